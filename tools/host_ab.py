"""Step time at a small batch with one switch of tracevae_amd.model off and on, alternating in
one process (host-bound regime: what a change costs or saves on the host).
    python tools/host_ab.py PREPARED_STACK_WEIGHTS [batch] [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import tracevae_amd.model as model
from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
from tracevae_amd.model import TraceVAE
from tracevae_amd.synth import SyntheticSpec, generate_flat
from tracevae_amd.train import Trainer
flag = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
V, M = 1001, 100
cfg = bench.make_config()
flat, lr = generate_flat(SyntheticSpec(num_operations=V, max_spans=M, seed=1), 1024, seed=2)
store = DeviceTraceStore(flat, lr, V)
torch.manual_seed(0)
vae = TraceVAE(cfg.model, V, max_node_count=M).to('cuda:0')
for m in vae.modules():
    if hasattr(m, 'initialized'):
        m.initialized = True
trainer = Trainer(vae, cfg, world_size=1)
rng = np.random.RandomState(0)
def one():
    ids = rng.randint(0, len(store), size=B)
    return trainer.step(TraceGraphBatch.from_store(store, ids, M), check_nan=False)['loss']
for _ in range(5):
    one()
for rnd in range(3):
    for v in (True, False):
        setattr(model, flag, v)
        one(); one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        print(f'{flag}={int(v)} batch {B}: {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step', flush=True)
