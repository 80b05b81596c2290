"""Step time with the three NaN-decision modes of Trainer.step, alternating in one process:
    python tools/nan_ab.py [batch] [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
from tracevae_amd.model import TraceVAE
from tracevae_amd.synth import SyntheticSpec, generate_flat
from tracevae_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
V, M = 1001, 100
cfg = bench.make_config()
flat, lr = generate_flat(SyntheticSpec(num_operations=V, max_spans=M, seed=1), 8192, seed=2)
store = DeviceTraceStore(flat, lr, V)
torch.manual_seed(0)
vae = TraceVAE(cfg.model, V, max_node_count=M).to('cuda:0')
for m in vae.modules():
    if hasattr(m, 'initialized'):
        m.initialized = True
trainer = Trainer(vae, cfg, world_size=1)
rng = np.random.RandomState(0)
def one(mode):
    ids = rng.randint(0, len(store), size=B)
    return trainer.step(TraceGraphBatch.from_store(store, ids, M), check_nan=mode)['loss']
for _ in range(8):
    one(False)
for rnd in range(3):
    for mode in (True, False, 'now'):
        one(mode); one(mode)
        trainer.finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one(mode)
        trainer.finish()
        torch.cuda.synchronize()
        print(f'check_nan={mode!s:5} batch {B}: {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step', flush=True)
