"""Host-bound step time (batch 64) under three settings of Python's cyclic collector:
default, gc.freeze() after set-up, gc.disable().    python tools/gc_ab.py [batch] [steps]"""
import gc, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
from tracevae_amd.model import TraceVAE
from tracevae_amd.synth import SyntheticSpec, generate_flat
from tracevae_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
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
    return trainer.step(TraceGraphBatch.from_store(store, ids, M))['loss']
for _ in range(8):
    one()
def run(tag):
    one(); one()
    trainer.finish(); torch.cuda.synchronize()
    c0 = [s['collections'] for s in gc.get_stats()]
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    trainer.finish(); torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0) / steps
    c1 = [s['collections'] for s in gc.get_stats()]
    print(f'{tag:10s} batch {B}: {dt:.3f} ms/step, collections per generation {[b - a for a, b in zip(c0, c1)]}', flush=True)
for rnd in range(2):
    gc.enable(); gc.unfreeze(); run('default')
    gc.collect(); gc.freeze(); run('freeze')
    gc.disable(); run('disabled')
# does a step leave cyclic garbage that holds device memory?  (collector still disabled here)
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated()
for _ in range(200):
    one()
trainer.finish(); torch.cuda.synchronize()
m1 = torch.cuda.memory_allocated()
n = gc.collect()
m2 = torch.cuda.memory_allocated()
print(f'200 steps without the collector: device memory {m0 / 2**20:.1f} -> {m1 / 2**20:.1f} MiB, '
      f'after gc.collect() ({n} objects) {m2 / 2**20:.1f} MiB', flush=True)
gc.enable()
for thr in (700, 5000, 50000):
    gc.set_threshold(thr, 10, 10)
    run(f'thr {thr}')
