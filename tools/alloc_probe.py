"""Device allocations (hipMalloc calls of torch's caching allocator) and step time, step by
step, from a cold start: shows how many steps a bench needs before its timed region is free of
allocator growth.    python tools/alloc_probe.py [batch] [steps]"""
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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
for i in range(steps):
    ids = rng.randint(0, len(store), size=B)
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_stats()['num_device_alloc']
    t0 = time.perf_counter()
    G = TraceGraphBatch.from_store(store, ids, M)
    trainer.step(G)
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    a1 = torch.cuda.memory_stats()['num_device_alloc']
    print(f'step {i:3d}: {dt:7.3f} ms  device allocations {a1 - a0:3d}  reserved '
          f'{torch.cuda.memory_reserved() / 2**30:.2f} GiB  rows {G.num_nodes if hasattr(G, "num_nodes") else ""}', flush=True)
trainer.finish()
