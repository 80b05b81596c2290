"""Host-side profile of the training step at a small batch (where the host is the bound):
    python tools/profile_host.py [batch] [steps]   -> cProfile table of the timed steps only"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
from tracevae_amd.model import TraceVAE
from tracevae_amd.synth import SyntheticSpec, generate_flat
from tracevae_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
V, M = 1001, 100
cfg = bench.make_config()
flat, lr = generate_flat(SyntheticSpec(num_operations=V, max_spans=M, seed=1), 1024, seed=2)
store = DeviceTraceStore(flat, lr, V)
torch.manual_seed(0)
dev = torch.device('cuda', 0)
vae = TraceVAE(cfg.model, V, max_node_count=M).to(dev)
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
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one()
torch.cuda.synchronize()
print('ms/step (no profiler):', 1e3 * (time.perf_counter() - t0) / steps)
torch.autograd.set_multithreading_enabled(False)   # backward in this thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    one()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
