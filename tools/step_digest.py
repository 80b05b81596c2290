"""Digest of one seeded training step (loss, flat gradient, updated parameters) for bit-level
A/B between library builds: TRACEVAE_HIP_LIB=<variant .so> python tools/step_digest.py [batch]"""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
from tracevae_amd.model import TraceVAE
from tracevae_amd.synth import SyntheticSpec, generate_flat
from tracevae_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
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
g = torch.Generator().manual_seed(7)
for step in range(2):
    ids = rng.randint(0, len(store), size=B)
    eps = {'z': torch.randn(B, 10, generator=g), 'z2': torch.randn(B, 10, generator=g)}
    out = trainer.local_gradients(TraceGraphBatch.from_store(store, ids, M), eps)
    grad = trainer.flat.grad.clone()
    trainer.update()
    h = hashlib.sha256()
    for t in (out['loss'].reshape(1), grad, trainer.flat.flat):
        h.update(t.detach().cpu().numpy().tobytes())
    print(f'step {step}: loss {float(out["loss"]):.6f} digest {h.hexdigest()[:24]}', flush=True)
