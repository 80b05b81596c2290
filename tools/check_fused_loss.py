"""Development check: one training job with the fused scalar loss against the same job with the
tensor-op composition (same seed): per-step loss and the relative difference of the flat gradient."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import tracevae_amd.train as T  # noqa: E402
from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch  # noqa: E402
from tracevae_amd.model import TraceVAE  # noqa: E402
from tracevae_amd.synth import SyntheticSpec, generate_flat  # noqa: E402


def run(fused, steps=8):
    T.FUSED_TRAIN_LOSS = fused
    V, M, B = 1001, 100, 512
    dev = torch.device('cuda', 0)
    cfg = bench.make_config()
    flat, lr = generate_flat(SyntheticSpec(num_operations=V, max_spans=M, seed=1234), 2048, seed=1235)
    store = DeviceTraceStore(flat, lr, V)
    torch.manual_seed(1234)
    vae = TraceVAE(cfg.model, V, max_node_count=M).to(dev)
    for m in vae.modules():
        if hasattr(m, 'initialized'):
            m.initialized = True
    tr = T.Trainer(vae, cfg, world_size=1)
    rng = np.random.RandomState(1234)
    out = []
    for _ in range(steps):
        ids = rng.randint(0, len(store), size=B)
        torch.manual_seed(7)
        res = tr.step(TraceGraphBatch.from_store(store, ids, M))
        out.append((float(res['loss']), tr.flat.grad.clone()))
    return out


a, b = run(True), run(False)
for i, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
    rel = float((ga - gb).norm() / gb.norm())
    print(f'step {i}: loss fused {la:.4f} composed {lb:.4f}  |dgrad|/|grad| {rel:.3e}')
