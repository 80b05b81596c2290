"""Which GEMM shapes a train step of the bench workload issues and what each costs: wraps ops.gemm
with events (one synchronisation per call, so only the per-call device times mean anything).
Run on the GPU box:  python tools/gemm_shapes.py [V M B width z_dim]   (default 1001 100 512: the bench
workload; 10001 500 32 1000 20: BASELINE configs[4])"""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from tracevae_amd import ops
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    from tracevae_amd.model import TraceVAE
    from tracevae_amd.synth import SyntheticSpec, generate_flat
    from tracevae_amd.train import Trainer
    argv = [int(x) for x in sys.argv[1:]]
    V, M, B = (argv + [1001, 100, 512][len(argv):])[:3]
    width = argv[3] if len(argv) > 3 else None
    z_dim = argv[4] if len(argv) > 4 else 10
    dev = torch.device('cuda', 0)
    cfg = bench.make_config(width, z_dim)
    spec = SyntheticSpec(num_operations=V, max_spans=M, seed=0)
    flat, lr = generate_flat(spec, max(4 * B, 256), seed=1)
    store = DeviceTraceStore(flat, lr, V)
    torch.manual_seed(0)
    vae = TraceVAE(cfg.model, V, max_node_count=M).to(dev)
    for m in vae.modules():
        if hasattr(m, 'initialized'):
            m.initialized = True
    trainer = Trainer(vae, cfg, world_size=1)
    rng = np.random.RandomState(0)

    def step():
        ids = rng.randint(0, len(store), size=B)
        return trainer.step(TraceGraphBatch.from_store(store, ids, M), check_nan=False)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    inner = ops.gemm
    table = collections.defaultdict(lambda: [0, 0.0])

    def timed(a, b, trans_a, trans_b, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = inner(a, b, trans_a, trans_b, **kw)
        e.record()
        torch.cuda.synchronize()
        m = a.shape[1] if trans_a else a.shape[0]
        k = a.shape[0] if trans_a else a.shape[1]
        n = b.shape[0] if trans_b else b.shape[1]
        act = kw.get('act')
        key = (m, n, k, 'T' if trans_a else 'N', 'T' if trans_b else 'N', act[0] if act else 0,
               a.stride(0), b.stride(0))
        table[key][0] += 1
        table[key][1] += s.elapsed_time(e)
        return out

    ops.gemm = timed
    steps = 3
    for _ in range(steps):
        step()
    ops.gemm = inner
    total = sum(v[1] for v in table.values())
    print(f'GEMM device time per step: {total / steps:.3f} ms')
    print('      m      n      k  ta tb act    lda    ldb  calls/step  ms/step  TF')
    for key, (calls, ms) in sorted(table.items(), key=lambda kv: -kv[1][1]):
        m, n, k, ta, tb, act, lda, ldb = key
        print(f'{m:7d} {n:6d} {k:6d}   {ta}  {tb}   {act} {lda:6d} {ldb:6d}  {calls / steps:9.1f}  {ms / steps:7.3f}  '
              f'{2.0 * m * n * k * calls / ms / 1e9:6.1f}')


if __name__ == '__main__':
    main()
