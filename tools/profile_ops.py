"""Which host-side line launches which small kernel: a torch.profiler pass over a few train steps
of the bench workload, printing the device time of short kernels grouped by the Python frame that
issued them.  Run on the GPU box:  python tools/profile_ops.py [--steps 3] [--max-us 15]"""
import argparse
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--max-us', type=float, default=15.0)
    args = ap.parse_args()
    import bench
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    from tracevae_amd.model import TraceVAE
    from tracevae_amd.synth import SyntheticSpec, generate_flat
    from tracevae_amd.train import Trainer
    V, M, B = 1001, 100, 512
    dev = torch.device('cuda', 0)
    cfg = bench.make_config()
    spec = SyntheticSpec(num_operations=V, max_spans=M, seed=0)
    flat, lr = generate_flat(spec, 4096, seed=1)
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
        trainer.step(TraceGraphBatch.from_store(store, ids, M), check_nan=False)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    events = prof.events()
    by_frame = collections.defaultdict(lambda: [0, 0.0])
    by_op = collections.defaultdict(lambda: [0, 0.0])
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for ev in events:
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        short = [k for k in ev.kernels if k.duration <= args.max_us]
        if not short:
            continue
        t = sum(k.duration for k in short)
        frame = next((f for f in (ev.stack or []) if here in f and 'profile_ops' not in f), None)
        if frame is None:
            frame = 'autograd/backward: ' + ev.name
        else:
            frame = frame.replace(here + '/', '')
        by_frame[frame][0] += len(short)
        by_frame[frame][1] += t
        by_op[ev.name][0] += len(short)
        by_op[ev.name][1] += t
    n = args.steps
    total = sum(v[1] for v in by_frame.values())
    print(f'short kernels (<= {args.max_us} us): {total / n:.0f} us/step, '
          f'{sum(v[0] for v in by_frame.values()) / n:.0f} launches/step')
    print('--- by issuing frame')
    for k, (c, t) in sorted(by_frame.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f'{t / n:8.1f} us {c / n:6.1f}  {k}')
    print('--- by op')
    for k, (c, t) in sorted(by_op.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f'{t / n:8.1f} us {c / n:6.1f}  {k}')


if __name__ == '__main__':
    main()
