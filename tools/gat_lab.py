"""Times the message-passing kernels alone on the graphs of the bench: the decoder graph of a
scoring call (128 traces x K = 100 replicas, eval epilogue) and the encoder / decoder graphs of a
512-trace training batch (statistics + gradients).  `TRACEVAE_HIP_LIB=<variant .so>` selects the
build under test.   python tools/gat_lab.py [--k 100]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--k', type=int, default=100)
    args = ap.parse_args()
    from tracevae_amd import ops
    from tracevae_amd.batch import DeviceTraceStore, build_device_batch
    from tracevae_amd.synth import SyntheticSpec, generate_flat
    dev = torch.device('cuda', 0)
    V, M, H, F = 1001, 100, 2, 500
    flat, lr = generate_flat(SyntheticSpec(num_operations=V, max_spans=M, seed=1), 4096, seed=2)
    store = DeviceTraceStore(flat, lr, V)
    rng = np.random.RandomState(0)
    al = torch.randn(H, F, device=dev) * 0.05
    ar = torch.randn(H, F, device=dev) * 0.05
    bias = torch.zeros(H * F, device=dev)
    scale = torch.rand(H * F, device=dev) + 0.5
    shift = torch.randn(H * F, device=dev) * 0.1
    ld = H * F + 4

    def report(name, us, graph, n_rep, extra=0.0):
        rows = graph.num_rows * n_rep
        comp = 4.0 * (ld + H * F) * rows + extra
        print(f'{name:34s} rows {rows:8d} edges/row {graph.num_edges / graph.num_rows:4.2f} '
              f'{us:8.1f} us  {comp / us / 1e6:6.2f} TB/s (compulsory bytes)', flush=True)

    # scoring: decoder graph, K replicas, eval epilogue
    b = build_device_batch(store, rng.randint(0, len(store), size=128), M)
    g = b.dec
    ft = torch.randn(g.num_rows * args.k, ld, device=dev)
    with torch.no_grad():
        us = timed(lambda: ops.gat_conv(ft, al, ar, bias, g, n_rep=args.k, ep_scale=scale,
                                        ep_shift=shift), n=5)
    report(f'fwd eval, dec graph x{args.k}', us, g, args.k)
    # the same rows as a pure copy: a graph of self-loops only, and torch's strided copy
    R = g.num_rows
    ar_ = torch.arange(R + 1, device=dev, dtype=torch.int64)
    g1 = ops.DeviceGraph(ar_, ar_[:R].clone(), ar_[:R].clone(), ar_[:R].clone(), ar_[:R].clone(),
                         ar_[:R].clone())
    with torch.no_grad():
        us = timed(lambda: ops.gat_conv(ft, al, ar, bias, g1, n_rep=args.k, ep_scale=scale,
                                        ep_shift=shift), n=5)
    report(f'fwd eval, self-loops only x{args.k}', us, g1, args.k)
    dst = torch.empty(ft.shape[0], H * F, device=dev)
    us = timed(lambda: dst.copy_(ft[:, :H * F]), n=5)
    report('torch strided copy of the rows', us, g1, args.k)
    del dst
    del ft
    # training batch
    b = build_device_batch(store, rng.randint(0, len(store), size=512), M)
    for name, g in (('enc', b.enc), ('dec', b.dec)):
        ft = torch.randn(g.num_rows, ld, device=dev, requires_grad=True)
        out = [None]

        def fwd():
            out[0] = ops.gat_conv(ft, al, ar, bias, g, want_stats=True)[0]
        us = timed(fwd, n=20)
        report(f'fwd train, {name} graph', us, g, 1)
        dout = torch.randn_like(out[0])
        us = timed(lambda: torch.autograd.grad(out[0], ft, dout, retain_graph=True), n=20)
        report(f'bwd train, {name} graph', us, g, 1, extra=4.0 * ld * g.num_rows)


if __name__ == '__main__':
    main()
