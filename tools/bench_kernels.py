"""Per-kernel micro-benchmarks on the GPU box (development tool; not part of the product path).

    python tools/bench_kernels.py [gemm] [gat] [norm] [misc]

Times each HIP entry point at the shapes the C2 training step uses (B=512 traces, M=100:
51,200 decoder rows, ~27,000 encoder rows, width 1000) with HIP events over `reps` launches
and prints achieved TFLOP/s or GB/s (algorithmic work / time).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from tracevae_amd import hip, ops  # noqa: E402

DEV = torch.device('cuda:0')


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps          # ms


def bench_gemm():
    shapes = [
        # name, ta, tb, m, n, k
        ('fwd  x.W^T dec', False, True, 51200, 1000, 1000),
        ('dX   dy.W  dec', False, False, 51200, 1000, 1000),
        ('dW   dy^T.x dec', True, False, 1000, 1000, 51200),
        ('fwd  x.W^T enc', False, True, 27000, 1000, 1000),
        ('dW   dy^T.x enc', True, False, 1000, 1000, 27000),
        ('fwd  layer1 K=40', False, True, 51200, 1000, 40),
        ('fwd  layer1 K=41', False, True, 27000, 1000, 41),
        ('logits agg.W V=1001', False, False, 51200, 1001, 1000),
        ('dX logits V=1001', False, True, 51200, 1000, 1001),
        ('dW logits V=1001', True, False, 1000, 1001, 51200),
        ('mlp 512x500x500', False, True, 512, 500, 500),
        ('node-emb 512x4000x500', False, True, 512, 4000, 500),
        ('square 4096', False, True, 4096, 4096, 4096),
        ('fwd  x.W^T dec (again)', False, True, 51200, 1000, 1000),
        ('fwd  1024-aligned', False, True, 51200, 1024, 1024),
        # scoring: K * B * M = 10 * 128 * 100 decoder rows
        ('score x.W^T', False, True, 128000, 1000, 1000),
        ('score x.W', False, False, 128000, 1000, 1000),
        ('score logits V=1001', False, False, 128000, 1001, 1000),
        ('score x.W^T 64k rows', False, True, 64000, 1000, 1000),
        ('score x.W^T 256k rows', False, True, 256000, 1000, 1000),
    ]
    for name, ta, tb, m, n, k in shapes:
        a = torch.randn((k, m) if ta else (m, k), device=DEV)
        b = torch.randn((n, k) if tb else (k, n), device=DEV)
        c = torch.empty(m, n, device=DEV)
        ms = timeit(lambda: ops.gemm(a, b, ta, tb, out=c))
        print(f'gemm {name:24s} m={m:6d} n={n:5d} k={k:6d}  {ms:8.3f} ms  '
              f'{2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s', flush=True)


def _graph(batch=512, M=100, V=1001):
    from helpers import make_flat
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    flat, lr = make_flat(batch, V, M, seed=1)
    store = DeviceTraceStore(flat, lr, V)
    t0 = time.perf_counter()
    db = TraceGraphBatch.from_store(store, np.arange(batch), M).build_dgl()
    torch.cuda.synchronize()
    ms = timeit(lambda: TraceGraphBatch.from_store(store, np.arange(batch), M).build_dgl(), reps=10)
    n = db.num_nodes
    out_bytes = 8 * (5 * db.enc.num_edges + 5 * db.dec.num_edges + n + 1 + batch * M + 1) \
        + 4 * db.adj.numel() + 8 * batch * M + 4 * batch * M + 8 * batch + 8 * n + 12 * n + 8 * n
    print(f'batch_build B={batch} M={M}: N={n} E_enc={db.enc.num_edges} E_dec={db.dec.num_edges} '
          f'{ms:.3f} ms (host+kernel)  out {out_bytes / 1e6:.1f} MB -> {out_bytes / ms / 1e6:.1f} GB/s',
          flush=True)
    return db


def bench_gat():
    db = _graph()
    H, F = 2, 500
    for which in ('dec', 'enc'):
        g = getattr(db, which)
        rows, E = g.num_rows, g.num_edges
        ft = torch.randn(rows, H * F, device=DEV, requires_grad=True)
        al = torch.randn(1, H, F, device=DEV, requires_grad=True) * 0.1
        ar = torch.randn(1, H, F, device=DEV, requires_grad=True) * 0.1
        bias = torch.zeros(H * F, device=DEV, requires_grad=True)
        el = torch.empty(rows, H, device=DEV)
        er = torch.empty(rows, H, device=DEV)
        ms = timeit(lambda: hip.call('tv_gat_scores', rows, H, F, hip.ptr(ft), hip.ptr(al.detach().reshape(-1).contiguous()),
                                     hip.ptr(ar.detach().reshape(-1).contiguous()), hip.ptr(el), hip.ptr(er)))
        print(f'gat_scores {which}: rows={rows} {ms:.3f} ms  {4.0 * rows * H * F / ms / 1e6:.0f} GB/s', flush=True)
        fwd_bytes = 4.0 * H * F * (E + rows)
        with torch.no_grad():
            ms = timeit(lambda: ops.gat_conv(ft, al, ar, bias, g, 1))
        print(f'gat fwd (scores+aggregate, no stats) {which}: rows={rows} E={E} {ms:.3f} ms  '
              f'{fwd_bytes / ms / 1e6:.0f} GB/s (aggregate bytes only)', flush=True)
        out, stats = ops.gat_conv(ft, al, ar, bias, g, 1, want_stats=True)
        dout = torch.randn_like(out)
        ms = timeit(lambda: torch.autograd.grad(out, [ft, al, ar, bias], dout, retain_graph=True))
        bwd_bytes = 4.0 * H * F * (2 * E + 4 * rows)
        print(f'gat bwd {which}: {ms:.3f} ms  {bwd_bytes / ms / 1e6:.0f} GB/s', flush=True)


def bench_norm():
    rows, C = 51200, 1000
    x = torch.randn(rows, C, device=DEV, requires_grad=True)
    gamma = torch.ones(C, device=DEV, requires_grad=True)
    beta = torch.zeros(C, device=DEV, requires_grad=True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    with torch.no_grad():
        ms = timeit(lambda: ops.bn_lrelu(x, gamma, beta, rm, rv, True, 1e-6))
    print(f'bn_lrelu train fwd (stats+finalize+apply): {ms:.3f} ms  {12.0 * rows * C / ms / 1e6:.0f} GB/s', flush=True)
    out = ops.bn_lrelu(x, gamma, beta, rm, rv, True, 1e-6)
    dy = torch.randn_like(out)
    ms = timeit(lambda: torch.autograd.grad(out, [x, gamma, beta], dy, retain_graph=True))
    print(f'bn_lrelu bwd (reduce+finalize+apply): {ms:.3f} ms  {20.0 * rows * C / ms / 1e6:.0f} GB/s', flush=True)


def bench_misc():
    db = _graph()
    rows, V = db.dec.num_rows, 1001
    logits = torch.randn(rows, V, device=DEV, requires_grad=True)
    tgt = db.node_type_pad.reshape(-1)
    with torch.no_grad():
        ms = timeit(lambda: ops.categorical_ll(logits, tgt))
    print(f'categorical_ll fwd rows={rows} V={V}: {ms:.3f} ms  {4.0 * rows * V / ms / 1e6:.0f} GB/s', flush=True)
    ll = ops.categorical_ll(logits, tgt)
    g = torch.randn_like(ll)
    ms = timeit(lambda: torch.autograd.grad(ll, [logits], g, retain_graph=True))
    print(f'categorical_ll bwd: {ms:.3f} ms  {8.0 * rows * V / ms / 1e6:.0f} GB/s', flush=True)
    h = torch.randn(512, 100, 40, device=DEV, requires_grad=True)
    with torch.no_grad():
        ms = timeit(lambda: ops.adj_bernoulli_ll(h, db.adj, 1))
    print(f'adj_bernoulli_ll fwd B=512 M=100: {ms:.3f} ms', flush=True)
    ll = ops.adj_bernoulli_ll(h, db.adj, 1)
    g = torch.randn_like(ll)
    ms = timeit(lambda: torch.autograd.grad(ll, [h], g, retain_graph=True))
    print(f'adj_bernoulli_ll bwd: {ms:.3f} ms', flush=True)
    x = torch.randn(rows, 1000, device=DEV)
    with torch.no_grad():
        ms = timeit(lambda: ops.graph_sum(x, db.dec, 1))
    print(f'graph_sum dec rows={rows}: {ms:.3f} ms  {4000.0 * (db.dec.num_edges + rows) / ms / 1e6:.0f} GB/s', flush=True)
    w = torch.randn(1001, 40, device=DEV, requires_grad=True)
    e = ops.embedding(w, db.node_type)
    g = torch.randn_like(e)
    ms = timeit(lambda: torch.autograd.grad(e, [w], g, retain_graph=True))
    print(f'embedding bwd N={db.num_nodes} V=1001: {ms:.3f} ms', flush=True)


def bench_adj():
    """tv_adj_bernoulli_ll_* through the C-ABI: configs[1] training (512 traces, M = 100),
    configs[3] scoring (K = 100 x 128), configs[4] (8 and 32 traces of M = 500)."""
    import math
    for name, K, B, M, D, biased in (('train C2', 1, 512, 100, 40, 0), ('score K=100', 100, 128, 100, 40, 1),
                                     ('C5 B=8 D=40', 1, 8, 500, 40, 0), ('C5 B=8 D=80', 1, 8, 500, 80, 0),
                                     ('C5 B=32 D=80', 1, 32, 500, 80, 0), ('C5 score K=10', 10, 8, 500, 80, 1)):
        h = torch.randn(K * B, M, D, device=DEV) * 0.3
        adj = (torch.rand(B, M * (M - 1) // 2, device=DEV) < 0.02).float()
        ll = torch.empty(K * B, device=DEV)
        dll = torch.randn(K * B, device=DEV)
        dh = torch.empty_like(h)
        wsb = hip.query('tv_adj_bernoulli_ll_fwd_workspace_bytes', K, B, M)
        ws = torch.empty(max(wsb // 4, 1), device=DEV)
        wsb2 = hip.query('tv_adj_bernoulli_ll_bwd_workspace_bytes', K, B, M, D)
        ws2 = torch.empty(max(wsb2 // 4, 1), device=DEV)
        args = (K, B, M, D, hip.ptr(h), hip.ptr(adj), biased, float(M), math.log(0.5))
        f = timeit(lambda: hip.call('tv_adj_bernoulli_ll_fwd', *args, hip.ptr(ll), hip.ptr(ws), wsb), reps=50)
        b = timeit(lambda: hip.call('tv_adj_bernoulli_ll_bwd', *args, hip.ptr(dll), hip.ptr(dh),
                                    hip.ptr(ws2), wsb2), reps=50)
        pairs = K * B * M * (M - 1) / 2
        print(f'adj_ll {name:14s} K*B={K * B:6d} M={M} D={D}: fwd {1e3 * f:8.1f} us '
              f'({2 * D * pairs / f / 1e9:6.2f} TF), bwd {1e3 * b:8.1f} us '
              f'({6 * D * pairs / b / 1e9:6.2f} TF)', flush=True)


if __name__ == '__main__':
    what = sys.argv[1:] or ['gemm', 'gat', 'norm', 'misc']
    torch.cuda.set_device(0)
    for w in what:
        {'gemm': bench_gemm, 'gat': bench_gat, 'norm': bench_norm, 'misc': bench_misc,
         'adj': bench_adj}[w]()
