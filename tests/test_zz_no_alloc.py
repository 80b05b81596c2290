"""The entry point that allocates nothing (`tv_adj_bernoulli_ll_fwd_static`, include/
tracevae_hip.h): called through the C-ABI, its output must equal - bit for bit, the same kernels
run in the same order - that of `tv_adj_bernoulli_ll_fwd` with a caller workspace, on two calls
in a row, the second with a smaller input (so stale partial sums of the first call would show)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(device, K, B, M, D, seed):
    g = torch.Generator().manual_seed(seed)
    h = (torch.randn(K * B, M, D, generator=g) * 0.4).to(device)
    adj = (torch.rand(B, M * (M - 1) // 2, generator=g) < 0.05).float().to(device)
    return h, adj


def _both(device, K, B, M, D, seed, biased):
    from tracevae_amd import hip
    h, adj = _case(device, K, B, M, D, seed)
    args = (K, B, M, D, hip.ptr(h), hip.ptr(adj), int(biased), float(M), math.log(0.5))
    want = torch.empty(K * B, device=device)
    wsb = hip.query('tv_adj_bernoulli_ll_fwd_workspace_bytes', K, B, M)
    ws = torch.empty(wsb // 4, device=device)
    hip.call('tv_adj_bernoulli_ll_fwd', *args, hip.ptr(want), hip.ptr(ws), wsb)
    got = torch.full((K * B,), float('nan'), device=device)
    # torch's allocator sees nothing of a hipMalloc inside the library; what this guards is the
    # host side of the call (no tensor made on the way).  The count of allocation events only
    # grows, unlike memory_allocated(), which drops when Python collects an earlier temporary.
    before = torch.cuda.memory_stats()['allocation.all.allocated']
    hip.call('tv_adj_bernoulli_ll_fwd_static', *args, hip.ptr(got))
    assert torch.cuda.memory_stats()['allocation.all.allocated'] == before
    torch.cuda.synchronize()
    return got, want


def test_static_entry_point_equals_the_workspace_one(device):
    from tracevae_amd import hip
    # a large call first (fills 7 * 24 * 3 partial sums), then a smaller one
    got, want = _both(device, 7, 24, 100, 40, seed=1, biased=True)
    assert torch.isfinite(want).all() and torch.equal(got, want)
    got, want = _both(device, 2, 5, 33, 24, seed=2, biased=False)
    assert torch.isfinite(want).all() and torch.equal(got, want)
    # and a long-trace shape (136 tiles per trace)
    got, want = _both(device, 1, 3, 500, 80, seed=3, biased=True)
    assert torch.equal(got, want)
    # beyond the static maximum: a clear error, nothing launched
    lib = hip.lib()
    limit = 1 << 20
    K, B, M = 100, 4096, 100                       # 409600 traces x 3 tile groups > 2**20
    assert hip.query('tv_adj_bernoulli_ll_fwd_workspace_bytes', K, B, M) // 4 > limit
    one = 4096
    rc = lib.tv_adj_bernoulli_ll_fwd_static(K, B, M, 40, one, one, 0, 1.0, 0.0, one, None)
    assert rc == -3 and b'static scratch' in lib.tv_last_error()
