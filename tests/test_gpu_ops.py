"""Operator-level parity: every HIP kernel, called through the C-ABI (`tracevae_amd.ops`),
against the oracle's restatement of the same operator evaluated in float64 on the CPU.
Tolerance: 1e-5 relative to the largest reference magnitude (fp32 kernels vs a float64
reference; the path-level bound of north_star is 1e-4)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import flow_log_prob_reference, hand_made_flat, make_flat, oracle_batch, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def _dev(t, device):
    return t.to(torch.float32).to(device).contiguous()


# --------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('m,n,k', [(1, 1, 1), (7, 5, 3), (64, 10, 1000), (130, 257, 41),
                                   (300, 1000, 1000), (1000, 1000, 50), (33, 1001, 1000),
                                   (513, 129, 36), (256, 256, 32)])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
def test_gemm(device, m, n, k, ta, tb):
    from tracevae_amd import ops
    a = _rand(*((k, m) if ta else (m, k)), seed=1)
    b = _rand(*((n, k) if tb else (k, n)), seed=2)
    bias = _rand(n, seed=3)
    ref = (a.t() if ta else a) @ (b.t() if tb else b) + bias
    got = ops.gemm(_dev(a, device), _dev(b, device), ta, tb, bias=_dev(bias, device))
    assert rel_err(got, ref) < TOL


@pytest.mark.parametrize('m,n,k,ta,tb,strided', [
    (40037, 1000, 1000, False, True, False),   # ~10 units per workgroup, partial last row tile, tail split
    (40000, 1001, 1000, False, False, False),  # N % 4 != 0: the direct epilogue on every unit
    (40000, 1000, 40, False, True, False),     # 2 k-tiles per unit: staged pieces flushed after the loop
    (33000, 1004, 1000, False, True, True),    # output consumed in place by a strided reader
    (1000, 1000, 40037, True, False, False),   # split-K slabs over two rounds of workgroups
])
def test_gemm_many_units_per_workgroup(device, m, n, k, ta, tb, strided):
    """The GEMM grid is persistent: a workgroup walks several output tiles, and a finished tile
    leaves through LDS in pieces stored during the next tile's k-loop.  Shapes with many rounds of
    tiles, a partial last row tile, a tail split, short k-loops and edge epilogues between staged
    ones, against the float64 product (computed on the GPU: torch's float64 GEMM)."""
    from tracevae_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(*((k, m) if ta else (m, k)), generator=g).to(device)
    b = torch.randn(*((n, k) if tb else (k, n)), generator=g).to(device)
    bias = torch.randn(n, generator=g).to(device)
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double()) + bias.double()
    if strided:
        wide = torch.full([m, n + 20], float('nan'), device=device)
        got = ops.gemm(a, b, ta, tb, bias=bias, out=wide[:, :n])
        assert torch.isnan(wide[:, n:]).all()             # nothing written beside the output
    else:
        got = ops.gemm(a, b, ta, tb, bias=bias)
    assert rel_err(got, ref) < TOL
    assert torch.equal(got, ops.gemm(a, b, ta, tb, bias=bias))      # bit-stable


@pytest.mark.parametrize('m,n,k,act,strided', [
    (1000, 1000, 5017, False, False),   # partial last k-tile in the last split, edge tiles both ways
    (1000, 1000, 5017, True, False),    # ... with the BatchNorm + LeakyReLU prologue on B
    (1004, 500, 3000, True, True),      # fc product with score columns; operands are column views
    (128, 260, 777, False, False),      # second column tile holds 4 columns
    (300, 132, 257, True, False),       # a split of one full k-tile and one row
    (1000, 1000, 130, False, False),    # k too short for two splits of two k-tiles: the other kernel
    (64, 1000, 4096, True, False),      # one row tile
    (1000, 2304, 200, False, False),    # 72 tiles x 3 splits of 96, 96 and 8 rows: a unit with no full k-tile
    (1000, 1000, 51200, True, False),   # the weight gradient of the bench's layers
])
def test_gemm_kmajor_operands_wide_form(device, m, n, k, act, strided):
    """Weight gradients C = A^T B with both operands k-major go through gemm_kmajor_wide_h_kernel
    (planes k-major in LDS, transposing fragment reads, packed split-K slabs) when M, N are
    multiples of 4 and the tiles fit one round of the chip; against the float64 product."""
    from tracevae_amd import ops
    gen = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    if strided:
        a = torch.randn(k, m + 8, generator=gen).to(device)[:, 4:m + 4]
        b = torch.randn(k, n + 12, generator=gen).to(device)[:, 8:n + 8]
    else:
        a = torch.randn(k, m, generator=gen).to(device)
        b = torch.randn(k, n, generator=gen).to(device)
    bias = torch.randn(n, generator=gen).to(device)
    Bm = b.double()
    kw = {}
    if act:
        scale = (0.5 + torch.rand(n, generator=gen)).to(device)
        shift = (0.3 * torch.randn(n, generator=gen)).to(device)
        kw['act'] = (2, scale, shift, 0.01)
        Bm = F.leaky_relu(Bm * scale.double() + shift.double(), 0.01)
    want = a.double().t() @ Bm + bias.double()
    wide = torch.full([m, n + 4], float('nan'), device=device)
    got = ops.gemm(a, b, True, False, bias=bias, out=wide[:, :n], **kw)
    assert torch.isnan(wide[:, n:]).all()
    assert rel_err(got, want) < 2e-5
    assert torch.equal(got, ops.gemm(a, b, True, False, bias=bias, **kw))       # bit-stable


@pytest.mark.parametrize('m,n,k', [(16424, 1000, 1001), (16384, 1000, 1017), (33000, 500, 1003)])
def test_gemm_wide_form_with_padded_rows_of_odd_length(device, m, n, k):
    """A row-major A whose row LENGTH is not a multiple of 4 but whose row STRIDE is (the gradient
    of the 1001-class logits, padded by the categorical backward) still takes the wide form with
    16-byte loads: the float4 that straddles K is read, its elements beyond K are zeroed in
    registers.  The pad columns hold NaN here, so anything that leaked would show."""
    from tracevae_amd import ops
    gen = torch.Generator().manual_seed(m + n + k)
    kp = (k + 3) // 4 * 4
    buf = torch.full([m, kp], float('nan'), device=device)
    a = buf[:, :k]
    a.copy_(torch.randn(m, k, generator=gen).to(device))
    b = torch.randn(n, k, generator=gen).to(device)
    want = a.double() @ b.double().t()
    got = ops.gemm(a, b, False, True)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < TOL
    assert torch.equal(got, ops.gemm(a, b, False, True))                # bit-stable
    assert rel_err(ops.gemm(a.contiguous(), b, False, True), want) < TOL   # the 4-byte-load path
    # a view with the same strides whose storage ENDS with the last element must not be read past
    flat = torch.empty((m - 1) * kp + k, device=device)
    tight = torch.as_strided(flat, (m, k), (kp, 1))
    tight.copy_(a)
    assert rel_err(ops.gemm(tight, b, False, True), want) < TOL


def test_categorical_backward_pads_rows_with_zeros(device):
    from tracevae_amd import hip
    rows, classes, stride = 37, 1001, 1004
    gen = torch.Generator().manual_seed(5)
    logits = torch.randn(rows, classes, generator=gen).to(device)
    target = torch.randint(0, classes, (rows,), generator=gen).to(device)
    dll = torch.randn(rows, generator=gen).to(device)
    lse = torch.logsumexp(logits, dim=1)
    out = torch.full([rows, stride], float('nan'), device=device)
    hip.call('tv_categorical_ll_bwd', rows, classes, rows, hip.ptr(logits), hip.ptr(target), 0, 1.0, 0.0,
             hip.ptr(lse), hip.ptr(dll), hip.ptr(out), stride)
    want = dll[:, None].double() * (F.one_hot(target, classes).double() - torch.softmax(logits.double(), dim=1))
    assert rel_err(out[:, :classes], want) < 1e-5
    assert torch.equal(out[:, classes:], torch.zeros(rows, stride - classes, device=device))
    with pytest.raises(hip.HipLibraryError):
        hip.call('tv_categorical_ll_bwd', rows, classes, rows, hip.ptr(logits), hip.ptr(target), 0, 1.0, 0.0,
                 hip.ptr(lse), hip.ptr(dll), hip.ptr(out), classes - 1)


@pytest.mark.parametrize('n,offset', [(1000003, 0), (1000003, 1), (4096, 0), (5, 0), (70001, 4)])
def test_masked_sum_of_squares(device, n, offset):
    """tv_sumsq_partial + tv_sum_partials (the L2 term and the gradient norm of the train step):
    16-byte loads when the pointers allow it, the scalar form otherwise, against float64."""
    from tracevae_amd import hip
    gen = torch.Generator().manual_seed(n + offset)
    x = torch.randn(n + offset, generator=gen).to(device)[offset:]
    mask = (torch.rand(n + offset, generator=gen) < 0.7).to(torch.uint8).to(device)[offset:]
    nblk = hip.query('tv_reduce_blocks', n)
    partial = torch.empty(nblk, dtype=torch.float64, device=device)
    out = torch.empty(1, device=device)
    for m, want in ((mask, (x.double() ** 2 * mask.double()).sum()), (None, (x.double() ** 2).sum())):
        hip.call('tv_sumsq_partial', n, x.data_ptr(), 0 if m is None else m.data_ptr(), partial.data_ptr())
        hip.call('tv_sum_partials', nblk, partial.data_ptr(), 0, out.data_ptr())
        assert abs(float(out[0]) - float(want)) <= 2e-6 * float(want) + 1e-12
    # y += alpha * x where the mask is on (the L2 term's gradient), same two forms
    y0 = torch.randn(n + offset, generator=gen).to(device)[offset:]
    for m in (mask, None):
        y = y0.clone()
        hip.call('tv_axpy_masked', n, 0.25, x.data_ptr(), 0 if m is None else m.data_ptr(), y.data_ptr())
        want = torch.where(mask.bool(), torch.addcmul(y0, x, torch.tensor(0.25, device=device)), y0) \
            if m is not None else y0 + 0.25 * x
        assert torch.allclose(y, want, rtol=1e-6, atol=1e-6)


def test_gemm_kmajor_wide_form_random_shapes(device):
    """32 seeded random weight-gradient shapes inside the wide k-major form's conditions (M, N
    multiples of 4, N > 128, few enough tiles for a split over K), with and without the prologue,
    against float64: unit boundaries, partial k-tiles and edge tiles fall everywhere."""
    from tracevae_amd import ops
    rng = np.random.RandomState(20261111)
    for case in range(32):
        m = 4 * int(rng.randint(1, 380))
        n = 4 * int(rng.randint(33, 380))
        k = int(rng.randint(130, 9000))
        gen = torch.Generator().manual_seed(case)
        a = torch.randn(k, m, generator=gen).to(device)
        b = torch.randn(k, n, generator=gen).to(device)
        Bm, kw = b.double(), {}
        if case % 2:
            scale = (0.5 + torch.rand(n, generator=gen)).to(device)
            shift = (0.3 * torch.randn(n, generator=gen)).to(device)
            kw['act'] = (2, scale, shift, 0.01)
            Bm = F.leaky_relu(Bm * scale.double() + shift.double(), 0.01)
        want = a.double().t() @ Bm
        got = ops.gemm(a, b, True, False, **kw)
        assert rel_err(got, want) < 2e-5, (m, n, k, case)


def test_gemm_wide_form_random_shapes(device):
    """12 seeded random layer-product shapes that take the wide form (many rows of a row-major A
    against a small B in either storage order), K any length under a row stride padded to 4
    floats, with bias, alternately with the prologue on A when K allows it; against float64."""
    from tracevae_amd import ops
    rng = np.random.RandomState(7)
    for case in range(12):
        n = int(rng.randint(600, 1100))
        m = 128 * int(rng.randint(256 * 256 // ((n + 255) // 256 * 256) + 1, 200)) + int(rng.randint(0, 128))
        k = int(rng.randint(40, 1100))
        tb = bool(case & 1)
        gen = torch.Generator().manual_seed(100 + case)
        kp = (k + 3) // 4 * 4
        buf = torch.full([m, kp], float('nan'), device=device)
        a = buf[:, :k]
        a.copy_(torch.randn(m, k, generator=gen).to(device))
        b = torch.randn(*((n, k) if tb else (k, n)), generator=gen).to(device)
        bias = torch.randn(n, generator=gen).to(device)
        A, kw = a.double(), {}
        if case % 3 == 0 and k % 4 == 0:
            scale = (0.5 + torch.rand(k, generator=gen)).to(device)
            shift = (0.3 * torch.randn(k, generator=gen)).to(device)
            kw['act'] = (1, scale, shift, 0.01)
            A = F.leaky_relu(A * scale.double() + shift.double(), 0.01)
        want = A @ (b.double().t() if tb else b.double()) + bias.double()
        got = ops.gemm(a, b, False, tb, bias=bias, **kw)
        assert torch.isfinite(got).all(), (m, n, k, case)
        assert rel_err(got, want) < 2e-5, (m, n, k, case)


def test_gemm_split_k_and_accumulate(device):
    from tracevae_amd import ops
    # weight-gradient shape: small output, long reduction -> split-K path
    a, b = _rand(20000, 96, seed=4), _rand(20000, 200, seed=5)
    ref = a.t() @ b
    got = ops.gemm(_dev(a, device), _dev(b, device), True, False)
    assert rel_err(got, ref) < TOL
    again = ops.gemm(_dev(a, device), _dev(b, device), True, False)
    assert torch.equal(got, again)                       # fixed-order reduction: bit-stable
    c0 = _rand(96, 200, seed=6)
    c = _dev(c0, device)
    scale = _rand(96, seed=7)
    ops.gemm(_dev(a, device), _dev(b, device), True, False, row_scale=_dev(scale, device), out=c,
             accumulate=True)
    assert rel_err(c, c0 + scale[:, None] * ref) < TOL


def _gemm_error_inputs(kind, m, n, k, gen):
    """Operand pairs for the accuracy bound of the split-bf16 GEMM, A [m, k] and B [n, k]."""
    a = torch.randn(m, k, generator=gen, dtype=torch.float64)
    b = torch.randn(n, k, generator=gen, dtype=torch.float64)
    if kind == 'wide':                      # 26 binary orders of magnitude inside one dot product
        a = a * torch.exp2(torch.randint(-13, 14, (m, k), generator=gen).double())
        b = b * torch.exp2(torch.randint(-13, 14, (n, k), generator=gen).double())
    elif kind == 'cancelling':              # every dot product is a small difference of large sums
        half = k // 2
        a[:, half:2 * half] = a[:, :half]
        b[:, half:2 * half] = -b[:, :half] * (1 + 1e-3 * torch.randn(n, half, generator=gen, dtype=torch.float64))
    elif kind == 'positive':                # no cancellation: the accumulators only grow
        a, b = a.abs(), b.abs()
    elif kind == 'graded':                  # ill-conditioned operands: geometrically graded columns
        grade = torch.logspace(0, -7, k, dtype=torch.float64)
        a = a * grade
        b = b / grade.flip(0) * 1e-7
    elif kind == 'activations':             # what the layers feed it: post-LeakyReLU x small weights
        a = torch.nn.functional.leaky_relu(a + 0.5, 0.01)
        b = b * 0.03
    return a.float(), b.float()


@pytest.mark.parametrize('kind', ['normal', 'wide', 'cancelling', 'positive', 'graded', 'activations'])
@pytest.mark.parametrize('m,n,k,ta,tb', [(512, 1000, 1000, False, True),      # layer forward
                                          (512, 1000, 1000, False, False),     # input gradient
                                          (384, 500, 20000, True, False)])     # weight gradient
def test_gemm_error_against_float64_is_within_twice_an_fp32_gemm(device, kind, m, n, k, ta, tb):
    """The GEMM multiplies on the bf16 matrix cores after splitting every fp32 operand into three
    bf16 pieces (six partial products, fp32 accumulation: csrc/gemm_f32.hip).  Its error against
    the float64 product of the same fp32 operands must not exceed twice that of an fp32 GEMM
    (torch's, rocBLAS / hipBLASLt on fp32 matrix instructions), element-wise scaled by
    sum_k |a||b| (the quantity a rounding-error bound is proportional to), and in the worst
    result element relative to the largest result - on layer shapes, for well- and
    ill-conditioned operands.  A small absolute floor (2^-24, half an fp32 ulp of the scale)
    covers the cases where the fp32 GEMM happens to be exact."""
    from tracevae_amd import ops
    gen = torch.Generator().manual_seed(m + n + k + len(kind))
    a, b = _gemm_error_inputs(kind, m, n, k, gen)           # A [m, k], B [n, k]
    A, B = a.to(device), b.to(device)
    exact = A.double() @ B.double().t()
    scale = A.double().abs() @ B.double().abs().t()
    a_arg = A.t().contiguous() if ta else A
    b_arg = B if tb else B.t().contiguous()
    got = ops.gemm(a_arg, b_arg, ta, tb).double()
    ref = (A @ B.t()).double()
    err_got = ((got - exact).abs() / scale).max().item()
    err_ref = ((ref - exact).abs() / scale).max().item()
    assert err_got <= 2 * err_ref + 2.0 ** -24, (kind, err_got, err_ref)
    top = exact.abs().max().item()
    worst_got = (got - exact).abs().max().item() / top
    worst_ref = (ref - exact).abs().max().item() / top
    assert worst_got <= 2 * worst_ref + 2.0 ** -24, (kind, worst_got, worst_ref)
    print(f'{kind:12s} {m}x{n}x{k} ta={ta} tb={tb}: scaled {err_got:.2e} (fp32 GEMM {err_ref:.2e}), '
          f'worst/top {worst_got:.2e} ({worst_ref:.2e})')


def test_gemm_asymmetric_identity(device):
    """A = I with an asymmetric B catches a transposed accumulator layout."""
    from tracevae_amd import ops
    n = 96
    b = torch.arange(n * 70, dtype=torch.float64).reshape(n, 70) % 17 - 3.0
    got = ops.gemm(_dev(torch.eye(n, dtype=torch.float64), device), _dev(b, device), False, False)
    assert torch.equal(got.cpu().double(), b)


def test_linear_autograd(device):
    from tracevae_amd import ops
    x, w, b = _rand(37, 5, 41, seed=1), _rand(29, 41, seed=2), _rand(29, seed=3)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.linear(xr, wr, br)
    g = _rand(*ref.shape, seed=4)
    ref.backward(g)
    xd, wd, bd = (_dev(t, device).requires_grad_(True) for t in (x, w, b))
    got = ops.linear(xd, wd, bd)
    got.backward(_dev(g, device))
    assert rel_err(got, ref) < TOL
    assert rel_err(xd.grad, xr.grad) < TOL and rel_err(wd.grad, wr.grad) < TOL
    assert rel_err(bd.grad, br.grad) < TOL


@pytest.mark.parametrize('lead,k,n,bias', [((512,), 1000, 10, True), ((3, 40), 250, 10, True),
                                          ((700,), 64, 16, False), ((5,), 77, 1, True),
                                          ((512,), 10, 500, True), ((3, 40), 10, 1000, False),
                                          ((33,), 16, 250, True), ((9,), 1, 64, True)])
def test_skinny_linear_autograd(device, lead, k, n, bias):
    """Linear layers with one small side (<= 16 inputs or outputs) take the skinny kernels
    instead of the MFMA GEMM."""
    from tracevae_amd import ops
    x, w = _rand(*lead, k, seed=1), _rand(n, k, seed=2)
    b = _rand(n, seed=3) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    ref = F.linear(xr, wr, br)
    g = _rand(*ref.shape, seed=4)
    ref.backward(g)
    xd, wd = _dev(x, device).requires_grad_(True), _dev(w, device).requires_grad_(True)
    bd = _dev(b, device).requires_grad_(True) if bias else None
    got = ops.linear(xd, wd, bd)
    assert got.grad_fn.__class__.__name__.startswith('_SkinnyLinear')
    got.backward(_dev(g, device))
    assert rel_err(got, ref) < TOL
    assert rel_err(xd.grad, xr.grad) < TOL and rel_err(wd.grad, wr.grad) < TOL
    if bias:
        assert rel_err(bd.grad, br.grad) < TOL
    # frozen input: no dx
    xd2 = _dev(x, device)
    wd2 = _dev(w, device).requires_grad_(True)
    ops.linear(xd2, wd2, None).backward(_dev(g, device))
    assert rel_err(wd2.grad, wr.grad) < TOL


def test_matmul_nn_and_rowdot_autograd(device):
    from tracevae_amd import ops
    x, w, b = _rand(301, 64, seed=1), _rand(64, 131, seed=2), _rand(131, seed=3)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = xr @ wr + br
    g = _rand(*ref.shape, seed=4)
    ref.backward(g)
    xd, wd, bd = (_dev(t, device).requires_grad_(True) for t in (x, w, b))
    got = ops.matmul_nn(xd, wd, bd)
    got.backward(_dev(g, device))
    for a_, b_ in ((got, ref), (xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        assert rel_err(a_, b_) < TOL
    # skinny head
    x, w = _rand(515, 1000, seed=5), _rand(1000, 2, seed=6)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = xr @ wr
    g = _rand(515, 2, seed=7)
    ref.backward(g)
    xd, wd = _dev(x, device).requires_grad_(True), _dev(w, device).requires_grad_(True)
    got = ops.rowdot(xd, wd)
    got.backward(_dev(g, device))
    for a_, b_ in ((got, ref), (xd.grad, xr.grad), (wd.grad, wr.grad)):
        assert rel_err(a_, b_) < TOL


# ---------------------------------------------------------------------------------- GAT
def _graph(device, num_traces=12, V=40, M=16, seed=7, shapes=None):
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    if shapes is None:
        flat, lr = make_flat(num_traces, V, M, seed=seed)
    else:
        flat, lr = hand_made_flat(shapes, V, seed=seed)
    store = DeviceTraceStore(flat, lr, V)
    db = TraceGraphBatch.from_store(store, np.arange(len(flat)), M).build_dgl()
    ob = oracle_batch(flat, lr, V, M)
    return db, ob


@pytest.mark.parametrize('heads,feats', [(2, 500), (2, 8), (1, 64), (2, 1000)])
@pytest.mark.parametrize('which', ['enc', 'dec'])
def test_gat_conv_forward_backward(device, heads, feats, which):
    from oracle.model import gat_conv
    from tracevae_amd import ops
    db, ob = _graph(device)
    graph = getattr(db, which)
    src, dst = torch.as_tensor(ob[which + '_src']), torch.as_tensor(ob[which + '_dst'])
    rows = graph.num_rows
    ft = _rand(rows, heads * feats, seed=1)
    al, ar = _rand(1, heads, feats, seed=2, scale=0.3), _rand(1, heads, feats, seed=3, scale=0.3)
    bias = _rand(heads * feats, seed=4)
    g = _rand(rows, heads * feats, seed=5)

    ftr, alr, arr, br = (t.clone().requires_grad_(True) for t in (ft, al, ar, bias))
    # the oracle takes h and a weight; feed ft through an identity weight
    eye = torch.eye(heads * feats, dtype=torch.float64)
    ref = gat_conv(ftr, src, dst, eye, alr, arr, br, heads).reshape(rows, -1)
    ref.backward(g)

    ftd, ald, ard, bd = (_dev(t, device).requires_grad_(True) for t in (ft, al, ar, bias))
    got, stats = ops.gat_conv(ftd, ald, ard, bd, graph, 1, want_stats=True)
    got.backward(_dev(g, device))
    assert rel_err(got, ref) < TOL
    assert rel_err(ftd.grad, ftr.grad) < TOL
    assert rel_err(ald.grad, alr.grad) < TOL and rel_err(ard.grad, arr.grad) < TOL
    assert rel_err(bd.grad, br.grad) < TOL
    # fused column statistics of the output
    s = stats.double().sum(0).cpu()
    assert rel_err(s[0], ref.detach().sum(0)) < TOL
    assert rel_err(s[1], (ref.detach() ** 2).sum(0)) < TOL
    # determinism: bit-equal on a second run
    got2, _ = ops.gat_conv(ftd.detach(), ald.detach(), ard.detach(), bd.detach(), graph, 1)
    assert torch.equal(got2, got.detach())


@pytest.mark.parametrize('heads,feats,k', [(2, 500, 1000), (2, 500, 41), (1, 64, 7), (4, 8, 50),
                                           (3, 12, 33), (2, 1000, 120)])
def test_gat_layer_with_scores_as_fc_columns(device, heads, feats, k):
    """fc and the attention scores in ONE product ([ft | el | er] = x [W ; A_l W ; A_r W]^T,
    `ops.linear_scores`) read in place by the aggregation, against the oracle's GATConv on h:
    output and the gradients of x, W, attn_l, attn_r, bias; replicas + eval epilogue too."""
    from oracle.model import gat_conv
    from tracevae_amd import ops
    db, ob = _graph(device)
    graph = db.dec
    src, dst = torch.as_tensor(ob['dec_src']), torch.as_tensor(ob['dec_dst'])
    rows, hf = graph.num_rows, heads * feats
    x = _rand(rows, k, seed=11)
    w = _rand(hf, k, seed=12, scale=1.0 / k ** 0.5)
    al, ar = _rand(1, heads, feats, seed=2, scale=0.3), _rand(1, heads, feats, seed=3, scale=0.3)
    bias, g = _rand(hf, seed=4), _rand(rows, hf, seed=5)
    xr, wr, alr, arr, br = (t.clone().double().requires_grad_(True) for t in (x, w, al, ar, bias))
    ref = gat_conv(xr, src, dst, wr, alr, arr, br, heads).reshape(rows, -1)
    ref.backward(g.double())

    xd, wd, ald, ard, bd = (_dev(t, device).requires_grad_(True) for t in (x, w, al, ar, bias))
    wx = ops.gat_fc_score_weight(wd.detach(), ald.detach(), ard.detach())
    width = hf + ops.score_columns(heads)
    assert wx.shape == (width, k) and torch.equal(wx[:hf], wd.detach())
    assert not wx[hf + 2 * heads:].any()
    a2 = al.reshape(heads, feats).double()
    assert rel_err(wx[hf:hf + heads], torch.einsum('hf,hfk->hk', a2, w.double().reshape(heads, feats, k))) < TOL
    fte = ops.linear_scores(xd, wd, ald, ard)
    assert fte.shape == (rows, width)
    got, _ = ops.gat_conv(fte, ald, ard, bd, graph, 1, want_stats=True)
    got.backward(_dev(g, device))
    assert rel_err(got, ref) < TOL
    for a_, b_ in ((xd.grad, xr.grad), (wd.grad, wr.grad), (ald.grad, alr.grad), (ard.grad, arr.grad),
                   (bd.grad, br.grad)):
        assert rel_err(a_, b_) < TOL
    # K replicas, no_grad, fused eval epilogue
    K = 3
    xk = _rand(K * rows, k, seed=21)
    scale, shift = _rand(hf, seed=5), _rand(hf, seed=6)
    refk = torch.cat([gat_conv(xk[i * rows:(i + 1) * rows].double(), src, dst, w.double(), al.double(),
                               ar.double(), bias.double(), heads).reshape(rows, -1) for i in range(K)])
    refk = F.leaky_relu(refk * scale + shift, 0.01)
    with torch.no_grad():
        ftk = ops.linear_scores(_dev(xk, device), wd, ald, ard)
        gotk, _ = ops.gat_conv(ftk, ald, ard, bd, graph, K, ep_scale=_dev(scale, device),
                               ep_shift=_dev(shift, device), ep_slope=0.01)
    assert rel_err(gotk, refk) < TOL


def test_gat_high_degree(device):
    """A star with more than 64 leaves exercises the multi-chunk edge loops."""
    from oracle.model import gat_conv
    from tracevae_amd import ops
    n = 150
    db, ob = _graph(device, V=30, M=n, shapes=[[0] * (n - 1), [0, 0, 1]])
    heads, feats = 2, 16
    for which in ('enc', 'dec'):
        graph = getattr(db, which)
        src, dst = torch.as_tensor(ob[which + '_src']), torch.as_tensor(ob[which + '_dst'])
        rows = graph.num_rows
        ft = _rand(rows, heads * feats, seed=1)
        al, ar = _rand(1, heads, feats, seed=2), _rand(1, heads, feats, seed=3)
        bias = _rand(heads * feats, seed=4)
        g = _rand(rows, heads * feats, seed=5)
        ftr, alr, arr, br = (t.clone().requires_grad_(True) for t in (ft, al, ar, bias))
        ref = gat_conv(ftr, src, dst, torch.eye(heads * feats, dtype=torch.float64), alr, arr, br,
                       heads).reshape(rows, -1)
        ref.backward(g)
        ftd, ald, ard, bd = (_dev(t, device).requires_grad_(True) for t in (ft, al, ar, bias))
        got, _ = ops.gat_conv(ftd, ald, ard, bd, graph, 1)
        got.backward(_dev(g, device))
        assert rel_err(got, ref) < TOL
        assert rel_err(ftd.grad, ftr.grad) < TOL
        assert rel_err(ald.grad, alr.grad) < TOL and rel_err(ard.grad, arr.grad) < TOL
        # the same star through the layer form (scores and their gradients as columns): the fused
        # backward pass keeps a row's per-edge dot products in memory when it has more than 64
        k = 24
        x, w = _rand(rows, k, seed=11), _rand(heads * feats, k, seed=12)
        xr, wr, alr, arr, br = (t.clone().requires_grad_(True) for t in (x, w, al, ar, bias))
        ref = gat_conv(xr, src, dst, wr, alr, arr, br, heads).reshape(rows, -1)
        ref.backward(g)
        xd, wd, ald, ard, bd = (_dev(t, device).requires_grad_(True) for t in (x, w, al, ar, bias))
        got, _ = ops.gat_conv(ops.linear_scores(xd, wd, ald, ard), ald, ard, bd, graph, 1)
        got.backward(_dev(g, device))
        assert rel_err(got, ref) < TOL
        for a_, b_ in ((xd.grad, xr.grad), (wd.grad, wr.grad), (ald.grad, alr.grad), (ard.grad, arr.grad),
                       (bd.grad, br.grad)):
            assert rel_err(a_, b_) < TOL


@pytest.mark.parametrize('K', [3, 9])        # (from 8 replicas on the output rows are streamed)
def test_gat_replicas_and_eval_epilogue(device, K):
    """K replicas of the decoder graph in one launch + fused eval BatchNorm/LeakyReLU."""
    from oracle.model import gat_conv
    from tracevae_amd import ops
    db, ob = _graph(device)
    heads, feats = 2, 12
    graph = db.dec
    src, dst = torch.as_tensor(ob['dec_src']), torch.as_tensor(ob['dec_dst'])
    rows = graph.num_rows
    ft = _rand(K * rows, heads * feats, seed=1)
    al, ar = _rand(1, heads, feats, seed=2), _rand(1, heads, feats, seed=3)
    bias, scale, shift = _rand(heads * feats, seed=4), _rand(heads * feats, seed=5), _rand(heads * feats, seed=6)
    eye = torch.eye(heads * feats, dtype=torch.float64)
    ref = torch.cat([gat_conv(ft[k * rows:(k + 1) * rows], src, dst, eye, al, ar, bias, heads)
                     .reshape(rows, -1) for k in range(K)])
    ref = F.leaky_relu(ref * scale + shift, 0.01)
    with torch.no_grad():
        got, _ = ops.gat_conv(_dev(ft, device), _dev(al, device), _dev(ar, device), _dev(bias, device),
                              graph, K, ep_scale=_dev(scale, device), ep_shift=_dev(shift, device),
                              ep_slope=0.01)
    assert rel_err(got, ref) < TOL


@pytest.mark.parametrize('batch,n_lat,use_l2,beta', [(700, 2, True, 1.0), (9, 1, False, 0.25),
                                                     (1, 0, True, 1.0), (512, 2, True, 0.5)])
def test_train_loss_matches_composition(device, batch, n_lat, use_l2, beta):
    """tv_train_loss_fwd/bwd against train.py:229-270 composed from tensor ops (float64)."""
    from tracevae_amd import ops
    names = ['node_count', 'adj', 'node_type', 'latency', 'qz', 'qz2', 'pz', 'pz2']
    vec = {n: _rand(batch, seed=10 + i) * 50 for i, n in enumerate(names)}
    obs, ql, pl = names[:4], names[4:4 + max(n_lat, 1)], names[6:6 + max(n_lat, 1)]
    zs = [_rand(batch, d, seed=30 + i) for i, d in enumerate((10, 20)[:n_lat])]
    regs = [1.0, 0.3][:n_lat]
    l2, l2_reg = _rand(1, seed=40).abs() * 100, 1e-4

    def run(dtype, dev, fused):
        v = {n: t.detach().clone().to(dtype).to(dev).requires_grad_(True) for n, t in vec.items()}
        z = [t.detach().clone().to(dtype).to(dev).requires_grad_(True) for t in zs]
        l2d = l2.to(dtype).to(dev)
        if fused:
            loss, recons, kl = ops.train_loss([v[n] for n in obs], [v[n] for n in ql],
                                              [v[n] for n in pl], beta, l2d if use_l2 else None,
                                              l2_reg if use_l2 else 0.0, list(zip(z, regs)))
        else:
            recons = sum(v[n] for n in obs).mean()
            kl = (sum(v[n] for n in ql) - sum(v[n] for n in pl)).mean()
            loss = beta * kl - recons
            if use_l2:
                loss = loss + l2_reg * l2d[0]
            for t, r in zip(z, regs):
                m = t.mean(-1)
                var = (t * t).mean(-1) - m * m
                loss = loss + r * ((m * m).mean() + ((var - 1) ** 2).mean())
        (loss * 3.0).backward()
        used = obs + ql + pl
        return loss, recons, kl, [v[n].grad for n in used], [t.grad for t in z]

    ref = run(torch.float64, 'cpu', False)
    got = run(torch.float32, device, True)
    for a, b in zip(got[:3], ref[:3]):
        assert rel_err(a.detach(), b.detach()) < TOL
    for a, b in zip(got[3] + got[4], ref[3] + ref[4]):
        assert rel_err(a, b) < TOL
    assert not got[1].requires_grad and not got[2].requires_grad
    again = run(torch.float32, device, True)
    assert torch.equal(again[0], got[0])


# --------------------------------------------------------------------------- BatchNorm
@pytest.mark.parametrize('rows,cols', [(300, 1000), (77, 250), (5000, 16)])
def test_bn_lrelu_train_and_eval(device, rows, cols):
    from tracevae_amd import ops
    x = _rand(rows, cols, seed=1) * 2 + 0.5
    gamma, beta = _rand(cols, seed=2), _rand(cols, seed=3)
    rm, rv = _rand(cols, seed=4), _rand(cols, seed=5).abs() + 0.5
    g = _rand(rows, cols, seed=6)
    eps = 1e-6
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.leaky_relu(F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, eps), 0.01)
    ref.backward(g)
    xd, gd, bd = (_dev(t, device).requires_grad_(True) for t in (x, gamma, beta))
    rmd, rvd = _dev(rm, device), _dev(rv, device)
    got = ops.bn_lrelu(xd, gd, bd, rmd, rvd, True, eps)
    got.backward(_dev(g, device))
    assert rel_err(got, ref) < TOL
    assert rel_err(xd.grad, xr.grad) < 5 * TOL
    assert rel_err(gd.grad, gr.grad) < TOL and rel_err(bd.grad, br.grad) < TOL
    assert rel_err(rmd, rm_ref) < TOL and rel_err(rvd, rv_ref) < TOL
    # eval mode
    xr2 = x.clone().requires_grad_(True)
    ref2 = F.leaky_relu(F.batch_norm(xr2, rm, rv, gamma, beta, False, 0.1, eps), 0.01)
    ref2.backward(g)
    xd2 = _dev(x, device).requires_grad_(True)
    got2 = ops.bn_lrelu(xd2, _dev(gamma, device), _dev(beta, device), _dev(rm, device),
                        _dev(rv, device), False, eps)
    got2.backward(_dev(g, device))
    assert rel_err(got2, ref2) < TOL and rel_err(xd2.grad, xr2.grad) < TOL


def test_bn_lrelu_weighted_rows_equal_expanded_batch(device):
    """A row of weight w stands for w identical rows (the compact latency decoder): output,
    running statistics and every gradient equal those of the expanded batch, the gradient of a
    weighted row being the sum over its copies."""
    from tracevae_amd import ops
    rows, cols, eps = 40, 24, 1e-6
    x = _rand(rows, cols, seed=1) * 2 + 0.5
    w = torch.randint(0, 5, (rows,), generator=torch.Generator().manual_seed(2))
    w[:20] = 1
    gamma, beta = _rand(cols, seed=3), _rand(cols, seed=4)
    rm, rv = _rand(cols, seed=5), _rand(cols, seed=6).abs() + 0.5
    idx = torch.repeat_interleave(torch.arange(rows), w)           # expanded batch -> source row
    g_exp = _rand(len(idx), cols, seed=7)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.leaky_relu(F.batch_norm(xr[idx], rm_ref, rv_ref, gr, br, True, 0.1, eps), 0.01)
    ref.backward(g_exp)
    # compact gradient of a row = sum of the gradients of its copies
    g_cmp = torch.zeros(rows, cols, dtype=torch.float64).index_add_(0, idx, g_exp)
    first = torch.full((rows,), -1, dtype=torch.long)
    first[idx.flip(0)] = torch.arange(len(idx)).flip(0)
    xd, gd, bd = (_dev(t, device).requires_grad_(True) for t in (x, gamma, beta))
    rmd, rvd = _dev(rm, device), _dev(rv, device)
    got = ops.bn_lrelu(xd, gd, bd, rmd, rvd, True, eps, row_weight=_dev(w.double(), device),
                       count=int(w.sum()))
    got.backward(_dev(g_cmp, device))
    live = w > 0
    assert rel_err(got[live.to(device)], ref[first[live]]) < TOL
    assert rel_err(xd.grad, xr.grad) < 5 * TOL
    assert rel_err(gd.grad, gr.grad) < TOL and rel_err(bd.grad, br.grad) < TOL
    assert rel_err(rmd, rm_ref) < TOL and rel_err(rvd, rv_ref) < TOL


# ------------------------------------------------------- GraphConv / readout / embedding
def test_graph_sum_segment_mean_embedding(device):
    from oracle.model import avg_pooling, graph_conv
    from tracevae_amd import ops
    db, ob = _graph(device)
    for which in ('enc', 'dec'):
        graph = getattr(db, which)
        src, dst = torch.as_tensor(ob[which + '_src']), torch.as_tensor(ob[which + '_dst'])
        rows = graph.num_rows
        for cols, out_cols in ((64, 80), (64, 2)):
            x, w, b = _rand(rows, cols, seed=1), _rand(cols, out_cols, seed=2), _rand(out_cols, seed=3)
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            ref = graph_conv(xr, src, dst, wr, b)
            g = _rand(rows, out_cols, seed=4)
            ref.backward(g)
            xd, wd = _dev(x, device).requires_grad_(True), _dev(w, device).requires_grad_(True)
            if cols > out_cols:
                got = ops.graph_sum(ops.rowdot(xd, wd), graph) + _dev(b, device)
            else:
                got = ops.matmul_nn(ops.graph_sum(xd, graph), wd, _dev(b, device))
            got.backward(_dev(g, device))
            assert rel_err(got, ref) < TOL
            assert rel_err(xd.grad, xr.grad) < TOL and rel_err(wd.grad, wr.grad) < TOL
    # readout (16-byte column path and the scalar one)
    rows = db.enc.num_rows
    for width in (1000, 250, 37):
        x = _rand(rows, width, seed=5)
        xr = x.clone().requires_grad_(True)
        ref = avg_pooling(xr, ob['node_offsets'])
        g = _rand(*ref.shape, seed=6)
        ref.backward(g)
        xd = _dev(x, device).requires_grad_(True)
        got = ops.segment_mean(xd, db.node_offsets, db.graph_id)
        got.backward(_dev(g, device))
        assert rel_err(got, ref) < TOL and rel_err(xd.grad, xr.grad) < TOL
    # embedding gather + per-operation segmented sum
    w = _rand(40, 40, seed=7)
    wr = w.clone().requires_grad_(True)
    ids = torch.as_tensor(ob['node_type'])
    ref = wr[ids]
    g = _rand(*ref.shape, seed=8)
    ref.backward(g)
    wd = _dev(w, device).requires_grad_(True)
    got = ops.embedding(wd, db.node_type)
    got.backward(_dev(g, device))
    assert torch.equal(got.detach().cpu().double(), ref.detach().float().double())
    assert rel_err(wd.grad, wr.grad) < TOL


def test_latdec_input(device):
    from tracevae_amd import ops
    db, ob = _graph(device)
    K, B, M = 3, db.batch_size, db.max_node_count
    z2, emb = _rand(K, B, 10, seed=1), _rand(40, 40, seed=2)
    ntp = torch.as_tensor(ob['node_type_pad'])
    z2r = z2.clone().requires_grad_(True)
    ref = torch.cat([z2r.unsqueeze(-2).expand(K, B, M, 10), emb[ntp].expand(K, B, M, 40)], -1)
    ref = ref.reshape(K * B * M, 50)
    g = _rand(*ref.shape, seed=3)
    ref.backward(g)
    z2d = _dev(z2, device).requires_grad_(True)
    got = ops.latdec_input(z2d.reshape(-1, 10), _dev(emb, device), db.node_type_pad, K)
    got.backward(_dev(g, device))
    assert torch.equal(got.detach().cpu(), ref.detach().float())
    assert rel_err(z2d.grad, z2r.grad) < TOL


def test_latdec_rows_compact(device):
    from tracevae_amd import ops
    db, ob = _graph(device)
    K, B, N = 3, db.batch_size, db.num_nodes
    z2, emb = _rand(K, B, 10, seed=1), _rand(40, 40, seed=2)
    row_graph, row_type = db.lat_row_graph.cpu(), db.lat_row_type.cpu()
    assert row_graph.shape[0] == N + B
    assert torch.equal(row_graph[N:], torch.arange(B)) and int(row_type[N:].abs().sum()) == 0
    z2r = z2.clone().requires_grad_(True)
    ref = torch.cat([z2r[:, row_graph], emb[row_type].expand(K, N + B, 40)], -1).reshape(-1, 50)
    g = _rand(*ref.shape, seed=3)
    ref.backward(g)
    z2d = _dev(z2, device).requires_grad_(True)
    got = ops.latdec_rows(z2d.reshape(-1, 10), _dev(emb, device), db.lat_row_graph,
                          db.lat_row_type, db.node_offsets, K)
    got.backward(_dev(g, device))
    assert torch.equal(got.detach().cpu(), ref.detach().float())
    assert rel_err(z2d.grad, z2r.grad) < TOL


@pytest.mark.parametrize('mode', ['normal', 'safe', 'biased'])
def test_latency_ll_compact_rows(device, mode):
    """Compact feats rows ([real nodes | one per trace]) give the same log-prob, std and
    gradients as the padded [K*B*M] rows they stand for."""
    from tracevae_amd import ops
    db, ob = _graph(device)
    K, B, M, N = 2, db.batch_size, db.max_node_count, db.num_nodes
    R = N + B
    fc = _rand(K * R, 2, seed=1, scale=1.5)
    off, cnt = db.node_offsets.cpu(), db.node_count.cpu()
    slot = torch.arange(M)[None, :]
    idx = torch.where(slot < cnt[:, None], off[:B, None] + slot, N + torch.arange(B)[:, None])
    idx = (idx[None] + R * torch.arange(K)[:, None, None]).reshape(-1)
    kw = dict(mode=mode, std_threshold=1.5, alpha=float(M), bias_threshold=0.5, logstd_min=-7,
              use_mask=True, weighted=True)
    fp = _dev(fc[idx], device).requires_grad_(True)
    want, want_std = ops.latency_ll(fp, db.latency_pad, db.node_count, db.node_type_pad, K, **kw)
    fd = _dev(fc, device).requires_grad_(True)
    got, std = ops.latency_ll(fd, db.latency_pad, db.node_count, db.node_type_pad, K,
                              node_offsets=db.node_offsets, **kw)
    assert torch.equal(got, want) and torch.equal(std, want_std)
    g = _dev(_rand(K * B, seed=6), device)
    want.backward(g)
    got.backward(g)
    ref_grad = torch.zeros(K * R, 2, device=device).index_add_(0, idx.to(device), fp.grad)
    assert rel_err(fd.grad, ref_grad) < TOL
    with pytest.raises(Exception):
        ops.latency_ll(fd, db.latency_pad, db.node_count, db.node_type_pad, K,
                       node_offsets=db.node_offsets, **dict(kw, use_mask=False))


# -------------------------------------------------------------------------- likelihoods
@pytest.mark.parametrize('M,D,K,biased', [(16, 40, 1, False), (16, 40, 3, True), (100, 40, 2, True),
                                          (70, 80, 1, False), (33, 40, 2, True), (129, 24, 1, True),
                                          # BASELINE configs[4]: 500-span traces, doubled
                                          # latent_embedding_size (124 750 node pairs per trace)
                                          (500, 80, 2, True), (500, 80, 1, False)])
def test_adj_bernoulli_ll(device, M, D, K, biased):
    from oracle.model import bernoulli_log_prob, biased as biased_fn, dense_to_triu
    from tracevae_amd import ops
    B = 5
    T = M * (M - 1) // 2
    h = _rand(K * B, M, D, seed=1, scale=0.4)
    adj = (torch.rand(B, T, generator=torch.Generator().manual_seed(2)) < 0.05).double()
    hr = h.clone().requires_grad_(True)
    logits = dense_to_triu(hr @ hr.transpose(-1, -2), M).reshape(K, B, T)
    lp = bernoulli_log_prob(adj, logits)
    if biased:
        lp = biased_fn(lp, M, 0.5)
    ref = lp.sum(-1).reshape(-1)
    g = _rand(K * B, seed=3)
    ref.backward(g)
    hd = _dev(h, device).requires_grad_(True)
    got = ops.adj_bernoulli_ll(hd, _dev(adj, device), K, biased, float(M), 0.5)
    got.backward(_dev(g, device))
    assert rel_err(got, ref) < TOL
    assert rel_err(hd.grad, hr.grad) < 5 * TOL


@pytest.mark.parametrize('M,D,G', [(2, 3, 1), (16, 10, 7), (100, 10, 128), (129, 24, 3), (500, 40, 2)])
def test_edge_logits_triu(device, M, D, G):
    """`tv_edge_logits_triu` against dense_to_triu(h h^T) (tensor_utils.py:54-67): every pair in
    the oracle's order, values to fp32 rounding of a D-term dot product."""
    from oracle.model import dense_to_triu
    from tracevae_amd import ops
    h = _rand(G, M, D, seed=5, scale=0.7)
    ref = dense_to_triu(h @ h.transpose(-1, -2), M)
    got = ops.edge_logits_triu(_dev(h, device))
    assert got.shape == (G, M * (M - 1) // 2)
    assert rel_err(got, ref) < 1e-6


@pytest.mark.parametrize('V,biased', [(1001, False), (33, True), (1488, False),
                                      (10001, False)])       # configs[4]: 10k-operation vocabulary
def test_categorical_ll(device, V, biased):
    from oracle.model import biased as biased_fn, categorical_log_prob
    from tracevae_amd import ops
    K, R = 3, 50
    logits = _rand(K * R, V, seed=1, scale=3.0)
    target = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(2))
    lr_ = logits.clone().requires_grad_(True)
    ref = categorical_log_prob(target.repeat(K), lr_)
    if biased:
        ref = biased_fn(ref, 1024.0, 0.5)
    g = _rand(K * R, seed=3)
    ref.backward(g)
    ld = _dev(logits, device).requires_grad_(True)
    got = ops.categorical_ll(ld, target.to(device), biased, 1024.0, 0.5)
    got.backward(_dev(g, device))
    assert rel_err(got, ref) < TOL and rel_err(ld.grad, lr_.grad) < TOL


def test_normal_reparam_and_ll(device):
    from oracle.model import normal_log_pdf
    from tracevae_amd import ops
    K, B, D = 4, 9, 10
    mean, ls, eps = _rand(B, D, seed=1), _rand(B, D, seed=2, scale=3.0), _rand(K, B, D, seed=3)
    mr, lr_ = mean.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    lsc = lr_.clamp(-7, 2)
    z = mr + torch.exp(lsc) * eps
    logq = normal_log_pdf(z, mr, lsc).sum(-1)
    gz, gq = _rand(K, B, D, seed=4), _rand(K, B, seed=5)
    (z * gz).sum().add((logq * gq).sum()).backward()
    md, ld = _dev(mean, device).requires_grad_(True), _dev(ls, device).requires_grad_(True)
    zd, qd, lsd = ops.normal_reparam(md, ld, _dev(eps, device), -7, 2)
    ((zd * _dev(gz, device)).sum() + (qd * _dev(gq, device)).sum()).backward()
    assert rel_err(zd, z) < TOL and rel_err(qd, logq) < TOL and rel_err(lsd, lsc) < TOL
    assert rel_err(md.grad, mr.grad) < TOL and rel_err(ld.grad, lr_.grad) < 5 * TOL
    # plain log-likelihood with clamp
    x = _rand(K, B, D, seed=6)
    m2, l2 = _rand(K, B, D, seed=7), _rand(K, B, D, seed=8, scale=3.0)
    xr, m2r, l2r = (t.clone().requires_grad_(True) for t in (x, m2, l2))
    ref = normal_log_pdf(xr, m2r, l2r.clamp(-5, 2)).sum(-1)
    g = _rand(K, B, seed=9)
    ref.backward(g)
    xd, m2d, l2d = (_dev(t, device).requires_grad_(True) for t in (x, m2, l2))
    got = ops.normal_ll(xd, m2d, l2d, -5, 2)
    got.backward(_dev(g, device))
    assert rel_err(got, ref) < TOL
    for a_, b_ in ((xd.grad, xr.grad), (m2d.grad, m2r.grad), (l2d.grad, l2r.grad)):
        assert rel_err(a_, b_) < TOL


@pytest.mark.parametrize('mode', ['normal', 'safe', 'biased', 'anomaly'])
@pytest.mark.parametrize('weighted,use_limit', [(False, False), (True, True)])
def test_latency_ll(device, mode, weighted, use_limit):
    from oracle.model import biased as biased_fn, normal_log_pdf
    from tracevae_amd import ops
    K, B, M, V = 2, 6, 12, 20
    feats = _rand(K * B * M, 2, seed=1, scale=(0.7 if mode == 'anomaly' else 2.0))
    x = _rand(B, M, 1, seed=2, scale=(1.0 if mode == 'anomaly' else 3.0))
    if mode == 'anomaly':
        # keep |x-mu|/std small: float32 erf saturates to 1 (log-prob -inf) long before the
        # float64 reference does, exactly as the reference's own float32 arithmetic would
        feats[:, 1] = feats[:, 1].abs() + 0.5
    node_count = torch.randint(2, M + 1, (B,), generator=torch.Generator().manual_seed(3))
    ntp = torch.randint(0, V, (B, M), generator=torch.Generator().manual_seed(4))
    std_limit = (_rand(V, seed=5).abs() + (2.0 if mode == 'anomaly' else 0.05)) if use_limit else None
    thr, alpha = (6.0, 1.0) if mode == 'safe' else (1.5, float(M))

    fr = feats.clone().requires_grad_(True)
    mu = fr[:, 0].reshape(K, B, M, 1)
    ls = fr[:, 1].reshape(K, B, M, 1).clamp(min=-7)
    if use_limit:
        ls = torch.minimum(ls, torch.log(std_limit[ntp].clamp(min=1e-7)).unsqueeze(-1))
    sd = torch.exp(ls)
    if mode == 'safe':
        lo, hi = (mu - thr * sd).detach(), (mu + thr * sd).detach()
        lp = normal_log_pdf(torch.maximum(torch.minimum(x, hi), lo), mu, ls)
    elif mode == 'biased':
        lp = normal_log_pdf(x, mu, ls)
        lp = torch.where((x - mu).abs() > thr * sd, lp * alpha, lp)
    elif mode == 'anomaly':
        t = ((x - mu) / sd).abs() - thr
        lp = biased_fn(-math.log(2.0) + torch.log1p(-torch.erf(t / math.sqrt(2.0))), alpha, 0.5)
    else:
        lp = normal_log_pdf(x, mu, ls)
    mask = (torch.arange(M).unsqueeze(0) < node_count.unsqueeze(-1)).double().unsqueeze(-1)
    lp = lp * mask
    if weighted:
        lp = lp * (float(M) / node_count.double()).reshape(-1, 1, 1)
    ref = lp.sum((-1, -2)).reshape(-1)

    fd = _dev(feats, device).requires_grad_(True)
    got, std = ops.latency_ll(
        fd, _dev(x, device), node_count.to(device), ntp.to(device), K, mode=mode,
        std_threshold=thr, alpha=alpha, bias_threshold=0.5, logstd_min=-7,
        std_limit=None if std_limit is None else _dev(std_limit, device), use_mask=True,
        weighted=weighted)
    assert rel_err(got, ref) < 2 * TOL
    assert rel_err(std, sd.reshape(-1)) < TOL
    if mode != 'anomaly':
        g = _rand(K * B, seed=6)
        ref.backward(g)
        got.backward(_dev(g, device))
        assert rel_err(fd.grad, fr.grad) < 2 * TOL


def test_score_reduce(device):
    from tracevae_amd import ops
    w = _rand(10, 33, seed=1, scale=30.0)
    w[3, 5] = float('nan')
    w[:, 7] = 1e9
    ref_loss = -w.mean(0)
    ref_nll = -(torch.logsumexp(w, 0) - math.log(10))
    clip = torch.tensor(100000.0, dtype=torch.float64)
    ref_loss = torch.where(ref_loss < clip, ref_loss, clip)
    ref_nll = torch.where(ref_nll < clip, ref_nll, clip)
    loss, nll = ops.score_reduce(_dev(w, device), 100000.0)
    assert rel_err(loss, ref_loss) < TOL and rel_err(nll, ref_nll) < TOL
    assert float(nll[5]) == 100000.0                     # NaN -> clip


# ---------------------------------------------------------------------------- optimizer
@pytest.mark.parametrize('kind', ['rmsprop', 'adam'])
def test_flat_optimizer(device, kind):
    from tracevae_amd.train import FlatOptimizer, FlatParameters
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(30, 50), torch.nn.Linear(50, 7))
    ref = torch.nn.Sequential(torch.nn.Linear(30, 50), torch.nn.Linear(50, 7))
    ref.load_state_dict(net.state_dict())
    net.to(device)
    flat = FlatParameters(net, tail_prefix='1.')
    opt = FlatOptimizer(flat, kind)
    opt.set_lr(1e-2)
    ropt = (torch.optim.RMSprop(ref.parameters(), lr=1e-2, alpha=0.99, eps=1e-8) if kind == 'rmsprop'
            else torch.optim.Adam(ref.parameters(), lr=1e-2))
    for step in range(3):
        x = torch.randn(16, 30)
        flat.zero_grad()
        (net(x.to(device)) ** 2).sum().backward()
        opt.step(10.0)
        ropt.zero_grad()
        (ref(x) ** 2).sum().backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 10.0)
        ropt.step()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p, q) < 1e-4, n


# ---------------------------------------------------------------------------------------
# RealNVP prior: the fused kernels against the PyTorch formulation of the same flow in float64
# (itself checked against the oracle by the model-level tests); tolerance 1e-4 relative
# ---------------------------------------------------------------------------------------
def _random_flow(dim, levels, seed, device):
    from tracevae_amd.config import RealNVPFlowConfig
    from tracevae_amd.model import InverseRealNVP
    torch.manual_seed(seed)
    cfg = RealNVPFlowConfig()
    cfg.flow_levels = levels
    flow = InverseRealNVP(dim, cfg)
    with torch.no_grad():
        for name, p in flow.named_parameters():
            if name.endswith('.kernel'):
                p.add_(0.2 * torch.randn_like(p))                # not orthogonal any more
            elif 'shared' in name:
                p.add_(0.05 * torch.randn_like(p))
            else:                                                # zero-initialised tensors
                p.copy_(0.3 * torch.randn_like(p))
    for m in flow.modules():
        if hasattr(m, 'initialized'):
            m.initialized = True
    return flow.to(device)


@pytest.mark.parametrize('dim,levels,rows', [(10, 5, 512), (10, 5, 1), (10, 1, 63), (10, 5, 65),
                                             (10, 8, 1300), (16, 3, 200), (20, 5, 130)])
def test_realnvp_log_prob(device, dim, levels, rows):
    flow = _random_flow(dim, levels, 100 + rows, device)
    ref = _random_flow(dim, levels, 100 + rows, device).double()
    gen = torch.Generator(device='cpu').manual_seed(rows)
    z = (1.5 * torch.randn(rows, dim, generator=gen)).to(device)
    w = torch.randn(rows, generator=gen).to(device)
    zf = z.clone().requires_grad_(True)
    lp = flow.log_prob(zf)
    (lp * w).sum().backward()
    zr = z.double().requires_grad_(True)
    lp_ref = flow_log_prob_reference(ref, zr)
    (lp_ref * w.double()).sum().backward()
    assert rel_err(lp, lp_ref) < 1e-4
    assert rel_err(zf.grad, zr.grad) < 1e-4
    for (name, p), (_, q) in zip(flow.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, name
        assert rel_err(p.grad, q.grad) < 1e-4, name


def test_realnvp_k_samples_frozen_parameters_and_fallback(device):
    flow = _random_flow(10, 5, 7, device)
    ref = _random_flow(10, 5, 7, device).double()
    z = torch.randn(3, 40, 10, device=device)
    assert rel_err(flow.log_prob(z), flow_log_prob_reference(ref, z.double())) < 1e-4   # [K, B] leading dims
    assert flow.log_prob(z).shape == (3, 40)
    # frozen parameters (phase 2 of training): only dz comes back
    for p in flow.parameters():
        p.requires_grad_(False)
    zf = z.clone().requires_grad_(True)
    flow.log_prob(zf).sum().backward()
    zr = z.double().requires_grad_(True)
    flow_log_prob_reference(ref, zr).sum().backward()
    assert rel_err(zf.grad, zr.grad) < 1e-4
    assert all(p.grad is None for p in flow.parameters())
    # empty input
    assert flow.log_prob(torch.zeros(0, 10, device=device)).shape == (0,)
    # a flow the kernels do not cover is refused when it is built (one implementation only)
    from tracevae_amd.config import RealNVPFlowConfig
    from tracevae_amd.model import InverseRealNVP
    for field, value in (('coupling_hidden_layer_units', 32), ('coupling_hidden_layer_count', 2),
                         ('coupling_layer_scale', 'exp'), ('flow_levels', 9)):
        cfg = RealNVPFlowConfig()
        setattr(cfg, field, value)
        with pytest.raises(NotImplementedError):
            InverseRealNVP(10, cfg)
    with pytest.raises(NotImplementedError):
        InverseRealNVP(12, RealNVPFlowConfig())
    # bad shapes are refused on the host
    from tracevae_amd import ops
    params, levels, hidden = flow._kernel_params(), flow.levels, flow.hidden
    with pytest.raises(ValueError):
        ops.realnvp_log_prob(z, params[:-1], levels, hidden, 0.01)
    with pytest.raises(ValueError):
        ops.realnvp_log_prob(torch.randn(4, 10, device=device), [params[1]] + params[1:], 5, 32, 0.01)


@pytest.mark.parametrize('dim,levels,rows', [(10, 5, 64), (10, 5, 1), (20, 3, 700), (16, 8, 1024),
                                             # more than one lane's worth of rows per lane: the
                                             # strided form (init batches above 1024 rows, or a
                                             # first call on K*B rows)
                                             (10, 5, 1025), (20, 4, 5000), (10, 5, 66000)])
def test_realnvp_actnorm_data_dependent_init(device, dim, levels, rows):
    """The first batch initialises every ActNorm (tensorkit's data-dependent init, reference
    train.py:188-199): `tv_realnvp_actnorm_init` against the level-by-level tensor formulation
    in float64, and the log-density of that first batch."""
    flow = _random_flow(dim, levels, 300 + rows, device)
    ref = _random_flow(dim, levels, 300 + rows, device).double()
    for f in (flow, ref):
        for m in f.modules():
            if hasattr(m, 'initialized'):
                m.initialized = False
    gen = torch.Generator(device='cpu').manual_seed(rows)
    z = (2.0 * torch.randn(rows, dim, generator=gen) + 0.5).to(device)
    with torch.no_grad():
        lp = flow.log_prob(z)
        lp_ref = flow_log_prob_reference(ref, z.double(), init_act_norm=True)
    assert all(m.initialized for m in flow.modules() if hasattr(m, 'initialized'))
    for (name, p), (_, q) in zip(flow.named_parameters(), ref.named_parameters()):
        if name.endswith('.bias') and 'flows' in name or name.endswith('pre_scale'):
            assert rel_err(p, q) < 1e-4, name
    assert rel_err(lp, lp_ref) < 1e-4
    # the second batch leaves them alone
    before = [p.detach().clone() for p in flow.parameters()]
    with torch.no_grad():
        flow.log_prob(z[: max(rows // 2, 1)] * 3.0)
    assert all(torch.equal(a, b) for a, b in zip(before, flow.parameters()))
    # the one-lane-per-row form and the strided form agree where both apply
    if rows == 1024:
        from tracevae_amd import ops
        a = _random_flow(dim, levels, 7, device)
        b = _random_flow(dim, levels, 7, device)
        with torch.no_grad():
            ops.realnvp_actnorm_init(z, a._kernel_params(), levels, a.hidden, 0.01)
            # 1025 rows whose last one repeats row 0 twice over is another batch; instead
            # compare on statistics: the strided form on [z ; z] has the same mean and (biased)
            # variance as the direct form on z
            ops.realnvp_actnorm_init(torch.cat([z, z]), b._kernel_params(), levels, b.hidden, 0.01)
        for (name, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            assert rel_err(p, q) < 1e-5, name


# ---------------------------------------------------------------------------------------
# BatchNorm + LeakyReLU inside the GEMM (tv_gemm_f32_act) and the fused layer operator
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize('m,n,k,ta,tb,operand', [
    (300, 200, 96, False, True, 1),        # forward product shape: act(x) . W^T
    (257, 130, 77, False, True, 1),        # nothing aligned: 4-byte loads, partial k-tile
    (128, 128, 1000, False, False, 1),     # A prologue with a k-major B
    (1000, 1000, 5000, True, False, 2),    # weight gradient: split over K
    (64, 100, 333, True, False, 2),        # odd sizes, B prologue
    (200, 72, 130, False, False, 2),       # B prologue with a row-major A
    (51200, 1000, 1000, False, True, 1),   # the layer shape of the bench (tail split)
])
def test_gemm_act_prologue(device, m, n, k, ta, tb, operand):
    from tracevae_amd import ops
    gen = torch.Generator(device='cpu').manual_seed(m + n + k)
    a = torch.randn((k, m) if ta else (m, k), generator=gen).to(device)
    b = torch.randn((n, k) if tb else (k, n), generator=gen).to(device)
    feats = k if operand == 1 else n
    scale = (0.5 + torch.rand(feats, generator=gen)).to(device)
    shift = (0.3 * torch.randn(feats, generator=gen)).to(device)
    bias = torch.randn(n, generator=gen).to(device)
    got = ops.gemm(a, b, ta, tb, bias=bias, act=(operand, scale, shift, 0.01))
    A = (a.t() if ta else a).double()
    Bm = (b.t() if tb else b).double()
    if operand == 1:
        A = F.leaky_relu(A * scale.double() + shift.double(), 0.01)
    else:
        Bm = F.leaky_relu(Bm * scale.double() + shift.double(), 0.01)
    want = A @ Bm + bias.double()
    assert rel_err(got, want) < 2e-5


def test_gemm_act_refuses_bad_arguments(device):
    from tracevae_amd import hip, ops
    a = torch.randn(64, 32, device=device)
    b = torch.randn(32, 48, device=device)
    s = torch.ones(32, device=device)
    with pytest.raises(hip.HipLibraryError):      # A prologue on a transposed A
        ops.gemm(a.t().contiguous(), b, True, False, act=(1, s, s, 0.01))
    with pytest.raises(hip.HipLibraryError):      # B prologue on a transposed B
        ops.gemm(a, b.t().contiguous(), False, True, act=(2, torch.ones(48, device=device),
                                                          torch.ones(48, device=device), 0.01))
    with pytest.raises(ValueError):               # wrong vector length
        ops.gemm(a, b, False, False, act=(1, torch.ones(31, device=device), s, 0.01))


@pytest.mark.parametrize('rows,cols,out,weighted', [(700, 96, 64, False), (333, 50, 37, False),
                                                    (900, 128, 200, True)])
def test_bn_lrelu_linear_equals_composition(device, rows, cols, out, weighted):
    """The fused operator against bn_lrelu followed by linear: values, all four gradients and
    the running statistics."""
    from tracevae_amd import ops
    gen = torch.Generator(device='cpu').manual_seed(rows + cols)
    x0 = (torch.randn(rows, cols, generator=gen) * 2 + 0.5).to(device)
    w0 = (torch.randn(out, cols, generator=gen) / cols ** 0.5).to(device)
    g0 = (0.5 + torch.rand(cols, generator=gen)).to(device)
    b0 = (0.2 * torch.randn(cols, generator=gen)).to(device)
    dft = torch.randn(rows, out, generator=gen).to(device)
    rw = count = None
    if weighted:
        rw = torch.ones(rows, device=device)
        rw[-50:] = torch.randint(1, 9, (50,), generator=gen).float().to(device)
        count = int(rw.sum().item())
    res = []
    for fused in (True, False):
        x, w, g, b = (t.clone().requires_grad_(True) for t in (x0, w0, g0, b0))
        rm, rv = torch.zeros(cols, device=device), torch.ones(cols, device=device)
        if fused:
            ft = ops.bn_lrelu_linear(x, g, b, w, rm, rv, 1e-6, 0.1, 0.01, None, rw, count)
        else:
            ft = ops.linear(ops.bn_lrelu(x, g, b, rm, rv, True, 1e-6, 0.1, 0.01, None, rw, count), w)
        ft.backward(dft)
        res.append((ft.detach(), x.grad, g.grad, b.grad, w.grad, rm, rv))
    for i, name in enumerate(('ft', 'dx', 'dgamma', 'dbeta', 'dw', 'running_mean', 'running_var')):
        assert rel_err(res[0][i], res[1][i]) < 2e-5, name


@pytest.mark.parametrize('kind,weighted', [('graph_sum', False), ('segment_mean', False),
                                           ('segment_mean', True)])
def test_bn_lrelu_tail_equals_composition(device, kind, weighted):
    """A stack's tail with the BatchNorm + LeakyReLU of its last GAT layer applied inside the
    tail's kernel (`bn_lrelu_graph_sum` / `bn_lrelu_segment_mean`) against bn_lrelu followed by
    the plain tail: values, the three gradients and the running statistics."""
    from tracevae_amd import ops
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    from helpers import make_flat
    V, M, cols = 30, 24, 72
    flat, lr = make_flat(40, V, M, seed=9)
    store = DeviceTraceStore(flat, lr, V)
    batch = TraceGraphBatch.from_store(store, np.arange(len(flat)), M).build_dgl()
    g = batch.dec if kind == 'graph_sum' else batch.enc
    rows = g.num_rows
    gen = torch.Generator(device='cpu').manual_seed(rows)
    x0 = (torch.randn(rows, cols, generator=gen) * 2 + 0.5).to(device)
    g0 = (0.5 + torch.rand(cols, generator=gen)).to(device)
    b0 = (0.2 * torch.randn(cols, generator=gen)).to(device)
    rw = count = None
    if weighted:
        rw = torch.ones(rows, device=device)
        rw[-20:] = torch.randint(1, 9, (20,), generator=gen).float().to(device)
        count = int(rw.sum().item())
    res = []
    for fused in (True, False):
        x, gm, bt = (t.clone().requires_grad_(True) for t in (x0, g0, b0))
        rm, rv = torch.zeros(cols, device=device), torch.ones(cols, device=device)
        if kind == 'graph_sum':
            if fused:
                out = ops.bn_lrelu_graph_sum(x, gm, bt, rm, rv, 1e-6, g, 0.1, 0.01, None, rw, count)
                assert float(ops.operand_max(out).slots.max()) == float(out.detach().abs().max())
            else:
                out = ops.graph_sum(ops.bn_lrelu(x, gm, bt, rm, rv, True, 1e-6, 0.1, 0.01, None, rw, count), g)
        else:
            if fused:
                out = ops.bn_lrelu_segment_mean(x, gm, bt, rm, rv, 1e-6, batch.node_offsets,
                                                batch.graph_id, 0.1, 0.01, None, rw, count)
            else:
                out = ops.segment_mean(ops.bn_lrelu(x, gm, bt, rm, rv, True, 1e-6, 0.1, 0.01, None, rw, count),
                                       batch.node_offsets, batch.graph_id)
        dout = torch.randn(*out.shape, generator=torch.Generator().manual_seed(1)).to(device)
        out.backward(dout)
        res.append((out.detach(), x.grad, gm.grad, bt.grad, rm, rv))
    for i, name in enumerate(('out', 'dx', 'dgamma', 'dbeta', 'running_mean', 'running_var')):
        assert rel_err(res[0][i], res[1][i]) < 2e-5, name


def test_gemm_operand_maxima_kept_by_the_caller(device):
    """`tv_gemm_f32_ex`: the maxima a first product leaves in the caller's buffer (state 1) give
    the later products of the same operand views (state 2) bit for bit what a scan of their own
    gives, on both fp16 forms; a product on the bf16 form leaves the buffer unmarked."""
    from tracevae_amd import hip, ops
    gen = torch.Generator(device='cpu').manual_seed(5)
    rows, width = 33000, 1004
    x = torch.randn(rows, 1000, generator=gen).to(device) * 3.0
    w = (torch.randn(width, 1000, generator=gen) * 0.05).to(device)
    dy = (torch.randn(rows, width, generator=gen) * 1e-4).to(device)
    scale = (0.5 + torch.rand(1000, generator=gen)).to(device)
    shift = (0.3 * torch.randn(1000, generator=gen)).to(device)
    act1, act2 = (1, scale, shift, 0.01), (2, scale, shift, 0.01)
    xm, wm, dm = ops.OperandMax(device), ops.OperandMax(device), ops.OperandMax(device)
    ft = ops.gemm(x, w, False, True, act=act1, a_max=xm, b_max=wm)
    assert hip.lib().tv_gemm_f32_last_form() == 1 and xm.valid and wm.valid
    assert torch.equal(ft, ops.gemm(x, w, False, True, act=act1))
    amax = float(torch.nn.functional.leaky_relu(x * scale + shift, 0.01).abs().max())
    assert float(xm.slots.max()) == pytest.approx(amax, rel=1e-6)
    assert float(wm.slots.max()) == float(w.abs().max())
    dx = ops.gemm(dy, w, False, False, a_max=dm, b_max=wm)          # weights: state 2
    assert dm.valid and torch.equal(dx, ops.gemm(dy, w, False, False))
    dw = ops.gemm(dy, x, True, False, act=act2, a_max=dm, b_max=xm)  # both: state 2
    assert hip.lib().tv_gemm_f32_last_form() == 2
    assert torch.equal(dw, ops.gemm(dy, x, True, False, act=act2))
    # the bf16 form (few rows) does not touch the buffers
    sm = ops.OperandMax(device)
    ops.gemm(x[:300], w, False, True, a_max=sm)
    assert hip.lib().tv_gemm_f32_last_form() == 0 and not sm.valid


@pytest.mark.parametrize('n_rep,epilogue', [(1, True), (3, True), (1, False)])
def test_gat_forward_windowed_equals_per_row(device, n_rep, epilogue, monkeypatch):
    """The windowed forward aggregation (32 rows staged in LDS per workgroup; taken for
    [ft | el | er] rows without statistics from 512 windows on) adds a row's neighbours in the
    same edge order with the same arithmetic as the per-row kernel: bit-equal outputs and maxima.
    The graph has chains inside and across windows, a row with 80 in-neighbours outside its
    window (the three-pass path), rows without in-edges and a row count that is no multiple of
    the window."""
    from tracevae_amd import ops
    H, F = 2, 500
    R = 16400 // n_rep + 7
    rng = np.random.RandomState(4)
    nbrs = [[r] for r in range(R)]                        # self loops
    for r in range(R - 1):
        if r % 50 != 49:                                  # chains of 50 rows, both directions
            nbrs[r].append(r + 1)
            nbrs[r + 1].append(r)
    for r in range(0, R - 40, 97):                        # edges that leave the window
        nbrs[r].append(r + 37)
    nbrs[7] = [7] + list(range(100, 180))                 # more than 64 in-neighbours
    nbrs[33] = []                                         # no in-edge at all
    nbrs[R - 1] = []                                      # ... and the last row (indptr = E)
    indptr = np.zeros(R + 1, np.int64)
    indptr[1:] = np.cumsum([len(x) for x in nbrs])
    indices = np.concatenate([np.asarray(x, np.int64) for x in nbrs if len(x)])
    ti = lambda a_: torch.as_tensor(a_).to(device)
    z = ti(np.zeros(len(indices), np.int64))
    g = ops.DeviceGraph(ti(indptr), ti(indices), z, z, z, z)
    gen = torch.Generator().manual_seed(11)
    ft = torch.randn(R * n_rep, H * F + 4, generator=gen).to(device)
    al, ar = 0.05 * torch.randn(1, H, F, generator=gen).to(device), 0.05 * torch.randn(1, H, F, generator=gen).to(device)
    bias = 0.1 * torch.randn(H * F, generator=gen).to(device)
    sc = (0.5 + torch.rand(H * F, generator=gen)).to(device) if epilogue else None
    sh = (0.2 * torch.randn(H * F, generator=gen)).to(device) if epilogue else None
    assert (R * n_rep + 31) // 32 >= 512
    with torch.no_grad():
        out_w, _ = ops.gat_conv(ft, al, ar, bias, g, n_rep, ep_scale=sc, ep_shift=sh)
        monkeypatch.setenv('TV_GAT_LDS_MIN_WINDOWS', '1000000000')
        out_r, _ = ops.gat_conv(ft, al, ar, bias, g, n_rep, ep_scale=sc, ep_shift=sh)
    assert torch.equal(out_w, out_r)
    assert float(out_w[33 if n_rep == 1 else R + 33].abs().max()) == float(
        out_r[33 if n_rep == 1 else R + 33].abs().max())
    if epilogue:
        bw, br = ops.operand_max(out_w), ops.operand_max(out_r)
        assert float(bw.slots.max()) == float(br.slots.max()) == float(out_r.abs().max())


def test_gemm_profile_records_the_main_kernel_of_every_product(device):
    """`tv_gemm_profile_begin / _end` (the bench's kernel-alone timing): one record per product in
    launch order with its form and shape, a positive duration that is shorter than the call's own
    wall time, nothing recorded outside the bracket."""
    import time
    from tracevae_amd import hip, ops
    a = torch.randn(20000, 512, device=device)
    w = torch.randn(768, 512, device=device)
    s = torch.randn(300, 40, device=device)
    ops.gemm(a, w, False, True)                          # warm-up, outside the bracket
    hip.lib().tv_gemm_profile_begin()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.gemm(a, w, False, True)                          # wide form
    ops.gemm(s, s, True, False)                          # producer / consumer form
    ops.gemm(a, a, True, False)                          # k-major wide form
    torch.cuda.synchronize()
    wall_ms = 1e3 * (time.perf_counter() - t0)
    buf = np.zeros([8, 5])
    n = int(hip.lib().tv_gemm_profile_end(buf.ctypes.data, 8))
    assert n == 3
    assert [int(f) for f in buf[:3, 1]] == [1, 0, 2]
    assert [tuple(int(v) for v in r[2:]) for r in buf[:3]] == [(20000, 768, 512), (40, 40, 300),
                                                               (512, 512, 20000)]
    assert (buf[:3, 0] > 0).all() and buf[:3, 0].sum() < wall_ms
    ops.gemm(a, w, False, True)
    assert int(hip.lib().tv_gemm_profile_end(buf.ctypes.data, 8)) == 0


def test_operand_maxima_from_the_producing_kernels(device):
    """The kernels that write a GEMM operand leave its maxima with it: the aggregation with the
    eval epilogue (exact: the next layer's product is bit-equal to one that scans), the backward
    aggregation (exact, over the whole [dft | del | der | 0] rows), and the BatchNorm finalize
    (a bound: the product stays within the GEMM's error bound of the float64 result)."""
    from tracevae_amd import hip, ops
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    from helpers import make_flat
    V, M, H, F = 50, 40, 2, 500
    flat, lr = make_flat(900, V, M, seed=3)
    store = DeviceTraceStore(flat, lr, V)
    batch = TraceGraphBatch.from_store(store, np.arange(len(flat)), M).build_dgl()
    g = batch.dec                                     # 900 * 40 = 36000 rows: the wide GEMM form
    gen = torch.Generator(device='cpu').manual_seed(1)
    rnd = lambda *sh: torch.randn(*sh, generator=gen).to(device)
    x = rnd(g.num_rows, 64)
    w, al, ar, bias = 0.1 * rnd(H * F, 64), 0.1 * rnd(1, H, F), 0.1 * rnd(1, H, F), 0.1 * rnd(H * F)
    w2 = 0.05 * rnd(H * F, H * F)
    sc, sh = 0.5 + torch.rand(H * F, generator=gen).to(device), 0.2 * rnd(H * F)
    with torch.no_grad():
        ft = ops.linear_scores(x, w, al, ar)
        out, _ = ops.gat_conv(ft, al, ar, bias, g, 1, ep_scale=sc, ep_shift=sh, ep_slope=0.01)
        box = ops.operand_max(out)
        assert box is not None and float(box.slots.max()) == float(out.abs().max())
        before = dict(ops.OPERAND_SCANS)
        y = ops.linear_scores(out, w2, al, ar)
        assert hip.lib().tv_gemm_f32_last_form() == 1
        # out: from the aggregation; the extended weights: from the kernel that wrote them
        assert ops.OPERAND_SCANS['kept'] == before['kept'] + 2
        assert ops.OPERAND_SCANS['scanned'] == before['scanned']
        assert torch.equal(y, ops.linear_scores(out.clone(), w2, al, ar))
        out.add_(1.0)                                  # written to: the maxima no longer apply
        assert ops.operand_max(out) is None
    # GraphConv: the aggregated rows carry their maxima into the product that follows, which keeps
    # the maxima of x, of the weights and of the output gradient between its three products
    wn = (0.05 * rnd(H * F, 101)).requires_grad_(True)         # [in, out], an odd class count
    hsrc = rnd(g.num_rows, H * F).requires_grad_(True)
    agg = ops.graph_sum(hsrc, g)
    box = ops.operand_max(agg)
    assert box is not None and float(box.slots.max()) == float(agg.detach().abs().max())
    before = dict(ops.OPERAND_SCANS)
    yl = ops.matmul_nn(agg, wn)
    assert ops.OPERAND_SCANS['kept'] - before['kept'] >= 1     # agg: from the aggregation
    assert torch.equal(yl.detach(), ops.matmul_nn(agg.detach().clone(), wn.detach()))
    before = dict(ops.OPERAND_SCANS)
    yl.backward(rnd(*yl.shape))
    assert hsrc.grad is not None and wn.grad is not None
    # backward: the gradient rows carry their maxima into the fc layer's two products
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    ft = ops.linear_scores(xg, wg, al, ar)
    out, _ = ops.gat_conv(ft, al, ar, bias.clone().requires_grad_(True), g, 1)
    dout = rnd(*out.shape)
    before = dict(ops.OPERAND_SCANS)
    out.backward(dout)
    kept = ops.OPERAND_SCANS['kept'] - before['kept']
    assert kept >= 2, (before, ops.OPERAND_SCANS)      # dft in the dx and dW products (+ x, weights)
    # the BatchNorm bound against the float64 product
    rows = 40000
    xb = 3.0 * rnd(rows, H * F) + 1.0
    gamma, beta = 0.5 + torch.rand(H * F, generator=gen).to(device), 0.3 * rnd(H * F)
    rm, rv = torch.zeros(H * F, device=device), torch.ones(H * F, device=device)
    before = dict(ops.OPERAND_SCANS)
    got = ops.bn_lrelu_linear(xb, gamma, beta, w2, rm, rv, 1e-6)
    assert hip.lib().tv_gemm_f32_last_form() == 1
    assert ops.OPERAND_SCANS['kept'] == before['kept'] + 1               # x: the bound
    xd = xb.double()
    xhat = (xd - xd.mean(0)) / torch.sqrt(xd.var(0, unbiased=False) + 1e-6)
    act = torch.nn.functional.leaky_relu(xhat * gamma.double() + beta.double(), 0.01)
    want = act @ w2.double().t()
    scale = act.abs() @ w2.double().abs().t()
    assert float(((got.double() - want).abs() / scale).max()) < 2e-6


@pytest.mark.parametrize('m,n,k', [(4096, 600, 500), (4096, 96, 40), (2500, 50, 300), (3000, 128, 132)])
@pytest.mark.parametrize('trans_b', [0, 1])
def test_gemm_takes_a_prepared_image_of_b(device, m, n, k, trans_b):
    """`tv_gemm_prepare_b_batch` + `tv_gemm_f32_ex` states 3 / 4 (B is a plane image made
    beforehand, several operands per launch) against the call that splits B itself: the same
    bits, both forms, both orientations of B; a wrong form is refused."""
    import ctypes
    from tracevae_amd import hip, ops
    g = torch.Generator().manual_seed(m + n + k + trans_b)

    def product(mm, nn, kk, tb, a, b, ldb, box, state, out):
        ws = torch.empty([hip.query('tv_gemm_f32_workspace_bytes', mm, nn, kk)], dtype=torch.uint8,
                         device=device)
        hip.call('tv_gemm_f32_ex', 0, tb, mm, nn, kk, a.data_ptr(), kk, b.data_ptr(), ldb,
                 out.data_ptr(), nn, None, None, 0, 0, None, None, 0.0, None, 0,
                 box.slots.data_ptr(), state, ws.data_ptr(), ws.numel())
        return out

    a = torch.randn(m, k, generator=g).to(device)
    b = (torch.randn(n, k, generator=g) if trans_b else torch.randn(k, n, generator=g)).to(device)
    form = hip.query('tv_gemm_b_form', m, n, k)
    assert form in (0, 1)
    box = ops.OperandMax(device)
    want = product(m, n, k, trans_b, a, b, b.stride(0), box, 1, torch.empty(m, n, device=device))
    # a second operand in the same launch: the batch kernel finds its item by block ranges
    m2, n2, k2 = 4096, 132, 76
    a2 = torch.randn(m2, k2, generator=g).to(device)
    b2 = torch.randn(k2, n2, generator=g).to(device)
    form2 = hip.query('tv_gemm_b_form', m2, n2, k2)
    box2 = ops.OperandMax(device)
    want2 = product(m2, n2, k2, 0, a2, b2, n2, box2, 1, torch.empty(m2, n2, device=device))
    image = torch.empty([hip.query('tv_gemm_prepared_b_bytes', m, n, k)], dtype=torch.uint8, device=device)
    image2 = torch.empty([hip.query('tv_gemm_prepared_b_bytes', m2, n2, k2)], dtype=torch.uint8,
                         device=device)
    i64, i32, ptrs = ctypes.c_int64 * 2, ctypes.c_int * 2, ctypes.c_void_p * 2
    hip.call('tv_gemm_prepare_b_batch', 2, i64(m2, m), i64(n2, n), i64(k2, k), i32(0, trans_b),
             ptrs(b2.data_ptr(), b.data_ptr()), i64(n2, b.stride(0)),
             ptrs(box2.slots.data_ptr(), box.slots.data_ptr()), ptrs(image2.data_ptr(), image.data_ptr()))
    got = product(m, n, k, trans_b, a, image, b.stride(0), box, 3 + form, torch.empty(m, n, device=device))
    assert torch.equal(got, want)
    got2 = product(m2, n2, k2, 0, a2, image2, n2, box2, 3 + form2, torch.empty(m2, n2, device=device))
    assert torch.equal(got2, want2)
    assert rel_err(want, a @ (b.t() if trans_b else b)) < 1e-5
    with pytest.raises(hip.HipLibraryError):
        product(m, n, k, trans_b, a, image, b.stride(0), box, 4 - form, got)


def test_batched_preparation_and_guarded_update_refuse_what_they_cannot_take(device):
    """Error paths of this round's entry points: more operands than a launch takes, a guard kind
    without a guard, and a `guard_host` that is not pinned host memory (refused by the host-side
    lookup of its device address - never a faulting store from the kernel)."""
    import ctypes
    from tracevae_amd import hip
    lib = hip.lib()
    n = 33
    i64, i32, ptrs = ctypes.c_int64 * n, ctypes.c_int * n, ctypes.c_void_p * n
    w = torch.randn(128, 128, device=device)
    img = torch.empty([hip.query('tv_gemm_prepared_b_bytes', 4096, 128, 128)], dtype=torch.uint8, device=device)
    box = torch.zeros(256, device=device)
    args = (i64(*[4096] * n), i64(*[128] * n), i64(*[128] * n), i32(*[0] * n), ptrs(*[w.data_ptr()] * n),
            i64(*[128] * n), ptrs(*[box.data_ptr()] * n), ptrs(*[img.data_ptr()] * n))
    assert lib.tv_gemm_prepare_b_batch(n, *args, hip.stream()) != 0
    assert b'tv_gemm_prepare_b_batch' in lib.tv_last_error()
    m = 17
    i64m, ptrm = ctypes.c_int64 * m, ctypes.c_void_p * m
    a = torch.randn(2, 64, device=device)
    wx = torch.empty(128 + 4, 128, device=device)
    assert lib.tv_gat_fc_score_weight_batch(
        m, i64m(*[2] * m), i64m(*[64] * m), i64m(*[128] * m), ptrm(*[w.data_ptr()] * m),
        ptrm(*[a.data_ptr()] * m), ptrm(*[a.data_ptr()] * m), ptrm(*[wx.data_ptr()] * m),
        ptrm(*[box.data_ptr()] * m), hip.stream()) != 0
    assert b'tv_gat_fc_score_weight_batch' in lib.tv_last_error()

    p = torch.ones(64, device=device)
    g = torch.ones(64, device=device)
    s1 = torch.zeros(64, device=device)
    guard = torch.zeros(1, device=device)
    common = (0, 64, p.data_ptr(), g.data_ptr(), s1.data_ptr(), None, None, 0.0, 1e-3, 0.0, 0.99, 1e-8, 1)
    assert lib.tv_optim_step_guarded(*common, None, 1, None, hip.stream()) != 0          # kind, no guard
    assert lib.tv_optim_step_guarded(*common, guard.data_ptr(), 0, None, hip.stream()) != 0   # guard, no kind
    pageable = torch.zeros(1)                                                             # not pinned
    assert lib.tv_optim_step_guarded(*common, guard.data_ptr(), 2, pageable.data_ptr(), hip.stream()) != 0
    torch.cuda.synchronize()
    assert torch.equal(p, torch.ones(64, device=device))          # none of the refused calls ran
    # the accepted form: guard 0 -> the update runs and the pinned scalar receives the guard
    pinned = torch.full([1], 7.0).pin_memory()
    assert lib.tv_optim_step_guarded(*common, guard.data_ptr(), 2, pinned.data_ptr(), hip.stream()) == 0
    torch.cuda.synchronize()
    assert float(pinned[0]) == 0.0 and not torch.equal(p, torch.ones(64, device=device))
    # a non-zero guard of kind 2, a NaN of kind 1: nothing moves
    before = p.clone()
    guard.fill_(1.0)
    assert lib.tv_optim_step_guarded(*common, guard.data_ptr(), 2, pinned.data_ptr(), hip.stream()) == 0
    guard.fill_(float('nan'))
    assert lib.tv_optim_step_guarded(*common, guard.data_ptr(), 1, pinned.data_ptr(), hip.stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(p, before) and math.isnan(float(pinned[0]))
