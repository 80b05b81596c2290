"""Host-side operators of the accelerated path: thin `torch.autograd.Function`s over the
C-ABI of `libtracevae_hip.so`.  PyTorch only provides device memory, the stream and the
autograd tape here; every operator body is a call into the HIP library (`hip.call`), and
there is no other implementation to fall back to.
"""
import math
from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import hip

__all__ = [
    'DeviceGraph', 'TIMER', 'gemm', 'linear', 'matmul_nn', 'rowdot', 'colsum', 'gat_conv', 'bn_lrelu', 'bn_lrelu_linear',
    'gat_layer_train', 'DeferredInput', 'graph_sum', 'segment_mean', 'embedding', 'latdec_input', 'latdec_rows', 'adj_bernoulli_ll',
    'categorical_ll', 'normal_reparam', 'normal_ll', 'realnvp_log_prob', 'realnvp_supported',
    'realnvp_actnorm_init',
    'latency_ll', 'score_reduce',
    'LATENCY_MODES',
]

_f32 = torch.float32


class KernelTimer(object):
    """Optional per-kernel timing with HIP events on the launch stream (bench.py turns it on
    for the timed region): `totals[name] = [launches, work, [event pairs]]`."""

    def __init__(self):
        self.enabled = False
        self.records = {}

    def reset(self):
        self.records = {}

    class _Region(object):
        def __init__(self, timer, name, work, tag):
            self.timer, self.name, self.work, self.tag = timer, name, work, tag

        def __enter__(self):
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.start.record()
            return self

        def __exit__(self, *exc):
            self.end.record()
            rec = self.timer.records.setdefault(self.name, [0, 0.0, []])
            rec[0] += 1
            rec[1] += self.work
            rec[2].append((self.start, self.end, self.tag, self.work))

    class _Null(object):
        def __enter__(self):
            return None

        def __exit__(self, *exc):
            return False

    _null = _Null()

    def region(self, name, work, tag=None):
        if not self.enabled:
            return self._null
        return self._Region(self, name, float(work), tag)

    def summary(self):
        """{name: (launches, total work, total milliseconds)} - call after a device sync."""
        out = {}
        for name, (count, work, pairs) in self.records.items():
            out[name] = (count, work, sum(p[0].elapsed_time(p[1]) for p in pairs))
        return out

    def by_tag(self, name):
        """{tag: (launches, total work, total milliseconds)} of one kernel family."""
        out = {}
        for start, end, tag, work in self.records.get(name, [0, 0.0, []])[2]:
            rec = out.setdefault(tag, [0, 0.0, 0.0])
            rec[0] += 1
            rec[1] += work
            rec[2] += start.elapsed_time(end)
        return {k: tuple(v) for k, v in out.items()}


TIMER = KernelTimer()


def _check(t: torch.Tensor, name: str, dtype=_f32) -> torch.Tensor:
    if not t.is_cuda:
        raise hip.HipLibraryError(
            f'{name} is on {t.device}: the accelerated path runs on the GPU only '
            f'(there is no CPU fallback)')
    if t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


def _rows(t: torch.Tensor, name: str) -> torch.Tensor:
    """A 2-D fp32 GPU operand whose rows may be strided (a column slice of a wider buffer)."""
    if t.dim() == 2 and t.is_cuda and t.dtype == _f32 and t.stride(1) == 1 \
            and t.stride(0) >= t.shape[1]:
        return t
    return _check(t, name)


_WORKSPACES = {}


def _workspace(nbytes: int, device) -> Optional[torch.Tensor]:
    """Scratch for ONE library call (split-K slabs, partial sums, pre-split planes): nothing in
    it outlives the call's own kernels and every launch goes to the device's current stream, so
    the calls share one grow-only buffer per device instead of allocating per call."""
    if nbytes <= 0:
        return None
    key = (device.index, torch._C._cuda_getCurrentRawStream(device.index))
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty([max((nbytes + 3) // 4, 1 << 18)], dtype=_f32, device=device)
        _WORKSPACES[key] = buf
    return buf


class DeviceGraph(object):
    """Destination-sorted CSC of a (symmetric) batched graph, as `tv_batch_build` emits it."""
    __slots__ = ['indptr', 'indices', 'eids', 'rev', 'src', 'dst', 'num_rows', 'num_edges']

    def __init__(self, indptr, indices, eids, rev, src, dst):
        self.indptr, self.indices, self.eids, self.rev = indptr, indices, eids, rev
        self.src, self.dst = src, dst
        self.num_rows = int(indptr.shape[0]) - 1
        self.num_edges = int(indices.shape[0])


# ---------------------------------------------------------------------------------------
# dense
# ---------------------------------------------------------------------------------------
class OperandMax(object):
    """The partial maxima of |operand| that the fp16 GEMM forms scale an operand by
    (`tv_gemm_f32_ex`), kept by the caller so that a tensor entering several products is scanned
    once.  One object per operand VIEW: same tensor, same rows and columns, same prologue."""
    __slots__ = ['slots', 'valid', 'version']

    def __init__(self, device):
        self.slots = torch.empty([256], dtype=_f32, device=device)
        self.valid = False
        self.version = None

    def attach(self, t: torch.Tensor) -> torch.Tensor:
        """A producer kernel has written the maxima of `t`: they travel with the tensor object
        (`operand_max(t)` finds them) for as long as nothing writes into it."""
        self.valid = True
        self.version = t._version
        t.tv_operand_max = self
        return t


def operand_max(t: torch.Tensor) -> Optional[OperandMax]:
    """The maxima a producer kernel left with this very tensor object, if it has not been written
    to since; None otherwise (the product then scans the operand itself)."""
    box = getattr(t, 'tv_operand_max', None)
    if box is not None and box.valid and box.version == t._version:
        return box
    return None


OPERAND_SCANS = {'kept': 0, 'scanned': 0}


def gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool, trans_b: bool,
         bias: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, accumulate: bool = False,
         act: Optional[tuple] = None, a_max: Optional[OperandMax] = None,
         b_max: Optional[OperandMax] = None) -> torch.Tensor:
    """C = row_scale * (op(a) @ op(b)) + bias on the matrix cores (`tv_gemm_f32_ex`).
    `act = (operand, scale, shift, slope)`: operand 1 (a, not transposed) or 2 (b, not
    transposed) goes through lrelu(v * scale[f] + shift[f]) on its way into the kernel.
    `a_max` / `b_max`: see `OperandMax`."""
    a_in, b_in = a, b
    a, b = _rows(a, 'a'), _rows(b, 'b')
    # (maxima only travel with the caller's own view, not with a contiguous copy made here)
    if a is not a_in:
        a_max = None
    if b is not b_in:
        b_max = None
    m, k = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    if not trans_a and k % 4 and a.stride(0) % 4 == 0 and a.stride(0) >= (k + 3) // 4 * 4:
        # the library reads such rows 16 bytes at a time, up to 3 floats past k (tracevae_hip.h):
        # fine for a column slice of a wider buffer, not for a view whose storage ends with the
        # last element
        end = a.storage_offset() + (m - 1) * a.stride(0) + (k + 3) // 4 * 4
        if end * 4 > a.untyped_storage().nbytes():
            a = a.contiguous()
    n, kb = (b.shape[0], b.shape[1]) if trans_b else (b.shape[1], b.shape[0])
    if k != kb:
        raise ValueError(f'gemm: inner sizes differ: {k} vs {kb}')
    c = out if out is not None else torch.empty([m, n], dtype=_f32, device=a.device)
    if out is not None and (tuple(out.shape) != (m, n) or _rows(out, 'out') is not out):
        raise ValueError(f'gemm: out must be a [{m}, {n}] fp32 GPU tensor with contiguous rows')
    if m == 0 or n == 0:
        return c
    if k == 0:
        if not accumulate:
            c.zero_()
        return c
    wsb = hip.query('tv_gemm_f32_workspace_bytes', m, n, k)
    ws = _workspace(wsb, a.device)
    operand, scale, shift, slope = act if act is not None else (0, None, None, 0.0)
    if act is not None:
        feats = k if operand == 1 else n
        if scale.numel() != feats or shift.numel() != feats:
            raise ValueError(f'gemm: act scale / shift must have {feats} entries')
        scale, shift = _check(scale, 'act scale'), _check(shift, 'act shift')
    had = (a_max is not None and a_max.valid, b_max is not None and b_max.valid)
    with TIMER.region('gemm_f32', 2.0 * m * n * k, (m, n, k, bool(trans_a), bool(trans_b))) as reg:
        hip.call('tv_gemm_f32_ex', int(trans_a), int(trans_b), m, n, k, hip.ptr_rows(a),
                 a.stride(0), hip.ptr_rows(b), b.stride(0), hip.ptr_rows(c), c.stride(0), hip.ptr(bias),
                 hip.ptr(row_scale), int(accumulate), int(operand), hip.ptr(scale), hip.ptr(shift),
                 float(slope),
                 None if a_max is None else a_max.slots.data_ptr(),
                 0 if a_max is None else (2 if a_max.valid else 1),
                 None if b_max is None else b_max.slots.data_ptr(),
                 0 if b_max is None else (2 if b_max.valid else 1),
                 hip.ptr(ws), wsb)
        if reg is not None or a_max is not None or b_max is not None:
            # which arithmetic the product took: 0 = bf16 x 3 (six terms), 1 / 2 = fp16 x 2 (three
            # terms) on the wide / k-major wide form; the bench prices each against its own peak,
            # and only the fp16 forms fill the callers' operand maxima
            form = int(hip.lib().tv_gemm_f32_last_form())
            if reg is not None:
                reg.tag = reg.tag + (form,)
            if form:
                if a_max is not None:
                    a_max.valid = True
                if b_max is not None:
                    b_max.valid = True
                # operands whose maxima were at hand / that the product had to scan
                OPERAND_SCANS['kept'] += int(had[0]) + int(had[1])
                OPERAND_SCANS['scanned'] += 2 - int(had[0]) - int(had[1])
    return c


def colsum(x: torch.Tensor) -> torch.Tensor:
    x = _rows(x, 'x')
    rows, cols = x.shape
    out = torch.empty([cols], dtype=_f32, device=x.device)
    if rows == 0:
        return out.zero_()
    wsb = hip.query('tv_colsum_workspace_bytes', rows, cols)
    ws = _workspace(wsb, x.device)
    hip.call('tv_colsum', rows, cols, hip.ptr_rows(x), x.stride(0), hip.ptr(out), hip.ptr(ws), wsb)
    return out


class _Linear(Function):
    """y = x @ W^T + b with W stored [out, in] (torch / tensorkit Linear, GATConv.fc)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = _check(x.reshape(-1, x.shape[-1]), 'x')
        weight = _check(weight, 'weight')
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        y = gemm(x2, weight, False, True, bias=bias)
        return y.reshape(x.shape[:-1] + (weight.shape[0],))

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy2 = _check(dy.reshape(-1, dy.shape[-1]), 'dy')
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm(dy2, weight, False, False).reshape(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            dw = gemm(dy2, x2, True, False)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy2)
        return dx, dw, db


def score_columns(heads: int) -> int:
    """Columns that el / er take behind ft: 2H, padded so that rows stay 16-byte aligned."""
    return (2 * heads + 3) // 4 * 4


def gat_fc_score_weight(weight, attn_l, attn_r, w_max=None):
    """[W ; A_l W ; A_r W] (`tv_gat_fc_score_weight`): the fc weight of a GAT layer with 2H more
    rows, so that el / er are 2H more columns of the fc product.  `w_max` (an `OperandMax`)
    receives the maxima of the result."""
    weight = _check(weight, 'weight')
    heads, feats = int(attn_l.shape[-2]), int(attn_l.shape[-1])
    n, k = weight.shape
    if n != heads * feats:
        raise ValueError(f'gat_fc_score_weight: weight {tuple(weight.shape)} vs heads {heads} x feats {feats}')
    out = torch.empty([n + score_columns(heads), k], dtype=_f32, device=weight.device)
    hip.call('tv_gat_fc_score_weight', heads, feats, k, hip.ptr(weight),
             hip.ptr(_check(attn_l, 'attn_l')), hip.ptr(_check(attn_r, 'attn_r')), hip.ptr(out),
             None if w_max is None else w_max.slots.data_ptr())
    if w_max is not None:
        w_max.valid = True
    return out


def _score_weight_grads(dwx, weight, attn_l, attn_r):
    """(dW, dattn_l, dattn_r) from the weight gradient of the extended fc product
    (`tv_gat_fc_score_weight_bwd`; dW is the leading rows of `dwx`, updated in place)."""
    heads, feats = int(attn_l.shape[-2]), int(attn_l.shape[-1])
    n, k = weight.shape
    dal = torch.empty([n], dtype=_f32, device=weight.device)
    dar = torch.empty([n], dtype=_f32, device=weight.device)
    hip.call('tv_gat_fc_score_weight_bwd', heads, feats, k, hip.ptr(weight),
             hip.ptr(_check(attn_l, 'attn_l')), hip.ptr(_check(attn_r, 'attn_r')), hip.ptr(dwx),
             hip.ptr(dal), hip.ptr(dar))
    return dwx[:n], dal.reshape(attn_l.shape), dar.reshape(attn_r.shape)


class _LinearScores(Function):
    """[ft | el | er] = x @ [W ; A_l W ; A_r W]^T: the fc product of a GAT layer with the attention
    scores as 2H more columns.  The gradient that comes back from `_GATConv` is
    [dft | del | der | 0] (dft without the el / er terms): the two GEMMs with the extended weight
    carry the score gradients to x and W, and the attention vectors get theirs here."""

    @staticmethod
    def forward(ctx, x, weight, attn_l, attn_r):
        x2 = _check(x, 'x')
        weight = _check(weight, 'weight')
        ctx.w_max = OperandMax(x2.device)
        wx = gat_fc_score_weight(weight, attn_l, attn_r, ctx.w_max)
        ctx.save_for_backward(x2, weight, wx, attn_l, attn_r)
        # x and the weights enter the backward pass's products again: their maxima are kept
        need = any(ctx.needs_input_grad)
        ctx.x_max = (operand_max(x) if x2 is x else None) or (OperandMax(x2.device) if need else None)
        return gemm(x2, wx, False, True, a_max=ctx.x_max, b_max=ctx.w_max)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, weight, wx, attn_l, attn_r = ctx.saved_tensors
        dy_in = dy
        dy = _check(dy, 'dy')
        dx = dw = dal = dar = None
        dy_max = (operand_max(dy_in) if dy is dy_in else None) or OperandMax(dy.device)
        if ctx.needs_input_grad[0]:
            dx = gemm(dy, wx, False, False, a_max=dy_max, b_max=ctx.w_max)
        if any(ctx.needs_input_grad[1:]):
            dw, dal, dar = _score_weight_grads(
                gemm(dy, x2, True, False, a_max=dy_max, b_max=ctx.x_max), weight, attn_l, attn_r)
        return dx, dw, dal, dar


def linear_scores(x, weight, attn_l, attn_r):
    return _LinearScores.apply(x, weight, attn_l, attn_r)


class _SkinnyLinear(Function):
    """`_Linear` with one small side, min(in, out) <= 16 (`tv_skinny_linear_*`): the z / z2
    heads (1000 -> 10) and the first layers fed by z (10 -> 500)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = _check(x.reshape(-1, x.shape[-1]), 'x')
        weight = _check(weight, 'weight')
        rows, k = x2.shape
        n = weight.shape[0]
        y = torch.empty([rows, n], dtype=_f32, device=x2.device)
        hip.call('tv_skinny_linear_fwd', rows, k, n, hip.ptr(x2), hip.ptr(weight),
                 hip.ptr(None if bias is None else _check(bias, 'bias')), hip.ptr(y))
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y.reshape(x.shape[:-1] + (n,))

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        rows, k = x2.shape
        n = weight.shape[0]
        dy2 = _check(dy.reshape(-1, n), 'dy')
        if rows == 0:
            return (torch.zeros(ctx.x_shape, dtype=_f32, device=x2.device), torch.zeros_like(weight),
                    torch.zeros([n], dtype=_f32, device=x2.device) if ctx.has_bias else None)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        dwb = torch.empty([n * k + n], dtype=_f32, device=x2.device)
        wsb = hip.query('tv_skinny_linear_bwd_workspace_bytes', rows, k, n)
        ws = _workspace(wsb, x2.device)
        hip.call('tv_skinny_linear_bwd', rows, k, n, hip.ptr(x2), hip.ptr(weight), hip.ptr(dy2),
                 hip.ptr(dx), hip.ptr(dwb), hip.ptr(ws), wsb)
        return (None if dx is None else dx.reshape(ctx.x_shape), dwb[:n * k].view(n, k),
                dwb[n * k:] if ctx.has_bias else None)


def linear(x, weight, bias=None):
    if min(weight.shape) <= 16 and max(weight.shape) >= 64:
        return _SkinnyLinear.apply(x, weight, bias)
    return _Linear.apply(x, weight, bias)


class _MatmulNN(Function):
    """y = x @ W + b with W stored [in, out] (dgl GraphConv weight layout)."""

    @staticmethod
    def forward(ctx, x_in, weight, bias):
        x, weight = _check(x_in, 'x'), _check(weight, 'weight')
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        # x enters this product and the weight gradient, the weights this product and the input
        # gradient: their maxima (fp16 GEMM forms) are kept between the products
        ctx.x_max = (operand_max(x_in) if x is x_in else None) or OperandMax(x.device)
        ctx.w_max = OperandMax(x.device)
        return gemm(x, weight, False, False, bias=bias, a_max=ctx.x_max, b_max=ctx.w_max)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy_in):
        x, weight = ctx.saved_tensors
        dy = _rows(dy_in, 'dy')       # (the categorical log-likelihood hands over padded rows)
        dy_max = (operand_max(dy_in) if dy is dy_in else None) or OperandMax(dy.device)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm(dy, weight, False, True, a_max=dy_max, b_max=ctx.w_max)
        if ctx.needs_input_grad[1]:
            dw = gemm(x, dy, True, False, a_max=ctx.x_max, b_max=dy_max)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy)
        return dx, dw, db


def matmul_nn(x, weight, bias=None):
    return _MatmulNN.apply(x, weight, bias)


class _RowDot(Function):
    """out[r, j] = sum_k x[r,k] w[k,j] for a skinny w (at most 8 columns)."""

    @staticmethod
    def forward(ctx, x, w):
        x, w = _check(x, 'x'), _check(w, 'w')
        rows, k = x.shape
        n = w.shape[1]
        out = torch.empty([rows, n], dtype=_f32, device=x.device)
        hip.call('tv_rowdot_fwd', rows, k, n, hip.ptr(x), hip.ptr(w), None, hip.ptr(out))
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        dout = _check(dout, 'dout')
        rows, k = x.shape
        n = w.shape[1]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        wsb = hip.query('tv_rowdot_bwd_workspace_bytes', rows, k, n)
        ws = _workspace(wsb, x.device)
        hip.call('tv_rowdot_bwd', rows, k, n, hip.ptr(x), hip.ptr(w), None, hip.ptr(dout),
                 hip.ptr(dx), hip.ptr(dw), hip.ptr(ws), wsb)
        return dx, dw


def rowdot(x, w):
    return _RowDot.apply(x, w)


# ---------------------------------------------------------------------------------------
# GAT message passing
# ---------------------------------------------------------------------------------------
class _GATConv(Function):
    @staticmethod
    def forward(ctx, ft, attn_l, attn_r, bias, graph: DeviceGraph, n_rep: int, want_stats: bool,
                ep_scale, ep_shift, ep_slope: float, need_grad: bool, stats_row_weight=None):
        ft = _check(ft, 'ft')
        heads, feats = int(attn_l.shape[-2]), int(attn_l.shape[-1])
        rows = ft.shape[0]
        hf = heads * feats
        if rows != graph.num_rows * n_rep or ft.shape[1] not in (hf, hf + score_columns(heads)):
            raise ValueError(f'gat_conv: ft {tuple(ft.shape)} does not match graph rows '
                             f'{graph.num_rows} x {n_rep} / heads {heads} x feats {feats}')
        al, ar = _check(attn_l, 'attn_l').reshape(-1), _check(attn_r, 'attn_r').reshape(-1)
        bias = _check(bias, 'bias')
        dev = ft.device
        if ft.shape[1] == hf:
            el = torch.empty([rows, heads], dtype=_f32, device=dev)
            er = torch.empty([rows, heads], dtype=_f32, device=dev)
            hip.call('tv_gat_scores', rows, heads, feats, hip.ptr(ft), hip.ptr(al), hip.ptr(ar),
                     hip.ptr(el), hip.ptr(er))
        else:
            # [ft | el | er] from `linear_scores`: read in place
            el, er = ft[:, hf:hf + heads], ft[:, hf + heads:hf + 2 * heads]
        if need_grad and n_rep != 1:
            raise ValueError('gat_conv: gradients are only available for a single graph replica')
        out = torch.empty([rows, heads * feats], dtype=_f32, device=dev)
        alpha = torch.empty([graph.num_edges, heads], dtype=_f32, device=dev) if need_grad else None
        stats = None
        if want_stats:
            stats = torch.empty([hip.query('tv_stat_blocks', rows), 2, heads * feats],
                                dtype=_f32, device=dev)
        # algorithmic bytes: ft gathered once per in-edge + out written once (+ scores/indices)
        work = 4.0 * hf * (graph.num_edges + graph.num_rows) * n_rep \
            + (8.0 * heads * (graph.num_edges + graph.num_rows) + 8.0 * graph.num_edges) * n_rep
        # compulsory bytes (the timer's tag): every [ft | el | er] row in once, every output row
        # out once, the CSC arrays once, the attention coefficients out once when they are kept
        compulsory = 4.0 * (ft.stride(0) + hf) * rows + 8.0 * (graph.num_edges + graph.num_rows) \
            + (4.0 * heads * graph.num_edges if need_grad else 0.0)
        # with the eval epilogue `out` is the next layer's operand as it stands: its maxima come
        # out of this pass (`gat_conv` attaches them to the tensor it returns)
        ctx.out_max = OperandMax(dev) if ep_scale is not None else None
        with TIMER.region('gat_aggregate_fwd', work, compulsory):
            hip.call('tv_gat_aggregate_fwd', graph.num_rows, n_rep, heads, feats,
                     hip.ptr(graph.indptr), hip.ptr(graph.indices), hip.ptr(ft), hip.ptr_rows(el),
                     hip.ptr_rows(er), hip.ptr(bias), hip.ptr(out), hip.ptr(alpha), hip.ptr(stats),
                     hip.ptr(stats_row_weight if want_stats else None),
                     hip.ptr(ep_scale), hip.ptr(ep_shift), float(ep_slope),
                     ft.stride(0), el.stride(0),
                     None if ctx.out_max is None else ctx.out_max.slots.data_ptr())
        if ctx.out_max is not None:
            _LAST_OUT_MAX['box'] = ctx.out_max      # handed to `gat_conv`, which owns the tensor
        if need_grad:
            ctx.save_for_backward(ft, el, er, alpha, al, ar)
            ctx.graph = graph
            ctx.hf = (heads, feats)
            ctx.attn_shape = attn_l.shape
        if stats is None:
            stats = torch.empty([0], dtype=_f32, device=dev)
        ctx.mark_non_differentiable(stats)
        # no zero-filled gradient tensor for the statistics output on the way back
        ctx.set_materialize_grads(False)
        return out, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, _dstats):
        if dout is None:
            return (None,) * 12
        ft, el, er, alpha, al, ar = ctx.saved_tensors
        g = ctx.graph
        heads, feats = ctx.hf
        dout = _check(dout, 'dout')
        dev = ft.device
        dft = torch.empty_like(ft)
        # [ft | el | er] in, [dft | del | der | 0] out: the attention vectors get their gradients in
        # the fc layer (`_LinearScores`, `_BNLReLULinear`)
        cols = ft.shape[1] != heads * feats
        dal = dar = None
        if not cols:
            dal = torch.empty([heads * feats], dtype=_f32, device=dev)
            dar = torch.empty([heads * feats], dtype=_f32, device=dev)
        dbias = torch.empty([heads * feats], dtype=_f32, device=dev)
        wsb = hip.query('tv_gat_aggregate_bwd_workspace_bytes', g.num_rows, 1, heads, feats,
                        g.num_edges)
        ws = _workspace(wsb, dev)
        # algorithmic bytes: edge pass (dout row + ft per in-edge) + node pass (dout per
        # in-edge, ft row read, dft written)
        hf = heads * feats
        work = 4.0 * hf * (2 * g.num_edges + 3 * g.num_rows)
        # compulsory bytes: dout and ft rows in once, dft rows out once, coefficients and CSC / rev
        compulsory = 4.0 * (hf + 2 * ft.stride(0)) * g.num_rows + 4.0 * heads * g.num_edges \
            + 8.0 * (2 * g.num_edges + g.num_rows)
        dft_max = OperandMax(dev) if cols else None
        with TIMER.region('gat_aggregate_bwd', work, compulsory):
            hip.call('tv_gat_aggregate_bwd', g.num_rows, 1, heads, feats, g.num_edges,
                     hip.ptr(g.indptr), hip.ptr(g.indices), hip.ptr(g.rev), hip.ptr(ft),
                     hip.ptr_rows(el), hip.ptr_rows(er), hip.ptr(al), hip.ptr(ar), hip.ptr(alpha),
                     hip.ptr(dout), hip.ptr(dft), hip.ptr(dal), hip.ptr(dar), hip.ptr(dbias),
                     ft.stride(0), el.stride(0), dft.stride(0), int(cols),
                     None if dft_max is None else dft_max.slots.data_ptr(), hip.ptr(ws), wsb)
        if dft_max is not None:
            dft_max.attach(dft)       # the fc layer's two products find it on the gradient tensor
        if not cols:
            dal, dar = dal.reshape(ctx.attn_shape), dar.reshape(ctx.attn_shape)
        return (dft, dal, dar, dbias, None, None, None, None, None, None, None, None)


def gat_conv(ft, attn_l, attn_r, bias, graph: DeviceGraph, n_rep: int = 1,
             want_stats: bool = False, ep_scale=None, ep_shift=None, ep_slope: float = 0.01,
             stats_row_weight=None):
    """GATConv message passing after the fc projection.  Returns `(out, stats_partial)`;
    `stats_partial` is an empty tensor unless `want_stats` (rows counted `stats_row_weight`
    times when given)."""
    # (grad mode is always off inside Function.forward, so decide here whether the
    # attention coefficients must be kept for a backward pass)
    need_grad = torch.is_grad_enabled() and any(
        t.requires_grad for t in (ft, attn_l, attn_r, bias))
    out, stats = _GATConv.apply(ft, attn_l, attn_r, bias, graph, n_rep, want_stats, ep_scale,
                                ep_shift, ep_slope, need_grad, stats_row_weight)
    box = _LAST_OUT_MAX.pop('box', None)
    if box is not None:
        box.attach(out)
    return out, stats


_LAST_OUT_MAX = {}


class DeferredInput(object):
    """The BatchNorm(batch statistics) + LeakyReLU that a layer's input still has to go through
    (`gat_layer_train`): statistics partials of x, the BatchNorm's parameters and buffers."""
    __slots__ = ['stats', 'gamma', 'beta', 'running_mean', 'running_var', 'eps', 'momentum',
                 'row_weight', 'count']

    def __init__(self, stats, gamma, beta, running_mean, running_var, eps, momentum, row_weight,
                 count):
        self.stats, self.gamma, self.beta = stats, gamma, beta
        self.running_mean, self.running_var = running_mean, running_var
        self.eps, self.momentum, self.row_weight, self.count = eps, momentum, row_weight, count


class PreparedGATWeights(object):
    """What `prepare_gat_layers` left for one layer: the plane images of its extended weight for
    the fc product (`fwd`) and for the input gradient (`bwd`), their forms (tv_gemm_f32_ex states
    3 / 4) and the maxima the fp16 form scaled them by.  Valid for products of `rows` rows until
    the weights change."""
    __slots__ = ['rows', 'fwd', 'fwd_state', 'bwd', 'bwd_state', 'w_max']


_PREPARE_CACHE: dict = {}


def prepare_gat_layers(layers, rows: int):
    """The extended weights `[W ; A_l W ; A_r W ; 0]` of a stack of GAT layers and their plane
    images in both orientations, in TWO launches for the whole stack
    (`tv_gat_fc_score_weight_batch`, `tv_gemm_prepare_b_batch`) instead of three per layer.
    `layers`: [(weight, attn_l, attn_r)].  Returns a list with a `PreparedGATWeights` per layer, or
    None where a layer's products do not take a pre-split B (the layer call then builds its
    weights itself).  The buffers and the argument arrays of a stack are made once and written
    anew by every call (a stack's weights live at fixed addresses in the flat parameter buffer);
    what a call returns is valid until the next call for the same stack."""
    import ctypes
    ent = _PREPARE_CACHE.get((id(layers), rows))
    if (ent is not None and ent['layers'] is layers
            and ent['ptrs'] == [t.data_ptr() for l in layers for t in l]):
        return _run_prepared(ent, rows)
    if len(_PREPARE_CACHE) >= 4096:
        # entries of models that are gone, or of thousands of row counts: start over (what a
        # running step still needs it holds itself, through the objects it was handed)
        _PREPARE_CACHE.clear()
    shapes = []
    for weight, attn_l, attn_r in layers:
        hf, k = weight.shape
        heads = int(attn_l.shape[-2])
        width = hf + score_columns(heads)
        shapes.append((hf, k, heads, int(attn_l.shape[-1]), width,
                       hip.query('tv_gemm_b_form', rows, width, k),
                       hip.query('tv_gemm_b_form', rows, k, width)))
    key = (tuple((w.data_ptr(), a.data_ptr(), b.data_ptr()) for w, a, b in layers), tuple(shapes))
    ent = _PREPARE_CACHE.get(key)
    if ent is None:
        todo = []
        for i, ((weight, attn_l, attn_r), (hf, k, heads, feats, width, f_fwd, f_bwd)) in enumerate(
                zip(layers, shapes)):
            if (f_fwd < 0 or f_bwd < 0 or k % 4 or (k + 15) // 16 * heads > 256
                    or not (weight.is_contiguous() and attn_l.is_contiguous() and attn_r.is_contiguous())):
                continue
            if len(todo) == 16:         # what one tv_gat_fc_score_weight_batch launch takes; the
                break                   # layers beyond it prepare their own weights
            todo.append((i, weight, attn_l, attn_r, heads, feats, k, width, f_fwd, f_bwd))
        ent = {'n': len(todo), 'out': [None] * len(layers)}
        if todo:
            dev = todo[0][1].device
            n = len(todo)
            wx = [torch.empty([t[7], t[6]], dtype=_f32, device=dev) for t in todo]
            boxes = [OperandMax(dev) for _ in todo]
            img_f = [torch.empty([hip.query('tv_gemm_prepared_b_bytes', rows, t[7], t[6])],
                                 dtype=torch.uint8, device=dev) for t in todo]
            img_b = [torch.empty([hip.query('tv_gemm_prepared_b_bytes', rows, t[6], t[7])],
                                 dtype=torch.uint8, device=dev) for t in todo]
            i64, ptrs = ctypes.c_int64 * n, ctypes.c_void_p * n
            ent['score'] = (n, i64(*[t[4] for t in todo]), i64(*[t[5] for t in todo]),
                            i64(*[t[6] for t in todo]), ptrs(*[t[1].data_ptr() for t in todo]),
                            ptrs(*[t[2].data_ptr() for t in todo]), ptrs(*[t[3].data_ptr() for t in todo]),
                            ptrs(*[w.data_ptr() for w in wx]), ptrs(*[b.slots.data_ptr() for b in boxes]))
            # two images per layer: [n][k] for x . W_ext^T, [k][n] for dft . W_ext
            m2 = 2 * n
            i64b, ptrb, i32b = ctypes.c_int64 * m2, ctypes.c_void_p * m2, ctypes.c_int * m2
            ns, ks, tb, bs, lds, mx, pl = [], [], [], [], [], [], []
            for j, t in enumerate(todo):
                k, width = t[6], t[7]
                for (nn, kk, tt, img) in ((width, k, 1, img_f[j]), (k, width, 0, img_b[j])):
                    ns.append(nn); ks.append(kk); tb.append(tt)
                    bs.append(wx[j].data_ptr()); lds.append(k); mx.append(boxes[j].slots.data_ptr())
                    pl.append(img.data_ptr())
            ent['rows'] = i64b(*([rows] * m2))
            ent['split'] = (i64b(*ns), i64b(*ks), i32b(*tb), ptrb(*bs), i64b(*lds), ptrb(*mx), ptrb(*pl))
            ent['keep'] = (wx, boxes, img_f, img_b, [t[1:4] for t in todo])
            for j, t in enumerate(todo):
                p = PreparedGATWeights()
                p.fwd, p.fwd_state = img_f[j], 3 + t[8]
                p.bwd, p.bwd_state = img_b[j], 3 + t[9]
                boxes[j].valid = True
                p.w_max = boxes[j]
                ent['out'][t[0]] = p
        _PREPARE_CACHE[key] = ent
    # a caller that keeps its list of layers alive finds the entry by the list and the row count
    # alone the next time (the forms of the products depend on the row count)
    _PREPARE_CACHE[(id(layers), rows)] = dict(ent, layers=layers,
                                              ptrs=[t.data_ptr() for l in layers for t in l])
    return _run_prepared(ent, rows)


def _run_prepared(ent, rows: int):
    if ent['n']:
        m2 = 2 * ent['n']
        arr = ent['rows']
        if arr[0] != rows:
            for q in range(m2):
                arr[q] = rows
        hip.call('tv_gat_fc_score_weight_batch', *ent['score'])
        hip.call('tv_gemm_prepare_b_batch', m2, arr, *ent['split'])
        for p in ent['out']:
            if p is not None:
                p.rows = rows
    return ent['out']


class _GATLayerTrain(Function):
    """One GATConvAgg layer of a training step as ONE library call each way
    (`tv_gat_layer_train_fwd / _bwd`): the launches of `_BNLReLULinear` (or `_LinearScores`) and
    `_GATConv` with column statistics, issued from inside the library instead of one by one from
    here.  Same kernels, same order, same arguments: bit-equal to the operator-level path; what it
    saves is the host's time per launch, which is the bound of a step at small batches."""

    @staticmethod
    def forward(ctx, x, in_gamma, in_beta, weight, attn_l, attn_r, bias, graph, deferred, slope,
                stats_row_weight, prepared=None):
        x_in = x
        x = _check(x, 'x')
        weight, bias = _check(weight, 'weight'), _check(bias, 'bias')
        heads, feats = int(attn_l.shape[-2]), int(attn_l.shape[-1])
        al, ar = _check(attn_l, 'attn_l').reshape(-1), _check(attn_r, 'attn_r').reshape(-1)
        rows, k = x.shape
        hf = heads * feats
        width = hf + score_columns(heads)
        dev = x.device
        if rows != graph.num_rows or tuple(weight.shape) != (hf, k):
            raise ValueError(f'gat_layer_train: x {tuple(x.shape)} / weight {tuple(weight.shape)} do '
                             f'not match graph rows {graph.num_rows}, heads {heads} x feats {feats}')
        count = 0
        coeffs = None
        if deferred is not None:
            count = rows if deferred.row_weight is None else int(deferred.count)
            if count < 2:
                raise ValueError('Expected more than 1 value per channel when training')
            coeffs = torch.empty([4, (k + 3) // 4 * 4], dtype=_f32, device=dev)
            in_gamma, in_beta = _check(in_gamma, 'gamma'), _check(in_beta, 'beta')
        if prepared is not None and (prepared.rows != rows or x.stride(0) % 4 or x.data_ptr() % 16):
            prepared = None             # (made for other products than this one)
        wx = prepared.fwd if prepared is not None else torch.empty([width, k], dtype=_f32, device=dev)
        ft = torch.empty([rows, width], dtype=_f32, device=dev)
        out = torch.empty([rows, hf], dtype=_f32, device=dev)
        alpha = torch.empty([graph.num_edges, heads], dtype=_f32, device=dev)
        stats = torch.empty([hip.query('tv_stat_blocks', rows), 2, hf], dtype=_f32, device=dev)
        x_max = (operand_max(x_in) if (x is x_in and deferred is None) else None) or OperandMax(dev)
        w_max = prepared.w_max if prepared is not None else OperandMax(dev)
        wsb = hip.query('tv_gat_layer_workspace_bytes', rows, k, heads, feats, graph.num_edges)
        ws = _workspace(wsb, dev)
        d = deferred
        hip.call('tv_gat_layer_train_fwd', rows, k, heads, feats, hip.ptr(graph.indptr),
                 hip.ptr(graph.indices), hip.ptr(x), x.stride(0),
                 hip.ptr(d.stats) if d else None, d.stats.shape[0] if d else 0, count,
                 float(d.eps) if d else 0.0, float(d.momentum) if d else 0.0, float(slope),
                 hip.ptr(in_gamma) if d else None, hip.ptr(in_beta) if d else None,
                 hip.ptr(d.running_mean) if d else None, hip.ptr(d.running_var) if d else None,
                 hip.ptr(coeffs), hip.ptr(weight), hip.ptr(al), hip.ptr(ar), hip.ptr(bias),
                 hip.ptr(stats_row_weight), hip.ptr(wx), hip.ptr(ft), hip.ptr(out), hip.ptr(alpha),
                 hip.ptr(stats), x_max.slots.data_ptr(), 2 if x_max.valid else 1,
                 w_max.slots.data_ptr(), prepared.fwd_state if prepared is not None else 0,
                 hip.ptr(ws), wsb)
        form = int(hip.lib().tv_gemm_f32_last_form())
        x_max.valid = x_max.valid or form != 0 or (d is not None and k <= 2048)
        w_max.valid = True          # written with weight_ext
        # (backward: the image for the input gradient when the weights were prepared)
        ctx.prepared = prepared
        ctx.save_for_backward(x, wx if prepared is None else prepared.bwd, ft, alpha, weight, al, ar,
                              *((coeffs, in_gamma, in_beta) if d else ()))
        ctx.graph, ctx.dims = graph, (rows, k, heads, feats)
        ctx.deferred = d is not None
        ctx.row_weight = d.row_weight if d else None
        ctx.count, ctx.slope = count, float(slope)
        ctx.x_max, ctx.w_max = x_max, w_max
        ctx.attn_shape = attn_l.shape
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return out, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, _dstats):
        if dout is None:
            return (None,) * 12
        x, wx, ft, alpha, weight, al, ar = ctx.saved_tensors[:7]
        coeffs = in_gamma = in_beta = None
        if ctx.deferred:
            coeffs, in_gamma, in_beta = ctx.saved_tensors[7:]
        rows, k, heads, feats = ctx.dims
        g = ctx.graph
        hf = heads * feats
        width = hf + score_columns(heads)
        dev = x.device
        dout = _check(dout, 'dout')
        need_dx = any(ctx.needs_input_grad[:3])
        need_dw = any(ctx.needs_input_grad[3:6])
        dwx = torch.empty([width, k], dtype=_f32, device=dev) if need_dw else None
        dal = torch.empty([hf], dtype=_f32, device=dev) if need_dw else None
        dar = torch.empty([hf], dtype=_f32, device=dev) if need_dw else None
        dbias = torch.empty([hf], dtype=_f32, device=dev)
        dx = torch.empty_like(x) if need_dx else None
        dgamma = torch.empty_like(in_gamma) if (ctx.deferred and need_dx) else None
        dbeta = torch.empty_like(in_beta) if (ctx.deferred and need_dx) else None
        wsb = hip.query('tv_gat_layer_workspace_bytes', rows, k, heads, feats, g.num_edges)
        ws = _workspace(wsb, dev)
        hip.call('tv_gat_layer_train_bwd', rows, k, heads, feats, g.num_edges, hip.ptr(g.indptr),
                 hip.ptr(g.indices), hip.ptr(g.rev), hip.ptr(x), x.stride(0), hip.ptr(coeffs),
                 hip.ptr(in_gamma), hip.ptr(in_beta), ctx.slope, hip.ptr(ctx.row_weight), ctx.count,
                 hip.ptr(weight), hip.ptr(al), hip.ptr(ar), hip.ptr(wx), hip.ptr(ft), hip.ptr(alpha),
                 hip.ptr(dout), ctx.x_max.slots.data_ptr(), 2 if ctx.x_max.valid else 1,
                 ctx.w_max.slots.data_ptr(),
                 ctx.prepared.bwd_state if ctx.prepared is not None else (2 if ctx.w_max.valid else 1),
                 int(need_dx), int(need_dw),
                 hip.ptr(dwx), hip.ptr(dal), hip.ptr(dar), hip.ptr(dbias), hip.ptr(dx),
                 hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(ws), wsb)
        dw = dwx[:hf] if need_dw else None
        if need_dw:
            dal, dar = dal.reshape(ctx.attn_shape), dar.reshape(ctx.attn_shape)
        return dx, dgamma, dbeta, dw, dal, dar, dbias, None, None, None, None, None


def gat_layer_train(x, weight, attn_l, attn_r, bias, graph: DeviceGraph,
                    deferred: Optional[DeferredInput] = None, slope: float = 0.01,
                    stats_row_weight=None, prepared: Optional[PreparedGATWeights] = None):
    """`(out, stats_partial)` of one GAT layer in training mode: fc (+ the deferred BatchNorm +
    LeakyReLU of its input) -> attention scores -> aggregation with column statistics of `out`."""
    return _GATLayerTrain.apply(x, deferred.gamma if deferred else None,
                                deferred.beta if deferred else None, weight, attn_l, attn_r, bias,
                                graph, deferred, slope, stats_row_weight, prepared)


# ---------------------------------------------------------------------------------------
# BatchNorm + LeakyReLU
# ---------------------------------------------------------------------------------------
class _BNLReLUTrain(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, slope, stats,
                row_weight=None, count=None):
        x = _check(x, 'x')
        rows, cols = x.shape
        dev = x.device
        count = rows if row_weight is None else int(count)
        if count < 2:
            raise ValueError('Expected more than 1 value per channel when training')
        if stats is None or stats.numel() == 0:
            stats = torch.empty([hip.query('tv_stat_blocks', rows), 2, cols], dtype=_f32, device=dev)
            hip.call('tv_colstats', rows, cols, hip.ptr(x), hip.ptr(row_weight), hip.ptr(stats))
        nblk = stats.shape[0]
        mean = torch.empty([cols], dtype=_f32, device=dev)
        rstd = torch.empty([cols], dtype=_f32, device=dev)
        hip.call('tv_bn_finalize', cols, nblk, hip.ptr(stats), count, float(eps), float(momentum),
                 hip.ptr(mean), hip.ptr(rstd), hip.ptr(running_mean), hip.ptr(running_var))
        y = torch.empty_like(x)
        gamma, beta = _check(gamma, 'gamma'), _check(beta, 'beta')
        hip.call('tv_bn_apply_lrelu', rows, cols, hip.ptr(x), hip.ptr(mean), hip.ptr(rstd),
                 hip.ptr(gamma), hip.ptr(beta), float(slope), hip.ptr(y))
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.slope = float(slope)
        ctx.row_weight, ctx.count = row_weight, count
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        dy = _check(dy, 'dy')
        rows, cols = x.shape
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        wsb = hip.query('tv_bn_lrelu_bwd_workspace_bytes', rows, cols)
        ws = _workspace(wsb, x.device)
        hip.call('tv_bn_lrelu_bwd', rows, cols, hip.ptr(x), hip.ptr(dy), hip.ptr(mean),
                 hip.ptr(rstd), hip.ptr(gamma), hip.ptr(beta), ctx.slope, hip.ptr(ctx.row_weight),
                 ctx.count, hip.ptr(dx), hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(ws), wsb)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


class _BNLReLULinear(Function):
    """ft = LeakyReLU(BatchNorm_train(x)) @ W^T without writing the activation: the normalised,
    activated rows only exist inside the two GEMMs that consume them (forward product and
    weight gradient, `tv_gemm_f32_act`).  Same arithmetic as `_BNLReLUTrain` + `_Linear` with
    the affine folded (y = x*scale + shift)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight, running_mean, running_var, eps, momentum, slope,
                stats, row_weight=None, count=None, attn_l=None, attn_r=None):
        x, weight = _check(x, 'x'), _check(weight, 'weight')
        gamma, beta = _check(gamma, 'gamma'), _check(beta, 'beta')
        rows, cols = x.shape
        dev = x.device
        count = rows if row_weight is None else int(count)
        if count < 2:
            raise ValueError('Expected more than 1 value per channel when training')
        if stats is None or stats.numel() == 0:
            stats = torch.empty([hip.query('tv_stat_blocks', rows), 2, cols], dtype=_f32, device=dev)
            hip.call('tv_colstats', rows, cols, hip.ptr(x), hip.ptr(row_weight), hip.ptr(stats))
        # mean, rstd, scale, shift; rows padded so that each starts 16-byte aligned
        coeffs = torch.empty([4, (cols + 3) // 4 * 4], dtype=_f32, device=dev)
        mean, rstd, scale, shift = (coeffs[i, :cols] for i in range(4))
        # (the finalize kernel also leaves a bound on |lrelu(x * scale + shift)| for the products
        # that apply these coefficients to x: no pass over x for its maximum, forward or backward)
        ctx.x_max = OperandMax(dev)
        bound_ok = cols <= 2048
        hip.call('tv_bn_finalize_coeffs', cols, stats.shape[0], hip.ptr(stats), count, float(eps),
                 float(momentum), hip.ptr(mean), hip.ptr(rstd), hip.ptr(running_mean),
                 hip.ptr(running_var), hip.ptr(gamma), hip.ptr(beta), hip.ptr(scale),
                 hip.ptr(shift), ctx.x_max.slots.data_ptr() if bound_ok else None)
        ctx.x_max.valid = bound_ok
        # with attention vectors: [ft | el | er], as `_LinearScores`
        ctx.scores = attn_l is not None
        need = any(ctx.needs_input_grad)
        ctx.w_max = OperandMax(dev) if (need or ctx.scores) else None
        w_fwd = gat_fc_score_weight(weight, attn_l, attn_r, ctx.w_max) if ctx.scores else weight
        ft = gemm(x, w_fwd, False, True, act=(1, scale, shift, slope), a_max=ctx.x_max,
                  b_max=ctx.w_max)
        if ctx.scores:
            ctx.save_for_backward(x, coeffs, gamma, beta, weight, w_fwd, attn_l, attn_r)
        else:
            ctx.save_for_backward(x, coeffs, gamma, beta, weight)
        ctx.slope = float(slope)
        ctx.row_weight, ctx.count = row_weight, count
        return ft

    @staticmethod
    @once_differentiable
    def backward(ctx, dft):
        x, coeffs, gamma, beta, weight = ctx.saved_tensors[:5]
        w_fwd = weight
        if ctx.scores:
            w_fwd, attn_l, attn_r = ctx.saved_tensors[5:]
        rows, cols = x.shape
        mean, rstd, scale, shift = (coeffs[i, :cols] for i in range(4))
        dft_in = dft
        dft = _check(dft, 'dft')           # with scores: [dft | del | der | 0]
        dx = dgamma = dbeta = dw = dal = dar = None
        dft_max = (operand_max(dft_in) if dft is dft_in else None) or OperandMax(dft.device)
        if ctx.needs_input_grad[3] or (ctx.scores and any(ctx.needs_input_grad[12:14])):
            dw = gemm(dft, x, True, False, act=(2, scale, shift, ctx.slope), a_max=dft_max,
                      b_max=ctx.x_max)
            if ctx.scores:
                dw, dal, dar = _score_weight_grads(dw, weight, attn_l, attn_r)
        if any(ctx.needs_input_grad[:3]):
            dy = gemm(dft, w_fwd, False, False, a_max=dft_max, b_max=ctx.w_max)
            dx = torch.empty_like(x)
            dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
            wsb = hip.query('tv_bn_lrelu_bwd_workspace_bytes', rows, cols)
            ws = _workspace(wsb, x.device)
            hip.call('tv_bn_lrelu_bwd', rows, cols, hip.ptr(x), hip.ptr(dy), hip.ptr(mean),
                     hip.ptr(rstd), hip.ptr(gamma), hip.ptr(beta), ctx.slope,
                     hip.ptr(ctx.row_weight), ctx.count, hip.ptr(dx), hip.ptr(dgamma),
                     hip.ptr(dbeta), hip.ptr(ws), wsb)
        return (dx, dgamma, dbeta, dw) + (None,) * 8 + (dal, dar)


def bn_lrelu_linear(x, gamma, beta, weight, running_mean, running_var, eps: float,
                    momentum: float = 0.1, slope: float = 0.01, stats=None, row_weight=None,
                    count=None, attn_l=None, attn_r=None):
    """`linear(bn_lrelu(x, training=True), weight)` as one operator (see `_BNLReLULinear`); with
    `attn_l` / `attn_r` the result is `[ft | el | er]` (`linear_scores`)."""
    return _BNLReLULinear.apply(x, gamma, beta, weight, running_mean, running_var, eps, momentum,
                                slope, stats, row_weight, count, attn_l, attn_r)


class _AffineLReLU(Function):
    @staticmethod
    def forward(ctx, x, scale, shift, slope):
        x = _check(x, 'x')
        rows, cols = x.shape
        y = torch.empty_like(x)
        hip.call('tv_affine_lrelu', rows, cols, hip.ptr(x), hip.ptr(scale), hip.ptr(shift),
                 float(slope), hip.ptr(y))
        ctx.save_for_backward(x, scale, shift)
        ctx.slope = float(slope)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, scale, shift = ctx.saved_tensors
        dy = _check(dy, 'dy')
        dx = torch.empty_like(x)
        hip.call('tv_affine_lrelu_bwd', x.shape[0], x.shape[1], hip.ptr(x), hip.ptr(dy),
                 hip.ptr(scale), hip.ptr(shift), ctx.slope, hip.ptr(dx))
        return dx, None, None, None


def bn_eval_coeffs(running_mean, running_var, gamma, beta, eps: float):
    cols = running_mean.shape[0]
    scale = torch.empty_like(running_mean)
    shift = torch.empty_like(running_mean)
    hip.call('tv_bn_eval_coeffs', cols, hip.ptr(running_mean), hip.ptr(running_var),
             hip.ptr(_check(gamma, 'gamma')), hip.ptr(_check(beta, 'beta')), float(eps),
             hip.ptr(scale), hip.ptr(shift))
    return scale, shift


def bn_lrelu(x, gamma, beta, running_mean, running_var, training: bool, eps: float,
             momentum: float = 0.1, slope: float = 0.01, stats=None, row_weight=None, count=None):
    """BatchNorm over rows followed by LeakyReLU.  Training mode uses (and updates) batch
    statistics; eval mode folds the running statistics into one affine pass.  With
    `row_weight` [rows] a row stands for that many identical rows of the batch (`count` =
    their total): statistics and the backward pass are those of the expanded batch."""
    if training:
        return _BNLReLUTrain.apply(x, gamma, beta, running_mean, running_var, eps, momentum,
                                   slope, stats, row_weight, count)
    with torch.no_grad():
        scale, shift = bn_eval_coeffs(running_mean, running_var, gamma, beta, eps)
    return _AffineLReLU.apply(x, scale, shift, slope)


# ---------------------------------------------------------------------------------------
# GraphConv aggregation, readout, embedding
# ---------------------------------------------------------------------------------------
class _GraphSum(Function):
    @staticmethod
    def forward(ctx, x, graph: DeviceGraph, n_rep: int, scale_src: bool, scale_dst: bool):
        x = _check(x, 'x')
        rows, cols = x.shape
        if rows != graph.num_rows * n_rep:
            raise ValueError(f'graph_sum: {rows} rows do not match graph {graph.num_rows} x {n_rep}')
        out = torch.empty_like(x)
        # wide rows are the operand of the GraphConv product that follows: their maxima come out
        # of this pass (`graph_sum` attaches them to the tensor it returns)
        box = OperandMax(x.device) if (cols % 4 == 0 and cols >= 64) else None
        hip.call('tv_graph_sum', graph.num_rows, n_rep, cols, hip.ptr(graph.indptr),
                 hip.ptr(graph.indices), hip.ptr(x), int(scale_src), int(scale_dst), hip.ptr(out),
                 None if box is None else box.slots.data_ptr(), None, None, 0.0)
        if box is not None:
            _LAST_OUT_MAX['graph_sum'] = box
        ctx.graph, ctx.n_rep, ctx.flags = graph, n_rep, (scale_src, scale_dst)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        # symmetric graph: the transpose aggregation is the same kernel, scalings swapped
        dout = _check(dout, 'dout')
        g = ctx.graph
        dx = torch.empty_like(dout)
        hip.call('tv_graph_sum', g.num_rows, ctx.n_rep, dout.shape[1], hip.ptr(g.indptr),
                 hip.ptr(g.indices), hip.ptr(dout), int(ctx.flags[1]), int(ctx.flags[0]),
                 hip.ptr(dx), None, None, None, 0.0)
        return dx, None, None, None, None


def graph_sum(x, graph: DeviceGraph, n_rep: int = 1, scale_src: bool = True, scale_dst: bool = True):
    out = _GraphSum.apply(x, graph, n_rep, scale_src, scale_dst)
    box = _LAST_OUT_MAX.pop('graph_sum', None)
    if box is not None:
        box.attach(out)
    return out


class _SegmentMean(Function):
    @staticmethod
    def forward(ctx, x, node_offsets, graph_id):
        x = _check(x, 'x')
        batch = node_offsets.shape[0] - 1
        out = torch.empty([batch, x.shape[1]], dtype=_f32, device=x.device)
        hip.call('tv_segment_mean_fwd', batch, x.shape[1], hip.ptr(node_offsets), hip.ptr(x),
                 hip.ptr(out), None, None, 0.0)
        ctx.save_for_backward(node_offsets, graph_id)
        ctx.rows = x.shape[0]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        node_offsets, graph_id = ctx.saved_tensors
        dout = _check(dout, 'dout')
        dx = torch.empty([ctx.rows, dout.shape[1]], dtype=_f32, device=dout.device)
        hip.call('tv_segment_mean_bwd', ctx.rows, dout.shape[1], hip.ptr(node_offsets),
                 hip.ptr(graph_id), hip.ptr(dout), hip.ptr(dx))
        return dx, None, None


def segment_mean(x, node_offsets, graph_id):
    return _SegmentMean.apply(x, node_offsets, graph_id)


class _BNLReLUTail(Function):
    """`graph_sum(LeakyReLU(BatchNorm_train(x)))` or `segment_mean(...)` without writing the
    activation: the last GAT layer of a stack hands its pre-normalisation output to the stack's
    tail (GraphConv aggregation / AvgPooling), whose kernel applies the folded BatchNorm +
    LeakyReLU to the rows as it reads them.  Same function as `_BNLReLUTrain` followed by
    `_GraphSum` / `_SegmentMean`; the backward pass is the tail's own followed by
    `tv_bn_lrelu_bwd` on the gradient of the (never materialised) activation."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, slope, stats,
                row_weight, count, kind, graph, node_offsets, graph_id):
        x = _check(x, 'x')
        gamma, beta = _check(gamma, 'gamma'), _check(beta, 'beta')
        rows, cols = x.shape
        dev = x.device
        if cols % 4:
            raise ValueError('the deferred BatchNorm tail needs a multiple of 4 columns')
        count = rows if row_weight is None else int(count)
        if count < 2:
            raise ValueError('Expected more than 1 value per channel when training')
        if stats is None or stats.numel() == 0:
            stats = torch.empty([hip.query('tv_stat_blocks', rows), 2, cols], dtype=_f32, device=dev)
            hip.call('tv_colstats', rows, cols, hip.ptr(x), hip.ptr(row_weight), hip.ptr(stats))
        coeffs = torch.empty([4, cols], dtype=_f32, device=dev)
        mean, rstd, scale, shift = (coeffs[i] for i in range(4))
        hip.call('tv_bn_finalize_coeffs', cols, stats.shape[0], hip.ptr(stats), count, float(eps),
                 float(momentum), hip.ptr(mean), hip.ptr(rstd), hip.ptr(running_mean),
                 hip.ptr(running_var), hip.ptr(gamma), hip.ptr(beta), hip.ptr(scale),
                 hip.ptr(shift), None)
        box = None
        if kind == 'graph_sum':
            if rows != graph.num_rows:
                raise ValueError(f'graph_sum: {rows} rows do not match graph {graph.num_rows}')
            out = torch.empty_like(x)
            box = OperandMax(dev) if cols >= 64 else None
            hip.call('tv_graph_sum', graph.num_rows, 1, cols, hip.ptr(graph.indptr),
                     hip.ptr(graph.indices), hip.ptr(x), 1, 1, hip.ptr(out),
                     None if box is None else box.slots.data_ptr(), hip.ptr(scale), hip.ptr(shift),
                     float(slope))
            if box is not None:
                _LAST_OUT_MAX['graph_sum'] = box
        else:
            batch = node_offsets.shape[0] - 1
            out = torch.empty([batch, cols], dtype=_f32, device=dev)
            hip.call('tv_segment_mean_fwd', batch, cols, hip.ptr(node_offsets), hip.ptr(x),
                     hip.ptr(out), hip.ptr(scale), hip.ptr(shift), float(slope))
        ctx.save_for_backward(x, coeffs, gamma, beta, *(() if kind == 'graph_sum'
                                                       else (node_offsets, graph_id)))
        ctx.kind, ctx.graph = kind, graph
        ctx.slope = float(slope)
        ctx.row_weight, ctx.count = row_weight, count
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, coeffs, gamma, beta = ctx.saved_tensors[:4]
        mean, rstd = coeffs[0], coeffs[1]
        rows, cols = x.shape
        dout = _check(dout, 'dout')
        dact = torch.empty_like(x)
        if ctx.kind == 'graph_sum':
            g = ctx.graph          # symmetric graph: the transpose aggregation is the same kernel
            hip.call('tv_graph_sum', g.num_rows, 1, cols, hip.ptr(g.indptr), hip.ptr(g.indices),
                     hip.ptr(dout), 1, 1, hip.ptr(dact), None, None, None, 0.0)
        else:
            node_offsets, graph_id = ctx.saved_tensors[4:]
            hip.call('tv_segment_mean_bwd', rows, cols, hip.ptr(node_offsets), hip.ptr(graph_id),
                     hip.ptr(dout), hip.ptr(dact))
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        wsb = hip.query('tv_bn_lrelu_bwd_workspace_bytes', rows, cols)
        ws = _workspace(wsb, x.device)
        hip.call('tv_bn_lrelu_bwd', rows, cols, hip.ptr(x), hip.ptr(dact), hip.ptr(mean),
                 hip.ptr(rstd), hip.ptr(gamma), hip.ptr(beta), ctx.slope, hip.ptr(ctx.row_weight),
                 ctx.count, hip.ptr(dx), hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(ws), wsb)
        return (dx, dgamma, dbeta) + (None,) * 12


def bn_lrelu_graph_sum(x, gamma, beta, running_mean, running_var, eps: float, graph: DeviceGraph,
                       momentum: float = 0.1, slope: float = 0.01, stats=None, row_weight=None,
                       count=None):
    """`graph_sum(bn_lrelu(x, training=True), graph)` as one operator (see `_BNLReLUTail`)."""
    out = _BNLReLUTail.apply(x, gamma, beta, running_mean, running_var, eps, momentum, slope,
                             stats, row_weight, count, 'graph_sum', graph, None, None)
    box = _LAST_OUT_MAX.pop('graph_sum', None)
    if box is not None:
        box.attach(out)
    return out


def bn_lrelu_segment_mean(x, gamma, beta, running_mean, running_var, eps: float, node_offsets,
                          graph_id, momentum: float = 0.1, slope: float = 0.01, stats=None,
                          row_weight=None, count=None):
    """`segment_mean(bn_lrelu(x, training=True), ...)` as one operator (see `_BNLReLUTail`)."""
    return _BNLReLUTail.apply(x, gamma, beta, running_mean, running_var, eps, momentum, slope,
                              stats, row_weight, count, 'segment_mean', None, node_offsets, graph_id)


class _Embedding(Function):
    @staticmethod
    def forward(ctx, weight, ids, extra_cols: int):
        weight = _check(weight, 'weight')
        ids_flat = _check(ids.reshape(-1), 'ids', torch.int64)
        rows, dim = ids_flat.shape[0], weight.shape[1]
        ld = dim + extra_cols
        out = torch.empty([rows, ld], dtype=_f32, device=weight.device)
        hip.call('tv_embedding_fwd', rows, dim, weight.shape[0], hip.ptr(weight), hip.ptr(ids_flat),
                 hip.ptr(out), ld, 0)
        ctx.save_for_backward(ids_flat)
        ctx.shape = weight.shape
        ctx.ld = ld
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (ids_flat,) = ctx.saved_tensors
        dout = _check(dout, 'dout')
        v, dim = ctx.shape
        dw = torch.empty([v, dim], dtype=_f32, device=dout.device)
        hip.call('tv_embedding_bwd', ids_flat.shape[0], dim, v, hip.ptr(ids_flat), hip.ptr(dout),
                 ctx.ld, 0, hip.ptr(dw))
        return dw, None, None


def embedding(weight, ids, extra_cols: int = 0):
    """Row gather into the first `dim` columns of a `[rows, dim + extra_cols]` buffer (the
    extra columns are left for the caller to fill, e.g. the latency feature)."""
    return _Embedding.apply(weight, ids, extra_cols)


class _LatDecInput(Function):
    @staticmethod
    def forward(ctx, z2, emb_weight, node_type_pad, n_samples: int):
        z2 = _check(z2, 'z2')
        emb_weight = _check(emb_weight, 'emb_weight')
        batch, m = node_type_pad.shape
        zdim, edim = z2.shape[-1], emb_weight.shape[1]
        out = torch.empty([n_samples * batch * m, zdim + edim], dtype=_f32, device=z2.device)
        hip.call('tv_latdec_input_fwd', n_samples, batch, m, zdim, edim, emb_weight.shape[0],
                 hip.ptr(z2), hip.ptr(emb_weight), hip.ptr(node_type_pad), hip.ptr(out))
        ctx.dims = (n_samples, batch, m, zdim, edim)
        ctx.z2_shape = z2.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        k, b, m, zdim, edim = ctx.dims
        dout = _check(dout, 'dout')
        dz2 = torch.empty(ctx.z2_shape, dtype=_f32, device=dout.device)
        hip.call('tv_latdec_input_bwd', k, b, m, zdim, edim, hip.ptr(dout), hip.ptr(dz2))
        return dz2, None, None, None


class _LatDecRows(Function):
    @staticmethod
    def forward(ctx, z2, emb_weight, row_graph, row_type, node_offsets, n_samples: int):
        z2 = _check(z2, 'z2')
        emb_weight = _check(emb_weight, 'emb_weight')
        rows, batch = row_graph.shape[0], node_offsets.shape[0] - 1
        zdim, edim = z2.shape[-1], emb_weight.shape[1]
        out = torch.empty([n_samples * rows, zdim + edim], dtype=_f32, device=z2.device)
        hip.call('tv_latdec_rows_fwd', n_samples, rows, batch, zdim, edim, emb_weight.shape[0],
                 hip.ptr(z2), hip.ptr(emb_weight), hip.ptr(row_graph), hip.ptr(row_type),
                 hip.ptr(out))
        ctx.dims = (n_samples, rows, batch, zdim, edim)
        ctx.z2_shape = z2.shape
        ctx.save_for_backward(node_offsets)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        k, rows, b, zdim, edim = ctx.dims
        node_offsets, = ctx.saved_tensors
        dout = _check(dout, 'dout')
        dz2 = torch.empty(ctx.z2_shape, dtype=_f32, device=dout.device)
        hip.call('tv_latdec_rows_bwd', k, rows, b, zdim, edim, hip.ptr(node_offsets),
                 hip.ptr(dout), hip.ptr(dz2))
        return dz2, None, None, None, None, None


def latdec_rows(z2, emb_weight, row_graph, row_type, node_offsets, n_samples: int):
    """Compact latency-decoder input: one row per real node plus one per trace."""
    return _LatDecRows.apply(z2, emb_weight, row_graph, row_type, node_offsets, n_samples)


def latdec_input(z2, emb_weight, node_type_pad, n_samples: int):
    """Rows `[z2 (broadcast over the M slots) || stop_grad(emb[node_type_pad])]`."""
    return _LatDecInput.apply(z2, emb_weight, node_type_pad, n_samples)


# ---------------------------------------------------------------------------------------
# likelihoods
# ---------------------------------------------------------------------------------------
class _AdjLL(Function):
    @staticmethod
    def forward(ctx, h, adj, n_samples: int, biased: bool, alpha: float, log_thr: float):
        h = _check(h, 'h')
        adj = _check(adj, 'adj')
        batch = adj.shape[0]
        m, dim = h.shape[-2], h.shape[-1]
        ll = torch.empty([n_samples * batch], dtype=_f32, device=h.device)
        wsb = hip.query('tv_adj_bernoulli_ll_fwd_workspace_bytes', n_samples, batch, m)
        ws = _workspace(wsb, h.device)
        hip.call('tv_adj_bernoulli_ll_fwd', n_samples, batch, m, dim, hip.ptr(h), hip.ptr(adj),
                 int(biased), float(alpha), float(log_thr), hip.ptr(ll), hip.ptr(ws), wsb)
        ctx.save_for_backward(h, adj)
        ctx.args = (n_samples, batch, m, dim, int(biased), float(alpha), float(log_thr))
        return ll

    @staticmethod
    @once_differentiable
    def backward(ctx, dll):
        h, adj = ctx.saved_tensors
        k, b, m, dim, biased, alpha, log_thr = ctx.args
        dll = _check(dll, 'dll')
        dh = torch.empty_like(h)
        wsb = hip.query('tv_adj_bernoulli_ll_bwd_workspace_bytes', k, b, m, dim)
        ws = _workspace(wsb, h.device)
        hip.call('tv_adj_bernoulli_ll_bwd', k, b, m, dim, hip.ptr(h), hip.ptr(adj), biased, alpha,
                 log_thr, hip.ptr(dll), hip.ptr(dh), hip.ptr(ws), wsb)
        return dh, None, None, None, None, None


def adj_bernoulli_ll(h, adj, n_samples: int, biased: bool = False, alpha: float = 1.0,
                     threshold: float = 0.5):
    """h [K*B, M, D] (or [K,B,M,D]), adj [B,T] -> log p(adj) [K*B]."""
    return _AdjLL.apply(h, adj, n_samples, biased, alpha, math.log(threshold))


def edge_logits_triu(h: torch.Tensor) -> torch.Tensor:
    """[graphs, M, D] -> [graphs, M(M-1)/2]: edge_logits_by_dot_product + dense_to_triu
    (tensor_utils.py:54-67) as values (`tv_edge_logits_triu`); no gradient - the likelihood and
    its gradient come from `adj_bernoulli_ll`, which never materialises the logits."""
    h = _check(h.detach(), 'h')
    graphs, m, d = h.shape
    out = torch.empty([graphs, m * (m - 1) // 2], dtype=_f32, device=h.device)
    for g0 in range(0, graphs, 65535):
        g1 = min(graphs, g0 + 65535)
        hip.call('tv_edge_logits_triu', g1 - g0, m, d, hip.ptr(h[g0:g1]), hip.ptr(out[g0:g1]))
    return out


class _CategoricalLL(Function):
    @staticmethod
    def forward(ctx, logits, target, biased: bool, alpha: float, log_thr: float):
        logits = _check(logits, 'logits')
        target = _check(target.reshape(-1), 'target', torch.int64)
        rows, classes = logits.shape
        ll = torch.empty([rows], dtype=_f32, device=logits.device)
        lse = torch.empty([rows], dtype=_f32, device=logits.device)
        hip.call('tv_categorical_ll_fwd', rows, classes, target.shape[0], hip.ptr(logits),
                 hip.ptr(target), int(biased), float(alpha), float(log_thr), hip.ptr(ll),
                 hip.ptr(lse))
        ctx.save_for_backward(logits, target, lse)
        ctx.args = (int(biased), float(alpha), float(log_thr))
        return ll

    @staticmethod
    @once_differentiable
    def backward(ctx, dll):
        logits, target, lse = ctx.saved_tensors
        biased, alpha, log_thr = ctx.args
        dll = _check(dll, 'dll')
        # rows padded to a multiple of 4 floats (the pad is written as zeros): the products that
        # take this gradient (dx = dlogits . W^T over the class axis) can then read it 16 bytes
        # at a time whatever the number of classes is (1001 operations: 4-byte loads otherwise)
        rows, classes = logits.shape
        padded = torch.empty([rows, (classes + 3) // 4 * 4], dtype=_f32, device=logits.device)
        hip.call('tv_categorical_ll_bwd', rows, classes, target.shape[0],
                 hip.ptr(logits), hip.ptr(target), biased, alpha, log_thr, hip.ptr(lse),
                 hip.ptr(dll), hip.ptr(padded), padded.shape[1])
        return padded[:, :classes], None, None, None, None


def categorical_ll(logits, target, biased: bool = False, alpha: float = 1.0,
                   threshold: float = 0.5):
    """logits [rows, V]; target [target_rows] with rows % target_rows == 0 (the sample axis
    of `logits` is outermost) -> log-prob per row."""
    return _CategoricalLL.apply(logits, target, biased, alpha, math.log(threshold))


def _bounds(lo, hi):
    return (-math.inf if lo is None else float(lo), math.inf if hi is None else float(hi))


class _NormalReparam(Function):
    @staticmethod
    def forward(ctx, mean, logstd_raw, eps, lo: float, hi: float):
        mean, logstd_raw, eps = _check(mean, 'mean'), _check(logstd_raw, 'logstd'), _check(eps, 'eps')
        batch, dim = mean.shape
        k = eps.numel() // (batch * dim)
        z = torch.empty_like(eps)
        logq = torch.empty(eps.shape[:-1], dtype=_f32, device=mean.device)
        logstd = torch.empty_like(mean)
        hip.call('tv_normal_reparam_fwd', k, batch, dim, hip.ptr(mean), hip.ptr(logstd_raw), lo, hi,
                 hip.ptr(eps), hip.ptr(z), hip.ptr(logq), hip.ptr(logstd))
        ctx.save_for_backward(logstd_raw, eps)
        ctx.args = (k, batch, dim, lo, hi)
        ctx.mark_non_differentiable(logstd)
        ctx.set_materialize_grads(False)
        return z, logq, logstd

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dlogq, _dlogstd):
        if dz is None and dlogq is None:
            return None, None, None, None, None
        logstd_raw, eps = ctx.saved_tensors
        k, batch, dim, lo, hi = ctx.args
        dz = _check(dz, 'dz') if dz is not None else None
        dlogq = _check(dlogq, 'dlogq') if dlogq is not None else None
        dmean = torch.empty_like(logstd_raw)
        dls = torch.empty_like(logstd_raw)
        hip.call('tv_normal_reparam_bwd', k, batch, dim, hip.ptr(logstd_raw), lo, hi, hip.ptr(eps),
                 hip.ptr(dz), hip.ptr(dlogq), hip.ptr(dmean), hip.ptr(dls))
        return dmean, dls, None, None, None


def normal_reparam(mean, logstd_raw, eps, lo=None, hi=None):
    """z = mean + exp(clamp(logstd)) * eps, log q(z) summed over the last axis.
    mean/logstd_raw [B,D]; eps [B,D] or [K,B,D].  Returns (z, logq, clamped logstd)."""
    lo, hi = _bounds(lo, hi)
    return _NormalReparam.apply(mean, logstd_raw, eps, lo, hi)


class _NormalLL(Function):
    @staticmethod
    def forward(ctx, x, mean, logstd_raw, lo: float, hi: float):
        x, mean, logstd_raw = _check(x, 'x'), _check(mean, 'mean'), _check(logstd_raw, 'logstd')
        dim = x.shape[-1]
        rows = x.numel() // dim
        ll = torch.empty(x.shape[:-1], dtype=_f32, device=x.device)
        hip.call('tv_normal_ll_fwd', rows, dim, hip.ptr(x), hip.ptr(mean), hip.ptr(logstd_raw),
                 lo, hi, hip.ptr(ll))
        ctx.save_for_backward(x, mean, logstd_raw)
        ctx.args = (rows, dim, lo, hi)
        return ll

    @staticmethod
    @once_differentiable
    def backward(ctx, dll):
        x, mean, logstd_raw = ctx.saved_tensors
        rows, dim, lo, hi = ctx.args
        dll = _check(dll, 'dll')
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dmean = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        dls = torch.empty_like(x) if ctx.needs_input_grad[2] else None
        hip.call('tv_normal_ll_bwd', rows, dim, hip.ptr(x), hip.ptr(mean), hip.ptr(logstd_raw),
                 lo, hi, hip.ptr(dll), hip.ptr(dx), hip.ptr(dmean), hip.ptr(dls))
        return dx, dmean, dls, None, None


def normal_ll(x, mean, logstd_raw, lo=None, hi=None):
    """sum over the last axis of Normal(x; mean, clamp(logstd_raw)); all three same shape."""
    lo, hi = _bounds(lo, hi)
    return _NormalLL.apply(x, mean, logstd_raw, lo, hi)


class _RealNVPLogProb(Function):
    @staticmethod
    def forward(ctx, z, levels: int, hidden: int, slope: float, *params):
        import ctypes
        z = _check(z, 'z')
        dim = z.shape[-1]
        rows = z.numel() // dim
        params = [_check(p, 'flow parameter') for p in params]
        if len(params) != 9 * levels:
            raise ValueError(f'expected {9 * levels} flow parameters, got {len(params)}')
        n1 = dim // 2
        shapes = [(dim,), (dim,), (hidden, n1), (hidden,), (dim - n1, hidden), (dim - n1,),
                  (dim - n1, hidden), (dim - n1,), (dim, dim)]
        for i, p in enumerate(params):
            if tuple(p.shape) != shapes[i % 9]:
                raise ValueError(f'flow parameter {i % 9} of level {i // 9} has shape '
                                 f'{tuple(p.shape)}, expected {shapes[i % 9]}')
        size = hip.query('tv_realnvp_workspace_floats', rows, levels, dim, hidden)
        if size < 0:
            raise hip.HipLibraryError(f'no RealNVP kernel for dim {dim} / hidden {hidden} / '
                                      f'{levels} levels')
        ws = torch.empty([size], dtype=_f32, device=z.device)
        table = hip.array_type(ctypes.c_void_p, len(params))(*[p.data_ptr() for p in params])
        logp = torch.empty(z.shape[:-1], dtype=_f32, device=z.device)
        with TIMER.region('realnvp_fwd', 0.0):
            hip.call('tv_realnvp_logprob_fwd', rows, levels, dim, hidden, slope, hip.ptr(z), table,
                     hip.ptr(ws), hip.ptr(logp))
        ctx.save_for_backward(z, ws, *params)
        ctx.args = (rows, levels, dim, hidden, slope)
        return logp

    @staticmethod
    @once_differentiable
    def backward(ctx, dlogp):
        import ctypes
        z, ws, *params = ctx.saved_tensors
        rows, levels, dim, hidden, slope = ctx.args
        dlogp = _check(dlogp, 'dlogp')
        need = ctx.needs_input_grad[4:]
        packed = torch.empty([sum(p.numel() for p in params)], dtype=_f32, device=z.device)
        grads, ptrs, off = [], [], 0
        for p, wanted in zip(params, need):
            view = packed[off:off + p.numel()].view(p.shape)
            off += p.numel()
            grads.append(view if wanted else None)
            ptrs.append(view.data_ptr() if wanted else None)
        dz = torch.empty_like(z)
        table = hip.array_type(ctypes.c_void_p, len(params))(*[p.data_ptr() for p in params])
        gtable = hip.array_type(ctypes.c_void_p, len(params))(*ptrs)
        with TIMER.region('realnvp_bwd', 0.0):
            hip.call('tv_realnvp_logprob_bwd', rows, levels, dim, hidden, slope, hip.ptr(z), table,
                     hip.ptr(dlogp), hip.ptr(ws), hip.ptr(dz), gtable)
        return (dz if ctx.needs_input_grad[0] else None, None, None, None) + tuple(grads)


def realnvp_actnorm_init(z, params, levels: int, hidden: int, slope: float) -> None:
    """Data-dependent ActNorm initialisation on the first batch (`tv_realnvp_actnorm_init`):
    writes bias / pre_scale of every level in place."""
    import ctypes
    z = _check(z.detach().reshape(-1, z.shape[-1]), 'z')
    params = [_check(p.detach(), 'flow parameter') for p in params]
    if len(params) != 9 * levels:
        raise ValueError(f'expected {9 * levels} flow parameters, got {len(params)}')
    table = hip.array_type(ctypes.c_void_p, len(params))(*[p.data_ptr() for p in params])
    hip.call('tv_realnvp_actnorm_init', z.shape[0], levels, z.shape[1], hidden, float(slope),
             hip.ptr(z), table)


def realnvp_supported(dim: int, hidden: int, levels: int) -> bool:
    return 1 <= levels <= 8 and bool(hip.query('tv_realnvp_supported', dim, hidden))


def realnvp_log_prob(z, params, levels: int, hidden: int, slope: float):
    """log p(z) of the RealNVP prior, z [..., dim]; `params`: 9 tensors per level in the order
    of `tv_realnvp_logprob_fwd` (include/tracevae_hip.h)."""
    return _RealNVPLogProb.apply(z, levels, hidden, slope, *params)


class _TrainLoss(Function):
    """(loss, recons, kl) of train_step from the chain's log-prob vectors, one kernel each way."""

    @staticmethod
    def forward(ctx, kinds, regs, beta, l2_value, l2_reg, n_vec, *tensors):
        import ctypes
        vecs = [_check(t.reshape(-1), 'log_prob') for t in tensors[:n_vec]]
        zs = [_check(t, 'z') for t in tensors[n_vec:]]
        batch = vecs[0].shape[0]
        for v in vecs:
            if v.shape[0] != batch:
                raise ValueError('train_loss: log-prob vectors of different lengths')
        for z in zs:
            if z.dim() != 2 or z.shape[0] != batch:
                raise ValueError(f'train_loss: latent of shape {tuple(z.shape)} for batch {batch}')
        out = torch.empty([3], dtype=_f32, device=vecs[0].device)
        ctx.tables = (
            hip.array_type(ctypes.c_void_p, len(vecs))(*[v.data_ptr() for v in vecs]),
            hip.array_type(ctypes.c_int, len(vecs))(*kinds),
            hip.array_type(ctypes.c_void_p, max(len(zs), 1))(*[z.data_ptr() for z in zs]),
            hip.array_type(ctypes.c_int64, max(len(zs), 1))(*[z.shape[1] for z in zs]),
            hip.array_type(ctypes.c_float, max(len(zs), 1))(*regs))
        vt, kt, zt, dt, rt = ctx.tables
        hip.call('tv_train_loss_fwd', batch, len(vecs), vt, kt, len(zs), zt, dt, rt,
                 hip.ptr(l2_value), float(l2_reg), float(beta), hip.ptr(out))
        ctx.save_for_backward(*vecs, *zs)
        ctx.meta = (batch, len(vecs), len(zs), float(beta), [t.shape for t in tensors])
        loss, recons, kl = out[0], out[1], out[2]
        ctx.mark_non_differentiable(recons, kl)
        ctx.set_materialize_grads(False)
        return loss, recons, kl

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss, _dr, _dk):
        import ctypes
        batch, n_vec, n_z, beta, shapes = ctx.meta
        if dloss is None:
            return (None,) * (6 + len(shapes))
        saved = ctx.saved_tensors
        need = ctx.needs_input_grad[6:]
        dev = saved[0].device
        dloss = _check(dloss.reshape(1), 'dloss')
        grads = [torch.empty_like(t) if w else None for t, w in zip(saved, need)]
        vt, kt, zt, dt, rt = ctx.tables
        gv = hip.array_type(ctypes.c_void_p, n_vec)(*[g.data_ptr() if g is not None else None
                                         for g in grads[:n_vec]])
        gz = hip.array_type(ctypes.c_void_p, max(n_z, 1))(*[g.data_ptr() if g is not None else None
                                               for g in grads[n_vec:]])
        hip.call('tv_train_loss_bwd', batch, n_vec, vt, kt, gv, n_z, zt, dt, rt, gz, beta,
                 hip.ptr(dloss))
        out = [g.reshape(sh) if g is not None else None for g, sh in zip(grads, shapes)]
        return (None, None, None, None, None, None) + tuple(out)


def train_loss(obs, q_latent, p_latent, beta: float, l2_value=None, l2_reg: float = 0.0,
               latents=()):
    """train.py:229-270 in one kernel: `obs` / `q_latent` / `p_latent` are lists of log-prob
    vectors [B]; `latents` is a list of `(z [B, D], reg)` for the unit-ball regulariser.
    Returns `(loss, recons, kl)` scalars; only `loss` carries gradients."""
    vecs = list(obs) + list(q_latent) + list(p_latent)
    kinds = [0] * len(obs) + [1] * len(q_latent) + [2] * len(p_latent)
    zs = [z for z, _ in latents]
    regs = [float(r) for _, r in latents]
    return _TrainLoss.apply(kinds, regs, beta, l2_value, l2_reg, len(vecs), *vecs, *zs)


LATENCY_MODES = {'normal': 0, 'safe': 1, 'biased': 2, 'anomaly': 3}


class _LatencyLL(Function):
    @staticmethod
    def forward(ctx, feats, latency_pad, node_count, node_type_pad, std_limit, n_samples, cfg,
                node_offsets=None):
        feats = _check(feats, 'feats')
        batch, m = latency_pad.shape[0], latency_pad.shape[1]
        ll = torch.empty([n_samples * batch], dtype=_f32, device=feats.device)
        std_out = torch.empty([n_samples * batch * m], dtype=_f32, device=feats.device)
        args = (n_samples, batch, m, hip.ptr(feats), hip.ptr(latency_pad), hip.ptr(node_count),
                hip.ptr(node_type_pad), hip.ptr(node_offsets), hip.ptr(std_limit)) + cfg
        ctx.node_offsets = node_offsets
        hip.call('tv_latency_ll_fwd', *args, hip.ptr(ll), hip.ptr(std_out))
        ctx.save_for_backward(feats, latency_pad, node_count, node_type_pad,
                              std_limit if std_limit is not None else feats.new_empty(0))
        ctx.has_limit = std_limit is not None
        ctx.n_samples, ctx.cfg = n_samples, cfg
        ctx.mark_non_differentiable(std_out)
        ctx.set_materialize_grads(False)
        return ll, std_out

    @staticmethod
    @once_differentiable
    def backward(ctx, dll, _dstd):
        if dll is None:
            return (None,) * 8
        feats, latency_pad, node_count, node_type_pad, std_limit = ctx.saved_tensors
        if not ctx.has_limit:
            std_limit = None
        dll = _check(dll, 'dll')
        batch, m = latency_pad.shape[0], latency_pad.shape[1]
        dfeats = torch.empty_like(feats)
        args = (ctx.n_samples, batch, m, hip.ptr(feats), hip.ptr(latency_pad), hip.ptr(node_count),
                hip.ptr(node_type_pad), hip.ptr(ctx.node_offsets), hip.ptr(std_limit)) + ctx.cfg
        hip.call('tv_latency_ll_bwd', *args, hip.ptr(dll), hip.ptr(dfeats))
        return dfeats, None, None, None, None, None, None, None


def latency_ll(feats, latency_pad, node_count, node_type_pad, n_samples: int, *, mode: str,
               std_threshold: float = 0.0, alpha: float = 1.0, bias_threshold: float = 0.5,
               logstd_min=None, std_limit=None, use_mask: bool = True, weighted: bool = False,
               node_offsets=None):
    """feats [K*B*M, 2] -> (log p(latency) [K*B], std [K*B*M]).  With `node_offsets` [B+1]
    feats are compact rows [K*(N+B), 2] (real nodes, then one row per trace for its padded
    slots); std stays in the padded layout."""
    cfg = (int(logstd_min is not None), float(logstd_min if logstd_min is not None else 0.0),
           LATENCY_MODES[mode], float(std_threshold), float(alpha), math.log(bias_threshold),
           int(use_mask), int(weighted))
    lp = _check(latency_pad.reshape(latency_pad.shape[0], -1), 'latency_pad')
    return _LatencyLL.apply(feats, lp, node_count, node_type_pad, std_limit, n_samples, cfg,
                            node_offsets)


def score_reduce(w: torch.Tensor, clip: Optional[float]):
    """w = log_joint - latent_log_joint [K,B] -> (loss [B], nll [B]) (evaluation.py:141-148)."""
    w = _check(w, 'w')
    k, b = w.shape
    loss = torch.empty([b], dtype=_f32, device=w.device)
    nll = torch.empty([b], dtype=_f32, device=w.device)
    hip.call('tv_score_reduce', k, b, hip.ptr(w), int(clip is not None),
             float(clip if clip is not None else 0.0), hip.ptr(loss), hip.ptr(nll))
    return loss, nll
