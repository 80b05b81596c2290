"""TraceVAE on the HIP kernels: same Python model API and the same `state_dict()` layout as
the reference, different engine underneath.

Reference classes mirrored here (constructor arguments, method names and keyword
signatures are the reference's):

* `TraceVAE`          model/trace_vae.py:40-161        `.q(G, observed, n_z, no_latency)`,
                                                       `.p(observed, G, n_z, no_latency, use_biased, ...)`
* `TraceStructVAE`    model/struct_vae.py:92-414
* `TraceLatencyVAE`   model/latency_vae.py:125-444
* `GATConvAgg`, `GraphConv`, `GNNSequential`           model/gnn_layers.py:102-190
* `OperationEmbedding`                                 model/operation_embedding.py:10-25
* RealNVP prior flow                                   model/realnvp_flow.py:18-49

Module attribute names reproduce the reference's `state_dict()` keys (SURVEY.md appendix C;
the tensorkit-side names are unverified there and collected in `checkpoint.KEY_NOTES`).

What runs where: every operator on a `[rows, features]` tensor is a HIP kernel (`ops.py`),
the RealNVP prior included (`csrc/flow.hip`); a handful of `[B]`-sized index / reshape steps
stay tensor ops.
"""
import math
from typing import Dict, List, Mapping, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import constants as C
from . import hip, ops
from .batch import DeviceBatch, TraceGraphBatch, build_sampled_graphs
from .bayes import BayesianNet, Distribution, StochasticTensor
from .config import TraceLatencyVAEConfig, TraceStructVAEConfig, TraceVAEConfig

__all__ = ['TraceVAE', 'TraceStructVAE', 'TraceLatencyVAE', 'OperationEmbedding', 'BN_EPSILON']

# tensorkit's BatchNorm epsilon (believed 1e-6, SURVEY.md appendix B3 - unverified)
BN_EPSILON = 1e-6
# latency decoder on compact rows (real nodes + one representative per trace) wherever the
# padded slots are provably identical; the padded [B*M] rows are the form for every other case
# (observed node types that are not the batch's own, no latency mask).  A module constant, not a
# run-time switch: the equivalence test patches it.
COMPACT_LATENCY_DECODER = True
LRELU_SLOPE = 0.01
# training: BatchNorm + LeakyReLU between two GAT layers applied inside the next layer's GEMMs
# instead of as a pass of its own (the materialised form serves the layers in front of a pooling
# or GraphConv tail and eval mode; the equivalence test patches this constant)
FUSED_BN_LINEAR = True
# training: a GAT layer as one library call each way (`ops.gat_layer_train`) instead of one call
# per kernel - the same launches, issued from inside the library (the equivalence test patches
# this constant; a kernel-timing run, `ops.TIMER.enabled`, takes the operator-level calls, whose
# HIP events bracket single kernels)
FUSED_GAT_LAYER = True
# the BatchNorm + LeakyReLU that ends a stack applied inside the stack's tail kernel
DEFERRED_TAILS = True
# a stack's extended weights and their plane images built by two launches per stack
PREPARED_STACK_WEIGHTS = True


# ---------------------------------------------------------------------------------------
# tensorkit-shaped layers
# ---------------------------------------------------------------------------------------
class ParamStore(nn.Module):
    def __init__(self, value: torch.Tensor):
        super().__init__()
        self.value = nn.Parameter(value)


def _kaiming_uniform(out_features: int, in_features: int) -> torch.Tensor:
    w = torch.empty(out_features, in_features)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    return w


class Linear(nn.Module):
    """tk.layers.Linear: y = x W^T + b, weight stored [out, in]."""

    def __init__(self, in_features: int, out_features: int, use_bias: bool = True,
                 zero_weight: bool = False):
        super().__init__()
        w = torch.zeros(out_features, in_features) if zero_weight else \
            _kaiming_uniform(out_features, in_features)
        self.weight_store = ParamStore(w)
        self.bias_store = ParamStore(torch.zeros(out_features)) if use_bias else None

    def forward(self, x):
        return ops.linear(x, self.weight_store.value,
                          self.bias_store.value if self.bias_store is not None else None)


class BatchNorm(nn.Module):
    """tk.layers.BatchNorm (= torch BatchNorm1d over rows), fused with the LeakyReLU that
    always follows it on this path."""

    def __init__(self, num_features: int, momentum: float = 0.1, eps: float = BN_EPSILON):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self.momentum, self.eps = momentum, eps
        self._pending_batches = 0

    def count_batch(self):
        """One more training batch.  The counter only matters when the state is saved, so it is
        kept on the host and folded into the `num_batches_tracked` buffer on demand (21 one-element
        device kernels per step otherwise)."""
        self._pending_batches += 1

    def _fold_counter(self):
        if self._pending_batches:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._fold_counter()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        super()._load_from_state_dict(*args, **kwargs)

    def forward_lrelu(self, x, stats=None, row_weight=None, count=None):
        if self.training:
            self.count_batch()
        return ops.bn_lrelu(x, self.weight, self.bias, self.running_mean, self.running_var,
                            self.training, self.eps, self.momentum, LRELU_SLOPE, stats,
                            row_weight, count)

    def eval_coeffs(self):
        return ops.bn_eval_coeffs(self.running_mean, self.running_var, self.weight, self.bias,
                                  self.eps)


class _Identity(object):
    """LeakyReLU without normalisation, as one affine pass with scale 1 / shift 0."""
    _cache: Dict[tuple, tuple] = {}

    @classmethod
    def lrelu(cls, x):
        key = (x.device, x.shape[-1])
        if key not in cls._cache:
            cls._cache[key] = (torch.ones(x.shape[-1], device=x.device),
                               torch.zeros(x.shape[-1], device=x.device))
        scale, shift = cls._cache[key]
        return ops._AffineLReLU.apply(x, scale, shift, LRELU_SLOPE)


class Dense(nn.Module):
    """tk `dense`: Linear(bias iff no normaliser) -> BatchNorm -> LeakyReLU (appendix B3)."""

    def __init__(self, in_features: int, out_features: int, use_batch_norm: bool):
        super().__init__()
        self.weight_store = ParamStore(_kaiming_uniform(out_features, in_features))
        self.bias_store = None if use_batch_norm else ParamStore(torch.zeros(out_features))
        self.normalizer = BatchNorm(out_features) if use_batch_norm else None

    def forward(self, x):
        if self.normalizer is not None:
            return self.normalizer.forward_lrelu(ops.linear(x, self.weight_store.value))
        return _Identity.lrelu(ops.linear(x, self.weight_store.value, self.bias_store.value))


class MLP(nn.Module):
    """`SequentialBuilder(...).dense(..)...[.linear(n)].build(flatten_to_ndims=True)`."""

    def __init__(self, in_features: int, hidden: List[int], use_batch_norm: bool,
                 out_features: Optional[int] = None):
        super().__init__()
        layers: List[nn.Module] = []
        for size in hidden:
            layers.append(Dense(in_features, size, use_batch_norm))
            in_features = size
        if out_features is not None:
            layers.append(Linear(in_features, out_features))
            in_features = out_features
        self.layer = nn.ModuleList(layers)
        self.out_features = in_features

    def forward(self, x):
        lead = x.shape[:-1]
        h = x.reshape(-1, x.shape[-1])
        for layer in self.layer:
            h = layer(h)
        return h.reshape(lead + (h.shape[-1],))


class OperationEmbedding(nn.Module):
    def __init__(self, num_operations: int, embedding_dim: int):
        super().__init__()
        self.num_operations, self.embedding_dim = num_operations, embedding_dim
        self.node_embedding = nn.Embedding(num_operations, embedding_dim)

    @property
    def weight(self) -> torch.Tensor:
        return self.node_embedding.weight

    def forward(self, node_type: torch.Tensor) -> torch.Tensor:
        out = ops.embedding(self.weight, node_type)
        return out.reshape(node_type.shape + (self.embedding_dim,))


# ---------------------------------------------------------------------------------------
# GNN layers
# ---------------------------------------------------------------------------------------
class _GATParams(nn.Module):
    """Parameters of dgl.nn.GATConv(in, out, heads): fc.weight [H*out, in] (no fc bias),
    attn_l / attn_r [1, H, out], bias [H*out]; Xavier-normal(gain sqrt 2) / zeros."""

    def __init__(self, in_feats: int, out_feats: int, heads: int):
        super().__init__()
        self.fc = nn.Linear(in_feats, out_feats * heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, heads, out_feats))
        self.bias = nn.Parameter(torch.zeros(heads * out_feats))
        gain = nn.init.calculate_gain('relu')
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)


class _DeferredBN(object):
    """Output of a GAT layer whose BatchNorm(batch statistics) + LeakyReLU has not been applied:
    the next layer's fc applies it to its operand inside the GEMM (`ops.bn_lrelu_linear`)."""
    __slots__ = ['x', 'stats', 'bn', 'row_weight', 'count']

    def __init__(self, x, stats, bn, row_weight, count):
        self.x, self.stats, self.bn, self.row_weight, self.count = x, stats, bn, row_weight, count

    def linear(self, weight, attn_l=None, attn_r=None):
        bn = self.bn
        bn.count_batch()
        return ops.bn_lrelu_linear(self.x, bn.weight, bn.bias, weight, bn.running_mean,
                                   bn.running_var, bn.eps, bn.momentum, LRELU_SLOPE, self.stats,
                                   self.row_weight, self.count, attn_l, attn_r)

    # the stack's tail applies it as it reads the rows (`ops._BNLReLUTail`)
    def graph_sum(self, graph):
        bn = self.bn
        bn.count_batch()
        return ops.bn_lrelu_graph_sum(self.x, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      bn.eps, graph, bn.momentum, LRELU_SLOPE, self.stats,
                                      self.row_weight, self.count)

    def segment_mean(self, node_offsets, graph_id):
        bn = self.bn
        bn.count_batch()
        return ops.bn_lrelu_segment_mean(self.x, bn.weight, bn.bias, bn.running_mean,
                                         bn.running_var, bn.eps, node_offsets, graph_id,
                                         bn.momentum, LRELU_SLOPE, self.stats, self.row_weight,
                                         self.count)

    def materialise(self):
        return self.bn.forward_lrelu(self.x, self.stats, self.row_weight, self.count)


class GATConvAgg(nn.Module):
    """GATConv -> concat heads -> BatchNorm -> LeakyReLU (gnn_layers.py:112-157)."""

    def __init__(self, input_dim: int, output_dim: int, num_heads: int, use_batch_norm: bool):
        super().__init__()
        self.output_dim = output_dim * num_heads
        self.gnn = _GATParams(input_dim, output_dim, num_heads)
        self.normalization = BatchNorm(self.output_dim) if use_batch_norm else None

    def takes_fused_call(self, n_rep: int) -> bool:
        bn = self.normalization
        return (bn is not None and bn.training and n_rep == 1 and FUSED_GAT_LAYER and FUSED_BN_LINEAR
                and not ops.TIMER.enabled)

    def forward(self, graph: ops.DeviceGraph, h, n_rep: int = 1, row_weight=None, count=None,
                defer: bool = False, prepared=None):
        """`row_weight` / `count`: rows that stand for several identical rows of the batch
        (compact latency decoder); only the BatchNorm statistics see the difference.
        `defer`: in training mode return a `_DeferredBN` for the next GAT layer to consume."""
        g = self.gnn
        bn = self.normalization
        if self.takes_fused_call(n_rep):
            deferred = None
            x = h
            if isinstance(h, _DeferredBN):
                b = h.bn
                b.count_batch()
                deferred = ops.DeferredInput(h.stats, b.weight, b.bias, b.running_mean, b.running_var,
                                             b.eps, b.momentum, h.row_weight, h.count)
                x = h.x
            out, stats = ops.gat_layer_train(x, g.fc.weight, g.attn_l, g.attn_r, g.bias, graph,
                                             deferred, LRELU_SLOPE, row_weight, prepared)
            if defer:
                return _DeferredBN(out, stats, bn, row_weight, count)
            return bn.forward_lrelu(out, stats, row_weight, count)
        # [ft | el | er]: the attention scores are 2H more columns of the fc product
        if isinstance(h, _DeferredBN):
            ft = h.linear(g.fc.weight, g.attn_l, g.attn_r)
        else:
            ft = ops.linear_scores(h, g.fc.weight, g.attn_l, g.attn_r)
        if bn is None:
            out, _ = ops.gat_conv(ft, g.attn_l, g.attn_r, g.bias, graph, n_rep)
            return _Identity.lrelu(out)
        if bn.training:
            out, stats = ops.gat_conv(ft, g.attn_l, g.attn_r, g.bias, graph, n_rep, want_stats=True,
                                      stats_row_weight=row_weight)
            if defer and FUSED_BN_LINEAR:
                return _DeferredBN(out, stats, bn, row_weight, count)
            return bn.forward_lrelu(out, stats, row_weight, count)
        if not torch.is_grad_enabled():
            # eval + no_grad: BatchNorm(running stats) + LeakyReLU ride in the aggregate epilogue
            scale, shift = bn.eval_coeffs()
            out, _ = ops.gat_conv(ft, g.attn_l, g.attn_r, g.bias, graph, n_rep,
                                  ep_scale=scale, ep_shift=shift, ep_slope=LRELU_SLOPE)
            return out
        out, _ = ops.gat_conv(ft, g.attn_l, g.attn_r, g.bias, graph, n_rep)
        return bn.forward_lrelu(out)


class _GraphConvParams(nn.Module):
    def __init__(self, in_feats: int, out_feats: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        self.bias = nn.Parameter(torch.zeros(out_feats))
        nn.init.xavier_uniform_(self.weight)


class GraphConv(nn.Module):
    """Final decoder layer: dgl GraphConv(norm='both') with weight and bias, no BatchNorm /
    activation (gnn_layers.py:160-190 as used at struct_vae.py:201-206, latency_vae.py:221-229)."""

    def __init__(self, input_dim: int, output_dim: int):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.gnn = _GraphConvParams(input_dim, output_dim)

    def takes_deferred(self) -> bool:
        # (the float4 form of the aggregation kernel: rows of a multiple of 4, at least 64, floats)
        return (self.input_dim <= self.output_dim and self.input_dim % 4 == 0
                and self.input_dim >= 64)

    def forward(self, graph: ops.DeviceGraph, h, n_rep: int = 1) -> torch.Tensor:
        w, b = self.gnn.weight, self.gnn.bias
        if isinstance(h, _DeferredBN):
            if self.takes_deferred() and n_rep == 1:
                return ops.matmul_nn(h.graph_sum(graph), w, b)
            h = h.materialise()
        if self.input_dim > self.output_dim:
            # DGL multiplies by W first when in > out (appendix A2)
            if self.output_dim <= 8:
                t = ops.rowdot(h, w)
            else:
                t = ops.matmul_nn(h, w)
            return ops.graph_sum(t, graph, n_rep) + b
        agg = ops.graph_sum(h, graph, n_rep)
        return ops.matmul_nn(agg, w, b)


class _Pooling(nn.Module):
    """dgl.nn.AvgPooling (pooling.py:45): no parameters."""

    def forward(self, batch: DeviceBatch, h) -> torch.Tensor:
        if isinstance(h, _DeferredBN):
            if h.x.shape[1] % 4 == 0:
                return h.segment_mean(batch.node_offsets, batch.graph_id)
            h = h.materialise()
        return ops.segment_mean(h, batch.node_offsets, batch.graph_id)


class GNNSequential(nn.Module):
    """`GNNSequential(layers + [tail])` (gnn_layers.py:102-109): `.gnn.{i}` children."""

    def __init__(self, layers: List[nn.Module], tail: nn.Module):
        super().__init__()
        self.gnn = nn.ModuleList(list(layers) + [tail])

    def _run(self, graph, h, n_rep, tail_arg, row_weight, count):
        layers = list(self.gnn)[:-1]
        tail = self.gnn[len(self.gnn) - 1]
        # the last layer's BatchNorm + LeakyReLU rides in the tail's kernel where the tail reads
        # wide rows (AvgPooling; GraphConv that aggregates before it multiplies)
        tail_defers = DEFERRED_TAILS and n_rep == 1 and (
            isinstance(tail, _Pooling) or (isinstance(tail, GraphConv) and tail.takes_deferred()))
        # the weights of the whole stack prepared in two launches (extended weights, plane images
        # for the fc products and the input gradients) instead of three per layer
        prepared = [None] * len(layers)
        if PREPARED_STACK_WEIGHTS and layers and all(
                isinstance(l, GATConvAgg) and l.takes_fused_call(n_rep) for l in layers):
            weights = self.__dict__.get('_gat_weights')
            if weights is None or any(a is not l.gnn.fc.weight for (a, _, _), l in zip(weights, layers)):
                weights = self.__dict__['_gat_weights'] = [
                    (l.gnn.fc.weight, l.gnn.attn_l, l.gnn.attn_r) for l in layers]
            prepared = ops.prepare_gat_layers(weights, graph.num_rows)
        for i, layer in enumerate(layers):
            h = layer(graph, h, n_rep, row_weight, count, defer=i + 1 < len(layers) or tail_defers,
                      prepared=prepared[i])
        if isinstance(tail, _Pooling):
            return tail(tail_arg, h)
        return tail(graph, h, n_rep)

    def forward(self, graph: ops.DeviceGraph, h: torch.Tensor, n_rep: int = 1, batch=None,
                row_weight=None, count=None):
        bn_training = any(isinstance(m, BatchNorm) and m.training for m in self.modules())
        if n_rep > 1 and bn_training:
            # apply_gnn_layer (gnn_layers.py:90-97): one pass per sample, so that batch
            # statistics are per sample like in the reference
            rows = h.shape[0] // n_rep
            return torch.cat([self._run(graph, h[i * rows:(i + 1) * rows], 1, batch, row_weight, count)
                              for i in range(n_rep)], dim=0)
        return self._run(graph, h, n_rep, batch, row_weight, count)


def make_gnn_layers(gnn_cfg, input_dim: int, sizes: List[int]):
    if gnn_cfg.type != 'GAT':
        raise NotImplementedError(
            "gnn.type='GraphConv' hidden layers are outside the accelerated path "
            '(SURVEY.md section 8: the default GAT stack is the hot path)')
    layers = []
    for size in sizes:
        layers.append(GATConvAgg(input_dim, size, gnn_cfg.gat.num_attention_heads,
                                 gnn_cfg.use_batch_norm))
        input_dim = layers[-1].output_dim
    return input_dim, layers


# ---------------------------------------------------------------------------------------
# RealNVP prior (realnvp_flow.py:18-49): parameters here, arithmetic in csrc/flow.hip
# ---------------------------------------------------------------------------------------
class ActNorm(nn.Module):
    """tensorkit ActNorm: y = (x + bias) * exp(pre_scale); both are set from the first batch
    (`tv_realnvp_actnorm_init`, driven by `InverseRealNVP.log_prob`)."""

    def __init__(self, dim: int):
        super().__init__()
        self.pre_scale = nn.Parameter(torch.zeros(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.initialized = False

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.initialized = True


class _Branch(nn.Module):
    def __init__(self, n1: int, n2: int, hidden_count: int, hidden_units: int):
        super().__init__()
        shared, d = [], n1
        for _ in range(hidden_count):
            shared.append(Linear(d, hidden_units))
            d = hidden_units
        self.shared = nn.ModuleList(shared)
        self.branches = nn.ModuleList([
            nn.ModuleList([Linear(d, n2, zero_weight=True)]),   # shift
            nn.ModuleList([Linear(d, n2, zero_weight=True)]),   # pre_scale
        ])

    def forward(self, x1):
        """shift and pre-scale of a few rows (only `InverseRealNVP.sample` comes here: drawing
        from the prior runs the flow backwards, off the training / scoring path)."""
        h = x1
        for lin in self.shared:
            h = F.leaky_relu(F.linear(h, lin.weight_store.value, lin.bias_store.value), LRELU_SLOPE)
        outs = []
        for br in self.branches:
            lin = br[0]
            outs.append(F.linear(h, lin.weight_store.value, lin.bias_store.value))
        return outs


class CouplingLayer(nn.Module):
    def __init__(self, dim: int, cfg):
        super().__init__()
        self.n1 = dim // 2
        self.scale_type = cfg.coupling_layer_scale
        self.shift_and_pre_scale = _Branch(self.n1, dim - self.n1, cfg.coupling_hidden_layer_count,
                                           cfg.coupling_hidden_layer_units)


class InvertibleDense(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        q, _ = torch.linalg.qr(torch.randn(dim, dim))
        self.kernel = nn.Parameter(q.contiguous())


class _SequentialFlow(nn.Module):
    def __init__(self, flows):
        super().__init__()
        self.flows = nn.ModuleList(flows)


class InverseRealNVP(nn.Module):
    """`make_realnvp_flow(z_dim, cfg).invert()` used as a prior: `log_prob(z)` runs the
    original flow forward (realnvp_flow.py:18-49, struct_vae.py:148-149, 251-254).  One
    implementation: `tv_realnvp_logprob_*` (and `tv_realnvp_actnorm_init` for the first batch);
    a flow shape those kernels do not cover is refused when the model is built."""

    def __init__(self, dim: int, cfg):
        super().__init__()
        if cfg.coupling_layer_scale != 'sigmoid' or cfg.coupling_hidden_layer_count != 1 or \
                not 1 <= cfg.flow_levels <= 8 or \
                not ops.realnvp_supported(dim, cfg.coupling_hidden_layer_units, cfg.flow_levels):
            raise NotImplementedError(
                f'RealNVP prior with dim {dim}, {cfg.flow_levels} levels, '
                f'{cfg.coupling_hidden_layer_count} hidden layer(s) of '
                f'{cfg.coupling_hidden_layer_units} units and {cfg.coupling_layer_scale!r} scale: '
                f'the accelerated path covers dim 10 / 16 / 20, 1-8 levels, ONE hidden layer of '
                f'64 units and the sigmoid scale (the train.sh configuration)')
        flows = []
        for _ in range(cfg.flow_levels):
            flows += [ActNorm(dim), CouplingLayer(dim, cfg), InvertibleDense(dim)]
        self.original_flow = _SequentialFlow(flows)
        self.levels = cfg.flow_levels
        self.hidden = cfg.coupling_hidden_layer_units

    def _kernel_params(self):
        """The 9 tensors per level in the order of `tv_realnvp_logprob_fwd`."""
        flows = list(self.original_flow.flows)
        params = []
        for lvl in range(self.levels):
            an, cp, dense = flows[3 * lvl:3 * lvl + 3]
            br = cp.shift_and_pre_scale
            params += [an.bias, an.pre_scale, br.shared[0].weight_store.value,
                       br.shared[0].bias_store.value,
                       br.branches[0][0].weight_store.value, br.branches[0][0].bias_store.value,
                       br.branches[1][0].weight_store.value, br.branches[1][0].bias_store.value,
                       dense.kernel]
        return params

    def log_prob(self, z: torch.Tensor) -> torch.Tensor:
        params = self._kernel_params()
        act_norms = [m for m in self.original_flow.flows if isinstance(m, ActNorm)]
        if not all(m.initialized for m in act_norms):
            # the first batch sets bias / pre_scale of every level (train.py:188-199)
            with torch.no_grad():
                ops.realnvp_actnorm_init(z, params, self.levels, self.hidden, LRELU_SLOPE)
            for m in act_norms:
                m.initialized = True
        return ops.realnvp_log_prob(z, params, self.levels, self.hidden, LRELU_SLOPE)

    def sample(self, eps: torch.Tensor) -> torch.Tensor:
        """z = flow(eps) of `FlowDistribution(UnitNormal, original_flow.invert())`
        (struct_vae.py:251-254): the original flow run backwards, level by level.  A few
        10-dimensional rows per prior-sampling call (section 8f row 4, not the training /
        scoring path): tensor ops, the dense inverse in float64."""
        lead = eps.shape[:-1]
        x = eps.reshape(-1, eps.shape[-1])
        for flow in reversed(list(self.original_flow.flows)):
            if isinstance(flow, InvertibleDense):
                inv = torch.linalg.inv(flow.kernel.double()).to(x.dtype)
                x = x @ inv.t()
            elif isinstance(flow, CouplingLayer):
                y1, y2 = x[:, :flow.n1], x[:, flow.n1:]
                shift, pre = flow.shift_and_pre_scale(y1)
                x = torch.cat([y1, y2 * torch.exp(-F.logsigmoid(pre + 2.0)) - shift], dim=-1)
            else:
                if not flow.initialized:
                    raise RuntimeError('the prior flow has not been initialised: train or '
                                       'load a model before sampling from it')
                x = x * torch.exp(-flow.pre_scale) - flow.bias
        return x.reshape(lead + (x.shape[-1],))


# ---------------------------------------------------------------------------------------
# the two VAEs
# ---------------------------------------------------------------------------------------
def _sample_eps(net: BayesianNet, name: str, n_z: Optional[int], batch: int, dim: int, device):
    eps = net.meta.get('eps', {}).get(name) if net.meta.get('eps') else None
    shape = (batch, dim) if n_z is None else (n_z, batch, dim)
    if eps is None:
        return torch.randn(shape, device=device, dtype=torch.float32)
    eps = torch.as_tensor(eps, dtype=torch.float32, device=device)
    if tuple(eps.shape) != shape:
        raise ValueError(f'eps[{name!r}] has shape {tuple(eps.shape)}, expected {shape}')
    return eps.contiguous()


def _latent_node(net, name, mean, logstd_raw, lo, hi, n_z):
    """`net.add(name, tk.Normal(mean, logstd), n_samples=n_z)` with a reparameterised draw,
    or the log-density of an observed value."""
    if name in net.observed:
        x = net.observed[name]
        logstd = logstd_raw.clamp(min=lo, max=hi) if (lo is not None or hi is not None) else logstd_raw
        lp = ops.normal_ll(x, mean.expand_as(x).contiguous(), logstd_raw.expand_as(x).contiguous(),
                           lo, hi)
        node = StochasticTensor(x, Distribution('Normal', mean=mean, logstd=logstd,
                                                std=lambda: torch.exp(logstd)), lp, n_z)
    else:
        eps = _sample_eps(net, name, n_z, mean.shape[0], mean.shape[1], mean.device)
        z, logq, logstd = ops.normal_reparam(mean, logstd_raw, eps, lo, hi)
        node = StochasticTensor(z, Distribution('Normal', mean=mean, logstd=logstd,
                                                std=lambda: torch.exp(logstd)), logq, n_z)
    return net.add_node(name, node)


class TraceStructVAE(nn.Module):
    def __init__(self, config: TraceStructVAEConfig, operation_embedding: OperationEmbedding,
                 max_node_count: int):
        super().__init__()
        self.config = config
        self.max_node_count = int(max_node_count)
        self.operation_embedding = operation_embedding
        self.num_operations = operation_embedding.num_operations
        enc, dec = config.encoder, config.decoder
        if enc.pool_type != 'avg':
            raise NotImplementedError('only AVG pooling is on the accelerated path')
        if enc.use_posterior_flow or dec.z_as_context:
            raise NotImplementedError('posterior flow / z_as_context are non-default and '
                                      'outside the accelerated path')
        out, layers = make_gnn_layers(enc.gnn, operation_embedding.embedding_dim, enc.gnn_layers)
        self.qz_gnn_layers = GNNSequential(layers, _Pooling())
        self.qz_mean = Linear(out, config.z_dim)
        self.qz_logstd = Linear(out, config.z_dim)
        if dec.use_prior_flow:
            self.pz_flow = InverseRealNVP(config.z_dim, config.realnvp)
        bn = config.use_batch_norm
        self.pG_node_count_logits = MLP(config.z_dim, dec.node_count_layers, bn,
                                        self.max_node_count + 1)
        self.pG_graph_embedding = MLP(config.z_dim, dec.graph_embedding_layers, bn)
        self.pG_node_embedding = Linear(self.pG_graph_embedding.out_features,
                                        self.max_node_count * dec.latent_embedding_size)
        out, layers = make_gnn_layers(dec.gnn, dec.latent_embedding_size, dec.gnn_layers)
        self.pG_node_type_logits = GNNSequential(layers, GraphConv(out, self.num_operations))

    def q(self, net: BayesianNet, batch: DeviceBatch, n_z: Optional[int] = None):
        enc = self.config.encoder
        h = ops.embedding(self.operation_embedding.weight, batch.node_type)
        h = self.qz_gnn_layers(batch.enc, h, 1, batch)
        _latent_node(net, 'z', self.qz_mean(h), self.qz_logstd(h), enc.z_logstd_min,
                     enc.z_logstd_max, n_z)

    def p(self, net: BayesianNet, batch: Optional[DeviceBatch] = None, n_z: Optional[int] = None,
          use_biased: bool = False):
        cfg, M = self.config, self.max_node_count
        dec = cfg.decoder
        if batch is None:
            return self._p_sampled(net, n_z, use_biased)
        if 'z' not in net.observed:
            raise ValueError('p(G=...) needs the observed z of q(): chain it as q.chain(p, G=G)')
        z = net.observed['z']
        lead = z.shape[:-1]
        n_rep = z.shape[0] if z.dim() == 3 else 1
        if z.shape[-2] != batch.batch_size:
            raise ValueError('z does not match the batch size')
        z2d = z.reshape(-1, z.shape[-1])

        # p(z)
        if dec.use_prior_flow:
            log_pz = self.pz_flow.log_prob(z)
            pz = Distribution('FlowDistribution', base_distribution=Distribution('UnitNormal'))
        else:
            zero = torch.zeros_like(z)
            log_pz = ops.normal_ll(z, zero, zero)
            pz = Distribution('UnitNormal')
        net.add_node('z', StochasticTensor(z, pz, log_pz, n_z))

        # p(node_count | z)
        nc_logits = self.pG_node_count_logits(z2d)
        lp = ops.categorical_ll(nc_logits, batch.node_count, use_biased, float(M * M), 0.5)
        nc_logits_v = nc_logits.reshape(lead + (M + 1,))
        net.add_node('node_count', StochasticTensor(
            batch.node_count,
            Distribution('BiasedCategorical' if use_biased else 'Categorical', logits=nc_logits_v,
                         probs=lambda: torch.softmax(nc_logits_v, dim=-1)),
            lp.reshape(lead)))

        # p(adj | z): outer dot product of the node embeddings, upper triangle
        h = self.pG_node_embedding(self.pG_graph_embedding(z2d))
        h = h.reshape(-1, M, dec.latent_embedding_size)
        lp = ops.adj_bernoulli_ll(h, batch.adj, n_rep, use_biased, float(M), 0.5)

        def edge_logits():
            return ops.edge_logits_triu(h).reshape(lead + (-1,))

        net.add_node('adj', StochasticTensor(
            batch.adj,
            Distribution('BiasedBernoulli' if use_biased else 'Bernoulli', logits=edge_logits,
                         probs=lambda: torch.sigmoid(edge_logits())),
            lp.reshape(lead)))
        net.meta['g'] = batch.dec

        # p(node_type | e, adj): GNN over the padded graphs, all M slots count
        logits = self.pG_node_type_logits(batch.dec, h.reshape(-1, dec.latent_embedding_size), n_rep)
        lp_rows = ops.categorical_ll(logits, batch.node_type_pad.reshape(-1))
        lp = lp_rows.reshape(-1, M).sum(-1)
        logits_v = logits.reshape(lead + (M, self.num_operations))
        net.add_node('node_type', StochasticTensor(
            batch.node_type_pad, Distribution('Categorical', logits=logits_v), lp.reshape(lead)))


    def _p_sampled(self, net: BayesianNet, n_z: Optional[int], use_biased: bool):
        """`p()` without an observed graph (struct_vae.py:243-333, 394-414): every variable
        that is not in `net.observed` is drawn, the decoder graphs come from the drawn
        adjacency (`tv_sampled_graph_*`).  Shapes follow the reference: z `[n_z, 1, Z]` (or
        `[1, Z]`), node_count `lead`, adj `lead + [T]`, node_type `lead + [M]`."""
        cfg, M = self.config, self.max_node_count
        dec = cfg.decoder
        obs = net.observed
        dev = self.pG_node_embedding.weight_store.value.device
        if dev.type != 'cuda':
            raise hip.HipLibraryError('the model lives on the CPU: the accelerated path runs on '
                                      'the GPU only (there is no CPU fallback)')

        # z ~ p(z)
        if 'z' in obs:
            z = torch.as_tensor(obs['z'], dtype=torch.float32, device=dev)
        else:
            eps = _sample_eps(net, 'z', n_z, 1, cfg.z_dim, dev)
            z = self.pz_flow.sample(eps) if dec.use_prior_flow else eps
        z = z.contiguous()
        if dec.use_prior_flow:
            log_pz = self.pz_flow.log_prob(z)
            pz = Distribution('FlowDistribution', base_distribution=Distribution('UnitNormal'))
        else:
            zero = torch.zeros_like(z)
            log_pz = ops.normal_ll(z, zero, zero)
            pz = Distribution('UnitNormal')
        net.add_node('z', StochasticTensor(z, pz, log_pz, n_z))
        lead = tuple(z.shape[:-1])
        z2d = z.reshape(-1, z.shape[-1])
        graphs = z2d.shape[0]

        def draw(name, make):
            if name in obs:
                return torch.as_tensor(obs[name], device=dev)
            given = (net.meta.get('draws') or {}).get(name)
            return torch.as_tensor(given, device=dev) if given is not None else make()

        # node_count ~ p(node_count | z)
        nc_logits = self.pG_node_count_logits(z2d)
        node_count = draw('node_count', lambda: torch.multinomial(
            torch.softmax(nc_logits, dim=-1), 1)).reshape(graphs).to(torch.int64).contiguous()
        lp = ops.categorical_ll(nc_logits, node_count, use_biased, float(M * M), 0.5)
        nc_logits_v = nc_logits.reshape(lead + (M + 1,))
        net.add_node('node_count', StochasticTensor(
            node_count.reshape(lead),
            Distribution('BiasedCategorical' if use_biased else 'Categorical', logits=nc_logits_v,
                         probs=lambda: torch.softmax(nc_logits_v, dim=-1)),
            lp.reshape(lead)))

        # adj ~ p(adj | z)
        h = self.pG_node_embedding(self.pG_graph_embedding(z2d))
        h = h.reshape(-1, M, dec.latent_embedding_size)
        edge_logits = ops.edge_logits_triu(h)
        adj = draw('adj', lambda: torch.bernoulli(torch.sigmoid(edge_logits)))
        adj = adj.reshape(graphs, -1).to(torch.float32).contiguous()
        lp = ops.adj_bernoulli_ll(h, adj, 1, use_biased, float(M), 0.5)
        edge_logits_v = edge_logits.reshape(lead + (-1,))
        net.add_node('adj', StochasticTensor(
            adj.reshape(lead + (-1,)),
            Distribution('BiasedBernoulli' if use_biased else 'Bernoulli', logits=edge_logits_v,
                         probs=lambda: torch.sigmoid(edge_logits_v)),
            lp.reshape(lead)))

        # the decoder graphs of the drawn adjacencies (make_graph, struct_vae.py:298-331)
        graph, node_idx = build_sampled_graphs(adj, M)
        net.meta['g'] = graph
        net.meta['node_idx'] = node_idx

        # node_type ~ p(node_type | e, adj)
        logits = self.pG_node_type_logits(graph, h.reshape(-1, dec.latent_embedding_size), 1)
        node_type = draw('node_type', lambda: torch.multinomial(
            torch.softmax(logits, dim=-1), 1)).reshape(graphs, M).to(torch.int64).contiguous()
        lp = ops.categorical_ll(logits, node_type.reshape(-1)).reshape(-1, M).sum(-1)
        logits_v = logits.reshape(lead + (M, self.num_operations))
        net.add_node('node_type', StochasticTensor(
            node_type.reshape(lead + (M,)), Distribution('Categorical', logits=logits_v),
            lp.reshape(lead)))


class TraceLatencyVAE(nn.Module):
    def __init__(self, config: TraceLatencyVAEConfig, z_dim: int,
                 operation_embedding: OperationEmbedding, max_node_count: int):
        super().__init__()
        self.config = config
        self.z_dim = z_dim
        self.max_node_count = int(max_node_count)
        self.operation_embedding = operation_embedding
        self.num_node_types = operation_embedding.num_operations
        enc, dec = config.encoder, config.decoder
        if not config.use_operation_embedding or enc.pool_type != 'avg' or enc.use_posterior_flow \
                or dec.use_prior_flow or dec.z2_prior_mixtures != 1:
            raise NotImplementedError('non-default latency_vae options are outside the '
                                      'accelerated path (SURVEY.md section 8)')
        emb = operation_embedding.embedding_dim
        out, layers = make_gnn_layers(enc.gnn, emb + C.LATENCY_DIM, enc.gnn_layers)
        self.qz2_gnn_layers = GNNSequential(layers, _Pooling())
        self.qz2_mean = Linear(out, config.z2_dim)
        self.qz2_logstd = Linear(out, config.z2_dim)
        if dec.condition_on_z:
            self.z2_given_z_hidden_layers = MLP(z_dim, dec.z2_given_z_hidden_layers,
                                                config.use_batch_norm)
            hid = self.z2_given_z_hidden_layers.out_features
            self.pz2_mean = nn.ModuleList([Linear(hid, config.z2_dim)])
            self.pz2_logstd = nn.ModuleList([Linear(hid, config.z2_dim)])
        out, layers = make_gnn_layers(dec.gnn, config.z2_dim + emb, dec.gnn_layers)
        self.pG_node_features = GNNSequential(layers, GraphConv(out, 2 * C.LATENCY_DIM))

    def q(self, net: BayesianNet, batch: DeviceBatch, n_z: Optional[int] = None):
        enc = self.config.encoder
        emb = self.operation_embedding.embedding_dim
        if enc.operation_embedding_stop_grad:
            with torch.no_grad():
                h = ops.embedding(self.operation_embedding.weight, batch.node_type, C.LATENCY_DIM)
                h[:, emb:] = batch.latency[:, :C.LATENCY_DIM]
        else:
            h = torch.cat([ops.embedding(self.operation_embedding.weight, batch.node_type),
                           batch.latency[:, :C.LATENCY_DIM]], dim=-1)
        h = self.qz2_gnn_layers(batch.enc, h, 1, batch)
        _latent_node(net, 'z2', self.qz2_mean(h), self.qz2_logstd(h), enc.z2_logstd_min,
                     enc.z2_logstd_max, n_z)

    def p(self, net: BayesianNet, batch: Optional[DeviceBatch], n_z: Optional[int] = None,
          use_biased: bool = False, latency_logstd_min: Optional[float] = None,
          latency_log_prob_weight: bool = False, std_limit: Optional[torch.Tensor] = None):
        cfg, M = self.config, self.max_node_count
        dec = cfg.decoder
        if batch is None:
            return self._p_sampled(net, n_z, use_biased, latency_logstd_min,
                                   latency_log_prob_weight, std_limit)
        if 'z2' not in net.observed:
            raise ValueError('p(G=...) needs the observed z2 of q(): chain it as q.chain(p, G=G)')
        z2 = net.observed['z2']
        lead = z2.shape[:-1]
        n_rep = z2.shape[0] if z2.dim() == 3 else 1

        # p(z2 | z)
        if dec.condition_on_z:
            hz = net['z'].tensor
            if dec.z2_given_z_stop_grad:
                hz = hz.detach()
            hz = self.z2_given_z_hidden_layers(hz)
            pz2_mean = self.pz2_mean[0](hz)
            pz2_logstd_raw = self.pz2_logstd[0](hz)
            lp = ops.normal_ll(z2, pz2_mean, pz2_logstd_raw, dec.z2_logstd_min, dec.z2_logstd_max)
            pz2 = Distribution('Normal', mean=pz2_mean,
                               logstd=lambda: _clip(pz2_logstd_raw, dec.z2_logstd_min, dec.z2_logstd_max))
        else:
            zero = torch.zeros_like(z2)
            lp = ops.normal_ll(z2, zero, zero)
            pz2 = Distribution('UnitNormal')
        net.add_node('z2', StochasticTensor(z2, pz2, lp, n_z))

        # node features: [z2 || stop_grad(emb[node_type])] through the GNN
        node_type = net['node_type'].tensor
        emb_w = self.operation_embedding.weight
        if not dec.operation_embedding_stop_grad:
            raise NotImplementedError('operation_embedding_stop_grad=False is non-default')
        # The padded slots of a trace all carry the same row [z2 || emb[0]] and only a self
        # loop, so they stay identical through every layer: when the observed node types are
        # the batch's own (teacher forcing, the only case on this path) and padded slots are
        # masked out of the likelihood, ONE representative row per trace stands for its
        # M - n padded slots, weighted accordingly in the BatchNorm statistics.  Same function
        # of the parameters, about half the rows.
        compact = COMPACT_LATENCY_DECODER and dec.use_latency_mask and \
            node_type is batch.node_type_pad and batch.batch_size > 0
        self.last_compact = bool(compact)
        if compact:
            h = ops.latdec_rows(z2.reshape(-1, z2.shape[-1]), emb_w.detach(), batch.lat_row_graph,
                                batch.lat_row_type, batch.node_offsets, n_rep)
            feats = self.pG_node_features(batch.lat, h, n_rep, None, batch.lat_row_weight,
                                          batch.batch_size * M)
        else:
            h = ops.latdec_input(z2.reshape(-1, z2.shape[-1]), emb_w.detach(), node_type, n_rep)
            feats = self.pG_node_features(batch.dec, h, n_rep)

        if latency_logstd_min is not None:
            if dec.latency_logstd_min is not None:
                latency_logstd_min = max(latency_logstd_min, dec.latency_logstd_min)
        else:
            latency_logstd_min = dec.latency_logstd_min

        if self.training:
            mode, thr, alpha = 'safe', dec.safe_normal_std_threshold, 1.0
        elif use_biased and dec.use_biased_latency:
            mode = 'anomaly' if dec.use_anomaly_detection_normal else 'biased'
            thr, alpha = dec.biased_normal_std_threshold, float(M)
        else:
            mode, thr, alpha = 'normal', 0.0, 1.0
        if std_limit is not None:
            std_limit = torch.as_tensor(std_limit, dtype=torch.float32, device=feats.device).contiguous()
        ll, std = ops.latency_ll(
            feats.contiguous(), batch.latency_pad, net['node_count'].tensor, node_type, n_rep,
            mode=mode, std_threshold=thr, alpha=alpha, bias_threshold=0.5,
            logstd_min=latency_logstd_min, std_limit=std_limit,
            use_mask=dec.use_latency_mask, weighted=latency_log_prob_weight,
            node_offsets=batch.node_offsets if compact else None)
        shape4 = lead + (M, C.LATENCY_DIM)
        if compact:
            feats_c, nb = feats, batch

            def padded_mean():
                # scatter the compact rows back to [.., B, M, 1] (off the hot path)
                n, b = nb.num_nodes, nb.batch_size
                slot = torch.arange(M, device=feats_c.device)[None, :]
                idx = torch.where(slot < nb.node_count[:, None], nb.node_offsets[:b, None] + slot,
                                  n + torch.arange(b, device=feats_c.device)[:, None])
                idx = idx[None] + (n + b) * torch.arange(n_rep, device=feats_c.device)[:, None, None]
                return feats_c[:, 0][idx.reshape(-1)].reshape(shape4)
        else:
            padded_mean = lambda: feats[:, :1].reshape(shape4)
        base = Distribution(
            {'safe': 'SafeNormal', 'biased': 'BiasedNormal', 'anomaly': 'AnomalyDetectionNormal',
             'normal': 'Normal'}[mode],
            mean=padded_mean, std=lambda: std.reshape(shape4),
            logstd=lambda: torch.log(std).reshape(shape4))
        dist = Distribution('MaskedDistribution', base_distribution=base) \
            if dec.use_latency_mask else base
        net.add_node('latency', StochasticTensor(batch.latency_pad, dist, ll.reshape(lead)))


    def _p_sampled(self, net: BayesianNet, n_z: Optional[int], use_biased: bool,
                   latency_logstd_min: Optional[float], latency_log_prob_weight: bool,
                   std_limit: Optional[torch.Tensor]):
        """`p()` on the graphs `struct_vae.p` drew (latency_vae.py:272-444 with `g =
        net.meta['g']`).  z2 is drawn ONCE per drawn z (shape `lead + [Z2]`): the reference
        passes `n_samples=n_z` again on a distribution whose batch shape already carries the
        n_z axis, which gives `[n_z, n_z, 1, Z2]` and no usable decoder input."""
        cfg, M = self.config, self.max_node_count
        dec = cfg.decoder
        obs = net.observed
        if not dec.operation_embedding_stop_grad:
            raise NotImplementedError('operation_embedding_stop_grad=False is non-default')
        z = net['z'].tensor
        lead = tuple(z.shape[:-1])
        dev = z.device
        graphs = z.numel() // z.shape[-1]

        # z2 ~ p(z2 | z)
        if dec.condition_on_z:
            hz = self.z2_given_z_hidden_layers(z.detach() if dec.z2_given_z_stop_grad else z)
            pz2_mean = self.pz2_mean[0](hz)
            pz2_logstd_raw = self.pz2_logstd[0](hz)
        else:
            pz2_mean = torch.zeros(lead + (cfg.z2_dim,), device=dev)
            pz2_logstd_raw = torch.zeros_like(pz2_mean)
        pz2_logstd = _clip(pz2_logstd_raw, dec.z2_logstd_min, dec.z2_logstd_max) \
            if dec.condition_on_z else pz2_logstd_raw
        if 'z2' in obs:
            z2 = torch.as_tensor(obs['z2'], dtype=torch.float32, device=dev).contiguous()
        else:
            eps = (net.meta.get('eps') or {}).get('z2')
            eps = torch.randn_like(pz2_mean) if eps is None else \
                torch.as_tensor(eps, dtype=torch.float32, device=dev).reshape(pz2_mean.shape)
            z2 = (pz2_mean + torch.exp(pz2_logstd) * eps).contiguous()
        if dec.condition_on_z:
            lp = ops.normal_ll(z2, pz2_mean.contiguous(), pz2_logstd_raw.contiguous(),
                               dec.z2_logstd_min, dec.z2_logstd_max)
            pz2 = Distribution('Normal', mean=pz2_mean, logstd=pz2_logstd)
        else:
            lp = ops.normal_ll(z2, pz2_mean, pz2_logstd_raw)
            pz2 = Distribution('UnitNormal')
        net.add_node('z2', StochasticTensor(z2, pz2, lp, n_z))

        # node features on the drawn graphs
        node_type = net['node_type'].tensor.reshape(graphs, M).contiguous()
        node_count = net['node_count'].tensor.reshape(graphs).contiguous()
        h = ops.latdec_input(z2.reshape(-1, z2.shape[-1]), self.operation_embedding.weight.detach(),
                             node_type, 1)
        feats = self.pG_node_features(net.meta['g'], h, 1).contiguous()
        self.last_compact = False

        if latency_logstd_min is not None:
            if dec.latency_logstd_min is not None:
                latency_logstd_min = max(latency_logstd_min, dec.latency_logstd_min)
        else:
            latency_logstd_min = dec.latency_logstd_min
        if self.training:
            mode, thr, alpha = 'safe', dec.safe_normal_std_threshold, 1.0
        elif use_biased and dec.use_biased_latency:
            mode = 'anomaly' if dec.use_anomaly_detection_normal else 'biased'
            thr, alpha = dec.biased_normal_std_threshold, float(M)
        else:
            mode, thr, alpha = 'normal', 0.0, 1.0
        if std_limit is not None:
            std_limit = torch.as_tensor(std_limit, dtype=torch.float32, device=dev).contiguous()

        # latency ~ Normal(mean, std), masked to the drawn node_count (distributions.py:80-103)
        shape4 = lead + (M, C.LATENCY_DIM)
        mean = feats[:, :1].reshape(graphs, M, 1)
        if 'latency' in obs:
            x = torch.as_tensor(obs['latency'], dtype=torch.float32, device=dev)
            x = x.reshape(graphs, M, 1).contiguous()
        else:
            logstd = feats[:, 1:2].reshape(graphs, M, 1)
            if latency_logstd_min is not None:
                logstd = logstd.clamp(min=latency_logstd_min)
            if std_limit is not None:
                logstd = torch.minimum(
                    logstd, torch.log(std_limit[node_type].clamp(min=1e-7)).unsqueeze(-1))
            eps = (net.meta.get('eps') or {}).get('latency')
            eps = torch.randn_like(mean) if eps is None else \
                torch.as_tensor(eps, dtype=torch.float32, device=dev).reshape(mean.shape)
            x = mean + torch.exp(logstd) * eps
            if dec.use_latency_mask:
                slot = torch.arange(M, device=dev)[None, :, None]
                x = x * (slot < node_count[:, None, None]).to(x.dtype)
            x = x.contiguous()
        ll, std = ops.latency_ll(
            feats, x, node_count, node_type, 1, mode=mode, std_threshold=thr, alpha=alpha,
            bias_threshold=0.5, logstd_min=latency_logstd_min, std_limit=std_limit,
            use_mask=dec.use_latency_mask, weighted=latency_log_prob_weight)
        base = Distribution(
            {'safe': 'SafeNormal', 'biased': 'BiasedNormal', 'anomaly': 'AnomalyDetectionNormal',
             'normal': 'Normal'}[mode],
            mean=lambda: mean.reshape(shape4), std=lambda: std.reshape(shape4),
            logstd=lambda: torch.log(std).reshape(shape4))
        dist = Distribution('MaskedDistribution', base_distribution=base) \
            if dec.use_latency_mask else base
        net.add_node('latency', StochasticTensor(x.reshape(shape4), dist, ll.reshape(lead)))


def _clip(x, lo, hi):
    if lo is None and hi is None:
        return x
    return x.clamp(min=lo, max=hi)


class TraceVAE(nn.Module):
    """`TraceVAE(config, num_operations)` (trace_vae.py:45).  `max_node_count` defaults to the
    MAX_NODE_COUNT constant, which is what the reference bakes into its layer shapes."""

    def __init__(self, config: TraceVAEConfig, num_operations: int,
                 max_node_count: Optional[int] = None):
        super().__init__()
        self.config = config
        self.num_operations = int(num_operations)
        self.max_node_count = int(max_node_count or C.MAX_NODE_COUNT)
        if config.arch != 'default':
            raise ValueError(f'Unsupported arch: {config.arch!r}')
        self.operation_embedding = OperationEmbedding(num_operations, config.operation_embedding_dim)
        self.struct_vae = TraceStructVAE(config.struct, self.operation_embedding,
                                         self.max_node_count)
        if config.use_latency:
            self.latency_vae = TraceLatencyVAE(config.latency, config.struct.z_dim,
                                               self.operation_embedding, self.max_node_count)

    def _call_graph_batch_build(self, G: TraceGraphBatch) -> DeviceBatch:
        if G._max_node_count is None:
            G._max_node_count = self.max_node_count
        batch = G.build_dgl(add_self_loop=True, directed=False)
        if batch.max_node_count != self.max_node_count:
            raise ValueError(f'batch was built for MAX_NODE_COUNT={batch.max_node_count}, '
                             f'the model for {self.max_node_count}')
        return batch

    def q(self, G: TraceGraphBatch, observed: Optional[Mapping[str, torch.Tensor]] = None,
          n_z: Optional[int] = None, no_latency: bool = False,
          eps: Optional[Mapping[str, torch.Tensor]] = None) -> BayesianNet:
        """q(z, z2 | G).  `eps` (an extension) supplies the N(0,1) draws, `{'z': ..., 'z2': ...}`,
        so that CPU oracle and GPU path can be compared on identical noise."""
        batch = self._call_graph_batch_build(G)
        net = BayesianNet(observed=observed)
        net.meta['eps'] = dict(eps) if eps else None
        self.struct_vae.q(net, batch, n_z=n_z)
        if self.config.use_latency and not no_latency:
            self.latency_vae.q(net, batch, n_z=n_z)
        return net

    def p(self, observed: Optional[Mapping[str, torch.Tensor]] = None,
          G: Optional[TraceGraphBatch] = None, n_z: Optional[int] = None, no_latency: bool = False,
          use_biased: bool = False, use_latency_biased: bool = False,
          latency_logstd_min: Optional[float] = None, latency_log_prob_weight: bool = False,
          std_limit: Optional[torch.Tensor] = None,
          eps: Optional[Mapping[str, torch.Tensor]] = None,
          draws: Optional[Mapping[str, torch.Tensor]] = None) -> BayesianNet:
        """p(...).  With `G`, every variable is observed (the chain of q).  Without it the
        variables not in `observed` are drawn (prior sampling, evaluation.py:349); `eps`
        (`z`, `z2`, `latency`: N(0,1) noise) and `draws` (`node_count`, `adj`, `node_type`:
        the discrete outcomes) are extensions that pin the random choices for parity tests."""
        batch = None
        observed = dict(observed or {})
        if G is not None:
            batch = self._call_graph_batch_build(G)
            observed['node_count'] = batch.node_count
            observed['adj'] = batch.adj
            observed['node_type'] = batch.node_type_pad
            observed['latency'] = batch.latency_pad[..., :C.LATENCY_DIM]
        net = BayesianNet(observed=observed)
        net.meta['eps'] = dict(eps) if eps else None
        net.meta['draws'] = dict(draws) if draws else None
        self.struct_vae.p(net, batch, n_z=n_z, use_biased=use_biased)
        if self.config.use_latency and not no_latency:
            self.latency_vae.p(net, batch, n_z=n_z, use_biased=use_biased and use_latency_biased,
                               latency_logstd_min=latency_logstd_min,
                               latency_log_prob_weight=latency_log_prob_weight,
                               std_limit=std_limit)
        return net
