"""CPU PyTorch restatement of the TraceVAE model path (TEST INFRASTRUCTURE ONLY).

Stock torch CPU ops only, float32 or float64.  Follows, function by function:

* GATConv / GraphConv / AvgPooling semantics      SURVEY.md appendix A1-A3 (DGL, un-vendored)
* `GATConvAgg.forward`                             model/gnn_layers.py:142-157
* `GraphConv.forward` (final decoder layer)        model/gnn_layers.py:182-190
* `apply_gnn_layer` (loop over the sample axis)    model/gnn_layers.py:69-99
* `TraceStructVAE.q / .p`                          model/struct_vae.py:213-241, 243-414
* `TraceLatencyVAE.q / .p`                         model/latency_vae.py:236-270, 272-444
* `TraceVAE.q / .p` (observed tensors)             model/trace_vae.py:84-161
* distributions                                    distributions.py:105-113, 125-134, 267-286, 300-315, 341-356
* RealNVP prior                                    model/realnvp_flow.py:18-49 + appendix B4
* `train_step` loss                                train.py:213-277
* `eval_step` (sgvb, is_loglikelihood, clip)       evaluation.py:115-148

Parameters come in as a flat `{state_dict key: tensor}` mapping (key layout: SURVEY.md
appendix C), random draws come in as explicit epsilon tensors, the batch comes in as
the arrays `oracle.batching.build_batch` produces.
"""
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

__all__ = ['OracleTraceVAE', 'edge_softmax', 'dense_to_triu']

LOG_2PI = math.log(2.0 * math.pi)


# ---------------------------------------------------------------------------------------
# graph primitives (DGL semantics)
# ---------------------------------------------------------------------------------------
def edge_softmax(e: torch.Tensor, dst: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """softmax of edge scores over the in-edges of each destination (max-subtracted)."""
    idx = dst.view(-1, *([1] * (e.dim() - 1))).expand_as(e)
    emax = torch.full((num_nodes,) + e.shape[1:], -float('inf'), dtype=e.dtype)
    emax = emax.scatter_reduce(0, idx, e, reduce='amax', include_self=True)
    ex = torch.exp(e - emax[dst])
    den = torch.zeros((num_nodes,) + e.shape[1:], dtype=e.dtype).index_add_(0, dst, ex)
    return ex / den[dst]


def gat_conv(h, src, dst, fc_weight, attn_l, attn_r, bias, heads, negative_slope=0.2):
    """dgl.nn.GATConv(in, out, heads, activation=None): appendix A1."""
    rows = h.shape[0]
    out_feats = fc_weight.shape[0] // heads
    ft = (h @ fc_weight.t()).view(rows, heads, out_feats)
    el = (ft * attn_l).sum(-1)
    er = (ft * attn_r).sum(-1)
    e = F.leaky_relu(el[src] + er[dst], negative_slope)
    a = edge_softmax(e, dst, rows)
    msg = ft[src] * a.unsqueeze(-1)
    rst = torch.zeros_like(ft).index_add_(0, dst, msg)
    rst = rst + bias.view(1, heads, out_feats)
    return rst


def graph_conv(h, src, dst, weight, bias):
    """dgl.nn.GraphConv(norm='both', weight=True, bias=True): appendix A2."""
    rows = h.shape[0]
    ones = torch.ones([len(src)], dtype=h.dtype)
    out_deg = torch.zeros([rows], dtype=h.dtype).index_add_(0, src, ones).clamp(min=1)
    in_deg = torch.zeros([rows], dtype=h.dtype).index_add_(0, dst, ones).clamp(min=1)
    feat = h * torch.pow(out_deg, -0.5).unsqueeze(-1)
    in_feats, out_feats = weight.shape
    if in_feats > out_feats:
        feat = feat @ weight
        rst = torch.zeros([rows, out_feats], dtype=h.dtype).index_add_(0, dst, feat[src])
    else:
        agg = torch.zeros([rows, in_feats], dtype=h.dtype).index_add_(0, dst, feat[src])
        rst = agg @ weight
    rst = rst * torch.pow(in_deg, -0.5).unsqueeze(-1)
    return rst + bias


def avg_pooling(h, node_offsets):
    """dgl.nn.AvgPooling: per-graph mean of node rows (appendix A3)."""
    outs = []
    for i in range(len(node_offsets) - 1):
        outs.append(h[int(node_offsets[i]):int(node_offsets[i + 1])].mean(dim=0))
    return torch.stack(outs, dim=0)


def dense_to_triu(x: torch.Tensor, node_count: int) -> torch.Tensor:
    """tensor_utils.py:64-67: strict upper triangle, row-major."""
    mask = torch.triu(torch.ones([node_count, node_count], dtype=torch.bool), 1).reshape(-1)
    return x.reshape(x.shape[:-2] + (-1,))[..., mask]


# ---------------------------------------------------------------------------------------
# log-likelihoods (tensorkit semantics, appendix B2; reference distributions.py)
# ---------------------------------------------------------------------------------------
def normal_log_pdf(x, mean, logstd):
    return -0.5 * LOG_2PI - logstd - 0.5 * (x - mean) ** 2 * torch.exp(-2.0 * logstd)


def bernoulli_log_prob(x, logits):
    return x * logits - F.softplus(logits)


def categorical_log_prob(x, logits):
    return torch.gather(F.log_softmax(logits, dim=-1), -1, x.unsqueeze(-1)).squeeze(-1)


def biased(log_prob, alpha, threshold):
    """distributions.py:125-134."""
    return torch.where(log_prob < math.log(threshold), log_prob * alpha, log_prob)


class OracleTraceVAE(object):
    """Functional model: every call takes the parameter mapping it should use."""

    def __init__(self, config, num_operations: int, max_node_count: int,
                 dtype=torch.float32, bn_eps: float = 1e-6, bn_momentum: float = 0.1):
        self.cfg = config
        self.V = int(num_operations)
        self.M = int(max_node_count)
        self.dtype = dtype
        self.bn_eps = bn_eps
        self.bn_momentum = bn_momentum
        s, l = config.struct, config.latency
        for sec in (s.encoder, s.decoder, l.encoder, l.decoder):
            if sec.gnn.type != 'GAT':
                raise NotImplementedError('oracle covers the default GAT layers only')
        if s.decoder.z_as_context or s.encoder.use_posterior_flow or l.encoder.use_posterior_flow \
                or l.decoder.use_prior_flow or l.decoder.z2_prior_mixtures != 1 \
                or not l.use_operation_embedding or s.encoder.pool_type != 'avg' \
                or l.encoder.pool_type != 'avg':
            raise NotImplementedError('configuration outside SURVEY.md section 8 defaults')

    # ---------------------------------------------------------------- building blocks
    def _t(self, x):
        return torch.as_tensor(x, dtype=self.dtype)

    def _bn(self, P, buf, prefix, x, training):
        rm, rv = buf[prefix + '.running_mean'], buf[prefix + '.running_var']
        if training:
            rm, rv = rm.clone(), rv.clone()
            y = F.batch_norm(x, rm, rv, P[prefix + '.weight'], P[prefix + '.bias'],
                             True, self.bn_momentum, self.bn_eps)
            buf[prefix + '.running_mean'], buf[prefix + '.running_var'] = rm, rv
            buf[prefix + '.num_batches_tracked'] = buf[prefix + '.num_batches_tracked'] + 1
            return y
        return F.batch_norm(x, rm, rv, P[prefix + '.weight'], P[prefix + '.bias'],
                            False, self.bn_momentum, self.bn_eps)

    def _gnn_stack_2d(self, P, buf, stack, gnn_cfg, n_layers, src, dst, h, training, tail):
        heads = gnn_cfg.gat.num_attention_heads
        for i in range(n_layers):
            p = f'{stack}.gnn.{i}'
            h = gat_conv(h, src, dst, P[p + '.gnn.fc.weight'], P[p + '.gnn.attn_l'],
                         P[p + '.gnn.attn_r'], P[p + '.gnn.bias'], heads)
            h = h.reshape(h.shape[0], -1)                      # head concat (gnn_layers.py:144-148)
            if gnn_cfg.use_batch_norm:
                h = self._bn(P, buf, p + '.normalization', h, training)
            h = F.leaky_relu(h, 0.01)
        return tail(h, f'{stack}.gnn.{n_layers}')

    def _gnn_stack(self, P, buf, stack, gnn_cfg, n_layers, src, dst, h, training, tail):
        # apply_gnn_layer: a 3-D input is a Python loop over axis 0 (gnn_layers.py:90-97)
        if h.dim() == 3:
            return torch.stack([
                self._gnn_stack_2d(P, buf, stack, gnn_cfg, n_layers, src, dst, h[i], training, tail)
                for i in range(h.shape[0])], dim=0)
        return self._gnn_stack_2d(P, buf, stack, gnn_cfg, n_layers, src, dst, h, training, tail)

    def _linear(self, P, name, x):
        return x @ P[name + '.weight_store.value'].t() + P[name + '.bias_store.value']

    def _mlp(self, P, buf, name, x, n_dense, use_bn, training, final_linear):
        """tk SequentialBuilder: dense = Linear(bias iff no normalizer) -> BatchNorm ->
        LeakyReLU(0.01); `.build(flatten_to_ndims=True)` feeds 2-D rows (appendix B3)."""
        lead = x.shape[:-1]
        h = x.reshape(-1, x.shape[-1])
        for i in range(n_dense):
            p = f'{name}.{i}'
            h = h @ P[p + '.weight_store.value'].t()
            if use_bn:
                h = self._bn(P, buf, p + '.normalizer', h, training)
            else:
                h = h + P[p + '.bias_store.value']
            h = F.leaky_relu(h, 0.01)
        if final_linear:
            h = self._linear(P, f'{name}.{n_dense}', h)
        return h.reshape(lead + (h.shape[-1],))

    def _flow_log_prob(self, P, name, z):
        """log p(z) under FlowDistribution(UnitNormal, flow.invert()) (appendix B4)."""
        cfg = self.cfg.struct.realnvp
        zdim = z.shape[-1]
        n1 = zdim // 2
        lead = z.shape[:-1]
        x = z.reshape(-1, zdim)
        logdet = torch.zeros([x.shape[0]], dtype=z.dtype)
        for lvl in range(cfg.flow_levels):
            a, c, d = (f'{name}.original_flow.flows.{3 * lvl + k}' for k in range(3))
            pre_scale, bias = P[a + '.pre_scale'], P[a + '.bias']
            x = (x + bias) * torch.exp(pre_scale)
            logdet = logdet + pre_scale.sum()
            x1, x2 = x[:, :n1], x[:, n1:]
            hdn = x1
            for j in range(cfg.coupling_hidden_layer_count):
                s = f'{c}.shift_and_pre_scale.shared.{j}'
                hdn = F.leaky_relu(hdn @ P[s + '.weight_store.value'].t() + P[s + '.bias_store.value'], 0.01)
            b0, b1 = (f'{c}.shift_and_pre_scale.branches.{k}.0' for k in range(2))
            shift = hdn @ P[b0 + '.weight_store.value'].t() + P[b0 + '.bias_store.value']
            pre = hdn @ P[b1 + '.weight_store.value'].t() + P[b1 + '.bias_store.value']
            if cfg.coupling_layer_scale == 'sigmoid':
                log_scale = F.logsigmoid(pre + 2.0)
            elif cfg.coupling_layer_scale == 'exp':
                log_scale = pre
            else:
                raise NotImplementedError(cfg.coupling_layer_scale)
            x2 = (x2 + shift) * torch.exp(log_scale)
            logdet = logdet + log_scale.sum(-1)
            x = torch.cat([x1, x2], dim=-1)
            w = P[d + '.kernel']
            x = x @ w.t()
            logdet = logdet + torch.linalg.slogdet(w)[1]
        base = (-0.5 * LOG_2PI - 0.5 * x ** 2).sum(-1)
        return (base + logdet).reshape(lead)

    def _flow_sample(self, P, name, eps):
        """z = flow(eps) of FlowDistribution(UnitNormal, flow.invert()) (struct_vae.py:251-254):
        the levels of `_flow_log_prob` undone in reverse order."""
        cfg = self.cfg.struct.realnvp
        zdim = eps.shape[-1]
        n1 = zdim // 2
        lead = eps.shape[:-1]
        x = eps.reshape(-1, zdim)
        for lvl in reversed(range(cfg.flow_levels)):
            a, c, d = (f'{name}.original_flow.flows.{3 * lvl + k}' for k in range(3))
            w = P[d + '.kernel']
            x = torch.linalg.solve(w.double(), x.double().t()).t().to(x.dtype)
            y1, y2 = x[:, :n1], x[:, n1:]
            hdn = y1
            for j in range(cfg.coupling_hidden_layer_count):
                s = f'{c}.shift_and_pre_scale.shared.{j}'
                hdn = F.leaky_relu(hdn @ P[s + '.weight_store.value'].t() + P[s + '.bias_store.value'], 0.01)
            b0, b1 = (f'{c}.shift_and_pre_scale.branches.{k}.0' for k in range(2))
            shift = hdn @ P[b0 + '.weight_store.value'].t() + P[b0 + '.bias_store.value']
            pre = hdn @ P[b1 + '.weight_store.value'].t() + P[b1 + '.bias_store.value']
            if cfg.coupling_layer_scale == 'sigmoid':
                log_scale = F.logsigmoid(pre + 2.0)
            elif cfg.coupling_layer_scale == 'exp':
                log_scale = pre
            else:
                raise NotImplementedError(cfg.coupling_layer_scale)
            x = torch.cat([y1, y2 * torch.exp(-log_scale) - shift], dim=-1)
            x = x * torch.exp(-P[a + '.pre_scale']) - P[a + '.bias']
        return x.reshape(lead + (zdim,))

    # ------------------------------------------------------------------------- q / p
    def q(self, P, buf, batch, eps_z, eps_z2, training, no_latency=False):
        """TraceVAE.q: returns z, z2 and log q(z), log q(z2).  `training` is a bool or a
        `(struct_vae mode, latency_vae mode)` pair (phase 2 of training mixes them)."""
        cfg = self.cfg
        train_s, train_l = _modes(training)
        src, dst = torch.as_tensor(batch['enc_src']), torch.as_tensor(batch['enc_dst'])
        node_type = torch.as_tensor(batch['node_type'])
        off = batch['node_offsets']
        emb_w = P['operation_embedding.node_embedding.weight']
        pool = lambda h, _name: avg_pooling(h, off)
        out = {}

        # struct_vae.q (struct_vae.py:213-241)
        se = cfg.struct.encoder
        h = emb_w[node_type]
        h = self._gnn_stack(P, buf, 'struct_vae.qz_gnn_layers', se.gnn, len(se.gnn_layers),
                            src, dst, h, train_s, pool)
        z_mean = self._linear(P, 'struct_vae.qz_mean', h)
        z_logstd = self._linear(P, 'struct_vae.qz_logstd', h)
        z_logstd = _clip(z_logstd, se.z_logstd_min, se.z_logstd_max)
        z = z_mean + torch.exp(z_logstd) * eps_z
        out.update(z=z, z_mean=z_mean, z_logstd=z_logstd,
                   log_q_z=normal_log_pdf(z, z_mean, z_logstd).sum(-1))

        # latency_vae.q (latency_vae.py:236-270)
        if cfg.use_latency and not no_latency:
            le = cfg.latency.encoder
            h2 = emb_w[node_type]
            if le.operation_embedding_stop_grad:
                h2 = h2.detach()
            lat = self._t(batch['latency'])[:, :1]
            h = torch.cat([h2, lat], dim=-1)
            h = self._gnn_stack(P, buf, 'latency_vae.qz2_gnn_layers', le.gnn, len(le.gnn_layers),
                                src, dst, h, train_l, pool)
            z2_mean = self._linear(P, 'latency_vae.qz2_mean', h)
            z2_logstd = _clip(self._linear(P, 'latency_vae.qz2_logstd', h),
                              le.z2_logstd_min, le.z2_logstd_max)
            z2 = z2_mean + torch.exp(z2_logstd) * eps_z2
            out.update(z2=z2, z2_mean=z2_mean, z2_logstd=z2_logstd,
                       log_q_z2=normal_log_pdf(z2, z2_mean, z2_logstd).sum(-1))
        return out

    def p(self, P, buf, batch, z, z2, training, no_latency=False, use_biased=False,
          use_latency_biased=False, latency_logstd_min=None, latency_log_prob_weight=False,
          std_limit=None):
        """TraceVAE.p with every variable observed (the chain `q.chain(p, G=G)`)."""
        cfg, M, V = self.cfg, self.M, self.V
        train_s, train_l = _modes(training)
        sd = cfg.struct.decoder
        src, dst = torch.as_tensor(batch['dec_src']), torch.as_tensor(batch['dec_dst'])
        node_count = torch.as_tensor(batch['node_count'])
        adj = self._t(batch['adj'])
        node_type_pad = torch.as_tensor(batch['node_type_pad'])
        out = {}

        # p(z) (struct_vae.py:251-254)
        if sd.use_prior_flow:
            out['log_p_z'] = self._flow_log_prob(P, 'struct_vae.pz_flow', z)
        else:
            out['log_p_z'] = (-0.5 * LOG_2PI - 0.5 * z ** 2).sum(-1)

        # p(node_count|z) (struct_vae.py:257-266)
        use_bn = cfg.struct.use_batch_norm
        nc_logits = self._mlp(P, buf, 'struct_vae.pG_node_count_logits.layer', z,
                              len(sd.node_count_layers), use_bn, train_s, True)
        lp = categorical_log_prob(node_count.expand(nc_logits.shape[:-1]), nc_logits)
        if use_biased:
            lp = biased(lp, M * M, 0.5)
        out['log_p_node_count'] = lp
        out['node_count_logits'] = nc_logits

        # p(adj|z) (struct_vae.py:269-294)
        h = self._mlp(P, buf, 'struct_vae.pG_graph_embedding.layer', z,
                      len(sd.graph_embedding_layers), use_bn, train_s, False)
        h = self._linear(P, 'struct_vae.pG_node_embedding', h)
        h = h.reshape(h.shape[:-1] + (M, sd.latent_embedding_size))
        edge_logits = dense_to_triu(h @ h.transpose(-1, -2), M)
        lp = bernoulli_log_prob(adj, edge_logits)
        if use_biased:
            lp = biased(lp, M, 0.5)
        out['log_p_adj'] = lp.sum(-1)
        out['edge_logits'] = edge_logits

        # p(node_type|e, adj) (struct_vae.py:394-414): all M positions count, padded target 0
        h_shape = h.shape
        hh = h.reshape(h_shape[:-3] + (h_shape[-3] * h_shape[-2], h_shape[-1]))
        tail = lambda x, name: graph_conv(x, src, dst, P[name + '.gnn.weight'], P[name + '.gnn.bias'])
        logits = self._gnn_stack(P, buf, 'struct_vae.pG_node_type_logits', sd.gnn,
                                 len(sd.gnn_layers), src, dst, hh, train_s, tail)
        logits = logits.reshape(h_shape[:-1] + (V,))
        lp = categorical_log_prob(node_type_pad.expand(logits.shape[:-1]), logits)
        out['log_p_node_type'] = lp.sum(-1)
        out['node_type_logits'] = logits

        if not (cfg.use_latency and not no_latency):
            return out

        # p(z2|z) (latency_vae.py:284-318)
        ld = cfg.latency.decoder
        if ld.condition_on_z:
            hz = z.detach() if ld.z2_given_z_stop_grad else z
            hz = self._mlp(P, buf, 'latency_vae.z2_given_z_hidden_layers.layer', hz,
                           len(ld.z2_given_z_hidden_layers), cfg.latency.use_batch_norm, train_l, False)
            pz2_mean = self._linear(P, 'latency_vae.pz2_mean.0', hz)
            pz2_logstd = _clip(self._linear(P, 'latency_vae.pz2_logstd.0', hz),
                               ld.z2_logstd_min, ld.z2_logstd_max)
            out['log_p_z2'] = normal_log_pdf(z2, pz2_mean, pz2_logstd).sum(-1)
        else:
            out['log_p_z2'] = (-0.5 * LOG_2PI - 0.5 * z2 ** 2).sum(-1)

        # node features (latency_vae.py:320-344)
        emb_w = P['operation_embedding.node_embedding.weight']
        h2 = emb_w[node_type_pad]
        if ld.operation_embedding_stop_grad:
            h2 = h2.detach()
        hz2 = z2.unsqueeze(-2)                                          # [..., B, 1, Z2]
        lead = hz2.shape[:-2]
        h = torch.cat([hz2.expand(lead + (M, z2.shape[-1])),
                       h2.expand(lead + (M, h2.shape[-1]))], dim=-1)    # [..., B, M, Z2+E]
        h_shape = h.shape
        hh = h.reshape(h_shape[:-3] + (h_shape[-3] * h_shape[-2], h_shape[-1]))
        feats = self._gnn_stack(P, buf, 'latency_vae.pG_node_features', ld.gnn,
                                len(ld.gnn_layers), src, dst, hh, train_l, tail)

        # mean / logstd (latency_vae.py:346-377)
        if latency_logstd_min is not None:
            if ld.latency_logstd_min is not None:
                latency_logstd_min = max(latency_logstd_min, ld.latency_logstd_min)
        else:
            latency_logstd_min = ld.latency_logstd_min
        lat_mean = feats[..., :1].reshape(h_shape[:-1] + (1,))
        lat_logstd = feats[..., 1:2].reshape(h_shape[:-1] + (1,))
        if latency_logstd_min is not None:
            lat_logstd = lat_logstd.clamp(min=latency_logstd_min)
        if std_limit is not None:
            limit = torch.log(self._t(std_limit)[node_type_pad].clamp(min=1e-7)).unsqueeze(-1)
            lat_logstd = torch.minimum(lat_logstd, limit)
        out['latency_mean'], out['latency_logstd'] = lat_mean, lat_logstd

        # p(latency|z2, G) (latency_vae.py:384-444)
        x = self._t(batch['latency_pad'])
        lat_std = torch.exp(lat_logstd)
        if train_l:
            thr = ld.safe_normal_std_threshold                          # SafeNormal
            lo = (lat_mean - thr * lat_std).detach()
            hi = (lat_mean + thr * lat_std).detach()
            lp = normal_log_pdf(torch.maximum(torch.minimum(x, hi), lo), lat_mean, lat_logstd)
        elif use_biased and use_latency_biased and ld.use_biased_latency:
            thr = ld.biased_normal_std_threshold
            if ld.use_anomaly_detection_normal:                         # AnomalyDetectionNormal
                t = torch.abs((x - lat_mean) / lat_std) - thr
                lp = -math.log(2.0) + torch.log1p(-torch.erf(t / math.sqrt(2.0)))
                lp = biased(lp, M, 0.5)
            else:                                                       # BiasedNormal
                lp = normal_log_pdf(x, lat_mean, lat_logstd)
                lp = torch.where(torch.abs(x - lat_mean) > thr * lat_std, lp * M, lp)
        else:
            lp = normal_log_pdf(x, lat_mean, lat_logstd)
        if ld.use_latency_mask:
            mask = (torch.arange(M).unsqueeze(0) < node_count.unsqueeze(-1)).to(self.dtype).unsqueeze(-1)
            lp = lp * mask
            if latency_log_prob_weight:
                w = (float(M) / node_count.to(self.dtype)).reshape(-1, 1, 1)
                lp = lp * w
        out['log_p_latency'] = lp.sum((-1, -2))
        return out

    # ---------------------------------------------------------------- train / eval step
    def l2_term(self, P, trainable: List[str]):
        """train.py:251-256 + appendix B3: sum of squares of the parameters whose name does
        not contain 'bias'."""
        tot = None
        for name in trainable:
            if 'bias' in name:
                continue
            t = (P[name] ** 2).sum()
            tot = t if tot is None else tot + t
        return tot

    def train_step(self, P, buf, batch, eps_z, eps_z2, trainable: List[str], *,
                   kl_beta=1.0, l2_reg=1e-4, z_unit_ball_reg=None, z2_unit_ball_reg=None,
                   struct_training=True, latency_training=True):
        """train.py:213-277.  `struct_training=False` restates phase 2 (train.py:204-208):
        everything but latency_vae runs in eval mode."""
        mode = (struct_training, latency_training)
        qo = self.q(P, buf, batch, eps_z, eps_z2, mode)
        po = self.p(P, buf, batch, qo['z'], qo.get('z2'), mode)
        p_obs = po['log_p_node_count'] + po['log_p_adj'] + po['log_p_node_type']
        q_lat, p_lat = qo['log_q_z'], po['log_p_z']
        if 'log_p_latency' in po:
            p_obs = p_obs + po['log_p_latency']
            q_lat = q_lat + qo['log_q_z2']
            p_lat = p_lat + po['log_p_z2']
        recons = p_obs.mean()
        kl = (q_lat - p_lat).mean()
        loss = kl_beta * kl - recons
        if l2_reg:
            loss = loss + l2_reg * self.l2_term(P, trainable)

        def ball(loss, t, reg):
            if reg is not None:
                m = t.mean(-1)
                v = (t ** 2).mean(-1) - m ** 2
                loss = loss + reg * ((m ** 2).mean() + ((v - 1) ** 2).mean())
            return loss

        loss = ball(loss, qo['z'], z_unit_ball_reg)
        if 'z2' in qo:
            loss = ball(loss, qo['z2'], z2_unit_ball_reg)
        return dict(loss=loss, recons=recons, kl=kl, q=qo, p=po)

    def eval_step(self, P, buf, batch, eps_z, eps_z2, *, use_biased=True, use_latency_biased=True,
                  no_latency=False, latency_logstd_min=None, latency_log_prob_weight=False,
                  std_limit=None, clip_nll=100_000.0):
        """evaluation.py:115-148: eps_* have a leading sample axis K."""
        with torch.no_grad():
            qo = self.q(P, buf, batch, eps_z, eps_z2, False, no_latency=no_latency)
            po = self.p(P, buf, batch, qo['z'], qo.get('z2'), False, no_latency=no_latency,
                        use_biased=use_biased, use_latency_biased=use_latency_biased,
                        latency_logstd_min=latency_logstd_min,
                        latency_log_prob_weight=latency_log_prob_weight, std_limit=std_limit)
            log_joint = po['log_p_z'] + po['log_p_node_count'] + po['log_p_adj'] + po['log_p_node_type']
            latent = qo['log_q_z']
            if 'log_p_latency' in po:
                log_joint = log_joint + po['log_p_z2'] + po['log_p_latency']
                latent = latent + qo['log_q_z2']
            w = log_joint - latent                                       # [K, B]
            loss = -w.mean(0)                                            # sgvb
            nll = -(torch.logsumexp(w, dim=0) - math.log(w.shape[0]))    # -is_loglikelihood
            if clip_nll is not None:
                limit = torch.tensor(clip_nll, dtype=w.dtype)
                loss = torch.where(loss < limit, loss, limit)
                nll = torch.where(nll < limit, nll, limit)
        return dict(loss=loss, nll=nll, q=qo, p=po)


def _modes(training):
    if isinstance(training, (tuple, list)):
        return bool(training[0]), bool(training[1])
    return bool(training), bool(training)


def _clip(x, lo, hi):
    if lo is not None or hi is not None:
        return x.clamp(min=lo, max=hi)
    return x
