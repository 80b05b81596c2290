"""Shared builders for the parity tests: seeded synthetic batches, small model configs and
the oracle <-> product plumbing."""
import numpy as np
import torch

from tracevae_amd.config import ExpConfig, parse_flags
from tracevae_amd.data import FlatTraces, TraceGraphLatencyRangeFile
from tracevae_amd.synth import SyntheticSpec, generate_flat

TRAIN_SH_FLAGS = ('--model.struct.z_dim=10 --model.struct.decoder.use_prior_flow=true '
                  '--train.z_unit_ball_reg=1 --model.latency.z2_dim=10 '
                  '--model.latency.decoder.condition_on_z=true').split()


def make_config(width=None, layers=None, flags=TRAIN_SH_FLAGS, extra=()):
    """ExpConfig with the train.sh overrides; `width` / `layers` shrink the GNN stacks
    (per-head width, number of GAT layers) for fast tests."""
    cfg = ExpConfig()
    parse_flags(cfg, list(flags) + list(extra))
    mc = cfg.model
    if width is not None or layers is not None:
        for sec in (mc.struct.encoder, mc.struct.decoder, mc.latency.encoder, mc.latency.decoder):
            n = layers if layers is not None else len(sec.gnn_layers)
            w = width if width is not None else sec.gnn_layers[0]
            sec.gnn_layers = [w] * n
    return cfg


def make_flat(num_traces, num_operations, max_nodes, seed=3, anomaly_ratio=0.0):
    spec = SyntheticSpec(num_operations=num_operations, max_spans=max_nodes, seed=seed)
    return generate_flat(spec, num_traces, anomaly_ratio=anomaly_ratio)


def hand_made_flat(shapes, num_operations, seed=0):
    """Traces with a prescribed topology: each item of `shapes` is a parent list
    (parent[j] for node j+1, BFS-numbered so parent[j] <= j)."""
    rng = np.random.default_rng(seed)
    off, us, vs = [0], [], []
    for parents in shapes:
        n = len(parents) + 1
        off.append(off[-1] + n)
        us.append(np.asarray(parents, dtype=np.int64))
        vs.append(np.arange(1, n, dtype=np.int64))
    n_total = off[-1]
    node_type = rng.integers(1, num_operations, size=n_total).astype(np.int64)
    lat = np.exp(rng.normal(2.0, 0.5, size=n_total)).astype(np.float32)
    flat = FlatTraces(
        node_offsets=np.asarray(off, dtype=np.int64),
        u=np.concatenate(us) if us else np.zeros(0, np.int64),
        v=np.concatenate(vs) if vs else np.zeros(0, np.int64),
        node_type=node_type, avg_latency=lat,
        max_latency=(lat * 1.5).astype(np.float32), min_latency=(lat * 0.5).astype(np.float32),
        labels=np.zeros(len(shapes), dtype=np.int64))
    mean = rng.uniform(1.0, 20.0, size=num_operations)
    std = rng.uniform(0.5, 5.0, size=num_operations)
    return flat, TraceGraphLatencyRangeFile.from_tables(mean, std)


def subset_flat(flat: FlatTraces, ids) -> FlatTraces:
    """Gather traces `ids` into a new contiguous FlatTraces (host side)."""
    ids = np.asarray(ids, dtype=np.int64)
    off = flat.node_offsets
    counts = off[ids + 1] - off[ids]
    new_off = np.zeros(len(ids) + 1, dtype=np.int64)
    np.cumsum(counts, out=new_off[1:])
    nsel = np.concatenate([np.arange(off[i], off[i + 1]) for i in ids]) if len(ids) else np.zeros(0, np.int64)
    esel = np.concatenate([np.arange(off[i] - i, off[i + 1] - i - 1) for i in ids]) if len(ids) else np.zeros(0, np.int64)
    return FlatTraces(
        node_offsets=new_off, u=flat.u[esel], v=flat.v[esel], node_type=flat.node_type[nsel],
        avg_latency=flat.avg_latency[nsel], max_latency=flat.max_latency[nsel],
        min_latency=flat.min_latency[nsel],
        labels=None if flat.labels is None else flat.labels[ids])


def oracle_batch(flat: FlatTraces, latency_range, num_operations, max_node_count):
    from oracle.batching import build_batch
    mean, std = latency_range.tables(num_operations)
    b = build_batch(flat.node_offsets, flat.u, flat.v, flat.node_type, flat.avg_latency,
                    flat.max_latency, flat.min_latency, mean, std, max_node_count)
    b['node_offsets'] = flat.node_offsets
    return b


def oracle_state(vae, dtype=torch.float32):
    """(params with requires_grad, buffers, trainable names) from the product module."""
    sd = vae.state_dict()
    names = [n for n, _ in vae.named_parameters()]
    P, buf = {}, {}
    for k, v in sd.items():
        v = v.detach().cpu()
        if 'running_' in k or 'num_batches_tracked' in k:
            buf[k] = v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()
        else:
            P[k] = v.clone().to(dtype)
    for n in names:
        P[n].requires_grad_(True)
    return P, buf, names


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    denom = b.abs().max().clamp(min=1e-12)
    return float((a - b).abs().max() / denom)


def elem_rel_err(a, b, floor):
    """Per-ELEMENT relative error, max_i |a_i - b_i| / max(|b_i|, floor): the form north_star's
    "per-trace anomaly scores within 1e-4 relative" asks for.  `floor` is the magnitude below
    which an element is held to an absolute error of tol * floor instead (a log-probability
    that happens to pass through zero has no meaningful relative error)."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu().reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).cpu().reshape(-1)
    if a.numel() == 0:
        return 0.0
    return float(((a - b).abs() / b.abs().clamp(min=floor)).max())


def golden_sample_flat():
    """The 17 traces of the reference's `sample_dataset`, as the reference's own
    `graph_vectors()` produced them (tests/golden/sample_dataset_vectors.json), as one
    `FlatTraces` + the data set's latency ranges.  Returns (flat, latency_range, V, graphs)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                        'sample_dataset_vectors.json')
    with open(path, 'r', encoding='utf-8') as f:
        golden = json.load(f)
    graphs = [g for entry in golden['sets'].values() for g in entry['graphs']]
    off = np.zeros(len(graphs) + 1, dtype=np.int64)
    np.cumsum([g['node_count'] for g in graphs], out=off[1:])
    cat = lambda k, dt: np.concatenate([np.asarray(g[k], dtype=dt) for g in graphs])
    flat = FlatTraces(node_offsets=off, u=cat('u', np.int64), v=cat('v', np.int64),
                      node_type=cat('node_type', np.int64),
                      avg_latency=cat('avg_latency', np.float32),
                      max_latency=cat('max_latency', np.float32),
                      min_latency=cat('min_latency', np.float32))
    lr = TraceGraphLatencyRangeFile(
        latency_data={int(k): v for k, v in golden['latency_range'].items()})
    return flat, lr, golden['num_operations'], graphs


def flow_log_prob_reference(flow, z, init_act_norm=False):
    """log p(z) of an `InverseRealNVP` module written out with stock tensor ops (SURVEY.md
    appendix B4; the same formulas as `oracle.model.OracleTraceVAE._flow_log_prob`), on the
    module's own parameters so that autograd gives reference gradients.  `init_act_norm`:
    first set every ActNorm from its input batch like tensorkit's data-dependent
    initialisation does (bias = -mean, pre_scale = -log(max(std, 1e-6)), biased std)."""
    import torch.nn.functional as F
    lead = z.shape[:-1]
    x = z.reshape(-1, z.shape[-1])
    logdet = torch.zeros(x.shape[0], dtype=z.dtype, device=z.device)
    flows = list(flow.original_flow.flows)
    for lvl in range(len(flows) // 3):
        an, cp, dense = flows[3 * lvl:3 * lvl + 3]
        if init_act_norm:
            with torch.no_grad():
                an.bias.copy_(-x.mean(0))
                an.pre_scale.copy_(-torch.log(x.std(0, unbiased=False).clamp(min=1e-6)))
        x = (x + an.bias) * torch.exp(an.pre_scale)
        logdet = logdet + an.pre_scale.sum()
        n1 = cp.n1
        x1, x2 = x[:, :n1], x[:, n1:]
        br = cp.shift_and_pre_scale
        h = x1
        for lin in br.shared:
            h = F.leaky_relu(F.linear(h, lin.weight_store.value, lin.bias_store.value), 0.01)
        shift = F.linear(h, br.branches[0][0].weight_store.value, br.branches[0][0].bias_store.value)
        pre = F.linear(h, br.branches[1][0].weight_store.value, br.branches[1][0].bias_store.value)
        log_scale = F.logsigmoid(pre + 2.0)
        x = torch.cat([x1, (x2 + shift) * torch.exp(log_scale)], dim=-1)
        logdet = logdet + log_scale.sum(-1)
        x = x @ dense.kernel.t()
        logdet = logdet + torch.linalg.slogdet(dense.kernel)[1]
    base = (-0.5 * float(np.log(2.0 * np.pi)) - 0.5 * x * x).sum(-1)
    return (base + logdet).reshape(lead)
