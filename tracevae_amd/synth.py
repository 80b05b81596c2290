"""Seeded synthetic trace graphs (SURVEY.md section 8d / BASELINE.md section 3).

Per trace: span count s ~ U{2..S}; random recursive tree (parent of span j ~ U{0..j-1});
operation id per span Zipf-like over 1..V-1 (id 0 is the NULL operation); the reference's
span-merge rule and BFS renumbering are applied (`TraceGraph.merge_spans_and_assign_id`),
so n <= s nodes remain; latency per span ~ LogNormal(mu_op, 0.5) with mu_op ~ U(0, 6);
`latency_range[op]` is the analytic (mean, std) of that log-normal, so normalised
latencies are O(1).

The generator emits `FlatTraces` (the layout the batching kernel reads); there are no
pickles or sqlite files on the timed path.
"""
from typing import Optional, Tuple

import numpy as np

from .data import (FlatTraces, TraceGraph, TraceGraphLatencyRangeFile, TraceGraphNode,
                   TraceGraphNodeFeatures, flatten_trace_graphs)

__all__ = ['SyntheticSpec', 'make_latency_range', 'generate_trace_graphs', 'generate_flat',
           'tile_flat']

SIGMA = 0.5


class SyntheticSpec(object):
    def __init__(self, num_operations: int = 1001, max_spans: int = 100, seed: int = 1234,
                 zipf_a: float = 1.1, zipf_shift: float = 10.0):
        self.num_operations = int(num_operations)
        self.max_spans = int(max_spans)
        self.seed = int(seed)
        rng = np.random.default_rng(seed)
        self.mu_op = rng.uniform(0.0, 6.0, size=self.num_operations)
        ranks = np.arange(1, self.num_operations, dtype=np.float64)
        p = 1.0 / np.power(ranks + zipf_shift, zipf_a)
        self.op_prob = p / p.sum()
        # the popularity order of operations is itself a seeded permutation
        self.op_ids = rng.permutation(np.arange(1, self.num_operations, dtype=np.int64))


def make_latency_range(spec: SyntheticSpec) -> TraceGraphLatencyRangeFile:
    mean = np.exp(spec.mu_op + 0.5 * SIGMA * SIGMA)
    std = mean * np.sqrt(np.exp(SIGMA * SIGMA) - 1.0)
    mean[0] = np.nan
    std[0] = np.nan
    lr = TraceGraphLatencyRangeFile.from_tables(mean, std)
    # NULL operation: never sampled, but keep the table total
    lr.latency_data[0] = {'mean': 0.0, 'std': 1.0}
    return lr


def _one_trace(spec: SyntheticSpec, rng: np.random.Generator, label: int) -> TraceGraph:
    s = int(rng.integers(2, spec.max_spans + 1))
    parent = np.floor(rng.random(s - 1) * np.arange(1, s)).astype(np.int64)
    ops = spec.op_ids[rng.choice(spec.num_operations - 1, size=s, p=spec.op_prob)]
    lat = np.exp(spec.mu_op[ops] + SIGMA * rng.standard_normal(s))
    if label == 2:      # latency anomaly: one span is 8..40x slower
        lat[int(rng.integers(0, s))] *= float(rng.uniform(8.0, 40.0))
    nodes = [
        TraceGraphNode(
            node_id=None, service_id=None, operation_id=int(ops[j]),
            features=TraceGraphNodeFeatures(
                span_count=1, avg_latency=float(lat[j]),
                max_latency=float(lat[j]), min_latency=float(lat[j])),
        ) for j in range(s)
    ]
    dropped = -1
    if label == 1 and s > 2:      # drop anomaly: one subtree disappears
        dropped = int(rng.integers(1, s))
    for j in range(1, s):
        if j != dropped:
            nodes[int(parent[j - 1])].children.append(nodes[j])
    data = {} if label == 0 else {
        'is_anomaly': True, 'anomaly_type': 'drop' if label == 1 else 'latency'}
    g = TraceGraph(version=TraceGraph.default_version(), trace_id=None, parent_id=None,
                   root=nodes[0], node_count=None, max_depth=None, data=data)
    g.merge_spans_and_assign_id()
    return g


def generate_trace_graphs(spec: SyntheticSpec, count: int, seed: Optional[int] = None,
                          anomaly_ratio: float = 0.0, min_nodes: int = 2):
    """`count` trace graphs with at least `min_nodes` nodes after merging (the reference
    drops smaller ones in preprocessing, trace_graph.py:602-608)."""
    rng = np.random.default_rng(spec.seed + 1 if seed is None else seed)
    out = []
    while len(out) < count:
        label = 0
        if anomaly_ratio > 0 and rng.random() < anomaly_ratio:
            label = int(rng.integers(1, 3))
        g = _one_trace(spec, rng, label)
        if g.node_count >= min_nodes:
            out.append(g)
    return out


def generate_flat(spec: SyntheticSpec, count: int, seed: Optional[int] = None,
                  anomaly_ratio: float = 0.0) -> Tuple[FlatTraces, TraceGraphLatencyRangeFile]:
    graphs = generate_trace_graphs(spec, count, seed=seed, anomaly_ratio=anomaly_ratio)
    return flatten_trace_graphs(graphs), make_latency_range(spec)


def tile_flat(flat: FlatTraces, copies: int, seed: int = 0, jitter: float = 0.1) -> FlatTraces:
    """A data set of `copies * len(flat)` traces for store-sized runs (BASELINE.json configs[1]:
    one million traces resident in HBM): the object-level generator above builds a trace in
    ~0.4 ms of Python, so a million of them would take minutes; here the topologies and operation
    ids of `flat` repeat `copies` times and every copy draws its own latencies around the
    originals (x exp(jitter * N(0,1)) per node, min <= avg <= max kept).  Copy 0 is `flat`."""
    if copies <= 1:
        return flat
    rng = np.random.default_rng(seed)
    n, t = int(flat.node_offsets[-1]), len(flat)
    off = (flat.node_offsets[None, :-1] + n * np.arange(copies, dtype=np.int64)[:, None]).reshape(-1)
    off = np.concatenate([off, [n * copies]]).astype(np.int64)
    scale = np.exp(jitter * rng.standard_normal(n * copies)).astype(np.float32)
    scale[:n] = 1.0
    rep = lambda a: np.tile(a, copies)
    return FlatTraces(
        node_offsets=off, u=rep(flat.u), v=rep(flat.v), node_type=rep(flat.node_type),
        avg_latency=rep(flat.avg_latency) * scale, max_latency=rep(flat.max_latency) * scale,
        min_latency=rep(flat.min_latency) * scale,
        labels=None if flat.labels is None else rep(flat.labels))
