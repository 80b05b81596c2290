"""From a sampled `p` net back to trace graphs, and graph comparison helpers
(reference: `tracegnn/models/trace_vae/graph_utils.py`).

Host-side post-processing of what `TraceVAE.p(n_z=...)` drew on the GPU: one device-to-host
copy per tensor, then plain NumPy / Python per generated trace, as in the reference.

* `p_net_to_trace_graphs`  graph_utils.py:98-299
* `flat_to_nx_graphs`      graph_utils.py:35-95 (needs networkx)
* `diff_graph`             graph_utils.py:344-396
* `trace_graph_key`        graph_utils.py:412-424
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import constants as C
from .bayes import BayesianNet, StochasticTensor
from .data import (TraceGraph, TraceGraphIDManager, TraceGraphLatencyRangeFile, TraceGraphNode,
                   TraceGraphNodeFeatures, TraceGraphNodeReconsScores)

__all__ = ['triu_to_dense', 'p_net_to_trace_graphs', 'flat_to_nx_graphs', 'GraphNodeMatch',
           'GraphNodeDiff', 'diff_graph', 'trace_graph_key']

_PARAMETRIC = ('Normal', 'SafeNormal', 'BiasedNormal', 'AnomalyDetectionNormal', 'Bernoulli',
               'BiasedBernoulli', 'Categorical', 'BiasedCategorical', 'OneHotCategorical')


def triu_to_dense(triu: np.ndarray, max_node_count: int, pad_value=0) -> np.ndarray:
    """`[..., T]` row-major strict upper triangles -> `[..., M, M]` (tensor_utils.py:70-76)."""
    m = int(max_node_count)
    triu = np.asarray(triu)
    out = np.full(triu.shape[:-1] + (m, m), pad_value, dtype=triu.dtype)
    rows, cols = np.triu_indices(m, 1)
    out[..., rows, cols] = triu
    return out


def _host(x, ndims: int) -> np.ndarray:
    """A tensor flattened to `ndims` axes (all leading axes merged), on the host."""
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x.reshape((-1,) + x.shape[x.ndim - ndims + 1:])


def _base(t: StochasticTensor):
    d = t.distribution
    while d.kind not in _PARAMETRIC:
        d = d.base_distribution
    return d


def _log_sigmoid(x: np.ndarray) -> np.ndarray:
    # log(1 / (1 + exp(-x))), stable on both sides (graph_utils.py:135-141)
    return np.where(x >= 0, -np.log1p(np.exp(-np.abs(x))), x - np.log1p(np.exp(-np.abs(x))))


def _log_softmax(x: np.ndarray) -> np.ndarray:
    m = x.max(axis=-1, keepdims=True)
    return x - (m + np.log(np.exp(x - m).sum(axis=-1, keepdims=True)))


def p_net_to_trace_graphs(p: BayesianNet,
                          id_manager: Optional[TraceGraphIDManager],
                          latency_range: TraceGraphLatencyRangeFile,
                          discard_node_with_type_0: bool = True,
                          discard_node_with_unknown_latency_range: bool = True,
                          discard_graph_with_error_node_count: bool = False,
                          keep_front_shape: bool = False,
                          rng=None,
                          max_node_count: Optional[int] = None,
                          ) -> Union[List[Optional[TraceGraph]], np.ndarray]:
    """One `TraceGraph` (or None) per sample of a `p` net drawn without an observed graph.

    Per sample, as the reference does (graph_utils.py:164-293): nodes beyond the drawn node
    count are ignored; nodes of operation 0 / without a latency range are cut out; every
    remaining node whose column of the drawn adjacency does not hold exactly one parent gets
    one drawn in proportion to the edge probabilities (`rng.multinomial`, the reference uses
    the global `np.random`); when edges are still missing only the component of node 0 is
    kept; the tree is assembled, latencies are mapped back with the operation's (mean, std),
    siblings of equal operation are merged.  `rng`: anything with `.multinomial` (default
    `np.random`)."""
    if C.USE_MULTI_DIM_LATENCY_CODEC:
        raise RuntimeError('`USE_MULTI_DIM_LATENCY_CODEC` is not supported.')
    rng = np.random if rng is None else rng
    adj_t = p['adj'].tensor
    front_shape = tuple(adj_t.shape[:-1])
    t = int(adj_t.shape[-1])
    m = int(max_node_count) if max_node_count else int(round((1 + math.sqrt(1 + 8 * t)) / 2))
    if m * (m - 1) // 2 != t:
        raise ValueError(f'adj has {t} entries per graph, not a strict upper triangle')

    adj_base = _base(p['adj'])
    adjs = triu_to_dense(_host(adj_t, 2), m)
    adj_probs = triu_to_dense(_host(adj_base.probs, 2), m)
    adj_logits = triu_to_dense(_log_sigmoid(_host(adj_base.logits, 2)), m, pad_value=-100000)
    node_counts = _host(p['node_count'].tensor, 1)
    node_types = _host(p['node_type'].tensor, 2)
    node_type_logits = _log_softmax(_host(_base(p['node_type']).logits, 3))
    has_latency = 'latency' in p
    if has_latency:
        lat_base = _base(p['latency'])
        avg_latencies = _host(p['latency'].tensor, 3)[..., 0]
        latency_means = _host(lat_base.mean, 3)
        latency_stds = _host(lat_base.std, 3)

    ret: List[Optional[TraceGraph]] = []
    for i, node_count in enumerate(node_counts):
        node_count = int(node_count)
        adj = adjs[i][:node_count, :node_count]
        adj_prob = adj_probs[i][:node_count, :node_count]
        adj_logit = adj_logits[i]
        node_type = node_types[i]
        keep = np.ones([node_count], dtype=bool)

        # cut out the nodes that cannot be part of a trace
        remaining = node_count
        for j in range(node_count):
            op = int(node_type[j])
            if (discard_node_with_type_0 and op == 0) or \
                    (discard_node_with_unknown_latency_range and op not in latency_range):
                keep[j] = False
                remaining -= 1
                adj[:, j] = 0
                adj[j, :] = 0
                adj_prob[:, j] = 0
                adj_prob[j, :] = 0

        # exactly one parent per node: redraw the column when it holds none or several
        for j in range(node_count):
            if keep[j] and np.sum(adj[:, j]) != 1:
                weights = adj_prob[:, j]
                total = np.sum(weights)
                if total > 1e-7:
                    pvals = weights / total
                    cand = pvals > 1e-7
                    pick = np.arange(len(pvals))[cand][np.argmax(rng.multinomial(1, pvals[cand]))]
                    adj[:, j] = 0
                    adj[pick, j] = 1

        edges = [(int(a), int(b)) for a, b in zip(*np.where(adj))]
        if len(edges) < remaining - 1:
            # a forest: keep the component of node 0
            parent = {j: -1 for j in range(node_count) if keep[j]}

            def root_of(s):
                path = []
                while parent[s] != -1:
                    path.append(s)
                    s = parent[s]
                for q in path[:-1]:
                    parent[q] = s
                return s

            for a, b in edges:
                parent[b] = a
            edges = [(a, b) for a, b in edges if a == 0 or root_of(a) == 0]
            remaining = len(edges) + 1

        if discard_graph_with_error_node_count and remaining != node_count:
            ret.append(None)
            continue

        nodes: Dict[int, TraceGraphNode] = {}

        def node_of(s: int) -> TraceGraphNode:
            if s not in nodes:
                op = node_type[s]
                if has_latency:
                    latency = avg_latencies[i][s]
                    if op in latency_range:
                        mu, std = latency_range[op]
                        latency = latency * std + mu
                    features = TraceGraphNodeFeatures(span_count=1, avg_latency=latency,
                                                      max_latency=latency, min_latency=latency)
                    nstd = float(abs(avg_latencies[i][s] - latency_means[i][s, 0]) /
                                 latency_stds[i][s, 0])
                else:
                    features = TraceGraphNodeFeatures(span_count=1, avg_latency=math.nan,
                                                      max_latency=math.nan, min_latency=math.nan)
                    nstd = 0
                nodes[s] = TraceGraphNode.new_sampled(
                    node_id=s, operation_id=op, features=features,
                    scores=TraceGraphNodeReconsScores(
                        edge_logit=0, operation_logit=node_type_logits[i][s, op],
                        avg_latency_nstd=nstd))
            return nodes[s]

        for a, b in sorted(edges):
            if keep[a] and keep[b]:
                child = node_of(b)
                node_of(a).children.append(child)
                child.scores.edge_logit = adj_logit[a, b]

        if 0 in nodes:
            g = TraceGraph.new_sampled(nodes[0], len(nodes), -1)
            g.merge_spans_and_assign_id()
            ret.append(g)
        else:
            ret.append(None)

    if keep_front_shape:
        out = np.empty([len(ret)], dtype=object)
        out[:] = ret
        return out.reshape(front_shape)
    return ret


def flat_to_nx_graphs(p: BayesianNet, id_manager: TraceGraphIDManager,
                      latency_range: TraceGraphLatencyRangeFile, min_edge_weight: float = 0.2,
                      max_node_count: Optional[int] = None):
    """Weighted `nx.Graph`s of the edge PROBABILITIES of a sampled `p` net
    (graph_utils.py:35-95): nodes `0..node_count-1`, an edge wherever the probability reaches
    `min_edge_weight`, node attributes `node_type`, `operation`, `avg_latency`."""
    import networkx as nx
    t = int(p['adj'].tensor.shape[-1])
    m = int(max_node_count) if max_node_count else int(round((1 + math.sqrt(1 + 8 * t)) / 2))
    probs = triu_to_dense(_host(p['adj'].distribution.probs, 2), m)
    node_counts = _host(p['node_count'].tensor, 1)
    node_types = _host(p['node_type'].tensor, 2)
    latencies = None
    if 'latency' in p:
        src = _host(p['latency'].distribution.base_distribution.mean, 3)
        latencies = np.full(src.shape, -1., dtype=np.float32)
        for i in range(node_types.shape[0]):
            for j in range(node_types.shape[1]):
                op = int(node_types[i, j])
                if op in latency_range:
                    mu, std = latency_range[op]
                    latencies[i, j] = src[i, j] * std + mu
    ret = []
    for i, node_count in enumerate(node_counts):
        node_count = int(node_count)
        g = nx.Graph()
        g.add_nodes_from(range(node_count))
        for u in range(node_count):
            for v in range(u + 1, node_count):
                w = float(probs[i, u, v])
                if w >= min_edge_weight:
                    g.add_edge(u, v, weight=w)
        for j in range(node_count):
            op = int(node_types[i, j])
            g.nodes[j]['node_type'] = op
            g.nodes[j]['operation'] = id_manager.operation_id.reverse_map(op)
            if latencies is not None:
                for k, pfx in enumerate(('avg_', 'max_', 'min_')):
                    if k < C.LATENCY_DIM:
                        g.nodes[j][f'{pfx}latency'] = latencies[i, j, k]
        ret.append(g)
    return ret


class GraphNodeMatch(object):
    """Matched node pairs of two traces (graph_utils.py:302-318); nodes are keyed by identity."""
    __slots__ = ['g1_to_g2', 'g2_to_g1']

    def __init__(self):
        self.g1_to_g2: Dict[int, TraceGraphNode] = {}
        self.g2_to_g1: Dict[int, TraceGraphNode] = {}

    def add_match(self, node1: TraceGraphNode, node2: TraceGraphNode) -> None:
        self.g1_to_g2[id(node1)] = node2
        self.g2_to_g1[id(node2)] = node1

    def __len__(self):
        return len(self.g1_to_g2)


@dataclass(init=False)
class GraphNodeDiff(object):
    """A branch present in one trace only: `offset` -1 = only in g1, +1 = only in g2."""
    parent: Optional[TraceGraphNode]
    depth: int
    node: TraceGraphNode
    offset: int
    node_count: int

    def __init__(self, parent, depth, node, offset):
        self.parent, self.depth, self.node, self.offset = parent, depth, node, offset
        self.node_count = node.count_nodes()

    def __repr__(self):
        return f'GraphNodeDiff(depth={self.depth}, offset={self.offset})'


def diff_graph(g1: TraceGraph, g2: TraceGraph) -> Tuple[GraphNodeMatch, List[GraphNodeDiff]]:
    """Top-down comparison of two traces whose children are sorted by operation id
    (graph_utils.py:344-396): nodes match while their operations agree along the path from
    the root; the children of a matched pair are merged like two sorted lists."""
    match, diffs = GraphNodeMatch(), []
    stack = [(0, None, None, g1.root, g2.root)]
    # an explicit stack visiting in the order of the reference's recursion
    while stack:
        depth, par1, par2, n1, n2 = stack.pop()
        if n1 is None and n2 is None:
            continue
        if n1 is None:
            diffs.append(GraphNodeDiff(par2, depth, n2, 1))
        elif n2 is None:
            diffs.append(GraphNodeDiff(par1, depth, n1, -1))
        elif n1.operation_id != n2.operation_id:
            diffs.append(GraphNodeDiff(par1, depth, n1, -1))
            diffs.append(GraphNodeDiff(par2, depth, n2, 1))
        else:
            match.add_match(n1, n2)
            i = j = 0
            c1, c2 = n1.children, n2.children
            todo = []
            while i < len(c1) or j < len(c2):
                a = c1[i] if i < len(c1) else None
                b = c2[j] if j < len(c2) else None
                if b is None or (a is not None and a.operation_id < b.operation_id):
                    todo.append((depth + 1, n1, None, a, None))
                    i += 1
                elif a is None or b.operation_id < a.operation_id:
                    todo.append((depth + 1, None, n2, None, b))
                    j += 1
                else:
                    todo.append((depth + 1, n1, n2, a, b))
                    i += 1
                    j += 1
            stack.extend(reversed(todo))
    return match, diffs


def trace_graph_key(graph: TraceGraph) -> str:
    """Order-free signature of a trace: the sorted multiset of `parent_op,op` pairs, each
    counted once per span of the node times the span counts above it (graph_utils.py:412-424)."""
    pairs: List[str] = []
    stack = [(graph.root, 0, 1)]
    while stack:
        node, parent_op, mult = stack.pop()
        mult = mult * len(node.spans)
        pairs.extend([f'{parent_op},{node.operation_id}'] * mult)
        stack.extend((c, node.operation_id, mult) for c in node.children)
    return ';'.join(sorted(pairs))
