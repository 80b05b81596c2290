"""Trace-graph domain types and the flat-array trace store that feeds the batching kernel.

Two input forms reach `TraceGraphBatch`:

* a list of `TraceGraph` objects - the form the reference's drivers pass
  (`tracegnn/data/trace_graph.py:177-401`); `TraceGraph.graph_vectors()` returns the same
  arrays, with the same dtypes and the same ordering, as the reference's
  (`trace_graph.py:286-352`): node ids are BFS order, `u[e] < v[e]`, edges are emitted
  parent by parent in BFS order;
* a `FlatTraces` store - all traces of a data set as concatenated arrays
  (`node_offsets`, `u`, `v`, `node_type`, latencies).  This is the layout the HIP batching
  kernel (`csrc/batch_build.hip`) reads directly from HBM; a batch is just a list of trace
  indices.

`TraceGraphIDManager` / `TraceGraphLatencyRangeFile` read the reference's
`id_manager/{service_id,operation_id,latency_range}.yml` files (`utils/id_assign.py:8-58`,
`utils/latency_range_file.py:11-80`) - read-only here: nothing on this path writes them.
"""
import os
from dataclasses import dataclass, field
from typing import Any, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import yaml

__all__ = [
    'TraceGraphNodeFeatures', 'TraceGraphNodeReconsScores', 'TraceGraphSpan', 'TraceGraphNode', 'TraceGraphVectors', 'TraceGraph',
    'IDAssign', 'TraceGraphIDManager', 'TraceGraphLatencyRangeFile',
    'FlatTraces', 'flatten_trace_graphs', 'load_trace_csv', 'df_to_trace_graphs',
]


class _SlotState(object):
    """Pickle state in the form the reference's `__slots__` classes produce and accept
    (`(None, {field: value})`), so `bytes_db` files go both ways."""

    def __getstate__(self):
        return None, dict(vars(self))


@dataclass
class TraceGraphNodeFeatures(_SlotState):
    span_count: int
    avg_latency: float
    max_latency: float
    min_latency: float


@dataclass
class TraceGraphNodeReconsScores(object):
    edge_logit: float
    operation_logit: float
    avg_latency_nstd: float


@dataclass
class TraceGraphSpan(_SlotState):
    span_id: Optional[int]
    start_time: Optional[Any]
    latency: float


@dataclass(eq=False)
class TraceGraphNode(_SlotState):
    node_id: Optional[int]
    service_id: Optional[int]
    operation_id: int
    features: TraceGraphNodeFeatures
    children: List['TraceGraphNode'] = field(default_factory=list)
    spans: Optional[list] = None
    scores: Optional[Any] = None
    anomaly: Optional[int] = None

    @staticmethod
    def new_sampled(node_id: int, operation_id: int, features: TraceGraphNodeFeatures,
                    scores: Optional[TraceGraphNodeReconsScores] = None) -> 'TraceGraphNode':
        """A node of a generated trace (trace_graph.py:86-100): no service, no spans."""
        return TraceGraphNode(node_id=node_id, service_id=None, operation_id=operation_id,
                              features=features, children=[], spans=None, scores=scores,
                              anomaly=None)

    def count_nodes(self) -> int:
        return sum(1 for _ in self.iter_bfs())

    def iter_bfs(self, with_parent: bool = False) -> Iterator[tuple]:
        """Level-order walk.  Yields `(depth, node)` or, with `with_parent`,
        `(depth, index_among_siblings, node, parent)` (reference: trace_graph.py:102-137)."""
        frontier = [(self, None, 0)]
        depth = 0
        while frontier:
            nxt = []
            for node, parent, idx in frontier:
                if with_parent:
                    yield depth, idx, node, parent
                else:
                    yield depth, node
                nxt.extend((c, node, i) for i, c in enumerate(node.children))
            frontier = nxt
            depth += 1


@dataclass
class TraceGraphVectors(_SlotState):
    u: np.ndarray            # int64 [n-1] parent ids   (u[e] < v[e])
    v: np.ndarray            # int64 [n-1] child ids
    node_type: np.ndarray    # int64 [n]
    node_depth: np.ndarray   # int64 [n]
    node_idx: np.ndarray     # int64 [n]
    span_count: np.ndarray   # int64 [n]
    avg_latency: np.ndarray  # float32 [n]
    max_latency: np.ndarray  # float32 [n]
    min_latency: np.ndarray  # float32 [n]


@dataclass
class TraceGraph(_SlotState):
    version: int
    trace_id: Optional[Tuple[int, int]]
    parent_id: Optional[int]
    root: TraceGraphNode
    node_count: Optional[int]
    max_depth: Optional[int]
    data: Dict[str, Any] = field(default_factory=dict)

    @staticmethod
    def default_version() -> int:
        return 0x2

    @staticmethod
    def new_sampled(root: TraceGraphNode, node_count: int, max_depth: int) -> 'TraceGraph':
        """A generated trace (trace_graph.py:197-206)."""
        return TraceGraph(version=TraceGraph.default_version(), trace_id=None, parent_id=None,
                          root=root, node_count=node_count, max_depth=max_depth, data={})

    @property
    def edge_count(self) -> Optional[int]:
        return None if self.node_count is None else self.node_count - 1

    def iter_bfs(self, with_parent: bool = False):
        yield from self.root.iter_bfs(with_parent=with_parent)

    def assign_node_id(self) -> None:
        """Number the nodes in BFS order (trace_graph.py:270-284)."""
        count = depth_max = 0
        for depth, node in self.iter_bfs():
            node.node_id = count
            count += 1
            depth_max = max(depth_max, depth)
        self.node_count, self.max_depth = count, depth_max

    def merge_spans_and_assign_id(self) -> None:
        """Merge sibling nodes with equal operation id, then number in BFS order
        (the reference's merge rule, trace_graph.py:219-268): children are sorted by
        operation id (stable), equal neighbours are folded into the first one - span
        counts add, the average is the running weighted mean, max/min combine, and the
        grand-children lists are concatenated."""
        count = depth_max = 0
        for depth, node in self.iter_bfs():
            depth_max = max(depth_max, depth)
            node.node_id = count
            count += 1
            merged: List[TraceGraphNode] = []
            for child in sorted(node.children, key=lambda c: c.operation_id):
                if merged and merged[-1].operation_id == child.operation_id:
                    keep = merged[-1]
                    a, b = keep.features, child.features
                    a.span_count += b.span_count
                    a.avg_latency += (b.avg_latency - a.avg_latency) * (b.span_count / a.span_count)
                    a.max_latency = max(a.max_latency, b.max_latency)
                    a.min_latency = min(a.min_latency, b.min_latency)
                    if child.children:
                        keep.children = (keep.children + child.children) if keep.children \
                            else child.children
                    if child.spans:
                        keep.spans = (keep.spans + child.spans) if keep.spans else child.spans
                else:
                    merged.append(child)
            node.children = merged
        self.node_count, self.max_depth = count, depth_max

    def graph_vectors(self) -> TraceGraphVectors:
        n = self.node_count
        gv = TraceGraphVectors(
            u=np.empty([n - 1], dtype=np.int64), v=np.empty([n - 1], dtype=np.int64),
            node_type=np.zeros([n], dtype=np.int64), node_depth=np.zeros([n], dtype=np.int64),
            node_idx=np.zeros([n], dtype=np.int64), span_count=np.zeros([n], dtype=np.int64),
            avg_latency=np.zeros([n], dtype=np.float32),
            max_latency=np.zeros([n], dtype=np.float32),
            min_latency=np.zeros([n], dtype=np.float32),
        )
        e = 0
        for depth, idx, node, _parent in self.iter_bfs(with_parent=True):
            j = node.node_id
            gv.node_type[j] = node.operation_id
            gv.node_depth[j] = depth
            gv.node_idx[j] = idx
            gv.span_count[j] = node.features.span_count
            gv.avg_latency[j] = node.features.avg_latency
            gv.max_latency[j] = node.features.max_latency
            gv.min_latency[j] = node.features.min_latency
            for child in node.children:
                gv.u[e] = j
                gv.v[e] = child.node_id
                e += 1
        if e != n - 1:
            raise ValueError(f'edge count mismatch: {e} != {n - 1}')
        return gv


# ---------------------------------------------------------------------------------------
# id manager / latency range (read side only)
# ---------------------------------------------------------------------------------------
class IDAssign(object):
    def __init__(self, path: str):
        self._path = path
        self._mapping: Dict[str, int] = {'': 0}
        if os.path.isfile(path):
            with open(path, 'r', encoding='utf-8') as f:
                self._mapping = yaml.safe_load(f.read())
        self._next_index = (max(self._mapping.values()) + 1) if self._mapping else 0
        self._rev_mapping = {v: k for k, v in self._mapping.items()}

    @classmethod
    def from_mapping(cls, mapping: Dict[str, int], size: Optional[int] = None) -> 'IDAssign':
        self = cls.__new__(cls)
        self._path = None
        self._mapping = dict(mapping)
        self._next_index = size if size is not None else \
            ((max(self._mapping.values()) + 1) if self._mapping else 0)
        self._rev_mapping = {v: k for k, v in self._mapping.items()}
        return self

    def __len__(self):
        return self._next_index

    def __getitem__(self, key):
        return self._mapping[key]

    def __contains__(self, key):
        return key in self._mapping

    def reverse_map(self, index: int):
        return self._rev_mapping[index]


class TraceGraphIDManager(object):
    def __init__(self, root_dir: str):
        self.root_dir = os.path.abspath(root_dir)
        self.service_id = IDAssign(os.path.join(self.root_dir, 'service_id.yml'))
        self.operation_id = IDAssign(os.path.join(self.root_dir, 'operation_id.yml'))

    @classmethod
    def from_mappings(cls, service_id: Dict[str, int], operation_id: Dict[str, int],
                      num_operations: Optional[int] = None) -> 'TraceGraphIDManager':
        self = cls.__new__(cls)
        self.root_dir = None
        self.service_id = IDAssign.from_mapping(service_id)
        self.operation_id = IDAssign.from_mapping(operation_id, num_operations)
        return self

    @property
    def num_operations(self) -> int:
        return len(self.operation_id)


class TraceGraphLatencyRangeFile(object):
    """Per-operation latency `(mean, std)` used to normalise node latencies
    (`latency_range_file.py:41-43`, consumed at `dataset.py:73-79`)."""

    def __init__(self, root_dir: Optional[str] = None, require_exists: bool = False,
                 latency_data: Optional[Dict[int, Dict[str, float]]] = None):
        self.latency_data: Dict[int, Dict[str, float]] = {}
        self.root_dir = None
        if latency_data is not None:
            self.latency_data = {int(k): dict(v) for k, v in latency_data.items()}
        if root_dir is not None:
            self.root_dir = os.path.abspath(root_dir)
            path = os.path.join(self.root_dir, 'latency_range.yml')
            if os.path.exists(path):
                with open(path, 'r', encoding='utf-8') as f:
                    obj = yaml.safe_load(f.read())
                self.latency_data = {int(k): v for k, v in obj.items()}
            elif require_exists:
                raise IOError(f'LatencyRangeFile does not exist: {path}')

    def __contains__(self, item):
        return int(item) in self.latency_data

    def __getitem__(self, operation_id: int) -> Tuple[float, float]:
        v = self.latency_data[int(operation_id)]
        return v['mean'], v['std']

    def tables(self, num_operations: int) -> Tuple[np.ndarray, np.ndarray]:
        """Dense float64 `(mean[V], std[V])` tables; NaN marks an operation with no entry
        (the reference raises `KeyError` for those when it meets one, `dataset.py:77`)."""
        mean = np.full([num_operations], np.nan, dtype=np.float64)
        std = np.full([num_operations], np.nan, dtype=np.float64)
        for k, v in self.latency_data.items():
            if 0 <= k < num_operations:
                mean[k], std[k] = float(v['mean']), float(v['std'])
        return mean, std

    @staticmethod
    def from_tables(mean: np.ndarray, std: np.ndarray) -> 'TraceGraphLatencyRangeFile':
        return TraceGraphLatencyRangeFile(latency_data={
            i: {'mean': float(m), 'std': float(s)}
            for i, (m, s) in enumerate(zip(mean, std)) if not (np.isnan(m) or np.isnan(s))
        })


# ---------------------------------------------------------------------------------------
# flat trace store
# ---------------------------------------------------------------------------------------
@dataclass
class FlatTraces(object):
    """All traces of a data set as concatenated arrays (host copies; `TraceGraphBatch`
    keeps the device copies).  Trace `i` owns nodes `node_offsets[i]:node_offsets[i+1]`
    and edges `node_offsets[i]-i : node_offsets[i+1]-(i+1)` (a tree has n-1 edges).
    `u`, `v` are node ids local to their trace with `u < v`."""
    node_offsets: np.ndarray   # int64 [G+1]
    u: np.ndarray              # int64 [N-G]
    v: np.ndarray              # int64 [N-G]
    node_type: np.ndarray      # int64 [N]
    avg_latency: np.ndarray    # float32 [N]   raw (not normalised)
    max_latency: np.ndarray    # float32 [N]
    min_latency: np.ndarray    # float32 [N]
    labels: Optional[np.ndarray] = None   # int64 [G]  0 normal / 1 drop / 2 latency

    def __len__(self) -> int:
        return len(self.node_offsets) - 1

    @property
    def node_counts(self) -> np.ndarray:
        return np.diff(self.node_offsets)

    def validate(self, max_node_count: Optional[int] = None) -> None:
        off = self.node_offsets
        g = len(self)
        if off[0] != 0 or np.any(np.diff(off) < 1):
            raise ValueError('node_offsets must start at 0 and be strictly increasing')
        n = int(off[-1])
        if len(self.node_type) != n or len(self.avg_latency) != n:
            raise ValueError('node arrays do not match node_offsets')
        if len(self.u) != n - g or len(self.v) != n - g:
            raise ValueError('edge arrays must hold n-1 edges per trace')
        if np.any(self.u >= self.v) or np.any(self.u < 0):
            raise ValueError('edges must satisfy 0 <= u < v')
        counts = np.diff(off)
        edge_owner = np.repeat(np.arange(g), counts - 1)
        if np.any(self.v >= counts[edge_owner]):
            raise ValueError('edge endpoint outside its trace')
        if max_node_count is not None and counts.max(initial=0) > max_node_count:
            raise ValueError(f'a trace has more than MAX_NODE_COUNT={max_node_count} nodes')


def flatten_trace_graphs(trace_graphs: Sequence[TraceGraph]) -> FlatTraces:
    """Collate `TraceGraph` objects (via `graph_vectors()`) into the flat layout."""
    gvs = [g.graph_vectors() for g in trace_graphs]
    counts = np.asarray([len(gv.node_type) for gv in gvs], dtype=np.int64)
    off = np.zeros([len(gvs) + 1], dtype=np.int64)
    np.cumsum(counts, out=off[1:])

    def cat(name, dtype):
        if not gvs:
            return np.zeros([0], dtype=dtype)
        return np.ascontiguousarray(np.concatenate([getattr(gv, name) for gv in gvs]), dtype=dtype)

    labels = np.asarray([
        0 if not g.data.get('is_anomaly') else (1 if g.data.get('anomaly_type') == 'drop' else 2)
        for g in trace_graphs], dtype=np.int64)
    return FlatTraces(
        node_offsets=off, u=cat('u', np.int64), v=cat('v', np.int64),
        node_type=cat('node_type', np.int64), avg_latency=cat('avg_latency', np.float32),
        max_latency=cat('max_latency', np.float32), min_latency=cat('min_latency', np.float32),
        labels=labels,
    )


# ---------------------------------------------------------------------------------------
# CSV spans -> trace graphs (config-1 plumbing; tracegnn/data/trace_graph.py:443-617)
# ---------------------------------------------------------------------------------------
_CSV_COLUMNS = ['traceIdHigh', 'traceIdLow', 'spanId', 'parentSpanId', 'serviceName',
                'operationName', 'startTime', 'duration', 'nanosecond', 'DBhash']
_CSV_TEST_COLUMNS = ['nodeLatencyLabel', 'graphLatencyLabel', 'graphStructureLabel']


def load_trace_csv(input_path: str, is_test: bool = False):
    import pandas as pd
    dtype = {c: int for c in _CSV_COLUMNS}
    dtype.update(serviceName=str, operationName=str, startTime=str, duration=float)
    if is_test:
        dtype.update({c: int for c in _CSV_TEST_COLUMNS})
    return pd.read_csv(input_path, engine='c', usecols=list(dtype), dtype=dtype)


def df_to_trace_graphs(df, id_manager: TraceGraphIDManager, test_label: Optional[int] = None,
                       min_node_count: int = 2, max_node_count: int = 32,
                       merge_spans: bool = False) -> List[TraceGraph]:
    """Spans (one row each) -> trace graphs, the way the reference's preprocessing does it:
    rows with DBhash != 0 and spans of unknown services / operations are skipped; for test
    files a trace is labelled 1 (structure) / 2 (latency) / 0 and only `test_label` is kept;
    nodes attach to their parent span, orphans become roots of their own graph; nodes are
    visited in (service id, operation id, start time) order so children lists are ordered;
    graphs outside [min_node_count, max_node_count] are dropped (defaults 2 and 32: the
    reference uses default arguments here, not MAX_NODE_COUNT - SURVEY.md appendix D.10)."""
    from datetime import datetime, timedelta
    rows = df.to_dict(orient='records') if hasattr(df, 'to_dict') else list(df)
    traces: Dict[Tuple[int, int], Dict[int, tuple]] = {}
    for row in rows:
        if int(row['DBhash']) != 0:
            continue
        if test_label is not None:
            label = 1 if int(row['graphStructureLabel']) != 0 else (
                2 if int(row['graphLatencyLabel']) != 0 else 0)
            if label != test_label:
                continue
        service = str(row['serviceName'])
        operation = f"{row['serviceName']}/{row['operationName']}"
        if service not in id_manager.service_id or operation not in id_manager.operation_id:
            continue
        trace_id = (int(row['traceIdHigh']), int(row['traceIdLow']))
        start = datetime.strptime(str(row['startTime']), '%Y-%m-%d %H:%M:%S') + \
            timedelta(microseconds=int(row['nanosecond']) / 1_000)
        latency = float(row['duration'])
        node = TraceGraphNode(
            node_id=None, service_id=id_manager.service_id[service],
            operation_id=id_manager.operation_id[operation],
            features=TraceGraphNodeFeatures(span_count=1, avg_latency=latency,
                                            max_latency=latency, min_latency=latency),
            spans=[(int(row['spanId']), start, latency)])
        traces.setdefault(trace_id, {})[int(row['spanId'])] = (int(row['parentSpanId']), node, start)

    data = {} if not test_label else {
        'is_anomaly': True, 'anomaly_type': 'drop' if test_label == 1 else 'latency'}
    graphs: List[TraceGraph] = []
    for trace_id, spans in traces.items():
        ordered = sorted(spans.values(),
                         key=lambda t: (t[1].service_id, t[1].operation_id, t[2]))
        for parent_id, node, _start in ordered:
            if parent_id == 0 or parent_id not in spans:
                graphs.append(TraceGraph(version=TraceGraph.default_version(), trace_id=trace_id,
                                         parent_id=parent_id, root=node, node_count=None,
                                         max_depth=None, data=dict(data)))
            else:
                spans[parent_id][1].children.append(node)
    for g in graphs:
        if merge_spans:
            g.merge_spans_and_assign_id()
        else:
            g.assign_node_id()
    return [g for g in graphs if min_node_count <= g.node_count <= max_node_count]
