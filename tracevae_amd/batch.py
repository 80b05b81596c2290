"""Mini-batch container of the accelerated path.

`TraceGraphBatch` keeps the reference's constructor and `build_dgl()` entry point
(`tracegnn/models/trace_vae/types.py:14-62`); what `build_dgl()` builds is no longer a DGL
object but the device-side layout of `tv_batch_build` (`csrc/batch_build.hip`): batched COO
in DGL edge order, destination-sorted CSC for the encoder graph and for the padded decoder
graph, and the observed tensors of `TraceVAE.p` (`model/trace_vae.py:121-140`).

Two ways to make one:

* `TraceGraphBatch(id_manager=..., latency_range=..., trace_graphs=[TraceGraph, ...])`
  - the reference's form; the graphs are collated on the host (`graph_vectors()`), copied
    to the GPU once, and the rest happens in the kernel;
* `TraceGraphBatch.from_store(store, trace_ids)` - the traces already live in HBM
  (`DeviceTraceStore`); a batch is a list of trace indices and nothing but two small index
  arrays crosses PCIe.
"""
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import constants as C
from . import hip
from .data import (FlatTraces, TraceGraph, TraceGraphIDManager, TraceGraphLatencyRangeFile,
                   flatten_trace_graphs)
from .ops import DeviceGraph

__all__ = ['DeviceTraceStore', 'DeviceBatch', 'TraceGraphBatch', 'build_sampled_graphs']

_STATUS_TEXT = {
    1: 'a trace has more than MAX_NODE_COUNT nodes',
    2: 'an edge violates 0 <= u < v < node_count',
    4: 'an operation id has no entry in latency_range (reference: KeyError in latency_range[...])',
    8: 'a node_type is outside [0, num_operations)',
}


def _current_device() -> torch.device:
    if not torch.cuda.is_available():
        raise hip.HipLibraryError(
            'no GPU is visible: the accelerated batching/model path runs on MI355X only '
            '(there is no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


class DeviceTraceStore(object):
    """A `FlatTraces` data set resident in HBM, plus the per-operation latency tables."""

    def __init__(self, flat: FlatTraces, latency_range: TraceGraphLatencyRangeFile,
                 num_operations: int, device: Optional[torch.device] = None,
                 validate: bool = True):
        if validate:
            flat.validate()
        self.device = device or _current_device()
        self.num_operations = int(num_operations)
        self.host = flat
        self.node_offsets_host = np.ascontiguousarray(flat.node_offsets, dtype=np.int64)
        self.node_counts_host = np.diff(self.node_offsets_host)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        self.u = t(flat.u, np.int64)
        self.v = t(flat.v, np.int64)
        self.node_type = t(flat.node_type, np.int64)
        self.avg_latency = t(flat.avg_latency, np.float32)
        self.min_latency = t(flat.min_latency, np.float32)
        self.max_latency = t(flat.max_latency, np.float32)
        mean, std = latency_range.tables(self.num_operations)
        self.lat_mean = torch.as_tensor(mean).to(self.device)
        self.lat_std = torch.as_tensor(std).to(self.device)
        self.labels = flat.labels
        # node types / latency ranges are checked once here; a batch only needs the kernel's
        # status word (a host sync) when the store is known to hold a problem
        used = np.unique(flat.node_type) if len(flat.node_type) else np.zeros(0, np.int64)
        in_range = (used >= 0) & (used < self.num_operations)
        self.needs_status_check = bool((~in_range).any()) or bool(
            np.isnan(mean[used[in_range]]).any() or np.isnan(std[used[in_range]]).any())

    def __len__(self) -> int:
        return len(self.node_counts_host)


class DeviceBatch(object):
    """What `tv_batch_build` produced for one mini-batch (all tensors on the GPU)."""
    __slots__ = [
        'batch_size', 'num_nodes', 'max_node_count', 'node_offsets', 'node_offsets_host',
        'enc', 'dec', 'adj', 'node_type_pad', 'latency_pad', 'node_count', 'node_type',
        'latency', 'graph_id', 'status',
        # compact latency-decoder graph: rows [real nodes | one per trace], see tv_batch_build
        'lat', 'lat_row_weight', 'lat_row_graph', 'lat_row_type',
    ]

    def batch_num_nodes(self) -> torch.Tensor:
        return self.node_count

    def check_status(self) -> None:
        """Read the kernel's status word back (a host sync) and raise like the reference
        would: `KeyError` for a missing latency range, `ValueError` otherwise."""
        code = int(self.status.item())
        if code == 0:
            return
        msgs = [text for bit, text in _STATUS_TEXT.items() if code & bit]
        if code & 4:
            raise KeyError('; '.join(msgs))
        raise ValueError('; '.join(msgs))


def build_device_batch(store: DeviceTraceStore, trace_ids: np.ndarray,
                       max_node_count: Optional[int] = None, check: bool = True) -> DeviceBatch:
    m = int(max_node_count or C.MAX_NODE_COUNT)
    dev = store.device
    ids = np.ascontiguousarray(trace_ids, dtype=np.int64)
    if ids.ndim != 1:
        raise ValueError('trace_ids must be one-dimensional')
    if len(ids) and (ids.min() < 0 or ids.max() >= len(store)):
        raise IndexError('trace id out of range')
    b = len(ids)
    counts = store.node_counts_host[ids]
    if b and counts.max() > m:
        raise ValueError(f'a trace has {int(counts.max())} nodes > MAX_NODE_COUNT={m}')
    off = np.zeros([b + 1], dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    n = int(off[-1])
    node_start = store.node_offsets_host[ids]
    edge_start = node_start - ids
    # one small pinned upload: [node_offsets | node_start | edge_start]
    meta = torch.from_numpy(np.concatenate([off, node_start, edge_start])).to(dev, non_blocking=True)
    d_off, d_nstart, d_estart = meta[:b + 1], meta[b + 1:2 * b + 1], meta[2 * b + 1:]

    e_enc, e_dec, t = 3 * n - 2 * b, b * m + 2 * (n - b), m * (m - 1) // 2
    i64 = lambda *shape: torch.empty(shape, dtype=torch.int64, device=dev)
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    # one allocation for all index outputs of each graph keeps the allocator quiet
    # the encoder arrays carry B more entries: with them they are the compact latency-decoder
    # graph (encoder graph + one self-loop node per trace); the encoder graph is the prefix
    lat_src, lat_dst, lat_indices, lat_eids, lat_rev = (i64(e_enc + b) for _ in range(5))
    lat_indptr = i64(n + b + 1)
    enc_src, enc_dst, enc_indices, enc_eids, enc_rev = (
        t[:e_enc] for t in (lat_src, lat_dst, lat_indices, lat_eids, lat_rev))
    enc_indptr = lat_indptr[:n + 1]
    dec_src, dec_dst, dec_indices, dec_eids, dec_rev = (i64(e_dec) for _ in range(5))
    dec_indptr = i64(b * m + 1)
    out = DeviceBatch()
    out.batch_size, out.num_nodes, out.max_node_count = b, n, m
    out.node_offsets, out.node_offsets_host = d_off, off
    out.adj = f32(b, t)
    out.node_type_pad = i64(b, m)
    out.latency_pad = f32(b, m, 1)
    out.node_count = i64(b)
    out.lat_row_type = i64(n + b)
    out.lat_row_graph = i64(n + b)
    out.lat_row_weight = f32(n + b)
    out.node_type = out.lat_row_type[:n]
    out.latency = f32(n, 3)
    out.graph_id = out.lat_row_graph[:n]
    out.status = torch.zeros([1], dtype=torch.int32, device=dev)
    p = hip.ptr
    if b == 0:
        # nothing to launch: the only defined entries are the two CSC terminators
        enc_indptr.zero_()
        dec_indptr.zero_()
        out.enc = DeviceGraph(enc_indptr, enc_indices, enc_eids, enc_rev, enc_src, enc_dst)
        out.dec = DeviceGraph(dec_indptr, dec_indices, dec_eids, dec_rev, dec_src, dec_dst)
        out.lat = out.enc
        return out
    hip.call('tv_batch_build', b, m, n, store.num_operations, p(d_off), p(d_nstart), p(d_estart),
             p(store.u), p(store.v), p(store.node_type), p(store.avg_latency),
             p(store.min_latency), p(store.max_latency), p(store.lat_mean), p(store.lat_std),
             p(enc_src), p(enc_dst), p(enc_indptr), p(enc_indices), p(enc_eids), p(enc_rev),
             p(dec_src), p(dec_dst), p(dec_indptr), p(dec_indices), p(dec_eids), p(dec_rev),
             p(out.adj), p(out.node_type_pad), p(out.latency_pad), p(out.node_count),
             p(out.node_type), p(out.latency), p(out.graph_id), p(out.lat_row_weight),
             p(out.status))
    out.enc = DeviceGraph(enc_indptr, enc_indices, enc_eids, enc_rev, enc_src, enc_dst)
    out.lat = DeviceGraph(lat_indptr, lat_indices, lat_eids, lat_rev, lat_src, lat_dst)
    out.dec = DeviceGraph(dec_indptr, dec_indices, dec_eids, dec_rev, dec_src, dec_dst)
    if check and store.needs_status_check:
        out.check_status()
    return out


def build_sampled_graphs(adj: torch.Tensor, max_node_count: int):
    """Decoder graphs of SAMPLED adjacencies: `make_graph` + `dgl.batch` of
    `TraceStructVAE.p` without an observed graph (model/struct_vae.py:296-333), on the GPU
    (`tv_sampled_graph_count` / `tv_sampled_graph_fill`).

    `adj` [G, M(M-1)/2] float32 with 0/1 entries.  Returns `(graph, node_idx)`: the batched
    `DeviceGraph` (G*M nodes) and the `node_idx` node data [G*M].  One host read-back (the
    edge total sizes the arrays), as the reference's `T.where` has per sample."""
    m = int(max_node_count)
    t = m * (m - 1) // 2
    if adj.dim() != 2 or adj.shape[1] != t:
        raise ValueError(f'adj must be [graphs, {t}] for MAX_NODE_COUNT={m}: got {tuple(adj.shape)}')
    if not adj.is_cuda:
        raise hip.HipLibraryError('build_sampled_graphs runs on the GPU only (no CPU fallback)')
    adj = adj.to(torch.float32).contiguous()
    dev, g = adj.device, int(adj.shape[0])
    i64 = lambda n: torch.empty([n], dtype=torch.int64, device=dev)
    indptr, offs = i64(g * m + 1), i64(g + 1)
    ws_bytes = hip.query('tv_sampled_graph_workspace_bytes', g, m)
    ws = torch.empty([max(ws_bytes, 8) // 8 + 1], dtype=torch.int64, device=dev)
    status = torch.zeros([1], dtype=torch.int32, device=dev)
    p = hip.ptr
    hip.call('tv_sampled_graph_count', g, m, p(adj), p(indptr), p(offs), p(ws), ws_bytes,
             p(status))
    e, code = int(offs[-1].item()), int(status.item())
    if code:
        raise ValueError('a sampled adjacency entry is neither 0 nor 1')
    src, dst, indices, eids, rev = (i64(e) for _ in range(5))
    node_idx = i64(g * m)
    hip.call('tv_sampled_graph_fill', g, m, e, p(adj), p(offs), p(indptr), p(src), p(dst),
             p(indices), p(eids), p(rev), p(node_idx), p(ws), ws_bytes)
    return DeviceGraph(indptr, indices, eids, rev, src, dst), node_idx


class TraceGraphBatch(object):
    """Drop-in for the reference's `TraceGraphBatch` (types.py:14-62).  The `dgl_graphs` /
    `dgl_batch` slots of the reference hold DGL objects; here `dgl_batch` holds the
    `DeviceBatch` (same role: the batched device-side graph + node data)."""
    __slots__ = ['id_manager', 'latency_range', 'trace_graphs', 'dgl_graphs', 'dgl_batch',
                 '_store', '_trace_ids', '_max_node_count', '_stream_index']

    def __init__(self, *, id_manager: Optional[TraceGraphIDManager] = None,
                 latency_range: Optional[TraceGraphLatencyRangeFile] = None,
                 trace_graphs: Optional[List[TraceGraph]] = None,
                 dgl_graphs=None, dgl_batch: Optional[DeviceBatch] = None,
                 num_operations: Optional[int] = None):
        if ((trace_graphs is None) or (id_manager is None and num_operations is None)) and \
                (dgl_batch is None):
            raise ValueError('Insufficient arguments.')
        self.id_manager = id_manager
        self.latency_range = latency_range
        self.trace_graphs = trace_graphs
        self.dgl_graphs = dgl_graphs
        self.dgl_batch = dgl_batch
        self._store = None
        self._trace_ids = None
        self._max_node_count = None
        self._stream_index = None
        if trace_graphs is not None and dgl_batch is None:
            v = num_operations if num_operations is not None else id_manager.num_operations
            if latency_range is None:
                raise ValueError('`latency_range` is required to build the batch '
                                 '(latencies are normalised per operation).')
            flat = flatten_trace_graphs(list(trace_graphs))
            # the host-side checks of the reference path, before anything reaches the GPU
            flat.validate()
            self._store = DeviceTraceStore(flat, latency_range, v, validate=False)
            self._trace_ids = np.arange(len(flat), dtype=np.int64)

    @staticmethod
    def from_store(store: DeviceTraceStore, trace_ids: Union[np.ndarray, Sequence[int]],
                   max_node_count: Optional[int] = None) -> 'TraceGraphBatch':
        self = TraceGraphBatch.__new__(TraceGraphBatch)
        self.id_manager = self.latency_range = self.trace_graphs = None
        self.dgl_graphs = self.dgl_batch = None
        self._store = store
        self._trace_ids = np.asarray(trace_ids, dtype=np.int64)
        self._max_node_count = max_node_count
        self._stream_index = None      # position in its StoreStream (seeds reproducible noise)
        return self

    def build_dgl(self, add_self_loop: bool = True, directed: Union[bool, str] = False,
                  check: bool = True) -> DeviceBatch:
        if not add_self_loop or directed is not False:
            raise NotImplementedError(
                'only add_self_loop=True, directed=False is built: it is the only form '
                'TraceVAE uses (model/trace_vae.py:77-82)')
        if self.dgl_batch is None:
            self.dgl_batch = build_device_batch(self._store, self._trace_ids,
                                                self._max_node_count, check=check)
        return self.dgl_batch

    @property
    def labels(self) -> Optional[np.ndarray]:
        if self._store is not None and self._store.labels is not None:
            return self._store.labels[self._trace_ids]
        return None

    def __len__(self) -> int:
        if self.dgl_batch is not None:
            return self.dgl_batch.batch_size
        return len(self._trace_ids)
