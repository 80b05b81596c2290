"""NumPy restatement of the reference batching step (TEST INFRASTRUCTURE ONLY).

Plain per-trace Python loops, like the reference; follows

* `trace_graph_to_dgl`            tracegnn/models/trace_vae/dataset.py:20-89
* `TraceGraphBatch.build_dgl`     tracegnn/models/trace_vae/types.py:42-62
* `dense_triu_adj`/`dense_to_triu`  tensor_utils.py:79-89, 64-67
* `pad_node_feature`              tensor_utils.py:92-114
* decoder graph rebuild           model/struct_vae.py:335-362
* graphs of sampled adjacencies   model/struct_vae.py:296-333, tensor_utils.py:70-76

and the DGL graph semantics of SURVEY.md appendix A0: `dgl.graph` keeps COO order (edge
id = position), `add_self_loop` / `add_reverse_edges` append, `dgl.batch` shifts node ids
by the cumulative node count and concatenates edges graph by graph, and message passing
reduces over the in-edges of each destination in a CSC built by a *stable* sort on dst.
"""
from typing import Dict, List, Sequence

import numpy as np

__all__ = [
    'encoder_coo', 'decoder_coo', 'csc_from_coo', 'reverse_positions', 'dense_triu_adj',
    'normalize_latency', 'build_batch', 'triu_to_dense', 'sampled_graphs',
]


def encoder_coo(u: np.ndarray, v: np.ndarray, n: int):
    """dataset.py:39-52 with directed=False, add_self_loop=True:
    edges [u->v ; v->u ; (i,i) for i<n]."""
    loop = np.arange(n, dtype=np.int64)
    src = np.concatenate([u, v, loop]).astype(np.int64)
    dst = np.concatenate([v, u, loop]).astype(np.int64)
    return src, dst


def decoder_coo(src: np.ndarray, dst: np.ndarray, max_node_count: int):
    """struct_vae.py:338-347: keep the encoder edges with src < dst, rebuild on
    `max_node_count` nodes, `add_reverse_edges`, `add_self_loop`."""
    keep = src < dst
    s, d = src[keep], dst[keep]
    loop = np.arange(max_node_count, dtype=np.int64)
    return (np.concatenate([s, d, loop]).astype(np.int64),
            np.concatenate([d, s, loop]).astype(np.int64))


def csc_from_coo(src: np.ndarray, dst: np.ndarray, num_nodes: int):
    """Destination-major layout: stable sort of the COO by dst (appendix A0)."""
    order = np.argsort(dst, kind='stable').astype(np.int64)
    counts = np.bincount(dst, minlength=num_nodes).astype(np.int64)
    indptr = np.zeros([num_nodes + 1], dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr, src[order].astype(np.int64), order


def reverse_positions(indptr: np.ndarray, indices: np.ndarray) -> np.ndarray:
    """For CSC entry p = (src s -> dst d) the CSC position of the opposite edge (d -> s).
    The graphs here are symmetric with one edge per ordered pair, so it always exists."""
    num_nodes = len(indptr) - 1
    where: Dict[tuple, int] = {}
    for d in range(num_nodes):
        for p in range(indptr[d], indptr[d + 1]):
            where[(int(indices[p]), d)] = p
    rev = np.empty_like(indices)
    for d in range(num_nodes):
        for p in range(indptr[d], indptr[d + 1]):
            rev[p] = where[(d, int(indices[p]))]
    return rev


def dense_triu_adj(src: np.ndarray, dst: np.ndarray, max_node_count: int) -> np.ndarray:
    """tensor_utils.py:79-89 + 64-67: adj[src, dst] = 1 on an M x M zero matrix, then the
    strict upper triangle, row-major."""
    adj = np.zeros([max_node_count, max_node_count], dtype=np.float32)
    adj[src, dst] = 1
    mask = np.triu(np.ones([max_node_count, max_node_count], dtype=bool), 1)
    return adj.reshape(-1)[mask.reshape(-1)]


def normalize_latency(latency: np.ndarray, node_type: np.ndarray,
                      mean: np.ndarray, std: np.ndarray) -> np.ndarray:
    """dataset.py:73-79: `(latency[i] - mu) / (std + 1e-5)` with mu, std Python floats,
    i.e. float64 arithmetic (NumPy 1.x promotion), then cast to float32."""
    out = np.empty([len(latency)], dtype=np.float32)
    for i in range(len(latency)):
        mu, sd = float(mean[node_type[i]]), float(std[node_type[i]])
        if np.isnan(mu) or np.isnan(sd):
            raise KeyError(int(node_type[i]))
        out[i] = np.float32((np.float64(latency[i]) - mu) / (sd + 1e-5))
    return out


def build_batch(node_offsets: np.ndarray, u: np.ndarray, v: np.ndarray,
                node_type: np.ndarray, avg_latency: np.ndarray, max_latency: np.ndarray,
                min_latency: np.ndarray, lat_mean: np.ndarray, lat_std: np.ndarray,
                max_node_count: int) -> Dict[str, np.ndarray]:
    """Everything the batching step hands to the model, for traces given in the flat
    layout (trace i: nodes node_offsets[i]:node_offsets[i+1], edges at the same range
    shifted by -i)."""
    b = len(node_offsets) - 1
    m = int(max_node_count)
    enc_src: List[np.ndarray] = []
    enc_dst: List[np.ndarray] = []
    dec_src: List[np.ndarray] = []
    dec_dst: List[np.ndarray] = []
    adj = np.zeros([b, m * (m - 1) // 2], dtype=np.float32)
    node_type_pad = np.zeros([b, m], dtype=np.int64)
    latency_pad = np.zeros([b, m, 1], dtype=np.float32)
    node_count = np.zeros([b], dtype=np.int64)
    latency = np.zeros([int(node_offsets[-1]), 3], dtype=np.float32)
    graph_id = np.zeros([int(node_offsets[-1])], dtype=np.int64)

    for i in range(b):
        n0, n1 = int(node_offsets[i]), int(node_offsets[i + 1])
        e0, e1 = n0 - i, n1 - (i + 1)
        n = n1 - n0
        if n > m:
            raise ValueError(f'trace {i} has {n} nodes > MAX_NODE_COUNT={m}')
        s, d = encoder_coo(u[e0:e1], v[e0:e1], n)
        enc_src.append(s + n0)           # dgl.batch: shift by cumulative node count
        enc_dst.append(d + n0)
        ds, dd = decoder_coo(s, d, m)
        dec_src.append(ds + i * m)
        dec_dst.append(dd + i * m)
        adj[i] = dense_triu_adj(s, d, m)
        node_type_pad[i, :n] = node_type[n0:n1]
        node_count[i] = n
        graph_id[n0:n1] = i
        # latency = [avg, min, max] (dataset.py:80-87)
        for col, arr in enumerate((avg_latency, min_latency, max_latency)):
            latency[n0:n1, col] = normalize_latency(arr[n0:n1], node_type[n0:n1], lat_mean, lat_std)
        latency_pad[i, :n, 0] = latency[n0:n1, 0]

    z = np.zeros([0], dtype=np.int64)
    es = np.concatenate(enc_src) if b else z
    ed = np.concatenate(enc_dst) if b else z
    ds_ = np.concatenate(dec_src) if b else z
    dd_ = np.concatenate(dec_dst) if b else z
    n_total = int(node_offsets[-1])
    enc_indptr, enc_indices, enc_eids = csc_from_coo(es, ed, n_total)
    dec_indptr, dec_indices, dec_eids = csc_from_coo(ds_, dd_, b * m)
    return dict(
        enc_src=es, enc_dst=ed, enc_indptr=enc_indptr, enc_indices=enc_indices,
        enc_eids=enc_eids, enc_rev=reverse_positions(enc_indptr, enc_indices),
        dec_src=ds_, dec_dst=dd_, dec_indptr=dec_indptr, dec_indices=dec_indices,
        dec_eids=dec_eids, dec_rev=reverse_positions(dec_indptr, dec_indices),
        adj=adj, node_type_pad=node_type_pad, latency_pad=latency_pad,
        node_count=node_count, node_type=node_type[:n_total].astype(np.int64),
        latency=latency, graph_id=graph_id,
    )


def triu_to_dense(triu: np.ndarray, max_node_count: int, pad_value=0) -> np.ndarray:
    """tensor_utils.py:70-76: the row-major strict upper triangle back into an M x M matrix."""
    m = max_node_count
    out = np.full([m, m], pad_value, dtype=triu.dtype)
    out[np.triu(np.ones([m, m], dtype=bool), 1)] = triu
    return out


def sampled_graphs(adj_triu: np.ndarray, max_node_count: int) -> Dict[str, np.ndarray]:
    """`make_graph` of TraceStructVAE.p (model/struct_vae.py:298-316) for each sampled
    adjacency, then `dgl.batch` (:320-331): edges `where(adj)` in row-major order, the reversed
    list appended (`add_reverse_edges`), M self loops appended (`add_self_loop`); node ids and
    edge ids shifted graph by graph; `node_idx = max(max_u(adj * cumsum(adj, -1)) - 1, 0)`."""
    m = max_node_count
    srcs: List[np.ndarray] = []
    dsts: List[np.ndarray] = []
    node_idx = []
    offsets = [0]
    for g, row in enumerate(np.asarray(adj_triu)):
        adj = triu_to_dense(row, m)
        u, v = np.nonzero(adj)
        loop = np.arange(m, dtype=np.int64)
        srcs.append(np.concatenate([u, v, loop]).astype(np.int64) + g * m)
        dsts.append(np.concatenate([v, u, loop]).astype(np.int64) + g * m)
        offsets.append(offsets[-1] + len(srcs[-1]))
        ranks = (adj * np.cumsum(adj, axis=-1)).max(axis=-2).astype(np.int64) - 1
        node_idx.append(np.maximum(ranks, 0))
    n_graphs = len(offsets) - 1
    src = np.concatenate(srcs) if srcs else np.zeros([0], dtype=np.int64)
    dst = np.concatenate(dsts) if dsts else np.zeros([0], dtype=np.int64)
    indptr, indices, eids = csc_from_coo(src, dst, n_graphs * m)
    return dict(src=src, dst=dst, indptr=indptr, indices=indices, eids=eids,
                rev=reverse_positions(indptr, indices),
                node_idx=np.concatenate(node_idx) if node_idx else np.zeros([0], dtype=np.int64),
                graph_edge_offsets=np.asarray(offsets, dtype=np.int64))
