"""(a) batching: tv_batch_build against the NumPy oracle - every index array bit-exact, the
float outputs bit-exact too (float64 normalisation, then one cast)."""
import numpy as np
import pytest
import torch

from helpers import hand_made_flat, make_flat, oracle_batch, subset_flat

pytestmark = pytest.mark.gpu

INT_KEYS = ['enc_src', 'enc_dst', 'enc_indptr', 'enc_indices', 'enc_eids', 'enc_rev',
            'dec_src', 'dec_dst', 'dec_indptr', 'dec_indices', 'dec_eids', 'dec_rev']


def _device_batch_arrays(db):
    out = {}
    for name in ('src', 'dst', 'indptr', 'indices', 'eids', 'rev'):
        out['enc_' + name] = getattr(db.enc, name)
        out['dec_' + name] = getattr(db.dec, name)
    for name in ('adj', 'node_type_pad', 'latency_pad', 'node_count', 'node_type', 'latency',
                 'graph_id'):
        out[name] = getattr(db, name)
    return {k: v.cpu().numpy() for k, v in out.items()}


def _compare(flat, lr, V, M, ids=None):
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    store = DeviceTraceStore(flat, lr, V)
    if ids is None:
        ids = np.arange(len(flat))
    db = TraceGraphBatch.from_store(store, ids, M).build_dgl()
    got = _device_batch_arrays(db)
    want = oracle_batch(subset_flat(flat, ids), lr, V, M)
    for k in INT_KEYS + ['node_type_pad', 'node_count', 'node_type', 'graph_id']:
        assert got[k].dtype == np.int64, k
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    for k in ('adj', 'latency_pad', 'latency'):
        assert got[k].dtype == np.float32, k
        np.testing.assert_array_equal(got[k].view(np.uint32), want[k].view(np.uint32), err_msg=k)


def test_synthetic_batches(device):
    flat, lr = make_flat(64, 200, 32, seed=11)
    _compare(flat, lr, 200, 32)
    # a shuffled subset, gathered out of the store by index
    ids = np.random.RandomState(0).permutation(64)[:23]
    _compare(flat, lr, 200, 32, ids)


def test_reference_sample_dataset_traces(device):
    """The 17 traces of the reference's sample data set (arrays from the reference's own
    graph_vectors(), tests/golden) through tv_batch_build, MAX_NODE_COUNT = 32 and the data
    set's 1488 operations: bit-equal to the oracle, plus the structure facts the CPU suite
    checks on the oracle side (edge counts, one in-edge per padded slot, the normalisation)."""
    from helpers import golden_sample_flat
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    flat, lr, V, graphs = golden_sample_flat()
    assert len(graphs) == 17 and V == 1488
    _compare(flat, lr, V, 32)
    _compare(flat, lr, V, 32, ids=[16, 3, 3, 15, 0])
    db = TraceGraphBatch.from_store(DeviceTraceStore(flat, lr, V), np.arange(17), 32).build_dgl()
    n, B, M = int(flat.node_offsets[-1]), 17, 32
    assert db.enc.num_edges == 3 * n - 2 * B and db.dec.num_edges == B * M + 2 * (n - B)
    assert db.adj.shape == (B, 496) and float(db.adj.sum()) == n - B
    assert db.node_count.cpu().tolist() == [g['node_count'] for g in graphs]
    deg = np.diff(db.dec.indptr.cpu().numpy()).reshape(B, M)
    for i, g in enumerate(graphs):
        assert np.all(deg[i, g['node_count']:] == 1)
    mu, sd = lr[int(flat.node_type[0])]
    want = np.float32((np.float64(flat.avg_latency[0]) - mu) / (sd + 1e-5))
    assert db.latency.cpu().numpy()[0, 0] == want


def test_edge_topologies(device):
    M = 12
    shapes = [
        [0],                                  # n = 2 (smallest trace the reference keeps)
        [0] * (M - 1),                        # star with n = M
        list(range(M - 1)),                   # chain with n = M
        [0, 0, 1, 1, 2, 2, 3],                # binary-ish tree
        [0, 1, 1, 1, 2, 5],                   # mixed
        [],                                   # n = 1: a lone root (a generated trace can be one)
    ]
    flat, lr = hand_made_flat(shapes, 30, seed=5)
    _compare(flat, lr, 30, M)
    _compare(flat, lr, 30, M, ids=[4, 0, 5, 0, 2, 5])     # repeats and reordering


def test_edge_lists_that_are_not_in_bfs_order(device):
    """graph_vectors() emits edges grouped by parent in BFS order; any valid edge list (u < v)
    must still give the stable sort by destination (the general rank path of the kernel)."""
    from tracevae_amd.data import FlatTraces
    flat, lr = hand_made_flat([[0, 0, 1, 1, 2, 2, 3], [0, 1, 1, 1, 2, 5], [0] * 9], 30, seed=6)
    rng = np.random.RandomState(4)
    u, v = flat.u.copy(), flat.v.copy()
    for i in range(len(flat)):                         # shuffle the edges of every trace
        e0, e1 = int(flat.node_offsets[i]) - i, int(flat.node_offsets[i + 1]) - i - 1
        perm = e0 + rng.permutation(e1 - e0)
        u[e0:e1], v[e0:e1] = flat.u[perm], flat.v[perm]
    assert not np.array_equal(v, flat.v)
    shuffled = FlatTraces(node_offsets=flat.node_offsets, u=u, v=v, node_type=flat.node_type,
                          avg_latency=flat.avg_latency, max_latency=flat.max_latency,
                          min_latency=flat.min_latency, labels=flat.labels)
    _compare(shuffled, lr, 30, 12)
    _compare(shuffled, lr, 30, 300)                    # the four-wave launch


def test_large_max_node_count(device):
    flat, lr = make_flat(8, 500, 100, seed=2)
    _compare(flat, lr, 500, 100)
    _compare(flat, lr, 500, 500)                     # long-trace configuration (C5 shape)


def test_empty_batch(device):
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    flat, lr = make_flat(4, 50, 8, seed=1)
    store = DeviceTraceStore(flat, lr, 50)
    db = TraceGraphBatch.from_store(store, np.zeros(0, np.int64), 8).build_dgl()
    assert db.batch_size == 0 and db.num_nodes == 0
    assert db.enc.indptr.cpu().tolist() == [0] and db.dec.indptr.cpu().tolist() == [0]


def test_errors(device):
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    from tracevae_amd.data import TraceGraphLatencyRangeFile
    flat, lr = hand_made_flat([[0, 0, 1], [0]], 20, seed=3)
    store = DeviceTraceStore(flat, lr, 20)
    with pytest.raises(ValueError):                  # more nodes than MAX_NODE_COUNT
        TraceGraphBatch.from_store(store, [0, 1], 3).build_dgl()
    # an operation missing from latency_range: the reference raises KeyError (dataset.py:77)
    mean, std = lr.tables(20)
    mean[int(flat.node_type[1])] = np.nan
    store2 = DeviceTraceStore(flat, TraceGraphLatencyRangeFile.from_tables(mean, std), 20)
    with pytest.raises(KeyError):
        TraceGraphBatch.from_store(store2, [0, 1], 8).build_dgl()


def test_from_trace_graph_objects(device):
    """The reference's constructor form: a list of TraceGraph objects."""
    from tracevae_amd.batch import TraceGraphBatch
    from tracevae_amd.data import flatten_trace_graphs
    from tracevae_amd.synth import SyntheticSpec, generate_trace_graphs, make_latency_range
    spec = SyntheticSpec(num_operations=80, max_spans=20, seed=9)
    graphs = generate_trace_graphs(spec, 10)
    lr = make_latency_range(spec)
    G = TraceGraphBatch(latency_range=lr, trace_graphs=graphs, num_operations=80)
    G._max_node_count = 20
    db = G.build_dgl()
    want = oracle_batch(flatten_trace_graphs(graphs), lr, 80, 20)
    got = _device_batch_arrays(db)
    for k in INT_KEYS:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    np.testing.assert_array_equal(got['adj'], want['adj'])


def test_compact_latency_decoder_graph(device):
    """tv_batch_build's compact latency-decoder graph: the encoder graph plus one self-loop
    node per trace, with the BatchNorm weight of every row."""
    from tracevae_amd.batch import DeviceTraceStore, build_device_batch
    V, M = 30, 12
    flat, lr = make_flat(9, V, M, seed=21)
    db = build_device_batch(DeviceTraceStore(flat, lr, V), np.arange(len(flat)), M)
    B, N = db.batch_size, db.num_nodes
    E = db.enc.num_edges
    lat = db.lat
    assert lat.num_rows == N + B and lat.num_edges == E + B
    for name in ('indptr', 'indices', 'eids', 'rev', 'src', 'dst'):
        full, enc = getattr(lat, name).cpu(), getattr(db.enc, name).cpu()
        assert torch.equal(full[:enc.shape[0]], enc), name
    tail = torch.arange(B)
    assert torch.equal(lat.indptr.cpu()[N:], E + torch.arange(B + 1))
    for name in ('indices', 'src', 'dst'):
        assert torch.equal(getattr(lat, name).cpu()[E:], N + tail), name
    for name in ('eids', 'rev'):
        assert torch.equal(getattr(lat, name).cpu()[E:], E + tail), name
    w = db.lat_row_weight.cpu()
    assert torch.equal(w[:N], torch.ones(N))
    assert torch.equal(w[N:], (M - db.node_count.cpu()).float())
    assert float(w.sum()) == B * M
    assert torch.equal(db.lat_row_graph.cpu(), torch.cat([db.graph_id.cpu(), tail]))
    assert torch.equal(db.lat_row_type.cpu(), torch.cat([db.node_type.cpu(), torch.zeros(B, dtype=torch.long)]))
