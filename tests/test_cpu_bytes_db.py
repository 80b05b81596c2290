"""The reader for data sets the reference has already processed (sqlite + snappy + pickled
TraceGraph): snappy decoder against hand-assembled streams of every element kind, encoder by
round trip, the unpickler against pickles written by stand-in classes that have the reference's
module path and `__slots__` layout, and the database classes end to end.

Parity unpinned: the reference ships no `_bytes.db` file and python-snappy is not installed,
so no stream here was produced by the reference's own writer.
"""
import os
import pickle
import sys
import types
from dataclasses import dataclass
from datetime import datetime

import numpy as np
import pytest

from tracevae_amd import bytes_db
from tracevae_amd.bytes_db import (BytesMultiDB, BytesSqliteDB, TraceGraphDB, dumps_trace_graph,
                                   loads_trace_graph, open_trace_graph_db, snappy_compress,
                                   snappy_decompress)
from tracevae_amd.data import flatten_trace_graphs
from tracevae_amd.synth import SyntheticSpec, generate_trace_graphs


def test_snappy_elements():
    # literal (short tag) + copy with 1-byte offset, overlapping: "abc" then 8 bytes from 3 back
    assert snappy_decompress(b'\x0b\x08abc\x11\x03') == b'abcabcabcab'
    # literal with a 1-byte length field (tag 60): 100 bytes
    lit = bytes(range(100))
    assert snappy_decompress(b'\x64' + bytes([60 << 2, 99]) + lit) == lit
    # literal with a 2-byte length field (tag 61): 300 bytes; header varint takes two bytes
    lit = bytes(i % 251 for i in range(300))
    assert snappy_decompress(b'\xac\x02' + bytes([61 << 2, 0x2b, 0x01]) + lit) == lit
    # copy with 2-byte offset: "0123456789" then 10 bytes from 10 back
    assert snappy_decompress(b'\x14\x240123456789' + bytes([(9 << 2) | 2, 10, 0])) == b'0123456789' * 2
    # copy with 4-byte offset, and a 1-byte offset using its high bits (offset 256 + 44 = 300)
    body = bytes(i % 251 for i in range(300))
    stream = b'\xb4\x02' + bytes([61 << 2, 0x2b, 0x01]) + body \
        + bytes([(3 << 2) | 3, 0x2c, 0x01, 0, 0]) + bytes([(1 << 5) | (0 << 2) | 1, 44])
    assert snappy_decompress(stream) == body + body[:4] + (body + body[:4])[4:8]
    assert snappy_decompress(b'\x00') == b''


@pytest.mark.parametrize('stream', [
    b'',                                    # no length
    b'\x05\x08abc',                         # shorter than announced
    b'\x03\x08ab',                          # truncated literal
    b'\x04\x00a\x09\x05',                   # copy from before the start
    b'\x04\x00a\x01\x00',                   # zero offset
    b'\xff\xff\xff\xff\xff\xff\x01',        # length header too long
])
def test_snappy_rejects_bad_streams(stream):
    with pytest.raises(ValueError):
        snappy_decompress(stream)


def test_snappy_round_trip():
    rs = np.random.RandomState(0)
    cases = [b'', b'a', b'abcd', b'ab' * 5000, bytes(200000), rs.bytes(70000),
             rs.bytes(100) * 300, b'x' * 59 + b'y' * 61 + b'z' * 257]
    for data in cases:
        packed = snappy_compress(data)
        assert snappy_decompress(packed) == data
    assert len(snappy_compress(bytes(200000))) < 12000


@pytest.fixture
def reference_classes():
    """Stand-ins living at the reference's module path, with its `__slots__` layout
    (trace_graph.py:29-190) - what a pickle written by the reference names."""
    saved = {k: sys.modules.get(k) for k in ('tracegnn', 'tracegnn.data', 'tracegnn.data.trace_graph')}
    mod = None
    for k in saved:
        mod = types.ModuleType(k)
        sys.modules[k] = mod

    def make(name, slots):
        cls = dataclass(type(name, (object,), {
            '__slots__': slots, '__annotations__': {s: object for s in slots},
            '__module__': 'tracegnn.data.trace_graph', '__qualname__': name}))
        setattr(mod, name, cls)
        return cls

    ns = types.SimpleNamespace(
        Features=make('TraceGraphNodeFeatures', ['span_count', 'max_latency', 'min_latency', 'avg_latency']),
        Span=make('TraceGraphSpan', ['span_id', 'start_time', 'latency']),
        Node=make('TraceGraphNode', ['node_id', 'service_id', 'operation_id', 'features', 'children',
                                     'spans', 'scores', 'anomaly']),
        Graph=make('TraceGraph', ['version', 'trace_id', 'parent_id', 'root', 'node_count',
                                  'max_depth', 'data']))
    yield ns
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _reference_graph(ns, version=2):
    def node(i, op, lat, children=()):
        return ns.Node(node_id=i, service_id=1, operation_id=op,
                       features=ns.Features(span_count=1, max_latency=lat, min_latency=lat, avg_latency=lat),
                       children=list(children),
                       spans=[ns.Span(span_id=100 + i, start_time=datetime(2021, 5, 1, 12, 0, i), latency=lat)],
                       scores=None, anomaly=None)
    root = node(0, 7, 30.0, [node(1, 3, 10.0, [node(3, 9, 2.5)]), node(2, 5, np.float64(12.0))])
    return ns.Graph(version=version, trace_id=(1, 2), parent_id=0, root=root, node_count=4,
                    max_depth=2, data={'is_anomaly': True, 'anomaly_type': 'drop'})


@pytest.mark.parametrize('protocol', [2, 3, 4, 5])
def test_loads_reference_pickle(reference_classes, protocol):
    g = loads_trace_graph(pickle.dumps(_reference_graph(reference_classes), protocol=protocol))
    assert type(g).__module__ == 'tracevae_amd.data'
    assert (g.version, g.trace_id, g.node_count, g.max_depth) == (2, (1, 2), 4, 2)
    gv = g.graph_vectors()
    assert gv.u.tolist() == [0, 0, 1] and gv.v.tolist() == [1, 2, 3]
    assert gv.node_type.tolist() == [7, 3, 5, 9]
    assert gv.avg_latency.tolist() == [30.0, 10.0, 12.0, 2.5]
    assert g.root.spans[0].start_time == datetime(2021, 5, 1, 12, 0, 0)
    assert flatten_trace_graphs([g]).labels.tolist() == [1]


def test_loads_old_versions(reference_classes):
    """trace_graph.py:383-396: graphs written before `version` / `scores` / `anomaly` existed."""
    ns = reference_classes
    g = _reference_graph(ns)
    del g.version
    for nd in (g.root, *g.root.children, g.root.children[0].children[0]):
        del nd.scores
        del nd.anomaly
    out = loads_trace_graph(pickle.dumps(g, protocol=3))
    assert out.version == 2
    assert all(nd.scores is None and nd.anomaly is None for _, nd in out.iter_bfs())
    g = _reference_graph(ns, version=1)
    del g.root.anomaly
    out = loads_trace_graph(pickle.dumps(g, protocol=3))
    assert out.version == 2 and out.root.anomaly is None


def test_unpickler_refuses_other_globals():
    for payload in (pickle.dumps(os.getcwd), pickle.dumps(print), pickle.dumps(types.SimpleNamespace(a=1))):
        with pytest.raises(pickle.UnpicklingError):
            loads_trace_graph(payload)
    with pytest.raises(pickle.UnpicklingError):         # a harmless pickle that is not a graph
        loads_trace_graph(pickle.dumps({'a': 1}))


def test_dumps_names_reference_classes():
    g = generate_trace_graphs(SyntheticSpec(num_operations=40, max_spans=12, seed=3), 1)[0]
    blob = dumps_trace_graph(g)
    assert b'tracegnn.data.trace_graph' in blob and b'tracevae_amd' not in blob
    assert type(g).__module__ == 'tracevae_amd.data' and 'tracegnn' not in sys.modules
    back = loads_trace_graph(blob)
    for k in ('u', 'v', 'node_type', 'avg_latency', 'max_latency', 'min_latency'):
        np.testing.assert_array_equal(getattr(back.graph_vectors(), k), getattr(g.graph_vectors(), k))


def test_dumps_is_loadable_by_slot_classes(reference_classes):
    """What this package writes must load into classes shaped like the reference's."""
    g = generate_trace_graphs(SyntheticSpec(num_operations=40, max_spans=12, seed=4), 1)[0]
    g.root.spans = None
    blob = dumps_trace_graph(g)
    back = pickle.loads(blob)                                   # resolves to the stand-ins
    assert type(back) is reference_classes.Graph
    assert back.node_count == g.node_count and back.root.operation_id == g.root.operation_id


def test_databases(tmp_path):
    spec = SyntheticSpec(num_operations=60, max_spans=16, seed=5)
    graphs = generate_trace_graphs(spec, 30)
    root = tmp_path / 'ds'
    for name, part in (('train', graphs[:20]), ('val', graphs[20:])):
        with TraceGraphDB(BytesSqliteDB(str(root / 'processed' / name), write=True)) as db:
            with db.write_batch():
                for g in part:
                    db.add(g)
            assert len(db) == len(part)
    with pytest.raises(FileNotFoundError):
        BytesSqliteDB(str(root / 'processed' / 'nope'))
    (root / 'id_manager').mkdir()
    (root / 'id_manager' / 'service_id.yml').write_text("'': 0\nsvc: 1\n")
    (root / 'id_manager' / 'operation_id.yml').write_text(
        ''.join(f"'op{i}': {i}\n" for i in range(60)))
    db, idm = open_trace_graph_db(str(root), ['train'])
    assert len(db) == 20 and idm.num_operations == 60
    assert isinstance(db.bytes_db, BytesSqliteDB) and db.bytes_db.get(99) is None
    db.close()
    db, _ = open_trace_graph_db(str(root), ['train', 'val'])
    assert isinstance(db.bytes_db, BytesMultiDB) and len(db) == 30
    want = flatten_trace_graphs(graphs)
    got = db.to_flat()
    for k in ('node_offsets', 'u', 'v', 'node_type', 'avg_latency', 'max_latency', 'min_latency', 'labels'):
        np.testing.assert_array_equal(getattr(got, k), getattr(want, k), err_msg=k)
    np.testing.assert_array_equal(db[25].graph_vectors().node_type, graphs[25].graph_vectors().node_type)
    with pytest.raises(IndexError):
        db.bytes_db.get(30)
    db.close()


def test_load_dataset_reads_reference_databases(tmp_path):
    from tracevae_amd.data_process import convert_db, db_to_flat, load_dataset, load_flat
    from tracevae_amd.synth import make_latency_range
    spec = SyntheticSpec(num_operations=60, max_spans=16, seed=6)
    graphs = generate_trace_graphs(spec, 25)
    root = tmp_path / 'ds'
    with TraceGraphDB(BytesSqliteDB(str(root / 'processed' / 'train'), write=True)) as db:
        for g in graphs:
            db.add(g)
    want = flatten_trace_graphs(graphs)
    got = db_to_flat(str(root / 'processed' / 'train'), chunk=7)      # several chunks
    for k in ('node_offsets', 'u', 'v', 'node_type', 'avg_latency', 'labels'):
        np.testing.assert_array_equal(getattr(got, k), getattr(want, k), err_msg=k)
    idm = root / 'id_manager'
    idm.mkdir()
    (idm / 'service_id.yml').write_text("'': 0\n")
    (idm / 'operation_id.yml').write_text(''.join(f"'op{i}': {i}\n" for i in range(60)))
    mean, std = make_latency_range(spec).tables(60)
    import yaml
    (idm / 'latency_range.yml').write_text(yaml.safe_dump(
        {i: {'mean': float(mean[i]), 'std': float(std[i])} for i in range(60) if np.isfinite(mean[i])}))
    sets, id_manager, _lr = load_dataset(str(root), names=('train',))
    np.testing.assert_array_equal(sets['train'].node_type, want.node_type)
    convert_db(str(root))
    np.testing.assert_array_equal(load_flat(str(root / 'processed' / 'train.npz')).u, want.u)
