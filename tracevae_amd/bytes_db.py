"""Reader for data sets the reference has already processed: `<root>/processed/<name>/_bytes.db`.

The reference stores each processed set as a SQLite table `data(key INT PRIMARY KEY, value BLOB)`
whose values are snappy-compressed pickles of `TraceGraph` objects
(`tracegnn/data/bytes_db.py:68-201`, `tracegnn/data/trace_graph.py:376-398`,
`tracegnn/data/trace_graph_db.py:16-108`).  Neither `python-snappy` nor the reference package
is importable where this library runs, so this module carries

* a decoder for the raw snappy block format (the published format description: a varint
  length, then literal / copy elements) and an encoder that emits literals and
  back-references found with a small hash table - enough to write files the reference can
  read back;
* an unpickler that resolves the reference's class paths (`tracegnn.data.trace_graph.*`) to
  this package's types and refuses every other global, so opening a data file can construct
  trace graphs and nothing else;
* `BytesSqliteDB`, `BytesMultiDB`, `TraceGraphDB`, `open_trace_graph_db` with the reference's
  names and meaning.

Parity unpinned: the reference ships no `_bytes.db` and no snappy vectors; the tests check the
decoder against hand-assembled streams of every element kind and the rest by round trips.
Host-side input conversion only - nothing here is on the GPU path; `to_flat()` turns a
database into the `FlatTraces` arrays the batching kernel reads.
"""
import io
import os
import pickle
import sqlite3
from contextlib import contextmanager
from typing import Iterator, List, Optional, Sequence, Tuple

from . import data as _data
from .data import FlatTraces, TraceGraph, TraceGraphIDManager, flatten_trace_graphs

__all__ = [
    'snappy_decompress', 'snappy_compress', 'loads_trace_graph', 'dumps_trace_graph',
    'BytesSqliteDB', 'BytesMultiDB', 'TraceGraphDB', 'open_trace_graph_db',
]


# ---------------------------------------------------------------------------------------
# snappy, raw block format
# ---------------------------------------------------------------------------------------
def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = value = 0
    while True:
        if pos >= len(buf) or shift > 35:
            raise ValueError('snappy: bad length header')
        b = buf[pos]
        pos += 1
        value |= (b & 0x7f) << shift
        if not b & 0x80:
            return value, pos
        shift += 7


def snappy_decompress(buf: bytes) -> bytes:
    """Raw snappy block -> bytes (what `snappy.decompress` does in `bytes_db.py:157`)."""
    buf = bytes(buf)
    total, pos = _read_varint(buf, 0)
    out = bytearray()
    n = len(buf)
    while pos < n:
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            length = tag >> 2
            if length >= 60:
                extra = length - 59
                if pos + extra > n:
                    raise ValueError('snappy: truncated literal length')
                length = int.from_bytes(buf[pos:pos + extra], 'little')
                pos += extra
            length += 1
            if pos + length > n:
                raise ValueError('snappy: truncated literal')
            out += buf[pos:pos + length]
            pos += length
            continue
        if kind == 1:                                   # copy, 11-bit offset, length 4..11
            if pos >= n:
                raise ValueError('snappy: truncated copy')
            length = ((tag >> 2) & 7) + 4
            offset = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        else:                                           # copy, 16- or 32-bit offset, length 1..64
            width = 2 if kind == 2 else 4
            if pos + width > n:
                raise ValueError('snappy: truncated copy')
            length = (tag >> 2) + 1
            offset = int.from_bytes(buf[pos:pos + width], 'little')
            pos += width
        if offset == 0 or offset > len(out):
            raise ValueError('snappy: copy offset out of range')
        start = len(out) - offset
        if offset >= length:
            out += out[start:start + length]
        else:                                           # overlapping copy repeats the pattern
            pattern = bytes(out[start:])
            reps, rem = divmod(length, offset)
            out += pattern * reps + pattern[:rem]
    if len(out) != total:
        raise ValueError(f'snappy: length mismatch ({len(out)} != {total})')
    return bytes(out)


def _emit_literal(out: bytearray, chunk: bytes) -> None:
    n = len(chunk) - 1
    if n < 60:
        out.append(n << 2)
    else:
        width = (n.bit_length() + 7) // 8
        out.append((59 + width) << 2)
        out += n.to_bytes(width, 'little')
    out += chunk


def snappy_compress(data: bytes) -> bytes:
    """bytes -> a valid raw snappy block (greedy 4-byte hash matching, 64 KiB window)."""
    data = bytes(data)
    n = len(data)
    out = bytearray()
    v = n
    while True:                                         # varint length
        if v < 0x80:
            out.append(v)
            break
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    table = {}
    pos = lit = 0
    while pos + 4 <= n:
        key = data[pos:pos + 4]
        cand = table.get(key)
        table[key] = pos
        if cand is None or pos - cand > 0xffff:
            pos += 1
            continue
        length = 4
        while pos + length < n and data[cand + length] == data[pos + length]:
            length += 1
        if lit < pos:
            _emit_literal(out, data[lit:pos])
        offset = pos - cand
        pos += length
        while length > 0:                               # 2-byte-offset copies, <= 64 bytes each
            step = min(length, 64)
            out.append(((step - 1) << 2) | 2)
            out += offset.to_bytes(2, 'little')
            length -= step
        lit = pos
    if lit < n:
        _emit_literal(out, data[lit:])
    return bytes(out)


# ---------------------------------------------------------------------------------------
# pickles of TraceGraph objects
# ---------------------------------------------------------------------------------------
_REFERENCE_MODULE = 'tracegnn.data.trace_graph'
_GRAPH_CLASSES = ('TraceGraph', 'TraceGraphNode', 'TraceGraphNodeFeatures', 'TraceGraphSpan',
                  'TraceGraphNodeReconsScores', 'TraceGraphVectors')
_SAFE_GLOBALS = {
    ('builtins', n) for n in ('object', 'set', 'frozenset', 'list', 'dict', 'tuple', 'int',
                              'float', 'bool', 'str', 'bytes', 'bytearray', 'complex')
} | {
    ('copyreg', '_reconstructor'), ('copy_reg', '_reconstructor'),
    ('collections', 'OrderedDict'),
    ('_codecs', 'encode'),                              # protocol <= 2 spelling of bytes
    ('datetime', 'datetime'), ('datetime', 'date'), ('datetime', 'timedelta'),
    ('datetime', 'timezone'),
    ('numpy', 'dtype'), ('numpy', 'ndarray'),
    ('numpy.core.multiarray', 'scalar'), ('numpy.core.multiarray', '_reconstruct'),
    ('numpy._core.multiarray', 'scalar'), ('numpy._core.multiarray', '_reconstruct'),
}


class _TraceGraphUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module in (_REFERENCE_MODULE, _data.__name__) and name in _GRAPH_CLASSES:
            return getattr(_data, name)
        if (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'refusing global {module}.{name} in a trace-graph pickle')


def loads_trace_graph(content: bytes) -> TraceGraph:
    """`TraceGraph.from_bytes` (trace_graph.py:379-398), with the version upgrades."""
    g = _TraceGraphUnpickler(io.BytesIO(content)).load()
    if not isinstance(g, TraceGraph):
        raise pickle.UnpicklingError(f'not a TraceGraph: {type(g).__name__}')
    version = getattr(g, 'version', 0)
    if version < 0x2:
        for _, node in g.root.iter_bfs():
            if version < 0x1:
                node.scores = None
            node.anomaly = None
        g.version = 0x2
    if not hasattr(g, 'data'):
        g.data = {}
    return g


def dumps_trace_graph(g: TraceGraph, protocol: int = pickle.DEFAULT_PROTOCOL) -> bytes:
    """`TraceGraph.to_bytes` (trace_graph.py:376-377): a pickle naming the reference's classes."""
    return _dumps_as_reference(g, protocol)


def _dumps_as_reference(g, protocol: int) -> bytes:
    # pickle.Pickler looks a class up by module + qualname when it writes a global; present
    # this package's classes under the reference's path for the duration of the dump
    import sys
    import types
    saved = {}
    mod_names = ['tracegnn', 'tracegnn.data', _REFERENCE_MODULE]
    for name in mod_names:
        saved[name] = sys.modules.get(name)
    classes = [getattr(_data, n) for n in _GRAPH_CLASSES]
    try:
        shim = None
        for name in mod_names:
            shim = types.ModuleType(name)
            sys.modules[name] = shim
        for cls in classes:
            setattr(shim, cls.__name__, cls)
            cls.__module__ = _REFERENCE_MODULE
        buf = io.BytesIO()
        pickle.Pickler(buf, protocol=protocol).dump(g)
        return buf.getvalue()
    finally:
        for cls in classes:
            cls.__module__ = _data.__name__
        for name in mod_names:
            if saved[name] is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = saved[name]


# ---------------------------------------------------------------------------------------
# databases
# ---------------------------------------------------------------------------------------
class BytesSqliteDB(object):
    """`bytes_db.py:68-201`: table `data(key, value)`; keys are 0..count-1."""

    def __init__(self, path: str, write: bool = False, table_name: str = 'data',
                 file_name: str = '_bytes.db'):
        self.path = path
        self.table_name = table_name
        self.file_name = file_name
        db_file = os.path.join(path, file_name)
        if write:
            os.makedirs(path, exist_ok=True)
        elif not os.path.isfile(db_file):
            raise FileNotFoundError(db_file)
        self.conn = sqlite3.connect(db_file)
        self.conn.text_factory = bytes
        cur = self.conn.cursor()
        try:
            if write:
                cur.execute(f'CREATE TABLE IF NOT EXISTS "{table_name}" ('
                            '  "key" INT PRIMARY KEY,  "value" BLOB);')
                self.conn.commit()
            self._data_count = cur.execute(f'SELECT COUNT(*) FROM "{table_name}"').fetchone()[0]
        finally:
            cur.close()
        self._pending: Optional[List[Tuple[int, bytes]]] = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self) -> int:
        return self._data_count

    def data_count(self) -> int:
        return self._data_count

    def describe(self) -> str:
        p = self.path if self.file_name == '_bytes.db' else os.path.join(self.path, self.file_name)
        return repr(p) if any(c in p for c in '(),') else p

    def get(self, item: int) -> Optional[bytes]:
        row = self.conn.execute(f'SELECT "value" FROM "{self.table_name}" WHERE "key" = ?',
                                (int(item),)).fetchone()
        return None if row is None else snappy_decompress(row[0])

    def __getitem__(self, item: int) -> bytes:
        value = self.get(item)
        if value is None:
            raise IndexError(item)
        return value

    def __iter__(self) -> Iterator[bytes]:
        """All values in key order, through one cursor (the whole-set read `to_flat` does)."""
        for (value,) in self.conn.execute(
                f'SELECT "value" FROM "{self.table_name}" ORDER BY "key"'):
            yield snappy_decompress(value)

    def add(self, val: bytes) -> int:
        if self._pending is None:
            with self.write_batch():
                return self.add(val)
        key = self._data_count
        self._pending.append((key, snappy_compress(val)))
        self._data_count += 1
        if len(self._pending) >= 1024:
            self._flush()
        return key

    def _flush(self) -> None:
        if self._pending:
            self.conn.executemany(
                f'INSERT INTO "{self.table_name}"("key", "value") VALUES (?, ?)', self._pending)
            self.conn.commit()
            self._pending.clear()

    @contextmanager
    def write_batch(self):
        if self._pending is not None:
            raise RuntimeError('Another write_batch is already open!')
        self._pending = []
        try:
            yield self
            self._flush()
        except BaseException:
            self.conn.rollback()
            raise
        finally:
            self._pending = None

    def commit(self) -> None:
        self.conn.commit()

    def close(self) -> None:
        self.conn.commit()
        self.conn.close()


class BytesMultiDB(object):
    """Several databases read as one, in order (`bytes_db.py:204-279`); read-only."""

    def __init__(self, *db_list):
        self.db_list = list(db_list)
        self._sizes = [db.data_count() for db in self.db_list]

    def __len__(self) -> int:
        return sum(self._sizes)

    def data_count(self) -> int:
        return len(self)

    def describe(self) -> str:
        return ','.join(db.describe() for db in self.db_list)

    def get(self, item: int) -> Optional[bytes]:
        if item < 0:
            item += len(self)
        for db, size in zip(self.db_list, self._sizes):
            if item < size:
                return db.get(item)
            item -= size
        raise IndexError('index out of range')

    def __getitem__(self, item: int) -> bytes:
        return self.get(item)

    def __iter__(self) -> Iterator[bytes]:
        for db in self.db_list:
            yield from db

    def close(self) -> None:
        for db in self.db_list:
            db.close()


class TraceGraphDB(object):
    """`trace_graph_db.py:16-83`: a bytes database whose values are trace graphs."""

    def __init__(self, bytes_db, protocol: Optional[int] = None):
        self.bytes_db = bytes_db
        self.protocol = pickle.DEFAULT_PROTOCOL if protocol is None else protocol

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self) -> int:
        return self.bytes_db.data_count()

    def data_count(self) -> int:
        return self.bytes_db.data_count()

    def __repr__(self):
        return f'TraceGraphDB({self.bytes_db.describe()})'

    def get(self, item: int) -> TraceGraph:
        return loads_trace_graph(self.bytes_db.get(item))

    def __getitem__(self, item: int) -> TraceGraph:
        return self.get(item)

    def __iter__(self) -> Iterator[TraceGraph]:
        for content in self.bytes_db:
            yield loads_trace_graph(content)

    def add(self, g: TraceGraph) -> int:
        return self.bytes_db.add(dumps_trace_graph(g, self.protocol))

    def write_batch(self):
        return self.bytes_db.write_batch()

    def to_flat(self) -> FlatTraces:
        """Every trace of the database as the flat arrays the batching kernel reads."""
        return flatten_trace_graphs(list(self))

    def close(self) -> None:
        self.bytes_db.close()


def open_trace_graph_db(input_dir: str, names: Sequence[str] = (), protocol: Optional[int] = None
                        ) -> Tuple[TraceGraphDB, TraceGraphIDManager]:
    """`trace_graph_db.py:86-108`: `<input_dir>/processed/<name>/_bytes.db` for each name."""
    file_name = f'_bytes_{protocol}.db' if protocol else '_bytes.db'
    id_manager = TraceGraphIDManager(os.path.join(input_dir, 'id_manager'))
    dbs = [BytesSqliteDB(os.path.join(input_dir, 'processed', name), file_name=file_name)
           for name in names]
    bytes_db = dbs[0] if len(dbs) == 1 else BytesMultiDB(*dbs)
    return TraceGraphDB(bytes_db, protocol=protocol), id_manager
