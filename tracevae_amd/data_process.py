"""`python -m tracevae_amd.data_process preprocess -i <dataset> -o <dataset>`: the role of
`tracegnn/cli/data_process.py:31-106` (csv-to-db x train / val / test) with the flat-array data
set of the accelerated path as its output instead of sqlite + snappy + pickled `TraceGraph`s:

    <out>/processed/{train,val,test,test-drop,test-latency}.npz

Each file holds one `FlatTraces` (concatenated per-trace arrays), i.e. what
`tv_batch_build` consumes, so training never unpickles a graph.  `load_dataset()` reads them
back (and falls back to the CSV files when a set was not preprocessed).

`python -m tracevae_amd.data_process convert-db -i <dataset>` does the same for sets the
reference has already processed (`<dataset>/processed/<name>/_bytes.db`, read by `bytes_db`);
`load_dataset()` also reads those directly when no `.npz` exists.
"""
import argparse
import os
import sys
from typing import Dict, Optional, Tuple

import numpy as np

from .data import (FlatTraces, TraceGraphIDManager, TraceGraphLatencyRangeFile,
                   df_to_trace_graphs, flatten_trace_graphs, load_trace_csv)

__all__ = ['csv_to_flat', 'db_to_flat', 'save_flat', 'load_flat', 'preprocess', 'convert_db',
           'load_dataset', 'concat_flat']

_FIELDS = ('node_offsets', 'u', 'v', 'node_type', 'avg_latency', 'max_latency', 'min_latency',
           'labels')
TEST_SETS = ('test', 'test-drop', 'test-latency')      # labels 0, 1, 2 (data_process.py:66-85)


def save_flat(path: str, flat: FlatTraces) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, **{k: getattr(flat, k) for k in _FIELDS})


def load_flat(path: str) -> FlatTraces:
    with np.load(path, allow_pickle=False) as z:
        return FlatTraces(**{k: z[k] for k in _FIELDS})


def concat_flat(parts) -> FlatTraces:
    """Concatenate data sets (the reference's BytesMultiDB over test / test-drop / test-latency)."""
    parts = list(parts)
    off = [np.zeros(1, np.int64)]
    base = 0
    for p in parts:
        off.append(p.node_offsets[1:] + base)
        base += int(p.node_offsets[-1])
    cat = lambda name: np.concatenate([getattr(p, name) for p in parts])
    return FlatTraces(node_offsets=np.concatenate(off), u=cat('u'), v=cat('v'),
                      node_type=cat('node_type'), avg_latency=cat('avg_latency'),
                      max_latency=cat('max_latency'), min_latency=cat('min_latency'),
                      labels=cat('labels'))


def csv_to_flat(input_dir: str, name: str, id_manager: TraceGraphIDManager) -> Dict[str, FlatTraces]:
    """One CSV -> its data set(s): `train` / `val` give one, `test` gives the three test sets
    (normal, structure anomalies, latency anomalies), like csv_to_db."""
    path = os.path.join(input_dir, f'{name}.csv')
    out: Dict[str, FlatTraces] = {}
    if 'test' not in name:
        graphs = df_to_trace_graphs(load_trace_csv(path), id_manager, merge_spans=True)
        flat = flatten_trace_graphs(graphs)
        flat.labels = np.zeros(len(flat), np.int64)
        out[name] = flat
    else:
        df = load_trace_csv(path, is_test=True)
        for label, set_name in enumerate(TEST_SETS):
            graphs = df_to_trace_graphs(df, id_manager, merge_spans=True, test_label=label)
            flat = flatten_trace_graphs(graphs)
            flat.labels = np.full(len(flat), label, np.int64)
            out[set_name] = flat
    return out


def db_to_flat(db_dir: str, chunk: int = 8192) -> FlatTraces:
    """`<dataset>/processed/<name>/` (a reference `_bytes.db`) -> flat arrays, streaming: at
    most `chunk` graph objects are alive at a time."""
    from .bytes_db import BytesSqliteDB, TraceGraphDB
    parts, pending = [], []
    with TraceGraphDB(BytesSqliteDB(db_dir)) as db:
        for g in db:
            pending.append(g)
            if len(pending) >= chunk:
                parts.append(flatten_trace_graphs(pending))
                pending = []
    if pending or not parts:
        parts.append(flatten_trace_graphs(pending))
    return parts[0] if len(parts) == 1 else concat_flat(parts)


def convert_db(input_dir: str, output_dir: Optional[str] = None) -> None:
    output_dir = output_dir or input_dir
    processed = os.path.join(input_dir, 'processed')
    for name in sorted(os.listdir(processed)):
        if os.path.isfile(os.path.join(processed, name, '_bytes.db')):
            flat = db_to_flat(os.path.join(processed, name))
            dst = os.path.join(output_dir, 'processed', f'{name}.npz')
            save_flat(dst, flat)
            print(f'{name}: {len(flat)} traces, {int(flat.node_offsets[-1])} nodes -> {dst}')


def preprocess(input_dir: str, output_dir: str) -> None:
    id_manager = TraceGraphIDManager(os.path.join(input_dir, 'id_manager'))
    for name in ('train', 'val', 'test'):
        for set_name, flat in csv_to_flat(input_dir, name, id_manager).items():
            dst = os.path.join(output_dir, 'processed', f'{set_name}.npz')
            save_flat(dst, flat)
            print(f'{set_name}: {len(flat)} traces, {int(flat.node_offsets[-1])} nodes -> {dst}')


def load_dataset(root_dir: str, names=('train', 'val') + TEST_SETS
                 ) -> Tuple[Dict[str, FlatTraces], TraceGraphIDManager, TraceGraphLatencyRangeFile]:
    """`<root>/id_manager` + the data sets, from `<root>/processed/<name>.npz` when present, else
    from the reference's `<root>/processed/<name>/_bytes.db`, else straight from `<root>/*.csv`."""
    id_manager = TraceGraphIDManager(os.path.join(root_dir, 'id_manager'))
    latency_range = TraceGraphLatencyRangeFile(os.path.join(root_dir, 'id_manager'),
                                               require_exists=True)
    sets: Dict[str, FlatTraces] = {}
    csv_cache: Dict[str, Dict[str, FlatTraces]] = {}
    for name in names:
        path = os.path.join(root_dir, 'processed', f'{name}.npz')
        if os.path.exists(path):
            sets[name] = load_flat(path)
            continue
        db_dir = os.path.join(root_dir, 'processed', name)
        if os.path.isfile(os.path.join(db_dir, '_bytes.db')):
            sets[name] = db_to_flat(db_dir)
            continue
        csv_name = 'test' if name in TEST_SETS else name
        if csv_name not in csv_cache:
            csv_cache[csv_name] = csv_to_flat(root_dir, csv_name, id_manager)
        sets[name] = csv_cache[csv_name][name]
    return sets, id_manager, latency_range


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog='tracevae_amd.data_process')
    sub = ap.add_subparsers(dest='command', required=True)
    p = sub.add_parser('preprocess')
    p.add_argument('-i', '--input-dir', required=True)
    p.add_argument('-o', '--output-dir', '--output_dir', dest='output_dir', required=True)
    p = sub.add_parser('convert-db')
    p.add_argument('-i', '--input-dir', required=True)
    p.add_argument('-o', '--output-dir', '--output_dir', dest='output_dir', default=None)
    args = ap.parse_args(argv)
    if args.command == 'convert-db':
        convert_db(os.path.abspath(args.input_dir),
                   os.path.abspath(args.output_dir) if args.output_dir else None)
        return 0
    preprocess(os.path.abspath(args.input_dir), os.path.abspath(args.output_dir))
    return 0


if __name__ == '__main__':
    sys.exit(main())
