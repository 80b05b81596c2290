"""Generates tests/golden/sample_dataset_vectors.json by running the REFERENCE's own
`load_trace_csv` + `df_to_trace_graphs(merge_spans=True)` + `TraceGraph.graph_vectors()`
(tracegnn/data/trace_graph.py) on the reference's `sample_dataset/`.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_vectors.py

The reference package cannot be imported as a whole (mltk / dgl / tensorkit / snappy are
not installable offline), so the pure-Python files are loaded by path into stub packages.
The sample data set is copied to a scratch directory first: the reference's id manager
rewrites its yml files on exit.  The fixture stores DATA only: the input spans (the CSV
rows) and the arrays the reference produced for them.
"""
import importlib.util
import json
import os
import shutil
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sample_dataset_vectors.json')


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    for pkg in ('tracegnn', 'tracegnn.utils', 'tracegnn.data'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    utils = sys.modules['tracegnn.utils']
    names = []
    for fname in ('id_assign', 'latency_range_file', 'array_buffer'):
        mod = _load(f'tracegnn.utils.{fname}', os.path.join(REF, 'tracegnn', 'utils', fname + '.py'))
        for k in getattr(mod, '__all__', []):
            setattr(utils, k, getattr(mod, k))
            names.append(k)
    utils.__all__ = names
    tg = _load('tracegnn.data.trace_graph', os.path.join(REF, 'tracegnn', 'data', 'trace_graph.py'))

    scratch = tempfile.mkdtemp(prefix='tracevae_golden_')
    try:
        ds = os.path.join(scratch, 'sample_dataset')
        shutil.copytree(os.path.join(REF, 'sample_dataset'), ds)
        id_manager = tg.TraceGraphIDManager(os.path.join(ds, 'id_manager'))
        lr = utils.TraceGraphLatencyRangeFile(os.path.join(ds, 'id_manager'))
        out = {'num_operations': id_manager.num_operations, 'sets': {}, 'latency_range': {}}
        used_ops = set()
        jobs = [('train', 'train.csv', False, None), ('val', 'val.csv', False, None),
                ('test', 'test.csv', True, 0), ('test-drop', 'test.csv', True, 1),
                ('test-latency', 'test.csv', True, 2)]
        for set_name, csv_name, is_test, label in jobs:
            df = tg.load_trace_csv(os.path.join(ds, csv_name), is_test=is_test)
            graphs = tg.df_to_trace_graphs(df, id_manager, test_label=label, merge_spans=True)
            rows = json.loads(df.to_json(orient='records'))
            entry = {'csv': csv_name, 'is_test': is_test, 'test_label': label, 'rows': rows,
                     'graphs': []}
            for g in graphs:
                gv = g.graph_vectors()
                used_ops.update(int(t) for t in gv.node_type)
                entry['graphs'].append({
                    'trace_id': [int(g.trace_id[0]), int(g.trace_id[1])],
                    'node_count': int(g.node_count), 'max_depth': int(g.max_depth),
                    'data': g.data,
                    'u': gv.u.tolist(), 'v': gv.v.tolist(), 'node_type': gv.node_type.tolist(),
                    'node_depth': gv.node_depth.tolist(), 'node_idx': gv.node_idx.tolist(),
                    'span_count': gv.span_count.tolist(),
                    'avg_latency': [float(x) for x in gv.avg_latency],
                    'max_latency': [float(x) for x in gv.max_latency],
                    'min_latency': [float(x) for x in gv.min_latency],
                })
            out['sets'][set_name] = entry
        for op in sorted(used_ops):
            mean, std = lr[op]
            out['latency_range'][str(op)] = {'mean': float(mean), 'std': float(std)}
        # the id maps restricted to what the sample spans use
        svc, ops = {}, {}
        for entry in out['sets'].values():
            for r in entry['rows']:
                s, o = str(r['serviceName']), f"{r['serviceName']}/{r['operationName']}"
                if s in id_manager.service_id._mapping:
                    svc[s] = id_manager.service_id._mapping[s]
                if o in id_manager.operation_id._mapping:
                    ops[o] = id_manager.operation_id._mapping[o]
        out['service_id'], out['operation_id'] = svc, ops
        with open(OUT, 'w', encoding='utf-8') as f:
            json.dump(out, f)
        print('wrote', OUT, {k: len(v['graphs']) for k, v in out['sets'].items()})
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
