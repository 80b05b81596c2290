"""BASELINE.json configs[0] on the accelerated path: the reference's sample data set through
`data_process preprocess` -> `train` (train.sh flags) -> `test evaluate-nll` (test.sh flags).
The data set directory is rebuilt from the committed fixture (the reference's CSV rows and id
tables, tests/golden/sample_dataset_vectors.json), because the reference itself does not
travel to the GPU box."""
import json
import os

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                      'sample_dataset_vectors.json')


def _write_dataset(root):
    import pandas as pd
    with open(GOLDEN, 'r', encoding='utf-8') as f:
        g = json.load(f)
    os.makedirs(os.path.join(root, 'id_manager'))
    for name in ('train', 'val', 'test'):
        pd.DataFrame(g['sets'][name]['rows']).to_csv(os.path.join(root, f'{name}.csv'), index=False)
    op = dict(g['operation_id'])
    if max(op.values()) < g['num_operations'] - 1:
        op['__last__'] = g['num_operations'] - 1     # the id table's size fixes the vocabulary
    tables = {'service_id.yml': g['service_id'], 'operation_id.yml': op,
              'latency_range.yml': {int(k): v for k, v in g['latency_range'].items()}}
    for fname, obj in tables.items():
        with open(os.path.join(root, 'id_manager', fname), 'w', encoding='utf-8') as f:
            yaml.safe_dump(obj, f)
    return g


def test_preprocess_train_test_on_the_sample_dataset(device, tmp_path, capsys):
    from tracevae_amd import data_process, test as test_cli, train as train_cli
    root, out = str(tmp_path / 'data'), str(tmp_path / 'results' / 'train')
    g = _write_dataset(root)
    assert data_process.main(['preprocess', '-i', root, '-o', root]) == 0
    sets, idm, lr = data_process.load_dataset(root)
    assert idm.num_operations == g['num_operations']
    for name in ('train', 'val', 'test', 'test-drop', 'test-latency'):
        assert len(sets[name]) == len(g['sets'][name]['graphs']), name
        assert os.path.exists(os.path.join(root, 'processed', f'{name}.npz'))
    assert set(sets['test-drop'].labels) <= {1} and set(sets['test-latency'].labels) <= {2}

    flags = ['--device=cpu', f'--dataset.root_dir={root}', '--seed=1234', '--model.struct.z_dim=10',
             '--model.struct.decoder.use_prior_flow=true', '--train.z_unit_ball_reg=1',
             '--model.latency.z2_dim=10', '--model.latency.decoder.condition_on_z=true',
             # (short schedule for the test; everything else is train.sh)
             '--train.max_epoch=3', '--train.struct_pretrain_epochs=2', '--train.val_epoch_freq=1',
             '--train.test_epoch_freq=2', f'--output-dir={out}']
    assert train_cli.main(flags) == 0
    for f in ('config.json', 'models/1.pt', 'models/3.pt', 'models/final.pt',
              'result/test-anomaly/2.json', 'result/test-anomaly/final.json'):
        assert os.path.exists(os.path.join(out, f)), f
    with open(os.path.join(out, 'result', 'test-anomaly', 'final.json')) as f:
        final = json.load(f)
    assert {'auc', 'best_fscore', 'nll_normal', 'test_loss'} <= set(final)
    assert np.isfinite(final['nll_normal'])

    capsys.readouterr()
    nll_out = str(tmp_path / 'nll.npz')
    assert test_cli.main(['evaluate-nll', '-M', out, '--use-train-val', '-D', root, '--device', 'cpu',
                          '--use-std-limit', '--std-limit-global', '-o', nll_out]) == 0
    printed = capsys.readouterr().out
    res = json.loads(printed[printed.index('{'):])
    assert {'train_nll_normal', 'val_nll_normal', 'test_auc', 'std_limit_max'} <= set(res)
    saved = np.load(nll_out)
    n_test = sum(len(g['sets'][n]['graphs']) for n in ('test', 'test-drop', 'test-latency'))
    assert saved['nll_list'].shape == (n_test,) and np.isfinite(saved['nll_list']).all()
    assert sorted(set(saved['label_list'].tolist())) == [0, 1, 2]

    # the same command sharded over two ranks (gloo rehearsal of the RCCL job on one GPU) with
    # the noise of every batch seeded by its position in the stream: every trace gets the same
    # score - bit for bit, the same kernels run on the same batches - in the same order
    import subprocess
    import sys
    nll2 = str(tmp_path / 'nll2.npz')
    env = dict(os.environ, TRACEVAE_DIST_BACKEND='gloo', PYTHONPATH=os.pathsep.join(sys.path))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29533', '-m', 'tracevae_amd.test',
           'evaluate-nll', '-M', out, '-D', root, '--batch-size', '3', '--n_z', '4',
           '--noise-seed', '5', '--clip-nll', '1e30', '-o', nll2]
    single = str(tmp_path / 'nll1.npz')
    assert test_cli.main(['evaluate-nll', '-M', out, '-D', root, '--batch-size', '3', '--n_z', '4',
                          '--noise-seed', '5', '--clip-nll', '1e30', '-o', single]) == 0
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(single), np.load(nll2)
    assert np.array_equal(a['label_list'], b['label_list'])
    assert a['nll_list'].shape == (n_test,) and np.array_equal(a['nll_list'], b['nll_list'])
    # (a barely trained model scores above the default clip of 1e5: these runs lift the clip, so
    # the comparison is between real scores, not between clip values)
    assert np.isfinite(a['nll_list']).all() and len(set(a['nll_list'].tolist())) == n_test
    # and another seed gives other scores (the seed is what pins them)
    other = str(tmp_path / 'nll3.npz')
    assert test_cli.main(['evaluate-nll', '-M', out, '-D', root, '--batch-size', '3', '--n_z', '4',
                          '--noise-seed', '6', '--clip-nll', '1e30', '-o', other]) == 0
    assert not np.array_equal(np.load(other)['nll_list'], a['nll_list'])
