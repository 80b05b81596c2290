"""End-to-end sanity run on synthetic traces: two-phase training with `fit`, then the test.py
flow (std limit from the validation set) and the AUC / F-score of injected anomalies.
    python tools/train_synthetic.py [--traces 4096] [--epochs 12]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tracevae_amd import constants as C  # noqa: E402
from tracevae_amd.batch import DeviceTraceStore  # noqa: E402
from tracevae_amd.config import ExpConfig, parse_flags  # noqa: E402
from tracevae_amd.synth import SyntheticSpec, generate_flat  # noqa: E402
from tracevae_amd.test import evaluate_nll  # noqa: E402
from tracevae_amd.train import StoreStream, fit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--traces', type=int, default=4096)
    ap.add_argument('--epochs', type=int, default=12)
    ap.add_argument('--max-spans', type=int, default=32)
    ap.add_argument('--num-operations', type=int, default=200)
    ap.add_argument('--width', type=int, default=248, help='per-head GAT width (a multiple of 4)')
    args = ap.parse_args()
    C.set_max_node_count(args.max_spans)
    V = args.num_operations
    spec = SyntheticSpec(num_operations=V, max_spans=args.max_spans, seed=1)
    train, lr = generate_flat(spec, args.traces, seed=2)
    val, _ = generate_flat(spec, args.traces // 8, seed=3)
    test, _ = generate_flat(spec, args.traces // 4, seed=4, anomaly_ratio=0.3)
    cfg = ExpConfig()
    w = args.width
    parse_flags(cfg, ['--model.struct.z_dim=10', '--model.struct.decoder.use_prior_flow=true',
                      '--train.z_unit_ball_reg=1', '--model.latency.z2_dim=10',
                      '--model.latency.decoder.condition_on_z=true', '--seed=1234',
                      f'--train.max_epoch={args.epochs}',
                      f'--train.struct_pretrain_epochs={max(1, args.epochs * 2 // 3)}',
                      f'--train.lr_anneal_epochs={max(1, args.epochs // 2)}',
                      '--train.batch_size=128', '--train.val_epoch_freq=2',
                      f'--train.test_epoch_freq={args.epochs}'])
    for sec in (cfg.model.struct.encoder, cfg.model.struct.decoder, cfg.model.latency.encoder,
                cfg.model.latency.decoder):
        sec.gnn_layers = [w] * len(sec.gnn_layers)
    stores = {k: DeviceTraceStore(v, lr, V) for k, v in (('train', train), ('val', val), ('test', test))}
    out = tempfile.mkdtemp(prefix='tracevae_synth_')
    t0 = time.perf_counter()
    res = fit(cfg, stores, V, out)
    vae = res.pop('_vae')
    print(f'trained in {time.perf_counter() - t0:.1f}s ->', json.dumps(
        {k: round(v, 4) for k, v in res.items() if isinstance(v, float)}))
    stream = lambda st, shuffle: StoreStream(st, 128, shuffle=shuffle)
    r = evaluate_nll(vae, stream(stores['test'], False), stream(stores['train'], True),
                     stream(stores['val'], True), n_z=10, use_std_limit=True, infer_threshold_=True,
                     rng=np.random.RandomState(0))
    keys = ('test_auc', 'test_best_fscore', 'test_best_fscore_drop', 'test_best_fscore_latency',
            'test_fscore', 'test_nll_normal', 'test_nll_drop', 'test_nll_latency', 'val_threshold')
    print(json.dumps({k: round(float(r[k]), 4) for k in keys}))
    frac = float(np.mean(test.labels != 0))
    print(f'anomaly fraction {frac:.3f} (AUC of a random scorer = that)')


if __name__ == '__main__':
    main()
