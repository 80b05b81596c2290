"""Test-time driver around `do_evaluate_nll`: the `evaluate-nll` command of
`tracegnn/models/trace_vae/test.py:58-266` minus its plumbing (click / mltk experiment,
sqlite data sets, plots).  What it keeps is the part that decides the numbers:

1. a K=1 pass over the train and validation streams that collects, per real node, the
   operation id, the predicted latency std and |latency - mean| / std  (test.py:203-206),
2. `bias_std`  = percentile of the validation-normal relative differences (test.py:208-212),
3. NLL thresholds = median over 10 bootstrap resamples of a percentile of the train /
   validation scores (test.py:214-233),
4. the per-operation std limit = amplified percentile of the validation-normal predicted
   std of that operation, operations never seen take the largest limit (test.py:235-262),
5. the K-sample pass over the test stream with those settings (test.py:266).

The inference functions are NumPy restatements of those lines (host side, a few thousand
values); the scoring passes run on the accelerated path.
"""
import argparse
import json
import sys
from typing import Dict, Iterable, Optional

import numpy as np

from .batch import TraceGraphBatch
from .evaluation import DeviceCollector, do_evaluate_nll
from .model import TraceVAE

__all__ = ['infer_bias_std', 'infer_threshold', 'infer_std_limit', 'evaluate_nll', 'main']


def infer_bias_std(val_normal_reldiff: np.ndarray, bias_std_normal_p: float = 0.995) -> float:
    """test.py:209: np.percentile(latency_reldiff['val_normal'], p * 100)."""
    return float(np.percentile(np.asarray(val_normal_reldiff), bias_std_normal_p * 100))


def infer_threshold(nll_list: np.ndarray, threshold_p: float = 0.995,
                    threshold_amplify: float = 1.0, clip_nll: Optional[float] = 100_000,
                    rng=None) -> float:
    """test.py:216-227: median over 10 bootstrap resamples (with replacement, full size) of
    percentile(scores below the clip value | finite scores, p) * amplify.  `rng` defaults to
    NumPy's global generator, which is what the reference draws from."""
    rng = np.random if rng is None else rng
    nll_list = np.asarray(nll_list)
    cand = []
    for _ in range(10):
        sub = rng.choice(nll_list, replace=True, size=len(nll_list))
        sub = sub[sub < clip_nll - 1e-7] if clip_nll else sub[np.isfinite(sub)]
        cand.append(np.percentile(sub, threshold_p * 100) * threshold_amplify)
    return float(np.median(cand))


def infer_std_limit(operation_id: np.ndarray, latency_std: np.ndarray, num_operations: int,
                    std_limit_p: float = 0.99, std_limit_amplify: float = 1.0,
                    std_limit_global: bool = False,
                    std_limit_fixed: Optional[float] = None) -> np.ndarray:
    """test.py:235-262 -> float32 [num_operations].  Fixed value, one global percentile, or per
    operation amplify * percentile of the validation-normal predicted std of its nodes; an
    operation without nodes takes the largest per-operation limit."""
    limit = np.full([num_operations], np.nan, dtype=np.float32)
    if std_limit_fixed is not None:
        limit[:] = std_limit_fixed
        return limit
    latency_std = np.asarray(latency_std)
    if std_limit_global:
        limit[:] = float(np.percentile(latency_std, std_limit_p * 100))
        return limit
    operation_id = np.asarray(operation_id)
    max_limit = 0
    for op in range(num_operations):
        v = latency_std[operation_id == op]
        if len(v) > 0:
            op_limit = std_limit_amplify * float(np.percentile(v, std_limit_p * 100))
            limit[op] = op_limit
            max_limit = max(max_limit, op_limit)
    limit[np.isnan(limit)] = max_limit
    return limit


def evaluate_nll(vae: TraceVAE, test_stream: Iterable[TraceGraphBatch],
                 train_stream: Optional[Iterable[TraceGraphBatch]] = None,
                 val_stream: Optional[Iterable[TraceGraphBatch]] = None, *,
                 nll_out: Optional[str] = None, n_z: int = 10, clip_nll: Optional[float] = 100_000,
                 no_biased: bool = False, no_latency_biased: bool = False,
                 no_latency: bool = False, use_train_val: bool = False,
                 infer_bias_std_: bool = False, bias_std_normal_p: float = 0.995,
                 infer_threshold_: bool = False, threshold_p: float = 0.995,
                 threshold_amplify: float = 1.0, no_latency_log_prob_weight: bool = False,
                 use_std_limit: bool = False, std_limit_global: bool = False,
                 std_limit_fixed: Optional[float] = None, std_limit_p: float = 0.99,
                 std_limit_amplify: float = 1.0, rng=None, process_group=None,
                 noise_seed: Optional[int] = None) -> Dict[str, float]:
    """`evaluate_nll` of test.py with its option names (the `infer_*` flags carry a trailing
    underscore here because the functions above own the plain names).  Returns the merged
    result dict: `train_*` / `val_*` / `test_*` metrics, `bias_std`, `train_threshold`,
    `val_threshold`, and `std_limit` (the array that was applied, or None)."""
    if infer_bias_std_ or infer_threshold_ or use_std_limit:
        use_train_val = True
    if use_train_val and (train_stream is None or val_stream is None):
        raise ValueError('the train and validation streams are needed for use_train_val / '
                         'infer_bias_std / infer_threshold / use_std_limit')
    num_operations = vae.num_operations
    operation_id, latency_std, latency_reldiff = (DeviceCollector(), DeviceCollector(),
                                                  DeviceCollector())
    results: Dict[str, float] = {}
    nll_result: Dict[str, np.ndarray] = {}

    def run(stream, category, k, threshold=None, std_limit=None, out_file=None):
        for c in (operation_id, latency_std, latency_reldiff):
            c.prefix = f'{category}_'
        r = do_evaluate_nll(
            stream, vae, k, use_biased=(not no_biased) and category == 'test',
            use_latency_biased=not no_latency_biased, no_latency=no_latency,
            latency_log_prob_weight=not no_latency_log_prob_weight, std_limit=std_limit,
            test_threshold=threshold, clip_nll=clip_nll, nll_output_file=out_file,
            operation_id_dict_out=operation_id, latency_std_dict_out=latency_std,
            latency_reldiff_dict_out=latency_reldiff, process_group=process_group,
            noise_seed=noise_seed)
        nll_result[category] = r.pop('_nll_list')
        r.pop('_label_list')
        results.update({f'{category}_{key}': v for key, v in r.items()})

    std_group_limit = None
    thresholds: Dict[str, float] = {}
    if use_train_val:
        run(train_stream, 'train', 1)
        run(val_stream, 'val', 1)
        if infer_bias_std_:
            bias_std = infer_bias_std(latency_reldiff.array('val_normal', process_group),
                                      bias_std_normal_p)
            results['bias_std'] = bias_std
            vae.config.latency.decoder.biased_normal_std_threshold = bias_std
        if infer_threshold_:
            for category in ('train', 'val'):
                thresholds[f'{category}_threshold'] = infer_threshold(
                    nll_result[category], threshold_p, threshold_amplify, clip_nll, rng)
            results.update(thresholds)
        if use_std_limit:
            std_group_limit = infer_std_limit(
                operation_id.array('val_normal', process_group),
                latency_std.array('val_normal', process_group), num_operations,
                std_limit_p, std_limit_amplify, std_limit_global, std_limit_fixed)
    run(test_stream, 'test', n_z, thresholds.get('val_threshold'), std_group_limit, nll_out)
    results['std_limit'] = std_group_limit
    return results


def main(argv=None) -> int:
    """`python -m tracevae_amd.test evaluate-nll -M <model dir|.pt> -D <dataset> [flags]`: the
    options of test.sh / test.py:58-84 (plot outputs are not produced)."""
    import os
    import torch
    from .batch import DeviceTraceStore
    from .data_process import TEST_SETS, concat_flat, load_dataset
    from .train import StoreStream, load_model
    ap = argparse.ArgumentParser(prog='tracevae_amd.test')
    sub = ap.add_subparsers(dest='command', required=True)
    p = sub.add_parser('evaluate-nll')
    p.add_argument('-D', '--data-dir')
    p.add_argument('-M', '--model-path', required=True)
    p.add_argument('-o', '--nll-out')
    p.add_argument('--device')
    p.add_argument('--n_z', type=int, default=10)
    p.add_argument('--batch-size', type=int, default=128)
    p.add_argument('--clip-nll', type=float, default=100_000)
    for flag in ('--no-biased', '--no-latency-biased', '--no-latency', '--use-train-val',
                 '--infer-bias-std', '--infer-threshold', '--no-latency-log-prob-weight',
                 '--use-std-limit', '--std-limit-global'):
        p.add_argument(flag, action='store_true')
    p.add_argument('--bias-std-normal-p', type=float, default=0.995)
    p.add_argument('--threshold-p', type=float, default=0.995)
    p.add_argument('--threshold-amplify', type=float, default=1.0)
    p.add_argument('--std-limit-fixed', type=float, default=None)
    p.add_argument('--std-limit-p', type=float, default=0.99)
    p.add_argument('--std-limit-amplify', type=float, default=1.0)
    p.add_argument('--noise-seed', type=int, default=None,
                   help='(extension) seed the N(0,1) draws per batch of the stream, so that the '
                        'scores do not depend on the number of ranks')
    args, extra = ap.parse_known_args(argv)
    if not torch.cuda.is_available():
        from . import hip
        raise hip.HipLibraryError('the accelerated path needs a GPU (there is no CPU fallback); '
                                  f'--device={args.device} cannot be honoured')
    import torch.distributed as dist
    from .train import init_distributed
    rank, world = init_distributed()     # sharded scoring: rank r takes batches r, r+W, ...
    device = torch.device('cuda', torch.cuda.current_device())
    # the data set root comes from the train config unless -D is given (test.py:108-109)
    cfg_dir = os.path.dirname(os.path.dirname(os.path.abspath(args.model_path))) \
        if args.model_path.endswith('.pt') else args.model_path
    data_dir = args.data_dir
    if data_dir is None:
        with open(os.path.join(cfg_dir, 'config.json'), 'r', encoding='utf-8') as f:
            data_dir = json.load(f)['dataset']['root_dir']
    sets, id_manager, latency_range = load_dataset(data_dir)
    v = id_manager.num_operations
    vae, _config = load_model(args.model_path, v, extra_args=extra, device=device)
    store = lambda flat: DeviceTraceStore(flat, latency_range, v)
    stream = lambda flat, shuffle: StoreStream(store(flat), args.batch_size, shuffle=shuffle,
                                               rank=rank, world_size=world)
    need_tv = args.use_train_val or args.infer_bias_std or args.infer_threshold or args.use_std_limit
    result = evaluate_nll(
        vae, stream(concat_flat(sets[n] for n in TEST_SETS), False),
        stream(sets['train'], True) if need_tv else None,
        stream(sets['val'], True) if need_tv else None,
        nll_out=args.nll_out, n_z=args.n_z, clip_nll=args.clip_nll, no_biased=args.no_biased,
        no_latency_biased=args.no_latency_biased, no_latency=args.no_latency,
        use_train_val=args.use_train_val, infer_bias_std_=args.infer_bias_std,
        bias_std_normal_p=args.bias_std_normal_p, infer_threshold_=args.infer_threshold,
        threshold_p=args.threshold_p, threshold_amplify=args.threshold_amplify,
        no_latency_log_prob_weight=args.no_latency_log_prob_weight,
        use_std_limit=args.use_std_limit, std_limit_global=args.std_limit_global,
        std_limit_fixed=args.std_limit_fixed, std_limit_p=args.std_limit_p,
        std_limit_amplify=args.std_limit_amplify,
        rng=np.random.RandomState(0) if world > 1 else None,   # same bootstrap on every rank
        noise_seed=args.noise_seed)
    limit = result.pop('std_limit')
    if limit is not None:
        result['std_limit_max'] = float(limit.max())
    if rank == 0:
        print(json.dumps(result, indent=1))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
