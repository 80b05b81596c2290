"""CPU suite, part 2: host logic without a GPU - the C-ABI library loads and exports every
symbol of include/tracevae_hip.h, argument validation, config / flag parsing, state_dict
layout, the flat parameter bucket, and the data-parallel gradient exchange over gloo with
two ranks."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from helpers import make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tracevae_amd import hip
    text = open(hip.header_path()).read()
    declared = set(re.findall(r'\b(tv_\w+)\s*\(', re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)))
    assert len(declared) >= 40
    assert declared == set(hip.SIGNATURES)
    lib = hip.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.tv_version() == 1
    assert hip.query('tv_stat_blocks', 1) == 1 and hip.query('tv_stat_blocks', 10 ** 9) == 512
    # a small product needs no workspace; a small B under >= 2048 rows is split once per call
    # into the LDS images of its tiles (24 KB per 128 x 32 tile); a partial last round of
    # workgroups is split over K into 128 x 128 slabs in front of them; a weight-gradient shape
    # gets split-K slabs
    image = 3 * 128 * 64
    assert hip.query('tv_gemm_f32_workspace_bytes', 100, 100, 100) == 0
    assert hip.query('tv_gemm_f32_workspace_bytes', 4096, 1024, 1000) == 8 * 32 * image
    # the wide forms add 2 x 256 partial operand maxima (2 KB) behind the planes
    full = hip.query('tv_gemm_f32_workspace_bytes', 51200, 1000, 1000)
    assert full > 8 * 32 * image and (full - 2048 - 8 * 32 * image) % (128 * 128 * 4) == 0
    assert hip.query('tv_gemm_f32_workspace_bytes', 1000, 1000, 51200) >= 2 * 1000 * 1000 * 4


def test_host_side_validation_without_gpu():
    """Bad shapes are rejected before any launch (so these calls are safe on a CPU box)."""
    from tracevae_amd import hip
    lib = hip.lib()
    one = 4096   # never dereferenced: validation fails first
    assert lib.tv_gemm_f32(0, 1, 4, 4, 4, one, 2, one, 4, one, 4, None, None, 0, None, 0, None) == -1
    assert b'lda' in lib.tv_last_error()
    assert lib.tv_gat_scores(10, 9, 8, one, one, one, one, one, None) == -1
    assert lib.tv_gat_scores(10, 2, 6, one, one, one, one, one, None) == -1
    assert lib.tv_adj_bernoulli_ll_fwd(1, 2, 8, 500, one, one, 0, 1.0, 0.0, one, one, 64, None) == -1
    assert lib.tv_adj_bernoulli_ll_fwd(1, 2, 8, 40, one, one, 0, 1.0, 0.0, one, one, 4, None) == -3   # workspace too small
    assert lib.tv_adj_bernoulli_ll_fwd_static(1, 2, 2000, 40, one, one, 0, 1.0, 0.0, one, None) == -1
    assert lib.tv_rowdot_fwd(5, 10, 9, one, one, None, one, None) == -1
    assert lib.tv_optim_step(3, 10, one, one, one, None, None, 0.0, 1e-3, 0.0, 0.99, 1e-8, 1, None) == -1


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from tracevae_amd import hip, ops
    with pytest.raises(hip.HipLibraryError):
        ops.gemm(torch.zeros(2, 2), torch.zeros(2, 2), False, False)
    from tracevae_amd.batch import DeviceTraceStore
    from helpers import make_flat
    flat, lr = make_flat(2, 10, 4)
    with pytest.raises(hip.HipLibraryError):
        DeviceTraceStore(flat, lr, 10)


def test_config_flags_and_json_round_trip():
    from tracevae_amd.config import ExpConfig, config_from_dict, config_to_dict, parse_flags
    cfg = make_config()
    assert cfg.model.struct.z_dim == 10 and cfg.model.struct.decoder.use_prior_flow is True
    assert cfg.train.z_unit_ball_reg == 1 and cfg.model.latency.decoder.condition_on_z is True
    assert cfg.train.batch_size == 64 and cfg.train.optimizer == 'rmsprop'
    assert cfg.model.struct.encoder.gnn_layers == [500, 500, 500, 500]
    rest = parse_flags(cfg, ['--device=cpu', '--dataset.root_dir=/x', '--seed=1234',
                             '--train.warm_up_epochs=none', '--model.struct.encoder.gnn_layers=[8,8]'])
    assert rest == [] and cfg.seed == 1234 and cfg.dataset.root_dir == '/x'
    assert cfg.model.struct.encoder.gnn_layers == [8, 8]
    with pytest.raises(AttributeError):
        parse_flags(cfg, ['--model.no_such_field=1'])
    blob = json.loads(json.dumps(config_to_dict(cfg)))
    again = config_from_dict(ExpConfig(), blob)
    assert config_to_dict(again) == config_to_dict(cfg)
    # instances do not share mutable defaults
    other = ExpConfig()
    assert other.model.struct.encoder.gnn_layers == [500, 500, 500, 500]


def test_state_dict_layout_matches_appendix_c():
    from tracevae_amd.model import TraceVAE
    cfg = make_config(width=8, layers=4)
    vae = TraceVAE(cfg.model, 50, max_node_count=8)
    sd = vae.state_dict()
    emb = [k for k in sd if k.endswith('node_embedding.weight')]
    assert sorted(emb) == ['latency_vae.operation_embedding.node_embedding.weight',
                           'operation_embedding.node_embedding.weight',
                           'struct_vae.operation_embedding.node_embedding.weight']
    assert sd[emb[0]].data_ptr() == sd[emb[1]].data_ptr() == sd[emb[2]].data_ptr()
    for stack in ('struct_vae.qz_gnn_layers', 'struct_vae.pG_node_type_logits',
                  'latency_vae.qz2_gnn_layers', 'latency_vae.pG_node_features'):
        for i in range(4):
            assert sd[f'{stack}.gnn.{i}.gnn.fc.weight'].shape[0] == 16
            assert sd[f'{stack}.gnn.{i}.gnn.attn_l'].shape == (1, 2, 8)
            assert sd[f'{stack}.gnn.{i}.gnn.bias'].shape == (16,)
            for leaf in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
                assert f'{stack}.gnn.{i}.normalization.{leaf}' in sd
    assert sd['struct_vae.pG_node_type_logits.gnn.4.gnn.weight'].shape == (16, 50)
    assert sd['latency_vae.pG_node_features.gnn.4.gnn.weight'].shape == (16, 2)
    assert sd['struct_vae.qz_mean.weight_store.value'].shape == (10, 16)
    assert sd['struct_vae.pG_node_embedding.weight_store.value'].shape == (8 * 40, 500)
    assert sd['struct_vae.pG_node_count_logits.layer.1.bias_store.value'].shape == (9,)
    assert 'struct_vae.pz_flow.original_flow.flows.2.kernel' in sd
    # named_parameters de-duplicates the shared embedding; L2 skips names containing 'bias'
    names = [n for n, _ in vae.named_parameters()]
    assert names.count('operation_embedding.node_embedding.weight') == 1
    assert not any('struct_vae.operation_embedding' in n for n in names)
    # strict load of its own state dict
    vae2 = TraceVAE(cfg.model, 50, max_node_count=8)
    vae2.load_state_dict(sd, strict=True)


def test_flat_parameters_and_phase2_quirk():
    from tracevae_amd.model import TraceVAE
    from tracevae_amd.train import FlatParameters, l2_parameter_names, learning_rate_for_epoch
    cfg = make_config(width=8, layers=2)
    vae = TraceVAE(cfg.model, 30, max_node_count=8)
    before = {n: p.detach().clone() for n, p in vae.named_parameters()}
    flat = FlatParameters(vae)
    for n, p in vae.named_parameters():
        assert torch.equal(p.detach(), before[n])
        a, b = flat.slices[n]
        assert p.data_ptr() == flat.flat.data_ptr() + 4 * a and a % 64 == 0
        assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * a
        assert (a >= flat.tail_start) == n.startswith('latency_vae.')
    # writing through the flat buffer is visible in the module (and in state_dict)
    flat.flat.fill_(0.5)
    assert float(vae.state_dict()['struct_vae.qz_mean.weight_store.value'].mean()) == 0.5
    # the reference zips trained parameters with the full name list (train.py:253-255)
    all_names = flat.all_names
    part = [n for n in all_names if n.startswith('latency_vae.')]
    l2 = l2_parameter_names(all_names, part)
    assert l2 == [p for p, n in zip(part, all_names) if 'bias' not in n]
    assert l2_parameter_names(all_names, all_names) == [n for n in all_names if 'bias' not in n]
    assert learning_rate_for_epoch(cfg.train, 1) == 1e-3
    assert abs(learning_rate_for_epoch(cfg.train, 30) - 1e-4) < 1e-12
    assert abs(learning_rate_for_epoch(cfg.train, 60) - 1e-5) < 1e-12


def _dp_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from tracevae_amd.model import TraceVAE
    from tracevae_amd.train import Trainer
    torch.manual_seed(0)                      # identical initial weights on every rank
    cfg = make_config(width=8, layers=2)
    vae = TraceVAE(cfg.model, 30, max_node_count=8)
    trainer = Trainer(vae, cfg, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(trainer.flat.total, generator=g)
    trainer.flat.grad.copy_(local)
    trainer.all_reduce_gradients()
    torch.save({'local': local, 'reduced': trainer.flat.grad.clone()}, os.path.join(tmp, f'r{rank}.pt'))
    # phase 2 only exchanges the latency_vae suffix of the bucket
    trainer.set_phase(latency_only=True)
    trainer.flat.grad.copy_(local)
    trainer.all_reduce_gradients()
    torch.save({'reduced': trainer.flat.grad.clone(), 'tail': trainer.flat.tail_start},
               os.path.join(tmp, f'p2_r{rank}.pt'))
    dist.destroy_process_group()


def test_dp_gradient_bucket_allreduce_gloo(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, 29000 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f'r{i}.pt') for i in range(world)]
    mean = (r[0]['local'] + r[1]['local']) / 2
    for i in range(world):
        assert torch.allclose(r[i]['reduced'], mean, atol=1e-7)
    assert torch.equal(r[0]['reduced'], r[1]['reduced'])
    p2 = [torch.load(tmp_path / f'p2_r{i}.pt') for i in range(world)]
    tail = p2[0]['tail']
    for i in range(world):
        assert torch.equal(p2[i]['reduced'][:tail], r[i]['local'][:tail])      # untouched
        assert torch.allclose(p2[i]['reduced'][tail:], mean[tail:], atol=1e-7)


def test_store_stream_shards_batches_across_ranks():
    from tracevae_amd.train import StoreStream

    class FakeStore(object):
        labels = None

        def __len__(self):
            return 103

    seen = []
    for rank in range(4):
        s = StoreStream(FakeStore(), 10, shuffle=True, seed=7, rank=rank, world_size=4)
        ids = [G._trace_ids for G in s]
        assert len(ids) == s.batch_count
        seen += ids
    allids = np.concatenate(seen)
    assert sorted(allids.tolist()) == list(range(103))        # a partition of the data set


def test_analyze_anomaly_nll_matches_reference_vectors():
    """`analyze_anomaly_nll` / `best_fscore` against the reference's own functions
    (fixture tests/golden/analyze_nll_vectors.json, generator make_analyze_vectors.py)."""
    import json
    import math
    from tracevae_amd.evaluation import analyze_anomaly_nll
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                        'analyze_nll_vectors.json')
    with open(path) as f:
        cases = json.load(f)['cases']
    assert len(cases) >= 4
    for c in cases:
        got = analyze_anomaly_nll(np.array(c['nll']), np.array(c['labels']), threshold=c['threshold'])
        assert set(got) == set(c['result'])
        for k, want in c['result'].items():
            assert got[k] == want or (math.isnan(want) and math.isnan(got[k])), k
    # an absent class gives NaN metrics, not an exception (the reference's log_error wrapper)
    r = analyze_anomaly_nll(np.arange(6.0), np.array([0, 0, 0, 0, 1, 1]))
    assert math.isnan(r['nll_latency']) and r['best_fscore_drop'] == 1.0 and 'fscore' not in r


def test_threshold_std_limit_and_bias_std_inference():
    """test.py:208-262 restated: known answers on hand-made inputs."""
    from tracevae_amd.test import infer_bias_std, infer_std_limit, infer_threshold
    # bias_std: a plain percentile
    assert infer_bias_std(np.arange(1001.0), 0.995) == pytest.approx(995.0)
    # threshold: constant scores -> that constant times amplify; clipped scores are dropped
    assert infer_threshold(np.full(50, 7.0), 0.9, 2.0, 100_000, np.random.RandomState(0)) == 14.0
    nll = np.concatenate([np.full(90, 3.0), np.full(10, 100_000.0)])
    assert infer_threshold(nll, 0.995, 1.0, 100_000, np.random.RandomState(1)) == 3.0
    nll = np.concatenate([np.full(90, 3.0), np.full(10, np.inf)])
    assert infer_threshold(nll, 0.995, 1.0, None, np.random.RandomState(1)) == 3.0
    # the same bootstrap as the reference's loop, drawn from an identical generator
    scores = np.random.RandomState(5).normal(10, 2, size=400)
    r1, r2 = np.random.RandomState(9), np.random.RandomState(9)
    cand = []
    for _ in range(10):
        sub = r2.choice(scores, replace=True, size=len(scores))
        sub = sub[sub < 100_000 - 1e-7]
        cand.append(np.percentile(sub, 99.5) * 1.5)
    assert infer_threshold(scores, 0.995, 1.5, 100_000, r1) == float(np.median(cand))
    # std limit: per operation percentile * amplify, unseen operations take the maximum
    op = np.array([0, 0, 0, 2, 2, 3])
    sd = np.array([1.0, 2.0, 3.0, 10.0, 20.0, 0.5], dtype=np.float32)
    lim = infer_std_limit(op, sd, 5, std_limit_p=1.0, std_limit_amplify=2.0)
    assert lim.dtype == np.float32 and lim.tolist() == [6.0, 40.0, 40.0, 1.0, 40.0]
    assert infer_std_limit(op, sd, 3, std_limit_fixed=0.25).tolist() == [0.25] * 3
    assert infer_std_limit(op, sd, 2, std_limit_p=0.5, std_limit_global=True).tolist() == [2.5] * 2


def test_flat_dataset_files_round_trip(tmp_path):
    """`data_process`: FlatTraces <-> .npz, and the concatenation of the three test sets."""
    from helpers import make_flat
    from tracevae_amd.data_process import concat_flat, load_flat, save_flat
    a, _ = make_flat(5, 30, 12, seed=1)
    b, _ = make_flat(3, 30, 12, seed=2, anomaly_ratio=1.0)
    path = str(tmp_path / 'processed' / 'a.npz')
    save_flat(path, a)
    back = load_flat(path)
    for f in ('node_offsets', 'u', 'v', 'node_type', 'avg_latency', 'max_latency', 'min_latency', 'labels'):
        assert np.array_equal(getattr(back, f), getattr(a, f)) and getattr(back, f).dtype == getattr(a, f).dtype
    both = concat_flat([a, b])
    both.validate(12)
    assert len(both) == 8 and int(both.node_offsets[-1]) == int(a.node_offsets[-1] + b.node_offsets[-1])
    assert np.array_equal(both.node_type[int(a.node_offsets[-1]):], b.node_type)
    assert np.array_equal(both.labels, np.concatenate([a.labels, b.labels]))


# ---------------------------------------------------------------------------------------
# data-parallel epoch loop: equal step counts, one collective per step, collective NaN vote
# ---------------------------------------------------------------------------------------
def _spawn_with_deadline(fn, args, nprocs, seconds=120):
    """mp.spawn that cannot hang the suite: a rank blocked in an unmatched collective is
    killed at the deadline and the test fails."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    deadline = time.time() + seconds
    while not ctx.join(timeout=1.0):
        if time.time() > deadline:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            pytest.fail('a rank was still blocked (unmatched collective?) at the deadline')


class _FakeStore(object):
    labels = None

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def _epoch_loop_worker(rank, world, port, tmp, nan_at):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from tracevae_amd.model import TraceVAE
    from tracevae_amd.train import NANLossError, StoreStream, Trainer, run_training
    torch.manual_seed(0)
    cfg = make_config(width=8, layers=2)
    cfg.train.max_epoch, cfg.train.struct_pretrain_epochs = 3, 1
    vae = TraceVAE(cfg.model, 30, max_node_count=8)

    log = []

    class HostTrainer(Trainer):
        """The real step skeleton (exchange, NaN vote) around stand-ins for the two GPU parts."""

        def local_gradients(self, G, eps=None):
            i = G._stream_index
            self.flat.grad.fill_(float(100 * self.epoch + 10 * i + rank))
            bad = nan_at is not None and (self.epoch, i) == nan_at
            return {'loss': torch.tensor(float('nan') if bad else float(i))}

        def update(self):
            log.append((self.epoch, self.latency_only, self.flat.grad.clone()))

    trainer = HostTrainer(vae, cfg, world_size=world)
    # 7 batches of 10 traces (the last one short): 7 % 2 == 1, the case that used to deadlock
    stream = StoreStream(_FakeStore(65), 10, shuffle=True, seed=3, rank=rank, world_size=world,
                         equal_steps=True)
    raised = None
    try:
        run_training(trainer, stream, cfg.train, cfg.train.struct_pretrain_epochs, log=lambda m: None)
    except NANLossError as ex:
        raised = ex.epoch
    torch.save({'log': log, 'raised': raised, 'tail': trainer.flat.tail_start,
                'batch_count': stream.batch_count, 'total': stream.total_batches},
               os.path.join(tmp, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_epoch_loop_equal_steps_across_phase_switch_gloo(tmp_path):
    world, port = 2, 33000 + os.getpid() % 2000
    _spawn_with_deadline(_epoch_loop_worker, (world, port, str(tmp_path), None), world)
    r = [torch.load(tmp_path / f'r{i}.pt') for i in range(world)]
    assert r[0]['total'] == r[1]['total'] == 6 and r[0]['batch_count'] == r[1]['batch_count'] == 3
    assert r[0]['raised'] is None and r[1]['raised'] is None
    assert len(r[0]['log']) == len(r[1]['log']) == 9            # 3 epochs x 3 steps on each rank
    tail = r[0]['tail']
    for step, (a, b) in enumerate(zip(r[0]['log'], r[1]['log'])):
        epoch, i0 = 1 + step // 3, 2 * (step % 3)
        assert a[0] == b[0] == epoch and a[1] == b[1] == (epoch > 1)
        mean = 100.0 * epoch + 10.0 * (i0 + 0.5) + 0.5        # ranks hold batches i0 and i0 + 1
        start = tail if epoch > 1 else 0                       # phase 2 exchanges the suffix only
        assert torch.all(a[2][start:] == mean) and torch.all(b[2][start:] == mean)
        if epoch > 1:
            assert torch.all(a[2][:tail] == 100.0 * epoch + 10.0 * i0)
            assert torch.all(b[2][:tail] == 100.0 * epoch + 10.0 * (i0 + 1) + 1)


def test_dp_nan_on_one_rank_raises_on_every_rank_gloo(tmp_path):
    world, port = 2, 35000 + os.getpid() % 2000
    # batch 3 belongs to rank 1 only; epoch 2 is already in phase 2
    _spawn_with_deadline(_epoch_loop_worker, (world, port, str(tmp_path), (2, 3)), world)
    r = [torch.load(tmp_path / f'r{i}.pt') for i in range(world)]
    assert r[0]['raised'] == 2 and r[1]['raised'] == 2
    assert len(r[0]['log']) == len(r[1]['log']) == 4            # epoch 1 + the first step of epoch 2


def test_nan_restart_loop(tmp_path):
    """train.py:510-531: restart (fresh try, output wiped) on a NaN in epoch 1, at most 10
    times; a NaN in a later epoch propagates."""
    from tracevae_amd.train import NANLossError, run_with_nan_restart
    out = tmp_path / 'results'
    calls, messages = [], []

    def job_factory(failures, epoch):
        def job():
            calls.append(1)
            (out / 'models').mkdir(parents=True, exist_ok=True)
            (out / 'models' / f'{len(calls)}.pt').write_text('x')
            (out / 'result' / 'test-anomaly').mkdir(parents=True, exist_ok=True)
            (out / 'config.json').write_text('{}')
            if len(calls) <= failures:
                raise NANLossError(epoch)
            return {'ok': len(calls)}
        return job

    assert run_with_nan_restart(job_factory(2, 1), str(out), log=messages.append) == {'ok': 3}
    assert len(messages) == 2 and '2-th time' in messages[1]
    # only the successful try's files are left; config.json is not wiped
    assert sorted(os.listdir(out / 'models')) == ['3.pt'] and (out / 'config.json').exists()
    del calls[:]
    with pytest.raises(NANLossError) as e:
        run_with_nan_restart(job_factory(1, 2), str(out), log=messages.append)
    assert e.value.epoch == 2 and len(calls) == 1
    del calls[:]
    with pytest.raises(NANLossError):
        run_with_nan_restart(job_factory(99, 1), str(out), log=messages.append)
    assert len(calls) == 11                                    # the first try + 10 restarts
