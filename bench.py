"""Throughput bench of the TraceVAE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the training hot path over one mini-batch of synthetic trace graphs
already resident in HBM: batch build (tv_batch_build) + q/p forward + backward + L2 +
gradient all-reduce (N > 1) + global-norm clip + optimizer update.  Workload = BASELINE.json
configs[1]: synthetic traces with <= 100 spans, 1k-operation vocabulary, default TraceVAE
dims with the train.sh overrides, MAX_NODE_COUNT = 100.  Weak scaling: every rank runs the
same per-GPU batch.  Rank 0 prints ONE JSON line; it also carries

  roofline     - the dominant kernel (fp32 MFMA GEMM): algorithmic FLOPs / HIP-event time
  cpu_baseline - the CPU oracle (a port: the reference cannot run here) on a bounded sample
  scoring      - K-sample NLL scoring throughput (the second half of BASELINE.json's metric)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The GEMM multiplies fp32 operands on the 16-bit matrix cores (csrc/gemm_f32.hip): the layer
# products and weight gradients (forms 1 / 2: the wide kernels, which carry the time) as two fp16
# pieces per element and THREE partial products per algorithmic product, every other shape
# (form 0) as three bf16 pieces and SIX partial products; fp32 accumulation either way.  The
# matrix-core ceiling in algorithmic (2*M*N*K) flops is the dense 16-bit peak of
# MI355X_MICROARCH.md (2500 TFLOP/s for fp16 and bf16 alike) over the number of partial products.
MFMA16_PEAK_TFLOPS = 2500.0
GEMM_KERNEL = 'gemm_wide_h_kernel / gemm_kmajor_wide_h_kernel'
GEMM_PEAK_TFLOPS = MFMA16_PEAK_TFLOPS / 3.0
GEMM_KERNEL_OTHER = 'gemm_kernel'
GEMM_PEAK_OTHER_TFLOPS = MFMA16_PEAK_TFLOPS / 6.0
GEMM_DTYPE = ('f32 in / f32 out; wide GEMM forms: fp16x2 split (x0 + 2^-11 x1, power-of-two operand '
              'scale), 3 terms, fp32 accumulate; other GEMM shapes: bf16x3 split, 6 terms')


def form_peak(form):
    return GEMM_PEAK_TFLOPS if form in (1, 2) else GEMM_PEAK_OTHER_TFLOPS


def gemm_profile_begin():
    from tracevae_amd import hip
    hip.lib().tv_gemm_profile_begin()


def gemm_profile_end():
    """[(ms, form, m, n, k)] of the MAIN kernel of every product since `gemm_profile_begin`
    (HIP events inside the library, on the launch stream, around that kernel alone)."""
    from tracevae_amd import hip
    cap = 1 << 16
    buf = np.zeros([cap, 5], dtype=np.float64)
    n = int(hip.lib().tv_gemm_profile_end(buf.ctypes.data, cap))
    return [tuple(buf[i]) for i in range(min(n, cap))]


def gemm_roofline(tags, elapsed, traffic, traffic_unit, steps, kernel_records=None):
    """The GEMM family from the per-(shape, form) timer table: the dominant kernels (the fp16
    forms) against their peak, the bf16 form beside them, and the ideal time of all of it.
    The timed region of a product includes its own preparation passes (max |operand|, split of
    B, tail / split-K reductions)."""
    by_form = {}
    for tag, (c, w, ms) in tags.items():
        rec = by_form.setdefault(1 if tag[5] in (1, 2) else 0, [0, 0.0, 0.0])
        rec[0] += c; rec[1] += w; rec[2] += ms
    ideal_s = sum(w / (form_peak(f) * 1e12) for f, (c, w, ms) in by_form.items())
    total_ms = sum(ms for c, w, ms in by_form.values())
    total_w = sum(w for c, w, ms in by_form.values())
    if total_ms <= 0:
        return None, 0.0, 0.0
    dom = 1 if 1 in by_form else 0
    c, w, ms = by_form[dom]
    tf = w / (ms * 1e-3) / 1e12
    roof = {'bound': 'mfma', 'kernel': GEMM_KERNEL if dom else GEMM_KERNEL_OTHER, 'achieved': tf,
            'peak': form_peak(dom), 'unit': 'TFLOP/s',
            'peak_is': 'dense 16-bit MFMA peak (2500 TFLOP/s) / partial products per algorithmic '
                       'product (3 on the fp16 forms, 6 on the bf16 form); achieved counts '
                       'algorithmic 2*M*N*K flops over the whole library call (operand-maximum, '
                       'B-split and reduction passes included)',
            'frac': tf / form_peak(dom), 'traffic': traffic, 'traffic_unit': traffic_unit,
            'launches': c, 'avg_ms': ms / max(c, 1), 'share_of_step': ms * 1e-3 / elapsed,
            'all_gemm': {'achieved': total_w / (total_ms * 1e-3) / 1e12, 'launches': sum(
                             v[0] for v in by_form.values()),
                         'share_of_step': total_ms * 1e-3 / elapsed,
                         'frac_of_ideal': ideal_s / (total_ms * 1e-3)}}
    if kernel_records:
        # the dominant kernel ALONE (events around it inside the library): this is the figure the
        # rocprofv3 per-kernel averages of profiles/ cross-check; the call-level one stays beside it
        sel = [r for r in kernel_records if (r[1] in (1.0, 2.0)) == bool(dom)]
        kms = sum(r[0] for r in sel)
        kw = sum(2.0 * r[2] * r[3] * r[4] for r in sel)
        if kms > 0:
            roof['whole_call'] = {'achieved': tf, 'frac': tf / form_peak(dom), 'avg_ms': ms / max(c, 1),
                                  'note': 'HIP events around the library call: operand-maximum, '
                                          'B-split and reduction passes included'}
            ktf = kw / (kms * 1e-3) / 1e12
            roof.update({'achieved': ktf, 'frac': ktf / form_peak(dom), 'launches': len(sel),
                         'avg_ms': kms / len(sel), 'share_of_step': kms * 1e-3 / elapsed,
                         'peak_is': 'dense 16-bit MFMA peak (2500 TFLOP/s) / partial products per '
                                    'algorithmic product (3 on the fp16 forms, 6 on the bf16 form); '
                                    'achieved = algorithmic 2*M*N*K flops / duration of the kernel '
                                    'alone (HIP events around it on its stream, tv_gemm_profile_*)'})
    if dom and 0 in by_form:
        c0, w0, ms0 = by_form[0]
        roof['bf16_form'] = {'kernel': GEMM_KERNEL_OTHER, 'achieved': w0 / (ms0 * 1e-3) / 1e12,
                             'peak': GEMM_PEAK_OTHER_TFLOPS,
                             'frac': w0 / (ms0 * 1e-3) / 1e12 / GEMM_PEAK_OTHER_TFLOPS,
                             'launches': c0, 'share_of_step': ms0 * 1e-3 / elapsed}
    top = sorted(tags.items(), key=lambda kv: -kv[1][2])[:4]
    roof['top_shapes'] = [
        {'m': t[0], 'n': t[1], 'k': t[2], 'trans_a': t[3], 'trans_b': t[4], 'form': t[5],
         'launches_per_step': c / steps, 'ms_per_step': ms / steps,
         'achieved': w / (ms * 1e-3) / 1e12, 'frac': w / (ms * 1e-3) / 1e12 / form_peak(t[5])}
        for t, (c, w, ms) in top]
    return roof, ideal_s, total_ms
HBM_PEAK_GBS = 8000.0
# HBM bytes per GEMM launch of the default workload, from the committed --pmc passes
# (tools/summarize_profile.py traffic; FETCH_SIZE x2 + WRITE_SIZE, separate passes)
TRAFFIC_PROFILE = os.path.join(ROOT, 'profiles', 'r03_hbm_traffic.json')
SCORING_TRAFFIC_PROFILE = os.path.join(ROOT, 'profiles', 'r03_scoring_hbm_traffic.json')


def measured_traffic(args, kernel=None, profile=TRAFFIC_PROFILE):
    """PMC counters cannot be read from inside the timed run; the figures come from the
    committed rocprofv3 --pmc passes over this same command (tools/profile_bench.sh), and only
    for the workload they were collected on.  `kernel=None`: the GEMM family; else the HBM
    bytes per launch of the kernel whose name starts with `kernel`."""
    if (args.batch_size, args.max_node_count, args.num_operations, args.gnn_width, args.z_dim,
            args.score_k, args.score_batch) != (512, 100, 1001, None, 10, 100, 128):
        return None
    try:
        with open(profile) as f:
            doc = json.load(f)
        if kernel is None:
            return float(doc['family_bytes_per_launch'])
        hits = [v for k, v in doc['kernels'].items() if k.startswith(kernel)]
        n = sum(v['launches'] for v in hits)
        return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in hits) / n if n else None
    except (OSError, KeyError, ValueError):
        return None


def message_passing_entry(name, summary, tags, elapsed, traffic):
    """One message-passing kernel against the HBM peak.  `frac` prices the COMPULSORY bytes of a
    launch (every row in once, every row out once, index and coefficient arrays once: what must
    cross the HBM interface however good the caches are) and cannot exceed 1; the per-edge model
    of SURVEY.md section 8d (a neighbour row counted once per in-edge, although the L2 serves
    most of those gathers) is kept beside it as `per_edge_model`."""
    c, w, ms = summary[name]
    compulsory = sum(tag * rec[0] for tag, rec in tags.items() if tag is not None)
    rate = compulsory / (ms * 1e-3) / 1e9
    out = {'bound': 'hbm', 'achieved': rate, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': rate / HBM_PEAK_GBS, 'bytes_model': 'compulsory (rows once in, once out)',
           'per_edge_model': {'achieved': w / (ms * 1e-3) / 1e9,
                              'ratio_to_peak': w / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              'note': 'SURVEY.md 8d bytes (a row per in-edge); gathers the L2 '
                                      'serves are counted, so this is not an HBM fraction'},
           'launches': c, 'avg_ms': ms / c, 'share_of_step': ms * 1e-3 / elapsed,
           'traffic': traffic, 'traffic_unit': 'PMC bytes/launch (FETCH_SIZE x2 + WRITE_SIZE)'}
    if traffic:
        out['traffic_rate_GBs'] = traffic / (ms / c * 1e-3) / 1e9
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-size', type=int, default=512, help='traces per GPU per step')
    ap.add_argument('--max-node-count', type=int, default=100)
    ap.add_argument('--num-operations', type=int, default=1001)
    ap.add_argument('--pool', type=int, default=1 << 20,
                    help='traces of the HBM-resident store per rank (configs[1]: one million)')
    ap.add_argument('--unique-traces', type=int, default=8192,
                    help='traces built by the object-level generator; the store tiles them up to '
                         '--pool with per-copy latencies (synth.tile_flat)')
    ap.add_argument('--sweep-batches', type=str, default='64,128',
                    help='other per-GPU batch sizes timed after the main leg (N=1; "" = skip); '
                         '64 is the reference default (train.py:73)')
    ap.add_argument('--score-k', type=int, default=100,
                    help='n_z of the scoring leg: BASELINE.json configs[3] scores with K = 100 (0 = skip)')
    ap.add_argument('--score-batch', type=int, default=128)
    ap.add_argument('--score-steps', type=int, default=3)
    ap.add_argument('--prior-samples', type=int, default=128,
                    help='traces drawn per call in the prior-sampling leg (N=1 only; 0 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--scoring-only', action='store_true',
                    help='run the scoring leg alone and print its object (profiling passes)')
    ap.add_argument('--gnn-width', type=int, default=None, help='per-head GAT width (default 500)')
    ap.add_argument('--z-dim', type=int, default=10, help='z / z2 dimension (train.sh: 10)')
    ap.add_argument('--gemm-shapes', action='store_true', help='per-shape GEMM table on stderr')
    ap.add_argument('--seed', type=int, default=1234)
    return ap.parse_args()


def make_config(width=None, z_dim=10):
    from tracevae_amd.config import ExpConfig, parse_flags
    cfg = ExpConfig()
    parse_flags(cfg, [f'--model.struct.z_dim={z_dim}', '--model.struct.decoder.use_prior_flow=true',
                      '--train.z_unit_ball_reg=1', f'--model.latency.z2_dim={z_dim}',
                      '--model.latency.decoder.condition_on_z=true'])
    if width is not None:     # configs[4]: doubled latent / GNN / MLP widths
        mc = cfg.model
        for sec in (mc.struct.encoder, mc.struct.decoder, mc.latency.encoder, mc.latency.decoder):
            sec.gnn_layers = [width] * len(sec.gnn_layers)
        if width >= 1000:
            sd, ld = mc.struct.decoder, mc.latency.decoder
            sd.latent_embedding_size *= 2
            sd.node_count_layers = [2 * w for w in sd.node_count_layers]
            sd.graph_embedding_layers = [2 * w for w in sd.graph_embedding_layers]
            ld.z2_given_z_hidden_layers = [2 * w for w in ld.z2_given_z_hidden_layers]
    return cfg


def _median_rate(fn, units, warmup, timed):
    """`units` processed per call of `fn`: median units/s over `timed` calls after `warmup`."""
    for _ in range(warmup):
        fn()
    times = []
    for _ in range(timed):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return units / float(np.median(times)), times


def cpu_baseline(cfg, flat, lr, V, M, score_k):
    """The CPU oracle (a port: the reference itself needs dgl / tensorkit / mltk and cannot run
    here) on bounded samples of the same synthetic set, SURVEY.md section 8d's protocol: warm-up
    steps, then the median of the timed ones; all host cores, and a one-thread figure.
      training: batch build + forward + backward + clip + RMSprop on the reference's own batch
                size of 64 traces (train.py:73);
      scoring:  batch build + encoders + K decoder passes + log-mean-exp on 16 traces."""
    from oracle.batching import build_batch
    from oracle.model import OracleTraceVAE
    from tracevae_amd.model import TraceVAE
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import oracle_state, subset_flat
    cores = os.cpu_count() or 1
    # the GPU box reports the whole machine's CPUs; its share is OMP_NUM_THREADS (16)
    threads = min(cores, int(os.environ.get('OMP_NUM_THREADS', '16') or 16))
    torch.manual_seed(0)
    vae = TraceVAE(cfg.model, V, max_node_count=M)
    for m in vae.modules():
        if hasattr(m, 'initialized'):
            m.initialized = True
    P, buf, names = oracle_state(vae, torch.float32)
    params = [P[n] for n in names]
    opt = torch.optim.RMSprop(params, lr=1e-3, alpha=0.99, eps=1e-8)
    orc = OracleTraceVAE(cfg.model, V, M, dtype=torch.float32)
    mean, std = lr.tables(V)
    zd, z2d = cfg.model.struct.z_dim, cfg.model.latency.z2_dim

    def batch_of(n, first=0):
        sub = subset_flat(flat, np.arange(first, first + n))
        ob = build_batch(sub.node_offsets, sub.u, sub.v, sub.node_type, sub.avg_latency,
                         sub.max_latency, sub.min_latency, mean, std, M)
        ob['node_offsets'] = sub.node_offsets
        return ob

    def train_step(n):
        ob = batch_of(n)
        opt.zero_grad()
        out = orc.train_step(P, buf, ob, torch.randn(n, zd), torch.randn(n, z2d), names,
                             l2_reg=cfg.train.l2_reg, z_unit_ball_reg=1.0)
        out['loss'].backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()

    def score_step(n, k):
        ob = batch_of(n)
        orc.eval_step(P, buf, ob, torch.randn(k, n, zd), torch.randn(k, n, z2d),
                      latency_log_prob_weight=True)

    out = {'unit': 'traces/s', 'cores': cores, 'threads': threads, 'kind': 'port'}
    torch.set_num_threads(threads)
    out['value'], t_train = _median_rate(lambda: train_step(64), 64, 2, 5)
    if score_k > 0:
        out['scoring_value'], t_score = _median_rate(lambda: score_step(16, score_k), 16, 1, 3)
    torch.set_num_threads(1)
    out['value_1_thread'], t_one = _median_rate(lambda: train_step(8), 8, 1, 3)
    torch.set_num_threads(threads)
    out['sample'] = (
        f'oracle (fp32, stock CPU PyTorch ops) on traces of the same synthetic set: train step '
        f'(batch build + fwd + bwd + clip + RMSprop) on 64 traces, 2 warm-up + 5 timed steps, '
        f'median {1e3 * float(np.median(t_train)):.0f} ms, {threads} threads; one thread: 8 traces, '
        f'1 + 3 steps, median {1e3 * float(np.median(t_one)):.0f} ms'
        + (f'; scoring K={score_k} on 16 traces, 1 + 3 calls, median '
           f'{1e3 * float(np.median(t_score)):.0f} ms' if score_k > 0 else ''))
    return out


def main():
    args = parse_args()
    if os.environ.get('TRACEVAE_BENCH_WATCHDOG'):      # debugging aid: dump all stacks and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['TRACEVAE_BENCH_WATCHDOG']), exit=True)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    if world > 1:
        # one process per GPU over RCCL; TRACEVAE_DIST_BACKEND=gloo rehearses the N > 1 control
        # flow on a box with fewer GPUs than ranks (ranks then share devices)
        backend = os.environ.get('TRACEVAE_DIST_BACKEND', 'nccl')
        dev_index = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    from tracevae_amd import ops
    from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
    from tracevae_amd.evaluation import eval_step
    from tracevae_amd.model import TraceVAE
    from tracevae_amd.synth import SyntheticSpec, generate_flat, tile_flat
    from tracevae_amd.train import Trainer

    V, M, B = args.num_operations, args.max_node_count, args.batch_size
    cfg = make_config(args.gnn_width, args.z_dim)
    spec = SyntheticSpec(num_operations=V, max_spans=M, seed=args.seed)
    # every rank generates its own shard of the synthetic set (seeded by rank)
    unique = min(args.unique_traces, args.pool)
    flat, lr = generate_flat(spec, unique, seed=args.seed + 1 + rank)
    copies = max(args.pool // unique, 1)
    flat = tile_flat(flat, copies, seed=args.seed + 101 + rank)
    store = DeviceTraceStore(flat, lr, V)
    store_bytes = sum(t.numel() * t.element_size() for t in (
        store.u, store.v, store.node_type, store.avg_latency, store.min_latency, store.max_latency))
    torch.manual_seed(args.seed)          # identical initial weights on every rank
    vae = TraceVAE(cfg.model, V, max_node_count=M).to(dev)
    for m in vae.modules():
        if hasattr(m, 'initialized'):
            m.initialized = True
    trainer = Trainer(vae, cfg, world_size=world)
    rng = np.random.RandomState(args.seed + rank)

    def train_one(batch=B, check_nan=True):
        # check_nan=True is Trainer.step's default: the NaN decision of train.py:272-274 with
        # the update guarded on the device and the exception raised one step later
        ids = rng.randint(0, len(store), size=batch)
        G = TraceGraphBatch.from_store(store, ids, M)
        return trainer.step(G, check_nan=check_nan)['loss']

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.scoring_only:
        args.steps = args.warmup = 0
        args.sweep_batches = ''
    for _ in range(args.warmup):
        train_one()
    barrier()
    t0 = time.perf_counter()
    losses = [train_one() for _ in range(args.steps)]
    trainer.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    # the same K steps once more with HIP events around every library call (`ops.TIMER`): the
    # rooflines below come from this pass.  It takes the operator-level entry points (one call
    # per kernel) where the product's default path crosses the boundary once per GAT layer
    # (tv_gat_layer_train_*): the same launches, bracketed one by one.
    ops.TIMER.reset()
    ops.TIMER.enabled = True
    gemm_profile_begin()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        train_one()
    barrier()
    elapsed_timed = max(time.perf_counter() - t1, 1e-9)
    ops.TIMER.enabled = False
    train_gemm_kernels = gemm_profile_end()
    kernel_summary = ops.TIMER.summary()
    # per-shape / per-tag tables of the TRAINING leg, taken before any other leg resets the timer
    train_gemm_tags = ops.TIMER.by_tag('gemm_f32')
    train_mp_tags = {n: ops.TIMER.by_tag(n) for n in ('gat_aggregate_fwd', 'gat_aggregate_bwd')}
    if args.gemm_shapes and rank == 0:
        agg = {}
        for ms, form, m, n, k in train_gemm_kernels:
            rec = agg.setdefault((int(form), int(m), int(n), int(k)), [0, 0.0])
            rec[0] += 1; rec[1] += ms
        for (form, m, n, k), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
            print(f'kernel alone form={form} m={m:7d} n={n:5d} k={k:7d} x{cnt / args.steps:5.1f}/step '
                  f'{ms / cnt * 1e3:8.1f} us  {2.0 * m * n * k * cnt / ms / 1e9:6.1f} TF', file=sys.stderr)
        rows = sorted(ops.TIMER.by_tag('gemm_f32').items(), key=lambda kv: -kv[1][2])
        for (m, n, k, ta, tb, form), (cnt, work, ms) in rows:
            print(f'gemm m={m:7d} n={n:5d} k={k:7d} ta={int(ta)} tb={int(tb)} form={form}  x{cnt / args.steps:5.1f}/step '
                  f'{ms / args.steps:7.3f} ms/step  {work / ms / 1e9:6.1f} TF', file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_vals = torch.stack(losses).cpu().numpy() if losses else np.zeros(0)
    if not np.all(np.isfinite(loss_vals)):
        raise RuntimeError(f'non-finite training loss in the timed region: {loss_vals}')

    # ---- the same step as the product's default `Trainer.step` issues it: with the reference's
    # per-step NaN check (train.py:273, a host sync every step); and other batch sizes
    def timed(fn, steps):
        barrier()
        t1 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64,
                              device=dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    dt_nan = timed(lambda: train_one(B, 'now'), args.steps)
    dt_unchecked = timed(lambda: train_one(B, False), args.steps)
    elapsed = max(elapsed, 1e-9)
    sweep = {}
    if world == 1 and args.sweep_batches:
        for b in [int(x) for x in args.sweep_batches.split(',') if x]:
            for _ in range(3):
                train_one(b)
            n_sw = max(args.steps, 20)
            dt_b = timed(lambda: train_one(b), n_sw)
            sweep[str(b)] = {'ms_per_step': 1e3 * dt_b / n_sw, 'traces_per_sec': b * n_sw / dt_b}
        sweep[str(B)] = {'ms_per_step': 1e3 * elapsed / args.steps,
                         'traces_per_sec': B * args.steps / elapsed}

    # ---- scoring leg (not part of `value`; reported next to it): every rank scores its own
    # shard of traces, no communication; whole-job rate over all ranks.
    # BASELINE.json configs[3]: K = 100 samples per trace.
    scoring = None
    if args.score_k > 0:
        vae.eval()
        SB, K = args.score_batch, args.score_k
        def score_one():
            ids = rng.randint(0, len(store), size=SB)
            return eval_step(vae, TraceGraphBatch.from_store(store, ids, M), K,
                             latency_log_prob_weight=True, clip_nll=100_000.0)[1]
        score_one()
        barrier()
        ops.TIMER.reset()
        ops.TIMER.enabled = True
        gemm_profile_begin()
        ts = time.perf_counter()
        for _ in range(args.score_steps):
            score_one()
        barrier()
        dt = time.perf_counter() - ts
        ops.TIMER.enabled = False
        score_gemm_kernels = gemm_profile_end()
        score_kernels = ops.TIMER.summary()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64,
                             device=dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        scoring = {'metric': 'nll_scoring_traces_per_sec',
                   'value': SB * world * args.score_steps / dt, 'unit': 'traces/s', 'n_z': K,
                   'batch_per_gpu': SB, 'n_gpus': world, 'steps': args.score_steps,
                   'ms_per_step': 1e3 * dt / args.score_steps,
                   'workload': f'BASELINE.json configs[3]: K={K}-sample NLL scoring, {SB} traces '
                               f'per call per GPU, decoders on K*B*M = {K * SB * M} rows'}
        sroof, ideal, _ = gemm_roofline(
            ops.TIMER.by_tag('gemm_f32'), dt, measured_traffic(args, None, SCORING_TRAFFIC_PROFILE),
            'bytes/launch (profiles/r03_scoring_hbm_traffic.json)', args.score_steps,
            score_gemm_kernels)
        if sroof is not None:
            scoring['roofline'] = sroof
            a = score_kernels.get('gat_aggregate_fwd')
            if a and a[2] > 0:
                entry = message_passing_entry(
                    'gat_aggregate_fwd', score_kernels, ops.TIMER.by_tag('gat_aggregate_fwd'), dt,
                    measured_traffic(args, 'gat_aggregate_fwd', SCORING_TRAFFIC_PROFILE))
                ideal += entry['achieved'] * 1e9 * a[2] * 1e-3 / (HBM_PEAK_GBS * 1e9)
                scoring['gat_aggregate_fwd'] = entry
            scoring['step_roofline'] = {'ideal_ms_per_step': 1e3 * ideal / args.score_steps,
                                        'measured_ms_per_step': 1e3 * dt / args.score_steps,
                                        'frac': ideal / dt}

    # ---- prior-sampling leg (section 8f row 4; N=1 only, not part of `value`): p() without
    # an observed graph, and the sampled-graph index kernels on their own against HBM
    prior = None
    if args.prior_samples > 0 and world == 1:
        from tracevae_amd.batch import build_sampled_graphs
        vae.eval()
        n = args.prior_samples
        with torch.no_grad():
            net = vae.p(n_z=n)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(3):
                net = vae.p(n_z=n)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - ts) / 3
        adj = net['adj'].tensor.reshape(n, -1)
        graph, _ = build_sampled_graphs(adj, M)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            build_sampled_graphs(adj, M)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        algo = 4.0 * adj.numel() + 8.0 * (5 * graph.num_edges + 2 * n * M + 1)
        prior = {'metric': 'prior_sampled_traces_per_sec', 'value': n / dt, 'unit': 'traces/s',
                 'samples_per_call': n, 'ms_per_call': 1e3 * dt,
                 'sampled_graph_build': {'bound': 'hbm', 'edges': graph.num_edges,
                                         'ms_incl_host_readback': ms,
                                         'achieved': algo / (ms * 1e-3) / 1e9,
                                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                         'frac': algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}

    if args.scoring_only:
        if rank == 0:
            print(json.dumps({'scoring': scoring}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if rank == 0:
        value = B * world * args.steps / elapsed
        roof, gemm_ideal_s, gemm_ms = gemm_roofline(
            train_gemm_tags, elapsed, measured_traffic(args),
            'bytes/launch (profiles/r03_hbm_traffic.json)', args.steps, train_gemm_kernels)
        others = {}
        pmc_names = {'gat_aggregate_fwd': 'gat_aggregate_fwd',
                     'gat_aggregate_bwd': 'gat_bwd_fused_kernel'}
        for name in ('gat_aggregate_fwd', 'gat_aggregate_bwd'):
            if name in kernel_summary and kernel_summary[name][2] > 0:
                others[name] = message_passing_entry(
                    name, kernel_summary, train_mp_tags[name], elapsed,
                    measured_traffic(args, pmc_names[name]))
        # step-level figure of SURVEY.md section 8d: sum over the attributed kernels of their
        # ideal time (flops / MFMA peak, or algorithmic bytes / HBM peak) over the measured step;
        # kernels without a work model count as zero ideal time, so this is a lower bound
        ideal_s = attributed_ms = 0.0
        ideal_s += gemm_ideal_s
        attributed_ms += gemm_ms
        for name, entry in others.items():        # compulsory bytes at the HBM peak
            ideal_s += entry['achieved'] * entry['avg_ms'] * entry['launches'] * 1e-3 / HBM_PEAK_GBS
            attributed_ms += kernel_summary[name][2]
        # (the same sum with every GEMM priced at the six-product bound of round 2's arithmetic,
        # 416.7 TFLOP/s, is kept beside it so that rounds can be compared)
        ideal_six = ideal_s
        if 'gemm_f32' in kernel_summary:
            ideal_six += kernel_summary['gemm_f32'][1] / (GEMM_PEAK_OTHER_TFLOPS * 1e12) - gemm_ideal_s
        step_roof = {'ideal_ms_per_step': 1e3 * ideal_s / args.steps,
                     'measured_ms_per_step': 1e3 * elapsed / args.steps,
                     'frac': ideal_s / elapsed,
                     'frac_with_gemm_at_six_product_bound': ideal_six / elapsed,
                     'attributed_share_of_step': attributed_ms * 1e-3 / elapsed,
                     'covers': 'gemm_f32 against the 16-bit MFMA peak / 3 (fp16 forms) or / 6 (bf16 form), gat_aggregate_fwd/bwd '
                               '(compulsory bytes) against HBM; every other kernel counts as zero '
                               'ideal time (lower bound). '
                               'The step is compute-dominated: the HBM figures refer to the '
                               'message-passing kernels only'}
        line = {
            'metric': 'train_step_traces_per_sec', 'value': value, 'unit': 'traces/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'ms_per_step_with_sync_nan_check': 1e3 * dt_nan / args.steps,
            'ms_per_step_without_nan_check': 1e3 * dt_unchecked / args.steps,
            'ms_per_step_instrumented': 1e3 * elapsed_timed / max(args.steps, 1),
            'instrumented': 'roofline / other_kernels / step_roofline come from a second pass of '
                            'the same K steps with HIP events around every library call, through '
                            'the operator-level entry points (the default path issues a GAT layer '
                            'as one library call each way: the same launches)',
            'nan_check': 'value / ms_per_step / batch_sweep: Trainer.step as the product calls it '
                         '(check_nan=True): the update is guarded by the loss on the device '
                         '(tv_optim_step_guarded), the loss travels to pinned host memory behind '
                         'the step and NANLossError is raised by the next step ahead of its own '
                         'update - the outcome of train.py:272-274 without a host sync per step; '
                         'ms_per_step_with_sync_nan_check: check_nan="now", the host waits for the '
                         'loss inside every step as the reference does; '
                         'ms_per_step_without_nan_check: no check and no guard',
            'batch_sweep': sweep or None,
            'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': GEMM_DTYPE, 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[1]: synthetic traces <=100 spans, '
                                   '1k-op vocab, default TraceVAE dims (train.sh flags), '
                                   'MAX_NODE_COUNT=100, training step',
                       'batch_per_gpu': B, 'global_batch': B * world, 'max_node_count': M,
                       'num_operations': V, 'store_traces_per_gpu': len(store),
                       'store_bytes_per_gpu': store_bytes,
                       'store': f'{len(store)} traces resident in HBM = {copies} copies of '
                                f'{unique} generated traces with per-copy latencies; batches '
                                f'are drawn uniformly from the whole store',
                       'parallelism': f'dp{world}'},
            'roofline': roof, 'step_roofline': step_roof, 'other_kernels': others,
            'scoring': scoring,
            'prior_sampling': prior,
            'final_loss': float(loss_vals[-1]),
        }
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only (bench contract)
            line['cpu_baseline'] = cpu_baseline(cfg, flat, lr, V, M, args.score_k)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
