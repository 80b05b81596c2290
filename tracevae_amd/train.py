"""Training step, optimizer and loop of the accelerated path.

Restates `tracegnn/models/trace_vae/train.py`:

* `train_step` ELBO terms, KL beta, L2, unit-ball regularisers, NaN check      :213-277
* two-phase schedule (all parameters, then `latency_vae.*` only with the rest
  of the model in eval mode), LR annealing from the global epoch                :204-208, 374-383, 455-487
* clip by global norm 10 + RMSprop / Adam (tensorkit `fit_model`)               :369-372, 417-426
* `models/{epoch}.pt`, `models/final.pt` = `torch.save(vae.state_dict())`        :343-352

MI355X-side design: all parameters live in ONE flat fp32 buffer (struct part first, the
`latency_vae.*` part last, so the phase-2 subset is a contiguous suffix), gradients in a
second one.  That makes the data-parallel exchange a single RCCL all-reduce over xGMI, and
L2 / global-norm / clip / update three streaming kernels over the flat buffers
(`csrc/optim.hip`).
"""
import json
import math
import os
import sys
import time
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import hip, ops
from .batch import DeviceTraceStore, TraceGraphBatch
from .config import ExpConfig, config_to_dict
from .model import TraceVAE

__all__ = ['NANLossError', 'FlatParameters', 'FlatOptimizer', 'compute_train_loss', 'Trainer',
           'l2_parameter_names', 'learning_rate_for_epoch', 'save_model', 'StoreStream', 'fit',
           'run_training', 'run_with_nan_restart', 'init_distributed', 'all_reduce_sum',
           'broadcast_model', 'average_batch_norm_buffers', 'load_model', 'main']


class NANLossError(Exception):
    def __init__(self, epoch):
        super().__init__(epoch)

    @property
    def epoch(self) -> Optional[int]:
        return self.args[0]

    def __str__(self):
        return f'NaN loss encountered at epoch {self.epoch}'


def l2_parameter_names(all_names: Sequence[str], part_names: Sequence[str]) -> List[str]:
    """Which parameters enter the L2 term.  The reference zips the *trained* parameter list
    with the *full* name list (`zip(params, param_names)`, train.py:253-255), so in phase 2
    the i-th trained parameter is filtered by the i-th name of the whole model.  Restated
    as is (SURVEY.md appendix D lists the reference's quirks that affect parity)."""
    return [p for p, n in zip(part_names, all_names) if 'bias' not in n]


def learning_rate_for_epoch(cfg, epoch: int) -> float:
    """train.py:374-380: lr = initial_lr * ratio ** (epoch // anneal_epochs), global epoch."""
    return cfg.initial_lr * cfg.lr_anneal_ratio ** int(epoch // cfg.lr_anneal_epochs)


class FlatParameters(object):
    """Re-homes every parameter of `module` into one flat fp32 buffer (and `.grad` into a
    second one).  Order: everything outside `latency_vae.` first, `latency_vae.` last."""

    def __init__(self, module: torch.nn.Module, tail_prefix: str = 'latency_vae.'):
        named = list(module.named_parameters())          # de-duplicated, registration order
        self.all_names = [n for n, _ in named]
        head = [(n, p) for n, p in named if not n.startswith(tail_prefix)]
        tail = [(n, p) for n, p in named if n.startswith(tail_prefix)]
        ordered = head + tail
        device = ordered[0][1].device
        sizes = [p.numel() for _, p in ordered]
        # 64-element (256 B) aligned segments keep every parameter 16-B aligned for float4
        starts, cur = [], 0
        for s in sizes:
            starts.append(cur)
            cur += (s + 63) // 64 * 64
        self.total = cur
        self.flat = torch.zeros(cur, dtype=torch.float32, device=device)
        # the gradient bucket carries one more (256-B aligned) slot behind the gradients: the
        # NaN flag of the step, so that ONE all-reduce moves both and every rank takes the same
        # NaN decision (Trainer.exchange)
        self.bucket = torch.zeros(cur + 64, dtype=torch.float32, device=device)
        self.grad = self.bucket[:cur]
        self.flag_index = cur
        self.names = [n for n, _ in ordered]
        self.slices: Dict[str, Tuple[int, int]] = {}
        self.tail_start = cur
        # the gradient views are made once: slicing + reshaping the bucket per parameter and step
        # cost the host 2 ms per step (177 parameters), a quarter of a batch-64 step
        self.grad_views: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for (name, p), start, size in zip(ordered, starts, sizes):
                view = self.flat[start:start + size].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self.grad_views[name] = self.grad[start:start + size].view(p.shape)
                p.grad = self.grad_views[name]
                self.slices[name] = (start, start + size)
                if name.startswith(tail_prefix) and start < self.tail_start:
                    self.tail_start = start
        self.params = dict(ordered)

    def mask(self, names: Iterable[str]) -> torch.Tensor:
        m = torch.zeros(self.total, dtype=torch.uint8)
        for n in names:
            a, b = self.slices[n]
            m[a:b] = 1
        return m.to(self.flat.device)

    def zero_grad(self) -> None:
        self.grad.zero_()
        for n, p in self.params.items():          # autograd may have replaced a view
            a, b = self.slices[n]
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * a:
                p.grad = self.grad_views[n]


    # -- one backward pass without a per-parameter accumulate kernel ----------------------
    def begin_backward(self) -> None:
        """Detach the `.grad` views: autograd then hands every parameter its freshly computed
        gradient tensor instead of launching one `grad += g` kernel per parameter."""
        for p in self.params.values():
            p.grad = None

    def collect_grads(self) -> None:
        """Gather the gradients into the flat bucket with multi-tensor copies (same values as
        accumulating into a zeroed bucket), then re-attach the views."""
        self.grad.zero_()
        dst, src = [], []
        for n, p in self.params.items():
            view = self.grad_views[n]
            if p.grad is not None:
                dst.append(view)
                src.append(p.grad)
            p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)


class FlatOptimizer(object):
    """RMSprop / Adam over a range of the flat buffer, with clip-by-global-norm
    (`tv_sumsq_partial` -> `tv_sum_partials` -> `tv_optim_step`)."""

    def __init__(self, flat: FlatParameters, kind: str = 'rmsprop', start: int = 0,
                 end: Optional[int] = None, alpha: float = 0.99, betas=(0.9, 0.999),
                 eps: float = 1e-8):
        self.flat, self.kind = flat, kind.lower()
        if self.kind not in ('rmsprop', 'adam'):
            raise ValueError(f'unknown optimizer {kind!r}')
        self.start, self.end = start, flat.total if end is None else end
        n = self.end - self.start
        dev = flat.flat.device
        self.state1 = torch.zeros(n, dtype=torch.float32, device=dev)
        self.state2 = torch.zeros(n, dtype=torch.float32, device=dev) if self.kind == 'adam' else None
        self.alpha, self.betas, self.eps = alpha, betas, eps
        self.lr = 1e-3
        self.steps = 0
        self._partial = torch.empty(hip.query('tv_reduce_blocks', n), dtype=torch.float64, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def set_lr(self, lr: float) -> None:
        self.lr = float(lr)

    def step(self, global_clip_norm: Optional[float], guard: Optional[torch.Tensor] = None,
             guard_kind: int = 0, guard_host: Optional[torch.Tensor] = None) -> None:
        """`guard` (a device scalar): the update is skipped on the device when it is NaN
        (guard_kind 1) or non-zero (guard_kind 2) - `tv_optim_step_guarded`; `guard_host`
        (one pinned host float) receives the scalar from the kernel."""
        n = self.end - self.start
        p = self.flat.flat[self.start:self.end]
        g = self.flat.grad[self.start:self.end]
        norm_ptr = None
        if global_clip_norm:
            hip.call('tv_sumsq_partial', n, hip.ptr(g), None, hip.ptr(self._partial))
            hip.call('tv_sum_partials', self._partial.shape[0], hip.ptr(self._partial), 1,
                     hip.ptr(self.grad_norm))
            norm_ptr = hip.ptr(self.grad_norm)
        self.steps += 1
        if self.kind == 'rmsprop':
            kind, b1, b2 = 0, 0.0, self.alpha
        else:
            kind, (b1, b2) = 1, self.betas
        hip.call('tv_optim_step_guarded', kind, n, hip.ptr(p), hip.ptr(g), hip.ptr(self.state1),
                 hip.ptr(self.state2), norm_ptr, float(global_clip_norm or 0.0), self.lr,
                 float(b1), float(b2), float(self.eps), self.steps,
                 hip.ptr(guard), guard_kind if guard is not None else 0,
                 guard_host.data_ptr() if guard_host is not None else None)


def compute_train_loss(chain, train_cfg, l2_value: Optional[torch.Tensor] = None,
                       epoch: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """The loss of `train_step` (train.py:229-270) given the chain `q(G).chain(p, G=G)`:
    recons, kl, beta * kl - recons + L2 + the unit-ball regularisers (train.py:259-270 with
    get_moments, tensor_utils.py:117-125) - one kernel each way (`csrc/loss.hip`)."""
    p_obs, p_latent, q_latent = [], [], []
    for name in chain.p:
        if name in chain.q:
            q_latent.append(chain.q[name].log_prob())
            p_latent.append(chain.p[name].log_prob())
        else:
            p_obs.append(chain.p[name].log_prob())
    beta = train_cfg.kl_beta
    if train_cfg.warm_up_epochs and epoch is not None and epoch < train_cfg.warm_up_epochs:
        beta = beta * (epoch / train_cfg.warm_up_epochs)
    z = chain.q['z'].tensor
    if z.dim() != 2:
        raise ValueError('the training loss takes one sample per trace (q(G) without n_z): z is '
                         f'{tuple(z.shape)}')
    latents = []
    if train_cfg.z_unit_ball_reg is not None:
        latents.append((z, train_cfg.z_unit_ball_reg))
    if 'z2' in chain.q and train_cfg.z2_unit_ball_reg is not None:
        latents.append((chain.q['z2'].tensor, train_cfg.z2_unit_ball_reg))
    use_l2 = bool(train_cfg.l2_reg) and l2_value is not None
    loss, recons, kl = ops.train_loss(
        p_obs, q_latent, p_latent, beta, l2_value.reshape(1) if use_l2 else None,
        train_cfg.l2_reg if use_l2 else 0.0, latents)
    return {'loss': loss, 'recons': recons, 'kl': kl}


class Trainer(object):
    """One rank of the (optionally data-parallel) training job."""

    def __init__(self, vae: TraceVAE, config: ExpConfig, latency_only: bool = False,
                 process_group=None, world_size: Optional[int] = None):
        self.vae, self.config = vae, config
        self.train_cfg = config.train
        self.flat = FlatParameters(vae)
        self.pg = process_group
        self.world_size = world_size if world_size is not None else (
            dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1)
        self.epoch = 1
        # run the collective of `exchange` even with ONE rank (a sum over one rank changes
        # nothing): lets a one-GPU box execute the RCCL code path itself - communicator bound to
        # the device, all-reduce of [gradient range | NaN flag] in device memory
        self.exchange_single_rank = False
        self._nan_pending, self._nan_host = None, None
        self.set_phase(latency_only)

    # -- phases ------------------------------------------------------------------------
    def set_phase(self, latency_only: bool) -> None:
        """Phase 1: every parameter; phase 2: `latency_vae.*` only (train.py:455-487)."""
        self.latency_only = latency_only
        f = self.flat
        part = [n for n in f.all_names if n.startswith('latency_vae.')] if latency_only \
            else list(f.all_names)
        self.part_names = part
        # parameters outside the phase are frozen: autograd then skips every backward kernel that
        # only feeds them (in phase 2 the whole struct-VAE backward), with the same update as
        # the reference, whose optimizer simply ignores those gradients (train.py:476-487)
        in_part = set(part)
        for n, p in f.params.items():
            p.requires_grad_(n in in_part)
        start = f.tail_start if latency_only else 0
        self.optimizer = FlatOptimizer(f, self.train_cfg.optimizer, start, f.total)
        self.optimizer.set_lr(learning_rate_for_epoch(self.train_cfg, self.epoch - 1))
        self.l2_names = l2_parameter_names(f.all_names, part)
        self.l2_mask = f.mask(self.l2_names)
        self._l2_partial = torch.empty(hip.query('tv_reduce_blocks', f.total), dtype=torch.float64,
                                       device=f.flat.device)
        self._l2_value = torch.zeros(1, dtype=torch.float32, device=f.flat.device)
        self.apply_mode()

    def apply_mode(self) -> None:
        """on_train_epoch_begin (train.py:202-208)."""
        if self.latency_only:
            self.vae.eval()
            self.vae.latency_vae.train()
        else:
            self.vae.train()

    def set_epoch(self, epoch: int) -> None:
        """Epoch `epoch` (1-based) starts.  The reference sets the learning rate when the
        previous epoch ends, from the count of finished epochs (train.py:374-383): epoch e
        runs at initial_lr * ratio ** ((e - 1) // anneal_epochs)."""
        self.epoch = epoch
        self.optimizer.set_lr(learning_rate_for_epoch(self.train_cfg, epoch - 1))

    # -- one step ----------------------------------------------------------------------
    def l2_value(self) -> torch.Tensor:
        f = self.flat
        hip.call('tv_sumsq_partial', f.total, hip.ptr(f.flat), hip.ptr(self.l2_mask),
                 hip.ptr(self._l2_partial))
        hip.call('tv_sum_partials', self._l2_partial.shape[0], hip.ptr(self._l2_partial), 0,
                 hip.ptr(self._l2_value))
        return self._l2_value[0]

    def forward_loss(self, G: TraceGraphBatch, eps=None) -> Dict[str, torch.Tensor]:
        chain = self.vae.q(G, eps=eps).chain(self.vae.p, G=G)
        l2 = self.l2_value() if self.train_cfg.l2_reg else None
        out = compute_train_loss(chain, self.train_cfg, l2, self.epoch)
        out['chain'] = chain
        return out

    def local_gradients(self, G: TraceGraphBatch, eps=None) -> Dict[str, torch.Tensor]:
        """Forward + backward of this rank's mini-batch; the gradient of the loss (L2 term
        included) ends up in the flat bucket."""
        f = self.flat
        f.begin_backward()
        out = self.forward_loss(G, eps)
        out['loss'].backward()
        f.collect_grads()
        if self.train_cfg.l2_reg:
            # d(l2_reg * sum p^2)/dp = 2 * l2_reg * p on the masked elements
            hip.call('tv_axpy_masked', f.total, 2.0 * self.train_cfg.l2_reg, hip.ptr(f.flat),
                     hip.ptr(self.l2_mask), hip.ptr(f.grad))
        return out

    def update(self, guard: Optional[torch.Tensor] = None, guard_kind: int = 0,
               guard_host: Optional[torch.Tensor] = None) -> None:
        """Clip by global norm + RMSprop / Adam over the trained range of the flat buffer."""
        if guard is None:
            self.optimizer.step(self.train_cfg.global_clip_norm)
        else:
            self.optimizer.step(self.train_cfg.global_clip_norm, guard, guard_kind, guard_host)

    def step(self, G: TraceGraphBatch, eps=None, check_nan=True) -> Dict[str, torch.Tensor]:
        """One training step.  With several ranks every rank makes exactly the same sequence of
        collectives (one all-reduce per step) and takes the same NaN decision, so a NaN on one
        rank raises `NANLossError` on all of them instead of leaving the others blocked.

        The reference reads the loss back in every step and raises before `optimizer.step()`
        when it is NaN (train.py:272-274).  `check_nan=True` gives the same outcome without
        making the host wait for the GPU in every step: the update is guarded by the loss (or
        the all-reduced NaN count) ON THE DEVICE - a NaN step changes neither parameters nor
        optimizer state -, the optimizer kernel stores the scalar to pinned host memory, and
        `NANLossError` is raised by the next `step` - once its forward and backward are issued,
        so that the GPU has work while the host looks at the step before; ahead of its
        collective and its update - or by `finish()` (the epoch loop calls it when an epoch's
        steps are issued).  `check_nan='now'` waits for the loss inside the step, as the
        reference does (and as any step does whose buffers are in host memory);
        `check_nan=False` neither checks nor guards."""
        out = self.local_gradients(G, eps)
        if check_nan:
            self.finish()
        self.exchange(out['loss'])
        if check_nan is True and self.flat.flat.is_cuda:
            guard, kind = self._nan_guard(out['loss'])
            if self._nan_host is None:
                self._nan_host = torch.zeros(1, dtype=torch.float32, pin_memory=True)
                self._nan_value = self._nan_host.numpy()
                self._nan_event = torch.cuda.Event()
            self.update(guard, kind, self._nan_host)     # the kernel stores the scalar there
            self._nan_event.record()
            self._nan_pending = (kind, self.epoch)
        else:
            if check_nan and self.loss_is_nan(out['loss']):     # host sync, like train.py:273
                raise NANLossError(self.epoch)
            self.update()
        return {k: v.detach() for k, v in out.items() if k != 'chain'}

    def _nan_guard(self, loss: torch.Tensor):
        if self.world_size > 1 or self.exchange_single_rank:
            f = self.flat
            return f.bucket[f.flag_index:f.flag_index + 1], 2    # ranks whose loss was NaN
        return loss.detach().reshape(1), 1

    def finish(self) -> None:
        """Take the NaN decision of the last step issued with `check_nan=True`: waits for that
        step's loss (long known when the next step begins) and raises `NANLossError` with the
        epoch of the step."""
        pending = self._nan_pending
        if pending is None:
            return
        self._nan_pending = None
        self._nan_event.synchronize()
        kind, epoch = pending
        v = float(self._nan_value[0])
        if math.isnan(v) if kind == 1 else v != 0.0:
            raise NANLossError(epoch)

    def loss_is_nan(self, loss: torch.Tensor) -> bool:
        if self.world_size > 1 or self.exchange_single_rank:
            # the all-reduced flag: the number of ranks whose loss was NaN
            return float(self.flat.bucket[self.flat.flag_index].item()) != 0.0
        return math.isnan(float(loss.item()))

    def exchange(self, loss: Optional[torch.Tensor] = None) -> None:
        """One collective over [gradients of the trained range | NaN flag]: sum over ranks,
        gradients divided by the world size."""
        if self.world_size <= 1 and not self.exchange_single_rank:
            return
        f = self.flat
        flag = f.bucket[f.flag_index:f.flag_index + 1]
        if loss is None:
            flag.zero_()
        else:
            flag.copy_(torch.isnan(loss.detach()).reshape(1))
        g = f.bucket[self.optimizer.start:f.flag_index + 1]
        all_reduce_sum(g, self.pg)
        # SUM then scale (one 25 us pass over the bucket) instead of ReduceOp.AVG: the sum is
        # the one reduction every RCCL build has
        f.grad[self.optimizer.start:self.optimizer.end].mul_(1.0 / self.world_size)

    def all_reduce_gradients(self) -> None:
        self.exchange(None)


def _uses_device_collectives(group) -> bool:
    return dist.get_backend(group) == 'nccl'      # RCCL over xGMI


def all_reduce_sum(t: torch.Tensor, group=None) -> None:
    """In-place sum over the ranks.  RCCL reduces device memory directly; under gloo (CPU
    tests, rehearsals with more ranks than GPUs) device tensors are staged through the host:
    gloo's device path is not dependable on ROCm."""
    if _uses_device_collectives(group) or not t.is_cuda:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    h = t.detach().cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    t.copy_(h)


def broadcast_from_rank0(t: torch.Tensor, group=None) -> None:
    if _uses_device_collectives(group) or not t.is_cuda:
        dist.broadcast(t, 0, group=group)
        return
    h = t.detach().cpu()
    dist.broadcast(h, 0, group=group)
    t.copy_(h)


def _float_buffers(module: torch.nn.Module) -> List[torch.Tensor]:
    seen, out = set(), []
    for b in module.buffers():
        if b.dtype.is_floating_point and b.data_ptr() not in seen:
            seen.add(b.data_ptr())
            out.append(b)
    return out


def broadcast_model(trainer: 'Trainer', group=None) -> None:
    """Rank 0's parameters (one flat buffer) and floating-point buffers to every rank."""
    broadcast_from_rank0(trainer.flat.flat, group)
    bufs = _float_buffers(trainer.vae)
    if bufs:
        packed = torch.cat([b.detach().reshape(-1).float() for b in bufs])
        broadcast_from_rank0(packed, group)
        off = 0
        for b in bufs:
            b.copy_(packed[off:off + b.numel()].view(b.shape))
            off += b.numel()


def average_batch_norm_buffers(module: torch.nn.Module, world_size: int, group=None) -> None:
    """BatchNorm batch statistics are per rank during training (no SyncBN); before a model is
    scored or saved the running means / variances are averaged over the ranks, so that every
    rank scores with - and rank 0 saves - the same buffers.  One small all-reduce."""
    if world_size <= 1:
        return
    bufs = _float_buffers(module)
    if not bufs:
        return
    packed = torch.cat([b.detach().reshape(-1).float() for b in bufs])
    all_reduce_sum(packed, group)
    packed.mul_(1.0 / world_size)
    off = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(packed[off:off + b.numel()].view(b.shape))
            off += b.numel()


def save_model(vae: TraceVAE, config: ExpConfig, out_dir: str, name: str = 'final') -> str:
    """`results/train/config.json` + `results/train/models/{name}.pt` (train.py:343-352,
    README.md:15, test_utils.py:22-33)."""
    os.makedirs(os.path.join(out_dir, 'models'), exist_ok=True)
    with open(os.path.join(out_dir, 'config.json'), 'w', encoding='utf-8') as f:
        json.dump(config_to_dict(config), f, indent=2)
    path = os.path.join(out_dir, 'models', f'{name}.pt')
    torch.save({k: v.detach().cpu().clone() for k, v in vae.state_dict().items()}, path)
    return path


class StoreStream(object):
    """Mini-batches of trace indices over a `DeviceTraceStore` (the role of
    `TraceGraphDataStream`, dataset.py:92-133); rank r of W takes batches r, r+W, ...

    `equal_steps` (the training stream of a data-parallel job): every rank must run the same
    number of steps per epoch, because every step is one all-reduce.  The tail rule: an epoch
    uses the first `total - total % W` batches of its permutation, the up to W-1 batches left
    over are dropped for that epoch (the next epoch's permutation picks them up again)."""

    def __init__(self, store: DeviceTraceStore, batch_size: int, shuffle: bool = False,
                 seed: int = 0, rank: int = 0, world_size: int = 1,
                 max_node_count: Optional[int] = None, skip_incomplete: bool = False,
                 equal_steps: bool = False):
        self.store, self.batch_size, self.shuffle = store, batch_size, shuffle
        self.rng = np.random.RandomState(seed)
        self.rank, self.world_size = rank, world_size
        self.max_node_count = max_node_count
        self.skip_incomplete = skip_incomplete
        self.equal_steps = equal_steps

    @property
    def total_batches(self) -> int:
        """Batches per epoch over all ranks."""
        n = len(self.store)
        total = n // self.batch_size if self.skip_incomplete else -(-n // self.batch_size)
        if self.equal_steps:
            total -= total % self.world_size
        return total

    @property
    def batch_count(self) -> int:
        return len(range(self.rank, self.total_batches, self.world_size))

    def __iter__(self):
        n = len(self.store)
        order = self.rng.permutation(n) if self.shuffle else np.arange(n)
        for i in range(self.rank, self.total_batches, self.world_size):
            ids = order[i * self.batch_size:(i + 1) * self.batch_size]
            G = TraceGraphBatch.from_store(self.store, ids, self.max_node_count)
            G._stream_index = i
            yield G


# ---------------------------------------------------------------------------------------
# the training job (train.py:100-500 without the mltk experiment / TensorBoard plumbing)
# ---------------------------------------------------------------------------------------
def load_model(model_path: str, num_operations: int, extra_args=(), device=None):
    """`load_config` + `load_model2` (test_utils.py:22-100): `model_path` is the train output
    directory (`config.json`, `models/final.pt`) or one `models/<name>.pt`."""
    if model_path.endswith('.pt'):
        model_file = model_path
        config_file = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(model_path))),
                                   'config.json')
    else:
        model_file = os.path.join(model_path, 'models', 'final.pt')
        config_file = os.path.join(model_path, 'config.json')
    from .config import config_from_dict, parse_flags
    with open(config_file, 'r', encoding='utf-8') as f:
        config = config_from_dict(ExpConfig(), json.load(f))
    if extra_args:
        parse_flags(config, list(extra_args), strict=False)
    vae = TraceVAE(config.model, num_operations)
    vae.load_state_dict(torch.load(model_file, map_location='cpu', weights_only=True))
    if device is not None:
        vae = vae.to(device)
    return vae, config


def run_training(trainer: 'Trainer', train_stream: Iterable[TraceGraphBatch], train_cfg,
                 pretrain_epochs: Optional[int] = None, epoch_end=None, log=print) -> None:
    """The epoch loop of train.py:455-487: epochs 1..pretrain_epochs move every parameter, the
    rest only `latency_vae.*`; the learning rate is set per epoch; `epoch_end(epoch, mean_loss,
    seconds)` runs after every epoch on every rank.  Under data parallelism `train_stream` must
    give every rank the same number of batches (`StoreStream(equal_steps=True)`): each step
    is one collective."""
    for epoch in range(1, train_cfg.max_epoch + 1):
        if pretrain_epochs and epoch == pretrain_epochs + 1:
            trainer.epoch = epoch
            trainer.set_phase(latency_only=True)
            log(f'Start to train latency_vae with {len(trainer.part_names)} params ...')
        trainer.set_epoch(epoch)
        trainer.apply_mode()
        t0, losses = time.perf_counter(), []
        for G in train_stream:
            losses.append(trainer.step(G)['loss'])
        trainer.finish()                                 # the NaN decision of the last step
        mean_loss = float(torch.stack(losses).mean()) if losses else float('nan')
        if epoch_end is not None:
            epoch_end(epoch, mean_loss, time.perf_counter() - t0)


def fit(config: ExpConfig, stores: Dict[str, DeviceTraceStore], num_operations: int,
        out_dir: str, rank: int = 0, world_size: int = 1, process_group=None,
        log=print) -> Dict[str, float]:
    """The two-phase training job: epochs 1..struct_pretrain_epochs move every parameter,
    the rest only `latency_vae.*` with everything else in eval mode (train.py:455-487); the
    learning rate is annealed per epoch; validation, periodic test scoring and a checkpoint
    after every epoch; final K-sample scoring of the test sets and `models/final.pt`.
    `stores`: 'train', 'val', 'test' (the three test sets concatenated).

    With world_size > 1 (one process per GPU): rank r trains on batches r, r+W, ... of every
    epoch's permutation (equal step counts, see `StoreStream`); weights - including the
    data-dependent ActNorm initialisation - come from rank 0; every rank draws its own noise
    (seed + rank); BatchNorm running statistics are averaged over the ranks at the end of
    every epoch, before validation / test scoring / saving; validation and test traces are
    sharded over the ranks and the results gathered; rank 0 writes the files."""
    from .evaluation import do_evaluate_nll
    tc = config.train
    if config.seed is not None:
        torch.manual_seed(config.seed)
        np.random.seed(config.seed)
    device = stores['train'].device
    vae = TraceVAE(config.model, num_operations).to(device)
    trainer = Trainer(vae, config, world_size=world_size, process_group=process_group)
    train_stream = StoreStream(stores['train'], tc.batch_size, shuffle=True,
                               seed=config.seed or 0, rank=rank, world_size=world_size,
                               equal_steps=True)
    val_stream = StoreStream(stores['val'], tc.val_batch_size, rank=rank, world_size=world_size)
    test_stream = StoreStream(stores['test'], config.test.batch_size, rank=rank,
                              world_size=world_size)
    result: Dict[str, float] = {}

    def evaluate(n_z, tag):
        r = do_evaluate_nll(test_stream, vae, n_z, use_biased=config.test.use_biased,
                            latency_log_prob_weight=config.test.latency_log_prob_weight,
                            clip_nll=config.test.clip_nll, process_group=process_group)
        result['_nll_list'], result['_label_list'] = r['_nll_list'], r['_label_list']
        r = {k: v for k, v in r.items() if not k.startswith('_')}
        if rank == 0:
            path = os.path.join(out_dir, 'result', 'test-anomaly', f'{tag}.json')
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, 'w', encoding='utf-8') as f:
                f.write(json.dumps(r))
        return r

    def validation_loss() -> float:
        vae.eval()
        with torch.no_grad():
            parts = [(vae.q(G).chain(vae.p, G=G).vi.training.sgvb().mean(), len(G))
                     for G in val_stream]
        # batch-size weighted mean over the batches of all ranks
        acc = torch.zeros(2, dtype=torch.float64, device=device)
        for v, n in parts:
            acc[0] += v.double() * n
            acc[1] += n
        if world_size > 1:
            all_reduce_sum(acc, process_group)
        return float(acc[0] / acc[1]) if float(acc[1]) > 0 else float('nan')

    # data-dependent initialisation on one batch (train.py:188-199), in the training mode
    trainer.apply_mode()
    rng = np.random.RandomState(config.seed or 0)
    init_ids = rng.choice(len(stores['train']), size=min(tc.init_batch_size, len(stores['train'])),
                          replace=False)
    with torch.no_grad():
        G = TraceGraphBatch.from_store(stores['train'], init_ids)
        loss = vae.q(G).chain(vae.p, G=G).vi.training.sgvb(reduction='mean')
    log(f'Network initialized: loss = {float(loss):.6g}')
    if world_size > 1:
        # every rank continues from rank 0's weights and buffers (the ActNorm initialisation
        # above included), and draws its own noise from here on
        broadcast_model(trainer, process_group)
        if config.seed is not None:
            torch.manual_seed(config.seed + rank)

    # train.py:443-448: `mltk.callbacks.EarlyStopping(checkpoint, metric_name='val_loss')` keeps
    # the state of the epoch with the smallest validation loss and restores it when the train
    # loop ends, before the final model is saved and scored (it does not cut the run short: no
    # patience is given).  Active only together with `val_epoch_freq`, like the reference.
    # mltk is not in the container: semantics from its API, parity unpinned.  The validation
    # loss is the all-reduced one, so every rank keeps and restores the same epoch.
    early = {'best': math.inf, 'epoch': None, 'state': None} \
        if (tc.val_epoch_freq and tc.use_early_stopping) else None

    def epoch_end(epoch, mean_loss, seconds):
        average_batch_norm_buffers(vae, world_size, process_group)
        msg = f'epoch {epoch}/{tc.max_epoch}: loss {mean_loss:.6g} ({seconds:.2f}s)'
        if tc.val_epoch_freq and epoch % tc.val_epoch_freq == 0:
            result['val_loss'] = validation_loss()
            msg += f' val_loss {result["val_loss"]:.6g}'
            if early is not None and result['val_loss'] < early['best']:
                early.update(best=result['val_loss'], epoch=epoch,
                             state={k: v.detach().clone() for k, v in vae.state_dict().items()})
                if rank == 0:
                    os.makedirs(os.path.join(out_dir, 'early-stopping'), exist_ok=True)
                    torch.save({k: v.cpu() for k, v in early['state'].items()},
                               os.path.join(out_dir, 'early-stopping', 'best.pt'))
                msg += ' (best so far)'
        log(msg)
        if tc.test_epoch_freq and epoch % tc.test_epoch_freq == 0:
            r = evaluate(tc.test_n_z, str(epoch))
            log(f'  test: auc {r.get("auc", float("nan")):.4f} best_fscore '
                f'{r.get("best_fscore", float("nan")):.4f}')
        if rank == 0:
            save_model(vae, config, out_dir, name=str(epoch))

    pretrain = tc.struct_pretrain_epochs if config.model.arch == 'default' else None
    run_training(trainer, train_stream, tc, pretrain, epoch_end, log)
    if early is not None and early['state'] is not None:
        log(f'early stopping: restoring the parameters of epoch {early["epoch"]} '
            f'(val_loss {early["best"]:.6g})')
        with torch.no_grad():
            vae.load_state_dict(early['state'])     # copies into the flat buffer's views
        result['early_stopping_epoch'] = early['epoch']
    vae.eval()
    if rank == 0:
        save_model(vae, config, out_dir, name='final')
    result.update(evaluate(config.test.eval_n_z, 'final'))
    result['_vae'] = vae
    return result


# directories of an experiment that a restart wipes (train.py:524-529; this package writes
# the first two, the others are the reference's and are cleared when present)
RESTART_WIPES = ('models', 'result', 'checkpoint', 'early-stopping', 'plotting', 'summary')
MAX_NAN_RESTARTS = 10


def run_with_nan_restart(job, out_dir: str, rank: int = 0, world_size: int = 1,
                         process_group=None, log=None, config: Optional[ExpConfig] = None):
    """The outer loop of train.py:510-531: when the loss turns NaN during epoch 1 the whole
    experiment starts over (a fresh model, up to 10 times), with the output directories of the
    failed try removed; a NaN in a later epoch, or an eleventh one, propagates.  `job()` runs
    one try.  Under data parallelism `NANLossError` is raised on every rank in the same step
    (`Trainer.step`), so all ranks restart together.

    The reference re-runs `main()` with the same `--seed`, and on its engine (non-deterministic
    atomics, a fresh CUDA RNG stream) the retry takes another trajectory.  Here every kernel
    sums in a fixed order, so a retry from the same seed would reach the same NaN ten times:
    with `config` given, try r runs at `seed + r` (the same on every rank).  `result/` is
    wiped too, which the reference leaves in place (train.py:525): a failed try's per-epoch
    test results would otherwise sit beside the successful run's."""
    import shutil
    retrial = 0
    base_seed = config.seed if config is not None else None
    while True:
        try:
            return job()
        except NANLossError as ex:
            if ex.epoch != 1 or retrial >= MAX_NAN_RESTARTS:
                raise
            retrial += 1
            (log or (lambda m: print(m, file=sys.stderr)))(
                f'\nRestart the experiment for the {retrial}-th time due to NaN loss at epoch '
                f'{ex.epoch}.\n')
            if rank == 0:
                for name in RESTART_WIPES:
                    path = os.path.join(out_dir, name)
                    if os.path.isdir(path):
                        shutil.rmtree(path)
            if base_seed is not None:
                config.seed = base_seed + retrial
            if world_size > 1:
                dist.barrier(group=process_group)


def init_distributed() -> Tuple[int, int]:
    """(rank, world size) from the torch.distributed.run environment; one process per GPU over
    RCCL (`TRACEVAE_DIST_BACKEND=gloo` rehearses the control flow on a box with fewer GPUs than
    ranks)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        index = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
        torch.cuda.set_device(index)
        backend = os.environ.get('TRACEVAE_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            # device_id binds the communicator to this rank's GPU up front
            dist.init_process_group('nccl', device_id=torch.device('cuda', index))
        else:
            dist.init_process_group(backend)
    return rank, world


def main(argv=None) -> int:
    """`python -m tracevae_amd.train --dataset.root_dir=<dir> [--a.b.c=value ...]`: the flags of
    train.sh.  Output under ./results/train (config.json, models/<epoch>.pt, models/final.pt,
    result/test-anomaly/*.json), like the reference's experiment directory."""
    from .config import parse_flags
    from .data_process import TEST_SETS, concat_flat, load_dataset
    argv = list(sys.argv[1:] if argv is None else argv)
    out_dir = 'results/train'
    for a in list(argv):
        if a.startswith('--output-dir='):
            out_dir = a.split('=', 1)[1]
            argv.remove(a)
    config = ExpConfig()
    parse_flags(config, argv)
    if not torch.cuda.is_available():
        raise hip.HipLibraryError('the accelerated path needs a GPU (there is no CPU fallback); '
                                  f'--device={config.device} cannot be honoured')
    rank, world = init_distributed()
    if config.device in (None, 'cpu') and rank == 0:
        print(f'note: --device={config.device} ignored, running on cuda:{torch.cuda.current_device()}')
    sets, id_manager, latency_range = load_dataset(config.dataset.root_dir)
    v = id_manager.num_operations
    stores = {'train': DeviceTraceStore(sets['train'], latency_range, v),
              'val': DeviceTraceStore(sets['val'], latency_range, v),
              'test': DeviceTraceStore(concat_flat(sets[n] for n in TEST_SETS), latency_range, v)}
    result = run_with_nan_restart(
        lambda: fit(config, stores, v, out_dir, rank=rank, world_size=world), out_dir, rank, world,
        config=config)
    if rank == 0:
        print(json.dumps({k: val for k, val in result.items() if not k.startswith('_')}, indent=1))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
