"""Light stand-ins for the tensorkit objects that the reference's drivers touch
(`tk.BayesianNet`, `StochasticTensor`, `VariationalChain`).

The model computes every log-likelihood with fused HIP kernels while it builds the net, so
a node here simply carries its tensor, its (already computed) log-probability and the
distribution parameters that the reference's collectors read
(`chain.p['latency'].distribution.base_distribution.std`, `.logits`, `.probs`, ...).

Same surface as the reference uses:
  net[name] / name in net / iteration in insertion order   (train.py:232-238)
  net[name].tensor, .log_prob(), .distribution             (tensor_utils.py:139-225)
  q_net.chain(p_builder, latent_axis=..., **kw)            (train.py:219-222, evaluation.py:121-131)
  chain.q, chain.p, chain.vi.training.sgvb(reduction=), chain.vi.evaluation.is_loglikelihood()
"""
import math
from collections import OrderedDict
from typing import Callable, Dict, Iterator, Optional

import torch

from . import ops

__all__ = ['Distribution', 'StochasticTensor', 'BayesianNet', 'VariationalChain']


class Distribution(object):
    """Parameter holder; `kind` names the reference distribution class it stands for."""

    def __init__(self, kind: str, base_distribution: Optional['Distribution'] = None, **params):
        self.kind = kind
        self.base_distribution = base_distribution
        self._params = params

    def __getattr__(self, item):
        params = self.__dict__.get('_params', {})
        if item in params:
            v = params[item]
            return v() if callable(v) else v
        raise AttributeError(f'{self.kind} distribution has no attribute {item!r}')

    def __repr__(self):
        return f'Distribution({self.kind})'


class StochasticTensor(object):
    def __init__(self, tensor: torch.Tensor, distribution: Distribution,
                 log_prob: Optional[torch.Tensor], n_samples: Optional[int] = None):
        self.tensor = tensor
        self.distribution = distribution
        self.n_samples = n_samples
        self._log_prob = log_prob

    def log_prob(self) -> torch.Tensor:
        if self._log_prob is None:
            raise RuntimeError('this node carries no log-probability')
        return self._log_prob


class BayesianNet(object):
    def __init__(self, observed: Optional[Dict[str, torch.Tensor]] = None):
        self.observed = dict(observed or {})
        self._nodes: 'OrderedDict[str, StochasticTensor]' = OrderedDict()
        self.meta: Dict[str, object] = {}

    def add_node(self, name: str, node: StochasticTensor) -> StochasticTensor:
        if name in self._nodes:
            raise ValueError(f'node {name!r} already exists')
        self._nodes[name] = node
        return node

    def __getitem__(self, name: str) -> StochasticTensor:
        return self._nodes[name]

    def __contains__(self, name: str) -> bool:
        return name in self._nodes

    def __iter__(self) -> Iterator[str]:
        return iter(self._nodes)

    def __len__(self):
        return len(self._nodes)

    def __delitem__(self, name: str):
        del self._nodes[name]

    def log_prob(self, name: str) -> torch.Tensor:
        return self._nodes[name].log_prob()

    def chain(self, p_builder: Callable[..., 'BayesianNet'], latent_axis: Optional[int] = None,
              observed: Optional[Dict[str, torch.Tensor]] = None, **kwargs) -> 'VariationalChain':
        obs = dict(observed or {})
        for name in self._nodes:
            obs[name] = self._nodes[name].tensor
        p_net = p_builder(observed=obs, **kwargs)
        return VariationalChain(self, p_net, latent_axis)


class _Training(object):
    def __init__(self, chain):
        self._chain = chain

    def sgvb(self, reduction: str = 'none') -> torch.Tensor:
        c = self._chain
        w = c.log_joint() - c.latent_log_joint()
        loss = -w
        if c.latent_axis is not None:
            loss = loss.mean(dim=c.latent_axis)
        return _reduce(loss, reduction)


class _Evaluation(object):
    def __init__(self, chain):
        self._chain = chain

    def is_loglikelihood(self, reduction: str = 'none') -> torch.Tensor:
        c = self._chain
        if c.latent_axis is None:
            raise ValueError('is_loglikelihood needs a latent (sample) axis')
        w = c.log_joint() - c.latent_log_joint()
        if c.latent_axis != 0:
            w = w.movedim(c.latent_axis, 0)
        k = w.shape[0]
        w2 = w.reshape(k, -1).contiguous()
        _loss, nll = ops.score_reduce(w2, None)
        return _reduce((-nll).reshape(w.shape[1:]), reduction)


class _VI(object):
    def __init__(self, chain):
        self.training = _Training(chain)
        self.evaluation = _Evaluation(chain)


class VariationalChain(object):
    def __init__(self, q: BayesianNet, p: BayesianNet, latent_axis: Optional[int]):
        self.q, self.p, self.latent_axis = q, p, latent_axis

    @property
    def vi(self) -> _VI:
        # made on access, not kept: a stored `_VI(self)` is a reference cycle chain <-> vi, and
        # everything a step's chain reaches (the nets, their nodes, the logits of every
        # likelihood: 26 MB at batch 64, 200 MB at 512) would then wait for Python's cyclic
        # collector instead of being freed when the step lets go of the chain
        return _VI(self)

    def log_joint(self) -> torch.Tensor:
        return _add_n([self.p[name].log_prob() for name in self.p])

    def latent_log_joint(self) -> torch.Tensor:
        return _add_n([self.q[name].log_prob() for name in self.q])

    def score(self, clip: Optional[float]):
        """`(sgvb loss [B], nll [B])` of evaluation.py:141-148 in one kernel."""
        if self.latent_axis != 0:
            raise ValueError('score() expects the sample axis first')
        w = (self.log_joint() - self.latent_log_joint()).contiguous()
        return ops.score_reduce(w, clip)


def _add_n(ts):
    out = ts[0]
    for t in ts[1:]:
        out = out + t
    return out


def _reduce(x, reduction):
    if reduction == 'none':
        return x
    if reduction == 'mean':
        return x.mean()
    if reduction == 'sum':
        return x.sum()
    raise ValueError(f'unknown reduction {reduction!r}')
