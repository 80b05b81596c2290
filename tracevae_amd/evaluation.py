"""K-sample NLL anomaly scoring (the `eval_step` of `do_evaluate_nll`,
`tracegnn/models/trace_vae/evaluation.py:115-148, 223`) on the accelerated path.

Per mini-batch: encoders once, K reparameterised draws of (z, z2), K decoder passes run as
one batch of K*B*M rows over the block-diagonally replicated decoder graph (legal because
eval-mode BatchNorm is per row), then `tv_score_reduce`: sgvb loss and
`nll = -log mean_k exp(log p(x,z_k) - log q(z_k|x))`, both clipped at `clip_nll`
(NaN -> clip).  Scoring shards by trace with no communication.
"""
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from .batch import TraceGraphBatch
from .model import TraceVAE

__all__ = ['eval_step', 'do_evaluate_nll', 'analyze_anomaly_nll']


@torch.no_grad()
def eval_step(vae: TraceVAE, G: TraceGraphBatch, n_z: int, *, use_biased: bool = True,
              use_latency_biased: bool = True, no_latency: bool = False,
              std_limit: Optional[torch.Tensor] = None, latency_log_prob_weight: bool = False,
              latency_logstd_min: Optional[float] = None, clip_nll: Optional[float] = None,
              eps=None, return_chain: bool = False):
    """Returns `(loss [B], nll [B])` as device tensors (and the chain when asked)."""
    chain = vae.q(G, n_z=n_z, no_latency=no_latency, eps=eps).chain(
        vae.p, latent_axis=0, G=G, use_biased=use_biased, use_latency_biased=use_latency_biased,
        no_latency=no_latency, latency_logstd_min=latency_logstd_min,
        latency_log_prob_weight=latency_log_prob_weight, std_limit=std_limit)
    loss, nll = chain.score(clip_nll)
    if return_chain:
        return loss, nll, chain
    return loss, nll


def analyze_anomaly_nll(nll_list: np.ndarray, label_list: np.ndarray) -> Dict[str, float]:
    """AUC / best-F1 of the scores per anomaly class (utils/analyze_nll.py:62-110,
    utils/fscore_utils.py:28-37); classes that are absent are skipped."""
    from sklearn.metrics import average_precision_score, precision_recall_curve, roc_auc_score
    out: Dict[str, float] = {}
    nll_list = np.asarray(nll_list, dtype=np.float64)
    label_list = np.asarray(label_list)
    normal = nll_list[label_list == 0]
    if len(normal):
        out['nll_normal'] = float(np.mean(normal))
    for tag, sel in (('', label_list != 0), ('_drop', label_list == 1), ('_latency', label_list == 2)):
        pos = nll_list[sel]
        if len(pos) == 0 or len(normal) == 0:
            continue
        if tag:
            out[f'nll{tag}'] = float(np.mean(pos))
        y = np.concatenate([np.zeros(len(normal)), np.ones(len(pos))])
        s = np.concatenate([normal, pos])
        out[f'auc{tag}'] = float(roc_auc_score(y, s))
        out[f'ap{tag}'] = float(average_precision_score(y, s))
        precision, recall, thresholds = precision_recall_curve(y, s)
        f1 = 2 * precision * recall / np.maximum(precision + recall, 1e-12)
        best = int(np.argmax(f1[:-1])) if len(f1) > 1 else 0
        out[f'best_fscore{tag}'] = float(f1[best])
        out[f'best_threshold{tag}'] = float(thresholds[best]) if len(thresholds) else float('nan')
    return out


def do_evaluate_nll(test_stream: Iterable[TraceGraphBatch], vae: TraceVAE, n_z: int,
                    use_biased: bool = True, use_latency_biased: bool = True,
                    no_latency: bool = False, std_limit=None, latency_log_prob_weight: bool = False,
                    latency_logstd_min: Optional[float] = None, clip_nll: Optional[float] = None,
                    nll_output_file: Optional[str] = None, labels: Optional[np.ndarray] = None,
                    **_ignored) -> Dict[str, float]:
    """The scoring driver: score every batch of the stream, keep the scores on the GPU until
    the end (one device->host copy), then the AUC / F1 summary and the optional `.npz`
    (evaluation.py:249-283).  The plotting / collector options of the reference are
    diagnostics and not part of this path; they are accepted and ignored."""
    was_training = vae.training
    vae.eval()
    nll_chunks: List[torch.Tensor] = []
    loss_chunks: List[torch.Tensor] = []
    label_chunks: List[np.ndarray] = []
    try:
        for G in test_stream:
            loss, nll = eval_step(
                vae, G, n_z, use_biased=use_biased, use_latency_biased=use_latency_biased,
                no_latency=no_latency, std_limit=std_limit,
                latency_log_prob_weight=latency_log_prob_weight,
                latency_logstd_min=latency_logstd_min, clip_nll=clip_nll)
            nll_chunks.append(nll)
            loss_chunks.append(loss)
            if labels is None and G.labels is not None:
                label_chunks.append(G.labels)
    finally:
        vae.train(was_training)
    nll_list = torch.cat(nll_chunks).cpu().numpy() if nll_chunks else np.zeros([0], np.float32)
    loss_list = torch.cat(loss_chunks).cpu().numpy() if loss_chunks else np.zeros([0], np.float32)
    if labels is None:
        labels = np.concatenate(label_chunks) if label_chunks else np.zeros(len(nll_list), np.int64)
    result: Dict[str, float] = {}
    normal_losses = loss_list[labels == 0]
    if len(normal_losses):
        result['test_loss'] = float(np.nanmean(normal_losses))
    result.update(analyze_anomaly_nll(nll_list, labels))
    if nll_output_file is not None:
        np.savez_compressed(nll_output_file, nll_list=nll_list, label_list=labels)
    result['_nll_list'] = nll_list
    result['_label_list'] = labels
    return result
