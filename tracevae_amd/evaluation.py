"""K-sample NLL anomaly scoring (the `eval_step` of `do_evaluate_nll`,
`tracegnn/models/trace_vae/evaluation.py:115-148, 223`) on the accelerated path.

Per mini-batch: encoders once, K reparameterised draws of (z, z2), K decoder passes run as
one batch of K*B*M rows over the block-diagonally replicated decoder graph (legal because
eval-mode BatchNorm is per row), then `tv_score_reduce`: sgvb loss and
`nll = -log mean_k exp(log p(x,z_k) - log q(z_k|x))`, both clipped at `clip_nll`
(NaN -> clip).  Scoring shards by trace with no communication.
"""
import math
import os
import warnings
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .batch import TraceGraphBatch
from .model import TraceVAE

__all__ = ['eval_step', 'do_evaluate_nll', 'do_evaluate_prior', 'analyze_anomaly_nll',
           'best_fscore', 'fscore_for_precision_and_recall', 'DeviceCollector']


@torch.no_grad()
def eval_step(vae: TraceVAE, G: TraceGraphBatch, n_z: int, *, use_biased: bool = True,
              use_latency_biased: bool = True, no_latency: bool = False,
              std_limit: Optional[torch.Tensor] = None, latency_log_prob_weight: bool = False,
              latency_logstd_min: Optional[float] = None, clip_nll: Optional[float] = None,
              eps=None, return_chain: bool = False, no_struct: bool = False):
    """Returns `(loss [B], nll [B])` as device tensors (and the chain when asked).
    `no_struct` scores the latency model alone: z and the structure terms are taken out of
    both nets before the reduction (evaluation.py:132-139)."""
    chain = vae.q(G, n_z=n_z, no_latency=no_latency, eps=eps).chain(
        vae.p, latent_axis=0, G=G, use_biased=use_biased, use_latency_biased=use_latency_biased,
        no_latency=no_latency, latency_logstd_min=latency_logstd_min,
        latency_log_prob_weight=latency_log_prob_weight, std_limit=std_limit)
    if no_struct:
        if no_latency:
            raise ValueError('no_struct and no_latency together leave nothing to score')
        del chain.q['z']
        for name in ('z', 'adj', 'node_count', 'node_type'):
            del chain.p[name]
    loss, nll = chain.score(clip_nll)
    if return_chain:
        return loss, nll, chain
    return loss, nll


def fscore_for_precision_and_recall(precision, recall) -> np.ndarray:
    """utils/fscore_utils.py:13-26 (log-domain harmonic mean, 0 where either is 0)."""
    precision = np.asarray(precision, dtype=np.float64)
    recall = np.asarray(recall, dtype=np.float64)
    return np.where(
        (precision == 0) | (recall == 0), 0.0,
        2. * np.exp(np.log(np.maximum(precision, 1e-8)) + np.log(np.maximum(recall, 1e-8)) -
                    np.log(np.maximum(precision + recall, 1e-8))))


def best_fscore(proba, truth):
    """(f-score, threshold, precision, recall) at the best point of the PR curve
    (utils/fscore_utils.py:29-34)."""
    from sklearn.metrics import precision_recall_curve
    precision, recall, threshold = precision_recall_curve(truth, proba)
    fscore = fscore_for_precision_and_recall(precision, recall)
    idx = int(np.argmax(fscore[:-1]))
    return fscore[idx], threshold[idx], precision[idx], recall[idx]


def analyze_anomaly_nll(nll_list: np.ndarray, label_list: np.ndarray,
                        threshold: Optional[float] = None) -> Dict[str, float]:
    """The score summary of `do_evaluate_nll` (utils/analyze_nll.py:62-110): mean NLL per class
    (0 normal, 1 structure anomaly "drop", 2 latency anomaly), `auc` (which the reference
    computes as AVERAGE PRECISION, fscore_utils.py:37-38), best F-score / precision / recall
    over all anomalies and per class (the other anomaly class removed), and the F-scores at
    `threshold` when one is given.  Pinned against the reference's own functions by
    tests/golden/analyze_nll_vectors.json.  A failing metric (an absent class) is NaN, as the
    reference's `log_error` wrapper makes it."""
    from sklearn.metrics import average_precision_score, f1_score
    nll_list = np.asarray(nll_list)
    label_list = np.asarray(label_list)
    is_anomaly = label_list != 0
    nan4 = (math.nan, math.nan, math.nan, math.nan)

    def guarded(fn, default, *args):
        try:
            with warnings.catch_warnings():      # sklearn warns about absent classes
                warnings.simplefilter('ignore')
                return fn(*args)
        except Exception:
            return default

    def mean_of(mask):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            return float(np.mean(nll_list[mask])) if len(nll_list) else math.nan

    out: Dict[str, float] = {
        'nll_normal': mean_of(label_list == 0),
        'nll_drop': mean_of(label_list == 1),
        'nll_latency': mean_of(label_list == 2),
    }
    out['auc'] = guarded(lambda: float(average_precision_score(is_anomaly, nll_list)), math.nan)

    def best_for_label(label):
        mask = label_list != (2 if label == 1 else 1)
        return guarded(best_fscore, nan4, nll_list[mask], label_list[mask] != 0)

    f_t, _, pr_t, rc_t = guarded(best_fscore, nan4, nll_list, is_anomaly)
    f_d, _, pr_d, rc_d = best_for_label(1)
    f_l, th_l, pr_l, rc_l = best_for_label(2)
    out.update({
        'best_fscore': float(f_t), 'best_fscore_drop': float(f_d), 'best_fscore_latency': float(f_l),
        'best_pr': float(pr_t), 'best_rc': float(rc_t),
        'best_pr_drop': float(pr_d), 'best_rc_drop': float(rc_d),
        'best_pr_latency': float(pr_l), 'best_rc_latency': float(rc_l),
        'best_threshold_latency': float(th_l),
    })
    if threshold is not None:
        def f_at(label):
            mask = label_list != (2 if label == 1 else 1)
            return guarded(f1_score, math.nan, label_list[mask] != 0, nll_list[mask] > threshold)
        out.update({
            'fscore': float(guarded(f1_score, math.nan, is_anomaly, nll_list > threshold)),
            'fscore_drop': float(f_at(1)), 'fscore_latency': float(f_at(2)),
        })
    return out


class DeviceCollector(object):
    """The `ArrayBuffer` collectors of the reference (evaluation.py:76-99, filled by
    tensor_utils.py:139-225), kept on the GPU: per category (`normal` / `drop` / `latency`, by
    trace label) the per-node values of the real nodes of sample 0.  `.array(key)` makes the
    one device->host copy."""

    CATEGORIES = ((0, 'normal'), (1, 'drop'), (2, 'latency'))

    def __init__(self, prefix: str = ''):
        self.prefix = prefix
        self.chunks: Dict[str, List[torch.Tensor]] = {}

    def add(self, values_bm: torch.Tensor, node_count: torch.Tensor, labels: np.ndarray) -> None:
        """values_bm [B, M]; rows of real nodes (slot < node_count) are kept, in trace order."""
        b, m = values_bm.shape
        real = torch.arange(m, device=values_bm.device)[None, :] < node_count[:, None]
        lab = torch.as_tensor(np.asarray(labels), device=values_bm.device)
        for code, name in self.CATEGORIES:
            sel = real & (lab == code)[:, None]
            self.chunks.setdefault(self.prefix + name, []).append(values_bm[sel])

    def add_selected(self, values: torch.Tensor, keep: torch.Tensor, labels: np.ndarray) -> None:
        """values / keep [B, ...]: the kept entries of every trace, in trace order, row-major
        within a trace."""
        lab = torch.as_tensor(np.asarray(labels), device=values.device)
        shape = (-1,) + (1,) * (values.dim() - 1)
        for code, name in self.CATEGORIES:
            sel = keep & (lab == code).reshape(shape)
            self.chunks.setdefault(self.prefix + name, []).append(values[sel])

    def array(self, key: str, group=None) -> np.ndarray:
        """All collected values of `key`; with a process group (sharded scoring) the shards of
        every rank, concatenated in rank order."""
        parts = self.chunks.get(key, [])
        local = torch.cat(parts).cpu().numpy() if parts else np.zeros([0], np.float32)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            shards = [None] * dist.get_world_size(group)
            dist.all_gather_object(shards, local, group=group)
            local = np.concatenate(shards)
        return local


def do_evaluate_nll(test_stream: Iterable[TraceGraphBatch], vae: TraceVAE, n_z: int,
                    use_biased: bool = True, use_latency_biased: bool = True,
                    no_latency: bool = False, no_struct: bool = False, std_limit=None,
                    latency_log_prob_weight: bool = False,
                    latency_logstd_min: Optional[float] = None, clip_nll: Optional[float] = None,
                    test_threshold: Optional[float] = None,
                    nll_output_file: Optional[str] = None, labels: Optional[np.ndarray] = None,
                    operation_id_dict_out: Optional[DeviceCollector] = None,
                    latency_std_dict_out: Optional[DeviceCollector] = None,
                    latency_reldiff_dict_out: Optional[DeviceCollector] = None,
                    p_node_count_dict_out: Optional[DeviceCollector] = None,
                    p_edge_dict_out: Optional[DeviceCollector] = None,
                    process_group=None, noise_seed: Optional[int] = None,
                    **_ignored) -> Dict[str, float]:
    """The scoring driver (evaluation.py:40-283): score every batch of the stream, keep scores
    and collected per-node values on the GPU until the end (one device->host copy each), then
    the summary of `analyze_anomaly_nll` and the optional `.npz` (`nll_list`, `label_list`,
    `anomaly_degree = nll / best_threshold_latency`, evaluation.py:276-283).  The three
    collectors feed test.py's std-limit / bias-std inference (operation id, predicted latency
    std and |latency - mean| / std of every real node, by trace label); `p_node_count_dict_out`
    gathers p(node_count = observed) and `p_edge_dict_out` the |(1 - adj) - p(edge)| entries of
    every trace's dense n x n block, both of sample 0 (tensor_utils.py:171-198).  The
    plotting / TensorBoard arguments of the reference are accepted and ignored.

    `noise_seed` (an extension): the N(0,1) draws of batch i of the stream come from a CPU
    generator seeded with (noise_seed, i), so the scores do not depend on how the stream is
    sharded over ranks; by default the device generator draws them."""
    was_training = vae.training
    vae.eval()
    nll_chunks: List[torch.Tensor] = []
    loss_chunks: List[torch.Tensor] = []
    weights: List[int] = []
    label_chunks: List[np.ndarray] = []
    # the reference skips every collector under no_struct (evaluation.py:161)
    collect_lat = any(c is not None for c in (operation_id_dict_out, latency_std_dict_out,
                                             latency_reldiff_dict_out)) and not no_latency \
        and not no_struct
    collect_struct = any(c is not None for c in (p_node_count_dict_out, p_edge_dict_out)) \
        and not no_struct
    collect = collect_lat or collect_struct
    offset = 0
    try:
        for step_index, G in enumerate(test_stream):
            eps = None
            if noise_seed is not None:
                index = getattr(G, '_stream_index', None)
                gen = torch.Generator().manual_seed(
                    int(noise_seed) * 1000003 + int(step_index if index is None else index))
                eps = {'z': torch.randn(n_z, len(G), vae.config.struct.z_dim, generator=gen)}
                if vae.config.use_latency and not no_latency:
                    eps['z2'] = torch.randn(n_z, len(G), vae.config.latency.z2_dim, generator=gen)
            res = eval_step(
                vae, G, n_z, use_biased=use_biased, use_latency_biased=use_latency_biased,
                no_latency=no_latency, std_limit=std_limit,
                latency_log_prob_weight=latency_log_prob_weight,
                latency_logstd_min=latency_logstd_min, clip_nll=clip_nll, return_chain=collect,
                no_struct=no_struct, eps=eps)
            loss, nll = res[0], res[1]
            nll_chunks.append(nll)
            loss_chunks.append(loss.mean().reshape(1))
            weights.append(len(nll))
            step_labels = labels[offset:offset + len(nll)] if labels is not None else G.labels
            if step_labels is None:
                step_labels = np.zeros(len(nll), np.int64)
            label_chunks.append(np.asarray(step_labels))
            offset += len(nll)
            if collect_struct:
                chain, batch = res[2], G.build_dgl()
                count = batch.node_count
                if p_node_count_dict_out is not None:
                    probs = chain.p['node_count'].distribution.probs[0]
                    picked = probs.gather(1, count[:, None])[:, 0]
                    p_node_count_dict_out.add_selected(picked, torch.ones_like(picked, dtype=torch.bool),
                                                       step_labels)
                if p_edge_dict_out is not None:
                    m = batch.max_node_count
                    iu = torch.triu_indices(m, m, 1, device=count.device)
                    # triu_to_dense pads with zeros: below the diagonal |(1 - 0) - 0| = 1
                    dense = torch.ones([len(count), m, m], device=count.device)
                    dense[:, iu[0], iu[1]] = \
                        ((1.0 - batch.adj) - chain.p['adj'].distribution.probs[0]).abs()
                    slot = torch.arange(m, device=count.device)
                    real = slot[None, :] < count[:, None]
                    p_edge_dict_out.add_selected(dense, real[:, :, None] & real[:, None, :],
                                                 step_labels)
            if collect_lat:
                chain, batch = res[2], G.build_dgl()
                count = batch.node_count
                if operation_id_dict_out is not None:
                    operation_id_dict_out.add(batch.node_type_pad, count, step_labels)
                base = chain.p['latency'].distribution.base_distribution
                std = base.std[0, ..., 0]
                if latency_std_dict_out is not None:
                    latency_std_dict_out.add(std, count, step_labels)
                if latency_reldiff_dict_out is not None:
                    rel = (batch.latency_pad[..., 0] - base.mean[0, ..., 0]) / std.clamp(min=1e-7)
                    latency_reldiff_dict_out.add(rel.abs(), count, step_labels)
    finally:
        vae.train(was_training)
    nll_parts = [c.cpu().numpy() for c in nll_chunks]
    loss_parts = torch.cat(loss_chunks).cpu().numpy().tolist() if loss_chunks else []
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        # sharded scoring (rank r scored batches r, r+W, ... of the stream, no communication
        # until here): put the batches of all ranks back into stream order
        world = dist.get_world_size(process_group)
        shards = [None] * world
        dist.all_gather_object(shards, (nll_parts, label_chunks, loss_parts, weights),
                               group=process_group)
        nll_parts, label_chunks, loss_parts, weights = [], [], [], []
        for i in range(max(len(sh[0]) for sh in shards)):
            for sh in shards:
                if i < len(sh[0]):
                    nll_parts.append(sh[0][i]); label_chunks.append(sh[1][i])
                    loss_parts.append(sh[2][i]); weights.append(sh[3][i])
    nll_list = np.concatenate(nll_parts) if nll_parts else np.zeros([0], np.float32)
    label_list = np.concatenate(label_chunks) if label_chunks else np.zeros([0], np.int64)
    result: Dict[str, float] = {}
    if loss_parts:
        # batch-size weighted mean of the per-batch mean losses (evaluation.py:247-260)
        w = np.asarray(weights, dtype=np.float64)
        result['test_loss'] = float(np.sum(np.asarray(loss_parts) * (w / w.sum())))
    result.update(analyze_anomaly_nll(nll_list, label_list, threshold=test_threshold))
    is_writer = not (dist.is_available() and dist.is_initialized()) or dist.get_rank(process_group) == 0
    if nll_output_file is not None and is_writer:
        os.makedirs(os.path.dirname(os.path.abspath(nll_output_file)), exist_ok=True)
        np.savez_compressed(nll_output_file, nll_list=nll_list, label_list=label_list,
                            anomaly_degree=nll_list / result['best_threshold_latency'])
    result['_nll_list'] = nll_list
    result['_label_list'] = label_list
    return result


@torch.no_grad()
def do_evaluate_prior(vae: TraceVAE, id_manager, latency_range, n_samples: int, batch_size: int,
                      eval_n_z: int, nll_threshold: Optional[float] = None,
                      use_biased: bool = True, output_file: Optional[str] = None,
                      latency_hist_out: Optional[str] = None, rng=None,
                      return_samples: bool = False):
    """Sample traces from the prior and report how many have to be dropped
    (`do_evaluate_prior`, evaluation.py:319-395): per batch `vae.p(n_z=batch_size)` draws the
    traces on the GPU, `p_net_to_trace_graphs` turns them into trees (malformed ones count
    as dropped), the survivors are scored with the K-sample NLL and those at or above
    `nll_threshold` are dropped as well.  Writes `{'drop_rate': ...}` as JSON / YAML.

    One deliberate difference: the reference chains `vae.p` WITHOUT `G=G`
    (evaluation.py:369-370), so its "NLL of the generated trace" is in fact the importance
    weight of yet another freshly drawn trace - here the generated traces are scored
    themselves.

    `rng` feeds `p_net_to_trace_graphs`; with `return_samples` the kept traces are returned
    next to the result dict.  `latency_hist_out` only switches the per-operation latency
    collection on, as in the reference (which gathers the values and never writes them)."""
    from .graph_utils import p_net_to_trace_graphs
    if nll_threshold is None:
        nll_threshold = math.inf
    sample_count = drop_count = 0
    latency_map: Dict[int, List[float]] = {}
    kept = []
    n_batches = (n_samples + batch_size - 1) // batch_size
    for _ in range(n_batches):
        p = vae.p(n_z=batch_size)
        graphs = p_net_to_trace_graphs(
            p, id_manager=id_manager, latency_range=latency_range,
            discard_node_with_type_0=True, discard_node_with_unknown_latency_range=True,
            discard_graph_with_error_node_count=True, rng=rng,
            max_node_count=vae.max_node_count)
        sample_count += len(graphs)
        drop_count += sum(g is None for g in graphs)
        graphs = [g for g in graphs if g is not None]
        if not graphs:
            continue
        G = TraceGraphBatch(id_manager=id_manager, latency_range=latency_range,
                            trace_graphs=graphs)
        G._max_node_count = vae.max_node_count
        _, nll = eval_step(vae, G, eval_n_z, use_biased=use_biased, use_latency_biased=False)
        for g, v in zip(graphs, nll.cpu().numpy()):
            if v >= nll_threshold:
                drop_count += 1
                continue
            kept.append(g)
            if latency_hist_out is not None:
                for _, nd in g.iter_bfs():
                    latency_map.setdefault(int(nd.operation_id), []).append(
                        float(nd.features.avg_latency))
    result = {'drop_rate': float(drop_count / sample_count) if sample_count else 0.0}
    if output_file is not None:
        import json
        import yaml
        _, ext = os.path.splitext(output_file)
        text = json.dumps(result) if ext == '.json' else yaml.safe_dump(result)
        with open(output_file, 'w', encoding='utf-8') as f:
            f.write(text)
    if return_samples:
        return result, kept
    return result
