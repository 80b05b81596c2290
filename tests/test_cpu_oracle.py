"""CPU suite, part 1: pins the oracle and the host logic.

* `graph_vectors()` / span merge / CSV plumbing against vectors produced by the reference's
  own code on its `sample_dataset` (fixture: tests/golden/sample_dataset_vectors.json,
  generator: tests/golden/make_reference_vectors.py);
* hand-derivable known answers for the pieces SURVEY.md section 8c lists: `dense_to_triu`
  ordering, `node_count_mask`, biased log-prob thresholding, `SafeNormal` clipping;
* oracle self-consistency: CSC = stable sort, reverse positions, K-sample replicas.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from helpers import hand_made_flat, make_config, make_flat, oracle_batch, oracle_state

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                      'sample_dataset_vectors.json')


@pytest.fixture(scope='module')
def golden():
    with open(GOLDEN, 'r', encoding='utf-8') as f:
        return json.load(f)


def test_graph_vectors_match_reference(golden):
    """a1: our TraceGraph pipeline reproduces the reference's arrays for all 17 sample traces."""
    from tracevae_amd.data import TraceGraphIDManager, df_to_trace_graphs
    idm = TraceGraphIDManager.from_mappings(golden['service_id'], golden['operation_id'],
                                            golden['num_operations'])
    assert idm.num_operations == 1488
    total = 0
    for set_name, entry in golden['sets'].items():
        graphs = df_to_trace_graphs(entry['rows'], idm, test_label=entry['test_label'],
                                    merge_spans=True)
        assert len(graphs) == len(entry['graphs']), set_name
        for g, want in zip(graphs, entry['graphs']):
            gv = g.graph_vectors()
            assert list(g.trace_id) == want['trace_id']
            assert g.node_count == want['node_count'] and g.max_depth == want['max_depth']
            assert g.data == want['data']
            for key in ('u', 'v', 'node_type', 'node_depth', 'node_idx', 'span_count'):
                arr = getattr(gv, key)
                assert arr.dtype == np.int64
                assert arr.tolist() == want[key], (set_name, key)
            for key in ('avg_latency', 'max_latency', 'min_latency'):
                arr = getattr(gv, key)
                assert arr.dtype == np.float32
                np.testing.assert_array_equal(arr, np.asarray(want[key], dtype=np.float32))
            assert np.all(gv.u < gv.v)
            total += 1
    assert total == 17


def test_sample_dataset_batching_oracle(golden):
    """a2-a6 on the reference's sample traces: structure of the batched encoder / decoder
    graphs and the observed tensors."""
    from helpers import golden_sample_flat
    flat, lr, _, graphs = golden_sample_flat()
    off = flat.node_offsets
    flat.validate(32)
    b = oracle_batch(flat, lr, golden['num_operations'], 32)
    n, B, M = int(off[-1]), len(graphs), 32
    assert len(b['enc_src']) == 3 * n - 2 * B
    assert len(b['dec_src']) == B * M + 2 * (n - B)
    assert b['adj'].shape == (B, 496) and b['adj'].sum() == n - B
    assert b['node_count'].tolist() == [g['node_count'] for g in graphs]
    # in-degree = tree degree + 1 (self loop), every node has at least one in-edge
    deg = np.diff(b['enc_indptr'])
    assert deg.min() >= 1 and deg.sum() == 3 * n - 2 * B
    # padded decoder rows are isolated nodes with a self loop
    for i, g in enumerate(graphs):
        pad_deg = np.diff(b['dec_indptr'])[i * M + g['node_count']:(i + 1) * M]
        assert np.all(pad_deg == 1)
    # normalised latency: float64 arithmetic, one cast (dataset.py:73-79)
    t0 = int(flat.node_type[0])
    mu, sd = lr[t0]
    assert b['latency'][0, 0] == np.float32((np.float64(flat.avg_latency[0]) - mu) / (sd + 1e-5))


def test_csc_is_stable_sort_and_reverse_positions():
    flat, lr = hand_made_flat([[0, 0, 1, 1], [0], [0, 1, 2]], 20, seed=1)
    b = oracle_batch(flat, lr, 20, 6)
    for pfx in ('enc', 'dec'):
        src, dst = b[pfx + '_src'], b[pfx + '_dst']
        indptr, indices, eids, rev = (b[f'{pfx}_{k}'] for k in ('indptr', 'indices', 'eids', 'rev'))
        assert np.array_equal(dst[eids], np.repeat(np.arange(len(indptr) - 1), np.diff(indptr)))
        assert np.array_equal(src[eids], indices)
        for d in range(len(indptr) - 1):               # stable: edge ids ascend within a dst
            seg = eids[indptr[d]:indptr[d + 1]]
            assert np.all(np.diff(seg) > 0)
        owner = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
        assert np.array_equal(indices[rev], owner) and np.array_equal(owner[rev], indices)
        assert np.array_equal(rev[rev], np.arange(len(rev)))
    # first trace, node 1: in-edges = [parent 0, children 3 and 4, self]
    p0, p1 = b['enc_indptr'][1], b['enc_indptr'][2]
    assert b['enc_indices'][p0:p1].tolist() == [0, 3, 4, 1]


def test_known_answers_triu_mask_biased_safe():
    from oracle.model import biased, dense_to_triu, normal_log_pdf
    from oracle.batching import dense_triu_adj
    # dense_to_triu: strict upper triangle, row-major (tensor_utils.py:60-67)
    x = torch.arange(16.0).reshape(4, 4)
    assert dense_to_triu(x, 4).tolist() == [1.0, 2.0, 3.0, 6.0, 7.0, 11.0]
    adj = dense_triu_adj(np.array([0, 1, 0, 1, 2]), np.array([1, 0, 2, 1, 0]), 4)
    assert adj.tolist() == [1.0, 1.0, 0.0, 0.0, 0.0, 0.0]       # only (0,1), (0,2) are upper
    # index formula used by the kernel: u(2M-u-1)/2 + (v-u-1)
    M = 7
    pos = {(i, j): k for k, (i, j) in enumerate((i, j) for i in range(M) for j in range(i + 1, M))}
    for (i, j), k in pos.items():
        assert i * (2 * M - i - 1) // 2 + (j - i - 1) == k
    # node_count_mask (tensor_utils.py:128-136)
    nc = torch.tensor([2, 4])
    mask = torch.arange(4).unsqueeze(0) < nc.unsqueeze(-1)
    assert mask.tolist() == [[True, True, False, False], [True, True, True, True]]
    # biased log-prob (distributions.py:125-134): below log(0.5) is multiplied by alpha
    lp = torch.tensor([math.log(0.9), math.log(0.5), math.log(0.4)])
    out = biased(lp, 10.0, 0.5)
    assert out[0] == lp[0] and out[1] == lp[1] and torch.isclose(out[2], 10 * lp[2])
    # SafeNormal (distributions.py:300-315): x clipped to mean +- thr*std before the pdf
    mean, logstd, thr = torch.tensor([0.0]), torch.tensor([0.0]), 6.0
    x = torch.tensor([100.0])
    clipped = torch.maximum(torch.minimum(x, mean + thr), mean - thr)
    assert torch.isclose(normal_log_pdf(clipped, mean, logstd),
                         torch.tensor(-0.5 * math.log(2 * math.pi) - 18.0))


def test_oracle_sample_axis_and_determinism():
    """K-sample eval: every sample is an independent decoder pass (gnn_layers.py:90-97), so
    sample k of a K-sample call equals a single-sample call with the same epsilon."""
    from oracle.model import OracleTraceVAE
    from tracevae_amd.model import TraceVAE
    torch.manual_seed(0)
    V, M = 30, 8
    cfg = make_config(width=8, layers=2)
    vae = TraceVAE(cfg.model, V, max_node_count=M)
    P, buf, names = oracle_state(vae)
    flat, lr = make_flat(5, V, M, seed=4)
    ob = oracle_batch(flat, lr, V, M)
    orc = OracleTraceVAE(cfg.model, V, M)
    ez, ez2 = torch.randn(3, 5, 10), torch.randn(3, 5, 10)
    full = orc.eval_step(P, buf, ob, ez, ez2, latency_log_prob_weight=True)
    one = orc.eval_step(P, buf, ob, ez[1:2], ez2[1:2], latency_log_prob_weight=True)
    w_full = (full['p']['log_p_adj'] + full['p']['log_p_node_type'])[1]
    w_one = (one['p']['log_p_adj'] + one['p']['log_p_node_type'])[0]
    assert torch.allclose(w_full, w_one, rtol=1e-5, atol=1e-5)
    # nll = -log mean exp: K = 1 reduces to the sgvb loss
    assert torch.allclose(one['nll'], one['loss'], rtol=1e-6)
    again = orc.eval_step(P, buf, ob, ez, ez2, latency_log_prob_weight=True)
    assert torch.equal(full['nll'], again['nll'])


def test_synthetic_generator_contract():
    from tracevae_amd.synth import SyntheticSpec, generate_flat
    spec = SyntheticSpec(num_operations=101, max_spans=30, seed=5)
    a, lr = generate_flat(spec, 40, anomaly_ratio=0.25)
    b, _ = generate_flat(spec, 40, anomaly_ratio=0.25)
    a.validate(30)
    assert np.array_equal(a.node_offsets, b.node_offsets) and np.array_equal(a.node_type, b.node_type)
    counts = a.node_counts
    assert counts.min() >= 2 and counts.max() <= 30
    assert a.node_type.min() >= 1 and a.node_type.max() < 101
    assert set(np.unique(a.labels)) <= {0, 1, 2}
    # merge rule: no two siblings share an operation id
    for i in range(len(a)):
        n0, e0, e1 = a.node_offsets[i], a.node_offsets[i] - i, a.node_offsets[i + 1] - i - 1
        pairs = set()
        for u, v in zip(a.u[e0:e1], a.v[e0:e1]):
            key = (int(u), int(a.node_type[n0 + v]))
            assert key not in pairs
            pairs.add(key)
        assert np.array_equal(a.v[e0:e1], np.arange(1, e1 - e0 + 1))      # BFS numbering
