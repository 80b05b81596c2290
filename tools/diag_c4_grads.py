import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import tracevae_amd.model as model
from test_gpu_model import _setup, LONG_TRACE_FLAGS
from helpers import oracle_batch
from oracle.model import OracleTraceVAE
from tracevae_amd.batch import TraceGraphBatch
from tracevae_amd.train import Trainer
dev = torch.device('cuda:0')
# usage: diag_c4_grads.py [V M width layers batch zdim [watched parameter names...]]
#   (default: the configs[4] case of tests/test_gpu_model.py; 'mid' = 300 32 500 4 16 10)
argv = sys.argv[1:]
V, M, WIDTH, LAYERS, batch, ZD = (int(x) for x in argv[:6]) if len(argv) >= 6 else (10001, 500, 1000, 4, 5, 20)
EXTRA = LONG_TRACE_FLAGS if V > 5000 else ()
WIDTH, LAYERS = WIDTH or None, LAYERS or None      # 0 0 = the default widths (configs[1])
watch = argv[6:] or ['struct_vae.pG_node_type_logits.gnn.3.gnn.attn_r', 'struct_vae.pG_node_type_logits.gnn.0.gnn.attn_l',
         'latency_vae.pG_node_features.gnn.2.gnn.fc.weight', 'latency_vae.pG_node_features.gnn.2.normalization.bias',
         'struct_vae.pG_node_type_logits.gnn.1.gnn.fc.weight', 'latency_vae.pG_node_features.gnn.3.gnn.fc.weight',
         'latency_vae.qz2_gnn_layers.gnn.2.gnn.fc.weight']
for compact, fused in ((True, True),):
    model.COMPACT_LATENCY_DECODER = compact
    model.FUSED_BN_LINEAR = fused
    cfg, vae, store, flat, lr, P, buf, names = _setup(dev, V, M, WIDTH, LAYERS, batch, extra=EXTRA)
    B = len(flat)
    g = torch.Generator().manual_seed(1)
    eps_z = torch.randn(B, cfg.model.struct.z_dim, generator=g); eps_z2 = torch.randn(B, cfg.model.latency.z2_dim, generator=g)
    orc = OracleTraceVAE(cfg.model, V, M, dtype=torch.float64)
    ob = oracle_batch(flat, lr, V, M)
    ref = orc.train_step(P, buf, ob, eps_z.double(), eps_z2.double(), names, kl_beta=cfg.train.kl_beta, l2_reg=0.0,
                         z_unit_ball_reg=cfg.train.z_unit_ball_reg, z2_unit_ball_reg=cfg.train.z2_unit_ball_reg)
    ref['loss'].backward()
    trainer = Trainer(vae, cfg, world_size=1)
    G = TraceGraphBatch.from_store(store, np.arange(B), M)
    trainer.flat.zero_grad()
    out = trainer.forward_loss(G, eps={'z': eps_z, 'z2': eps_z2})
    out['loss'].backward()
    print('compact', compact, 'fused', fused)
    for n, p in vae.named_parameters():
        if n in watch:
            want = P[n].grad
            d = (p.grad.double().cpu() - want).abs().reshape(-1)
            top = torch.topk(d, min(5, d.numel()))
            med = float(d.median())
            print('  %-60s err %.3e wmax %.3e  top abs errs %s at %s median %.2e' % (n, float(d.max() / want.abs().max()), float(want.abs().max()), [('%.2e' % v) for v in top.values.tolist()], top.indices.tolist(), med))
