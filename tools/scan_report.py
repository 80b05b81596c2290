"""Which products of a training step still scan an operand for its maximum (fp16 GEMM forms)?
    python tools/scan_report.py [batch]"""
import collections, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tracevae_amd import hip, ops
from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
from tracevae_amd.model import TraceVAE
from tracevae_amd.synth import SyntheticSpec, generate_flat
from tracevae_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
V, M = 1001, 100
cfg = bench.make_config()
flat, lr = generate_flat(SyntheticSpec(num_operations=V, max_spans=M, seed=1), 1024, seed=2)
store = DeviceTraceStore(flat, lr, V)
torch.manual_seed(0)
vae = TraceVAE(cfg.model, V, max_node_count=M).to('cuda:0')
for m in vae.modules():
    if hasattr(m, 'initialized'):
        m.initialized = True
trainer = Trainer(vae, cfg, world_size=1)
rng = np.random.RandomState(0)
def one():
    ids = rng.randint(0, len(store), size=B)
    return trainer.step(TraceGraphBatch.from_store(store, ids, M), check_nan=False)['loss']
one(); one()
log = collections.Counter()
orig = hip.call
def spy(name, *args):
    rc = orig(name, *args)
    if name == 'tv_gemm_f32_ex':
        ta, tb, m, n, k = args[:5]
        form = int(hip.lib().tv_gemm_f32_last_form())
        a_state, b_state = args[19], args[21]
        if form:
            if a_state != 2: log[('A', ta, tb, m, n, k, form)] += 1
            if b_state != 2: log[('B', ta, tb, m, n, k, form)] += 1
    return rc
hip.call = spy
ops.hip.call = spy
before = dict(ops.OPERAND_SCANS)
one()
torch.cuda.synchronize()
print('per step: kept', ops.OPERAND_SCANS['kept'] - before['kept'], 'scanned', ops.OPERAND_SCANS['scanned'] - before['scanned'])
print('(the layer calls issue their products inside the library: not seen here)')
for key, cnt in sorted(log.items(), key=lambda kv: -kv[0][3] * kv[0][5]):
    print(cnt, key)
