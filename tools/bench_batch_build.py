import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from helpers import make_flat
from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
for B, M, V in ((512, 100, 1001), (8, 500, 10001), (32, 500, 10001)):
    flat, lr = make_flat(B, V, M, seed=1)
    store = DeviceTraceStore(flat, lr, V)
    f = lambda: TraceGraphBatch.from_store(store, np.arange(B), M).build_dgl()
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print(f'batch build B={B} M={M}: {s.elapsed_time(e)/20*1e3:.1f} us per call (incl. allocs + memsets)')
