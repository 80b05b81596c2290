"""Times the narrow layer products (first decoder layers: 40 / 50 columns on one side) through
ops.gemm (the measurement behind the n, k >= 64 condition of the wide form in gemm_f32.hip)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracevae_amd import hip, ops
dev = torch.device('cuda', 0)
def timed(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows in (51200, 26500):
    x40 = torch.randn(rows, 40, device=dev); w = torch.randn(1004, 40, device=dev) * 0.1
    d = torch.randn(rows, 1004, device=dev)
    for name, fn in (('fwd  [r,40]x[1004,40]^T', lambda: ops.gemm(x40, w, False, True)),
                     ('dx   [r,1004]x[1004,40]', lambda: ops.gemm(d, w, False, False)),
                     ('dW   [r,1004]^T x [r,40]', lambda: ops.gemm(d, x40, True, False))):
        us = timed(fn)
        print(f'rows {rows} {name:28s} {us:7.1f} us form {hip.lib().tv_gemm_f32_last_form()}', flush=True)
