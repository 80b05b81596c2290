"""One GEMM shape, repeated (development tool for counter runs):
    python tools/bench_gemm_one.py [fwd|dx|dw|fwdact|dwact] [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracevae_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
m, n, k = 51200, 1000, 1000
act = None
if which == 'fwdact':
    a, b, ta, tb = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), False, True
    act = (1, torch.rand(k, device=dev) + 0.5, torch.randn(k, device=dev), 0.01)
elif which == 'dwact':
    a, b, ta, tb = torch.randn(m, 1000, device=dev), torch.randn(m, 1000, device=dev), True, False
    m, n, k = 1000, 1000, 51200
    act = (2, torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev), 0.01)
elif which == 'fwd':
    a, b, ta, tb = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), False, True
elif which == 'dx':
    a, b, ta, tb = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), False, False
else:
    a, b, ta, tb = torch.randn(m, 1000, device=dev), torch.randn(m, 1000, device=dev), True, False
    m, n, k = 1000, 1000, 51200
if len(sys.argv) > 3 and sys.argv[3] == 'zero':      # same kernel, no switching activity in the data
    a.zero_(); b.zero_()
c = torch.empty(m, n, device=dev)
for _ in range(3):
    ops.gemm(a, b, ta, tb, out=c, act=act)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    ops.gemm(a, b, ta, tb, out=c, act=act)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / reps
print(f'{which}: {ms:.3f} ms  {2.0 * m * n * k / ms / 1e9:.1f} TF')
