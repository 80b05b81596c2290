"""bench.py with one module-level switch of tracevae_amd.model set, for same-box A/B runs:
    python tools/ab_flag.py DEFERRED_TAILS=0 [bench flags...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tracevae_amd.model as model
name, value = sys.argv[1].split('=')
assert hasattr(model, name), name
setattr(model, name, bool(int(value)))
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
import bench
bench.main()
