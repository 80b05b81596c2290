"""Which objects of a training step end up as cyclic garbage (freed by Python's collector, not
by reference counts)?    python tools/cycle_probe.py [batch]"""
import collections, gc, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tracevae_amd.batch import DeviceTraceStore, TraceGraphBatch
from tracevae_amd.model import TraceVAE
from tracevae_amd.synth import SyntheticSpec, generate_flat
from tracevae_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
    return trainer.step(TraceGraphBatch.from_store(store, ids, M))['loss']
for _ in range(4):
    one()
trainer.finish(); torch.cuda.synchronize()
gc.collect()
gc.disable()
m0 = torch.cuda.memory_allocated()
one()
trainer.finish(); torch.cuda.synchronize()
m1 = torch.cuda.memory_allocated()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
gc.set_debug(0)
kinds = collections.Counter(type(o).__module__ + '.' + type(o).__qualname__ for o in gc.garbage)
print(f'one step: device memory {m0 / 2**20:.1f} -> {m1 / 2**20:.1f} MiB; {n} unreachable objects')
for k, v in kinds.most_common(25):
    print(f'{v:6d}  {k}')
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
print('tensors in cycles:', len(tens), 'bytes', sum(t.numel() * t.element_size() for t in tens if t.is_cuda) / 2**20, 'MiB')
shapes = collections.Counter((tuple(t.shape), type(t.grad_fn).__name__ if t.grad_fn is not None else None) for t in tens)
for k, v in shapes.most_common(30):
    print(v, k)
import types
for o in gc.garbage:
    if isinstance(o, types.FunctionType):
        print('function', o.__qualname__, o.__code__.co_filename, o.__code__.co_firstlineno)
tl = [o for o in gc.garbage if isinstance(o, tuple)]
for t in tl[:12]:
    print('tuple', len(t), [type(x).__name__ for x in t][:8])
