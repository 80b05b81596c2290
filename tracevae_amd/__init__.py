"""tracevae_amd: the TraceVAE graph-VAE training / scoring path on MI355X (gfx950).

Python host code on PyTorch-ROCm keeps the reference's model API (`TraceGraphBatch`,
`TraceVAE.q / .p`, the chain object, `train.sh` / `test.sh` flags, `final.pt` layout); the
operators underneath are hand-written HIP kernels in `csrc/`, reached through the C-ABI
declared in `include/tracevae_hip.h`.  See DESIGN.md.
"""
from . import constants
from .config import (ExpConfig, TraceLatencyVAEConfig, TraceStructVAEConfig, TraceVAEConfig,
                     parse_flags)
from .data import (FlatTraces, TraceGraph, TraceGraphIDManager, TraceGraphLatencyRangeFile,
                   TraceGraphNode, TraceGraphNodeFeatures, flatten_trace_graphs)

__all__ = [
    'constants', 'ExpConfig', 'TraceVAEConfig', 'TraceStructVAEConfig', 'TraceLatencyVAEConfig',
    'parse_flags', 'FlatTraces', 'TraceGraph', 'TraceGraphNode', 'TraceGraphNodeFeatures',
    'TraceGraphIDManager', 'TraceGraphLatencyRangeFile', 'flatten_trace_graphs',
    'TraceGraphBatch', 'DeviceTraceStore', 'TraceVAE',
]


def __getattr__(name):
    # the torch-facing modules are imported lazily so that `import tracevae_amd` stays light
    if name in ('TraceGraphBatch', 'DeviceTraceStore'):
        from . import batch
        return getattr(batch, name)
    if name == 'TraceVAE':
        from .model import TraceVAE
        return TraceVAE
    raise AttributeError(name)
