"""Shape constants of the trace-graph VAE path.

Same environment variables and defaults as the reference
(`tracegnn/constants.py:4-13`, `tracegnn/models/trace_vae/constants.py:3`):
every kernel in `csrc/` takes these as runtime arguments, so changing the
environment only changes what the host passes down.
"""
import os

__all__ = [
    'MAX_NODE_COUNT', 'MAX_SPAN_COUNT', 'USE_MULTI_DIM_LATENCY_CODEC',
    'MAX_LATENCY_DIM', 'MAX_DEPTH', 'LATENCY_DIM', 'set_max_node_count',
]

MAX_NODE_COUNT = int(os.environ.get('MAX_NODE_COUNT', '32'))
MAX_SPAN_COUNT = int(os.environ.get('MAX_SPAN_COUNT', '32'))
USE_MULTI_DIM_LATENCY_CODEC = os.environ.get('USE_MULTI_DIM_LATENCY_CODEC', '0') == '1'
MAX_LATENCY_DIM = int(os.environ.get('MAX_LATENCY_DIM', '5'))
MAX_DEPTH = int(os.environ.get('MAX_DEPTH', '4'))

# only the average latency is modelled (trace_vae/constants.py:3)
LATENCY_DIM = 1

if USE_MULTI_DIM_LATENCY_CODEC:
    raise NotImplementedError(
        'USE_MULTI_DIM_LATENCY_CODEC=1 is outside the accelerated path '
        '(SURVEY.md section 8: the codec is off by default and off the hot path).')


def set_max_node_count(m: int) -> None:
    """Change MAX_NODE_COUNT after import (tests and bench.py use this instead of
    re-importing with another environment)."""
    global MAX_NODE_COUNT
    if m < 2:
        raise ValueError(f'MAX_NODE_COUNT must be >= 2: got {m}')
    MAX_NODE_COUNT = int(m)
