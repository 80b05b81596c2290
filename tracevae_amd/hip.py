"""ctypes binding of `libtracevae_hip.so` (C-ABI: `include/tracevae_hip.h`).

The product path has no CPU fallback: if the library is missing, or a call fails, this
module raises.  Tensors are passed as raw device pointers (`tensor.data_ptr()`), the
stream is torch's current HIP stream.
"""
import ctypes
import os
from typing import Dict, Optional, Tuple

import torch

__all__ = ['lib', 'call', 'query', 'array_type', 'ptr', 'stream', 'library_path', 'header_path', 'SIGNATURES',
           'HipLibraryError']

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = 'libtracevae_hip.so'


class HipLibraryError(RuntimeError):
    pass


def library_path() -> str:
    # TRACEVAE_HIP_LIB selects another build of the same library (kernel A/B experiments)
    return os.environ.get('TRACEVAE_HIP_LIB') or os.path.join(_HERE, 'csrc', _LIB_NAME)


def header_path() -> str:
    return os.path.join(os.path.dirname(_HERE), 'include', 'tracevae_hip.h')


def _parse_header(path: str) -> Dict[str, str]:
    """Derive the ctypes signatures from the C declarations, so that the binding cannot
    drift from `include/tracevae_hip.h`.
    i = int, l = int64, f = float, p = pointer, s = stream (last `void *stream`), c = char*."""
    import re
    text = open(path, 'r', encoding='utf-8').read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    sigs: Dict[str, str] = {}
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(tv_\w+)\s*\(([^)]*)\)\s*;', text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()

        def code(decl: str, is_ret: bool = False) -> str:
            decl = ' '.join(decl.split())
            if '*' in decl:
                if 'char' in decl:
                    return 'c'
                if not is_ret and decl.replace(' ', '').startswith('void*') and decl.endswith('stream'):
                    return 's'
                return 'p'
            base = decl.replace('const', ' ').split()
            ty = base[0]
            return {'int': 'i', 'int32_t': 'i', 'int64_t': 'l', 'float': 'f'}[ty]

        args = '' if params in ('', 'void') else ''.join(code(p) for p in params.split(','))
        sigs[name] = args + '->' + code(ret + ' x' if '*' not in ret else ret, True)
    return sigs


SIGNATURES: Dict[str, str] = _parse_header(header_path())

_CTYPES = {'i': ctypes.c_int, 'l': ctypes.c_int64, 'f': ctypes.c_float, 'p': ctypes.c_void_p,
           's': ctypes.c_void_p, 'c': ctypes.c_char_p}

_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Load the library once; raise (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.isfile(path):
            raise HipLibraryError(
                f'{path} is missing: build it with `make -C tracevae_amd/csrc` '
                f'(or `python -c "import __graft_entry__ as g; g.build()"`). '
                f'There is no CPU fallback for the accelerated path.')
        handle = ctypes.CDLL(path)
        for name, sig in SIGNATURES.items():
            args, ret = sig.split('->')
            fn = getattr(handle, name)
            fn.argtypes = [_CTYPES[c] for c in args]
            fn.restype = _CTYPES[ret]
        _lib = handle
    return _lib


def ptr(t) -> Optional[int]:
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    if not t.is_contiguous():
        raise ValueError('tensors handed to the HIP library must be contiguous')
    return t.data_ptr()


def ptr_rows(t) -> int:
    """Device pointer of a 2-D tensor whose rows are contiguous but may be strided (a column
    slice of a wider buffer); the caller passes `t.stride(0)` as the row stride."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError('a row-strided operand must be 2-D with contiguous rows')
    return t.data_ptr()


def stream() -> int:
    """Raw handle of torch's current HIP stream on the current device (the C-level accessor:
    `torch.cuda.current_stream()` builds a Stream object per call, 15 us that a step of several
    hundred launches pays several hundred times)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


_FUNCTIONS: Dict[str, object] = {}


def call(name: str, *args) -> None:
    """Call an int-returning entry point on the current stream; raise on error."""
    fn = _FUNCTIONS.get(name)
    if fn is None:
        fn = _FUNCTIONS[name] = getattr(lib(), name)
    rc = fn(*args, stream())
    if rc != 0:
        msg = lib().tv_last_error().decode('utf-8', 'replace')
        raise HipLibraryError(f'{name} failed ({rc}): {msg}')


_ARRAY_TYPES: Dict[tuple, type] = {}


def array_type(ctype, n: int):
    """`ctype * n`, kept: ctypes drops an array type when its last instance goes, and a type is a
    reference cycle - a host table built per call would leave one for Python's cyclic collector
    every time."""
    t = _ARRAY_TYPES.get((ctype, n))
    if t is None:
        t = _ARRAY_TYPES[(ctype, n)] = ctype * n
    return t


_QUERIES: Dict[tuple, int] = {}


def query(name: str, *args) -> int:
    """An int64-returning size query (workspace bytes, slot counts): pure functions of their
    arguments, so the answers are kept."""
    key = (name,) + args
    v = _QUERIES.get(key)
    if v is None:
        v = _QUERIES[key] = int(getattr(lib(), name)(*args))
    return v
