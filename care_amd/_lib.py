"""ctypes binding of libcare_hip.so (the C ABI declared in include/care_hip.h).

There is deliberately NO fallback: if the library is missing or a call is rejected the
product path raises.  Nothing here imports `oracle/`.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch

CARE_F32, CARE_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
ACT_CODES = {"linear": ACT_NONE, "relu": ACT_RELU, "gelu": ACT_GELU}
ABI_VERSION = 1

_ERRORS = {-1: "CARE_EINVAL (null pointer / bad size)", -2: "CARE_EALIGN (alignment)",
           -3: "CARE_ESHAPE (unsupported shape)", -4: "CARE_EDTYPE (unknown dtype/activation)"}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcare_hip.so")

# name -> argtypes, in the exact order of include/care_hip.h
_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float
SIGNATURES = {
    "care_gemm": [_P, _L, _P, _I, _P, _P, _L, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "care_gemm_argmax": [_P, _L, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    "care_greedy_update": [_P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "care_add_ln": [_P, _L, _P, _L, _P, _P, _P, _F, _P, _L, _I, _I, _I, _I, _I, _P],
    "care_group_mean": [_P, _L, _I, _I, _I, _P, _L, _I, _I, _I, _P],
    "care_concept_finish": [_P, _L, _P, _L, _P, _I, _I, _P],
    "care_concept_topk_embed": [_P, _L, _I, _I, _P, _P, _P, _P, _F, _P, _P, _L, _I, _I, _I, _I, _P],
    "care_embed_ln": [_P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P, _F, _P, _L, _I, _I, _I, _P],
    "care_attention": [_P, _L, _P, _P, _I, _L, _L, _I, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _L,
                       _I, _I, _P],
    "care_beam_select": [_P, _L, _I, _I, _P, _P, _I, _P],
    "care_beam_advance": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
}
PLAIN = {"care_version": (c_int, []), "care_arch": (c_char_p, []), "care_argmax_parts": (c_int, [c_int])}

_lib = None


class CareHipError(RuntimeError):
    pass


def load(path: str = LIB_PATH):
    """Load the library once; raises if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise CareHipError(
            "libcare_hip.so not found at {} - run `python -m care_amd.build` "
            "(the HIP path has no CPU/PyTorch fallback)".format(path))
    lib = ctypes.CDLL(path)
    for name, (res, args) in PLAIN.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype, fn.argtypes = c_int, args
    if lib.care_version() != ABI_VERSION:
        raise CareHipError("libcare_hip.so ABI {} != expected {}".format(lib.care_version(), ABI_VERSION))
    _lib = lib
    return lib


def exported_symbols():
    return sorted(list(SIGNATURES) + list(PLAIN))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args):
    """Invoke an ABI function on torch's current stream; raise on any non-zero status."""
    rc = getattr(load(), name)(*args, stream_ptr())
    if rc != 0:
        what = _ERRORS.get(rc, "hipError_t {}".format(rc))
        raise CareHipError("{} failed: {}".format(name, what))


def argmax_parts(n: int) -> int:
    return load().care_argmax_parts(n)
