"""ctypes binding of libcare_hip.so (the C ABI declared in include/care_hip.h).

There is deliberately NO fallback: if the library is missing or a call is rejected the
product path raises.  Nothing here imports `oracle/`.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

import torch

CARE_F32, CARE_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
ACT_CODES = {"linear": ACT_NONE, "relu": ACT_RELU, "gelu": ACT_GELU}
ABI_VERSION = 22

_ERRORS = {-1: "CARE_EINVAL (null pointer / bad size)", -2: "CARE_EALIGN (alignment)",
           -3: "CARE_ESHAPE (unsupported shape)", -4: "CARE_EDTYPE (unknown dtype/activation)"}

LIB_PATH = os.environ.get("CARE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcare_hip.so")
# library variants (care_amd/build.py): "" = libcare_hip.so (16-bit type bf16), "f16" = libcare_hip_f16.so (IEEE half)
VARIANT_H16 = {"": "bf16", "f16": "fp16"}

# name -> argtypes, in the exact order of include/care_hip.h
_P, _I, _L, _F, _U = c_void_p, c_int, c_int64, c_float, c_uint64
SIGNATURES = {
    "care_gemm": [_P, _L, _P, _I, _P, _P, _L, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "care_split3_weight": [_P, _P, _I, _I, _P],
    "care_gemm_split3": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _P],
    "care_gemm_argmax": [_P, _L, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    "care_gemm_bf16": [_P, _L, _I, _P, _P, _P, _L, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "care_gemm_argmax_bf16": [_P, _L, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "care_gemm_tile": [_P, _L, _P, _P, _P, _L, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "care_split2_act": [_P, _L, _P, _I, _I, _P],
    "care_gemm_tile_split3": [_P, _P, _P, _P, _L, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "care_gemm_tile_split3_argmax": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "care_absmax": [_P, _L, _I, _I, _P, _P],
    "care_split_pieces": [_P, _L, _I, _I, _I, _I, _I, _P, _I, _P, _P],
    "care_gemm_tile_split3_scaled": [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _I, _P],
    "care_gemm_tile_batched": [_P, _L, _L, _P, _L, _L, _P, _I, _P, _L, _L, _I, _I, _I, _I, _I, _P],
    "care_gemm_tile_argmax": [_P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "care_score_partials": [_P, _P, _P, _P, _I, _P, _P, _I, _P],
    "care_label_logits": [_P, _L, _P, _P, _P, _I, _I, _I, _P],
    "care_score_partials_lab": [_P, _P, _P, _I, _P, _P, _P, _I, _P],
    "care_score_logits": [_P, _L, _I, _P, _P, _P, _I, _P],
    "care_greedy_update": [_P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "care_greedy_update_embed": [_P, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _F,
                                 _P, _P, _L, _I, _P],
    "care_add_ln": [_P, _L, _P, _L, _P, _P, _P, _F, _P, _P, _L, _I, _I, _I, _I, _I, _I, _L, _P],
    "care_gemm_ln": [_P, _L, _I, _P, _P, _P, _L, _P, _P, _P, _F, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "care_gemm_ln_packed": [_P, _L, _I, _P, _P, _P, _L, _P, _P, _F, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "care_pack_ln_weight": [_P, _P, _I, _I, _P],
    "care_pack_ln_weight_split": [_P, _P, _I, _I, _P],
    "care_gemm_ln_split": [_P, _L, _P, _P, _P, _P, _F, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "care_gemm_bf16_splitk": [_P, _L, _I, _P, _P, _P, _L, _L, _I, _I, _I, _P],
    "care_group_mean": [_P, _L, _I, _I, _I, _P, _L, _I, _I, _I, _P],
    "care_concept_finish": [_P, _L, _P, _L, _P, _I, _I, _P],
    "care_concept_topk_embed": [_P, _L, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "care_embed_ln": [_P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P, _F, _P, _P, _L, _I, _I, _I, _P],
    "care_attention": [_P, _L, _P, _P, _I, _L, _L, _I, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _L,
                       _I, _I, _I, _P],
    "care_attention_seq": [_P, _L, _P, _P, _L, _L, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _L, _I, _I, _P],
    "care_attention_latent": [_P, _L, _P, _L, _L, _I, _I, _P, _I, _P, _L, _I, _I, _I, _P],
    "care_head_expand": [_P, _L, _P, _P, _L, _I, _I, _P],
    "care_head_reduce": [_P, _L, _P, _P, _P, _L, _I, _I, _P],
    "care_gemm_argmax_bf16_min": [_P, _L, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "care_beam_threshold": [_P, _I, _I, _P, _P, _I, _P],
    "care_gemm_argmax_bf16_tiles": [_P, _L, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "care_beam_sparse_collect": [_P, _L, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P],
    "care_gemm_collect_bf16": [_P, _L, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "care_beam_pick": [_P, _P, _I, _P, _P, _P, _I, _I, _P, _L, _I, _P, _I, _I, _P, _P, _I, _P],
    "care_gemm_tile_beam": [_P, _L, _P, _P, _P, _P, _I, _I, _I, _P],
    "care_beam_pick_groups": [_P, _P, _P, _I, _I, _P, _L, _P, _I, _I, _P, _P, _I, _P],
    "care_beam_select": [_P, _L, _I, _I, _P, _P, _I, _I, _P],
    "care_ensemble_select": [_P, _I, _L, _I, _I, _P, _P, _I, _P],
    "care_attention_probs": [_P, _L, _P, _I, _L, _L, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _I, _I, _P],
    "care_timestamp": [_P, _P],
    "care_decode_resident": [_P, _I, _P, _P, _P, _I, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P,
                             _P, _L, _I, _I, _P],
    "care_decode_resident_beam": [_P, _I, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I,
                                  _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L, _I, _I, _P],
    "care_decode_chain_beam": [_P, _I, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I,
                               _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L, _I, _P],
    "care_gemm_kn": [_P, _L, _I, _P, _L, _P, _L, _I, _I, _I, _P],
    "care_gemm_kn_splitk": [_P, _L, _I, _P, _L, _P, _L, _L, _I, _I, _I, _I, _P],
    "care_ln_bwd": [_P, _L, _P, _L, _P, _P, _L, _F, _P, _L, _P, _P, _I, _I, _P],
    "care_act": [_P, _P, _P, _L, _I, _P],
    "care_dropout": [_P, _P, _L, _F, _U, _P],
    "care_strided_sum": [_P, _L, _P, _L, _I, _I, _I, _L, _L, _F, _P],
    "care_bcast_rows": [_P, _L, _P, _L, _I, _I, _I, _F, _P],
    "care_add_pos_sem": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "care_scatter_add_rows": [_P, _L, _P, _P, _L, _I, _I, _I, _P],
    "care_concept_bwd": [_P, _L, _P, _L, _P, _P, _L, _I, _I, _P],
    "care_attn_pv": [_P, _P, _L, _L, _P, _L, _I, _I, _I, _I, _F, _U, _P],
    "care_attn_bwd": [_P, _L, _P, _P, _L, _L, _P, _P, _L, _P, _L, _P, _P, _L, _L, _P, _I, _I, _I, _I, _I, _F, _U, _P],
    "care_active_slots": [_P, _I, _P, _P, _P],
    "care_gather_rows": [_P, _L, _P, _L, _P, _I, _L, _P],
    "care_scatter_rows": [_P, _L, _P, _L, _P, _I, _L, _P],
    "care_expand_index": [_P, _I, _I, _P, _P],
    "care_remap_rows": [_P, _L, _P, _I, _P],
    "care_beam_advance": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
}
PLAIN = {"care_version": (c_int, []), "care_arch": (c_char_p, []), "care_source_hash": (c_char_p, []),
         "care_build_flags": (c_char_p, []), "care_h16": (c_char_p, []), "care_argmax_parts": (c_int, [c_int]),
         "care_argmax_parts_bf16": (c_int, [c_int, c_int]),
         "care_argmax_parts_tile": (c_int, [c_int]),
         "care_argmax_parts_bf16_min": (c_int, [c_int, c_int, c_int, c_int, c_int]),
         "care_beam_sparse_applies": (c_int, [c_int, c_int, c_int, c_int]),
         "care_decode_resident_scratch": (c_int64, [c_int, c_int, c_int, c_int]),
         "care_decode_resident_beam_scratch": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
         "care_decode_chain_beam_scratch": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
         "care_decode_resident_debug": (None, [c_int, c_int]),
         "care_resident_set_fenced": (None, [c_int]), "care_resident_fenced": (c_int, []),
         "care_gemm_kn_splits": (c_int, [c_int, c_int, c_int])}


class ResidentAttn(ctypes.Structure):
    """care_resident_attn (include/care_hip.h)."""
    _fields_ = [("q_w", c_void_p), ("q_b", c_void_p), ("o_w", c_void_p), ("o_b", c_void_p), ("ln_g", c_void_p), ("ln_b", c_void_p),
                ("kv", c_void_p), ("kv_batch_stride", c_int64), ("nkeys", ctypes.c_int32), ("rows_per_kv", ctypes.c_int32),
                ("bias", c_void_p), ("bias_ld", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class ResidentLayer(ctypes.Structure):
    """care_resident_layer (include/care_hip.h)."""
    _fields_ = [("qkv_w", c_void_p), ("qkv_b", c_void_p), ("o_w", c_void_p), ("o_b", c_void_p), ("ln_g", c_void_p), ("ln_b", c_void_p),
                ("self_kv", c_void_p), ("att", ResidentAttn * 2), ("n_att", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p), ("ffn_g", c_void_p), ("ffn_b", c_void_p)]

_libs = {}


class CareHipError(RuntimeError):
    pass


def variant_path(variant: str = "") -> str:
    if variant == "":
        return LIB_PATH
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcare_hip_{}.so".format(variant))


def _ensure_current(variant: str, path: str) -> None:
    """The library must come from THIS tree's sources: care_amd/build.py's source hash (kernel sources, headers, flags)
    is compiled into it.  Stale or missing -> rebuilt when hipcc is at hand, refused otherwise.  A library named through
    CARE_HIP_LIB (tools: ablation builds with flags of their own) is taken as it is."""
    if variant == "" and os.environ.get("CARE_HIP_LIB"):
        return
    from . import build as _build
    if not _build.have_sources():
        return
    if not _build.needs_build(variant):
        return
    why = "missing" if not os.path.exists(path) else "built from other sources (hash {} != tree {})".format(
        _build.embedded_hash(path), _build.source_hash(variant))
    if os.environ.get("CARE_NO_REBUILD") or not _build.have_hipcc():
        raise CareHipError("{} is {} - run `python -m care_amd.build{}` (the HIP path has no CPU/PyTorch fallback)".format(
            path, why, " --variant " + variant if variant else ""))
    import sys
    print("care_amd: {} is {}: rebuilding".format(os.path.basename(path), why), file=sys.stderr, flush=True)
    _build.build(variant=variant, verbose=False)


def load(path: str = None, variant: str = ""):
    """Load the library of `variant` once; raises if it is absent, stale (and cannot be rebuilt) or has the wrong ABI."""
    if variant in _libs and path is None:
        return _libs[variant]
    if variant not in VARIANT_H16:
        raise CareHipError("unknown library variant {!r}".format(variant))
    explicit = path is not None
    path = path or variant_path(variant)
    if not explicit:
        _ensure_current(variant, path)
    if not os.path.exists(path):
        raise CareHipError(
            "{} not found - run `python -m care_amd.build` "
            "(the HIP path has no CPU/PyTorch fallback)".format(path))
    lib = ctypes.CDLL(path)
    for name, (res, args) in PLAIN.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype, fn.argtypes = c_int, args
    if lib.care_version() != ABI_VERSION:
        raise CareHipError("{} ABI {} != expected {}".format(os.path.basename(path), lib.care_version(), ABI_VERSION))
    if lib.care_h16().decode() != VARIANT_H16[variant]:
        raise CareHipError("{} was compiled for 16-bit type {}, variant {!r} needs {}".format(
            path, lib.care_h16().decode(), variant, VARIANT_H16[variant]))
    if not explicit:
        _libs[variant] = lib
    return lib


def source_hash(variant: str = "") -> str:
    """The source hash the loaded library of `variant` carries (care_amd/build.py)."""
    return load(variant=variant).care_source_hash().decode()


def exported_symbols():
    return sorted(list(SIGNATURES) + list(PLAIN))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


# When set to a dict, every tagged launch is bracketed by HIP events recorded on the
# launch stream (torch's current stream): TIMING[tag] = [(start_event, end_event), ...].
# bench.py uses it to measure per-kernel durations live; it is None in normal operation.
TIMING = None
# When set to {"tag": name, "buf": int64 device tensor [2 * cap], "n": 0}, every launch tagged `name` is bracketed by
# two care_timestamp kernels (device wall clock, 100 MHz) - also under hipGraph capture, where events cannot be
# used: slot 2 i / 2 i + 1 = before / after the i-th such launch of a pass.  bench.py only.
STAMP = None
LAST_CALL = {}  # tag -> (function name, args): lets bench.py re-launch one kernel back to back


def call(name: str, *args, tag: str = None, variant: str = ""):
    """Invoke an ABI function (of the library `variant`) on torch's current stream; raise on any non-zero status."""
    lib = load(variant=variant)
    fn = getattr(lib, name)
    if STAMP is not None and tag == STAMP["tag"] and 2 * STAMP["n"] + 2 <= STAMP["buf"].numel():
        base, i = STAMP["buf"].data_ptr(), STAMP["n"]
        STAMP["n"] = i + 1
        lib.care_timestamp(base + 16 * i, stream_ptr())
        rc = fn(*args, stream_ptr())
        lib.care_timestamp(base + 16 * i + 8, stream_ptr())
    elif TIMING is not None and tag is not None:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        rc = fn(*args, stream_ptr())
        end.record()
        TIMING.setdefault(tag, []).append((start, end))
        LAST_CALL[tag] = (name, args, variant)
    else:
        rc = fn(*args, stream_ptr())
    if rc != 0:
        what = _ERRORS.get(rc, "hipError_t {}".format(rc))
        raise CareHipError("{} failed: {}".format(name, what))


def argmax_parts(n: int, m: int = 0, bf16: bool = False) -> int:
    return load().care_argmax_parts_bf16(m, n) if bf16 else load().care_argmax_parts(n)


def relaunch_avg_us(tag: str, iters: int = 50) -> float:
    """Average duration of the last launch recorded under `tag`, re-issued `iters` times back to
    back on the current stream between two HIP events (kernels on the path are idempotent)."""
    name, args, variant = LAST_CALL[tag]
    fn = getattr(load(variant=variant), name)
    for _ in range(3):
        fn(*args, stream_ptr())
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn(*args, stream_ptr())
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1e3 / iters
