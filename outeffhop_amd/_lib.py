"""ctypes binding of liboeh_hip.so (C ABI: include/oeh.h).

The HIP library IS the product path: if it cannot be loaded this module raises - there is no CPU or
PyTorch fallback behind any op in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OEH_LIB: explicit path of another build of the same library (A/B timing of compiler flags; tools/ only)
LIB_PATH = os.environ.get("OEH_LIB") or os.path.join(_HERE, "lib", "liboeh_hip.so")

ABI_VERSION = 6  # include/oeh.h: OEH_ABI_VERSION
CALIB_WORK_BYTES = 36864  # include/oeh.h: OEH_CALIB_WORK_BYTES
OEH_F16, OEH_BF16, OEH_F32, OEH_I8 = 0, 1, 2, 3
OEH_SOFTMAX_VANILLA, OEH_SOFTMAX_ONE = 0, 1


class OehError(RuntimeError):
    code = 0  # the negative OEH_E* code when the error came from the library


class oeh_fq(C.Structure):
    _fields_ = [
        ("enable", C.c_int32),
        ("scale", C.c_float),
        ("zero_point", C.c_float),
        ("qmax", C.c_float),
        ("dump_idx", C.c_void_p),
    ]


class oeh_fq_desc(C.Structure):
    _fields_ = [("scores", oeh_fq), ("probs", oeh_fq), ("ctx", oeh_fq), ("ctx_quant_before_gate", C.c_int32), ("ctx_emit_index", C.c_int32)]


class oeh_grid(C.Structure):
    _fields_ = [("scale", C.c_float), ("zero_point", C.c_float)]


class oeh_attn_desc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("Sq", C.c_int32), ("Sk", C.c_int32), ("D", C.c_int32),
        ("dtype", C.c_int32),
        ("q_stride", C.c_int64 * 3), ("k_stride", C.c_int64 * 3), ("v_stride", C.c_int64 * 3), ("o_stride", C.c_int64 * 3),
        ("scale", C.c_float), ("scale_div", C.c_float),
        ("softmax_base", C.c_int32), ("clip", C.c_int32), ("gamma", C.c_float), ("eta", C.c_float),
        ("key_pad_mask", C.c_void_p), ("key_pad_dtype", C.c_int32), ("key_pad_stride", C.c_int64),
        ("full_mask", C.c_void_p), ("full_mask_dtype", C.c_int32), ("full_mask_stride", C.c_int64 * 2),
        ("causal", C.c_int32), ("clamp_min", C.c_int32), ("mask_min", C.c_float),
        ("gate", C.c_void_p), ("gate_stride", C.c_int64 * 3),
        ("gate_hidden", C.c_void_p), ("gate_hidden_stride", C.c_int64 * 2),
        ("gate_w1", C.c_void_p), ("gate_b1", C.c_void_p), ("gate_w2", C.c_void_p), ("gate_b2", C.c_void_p),
        ("gate_units", C.c_int32), ("gate_scaling", C.c_float), ("gate_out", C.c_void_p),
        ("q_grid", oeh_grid), ("k_grid", oeh_grid), ("v_grid", oeh_grid), ("o_dtype", C.c_int32),
        ("key_pad_boolean", C.c_int32),
    ]


class oeh_proj_seg(C.Structure):
    """include/oeh.h: one column segment (a projection) of oeh_proj_quant_i8."""
    _fields_ = [("alpha", C.c_float), ("scale", C.c_float), ("zero_point", C.c_float), ("out", C.c_void_p), ("y", C.c_void_p),
                ("y_stride_row", C.c_int64), ("transpose", C.c_int32), ("acc_add", C.c_void_p)]


# every symbol include/oeh.h declares (tests/test_abi.py checks the .so exports exactly these)
EXPORTS = (
    "oeh_attn_fwd", "oeh_softmax_rows", "oeh_fake_quant", "oeh_gate_fwd", "oeh_minmax", "oeh_percentile_ema", "oeh_fake_quant_range", "oeh_attn_calibrate", "oeh_quantize_heads_i8", "oeh_split_pairs", "oeh_split_triples", "oeh_proj_quant_i8",
    "oeh_abi_version", "oeh_build_info", "oeh_strerror", "oeh_attn_variant",
)

_lib = None


def load() -> C.CDLL:
    """Load liboeh_hip.so once.  torch (if it is going to be used) must be imported first so that both
    share ONE HIP runtime: the library's DT_NEEDED libamdhip64.so.7 then resolves to the copy torch loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OehError(
            f"{LIB_PATH} is missing: build it with `make -C outeffhop_amd/csrc -j8` (or `python -c 'import "
            "__graft_entry__ as g; g.build()'`).  outeffhop_amd has no CPU/PyTorch fallback."
        )
    try:
        import torch  # noqa: F401  (brings in torch's libamdhip64 first)
    except Exception:  # pragma: no cover - library is usable from plain C hosts too
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.oeh_attn_fwd.argtypes = [C.POINTER(oeh_attn_desc), vp, vp, vp, vp, C.POINTER(oeh_fq_desc), vp]
    lib.oeh_attn_fwd.restype = C.c_int
    lib.oeh_softmax_rows.argtypes = [vp, vp, i64, i32, i32, i32, i32, f32, f32, vp]
    lib.oeh_softmax_rows.restype = C.c_int
    lib.oeh_fake_quant.argtypes = [vp, vp, vp, i64, i32, f32, f32, f32, vp]
    lib.oeh_fake_quant.restype = C.c_int
    lib.oeh_gate_fwd.argtypes = [vp, i32, i32, i32, i32, i32, i64, i64, vp, vp, vp, vp, i32, i32, f32, vp, vp]
    lib.oeh_gate_fwd.restype = C.c_int
    lib.oeh_minmax.argtypes = [vp, i64, i32, vp, vp]
    lib.oeh_minmax.restype = C.c_int
    f64 = C.c_double
    lib.oeh_percentile_ema.argtypes = [vp, i64, i32, f64, f64, f64, i32, vp, vp, vp]
    lib.oeh_percentile_ema.restype = C.c_int
    lib.oeh_fake_quant_range.argtypes = [vp, vp, i64, i32, vp, i32, f64, vp]
    lib.oeh_fake_quant_range.restype = C.c_int
    lib.oeh_attn_calibrate.argtypes = [C.POINTER(oeh_attn_desc), vp, vp, vp, vp, i32, vp, vp, i32, f64, f64, f64, f64, i32, vp, vp, vp]
    lib.oeh_attn_calibrate.restype = C.c_int
    lib.oeh_quantize_heads_i8.argtypes = [vp, vp, vp, i64, i32, i32, C.POINTER(i64), C.POINTER(i64), i32, f32, f32, i32, f32, vp, vp]
    lib.oeh_quantize_heads_i8.restype = C.c_int
    lib.oeh_split_pairs.argtypes = [vp, vp, i64, i32, i64, vp]
    lib.oeh_split_pairs.restype = C.c_int
    lib.oeh_split_triples.argtypes = [vp, vp, i64, i32, i64, vp]
    lib.oeh_split_triples.restype = C.c_int
    lib.oeh_proj_quant_i8.argtypes = [vp, i32, vp, vp, i64, i32, i32, i32, i32, C.POINTER(oeh_proj_seg), i64, i64, vp]
    lib.oeh_proj_quant_i8.restype = C.c_int
    lib.oeh_abi_version.restype = C.c_int
    lib.oeh_build_info.restype = C.c_char_p
    lib.oeh_strerror.argtypes = [C.c_int]
    lib.oeh_strerror.restype = C.c_char_p
    lib.oeh_attn_variant.argtypes = [C.POINTER(oeh_attn_desc), C.POINTER(oeh_fq_desc)]
    lib.oeh_attn_variant.restype = C.c_char_p
    if lib.oeh_abi_version() != ABI_VERSION:
        raise OehError(f"liboeh_hip.so ABI {lib.oeh_abi_version()} != {ABI_VERSION} (stale build?)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        err = OehError(f"{what} failed: {load().oeh_strerror(rc).decode()} ({rc})")
        err.code = rc
        raise err
