"""ctypes binding of experiments/libcim_exp.so (experiments/include/cim_exp.h): the superseded engines.  Test infrastructure."""
import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_void_p

from cim_amd._lib import CimHipError, ptr, stream_ptr      # noqa: F401  (same helpers as the product binding)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcim_exp.so")
_P = c_void_p
SIGNATURES = {
    "cim_gemm_f32_splits": [c_int, c_int, c_int, c_int],
    "cim_gemm_f32": [_P, _P, _P, _P] + [c_int] * 6 + [c_int, c_int, c_int, c_int, _P, c_int, _P],
    "cim_conv3x3_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "cim_conv3x3_wgrad_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "cim_gemm_f32_batched": [_P, _P, _P] + [c_int] * 6 + [c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_int, _P],
    "cim_gemm_f16x2_splits": [c_int, c_int, c_int],
    "cim_amax_rowcol": [_P, c_int, c_int, c_int, c_int, c_longlong, _P, _P, _P],
    "cim_gemm_f16x2": [_P, _P, _P, _P] + [c_int] * 6 + [c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "cim_gemm_f16x2_batched": [_P, _P, _P] + [c_int] * 6 + [c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, _P, _P, _P],
    "cim_wino_input_transform": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_flatten_chw": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_wino_input_transform_amax": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_wino_scale_bounds": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_wino_dy_adjoint_transform": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_wino_dx_adjoint_output": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_wino_filter_transform": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_wino_output_transform": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "cim_wino_dy_transform": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_wino_wgrad_output": [_P, _P, c_int, c_int, c_int, _P],
}
VALUE_RETURNING = {"cim_gemm_f32_splits", "cim_gemm_f16x2_splits"}
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (PyTorch's HIP runtime first, as in cim_amd/_lib.py)
    if not os.path.exists(LIB_PATH):
        from . import build
        build.build()
    lib = ctypes.CDLL(LIB_PATH)
    lib.cim_last_error.restype = ctypes.c_char_p
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in VALUE_RETURNING:
        return rc
    if rc != 0:
        raise CimHipError("%s failed (rc=%d): %s" % (name, rc, lib.cim_last_error().decode()))
