"""ctypes binding of libcim_hip.so (include/cim_hip.h).  There is NO fallback: if the HIP
library is missing or a call fails, this raises."""
import ctypes
import os

from ctypes import c_float, c_int, c_longlong, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# CIM_HIP_LIB: an ablation build (python -m cim_amd.build --out=...) for whole-step A/B runs; must export the same ABI
LIB_PATH = os.environ.get("CIM_HIP_LIB") or os.path.join(HERE, "libcim_hip.so")

# name -> argtypes (all return int)
_P = c_void_p
SIGNATURES = {
    "cim_roi_align_fwd": [_P, _P, _P] + [c_int] * 6 + [c_float, c_int, c_int, _P],
    "cim_roi_align_fwd_ws": [_P, _P, _P] + [c_int] * 6 + [c_float, c_int, c_int, _P, _P],
    "cim_roi_align_maskcat_fwd_ws": [_P, _P, _P, _P] + [c_int] * 6 + [c_float, c_int, c_int, _P, _P],
    "cim_roi_align_wino7_pair_fwd": [_P, _P, _P, _P, _P] + [c_int] * 7 + [c_float, c_int, c_int, _P, _P],
    "cim_roi_align_bwd_ws": [_P, _P, _P] + [c_int] * 6 + [c_float, c_int, c_int, _P, c_int, _P, _P],
    "cim_roi_align_maskcat_bwd_ws": [_P, _P, _P, _P] + [c_int] * 6 + [c_float, c_int, c_int, _P, c_int, _P, _P],
    "cim_roi_align_bwd_workspace": [c_int, c_int, c_int, c_int],
    "cim_roi_align_bwd_scratch": [c_int, c_int, c_int, c_int, c_int],
    "cim_roi_align_bwd": [_P, _P, _P] + [c_int] * 6 + [c_float, c_int, c_int, _P, _P],
    "cim_roi_align_maskcat_fwd": [_P, _P, _P, _P] + [c_int] * 6 + [c_float, c_int, c_int, _P],
    "cim_roi_align_maskcat_bwd": [_P, _P, _P, _P] + [c_int] * 6 + [c_float, c_int, c_int, _P, _P],
    "cim_maxpool2d_out_size": [c_int, c_int, c_int, c_int],
    "cim_maxpool2d_fwd": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "cim_maxpool2d_bwd": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "cim_upsample_nearest_fwd": [_P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "cim_upsample_nearest_bwd": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_image_prep": [_P, c_int, c_int, _P, c_int, c_int, c_longlong, c_int, ctypes.c_double, c_int, _P, _P],
    "cim_mask_pack": [_P, _P, c_int, c_int, _P],
    "cim_mask_iou_pair": [_P, c_int, c_int, _P, _P, _P, _P],
    "cim_asy_flag": [_P, c_int, c_float, _P, _P],
    "cim_asy_prep": [_P, c_int, _P, c_int, _P, _P, _P],
    "cim_mining_step": [_P, _P, _P],
    "cim_mining_lds_bytes": [c_int, c_int],
    "cim_mining_sync_bytes": [],
    "cim_gemm_small_splits": [c_int, c_int, c_int],
    "cim_gemm_small_f32": [_P, _P, _P] + [c_int] * 8 + [_P, _P, _P, _P, _P, c_float, _P, c_int, c_int, _P, _P],
    "cim_conv1x1_bwd_workspace": [c_int, c_int, c_int, c_int],
    "cim_conv1x1_bn_act_bwd": [_P] * 8 + [c_float, c_int] + [_P] * 5 + [c_int] * 4 + [_P, _P, _P, _P, _P, c_int] + [c_int, _P, _P, c_float] + [_P] * 4 + [c_int],
    "cim_bn_part_finish": [_P, c_int, _P],
    "cim_conv3x3_nchw_splits": [c_int] * 5,
    "cim_conv3x3_nchw_f32": [_P, _P, _P] + [c_int] * 6 + [_P, _P, _P, _P, _P, c_float, _P, c_int, c_int, _P, _P],
    "cim_conv7x7_nchw_f32": [_P, _P, _P] + [c_int] * 5 + [_P, _P, _P, _P, c_float, c_int, _P],
    "cim_conv3x3_nchw_bwd_workspace": [c_int] * 6,
    "cim_conv3x3_dx_parts": [c_int] * 3,
    "cim_conv3x3_nchw_bn_act_bwd": [_P] * 8 + [c_float, c_int] + [_P] * 5 + [c_int] * 7 + [_P, _P, _P, _P, _P, c_int] + [c_int, _P, _P, c_float] + [_P] * 4,
    "cim_conv3x3_wt_multi": [_P, c_int, _P],
    "cim_bn_act_fwd": [_P, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int, c_int, c_int, _P],
    "cim_bn_act_bwd_chunks": [c_int, c_int, c_int],
    "cim_bn_act_bwd": [_P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_sgd_multi": [_P, _P, c_int, c_float, c_int, _P],
    "cim_gemm_pair_splits": [c_int, c_int, c_int],
    "cim_gemm_pair": [_P, _P, _P, _P] + [c_int] * 6 + [c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cim_gemm_pair_batched": [_P, _P, _P] + [c_int] * 6 + [c_int, c_int, c_int, c_longlong, c_longlong, c_longlong,
                              _P, _P, c_int, c_int, c_int, _P],
    "cim_pair_scales": [_P, c_int, _P, _P, c_int, c_int, _P],
    "cim_pair_split": [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_longlong, c_longlong, _P, _P, _P],
    "cim_pair_amax": [_P, c_longlong, _P, _P],
    "cim_pair_masked_stats": [_P, _P, c_int, c_int, _P, _P, _P],
    "cim_wino7_flatten_bwd_dy_pair": [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cim_wino7_pair_scales": [_P, c_int, _P, c_int, _P, _P],
    "cim_wino7_input_pair": [_P, _P, _P, c_int, c_int, c_int, _P],
    "cim_wino7_filter_pair": [_P, _P, _P, c_int, c_int, _P],
    "cim_wino7_dy_pair": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_wino7_output_amax": [_P, _P, _P, c_int, c_int, c_int, _P, _P],
    "cim_flatten_chw_pair": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_flatten_chw_bwd_bias": [_P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cim_wino_dx_adjoint_output": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "cim_wino7_dx_maskfold": [_P, _P, _P, c_int, c_int, _P],
    "cim_wino_wgrad_output": [_P, _P, c_int, c_int, c_int, _P],
    "cim_losses_fwd": [_P, _P],
    "cim_linear_bias_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P],
    "cim_loss_finish": [_P, c_int, _P, _P],
    "cim_loss_grad_combine": [_P] * 8 + [c_int, c_int, c_int, _P],
    "cim_head_act_fwd": [_P, _P, _P, c_int, c_int, c_int, _P],
    "cim_head_act_bwd": [_P, _P, _P, _P, c_int, c_int, c_int, _P],
}

ABI_VERSION = 15         # cim_abi_version() of include/cim_hip.h this binding was written against
_lib = None


class CimHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime of this process, so
    # import torch BEFORE dlopen-ing our library (otherwise /opt/rocm's copy is loaded first and
    # the two runtimes do not share devices / streams).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise CimHipError(
            "libcim_hip.so not found at %s - build it with `python -m cim_amd.build` "
            "(there is no CPU fallback for the CIM hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.cim_last_error.restype = ctypes.c_char_p
    lib.cim_last_error.argtypes = []
    lib.cim_abi_version.restype = c_int
    if lib.cim_abi_version() != ABI_VERSION:
        raise CimHipError("%s exports ABI %d, this package binds ABI %d: rebuild with `python -m cim_amd.build`"
                          % (LIB_PATH, lib.cim_abi_version(), ABI_VERSION))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = c_longlong if name in ("cim_conv1x1_bwd_workspace", "cim_conv3x3_nchw_bwd_workspace", "cim_roi_align_bwd_workspace", "cim_roi_align_bwd_scratch", "cim_mining_lds_bytes", "cim_mining_sync_bytes") else c_int
    _lib = lib
    return lib


VALUE_RETURNING = {"cim_maxpool2d_out_size", "cim_conv3x3_dx_parts", "cim_mining_sync_bytes", "cim_conv1x1_bwd_workspace", "cim_conv3x3_nchw_bwd_workspace", "cim_conv3x3_nchw_splits", "cim_gemm_small_splits", "cim_mining_lds_bytes", "cim_bn_act_bwd_chunks", "cim_gemm_pair_splits", "cim_roi_align_bwd_workspace", "cim_roi_align_bwd_scratch"}      # return a count, not a status


# split counts / workspace sizes of the body's layers: pure functions of their integer arguments (their tuning switches are read
# once per process), asked ~100 times per training step with the step's few dozen layer shapes
PURE = {"cim_maxpool2d_out_size", "cim_conv3x3_dx_parts", "cim_gemm_small_splits", "cim_conv3x3_nchw_splits", "cim_conv1x1_bwd_workspace", "cim_conv3x3_nchw_bwd_workspace",
        "cim_bn_act_bwd_chunks", "cim_gemm_pair_splits"}
_PURE_VALUES = {}


def call(name, *args):
    if name in PURE:
        key = (name,) + args
        v = _PURE_VALUES.get(key)
        if v is None:
            v = _PURE_VALUES[key] = getattr(load(), name)(*args)
        return v
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in VALUE_RETURNING:
        return rc
    if rc != 0:
        raise CimHipError("%s failed (rc=%d): %s" % (name, rc, lib.cim_last_error().decode()))


def stream_ptr():
    """Raw hipStream_t of the calling thread's current stream on its current device (called ~120 times per training step:
    torch.cuda.current_stream().cuda_stream builds a Stream object through three Python layers, ~9 us; the raw query ~0.3 us)."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:              # (a torch build without the private query)
        return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a dense tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
