"""ORACLE (test infrastructure, not product code): ctypes front-end of oracle/roi_align_ref.c.

PARITY UNPINNED for aligned=True (mmcv is not in /root/reference; see the C file's header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def roi_align_fwd(feat, rois, P=7, scale=1.0 / 16, sampling_ratio=0, aligned=True):
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    B, C, H, W = feat.shape
    K = rois.shape[0]
    out = np.zeros((K, C, P, P), dtype=np.float32)
    rc = lib().oracle_roi_align_fwd(_p(feat, ctypes.c_float), _p(rois, ctypes.c_float), _p(out, ctypes.c_float),
                                    B, C, H, W, K, P, ctypes.c_float(scale), sampling_ratio, int(aligned))
    assert rc == 0
    return out


def roi_align_bwd(grad_out, rois, feat_shape, P=7, scale=1.0 / 16, sampling_ratio=0, aligned=True):
    """Returns grad_in as float64 (exact accumulation order-independent target)."""
    grad_out = np.ascontiguousarray(grad_out, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    B, C, H, W = feat_shape
    K = rois.shape[0]
    gin = np.zeros((B, C, H, W), dtype=np.float64)
    rc = lib().oracle_roi_align_bwd(_p(grad_out, ctypes.c_float), _p(rois, ctypes.c_float), _p(gin, ctypes.c_double),
                                    B, C, H, W, K, P, ctypes.c_float(scale), sampling_ratio, int(aligned))
    assert rc == 0
    return gin
