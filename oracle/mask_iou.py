"""ORACLE (test infrastructure, not product code).

NumPy restatement of the offline mask-IoU / containment maps the reference trains on:
/root/reference/lib/utils/mask_utils.py:6-18 (`mask_iou`), :20-32 (`mask_asymmetric_iou`) as
driven by tools/pre/create_cob_iou.py:43-49 and tools/pre/create_cob_asy_iou.py:43-53
(one column per call, concatenated on axis 1, cast to float16).

  iou[i, j] = |m_i & m_j| / |m_i | m_j|          asy[i, j] = |m_i & m_j| / |m_j|

Rounding chain of the reference: int64 counts -> float64 divide -> float32 store -> float16.
Parity is PINNED by tests/golden/mask_iou_*.npz (captured by running mask_utils.py itself).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np


def _counts(masks):
    """masks: [N,H,W] bool -> (inter[N,N] int64, area[N] int64). float32 matmul is exact
    while every count < 2**24 (any image up to 4096x4096)."""
    n = masks.shape[0]
    m = masks.reshape(n, -1)
    assert m.shape[1] < (1 << 24)
    mf = m.astype(np.float32)
    inter = (mf @ mf.T).astype(np.int64)
    area = m.sum(axis=1).astype(np.int64)
    return inter, area


def mask_iou_maps(masks):
    """Returns (iou_f16[N,N], asy_f16[N,N]) with the reference's rounding chain."""
    inter, area = _counts(np.asarray(masks).astype(bool))
    union = area[:, None] + area[None, :] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = (inter.astype(np.float64) / union.astype(np.float64)).astype(np.float32).astype(np.float16)
        asy = (inter.astype(np.float64) / area[None, :].astype(np.float64)).astype(np.float32).astype(np.float16)
    return iou, asy


def mask_iou_maps_loops(masks):
    """Literal double loop (small N only) used to cross-check the matmul formulation."""
    masks = np.asarray(masks).astype(bool)
    n = masks.shape[0]
    iou = np.empty((n, n), dtype=np.float32)
    asy = np.empty((n, n), dtype=np.float32)
    for i in range(n):
        for j in range(n):
            inter = np.bitwise_and(masks[i], masks[j]).sum()
            union = np.bitwise_or(masks[i], masks[j]).sum()
            with np.errstate(divide="ignore", invalid="ignore"):
                iou[i, j] = inter / union
                asy[i, j] = inter / masks[j].sum()
    return iou.astype(np.float16), asy.astype(np.float16)
