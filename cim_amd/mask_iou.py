"""On-device construction of the N x N mask-IoU and containment maps from proposal masks.

Replaces the offline cupy loops of /root/reference/tools/pre/create_cob_iou.py:43-49 and
create_cob_asy_iou.py:43-53 (lib/utils/mask_utils.py:6-32) and the per-step pickle.load + H2D
of lib/modeling/model_builder.py:147-159.  Bit-identical float16 maps.
"""
import torch

from . import _lib


def pack_masks(masks):
    """masks [N,H,W] (bool/uint8, CUDA) -> word-major bit-packed int64 tensor [words, N]."""
    if not masks.is_cuda:
        raise _lib.CimHipError("pack_masks: CUDA/HIP tensor expected (no CPU fallback)")
    n = masks.shape[0]
    hw = int(masks[0].numel())
    m = masks.reshape(n, hw)
    if m.dtype == torch.bool:
        m = m.contiguous().view(torch.uint8)                 # same bytes: no conversion pass over the 187 MB of cfg2's masks
    elif m.dtype != torch.uint8:
        m = (m != 0).to(torch.uint8).contiguous()
    else:
        m = m.contiguous()
    words = (hw + 63) // 64
    packed = torch.empty((words, n), dtype=torch.int64, device=masks.device)
    _lib.call("cim_mask_pack", m.data_ptr(), packed.data_ptr(), n, hw, _lib.stream_ptr())
    return packed


def maps_from_packed(packed):
    """packed [words,N] int64 -> (iou_f16 [N,N], asy_f16 [N,N], area [N] int32)."""
    words, n = packed.shape
    dev = packed.device
    area = torch.empty(n, dtype=torch.int32, device=dev)
    iou = torch.empty((n, n), dtype=torch.float16, device=dev)
    asy = torch.empty((n, n), dtype=torch.float16, device=dev)
    _lib.call("cim_mask_iou_pair", packed.data_ptr(), n, words, area.data_ptr(), iou.data_ptr(), asy.data_ptr(),
              _lib.stream_ptr())
    return iou, asy, area


def mask_iou_maps(masks):
    """masks [N,H,W] -> (iou_map, asy_iou_map) float16 [N,N], as the reference's pickles hold."""
    iou, asy, _ = maps_from_packed(pack_masks(masks))
    return iou, asy
