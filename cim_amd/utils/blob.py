"""Network-input blobs built on the device (f-3: the data side of the step).

Same names and argument meaning as /root/reference/lib/utils/blob.py:40-169 (`get_image_blob`, `prep_im_for_blob`,
`get_target_scale`, `im_list_to_blob`) for cfg.transform_mode == "ToTensor" - the mode of every shipped config -,
but the image arrives as the BGR uint8 array `cv2.imread` returns (host NumPy or a device tensor) and the whole chain
(flip, float conversion, bilinear resize, uint8 truncation, BGR2RGB, /255, mean / std) runs as ONE HIP launch writing
straight into the NCHW blob (cim_amd/csrc/image_prep.hip); nothing but the 3 bytes per source pixel crosses PCIe.
`project_im_rois` is lib/roi_data/minibatch.py:152-155 / lib/core/test.py:476-489 (rois * im_scale, batch index column)."""
import ctypes

import numpy as np
import torch

from .. import _lib

MEAN = (0.485, 0.456, 0.406)        # lib/utils/blob.py:131-132 (RGB)
STD = (0.229, 0.224, 0.225)
_MS = (ctypes.c_float * 6)(*MEAN, *STD)


def get_target_scale(im_size_min, im_size_max, target_size, max_size):
    """lib/utils/blob.py:162-169: the longest side is scaled to target_size (the max_size cap is commented out there)."""
    return float(target_size) / float(im_size_max)


def _round_half_even(x):
    return int(np.round(x))          # cvRound / saturate_cast<int>: round half to even, like np.round


def blob_size(h, w, im_scale):
    """dsize of cv2.resize(im, None, None, fx=im_scale, fy=im_scale)."""
    return _round_half_even(h * im_scale), _round_half_even(w * im_scale)


def _as_device_u8(im, device):
    if torch.is_tensor(im):
        t = im
    else:
        t = torch.from_numpy(np.ascontiguousarray(im))
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("image must be uint8 [h, w, 3] (BGR, as cv2.imread returns it)")
    return t.to(device, non_blocking=True).contiguous()


def prep_im_for_blob(im, pixel_means, target_sizes, max_size, flag="ToTensor", hflip=False, device="cuda", out=None):
    """One image at each of `target_sizes` -> ([float32 CHW device tensors], [im_scales]) (lib/utils/blob.py:93-147;
    the reference returns HWC NumPy arrays and transposes in im_list_to_blob)."""
    if flag != "ToTensor":
        raise NotImplementedError("prep_im_for_blob: only transform_mode 'ToTensor' is on the CIM path (got %r)" % (flag,))
    src = _as_device_u8(im, torch.device(device))
    if not src.is_cuda:
        raise _lib.CimHipError("prep_im_for_blob: the HIP path needs a CUDA/HIP device (no CPU fallback)")
    h, w = int(src.shape[0]), int(src.shape[1])
    ims, scales = [], []
    for k, target in enumerate(target_sizes):
        s = get_target_scale(min(h, w), max(h, w), target, max_size)
        H, W = blob_size(h, w, s)
        dst = out[k] if out is not None else torch.empty((3, H, W), dtype=torch.float32, device=src.device)
        assert dst.shape[0] == 3 and dst.shape[1] >= H and dst.shape[2] >= W and dst.stride(2) == 1
        _lib.call("cim_image_prep", src.data_ptr(), h, w, dst.data_ptr(), H, W, dst.stride(0), dst.stride(1),
                  1.0 / s, int(bool(hflip)), ctypes.cast(_MS, ctypes.c_void_p), _lib.stream_ptr())
        ims.append(dst[:, :H, :W])
        scales.append(s)
    return ims, scales


def im_list_to_blob(ims):
    """[CHW tensors] -> zero-padded NCHW blob (lib/utils/blob.py:59-83)."""
    if not isinstance(ims, (list, tuple)):
        ims = [ims]
    H = max(int(t.shape[1]) for t in ims)
    W = max(int(t.shape[2]) for t in ims)
    blob = torch.zeros((len(ims), 3, H, W), dtype=torch.float32, device=ims[0].device)
    for i, t in enumerate(ims):
        blob[i, :, :t.shape[1], :t.shape[2]] = t
    return blob


def get_image_blob(im, target_scale, target_max_size, flag="ToTensor", hflip=False, device="cuda"):
    """lib/utils/blob.py:40-56: (blob [1,3,H,W], im_scale)."""
    ims, scales = prep_im_for_blob(im, None, [target_scale], target_max_size, flag, hflip=hflip, device=device)
    return ims[0].unsqueeze(0), scales[0]


def project_im_rois(im_rois, im_scale, batch_index=0, device="cuda"):
    """boxes [N,4] in original-image pixels -> rois blob [N,5] = (batch index, x1, y1, x2, y2) * im_scale
    (lib/roi_data/minibatch.py:44-48,152-155; lib/core/test.py:476-489)."""
    b = torch.as_tensor(im_rois, dtype=torch.float32, device=device)
    out = torch.empty((b.shape[0], 5), dtype=torch.float32, device=b.device)
    out[:, 0] = float(batch_index)
    out[:, 1:] = b * np.float32(im_scale)
    return out
