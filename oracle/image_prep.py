"""ORACLE (test infrastructure, not product code).

NumPy restatement of the reference's network-input chain for transform_mode "ToTensor",
/root/reference/lib/utils/blob.py:93-147 (prep_im_for_blob): float32 conversion, cv2.resize(fx = fy = scale,
INTER_LINEAR), np.uint8 truncation, BGR2RGB, torchvision ToTensor (/255) and Normalize(mean, std).

PARITY UNPINNED: cv2 and torchvision are third-party packages outside /root/reference and absent from this image
(docs/INSTALL.md installs opencv-python unpinned), so no reference-generated vectors exist.  The resize follows the
published algorithm of OpenCV 4.x modules/imgproc/src/resize.cpp for CV_32FC3 + INTER_LINEAR (resizeGeneric_ with
HResizeLinear then VResizeLinear; coefficient and border rules cited inline), every product / sum in float32 without
fused multiply-add; a build of OpenCV whose SIMD path fuses them can differ in the last bit before the truncation.
Pinned here only by closed-form cases (identity scale, constant image, integer down-scales): tests/test_oracle_misc.py.
"""
import numpy as np

f32 = np.float32
MEAN = np.array([0.485, 0.456, 0.406], dtype=f32)      # blob.py:131-132
STD = np.array([0.229, 0.224, 0.225], dtype=f32)


def _coeffs(dsize, ssize, scale):
    """resize.cpp: fx = (float)((dx + 0.5) * scale_x - 0.5); sx = cvFloor(fx); fx -= sx."""
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(f32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(f32)).astype(f32)
    return s, f


def resize_linear(im, im_scale):
    """cv2.resize(im, None, None, fx = fy = im_scale, interpolation = INTER_LINEAR) for a float32 [h, w, c] image."""
    im = np.asarray(im, dtype=f32)
    h, w = im.shape[:2]
    H, W = int(np.round(h * im_scale)), int(np.round(w * im_scale))         # saturate_cast<int>(ssize * inv_scale)
    scale = 1.0 / float(im_scale)                                           # scale_x = 1. / inv_scale_x
    sx, fx = _coeffs(W, w, scale)
    neg = sx < 0                                                            # "if (sx < 0) fx = 0, sx = 0"
    fx[neg], sx[neg] = 0, 0
    edge = sx >= w - 1                                                      # xmax: D[dx] = S[sx] for these columns
    fx[edge], sx[edge] = 0, w - 1
    sx1 = np.minimum(sx + 1, w - 1)
    a0, a1 = (f32(1) - fx)[None, :, None], fx[None, :, None]
    hor = (im[:, sx, :] * a0).astype(f32) + (im[:, sx1, :] * a1).astype(f32)   # HResizeLinear
    hor = hor.astype(f32)
    hor[:, edge, :] = im[:, w - 1, :][:, None, :]
    sy, fy = _coeffs(H, h, scale)
    y0 = np.clip(sy, 0, h - 1)                                              # row indices clipped, weights kept
    y1 = np.clip(sy + 1, 0, h - 1)
    b0, b1 = (f32(1) - fy)[:, None, None], fy[:, None, None]
    return ((hor[y0] * b0).astype(f32) + (hor[y1] * b1).astype(f32)).astype(f32)   # VResizeLinear


def prep_image(im_bgr_u8, im_scale, hflip=False):
    """-> float32 [3,H,W] (RGB, normalised), H, W = round-half-even(h * scale), (w * scale)."""
    im = im_bgr_u8[:, ::-1, :] if hflip else im_bgr_u8                      # minibatch.py:121-122 / test.py:249
    ver = resize_linear(im.astype(f32), im_scale)
    u8 = np.clip(ver, 0, 255).astype(np.uint8)                              # np.uint8(): truncation
    rgb = u8[:, :, ::-1].astype(f32)                                        # cv2.COLOR_BGR2RGB
    unit = (rgb / f32(255)).astype(f32)                                     # ToTensor
    out = ((unit - MEAN) / STD).astype(f32)                                 # Normalize
    return np.ascontiguousarray(out.transpose(2, 0, 1))
