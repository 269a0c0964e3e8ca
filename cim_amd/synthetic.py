"""Synthetic VOC/COCO-shaped inputs for the per-image training step.

Produces exactly the blobs /root/reference/lib/roi_data/minibatch.py:19-89 hands to
`Generalized_RCNN.forward` (data, rois, masks, labels, mat, index) plus the full-resolution
proposal masks from which the two N x N maps are built (the reference reads those maps from
pickles written by tools/pre/create_cob_iou.py / create_cob_asy_iou.py).

Pure NumPy, seeded with a private RandomState (never touches the global NumPy RNG, which the
mining's anti-noise sampling consumes, heads.py:459).
"""
import numpy as np

CONFIGS = {
    # name: (orig H, orig W, target longest side, N proposals, classes, positive classes)
    "vgg16_voc": dict(orig_hw=(375, 500), target=480, n=300, classes=20, n_pos=2, body="vgg16"),
    "resnet50_voc": dict(orig_hw=(375, 500), target=688, n=1000, classes=20, n_pos=2, body="resnet50"),
    "resnet50_coco2017": dict(orig_hw=(480, 640), target=688, n=2000, classes=80, n_pos=3, body="resnet50"),
    "hrnet48_coco2017": dict(orig_hw=(480, 640), target=688, n=2000, classes=80, n_pos=3, body="hrnet48"),
    "hrnet48_voc": dict(orig_hw=(375, 500), target=688, n=1000, classes=20, n_pos=2, body="hrnet48"),
}


def make_masks(n, h, w, rng, min_side=16):
    """n proposal masks [n,h,w] bool: axis-aligned box AND a random ellipse inside it, so
    that IoU and containment between proposals are non-trivial.  A quarter of the proposals
    are drawn as sub-boxes of an earlier proposal so that containment chains exist."""
    masks = np.zeros((n, h, w), dtype=bool)
    boxes = np.zeros((n, 4), dtype=np.int64)
    for i in range(n):
        if i >= 8 and rng.rand() < 0.25:
            p = boxes[rng.randint(0, i)]
            px0, py0, px1, py1 = p
            bw = rng.randint(min(min_side, px1 - px0), px1 - px0 + 1)
            bh = rng.randint(min(min_side, py1 - py0), py1 - py0 + 1)
            x0 = rng.randint(px0, px1 - bw + 1)
            y0 = rng.randint(py0, py1 - bh + 1)
        else:
            bw = rng.randint(min_side, w + 1)
            bh = rng.randint(min_side, h + 1)
            x0 = rng.randint(0, w - bw + 1)
            y0 = rng.randint(0, h - bh + 1)
        yy, xx = np.mgrid[0:bh, 0:bw]
        cy, cx = (bh - 1) / 2.0, (bw - 1) / 2.0
        ry = (0.35 + 0.4 * rng.rand()) * bh
        rx = (0.35 + 0.4 * rng.rand()) * bw
        ell = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        if not ell.any():
            ell[:] = True
        masks[i, y0:y0 + bh, x0:x0 + bw] = ell
        ys, xs = np.nonzero(masks[i])
        boxes[i] = (xs.min(), ys.min(), xs.max() + 1, ys.max() + 1)   # generate_7_7_voc.py:36
    return masks, boxes


def masks_7x7(masks, boxes, size=7):
    """Nearest-neighbour resize of each mask cropped to its box (generate_7_7_voc.py:37-39)."""
    n = masks.shape[0]
    out = np.zeros((n, size, size), dtype=np.float32)
    for i in range(n):
        x0, y0, x1, y1 = boxes[i]
        crop = masks[i, y0:y1, x0:x1]
        ry = np.minimum(((np.arange(size) + 0.5) * crop.shape[0] / size).astype(np.int64), crop.shape[0] - 1)
        rx = np.minimum(((np.arange(size) + 0.5) * crop.shape[1] / size).astype(np.int64), crop.shape[1] - 1)
        out[i] = crop[ry][:, rx]
    return out


def make_mat(masks, labels_pos, num_classes, rng):
    """PRM cluster matrix [N, C+1] in the format of tools/pre/AGPL_label_assign.py:137-185:
    at most one non-zero per row holding the cluster id; column 0 holds the background cluster."""
    n = masks.shape[0]
    mat = np.zeros((n, num_classes + 1), dtype=np.float32)
    area = masks.reshape(n, -1).sum(1).astype(np.float64)
    cluster = 1
    bg_agg = np.zeros(n, dtype=bool)
    for c in labels_pos:
        ref = masks[rng.randint(0, n)]
        inter = (masks & ref[None]).reshape(n, -1).sum(1)
        iou = inter / (area + ref.sum() - inter)
        assign = iou > 0.5
        mat[assign, :] = 0
        mat[assign, c + 1] = cluster
        bg_agg |= (iou <= 0.5) & (iou != 0)
        cluster += 1
    bg = bg_agg & (mat.sum(1) == 0)
    mat[bg, 0] = cluster
    return mat


def make_image_inputs(config="resnet50_voc", seed=3, n=None, with_image=True, target=None):
    """One synthetic training image.  Returns a dict of NumPy arrays:
    data[1,3,H,W] f32, rois[N,5] f32, masks[N,7,7] f32, labels[1,C] f32, mat[N,C+1] f32,
    index[N] i64, full_masks[N,h,w] bool, boxes[N,4] i64, im_scale.
    target = longest image side after scaling (one of cfg.TRAIN.SCALES, configs/resnet50_voc.yaml:34; default: the
    config's median scale)."""
    cfg = CONFIGS[config]
    rng = np.random.RandomState(seed)
    h, w = cfg["orig_hw"]
    n = cfg["n"] if n is None else n
    C = cfg["classes"]
    im_scale = float(target or cfg["target"]) / float(max(h, w))       # utils/blob.py:165
    H, W = int(round(h * im_scale)), int(round(w * im_scale))
    full_masks, boxes = make_masks(n, h, w, rng)
    rois = np.zeros((n, 5), dtype=np.float32)
    rois[:, 1:] = boxes.astype(np.float32) * np.float32(im_scale)       # minibatch.py:52
    pos = np.sort(rng.choice(C, size=cfg["n_pos"], replace=False))
    labels = np.zeros((1, C), dtype=np.float32)
    labels[0, pos] = 1
    out = dict(
        rois=rois,
        masks=masks_7x7(full_masks, boxes),
        labels=labels,
        mat=make_mat(full_masks, pos, C, rng),
        index=np.arange(n, dtype=np.int64),
        full_masks=full_masks,
        boxes=boxes,
        im_scale=im_scale,
        image_hw=(H, W),
    )
    if with_image:
        out["data"] = rng.randn(1, 3, H, W).astype(np.float32)
    return out


def make_scores(n, num_classes, rng):
    """Random head outputs with the shapes of cls_iou_model's (heads.py:194-219), built from
    integer permutations and one correctly-rounded division so that they are bit-identical
    on every host (no libm calls) and tie-free within each column.
    Returns (cls, det, iou) float32 [n, C+1]: cls and iou in (0,1), det in (0, 2**-6)."""
    m = n * (num_classes + 1)
    assert m < (1 << 24)
    def perm():
        return ((rng.permutation(m) + 1).astype(np.float32) / np.float32(m + 1)).reshape(n, num_classes + 1)
    cls = perm()
    det = perm() * np.float32(2.0 ** -6)
    iou = perm()
    return cls, det, iou
