"""The per-image training minibatch built on the device (f-3: the data side of the step).

Same function names, arguments and returned blob dictionary as /root/reference/lib/roi_data/minibatch.py:10-150
(`get_minibatch_blob_names`, `get_minibatch(roidb, num_classes, flag)`): one roidb entry (image, COB proposal boxes,
7 x 7 proposal masks, PRM cluster matrix, image-level classes) -> {data, rois, masks, labels, gtrois, mat, index, path}
with the leading "image in batch" conventions of the reference (rois [N,5] with a batch-index column, labels [1,C]).

Differences by design: the image is the BGR uint8 array `cv2.imread` returns (host array or device tensor; a path is
read with cv2 when that package is present) and its whole preparation - flip, scale draw as the reference draws it
(np.random.randint on the global generator, minibatch.py:115-116), float conversion, bilinear resize, uint8 truncation,
BGR2RGB, /255, mean / std - is ONE HIP launch (`cim_amd.utils.blob`); every blob is a device tensor, so
`nn.DataParallel(minibatch=True)` has nothing left to upload."""
import numpy as np
import torch

from ..core.config import cfg
from ..utils import blob as blob_utils


def get_minibatch_blob_names(is_training=True):
    """minibatch.py:10-15."""
    return ["data", "rois", "masks", "labels", "gtrois", "mat"]


def _read_image(entry):
    im = entry["image"]
    if isinstance(im, str):
        try:
            import cv2
        except ImportError as e:
            raise ImportError("roidb['image'] is a path and cv2 is not installed: pass the decoded BGR uint8 array (%s)" % e)
        im = cv2.imread(im)
        assert im is not None, "Failed to read image '{}'".format(entry["image"])
    return im


def get_minibatch(roidb, num_classes, flag, device="cuda"):
    """minibatch.py:19-89.  Returns (blobs, True)."""
    assert len(roidb) == 1, "Single batch only"
    entry = roidb[0]
    dev = torch.device(device)
    # ---- image blob (minibatch.py:109-150): one random training scale, drawn from the global NumPy generator
    scale_ind = int(np.random.randint(0, high=len(cfg.TRAIN.SCALES), size=1)[0])
    target_size = cfg.TRAIN.SCALES[scale_ind]
    ims, im_scales = blob_utils.prep_im_for_blob(_read_image(entry), None, [target_size], cfg.TRAIN.MAX_SIZE, flag,
                                                 hflip=bool(entry.get("flipped", False)), device=dev)
    data = blob_utils.im_list_to_blob(ims)
    im_scale = im_scales[0]
    # ---- proposals (minibatch.py:92-106: at most TRAIN.BATCH_SIZE_PER_IM, a random subset beyond that)
    labels = np.asarray(entry["gt_classes"]).reshape(1, -1)
    rois = np.asarray(entry["boxes"])
    gt_rois = np.asarray(entry["gt_boxes"], dtype=np.float32).reshape(-1, 5) if "gt_boxes" in entry else np.zeros((0, 5), np.float32)
    batch_size = cfg.TRAIN.BATCH_SIZE_PER_IM if cfg.TRAIN.BATCH_SIZE_PER_IM > 0 else np.inf
    if batch_size < rois.shape[0]:
        rois = rois[np.random.permutation(rois.shape[0])[:batch_size], :]
    rois_blob = np.hstack((np.zeros((rois.shape[0], 1)), rois * im_scale))            # batch index 0 | minibatch.py:152-155
    gt = gt_rois.copy()
    gt[:, :4] = gt[:, :4] * im_scale
    gtbox_blob = np.hstack((np.zeros((gt.shape[0], 1)), gt)).astype(np.float32)
    masks = np.asarray(entry["masks"]).astype(np.float32)
    mat = np.asarray(entry["mat"])
    if cfg.DEDUP_BOXES > 0:                                                           # minibatch.py:52-61
        v = np.array([1, 1e3, 1e6, 1e9, 1e12])
        hashes = np.round(rois_blob * cfg.DEDUP_BOXES).dot(v)
        _, index, _ = np.unique(hashes, return_index=True, return_inverse=True)
        rois_blob, mat, masks = rois_blob[index, :], mat[index, :], masks[index, :, :]
    else:
        index = np.arange(rois_blob.shape[0])
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype=dt, non_blocking=True)
    blobs = dict(data=data, rois=t(rois_blob), masks=t(masks), labels=t(labels), gtrois=t(gtbox_blob), mat=t(mat),
                 index=torch.from_numpy(index).to(dev), path=entry["image"] if isinstance(entry["image"], str) else entry.get("path", ""))
    return blobs, True
