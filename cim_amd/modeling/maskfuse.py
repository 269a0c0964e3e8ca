"""`MaskFuse` box head: ROIAlign -> (box_x, box_x * mask) -> 3x3 conv 2C->C -> FC 49C->4096 -> FC.

Mirrors MaskFuse of /root/reference/lib/modeling/resnet50.py:94-138 (vgg16.py:135-179,
HRNet.py:588-632): same constructor `(dim_in, roi_xform_func, spatial_scale)`, same parameter
names (`mask_branch.0`, `seg_fc.0`, `seg_fc.2`) and the same (c, h, w) flatten order into
`seg_fc.0`, so reference checkpoints load unchanged.

MI355X design: the ROIAlign output, the mask product and the channel concat are produced by
ONE fused HIP kernel straight into the conv input (channels-last), instead of three
materialised tensors (box_x, mask_x, cat) as in the reference.
"""
import torch
import torch.nn as nn

from ..core.config import cfg
from ..ops import maskfuse_pair, pair, roi_align_maskcat


# The box head as ONE autograd node whose first launch writes the convolution's Winograd input image straight from the feature map
# (ops/maskfuse_pair.py: MaskFuseRoiPairFunction).  False: ROIAlign (+ mask multiply + concat) and the head as two nodes with the
# `cat` tensor in between - same bits (tests/test_gpu_gemm_pair.py), kept for shapes the fused launch does not take.
FUSE_ROI_WINO = True


class MaskFuse(nn.Module):
    def __init__(self, dim_in, roi_xform_func, spatial_scale):
        super().__init__()
        self.dim_in = dim_in
        self.roi_xform = roi_xform_func
        self.spatial_scale = spatial_scale
        self.dim_out = hidden_dim = 4096
        roi_size = cfg.FAST_RCNN.ROI_XFORM_RESOLUTION
        self.mask_branch = nn.Sequential(nn.Conv2d(dim_in * 2, dim_in, kernel_size=3, padding=1), nn.ReLU())
        self.seg_fc = nn.Sequential(nn.Linear(dim_in * roi_size ** 2, hidden_dim), nn.ReLU(),
                                    nn.Linear(hidden_dim, hidden_dim), nn.ReLU())

    def detectron_weight_mapping(self):
        return {name: name for name, _ in self.named_parameters()}, []

    _told_big_map = False

    def prefetch(self):
        """Weight-only work of the forward, launched ahead on the side stream (called before the backbone forward)."""
        if cfg.FAST_RCNN.ROI_XFORM_RESOLUTION == 7:
            maskfuse_pair.prefetch_weight_images(self.mask_branch[0].weight, self.seg_fc[0].weight, self.seg_fc[2].weight)

    @staticmethod
    def _cat_amax(x, masks):
        """int32[2] bit patterns (max |x|, max |mask|): the scale source of the convolution's input image.  ROIAlign averages feature
        pixels and the masks multiply them: max |cat| <= max |x| max(1, max |mask|) - two small passes instead of one over cat."""
        xd, md = x.detach(), masks.detach().to(torch.float32)
        if not (xd.is_contiguous() or xd.is_contiguous(memory_format=torch.channels_last)):
            xd = xd.contiguous()
        fa = torch.zeros(2, dtype=torch.int32, device=x.device)
        pair.amax_of(xd, out=fa[0:1])
        pair.amax_of(md.contiguous(), out=fa[1:2])
        return fa

    def forward(self, x, rois, masks):
        method = cfg.FAST_RCNN.ROI_XFORM_METHOD
        if method != "RoIAlign":
            raise NotImplementedError("MaskFuse: only ROI_XFORM_METHOD=RoIAlign is on the CIM path (got %s)" % method)
        # the nn.Conv2d / nn.Linear modules only hold the parameters (reference names and layouts)
        conv = self.mask_branch[0]
        fc1, fc2 = self.seg_fc[0], self.seg_fc[2]
        res, sr = cfg.FAST_RCNN.ROI_XFORM_RESOLUTION, cfg.FAST_RCNN.ROI_XFORM_SAMPLING_RATIO
        import sys
        if (FUSE_ROI_WINO and not sys.modules["cim_amd.ops.roi_align"].EXACT
                and maskfuse_pair.roi_supported(x, conv.weight, fc1.weight, fc2.weight, res)):
            # ONE autograd node for the whole box head: ROIAlign + mask multiply + concat + the convolution's Winograd input
            # transform in one launch (the conv input `cat` is never stored), then conv -> flatten -> fc1 -> fc2 on pair images.
            # ROIAlign averages feature pixels and the masks are {0, 1}: max |cat| <= max |x| max(1, max |mask|)
            return maskfuse_pair.maskfuse_roi_head(x, rois, masks.detach(), conv, fc1, fc2, self._cat_amax(x, masks), self.spatial_scale, sr)
        if FUSE_ROI_WINO and x.is_cuda and (x.shape[2] > 128 or x.shape[3] > 128) and not MaskFuse._told_big_map:
            # NOT a library fallback (both paths are this package's HIP kernels) - but the numbers quoted for the fused launch
            # (33 x 43 maps) do not carry over to maps above 128 rows / columns: say so once
            MaskFuse._told_big_map = True
            import logging
            logging.getLogger("cim_amd.maskfuse").warning(
                "feature map %d x %d exceeds the fused ROIAlign -> Winograd launch's 128 x 128 table limit: the two-launch path "
                "(roi_align_maskcat + wino7_input_pair, `cat` materialised) runs for such images", x.shape[2], x.shape[3])
        cat = roi_align_maskcat(x, rois, masks, res, self.spatial_scale, sr, aligned=True)
        if not maskfuse_pair.supported(cat, conv.weight, fc1.weight, fc2.weight):
            # ONE engine, ONE algorithm: every configuration of the reference (512 / 1024 / 2048 feature channels, 7 x 7 ROI maps,
            # 4096-wide fully connected layers) qualifies; anything else fails loudly instead of taking a second code path
            raise NotImplementedError(
                "MaskFuse: the MI355X contraction engine takes 7 x 7 ROI maps, dim_in a multiple of 16 (conv output channels a multiple "
                "of 64) and fully connected widths that are multiples of 32 - got cat %s, conv %s, fc %s / %s"
                % (tuple(cat.shape), tuple(conv.weight.shape), tuple(fc1.weight.shape), tuple(fc2.weight.shape)))
        # conv -> flatten -> fc1 -> fc2 on pair images (one scale per matrix).  ROIAlign averages feature pixels and the masks are
        # {0, 1}: max |cat| <= max |x| max(1, max |mask|) - a 6 MB pass instead of one over cat
        return maskfuse_pair.maskfuse_head(cat, conv, fc1, fc2, self._cat_amax(x, masks))
