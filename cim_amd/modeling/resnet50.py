"""ResNet-50 C4 body + MaskFuse head (mirror of /root/reference/lib/modeling/resnet50.py).

The reference wraps `torchvision.models.resnet50` (resnet50.py:20-33); torchvision is not in
this image, so the architecture (ResNet-50 v1.5: conv1-bn-relu-maxpool, layer1..layer3) is
defined here with torchvision's state_dict key names so that ImageNet / reference checkpoints
load unchanged: `res1.0`=conv1, `res1.1`=bn1, `res2|res3|res4.<i>.{conv1,bn1,conv2,bn2,conv3,
bn3,downsample.0,downsample.1}`.  All BatchNorm layers stay in eval mode and res1-2 are frozen,
as resnet50.py:53-77 does.
"""
import torch
import torch.nn as nn

from ..core.config import cfg
from ..ops import conv1x1_bn_act, conv3x3_bn_act, conv7x7_bn_act, max_pool2d
from ..ops.conv3x3 import prefetch_transposed_weights
from .maskfuse import MaskFuse  # noqa: F401  (resolved as "resnet50.MaskFuse" by get_func)


# BatchNorm backward of conv1 / conv2 inside the next layer's data gradient (ops/chain.py); a module attribute so that tests can
# compare both forms.  (The library paths - MIOpen convolutions, ATen BatchNorm - are not selectable here: tools/bench_library_paths.py
# patches them in for A/B runs.)
FUSE_BN_BWD = True


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        # every convolution of the block with its BatchNorm (eval statistics) (+ identity) (+ ReLU) is ONE HIP launch
        # (small-tile fp32-MFMA GEMM - implicit for the 3 x 3 - with the chain in its epilogue, cim_amd/csrc/conv1x1.hip).
        # They fall back to the ATen ops for a BN in training mode or CPU tensors.
        identity = x
        # (fuse_input_bn: conv1's / conv2's outputs have one consumer each - their BatchNorm + ReLU backward rides in the next
        # layer's data-gradient epilogue, ops/chain.py)
        # (branch: the gradient that reaches x through the other path - the identity, or the downsample layer's data gradient (every
        # second pixel for a stride-2 one) - is added in conv1's data-gradient epilogue, ops/conv1x1.py)
        branch = {} if FUSE_BN_BWD else None
        out = conv1x1_bn_act(x, self.conv1, self.bn1, branch=branch)
        out = conv3x3_bn_act(out, self.conv2, self.bn2, fuse_input_bn=FUSE_BN_BWD)
        if self.downsample is not None:
            identity = conv1x1_bn_act(x, self.downsample[0], self.downsample[1], relu=False, branch=branch)
        return conv1x1_bn_act(out, self.conv3, self.bn3, residual=identity, fuse_input_bn=FUSE_BN_BWD, branch=branch)


def _make_layer(inplanes, planes, blocks, stride):
    downsample = None
    if stride != 1 or inplanes != planes * 4:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                   nn.BatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, downsample)]
    for _ in range(1, blocks):
        layers.append(Bottleneck(planes * 4, planes))
    return nn.Sequential(*layers)


def freeze_params(m):
    for p in m.parameters():
        p.requires_grad = False


class resnet(nn.Module):
    def __init__(self, block_counts=4):
        super().__init__()
        if block_counts != 4:
            raise AssertionError
        self.res1 = nn.Sequential(nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False),
                                  nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                                  nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        self.res2 = _make_layer(64, 64, 3, 1)
        self.res3 = _make_layer(256, 128, 4, 2)
        self.res4 = _make_layer(512, 256, 6, 2)
        self.spatial_scale = 1 / 16
        self.dim_out = 1024
        self.block_counts = block_counts
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if cfg.MODEL.LOAD_IMAGENET_PRETRAINED_WEIGHTS:          # resnet50.py:20: torchvision's ImageNet weights
            self.load_torchvision_state_dict(_torchvision_resnet50_state_dict())
            print("Load pre-trained weight for resnet !!")
        self._init_modules()

    def load_torchvision_state_dict(self, sd):
        """torchvision ResNet-50 keys -> this body's keys (conv1 -> res1.0, bn1 -> res1.1, layer<i> -> res<i+1>;
        layer4 / fc are not part of the C4 body)."""
        ren = {"conv1.": "res1.0.", "bn1.": "res1.1.", "layer1.": "res2.", "layer2.": "res3.", "layer3.": "res4."}
        own = {}
        for k, v in sd.items():
            for a, b in ren.items():
                if k.startswith(a):
                    own[b + k[len(a):]] = v
        missing, unexpected = self.load_state_dict(own, strict=False)
        assert not unexpected and all("num_batches_tracked" in m for m in missing), (missing, unexpected)

    def _init_modules(self):
        assert cfg.ResNet.FREEZE_AT in [0, 2, 3, 4, 5]
        for i in range(1, cfg.ResNet.FREEZE_AT + 1):
            freeze_params(getattr(self, "res%d" % i))
        self.freeze(self)

    def freeze(self, m):
        for k in m.modules():
            if isinstance(k, nn.BatchNorm2d):
                k.eval()

    def train(self, mode=True):
        self.training = mode
        for i in range(cfg.ResNet.FREEZE_AT + 1, self.block_counts + 1):
            getattr(self, "res%d" % i).train(mode)
        self.freeze(self)
        return self

    def detectron_weight_mapping(self):
        return {name: name for name, _ in self.named_parameters()}, []

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            # the transposed 3 x 3 weights the backward's data gradients read: one launch beside the forward (ops/conv3x3.py)
            prefetch_transposed_weights([b.conv2 for i in range(cfg.ResNet.FREEZE_AT + 1, self.block_counts + 1)
                                         for b in getattr(self, "res%d" % i)])
        for i in range(self.block_counts):
            m = getattr(self, "res%d" % (i + 1))
            if i == 0:      # the stem: 7 x 7 convolution + BatchNorm + ReLU in one launch (frozen: forward only), then the max-pool (csrc/pool.hip)
                x = max_pool2d(conv7x7_bn_act(x, m[0], m[1], relu=True), m[3])
            else:
                x = m(x)
        return x


def _torchvision_resnet50_state_dict():
    try:
        from torchvision import models
    except ImportError as e:
        raise ImportError("cfg.MODEL.LOAD_IMAGENET_PRETRAINED_WEIGHTS for the ResNet body needs torchvision "
                          "(the reference wraps torchvision.models.resnet50, resnet50.py:12,20): %s" % e)
    return models.resnet50(pretrained=True).state_dict()


def torch_resnet50():
    return resnet()
