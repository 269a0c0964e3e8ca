"""Dilated VGG-16 conv5 body (stride 8) + MaskFuse head
(mirror of /root/reference/lib/modeling/vgg16.py:34-132,135-179; same parameter names
`conv{1..5}.{0,2,4}.{weight,bias}`; conv1-2 frozen for VGG.FREEZE_AT=2)."""
import torch.nn as nn

from ..core.config import cfg
from ..ops import conv3x3_bias_act, max_pool2d
from .maskfuse import MaskFuse  # noqa: F401  (resolved as "vgg16.MaskFuse" by get_func)

# (out_channels per conv, trailing max-pool, dilation)
_STAGES = [([64, 64], True, 1), ([128, 128], True, 1), ([256, 256, 256], True, 1),
           ([512, 512, 512], False, 1), ([512, 512, 512], False, 2)]


def freeze_params(m):
    for p in m.parameters():
        p.requires_grad = False


class dilated_conv5_body(nn.Module):
    def __init__(self):
        super().__init__()
        cin = 3
        for i, (widths, pool, dil) in enumerate(_STAGES, start=1):
            layers = []
            for w in widths:
                layers += [nn.Conv2d(cin, w, kernel_size=3, stride=1, padding=dil, dilation=dil, bias=True),
                           nn.ReLU(inplace=True)]
                cin = w
            if pool:
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            setattr(self, "conv%d" % i, nn.Sequential(*layers))
        self.dim_out = 512
        self.spatial_scale = 1.0 / 8.0
        assert cfg.VGG.FREEZE_AT in [0, 2, 3, 4, 5]
        for i in range(1, cfg.VGG.FREEZE_AT + 1):
            freeze_params(getattr(self, "conv%d" % i))

    def detectron_weight_mapping(self):
        return ({name: name.replace(".", "_").replace("_weight", "_w").replace("_bias", "_b")
                 for name, _ in self.named_parameters()}, [])

    def train(self, mode=True):
        self.training = mode
        for i in range(cfg.VGG.FREEZE_AT + 1, 6):
            getattr(self, "conv%d" % i).train(mode)
        return self

    def forward(self, x):
        # every Conv2d + ReLU pair is one launch of the implicit-GEMM kernel (bias + ReLU in its epilogue, dilation 2 in conv5;
        # cim_amd/csrc/conv1x1.hip), every max-pool one launch of csrc/pool.hip.  CPU tensors take the modules as they are.
        for i in range(1, 6):
            seq = getattr(self, "conv%d" % i)
            if not x.is_cuda:
                x = seq(x)
                continue
            mods = list(seq)
            k = 0
            while k < len(mods):
                m = mods[k]
                if isinstance(m, nn.Conv2d) and k + 1 < len(mods) and isinstance(mods[k + 1], nn.ReLU):
                    x = conv3x3_bias_act(x, m, relu=True)
                    k += 2
                elif isinstance(m, nn.MaxPool2d):
                    x = max_pool2d(x, m)
                    k += 1
                else:
                    x = m(x)
                    k += 1
        return x
