"""HRNet-W48 classification trunk -> one 2048-channel stride-32 map, + MaskFuse head
(mirror of /root/reference/lib/modeling/HRNet.py:257-586 `HighResolutionNet`, :588-632 `MaskFuse`,
`get_HRNet`).  Same module tree, hence the same state_dict keys as the reference / the public
HRNet ImageNet checkpoints: conv1, bn1, conv2, bn2, layer1, transition{1,2,3}, stage{2,3,4}
(.branches / .fuse_layers), incre_modules, downsamp_modules, final_layer, classifier (unused in
forward, HRNet.py:569-586).  Behaviour kept: input zero-padded right/bottom to a multiple of 32
(HRNet.py:501-513), every BatchNorm in eval mode, stem + layer1 (+ stage2 for FREEZE_AT = 2) frozen
and evaluated under no_grad (HRNet.py:322-336,516-535).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..core.config import cfg
from ..ops import bn_act, conv1x1_bn_act, conv3x3_bn_act, upsample_nearest
from .maskfuse import MaskFuse  # noqa: F401  (resolved as "HRNet.MaskFuse" by get_func)

BN_MOMENTUM = 0.1

# HRNet-W48 (configs/hrnet48_*.yaml MODEL.EXTRA); used when cfg.MODEL.EXTRA does not carry the stages
W48 = dict(
    STAGE1=dict(NUM_MODULES=1, NUM_BRANCHES=1, BLOCK="BOTTLENECK", NUM_BLOCKS=[4], NUM_CHANNELS=[64]),
    STAGE2=dict(NUM_MODULES=1, NUM_BRANCHES=2, BLOCK="BASIC", NUM_BLOCKS=[4, 4], NUM_CHANNELS=[48, 96]),
    STAGE3=dict(NUM_MODULES=4, NUM_BRANCHES=3, BLOCK="BASIC", NUM_BLOCKS=[4, 4, 4], NUM_CHANNELS=[48, 96, 192]),
    STAGE4=dict(NUM_MODULES=3, NUM_BRANCHES=4, BLOCK="BASIC", NUM_BLOCKS=[4, 4, 4, 4], NUM_CHANNELS=[48, 96, 192, 384]),
)


def _bn(c):
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


def _conv_bn(cin, cout, k, stride, relu, bias=False):
    layers = [nn.Conv2d(cin, cout, k, stride, k // 2, bias=bias), _bn(cout)]
    if relu is not None:
        layers.append(nn.ReLU(inplace=relu))
    return nn.Sequential(*layers)


def _downsample(ds, x):
    """conv -> BN projection shortcut (nn.Sequential of exactly those two)."""
    if len(ds) == 2 and isinstance(ds[1], nn.BatchNorm2d):
        if ds[0].kernel_size == (1, 1):      # 1 x 1 projection: small-tile GEMM with the BatchNorm in its epilogue
            return conv1x1_bn_act(x, ds[0], ds[1], relu=False)
        return conv3x3_bn_act(x, ds[0], ds[1], relu=False)      # (falls back to ATen + bn_act for other shapes)
    return ds(x)


def _up(seq, x):
    """A fuse layer's up path, nn.Sequential(1 x 1 conv, BN, nn.Upsample(nearest)) (HRNet.py:195-201): the projection with its
    BatchNorm as one small-tile GEMM launch, the up-sampling on csrc/pool.hip."""
    if len(seq) == 3 and isinstance(seq[0], nn.Conv2d) and seq[0].kernel_size == (1, 1) and isinstance(seq[1], nn.BatchNorm2d) \
            and isinstance(seq[2], nn.Upsample):
        return upsample_nearest(conv1x1_bn_act(x, seq[0], seq[1], relu=False), seq[2])
    return seq(x)


def _is_conv_bn(m):
    return isinstance(m, nn.Sequential) and len(m) in (2, 3) and isinstance(m[0], nn.Conv2d) and isinstance(m[1], nn.BatchNorm2d) \
        and m[0].kernel_size == (3, 3) and (len(m) == 2 or isinstance(m[2], nn.ReLU))


def _conv_bn_steps(seq, x):
    """A fuse / transition path: one `_conv_bn` (conv -> BN [-> ReLU]) or an nn.Sequential of them, each as one fused launch."""
    if _is_conv_bn(seq):
        return conv3x3_bn_act(x, seq[0], seq[1], relu=len(seq) == 3)
    for step in seq:
        x = conv3x3_bn_act(x, step[0], step[1], relu=len(step) == 3) if _is_conv_bn(step) else step(x)
    return x


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = _bn(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = _bn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):       # 3 x 3 convolution + BN (+ residual) + ReLU: one HIP launch each (csrc/conv1x1.hip) when the BN is in eval()
        out = conv3x3_bn_act(x, self.conv1, self.bn1)      # (one consumer: its BatchNorm + ReLU backward rides in conv2's data gradient, ops/chain.py)
        res = x if self.downsample is None else _downsample(self.downsample, x)
        return conv3x3_bn_act(out, self.conv2, self.bn2, residual=res, fuse_input_bn=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _bn(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):       # 1 x 1 convolutions + BatchNorm (+ identity) (+ ReLU): one HIP launch each (csrc/conv1x1.hip)
        out = conv1x1_bn_act(x, self.conv1, self.bn1)
        out = conv3x3_bn_act(out, self.conv2, self.bn2, fuse_input_bn=True)
        res = x if self.downsample is None else _downsample(self.downsample, x)
        return conv1x1_bn_act(out, self.conv3, self.bn3, residual=res, fuse_input_bn=True)


BLOCKS = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def _make_layer(block, inplanes, planes, blocks, stride=1):
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
                                   _bn(planes * block.expansion))
    layers = [block(inplanes, planes, stride, downsample)]
    layers += [block(planes * block.expansion, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class HighResolutionModule(nn.Module):
    """Parallel branches + all-to-all fusion (down: strided 3x3 convs, up: 1x1 conv + nearest upsample)."""

    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels, multi_scale_output=True):
        super().__init__()
        self.num_branches = num_branches
        self.branches = nn.ModuleList()
        out_ch = []
        for b in range(num_branches):
            self.branches.append(_make_layer(block, num_inchannels[b], num_channels[b], num_blocks[b]))
            out_ch.append(num_channels[b] * block.expansion)
        self.num_inchannels = out_ch
        self.fuse_layers = None
        if num_branches > 1:
            rows = []
            for i in range(num_branches if multi_scale_output else 1):
                row = []
                for j in range(num_branches):
                    if j > i:
                        row.append(nn.Sequential(nn.Conv2d(out_ch[j], out_ch[i], 1, 1, 0, bias=False), _bn(out_ch[i]),
                                                 nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                    elif j == i:
                        row.append(None)
                    else:
                        steps = []
                        for k in range(i - j):
                            last = k == i - j - 1
                            steps.append(_conv_bn(out_ch[j], out_ch[i] if last else out_ch[j], 3, 2, None if last else False))
                        row.append(nn.Sequential(*steps))
                rows.append(nn.ModuleList(row))
            self.fuse_layers = nn.ModuleList(rows)
        self.relu = nn.ReLU(False)

    def forward(self, x):
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        x = [self.branches[i](x[i]) for i in range(self.num_branches)]
        fused = []
        for i, row in enumerate(self.fuse_layers):
            y = x[0] if i == 0 else _conv_bn_steps(row[0], x[0])
            for j in range(1, self.num_branches):
                y = y + (x[j] if i == j else (_conv_bn_steps(row[j], x[j]) if j < i else _up(row[j], x[j])))
            fused.append(self.relu(y))
        return fused


def freeze_params(m):
    for p in m.parameters():
        p.requires_grad = False


class HighResolutionNet(nn.Module):
    def __init__(self):
        super().__init__()
        extra = cfg.MODEL.get("EXTRA", {}) or {}
        stage = lambda n: dict(extra[n]) if n in extra else W48[n]
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = _bn(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = _bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.stage1_cfg = s1 = stage("STAGE1")
        b1 = BLOCKS[s1["BLOCK"]]
        self.layer1 = _make_layer(b1, 64, s1["NUM_CHANNELS"][0], s1["NUM_BLOCKS"][0])
        pre = [b1.expansion * s1["NUM_CHANNELS"][0]]
        for idx in (2, 3, 4):
            sc = stage("STAGE%d" % idx)
            setattr(self, "stage%d_cfg" % idx, sc)
            blk = BLOCKS[sc["BLOCK"]]
            chans = [c * blk.expansion for c in sc["NUM_CHANNELS"]]
            setattr(self, "transition%d" % (idx - 1), self._make_transition(pre, chans))
            mods, inch = [], chans
            for _ in range(sc["NUM_MODULES"]):
                m = HighResolutionModule(sc["NUM_BRANCHES"], blk, sc["NUM_BLOCKS"], inch, sc["NUM_CHANNELS"], True)
                inch = m.num_inchannels
                mods.append(m)
            setattr(self, "stage%d" % idx, nn.Sequential(*mods))
            pre = inch
        # classification head reused as the fusion into ONE 2048-channel, stride-32 map
        head_channels = [32, 64, 128, 256]
        self.incre_modules = nn.ModuleList([_make_layer(Bottleneck, c, head_channels[i], 1) for i, c in enumerate(pre)])
        self.downsamp_modules = nn.ModuleList([
            nn.Sequential(nn.Conv2d(head_channels[i] * 4, head_channels[i + 1] * 4, 3, 2, 1), _bn(head_channels[i + 1] * 4),
                          nn.ReLU(inplace=True)) for i in range(len(pre) - 1)])
        self.final_layer = nn.Sequential(nn.Conv2d(head_channels[3] * 4, 2048, 1, 1, 0), _bn(2048), nn.ReLU(inplace=True))
        self.classifier = nn.Linear(2048, 1000)
        self.FREEZE_AT = cfg.HRNET.FREEZE_AT
        assert self.FREEZE_AT <= 2
        self.spatial_scale = 1 / 32
        self.dim_out = 2048
        self._freeze()

    @staticmethod
    def _make_transition(pre, cur):
        layers = []
        for i, c in enumerate(cur):
            if i < len(pre):
                layers.append(_conv_bn(pre[i], c, 3, 1, True) if c != pre[i] else None)
            else:
                steps = []
                for j in range(i + 1 - len(pre)):
                    steps.append(_conv_bn(pre[-1], c if j == i - len(pre) else pre[-1], 3, 2, True))
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    def _freeze(self):
        if self.FREEZE_AT >= 1:
            for name in ("conv1", "conv2", "layer1"):
                freeze_params(getattr(self, name))
        for i in range(2, self.FREEZE_AT + 1):
            freeze_params(getattr(self, "stage%d" % i))
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def train(self, mode=True):
        self.training = mode
        for child in self.children():
            child.train(mode)
        self._freeze()
        return self

    def detectron_weight_mapping(self):
        return {name: name for name, _ in self.named_parameters()}, []

    def _stem(self, x):
        x = conv3x3_bn_act(x, self.conv1, self.bn1)          # (3 input channels: forward only - the stem is frozen)
        x = conv3x3_bn_act(x, self.conv2, self.bn2)
        return self.layer1(x)

    def _stage(self, idx, ys):
        sc = getattr(self, "stage%d_cfg" % idx)
        tr = getattr(self, "transition%d" % (idx - 1))
        xs = []
        for i in range(sc["NUM_BRANCHES"]):
            if tr[i] is not None:
                xs.append(_conv_bn_steps(tr[i], ys[-1] if idx > 2 else ys[0]))
            else:
                xs.append(ys[i])
        return getattr(self, "stage%d" % idx)(xs)

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            # the transposed 3 x 3 weights the backward's data gradients read: a few launches beside the forward (ops/conv3x3.py)
            from ..ops.conv3x3 import prefetch_transposed_weights
            prefetch_transposed_weights([m for m in self.modules() if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3)
                                         and m.weight.requires_grad and m.stride == (1, 1)])
        h, w = x.shape[-2:]
        x = F.pad(x, [0, (32 - w % 32) % 32, 0, (32 - h % 32) % 32], mode="constant", value=0)
        with torch.set_grad_enabled(torch.is_grad_enabled() and self.FREEZE_AT < 1):
            x = self._stem(x)
        with torch.set_grad_enabled(torch.is_grad_enabled() and self.FREEZE_AT < 2):
            ys = self._stage(2, [x])
        ys = self._stage(3, ys)
        ys = self._stage(4, ys)
        y = self.incre_modules[0](ys[0])
        for i, down in enumerate(self.downsamp_modules):           # conv3x3 / 2 (with a bias) -> BN -> ReLU: one launch
            y = self.incre_modules[i + 1](ys[i + 1]) + _conv_bn_steps(down, y)
        fl = self.final_layer                                        # conv1x1 (with a bias) -> BN -> ReLU
        return conv1x1_bn_act(y, fl[0], fl[1], relu=True)


def get_HRNet():
    return HighResolutionNet()
