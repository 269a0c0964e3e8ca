"""-m gpu: the backbone's fused convolution + BatchNorm kernels (cim_amd/csrc/conv1x1.hip, true-fp32 MFMA small-tile GEMMs) through the
C ABI against ATen / fp64 references, their chained backward, branch hand-overs and stream scheduling."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from cim_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())

@pytest.mark.parametrize("B,cin,cout,H,W,stride,relu,res", [(1, 64, 256, 33, 43, 1, True, True), (1, 256, 64, 65, 86, 1, True, False),
                                                            (1, 512, 1024, 66, 86, 2, False, False), (2, 128, 128, 17, 23, 1, True, False),
                                                            (1, 1024, 256, 33, 43, 1, True, False), (1, 64, 64, 129, 172, 1, True, False)])
def test_conv1x1_bn_act_vs_aten(B, cin, cout, H, W, stride, relu, res):
    """Forward and all five gradients (x, weight, residual, gamma, beta) against the ATen formulation in float64."""
    from cim_amd.ops import conv1x1_bn_act
    torch.manual_seed(cin + cout)
    dev = torch.device("cuda:0")
    conv = torch.nn.Conv2d(cin, cout, 1, stride=stride, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5); bn.running_mean.uniform_(-0.3, 0.3); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
    Ho, Wo = -(-H // stride), -(-W // stride)
    r = torch.randn(B, cout, Ho, Wo, device=dev, requires_grad=True) if res else None
    y = conv1x1_bn_act(x, conv, bn, residual=r, relu=relu)
    up = torch.randn_like(y)
    y.backward(up)
    got = [y.detach(), x.grad, conv.weight.grad, bn.weight.grad, bn.bias.grad] + ([r.grad] if res else [])
    # float64 ATen reference
    c64, b64 = torch.nn.Conv2d(cin, cout, 1, stride=stride, bias=False).to(dev).double(), torch.nn.BatchNorm2d(cout).to(dev).double().eval()
    c64.weight.data.copy_(conv.weight.data); b64.load_state_dict({k: v.double() for k, v in bn.state_dict().items()})
    x64 = x.detach().double().requires_grad_(True)
    r64 = r.detach().double().requires_grad_(True) if res else None
    ref = b64(c64(x64))
    if res:
        ref = ref + r64
    if relu:
        ref = torch.relu(ref)
    ref.backward(up.double())
    want = [ref.detach(), x64.grad, c64.weight.grad, b64.weight.grad, b64.bias.grad] + ([r64.grad] if res else [])
    for name, g, w in zip(("y", "dx", "dw", "dgamma", "dbeta", "dres"), got, want):
        assert g.shape == w.shape, name
        err = float((g.double() - w).norm() / (w.norm() + 1e-30))
        assert err < 2e-6, (name, err)       # fp32 products and accumulation
    # frozen layer (no gradients requested): same values, nothing saved
    with torch.no_grad():
        y2 = conv1x1_bn_act(x.detach(), conv, bn, residual=(r.detach() if res else None), relu=relu)
    assert torch.equal(y2, y.detach())


@pytest.mark.parametrize("affine", [True, False])
@pytest.mark.parametrize("inplanes,planes,stride,H,W", [(256, 64, 1, 33, 43), (256, 128, 2, 34, 45), (64, 16, 1, 9, 7), (512, 128, 1, 65, 86),
                                                        (64, 64, 1, 17, 23), (128, 64, 2, 35, 47)])
def test_bottleneck_chained_bn_backward_is_bit_identical(dev, inplanes, planes, stride, H, W, affine, monkeypatch):
    """ops/chain.py: conv1's and conv2's BatchNorm + ReLU backward applied in the NEXT layer's data-gradient epilogue (no
    bn_act_bwd launch, lib/modeling/resnet50.py:17-44 torchvision Bottleneck) gives the same bits as the unchained backward -
    input gradient and all four weight gradients, with and without a downsample branch (stride 1 and 2, odd and even map sizes:
    the downsample layer's data gradient is added in conv1's data-gradient epilogue - at every second pixel for stride 2 - instead
    of autograd's zero-filled scatter + add); a second consumer of a chained tensor is an error, not a wrong gradient.
    affine=True is the REFERENCE's configuration (resnet50.py:59-60: statistics frozen, gamma / beta trainable): bn1's and bn2's
    affine gradients then come from the per-32-pixel partial sums the consumer's epilogue writes, finished in group order - equal
    to the separate launch up to the summation order (checked against float64 as well), bit-equal from run to run, everything else
    still bit-identical."""
    from cim_amd.modeling import resnet50
    from cim_amd.ops import conv1x1_bn_act, conv3x3_bn_act, fallback, gemm
    torch.manual_seed(inplanes + planes)
    ds = None
    if stride != 1 or inplanes != planes * 4:
        ds = torch.nn.Sequential(torch.nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), torch.nn.BatchNorm2d(planes * 4))
    blk = resnet50.Bottleneck(inplanes, planes, stride, ds).to(dev).eval()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5); m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 2.0)
                if not affine:
                    m.weight.requires_grad_(False); m.bias.requires_grad_(False)
    x0 = torch.randn(1, inplanes, H, W, device=dev)
    names = [n for n, p in blk.named_parameters() if p.requires_grad]
    up = None
    res = {}
    for flag in (True, False, "again"):
        monkeypatch.setattr(resnet50, "FUSE_BN_BWD", bool(flag))
        for p in blk.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = blk(x)
        up = torch.randn_like(y) if up is None else up
        y.backward(up)
        gemm.join_side()
        torch.cuda.synchronize()
        res[flag] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in blk.parameters() if p.requires_grad]
    assert len(res[True]) >= 5 and len(res[True]) == len(names) + 2
    for name, a, b, c in zip(["y", "dx"] + names, res[True], res[False], res["again"]):
        assert torch.equal(a, c), name                              # deterministic
        if name.startswith(("bn1.", "bn2.")):                       # partial sums in another order than the separate launch
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6, name
        else:
            assert torch.equal(a, b), name
    if affine:          # the chained affine gradients against float64 autograd of the same block
        blk64 = resnet50.Bottleneck(inplanes, planes, stride, None if ds is None else torch.nn.Sequential(
            torch.nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), torch.nn.BatchNorm2d(planes * 4))).to(dev).double().eval()
        blk64.load_state_dict({k: v.double() for k, v in blk.state_dict().items()})
        x64 = x0.double().requires_grad_(True)
        # (the ReLU masks are the fp32 run's own: a pre-activation within rounding of zero - one in ~10^5 elements here - flips between
        # fp32 summation orders, and ONE flipped element moves a per-channel sum by ~1e-3 of its norm)
        with torch.no_grad():
            h1 = conv1x1_bn_act(x0, blk.conv1, blk.bn1)
            h2 = conv3x3_bn_act(h1, blk.conv2, blk.bn2)
        h = blk64.bn1(blk64.conv1(x64)) * (h1 > 0)
        h = blk64.bn2(blk64.conv2(h)) * (h2 > 0)
        idn = x64 if ds is None else blk64.downsample(x64)
        ((blk64.bn3(blk64.conv3(h)) + idn) * (res[True][0] > 0)).backward(up.double())
        want = dict(blk64.named_parameters())
        for name, a in zip(names, res[True][2:]):
            if name.startswith(("bn1.", "bn2.")):
                err = float((a.double() - want[name].grad).norm() / (want[name].grad.norm() + 1e-30))
                assert err < 5e-6, (name, err)
    # a RETAINED graph runs its backward twice (hand-over marks, partial sums and the branch token are per pass)
    monkeypatch.setattr(resnet50, "FUSE_BN_BWD", True)
    x = x0.clone().requires_grad_(True)
    y = blk(x)
    twice = []
    for keep in (True, False):
        for p in blk.parameters():
            p.grad = None
        x.grad = None
        y.backward(up, retain_graph=keep)
        gemm.join_side()
        torch.cuda.synchronize()
        twice.append([x.grad.clone()] + [p.grad.clone() for p in blk.parameters() if p.requires_grad])
    for a, b, c in zip(twice[0], twice[1], res[True][1:]):
        assert torch.equal(a, b) and torch.equal(a, c)
    # misuse: the chained tensor feeds a second consumer -> its producer refuses the accumulated gradient
    monkeypatch.setattr(resnet50, "FUSE_BN_BWD", True)
    x = x0.clone().requires_grad_(True)
    h1 = conv1x1_bn_act(x, blk.conv1, blk.bn1)
    h2 = conv3x3_bn_act(h1, blk.conv2, blk.bn2, fuse_input_bn=True)
    with pytest.raises(RuntimeError, match="second consumer"):
        (h2.sum() + h1.sum()).backward()
    gemm.join_side(discard=True)
    # ... and the check is still armed after a RESTRICTED pass over the same (retained) graph in which the consumer stepped aside
    # (the per-pass decision must not clear the forward-time mark: ADVICE r5, ops/chain.py still_private / take)
    x = x0.clone().requires_grad_(True)
    h1 = conv1x1_bn_act(x, blk.conv1, blk.bn1)
    h2 = conv3x3_bn_act(h1, blk.conv2, blk.bn2, fuse_input_bn=True)
    loss = h2.sum() + h1.sum()
    g_plain, = torch.autograd.grad(loss, [x], retain_graph=True)       # (plain autograd path for this pass: no raise, right numbers)
    xr = x0.clone().requires_grad_(True)
    r1 = conv1x1_bn_act(xr, blk.conv1, blk.bn1)
    r2 = conv3x3_bn_act(r1, blk.conv2, blk.bn2)
    g_ref, = torch.autograd.grad(r2.sum() + r1.sum(), [xr])
    assert torch.equal(g_plain, g_ref)
    with pytest.raises(RuntimeError, match="second consumer"):
        loss.backward()
    gemm.join_side(discard=True)


def test_bottleneck_inner_activation_gradients_are_the_plain_ones(dev):
    """The chained BatchNorm backward and the branch hand-over rewrite what flows along edges INSIDE a bottleneck (ops/chain.py,
    ops/conv1x1.py).  Whoever looks at such an edge - torch.autograd.grad towards an inner activation or a single weight, a tensor
    hook, retain_grad() - must get the ordinary gradient, not the rewritten one: the fused layers notice at their backward
    (hooks / retain_grad on the activation, torch._C._will_engine_execute_node for the node that would receive the hand-over) and
    take the plain autograd path for that pass (task 8)."""
    from cim_amd.modeling import resnet50
    from cim_amd.ops import conv1x1_bn_act, conv3x3_bn_act, gemm
    torch.manual_seed(3)
    blk = resnet50.Bottleneck(256, 64, 1, None).to(dev).eval()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5); m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 2.0)
    x0 = torch.randn(1, 256, 17, 23, device=dev)

    def run(fused):
        x = x0.clone().requires_grad_(True)
        branch = {} if fused else None
        h1 = conv1x1_bn_act(x, blk.conv1, blk.bn1, branch=branch)
        h2 = conv3x3_bn_act(h1, blk.conv2, blk.bn2, fuse_input_bn=fused)
        y = conv1x1_bn_act(h2, blk.conv3, blk.bn3, residual=x, fuse_input_bn=fused, branch=branch)
        return x, h1, h2, y

    def full(y, x, up, retain=False):
        for p in blk.parameters():
            p.grad = None
        x.grad = None
        y.backward(up, retain_graph=retain)
        gemm.join_side()
        torch.cuda.synchronize()
        return [x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]

    x, h1, h2, y = run(False)
    up = torch.randn_like(y)
    ref_h1, ref_h2, ref_w3 = torch.autograd.grad(y, [h1, h2, blk.conv3.weight], up, retain_graph=True)
    ref_full = full(y, x, up)
    close = lambda a, b: float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7      # (affine sums: another order)
    # (a) torch.autograd.grad towards inner activations / one weight: the hand-overs' receivers are not part of the pass
    x, h1, h2, y = run(True)
    g1, = torch.autograd.grad(y, [h1], up, retain_graph=True)
    g2, = torch.autograd.grad(y, [h2], up, retain_graph=True)
    gw, = torch.autograd.grad(y, [blk.conv3.weight], up, retain_graph=True)
    assert torch.equal(g1, ref_h1) and torch.equal(g2, ref_h2) and torch.equal(gw, ref_w3)
    # ... and the complete pass on the same graph afterwards is the chained one, unchanged
    got = full(y, x, up)
    assert all(close(a, b) for a, b in zip(got, ref_full)) and torch.equal(got[0], ref_full[0])
    # (b) a hook on an inner activation sees the plain gradient; the pass is still right
    x, h1, h2, y = run(True)
    seen = {}
    h2.register_hook(lambda g: seen.__setitem__("h2", g.clone()))
    got = full(y, x, up)
    assert torch.equal(seen["h2"], ref_h2) and all(close(a, b) for a, b in zip(got, ref_full))
    # (c) retain_grad()
    x, h1, h2, y = run(True)
    h1.retain_grad()
    got = full(y, x, up)
    assert torch.equal(h1.grad, ref_h1) and all(close(a, b) for a, b in zip(got, ref_full))
    # (d) a receiver with a chained producer is refused up front, with a message
    x = x0.clone().requires_grad_(True)
    h1 = conv1x1_bn_act(x, blk.conv1, blk.bn1)
    conv = torch.nn.Conv2d(64, 64, 1, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(64).to(dev).eval()
    with pytest.raises(ValueError, match="RECEIVER"):
        conv1x1_bn_act(h1, conv, bn, fuse_input_bn=True, branch={})
    gemm.join_side(discard=True)


@pytest.mark.parametrize("cin,cout,H,W,stride,res,relu,B", [(64, 64, 20, 27, 1, False, True, 1), (128, 128, 33, 43, 2, False, True, 1),
                                                            (256, 256, 17, 22, 1, True, True, 2), (32, 48, 9, 5, 2, False, False, 1),
                                                            (128, 128, 66, 86, 1, False, True, 1), (4, 8, 1, 7, 1, True, False, 1)])
def test_conv3x3_bn_act_matches_aten(dev, cin, cout, H, W, stride, res, relu, B):
    """Implicit-GEMM 3 x 3 convolution + frozen BatchNorm (+ residual) (+ ReLU), forward and all gradients, against the ATen
    ops in float64 (torchvision Bottleneck.conv2 / bn2 shapes of the C4 body, both strides, ragged tiles, a one-row map)."""
    from cim_amd.ops import conv3x3_bn_act
    torch.manual_seed(cin + H)
    conv = torch.nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(B, cout, ho, wo, device=dev, requires_grad=True) if res else None
    y = conv3x3_bn_act(x, conv, bn, residual=r, relu=relu)
    g = torch.randn_like(y)
    y.backward(g)
    got = [y.detach(), x.grad, conv.weight.grad, bn.weight.grad, bn.bias.grad] + ([r.grad] if res else [])
    conv64, bn64 = copy.deepcopy(conv).double(), copy.deepcopy(bn).double()
    for m in (conv64, bn64):
        for p_ in m.parameters():
            p_.grad = None
    x64 = x.detach().double().requires_grad_(True)
    r64 = r.detach().double().requires_grad_(True) if res else None
    o = bn64(conv64(x64))
    if res:
        o = o + r64
    if relu:
        o = torch.relu(o)
    o.backward(g.double())
    ref = [o.detach(), x64.grad, conv64.weight.grad, bn64.weight.grad, bn64.bias.grad] + ([r64.grad] if res else [])
    for name, a, b_ in zip(("y", "dx", "dw", "dgamma", "dbeta", "dres"), got, ref):
        scale = float(b_.abs().max()) + 1e-30
        err = float((a.double() - b_).abs().max()) / scale
        assert err < 2e-5, (name, err)


@pytest.mark.parametrize("cin,cout,H,W,dil,relu", [(64, 64, 45, 60, 1, True), (512, 512, 23, 30, 2, True), (128, 256, 31, 17, 1, False),
                                                     (16, 32, 9, 11, 2, True)])
def test_conv3x3_bias_act_matches_aten(dev, cin, cout, H, W, dil, relu):
    """VGG16's convolutions (lib/modeling/vgg16.py:34-78: 3 x 3, bias, no BatchNorm, dilation 2 in conv5) on the
    implicit-GEMM kernel: forward, dx, dw and the bias gradient against ATen in float64."""
    from cim_amd.ops import conv3x3_bias_act
    torch.manual_seed(cin + H + dil)
    conv = torch.nn.Conv2d(cin, cout, 3, stride=1, padding=dil, dilation=dil, bias=True).to(dev)
    x = torch.randn(1, cin, H, W, device=dev, requires_grad=True)
    y = conv3x3_bias_act(x, conv, relu=relu)
    g = torch.randn_like(y)
    y.backward(g)
    c64 = copy.deepcopy(conv).double()
    for p_ in c64.parameters():
        p_.grad = None
    x64 = x.detach().double().requires_grad_(True)
    o = c64(x64)
    if relu:
        o = torch.relu(o)
    o.backward(g.double())
    for name, a, b_ in (("y", y.detach(), o.detach()), ("dx", x.grad, x64.grad), ("dw", conv.weight.grad, c64.weight.grad),
                        ("dbias", conv.bias.grad, c64.bias.grad)):
        err = float((a.double() - b_).abs().max()) / (float(b_.abs().max()) + 1e-30)
        assert err < 2e-5, (name, err)


def test_conv3x3_rgb_stem_forward_only(dev):
    """3 input channels (the frozen stems of VGG16, vgg16.py:37, and HRNet, HRNet.py:263-266): forward on the kernel."""
    from cim_amd.ops import conv3x3_bias_act, conv3x3_bn_act
    torch.manual_seed(1)
    x = torch.randn(1, 3, 75, 100, device=dev)
    conv = torch.nn.Conv2d(3, 64, 3, padding=1, bias=True).to(dev)
    for p_ in conv.parameters():
        p_.requires_grad = False
    y = conv3x3_bias_act(x, conv, relu=True)
    ref = torch.relu(copy.deepcopy(conv).double()(x.double()))
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    conv2 = torch.nn.Conv2d(3, 64, 3, stride=2, padding=1, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(64).to(dev).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
        y2 = conv3x3_bn_act(x, conv2, bn)
        ref2 = torch.relu(copy.deepcopy(bn).double()(copy.deepcopy(conv2).double()(x.double())))
    assert float((y2.double() - ref2).abs().max() / ref2.abs().max()) < 2e-5


@pytest.mark.parametrize("k,stride", [(3, 2), (1, 1)])
def test_conv_with_bias_in_front_of_batchnorm(dev, k, stride):
    """HRNet's downsamp_modules / final_layer (HRNet.py:283-312): a convolution WITH a bias followed by a frozen BatchNorm.
    The bias is folded into the BatchNorm mean; its gradient comes back through that fold."""
    from cim_amd.ops import conv1x1_bn_act, conv3x3_bn_act
    torch.manual_seed(k)
    cin, cout, H, W = 64, 128, 19, 26
    conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=True).to(dev)
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(1, cin, H, W, device=dev, requires_grad=True)
    y = (conv3x3_bn_act if k == 3 else conv1x1_bn_act)(x, conv, bn, relu=True)
    g = torch.randn_like(y)
    y.backward(g)
    c64, b64 = copy.deepcopy(conv).double(), copy.deepcopy(bn).double()
    for m in (c64, b64):
        for p_ in m.parameters():
            p_.grad = None
    x64 = x.detach().double().requires_grad_(True)
    o = torch.relu(b64(c64(x64)))
    o.backward(g.double())
    for name, a, b_ in (("y", y.detach(), o.detach()), ("dx", x.grad, x64.grad), ("dw", conv.weight.grad, c64.weight.grad),
                        ("dbias", conv.bias.grad, c64.bias.grad), ("dgamma", bn.weight.grad, b64.weight.grad), ("dbeta", bn.bias.grad, b64.bias.grad)):
        err = float((a.double() - b_).abs().max()) / (float(b_.abs().max()) + 1e-30)
        assert err < 2e-5, (name, err)


def test_gpu_fallbacks_are_errors_by_default(dev, monkeypatch):
    """A CUDA tensor that would take an ATen / MIOpen branch raises - by default, whatever CIM_STRICT says - and is counted; the
    opt-out is explicit and per operator."""
    monkeypatch.delenv("CIM_STRICT", raising=False)
    from cim_amd import _lib
    from cim_amd.ops import conv3x3_bn_act, fallback
    conv = torch.nn.Conv2d(8, 8, 3, padding=1, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(8).to(dev).train()                  # training-mode statistics: not on the fused path
    x = torch.randn(1, 8, 5, 5, device=dev)
    with pytest.raises(_lib.CimHipError):
        conv3x3_bn_act(x, conv, bn)
    with fallback.allowed("conv3x3_bn_act"):
        conv3x3_bn_act(x, conv, bn)
    assert any(k[0] == "conv3x3_bn_act" for k in fallback.counts())


def test_backbone_weight_gradients_deferred_to_the_side_stream(dev, monkeypatch):
    """The body's weight-gradient GEMMs run on the side stream and are joined once, at the end of the backward pass
    (cim_amd/ops/gemm.py: defer_side_join): same gradients as with the join inside every layer, nothing left pending after
    backward(), and accumulation into an existing .grad (second backward without zero_grad) still exact."""
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling import resnet50
    from cim_amd.ops import gemm
    apply_preset("resnet50_voc")
    torch.manual_seed(0)
    body = resnet50.resnet().to(dev).train()
    x = torch.randn(1, 3, 200, 264, device=dev)
    g = None

    def grads(defer, passes):
        monkeypatch.setattr(gemm, "DEFER_DW", defer)
        body.zero_grad(set_to_none=True)
        nonlocal g
        for _ in range(passes):
            y = body(x)
            if g is None:
                g = torch.randn_like(y)
            y.backward(g)
            assert all(not (ent[1] or ent[2]) for ent in gemm._DEFERRED.values())          # joined by the engine callback
        torch.cuda.synchronize()
        return [p.grad.clone() for p in body.parameters() if p.grad is not None]

    ref1, ref2 = grads(False, 1), grads(False, 2)
    for trial in range(3):                                                       # (a race would show as run-to-run differences)
        for a, b in zip(ref1, grads(True, 1)):
            assert torch.equal(a, b)
    for a, b in zip(ref2, grads(True, 2)):
        assert torch.equal(a, b)
    for a, b in zip(ref1, ref2):
        torch.testing.assert_close(2 * a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("H,W,stride", [(75, 100, 2), (516, 688, 2), (31, 45, 1)])
def test_conv7x7_stem_matches_aten(dev, H, W, stride):
    """The frozen ResNet stem (7 x 7, 3 -> 64 channels, padding 3) + BatchNorm + ReLU as one implicit-GEMM launch."""
    from cim_amd.ops import conv7x7_bn_act
    torch.manual_seed(H)
    conv = torch.nn.Conv2d(3, 64, 7, stride=stride, padding=3, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(64).to(dev).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    for p_ in list(conv.parameters()) + list(bn.parameters()):
        p_.requires_grad = False
    x = torch.randn(2, 3, H, W, device=dev)
    y = conv7x7_bn_act(x, conv, bn)
    ref = torch.relu(copy.deepcopy(bn).double()(copy.deepcopy(conv).double()(x.double())))
    assert y.shape == ref.shape
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    assert err < 5e-6, err
    conv.weight.requires_grad = True                     # a stem that trains: the ATen path (autograd), a counted fallback
    from cim_amd.ops import fallback
    with fallback.allowed("conv7x7_bn_act"):
        y2 = conv7x7_bn_act(x, conv, bn)
    assert y2.requires_grad and float((y2.detach().double() - ref).abs().max() / ref.abs().max()) < 1e-4
