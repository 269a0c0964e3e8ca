"""-m gpu: the exact-fp32 MFMA contractions (cim_amd/csrc/gemm_f32.hip) through the C ABI against
fp64 references: every operand layout, ragged M/N/K, split-K, the implicit 3x3 conv and its
data / weight gradients, and the autograd wrappers used by MaskFuse."""
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


class _engine:
    """Run the wrapped block on one arithmetic engine of the contraction library."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        from cim_amd import _lib
        from cim_amd.ops import gemm as G
        self.G, self.lib, self.saved = G, _lib, (G.ENGINE, G.PAIR)
        G.ENGINE = "f16x2" if self.name == "f16x2p" else self.name       # (the engine is an argument of every library call)
        G.PAIR = self.name == "f16x2p"

    def __exit__(self, *exc):
        self.G.ENGINE = self.saved[0]
        self.G.PAIR = self.saved[1]


@pytest.mark.parametrize("M,N,K", [(1000, 4096, 4096), (300, 260, 1000), (37, 8, 20), (513, 516, 48), (256, 256, 16)])
@pytest.mark.parametrize("a_m,b_k", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts_vs_fp64(dev, M, N, K, a_m, b_k):
    from cim_amd.ops import gemm as G
    if (a_m and M % 4) or (not a_m and K % 4) or (b_k and K % 4):
        pytest.skip("layout needs 16-byte rows")
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ B.double() + bias.double()
    a_dev = (A.t().contiguous() if a_m else A).to(dev)
    b_dev = (B.t().contiguous() if b_k else B).to(dev)
    c = G.gemm(a_dev, b_dev, M, N, K, M if a_m else K, K if b_k else N, bool(a_m), bool(b_k), bias.to(dev))
    assert _rel(c.cpu(), ref) < 2e-6        # exact-fp32 class error (K <= 4096)
    c2 = G.gemm(a_dev, b_dev, M, N, K, M if a_m else K, K if b_k else N, bool(a_m), bool(b_k), bias.to(dev), relu=True)
    assert _rel(c2.cpu(), ref.clamp(min=0)) < 2e-6


def test_gemm_is_asymmetric_and_deterministic(dev):
    """A = I with an asymmetric B catches a transposed C write; split-K reduces in a fixed order."""
    from cim_amd.ops import gemm as G
    n = 320
    B = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 1000.0
    c = G.gemm(torch.eye(n).to(dev), B.to(dev), n, n, n, n, n)
    if G.ENGINE == "f16x2":      # the two-term split carries 23 of fp32's 24 significant bits: I.B is within 1 ulp of B
        assert float(((c.cpu() - B).abs() / B.clamp(min=1e-30)).max()) <= 2.0 ** -23
        with _engine("bf16x3"):
            c = G.gemm(torch.eye(n).to(dev), B.to(dev), n, n, n, n, n)
    assert torch.equal(c.cpu(), B)
    g = torch.Generator().manual_seed(1)
    A = torch.randn(200, 50176, generator=g).to(dev)          # fc1-like: few tiles, long K -> split-K
    W = torch.randn(256, 50176, generator=g).to(dev)
    y1 = G.gemm(A, W, 200, 256, 50176, 50176, 50176, b_kcontig=True)
    y2 = G.gemm(A, W, 200, 256, 50176, 50176, 50176, b_kcontig=True)
    assert torch.equal(y1, y2)
    assert _rel(y1.cpu(), A.cpu().double() @ W.cpu().double().t()) < 3e-5   # K = 50176: fp32 accumulation error grows with K


def test_default_engine_is_f16x2p():
    """Default: pair images for MaskFuse's fused head (PAIR), the f16x2 engine for every other contraction."""
    from cim_amd.ops import gemm as G
    import os
    want = os.environ.get("CIM_GEMM_ENGINE", "f16x2p")
    assert (G.ENGINE, G.PAIR) == (("f16x2", True) if want == "f16x2p" else (want, False))


@pytest.mark.parametrize("K", [2048, 50176])
def test_split_engines_error_class(dev, K):
    """The split engines (f16x2: scaled two-term fp16 split, 3 MFMA products; bf16x3: exact three-term bf16
    split, 6 products; both fp32 accumulate) must sit in the same error class as the f32-multiply MFMA engine
    against fp64, on unit-variance data, on data with a large dynamic range (exponents spread over 2^+-20)
    and on operands whose rows / columns differ in magnitude by 2^+-30 (the per-row / per-column scales)."""
    from cim_amd.ops import gemm as G
    g = torch.Generator().manual_seed(K)
    M = N = 256
    for case in ("unit", "spread", "rowscale"):
        A = torch.randn(M, K, generator=g)
        B = torch.randn(K, N, generator=g)
        if case == "spread":
            A = A * torch.exp2(20.0 * (torch.rand(M, K, generator=g) - 0.5))
            B = B * torch.exp2(20.0 * (torch.rand(K, N, generator=g) - 0.5))
        if case == "rowscale":
            A = A * torch.exp2(torch.randint(-30, 31, (M, 1), generator=g).float())
            B = B * torch.exp2(torch.randint(-30, 31, (1, N), generator=g).float())
        ref = A.double() @ B.double()
        scale = (A.double().abs() @ B.double().abs())            # condition-aware (componentwise) error scale
        err = {}
        for engine in ("fp32", "bf16x3", "f16x2"):
            with _engine(engine):
                c = G.gemm(A.to(dev), B.to(dev), M, N, K, K, N)
            err[engine] = float(((c.cpu().double() - ref).abs() / scale).max())
        assert max(err.values()) < 2e-6, (case, err)             # all far below fp32 eps * sqrt(K)
        assert err["bf16x3"] < 2.0 * err["fp32"] + 1e-8, (case, err)
        assert err["f16x2"] < 2.0 * err["fp32"] + 1e-7, (case, err)


def test_f16x2_small_elements_below_row_max(dev):
    """Elements 2^-17 .. 2^-30 below their row / column maximum land in fp16's subnormal range after scaling:
    they keep an absolute accuracy of ~2^-40 of that maximum (nothing is flushed to zero)."""
    from cim_amd.ops import gemm as G
    g = torch.Generator().manual_seed(11)
    M, N, K = 256, 256, 512
    A = torch.randn(M, K, generator=g) * torch.exp2(-torch.randint(17, 31, (M, K), generator=g).float())
    B = torch.randn(K, N, generator=g)
    A[:, 0] = 1.0                                                # the row maximum
    B[0, :] = 0.0                                                # ... multiplies zero: only the small elements contribute
    ref = A.double() @ B.double()
    with _engine("f16x2"):
        c = G.gemm(A.to(dev), B.to(dev), M, N, K, K, N)
    # error budget: K elements x 2^-25 (half a subnormal step at scale 2^14) / 2^14 x |b| ~ K * 2^-39 * 4
    assert float((c.cpu().double() - ref).abs().max()) < K * 2.0 ** -39 * 6
    assert float((c.cpu().double() - ref).abs().max() / ref.abs().max()) < 1e-3      # and they are NOT flushed


@pytest.mark.parametrize("engine", ["bf16x3", "f16x2"])
def test_split_engines_exact_on_representable(dev, engine):
    """Operands whose split terms have small-integer products are reproduced exactly:
    checks the split planes, the swizzled LDS layout, the scales and the k-pair packing of every loader."""
    from cim_amd.ops import gemm as G
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 264, 176
    A = torch.randint(-8, 9, (M, K), generator=g).float() + torch.randint(-8, 9, (M, K), generator=g).float() / 4096.0
    B = torch.randint(-8, 9, (K, N), generator=g).float()
    ref = (A.double() @ B.double())
    with _engine(engine):
        for a_m in (0, 1):
            for b_k in (0, 1):
                a_dev = (A.t().contiguous() if a_m else A).to(dev)
                b_dev = (B.t().contiguous() if b_k else B).to(dev)
                c = G.gemm(a_dev, b_dev, M, N, K, M if a_m else K, K if b_k else N, bool(a_m), bool(b_k))
                assert torch.equal(c.cpu().double(), ref), (a_m, b_k)


def test_amax_rowcol(dev):
    from cim_amd.ops import gemm as G
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 301, 1100, generator=g)
    x[1, 7, :] = 0.0
    xd = x.to(dev)[:, :, :1028].contiguous()
    x = x[:, :, :1028]
    ra, ca = G.amax(xd, 301, 1028, 1028, True, True, batch=3, bs=301 * 1028)
    assert torch.equal(ra.view(torch.float32).cpu().view(3, 301), x.abs().amax(dim=2))
    assert torch.equal(ca.view(torch.float32).cpu().view(3, 1028), x.abs().amax(dim=1))


@pytest.mark.parametrize("algo", ["winograd", "winograd4", "winograd7", "direct"])
@pytest.mark.parametrize("R,Cin,Cout", [(11, 32, 48), (40, 64, 272), (6, 16, 16)])
def test_conv3x3_fwd_bwd_vs_fp64(dev, R, Cin, Cout, algo, monkeypatch):
    from cim_amd.ops import conv3x3, gemm as G
    monkeypatch.setattr(G, "CONV_ALGO", algo)
    # fp32 error classes: direct sum ~6e-7, F(2x2,3x3) ~1.5e-6, F(4x4,3x3) on {0,1,-1,2,-1/2,inf} ~7e-6
    tol = {"direct": 2e-6, "winograd": 6e-6, "winograd4": 3e-5, "winograd7": 3e-5}[algo]
    g = torch.Generator().manual_seed(R + Cin)
    x = torch.randn(R, Cin, 7, 7, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    go = torch.randn(R, Cout, 7, 7, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.relu(F.conv2d(xr, wr, br, padding=1))
    yr.backward(go.double())
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = conv3x3(xd, wd, bd, relu=True)
    assert y.shape == (R, Cout, 7, 7)
    y.backward(go.to(dev))
    assert _rel(y.detach().cpu(), yr.detach()) < tol
    assert _rel(xd.grad.cpu(), xr.grad) < tol
    assert _rel(wd.grad.cpu(), wr.grad) < tol
    assert _rel(bd.grad.cpu(), br.grad) < 2e-6


def test_linear_fwd_bwd_vs_fp64(dev):
    from cim_amd.ops import linear
    g = torch.Generator().manual_seed(3)
    x = torch.randn(130, 392, generator=g)
    w = torch.randn(96, 392, generator=g) * 0.05
    b = torch.randn(96, generator=g)
    go = torch.randn(130, 96, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.relu(F.linear(xr, wr, br))
    yr.backward(go.double())
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = linear(xd, wd, bd, relu=True)
    y.backward(go.to(dev))
    for got, ref in ((y.detach(), yr.detach()), (xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        assert _rel(got.cpu(), ref) < 2e-6


def test_conv3x3_full_size_linearity(dev):
    """cfg2 size (1000 x 7 x 7, 2048 -> 1024): conv(a + 2b) == conv(a) + 2 conv(b)."""
    from cim_amd.ops import conv3x3
    g = torch.Generator(device=dev).manual_seed(0)
    w = torch.randn(1024, 2048, 3, 3, device=dev, generator=g) * 0.01
    a = torch.randn(1000, 7, 7, 2048, device=dev, generator=g).permute(0, 3, 1, 2)
    b = torch.randn(1000, 7, 7, 2048, device=dev, generator=g).permute(0, 3, 1, 2)
    from cim_amd.ops import gemm as gemm_mod
    ya, yb, yab = conv3x3(a, w), conv3x3(b, w), conv3x3(a + 2 * b, w)
    # outputs are O(3); F(4x4,3x3) carries ~5x the rounding error of F(2x2,3x3)
    atol = 8e-4 if gemm_mod.CONV_ALGO in ("winograd4", "winograd7") else 2e-4
    torch.testing.assert_close(yab, ya + 2 * yb, rtol=1e-4, atol=atol)
    # spot-check 8 output rows against fp64
    idx = torch.tensor([0, 17, 48, 49, 500 * 49 + 24, 999 * 49 + 48, 999 * 49, 12345])
    ref = F.conv2d(a[idx // 49].double().cpu(), w.double().cpu(), padding=1)
    for j, i in enumerate(idx.tolist()):
        p = i % 49
        got = ya[i // 49, :, p // 7, p % 7].cpu().double()
        assert float((got - ref[j, :, p // 7, p % 7]).abs().max()) < 1e-4


@pytest.mark.parametrize("tile", [4, 7])
def test_wino_fused_scales(dev, tile):
    """The f16x2 engine's Winograd operand scales (both tilings): the row bounds stored by the input / adjoint-dy
    transforms and the column bounds derived from the untransformed tensors dominate the true maxima of the transformed
    operands (never below - that would overflow fp16 - and within the transforms' gain above)."""
    from cim_amd import _lib
    from cim_amd.ops import gemm as G
    g = torch.Generator().manual_seed(9)
    R, P, C, Cout = 13, 7, 72, 40
    x = (torch.randn(R, P, P, C, generator=g) * torch.exp2(torch.randint(-6, 7, (1, 1, 1, C), generator=g).float())).to(dev)
    w = torch.randn(Cout, C, 3, 3, generator=g).to(dev)
    npos, mt = (36, R * 4) if tile == 4 else (121, R)
    st = _lib.stream_ptr()
    V = torch.empty(npos, mt, C, device=dev)
    vr = torch.empty(npos * mt, dtype=torch.int32, device=dev)
    _lib.call("cim_wino_input_transform_amax", x.data_ptr(), V.data_ptr(), vr.data_ptr(), R, P, C, tile, st)
    V0 = torch.empty_like(V)
    _lib.call("cim_wino_input_transform", x.data_ptr(), V0.data_ptr(), R, P, C, tile, st)
    assert torch.equal(V, V0)
    rb, true_rows = vr.view(torch.float32).view(npos, mt), V.abs().amax(dim=2)
    assert bool((rb >= true_rows).all()) and bool((rb <= 49.01 * x.abs().max()).all())
    xc = G.amax(x, R * P * P, C, C, want_cols=True)[1]
    vb = G._bounds(xc, C, 1, 0, npos, dev).view(torch.float32).view(npos, C)
    true = V.abs().amax(dim=1)
    assert bool((vb >= true).all()) and bool((vb <= 49.01 * x.abs().amax(dim=(0, 1, 2))[None, :]).all())
    for mode, n, kd in ((0, Cout, C), (1, C, Cout)):
        U = torch.empty(npos, kd, n, device=dev)
        _lib.call("cim_wino_filter_transform", w.data_ptr(), U.data_ptr(), Cout, C, mode, tile, st)
        wr, wc = G.amax(w, Cout, C * 9, C * 9, True, True)
        ub = (G._bounds(wr, Cout, 1, 1, npos, dev) if mode == 0 else G._bounds(wc, C, 9, 1, npos, dev))
        assert bool((ub.view(torch.float32).view(npos, n) >= U.abs().amax(dim=1)).all()), mode
        if mode == 0 and tile == 7:          # the adjoint data gradient reads U K-contiguously: its "columns" are the rows [ci] of U[pos]
            ub2 = G._bounds(wc, C, 9, 1, npos, dev).view(torch.float32).view(npos, C)
            assert bool((ub2 >= U.abs().amax(dim=2)).all())
    dy = torch.randn(R, P, P, Cout, generator=g).to(dev)
    D = torch.empty(npos, mt, Cout, device=dev)
    _lib.call("cim_wino_dy_transform", dy.data_ptr(), D.data_ptr(), R, P, Cout, tile, st)
    db = G._bounds(G.amax(dy, R * P * P, Cout, Cout, want_cols=True)[1], Cout, 1, 2, npos, dev)
    assert bool((db.view(torch.float32).view(npos, Cout) >= D.abs().amax(dim=1)).all())
    if tile == 7:
        E = torch.empty(npos, mt, Cout, device=dev)
        er = torch.empty(npos * mt, dtype=torch.int32, device=dev)
        _lib.call("cim_wino_dy_adjoint_transform", dy.data_ptr(), E.data_ptr(), er.data_ptr(), R, P, Cout, tile, st)
        assert bool((er.view(torch.float32).view(npos, mt) >= E.abs().amax(dim=2)).all())


@pytest.mark.parametrize("Cout", [64, 192, 48])
def test_conv3x3_flatten_chw(dev, Cout):
    """conv3x3(..., flatten_chw=True) == F.relu(F.conv2d(...)).view(R, -1) of the reference's NCHW tensor, forward and
    backward (transposing kernel fused with the ReLU mask; Cout % 64 != 0 takes the strided-copy path)."""
    from cim_amd.ops import conv3x3
    g = torch.Generator().manual_seed(Cout)
    R, Cin = 9, 32
    x = torch.randn(R, Cin, 7, 7, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    go = torch.randn(R, Cout * 49, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.relu(F.conv2d(xr, wr, br, padding=1)).view(R, -1)
    yr.backward(go.double())
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = conv3x3(xd, wd, bd, relu=True, flatten_chw=True)
    assert y.shape == (R, Cout * 49) and y.is_contiguous()
    y.backward(go.to(dev))
    for got, ref in ((y.detach(), yr.detach()), (xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        assert _rel(got.cpu(), ref) < 3e-5


# ------------------------------------------------------------------ backbone 1 x 1 convolutions (csrc/conv1x1.hip)
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


def test_gpu_fallbacks_are_errors_under_strict(dev):
    """A CUDA tensor that would take an ATen / MIOpen branch raises under CIM_STRICT=1 (the suite's setting) and is counted."""
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
