"""-m gpu: the SUPERSEDED engines and algorithms of experiments/ (bf16x3 / f16x2 / fp32 MFMA GEMMs, direct and F(2x2,3x3) /
F(4x4,3x3) convolution, the fp32 stages of the mixed tiling) through experiments/libcim_exp.so against fp64 references: every operand
layout, ragged M/N/K, split-K, the implicit 3x3 conv and its gradients, the per-layer autograd Functions.  Test infrastructure keeping
test infrastructure honest: tests/test_gpu_tolerance.py compares these engines with the reference and with the product's pair engine."""
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
        from experiments import engines as G
        self.G, self.saved = G, G.ENGINE
        G.ENGINE = self.name                                              # (the engine is an argument of every library call)

    def __exit__(self, *exc):
        self.G.ENGINE = self.saved


@pytest.mark.parametrize("M,N,K", [(1000, 4096, 4096), (300, 260, 1000), (37, 8, 20), (513, 516, 48), (256, 256, 16)])
@pytest.mark.parametrize("a_m,b_k", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts_vs_fp64(dev, M, N, K, a_m, b_k):
    from experiments import engines as G
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
    from experiments import engines as G
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


def test_nothing_in_the_product_loads_the_experiments():
    """One engine, one algorithm in the product: no module under cim_amd/ imports experiments/ or names its switches."""
    import glob
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cim_amd")
    for path in glob.glob(os.path.join(root, "**", "*.py"), recursive=True):
        text = open(path).read()
        assert "import experiments" not in text and "from experiments" not in text, path
        assert "CIM_GEMM_ENGINE" not in text and "CIM_CONV_ALGO" not in text, path


@pytest.mark.parametrize("K", [2048, 50176])
def test_split_engines_error_class(dev, K):
    """The split engines (f16x2: scaled two-term fp16 split, 3 MFMA products; bf16x3: exact three-term bf16
    split, 6 products; both fp32 accumulate) must sit in the same error class as the f32-multiply MFMA engine
    against fp64, on unit-variance data, on data with a large dynamic range (exponents spread over 2^+-20)
    and on operands whose rows / columns differ in magnitude by 2^+-30 (the per-row / per-column scales)."""
    from experiments import engines as G
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
    from experiments import engines as G
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
    from experiments import engines as G
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
    from experiments import engines as G
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
    from experiments.engines import conv3x3
    from experiments import engines as G
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
    from experiments.engines import linear
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
    from experiments.engines import conv3x3
    g = torch.Generator(device=dev).manual_seed(0)
    w = torch.randn(1024, 2048, 3, 3, device=dev, generator=g) * 0.01
    a = torch.randn(1000, 7, 7, 2048, device=dev, generator=g).permute(0, 3, 1, 2)
    b = torch.randn(1000, 7, 7, 2048, device=dev, generator=g).permute(0, 3, 1, 2)
    from experiments import engines as gemm_mod
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
    from experiments import _lib
    from experiments import engines as G
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
    from experiments.engines import conv3x3
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
