"""-m gpu: the f16x2p contraction engine (cim_amd/csrc/gemm_pair.hip: pre-split fp16 pair images, LDS-DMA staging,
hardware-transposed LDS reads) through the C ABI against fp64: the split producer, every operand layout, ragged M / N,
batched and split-K launches, and its error class next to the f32-multiply engine."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from cim_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _decode(p):
    """Pair image -> float64 [batch][rows_pad][ld] (h + l) / scale."""
    raw = p.buf.cpu().numpy().view(np.float16).reshape(p.batch, p.rows_pad, p.ld // 8, 2, 8).astype(np.float64)
    v = (raw[:, :, :, 0, :] + raw[:, :, :, 1, :]).reshape(p.batch, p.rows_pad, p.ld)
    return v / p.scale.cpu().numpy().astype(np.float64)[:, None, None]


def test_split_image_layout_and_precision(dev):
    from cim_amd.ops import pair
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 70, 64, generator=g) * torch.exp2(8.0 * (torch.rand(3, 70, 64, generator=g) - 0.5))
    p = pair.split(x.to(dev), 70, 64, 64, batch=3, x_bs=70 * 64)
    assert p.rows_pad == 96 and p.ld == 64
    s = p.scale.cpu().numpy()
    assert np.all(np.log2(s) == np.round(np.log2(s)))                    # powers of two
    assert float((x.abs().max() * float(s[0]))) < 2.0 ** 15               # one scale for the whole call's maximum
    d = _decode(p)
    assert np.all(d[:, 70:, :] == 0.0)                                    # zero pad rows
    err = np.abs(d[:, :70, :] - x.double().numpy())
    # 22 significant bits relative to the element (|x| >= 2^-13 max), absolute 2^-39 max below
    bound = np.maximum(np.abs(x.double().numpy()) * 2.0 ** -22, float(x.abs().max()) * 2.0 ** -38)
    assert np.all(err <= bound)


def _operands(M, N, K, a_m, b_k, g, spread=0.0):
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    if spread:
        A = A * torch.exp2(spread * (torch.rand(M, K, generator=g) - 0.5))
        B = B * torch.exp2(spread * (torch.rand(K, N, generator=g) - 0.5))
    return A, B


def _pair_gemm(dev, A, B, a_m, b_k, bias=None, relu=False, c_amax=None, balance=False):
    from cim_amd.ops import pair
    M, K = A.shape
    N = B.shape[1]
    pa = pair.split((A.t().contiguous() if a_m else A).to(dev))
    pb = pair.split((B.t().contiguous() if b_k else B).to(dev))
    kk = pair.pad32(K)
    return pair.gemm(pa, pb, M, N, kk, bool(a_m), bool(b_k), bias=None if bias is None else bias.to(dev), relu=relu,
                     c_amax=c_amax, balance=balance)


@pytest.mark.parametrize("M,N,K", [(1000, 1024, 2048), (304, 264, 992), (40, 8, 32), (520, 520, 64), (256, 256, 32)])
@pytest.mark.parametrize("a_m,b_k", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_pair_gemm_layouts_vs_fp64(dev, M, N, K, a_m, b_k):
    # K-contiguous operands contract over their columns: K itself must be a multiple of 32 there; an operand
    # contracted over its rows is padded with zero rows by the image
    if (not a_m or b_k) and K % 32:
        pytest.skip("a K-contiguous operand needs K % 32 == 0")
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A, B = _operands(M, N, K, a_m, b_k, g)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ B.double() + bias.double()
    c = _pair_gemm(dev, A, B, a_m, b_k, bias)
    scale = A.double().abs() @ B.double().abs()
    assert float(((c.cpu().double() - ref).abs() / scale).max()) < 2e-6
    am = torch.zeros(1, dtype=torch.int32, device=dev)
    c2 = _pair_gemm(dev, A, B, a_m, b_k, bias, relu=True, c_amax=am)
    assert float(((c2.cpu().double() - ref.clamp(min=0)).abs() / scale).max()) < 2e-6
    assert float(am.view(torch.float32)) == float(c2.abs().max())


def test_pair_gemm_is_asymmetric_and_deterministic(dev):
    """A = I with an asymmetric B catches a transposed operand read or C write in every layout; split-K is fixed-order."""
    n = 320
    B = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 1000.0
    for a_m in (0, 1):
        for b_k in (0, 1):
            c = _pair_gemm(dev, torch.eye(n), B, a_m, b_k)
            assert float(((c.cpu() - B).abs() / B.clamp(min=1e-30)).max()) <= 2.0 ** -21, (a_m, b_k)
    g = torch.Generator().manual_seed(1)
    A = torch.randn(200, 50176, generator=g)
    W = torch.randn(50176, 256, generator=g)
    y1 = _pair_gemm(dev, A, W, 0, 1)
    y2 = _pair_gemm(dev, A, W, 0, 1)
    assert torch.equal(y1, y2)
    ref = A.double() @ W.double()
    assert float((y1.cpu().double() - ref).abs().max() / ref.abs().max()) < 3e-5


@pytest.mark.parametrize("rows,cols", [(1000, 4096), (37, 72), (130, 260)])
def test_masked_stats_in_front_of_a_split(dev, rows, cols):
    """pair.masked_stats: max |dz| and the column sums of dz = y > 0 ? dy : 0 without storing dz; split(dy, relu_y=y) writes the
    image of the same dz."""
    from cim_amd.ops import pair
    g = torch.Generator().manual_seed(rows + cols)
    dy = torch.randn(rows, cols, generator=g).to(dev)
    y = torch.randn(rows, cols, generator=g).clamp(min=0).to(dev)
    dz = torch.where(y > 0, dy, torch.zeros_like(dy))         # (dy * (y > 0) leaves -0.0 behind)
    am = torch.zeros(1, dtype=torch.int32, device=dev)
    db = pair.masked_stats(dy, y, am, True)
    assert float(am.view(torch.float32)) == float(dz.abs().max())
    ref = dz.double().sum(dim=0)
    assert float((db.double() - ref).abs().max()) <= 1e-5 * float(dz.abs().sum(dim=0).max())
    am2 = torch.zeros(1, dtype=torch.int32, device=dev)
    assert pair.masked_stats(dy, y, am2, False) is None and torch.equal(am, am2)
    if cols % 8 == 0:
        sc = pair.scales_from(am, 1)
        a = pair.split(dy, rows, cols, cols, scale=sc, relu_y=y)
        b = pair.split(dz, rows, cols, cols, scale=sc)
        assert torch.equal(a.buf, b.buf)


@pytest.mark.parametrize("b_k", [0, 1])
def test_pair_gemm_balanced_tail_columns(dev, b_k):
    """balance=True (ops/pair.py: tail_columns): the tiles of a short last round run as a separate split-K product over the last
    columns - the main columns are bit-identical to the one-launch product, the tail differs by the split-K summation order
    only, bias / ReLU / the output maximum cover both parts."""
    from cim_amd.ops import pair
    M, N, K = 300, 256 * 130, 1024                     # 2 x 130 = 260 tiles: one round of 256 + 4 tiles
    assert pair.tail_columns(M, N, K) == (256 * 128, 4)
    assert pair.tail_columns(1000, 50176, 1024) == (192 * 256, 4)       # fc1's data gradient at 1000 proposals
    assert pair.tail_columns(1000, 12544, 1024) is None and pair.tail_columns(1024, 256 * 64, 1024) is None
    g = torch.Generator().manual_seed(11 + b_k)
    A, B = _operands(M, N, K, 0, b_k, g)
    bias = torch.randn(N, generator=g)
    am0 = torch.zeros(1, dtype=torch.int32, device=dev)
    am1 = torch.zeros(1, dtype=torch.int32, device=dev)
    c0 = _pair_gemm(dev, A, B, 0, b_k, bias, relu=True, c_amax=am0)
    c1 = _pair_gemm(dev, A, B, 0, b_k, bias, relu=True, c_amax=am1, balance=True)
    n_main = 256 * 128
    assert torch.equal(c0[:, :n_main], c1[:, :n_main])
    ref = (A.double() @ B.double() + bias.double()).clamp(min=0)
    scale = A.double().abs() @ B.double().abs()
    assert float(((c1.cpu().double() - ref).abs() / scale).max()) < 2e-6
    assert float(am1.view(torch.float32)) == float(c1.abs().max())
    assert abs(float(am1.view(torch.float32)) - float(am0.view(torch.float32))) <= 1e-6 * float(am0.view(torch.float32))


def test_pair_gemm_batched_balanced_tail_entries(dev):
    """balance=True on a batched product (ops/pair.py: tail_entries): the entries whose tiles would form a short last round run as
    single split-K products - the others are bit-identical to the one-launch product, the tail differs by the summation order."""
    from cim_amd.ops import pair
    assert pair.tail_entries(1000, 2048, 1024, 121) == (1, 4)           # the Winograd data gradient at 1000 proposals
    assert pair.tail_entries(1000, 1024, 2048, 121) is None and pair.tail_entries(1200, 2048, 1024, 121) is None
    nb, M, N, K = 33, 300, 1024, 1024                                    # 33 x 8 tiles = 256 + 8: the last entry
    assert pair.tail_entries(M, N, K, nb) == (1, 4)
    g = torch.Generator().manual_seed(21)
    A = torch.randn(nb, M, K, generator=g)
    B = torch.randn(nb, K, N, generator=g) * torch.exp2(torch.arange(nb).float() % 5)[:, None, None]
    amax_b = torch.stack([b.abs().max() for b in B]).to(dev).view(torch.int32)
    pa = pair.split(A.to(dev), M, K, K, batch=nb, x_bs=M * K)
    pb = pair.split(B.to(dev), K, N, N, batch=nb, x_bs=K * N, scale=pair.scales_from(amax_b, nb))
    c0 = pair.gemm(pa, pb, M, N, K, False, False)
    c1 = pair.gemm(pa, pb, M, N, K, False, False, balance=True)
    assert torch.equal(c0[:nb - 1], c1[:nb - 1])
    ref = A.double() @ B.double()
    scale = A.double().abs() @ B.double().abs()
    assert float(((c1.cpu().double() - ref).abs() / scale).max()) < 2e-6


def test_pair_gemm_batched(dev):
    from cim_amd.ops import pair
    g = torch.Generator().manual_seed(5)
    nb, M, N, K = 11, 300, 264, 96
    A = torch.randn(nb, M, K, generator=g) * torch.exp2(torch.arange(nb).float() - 5)[:, None, None]
    B = torch.randn(nb, K, N, generator=g)
    ref = A.double() @ B.double()
    amax = torch.stack([a.abs().max() for a in A]).to(dev).view(torch.int32)
    sa = pair.scales_from(amax, nb)                                              # one scale per batch entry
    pa = pair.split(A.to(dev), M, K, K, batch=nb, x_bs=M * K, scale=sa)
    pb = pair.split(B.to(dev), K, N, N, batch=nb, x_bs=K * N)
    c = pair.gemm(pa, pb, M, N, K, False, False)
    scale = A.double().abs() @ B.double().abs()
    assert float(((c.cpu().double() - ref).abs() / scale).max()) < 2e-6
    # the same images contracted the other way: C2[b] = A[b]^T . A[b]  (A as the M-contiguous AND the N-contiguous operand)
    c2 = pair.gemm(pa, pa, K, K, pair.pad32(M), True, False)
    ref2 = A.double().transpose(1, 2) @ A.double()
    scale2 = A.double().abs().transpose(1, 2) @ A.double().abs()
    assert float(((c2.cpu().double() - ref2).abs() / scale2).max()) < 2e-6


def test_pair_gemm_chunked_launches_are_bit_identical(dev):
    """max_workgroups = n of cim_gemm_pair*: a product goes out as consecutive launches of at most n workgroups (the MaskFuse
    weight gradients beside the backbone backward) - same tiles, same arithmetic, same bits, for a batched product (>= 8 slices: the
    XCD-slice work order), a split-K product and a plain one; the cap is an ARGUMENT of the call (no library state), 0 = one launch."""
    from cim_amd import _lib
    from cim_amd.ops import pair
    g = torch.Generator().manual_seed(11)
    nb, M, N, K = 9, 520, 520, 64
    A, B = torch.randn(nb, M, K, generator=g).to(dev), torch.randn(nb, K, N, generator=g).to(dev)
    pa, pb = pair.split(A, M, K, K, batch=nb, x_bs=M * K), pair.split(B, K, N, N, batch=nb, x_bs=K * N)
    X, W = torch.randn(300, 8192, generator=g).to(dev), torch.randn(264, 8192, generator=g).to(dev)
    px, pw = pair.split(X), pair.split(W)
    assert _lib.call("cim_gemm_pair_splits", 300, 264, 8192) > 1
    ref = (pair.gemm(pa, pb, M, N, K, False, False), pair.gemm(px, pw, 300, 264, 8192, False, True),
           pair.gemm(pa, pa, K, K, pair.pad32(M), True, False))
    for limit in (1, 5, 7, 100000):
        got = (pair.gemm(pa, pb, M, N, K, False, False, limit=limit), pair.gemm(px, pw, 300, 264, 8192, False, True, limit=limit),
               pair.gemm(pa, pa, K, K, pair.pad32(M), True, False, limit=limit))
        for a, b in zip(ref, got):
            assert torch.equal(a, b), limit
    with pytest.raises(_lib.CimHipError):
        pair.gemm(px, pw, 300, 264, 8192, False, True, limit=-1)


def test_pair_gemm_co_resident_form_is_bit_identical(dev):
    """form = 1 of cim_gemm_pair* (128 x 256 tiles of four waves: the MaskFuse weight gradients beside the backbone's backward) - the
    same products in the same order per output element as the 256 x 256 form: same bits, for a batched product (ragged last tiles both
    ways), a split-K product, a plain one and chunked launches; only for the weight gradients' layout (both operands K-major)."""
    from cim_amd import _lib
    from cim_amd.ops import pair
    g = torch.Generator().manual_seed(12)
    nb, R, M, N = 9, 333, 520, 264
    X, Y = torch.randn(nb, R, M, generator=g).to(dev), torch.randn(nb, R, N, generator=g).to(dev)
    px, py = pair.split(X, R, M, M, batch=nb, x_bs=R * M), pair.split(Y, R, N, N, batch=nb, x_bs=R * N)
    Xl, Yl = torch.randn(8192, 296, generator=g).to(dev), torch.randn(8192, 264, generator=g).to(dev)
    pxl, pyl = pair.split(Xl), pair.split(Yl)
    assert _lib.call("cim_gemm_pair_splits", 296, 264, 8192) > 1
    Xs, Ys = torch.randn(1000, 1024, generator=g).to(dev), torch.randn(1000, 640, generator=g).to(dev)
    pxs, pys = pair.split(Xs), pair.split(Ys)
    Xt, Yt = torch.randn(20, 8, generator=g).to(dev), torch.randn(20, 16, generator=g).to(dev)      # one ragged tile, two 16-k slabs
    pxt, pyt = pair.split(Xt), pair.split(Yt)
    Xu, Yu = torch.randn(90, 136, generator=g).to(dev), torch.randn(90, 520, generator=g).to(dev)   # six slabs: the ring wraps once
    pxu, pyu = pair.split(Xu), pair.split(Yu)
    cases = ((px, py, M, N, pair.pad32(R)), (pxl, pyl, 296, 264, 8192), (pxs, pys, 1024, 640, pair.pad32(1000)),
             (pxt, pyt, 8, 16, 32), (pxu, pyu, 136, 520, 96))
    for a, b, m, n, k in cases:
        ref = pair.gemm(a, b, m, n, k, True, False)
        for limit in (0, 7):
            assert torch.equal(ref, pair.gemm(a, b, m, n, k, True, False, limit=limit, form=1)), (m, n, k, limit)
        one = pair.gemm(a, b, m, n, k, True, False, products=1)                    # (the one-product instantiation of both kernels)
        assert torch.equal(one, pair.gemm(a, b, m, n, k, True, False, products=1, form=1)), (m, n, k, "products=1")
    want = Xs.double().t() @ Ys.double()
    scale = Xs.double().abs().t() @ Ys.double().abs()
    assert float(((pair.gemm(pxs, pys, 1024, 640, pair.pad32(1000), True, False, form=1).double() - want).abs() / scale).max()) < 2e-6
    with pytest.raises(_lib.CimHipError):
        pair.gemm(pxl, pyl, 8192, 8192, 264, False, True, form=1)        # (a forward product's layout)
    with pytest.raises(_lib.CimHipError):
        pair.gemm(pxs, pys, 1024, 640, pair.pad32(1000), True, False, form=2)


@pytest.mark.parametrize("K", [2048, 50176])
def test_pair_engine_error_class(dev, K):
    """One scale per matrix: on unit-variance data and on data with exponents spread over 2^+-10 the engine sits in the
    error class of the f32-multiply MFMA engine (componentwise scale sum |a||b|)."""
    from cim_amd.ops import gemm as G
    from cim_amd import _lib
    g = torch.Generator().manual_seed(K)
    M = N = 256
    for spread in (0.0, 10.0):
        A, B = _operands(M, N, K, 0, 0, g, spread)
        ref = A.double() @ B.double()
        scale = A.double().abs() @ B.double().abs()
        c = _pair_gemm(dev, A, B, 0, 0)
        err_pair = float(((c.cpu().double() - ref).abs() / scale).max())
        c32 = torch.empty(M, N, device=dev)          # the true-fp32 MFMA comparator: v_mfma_f32_32x32x2_f32 (csrc/conv1x1.hip)
        splits = _lib.call("cim_gemm_small_splits", M, N, K)
        ws = torch.empty(splits * M * N, device=dev)
        Ad, Bd = A.to(dev), B.to(dev)                # (names: the operands must outlive the launch)
        _lib.call("cim_gemm_small_f32", Ad.data_ptr(), Bd.data_ptr(), c32.data_ptr(), M, N, K, K, N, N, 0, 0,
                  None, None, None, None, None, 0.0, None, 0, splits, ws.data_ptr(), _lib.stream_ptr())
        err_f32 = float(((c32.cpu().double() - ref).abs() / scale).max())
        assert err_pair < 2e-6 and err_pair < 2.0 * err_f32 + 1e-7, (spread, err_pair, err_f32)


def test_pair_small_elements_keep_absolute_accuracy(dev):
    """Elements far below the matrix maximum fall into fp16's subnormal range: absolute accuracy 2^-39 of the maximum."""
    g = torch.Generator().manual_seed(3)
    M = N = 256
    K = 64
    A = torch.randn(M, K, generator=g)
    A[:, 1:] *= 2.0 ** -20
    B = torch.randn(K, N, generator=g)
    ref = A.double() @ B.double()
    c = _pair_gemm(dev, A, B, 0, 0)
    # the large products carry the split's relative error (dropped l*l term <= 2^-22 |a b|), the 2^-20-scaled elements an
    # ABSOLUTE error of 2^-38 max |A| each - not a relative one
    big = float((A[:, :1].double().abs() @ B[:1].double().abs()).max())
    bound = 2.0 ** -21 * big + K * float(A.abs().max()) * 2.0 ** -38 * float(B.abs().max())
    assert float((c.cpu().double() - ref).abs().max()) < bound
    small = (A[:, 1:].double() @ B[1:].double())                     # the small elements' own contribution survives
    got_small = c.cpu().double() - A[:, :1].double() @ B[:1].double()
    assert float((got_small - small).abs().max()) < bound


# ---- producers that write pair images directly (winograd.hip) against the fp32 transforms -----------------------------
def _close_to(d, ref, floor):
    """decoded image vs the fp32 kernel's output: 22 bits relative to the element, `floor` (2^-38 of the position's scale
    bound) below that, plus the fp32 rounding of the transform itself (the two kernels may contract FMAs differently:
    relative to the position's largest value, not to a cancelled element)."""
    ref = ref.astype(np.float64)
    big = np.abs(ref).reshape(ref.shape[0], -1).max(axis=1).reshape((-1,) + (1,) * (ref.ndim - 1))
    bad = np.abs(d - ref) > np.abs(ref) * 2.0 ** -21 + floor + 1e-6 * big
    assert not bad.any(), (int(bad.sum()), float(np.abs(d - ref).max()))


def test_wino7_pair_producers_match_fp32_transforms(dev):
    from cim_amd import _lib
    from cim_amd.ops import pair
    from experiments import _lib as _xlib           # the fp32 stages of the same tiling: the comparator (test infrastructure)
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(11)
    R, C, Co = 37, 64, 128
    Rs = pair.pad32(R)
    x = torch.randn(R, 7, 7, C, generator=g).to(dev)
    amax = x.abs().max().reshape(1).view(torch.int32)
    # input transform
    V = torch.empty(121, R, C, device=dev)
    _xlib.call("cim_wino_input_transform", x.data_ptr(), V.data_ptr(), R, 7, C, 7, st)
    sc = torch.empty(121, device=dev)
    _lib.call("cim_wino7_pair_scales", amax.data_ptr(), 1, None, 0, sc.data_ptr(), st)
    Vp = pair.Pair(torch.full((121, Rs, C), 0x7fff7fff, dtype=torch.int32, device=dev), R, C, 121, sc)
    _lib.call("cim_wino7_input_pair", x.data_ptr(), Vp.buf.data_ptr(), sc.data_ptr(), R, Rs, C, st)
    d = _decode(Vp)
    assert np.all(d[:, R:, :] == 0.0)
    assert float((V.abs().amax(dim=(1, 2)) * sc).max()) < 2.0 ** 15          # the bound holds at every position
    floor = (2.0 ** -38 * 2.0 ** 15 / sc.cpu().numpy().astype(np.float64))[:, None, None]
    _close_to(d[:, :R, :], V.cpu().numpy(), floor)
    # dy transforms (weight gradient and adjoint data gradient)
    for adj, kind in ((0, 2), (1, 3)):
        D = torch.empty(121, R, C, device=dev)
        if adj:
            _xlib.call("cim_wino_dy_adjoint_transform", x.data_ptr(), D.data_ptr(), None, R, 7, C, 7, st)
        else:
            _xlib.call("cim_wino_dy_transform", x.data_ptr(), D.data_ptr(), R, 7, C, 7, st)
        _lib.call("cim_wino7_pair_scales", amax.data_ptr(), 1, None, kind, sc.data_ptr(), st)
        Dp = pair.Pair(torch.full((121, Rs, C), 0x7fff7fff, dtype=torch.int32, device=dev), R, C, 121, sc)
        _lib.call("cim_wino7_dy_pair", x.data_ptr(), Dp.buf.data_ptr(), sc.data_ptr(), R, Rs, C, adj, st)
        d = _decode(Dp)
        assert np.all(d[:, R:, :] == 0.0)
        assert float((D.abs().amax(dim=(1, 2)) * sc).max()) < 2.0 ** 15
        floor = (2.0 ** -38 * 2.0 ** 15 / sc.cpu().numpy().astype(np.float64))[:, None, None]
        _close_to(d[:, :R, :], D.cpu().numpy(), floor)
    # filter transform: U' [121][Cout][Cin] = U [121][Cin][Cout] transposed
    w = torch.randn(Co, C, 3, 3, generator=g).to(dev)
    U = torch.empty(121, C, Co, device=dev)
    _xlib.call("cim_wino_filter_transform", w.data_ptr(), U.data_ptr(), Co, C, 0, 7, st)
    wamax = w.abs().max().reshape(1).view(torch.int32)
    _lib.call("cim_wino7_pair_scales", wamax.data_ptr(), 1, None, 1, sc.data_ptr(), st)
    Up = pair.Pair(torch.empty((121, Co, C), dtype=torch.int32, device=dev), Co, C, 121, sc)
    _lib.call("cim_wino7_filter_pair", w.data_ptr(), Up.buf.data_ptr(), sc.data_ptr(), Co, C, st)
    assert float((U.abs().amax(dim=(1, 2)) * sc).max()) < 2.0 ** 15
    floor = (2.0 ** -38 * 2.0 ** 15 / sc.cpu().numpy().astype(np.float64))[:, None, None]
    _close_to(_decode(Up), U.transpose(1, 2).cpu().numpy(), floor)


def test_output_amax_and_flatten_pair(dev):
    from cim_amd import _lib
    from cim_amd.ops import pair
    from experiments import _lib as _xlib           # the fp32 stages of the same tiling: the comparator (test infrastructure)
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(12)
    R, C = 21, 128
    Rs = pair.pad32(R)
    M = torch.randn(121, R, C, generator=g).to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    y0 = torch.empty(R, 7, 7, C, device=dev)
    y1 = torch.empty(R, 7, 7, C, device=dev)
    am = torch.zeros(1, dtype=torch.int32, device=dev)
    _xlib.call("cim_wino_output_transform", M.data_ptr(), bias.data_ptr(), y0.data_ptr(), R, 7, C, 1, 7, st)
    _lib.call("cim_wino7_output_amax", M.data_ptr(), bias.data_ptr(), y1.data_ptr(), R, C, 1, am.data_ptr(), st)
    assert torch.equal(y0, y1)
    assert float(am.view(torch.float32)) == float(y0.max())
    sc = pair.scales_from(am, 1)
    Xp = pair.Pair(torch.full((1, Rs, C * 49), 0x7fff7fff, dtype=torch.int32, device=dev), R, C * 49, 1, sc)
    _lib.call("cim_flatten_chw_pair", y0.data_ptr(), Xp.buf.data_ptr(), sc.data_ptr(), R, Rs, 49, C, st)
    flat = y0.permute(0, 3, 1, 2).reshape(R, C * 49)                       # (c, h, w) order of `.view(R, -1)` on NCHW
    d = _decode(Xp)[0]
    assert np.all(d[R:] == 0.0)
    _close_to(d[:R], flat.cpu().numpy(), float(y0.max()) * 2.0 ** -38)


@pytest.mark.parametrize("r", [37, 64])
def test_maskfuse_pair_function_vs_per_layer_path(dev, r):
    """The fused head Function on pair images against the per-layer f16x2 Functions (same module parameters) and against
    float64: forward values and every gradient.

    The instance is drawn so that no ReLU pre-activation lies within 4e-6 of zero: the two paths round a pre-activation
    differently (~1e-7 .. 1e-6 here), and ONE flipped mask of the convolution's ReLU moves the gradient of `cat` by ~1e-3 of
    its norm (a 3 x 3 window of one proposal over all input channels) - a property of ReLU, not of either path.  With
    unseeded module parameters this test failed in ~8 % of its runs for exactly that reason (round 4: 600 draws, every
    failure was one flipped mask with |pre-activation| <= 1.4e-6; none with the masks taken from the path under test)."""
    from cim_amd.ops import maskfuse_pair, pair
    from experiments.engines import conv3x3, linear        # the per-layer f16x2 Functions: the comparator
    cin, cout, h = 128, 64, 256
    lin = torch.nn.functional.linear
    for seed in range(r, r + 64):
        torch.manual_seed(seed)
        cat = torch.randn(r, cin, 7, 7).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
        conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
        fc1, fc2 = torch.nn.Linear(cout * 49, h).to(dev), torch.nn.Linear(h, h).to(dev)
        dy = torch.randn(r, h).to(dev)
        params = [conv.weight, conv.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias]
        cat64 = cat.detach().double().requires_grad_()
        p64 = [t.detach().double().requires_grad_() for t in params]
        z0 = torch.nn.functional.conv2d(cat64, p64[0], p64[1], padding=1)
        z1 = lin(z0.relu().reshape(r, -1), p64[2], p64[3])
        z2 = lin(z1.relu(), p64[4], p64[5])
        if min(float(z.detach().abs().min()) for z in (z0, z1, z2)) > 4e-6:
            break
    else:
        raise AssertionError("no instance with unambiguous ReLU masks in 64 draws")
    want = z2.relu()
    want.backward(dy.double())
    g64 = [t.grad.cpu() for t in [cat64] + p64]
    assert maskfuse_pair.supported(cat, conv.weight, fc1.weight, fc2.weight)

    def grads(out):
        for t in [cat] + params:
            t.grad = None
        out.backward(dy)                  # (.backward(): the node's weight gradients are installed at the end of the pass, not returned)
        torch.cuda.synchronize()
        return [t.grad.double().cpu() for t in [cat] + params]

    y = conv3x3(cat, conv.weight, conv.bias, relu=True, flatten_chw=True)
    ref = linear(linear(y, fc1.weight, fc1.bias, relu=True), fc2.weight, fc2.bias, relu=True)
    gref = grads(ref)
    fa = pair.amax_of(cat.detach())
    out = maskfuse_pair.maskfuse_head(cat, conv, fc1, fc2, fa)
    gout = grads(out)
    names = ["cat", "wc", "bc", "w1", "b1", "w2", "b2"]
    for path, o, gs in (("per-layer path", ref, gref), ("fused pair node", out, gout)):
        e = float((o.detach().double() - want.detach()).abs().max() / want.abs().max())
        assert e < 1e-5, "%s: forward deviates from float64 by %.3g" % (path, e)
        for a, b, name in zip(gs, g64, names):
            e = float((a - b).norm() / b.norm())
            assert e < 1e-5, "%s: gradient of %s deviates from float64 by %.3g" % (path, name, e)
    assert float((out - ref).abs().max() / ref.abs().max()) < 2e-5
    for a, b, name in zip(gout, gref, ["cat", "wc", "bc", "w1", "b1", "w2", "b2"]):
        assert float((a - b).norm() / b.norm()) < 2e-5, name


@pytest.mark.parametrize("C,H,W,K", [(256, 33, 43, 70), (64, 20, 25, 33), (512, 45, 60, 40), (256, 7, 9, 3), (1024, 64, 64, 32), (8, 33, 43, 65),
                                     (256, 57, 75, 40), (64, 100, 128, 12)])        # (the reference's largest training scale; the limit)
def test_roi_align_wino7_pair_image_is_bit_identical_to_the_two_kernel_path(dev, C, H, W, K):
    """cim_roi_align_wino7_pair_fwd (round 5: ROIAlign + mask multiply + concat + the Winograd 4 + 3 input transform in one launch,
    lib/modeling/resnet50.py:121-135) writes the very pair image cim_roi_align_maskcat_fwd_ws + cim_wino7_input_pair write -
    every byte, pad rows included - and, as one autograd node with the head, gives the same outputs and gradients as the two
    nodes with the `cat` tensor in between (task 3)."""
    from cim_amd import _lib
    from cim_amd.ops import maskfuse_pair, pair, roi_align_maskcat
    from cim_amd.ops import roi_align as _  # noqa: F401
    import sys
    RA = sys.modules["cim_amd.ops.roi_align"]
    rng = np.random.RandomState(C + K)
    feat = torch.from_numpy(rng.randn(2, C, H, W).astype(np.float32)).to(dev).contiguous(memory_format=torch.channels_last)
    x1 = rng.uniform(-20, W * 16 * 0.8, K); y1 = rng.uniform(-20, H * 16 * 0.8, K)
    rois = np.stack([rng.randint(0, 2, K), x1, y1, x1 + rng.uniform(1, W * 16, K), y1 + rng.uniform(1, H * 16, K)], 1).astype(np.float32)
    rois[0, 1:] = (0, 0, W * 16, H * 16)
    rois[1, 1:] = (40, 40, 40.5, 40.25)
    rois[2, 1:] = (W * 16 + 500, 10, W * 16 + 900, 200)
    rois_d = torch.from_numpy(rois).to(dev)
    masks = torch.from_numpy((rng.rand(K, 7, 7) > 0.4).astype(np.float32)).to(dev)
    st = _lib.stream_ptr()
    fa = (pair.amax_of(feat).view(torch.float32) * 1.0).view(torch.int32)
    sV = maskfuse_pair._input_scales(fa, dev)
    rp = pair.pad32(K)
    # two kernels
    cat = roi_align_maskcat(feat, rois_d, masks, 7, 1 / 16.0, 0, True).contiguous(memory_format=torch.channels_last)
    V0 = torch.full((121, rp, 2 * C), 0x7fff7fff, dtype=torch.int32, device=dev)
    _lib.call("cim_wino7_input_pair", cat.data_ptr(), V0.data_ptr(), sV.data_ptr(), K, rp, 2 * C, st)
    # one kernel
    V1 = torch.full((121, rp, 2 * C), 0x7fff7fff, dtype=torch.int32, device=dev)
    ws = RA._workspace(K, 7, H, W, dev)
    _lib.call("cim_roi_align_wino7_pair_fwd", feat.data_ptr(), rois_d.data_ptr(), masks.data_ptr(), V1.data_ptr(), sV.data_ptr(),
              2, C, H, W, K, rp, 7, 1 / 16.0, 0, 1, ws.data_ptr(), st)
    assert torch.equal(V0, V1)
    if C % 32:
        return
    # the whole box head: one node against two
    torch.manual_seed(C)
    conv = torch.nn.Conv2d(2 * C, 64, 3, padding=1).to(dev)
    fc1, fc2 = torch.nn.Linear(64 * 49, 128).to(dev), torch.nn.Linear(128, 96).to(dev)
    params = [conv.weight, conv.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias]
    dy = torch.randn(K, 96, device=dev)
    assert maskfuse_pair.roi_supported(feat, conv.weight, fc1.weight, fc2.weight, 7)

    def run(fused):
        x = feat.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
        for t in params:
            t.grad = None
        if fused:
            out = maskfuse_pair.maskfuse_roi_head(x, rois_d, masks, conv, fc1, fc2, fa, 1 / 16.0, 0)
        else:
            out = maskfuse_pair.maskfuse_head(roi_align_maskcat(x, rois_d, masks, 7, 1 / 16.0, 0, True), conv, fc1, fc2, fa)
        out.backward(dy)
        from cim_amd.ops import gemm
        gemm.join_side()
        torch.cuda.synchronize()
        return [out.detach()] + [x.grad] + [t.grad for t in params]

    for a, b, name in zip(run(True), run(False), ["seg_x", "dfeat", "wc", "bc", "w1", "b1", "w2", "b2"]):
        if name == "dfeat":     # the one node folds dcat_lo + mask * dcat_hi into the convolution's last backward stage (another
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), name      # summation order than fma(mask, hi, lo) later)
        else:
            assert torch.equal(a, b), name


@pytest.mark.parametrize("R,C,mask", [(37, 256, True), (21, 512, True), (5, 256, False), (1, 256, True), (33, 2048, True)])
def test_flatten_backward_dy_images_in_one_launch_are_bit_identical(dev, R, C, mask):
    """cim_wino7_flatten_bwd_dy_pair (round 5): the backward of the (c, h, w) flatten + the conv's ReLU mask + both output-gradient
    transforms in ONE launch writes the very images (and bias partial sums) that cim_flatten_chw_bwd_bias + cim_wino7_dy_pair x 2
    write - every byte, pad rows included - without storing the masked gradient (lib/modeling/resnet50.py:104-105,135 backward)."""
    from cim_amd import _lib
    from cim_amd.ops import pair
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(R + C)
    Rs = pair.pad32(R)
    dX = torch.randn(R, C * 49, generator=g).to(dev)
    y = torch.randn(R, 7, 7, C, generator=g).to(dev) if mask else None
    amax = dX.abs().max().reshape(1).view(torch.int32)
    sE, sD = torch.empty(121, device=dev), torch.empty(121, device=dev)
    _lib.call("cim_wino7_pair_scales", amax.data_ptr(), 1, None, 3, sE.data_ptr(), st)
    _lib.call("cim_wino7_pair_scales", amax.data_ptr(), 1, None, 2, sD.data_ptr(), st)
    # three launches
    dy = torch.empty(R, 7, 7, C, device=dev)
    b0 = torch.empty(R, C, device=dev)
    _lib.call("cim_flatten_chw_bwd_bias", dX.data_ptr(), _lib.ptr(y), dy.data_ptr(), b0.data_ptr(), R, 49, C, st)
    E0 = torch.full((121, Rs, C), 0x7fff7fff, dtype=torch.int32, device=dev)
    D0 = torch.full((121, Rs, C), 0x7fff7fff, dtype=torch.int32, device=dev)
    _lib.call("cim_wino7_dy_pair", dy.data_ptr(), E0.data_ptr(), sE.data_ptr(), R, Rs, C, 1, st)
    _lib.call("cim_wino7_dy_pair", dy.data_ptr(), D0.data_ptr(), sD.data_ptr(), R, Rs, C, 0, st)
    # one launch (both images; each alone)
    for want_e, want_d in ((True, True), (True, False), (False, True)):
        E1 = torch.full((121, Rs, C), 0x7fff7fff, dtype=torch.int32, device=dev)
        D1 = torch.full((121, Rs, C), 0x7fff7fff, dtype=torch.int32, device=dev)
        b1 = torch.full((R, C), float("nan"), device=dev)
        _lib.call("cim_wino7_flatten_bwd_dy_pair", dX.data_ptr(), _lib.ptr(y), E1.data_ptr() if want_e else None,
                  sE.data_ptr() if want_e else None, D1.data_ptr() if want_d else None, sD.data_ptr() if want_d else None,
                  b1.data_ptr(), R, Rs, C, st)
        if want_e:
            assert torch.equal(E0, E1)
        if want_d:
            assert torch.equal(D0, D1)
        assert torch.equal(b0, b1)


def test_head_backward_with_and_without_the_fused_dy_launch_is_bit_identical(dev):
    from cim_amd.ops import gemm, maskfuse_pair, pair
    torch.manual_seed(3)
    R, C = 45, 128
    conv = torch.nn.Conv2d(2 * C, 256, 3, padding=1).to(dev)
    fc1, fc2 = torch.nn.Linear(256 * 49, 128).to(dev), torch.nn.Linear(128, 96).to(dev)
    params = [conv.weight, conv.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias]
    cat0 = torch.randn(R, 2 * C, 7, 7, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(R, 96, device=dev)
    fa = pair.amax_of(cat0)

    def run(fused):
        maskfuse_pair.FUSE_DY = fused
        try:
            cat = cat0.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
            for t in params:
                t.grad = None
            out = maskfuse_pair.maskfuse_head(cat, conv, fc1, fc2, fa)
            out.backward(dy)
            gemm.join_side()
            torch.cuda.synchronize()
            return [out.detach(), cat.grad] + [t.grad for t in params]
        finally:
            maskfuse_pair.FUSE_DY = True

    for a, b, name in zip(run(True), run(False), ["seg_x", "dcat", "wc", "bc", "w1", "b1", "w2", "b2"]):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("C,H,W,K,cout", [(32, 7, 9, 5, 256), (512, 45, 60, 20, 512), (1024, 33, 43, 24, 1024), (256, 57, 75, 12, 256)],
                         ids=["tiny", "cfg1-shape", "cfg2-shape", "scale-1200-map"])
def test_fused_box_head_against_the_oracle_and_float64(dev, C, H, W, K, cout):
    """The four fused launches of round 5 - roi_align_wino7_pair (ROIAlign + mask + concat + Winograd input transform),
    wino7_flatten_bwd_dy_pair (flatten backward + ReLU mask + both dy transforms), wino7_dx_maskfold (adjoint output transform + the
    backward of mask multiply and concat) and wino7_wgrad_out_rows - checked DIRECTLY, not against the kernels they replaced
    (VERDICT r5 item 6): the box head as ONE autograd node (lib/modeling/resnet50.py:120-138) against oracle/roi_align_ref.c for
    the ROIAlign (forward and backward) with everything between them evaluated in float64 from the definitions (conv2d, linear,
    ReLU).  The ReLU masks of the reference's backward are the node's own (a pre-activation within rounding of zero would otherwise
    flip one mask and move a gradient by ~1e-3 of its norm - a property of ReLU, not of either side)."""
    from cim_amd.ops import gemm, maskfuse_pair, pair
    from oracle import roi_align as oracle_roi
    rng = np.random.RandomState(C + K)
    torch.manual_seed(C + K)
    feat_np = rng.randn(1, C, H, W).astype(np.float32)
    x1 = rng.uniform(-20, W * 16 * 0.8, K); y1 = rng.uniform(-20, H * 16 * 0.8, K)
    rois = np.stack([np.zeros(K), x1, y1, x1 + rng.uniform(1, W * 16, K), y1 + rng.uniform(1, H * 16, K)], 1).astype(np.float32)
    rois[0, 1:] = (0, 0, W * 16, H * 16)                       # the whole image
    rois[1, 1:] = (40, 40, 40.5, 40.25)                        # degenerate (< 1 pixel of the map)
    rois[2, 1:] = (W * 16 + 500, 10, W * 16 + 900, 200)        # outside the image
    masks_np = (rng.rand(K, 7, 7) > 0.4).astype(np.float32)
    conv = torch.nn.Conv2d(2 * C, cout, 3, padding=1).to(dev)
    fc1, fc2 = torch.nn.Linear(cout * 49, 128).to(dev), torch.nn.Linear(128, 96).to(dev)
    params = [conv.weight, conv.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias]
    dy = torch.randn(K, 96, device=dev)
    feat = torch.from_numpy(feat_np).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rois_d, masks = torch.from_numpy(rois).to(dev), torch.from_numpy(masks_np).to(dev)
    assert maskfuse_pair.roi_supported(feat, conv.weight, fc1.weight, fc2.weight, 7) and maskfuse_pair.FUSE_DY and cout % 256 == 0
    fa = torch.zeros(2, dtype=torch.int32, device=dev)
    pair.amax_of(feat.detach(), out=fa[0:1])
    pair.amax_of(masks, out=fa[1:2])
    out = maskfuse_pair.maskfuse_roi_head(feat, rois_d, masks, conv, fc1, fc2, fa, 1 / 16.0, 0)
    y_own, Y1_own, Y2_own = out.grad_fn.saved_tensors[:3]            # the node's ReLU outputs: [K,7,7,cout], [K,128], [K,96]
    m0 = (y_own > 0).permute(0, 3, 1, 2).double().cpu()
    m1, m2 = (Y1_own > 0).double().cpu(), (Y2_own > 0).double().cpu()
    out.backward(dy)
    gemm.join_side()
    torch.cuda.synchronize()
    got = [out.detach().double().cpu(), feat.grad.double().cpu()] + [t.grad.double().cpu() for t in params]
    # ---- the reference: oracle ROIAlign (the reference's fp32 operator) around a float64 evaluation of the definitions
    box = torch.from_numpy(oracle_roi.roi_align_fwd(feat_np, rois, 7, 1 / 16.0, 0, True)).double().requires_grad_(True)     # [K,C,7,7]
    mk = torch.from_numpy(masks_np).double()[:, None]
    cat = torch.cat([box, box * mk], 1)                                                          # resnet50.py:131-134
    p64 = [t.detach().double().cpu().requires_grad_(True) for t in params]
    z0 = torch.nn.functional.conv2d(cat, p64[0], p64[1], padding=1)
    z1 = torch.nn.functional.linear((z0 * m0).reshape(K, -1), p64[2], p64[3])                    # (.view(batch, -1) on NCHW: (c, h, w))
    z2 = torch.nn.functional.linear(z1 * m1, p64[4], p64[5])
    want = z2 * m2
    want.backward(dy.double().cpu())
    dfeat = oracle_roi.roi_align_bwd(box.grad.float().numpy(), rois, (1, C, H, W), 7, 1 / 16.0, 0, True)      # float64 accumulation
    ref = [want.detach(), torch.from_numpy(dfeat)] + [t.grad for t in p64]
    # the forward's own ReLU decisions agree with the float64 pre-activations except within rounding of zero
    flips = int(((z0.detach() > 0).double() != m0).sum())
    assert flips <= max(2, int(2e-5 * m0.numel())), flips
    names = ["seg_x", "dfeat", "wc", "bc", "w1", "b1", "w2", "b2"]
    worst = {}
    for a, b, name in zip(got, ref, names):
        if name == "seg_x":
            e = float((a - b).abs().max() / b.abs().max())
        else:
            e = float((a - b).norm() / b.norm())
        worst[name] = e
        assert e < 1e-5, "%s deviates from the oracle / float64 evaluation by %.3g" % (name, e)
    from cases import record_deviation
    record_deviation("fused_box_head_vs_oracle_fp64[C=%d,%dx%d,K=%d]" % (C, H, W, K), worst)


def test_maskfuse_weight_gradients_in_restricted_passes(dev):
    """torch.autograd.grad(...) towards MaskFuse's weights and .backward(inputs=[...]) are not complete passes: the node's weight
    gradients must come back THROUGH autograd (not be launched late / installed as .grad at the end of the pass), and a pass that
    does not ask for the weights must leave their .grad alone.  (the defer block used to run in such passes -
    autograd.grad raised 'appears to not have been used', .backward(inputs=[cat]) wrote the weights' .grad.)"""
    from cim_amd.ops import gemm, maskfuse_pair, pair
    torch.manual_seed(11)
    R, C = 37, 128
    conv = torch.nn.Conv2d(2 * C, 256, 3, padding=1).to(dev)
    fc1, fc2 = torch.nn.Linear(256 * 49, 128).to(dev), torch.nn.Linear(128, 96).to(dev)
    params = [conv.weight, conv.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias]
    cat0 = torch.randn(R, 2 * C, 7, 7, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(R, 96, device=dev)
    fa = pair.amax_of(cat0)

    def fresh():
        cat = cat0.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
        for t in params:
            t.grad = None
        return cat, maskfuse_pair.maskfuse_head(cat, conv, fc1, fc2, fa)

    # the complete pass: the reference values
    cat, out = fresh()
    out.backward(dy)
    gemm.join_side()
    torch.cuda.synchronize()
    want = dict(cat=cat.grad.clone(), wc=conv.weight.grad.clone(), w1=fc1.weight.grad.clone(), w2=fc2.weight.grad.clone(),
                b1=fc1.bias.grad.clone())
    # (a) torch.autograd.grad towards two of the weights (and the input): returned, equal, nothing installed or left pending
    cat, out = fresh()
    g1, gc, gcat = torch.autograd.grad(out, [fc1.weight, conv.weight, cat], dy)
    torch.cuda.synchronize()
    assert torch.equal(g1, want["w1"]) and torch.equal(gc, want["wc"]) and torch.equal(gcat, want["cat"])
    assert all(t.grad is None for t in params) and not gemm._PENDING_IDS
    # (b) towards ONE weight only
    cat, out = fresh()
    g2, = torch.autograd.grad(out, [fc2.weight], dy)
    torch.cuda.synchronize()
    assert torch.equal(g2, want["w2"])
    assert all(t.grad is None for t in params) and not gemm._PENDING_IDS
    # (c) .backward(inputs=[cat]): only cat.grad is written
    cat, out = fresh()
    out.backward(dy, inputs=[cat])
    gemm.join_side()
    torch.cuda.synchronize()
    assert torch.equal(cat.grad, want["cat"])
    assert all(t.grad is None for t in params), [n for n, t in zip("wc bc w1 b1 w2 b2".split(), params) if t.grad is not None]
    # (d) .backward(inputs=[fc1.weight, fc1.bias]): those two, nothing else
    cat, out = fresh()
    out.backward(dy, inputs=[fc1.weight, fc1.bias])
    gemm.join_side()
    torch.cuda.synchronize()
    assert torch.equal(fc1.weight.grad, want["w1"]) and torch.equal(fc1.bias.grad, want["b1"])
    assert cat.grad is None and conv.weight.grad is None and fc2.weight.grad is None
    # and a complete pass afterwards still takes the deferred schedule and gives the same numbers
    cat, out = fresh()
    out.backward(dy)
    gemm.join_side()
    torch.cuda.synchronize()
    assert torch.equal(conv.weight.grad, want["wc"]) and torch.equal(fc1.weight.grad, want["w1"]) and torch.equal(cat.grad, want["cat"])


def test_scale_sources_of_abi_13(dev):
    """cim_pair_amax takes any n (tail elements), cim_pair_scales(reduce_all) takes the maximum of all words (a weight's row maxima ->
    its one scale), cim_wino7_pair_scales multiplies max |d| by max(1, max |mask|) (MaskFuse's concat: lib/modeling/resnet50.py:131-134)."""
    from cim_amd import _lib
    from cim_amd.ops import pair
    st = _lib.stream_ptr()
    g = torch.Generator().manual_seed(5)
    for n in (49 * 850, 7, 4096 + 3):
        x = torch.randn(n, generator=g).to(dev)
        x[n - 1] = -37.5                                           # the maximum sits in the tail
        got = pair.amax_of(x).view(torch.float32)
        assert float(got) == 37.5, (n, float(got))
    # reduce_all: one scale from the maximum of the row maxima
    rows = (torch.rand(300, generator=g) * 3.0).to(dev)
    rows[123] = 5.0
    words = rows.view(torch.int32)
    s_all = pair.scales_from(words, 1, reduce_all=True)
    s_one = pair.scales_from(torch.tensor([5.0], device=dev).view(torch.int32), 1)
    assert torch.equal(s_all, s_one) and float(s_one) == 2.0 ** (14 - 2)
    # the Winograd scales with a mask maximum: equal to the scales of amax * max(1, max |mask|)
    a = torch.tensor([3.0], device=dev).view(torch.int32)
    for mmax, eff in ((0.5, 3.0), (1.0, 3.0), (2.5, 7.5)):
        m = torch.tensor([mmax], device=dev).view(torch.int32)
        e = torch.tensor([eff], device=dev).view(torch.int32)
        s0, s1 = torch.empty(121, device=dev), torch.empty(121, device=dev)
        _lib.call("cim_wino7_pair_scales", a.data_ptr(), 1, m.data_ptr(), 0, s0.data_ptr(), st)
        _lib.call("cim_wino7_pair_scales", e.data_ptr(), 1, None, 0, s1.data_ptr(), st)
        assert torch.equal(s0, s1), mmax
    # several amax words: their maximum
    many = torch.tensor([0.25, 3.0, 1.0], device=dev).view(torch.int32)
    s2 = torch.empty(121, device=dev)
    _lib.call("cim_wino7_pair_scales", many.data_ptr(), 3, None, 1, s2.data_ptr(), st)
    s3 = torch.empty(121, device=dev)
    _lib.call("cim_wino7_pair_scales", a.data_ptr(), 1, None, 1, s3.data_ptr(), st)
    assert torch.equal(s2, s3)
