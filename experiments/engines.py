"""The superseded arithmetic engines and convolution algorithms of the MaskFuse contractions, with autograd - TEST INFRASTRUCTURE
(moved out of cim_amd/ops/gemm.py in round 5; see experiments/__init__.py).

`linear(x, weight, bias, relu)` == F.relu(F.linear(x, weight, bias)) and `conv3x3(x, weight, bias, relu)` ==
F.relu(F.conv2d(x, weight, bias, padding=1)) for the N x 7 x 7 ROI maps of /root/reference/lib/modeling/resnet50.py:104-110,135-136 on
  ENGINE     "f16x2" (scaled two-term fp16 split, 3 MFMA products; operand scales per row / column), "bf16x3" (exact three-term bf16
             split, 6 products), "fp32" (f32 MFMA multiplies)
  CONV_ALGO  "winograd7" (mixed 4 + 3 tiling, fp32 transforms), "winograd4" (F(4x4,3x3)), "winograd" (F(2x2,3x3)), "direct"
Both are plain module attributes that a test sets (no environment switches).  `maskfuse_forward` is the box head on these per-layer
Functions: what cim_amd.modeling.maskfuse.MaskFuse.forward did for every engine but the pair engine.
"""
import weakref

import torch
from torch.autograd import Function

from cim_amd.ops import gemm as _infra          # side stream, weight-scale registry (shared with the product's optimizer)
from . import _lib

ENGINE = "f16x2"
CONV_ALGO = "winograd7"
OVERLAP = True
_side_stream = _infra._side_stream
_registered_scales = _infra._registered_scales


def _ws(m, n, splits, like):
    return torch.empty(splits * m * n, dtype=torch.float32, device=like.device) if splits > 1 else None


def engine_code():
    """`engine` argument of the cim_gemm_f32 / cim_conv3x3_f32 entry points (operands without scales): 0 = f32 MFMA multiplies,
    1 = the exact three-term bf16 split.  The library keeps no engine state: every call says which arithmetic it wants."""
    return 0 if ENGINE == "fp32" else 1


def _zeros_i32(dev, *sizes):
    """Zeroed int32 arrays of the given sizes carved out of ONE allocation / fill (0 -> None)."""
    pad = [(n + 3) & ~3 for n in sizes]
    buf = torch.zeros(max(sum(pad), 1), dtype=torch.int32, device=dev)
    out, o = [], 0
    for n, p in zip(sizes, pad):
        out.append(buf[o:o + n] if n else None)
        o += p
    return out


def amax(x, rows, cols, ld, want_rows=False, want_cols=False, batch=1, bs=0, out=None):
    """|max| bit patterns of a stored [batch][rows][ld] fp32 matrix: per row (over its columns) and / or per
    column (over its rows), ONE pass over x.  These are the operand scales of the f16x2 engine: an operand
    read K-contiguously takes the per-row array, one read M/N-contiguously the per-column array.
    out: (row array, col array) of pre-zeroed int32 storage (see _zeros_i32), else allocated here."""
    if out is None:
        out = _zeros_i32(x.device, batch * rows if want_rows else 0, batch * cols if want_cols else 0)
    ra, ca = out
    _lib.call("cim_amax_rowcol", x.data_ptr(), rows, cols, ld, batch, bs, _lib.ptr(ra), _lib.ptr(ca), _lib.stream_ptr())
    return ra, ca


def gemm(a, b, m, n, k, lda, ldb, a_mcontig=False, b_kcontig=False, bias=None, relu=False, out=None,
         a_amax=None, b_amax=None):
    """C[m,n] = A.B (+bias)(ReLU).  a/b are dense device tensors interpreted by the layout flags.
    a_amax / b_amax: operand scales from amax() when the caller already has them (f16x2 engine)."""
    if not a.is_cuda:
        raise _lib.CimHipError("experiments.engines: CUDA/HIP tensors required")
    c = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=a.device)
    if ENGINE != "f16x2":
        splits = _lib.call("cim_gemm_f32_splits", m, n, k, engine_code())
        ws = _ws(m, n, splits, a)
        _lib.call("cim_gemm_f32", a.data_ptr(), b.data_ptr(), c.data_ptr(), _lib.ptr(bias), m, n, k, lda, ldb, n,
                  int(a_mcontig), int(b_kcontig), int(relu), splits, _lib.ptr(ws), engine_code(), _lib.stream_ptr())
        return c
    if a_amax is None:
        a_amax = amax(a, k, m, lda, want_cols=True)[1] if a_mcontig else amax(a, m, k, lda, want_rows=True)[0]
    if b_amax is None:
        b_amax = amax(b, n, k, ldb, want_rows=True)[0] if b_kcontig else amax(b, k, n, ldb, want_cols=True)[1]
    splits = _lib.call("cim_gemm_f16x2_splits", m, n, k)
    ws = _ws(m, n, splits, a)
    _lib.call("cim_gemm_f16x2", a.data_ptr(), b.data_ptr(), c.data_ptr(), _lib.ptr(bias), m, n, k, lda, ldb, n,
              int(a_mcontig), int(b_kcontig), int(relu), splits, _lib.ptr(ws), a_amax.data_ptr(), b_amax.data_ptr(),
              _lib.stream_ptr())
    return c



class LinearFunction(Function):
    """y = relu?(x @ w.T + b); x [M,K], w [N,K] (nn.Linear layout)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x = x.contiguous()
        w = w.contiguous()
        m, k = x.shape
        n = w.shape[0]
        xr = xc = wr = wc = None
        if ENGINE == "f16x2":        # one pass per operand: the row scales serve this product, the column scales the backward
            reg = _registered_scales(w, n, k)        # by-product of the optimizer step, if it ran cim_amd.optim.SGD
            z = _zeros_i32(x.device, m, k if ctx.needs_input_grad[1] else 0,
                           0 if reg else n, 0 if reg or not ctx.needs_input_grad[0] else k)
            xr, xc = amax(x, m, k, k, True, ctx.needs_input_grad[1], out=z[0:2])
            wr, wc = reg if reg else amax(w, n, k, k, True, ctx.needs_input_grad[0], out=z[2:4])
        y = gemm(x, w, m, n, k, k, k, b_kcontig=True, bias=b, relu=relu, a_amax=xr, b_amax=wr)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.scales = (xc, wc)
        ctx.relu = relu
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        m, k = x.shape
        n = w.shape[0]
        dy = dy.contiguous()
        if ctx.relu:
            dy = dy * (y > 0)
        dx = dw = db = None
        xc, wc = ctx.scales
        dr = dc = None
        if ENGINE == "f16x2":
            dr, dc = amax(dy, m, n, n, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        both = OVERLAP and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
        if both:
            cur, side = torch.cuda.current_stream(), _side_stream(dy.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                dx = gemm(dy, w, m, k, n, n, k, a_amax=dr, b_amax=wc)
            dw = gemm(dy, x, n, k, m, n, k, a_mcontig=True, a_amax=dc, b_amax=xc)
            cur.wait_stream(side)
            dx.record_stream(cur)
        else:
            if ctx.needs_input_grad[0]:
                dx = gemm(dy, w, m, k, n, n, k, a_amax=dr, b_amax=wc)              # dY[M,N] . W[N,K]
            if ctx.needs_input_grad[1]:
                dw = gemm(dy, x, n, k, m, n, k, a_mcontig=True, a_amax=dc, b_amax=xc)   # dY^T[N,M] . X[M,K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=0)
        return dx, dw, db, None


def _bgemm(a, b, c, m, n, k, lda, ldb, a_mcontig, batch, a_bs, b_bs, c_bs, a_amax=None, b_amax=None, b_kcontig=False):
    """`batch` GEMMs C[i] = A[i] . B[i] (B N-contiguous, or [n][k] K-contiguous).  a_amax: per-row scales of A
    ([batch, m]: rows of a K-contiguous A, columns of the stored matrix for an M-contiguous one); b_amax: per-column
    scales [batch, n]."""
    if ENGINE != "f16x2":
        _lib.call("cim_gemm_f32_batched", a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, lda, ldb, n,
                  int(a_mcontig), int(b_kcontig), batch, a_bs, b_bs, c_bs, engine_code(), _lib.stream_ptr())
        return
    if a_amax is None:
        a_amax = (amax(a, k, m, lda, want_cols=True, batch=batch, bs=a_bs)[1] if a_mcontig
                  else amax(a, m, k, lda, want_rows=True, batch=batch, bs=a_bs)[0])
    if b_amax is None:
        b_amax = (amax(b, n, k, ldb, want_rows=True, batch=batch, bs=b_bs)[0] if b_kcontig
                  else amax(b, k, n, ldb, want_cols=True, batch=batch, bs=b_bs)[1])
    _lib.call("cim_gemm_f16x2_batched", a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, lda, ldb, n,
              int(a_mcontig), int(b_kcontig), batch, a_bs, b_bs, c_bs, a_amax.data_ptr(), b_amax.data_ptr(),
              _lib.stream_ptr())


def _bounds(amax_in, n, group, kind, npos, dev):
    """[npos, n] column-scale bounds of a Winograd-domain operand from the |max| of the untransformed tensor."""
    out = torch.empty(npos * n, dtype=torch.int32, device=dev)
    _lib.call("cim_wino_scale_bounds", amax_in.data_ptr(), out.data_ptr(), n, group, kind, 7 if npos == 121 else 4,
              _lib.stream_ptr())
    return out


# "winograd7" (default: a 4-wide + a 3-wide tile per axis of the 7 x 7 map, 121 positions; other map sizes take winograd4)
# | "winograd4" (F(4x4,3x3), 2 x 2 tiles of 36 positions) | "winograd" (F(2x2,3x3)) | "direct"

def _wino_geometry(algo, p, r):
    """(tile code of the C entry points, positions, GEMM rows) of a Winograd algorithm on r maps of p x p."""
    if algo == "winograd7":
        return 7, 121, r                       # one tile of each of the 4 types per map
    tile = 4 if algo == "winograd4" else 2
    t = (p + tile - 1) // tile
    return tile, (tile + 2) ** 2, r * t * t



class Conv3x3Function(Function):
    """y = relu?(conv2d(x, w, b, padding=1)) on channels-last ROI maps.
    x: logical [R,Cin,P,P] in torch.channels_last (physical [R,P,P,Cin]); w [Cout,Cin,3,3].

    Default algorithm: Winograd F(2x2,3x3) in fp32 - input / filter transforms, 16 batched exact-fp32
    MFMA GEMMs, output transform (cim_amd/csrc/winograd.hip); the transformed input V is kept for
    the weight gradient (F(3x3,2x2) shares its B^T).  `CIM_CONV_ALGO=direct` selects the implicit
    GEMM on the untransformed data (1.72x more multiplies, ~3x closer to the fp64 result)."""

    @staticmethod
    def forward(ctx, x, w, b, relu, x_col_amax=None, flatten_chw=False):
        x = x.contiguous(memory_format=torch.channels_last)
        r, cin, p, _ = x.shape
        cout = w.shape[0]
        w = w.contiguous()
        dev = x.device
        st = _lib.stream_ptr()
        y = torch.empty((r, p, p, cout), dtype=torch.float32, device=dev)
        ctx.algo = CONV_ALGO if (cin % 4 == 0 and cout % 4 == 0) else "direct"
        if ctx.algo == "winograd7" and p != 7:
            ctx.algo = "winograd4"
        ctx.tile = tile = 0
        V = None
        if ctx.algo.startswith("winograd"):
            ctx.tile, npos, mt = _wino_geometry(ctx.algo, p, r)
            tile = ctx.tile
            V = torch.empty((npos, mt, cin), dtype=torch.float32, device=dev)
            U = torch.empty((npos, cin, cout), dtype=torch.float32, device=dev)
            M = torch.empty((npos, mt, cout), dtype=torch.float32, device=dev)
            vr = uc = None
            ctx.fused_scales = ENGINE == "f16x2" and tile in (4, 7)
            if ctx.fused_scales:
                # operand scales of the f16x2 engine without a pass over the 1.2 GB transformed tensors: all of them are
                # upper BOUNDS from the |max| of the untransformed tensors (x per tile for the rows of V - inside the
                # transform kernel -, x per channel, w per filter) times the transform's absolute row sums
                vr = torch.empty(npos * mt, dtype=torch.int32, device=dev)
                _lib.call("cim_wino_input_transform_amax", x.data_ptr(), V.data_ptr(), vr.data_ptr(), r, p, cin, tile, st)
                need_xc = ctx.needs_input_grad[1] and x_col_amax is None
                reg = _registered_scales(w, cout, cin * 9)
                z = _zeros_i32(dev, 0 if reg else cout, 0 if reg or not ctx.needs_input_grad[0] else cin * 9, cin if need_xc else 0)
                w_rows, w_cols = reg if reg else amax(w, cout, cin * 9, cin * 9, True, ctx.needs_input_grad[0], out=z[0:2])
                uc = _bounds(w_rows, cout, 1, 1, npos, dev)
                ctx.w_cols = w_cols
                if ctx.needs_input_grad[1]:
                    # per-channel |max| of x: computed here, or an upper bound handed in by the producer of x
                    xc = x_col_amax if x_col_amax is not None else amax(x, r * p * p, cin, cin, want_cols=True, out=(None, z[2]))[1]
                    ctx.v_cols = _bounds(xc, cin, 1, 0, npos, dev)
            else:
                _lib.call("cim_wino_input_transform", x.data_ptr(), V.data_ptr(), r, p, cin, tile, st)
                if ENGINE == "f16x2":    # one pass over V: row scales for this product, column scales for the weight gradient
                    vr, ctx.v_cols = amax(V, mt, cin, cin, True, ctx.needs_input_grad[1], batch=npos, bs=mt * cin)
            _lib.call("cim_wino_filter_transform", w.data_ptr(), U.data_ptr(), cout, cin, 0, tile, st)
            _bgemm(V, U, M, mt, cout, cin, cin, cout, False, npos, mt * cin, cin * cout, mt * cout, a_amax=vr, b_amax=uc)
            _lib.call("cim_wino_output_transform", M.data_ptr(), _lib.ptr(b), y.data_ptr(), r, p, cout, int(relu), tile, st)
        else:
            whwio = w.permute(2, 3, 1, 0).contiguous()
            _lib.call("cim_conv3x3_f32", x.data_ptr(), whwio.data_ptr(), _lib.ptr(b), y.data_ptr(), r, p, cin, cout,
                      int(relu), engine_code(), st)
        # mixed tiling: the data gradient is evaluated as the adjoint of this product and reuses U (no second filter transform)
        ctx.save_for_backward(x, w, y if relu else None, V, U if (tile == 7 and ctx.needs_input_grad[0]) else None)
        ctx.relu = relu
        ctx.has_bias = b is not None
        ctx.flatten = bool(flatten_chw)
        if ctx.flatten:      # the reference's `.view(N, -1)` of the NCHW output: (c, h, w) order, one transposing pass
            flat = torch.empty((r, cout * p * p), dtype=torch.float32, device=dev)
            _lib.call("cim_flatten_chw", y.data_ptr(), None, flat.data_ptr(), r, p * p, cout, 0, st)
            return flat
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        x, w, y, V, U = ctx.saved_tensors
        r, cin, p, _ = x.shape
        cout = w.shape[0]
        dev = x.device
        st = _lib.stream_ptr()
        if ctx.flatten:      # transpose back to channels-last fused with the ReLU mask
            dflat = dy.contiguous()
            dy = torch.empty((r, p, p, cout), dtype=torch.float32, device=dev)
            _lib.call("cim_flatten_chw", dflat.data_ptr(), _lib.ptr(y) if ctx.relu else None, dy.data_ptr(), r, p * p, cout, 1, st)
        else:
            dy = dy.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)      # physical [R,P,P,Cout]
            if ctx.relu:
                dy = dy * (y > 0)
            dy = dy.contiguous()
        dx = dw = db = None
        wino = ctx.algo.startswith("winograd")
        tile = ctx.tile
        npos = mt = 0
        if wino:
            _, npos, mt = _wino_geometry(ctx.algo, p, r)
        fused = wino and getattr(ctx, "fused_scales", False)

        def data_grad():
            st = _lib.stream_ptr()
            dxp = torch.empty((r, p, p, cin), dtype=torch.float32, device=dev)
            if wino and tile == 7 and U is not None:
                # adjoint of the forward: E = A dy A^T, Md[pos] = E[pos] . U[pos]^T (U read K-contiguously), dx = overlap-add B Md B^T
                E = torch.empty((npos, mt, cout), dtype=torch.float32, device=dev)
                M2 = torch.empty((npos, mt, cin), dtype=torch.float32, device=dev)
                er = uc2 = None
                if fused:
                    er = torch.empty(npos * mt, dtype=torch.int32, device=dev)
                    uc2 = _bounds(ctx.w_cols, cin, 9, 1, npos, dev)      # per input channel: max over (co, taps)
                _lib.call("cim_wino_dy_adjoint_transform", dy.data_ptr(), E.data_ptr(), _lib.ptr(er), r, p, cout, tile, st)
                _bgemm(E, U, M2, mt, cin, cout, cout, cout, False, npos, mt * cout, cin * cout, mt * cin,
                       a_amax=er, b_amax=uc2, b_kcontig=True)
                _lib.call("cim_wino_dx_adjoint_output", M2.data_ptr(), dxp.data_ptr(), r, p, cin, tile, st)
            elif wino:
                # data gradient = the same convolution of dY with the 180-degree rotated, in/out-swapped filter
                Vd = torch.empty((npos, mt, cout), dtype=torch.float32, device=dev)
                U2 = torch.empty((npos, cout, cin), dtype=torch.float32, device=dev)
                M2 = torch.empty((npos, mt, cin), dtype=torch.float32, device=dev)
                dr = u2c = None
                if fused:
                    dr = torch.empty(npos * mt, dtype=torch.int32, device=dev)
                    _lib.call("cim_wino_input_transform_amax", dy.data_ptr(), Vd.data_ptr(), dr.data_ptr(), r, p, cout, tile, st)
                    u2c = _bounds(ctx.w_cols, cin, 9, 1, npos, dev)       # per input channel: max over (co, taps)
                else:
                    _lib.call("cim_wino_input_transform", dy.data_ptr(), Vd.data_ptr(), r, p, cout, tile, st)
                _lib.call("cim_wino_filter_transform", w.data_ptr(), U2.data_ptr(), cout, cin, 1, tile, st)
                _bgemm(Vd, U2, M2, mt, cin, cout, cout, cin, False, npos, mt * cout, cout * cin, mt * cin,
                       a_amax=dr, b_amax=u2c)
                _lib.call("cim_wino_output_transform", M2.data_ptr(), None, dxp.data_ptr(), r, p, cin, 0, tile, st)
            else:
                w2 = w.flip(2, 3).permute(2, 3, 0, 1).contiguous()                     # [3,3,Cout,Cin]
                _lib.call("cim_conv3x3_f32", dy.data_ptr(), w2.data_ptr(), None, dxp.data_ptr(), r, p, cout, cin, 0, engine_code(), st)
            return dxp.permute(0, 3, 1, 2)

        def weight_grad():
            st = _lib.stream_ptr()
            if wino:
                D = torch.empty((npos, mt, cout), dtype=torch.float32, device=dev)
                dU = torch.empty((npos, cin, cout), dtype=torch.float32, device=dev)
                dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dev)
                _lib.call("cim_wino_dy_transform", dy.data_ptr(), D.data_ptr(), r, p, cout, tile, st)
                dc = _bounds(amax(dy, r * p * p, cout, cout, want_cols=True)[1], cout, 1, 2, npos, dev) if fused else None
                # dU[pos] = V[pos]^T . D[pos]:  A = V[pos] read M-contiguously (element (ci, m) at V[m*Cin + ci])
                _bgemm(V, D, dU, cin, cout, mt, cin, cout, True, npos, mt * cin, mt * cout, cin * cout,
                       a_amax=getattr(ctx, "v_cols", None), b_amax=dc)
                _lib.call("cim_wino_wgrad_output", dU.data_ptr(), dw.data_ptr(), cout, cin, tile, st)
            else:
                m, n, k = 9 * cin, cout, r * p * p
                splits = _lib.call("cim_gemm_f32_splits", m, n, k, engine_code())
                ws = _ws(m, n, splits, x)
                dwh = torch.empty((3, 3, cin, cout), dtype=torch.float32, device=dev)
                _lib.call("cim_conv3x3_wgrad_f32", x.data_ptr(), dy.data_ptr(), dwh.data_ptr(), r, p, cin, cout, splits,
                          _lib.ptr(ws), engine_code(), st)
                dw = dwh.permute(3, 2, 0, 1)
            return dw

        if OVERLAP and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # the two gradient paths are independent: on two HIP streams the weight-gradient GEMM (1152 tiles = 4.5
            # rounds of 256 CUs) and the data-gradient GEMM fill each other's partly empty rounds, and the HBM-bound
            # transform kernels of one path run under the MFMA-bound GEMM of the other
            cur, side = torch.cuda.current_stream(), _side_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                dw = weight_grad()
            dx = data_grad()
            cur.wait_stream(side)
            dw.record_stream(cur)
        else:
            if ctx.needs_input_grad[0]:
                dx = data_grad()
            if ctx.needs_input_grad[1]:
                dw = weight_grad()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=(0, 1, 2))
        return dx, dw, db, None, None, None


def linear(x, weight, bias=None, relu=False):
    return LinearFunction.apply(x, weight, bias, relu)


def conv3x3(x, weight, bias=None, relu=False, x_col_amax=None, flatten_chw=False):
    """x_col_amax: optional int32 [Cin] bit patterns of an UPPER BOUND of max |x[:, c, :, :]| (f16x2 engine: saves the
    pass over x that derives the weight-gradient operand scales).
    flatten_chw: return the output flattened as [R, Cout*P*P] in (c, h, w) order - what `.view(R, -1)` gives on the
    reference's NCHW tensor - through one transposing kernel each way (needs Cout % 64 == 0, P*P <= 64)."""
    if flatten_chw and not (weight.shape[0] % 64 == 0 and x.shape[-1] * x.shape[-2] <= 64):
        y = Conv3x3Function.apply(x, weight, bias, relu, x_col_amax, False)
        return y.contiguous(memory_format=torch.contiguous_format).view(y.size(0), -1)
    return Conv3x3Function.apply(x, weight, bias, relu, x_col_amax, flatten_chw)


def maskfuse_forward(module, x, rois, masks):
    """MaskFuse.forward (/root/reference/lib/modeling/resnet50.py:120-138) on the per-layer Functions above: ROIAlign + mask multiply +
    concat (the product's kernel), then conv3x3 -> flatten -> fc1 -> fc2 on ENGINE / CONV_ALGO."""
    from cim_amd.core.config import cfg
    from cim_amd.ops import roi_align_maskcat
    cat = roi_align_maskcat(x, rois, masks, cfg.FAST_RCNN.ROI_XFORM_RESOLUTION, module.spatial_scale,
                            cfg.FAST_RCNN.ROI_XFORM_SAMPLING_RATIO, aligned=True)
    conv = module.mask_branch[0]
    fc1, fc2 = module.seg_fc[0], module.seg_fc[2]
    # ROIAlign averages feature pixels, so per channel max |box_x| <= max |x| over the map and max |box_x * mask| <= that times
    # max |mask| ({0,1} masks: 1): a 6 MB pass instead of one over the 400 MB cat tensor for the conv's weight-gradient scales
    xc = None
    if ENGINE == "f16x2" and CONV_ALGO in ("winograd4", "winograd7"):
        xn = x.detach().contiguous(memory_format=torch.channels_last)
        fa = amax(xn, xn.size(0) * xn.size(2) * xn.size(3), xn.size(1), xn.size(1), want_cols=True)[1]
        fm = (fa.view(torch.float32) * masks.detach().abs().max().clamp(min=1.0)).view(torch.int32)
        xc = torch.cat([fa, fm])
    y = conv3x3(cat, conv.weight, conv.bias, relu=True, x_col_amax=xc, flatten_chw=True)
    return linear(linear(y, fc1.weight, fc1.bias, relu=True), fc2.weight, fc2.bias, relu=True)
