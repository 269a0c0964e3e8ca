"""MaskFuse's contractions on the f16x2p engine: conv3x3 (mixed 4 + 3 Winograd tiling) -> (c, h, w) flatten -> fc1 -> fc2 as
ONE autograd Function whose intermediate operands are pair images written by their producers.

Replaces the `mask_branch` / `seg_fc` part of MaskFuse.forward, /root/reference/lib/modeling/resnet50.py:104-110,131-137,
and its autograd backward - the ONE engine and ONE algorithm of the product (the per-layer Functions on the superseded engines are
test infrastructure: experiments/engines.py).  Every operand is split by its producer (cim_amd/csrc/gemm_pair.hip header):

  forward   cat --wino7_input_pair--> V'   (scale per position from max |feature map|)
            w   --wino7_filter_pair--> U'  (side stream, under the backbone forward; scale from max |w|)
            M = V' . U'^T  ->  wino7_output (+bias, ReLU, max |y|)  ->  y fp32 (kept: ReLU mask of the backward)
            y --flatten_chw_pair--> X'     (never exists in fp32)
            Y1 = relu(X' . W1'^T + b1) (max |Y1| from the split-K reduce) -> Y1' ;  Y2 = relu(Y1' . W2'^T + b2)
  backward  dY2 -> dY2' (mask fused into the split);  dW2 = dY2'^T . Y1' | dY1 = dY2' . W2'
            dY1 -> dY1' ;  dW1 = dY1'^T . X' | dX = dY1' . W1'  (max |dX|)
            dX --flatten backward (ReLU mask)--> dy --wino7_dy_pair--> E', D'
            dU = V'^T . D' -> dW  |  Md = E' . U' -> dcat      (weight gradients on the side stream)
Every image is written once and read by two products (contracted over its columns by one, over its rows by the other).
"""
import torch
from torch.autograd import Function

from .. import _lib
from . import chain
from . import gemm as G
from . import pair

NPOS = 121
# Measured experiment (maskfuse_pair.DEFER_DW with gemm.HIGH_PRIO, both MODULE ATTRIBUTES - no environment switches): the node's weight-gradient GEMMs (fc2, fc1, conv: ~2.5 ms of MFMA
# work at cfg2) are only needed by the optimizer, so they can be joined once, at the end of the backward pass (ops/gemm.py:
# defer_side_join), and keep running on the normal-priority side stream while a high-priority main chain
# (model_builder.Generalized_RCNN.forward) goes on to the ROIAlign and backbone backward.  Measured at cfg2 (same box, interleaved,
# twice): 15.27 / 15.23 ms as shipped, 15.21 / 15.24 with both, 15.21 deferred only, 15.45 priority only - NO gain: a running
# 256 x 256 tile holds its CU for ~170 us whatever the queue priority, and the data-gradient chain still shares the chip half and half.
# (A first measurement said 14.40 ms: its runs were corrupted - buffers the side stream still read were handed out by the caching
# allocator, weights went NaN, and NaN operands draw less power and run the MFMAs at a higher clock.  The lifetimes are recorded now
# (record_stream below) and tests/test_gpu_parity.py::test_stream_scheduling_does_not_change_a_training_run runs 40 optimizer steps
# with the options on and off, bit-equal.)
DEFER_DW = True
# With the deferred join: the three weight-gradient products are LAUNCHED at the end of this node's backward (behind its
# data-gradient chain, not beside it) in consecutive launches of DW_WGS workgroups (the `max_workgroups` argument of cim_gemm_pair*; a workgroup owns its
# CU) - they then run beside the ROIAlign and backbone backward, whose small kernels get on the chip between two launches instead
# of queueing behind several thousand resident-for-70-us workgroups.  Measured at cfg2, interleaved runs on one box, ms per step:
# beside the data gradients, joined at the node (round 2's schedule) 14.84 / 14.53 / 14.78; late in launches of 256 14.25 / 14.23 /
# 14.19; late as whole products 14.44 / 14.62 / 14.31; late in launches of 96 / 128 / 192 / 224 (part of the chip left free):
# 16.9 / 15.8 / 15.1 / 14.9 - the products are MFMA-bound, CUs withheld from them are simply lost.
# 0: launched where their operands are ready, uncapped (beside the data-gradient products).
DW_WGS = 256
# Round 6: the late products run in the CO-RESIDENT form of the pair GEMM (include/cim_hip.h: `form` = 1 - 128 x 256 tiles of four waves, a
# ring of five 16-k slabs: half of a CU's registers and 40 KB of its LDS stay free), as whole products (DW_FORM1_WGS = 0: no launch cap) -
# the backbone's backward kernels (35 KB of LDS, 64 VGPRs) then run on the SAME CUs beside them instead of taking turns with
# 256-workgroup launches that own the chip.  Alone the form is slower (0.40 against 0.44 of the f16 peak on the weight gradients), beside
# the backbone's chains the last phase of the backward is shorter: 4.17 -> 3.86 ms (tools/diag_late.py: the chains end 0.9 ms earlier, the
# late stream 0.12 ms earlier), step 14.41 -> 14.20 ms (same box, three interleaved runs each; bench.py --dw-form 0 | 1).  Same bits
# (tests/test_gpu_gemm_pair.py).  0: the 256 x 256 form in launches of DW_WGS workgroups (round 4's schedule).
DW_FORM = 1
DW_FORM1_WGS = 0
# which schedule the backward passes of this process took (counts per pass; bench.py prints it in extra.comm, the 2-rank tests
# assert on it): the single-process schedule (late launches of DW_WGS workgroups, postponed behind the ROIAlign backward) and the
# multi-rank one (whole products, handed to nn.DataParallel as soon as they are enqueued) are different code paths
import collections as _collections
SCHEDULE = _collections.Counter()


def supported(cat, wc, w1, w2):
    r, cin, p, q = cat.shape
    cout = wc.shape[0]
    return (cat.is_cuda and p == 7 and q == 7 and cin % 32 == 0 and cout % 64 == 0 and wc.shape[1] == cin
            and w1.shape[1] == cout * 49 and w1.shape[0] % 32 == 0 and w2.shape[1] == w1.shape[0] and w2.shape[0] % 32 == 0)


def _weight_amax(w, rows, cols):
    """int32[n] bit patterns whose maximum is max |w|: the row maxima the fused SGD kernel registered for this version of w (the
    scale kernels take their maximum themselves), else one pass (n = 1)."""
    reg = G._registered_scales(w, rows, cols)
    if reg is not None and reg[0] is not None and reg[0].is_contiguous() and reg[0].numel() <= 65536:
        return reg[0].reshape(-1)
    return pair.amax_of(w.detach())


_IMAGES = {}      # id(weight) -> (version, data_ptr, Pair, event or None, weakref to the weight: an id can be reused by another tensor)
import weakref as _weakref


def _store(w, img, ev):
    key = id(w)
    _IMAGES[key] = (w._version, w.data_ptr(), img, ev, _weakref.ref(w, lambda _r, key=key: _IMAGES.pop(key, None)))


_STALE_FC = False      # measurement aid (tools set it): never refresh the fc weights' images


def _cached(w):
    e = _IMAGES.get(id(w))
    if e is not None and e[4]() is w and ((e[0] == w._version and e[1] == w.data_ptr()) or (_STALE_FC and w.dim() == 2)):
        return e
    return None


def _conv_image(w):
    cout, cin = w.shape[0], w.shape[1]
    sc = torch.empty(NPOS, dtype=torch.float32, device=w.device)
    st = _lib.stream_ptr()
    wa = _weight_amax(w, cout, cin * 9)
    _lib.call("cim_wino7_pair_scales", wa.data_ptr(), wa.numel(), None, 1, sc.data_ptr(), st)
    U = pair.Pair(torch.empty((NPOS, cout, cin), dtype=torch.int32, device=w.device), cout, cin, NPOS, sc)
    _lib.call("cim_wino7_filter_pair", w.data_ptr(), U.buf.data_ptr(), sc.data_ptr(), cout, cin, st)
    return U


def _fc_image(w):
    n, k = w.shape
    return pair.split(w.detach(), n, k, k, scale=pair.scales_from(_weight_amax(w, n, k), 1, reduce_all=True))


def weight_image(w, conv=False):
    """Pair image of a weight for its CURRENT version: the prefetched one (the caller's stream waits for its event), else
    built now on the caller's stream."""
    e = _cached(w)
    if e is not None:
        if e[3] is not None:
            torch.cuda.current_stream(w.device).wait_event(e[3])
            e[2].record_stream(torch.cuda.current_stream(w.device))
        return e[2]
    G.wait_pending_updates(w.device)          # (built on the caller's stream: an update of w may still run on the side stream)
    with torch.no_grad():
        img = _conv_image(w.contiguous()) if conv else _fc_image(w.contiguous())
    _store(w, img, None)
    return img


FUSE_DY = True         # flatten backward + ReLU mask + both dy transforms of the convolution in one launch (else three)
PREFETCH = True        # the weights' pair images on the side stream, under the backbone forward


def prefetch_weight_images(wc, w1, w2):
    """Build the three weights' pair images on the SIDE stream now (they depend on the weights only): called before the
    backbone forward, whose small latency-bound kernels leave most of the chip idle."""
    if not (G.OVERLAP and PREFETCH and wc.is_cuda) or torch.cuda.is_current_stream_capturing():
        return
    todo = [(w, c) for w, c in ((wc, True), (w1, False), (w2, False)) if _cached(w) is None and w.is_contiguous()]
    if not todo:
        return
    dev = wc.device
    cur, side = torch.cuda.current_stream(dev), G._side_stream(dev)
    side.wait_stream(cur)                       # (the optimizer's update of the weights was enqueued on `cur`)
    with torch.cuda.stream(side), torch.no_grad():
        for w, conv in todo:
            img = _conv_image(w) if conv else _fc_image(w)
            ev = torch.cuda.Event()
            ev.record(side)
            _store(w, img, ev)


def _i32(dev):
    return torch.zeros(1, dtype=torch.int32, device=dev)


def _head_forward(ctx, V, r, cin, p, wc, bc, w1, b1, w2, b2):
    """conv (Winograd-domain product of V' with the filter image) -> flatten -> fc1 -> fc2 from the input image V'; saves what the
    backward needs on ctx."""
    cout, h1, h2 = wc.shape[0], w1.shape[0], w2.shape[0]
    dev, rp = V.buf.device, pair.pad32(r)
    st = _lib.stream_ptr()
    Up, W1p, W2p = weight_image(wc, conv=True), weight_image(w1), weight_image(w2)
    M = pair.gemm(V, Up, r, cout, cin, False, True)
    y = torch.empty((r, p, p, cout), dtype=torch.float32, device=dev)
    am = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.call("cim_wino7_output_amax", M.data_ptr(), _lib.ptr(bc), y.data_ptr(), r, cout, 1, am[0:1].data_ptr(), st)
    del M
    # flatten -> X' ; fc1 ; fc2
    Xp = pair.Pair(torch.empty((1, rp, cout * p * p), dtype=torch.int32, device=dev), r, cout * p * p, 1,
                   pair.scales_from(am[0:1], 1))
    _lib.call("cim_flatten_chw_pair", y.data_ptr(), Xp.buf.data_ptr(), Xp.scale.data_ptr(), r, rp, p * p, cout, st)
    Y1 = pair.gemm(Xp, W1p, r, h1, cout * p * p, False, True, bias=b1, relu=True, c_amax=am[1:2])
    Y1p = pair.split(Y1, r, h1, h1, scale=pair.scales_from(am[1:2], 1))
    Y2 = pair.gemm(Y1p, W2p, r, h2, h1, False, True, bias=b2, relu=True)
    ctx.save_for_backward(y, Y1, Y2, V.buf, V.scale, Up.buf, Up.scale, W1p.buf, W1p.scale, W2p.buf, W2p.scale,
                          Xp.buf, Xp.scale, Y1p.buf, Y1p.scale)
    ctx.dims = (r, cin, cout, h1, h2, p)
    ctx.has_bias = (bc is not None, b1 is not None, b2 is not None)
    ctx.weights = (wc, w1, w2)           # the Parameter objects: big gradients may be published early (see backward)
    ctx.biases = (bc, b1, b2)            # (their gradients' last reduction may leave the critical chain with the late products)
    return Y2


def _roi_align_module():
    import importlib
    return importlib.import_module(__package__ + ".roi_align")      # (the package attribute `roi_align` is the function)


def _input_scales(feat_amax, dev):
    """Scales of the convolution's input image from max |feature map| (feat_amax[0], bit pattern); a second word, if given, is
    max |mask|: ROIAlign averages feature pixels, so max |cat| <= max |x| max(1, max |mask|)."""
    sV = torch.empty(NPOS, dtype=torch.float32, device=dev)
    mul = feat_amax[1:2].data_ptr() if feat_amax.numel() > 1 else None
    _lib.call("cim_wino7_pair_scales", feat_amax.data_ptr(), 1, mul, 0, sV.data_ptr(), _lib.stream_ptr())
    return sV


class MaskFusePairFunction(Function):
    """(cat, weights) -> seg_x: the head on a materialised `cat` = [box_x, box_x * mask] tensor (ops.roi_align_maskcat)."""
    arg_ofs = 0

    @staticmethod
    def forward(ctx, cat, wc, bc, w1, b1, w2, b2, feat_amax):
        cat = cat.contiguous(memory_format=torch.channels_last)
        r, cin, p, _ = cat.shape
        dev, rp = cat.device, pair.pad32(r)
        sV = _input_scales(feat_amax, dev)
        V = pair.Pair(torch.empty((NPOS, rp, cin), dtype=torch.int32, device=dev), r, cin, NPOS, sV)
        _lib.call("cim_wino7_input_pair", cat.data_ptr(), V.buf.data_ptr(), sV.data_ptr(), r, rp, cin, _lib.stream_ptr())
        return _head_forward(ctx, V, r, cin, p, wc, bc, w1, b1, w2, b2)

    @staticmethod
    def backward(ctx, dY2):
        return _head_backward(ctx, dY2, 0) + (None,)


class MaskFuseRoiPairFunction(Function):
    """(feature map, rois, masks, weights) -> seg_x: ROIAlign, mask multiply, channel concat AND the Winograd input transform of the
    convolution in ONE launch (csrc/roi_align.hip: roi_align_wino7_pair_kernel) - `cat` is never stored; the backward ends with the
    ROIAlign backward on the head's input gradient.  /root/reference/lib/modeling/resnet50.py:120-138 as one autograd node."""

    @staticmethod
    def forward(ctx, feat, rois, masks, wc, bc, w1, b1, w2, b2, feat_amax, spatial_scale, sampling_ratio):
        RA = _roi_align_module()
        RA._check(feat, rois)
        feat = RA._nhwc(feat)
        rois = rois.to(torch.float32).contiguous()
        masks = masks.to(torch.float32).contiguous()
        B, C, H, W = feat.shape
        K, P = rois.size(0), 7
        if tuple(masks.shape) != (K, P, P):
            raise ValueError("maskfuse: masks must be [K,7,7]")
        dev, rp = feat.device, pair.pad32(K)
        sV = _input_scales(feat_amax, dev)
        V = pair.Pair(torch.empty((NPOS, rp, 2 * C), dtype=torch.int32, device=dev), K, 2 * C, NPOS, sV)
        tables = RA._workspace(K, P, H, W, dev)
        _lib.call("cim_roi_align_wino7_pair_fwd", feat.data_ptr(), rois.data_ptr(), masks.data_ptr(), V.buf.data_ptr(), sV.data_ptr(),
                  B, C, H, W, K, rp, P, float(spatial_scale), int(sampling_ratio), 1, tables.data_ptr(), _lib.stream_ptr())
        ctx.roi = (rois, masks, tables, (B, C, H, W, K, P, float(spatial_scale), int(sampling_ratio), 1))
        return _head_forward(ctx, V, K, 2 * C, P, wc, bc, w1, b1, w2, b2)

    @staticmethod
    def backward(ctx, dY2):
        RA = _roi_align_module()
        rois, masks, tables, (B, C, H, W, K, P, scale, sr, aligned) = ctx.roi
        dbox, dwc, dbc, dw1, db1, dw2, db2 = _head_backward(ctx, dY2, 2, fold_masks=masks)
        dfeat = None
        if dbox is not None:
            gcat = dbox.permute(0, 2, 3, 1)                  # [K,7,7,C] (channels-last): dcat's halves already combined with the masks
            assert gcat.is_contiguous()
            dfeat = RA._empty_nhwc(B, C, H, W, gcat)
            scratch = RA._scratch(K, B, C, H, W, gcat.device)        # (a name, not a temporary inside the call: it must outlive the LAUNCH)
            _lib.call("cim_roi_align_bwd_ws", gcat.data_ptr(), rois.data_ptr(), dfeat.data_ptr(),
                      B, C, H, W, K, P, scale, sr, aligned, tables.data_ptr(), 1, _lib.ptr(scratch), _lib.stream_ptr())
            G.run_postponed(gcat.device)         # this node's late weight gradients start behind the ROIAlign backward
        return dfeat, None, None, dwc, dbc, dw1, db1, dw2, db2, None, None, None


def _head_backward(ctx, dY2, ofs, fold_masks=None):
    """-> (d input, dwc, dbc, dw1, db1, dw2, db2); `ofs`: position of wc among the Function's inputs minus 1.
    fold_masks ([r,7,7] masks): the input gradient is returned as  dbox = dcat[:, :C] + mask * dcat[:, C:]  ([r,C,7,7], half the
    channels: the backward of the mask multiply + concat folded into the convolution's last backward stage)."""
    if True:
        (y, Y1, Y2, Vb, Vs, Ub, Us, W1b, W1s, W2b, W2s, Xb, Xs, Y1b, Y1s) = ctx.saved_tensors
        r, cin, cout, h1, h2, p = ctx.dims
        dev, rp = y.device, pair.pad32(r)
        V = pair.Pair(Vb, r, cin, NPOS, Vs)
        Up = pair.Pair(Ub, cout, cin, NPOS, Us)
        W1p, W2p = pair.Pair(W1b, h1, cout * p * p, 1, W1s), pair.Pair(W2b, h2, h1, 1, W2s)
        Xp, Y1p = pair.Pair(Xb, r, cout * p * p, 1, Xs), pair.Pair(Y1b, r, h1, 1, Y1s)
        need_x, need_wc, need_w1, need_w2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1 + ofs], ctx.needs_input_grad[3 + ofs], ctx.needs_input_grad[5 + ofs]
        cur, side = torch.cuda.current_stream(dev), G._side_stream(dev)
        overlap = G.OVERLAP and not torch.cuda.is_current_stream_capturing()

        def on_side(fn):
            """Weight gradients run beside the data-gradient chain (second HIP stream)."""
            if not overlap:
                return fn()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                out = fn()
            return out

        def publish(w, dw):
            """Several ranks: hand a finished weight gradient to nn.DataParallel NOW (its all-reduce starts on the side stream
            right behind the GEMM that wrote it) instead of returning it when the whole node is done.  -> None when published."""
            pub = publisher
            if pub is None or dw is None or not isinstance(w, torch.nn.Parameter) or not w.is_leaf:
                return dw
            if pub(w, dw, side if overlap else cur):
                published.append(dw)
                SCHEDULE["weight_gradients_published_early"] += 1
                return None
            return dw

        published = []
        publisher = G.publisher_for(ctx.weights[1])      # several ranks: this model's nn.DataParallel wrapper takes the big gradients early
        if publisher is not None and chain.restricted_pass(ctx, (1 + ofs, 3 + ofs, 5 + ofs)):
            publisher = None                             # (torch.autograd.grad towards the weights: through autograd)
        late = []               # (slot, weight, closure) of the weight gradients launched at the end (DW_WGS > 0)
        # A pass that does not accumulate into the weights - torch.autograd.grad(...) towards them, .backward(inputs=[...]) without them -
        # CAPTURES what this node returns: there the weight gradients are neither launched late nor joined at the end of the pass (they
        # would be lost, or written into a .grad the pass never asked for) - they go back through autograd, joined here.
        restricted = chain.restricted_pass(ctx, (1 + ofs, 3 + ofs, 5 + ofs))
        defer = DEFER_DW and G.DEFER_DW and not restricted
        run_late = overlap and defer and DW_WGS > 0

        def side_grad(slot, w, fn):
            if run_late:
                late.append((slot, w, fn))
                return None
            return publish(w, on_side(fn))

        wc_p, w1_p, w2_p = ctx.weights
        dcat = dwc = dbc = dw1 = db1 = dw2 = db2 = None
        # The bias gradients are column sums whose partial sums fall out of kernels of the data-gradient chain; the LAST reduction of
        # each (a 10 us launch + its boundary, three per step) leaves the chain with the postponed late products: it runs on their
        # stream and the gradient is installed at the end-of-backward join like the weights' (same sums, same order).
        needed = [w for w, need in ((w2_p, need_w2), (w1_p, need_w1), (wc_p, need_wc)) if need]
        will_postpone = (run_late and G.POSTPONE_DW and publisher is None and len(needed) > 0
                         and all(isinstance(w, torch.nn.Parameter) for w in needed))
        bc_p, b1_p, b2_p = getattr(ctx, "biases", (None, None, None))
        late_bias = []          # (bias Parameter, partial sums)

        def bias_sum(b_p, part):
            if part is None:
                return None
            if will_postpone and isinstance(b_p, torch.nn.Parameter):
                # (the result belongs to the step's stream - the optimizer reads it there: allocated here, written on the late stream)
                late_bias.append((b_p, part, torch.empty(part.shape[1], dtype=part.dtype, device=part.device)))
                return None
            return part.sum(dim=0)
        dy_keep = []
        am = torch.zeros(3, dtype=torch.int32, device=dev)
        # ---- fc2
        # (the ReLU mask is applied by the split; one launch gives max |dz| and the bias gradient's partial sums)
        dY2 = dY2.contiguous()
        db2 = bias_sum(b2_p, pair.masked_stats(dY2, Y2, am[0:1], ctx.has_bias[2] and ctx.needs_input_grad[6 + ofs], reduce=False))
        dY2p = pair.split(dY2, r, h2, h2, scale=pair.scales_from(am[0:1], 1), relu_y=Y2)
        if need_w2:
            dw2 = side_grad(2, w2_p, lambda limit=0, form=0: pair.gemm(dY2p, Y1p, h2, h1, rp, True, False, limit=limit, form=form))
        dY1 = pair.gemm(dY2p, W2p, r, h1, h2, False, False)
        # ---- fc1
        db1 = bias_sum(b1_p, pair.masked_stats(dY1, Y1, am[1:2], ctx.has_bias[1] and ctx.needs_input_grad[4 + ofs], reduce=False))
        dY1p = pair.split(dY1, r, h1, h1, scale=pair.scales_from(am[1:2], 1), relu_y=Y1)
        if need_w1:
            dw1 = side_grad(1, w1_p, lambda limit=0, form=0: pair.gemm(dY1p, Xp, h1, cout * p * p, rp, True, False, limit=limit, form=form))
        if need_x or need_wc or (ctx.has_bias[0] and ctx.needs_input_grad[2 + ofs]):
            dX = pair.gemm(dY1p, W1p, r, cout * p * p, h1, False, False, c_amax=am[2:3], balance=True)
            # ---- flatten backward + ReLU mask of the conv; conv gradients
            st = _lib.stream_ptr()
            want_dbc = ctx.has_bias[0] and ctx.needs_input_grad[2 + ofs]
            bpart = torch.empty((r, cout), dtype=torch.float32, device=dev) if want_dbc else None
            fused = FUSE_DY and cout % 256 == 0 and p == 7
            E = D = dy = None
            if fused:
                # ONE launch: dX -> (masked, transposed in LDS) -> the adjoint image E' and the weight-gradient image D'; the masked
                # gradient dy [r,7,7,cout] is never stored (csrc/winograd.hip: wino7_flatten_bwd_dy_pair_kernel)
                if need_x:
                    sE = torch.empty(NPOS, dtype=torch.float32, device=dev)
                    _lib.call("cim_wino7_pair_scales", am[2:3].data_ptr(), 1, None, 3, sE.data_ptr(), st)
                    E = pair.Pair(torch.empty((NPOS, rp, cout), dtype=torch.int32, device=dev), r, cout, NPOS, sE)
                if need_wc:
                    sD = torch.empty(NPOS, dtype=torch.float32, device=dev)
                    _lib.call("cim_wino7_pair_scales", am[2:3].data_ptr(), 1, None, 2, sD.data_ptr(), st)
                    D = pair.Pair(torch.empty((NPOS, rp, cout), dtype=torch.int32, device=dev), r, cout, NPOS, sD)
                if E is not None or D is not None:
                    _lib.call("cim_wino7_flatten_bwd_dy_pair", dX.data_ptr(), y.data_ptr(), _lib.ptr(E.buf if E is not None else None),
                              _lib.ptr(E.scale if E is not None else None), _lib.ptr(D.buf if D is not None else None),
                              _lib.ptr(D.scale if D is not None else None),
                              _lib.ptr(bpart), r, rp, cout, st)
                else:
                    fused = False
            if not fused:
                dy = torch.empty((r, p, p, cout), dtype=torch.float32, device=dev)
                _lib.call("cim_flatten_chw_bwd_bias", dX.data_ptr(), y.data_ptr(), dy.data_ptr(), _lib.ptr(bpart), r, p * p, cout, st)
            del dX
            if want_dbc:
                dbc = bias_sum(bc_p, bpart)     # per-ROI partial sums from the flatten kernel: 4 MB instead of a pass over dy
            if need_wc:
                def wgrad(limit=0, form=0, D=D):
                    st2 = _lib.stream_ptr()
                    if D is None:
                        sD = torch.empty(NPOS, dtype=torch.float32, device=dev)
                        _lib.call("cim_wino7_pair_scales", am[2:3].data_ptr(), 1, None, 2, sD.data_ptr(), st2)
                        D = pair.Pair(torch.empty((NPOS, rp, cout), dtype=torch.int32, device=dev), r, cout, NPOS, sD)
                        _lib.call("cim_wino7_dy_pair", dy.data_ptr(), D.buf.data_ptr(), sD.data_ptr(), r, rp, cout, 0, st2)
                    dU = pair.gemm(V, D, cin, cout, rp, True, False, limit=limit, form=form)
                    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dev)
                    _lib.call("cim_wino_wgrad_output", dU.data_ptr(), dw.data_ptr(), cout, cin, 7, st2)
                    return dw
                dwc = side_grad(0, wc_p, wgrad)
            if need_x:
                if E is None:
                    sE = torch.empty(NPOS, dtype=torch.float32, device=dev)
                    _lib.call("cim_wino7_pair_scales", am[2:3].data_ptr(), 1, None, 3, sE.data_ptr(), st)
                    E = pair.Pair(torch.empty((NPOS, rp, cout), dtype=torch.int32, device=dev), r, cout, NPOS, sE)
                    _lib.call("cim_wino7_dy_pair", dy.data_ptr(), E.buf.data_ptr(), sE.data_ptr(), r, rp, cout, 1, st)
                M2 = pair.gemm(E, Up, r, cin, cout, False, False, balance=True)
                if fold_masks is None:
                    dxp = torch.empty((r, p, p, cin), dtype=torch.float32, device=dev)
                    _lib.call("cim_wino_dx_adjoint_output", M2.data_ptr(), dxp.data_ptr(), r, p, cin, 7, st)
                else:
                    dxp = torch.empty((r, p, p, cin // 2), dtype=torch.float32, device=dev)
                    _lib.call("cim_wino7_dx_maskfold", M2.data_ptr(), fold_masks.data_ptr(), dxp.data_ptr(), r, cin // 2, st)
                dcat = dxp.permute(0, 3, 1, 2)
            dy_keep = [t for t in ((dy,) if D is None else (D.buf, D.scale)) if t is not None]   # what the late weight gradient reads
            if overlap:
                for t in dy_keep:
                    t.record_stream(side)
        keep_alive = [t for t in (V.buf, V.scale, Xp.buf, Xp.scale, Y1p.buf, Y1p.scale, dY2p.buf, dY2p.scale, dY1p.buf, dY1p.scale, am)
                      if t is not None] + dy_keep
        if late:
            # biggest first: with several ranks a gradient's all-reduce starts right behind its product (publish), and fc1's 822 MB
            # is the one that needs the rest of the backward pass to hide under
            late.sort(key=lambda e: -e[1].numel())

            def launch(defer):
                c2 = torch.cuda.current_stream(dev)
                # (single process: the stream of the late launches may be confined to a part of the chip - ops/gemm.py: LATE_CUS -
                # one workgroup per CU of that part per launch)
                late_st = G._late_stream(dev) if publisher is None else side
                late_st.wait_stream(c2)
                if late_st is not side:
                    late_st.wait_stream(side)        # (the operands' images, the weights' images: written on either)
                    for t in keep_alive:
                        t.record_stream(late_st)
                with torch.cuda.stream(late_st):
                    # (several ranks: RCCL's all-reduce kernels hold CUs of their own while these products run - a launch of exactly
                    # one workgroup per CU would then need a second, nearly empty round each time: the products go out whole)
                    limit = (G.LATE_CUS or DW_WGS) if publisher is None else 0
                    form = DW_FORM          # (several ranks as well: whole products either way, and the co-resident form leaves RCCL's
                    if form == 1:           # kernels and the backbone's chains room on every CU instead of whole CUs taken from a round)
                        limit = DW_FORM1_WGS
                    SCHEDULE["late_launches_chunked" if publisher is None else "late_launches_whole_products"] += 1
                    got = {slot: publish(w, fn(limit, form)) for slot, w, fn in late}
                    if defer:
                        for b_p, part, out in late_bias:
                            part.record_stream(late_st)
                            out.record_stream(late_st)
                            torch.sum(part, dim=0, out=out)
                if defer:           # (postponed: this node has returned - the join bookkeeping of the block below happens here)
                    # (OUTSIDE the late stream's context: the first deferral of a pass records the CURRENT stream as the one to join)
                    for b_p, part, out in late_bias:
                        G.defer_side_join(dev, b_p, out, part)
                    keep = [t for t in (V.buf, V.scale, Xp.buf, Xp.scale, Y1p.buf, Y1p.scale, dY2p.buf, dY2p.scale, dY1p.buf,
                                        dY1p.scale, am) if t is not None] + dy_keep
                    for slot, w, _ in late:
                        if got[slot] is not None:
                            G.defer_side_join(dev, w, got[slot], *keep)
                return got

            # behind the ROIAlign backward (the next node: it gets the chip to itself) when every weight is a Parameter whose
            # gradient is installed at the join anyway; else here
            # (not with several ranks: the publisher hands the gradients to nn.DataParallel's bucket bookkeeping, which must see them
            # inside the backward pass proper, not from an end-of-backward fallback)
            if G.POSTPONE_DW and publisher is None and all(isinstance(w, torch.nn.Parameter) for _, w, _ in late):
                SCHEDULE["late_launches_postponed_behind_roi_align"] += 1
                G.postpone(dev, lambda: launch(True))
            else:
                assert not late_bias, "bias partial sums were left to a postponed launch that does not happen"
                got = launch(False)
                dwc, dw1, dw2 = got.get(0, dwc), got.get(1, dw1), got.get(2, dw2)
        else:
            assert not late_bias, "bias partial sums were left to a postponed launch that does not happen"
        if overlap:
            # operands the side stream's GEMMs read: the allocator must not hand their memory out before that work is done,
            # whichever way (join here, deferred join, DataParallel's all-reduce) the weight gradients leave this node
            for t in [V.buf, V.scale, Xp.buf, Xp.scale, Y1p.buf, Y1p.scale, dY2p.buf, dY2p.scale, dY1p.buf, dY1p.scale, am] + dy_keep:
                if t is not None:
                    t.record_stream(side)
            if defer:
                # the weight gradients keep running on the side stream while the main stream goes on to the ROIAlign and backbone
                # backward: joined (and installed as .grad) once, at the end of the backward pass (ops/gemm.py: defer_side_join)
                keep = [V.buf, Xp.buf, Y1p.buf, dY2p.buf, dY1p.buf] + dy_keep
                for w, t in ((w2_p, dw2), (w1_p, dw1), (wc_p, dwc)):
                    if t is not None and isinstance(w, torch.nn.Parameter):
                        G.defer_side_join(dev, w, t, *[k for k in keep if k is not None])
                dw2 = None if isinstance(w2_p, torch.nn.Parameter) else dw2
                dw1 = None if isinstance(w1_p, torch.nn.Parameter) else dw1
                dwc = None if isinstance(wc_p, torch.nn.Parameter) else dwc
                if any(t is not None for t in (dw2, dw1, dwc)):
                    cur.wait_stream(side)
            else:
                cur.wait_stream(side)
                for t in (dw2, dw1, dwc):
                    if t is not None:
                        t.record_stream(cur)
        return dcat, dwc, dbc, dw1, db1, dw2, db2


def maskfuse_head(cat, conv, fc1, fc2, feat_amax):
    return MaskFusePairFunction.apply(cat, conv.weight, conv.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, feat_amax)


def roi_supported(x, conv_w, fc1_w, fc2_w, resolution):
    """Shapes the fused ROIAlign -> Winograd-image launch takes (else: roi_align_maskcat + maskfuse_head)."""
    c, h, w = x.shape[1], x.shape[2], x.shape[3]
    cout = conv_w.shape[0]
    return (x.is_cuda and x.dtype == torch.float32 and resolution == 7 and c % 32 == 0 and h <= 128 and w <= 128 and h * w * c < (1 << 30)
            and cout % 64 == 0 and conv_w.shape[1] == 2 * c and fc1_w.shape[1] == cout * 49 and fc1_w.shape[0] % 32 == 0
            and fc2_w.shape[1] == fc1_w.shape[0] and fc2_w.shape[0] % 32 == 0)


def maskfuse_roi_head(x, rois, masks, conv, fc1, fc2, feat_amax, spatial_scale, sampling_ratio):
    return MaskFuseRoiPairFunction.apply(x, rois, masks, conv.weight, conv.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias,
                                         feat_amax, spatial_scale, sampling_ratio)
