"""Exact-fp32 MFMA contractions of the MaskFuse head (cim_amd/csrc/gemm_f32.hip) with autograd.

`linear(x, weight, bias, relu)` == F.relu(F.linear(x, weight, bias)) and
`conv3x3(x, weight, bias, relu)` == F.relu(F.conv2d(x, weight, bias, padding=1)) for the
N x 7 x 7 ROI maps of /root/reference/lib/modeling/resnet50.py:104-110,135-136; parameters keep
the reference's layouts ([out,in] and [out,in,3,3]) so checkpoints and optimizers are unchanged.
Forward and both backward contractions run on hand-written HIP; there is no CPU fallback.
"""
import os
import weakref

import torch
from torch.autograd import Function

from .. import _lib
from ..utils import engine
from . import chain


def _ws(m, n, splits, like):
    return torch.empty(splits * m * n, dtype=torch.float32, device=like.device) if splits > 1 else None


# Arithmetic engine of the contractions: "f16x2" (default: scaled two-term fp16 split,
# 3 MFMA products), "bf16x3" (exact three-term bf16 split, 6 products), "fp32" (f32 MFMA multiplies).
# "f16x2p" (default): the f16x2 arithmetic with ONE scale per matrix on operands pre-split by their producers ("pair images",
# cim_amd/csrc/gemm_pair.hip) - used by MaskFuse's fused head Function (cim_amd/ops/maskfuse_pair.py) wherever its shapes
# qualify; everything else (and the per-layer Functions below) then runs "f16x2".
ENGINE = os.environ.get("CIM_GEMM_ENGINE", "f16x2p")
if ENGINE not in ("f16x2p", "f16x2", "bf16x3", "fp32"):
    raise _lib.CimHipError("CIM_GEMM_ENGINE must be f16x2p, f16x2, bf16x3 or fp32, got %r" % ENGINE)
PAIR = ENGINE == "f16x2p"
if PAIR:
    ENGINE = "f16x2"


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


# |max| arrays of weight matrices handed in by a producer that already streamed over the weight (the fused SGD kernel):
# keyed by the weight tensor's identity with a weak reference (an entry dies with its tensor; a recycled address or id can
# never match) and holding (version counter, data_ptr, rows, cols, row array, column array).  An entry is valid only for that exact version of that storage: any
# tracked in-place change (optimizer step, load_state_dict, init) bumps the counter and the ops make their own pass.
# Writes through `w.data` are invisible to the counter - as for autograd itself - so code that edits weights that way
# must call forget_weight_scales(w).
_WEIGHT_SCALES = {}          # id(tensor) -> (weak reference, version, data_ptr, rows, cols, row array, column array)


def register_weight_scales(w, rows, cols, row_amax, col_amax):
    key = id(w)
    ref = weakref.ref(w, lambda _r, key=key: _WEIGHT_SCALES.pop(key, None))     # the entry dies with its tensor
    _WEIGHT_SCALES[key] = (ref, w._version, w.data_ptr(), rows, cols, row_amax, col_amax)


def forget_weight_scales(w=None):
    if w is None:
        _WEIGHT_SCALES.clear()
    else:
        _WEIGHT_SCALES.pop(id(w), None)


def _registered_scales(w, rows, cols):
    e = _WEIGHT_SCALES.get(id(w))
    if e is not None and e[0]() is w and e[1] == w._version and e[2] == w.data_ptr() and e[3] == rows and e[4] == cols:
        return e[5], e[6]
    return None


def gemm(a, b, m, n, k, lda, ldb, a_mcontig=False, b_kcontig=False, bias=None, relu=False, out=None,
         a_amax=None, b_amax=None):
    """C[m,n] = A.B (+bias)(ReLU).  a/b are dense device tensors interpreted by the layout flags.
    a_amax / b_amax: operand scales from amax() when the caller already has them (f16x2 engine)."""
    if not a.is_cuda:
        raise _lib.CimHipError("cim_amd.ops.gemm: CUDA/HIP tensors required (no CPU fallback)")
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


_SIDE = {}
OVERLAP = os.environ.get("CIM_GEMM_OVERLAP", "1") == "1"


def _side_stream(dev):
    """Second HIP stream per device: the data- and weight-gradient GEMMs of a layer are independent, and launched on
    two queues the workgroups of one fill the partly empty last round of the other (fc1: 784 tiles = 3.06 rounds of
    256 CUs next to 3136 = 12.25)."""
    s = _SIDE.get(dev)
    if s is None:
        s = _SIDE[dev] = torch.cuda.Stream(device=dev)
    return s


HIGH_PRIO = os.environ.get("CIM_HIGH_PRIO", "0") == "1"      # opt-in experiment: no measured gain (ops/maskfuse_pair.py)


def main_stream_high_priority(dev):
    """A HIGH-priority HIP stream for the training step's main chain (Generalized_RCNN.forward runs on it, so does its
    backward): side-stream work - the deferred weight-gradient GEMMs - then only takes the CUs the main chain leaves, instead of
    time-sharing them half and half with the data-gradient GEMMs.  Default off (CIM_HIGH_PRIO=1 to try): everything on the
    caller's stream."""
    s = _SIDE.get(("main", dev))
    if s is None:
        s = _SIDE[("main", dev)] = torch.cuda.Stream(device=dev, priority=-1)
    return s


# ---- weight gradients that run BESIDE the rest of the backward pass (backbone convolutions) ---------------------------
# A backbone layer's weight-gradient GEMM is only needed by the optimizer.  Joined layer by layer (as the MaskFuse layers
# above do) it cannot overlap much: the main stream waits for it before the next layer's BatchNorm backward.  Deferred, the
# ~30 weight-gradient GEMMs of the body form their own chain on the side stream and the main stream's data-gradient chain
# never waits: the join happens ONCE, in an autograd-engine callback at the end of the backward pass (and defensively
# at the start of SGD.step / before DataParallel reduces a bucket).  Buffers the side work reads or writes (saved
# activations, workspaces) are kept referenced here until that join, so the caching allocator cannot hand them out.
# The deferred gradients do not travel through autograd (see defer_side_join): join_side() installs them as param.grad, or
# adds them to an existing one (iter_size > 1, DataParallel's flat views) after the join.  Consequences: a weight's
# post-accumulate-grad hooks fire at the layer's backward although nothing was accumulated yet (gradient_is_deferred() tells) -
# DataParallel, which counts ready gradients per bucket through such hooks, skips them and launches the bucket that holds
# them at the end of the backward pass, after calling join_side() itself - and torch.autograd.grad() with such a weight
# among its inputs does not see the gradient (use .backward(), or CIM_DEFER_DW=0).
# Measured at cfg2: 17.33 -> 16.62 ms per step.  Deferring the MaskFuse layers' weight gradients as well (fc1, fc2, the
# Winograd convolution: they already run beside their layer's data gradient) changed nothing: 16.71 ms.
DEFER_DW = OVERLAP and os.environ.get("CIM_DEFER_DW", "1") == "1" and engine.HAS_ENGINE_CALLBACK
_DEFERRED = {}        # device -> [main stream, [(param, dw)], [tensors kept alive], [(finisher, record)]]
_PENDING_IDS = set()  # id(param) of the weights whose gradient is still on the side stream


# nn.DataParallel (several ranks) registers a publisher for ITS parameters: publish(param, grad, stream) installs `grad` as
# param.grad (or adds it) on `stream` and lets the parameter's all-reduce bucket go out AT ONCE - used by fused Functions whose
# backward computes several big weight gradients (MaskFuse: fc2, fc1, conv) long before it returns: returned through autograd
# they would all become visible at the end of the node, and the 822 MB fc1 all-reduce would lose ~5 ms of backward to hide
# under.  Keyed by parameter (weak references both ways), so two models in one process - one wrapped for several ranks, one
# not - each run their own schedule.
_PUBLISHERS = {}      # id(parameter) -> (weak reference to the parameter, weak reference to the wrapper's bound method)


def register_publisher(params, method):
    wm = weakref.WeakMethod(method)
    for q in params:
        key = id(q)
        _PUBLISHERS[key] = (weakref.ref(q, lambda _r, key=key: _PUBLISHERS.pop(key, None)), wm)


def publisher_for(param):
    """The publish callable of the wrapper that owns `param`, or None (single process / unwrapped model)."""
    e = _PUBLISHERS.get(id(param)) if param is not None else None
    if e is None or e[0]() is not param:
        return None
    return e[1]()


def gradient_is_deferred(param):
    """True between a layer's backward and join_side() for a weight whose gradient runs on the side stream (its
    post-accumulate-grad hooks fire at the layer's backward although nothing was accumulated yet)."""
    return id(param) in _PENDING_IDS


_FORK_EVENTS = {}     # device -> [next index, [(torch event, raw hipEvent_t), ...]]


def fork_event(dev):
    """Raw hipEvent_t (an int) for the fork inside a cim_*_bn_act_bwd call: the C library creates nothing, the events are the
    host's.  A small round-robin pool per device - an event is recorded and waited on inside the call that gets it, so it can go
    to the next call at once.  (torch creates the underlying hipEvent_t at the first record().)"""
    ent = _FORK_EVENTS.get(dev)
    if ent is None:
        evs = []
        cur = torch.cuda.current_stream(dev)
        for _ in range(16):
            e = torch.cuda.Event(enable_timing=False)
            e.record(cur)
            evs.append((e, e.cuda_event))
        ent = _FORK_EVENTS[dev] = [0, evs]
    ent[0] = (ent[0] + 1) % len(ent[1])
    return ent[1][ent[0]][1]


def side_stream_for_backward(dev, param, node=None):
    """(side stream pointer or None, fork event, join event, join flag) for a backbone layer's backward whose weight is `param`:
    deferred to the side stream (the caller joins later: no join event), or - HIP-graph capture, CIM_DEFER_DW=0, a weight that is
    not a Parameter, a pass that is not a complete .backward() (torch.autograd.grad captures the gradient the node RETURNS: a
    deferred one, installed as .grad at the end of the pass, would be lost; `node` = the layer's ctx, ops/chain.py:
    restricted_pass) - everything on the caller's stream (a fork / join inside every layer measured SLOWER than that: 17.0-17.1 vs
    16.8 ms per step)."""
    if not DEFER_DW or param is None or torch.cuda.is_current_stream_capturing():
        return None, None, None, 1
    if node is not None and chain.restricted_pass(node):
        return None, None, None, 1
    return _body_stream(dev).cuda_stream, fork_event(dev), None, 0


def _body_stream(dev):
    """Third HIP stream per device: the body's deferred weight-gradient GEMMs.  On the side stream they queued behind MaskFuse's
    late weight-gradient products (~3 ms of 256-workgroup launches, ops/maskfuse_pair.py) and formed the tail of the backward
    pass; on their own queue they fill the gaps between those launches: 4.49 -> 4.30 ms for the body's backward phase, 14.11 ->
    13.81 ms per step at cfg2 (same box, round 4; round 3 measured no gain from this, before the late launches existed)."""
    s = _SIDE.get(("body", dev))
    if s is None:
        s = _SIDE[("body", dev)] = torch.cuda.Stream(device=dev)
    return s


def defer_side_join(dev, param, dw, *keep):
    """The layer's weight gradient `dw` was enqueued on the side stream without a join.  It does NOT travel through
    autograd (the layer's backward returns None for the weight: AccumulateGrad may copy a gradient it cannot steal, on the
    main stream, before the side stream has written it): it is installed as param.grad by join_side(), after the join."""
    ent = _deferred_entry(dev)
    ent[1].append((param, dw))
    ent[2].extend(keep)
    _PENDING_IDS.add(id(param))


def _deferred_entry(dev):
    ent = _DEFERRED.get(dev)
    if ent is None or not (ent[1] or ent[2] or ent[3]):
        ent = _DEFERRED[dev] = [torch.cuda.current_stream(dev), [], [], []]
        engine.queue_callback(join_side)
    return ent


def defer_finisher(dev, finisher, record, params):
    """Gradients that ONE launch at the end of the backward pass finishes for many layers (the affine gradients of chained
    BatchNorm layers, ops/conv1x1.py): join_side() calls finisher(list of records) on the stream the backward ran on - every
    kernel that feeds it is enqueued by then - and installs the (param, grad) pairs it returns like the deferred weight gradients.
    `params`: the parameters whose gradient is pending until then."""
    ent = _deferred_entry(dev)
    ent[3].append((finisher, record))
    for q in params:
        _PENDING_IDS.add(id(q))


# Launches a node's backward postponed to "after the next node's kernels are enqueued": MaskFuse's late weight-gradient products
# (ops/maskfuse_pair.py) wait for the ROIAlign backward - the node that consumes MaskFuse's input gradient - so that it gets the chip
# to itself; ops/roi_align.py calls run_postponed() behind its launch, an end-of-backward callback does when no such node ran.
_POSTPONED = {}       # device -> [closures]
POSTPONE_DW = os.environ.get("CIM_MASKFUSE_DW_AFTER_ROI", "1") == "1"


def postpone(dev, fn):
    lst = _POSTPONED.setdefault(dev, [])
    if not lst:
        engine.queue_callback(run_postponed)
    lst.append(fn)


def run_postponed(dev=None):
    for d in ([dev] if dev is not None else list(_POSTPONED)):
        lst = _POSTPONED.get(d)
        while lst:
            lst.pop(0)()


def join_side(discard=False):
    """Make the stream the backward ran on wait for the deferred side-stream work, install the weight gradients, release
    the kept buffers.  Runs as an autograd-engine callback at the end of the backward pass.
    discard=True (the safety join at the start of a training forward): whatever is still pending belongs to a backward
    pass that was ABORTED (an exception before the engine's callbacks ran) - the streams are joined, but its gradients
    are dropped instead of being installed into the new step's `.grad`."""
    if discard:
        _POSTPONED.clear()
    else:
        run_postponed()             # (launches still waiting for "the next node": whoever joins needs them enqueued first)
    for dev, ent in list(_DEFERRED.items()):
        if ent[3]:
            if not discard:
                with torch.cuda.stream(ent[0]), torch.no_grad():
                    todo = {}
                    for fn, rec in ent[3]:
                        todo.setdefault(fn, []).append(rec)
                    for fn, recs in todo.items():
                        ent[1].extend(fn(recs))
            ent[3] = []
        if ent[1] or ent[2]:
            streams = [_side_stream(dev)] + ([_SIDE[("body", dev)]] if ("body", dev) in _SIDE else [])
            for st in streams:
                ent[0].wait_stream(st)
            cur = torch.cuda.current_stream(dev)
            if cur != ent[0]:
                for st in streams:
                    cur.wait_stream(st)
            with torch.no_grad():
                for param, dw in ent[1]:
                    if not discard:
                        if param.grad is None:
                            param.grad = dw
                        else:
                            param.grad += dw
                    _PENDING_IDS.discard(id(param))
            ent[1], ent[2] = [], []
    if discard:
        _PENDING_IDS.clear()


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
CONV_ALGO = os.environ.get("CIM_CONV_ALGO", "winograd7")
PREFETCH_U = os.environ.get("CIM_PREFETCH_U", "1") == "1"     # filter transform on the side stream, under the backbone forward


def _wino_geometry(algo, p, r):
    """(tile code of the C entry points, positions, GEMM rows) of a Winograd algorithm on r maps of p x p."""
    if algo == "winograd7":
        return 7, 121, r                       # one tile of each of the 4 types per map
    tile = 4 if algo == "winograd4" else 2
    t = (p + tile - 1) // tile
    return tile, (tile + 2) ** 2, r * t * t


_U_PREFETCH = {}     # id(weight) -> (tile, U, event): filter transforms launched ahead on the side stream


def prefetch_filter_transform(w, p):
    """Launch the Winograd filter transform of the conv weight `w` (maps of p x p) on the SIDE stream now: it only depends on the
    weight, so it can run under the backbone forward (small latency-bound kernels that leave most of the chip idle) instead
    of between the ROIAlign and the convolution's GEMM.  Conv3x3Function.forward picks it up (and waits for its event)."""
    if not (OVERLAP and PREFETCH_U and w.is_cuda and w.dim() == 4) or torch.cuda.is_current_stream_capturing():
        return
    cout, cin = w.shape[0], w.shape[1]
    algo = CONV_ALGO if (cin % 4 == 0 and cout % 4 == 0) else "direct"
    if algo == "winograd7" and p != 7:
        algo = "winograd4"
    if not algo.startswith("winograd") or not w.is_contiguous():
        return
    tile, npos, _ = _wino_geometry(algo, p, 1)
    dev = w.device
    cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
    side.wait_stream(cur)                    # (the optimizer's update of w was enqueued on `cur`)
    with torch.cuda.stream(side), torch.no_grad():
        U = torch.empty((npos, cin, cout), dtype=torch.float32, device=dev)
        _lib.call("cim_wino_filter_transform", w.data_ptr(), U.data_ptr(), cout, cin, 0, tile, _lib.stream_ptr())
        ev = torch.cuda.Event()
        ev.record(side)
    _U_PREFETCH[id(w)] = (tile, U, ev, weakref.ref(w), w._version)      # (validated at use: an id can be reused)


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
            pre = _U_PREFETCH.pop(id(w), None)
            if pre is not None and pre[0] == tile and pre[1].shape == U.shape and pre[3]() is w and pre[4] == w._version:
                U = pre[1]                                             # transformed ahead, on the side stream
                torch.cuda.current_stream(dev).wait_event(pre[2])
                U.record_stream(torch.cuda.current_stream(dev))
            else:
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
