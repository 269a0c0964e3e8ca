"""Host-side infrastructure of the step's backward schedule: the second / third HIP streams, weight gradients that run BESIDE the
rest of the backward pass and are joined once at its end, launches postponed behind the next autograd node, the publisher registry of
nn.DataParallel, caller-owned fork events for the C library, and the registry of weight |max| arrays the fused SGD kernel leaves
behind (the scale source of the weights' pair images, ops/maskfuse_pair.py).

The contractions themselves run on ONE engine - the pair engine, cim_amd/ops/pair.py + ops/maskfuse_pair.py - and ONE convolution
algorithm (the mixed 4 + 3 Winograd tiling).  The per-layer `linear` / `conv3x3` Functions on the bf16x3 / f16x2 / fp32 engines and the
direct / F(2x2,3x3) / F(4x4,3x3) algorithms that lived here in rounds 1-4 are test infrastructure now: experiments/engines.py.
"""
import weakref

import torch

from .. import _lib
from ..utils import engine
from . import chain


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



_SIDE = {}
OVERLAP = True          # two-stream schedule of the backward pass (tests / tools may switch it off: a module attribute, not an environment switch)


def _side_stream(dev):
    """Second HIP stream per device: the data- and weight-gradient GEMMs of a layer are independent, and launched on
    two queues the workgroups of one fill the partly empty last round of the other (fc1: 784 tiles = 3.06 rounds of
    256 CUs next to 3136 = 12.25)."""
    s = _SIDE.get(dev)
    if s is None:
        s = _SIDE[dev] = torch.cuda.Stream(device=dev)
    return s



# CU partition of the last backward phase (round 6).  MaskFuse's late weight-gradient products (ops/maskfuse_pair.py: launches of one
# 256 x 256 tile per CU, ~110 us each, 128 KB of LDS: the workgroup owns its CU) and the body's data- / weight-gradient chains
# (~100 latency-bound launches that fill a fraction of the chip each) used to take the SUM of their times: a body launch got on the
# chip between two launches of tiles and held all of it at a quarter of its MFMA rate.  With LATE_CUS > 0 the late products run on a
# stream created with hipExtStreamCreateWithCUMask over the LAST `LATE_CUS` logical CUs (the driver deals mask bits round-robin over
# the 8 XCDs: every XCD gives the same share), in launches of LATE_CUS workgroups; the body's launches - which cannot share a CU with
# a tile anyway (LDS) - keep the other CUs to themselves the whole time.  The stream is the HOST's (the C library creates nothing);
# 0: the late products run on the side stream over the whole chip (round 5's schedule).
LATE_CUS = 0
TOTAL_CUS = 256


def _late_stream(dev):
    """The stream of MaskFuse's late weight-gradient launches: the side stream, or - LATE_CUS > 0 - a stream confined to the
    last LATE_CUS CUs."""
    if not LATE_CUS:
        return _side_stream(dev)
    key = ("late", dev, LATE_CUS)
    s = _SIDE.get(key)
    if s is None:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        words = (TOTAL_CUS + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        for i in range(TOTAL_CUS - LATE_CUS, TOTAL_CUS):
            mask[i // 32] |= 1 << (i % 32)
        raw = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(raw), ctypes.c_uint32(words), mask)
        if rc != 0 or not raw.value:
            raise RuntimeError("cim_amd: hipExtStreamCreateWithCUMask failed (%d)" % rc)
        s = _SIDE[key] = torch.cuda.ExternalStream(raw.value, device=dev)
    return s


def _extra_streams(dev):
    """Every stream besides the side stream that deferred work may still run on (joined at the end of the backward pass)."""
    return [st for key, st in _SIDE.items() if isinstance(key, tuple) and key[0] in ("body", "late") and key[1] == dev]


# ---- parameter updates still running on the side stream (cim_amd.optim.SGD with overlap_update: the big MaskFuse weights are
# updated UNDER the next step's backbone forward).  Everything of this package that reads such a weight either runs on the side
# stream behind the update (the weights' pair images, ops/maskfuse_pair.py: prefetch_weight_images) or calls wait_pending_updates()
# first (an image built on the caller's stream, state_dict() of the model and of the optimizer).
_PENDING_UPDATES = {}   # device -> (event recorded behind the update on the side stream, frozenset of parameter ids)


def register_pending_update(dev, event, ids):
    _PENDING_UPDATES[dev] = (event, frozenset(ids))


def wait_pending_updates(dev=None):
    """Make the CURRENT stream wait for parameter updates that are still running on the side stream (no-op when none are)."""
    for d in ([dev] if dev is not None else list(_PENDING_UPDATES)):
        ent = _PENDING_UPDATES.pop(d, None)
        if ent is not None:
            torch.cuda.current_stream(d).wait_event(ent[0])


HIGH_PRIO = False       # opt-in experiment: no measured gain (ops/maskfuse_pair.py)


def main_stream_high_priority(dev):
    """A HIGH-priority HIP stream for the training step's main chain (Generalized_RCNN.forward runs on it, so does its
    backward): side-stream work - the deferred weight-gradient GEMMs - then only takes the CUs the main chain leaves, instead of
    time-sharing them half and half with the data-gradient GEMMs.  Default off (`gemm.HIGH_PRIO = True`, a module attribute, to
    try): everything on the caller's stream."""
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
# among its inputs is served through autograd instead (ops/chain.py: restricted_pass - the layer's backward sees that the pass does not
# accumulate into its weight and keeps everything on the caller's stream); `gemm.DEFER_DW = False` (a module attribute, no
# environment switch) turns the deferral off altogether.
# Measured at cfg2: 17.33 -> 16.62 ms per step.  Deferring the MaskFuse layers' weight gradients as well (fc1, fc2, the
# Winograd convolution: they already run beside their layer's data gradient) changed nothing: 16.71 ms.
DEFER_DW = OVERLAP and engine.HAS_ENGINE_CALLBACK
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
    deferred to the side stream (the caller joins later: no join event), or - HIP-graph capture, gemm.DEFER_DW = False, a weight that is
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
        cur = torch.cuda.current_stream(dev)
        # the first deferral of a pass names the stream that joins (and runs the finishers): the stream the backward runs on - a
        # caller inside `with torch.cuda.stream(side)` would make the SIDE stream wait for itself and the pass's own stream for nothing
        if any(cur == st for st in [_SIDE.get(dev)] + _extra_streams(dev) if st is not None):
            raise RuntimeError("cim_amd: defer_side_join / defer_finisher called under a side stream's context")
        ent = _DEFERRED[dev] = [cur, [], [], []]
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
POSTPONE_DW = True


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
            streams = [_side_stream(dev)] + _extra_streams(dev)
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

