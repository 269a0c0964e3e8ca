"""Fused multi-tensor SGD (cim_amd/csrc/sgd.hip) behind torch.optim.SGD's interface.

Drop-in for `torch.optim.SGD(params, momentum=cfg.SOLVER.MOMENTUM)` at /root/reference/tools/train.py:308-309: same
param-group keys (lr, momentum, weight_decay, dampening, nesterov), same `state[p]['momentum_buffer']` tensors (so
lib/utils/net.py:47-83 `update_learning_rate` / `_CorrectMomentum` and the checkpoint code keep working), same update
rule - but ONE kernel launch per step for all parameters instead of one multi-tensor launch per ~20 tensors.
Only what the reference uses is supported on the HIP path: fp32 CUDA/HIP parameters, dampening 0, no Nesterov,
dense gradients, one momentum value; anything else raises (there is no silent fallback).

Host cost per step: autograd hands out new gradient tensors every step and the row / column |max| arrays are fresh
storage every step, so the 64-byte record of every tensor is rebuilt and copied (one pass over the parameters, one
10 KB H2D copy; the copy is skipped only when the table happens to be byte-identical to the previous step's); the chunk
table only depends on the tensor sizes and is built once.  Matrix mode and 16-byte accesses need 16-byte aligned
parameter, gradient and history pointers: nn.DataParallel's flat gradient buffer aligns its views accordingly.

Weights of >= 2^20 elements are updated in the kernel's matrix mode, which also emits max |w_new| per row and per column:
exactly what the f16x2 contraction engine needs as operand scales of `nn.Linear` / conv weights.  They are registered with
`cim_amd.ops.gemm.register_weight_scales` under the weight's new version counter, so the next forward skips its pass over
the weight (0.29 ms per step at cfg2, mostly the 822 MB fc1 weight); any other in-place change of the weight bumps the
version and the ops fall back to their own pass.
"""
import numpy as np
import torch

from ..ops.gemm import join_side as _join_side

from .. import _lib

CHUNK = 16384          # elements per workgroup
_TENSOR = np.dtype([("p", "<u8"), ("g", "<u8"), ("buf", "<u8"), ("n", "<i8"), ("lr", "<f4"), ("wd", "<f4"),
                    ("rows", "<i4"), ("cols", "<i4"), ("row_amax", "<u8"), ("col_amax", "<u8")])
_CHUNK = np.dtype([("tensor", "<i4"), ("n", "<i4"), ("offset", "<i8")])
MATRIX_MIN = 1 << 20   # weights of at least this many elements are updated in matrix mode (row / column |max| by-product)
TRAIL_MIN = 1 << 24    # overlap_update: weights of at least this many elements are updated on the side stream (at cfg2: fc1 205 M,
                       # the MaskFuse convolution 18.9 M, fc2 16.8 M elements = 96 % of the update's 5.1 GB of traffic)
TILE_ROWS, TILE_COLS = 64, 1024


def _matrix_shape(p):
    """(rows, cols) when the parameter qualifies for the kernel's matrix mode, else None."""
    if p.dim() < 2 or p.numel() < MATRIX_MIN:
        return None
    rows = p.shape[0]
    cols = p.numel() // rows
    return (rows, cols) if cols % 4 == 0 else None


class _Pass:
    """Device tables and cached per-parameter records of ONE fused launch over a fixed subset of the parameters."""

    def __init__(self):
        self.layout = None          # tuple of (numel, rows, cols) the chunk table on the device was built for
        self.chunks = None          # device chunk table
        self.n_chunks = 0
        self.cache = None           # records that do not change from step to step (see SGD._scan)


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if dampening != 0.0 or nesterov:
            raise NotImplementedError("cim_amd.optim.SGD: dampening / Nesterov are not used by the reference and not provided")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        # Opt-in (round 6; `optimizer.overlap_update = True`, bench.py sets it): the update of the BIG weights (>= TRAIL_MIN elements)
        # leaves the caller's stream - step() enqueues it on the package's side stream, ordered behind everything the caller's stream
        # has done, and returns without making the caller's stream wait.  The next forward's backbone (~1.9 ms of small latency-bound
        # launches that leave HBM idle) then runs BESIDE the 0.8 ms HBM-bound update instead of behind it; the weights' pair images
        # (all the forward and backward ever read of these weights) are built on the same side stream behind the update, and the
        # caller's stream waits for them where MaskFuse starts - as before.  What the caller must know: between step() and the next
        # forward's box head these weights (and their momentum buffers) are NOT ordered on the caller's stream; state_dict() of the
        # model and of this optimizer wait by themselves, any other direct read needs `optimizer.wait_update()` first.
        self.overlap_update = False
        self.trail_workgroups = 256      # workgroups of the side-stream launch (one slot per CU; 0 / 256 / 512 / 1024 / 2048: 13.90 / 13.70 / 13.79 / 13.84 / 13.85 ms per step)
        self._passes = {}           # "all" | "early" | "rest" | "trail" -> _Pass
        self._early = None          # (frozenset of parameter ids updated early in this optimizer step, stream, event)
        self._check_every_step = True    # re-count the parameters with gradients every step (a parameter that starts to
                                         # receive gradients must not be skipped silently; ~20 us)

    def _build_chunks(self, ps, layout, dev):
        """layout: per tensor (numel, rows, cols) with rows = cols = 0 for flat tensors."""
        parts = []
        for ti, (n, rows, cols) in enumerate(layout):
            if cols > 0:        # matrix mode: 64 x 1024 tiles, offset = first row, n = first column
                r0, c0 = np.meshgrid(np.arange(0, rows, TILE_ROWS), np.arange(0, cols, TILE_COLS), indexing="ij")
                tab = np.empty(r0.size, dtype=_CHUNK)
                tab["offset"], tab["n"] = r0.reshape(-1), c0.reshape(-1)
            else:
                cnt = (n + CHUNK - 1) // CHUNK
                tab = np.empty(cnt, dtype=_CHUNK)
                tab["offset"], tab["n"] = np.arange(cnt, dtype=np.int64) * CHUNK, CHUNK
            tab["tensor"] = ti
            parts.append(tab)
        tab = np.concatenate(parts)
        ps.chunks = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(dev)
        ps.n_chunks, ps.layout = int(tab.shape[0]), tuple(layout)

    def _scan(self, ps, select):
        """Slow path (first step, or when the set of parameters with gradients changed): validate every selected tensor and
        cache what does not change from step to step - parameter and history pointers, sizes, matrix shapes, the device
        chunk table.  Returns False when no selected parameter has a gradient."""
        recs, momentum, dev = [], None, None
        for gi, group in enumerate(self.param_groups):
            if group.get("dampening", 0.0) != 0.0 or group.get("nesterov", False):
                raise NotImplementedError("cim_amd.optim.SGD: dampening / Nesterov")
            if momentum is None:
                momentum = float(group["momentum"])
            elif float(group["momentum"]) != momentum:
                raise NotImplementedError("cim_amd.optim.SGD: one momentum value for all groups")
            for p in group["params"]:
                g = p.grad
                if g is None or not select(p):
                    continue
                if not p.is_cuda:
                    raise _lib.CimHipError("cim_amd.optim.SGD: CUDA/HIP parameters required (no CPU fallback)")
                if p.dtype != torch.float32 or g.dtype != torch.float32 or g.is_sparse or not p.is_contiguous():
                    raise NotImplementedError("cim_amd.optim.SGD: dense contiguous fp32 parameters and gradients")
                st = self.state[p]
                buf = st.get("momentum_buffer")
                if buf is None:
                    buf = st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                dev = p.device
                recs.append((p, buf, gi))
        if not recs:
            ps.cache = None
            return False
        n = len(recs)
        tab = np.zeros(n, dtype=_TENSOR)
        tab["p"] = [p.data_ptr() for p, _, _ in recs]
        tab["buf"] = [b.data_ptr() for _, b, _ in recs]
        tab["n"] = [p.numel() for p, _, _ in recs]
        amax_off, off = [], 0
        for i, (p, buf, _) in enumerate(recs):
            ms = _matrix_shape(p) if ((p.data_ptr() | buf.data_ptr() | p.grad.data_ptr()) & 15) == 0 else None
            if ms:
                tab["rows"][i], tab["cols"][i] = ms
                amax_off.append((i, off, ms[0], ms[1]))
                off += ms[0] + ms[1]
        layout = tuple((int(tab["n"][i]), int(tab["rows"][i]), int(tab["cols"][i])) for i in range(n))
        if layout != ps.layout or ps.chunks is None or ps.chunks.device != dev:
            self._build_chunks(ps, layout, dev)
        touched = []
        for p, buf, _ in recs:
            touched += [p, buf]
        ps.cache = dict(recs=recs, tab=tab, amax_off=amax_off, n_amax=off, momentum=momentum, dev=dev, touched=touched,
                        pinned=torch.empty(tab.nbytes, dtype=torch.uint8).pin_memory(),
                        table=torch.empty(tab.nbytes, dtype=torch.uint8, device=dev), copied=None,
                        group_of=np.array([gi for _, _, gi in recs]), sig=None, ids=frozenset(id(p) for p, _, _ in recs))
        return True

    def _run(self, key, select, _retry=True):
        """One fused launch over the parameters `select` accepts (on the current stream)."""
        ps = self._passes.setdefault(key, _Pass())
        c = ps.cache
        # fast path: the same parameters have gradients as last step (the usual case) - only the gradient pointers, the
        # learning rates and the |max| arrays are refreshed; anything else re-scans
        if c is not None:
            grads = [p.grad for p, _, _ in c["recs"]]
            stale = any(g is None for g in grads) or any(float(g["momentum"]) != c["momentum"] for g in self.param_groups)
            if not stale:
                # the cached raw pointers must still be THE tensors: load_state_dict() replaces the momentum buffers,
                # .to() / .half() / set_() the parameter storage (one int compare per tensor)
                tabp, tabb, state = c["tab"]["p"], c["tab"]["buf"], self.state
                for i, (p, buf, _) in enumerate(c["recs"]):
                    if p.data_ptr() != tabp[i] or state[p].get("momentum_buffer") is not buf or buf.data_ptr() != tabb[i]:
                        stale = True
                        break
            if not stale and self._check_every_step:
                n_sel = sum(1 for g in self.param_groups for p in g["params"] if p.grad is not None and select(p))
                stale = n_sel != len(grads)
            if stale:
                c = None
        if c is None:
            if not self._scan(ps, select):
                return
            c = ps.cache
            grads = [p.grad for p, _, _ in c["recs"]]
        tab, dev = c["tab"], c["dev"]
        keep, gp = [], []
        for g in grads:
            if g.dtype != torch.float32 or not g.is_contiguous():
                if g.dtype != torch.float32 or g.is_sparse:
                    raise NotImplementedError("cim_amd.optim.SGD: dense fp32 gradients")
                g = g.contiguous()
                keep.append(g)
            gp.append(g.data_ptr())
        tab["g"] = gp
        lrs = np.array([float(g["lr"]) for g in self.param_groups], dtype=np.float32)
        wds = np.array([float(g["weight_decay"]) for g in self.param_groups], dtype=np.float32)
        tab["lr"], tab["wd"] = lrs[c["group_of"]], wds[c["group_of"]]
        # row / column |max| arrays of the matrix-mode tensors: fresh (zeroed) storage every step - consumers of the
        # previous step's arrays (autograd graphs kept alive) never see them change
        amax_buf = torch.zeros(max(c["n_amax"], 1), dtype=torch.int32, device=dev)
        base = amax_buf.data_ptr()
        slices = []
        for i, off, rows, cols in c["amax_off"]:
            if gp[i] & 15:          # a gradient view at an odd offset this step (rare): re-scan, the tensor takes flat mode
                ps.cache = None
                if not _retry:
                    raise _lib.CimHipError("cim_amd.optim.SGD: inconsistent gradient alignment")
                return self._run(key, select, _retry=False)
            tab["row_amax"][i], tab["col_amax"][i] = base + 4 * off, base + 4 * (off + rows)
            slices.append((c["recs"][i][0], amax_buf[off:off + rows], amax_buf[off + rows:off + rows + cols], rows, cols))
        raw = tab.view(np.uint8).reshape(-1)
        sig = raw.tobytes()
        if sig != c["sig"]:
            if c["copied"] is not None:
                c["copied"].synchronize()               # the previous H2D copy out of the staging buffer (long done in practice)
            c["pinned"].numpy()[:] = raw
            c["table"].copy_(c["pinned"], non_blocking=True)
            c["copied"] = torch.cuda.Event()
            c["copied"].record()
            c["sig"] = sig
        _lib.call("cim_sgd_multi", c["table"].data_ptr(), ps.chunks.data_ptr(), ps.n_chunks, c["momentum"],
                  self.trail_workgroups if key == "trail" else 0, _lib.stream_ptr())
        # the kernel wrote parameters and momentum buffers through raw pointers: tell autograd's version counters, so that
        # anything keyed by Tensor._version (saved-tensor checks, caches) sees the in-place update
        torch.autograd.graph.increment_version(c["touched"])
        if slices:      # hand the by-product scales to the contraction ops (valid for exactly this version of the weight)
            from ..ops import gemm
            for p, ra, ca, rows, cols in slices:
                gemm.register_weight_scales(p, rows, cols, ra, ca)
        return c["ids"]

    def _invalidate(self):
        for ps in self._passes.values():
            ps.cache = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._invalidate()              # new momentum-buffer tensors: the cached pointers are dead

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault("_passes", {})
        self.__dict__.setdefault("_early", None)
        self.__dict__.setdefault("_check_every_step", True)
        self._invalidate()

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_passes"):
            self._invalidate()

    @torch.no_grad()
    def step_early(self, params, stream):
        """Update `params` NOW, on `stream`, ahead of `step()` - called from inside the backward pass once their gradients
        are final (nn.DataParallel.attach_optimizer): the 1 GB of MaskFuse weights is updated while the backward of the
        backbone - small latency-bound launches that leave HBM idle - is still running.  The following `step()` updates
        only the remaining parameters and makes the caller's stream wait for this one."""
        ids = frozenset(id(p) for p in params)
        cur = torch.cuda.current_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            done = self._run("early", lambda p: id(p) in ids)
        ev = torch.cuda.Event()
        ev.record(stream)
        for p in params:                                 # the side stream reads / writes these; keep the allocator informed
            if p.grad is not None:
                p.grad.record_stream(stream)
        self._early = (done or frozenset(), stream, ev)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _join_side()            # weight gradients deferred to the side stream (cim_amd/ops/gemm.py; normally joined at the end of backward)
        if self._early is not None:
            done, stream, ev = self._early
            self._early = None
            self._run("rest", lambda p: id(p) not in done)
            torch.cuda.current_stream().wait_event(ev)  # everything after the step sees the early update too
        elif self.overlap_update and not torch.cuda.is_current_stream_capturing():
            self._step_overlapped()
        else:
            self._run("all", lambda p: True)
        return loss

    def _step_overlapped(self):
        from ..ops import gemm
        big = [p for g in self.param_groups for p in g["params"]
               if p.grad is not None and p.is_cuda and p.numel() >= TRAIL_MIN and _matrix_shape(p) is not None]
        ids = frozenset(id(p) for p in big)
        self._run("rest", lambda p: id(p) not in ids)
        if not big:
            return
        dev = big[0].device
        cur, side = torch.cuda.current_stream(dev), gemm._side_stream(dev)
        gemm.wait_pending_updates(dev)                   # (a previous trailing update nobody waited for: same stream order anyway)
        side.wait_stream(cur)                            # gradients final, the small parameters' launch enqueued
        with torch.cuda.stream(side):
            self._run("trail", lambda p: id(p) in ids)
            ev = torch.cuda.Event()
            ev.record(side)
        for p in big:                                    # zero_grad() drops these while the side stream may still read them
            p.grad.record_stream(side)
        gemm.register_pending_update(dev, ev, ids)

    def zero_grad(self, set_to_none=True):
        if not set_to_none:
            self.wait_update()          # (zeroing in place: the side stream may still read the big weights' gradients)
        return super().zero_grad(set_to_none=set_to_none)

    def wait_update(self):
        """Make the current stream wait for an update that is still running on the side stream (overlap_update)."""
        from ..ops import gemm
        gemm.wait_pending_updates()

    def state_dict(self):
        self.wait_update()              # (momentum buffers of the big weights may still be written on the side stream)
        return super().state_dict()
