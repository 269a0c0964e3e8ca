"""Data-parallel wrapper for the CIM training step: one process per MI355X, gradient
all-reduce on RCCL over xGMI.

Keeps the constructor / call surface of /root/reference/lib/nn/parallel/data_parallel.py:9-116
(`DataParallel(module, cpu_keywords=[...], minibatch=True)`, list-valued kwargs with one entry
per local device, `.module`), which tools/train.py:344,432 uses.  The reference runs one Python
thread per GPU inside one process, re-broadcasts all parameters every forward and gathers
outputs (broken for >1 GPU, S7).  Here each rank owns ONE device and one image;
the only collective is the gradient all-reduce:

  * the gradients of the many small parameters live in ONE flat fp32 buffer (param.grad are views into it), so
    zeroing is one memset and their all-reduce needs no packing copies; a BIG parameter (>= `big_bytes`: the two fc
    weights and the 3x3 conv of MaskFuse, 96 % of the 1 GB) is its own bucket and is reduced IN PLACE in the tensor
    autograd hands over (its .grad is reset to None each step, so autograd steals the incoming gradient instead of
    adding it into a zeroed view: no 1 GB memset and no 3 GB accumulate pass per step);
  * buckets are ordered in reverse parameter order (= backward completion order);
    a bucket's all-reduce is launched from a post-accumulate-grad hook as soon as its last
    gradient is written, so RCCL traffic overlaps the rest of backward;
  * the all-reduce AVERAGES over ranks (RCCL's AVG; SUM + one scale on backends without it) == the reference's
    `loss.mean(dim=0)` over GPUs (lib/utils/training_stats.py:100) - the driver does not rescale its loss;
  * the driver loop needs NO extra call: whatever was not reduced by the hooks is reduced, and all collectives are
    waited for, in an autograd-engine callback at the end of `loss.backward()` (tools/train.py:436-438 unchanged);
    with gradient accumulation (`--iter_size k`, train.py:420) set `iter_size=k` (or CIM_ITER_SIZE=k) and only the
    k-th backward after `zero_grad()` communicates (leaving it at 1 is still correct - averaging is linear - but
    reduces k times);
  * gradients that do NOT come through autograd - the backbone's convolution weights (side stream, ops/gemm.py) and the gamma /
    beta of chained BatchNorm layers (bn1 / bn2 of a bottleneck, finished by ONE launch at the end of the pass, ops/chain.py) -
    fire no hook: the buckets that hold them go out at the forced flush at the end of backward.  Every backbone bucket holds
    deferred convolution weights, and the backbone's buckets are the LAST ones in the strict issue order, so the affine
    parameters hold nothing back that was not waiting already (tests/test_gpu_dp.py checks each of them is the ranks' mean);
  * construction broadcasts rank 0's parameters and buffers (a per-rank checkpoint load or a nondeterministic
    initialisation cannot make the replicas diverge silently).
"""
import os
import contextlib

import torch
import torch.distributed as dist

from ...ops.gemm import join_side as _join_side, gradient_is_deferred as _gradient_is_deferred
from ...ops import gemm as _gemm_ops
from ...utils import engine
from torch import nn


class DataParallel(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None, dim=0, cpu_keywords=(), minibatch=False,
                 batch_outputs=True, bucket_bytes=64 << 20, process_group=None, force_flat_grads=False,
                 big_bytes=16 << 20, iter_size=None):
        super().__init__()
        self.module = module
        self.cpu_keywords = list(cpu_keywords)
        self.minibatch = minibatch
        self.batch_outputs = batch_outputs
        self.dim = dim
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.device = next(module.parameters()).device
        self.device_ids = [self.device.index] if self.device.type == "cuda" else []
        self.output_device = self.device
        self._sync = True
        self._avg_op = None
        self._pending = []
        self._next_bucket = 0
        self._force_flat = force_flat_grads
        self._big_bytes = big_bytes
        # (the reference's driver builds the wrapper without this argument, tools/train.py:344: CIM_ITER_SIZE carries its --iter_size)
        self.iter_size = int(os.environ.get("CIM_ITER_SIZE", "1")) if iter_size is None else int(iter_size)
        self._backwards = 0          # backward passes since zero_grad()
        self._cb_queued = False
        self._reduced_this_step = False     # this step's buckets are reduced: a second finish_gradient_sync() is a no-op
        self._fwd_since_step = 0            # forward calls since the last optimizer step (fallback without the engine callback)
        self.comm_works = None       # optional (bench.py): list that receives every collective's Work object
        self._early = None           # attach_optimizer(): (optimizer, early parameters, their ids, trigger count, side stream)
        self._early_seen = 0
        if self.world_size > 1:
            self._broadcast_module_state()
        self._build_flat_grads(bucket_bytes)

    def _broadcast_module_state(self):
        tensors = [p.data for p in self.module.parameters()] + [b.data for b in self.module.buffers()]
        engine.broadcast_coalesced(tensors, 0, self.process_group)

    # ------------------------------------------------------------------ flat gradient storage
    def _build_flat_grads(self, bucket_bytes):
        params = [p for p in self.module.parameters() if p.requires_grad]
        self._params = params
        self.flat_grad = None
        self.buckets = []
        if self.world_size == 1 and not self._force_flat:
            # single process: nothing to reduce -> let autograd hand its gradient tensors over as
            # they are (no accumulate-into-view add per parameter, no 1 GB memset per step); the
            # reference's DataParallel is a pass-through on one GPU too (data_parallel.py:107-108)
            return
        big = set(p for p in params if p.numel() * 4 >= self._big_bytes)
        self._big = [p for p in params if p in big]
        total = sum((p.numel() + 3) & ~3 for p in params if p not in big)
        self.flat_grad = torch.zeros(max(total, 1), dtype=torch.float32, device=self.device)
        # reverse registration order ~ order in which backward produces gradients
        off = 0
        spans = []
        self._view_of = {}
        for p in reversed(params):
            if p in big:
                p.grad = None
                spans.append((p, None, None))
                continue
            n = p.numel()
            p.grad = self._view_of[p] = self.flat_grad[off:off + n].view_as(p)
            spans.append((p, off, off + n))
            off += (n + 3) & ~3      # 16-byte aligned views: float4 accesses / matrix mode of the fused SGD stay enabled
        # buckets, in the order the collectives are issued: a big tensor (seg_fc.0: 822 MB) is its own bucket, issued where it
        # stands in the backward order; the small parameters fill flat buckets of ~bucket_bytes that do NOT break at a big
        # tensor - a flat bucket is issued where it is CLOSED (full, or right after the last big tensor, or at the end) -
        # so the few biases between the big weights do not become collectives of their own:
        # fc2 | fc1 | conv | heads + MaskFuse biases | backbone = 5 collectives per step (was 7)
        self.buckets = []
        last_big = max([i for i, (p, a, b) in enumerate(spans) if a is None], default=-1)
        cur = None

        def close():
            nonlocal cur
            if cur is not None:
                self.buckets.append(cur)
                cur = None

        for i, (p, a, b) in enumerate(spans):
            if a is None:
                self.buckets.append(dict(tensor=p, params=[p], ready=0))
                if i == last_big:
                    close()
                continue
            if cur is None:
                cur = dict(start=a, end=b, params=[p], ready=0)
            else:
                cur["end"] = b
                cur["params"].append(p)
            if (cur["end"] - cur["start"]) * 4 >= bucket_bytes:
                close()
        close()
        self._bucket_of = {}
        for bi, bk in enumerate(self.buckets):
            for p in bk["params"]:
                self._bucket_of[p] = bi
        if self.world_size > 1:
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad_ready)
            _gemm_ops.register_publisher(params, self._publish_early)      # big gradients finished inside a fused node go out at once
            if not engine.HAS_ENGINE_CALLBACK:
                # no engine callback (see cim_amd/utils/engine.py): the public-API form of "finish the reduction" is a global
                # optimizer-step pre-hook - whatever the gradient hooks have not reduced is reduced (and waited for) right
                # before ANY optimizer.step(); backward passes are counted by the forward calls since the last step
                from torch.optim.optimizer import register_optimizer_step_pre_hook
                self._step_hook = register_optimizer_step_pre_hook(lambda opt, args, kwargs: self._before_optimizer_step())

    def _before_optimizer_step(self):
        self._cb_queued = False
        if self.world_size > 1 and self._sync and not self._reduced_this_step:
            self.finish_gradient_sync()
        self._reduced_this_step = False
        self._fwd_since_step = 0

    # ------------------------------------------------------------------ optimizer step overlapped with the backward pass
    def attach_optimizer(self, optimizer, early_modules=("Box_Head", "cls_iou_model")):
        """Let the wrapper start the optimizer step of the `early_modules`' parameters INSIDE the last backward pass of an
        optimizer step, as soon as their gradients are final (and, with several ranks, averaged): MaskFuse and the heads
        hold 97 % of the parameter bytes and finish their backward first; their fused SGD update (HBM-bound, ~0.9 ms at
        cfg2) then runs on a side stream under the ROIAlign and backbone backward (latency-bound small launches).
        `optimizer.step()` stays where the driver calls it (tools/train.py:438) and updates the rest.
        Needs cim_amd.optim.SGD (step_early); any other optimizer is left alone."""
        if not hasattr(optimizer, "step_early") or self.device.type != "cuda" or not engine.HAS_ENGINE_CALLBACK:
            return False
        params = []
        for name in early_modules:
            mod = getattr(self.module, name, None)
            if mod is not None:
                params += [p for p in mod.parameters() if p.requires_grad]
        if not params:
            return False
        self._early = (optimizer, params, frozenset(id(p) for p in params), torch.cuda.Stream(device=self.device))
        # the early update needs the MaskFuse weight gradients INSIDE the backward pass: they must come through autograd (whose
        # hooks trigger it), not be installed at the end of the pass
        from ...ops import maskfuse_pair as _mfp
        _mfp.DEFER_DW = False
        if self.world_size == 1:          # (with several ranks every parameter already carries the hook)
            for p in params:
                p.register_post_accumulate_grad_hook(self._on_grad_ready)
        return True

    def _maybe_step_early(self, p):
        """Called from the gradient hooks: when the last early parameter of the last backward pass of the optimizer step
        has its gradient, reduce what is reduced so far and start the early update."""
        if self._early is None or id(p) not in self._early[2]:
            return
        self._early_seen += 1
        if self._early_seen < len(self._early[1]) or not self._sync_this_backward():
            return
        opt, params, _, side = self._early
        if self.world_size > 1:
            # their buckets were launched by the hooks (strict order); the side stream waits for exactly those collectives
            if any(self._bucket_of[q] >= self._next_bucket for q in params):
                return                    # a bucket mixes early and late parameters: not reduced yet - update at step()
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for w, buf in self._pending:
                    w.wait()
                    if buf is not None:
                        buf.div_(self.world_size)
            self._pending = []
        opt.step_early(params, side)

    def _sync_this_backward(self):
        if not engine.HAS_ENGINE_CALLBACK:        # (backward passes cannot be counted without the engine callback)
            return self._sync and self._fwd_since_step % max(self.iter_size, 1) == 0
        return self._sync and (self._backwards + 1) % max(self.iter_size, 1) == 0

    def _backward_started(self):
        if not self._cb_queued:       # first gradient of this backward pass: finish the reduction when the pass ends
            self._cb_queued = True
            self._early_seen = 0
            self._reduced_this_step = False
            if engine.HAS_ENGINE_CALLBACK:
                engine.queue_callback(self._end_of_backward)

    def _publish_early(self, p, g, stream):
        """ops.gemm.publisher_for(param): a fused Function finished the gradient `g` of the big parameter `p` on `stream` in the middle of
        its backward.  When `p` is one of this wrapper's parameters and this backward pass communicates: install it (autograd will
        not: the Function then returns None for it), let its bucket's all-reduce start on that stream at once, in the strict bucket
        order, and return True.  Otherwise return False: the Function keeps its own (joined or deferred) path."""
        if self.world_size == 1 or p not in self._bucket_of or not self._sync_this_backward():
            return False
        self._backward_started()
        with torch.cuda.stream(stream):
            view = self._view_of.get(p)
            if view is not None:
                # a parameter of a FLAT bucket (a MaskFuse weight under big_bytes: small-channel configurations): the all-reduce
                # runs over the flat buffer, so the gradient must be IN its view before the bucket goes out - a fresh tensor
                # installed as .grad (the driver dropped the views: zero_grad(set_to_none=True)) would only be copied in by
                # _rebind_views() at the end of the pass, after the bucket was reduced without it
                if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                    view.copy_(g) if p.grad is None else view.copy_(p.grad + g)
                    p.grad = view
                else:
                    view += g
            elif p.grad is None:
                p.grad = g
            else:
                p.grad += g
            self.buckets[self._bucket_of[p]]["ready"] += 1
            self._launch_ready_buckets()
        return True

    def _on_grad_ready(self, p):
        self._backward_started()
        if self.world_size == 1:
            self._maybe_step_early(p)
            return
        if not self._sync_this_backward():
            return
        if _gradient_is_deferred(p):      # its gradient is still on the side stream (cim_amd/ops/gemm.py): the hook fires at the layer's
            return                        # backward, the gradient arrives at the join - the bucket goes out at the end of backward
        bk = self.buckets[self._bucket_of[p]]
        bk["ready"] += 1
        if "tensor" not in bk and (p.grad.data_ptr() < self.flat_grad.data_ptr() or
                                   p.grad.data_ptr() >= self.flat_grad.data_ptr() + self.flat_grad.numel() * 4):
            # the driver dropped the flat view (optimizer.zero_grad() defaults to set_to_none=True - the reference loop,
            # tools/train.py:418-438, calls exactly that): autograd produced a fresh tensor.  Move it into the view (it
            # REPLACES the view's content: whatever the previous step left there) and bind the view again, so the bucket is
            # reduced in place and later backward passes of an accumulation accumulate into the flat buffer.
            view = self._view_of[p]
            view.copy_(p.grad)
            p.grad = view
        self._launch_ready_buckets()
        self._maybe_step_early(p)

    def _launch_ready_buckets(self, force=False):
        """Buckets are reduced STRICTLY in index order, so every rank issues the same sequence of
        collectives even if a rank's autograd produced gradients in a different order or skipped
        a parameter (e.g. a refinement head whose CIM layer returned None on that image)."""
        if force:
            # the backbone's weight gradients run on the side stream and do not come through autograd (cim_amd/ops/gemm.py):
            # their hooks never fire, so the bucket that holds them is only ever launched here, at the end of the backward
            # pass - after the join that adds them into the flat gradient views
            _join_side()
            self._rebind_views()
        while self._next_bucket < len(self.buckets):
            bk = self.buckets[self._next_bucket]
            if not force and bk["ready"] < len(bk["params"]):
                break
            bk["ready"] = 0
            if "tensor" in bk:
                p = bk["tensor"]
                if p.grad is None:          # this rank's autograd produced nothing for it: still take part in the collective
                    p.grad = torch.zeros_like(p)
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                buf = p.grad
            else:
                buf = self.flat_grad[bk["start"]:bk["end"]]
            self._all_reduce_mean(buf)
            self._next_bucket += 1

    def _rebind_views(self):
        """Gradients that arrived as fresh tensors instead of in the flat views - the driver dropped the views
        (optimizer.zero_grad(set_to_none=True)) and either the backward passes so far did not communicate (iter_size > 1)
        or the gradient was installed by join_side() - move into their views before the flat buckets are reduced."""
        lo = self.flat_grad.data_ptr()
        hi = lo + self.flat_grad.numel() * 4
        for p, view in self._view_of.items():
            g = p.grad
            if g is not None and not (lo <= g.data_ptr() < hi):
                view.copy_(g)
                p.grad = view

    def _all_reduce_mean(self, buf):
        if self._avg_op is None:
            self._avg_op = dist.get_backend(self.process_group) == "nccl"
        if self._avg_op:
            w = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.process_group, async_op=True)
            self._pending.append((w, None))
        else:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
            self._pending.append((w, buf))
        if self.comm_works is not None:
            self.comm_works.append(w)

    def _end_of_backward(self):
        self._cb_queued = False
        if self.world_size > 1 and self._sync_this_backward():
            self.finish_gradient_sync()
            self._reduced_this_step = True
        self._backwards += 1

    def zero_grad(self, set_to_none=False):
        """One memset of the flat buffer (param.grad stay views into it); without a flat buffer
        (single process) the gradients are simply dropped."""
        if self._pending or self._next_bucket:
            raise RuntimeError("DataParallel.zero_grad(): gradient all-reduces of the previous backward are still in "
                               "flight (a backward pass was interrupted?) - call finish_gradient_sync() first")
        self._backwards = 0
        self._reduced_this_step = False
        self._cb_queued = False             # (a backward pass that was aborted before the engine's callbacks ran leaves it set)
        if self.flat_grad is None:
            for p in self._params:
                p.grad = None
        else:
            self.flat_grad.zero_()
            for p in self._big:             # autograd steals the next gradient tensor instead of accumulating into a view
                p.grad = None

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (tools/train.py --iter_size): skip the all-reduce inside."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def finish_gradient_sync(self):
        """Reduce whatever has not been reduced yet (in bucket order) and wait for the in-flight
        all-reduces.  Runs by itself at the end of every backward pass that
        communicates (`_end_of_backward`); calling it again is a no-op."""
        if self.world_size > 1 and self._sync and not self._reduced_this_step and (
                self._next_bucket or any(bk["ready"] for bk in self.buckets) or not self._cb_queued):
            self._launch_ready_buckets(force=True)
        for w, buf in self._pending:
            w.wait()
            if buf is not None:
                buf.div_(self.world_size)
        self._pending = []
        self._next_bucket = 0
        for bk in self.buckets:
            bk["ready"] = 0

    def loss_scale(self):
        """Kept for drivers written against the first version of this wrapper: the all-reduce averages, so the loss
        needs no rescaling."""
        return 1.0

    # ------------------------------------------------------------------ forward
    def _to_device(self, k, v):
        if k in self.cpu_keywords or not torch.is_tensor(v) or self.device.type != "cuda":
            return v
        return v.to(self.device, non_blocking=True)

    def forward(self, *inputs, **kwargs):
        self._fwd_since_step += 1
        if self.minibatch:
            # the reference passes one list entry per local GPU; this process owns exactly one
            inputs = [x[0] if isinstance(x, (list, tuple)) else x for x in inputs]
            kwargs = {k: (v[0] if isinstance(v, (list, tuple)) else v) for k, v in kwargs.items()}
        inputs = [self._to_device(None, x) for x in inputs]
        kwargs = {k: self._to_device(k, v) for k, v in kwargs.items()}
        return self.module(*inputs, **kwargs)
