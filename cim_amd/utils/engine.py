"""The two PyTorch internals the training step leans on, behind one guard with public-API fallbacks.

* `queue_callback(fn)`: run `fn` once when the CURRENT backward pass has finished (torch.autograd's engine callback,
  `Variable._execution_engine.queue_callback` - what DistributedDataParallel itself uses, but private).  Users: the join of
  the side-stream weight gradients (ops/gemm.py), the NumPy-generator rewind of the device-side anti-noise sampling
  (modeling/heads.py), nn.DataParallel's end-of-backward reduction.
* `broadcast_coalesced(tensors, src, group)`: `dist._broadcast_coalesced`.

When an internal is missing (a torch upgrade), `HAS_ENGINE_CALLBACK` is False and every user takes its documented
synchronous form instead of failing: weight gradients travel through autograd on the caller's stream (the
behaviour of `ops.gemm.DEFER_DW = False`), the generator is settled inside the forward, DataParallel finishes its reduction
from a `torch.autograd.graph.register_multi_grad_hook` over its parameters; the broadcast falls back to one
`dist.broadcast` per tensor.  CIM_NO_ENGINE_CALLBACK=1 forces the fallbacks (tests/test_dp_gloo.py runs both)."""
import os

import torch


def _engine():
    return getattr(getattr(torch.autograd, "Variable", None), "_execution_engine", None)


_FAST_PATHS = os.environ.get("CIM_NO_ENGINE_CALLBACK", "0") != "1"      # (=1: the fallbacks for a torch build without the private hooks)
HAS_ENGINE_CALLBACK = _FAST_PATHS and callable(getattr(_engine(), "queue_callback", None))


def queue_callback(fn):
    """Run fn() when the backward pass that is executing now has finished.  Only valid from inside a backward pass and when
    HAS_ENGINE_CALLBACK (callers check it and take their synchronous form otherwise)."""
    if not HAS_ENGINE_CALLBACK:
        raise RuntimeError("torch's autograd engine callback is unavailable: callers must check engine.HAS_ENGINE_CALLBACK")
    _engine().queue_callback(fn)


def broadcast_coalesced(tensors, src=0, group=None, buffer_bytes=256 << 20):
    import torch.distributed as dist
    fn = getattr(dist, "_broadcast_coalesced", None)
    if fn is not None and _FAST_PATHS:
        pg = group if group is not None else dist.group.WORLD
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for same in by_dtype.values():
            fn(pg, same, buffer_bytes, src)
        return
    for t in tensors:                      # public API: one collective per tensor
        dist.broadcast(t, src=src, group=group)
