"""Fused multi-tensor SGD (cim_amd/csrc/sgd.hip) behind torch.optim.SGD's interface.

Drop-in for `torch.optim.SGD(params, momentum=cfg.SOLVER.MOMENTUM)` at /root/reference/tools/train.py:308-309: same
param-group keys (lr, momentum, weight_decay, dampening, nesterov), same `state[p]['momentum_buffer']` tensors (so
lib/utils/net.py:47-83 `update_learning_rate` / `_CorrectMomentum` and the checkpoint code keep working), same update
rule - but ONE kernel launch per step for all parameters instead of one multi-tensor launch per ~20 tensors.
Only what the reference uses is supported on the HIP path: fp32 CUDA/HIP parameters, dampening 0, no Nesterov,
dense gradients, one momentum value; anything else raises (there is no silent fallback).

Host cost per step: autograd hands out new gradient tensors every step, so the 40-byte record of every tensor is
refreshed (one pass over the parameters, one 6 KB H2D copy); the chunk table only depends on the tensor sizes and is
built once.
"""
import numpy as np
import torch

from .. import _lib

CHUNK = 16384          # elements per workgroup
_TENSOR = np.dtype([("p", "<u8"), ("g", "<u8"), ("buf", "<u8"), ("n", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
_CHUNK = np.dtype([("tensor", "<i4"), ("n", "<i4"), ("offset", "<i8")])


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if dampening != 0.0 or nesterov:
            raise NotImplementedError("cim_amd.optim.SGD: dampening / Nesterov are not used by the reference and not provided")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        self._layout = None         # tuple of element counts the chunk table on the device was built for
        self._chunks = None         # device chunk table
        self._n_chunks = 0
        self._stage = None          # (pinned host array, device array, event of the last H2D copy) of the tensor records
        self._last = None           # tensor records of the last step (skip the copy when nothing changed)

    def _build_chunks(self, sizes, dev):
        counts = np.array([(n + CHUNK - 1) // CHUNK for n in sizes], dtype=np.int64)
        total = int(counts.sum())
        owner = np.repeat(np.arange(len(sizes)), counts)
        first = np.concatenate([[0], np.cumsum(counts)[:-1]])
        tab = np.empty(total, dtype=_CHUNK)
        tab["tensor"] = owner
        tab["offset"] = (np.arange(total) - first[owner]) * CHUNK
        tab["n"] = CHUNK
        self._chunks = torch.from_numpy(tab.view(np.uint8).reshape(-1).copy()).to(dev)
        self._n_chunks, self._layout = total, tuple(sizes)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        recs, touched, momentum, dev = [], [], None, None
        for group in self.param_groups:
            if group.get("dampening", 0.0) != 0.0 or group.get("nesterov", False):
                raise NotImplementedError("cim_amd.optim.SGD: dampening / Nesterov")
            if momentum is None:
                momentum = float(group["momentum"])
            elif float(group["momentum"]) != momentum:
                raise NotImplementedError("cim_amd.optim.SGD: one momentum value for all groups")
            lr, wd = float(group["lr"]), float(group["weight_decay"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda:
                    raise _lib.CimHipError("cim_amd.optim.SGD: CUDA/HIP parameters required (no CPU fallback)")
                if p.dtype != torch.float32 or g.dtype != torch.float32 or g.is_sparse or not p.is_contiguous():
                    raise NotImplementedError("cim_amd.optim.SGD: dense contiguous fp32 parameters and gradients")
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                buf = st.get("momentum_buffer")
                if buf is None:
                    buf = st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                dev = p.device
                recs.append((p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), lr, wd, g))
                touched.append(p)
                touched.append(buf)
        if not recs:
            return loss
        sizes = tuple(r[3] for r in recs)
        if sizes != self._layout:
            self._build_chunks(sizes, dev)
            self._stage = self._last = None
        tab = np.array([r[:6] for r in recs], dtype=_TENSOR)
        if self._last is None or not np.array_equal(tab, self._last):
            nbytes = tab.nbytes
            if self._stage is None:
                self._stage = [torch.empty(nbytes, dtype=torch.uint8).pin_memory(), torch.empty(nbytes, dtype=torch.uint8, device=dev), None]
            pinned, table, copied = self._stage
            if copied is not None:
                copied.synchronize()            # the previous H2D copy out of the staging buffer (long done in practice)
            pinned.numpy()[:] = tab.view(np.uint8).reshape(-1)
            table.copy_(pinned, non_blocking=True)
            self._stage[2] = torch.cuda.Event()
            self._stage[2].record()
            self._last = tab
        _lib.call("cim_sgd_multi", self._stage[1].data_ptr(), self._chunks.data_ptr(), self._n_chunks, momentum, _lib.stream_ptr())
        # the kernel wrote parameters and momentum buffers through raw pointers: tell autograd's version counters, so that
        # anything keyed by Tensor._version (saved-tensor checks, caches) sees the in-place update
        torch.autograd.graph.increment_version(touched)
        return loss
