"""Fused multi-tensor SGD (cim_amd/csrc/sgd.hip) behind torch.optim.SGD's interface.

Drop-in for `torch.optim.SGD(params, momentum=cfg.SOLVER.MOMENTUM)` at /root/reference/tools/train.py:308-309: same
param-group keys (lr, momentum, weight_decay, dampening, nesterov), same `state[p]['momentum_buffer']` tensors (so
lib/utils/net.py:47-83 `update_learning_rate` / `_CorrectMomentum` and the checkpoint code keep working), same update
rule - but ONE kernel launch per step for all parameters instead of one multi-tensor launch per ~20 tensors.
Only what the reference uses is supported on the HIP path: fp32 CUDA/HIP parameters, dampening 0, no Nesterov,
dense gradients, one momentum value; anything else raises (there is no silent fallback).
"""
import numpy as np
import torch

from .. import _lib

CHUNK = 16384          # elements per workgroup
_REC = np.dtype([("p", "<u8"), ("g", "<u8"), ("buf", "<u8"), ("n", "<i4"), ("aligned", "<i4"), ("lr", "<f4"), ("wd", "<f4")])


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if dampening != 0.0 or nesterov:
            raise NotImplementedError("cim_amd.optim.SGD: dampening / Nesterov are not used by the reference and not provided")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)
        self._pinned = None
        self._table = None
        self._copied = None         # event: the last H2D copy of the staging buffer
        self._key = None            # identity of the table on the device (pointers, lr, wd of every tensor)
        self._total = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        recs, momentum, dev = [], None, None
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
                n, pp, gp, bp = p.numel(), p.data_ptr(), g.data_ptr(), buf.data_ptr()
                aligned = int(pp % 16 == 0 and gp % 16 == 0 and bp % 16 == 0)
                recs.append((pp, gp, bp, n, aligned, lr, wd, g))
        if not recs:
            return loss
        key = tuple(r[:7] for r in recs)
        if key == self._key:        # same tensors as last step (flat-gradient data-parallel mode): the table is still valid
            _lib.call("cim_sgd_multi", self._table.data_ptr(), self._total, momentum, _lib.stream_ptr())
            return loss
        # chunk table (vectorised): every tensor contributes ceil(n / CHUNK) records
        counts = np.array([(r[3] + CHUNK - 1) // CHUNK for r in recs], dtype=np.int64)
        total = int(counts.sum())
        owner = np.repeat(np.arange(len(recs)), counts)
        first = np.concatenate([[0], np.cumsum(counts)[:-1]])
        off = (np.arange(total) - first[owner]) * CHUNK
        n_of = np.array([r[3] for r in recs], dtype=np.int64)[owner]
        tab = np.empty(total, dtype=_REC)
        for name, idx in (("p", 0), ("g", 1), ("buf", 2)):
            tab[name] = np.array([r[idx] for r in recs], dtype=np.uint64)[owner] + (off * 4).astype(np.uint64)
        tab["n"] = np.minimum(CHUNK, n_of - off)
        tab["aligned"] = np.array([r[4] for r in recs], dtype=np.int32)[owner]
        tab["lr"] = np.array([r[5] for r in recs], dtype=np.float32)[owner]
        tab["wd"] = np.array([r[6] for r in recs], dtype=np.float32)[owner]
        nbytes = tab.nbytes
        if self._pinned is None or self._pinned.numel() < nbytes:
            self._pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            self._table = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._copied = None
        if self._copied is not None:
            self._copied.synchronize()          # the previous H2D copy out of the staging buffer (long done in practice)
        self._pinned[:nbytes].numpy()[:] = tab.view(np.uint8).reshape(-1)
        self._table[:nbytes].copy_(self._pinned[:nbytes], non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record()
        self._key, self._total = key, total
        _lib.call("cim_sgd_multi", self._table.data_ptr(), total, momentum, _lib.stream_ptr())
        return loss
