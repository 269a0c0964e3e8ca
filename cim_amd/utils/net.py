"""Learning-rate / momentum-history helpers of the training loop.

Same names, arguments and behaviour as /root/reference/lib/utils/net.py:14-89 (`clip_gradient`,
`decay_learning_rate`, `update_learning_rate`, `_CorrectMomentum`, `_get_lr_change_ratio`), which
tools/train.py:388-414 calls around every optimizer step.  They only touch `optimizer.param_groups`
and `optimizer.state[p]['momentum_buffer']`, both of which `cim_amd.optim.SGD` keeps exactly as
`torch.optim.SGD` does - so they work with either optimizer.  The per-step learning rate reaches the
fused HIP update through the per-tensor records `cim_amd.optim.SGD.step` rebuilds every step.
"""
import logging

import numpy as np
import torch

from ..core.config import cfg

logger = logging.getLogger(__name__)


def clip_gradient(model, clip_norm):
    """Scale all gradients by clip_norm / max(total L2 norm, clip_norm)  (net.py:14-27)."""
    grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    if not grads:
        return
    total = torch.sqrt(sum(g.detach().norm() ** 2 for g in grads)).item()
    norm = clip_norm / max(total, clip_norm)
    for g in grads:
        g.mul_(norm)


def _get_lr_change_ratio(cur_lr, new_lr):
    eps = 1e-10
    return float(np.max((new_lr / np.max((cur_lr, eps)), cur_lr / np.max((new_lr, eps)))))


def _CorrectMomentum(optimizer, param_keys, correction):
    """V := mu * V + lr * grad carries the learning rate inside the history V, so a changed learning rate
    rescales V by new_lr / old_lr (net.py:65-82).  Parameters without a history yet are skipped."""
    logger.info("Scaling update history by %.6f (new lr / old lr)", correction)
    if hasattr(optimizer, "wait_update"):
        optimizer.wait_update()          # (cim_amd.optim.SGD.overlap_update: the big weights' histories may still be written on the side stream)
    for p in param_keys:
        buf = optimizer.state.get(p, {}).get("momentum_buffer")
        if buf is not None:
            buf.mul_(correction)


def _scale_momentum(cur_lr, ratio):
    return (cfg.SOLVER.TYPE in ["SGD"] and cfg.SOLVER.SCALE_MOMENTUM and cur_lr > 1e-7
            and ratio > cfg.SOLVER.SCALE_MOMENTUM_THRESHOLD)


def decay_learning_rate(optimizer, cur_lr, decay_rate):
    """Multiply every group's own learning rate by decay_rate (net.py:30-45)."""
    new_lr = cur_lr * decay_rate
    ratio = 1 / decay_rate
    if ratio > cfg.SOLVER.LOG_LR_CHANGE_THRESHOLD:
        logger.info("Changing learning rate %.6f -> %.6f", cur_lr, new_lr)
    for group in optimizer.param_groups:
        cur = group["lr"]
        new = decay_rate * group["lr"]
        group["lr"] = new
        if _scale_momentum(cur, ratio):
            _CorrectMomentum(optimizer, group["params"], new / cur)


def update_learning_rate(optimizer, cur_lr, new_lr):
    """Set the learning rate (group 1 = biases: x2 under BIAS_DOUBLE_LR) and rescale the momentum history when
    the change exceeds SCALE_MOMENTUM_THRESHOLD (net.py:47-63)."""
    if cur_lr == new_lr:
        return
    ratio = _get_lr_change_ratio(cur_lr, new_lr)
    if ratio > cfg.SOLVER.LOG_LR_CHANGE_THRESHOLD:
        logger.info("Changing learning rate %.6f -> %.6f", cur_lr, new_lr)
    keys = []
    for ind, group in enumerate(optimizer.param_groups):
        group["lr"] = new_lr * 2 if (ind == 1 and cfg.SOLVER.BIAS_DOUBLE_LR) else new_lr
        keys += group["params"]
    if _scale_momentum(cur_lr, ratio):
        _CorrectMomentum(optimizer, keys, new_lr / cur_lr)
