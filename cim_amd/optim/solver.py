"""Optimizer construction and the learning-rate schedule of the reference's training loop.

`param_groups` / `make_optimizer` restate /root/reference/tools/train.py:282-311 (two groups: weights, and biases with
doubled learning rate and no weight decay; lr 0 until the schedule sets it); `LRSchedule.before_step` restates the
warm-up and step-decay bookkeeping of tools/train.py:388-414.  The update itself is the fused HIP kernel behind
`cim_amd.optim.SGD`.
"""
from ..core.config import cfg
from ..utils import net as net_utils
from .sgd import SGD


def param_groups(model):
    bias, nonbias = [], []
    for key, value in model.named_parameters():
        if value.requires_grad:
            (bias if "bias" in key else nonbias).append(value)
    return [
        {"params": nonbias, "lr": 0, "weight_decay": cfg.SOLVER.WEIGHT_DECAY},
        {"params": bias, "lr": 0 * (cfg.SOLVER.BIAS_DOUBLE_LR + 1),
         "weight_decay": cfg.SOLVER.WEIGHT_DECAY if cfg.SOLVER.BIAS_WEIGHT_DECAY else 0},
    ]


def make_optimizer(model):
    if cfg.SOLVER.TYPE != "SGD":
        raise NotImplementedError("cim_amd.optim: SOLVER.TYPE %r - the shipped configs train with SGD" % (cfg.SOLVER.TYPE,))
    return SGD(param_groups(model), momentum=cfg.SOLVER.MOMENTUM)


class LRSchedule:
    """`lr = schedule.before_step(step)` at the top of every iteration, as the loop of train.py:385-414 does inline."""

    def __init__(self, optimizer, start_step=0):
        self.optimizer = optimizer
        self.lr = optimizer.param_groups[0]["lr"]       # 0: "a dummy value to be set properly at the start of training"
        self.decay_steps_ind = None
        for i in range(1, len(cfg.SOLVER.STEPS)):       # train.py:376-381
            if cfg.SOLVER.STEPS[i] >= start_step:
                self.decay_steps_ind = i
                break
        if self.decay_steps_ind is None:
            self.decay_steps_ind = len(cfg.SOLVER.STEPS)

    def before_step(self, step):
        s = cfg.SOLVER
        if step < s.WARM_UP_ITERS:
            if s.WARM_UP_METHOD == "constant":
                factor = s.WARM_UP_FACTOR
            elif s.WARM_UP_METHOD == "linear":
                alpha = step / s.WARM_UP_ITERS
                factor = s.WARM_UP_FACTOR * (1 - alpha) + alpha
            else:
                raise KeyError("Unknown SOLVER.WARM_UP_METHOD: {}".format(s.WARM_UP_METHOD))
            self._set(s.BASE_LR * factor)
        elif step == s.WARM_UP_ITERS:
            self._set(s.BASE_LR)
        if self.decay_steps_ind < len(s.STEPS) and step == s.STEPS[self.decay_steps_ind]:
            self._set(self.lr * s.GAMMA)
            self.decay_steps_ind += 1
        return self.lr

    def _set(self, new_lr):
        net_utils.update_learning_rate(self.optimizer, self.lr, new_lr)
        self.lr = self.optimizer.param_groups[0]["lr"]
        assert self.lr == new_lr
