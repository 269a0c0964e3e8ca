"""Optimizers of the training step."""
from .sgd import SGD
from .solver import LRSchedule, make_optimizer, param_groups

__all__ = ["SGD", "LRSchedule", "make_optimizer", "param_groups"]
