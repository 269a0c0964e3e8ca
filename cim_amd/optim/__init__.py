"""Optimizers of the training step."""
from .sgd import SGD

__all__ = ["SGD"]
