"""Book-keeping of the places where a GPU tensor leaves the hand-written HIP kernels for an ATen / MIOpen library op.

Every wrapper in cim_amd.ops that has an ATen branch calls `note(op, reason)` when a CUDA/HIP tensor takes it: counted per
(op, reason), logged once each, and an ERROR under CIM_STRICT=1 (the test suite sets it: a silent fallback would let the
GPU tests pass on library code).  `counts()` / `reset()` are for profiles and tests (tools/bench reports the counts).
CPU tensors (host-side tests of the modules) are not fallbacks and are not recorded."""
import logging
import os

from .. import _lib

_COUNTS = {}
_LOG = logging.getLogger("cim_amd.fallback")
# shapes the shipped configurations never produce but callers may (tests of the wrappers themselves exercise them on purpose)
ALLOW = set(filter(None, os.environ.get("CIM_ALLOW_FALLBACK", "").split(",")))


def strict():
    return os.environ.get("CIM_STRICT", "0") == "1"


def note(op, reason):
    key = (op, reason)
    n = _COUNTS.get(key, 0)
    _COUNTS[key] = n + 1
    if n == 0:
        _LOG.warning("%s: ATen / library fallback on a GPU tensor (%s)", op, reason)
    if strict() and op not in ALLOW and "*" not in ALLOW:
        raise _lib.CimHipError("%s: ATen / library fallback on a GPU tensor (%s) under CIM_STRICT=1" % (op, reason))


def counts():
    return dict(_COUNTS)


def reset():
    _COUNTS.clear()


class allowed:
    """with fallback.allowed("conv3x3_bn_act"): ... - a test that exercises an ATen branch on purpose."""

    def __init__(self, *ops):
        self.ops = ops

    def __enter__(self):
        self.added = [o for o in self.ops if o not in ALLOW]
        ALLOW.update(self.added)

    def __exit__(self, *exc):
        for o in self.added:
            ALLOW.discard(o)
