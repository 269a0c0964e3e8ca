"""Book-keeping of the places where a GPU tensor would leave the hand-written HIP kernels for an ATen / MIOpen library op.

Every wrapper in cim_amd.ops that has an ATen branch calls `note(op, reason)` when a CUDA/HIP tensor is about to take it: counted
per (op, reason) and - round 6: BY DEFAULT - an ERROR (`CimHipError`): the product has one code path, a geometry the kernels do not
take (a BatchNorm in training mode, a dilated strided convolution, ...) fails loudly instead of quietly running MIOpen.  Opting
OUT is explicit and per operator: `CIM_ALLOW_FALLBACK=op,...` (or `*`) in the environment, or `with fallback.allowed("op"):` in
code (the tests of the wrappers' library branches, tools/bench_library_paths.py); an allowed fallback is logged once and counted.
`counts()` / `reset()` are for profiles and tests (bench.py reports the counts: `{}` at every shipped configuration).
CPU tensors (host-side tests of the modules) are not fallbacks and are not recorded."""
import logging
import os

from .. import _lib

_COUNTS = {}
_LOG = logging.getLogger("cim_amd.fallback")
# shapes the shipped configurations never produce but callers may (tests of the wrappers themselves exercise them on purpose)
ALLOW = set(filter(None, os.environ.get("CIM_ALLOW_FALLBACK", "").split(",")))


def strict():
    """CIM_STRICT=1 (the test suite's setting): leaving the reference's NumPy stream inside a step (modeling/heads.py: _RngLedger) is
    an error instead of a warning.  (Library fallbacks are errors whatever this says.)"""
    return os.environ.get("CIM_STRICT", "0") == "1"


def note(op, reason):
    key = (op, reason)
    n = _COUNTS.get(key, 0)
    _COUNTS[key] = n + 1
    if op not in ALLOW and "*" not in ALLOW:
        raise _lib.CimHipError("%s: a GPU tensor would take an ATen / library branch (%s) - the product has no second code path; "
                               "allow it explicitly with CIM_ALLOW_FALLBACK=%s or `with cim_amd.ops.fallback.allowed(%r):`"
                               % (op, reason, op, op))
    if n == 0:
        _LOG.warning("%s: ATen / library fallback on a GPU tensor (%s), allowed explicitly", op, reason)


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
