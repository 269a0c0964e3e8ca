"""cim_amd: MI355X-native implementation of the ZechengLi19/CIM per-image training step.

`install_as_lib()` makes the reference's own driver (`tools/train.py:29-39`: `import nn as mynn`,
`from modeling.model_builder import Generalized_RCNN`, ...) resolve the modules on the training hot path to
this package while everything else (`utils.*`, `datasets`, `roi_data`, `core.config`, `nn.modules`, `nn.init`)
stays the reference's own code.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

__version__ = "0.2.0"

# reference module name (under lib/) -> module of this package that replaces it
ALIASES = {
    "ops": "cim_amd.ops",                                              # lib/ops/__init__.py:6 (mmcv.ops re-export)
    "modeling.heads": "cim_amd.modeling.heads",                        # lib/modeling/heads.py
    "modeling.model_builder": "cim_amd.modeling.model_builder",        # lib/modeling/model_builder.py
    "modeling.resnet50": "cim_amd.modeling.resnet50",                  # lib/modeling/resnet50.py
    "modeling.vgg16": "cim_amd.modeling.vgg16",                        # lib/modeling/vgg16.py
    "modeling.HRNet": "cim_amd.modeling.HRNet",                        # lib/modeling/HRNet.py
    "nn.parallel": "cim_amd.nn.parallel",                              # lib/nn/parallel/__init__.py (DataParallel)
    "nn.parallel.data_parallel": "cim_amd.nn.parallel.data_parallel",  # lib/nn/parallel/data_parallel.py
}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)      # the real module object, imported under its real name

    def exec_module(self, module):
        pass


class _CfgBindingLoader(importlib.abc.Loader):
    """Runs the reference's own `core/config.py`, then points cim_amd's `cfg` proxy at its `cfg` object."""

    def __init__(self, inner):
        self.inner = inner

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        from .core import config as own
        own.cfg.bind(module.cfg)


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        real = ALIASES.get(fullname)
        if real is not None:
            is_pkg = fullname in ("ops", "nn.parallel")
            return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real), is_package=is_pkg)
        if fullname == "core.config":
            spec = importlib.machinery.PathFinder.find_spec(fullname, path)
            if spec is not None and spec.loader is not None:
                spec.loader = _CfgBindingLoader(spec.loader)
            return spec
        return None


_finder = _Finder()


def install_as_lib():
    """Call once before the reference's `lib/` modules are imported (e.g. first line of tools/train.py).
    Idempotent.  Modules of `ALIASES` that the process already imported from the reference tree are replaced."""
    if _finder not in sys.meta_path:
        sys.meta_path.insert(0, _finder)
    for name in ALIASES:
        mod = sys.modules.get(name)
        if mod is not None and mod.__name__ != ALIASES[name]:
            del sys.modules[name]
    ref_cfg = sys.modules.get("core.config")
    if ref_cfg is not None and hasattr(ref_cfg, "cfg"):
        from .core import config as own
        own.cfg.bind(ref_cfg.cfg)


def uninstall_as_lib():
    """Undo `install_as_lib()` (tests)."""
    if _finder in sys.meta_path:
        sys.meta_path.remove(_finder)
    for name, real in ALIASES.items():
        mod = sys.modules.get(name)
        if mod is not None and mod.__name__ == real:
            del sys.modules[name]
    from .core import config as own
    own.reset_cfg()
