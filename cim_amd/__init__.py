"""cim_amd: MI355X-native implementation of the ZechengLi19/CIM per-image training step.

`install_as_lib()` registers the sub-packages under the top-level names the reference's
tools/train.py imports after `_init_paths` puts `lib/` on sys.path (`core`, `modeling`, `ops`,
`nn`, `utils`), so the reference driver resolves to this implementation.
"""
import importlib
import sys

__version__ = "0.1.0"


def install_as_lib():
    for name in ("core", "modeling", "ops", "nn", "utils"):
        sys.modules.setdefault(name, importlib.import_module("cim_amd." + name))
    for sub in ("core.config", "modeling.heads", "modeling.model_builder", "modeling.resnet50", "modeling.vgg16"):
        sys.modules.setdefault(sub, importlib.import_module("cim_amd." + sub))
