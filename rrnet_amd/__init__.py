"""rrnet_amd — MI355X-native (gfx950) implementation of RRNet's detection hot path behind the
reference's own Python API (models.rrnet / operators.rrnet_operator / detectors /
ext.nms.nms_wrapper ...).  See DESIGN.md and INTEGRATION.md."""
import importlib
import importlib.abc
import importlib.util
import sys

_ALIASES = ("backbones", "configs", "datasets", "detectors", "ext", "models", "modules", "operators", "utils")


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`models.rrnet` IS `rrnet_amd.models.rrnet` — the same module object under both names, at any depth — so
    classes, Config tables and module state exist once."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".", 1)[0] not in _ALIASES:
            return None
        try:
            real = importlib.import_module("rrnet_amd." + fullname)
        except ModuleNotFoundError as e:
            if e.name and ("rrnet_amd." + fullname).startswith(e.name):
                return None                       # no such module in this package: let the normal machinery decide
            raise
        spec = importlib.util.spec_from_loader(fullname, self, is_package=hasattr(real, "__path__"))
        spec._rr_real = real
        return spec

    def create_module(self, spec):
        return spec._rr_real

    def exec_module(self, module):
        pass


_finder = _AliasFinder()


def install_aliases():
    """Make the reference's top-level import names (`from models.rrnet import RRNet`,
    `from operators.rrnet_operator import RRNetOperator`, ...) resolve to this package, so the
    reference's scripts/RRNet/{train,eval}.py run unchanged.  Overrides an unrelated pip
    package called `datasets` for this process.  The `shims/` directory at the repository root does the same
    without a single edited line: `PYTHONPATH=<repo>/shims python scripts/RRNet/train.py`."""
    if _finder not in sys.meta_path:
        sys.meta_path.insert(0, _finder)
    for name in _ALIASES:
        for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
            mod = sys.modules[k]
            if not getattr(mod, "__name__", "").startswith("rrnet_amd."):
                del sys.modules[k]                # a foreign package of the same name (pip `datasets`)
