"""rrnet_amd — MI355X-native (gfx950) implementation of RRNet's detection hot path behind the
reference's own Python API (models.rrnet / operators.rrnet_operator / detectors /
ext.nms.nms_wrapper ...).  See DESIGN.md and INTEGRATION.md."""
import importlib
import sys

_ALIASES = ("backbones", "configs", "datasets", "detectors", "ext", "models", "modules", "operators", "utils")


def install_aliases():
    """Make the reference's top-level import names (`from models.rrnet import RRNet`,
    `from operators.rrnet_operator import RRNetOperator`, ...) resolve to this package, so the
    reference's scripts/RRNet/{train,eval}.py run unchanged.  Overrides an unrelated pip
    package called `datasets` for this process."""
    for name in _ALIASES:
        for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
            del sys.modules[k]
        try:
            mod = importlib.import_module("rrnet_amd." + name)
        except ModuleNotFoundError:
            continue
        sys.modules[name] = mod
    # sub-modules resolve through the aliased package's __path__
