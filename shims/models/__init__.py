# Zero-edit drop-in: `PYTHONPATH=<repo>/shims python scripts/RRNet/train.py` resolves the reference's top-level
# package names to rrnet_amd (one module object under both names, see rrnet_amd.install_aliases).
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(1, _root)
import rrnet_amd  # noqa: E402

rrnet_amd.install_aliases()
sys.modules[__name__] = importlib.import_module("rrnet_amd." + __name__)
