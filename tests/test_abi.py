"""CPU-side boundary checks (no GPU, no compute calls): the C-ABI library builds for gfx950,
loads, and exports every symbol include/rrnet_hip.h declares; the Python bindings take their
prototypes from that header; the product package never imports the oracle."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from rrnet_amd.csrc import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(built_lib):
    from rrnet_amd import _C
    sigs = _C.header_signatures()
    assert "rr_soft_nms_segments" in sigs and "rr_conv_fprop" in sigs and "rr_last_error" in sigs
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    missing = [n for n in sigs if n not in exported]
    assert not missing, "declared in include/rrnet_hip.h but not exported: %s" % missing
    L = _C.lib()
    for n in sigs:
        assert getattr(L, n) is not None
    assert L.rr_abi_version() == 1


def test_bindings_only_call_declared_entry_points():
    from rrnet_amd import _C
    sigs = _C.header_signatures()
    used = set()
    for dp, _, files in os.walk(os.path.join(ROOT, "rrnet_amd")):
        for f in files:
            if f.endswith(".py"):
                used |= set(re.findall(r'_C\.fn\("(\w+)"', open(os.path.join(dp, f)).read()))
    assert used, "no bindings found"
    assert not (used - set(sigs)), "bindings call undeclared entry points: %s" % (used - set(sigs))


def test_product_never_touches_the_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "rrnet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "liboracle" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch):
    from rrnet_amd import _C
    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "LIB_PATH", "/nonexistent/librrnet_hip.so")
    with pytest.raises(_C.RRNetHipError):
        _C.lib()


def test_cpu_tensors_are_refused():
    import torch
    from rrnet_amd import _C
    with pytest.raises(_C.RRNetHipError):
        _C.require_cuda(torch.zeros(3))


def test_reference_script_imports_resolve_through_aliases():
    """The import lines of the reference's scripts/RRNet/train.py and eval.py (configs.rrnet_config.Config,
    operators.distributed_wrapper.DistributedWrapper, operators.rrnet_operator.RRNetOperator,
    utils.metrics.metrics.evaluate_results, ext.nms.nms_wrapper.soft_nms, models.rrnet.RRNet) resolve to this
    package after rrnet_amd.install_aliases() — run in a fresh interpreter, no GPU needed for importing."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import rrnet_amd; rrnet_amd.install_aliases()\n"
        "from configs.rrnet_config import Config\n"
        "from operators.distributed_wrapper import DistributedWrapper\n"
        "from operators.rrnet_operator import RRNetOperator\n"
        "from utils.metrics.metrics import evaluate_results, auto_evaluate_results\n"
        "from ext.nms.nms_wrapper import soft_nms, nms\n"
        "from models.rrnet import RRNet\n"
        "from backbones.hourglass import hourglass_net\n"
        "from detectors.fasterrcnn_detector import FasterRCNNDetector\n"
        "from modules.loss.focalloss import FocalLossHM\n"
        "w = DistributedWrapper(Config, RRNetOperator)\n"
        "import inspect, os\n"
        "assert os.sep + 'rrnet_amd' + os.sep in inspect.getfile(RRNetOperator) and Config.Model.backbone == 'hourglass'\n"
        "assert Config.Val.scales == [1, 1.1, 1.2, 1.3, 1.4, 1.5] and Config.Train.lr == 2.5e-4\n"
        "print('ok')\n" % ROOT)
    out = subprocess.check_output(["python", "-c", code], text=True, cwd=ROOT, timeout=300)
    assert out.strip().endswith("ok")


def test_reference_scripts_run_their_imports_through_shims():
    """Zero-edit drop-in: with PYTHONPATH=<repo>/shims the import blocks of the reference's scripts/RRNet/{train,eval,
    auto_eval}.py and scripts/CTNet/*.py resolve to this package — the same module objects as rrnet_amd.* — in a
    fresh interpreter started OUTSIDE the repository, exactly as `PYTHONPATH=shims python scripts/RRNet/train.py`
    would (no GPU needed for importing)."""
    code = (
        "from configs.rrnet_config import Config\n"
        "from operators.distributed_wrapper import DistributedWrapper\n"
        "from operators.rrnet_operator import RRNetOperator\n"
        "from utils.metrics.metrics import evaluate_results, auto_evaluate_results\n"
        "from configs.centernet_config import Config as CtConfig\n"
        "from operators.centernet_operator import CenterNetOperator\n"
        "import models.rrnet, rrnet_amd.models.rrnet, rrnet_amd.configs.rrnet_config as c2, datasets, ext.nms.nms_wrapper\n"
        "assert models.rrnet is rrnet_amd.models.rrnet and c2.Config is Config\n"
        "assert datasets.__name__ == 'rrnet_amd.datasets'\n"
        "for m in ('nms', '_topk', '_gather_feat', '_transpose_and_gather_feat', 'transform_bbox', 'forward_stage1'):\n"
        "    assert hasattr(models.rrnet.RRNet, m), m\n"
        "for m in ('criterion', 'training_process', 'evaluation_process', 'transform_bbox', '_ctnet_nms', '_ext_nms', 'save_result'):\n"
        "    assert hasattr(CenterNetOperator, m), m\n"
        "DistributedWrapper(Config, RRNetOperator); DistributedWrapper(CtConfig, CenterNetOperator)\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "shims"))
    out = subprocess.check_output(["python", "-c", code], text=True, cwd="/tmp", env=env, timeout=300)
    assert out.strip().endswith("ok")
