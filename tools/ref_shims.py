"""Build-container-only helper: makes /root/reference importable on CPU (SURVEY appendix recipe).

Stubs the third-party modules this image lacks (easydict, cv2, tensorboard, torchvision — the
torchvision.ops entry points are bound to the oracle's restatements, so goldens that pass
through roi_align / nms / box_iou are flagged "parity unpinned"), injects the compiled
reference Soft-NMS (oracle/_ref) as ext.nms.nms.cpu_nms, and puts /root/reference first on
sys.path (a pip package called `datasets` would otherwise shadow the reference's).
Never imported by tests/, the product, or anything that runs on the GPU box.
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def install():
    assert os.path.isdir(REF), "reference tree not present: goldens can only be generated in the build container"
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import nms as onms, ops as oops
    from oracle.build import build_ref
    build_ref()

    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    m = types.ModuleType("easydict"); m.EasyDict = EasyDict; sys.modules["easydict"] = m
    sys.modules["cv2"] = types.ModuleType("cv2")
    tb = types.ModuleType("torch.utils.tensorboard"); tb.SummaryWriter = object
    sys.modules["torch.utils.tensorboard"] = tb

    tv = types.ModuleType("torchvision")
    tv_ops = types.ModuleType("torchvision.ops")
    tv_ops.roi_align = lambda feat, rois, size, spatial_scale=1.0, sampling_ratio=-1: \
        oops.roi_align(feat, rois, size, spatial_scale, sampling_ratio)
    tv_ops.nms = oops.tv_nms
    tv_ops.box_iou = oops.box_iou
    tv_tr = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    tv_tr.Compose = Compose
    tv_trf = types.ModuleType("torchvision.transforms.functional")
    tv_tr.functional = tv_trf
    tv_utils = types.ModuleType("torchvision.utils"); tv_utils.make_grid = lambda *a, **k: None
    tv.ops, tv.transforms, tv.utils = tv_ops, tv_tr, tv_utils
    sys.modules.update({"torchvision": tv, "torchvision.ops": tv_ops, "torchvision.transforms": tv_tr,
                        "torchvision.transforms.functional": tv_trf, "torchvision.utils": tv_utils})

    for k in [k for k in sys.modules if k == "datasets" or k.startswith("datasets.")]:
        del sys.modules[k]
    if REF in sys.path:
        sys.path.remove(REF)
    sys.path.insert(0, REF)

    ref_nms = onms.load_reference_cpu_nms()
    assert ref_nms is not None
    cm = types.ModuleType("ext.nms.nms.cpu_nms")
    cm.cpu_soft_nms = ref_nms.cpu_soft_nms
    cm.cpu_nms = None
    gm = types.ModuleType("ext.nms.nms.gpu_nms"); gm.gpu_nms = None
    sys.modules["ext.nms.nms.cpu_nms"] = cm
    sys.modules["ext.nms.nms.gpu_nms"] = gm
    return ref_nms
