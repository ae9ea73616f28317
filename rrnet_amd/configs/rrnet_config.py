"""`Config` for RRNet with the keys and values of the reference's configs/rrnet_config.py:7-91, declared as one
nested table.  The augmentation transforms of the reference's training pipeline (MultiScale, MaskIgnore, FillDuck,
HorizontalFlip, RandomCrop) belong to the out-of-scope data layer; the pipelines here keep the two transforms that
define the tensor contract of the hot path (Normalize, ToHeatmap)."""
from torch.utils.data import DistributedSampler

from rrnet_amd.datasets.transforms import Compose, Normalize, ToHeatmap
from rrnet_amd.utils.attrdict import AttrDict

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
LOG_PREFIX = 'TwoStageNet'
STRIDE = 4


def _tree(d):
    """dict of dicts -> AttrDict of AttrDicts (attribute access like easydict, which the reference uses)."""
    return AttrDict({k: _tree(v) if isinstance(v, dict) else v for k, v in d.items()})


Config = _tree({
    "seed": 219, "dataset": 'drones_det', "data_root": './data/DronesDET', "log_prefix": LOG_PREFIX,
    "use_tensorboard": True, "num_classes": 10,
    "Train": {
        "pretrained": True, "batch_size": 4, "num_workers": 4, "sampler": DistributedSampler,
        # Adam: only lr is used by the operator (momentum / weight_decay are carried but ignored, as in the reference)
        "lr": 2.5e-4, "momentum": 0.9, "weight_decay": 0.0001, "lr_milestones": [60000, 80000], "iter_num": 100000,
        "crop_size": (512, 512), "mean": IMAGENET_MEAN, "std": IMAGENET_STD, "scale_factor": STRIDE, "with_road": True,
        "transforms": Compose([Normalize(IMAGENET_MEAN, IMAGENET_STD), ToHeatmap(scale_factor=STRIDE)]),
        "print_interval": 20, "checkpoint_interval": 5000,
    },
    "Val": {
        "model_path": './log/%s/ckp-89999.pth' % LOG_PREFIX, "is_eval": True, "auto_test": True, "batch_size": 1,
        "num_workers": 4, "sampler": DistributedSampler, "mean": IMAGENET_MEAN, "std": IMAGENET_STD,
        "scales": [1, 1.1, 1.2, 1.3, 1.4, 1.5], "transforms": Compose([Normalize(IMAGENET_MEAN, IMAGENET_STD)]),
        "result_dir": './results/',
    },
    "Model": {"backbone": 'hourglass', "num_stacks": 2,
              "nms_type_for_stage1": 'nms',          # or 'soft_nms'
              "nms_per_class_for_stage1": True},
    "Distributed": {"world_size": 1, "gpu_id": -1, "rank": 0, "ngpus_per_node": 1,
                    "dist_url": 'tcp://127.0.0.1:34564'},
})
