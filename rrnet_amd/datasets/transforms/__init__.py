"""Transform classes of the reference's datasets/transforms/transforms.py that belong to the target contract."""
import torch

from .functional import flip_annos, flip_img  # noqa: F401


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


class ToHeatmap:
    """datasets/transforms/transforms.py ToHeatmap: (img [3,H,W], annos [n,>=6]) -> (img, annos, hm, wh, ind, offset,
    reg_mask).  The targets are built by rr_ctnet_targets on the current device and returned as CPU tensors in the
    reference's shapes (a per-sample transform runs in the loader, before collation)."""

    def __init__(self, scale_factor=4, cls_num=10):
        self.scale_factor = scale_factor
        self.cls_num = cls_num

    def __call__(self, data):
        from rrnet_amd.datasets.synthetic import collate_ctnet_device
        img, annos = data[0], data[1]
        _, hm, wh, ind, off, mask = collate_ctnet_device([annos.float()], img.size(1), img.size(2), self.scale_factor,
                                                         self.cls_num)
        return (img, annos, hm[0].cpu().contiguous(), wh[0].cpu(), ind[0].cpu(), off[0].cpu(), mask[0].cpu())


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, data):
        img = data[0]
        mean = torch.tensor(self.mean, dtype=img.dtype).view(-1, 1, 1)
        std = torch.tensor(self.std, dtype=img.dtype).view(-1, 1, 1)
        return ((img - mean) / std,) + tuple(data[1:])
