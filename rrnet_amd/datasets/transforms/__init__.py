from .functional import to_heatmap, gaussian_radius, draw_umich_gaussian  # noqa: F401


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


class ToHeatmap:
    """datasets/transforms/transforms.py ToHeatmap -> functional.to_heatmap."""

    def __init__(self, scale_factor=4, cls_num=10):
        self.scale_factor = scale_factor
        self.cls_num = cls_num

    def __call__(self, data):
        return to_heatmap(data, self.scale_factor, self.cls_num)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, data):
        import torch
        img = data[0]
        mean = torch.tensor(self.mean, dtype=img.dtype).view(-1, 1, 1)
        std = torch.tensor(self.std, dtype=img.dtype).view(-1, 1, 1)
        return ((img - mean) / std,) + tuple(data[1:])
