"""Target generation contract: the reference's
datasets/transforms/functional.py gaussian_radius :177-198, gaussian2d :201-209,
draw_umich_gaussian :212-227, to_heatmap :230-262, restated (host side, torch-CPU/numpy fp32).
Quirks kept on purpose: the CornerNet radius formula divides by 2 instead of 2a; `ind` uses the
hard-coded `w // 4`, not scale_factor (:257); gaussian sigma = diameter / 6."""
import numpy as np
import torch


def gaussian_radius(det_size, min_overlap=0.7):
    height, width = det_size
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + (b1 ** 2 - 4 * c1).sqrt()) / 2.
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + (b2 ** 2 - 16 * c2).sqrt()) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + (b3 ** 2 - 4 * a3 * c3).sqrt()) / 2
    return torch.cat((r1, r2, r3), dim=1).min(dim=1)[0]


def _gaussian2d(diameter, sigma):
    m = ((diameter - 1.) / 2.).numpy()
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    s = sigma.numpy()
    h = np.exp(-(x * x + y * y) / (2 * s * s))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return torch.from_numpy(h).float()


def draw_umich_gaussian(heatmap, center, radius, k=1):
    diameter = 2 * radius + 1
    g = _gaussian2d(diameter, diameter / 6)
    x, y = center[0], center[1]
    height, width = heatmap.size()[0:2]
    left, right = torch.min(x, radius), torch.min(width - x, radius + 1)
    top, bottom = torch.min(y, radius), torch.min(height - y, radius + 1)
    hm = heatmap[int(y - top):int(y + bottom), int(x - left):int(x + right)]
    gs = g[int(radius - top):int(radius + bottom), int(radius - left):int(radius + right)]
    if min(gs.shape) > 0 and min(hm.shape) > 0:
        torch.max(hm, gs * k, out=hm)
    return heatmap


def to_heatmap(data, scale_factor=4, cls_num=10):
    """(img [3,H,W], annos [n,>=6] xywh,score,cls(1-based)) ->
    (img, annos, hm [cls,H/s,W/s], wh [n,2], ind [n,1], offset [n,2], reg_mask [n,1])."""
    img = data[0]
    annos = data[1].clone()
    h, w = img.size(1), img.size(2)
    hm = torch.zeros(cls_num, h // scale_factor, w // scale_factor)
    annos[:, 2] += annos[:, 0]
    annos[:, 3] += annos[:, 1]
    annos[:, :4] = annos[:, :4] / scale_factor
    cls_idx = annos[:, 5] - 1
    bh, bw = annos[:, 3:4] - annos[:, 1:2], annos[:, 2:3] - annos[:, 0:1]
    wh = torch.cat([bw, bh], dim=1)
    ct = torch.cat(((annos[:, 0:1] + annos[:, 2:3]) / 2., (annos[:, 1:2] + annos[:, 3:4]) / 2.), dim=1)
    ct_int = ct.floor()
    offset = ct - ct_int
    reg_mask = ((bh > 0) * (bw > 0))
    ind = ct_int[:, 1:2] * (w // 4) + ct_int[:, 0:1]
    radius = gaussian_radius((bh.ceil(), bw.ceil())).floor().clamp(min=0)
    for k, cls in enumerate(cls_idx):
        draw_umich_gaussian(hm[cls.long().item()], ct_int[k], radius[k])
    return data[0], data[1], hm, wh, ind, offset, reg_mask


def flip_img(data):
    """datasets/transforms/functional.py:13-19: horizontal flip of a [C,H,W] image tensor."""
    return data.flip(dims=(2,))


def flip_annos(data, w):
    """datasets/transforms/functional.py:22-29: x -> w - x - width for xywh rows (in place, like the reference)."""
    data[:, 0] = w - data[:, 0] - data[:, 2]
    return data
