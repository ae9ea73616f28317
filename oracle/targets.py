"""ORACLE — test infrastructure, not product code (only tests/, smoke() and bench.py's cpu_baseline may import it).

Host restatement of the reference's CenterNet training targets:
  datasets/transforms/functional.py  gaussian_radius :177-198, gaussian2d :201-209, draw_umich_gaussian :212-227,
                                     to_heatmap :230-262
  datasets/drones_det.py             collate_fn_ctnet :70-94
in numpy float32 (every step rounds like the reference's float32 torch / numpy expression).  Pinned bit-exact to
tests/golden/targets.npz, which tools/gen_goldens.py produced by calling the reference's own to_heatmap.

Quirks kept on purpose: the CornerNet radius formula divides by 2 instead of 2a; `ind` uses the hard-coded
`w // 4`, not scale_factor (:257); sigma = diameter / 6; entries below eps * max of the splat are dropped.
The product path is the device kernel rr_ctnet_targets (rrnet_amd/csrc/targets.hip); this file is its checker.
"""
import numpy as np
import torch

F = np.float32


def gaussian2d(diameter, sigma):
    """functional.py:201-209 for a square (diameter x diameter) kernel, float32 like the reference's tensor inputs."""
    m = F((F(diameter) - F(1)) / F(2))
    ax = np.arange(-m, m + 1, dtype=F)
    yy, xx = ax[:, None], ax[None, :]
    g = np.exp(-(xx * xx + yy * yy) / (F(2) * F(sigma) * F(sigma))).astype(F)
    g[g < np.finfo(F).eps * g.max()] = 0
    return g


def draw_umich_gaussian(plane, cx, cy, radius):
    """functional.py:212-227: element-wise max of `plane` [Hf,Wf] with the gaussian clipped at the borders (in place)."""
    r = int(radius)
    g = gaussian2d(F(2 * r + 1), F(2 * r + 1) / F(6))
    hf, wf = plane.shape
    x, y = int(cx), int(cy)
    left, right = min(x, r), min(wf - x, r + 1)
    top, bottom = min(y, r), min(hf - y, r + 1)
    dst = plane[y - top:y + bottom, x - left:x + right]
    src = g[r - top:r + bottom, r - left:r + right]
    if min(src.shape) > 0 and min(dst.shape) > 0:
        np.maximum(dst, src, out=dst)
    return plane


def ctnet_targets(annos, img_h, img_w, scale_factor=4, cls_num=10):
    """to_heatmap (functional.py:230-262) on one image's annotations.
    annos float32 [n,>=6] = x,y,w,h,score,cls(1-based) -> hm [cls,Hf,Wf], wh [n,2], ind [n,1], offset [n,2],
    reg_mask [n,1] (float32 numpy)."""
    a = np.array(annos, F, copy=True)
    s = F(scale_factor)
    x1, y1 = a[:, 0] / s, a[:, 1] / s
    x2, y2 = (a[:, 2] + a[:, 0]) / s, (a[:, 3] + a[:, 1]) / s
    bw, bh = x2 - x1, y2 - y1
    cx, cy = (x1 + x2) / F(2), (y1 + y2) / F(2)
    cxi, cyi = np.floor(cx), np.floor(cy)
    hm = np.zeros((cls_num, img_h // scale_factor, img_w // scale_factor), F)
    with np.errstate(invalid="ignore"):
        rad = gaussian_radius(np.ceil(bh), np.ceil(bw))
    rad = np.where(np.isnan(rad), F(0), np.maximum(np.floor(rad), F(0)))
    for k in range(a.shape[0]):
        draw_umich_gaussian(hm[int(a[k, 5] - 1)], cxi[k], cyi[k], rad[k])
    wh = np.stack([bw, bh], 1)
    off = np.stack([cx - cxi, cy - cyi], 1)
    ind = (cyi * F(img_w // 4) + cxi)[:, None]
    mask = ((bh > 0) & (bw > 0)).astype(F)[:, None]
    return hm, wh.astype(F), ind.astype(F), off.astype(F), mask


def gaussian_radius(height, width, ov=0.7):
    """functional.py:177-198 in its operation order (float32 tensor x python float = float32): three quadratics
    a r^2 + b r + c, each "solved" as (b + sqrt(b^2 - 4ac)) / 2 (sic: not / 2a), minimum of the three."""
    h, w = height.astype(F), width.astype(F)
    b1 = h + w
    c1 = w * h * F(1 - ov) / F(1 + ov)
    r1 = (b1 + np.sqrt(b1 * b1 - F(4) * c1)) / F(2)
    b2 = F(2) * (h + w)
    c2 = F(1 - ov) * w * h
    r2 = (b2 + np.sqrt(b2 * b2 - F(16) * c2)) / F(2)
    a3 = F(4 * ov)
    b3 = F(-2 * ov) * (h + w)
    c3 = F(ov - 1) * w * h
    r3 = (b3 + np.sqrt(b3 * b3 - F(4) * a3 * c3)) / F(2)
    return np.minimum(np.minimum(r1, r2), r3).astype(F)


def to_heatmap(data, scale_factor=4, cls_num=10):
    """Tuple-in / tuple-out form of functional.py:230-262: (img [3,H,W], annos [n,>=6]) ->
    (img, annos, hm, wh, ind, offset, reg_mask) as torch CPU tensors; the inputs are not mutated."""
    img, annos = data[0], data[1]
    hm, wh, ind, off, mask = ctnet_targets(annos.numpy(), img.size(1), img.size(2), scale_factor, cls_num)
    return (img, annos, torch.from_numpy(hm), torch.from_numpy(wh), torch.from_numpy(ind), torch.from_numpy(off),
            torch.from_numpy(mask))


def collate_ctnet(samples):
    """collate_fn_ctnet (drones_det.py:70-94): zero-pad the per-image rows to the longest annotation list.
    samples: [(img, annos, hm, wh, ind, offset, reg_mask, name)] -> batch tuple of the same order."""
    bs = len(samples)
    m = max(int(s[1].shape[0]) for s in samples)
    annos, whs, offs = torch.zeros(bs, m, 8), torch.zeros(bs, m, 2), torch.zeros(bs, m, 2)
    inds, masks = torch.zeros(bs, m, 1), torch.zeros(bs, m, 1)
    for i, (_, a, _, wh, ind, off, mask, _) in enumerate(samples):
        n = int(a.shape[0])
        annos[i, :n], whs[i, :n], inds[i, :n], offs[i, :n], masks[i, :n] = a[:, :8], wh, ind, off, mask
    imgs = torch.stack([s[0] for s in samples])
    hms = torch.stack([s[2] for s in samples])
    return imgs, annos, hms, whs, inds, offs, masks, [s[7] for s in samples]


def host_batch(imgs, annos_list, scale_factor=4, cls_num=10):
    """Reference host pipeline for a list of per-image annotation tensors: to_heatmap per image + collate_fn_ctnet."""
    samples = []
    for i, a in enumerate(annos_list):
        _, aa, hm, wh, ind, off, mask = to_heatmap((imgs[i], a), scale_factor, cls_num)
        samples.append((imgs[i], aa, hm, wh, ind, off, mask, "synthetic_%06d" % i))
    return collate_ctnet(samples)
