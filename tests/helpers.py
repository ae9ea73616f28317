"""Shared by the parity tests and by tools/gen_goldens.py: deterministic, platform-stable
weights (numpy PCG64, not torch's RNG) for any reference-keyed state_dict, so fixtures need
to store only inputs and expected outputs."""
import zlib

import numpy as np
import torch


def det_fill(shapes, seed=219):
    """shapes: {key: shape tuple} -> {key: float32 tensor}.  He-style conv weights, BN affine near
    (1, 0), running stats (0, 1), hm-style biases small.  Each key draws from its own stream."""
    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros((), dtype=torch.long)
        elif leaf == "running_mean":
            out[key] = torch.zeros(shape)
        elif leaf == "running_var":
            out[key] = torch.ones(shape)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            out[key] = torch.from_numpy((rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32))
        elif leaf == "weight":                       # BN gamma
            out[key] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32))
        else:                                        # BN beta / conv bias
            out[key] = torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32))
    return out


def shapes_of(state_dict):
    return {k: tuple(v.shape) for k, v in state_dict.items()}


def synth_annos(rng, n, img_h, img_w, min_wh=5.0, max_wh=60.0):
    """[n,8] VisDrone-style rows x,y,w,h,score,cls(1..10),trunc,occl."""
    w = np.exp(rng.uniform(np.log(min_wh), np.log(max_wh), n))
    h = np.exp(rng.uniform(np.log(min_wh), np.log(max_wh), n))
    x = rng.uniform(0, img_w - w)
    y = rng.uniform(0, img_h - h)
    cls = rng.integers(1, 11, n)
    a = np.stack([x, y, w, h, np.ones(n), cls, np.zeros(n), np.zeros(n)], 1).astype(np.float32)
    return a


def host_synth_batch(batch_size, height, width, boxes_per_image=100, seed=219, rank=0):
    """The synthetic batch recipe with the targets built by the ORACLE's host pipeline (oracle/targets.py), all on
    the CPU: what the parity tests feed to the oracle and (moved to the device) to the HIP path."""
    from oracle.targets import host_batch
    from rrnet_amd.datasets.synthetic import synth_frames
    return host_batch(*synth_frames(batch_size, height, width, boxes_per_image, seed, rank))


# ------------------------------------------------------------------------------------------------------------------
# Analytic pins for RoIAlign (torchvision.ops.roi_align is third-party and absent: models/rrnet.py:51).  Bilinear
# interpolation reproduces a map f(x, y) = a*x + b*y + c exactly, so the published definition (legacy coordinates,
# adaptive ceil(size / bins) sampling grid, RoI size clamped to >= 1, samples outside [-1, H] x [-1, W] contribute
# zero but still count, coordinates clamped into the map) can be evaluated in closed form WITHOUT any bilinear
# weights or indices: each bin = sum over its valid samples of f(clip(x), clip(y)) / (gh * gw).
# ------------------------------------------------------------------------------------------------------------------
def linear_map(n, ch, height, width, seed=0):
    """feat[b, k, y, x] = a[b,k]*x + b_[b,k]*y + c[b,k]  (float32 tensor [n,ch,H,W]) and the float64 coefficients."""
    rng = np.random.default_rng(seed)
    a, b, c = (rng.uniform(-1, 1, (n, ch)) for _ in range(3))
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    f = a[:, :, None, None] * xs + b[:, :, None, None] * ys + c[:, :, None, None]
    return torch.from_numpy(f.astype(np.float32)), (a, b, c)


def roi_align_on_linear_map(rois, coef, height, width, out_size, sampling_ratio=-1):
    """Closed-form RoIAlign of the map of `linear_map` -> float64 [K, ch, ph, pw]."""
    a, b, c = coef
    ph, pw = out_size
    rois = np.asarray(rois, np.float64)
    out = np.zeros((rois.shape[0], a.shape[1], ph, pw))
    for r, (bi, x1, y1, x2, y2) in enumerate(rois):
        bi = int(bi)
        rw, rh = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rw / pw))
        for i in range(ph):
            ys = y1 + i * rh / ph + (np.arange(gh) + 0.5) * (rh / ph) / gh
            for j in range(pw):
                xs = x1 + j * rw / pw + (np.arange(gw) + 0.5) * (rw / pw) / gw
                yy, xx = np.meshgrid(ys, xs, indexing="ij")
                ok = (yy >= -1.0) & (yy <= height) & (xx >= -1.0) & (xx <= width)
                yc, xc = np.clip(yy, 0, height - 1), np.clip(xx, 0, width - 1)
                val = a[bi][:, None, None] * xc + b[bi][:, None, None] * yc + c[bi][:, None, None]
                out[r, :, i, j] = (val * ok).sum((1, 2)) / (gh * gw)
    return out


ROI_PIN_CASES = np.array([
    [0, 2.3, 1.2, 9.7, 8.1],          # interior, fractional
    [1, 0.0, 0.0, 23.0, 19.0],        # the whole map
    [0, 4.0, 5.0, 13.0, 11.0],        # integer corners
    [1, 10.2, 11.9, 10.9, 12.3],      # smaller than one pixel: the size clamp (>= 1) decides the bins
    [0, 7.0, 3.0, 7.0, 3.0],          # zero size
    [0, -3.0, -2.5, 4.0, 3.0],        # partly beyond the top-left: samples < -1 are cut, [-1, 0) clamps to 0
    [1, 18.0, 15.0, 30.0, 28.0],      # beyond the bottom-right: samples > W / > H are cut, (W-1, W] clamps
    [0, -9.0, -9.0, -2.0, -2.0],      # entirely outside: all zero
    [1, 5.5, 5.5, 17.0, 6.0],         # flat: height clamp, wide bins
    [0, 0.5, 0.5, 22.5, 18.5],        # many samples per bin (adaptive grid 8 x 6)
], np.float32)
