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
