"""CPU: the oracle of the split-operand arithmetic (oracle/split.py) has the properties its header states — and the numbers
DESIGN 12 quotes for the kernels follow from them."""
import numpy as np
import torch

from oracle import split as osp


def test_scale_places_the_maximum_in_fp16_range():
    rng = np.random.default_rng(0)
    for gain in (1e-30, 1e-6, 1.0, 37.5, 1e4, 1e30):
        t = (rng.standard_normal(1000) * gain).astype(np.float32)
        s = osp.scale_of(osp.absmax_bits(t))
        m = np.abs(t).max() * s
        assert 2.0 ** 14 <= m < 2.0 ** 15, (gain, m)
        assert np.log2(float(s)) == np.round(np.log2(float(s)))            # a power of two: scaling and unscaling are exact
    # maxima below 2^-112 (zero, denormals): the exponent is clamped — a finite power of two, never the Inf / NaN encodings
    for tiny in (0.0, 1e-45, 1e-39, 2.0 ** -120, 2.0 ** -113):
        s = osp.scale_of(osp.absmax_bits(np.full(4, tiny, np.float32)))
        assert np.isfinite(s) and s == 2.0 ** 126, (tiny, s)
    assert osp.scale_of(osp.absmax_bits(np.full(4, 2.0 ** -112, np.float32))) == 2.0 ** 126
    assert osp.scale_of(osp.absmax_bits(np.full(4, 2.0 ** -111, np.float32))) == 2.0 ** 125


def test_split_reconstructs_to_22_bits():
    rng = np.random.default_rng(1)
    v = (rng.standard_normal(100000) * np.exp(rng.uniform(-6, 0, 100000))).astype(np.float32)
    s = osp.scale_of(osp.absmax_bits(v))
    hi, lo = osp.split(v, s)
    assert np.isfinite(hi.astype(np.float32)).all() and np.abs(hi.astype(np.float32)).max() < 65504
    rec = (hi.astype(np.float64) + lo.astype(np.float64)) / float(s)
    big = np.abs(v) > np.abs(v).max() * 2.0 ** -15                      # lo is a normal fp16 number there
    assert np.abs(rec[big] - v[big]).max() <= 2.0 ** -22 * np.abs(v[big]).max()
    rel = np.abs(rec[big] - v[big]) / np.abs(v[big])
    assert rel.max() <= 2.0 ** -21 and np.median(rel) <= 2.0 ** -23
    # below: absolute error bounded by fp16's subnormal spacing at the scaled maximum — 2^-25 / 2^14 of the maximum
    assert np.abs(rec - v).max() <= 2.0 ** -22 * np.abs(v).max()


def test_oracle_convolution_is_fp32_class():
    """Against an fp64 convolution of the unsplit operands: error ~1e-7 of the output's rms — what remains is the dropped lo*lo
    term and the two 2^-23 representation errors; bf16 operands would give 2e-3 (tests/test_conv_bf16_gpu.py)."""
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(1, 64, 20, 20, generator=g) * torch.rand(1, 64, 1, 1, generator=g) * 3).numpy()
    w = (torch.randn(32, 64, 3, 3, generator=g) * 0.05).numpy()
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, 1, (1, 1)).numpy()
    y = osp.conv2d(x, w, 1, (1, 1))
    rms = np.sqrt((ref ** 2).mean())
    err = np.sqrt(((y - ref) ** 2).mean()) / rms
    assert err <= 2.5e-7, err
    # scale invariance: the same relative error for operands 1e-6 / 1e4 times as large
    y2 = osp.conv2d(x * 1e-6, w * 1e4, 1, (1, 1))
    err2 = np.sqrt(((y2 - ref * 1e-2) ** 2).mean()) / (rms * 1e-2)
    assert abs(err2 - err) <= 0.5 * err + 1e-8, (err, err2)
    assert np.all(osp.conv2d(np.zeros_like(x), w, 1, (1, 1)) == 0)
