"""Pins the oracle of the deformable PS-RoI pooling (oracle/psroi.py) by properties — the reference op is CUDA-only,
no reference output exists (parity unpinned, see the oracle's header):
  * a constant map pools to that constant, count = sample_per_part^2 for RoIs inside the map;
  * a map linear in (x, y) pools to the mean of the (clamped) sample positions' values: bilinear interpolation is exact
    on linear functions, so the expected value has a closed form from the kernel's own sampling rule
    (dcn_v2_psroi_pooling_cuda.cu:115-140);
  * the data gradient is the transpose of the forward (linear in x): <dout, fwd(x)> == <bwd(dout), x>;
  * the trans gradient matches central differences of the forward where no sample is clamped."""
import numpy as np

from oracle import psroi as op


def _case(rng, no_trans=False, classes=1, out_dim=4, P=3, H=14, W=17):
    x = rng.normal(0, 1, (2, out_dim, H, W)).astype(np.float32)
    rois = np.array([[0, 2.2, 3.1, 9.7, 10.2], [1, 4.0, 1.0, 12.6, 8.4], [1, 0.3, 0.2, 5.5, 4.9]], np.float32)
    trans = rng.normal(0, 1, (3, 2 * classes, P, P)).astype(np.float32)
    return x, rois, trans, dict(no_trans=no_trans, scale=1.0, out_dim=out_dim, gs=1, P=P, part=P, spp=2, trans_std=0.1)


def test_constant_and_linear_maps():
    rng = np.random.default_rng(0)
    x, rois, trans, kw = _case(rng, no_trans=True)
    out, cnt = op.psroi_forward(np.full_like(x, 2.5), rois, trans, **kw)
    assert np.all(cnt == 4) and np.allclose(out, 2.5)
    H, W = x.shape[2:]
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    lin = np.broadcast_to(0.5 * xx - 0.25 * yy + 1.0, x.shape).astype(np.float32).copy()
    out, cnt = op.psroi_forward(lin, rois, trans, **kw)
    for n in range(rois.shape[0]):
        for ph in range(3):
            for pw in range(3):
                _, _, _, _, _, ws, hs, sw, sh, _, _ = op._bins(rois, trans, n, 0, ph, pw, True, 1.0, 4, 1, 3, 3, 2, 0.1)
                pts = list(op._samples(ws, hs, sw, sh, 2, H, W))
                exp = np.mean([0.5 * w - 0.25 * h + 1.0 for w, h in pts])
                assert abs(out[n, 0, ph, pw] - exp) < 1e-5


def test_backward_is_transpose_and_trans_gradient_matches_differences():
    rng = np.random.default_rng(1)
    x, rois, trans, kw = _case(rng, classes=2)
    out, cnt = op.psroi_forward(x, rois, trans, **kw)
    dout = rng.normal(0, 1, out.shape).astype(np.float32)
    dx, dtrans = op.psroi_backward(dout, x, rois, trans, cnt, **kw)
    assert abs(float((dout.astype(np.float64) * out).sum()) - float((dx.astype(np.float64) * x).sum())) < 1e-3
    eps = 1e-2
    for idx in [(0, 0, 1, 1), (1, 3, 0, 2), (2, 1, 2, 0)]:
        tp, tm = trans.copy(), trans.copy()
        tp[idx] += eps
        tm[idx] -= eps
        fp = (dout * op.psroi_forward(x, rois, tp, **kw)[0]).sum()
        fm = (dout * op.psroi_forward(x, rois, tm, **kw)[0]).sum()
        num = (fp - fm) / (2 * eps)
        assert abs(num - dtrans[idx]) < 5e-2 * max(1.0, abs(num)), (idx, num, dtrans[idx])
