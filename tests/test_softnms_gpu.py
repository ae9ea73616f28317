"""GPU parity: rr_soft_nms_segments (HIP, through the C ABI) vs the oracle and the reference's
golden vectors.  Bit-exact rows, exact N'."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_segments(boxes_list, sigma, Nt, thr, method):
    from rrnet_amd.ext.nms.nms_wrapper import soft_nms_segments
    offs = np.concatenate([[0], np.cumsum([b.shape[0] for b in boxes_list])]).astype(np.int32)
    stride = boxes_list[0].shape[1]
    allb = np.concatenate(boxes_list, 0).astype(np.float32) if offs[-1] > 0 else np.zeros((0, stride), np.float32)
    d = torch.from_numpy(allb).cuda()
    n_out = soft_nms_segments(d, torch.from_numpy(offs).cuda(), max(b.shape[0] for b in boxes_list),
                              sigma, Nt, thr, method)
    d = d.cpu().numpy()
    n_out = n_out.cpu().numpy()
    return [d[offs[i]:offs[i] + n_out[i]] for i in range(len(boxes_list))], n_out


def test_golden_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "softnms.npz"))
    for name in z["names"]:
        name = str(name)
        sigma, Nt, thr, method = z[name + "/params"]
        inp, exp = z[name + "/in"], z[name + "/out"]
        if inp.shape[0] == 0:
            continue
        outs, n_out = _run_segments([inp], sigma, Nt, thr, int(method))
        assert n_out[0] == int(z[name + "/n_out"]), name
        assert np.array_equal(outs[0][:, :5].view(np.uint32), exp[:, :5].view(np.uint32)), name
        if inp.shape[1] > 5:   # class column is never permuted
            assert np.array_equal(outs[0][:, 5], inp[:n_out[0], 5]), name


def test_wrapper_reference_contract():
    from rrnet_amd.ext.nms.nms_wrapper import soft_nms
    anchor = [[10, 9, 20, 19, 0.5], [10, 10, 15, 30, 0.45], [10, 10, 26, 26, 0.7],
              [8, 9, 14, 16, 0.3], [8, 8, 15, 15, 0.1]]
    res = soft_nms(anchor, Nt=0.4, sigma=0.3)          # list input: unmodified rows come back
    assert np.array_equal(res, np.array(anchor))
    a = np.array(anchor, dtype=np.float32)
    res = soft_nms(a, Nt=0.4, sigma=0.3)               # contiguous f32: mutated in place
    assert res.shape == (5, 5)
    assert a[0, 4] == np.float32(0.7) and abs(float(a[4, 4]) - 0.030986) < 1e-6
    assert soft_nms(np.zeros((0, 5), np.float32)).shape == (0, 5)
    # error behaviour of the Cython routine (cdivision=False): sigma == 0 with the gaussian method divides by zero
    b = np.array([[0, 0, 10, 10, 0.9], [0, 0, 10, 10, 0.8]], dtype=np.float32)
    with pytest.raises(ZeroDivisionError):
        soft_nms(b, sigma=0.0, method=2)
    # degenerate (negative-extent) boxes never overlap under the +1 convention: both survive, no error
    c = np.array([[0, 0, -1, -1, 0.9], [0, 0, -1, -1, 0.8]], dtype=np.float32)
    assert soft_nms(c, method=2).shape == (2, 5)


@pytest.mark.parametrize("method,Nt,thr", [(2, 0.7, 0.1), (1, 0.3, 0.001), (0, 0.3, 0.001)])
def test_batched_segments_vs_oracle(method, Nt, thr):
    from oracle import nms as onms
    rng = np.random.default_rng(100 + method)
    segs = []
    for n in [0, 1, 3, 64, 65, 190, 193, 700, 1500, 2600, 300]:
        span = 300.0 if n < 1000 else 700.0
        xy = rng.uniform(0, span, (n, 2))
        wh = rng.uniform(8, 120, (n, 2))
        s = rng.uniform(0.01, 1, (n, 1))
        if n == 700:
            s = np.round(s * 8) / 8 + 0.01      # exact ties
        c = rng.integers(0, 10, (n, 1)).astype(np.float64)
        segs.append(np.concatenate([xy, xy + wh, s, c], 1).astype(np.float32))
    outs, n_out = _run_segments(segs, 0.5, Nt, thr, method)
    for b, o, k in zip(segs, outs, n_out):
        w = b.copy()
        keep = onms.cpu_soft_nms(w, 0.5, Nt, thr, method)
        assert len(keep) == k
        assert np.array_equal(o[:, :5].view(np.uint32), w[:k, :5].view(np.uint32))


def test_large_segment_global_workspace_path():
    from oracle import nms as onms
    rng = np.random.default_rng(5)
    n = 6500                                         # > RR_SOFT_NMS_LDS_MAX
    xy = rng.uniform(0, 1500, (n, 2))
    wh = rng.uniform(8, 120, (n, 2))
    b = np.concatenate([xy, xy + wh, rng.uniform(0.01, 1, (n, 1))], 1).astype(np.float32)
    outs, n_out = _run_segments([b], 0.5, 0.7, 0.1, 2)
    w = b.copy()
    keep = onms.cpu_soft_nms(w, 0.5, 0.7, 0.1, 2)
    assert len(keep) == n_out[0]
    assert np.array_equal(outs[0].view(np.uint32), w[:len(keep)].view(np.uint32))


def test_idempotent_round_trip_property():
    """Size-independent property: re-running hard-mode NMS on its own output keeps everything."""
    rng = np.random.default_rng(9)
    n = 1500
    xy = rng.uniform(0, 900, (n, 2))
    wh = rng.uniform(8, 120, (n, 2))
    b = np.concatenate([xy, xy + wh, rng.uniform(0.2, 1, (n, 1))], 1).astype(np.float32)
    outs, n_out = _run_segments([b], 0.5, 0.5, 0.001, 0)
    outs2, n_out2 = _run_segments([outs[0].copy()], 0.5, 0.5, 0.001, 0)
    assert n_out2[0] == n_out[0]
    assert np.array_equal(outs2[0], outs[0])
    assert np.all(np.diff(outs[0][:, 4]) <= 0)        # selection order = non-increasing score


def test_legacy_hard_nms_family_vs_oracle():
    """rr_nms_sorted (+ the `_nms` host entry) against the oracle's restatement of gpu_nms / cpu_nms: kept index
    lists equal, both threshold conventions, sizes around the 64-box mask words."""
    import ctypes
    from oracle import nms as onms
    from rrnet_amd import _C
    from rrnet_amd.ext.nms import nms_wrapper as W
    a = np.array([[10, 9, 20, 19, 0.5], [10, 10, 15, 30, 0.45], [10, 10, 26, 26, 0.7], [8, 9, 14, 16, 0.3],
                  [8, 8, 15, 15, 0.1]], np.float32)
    assert [int(i) for i in W.gpu_nms(a, 0.3)] == [2, 3]
    np.testing.assert_array_equal(W.nms(a, 0.3), a[[2, 3]])
    b = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8]], np.float32)
    assert [int(i) for i in W.gpu_nms(b, 0.5)] == [0, 1] and [int(i) for i in W.cpu_nms(b, 0.5)] == [0]
    rng = np.random.default_rng(8)
    for n in [1, 2, 63, 64, 65, 128, 129, 1000, 5000]:
        xy = rng.uniform(0, 300, (n, 2)).astype(np.float32)
        wh = rng.uniform(5, 80, (n, 2)).astype(np.float32)
        d = np.concatenate([xy, xy + wh, rng.uniform(0.01, 1, (n, 1)).astype(np.float32)], 1)
        for thr in (0.3, 0.5, 0.7):
            assert [int(i) for i in W.gpu_nms(d, thr)] == onms.legacy_nms(d, thr)
            assert [int(i) for i in W.cpu_nms(d, thr)] == onms.legacy_nms(d, thr, inclusive=True)
        # the reference's C entry point: host pointers, pre-sorted boxes
        order = d[:, 4].argsort()[::-1]
        sd = np.ascontiguousarray(d[order])
        keep = np.zeros(n, np.int32)
        num = ctypes.c_int(0)
        f = _C.fn("_nms")
        f(keep.ctypes.data_as(ctypes.c_void_p), ctypes.cast(ctypes.pointer(num), ctypes.c_void_p),
          sd.ctypes.data_as(ctypes.c_void_p), n, 5, 0.5, torch.cuda.current_device())
        assert list(order[keep[:num.value]]) == onms.legacy_nms(d, 0.5)
    assert W.nms(np.zeros((0, 5), np.float32), 0.5) == []


def _hardnms_goldens(golden_dir):
    z = np.load(os.path.join(golden_dir, "hardnms.npz"))
    for name in z["names"]:
        name = str(name)
        yield name, z[name + "/dets"], float(z[name + "/thresh"]), [int(i) for i in z[name + "/keep"]]


def test_hard_nms_family_vs_reference_goldens(golden_dir):
    """rr_nms_sorted through gpu_nms / nms, and the reference's C entry `_nms` (host pointers, pre-sorted boxes,
    nms_kernel.cu:91-144), against the kept-index lists the reference's own py_cpu_nms.py returned
    (tools/gen_golden_hardnms.py): 32 cases incl. N = 0 / 1 / 1500, ties, identical boxes, IoU == thresh."""
    import ctypes
    from rrnet_amd import _C
    from rrnet_amd.ext.nms import nms_wrapper as W
    ncases = 0
    for name, d, thr, keep in _hardnms_goldens(golden_dir):
        ncases += 1
        if d.shape[0] == 0:
            assert W.nms(d, thr) == [] and keep == []
            continue
        assert [int(i) for i in W.gpu_nms(d, thr)] == keep, name
        np.testing.assert_array_equal(W.nms(d, thr), d[keep])
        order = d[:, 4].argsort()[::-1]
        sd = np.ascontiguousarray(d[order])
        out = np.zeros(d.shape[0], np.int32)
        num = ctypes.c_int(0)
        _C.fn("_nms")(out.ctypes.data_as(ctypes.c_void_p), ctypes.cast(ctypes.pointer(num), ctypes.c_void_p),
                      sd.ctypes.data_as(ctypes.c_void_p), d.shape[0], 5, thr, torch.cuda.current_device())
        assert [int(i) for i in order[out[:num.value]]] == keep, name
    assert ncases >= 30


def test_stage1_hard_nms_kernel_pinned_by_reference_on_integer_boxes(golden_dir):
    """The stage-1 default NMS (torchvision convention: plain IoU, `>`; models/rrnet.py:69) has no reference output
    here.  On integer coordinates IoU(+1) of (x1,y1,x2,y2) IS the plain IoU of (x1,y1,x2+1,y2+1), exactly, so the
    reference's py_cpu_nms goldens pin the greedy algorithm of rr_hard_nms_segments (visiting order, threshold
    convention, suppression by kept boxes only)."""
    from rrnet_amd import ops
    for name, d, thr, keep in _hardnms_goldens(golden_dir):
        if not name.startswith(("integer_grid", "exact_half")):
            continue
        order = np.argsort(-d[:, 4], kind="stable")
        if len(np.unique(d[:, 4])) != d.shape[0]:
            continue                                        # ties: visiting orders may legitimately differ
        b = d[order].copy()
        b[:, 2:4] += 1.0
        rows = torch.from_numpy(np.concatenate([b, np.zeros((b.shape[0], 1), np.float32)], 1)).cuda()
        seg = torch.tensor([0, b.shape[0]], dtype=torch.int32, device="cuda")
        n_out = ops.hard_nms_segments(rows, seg, b.shape[0], thr, None)
        got = rows[:int(n_out[0])].cpu().numpy()
        np.testing.assert_array_equal(got[:, :5], b[[list(order).index(k) for k in keep]][:, :5])


def test_ext_nms_batch_and_auto_evaluate(tmp_path):
    """utils/metrics: the batched per-file / per-class Soft-NMS equals the oracle's ext_nms file by file, bit for
    bit, and auto_evaluate_results == evaluating those results in memory."""
    import contextlib
    import io
    from oracle import nms as onms
    from rrnet_amd.utils.metrics import metrics as M
    rng = np.random.default_rng(12)
    preds, targets = [], []
    for n in [80, 0, 1, 300, 45]:
        xy = rng.uniform(0, 200, (n, 2)); wh = rng.uniform(10, 60, (n, 2))
        p = np.concatenate([xy, wh, rng.uniform(0.02, 1, (n, 1)), rng.integers(1, 11, (n, 1))], 1).astype(np.float32)
        preds.append(p[np.argsort(-p[:, 4], kind='stable')])
        g = max(n // 4, 2)
        txy = rng.uniform(0, 200, (g, 2)); twh = rng.uniform(10, 60, (g, 2))
        targets.append(np.concatenate([txy, twh, np.ones((g, 1)), rng.integers(0, 11, (g, 1)), np.zeros((g, 2))], 1)
                       .astype(np.float32))
    got = M.ext_nms_batch(preds, 0.1)
    for p, gk in zip(preds, got):
        ref = onms.ext_nms(p) if p.shape[0] else p.reshape(0, 6)
        assert gk.shape == ref.shape
        np.testing.assert_array_equal(gk.view(np.uint32), ref.view(np.uint32))
    pd_dir, gt_dir = tmp_path / "pred", tmp_path / "gt"
    pd_dir.mkdir(); gt_dir.mkdir()
    for i, (p, t) in enumerate(zip(preds, targets)):
        if p.shape[0] == 0:
            continue
        np.savetxt(pd_dir / ("f%d.txt" % i), np.concatenate([p, -np.ones((p.shape[0], 2))], 1), delimiter=',', fmt='%.6f')
        np.savetxt(gt_dir / ("f%d.txt" % i), t, delimiter=',', fmt='%d')
    with contextlib.redirect_stdout(io.StringIO()):
        ap, rc = M.auto_evaluate_results(str(pd_dir), str(gt_dir), 0.05, 0.1)
    flags, confs, tc, ic = M._fresh(11, 10)
    for name in M._names(str(pd_dir)):
        p = M._read(os.path.join(str(pd_dir), name + ".txt"))
        p = p[p[:, 4] > 0.05].astype(np.float32)
        p = p[np.argsort(-p[:, 4], kind='stable')]
        k = onms.ext_nms(p[:, :6])
        k = torch.from_numpy(M._snap(k.astype(np.float64))).float()
        k = k[torch.sort(k[:, 4], descending=True)[1]][:500]
        t = torch.from_numpy(M._read(os.path.join(str(gt_dir), name + ".txt"))).float()[:500]
        flags, confs, tc, ic = M.get_tp(k, t, flags, confs, tc, ic, M.THRESHOLDS, 11)
    ap2, rc2 = M.calculate_ap_rc(flags, confs, tc, ic)
    np.testing.assert_allclose(ap.numpy(), ap2.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(float(rc), float(rc2), rtol=1e-6, atol=1e-7)
