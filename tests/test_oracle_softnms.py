"""Pins the oracle's Soft-NMS restatement (oracle/soft_nms.c) to the reference:
 - the README known-answer vector, /root/reference/ext/nms/nms_wrapper.py:36-50;
 - golden vectors produced by the reference's own compiled cpu_soft_nms (tools/gen_golden_softnms.py);
 - oracle/_ref itself when present (build container).  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import nms


def _cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "softnms.npz"))
    for name in z["names"]:
        name = str(name)
        sigma, Nt, thr, method = z[name + "/params"]
        yield name, z[name + "/in"], z[name + "/out"], int(z[name + "/n_out"]), sigma, Nt, thr, int(method)


def test_readme_known_answer():
    anchor = [[10, 9, 20, 19, 0.5], [10, 10, 15, 30, 0.45], [10, 10, 26, 26, 0.7],
              [8, 9, 14, 16, 0.3], [8, 8, 15, 15, 0.1]]
    a = np.array(anchor, dtype=np.float32)
    keep = nms.cpu_soft_nms(a, sigma=0.3, Nt=0.4, threshold=0.001, method=1)
    assert keep == [0, 1, 2, 3, 4]          # documented at nms_wrapper.py:47-50
    # wrapper semantics: a python list is not a contiguous f32 array -> unmodified rows come back
    res = nms.soft_nms(np.array(anchor), Nt=0.4, sigma=0.3)
    assert res.shape == (5, 5) and np.array_equal(res, np.array(anchor))


def test_oracle_matches_reference_goldens(golden_dir):
    n = 0
    for name, inp, exp, n_out, sigma, Nt, thr, method in _cases(golden_dir):
        work = np.ascontiguousarray(inp.copy())
        keep = nms.cpu_soft_nms(work, sigma, Nt, thr, method)
        assert len(keep) == n_out, name
        assert np.array_equal(work[:n_out].view(np.uint32), exp.view(np.uint32)), name
        n += 1
    assert n >= 30


def test_class_column_not_permuted(golden_dir):
    z = np.load(os.path.join(golden_dir, "softnms.npz"))
    inp, out = z["sixcol_n120/in"], z["sixcol_n120/out"]
    assert np.array_equal(out[:, 5], inp[:len(out), 5])   # cpu_nms.pyx:55-66 move cols 0..4 only


def test_zero_division_raises():
    a = np.array([[0, 0, -1, -1, 0.9], [0, 0, -1, -1, 0.8]], dtype=np.float32)  # areas 0 -> ua == 0 ... not overlapping
    # iw = min(-1,-1)-max(0,0)+1 = 0 -> no overlap branch, no error
    assert nms.cpu_soft_nms(a.copy(), method=2) == [0, 1]
    b = np.array([[0, 0, 10, 10, 0.9], [0, 0, 10, 10, 0.8]], dtype=np.float32)
    with pytest.raises(ZeroDivisionError):
        nms.cpu_soft_nms(b, sigma=0.0, method=2)


def test_against_compiled_reference_if_present():
    ref = nms.load_reference_cpu_nms()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(7)
    for trial in range(60):
        n = int(rng.choice([1, 2, 5, 64, 65, 150, 400]))
        xy = rng.uniform(0, 500, (n, 2))
        wh = rng.uniform(8, 120, (n, 2))
        s = rng.uniform(0.01, 1, (n, 1))
        b = np.concatenate([xy, xy + wh, s], 1).astype(np.float32)
        for method in (0, 1, 2):
            a1, a2 = b.copy(), b.copy()
            k1 = ref.cpu_soft_nms(a1, np.float32(0.5), np.float32(0.7), np.float32(0.1), np.uint8(method))
            k2 = nms.cpu_soft_nms(a2, 0.5, 0.7, 0.1, method)
            assert k1 == k2
            assert np.array_equal(a1[:len(k1)].view(np.uint32), a2[:len(k2)].view(np.uint32))


def test_hard_nms_basic():
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.95], dtype=np.float32)
    keep = nms.hard_nms(boxes, scores, 0.5)
    assert keep.tolist() == [3, 2]


def test_legacy_hard_nms_known_answer():
    """ext/nms/nms_wrapper.py:36-57: nms(anchor, thresh=0.3) keeps boxes [2, 3] (both threshold conventions)."""
    from oracle import nms as onms
    a = np.array([[10, 9, 20, 19, 0.5], [10, 10, 15, 30, 0.45], [10, 10, 26, 26, 0.7], [8, 9, 14, 16, 0.3],
                  [8, 8, 15, 15, 0.1]], np.float32)
    assert onms.legacy_nms(a, 0.3) == [2, 3]
    assert onms.legacy_nms(a, 0.3, inclusive=True) == [2, 3]
    # the two conventions differ exactly at IoU == thresh: two boxes with IoU(+1) = 0.5
    b = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8]], np.float32)     # areas 100 and 50, inter 50 -> 0.5
    assert onms.legacy_nms(b, 0.5) == [0, 1]
    assert onms.legacy_nms(b, 0.5, inclusive=True) == [0]


def test_legacy_nms_vs_reference_goldens(golden_dir):
    """oracle/nms.py:legacy_nms against the kept-index lists the reference's own numpy NMS returned
    (/root/reference/ext/nms/nms/py_cpu_nms.py, run by tools/gen_golden_hardnms.py): 32 cases."""
    z = np.load(os.path.join(golden_dir, "hardnms.npz"))
    assert len(z["names"]) >= 30
    for name in z["names"]:
        name = str(name)
        d, thr = z[name + "/dets"], float(z[name + "/thresh"])
        assert nms.legacy_nms(d, thr) == [int(i) for i in z[name + "/keep"]], name


def test_tv_style_hard_nms_oracle_pinned_by_reference_on_integer_boxes(golden_dir):
    """oracle/nms.py:hard_nms (torchvision convention, unpinned by any torchvision output) agrees with the reference's
    py_cpu_nms on integer boxes after the x2+1 / y2+1 shift that makes the two IoU conventions identical."""
    z = np.load(os.path.join(golden_dir, "hardnms.npz"))
    seen = 0
    for name in z["names"]:
        name = str(name)
        if not name.startswith(("integer_grid", "exact_half")):
            continue
        d, thr = z[name + "/dets"], float(z[name + "/thresh"])
        if len(np.unique(d[:, 4])) != d.shape[0]:
            continue
        b = d[:, :4].copy()
        b[:, 2:4] += 1.0
        assert nms.hard_nms(b, d[:, 4], thr).tolist() == [int(i) for i in z[name + "/keep"]], name
        seen += 1
    assert seen >= 2
