"""Generates tests/golden/softnms.npz from the REFERENCE's own compiled cpu_soft_nms
(oracle/_ref, built by oracle/build.py from /root/reference/ext/nms/nms/cpu_nms.pyx:1-120).
Runs only in the build container; the .npz it writes is data (inputs + expected outputs).

Cases: the README known-answer vector (nms_wrapper.py:36-50) under the three
methods; seeded random sets N in {0,1,2,64,65,150,1500}; adversarial sets (exact score ties,
identical boxes, disjoint boxes below threshold, 6-column input whose class column must not move).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nms  # noqa: E402
from oracle.build import build_ref  # noqa: E402


def main():
    build_ref()
    ref = nms.load_reference_cpu_nms()
    assert ref is not None, "oracle/_ref not built (needs /root/reference)"
    rng = np.random.default_rng(219)
    cases = []

    readme = np.array([[10, 9, 20, 19, 0.5], [10, 10, 15, 30, 0.45], [10, 10, 26, 26, 0.7],
                       [8, 9, 14, 16, 0.3], [8, 8, 15, 15, 0.1]], dtype=np.float32)
    for method in (0, 1, 2):
        cases.append(("readme_m%d" % method, readme.copy(), 0.3, 0.4, 0.001, method))

    def rand_set(n, cols=5, span=1800.0):
        xy = rng.uniform(0, span, (n, 2))
        wh = rng.uniform(8, 120, (n, 2))
        s = rng.uniform(0.01, 1, (n, 1))
        parts = [xy, xy + wh, s]
        if cols == 6:
            parts.append(rng.integers(0, 10, (n, 1)).astype(np.float64))
        return np.concatenate(parts, 1).astype(np.float32)

    for n in (0, 1, 2, 64, 65, 150, 1500):
        for method, Nt, thr in ((2, 0.7, 0.1), (1, 0.3, 0.001), (0, 0.3, 0.001)):
            span = 1800.0 if n < 1000 else 900.0
            cases.append(("rand_n%d_m%d" % (n, method), rand_set(n, span=span), 0.5, Nt, thr, method))
    # dense clusters: many overlaps, many removals, deep decay chains
    for n in (150, 700):
        b = rand_set(n, span=200.0)
        cases.append(("dense_n%d_gauss" % n, b, 0.5, 0.7, 0.1, 2))
        cases.append(("dense_n%d_linear" % n, b.copy(), 0.5, 0.3, 0.05, 1))
    # exact score ties (first-max tie-break + swap order)
    b = rand_set(200, span=300.0)
    b[:, 4] = (np.round(b[:, 4] * 4) / 4 + 0.01).astype(np.float32)
    cases.append(("ties_n200", b, 0.5, 0.7, 0.1, 2))
    # identical boxes
    b = rand_set(64, span=300.0)
    b[1::2, :4] = b[0::2, :4]
    cases.append(("identical_n64", b, 0.5, 0.7, 0.1, 2))
    # disjoint boxes, all scores below the threshold: all survive (threshold only tested on overlap)
    n = 40
    b = np.zeros((n, 5), np.float32)
    b[:, 0] = np.arange(n) * 50
    b[:, 1] = 0
    b[:, 2] = b[:, 0] + 10
    b[:, 3] = 10
    b[:, 4] = rng.uniform(0.001, 0.05, n)
    cases.append(("disjoint_lowscore", b, 0.5, 0.7, 0.1, 2))
    # 6 columns: class column is not permuted (cpu_nms.pyx:55-66)
    cases.append(("sixcol_n120", rand_set(120, cols=6, span=250.0), 0.5, 0.7, 0.1, 2))

    out = {}
    names = []
    for name, boxes, sigma, Nt, thr, method in cases:
        work = np.ascontiguousarray(boxes.copy())
        if work.shape[0]:
            keep = ref.cpu_soft_nms(work, np.float32(sigma), np.float32(Nt), np.float32(thr), np.uint8(method))
        else:
            keep = []
        names.append(name)
        out[name + "/in"] = boxes
        out[name + "/out"] = work[:len(keep)].copy()
        out[name + "/params"] = np.array([sigma, Nt, thr, method], dtype=np.float64)
        out[name + "/n_out"] = np.array(len(keep), dtype=np.int64)
    out["names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "softnms.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(names), "cases")


if __name__ == "__main__":
    main()
