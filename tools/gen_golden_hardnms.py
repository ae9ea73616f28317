"""Generates tests/golden/hardnms.npz by RUNNING THE REFERENCE's own pure-numpy hard NMS
(/root/reference/ext/nms/nms/py_cpu_nms.py:4-38, imported in place — build container only).
The .npz holds data only: input boxes, threshold, and the kept index list the reference returned.

py_cpu_nms is the `IoU(+1) > thresh suppresses` convention of gpu_nms (nms_kernel.cu:23-31,62-75); it pins
oracle/nms.py:legacy_nms, rr_nms_sorted, the `_nms` C entry and the nms_wrapper.gpu_nms / nms drop-ins.

Cases: seeded random sets N in {0,1,2,63,64,65,200,1500}, three thresholds; clustered (heavily overlapping) sets;
exact score ties; identical boxes; IoU exactly equal to the threshold; integer-coordinate boxes.
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ext/nms/nms/py_cpu_nms.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_py_cpu_nms", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.py_cpu_nms


def main():
    py_cpu_nms = load_reference()
    rng = np.random.default_rng(2190)
    cases = []

    def rand_set(n, span=600.0, lo=5.0, hi=120.0):
        xy = rng.uniform(0, span, (n, 2))
        wh = rng.uniform(lo, hi, (n, 2))
        return np.concatenate([xy, xy + wh, rng.uniform(0.01, 1, (n, 1))], 1).astype(np.float32)

    for n in (0, 1, 2, 63, 64, 65, 200, 1500):
        for thr in (0.3, 0.5, 0.7):
            cases.append(("rand_n%d_t%d" % (n, int(thr * 10)), rand_set(n), thr))
    # clusters: 40 centres, jittered copies -> long suppression chains
    c = rand_set(40, span=300.0, lo=30.0, hi=60.0)
    cl = np.repeat(c, 25, axis=0)
    cl[:, :4] += rng.normal(0, 3.0, (cl.shape[0], 4)).astype(np.float32)
    cl[:, 4] = rng.uniform(0.01, 1, cl.shape[0]).astype(np.float32)
    cases.append(("clusters_t5", cl.copy(), 0.5))
    cases.append(("clusters_t7", cl.copy(), 0.7))
    # exact score ties (the visiting order is numpy's argsort()[::-1], which the drop-in reproduces on the host)
    t = rand_set(300, span=200.0)
    t[:, 4] = (rng.integers(0, 8, 300) / 8.0 + 0.05).astype(np.float32)
    cases.append(("ties_t5", t, 0.5))
    # identical boxes
    d = np.repeat(rand_set(10), 7, axis=0)
    d[:, 4] = rng.uniform(0.01, 1, d.shape[0]).astype(np.float32)
    cases.append(("identical_t5", d, 0.5))
    # IoU(+1) == thresh exactly: areas 100 and 50, inter 50 -> 0.5 (kept by `>`), plus integer grids
    cases.append(("exact_half", np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8], [20, 20, 29, 29, 0.7],
                                          [20, 20, 29, 24, 0.95]], np.float32), 0.5))
    g = rng.integers(0, 40, (400, 2)).astype(np.float32)
    gi = np.concatenate([g, g + rng.integers(1, 20, (400, 2)).astype(np.float32),
                         rng.uniform(0.01, 1, (400, 1)).astype(np.float32)], 1).astype(np.float32)
    cases.append(("integer_grid_t3", gi, 0.3))
    cases.append(("integer_grid_t5", gi, 0.5))
    # the wrapper's own known answer (nms_wrapper.py:36-57)
    cases.append(("readme", np.array([[10, 9, 20, 19, 0.5], [10, 10, 15, 30, 0.45], [10, 10, 26, 26, 0.7],
                                      [8, 9, 14, 16, 0.3], [8, 8, 15, 15, 0.1]], np.float32), 0.3))
    out = {"names": np.array([c[0] for c in cases])}
    for name, dets, thr in cases:
        keep = py_cpu_nms(dets.copy(), thr) if dets.shape[0] else []
        out[name + "/dets"] = dets
        out[name + "/thresh"] = np.float64(thr)
        out[name + "/keep"] = np.asarray(keep, dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "hardnms.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d cases, %d bytes" % (path, len(cases), os.path.getsize(path)))


if __name__ == "__main__":
    main()
