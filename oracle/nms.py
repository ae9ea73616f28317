"""ORACLE (test infrastructure only) — ctypes front-end to oracle/soft_nms.c and, when it was
built in the build container, to oracle/_ref (the reference's own compiled cpu_soft_nms).

Restates /root/reference/ext/nms/nms_wrapper.py:13-19 (soft_nms) and the per-class drivers
/root/reference/operators/rrnet_operator.py:211-232 (_ext_nms) and
/root/reference/models/rrnet.py:56-80 (RRNet.nms).
"""
import ctypes
import glob
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            from oracle.build import build_oracle
            build_oracle()
        L = ctypes.CDLL(path)
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int)
        L.oracle_soft_nms.argtypes = [f32p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_uint]
        L.oracle_soft_nms.restype = ctypes.c_int
        L.oracle_soft_nms_segments.argtypes = [f32p, i32p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                               ctypes.c_uint, i32p]
        L.oracle_soft_nms_segments.restype = ctypes.c_int
        L.oracle_hard_nms.argtypes = [f32p, ctypes.c_int, ctypes.c_int, i32p, ctypes.c_float, i32p]
        L.oracle_hard_nms.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _f32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def cpu_soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """In-place on a C-contiguous float32 [N, >=5] array; returns list(range(N')).
    Same contract as cpu_nms.pyx:17-120 (ZeroDivisionError where the reference raises)."""
    assert boxes.dtype == np.float32 and boxes.ndim == 2 and boxes.flags["C_CONTIGUOUS"]
    n, stride = boxes.shape
    if n == 0:
        return []
    r = lib().oracle_soft_nms(_f32p(boxes), n, stride, np.float32(sigma), np.float32(Nt),
                              np.float32(threshold), int(method))
    if r < 0:
        raise ZeroDivisionError("float division")
    return list(range(r))


def soft_nms(dets, sigma=0.5, Nt=0.3, threshold=0.001, method=1):
    """nms_wrapper.py:13-19, including its reliance on in-place mutation: when `dets` is not
    already C-contiguous float32 the algorithm runs on a copy and the *unmodified* first N'
    rows come back."""
    work = np.ascontiguousarray(dets, dtype=np.float32)
    keep = cpu_soft_nms(work, np.float32(sigma), np.float32(Nt), np.float32(threshold), np.uint8(method))
    dets = np.asarray(dets) if not isinstance(dets, np.ndarray) else dets
    return dets[keep]


def soft_nms_segments(boxes, seg_off, sigma, Nt, threshold, method):
    """Batched segments (in place).  Returns n_out[nseg]."""
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int32)
    n_out = np.zeros(len(seg_off) - 1, dtype=np.int32)
    r = lib().oracle_soft_nms_segments(_f32p(boxes), _i32p(seg_off), len(seg_off) - 1, boxes.shape[1],
                                       np.float32(sigma), np.float32(Nt), np.float32(threshold),
                                       int(method), _i32p(n_out))
    if r < 0:
        raise ZeroDivisionError("float division")
    return n_out


def hard_nms(boxes, scores, thresh):
    """torchvision.ops.nms semantics as called at models/rrnet.py:69,78 (parity unpinned by the
    reference — torchvision is not vendored).  Returns kept indices, score-descending."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable").astype(np.int32)
    keep = np.zeros(n, dtype=np.int32)
    nk = lib().oracle_hard_nms(_f32p(boxes), n, boxes.shape[1], _i32p(order), np.float32(thresh), _i32p(keep))
    return keep[:nk].astype(np.int64)


def ext_nms(pred_bbox, per_cls=True):
    """rrnet_operator.py:211-232 on a float32 numpy [n,6] xywh array -> [n',6] xywh array."""
    pred_bbox = np.ascontiguousarray(pred_bbox, dtype=np.float32)
    if pred_bbox.shape[0] == 0:
        return pred_bbox
    outs = []
    if per_cls:
        for c in np.unique(pred_bbox[:, 5]):
            b = pred_bbox[pred_bbox[:, 5] == c].copy()
            b[:, 2] = b[:, 0] + b[:, 2]
            b[:, 3] = b[:, 1] + b[:, 3]
            outs.append(soft_nms(b, Nt=0.7, threshold=0.1, method=2))
        out = np.concatenate(outs, axis=0)
    else:
        b = pred_bbox.copy()
        b[:, 2] = b[:, 0] + b[:, 2]
        b[:, 3] = b[:, 1] + b[:, 3]
        out = soft_nms(b, Nt=0.7, threshold=0.1, method=2)
    out[:, 2:4] -= out[:, 0:2]
    return out


def load_reference_cpu_nms():
    """oracle/_ref: the reference's own compiled module, or None if it was not built."""
    hits = glob.glob(os.path.join(_HERE, "_ref", "cpu_nms*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("cpu_nms", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def legacy_nms(dets, thresh, inclusive=False):
    """Restatement of the reference's hard-NMS family on a float32 [N,>=5] array -> kept indices.
    ext/nms/nms/cpu_nms.pyx:129-176 (`ovr >= thresh` suppresses: inclusive=True) and
    ext/nms/nms/nms_kernel.cu:23-31,62-75 + gpu_nms.pyx:17-31 / py_cpu_nms.py (`ovr > thresh`): the "+1" pixel
    convention, visiting order `scores.argsort()[::-1]`, all arithmetic in float32.
    Pinned by the known answer in ext/nms/nms_wrapper.py:36-57 (thresh 0.3 -> keep [2, 3]); cpu_nms.pyx as a whole
    does not cythonize under numpy 2 (np.int_t, np.int), so there is no compiled reference for it here."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    x1, y1, x2, y2, sc = (dets[:, i] for i in range(5))
    one = np.float32(1)
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    order = sc.argsort()[::-1]
    n = dets.shape[0]
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(int(i))
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1 + one)
        h = np.maximum(np.float32(0), yy2 - yy1 + one)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        hit = (ovr >= np.float32(thresh)) if inclusive else (ovr > np.float32(thresh))
        suppressed[rest[hit]] = True
    return keep
