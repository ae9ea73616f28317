"""Drop-in for the reference's ext/nms/nms_wrapper.py (soft_nms :13-19, nms :23-33), backed by
the wavefront-parallel HIP kernels in librrnet_hip.so instead of the Cython/CUDA extensions
ext/nms/nms/{cpu_nms.pyx,gpu_nms.pyx,nms_kernel.cu}.

`soft_nms` keeps the reference's host contract (numpy in, numpy out, in-place mutation of a
C-contiguous float32 `dets`, unmodified rows otherwise).  The batched device entry points
(`soft_nms_segments`, `hard_nms_segments`) are what the model / operator code uses so that the
per-image x per-class Python loops of the reference become one launch."""
import numpy as np
import torch

from rrnet_amd import _C

_P = _C.c_void_p


def soft_nms_segments(boxes, seg_off, max_seg, sigma=0.5, Nt=0.3, threshold=0.001, method=1, seg_len=None,
                      check=True):
    """boxes: cuda float32 [total, stride>=5] (modified in place); seg_off: cuda int32 [nseg+1];
    seg_len (optional cuda int32 [nseg]): explicit segment lengths (rows past the length are ignored).
    Returns n_out cuda int32 [nseg]; rows [seg_off[s], seg_off[s]+n_out[s]) are the kept boxes in
    the reference's order.  Raises ZeroDivisionError where the reference would (`check=False` skips the
    host read of the flag and returns (n_out, err) for a caller that synchronises later)."""
    _C.require_cuda(boxes, seg_off)
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.dim() == 2
    assert seg_off.dtype == torch.int32
    nseg = seg_off.numel() - 1
    n_out = torch.zeros(max(nseg, 0), dtype=torch.int32, device=boxes.device)
    if nseg <= 0:
        return n_out
    if int(method) == 2 and float(np.float32(sigma)) == 0.0:
        raise ZeroDivisionError("float division")
    err = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    f_ws = _C.fn("rr_soft_nms_workspace_bytes")
    nbytes = f_ws(boxes.size(0), int(max_seg))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=boxes.device) if nbytes else None
    tail = (int(max_seg), boxes.size(1), float(np.float32(sigma)), float(np.float32(Nt)), float(np.float32(threshold)),
            int(method), _C.ptr(n_out), _C.ptr(err), _C.ptr(ws), _C.stream())
    if seg_len is None:
        _C.check(_C.fn("rr_soft_nms_segments")(_C.ptr(boxes), _C.ptr(seg_off), nseg, *tail), "rr_soft_nms_segments")
    else:
        assert seg_len.dtype == torch.int32 and seg_len.numel() == nseg
        _C.check(_C.fn("rr_soft_nms_ragged")(_C.ptr(boxes), _C.ptr(seg_off), _C.ptr(seg_len), nseg, *tail),
                 "rr_soft_nms_ragged")
    if not check:
        return n_out, err
    if int(err.item()) != 0:
        raise ZeroDivisionError("float division")
    return n_out


def soft_nms(dets, sigma=0.5, Nt=0.3, threshold=0.001, method=1):
    """Reference signature (nms_wrapper.py:13-19).  `dets`: array-like [N, >=5]."""
    work = np.ascontiguousarray(dets, dtype=np.float32)
    n = work.shape[0]
    if n == 0:
        return np.asarray(dets)[[]] if not isinstance(dets, np.ndarray) else dets[[]]
    dev = torch.device("cuda", torch.cuda.current_device())
    d = torch.from_numpy(work).to(dev)
    seg = torch.tensor([0, n], dtype=torch.int32, device=dev)
    n_out = soft_nms_segments(d, seg, n, sigma, Nt, threshold, np.uint8(method))
    k = int(n_out.item())
    # in-place mutation, as the Cython routine does on the buffer it was handed: when `dets`
    # already was C-contiguous float32, `work` IS `dets` and the caller sees the permuted rows.
    work[:k] = d[:k].cpu().numpy()
    if not isinstance(dets, np.ndarray):
        dets = np.asarray(dets)
    return dets[list(range(k))]


def _nms_device(dets, thresh, inclusive):
    """Shared body of gpu_nms / cpu_nms: numpy [N,>=5] -> list of kept row indices (original numbering)."""
    dets = np.asarray(dets)
    n = dets.shape[0]
    if n == 0:
        return []
    work = np.ascontiguousarray(dets[:, :5], dtype=np.float32)
    order = work[:, 4].argsort()[::-1]                      # the reference's own ordering (gpu_nms.pyx:25-28)
    dev = torch.device("cuda", torch.cuda.current_device())
    d = torch.from_numpy(work[order]).to(dev)
    ws = torch.empty(_C.fn("rr_nms_workspace_bytes")(n), dtype=torch.uint8, device=dev)
    keep = torch.empty(n, dtype=torch.int32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    _C.check(_C.fn("rr_nms_sorted")(_C.ptr(d), n, 5, float(np.float32(thresh)), int(inclusive), _C.ptr(ws), _C.ptr(keep),
                                    _C.ptr(num), _C.stream()), "rr_nms_sorted")
    k = int(num.item())
    return list(order[keep[:k].cpu().numpy()])


def gpu_nms(dets, thresh, device_id=0):
    """ext/nms/nms/gpu_nms.pyx:17-31: kept indices, IoU(+1) > thresh suppresses."""
    return _nms_device(dets, thresh, inclusive=0)


def cpu_nms(dets, thresh):
    """ext/nms/nms/cpu_nms.pyx:129-176: kept indices, IoU(+1) >= thresh suppresses."""
    return _nms_device(dets, thresh, inclusive=1)


def nms(dets, thresh, gpu_id=0):
    """Reference signature (nms_wrapper.py:23-33): the kept rows; gpu_id None selects the cpu_nms convention."""
    dets = np.asarray(dets)
    if dets.shape[0] == 0:
        return []
    keep = gpu_nms(dets[:, :5], thresh, device_id=gpu_id) if gpu_id is not None else cpu_nms(dets[:, :5], thresh)
    return dets[keep]
