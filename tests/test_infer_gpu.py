"""GPU parity of the batched inference post-process (rrnet_amd/inference.py, kernels rr_refine_boxes /
rr_soft_nms_ragged / rr_finalize_frames + decode / NMS / RoIAlign / head) against the CPU oracle
(oracle/infer.py), frame by frame."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _head_and_params(seed=5):
    from oracle import model as omodel
    from rrnet_amd.detectors.fasterrcnn_detector import FasterRCNNDetector
    from tests.helpers import det_fill, shapes_of
    head = FasterRCNNDetector()
    sd = det_fill(shapes_of(head.state_dict()), seed)
    rng = np.random.default_rng(seed)
    for k in sd:                                  # non-trivial running statistics for eval-mode BN
        if k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rng.normal(0, 0.2, sd[k].shape).astype(np.float32))
        if k.endswith("running_var"):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, sd[k].shape).astype(np.float32))
    # keep the regression deltas moderate so that exp() stays well conditioned
    sd["regressor.weight"] = sd["regressor.weight"] * 0.2
    head.load_state_dict(sd)
    head = head.cuda().eval()
    P = omodel.Params({"head_detector." + k: v.clone() for k, v in sd.items()}, training=False)
    return head, P


@pytest.mark.parametrize("hf,wf,k,frames", [(32, 48, 200, 3), (68, 120, 1500, 2)])
def test_refine_frames_matches_oracle(hf, wf, k, frames):
    from oracle import infer as oinfer
    from rrnet_amd import inference
    from rrnet_amd.datasets.synthetic import synth_head_outputs
    head, P = _head_and_params()
    hm, wh, off, feat = synth_head_outputs(frames, hf, wf, seed=11, planted=60)
    feat = feat - 0.3                              # pre-ReLU feature with negatives: the ReLU step matters
    boxes, frame_off = inference.refine_frames(hm.cuda(), wh.cuda(), off.cuda(), feat.cuda(), head, k=k)
    fo = frame_off.cpu().numpy()
    boxes = boxes.cpu().numpy()
    assert fo[0] == 0 and fo[-1] == boxes.shape[0]
    for f in range(frames):
        ref = oinfer.postprocess_frame(P, hm[f:f + 1], wh[f:f + 1], off[f:f + 1], feat[f:f + 1], k=k)
        got = boxes[fo[f]:fo[f + 1]]
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert np.all(np.diff(got[:, 4]) <= 0)                       # score-descending
        np.testing.assert_array_equal(got[:, 5], ref[:, 5])          # classes (order included)
        np.testing.assert_allclose(got[:, 4], ref[:, 4], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(got[:, :4], ref[:, :4], rtol=1e-3, atol=1e-3)


def test_refine_kernels_bit_exact_vs_host_composition():
    """rr_refine_boxes + rr_soft_nms_ragged + rr_finalize_frames on given RoIs / regressions equal the oracle's
    generate_bbox -> filter -> ext_nms -> sort bit for bit when exp() is kept out of the picture (reg[:,2:]=0)."""
    from oracle import nms as onms
    from oracle import ops as oops
    from rrnet_amd import ops
    from rrnet_amd.ext.nms.nms_wrapper import soft_nms_segments
    rng = np.random.default_rng(3)
    nframes, ncls = 3, 4
    rois, scores, clses, seg = [], [], [], [0]
    for f in range(nframes):
        for c in range(ncls):
            n = int(rng.integers(0, 70)) if (f, c) != (1, 2) else 0
            xy = rng.uniform(0, 40, (n, 2)).astype(np.float32)
            wh_ = rng.uniform(1, 12, (n, 2)).astype(np.float32)
            s = np.sort(rng.uniform(0.0, 1.0, n).astype(np.float32))[::-1]
            rois.append(np.concatenate([np.full((n, 1), f, np.float32), xy, xy + wh_], 1))
            scores.append(s)
            clses.append(np.full(n, c, np.float32))
            seg.append(seg[-1] + n)
    rois = np.concatenate(rois); scores = np.concatenate(scores).copy(); clses = np.concatenate(clses)
    reg = rng.normal(0, 0.1, (rois.shape[0], 4)).astype(np.float32)
    reg[:, 2:] = 0
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    seg_off = torch.tensor(seg, dtype=torch.int32).cuda()
    b6, seg_len = ops.refine_boxes(T(rois), T(reg), T(scores), T(clses), seg_off, 4.0, 0.01)
    n_out = soft_nms_segments(b6, seg_off, 70, sigma=0.5, Nt=0.7, threshold=0.1, method=2, seg_len=seg_len)
    out6, frame_off = ops.finalize_frames(b6, seg_off, n_out, nframes, ncls, 70 * ncls)
    fo = frame_off.cpu().numpy()
    out6 = out6.cpu().numpy()
    for f in range(nframes):
        outs = (None, None, None, torch.from_numpy(reg), torch.from_numpy(rois), torch.from_numpy(scores),
                torch.from_numpy(clses))
        _, pred = oops.generate_bbox(outs, f, 4)
        pred = pred[pred[:, 4] > 0.01].numpy()
        ref = onms.ext_nms(pred)
        ref = ref[np.argsort(-ref[:, 4], kind='stable')] if ref.shape[0] else ref.reshape(0, 6)
        got = out6[fo[f]:fo[f + 1]]
        assert got.shape == ref.shape
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_group_by_class_large_matches_stable_sort():
    from rrnet_amd import ops
    rng = np.random.default_rng(9)
    for k, nc in [(1500, 10), (4096, 10), (9000, 32), (20000, 7)]:
        b = 3
        boxes = rng.uniform(0, 100, (b, k, 6)).astype(np.float32)
        boxes[:, :, 5] = rng.integers(0, nc + 2, (b, k))            # two classes out of range: dropped
        g, so = ops.group_by_class(torch.from_numpy(boxes).cuda(), nc)
        g, so = g.cpu().numpy(), so.cpu().numpy()
        for i in range(b):
            valid = boxes[i][boxes[i, :, 5] < nc]
            order = np.argsort(valid[:, 5], kind='stable')
            np.testing.assert_array_equal(g[i, :valid.shape[0]], valid[order])
            for c in range(nc):
                assert so[i * nc + c] == i * k + int((valid[:, 5] < c).sum())
        assert so[-1] == (b - 1) * k + int((boxes[b - 1, :, 5] < nc).sum())
