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


# the last case is BASELINE configs[4] at its named size: one 1920x1080 frame = a 270x480 map, K=1500
@pytest.mark.parametrize("hf,wf,k,frames", [(32, 48, 200, 3), (68, 120, 1500, 2), (270, 480, 1500, 1)])
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
        g, so, sl = ops.group_by_class(torch.from_numpy(boxes).cuda(), nc)
        g, so, sl = g.cpu().numpy(), so.cpu().numpy(), sl.cpu().numpy()
        for i in range(b):
            valid = boxes[i][boxes[i, :, 5] < nc]
            order = np.argsort(valid[:, 5], kind='stable')
            np.testing.assert_array_equal(g[i, :valid.shape[0]], valid[order])
            for c in range(nc):
                assert so[i * nc + c] == i * k + int((valid[:, 5] < c).sum())
                assert sl[i * nc + c] == int((valid[:, 5] == c).sum())      # exact lengths, also for the last class
        assert so[-1] == (b - 1) * k + int((boxes[b - 1, :, 5] < nc).sum())


def test_multi_scale_evaluate_images_vs_oracle():
    """the multi-scale evaluation body (operators/rrnet_operator.py:256-279) — bilinear
    align_corners rescale (rr_resize_bilinear_ac), full model, generate_bbox, score filter, cross-scale concat + sort,
    per-class Soft-NMS, sort — against the same composition of the CPU oracle on a tiny RRNet."""
    import types
    from types import SimpleNamespace
    import torch.nn.functional as F
    from oracle import model as om, nms as onms, ops as oo
    from rrnet_amd import ops
    from rrnet_amd.models.rrnet import RRNet
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    from tests.helpers import det_fill
    cfg = SimpleNamespace(num_classes=10, Model=SimpleNamespace(num_stacks=2, backbone="hourglass_tiny", nms_type_for_stage1="nms",
                          nms_per_class_for_stage1=True), Train=SimpleNamespace(scale_factor=4),
                          Val=SimpleNamespace(scales=[1, 1.25, 1.5], auto_test=False))
    model = RRNet(cfg)
    sd = det_fill({k: tuple(v.shape) for k, v in model.state_dict().items()}, 80)
    rng = np.random.default_rng(3)
    for i in range(2):
        sd["hm.detect_layer.%d.1.bias" % i].fill_(-2.19)
        sd["wh.detect_H_layer.%d.0.conv.bias" % i].fill_(3.0)
        sd["wh.detect_W_layer.%d.0.conv.bias" % i].fill_(3.0)
    sd["head_detector.regressor.weight"] = sd["head_detector.regressor.weight"] * 0.05   # moderate deltas: exp() stays finite
    img = torch.from_numpy(rng.normal(0, 1, (1, 3, 256, 256)).astype(np.float32))   # 64x64 map: k=1500 <= H*W as the reference needs
    # eval-mode BN needs running statistics that match the activations (a 60-layer net with unit running variance
    # saturates every score to 1.0): one training-mode pass of the oracle with momentum 1 stores the batch statistics
    om.BN_MOMENTUM = 1.0
    try:
        with torch.no_grad():
            pc = om.Params(sd, training=True)
            om.stage1(pc, om.hourglass_net(pc, img))
    finally:
        om.BN_MOMENTUM = 0.1
    model.load_state_dict(sd)
    # the resize kernel alone, odd and even target sizes
    for s in (1.25, 1.5, 0.7):
        ref = F.interpolate(img, scale_factor=s, mode='bilinear', align_corners=True)
        got = ops.resize_bilinear_ac(img.cuda(), s)
        assert tuple(got.shape) == tuple(ref.shape)
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=1e-4, rtol=1e-4)   # source coordinate rounds differently by an ulp
    # oracle composition
    P = om.Params({k: v.clone() for k, v in sd.items()}, training=False)
    per_scale = []
    with torch.no_grad():
        for s in cfg.Val.scales:
            x = F.interpolate(img, scale_factor=s, mode='bilinear', align_corners=True)
            outs = om.rrnet_forward(P, x, k=1500)
            _, pred = oo.generate_bbox(outs, 0, 4)
            pred = pred[pred[:, 4] > 0.01]
            pred[:, :4] = pred[:, :4] / s
            per_scale.append(pred)
    ref = torch.cat(per_scale)
    ref = ref[torch.sort(ref[:, 4], descending=True, stable=True)[1]]
    ref = onms.ext_nms(ref.numpy())
    ref = ref[np.argsort(-ref[:, 4], kind='stable')]
    # product
    ns = SimpleNamespace(cfg=cfg, model=model.cuda().to(memory_format=torch.channels_last).eval())
    ns.generate_bbox = types.MethodType(RRNetOperator.generate_bbox, ns)
    ns._ext_nms = RRNetOperator._ext_nms
    ns._ext_nms_device = RRNetOperator._ext_nms_device
    with torch.no_grad():
        got = RRNetOperator.evaluate_images(ns, img.cuda()).numpy()
    assert abs(got.shape[0] - ref.shape[0]) <= max(2, ref.shape[0] // 200), (got.shape, ref.shape)
    # rows are score-ordered on both sides; a rare threshold flip inserts / drops a row and near-equal scores may swap,
    # so every reference row is looked up in a small window of the product's rows
    used = np.zeros(got.shape[0], bool)
    hits = 0
    for i in range(ref.shape[0]):
        lo, hi = max(0, i - 40), min(got.shape[0], i + 41)
        cand = np.where(~used[lo:hi] & (np.abs(got[lo:hi, 4] - ref[i, 4]) < 1e-4) & (got[lo:hi, 5] == ref[i, 5]) &
                        np.all(np.abs(got[lo:hi, :4] - ref[i, :4]) < 5e-2 + 1e-3 * np.abs(ref[i, :4]), axis=1))[0]
        if cand.size:
            used[lo + cand[0]] = True
            hits += 1
    assert hits >= 0.98 * ref.shape[0], (hits, ref.shape[0])


def test_sort_rows_by_score_is_the_stable_descending_sort():
    from rrnet_amd import ops
    rng = np.random.default_rng(4)
    for n in (1, 7, 1500, 9000, 16384):
        rows = rng.uniform(0, 100, (n, 6)).astype(np.float32)
        rows[:, 4] = np.round(rng.uniform(0, 1, n) * 50) / 50            # many exact ties
        got = ops.sort_rows_by_score(torch.from_numpy(rows).cuda()).cpu().numpy()
        exp = rows[np.argsort(-rows[:, 4], kind="stable")]
        assert np.array_equal(got, exp), n
