"""Host logic of the AP / AR evaluator (rrnet_amd/utils/metrics/metrics.py) against goldens produced by the
reference's own bbox_iou / get_tp / calculate_ap_rc / evaluate_once (tools/gen_goldens.py g10)."""
import contextlib
import io
import os

import numpy as np
import torch


def _load(golden_dir):
    return np.load(os.path.join(golden_dir, "metrics.npz"))


def test_bbox_iou_and_evaluate_once(golden_dir):
    from rrnet_amd.utils.metrics import metrics as M
    z = _load(golden_dir)
    for i in range(4):
        pred, target = torch.from_numpy(z["c%d/pred" % i]), torch.from_numpy(z["c%d/target" % i])
        iou, ov = M.bbox_iou(pred[:, :4], target[:, :4], x1y1x2y2=False, overlap=True)
        np.testing.assert_allclose(iou.numpy(), z["c%d/iou" % i], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(ov.numpy(), z["c%d/overlap" % i], rtol=1e-6, atol=1e-7)
        with contextlib.redirect_stdout(io.StringIO()):
            ap, rc = M.evaluate_once(pred.clone(), target.clone(), max_det_num=100 if i == 2 else 500)
        np.testing.assert_allclose(ap.numpy(), z["c%d/ap" % i], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(float(rc), float(z["c%d/rc" % i]), rtol=1e-5, atol=1e-6)


def test_get_tp_accumulation_and_ap(golden_dir):
    from rrnet_amd.utils.metrics import metrics as M
    z = _load(golden_dir)
    flags, confs, tc, ic = M._fresh(11, 10)
    for i in range(4):
        pred, target = torch.from_numpy(z["c%d/pred" % i]), torch.from_numpy(z["c%d/target" % i])
        flags, confs, tc, ic = M.get_tp(pred.clone(), target.clone(), flags, confs, tc, ic, M.THRESHOLDS, 11)
        np.testing.assert_array_equal(tc.numpy(), z["c%d/target_count" % i])
        np.testing.assert_array_equal(ic.numpy(), z["c%d/in_img_count" % i])
    for c in range(10):
        np.testing.assert_array_equal(flags[c].numpy(), z["all/flags%d" % c])
        np.testing.assert_array_equal(confs[c].numpy(), z["all/confs%d" % c])
    ap, rc = M.calculate_ap_rc(flags, confs, tc, ic)
    np.testing.assert_allclose(ap.numpy(), z["all/ap"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(rc), float(z["all/rc"]), rtol=1e-5, atol=1e-6)


def test_evaluate_results_files(golden_dir, tmp_path):
    """File-based driver == the in-memory accumulation over the same (integer-snapped) detections."""
    from rrnet_amd.utils.metrics import metrics as M
    z = _load(golden_dir)
    pd_dir, gt_dir = tmp_path / "pred", tmp_path / "gt"
    pd_dir.mkdir(); gt_dir.mkdir()
    flags, confs, tc, ic = M._fresh(11, 10)
    for i in range(4):
        pred, target = z["c%d/pred" % i], z["c%d/target" % i]
        with open(pd_dir / ("f%d.txt" % i), "w") as f:
            for r in pred:
                f.write('%f,%f,%f,%f,%.4f,%d,-1,-1\n' % (r[0], r[1], r[2], r[3], r[4], int(r[5])))
        with open(gt_dir / ("f%d.txt" % i), "w") as f:
            for r in target:
                f.write(','.join('%d' % int(v) for v in r) + '\n')
    with contextlib.redirect_stdout(io.StringIO()) as out:
        ap, rc = M.evaluate_results(str(pd_dir), str(gt_dir))
    assert "Average Precision  (AP) @[ IoU=0.50:0.95]" in out.getvalue()
    for name in M._names(str(pd_dir)):                       # same file order as the driver
        p = M._snap(M._read(os.path.join(str(pd_dir), name + ".txt")).astype(np.float64))
        t = M._read(os.path.join(str(gt_dir), name + ".txt"))
        flags, confs, tc, ic = M.get_tp(torch.from_numpy(p).float()[:500], torch.from_numpy(t).float()[:500], flags, confs,
                                        tc, ic, M.THRESHOLDS, 11)
    ap2, rc2 = M.calculate_ap_rc(flags, confs, tc, ic)
    assert torch.equal(ap, ap2) and torch.equal(rc, rc2)
    assert 0.0 < float(ap.mean()) < 1.0
