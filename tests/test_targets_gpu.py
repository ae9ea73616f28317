"""Device target generation (rr_ctnet_targets — the product's loader path) against
  (i)  tests/golden/targets.npz, produced by the reference's own to_heatmap (tools/gen_goldens.py g9), directly;
  (ii) the oracle's host restatement + collate_fn_ctnet (oracle/targets.py, itself bit-exact vs that golden) on
       ragged batches at the bench size.
Regression targets bit-identical, heat-map within expf's last bit, peaks exactly 1, zeros exactly 0."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _hm_close(hm_dev, hm_ref):
    hm_c = hm_dev.cpu().contiguous()
    np.testing.assert_allclose(hm_c.numpy(), hm_ref.numpy(), atol=2e-7, rtol=0)
    assert torch.equal(hm_c == 1, hm_ref == 1)                       # focal loss compares gt == 1 exactly
    assert torch.equal(hm_c == 0, hm_ref == 0)


def test_ctnet_targets_vs_reference_golden(golden_dir):
    from rrnet_amd.datasets.synthetic import collate_ctnet_device
    z = np.load(os.path.join(golden_dir, "targets.npz"))
    for i in range(2):
        img = int(z["c%d/img" % i])
        annos = torch.from_numpy(z["c%d/annos" % i])
        annos_d, hm, wh, ind, off, mask = collate_ctnet_device([annos], img, img)
        assert torch.equal(annos_d[0].cpu(), annos[:, :8])
        for got, name in ((wh, "wh"), (ind, "ind"), (off, "off"), (mask, "mask")):
            np.testing.assert_array_equal(got[0].cpu().numpy(), z["c%d/%s" % (i, name)])
        _hm_close(hm[0], torch.from_numpy(z["c%d/hm" % i]))


def test_to_heatmap_transform_class_vs_reference_golden(golden_dir):
    from rrnet_amd.datasets.transforms import ToHeatmap
    z = np.load(os.path.join(golden_dir, "targets.npz"))
    img = int(z["c0/img"])
    annos = torch.from_numpy(z["c0/annos"])
    out = ToHeatmap(4, 10)((torch.zeros(3, img, img), annos))
    assert out[1] is annos and out[2].shape == (10, img // 4, img // 4)
    np.testing.assert_array_equal(out[3].numpy(), z["c0/wh"])
    _hm_close(out[2], torch.from_numpy(z["c0/hm"]))


@pytest.mark.parametrize("h,w,n", [(128, 160, 30), (512, 512, 100), (1024, 1024, 100)])
def test_ctnet_targets_vs_oracle(h, w, n):
    from oracle.targets import host_batch
    from rrnet_amd.datasets.synthetic import collate_ctnet_device, synth_annotations
    rng = np.random.default_rng(31)
    counts = [n, max(n // 3, 1), n - 1]
    annos_list = [torch.from_numpy(synth_annotations(rng, c, h, w)) for c in counts]
    # some boxes hugging the borders and a degenerate (zero-area) one
    annos_list[0][0, :4] = torch.tensor([0.0, 0.0, 9.0, 7.0])
    annos_list[0][1, :4] = torch.tensor([w - 12.0, h - 10.0, 12.0, 10.0])
    annos_list[1][0, 2] = 0.0
    _, annos_h, hms_h, whs_h, inds_h, offs_h, masks_h, _ = host_batch(torch.zeros(3, 3, h, w), annos_list)
    annos_d, hm, wh, ind, off, mask = collate_ctnet_device(annos_list, h, w)
    assert torch.equal(annos_d.cpu(), annos_h)
    assert torch.equal(wh.cpu(), whs_h) and torch.equal(off.cpu(), offs_h)
    assert torch.equal(ind.cpu(), inds_h) and torch.equal(mask.cpu(), masks_h)
    _hm_close(hm, hms_h)


def test_loader_batches_are_device_built_and_annotations_survive_the_criterion():
    """SyntheticDronesDET (the operators' training loader): targets from rr_ctnet_targets, resident; get_batch hands
    out a fresh copy of the annotations (the criterion converts them to xyxy in place, rrnet_operator.py:67)."""
    from types import SimpleNamespace
    from oracle.targets import host_batch
    from rrnet_amd.datasets.synthetic import SyntheticDronesDET, synth_frames
    cfg = SimpleNamespace(seed=219, num_classes=10, Train=SimpleNamespace(scale_factor=4))
    ld = SyntheticDronesDET(cfg, 2, 128, 160, boxes_per_image=9, pool=1)
    b0 = ld.get_batch()
    assert all(t.is_cuda for t in b0[:7])
    ref = host_batch(*synth_frames(2, 128, 160, boxes_per_image=9, seed=219))
    assert torch.equal(b0[0].cpu(), ref[0]) and torch.equal(b0[1].cpu(), ref[1])
    _hm_close(b0[2], ref[2])
    b0[1][:, :, 2:4] += b0[1][:, :, 0:2]
    assert torch.equal(ld.get_batch()[1].cpu(), ref[1])
