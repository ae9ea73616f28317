"""Device target generation (rr_ctnet_targets) against the host restatement of the reference's to_heatmap +
collate_fn_ctnet (rrnet_amd/datasets/transforms/functional.py, itself bit-exact vs the reference golden
tests/golden/targets.npz): regression targets bit-identical, heat-map within expf's last bit, peaks exactly 1."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,n", [(128, 160, 30), (512, 512, 100), (1024, 1024, 100)])
def test_ctnet_targets_vs_host(h, w, n):
    from rrnet_amd.datasets.synthetic import collate_ctnet, collate_ctnet_device, synth_annotations
    from rrnet_amd.datasets.transforms.functional import to_heatmap
    rng = np.random.default_rng(31)
    counts = [n, max(n // 3, 1), n - 1]
    annos_list = [torch.from_numpy(synth_annotations(rng, c, h, w)) for c in counts]
    # some boxes hugging the borders and a degenerate (zero-area) one
    annos_list[0][0, :4] = torch.tensor([0.0, 0.0, 9.0, 7.0])
    annos_list[0][1, :4] = torch.tensor([w - 12.0, h - 10.0, 12.0, 10.0])
    annos_list[1][0, 2] = 0.0
    img = torch.zeros(3, h, w)
    samples = []
    for a in annos_list:
        _, aa, hm, wh, ind, off, mask = to_heatmap((img, a), 4, 10)
        samples.append((img, aa, hm, wh, ind, off, mask.float(), "x"))
    _, annos_h, hms_h, whs_h, inds_h, offs_h, masks_h, _ = collate_ctnet(samples)
    annos_d, hm, wh, ind, off, mask = collate_ctnet_device(annos_list, h, w)
    assert torch.equal(annos_d.cpu(), annos_h)
    assert torch.equal(wh.cpu(), whs_h) and torch.equal(off.cpu(), offs_h)
    assert torch.equal(ind.cpu(), inds_h) and torch.equal(mask.cpu(), masks_h)
    hm_c = hm.cpu().contiguous()
    np.testing.assert_allclose(hm_c.numpy(), hms_h.numpy(), atol=2e-7, rtol=0)
    assert torch.equal(hm_c == 1, hms_h == 1)                       # focal loss compares gt == 1 exactly
    assert torch.equal(hm_c == 0, hms_h == 0)
