"""Host-side target contract: rrnet_amd.datasets.transforms.functional.to_heatmap against golden
vectors produced by the reference's to_heatmap (tools/gen_goldens.py g9)."""
import os

import numpy as np
import torch


def test_to_heatmap_matches_reference(golden_dir):
    from rrnet_amd.datasets.transforms.functional import to_heatmap
    z = np.load(os.path.join(golden_dir, "targets.npz"))
    for i in range(2):
        img = int(z["c%d/img" % i])
        annos = torch.from_numpy(z["c%d/annos" % i])
        _, a, hm, wh, ind, off, mask = to_heatmap((torch.zeros(3, img, img), annos), scale_factor=4)
        assert torch.equal(a, annos)                        # annotations are not mutated
        np.testing.assert_array_equal(hm.numpy(), z["c%d/hm" % i])
        np.testing.assert_array_equal(wh.numpy(), z["c%d/wh" % i])
        np.testing.assert_array_equal(ind.numpy(), z["c%d/ind" % i])
        np.testing.assert_array_equal(off.numpy(), z["c%d/off" % i])
        np.testing.assert_array_equal(mask.float().numpy(), z["c%d/mask" % i])


def test_synthetic_batch_contract():
    from rrnet_amd.datasets.synthetic import synth_batch
    imgs, annos, hms, whs, inds, offs, masks, names = synth_batch(2, 128, 160, boxes_per_image=7)
    assert imgs.shape == (2, 3, 128, 160) and hms.shape == (2, 10, 32, 40)
    assert annos.shape == (2, 7, 8) and whs.shape == (2, 7, 2) and inds.shape == (2, 7, 1)
    assert float(hms.max()) == 1.0 and float(inds.max()) < 32 * 40
    a2 = synth_batch(2, 128, 160, boxes_per_image=7)[1]
    assert torch.equal(annos, a2)                           # deterministic under the seed
