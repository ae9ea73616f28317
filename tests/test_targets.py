"""Host-side target contract: the oracle's restatement (oracle/targets.py) against golden vectors produced by the
reference's own to_heatmap (tools/gen_goldens.py g9), and the host half of the synthetic batch recipe."""
import os

import numpy as np
import torch


def test_oracle_to_heatmap_matches_reference(golden_dir):
    from oracle.targets import to_heatmap
    z = np.load(os.path.join(golden_dir, "targets.npz"))
    for i in range(2):
        img = int(z["c%d/img" % i])
        annos = torch.from_numpy(z["c%d/annos" % i])
        before = annos.clone()
        _, a, hm, wh, ind, off, mask = to_heatmap((torch.zeros(3, img, img), annos), scale_factor=4)
        assert torch.equal(a, before)                       # annotations are not mutated
        np.testing.assert_array_equal(hm.numpy(), z["c%d/hm" % i])
        np.testing.assert_array_equal(wh.numpy(), z["c%d/wh" % i])
        np.testing.assert_array_equal(ind.numpy(), z["c%d/ind" % i])
        np.testing.assert_array_equal(off.numpy(), z["c%d/off" % i])
        np.testing.assert_array_equal(mask.numpy(), z["c%d/mask" % i])


def test_synthetic_frames_and_host_collate_contract():
    from oracle.targets import host_batch
    from rrnet_amd.datasets.synthetic import synth_frames
    imgs, annos_list = synth_frames(2, 128, 160, boxes_per_image=7)
    assert imgs.shape == (2, 3, 128, 160) and len(annos_list) == 2 and annos_list[0].shape == (7, 8)
    imgs, annos, hms, whs, inds, offs, masks, names = host_batch(imgs, annos_list)
    assert hms.shape == (2, 10, 32, 40)
    assert annos.shape == (2, 7, 8) and whs.shape == (2, 7, 2) and inds.shape == (2, 7, 1)
    assert float(hms.max()) == 1.0 and float(inds.max()) < 32 * 40
    a2 = synth_frames(2, 128, 160, boxes_per_image=7)[1]
    assert all(torch.equal(x, y) for x, y in zip(annos_list, a2))       # deterministic under the seed


def test_product_package_holds_no_host_target_loop():
    """The reference's per-box host loop must not live in the product: targets are rr_ctnet_targets."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rrnet_amd")
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "to_heatmap" not in open(os.path.join(d, f)).read(), os.path.join(d, f)
