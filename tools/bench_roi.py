"""RoIAlign (3x3 bins) at config-5 size: the RoIs of 128 decoded frames of 270x480 on their 256-channel feature maps.
  python tools/bench_roi.py [--iters 10]   -> one JSON line: ms per launch, requested GB, TB/s"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--frames", type=int, default=128)
    a = ap.parse_args()
    from rrnet_amd import ops
    from rrnet_amd.datasets.synthetic import synth_head_outputs
    from rrnet_amd.models.rrnet import stage1_proposals
    dev = torch.device("cuda", 0)
    hm, wh, off, feat = (ops.to_nhwc(t) for t in synth_head_outputs(a.frames, 270, 480, seed=219, device=dev))
    with torch.no_grad():
        rois, scores, clses, row_off = stage1_proposals(hm, wh, off, 1500, 10, 'nms', True, want_offsets=True)
        order = ops.roi_spatial_order(rois, row_off[::10].contiguous())
        res = {}
        for name, o in (("ordered", order), ("decode_order", None)):
            out = ops.roi_align_fwd(feat, rois, (3, 3), order=o)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                out = ops.roi_align_fwd(feat, rois, (3, 3), order=o)
            e1.record()
            torch.cuda.synchronize()
            res[name + "_ms"] = round(e0.elapsed_time(e1) / a.iters, 4)
    res["rois"] = int(rois.shape[0])
    res["lds_kb"] = int(os.environ.get("RR_ROI_LDS_KB", "0"))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
