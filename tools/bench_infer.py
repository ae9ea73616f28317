"""Inference-only post-process throughput (BASELINE.json configs[4]):
heat-map decode + stage-1 NMS + RoIAlign + re-regression head + Soft-NMS on a stream of 1920x1080 frames
(stride-4 maps 270x480), K=1500 candidate boxes per frame, one MI355X.

  python tools/bench_infer.py [--frames 10000] [--batch 128] [--pool 128] [--cpu-frames 3]

Inputs (stage-1 head outputs + backbone feature, recipe in rrnet_amd/datasets/synthetic.py
`synth_head_outputs`) are generated on the device and stay resident in HBM: `--pool` distinct frames, cycled
until `--frames` frames have been processed.  Prints ONE JSON line: frames/sec, candidate boxes/sec
(= frames/sec * K: every candidate is decoded, NMS'd, pooled, re-regressed and Soft-NMS'd), output
boxes/frame, per-stage GPU time, and `cpu_baseline` = the CPU oracle (oracle/infer.py) on `--cpu-frames`
frames of the same pool plus the CPU Soft-NMS alone (the reference's own compiled cpu_soft_nms from
oracle/_ref when present, else the C restatement) on the same segments.
Replicas only across GPUs: frames are independent, there is no collective on this path."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--pool", type=int, default=512, help="distinct resident frames (default: 4 batches of 128, 68 GB of feature maps)")
    ap.add_argument("--hf", type=int, default=270)
    ap.add_argument("--wf", type=int, default=480)
    ap.add_argument("--k", type=int, default=1500)
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--stages", action="store_true", help="time every stage with events (adds syncs)")
    return ap.parse_args()


def cpu_baseline(head_sd, pool, n_frames, k):
    from oracle import infer as oinfer, model as omodel, nms as onms, ops as oops
    P = omodel.Params({"head_detector." + kk: v.cpu() for kk, v in head_sd.items()}, training=False)
    hm, wh, off, feat = pool
    t_full, t_nms, n_boxes, nms_rows = 0.0, 0.0, 0, 0
    mod = onms.load_reference_cpu_nms()
    ref = mod.cpu_soft_nms if mod is not None else None
    for f in range(n_frames):
        args = [t[f:f + 1].cpu() for t in (hm, wh, off, feat)]
        t0 = time.perf_counter()
        out = oinfer.postprocess_frame(P, *args, k=k, relu_feat=False)
        t_full += time.perf_counter() - t0
        n_boxes += out.shape[0]
        # Soft-NMS alone on the segments this frame hands to ext/nms: rebuild the stage-2 boxes once more
        with torch.no_grad():
            bb = oops.transform_bbox(args[0], args[1], args[2], k)
            kept = oops.stage1_nms(bb[0], 'nms', True)
        boxes = kept.numpy().copy()
        boxes[:, :4] *= 4
        for c in np.unique(boxes[:, 5]):
            seg = np.ascontiguousarray(boxes[boxes[:, 5] == c][:, :5], dtype=np.float32)
            t0 = time.perf_counter()
            if ref is not None:
                ref(seg, np.float32(0.5), np.float32(0.7), np.float32(0.1), np.uint8(2))
            else:
                onms.cpu_soft_nms(seg, 0.5, 0.7, 0.1, 2)
            t_nms += time.perf_counter() - t0
            nms_rows += seg.shape[0]
    return {"value": round(n_frames * k / t_full, 1), "unit": "boxes/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "oracle/infer.py post-process of %d frames %dx%d K=%d: %.2f s/frame, %.0f output boxes/frame"
                      % (n_frames, hm.shape[2], hm.shape[3], k, t_full / n_frames, n_boxes / n_frames),
            "ext_nms_only": {"value": round(nms_rows / t_nms, 1), "unit": "boxes/sec", "cores": 1,
                             "kind": "reference" if ref is not None else "port",
                             "sample": "cpu_soft_nms (Nt .7, thr .1, gaussian) per class on the same frames: %.2f ms/frame"
                                       % (1e3 * t_nms / n_frames)}}


def run(frames=10000, batch=128, pool_frames=128, hf=270, wf=480, k=1500, cpu_frames=3):
    """-> the result dict (also used by bench.py for its `config5` key)."""
    from rrnet_amd import inference, ops
    from rrnet_amd.datasets.synthetic import synth_head_outputs
    from rrnet_amd.detectors.fasterrcnn_detector import FasterRCNNDetector
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(219)
    head = FasterRCNNDetector().to(dev).eval()
    head_sd = {kk: v.detach().clone() for kk, v in head.state_dict().items()}
    nb = pool_frames // batch
    assert nb >= 1, "--pool must be >= --batch"
    # the pool is generated one batch at a time (own seed each): 128 frames of 270x480x256 fp32 are 17 GB, and the
    # generator's NCHW temporaries would triple the footprint of a 512-frame pool made in one piece
    batches, pool = [], None
    for i in range(nb):
        chunk = synth_head_outputs(batch, hf, wf, seed=219 + i, device=dev)
        if i == 0 and cpu_frames > 0:
            pool = tuple(t[:cpu_frames].clone() for t in chunk)
        batches.append(tuple(ops.to_nhwc(t) for t in chunk))
        del chunk
        torch.cuda.empty_cache()

    def step(bt):
        return inference.refine_frames(bt[0], bt[1], bt[2], bt[3], head, k=k, relu_feat=False)

    for i in range(2):
        out, fo = step(batches[i % nb])
    torch.cuda.synchronize()
    n_iter = max(frames // batch, 1)
    kept = 0
    t0 = time.perf_counter()
    for i in range(n_iter):
        out, fo = step(batches[i % nb])
        kept += out.shape[0]
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    frames = n_iter * batch
    res = {"metric": "boxes/sec (decode + re-regression + Soft-NMS)", "value": round(frames * k / t, 1),
           "unit": "boxes/sec", "frames_per_sec": round(frames / t, 1), "frames": frames, "n_gpus": 1,
           "ms_per_batch": round(1e3 * t / n_iter, 3), "output_boxes_per_frame": round(kept / frames, 1),
           "higher_is_better": True, "dtype": "f32", "data": "synthetic", "vs_baseline": None,
           "config": {"workload": "RRNet inference-only post-process, %d frames of 1920x1080 (maps %dx%d), K=%d, "
                                  "batch %d, pool of %d resident frames" % (frames, hf, wf, k, batch, pool_frames)}}
    if cpu_frames > 0:
        res["cpu_baseline"] = cpu_baseline(head_sd, pool, cpu_frames, k)
    return res


def main():
    a = parse()
    torch.cuda.set_device(0)
    print(json.dumps(run(a.frames, a.batch, a.pool, a.hf, a.wf, a.k, a.cpu_frames)))


if __name__ == "__main__":
    main()
