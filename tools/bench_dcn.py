"""DCNv2 layer benchmark (BASELINE config 4 shape: 256->256 3x3 deformable head conv on the 256x256 stride-4 map,
B=8): forward with fp32 and with bf16 MFMA operands, backward (all five gradients); TFLOP/s on the algorithmic
2*M*K*9*C FLOPs of the contraction (backward: 2x that — the weight and the column gradient).

  python tools/bench_dcn.py            prints one JSON line (also used by bench.py for its `config4` key)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def _time(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def run(n=8, c=256, h=256, w=256, k=256, reps=5):
    from rrnet_amd import ops
    from rrnet_amd.functional import dcn_v2_conv
    g = torch.Generator(device="cuda").manual_seed(0)
    x = ops.to_nhwc(torch.randn(n, c, h, w, device="cuda", generator=g))
    off = ops.to_nhwc(torch.randn(n, 18, h, w, device="cuda", generator=g))
    mask = ops.to_nhwc(torch.sigmoid(torch.randn(n, 9, h, w, device="cuda", generator=g)))
    wt = ops.to_nhwc(torch.randn(k, c, 3, 3, device="cuda", generator=g) / 48.0)
    flops = 2.0 * n * h * w * k * c * 9
    res = {"workload": "DCNv2 %d->%d 3x3 on B=%d x %dx%d (RRNet head shape), offsets N(0,1)" % (c, k, n, h, w),
           "gflop_fwd": round(flops / 1e9, 1)}
    for bf in (False, True):
        for _ in range(2):
            ops.dcn_fwd(x, off, mask, wt, None, 1, (1, 1), 1, 1, bf16=bf)
        ms = _time(lambda: ops.dcn_fwd(x, off, mask, wt, None, 1, (1, 1), 1, 1, bf16=bf), reps)
        tag = "bf16" if bf else "fp32"
        res["fwd_%s_ms" % tag] = round(ms, 3)
        res["fwd_%s_tflops" % tag] = round(flops / ms / 1e9, 1)
    for bf in (False, True):
        xg, og, mg, wg = [t.clone().requires_grad_() for t in (x, off, mask, wt)]
        y = dcn_v2_conv(xg, og, mg, wg, None, 1, 1, 1, 1, bf16=bf)
        gy = torch.randn_like(y)
        y.backward(gy, retain_graph=True)                          # warm-up
        torch.cuda.synchronize()

        def bwd():
            for t in (xg, og, mg, wg):
                t.grad = None
            y.backward(gy, retain_graph=True)
        ms = _time(bwd, 3)
        tag = "bf16" if bf else "fp32"
        res["bwd_%s_ms" % tag] = round(ms, 3)
        res["bwd_%s_tflops" % tag] = round(2 * flops / ms / 1e9, 1)
        del y, xg, og, mg, wg
    # LDS roofline of the bf16 forward (the kernel furthest from the MFMA and HBM roofs): bytes the
    # workgroups read from LDS per launch, from the kernel's structure — per output pixel, tap and 32-channel chunk:
    # 4 bilinear corners x 128 B from the fp32 input window (the gather), 8 geometry words per staging thread
    # (8 threads x 2 x 16 B), and per workgroup and K-step the MFMA fragments (8 waves x 8 ds_read_b128 x 1 KiB).
    m = n * h * w
    ksteps = (m // 128) * 9 * (c // 32)
    lds_read = m * 9 * c * 4 * 4 + ksteps * (512 * 16 * 4) + ksteps * (8 * 8 * 1024) * (k // 256)
    lds_peak = 150.0                                   # TB/s, ds_read_b64/b128 with every CU streaming (MI355X_MICROARCH.md §LDS)
    ach = lds_read / (res["fwd_bf16_ms"] * 1e-3) / 1e12
    res["roofline"] = {"bound": "lds", "kernel": "dcn_fprop_win_kernel<256,false> (bf16 operands)",
                       "achieved": round(ach, 2), "peak": lds_peak, "unit": "TB/s", "frac": round(ach / lds_peak, 4),
                       "lds_read_gb_per_launch": round(lds_read / 1e9, 2),
                       "mfma_frac_bf16": round(res["fwd_bf16_tflops"] / 2500.0, 4),
                       "note": "neither roof binds: the per-K-step chain gather -> blend -> LDS tile -> barrier -> MFMA "
                               "at one workgroup per CU sets the time (latency-bound)"}
    return res


if __name__ == "__main__":
    torch.cuda.set_device(0)
    print(json.dumps(run()))
