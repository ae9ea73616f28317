"""BASELINE configs[3] at model level: the headline RRNet hourglass-104 train step with the three heads' 3x3
convolutions replaced by ext/dcn `DCN` layers (cfg.Model.dcn_heads; bf16 matrix operands in the deformable kernels),
same batch, same loop as bench.py.  The offset/mask convolutions are NOT left at their zero initialisation (all samples
on integer positions: three of four bilinear corners drop out): bias ~ N(0, 1), weight ~ N(0, 0.01), i.e. fractional
offsets of about a pixel, as in tools/bench_dcn.py.

  python tools/bench_config4.py [--steps 3] [--plain]      (--plain: the same loop without DCN heads, for the delta)
Called by bench.py's `config4` extra; run under rocprofv3 by tools/prof_dcn.sh."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(batch=8, size=1024, steps=5, dcn=True, backbone="hourglass", bf16=None, conv_math=None, curve=None):
    """bf16 (default: with the DCN heads): cfg.Model.bf16 — bf16 matrix operands in EVERY convolution of the backbone and
    the heads (csrc/conv_bf16.hip), not only in the six deformable layers.
    curve: a list that receives the five losses of EVERY step (warm-up included; one host read per step: not for timings)."""
    bf16 = dcn if bf16 is None else bf16
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    saved = (cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, getattr(cfg.Model, "dcn_heads", False),
             getattr(cfg.Model, "dcn_bf16", False), getattr(cfg.Model, "bf16", False))
    cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = batch, (size, size), backbone
    cfg.Model.dcn_heads, cfg.Model.dcn_bf16, cfg.Model.bf16 = dcn, dcn, bf16
    saved_math = getattr(cfg.Model, "conv_math", None)
    cfg.Model.conv_math = conv_math          # "f16x3": split-operand kernels on the large layers (ops.math_mode)
    if cfg.Distributed.gpu_id < 0:
        cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    try:
        import gc
        gc.collect()                                     # an operator of an earlier run in this process (reference cycles): its
        torch.cuda.empty_cache()                         # tensors would count into this run's peak and fragment its pool
        torch.cuda.reset_peak_memory_stats()             # (the allocator peak reported below is THIS run's, not the process's)
        base_gib = torch.cuda.memory_allocated() / 2 ** 30   # what the process still holds from earlier workloads
        if os.environ.get("RR_BENCH_MEMDEBUG") == "1" and base_gib > 0.5:
            live = {}
            for o in gc.get_objects():
                try:
                    if torch.is_tensor(o) and o.is_cuda:
                        st = o.untyped_storage()
                        live[st.data_ptr()] = (st.nbytes(), tuple(o.shape), str(o.dtype))
                except Exception:
                    pass
            for nb, shp, dt in sorted(live.values(), reverse=True)[:25]:
                print("  live before config 4: %8.1f MiB %s %s" % (nb / 2 ** 20, shp, dt), file=sys.stderr, flush=True)
            print("  live before config 4: %.2f GiB allocated, %.2f GiB in python-visible tensors" %
                  (base_gib, sum(v[0] for v in live.values()) / 2 ** 30), file=sys.stderr, flush=True)
        torch.manual_seed(cfg.seed)
        op = RRNetOperator(cfg)                          # the synthetic pool is cached: no host-side regeneration
        op.model.train()
        n_dcn = 0
        g = torch.Generator(device="cuda").manual_seed(5)
        for m in op.model.modules():
            if type(m).__name__ == "DCN":
                m.conv_offset_mask.weight.data.normal_(0, 0.01, generator=g)
                m.conv_offset_mask.bias.data.normal_(0, 1.0, generator=g)
                n_dcn += 1
        batches = [op.training_loader.get_batch() for _ in range(len(op.training_loader))]
        step_no = 2000                                   # past the stage-2 warm-up: all four losses on

        last = [None]
        host = [0.0]
        from rrnet_amd import ops

        def one():
            nonlocal step_no
            b = batches[step_no % len(batches)]
            h0 = time.perf_counter()
            last[0] = op.train_step(step_no, (b[0], b[1].clone()) + tuple(b[2:]))[1]
            host[0] += time.perf_counter() - h0
            step_no += 1
            if curve is not None:
                curve.append([round(float(v.detach()), 5) for v in last[0]])
        for _ in range(3):      # three: the allocator's pool of an in-process run (after other workloads' empty_cache) settles by then
            one()
        torch.cuda.synchronize()
        host[0], ops.SYNC_WAIT_S = 0.0, 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps
        # a timing on non-finite numbers is not a timing (NaN arithmetic draws less power: everything runs faster)
        finite = bool(all(torch.isfinite(v.detach()).all() for v in last[0])) and bool(torch.isfinite(op.optimizer.fp.grad).all()) \
            and bool(torch.isfinite(op.optimizer.fp.flat).all())
        if not finite:
            raise RuntimeError("bench_config4: non-finite losses / gradients / parameters after the timed steps")
        out = {"finite_after_timed_steps": finite, "value": round(batch / t, 4), "unit": "images/sec", "ms_per_step": round(t * 1e3, 2), "steps": steps,
               # host side of a step: time inside train_step() minus the time blocked in its one device -> host read (RoI count)
               "host_enqueue_ms_per_step": round((host[0] - ops.SYNC_WAIT_S) / steps * 1e3, 2),
               "dcn_layers": n_dcn, "dtype": ("f32 (large layers: operands split into two fp16 parts, three MFMA products, fp32 accumulation)"
                                              if conv_math == "f16x3" else "bf16 matrix operands, fp32 accumulation / storage" if bf16 else "f32"),
               "workload": (("RRNet hourglass-104 + %d DCN head layers (offsets ~ N(0,1)) train step, B=%d, %dx%d"
                             % (n_dcn, batch, size, size)) if dcn else
                            "RRNet hourglass-104 train step, B=%d, %dx%d (no DCN heads)" % (batch, size, size))
                           + (", bf16 operands in every convolution" if bf16 else "") + (", conv_math f16x3" if conv_math == "f16x3" else "")}
        ms_ = torch.cuda.memory_stats()
        out["allocator"] = {"reserved_GiB": round(ms_["reserved_bytes.all.current"] / 2 ** 30, 2),
                            "allocated_peak_GiB": round(ms_["allocated_bytes.all.peak"] / 2 ** 30, 2),
                            "held_by_earlier_workloads_GiB": round(base_gib, 2),
                            "segments": ms_["segment.all.current"], "hipMallocs": ms_["num_device_alloc"],
                            "hipFrees": ms_["num_device_free"], "alloc_retries": ms_["num_alloc_retries"]}
        if curve is not None:
            out["timing_valid"] = False      # (a loss read per step synchronises the host with the device)
        if backbone == "hourglass" and size == 1024:
            # 7.02 TFLOP of convolution per image; the six DCN layers replace plain 3x3 layers of the same FLOPs
            peak = 2500.0 if (bf16 or conv_math == "f16x3") else 157.3
            out["step_mfma_frac"] = round(out["value"] * 7.02 / peak, 4)
            out["mfma_peak_tflops"] = peak
        return out
    finally:
        (cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, cfg.Model.dcn_heads, cfg.Model.dcn_bf16,
         cfg.Model.bf16) = saved
        cfg.Model.conv_math = saved_math


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--backbone", default="hourglass")
    ap.add_argument("--plain", action="store_true", help="no DCN heads")
    ap.add_argument("--fp32", action="store_true", help="fp32 convolutions (round 3's config-4 definition: only the DCN layers in bf16)")
    ap.add_argument("--bf16", action="store_true", help="bf16 convolutions even with --plain")
    ap.add_argument("--math", default=None, help="cfg.Model.conv_math: f32 | bf16 | f16x3")
    ap.add_argument("--curve", action="store_true", help="record the five losses of every step (key `loss_curve`): the numerics A/B of a "
                    "switch over a few hundred steps, e.g. RR_BF16_ONLY_ACT=0 / 1 (ADVICE r5)")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    bf16 = True if a.bf16 else (False if a.fp32 else None)
    curve = [] if a.curve else None
    res = run(a.batch, a.size, a.steps, not a.plain, a.backbone, bf16, a.math, curve)
    if curve is not None:
        res["loss_curve"] = curve
        res["env"] = {k: v for k, v in os.environ.items() if k.startswith("RR_")}
    print(json.dumps(res))
