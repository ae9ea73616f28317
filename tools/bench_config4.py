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


def run(batch=8, size=1024, steps=2, dcn=True, backbone="hourglass"):
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    saved = (cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, getattr(cfg.Model, "dcn_heads", False),
             getattr(cfg.Model, "dcn_bf16", False))
    cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = batch, (size, size), backbone
    cfg.Model.dcn_heads, cfg.Model.dcn_bf16 = dcn, dcn
    if cfg.Distributed.gpu_id < 0:
        cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    try:
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

        def one():
            nonlocal step_no
            b = batches[step_no % len(batches)]
            op.train_step(step_no, (b[0], b[1].clone()) + tuple(b[2:]))
            step_no += 1
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps
        return {"value": round(batch / t, 4), "unit": "images/sec", "ms_per_step": round(t * 1e3, 2), "steps": steps,
                "dcn_layers": n_dcn,
                "workload": ("RRNet hourglass-104 + %d DCN head layers (bf16 matrix operands, offsets ~ N(0,1)) train step, "
                             "B=%d, %dx%d" % (n_dcn, batch, size, size)) if dcn else
                            "RRNet hourglass-104 train step, B=%d, %dx%d (no DCN heads)" % (batch, size, size)}
    finally:
        (cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone, cfg.Model.dcn_heads, cfg.Model.dcn_bf16) = saved


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--backbone", default="hourglass")
    ap.add_argument("--plain", action="store_true")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    print(json.dumps(run(a.batch, a.size, a.steps, not a.plain, a.backbone)))
