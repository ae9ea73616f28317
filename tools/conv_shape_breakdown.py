"""Where the convolution time of one train step goes, per layer shape.

One train step (config 2 fp32, or --bf16: config 4's bf16 operands) with rrnet_amd.ops' convolution entry points wrapped
by a recorder (this tool only: the product is not touched); every distinct call (entry point, shapes, stride, flags) is
then replayed alone on an idle GPU and timed.  Prints count x time per shape, the rate of each against the MFMA peak of
its operand type, and the totals per entry point — the list a kernel change is aimed with.

  python tools/conv_shape_breakdown.py [--bf16] [--dcn] [--top 40] [--json out.json]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rrnet_amd import ops  # noqa: E402

CALLS = {}
ORDER = []
RECORD = [True]


def _sig(v):
    if isinstance(v, torch.Tensor):
        return ("T",) + tuple(v.shape) + (str(v.dtype)[6:],)
    if isinstance(v, (tuple, list)):
        return tuple(_sig(u) for u in v)
    if isinstance(v, (int, float, bool, str)) or v is None:
        return v
    return type(v).__name__


def _wrap(name):
    orig = getattr(ops, name)

    def rec(*a, **kw):
        if not RECORD[0]:
            return orig(*a, **kw)
        key = (name, ops.BF16, _sig(a), tuple(sorted((k, _sig(v)) for k, v in kw.items())))
        e = CALLS.get(key)
        if e is None:
            CALLS[key] = e = {"n": 0, "a": a, "kw": kw, "orig": orig, "bf16": ops.BF16, "name": name}
            ORDER.append(key)
        e["n"] += 1
        return orig(*a, **kw)
    setattr(ops, name, rec)


def _flops(name, a, kw):
    """2 * pixels * K * C * R * S of the call."""
    if name in ("conv_fprop", "conv_fprop_packed"):
        x, w = a[0], a[1]
        stride = a[3] if len(a) > 3 else kw.get("stride", 1)
        pad = a[4] if len(a) > 4 else kw.get("pad", (0, 0))
        if name == "conv_fprop_packed":
            stride, pad = a[2], a[3]
        n, c, h, wd = x.shape
        k, _, r, s = w.shape
        p, q = ops.out_hw(h, wd, r, s, stride, pad[0], pad[1])
        return 2.0 * n * p * q * k * c * r * s, "N%d C%d %dx%d K%d %dx%d s%d" % (n, c, h, wd, k, r, s, stride)
    if name == "conv_dgrad":
        dy, w, xs = a[0], a[1], a[2]
        stride = a[3] if len(a) > 3 else kw.get("stride", 1)
        n, k, p, q = dy.shape
        _, c, r, s = w.shape
        return 2.0 * n * p * q * k * c * r * s, "N%d C%d %dx%d K%d %dx%d s%d" % (n, c, xs[2], xs[3], k, r, s, stride)
    if name == "conv_wgrad":
        x, dy, dw = a[0], a[1], a[2]
        stride = a[3] if len(a) > 3 else kw.get("stride", 1)
        n, c, h, wd = x.shape
        _, k, p, q = dy.shape
        r, s = dw.shape[2], dw.shape[3]
        return 2.0 * n * p * q * k * c * r * s, "N%d C%d %dx%d K%d %dx%d s%d" % (n, c, h, wd, k, r, s, stride)
    if name == "conv_wgrad_packed":
        xp, dy, dw = a[0], a[1], a[2]
        n, k, p, q = dy.shape
        return 2.0 * n * p * q * k * dw.shape[1] * dw.shape[2] * dw.shape[3], "packed K%d %dx%d" % (k, p, q)
    return 0.0, "?"


def _time(e, iters):
    fn = lambda: e["orig"](*e["a"], **e["kw"])
    with ops.bf16_scope(e["bf16"]):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        t.record()
        torch.cuda.synchronize()
    return s.elapsed_time(t) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--dcn", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--json")
    a = ap.parse_args()
    from rrnet_amd.configs.rrnet_config import Config as cfg
    from rrnet_amd.operators.rrnet_operator import RRNetOperator
    cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = a.batch, (a.size, a.size), "hourglass"
    cfg.Model.dcn_heads, cfg.Model.dcn_bf16, cfg.Model.bf16 = a.dcn, a.dcn, a.bf16
    cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
    torch.manual_seed(cfg.seed)
    op = RRNetOperator(cfg)
    op.model.train()
    b = op.training_loader.get_batch()
    op.train_step(2000, (b[0], b[1].clone()) + tuple(b[2:]))         # warm (allocator, filter copies)
    torch.cuda.synchronize()
    for nm in ("conv_fprop", "conv_fprop_packed", "conv_dgrad", "conv_wgrad", "conv_wgrad_packed"):
        _wrap(nm)
    op.train_step(2001, (b[0], b[1].clone()) + tuple(b[2:]))
    torch.cuda.synchronize()
    RECORD[0] = False             # the replays below call entry points that call each other (the packed stem)
    rows = []
    for key in ORDER:
        e = CALLS[key]
        fl, desc = _flops(e["name"], e["a"], e["kw"])
        flags = []
        for k, v in e["kw"].items():
            if v is not None and v is not False and k not in ("stride", "pad"):
                flags.append(k)
        if e["name"] == "conv_fprop" and len(e["a"]) > 5 and e["a"][5]:
            flags.append("relu")
        ms = _time(e, a.iters)
        rows.append({"entry": e["name"], "bf16": bool(e["bf16"]), "shape": desc, "flags": ",".join(flags), "calls": e["n"],
                     "ms": ms, "total_ms": ms * e["n"], "tflops": fl / ms / 1e9 if ms > 0 else 0.0, "gflop": fl / 1e9})
    rows.sort(key=lambda r: -r["total_ms"])
    tot = sum(r["total_ms"] for r in rows)
    print("%-18s %-34s %-28s %5s %8s %9s %7s %6s" % ("entry", "shape", "flags", "calls", "ms", "total ms", "TF/s", "cum %"))
    cum = 0.0
    for r in rows[:a.top]:
        cum += r["total_ms"]
        print("%-18s %-34s %-28s %5d %8.3f %9.2f %7.1f %6.1f" % (r["entry"] + ("/bf16" if r["bf16"] else ""), r["shape"], r["flags"].replace("bnsum_z", "z").replace("accumulate", "acc").replace("want_stats", "stats")[:28],
                                                             r["calls"], r["ms"], r["total_ms"], r["tflops"], 100 * cum / tot))
    print("\nper entry point:")
    for nm in sorted({r["entry"] for r in rows}):
        sel = [r for r in rows if r["entry"] == nm]
        t = sum(r["total_ms"] for r in sel)
        f = sum(r["gflop"] * r["calls"] for r in sel)
        print("  %-20s %8.2f ms  %8.1f GFLOP  %7.1f TF/s  (%d shapes, %d calls)" % (nm, t, f, f / t if t else 0, len(sel), sum(r["calls"] for r in sel)))
    print("  %-20s %8.2f ms  %8.1f GFLOP" % ("all", tot, sum(r["gflop"] * r["calls"] for r in rows)))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
