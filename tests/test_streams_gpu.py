"""GPU: the side-stream train step must equal the one-stream train step.

Since round 3 every weight gradient runs on a free-running second HIP stream (rrnet_amd/functional.py:_wgrad_async,
default RR_WGRAD_STREAM=2): it reads the layer input and dy while the main stream goes on, adds into the flat gradient
buffer with float atomics, and is joined only at the end of backward / before a gradient bucket / before Adam.  The
reference (operators/rrnet_operator.py:137-138) runs everything on one ordered stream.  A missing `record_stream`, a
missing event or an in-place update under a reader would be silent noise in the gradients, and the every-kernel audit
cannot see it (it copies operands to the host per call, which serialises the streams).

What "equal" can mean (measured, round 4 — DESIGN §6): the gradient of hourglass-104 at its random initialisation is
ILL-CONDITIONED in the deep levels of the second stack.  With every kernel deterministic (RR_CONV_SPLITK=0), scaling the
input image by (1 + 1e-7 N(0,1)) changes the weight gradients of `hgs.1.low2.low2.low2.low2.*` by 15-30 % of their scale
(losses unchanged to 1e-7).  The default path's split-K layers (<= 32x32: fp32 atomics, order varies run to run) inject
exactly such rounding-sized noise, so two runs of the DEFAULT one-stream step already differ by ~10 % there.  Hence:

  * the tight A/B runs with RR_CONV_SPLITK=0: forward and data gradients are then bit-reproducible and the only freedom
    left is the arrival order of the weight-gradient atomics themselves — side stream (x5, allocator-warm) and side
    stream under stress must match the one-stream step within BOUND = 1e-5 of EVERY parameter's gradient scale;
  * the default path (split-K on) is checked on the parameters that are well-conditioned there (one-stream run-to-run
    spread <= BOUND): same bound, and the ill-conditioned rest must not be worse than the one-stream arm's own spread;
  * stress arm: RR_WGRAD_STRESS puts a ~1 ms spin kernel in front of every side-stream launch — the side stream falls
    hundreds of milliseconds behind and its operands have long been released by the main stream when they are read.  A
    lifetime / ordering bug turns into a wrong gradient instead of passing by luck;
  * the test's own sensitivity: with the joins dropped the comparison fails;
  * the same pair for the DCN backward's two streams (RR_DCN_BWD_STREAMS) at model level (config 4 heads)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "stream_step_worker.py")
sys.path.insert(0, os.path.join(ROOT, "tests"))

# max |g - g_ref| over a parameter / max |g_ref| over that parameter.  With deterministic forward / data-gradient kernels
# the one-stream arm repeats itself within ~5e-7 (weight-gradient tiles of up to 524 288 fp32 terms arriving in a
# different order); 1e-5 is VERDICT r3's bound.  A real race (a tile of x or dy overwritten, a lost wgrad, Adam before the
# join) moves a parameter's gradient by >= 1e-2 of its scale.
BOUND = 1e-5
# share of parameter elements whose first Adam update (~ lr * sign(g)) may differ between two correct runs: elements with a
# gradient of ~0 flip sign with the atomics' rounding; a gradient that has not landed when Adam runs moves whole tensors
MOVED = 1e-3
DET = {"RR_CONV_SPLITK": "0"}          # deterministic forward / data gradients (no split-K atomics)
KEYS = ("RR_WGRAD_STREAM", "RR_WGRAD_STRESS", "RR_DCN_BWD_STREAMS", "RR_CONV_SPLITK", "RR_DP_FORCE")
ONE = {"RR_WGRAD_STREAM": "0"}      # the reference's shape: every kernel on one ordered stream
MANY = {"RR_WGRAD_STREAM": "2"}     # the product's default: the weight-gradient side stream

def _run(tmp, tag, env, size, batch, repeats, extra=(), must_be_finite=True):
    out = os.path.join(str(tmp), tag)
    os.makedirs(out, exist_ok=True)
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in KEYS:
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, WORKER, "--out", out, "--size", str(size), "--batch", str(batch), "--repeats",
                        str(repeats)] + list(extra), cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (tag, r.stdout[-1500:], r.stderr[-3000:])
    meta = json.load(open(os.path.join(out, "meta.json")))
    assert meta["finite"] or not must_be_finite, tag
    return out, meta


def _load(out, name):
    return torch.from_numpy(np.fromfile(os.path.join(out, name), dtype=np.float32)).cuda()


def _rel(a, b, slices):
    from stream_step_worker import rel_per_param
    return rel_per_param(a, b, [tuple(s) for s in slices])


def _worst(a, b, slices):
    from stream_step_worker import worst_rel
    return worst_rel(a, b, [tuple(s) for s in slices])


def _moved(a, b):
    from stream_step_worker import frac_moved
    return frac_moved(a, b)


def _compare(ref, other, tag, bound=BOUND):
    (ro, rm), (oo, om) = ref, other
    la, lb = np.array(rm["losses"][0]), np.array(om["losses"][0])
    assert np.all(np.abs(la - lb) <= 2e-5 * np.maximum(np.abs(la), 1e-3)), (tag, la, lb)
    g = _worst(_load(oo, "grad.bin"), _load(ro, "grad.bin"), rm["slices"])
    p = _moved(_load(oo, "param.bin"), _load(ro, "param.bin"))
    ba, bb = _load(oo, "buffers.bin"), _load(ro, "buffers.bin")
    bw = float(((ba - bb).abs() / bb.abs().clamp_min(1e-3)).max())
    print("%s: worst gradient difference %.2e of its parameter's scale (%s, scale %.3g); share of parameter "
          "elements whose Adam update differs %.2e; BN running statistics %.2e" % (tag, g[0], rm["names"][g[1]], g[2], p, bw))
    assert g[0] <= bound, (tag, g, rm["names"][g[1]])
    assert p <= MOVED, (tag, p)
    assert bw <= 2e-5, (tag, bw)
    return g[0]


@pytest.mark.parametrize("size,batch", [(256, 2), (1024, 8)])
def test_side_stream_wgrad_equals_one_stream(tmp_path, size, batch):
    one = _run(tmp_path, "one_stream", dict(DET, **ONE), size, batch, 2)
    noise = max(r["grad"][0] for r in one[1]["repeat_vs_first"])
    print("one stream, deterministic forward, run to run: %.2e" % noise)
    assert noise <= BOUND, noise
    side = _run(tmp_path, "side_stream", dict(DET, **MANY), size, batch, 5)
    assert side[1]["env"]["RR_WGRAD_STREAM"] == "2"
    _compare(one, side, "side stream vs one stream (%dx%d, B=%d)" % (size, size, batch))
    for i, r in enumerate(side[1]["repeat_vs_first"]):
        assert r["grad"][0] <= BOUND and r["param"] <= MOVED and r["buffers"] <= 2e-5, (i, r)
    print("side stream, 5 runs: worst run-to-run %.2e" % max(r["grad"][0] for r in side[1]["repeat_vs_first"]))
    stress = _run(tmp_path, "stress", dict(DET, **MANY, RR_WGRAD_STRESS="1"), size, batch, 2)
    _compare(one, stress, "side stream under stress vs one stream")
    assert stress[1]["repeat_vs_first"][0]["grad"][0] <= BOUND


@pytest.mark.parametrize("size,batch", [(1024, 8)])
def test_side_stream_default_path_with_split_k(tmp_path, size, batch):
    """The bench's exact configuration (split-K on).  Its one-stream arm is not reproducible in the ill-conditioned deep
    layers (module docstring), so: parameters whose one-stream run-to-run spread is within BOUND must match within
    3 x BOUND between the arms; over ALL parameters the side-stream arm's distance to the one-stream arm must look like
    the one-stream arm's distance to itself (median and count above BOUND), not worse."""
    one = _run(tmp_path, "one_default", dict(ONE), size, batch, 2)
    side = _run(tmp_path, "side_default", dict(MANY, RR_WGRAD_STRESS="1"), size, batch, 2)
    sl = one[1]["slices"]
    g1, g2 = _load(one[0], "grad.bin"), _load(one[0], "grad2.bin")
    self_rel, _ = _rel(g2, g1, sl)
    cross_rel, _ = _rel(_load(side[0], "grad.bin"), g1, sl)
    # "well-conditioned here" = reproducible to a third of the bound in the one-stream arm itself; on those parameters the
    # arms must agree to 2e-3 of the scale (the spread of a parameter that just made the cut is 1e-5 ... 5e-4; a lost or torn
    # weight gradient is >= 1e-2).  Everywhere else the comparison is between two samples of the same noisy quantity:
    # generous factors, they still separate "same distribution" from "a wrong gradient somewhere" (O(1) on that parameter)
    good = self_rel <= BOUND / 3
    n_good, n = int(good.sum()), good.numel()
    worst_good = float(cross_rel[good].max())
    print("default path: %d of %d parameters reproducible within %.1e in the one-stream arm; on those the side stream "
          "(under stress) differs by at most %.2e; all parameters: median %.2e (one stream vs itself %.2e), above bound "
          "%d (vs %d), worst %.2e (vs %.2e)" % (n_good, n, BOUND / 3, worst_good, float(cross_rel.median()),
                                                  float(self_rel.median()), int((cross_rel > BOUND).sum()),
                                                  int((self_rel > BOUND).sum()), float(cross_rel.max()), float(self_rel.max())))
    assert n_good >= 15, (n_good, n)             # the last stack's heads and their feeders: downstream of the deep levels' noise
    # (two one-stream samples can agree to 3e-6 by chance on a parameter whose real spread is far larger — the median spread over
    # all 519 parameters is 2.5e-2, so about one run in eight has such a falsely "reproducible" parameter among the ~26: measured
    # 4.9e-4 and, once in four full-suite runs of round 5, 6.7e-3.  At most ONE may exceed 2e-3, and it must stay below the
    # 1e-2 a lost or torn weight gradient leaves; a race moves every step's gradient of its parameter, not one sample.)
    over = cross_rel[good] > 2e-3
    assert int(over.sum()) <= 1 and worst_good < 1e-2, (int(over.sum()), worst_good)
    assert float(cross_rel.median()) <= 5 * max(float(self_rel.median()), BOUND)
    assert int((cross_rel > BOUND).sum()) <= int(1.5 * (self_rel > BOUND).sum()) + 30
    assert float(cross_rel.max()) <= max(10 * float(self_rel.max()), 0.5) and float(cross_rel.max()) < 2.0
    la, lb = np.array(one[1]["losses"][0]), np.array(side[1]["losses"][0])
    assert np.all(np.abs(la - lb) <= 2e-5 * np.maximum(np.abs(la), 1e-3)), (la, lb)


def test_stress_mode_detects_a_missing_join(tmp_path):
    """The test's own sensitivity: with every wait ON the side stream dropped (the worker patches
    torch.cuda.Stream.wait_stream; the product code is untouched) and the side stream delayed, Adam reads the gradient
    buffer before the weight gradients have landed — the comparison above must FAIL, i.e. it can see an ordering bug."""
    one = _run(tmp_path, "one_stream", dict(DET, **ONE), 256, 2, 1)
    bad = _run(tmp_path, "nojoin", dict(DET, **MANY, RR_WGRAD_STRESS="1"), 256, 2, 1, ["--sabotage"])
    p = _moved(_load(bad[0], "param.bin"), _load(one[0], "param.bin"))
    print("joins removed: %.1f %% of the parameter elements got a different Adam update" % (100 * p))
    assert p > 10 * MOVED, p


def test_gradient_conditioning_at_initialisation(tmp_path):
    """The measurement the bounds above rest on: deterministic kernels, input images scaled by (1 + 1e-7 N(0,1)).  The deep
    levels' weight gradients move by >= 1e-3 of their scale — and, since every layer upstream of them receives its
    gradient THROUGH them, so do most parameters of the network (the stem included); only the layers downstream of the
    last hourglass (the last stack's heads) respond at rounding size.  This is a property of the reference's network at its
    random initialisation (BatchNorm over few samples in the deep levels, 100+ layers), not of the kernels."""
    out, meta = _run(tmp_path, "perturb", dict(DET, **ONE), 256, 2, 2, ["--perturb", "1e-7"])
    rel, _ = _rel(_load(out, "grad2.bin"), _load(out, "grad.bin"), meta["slices"])
    names = meta["names"]
    worst = int(rel.argmax())
    stem = [i for i, n in enumerate(names) if "pre_layer" in n]
    good = [names[i] for i in torch.nonzero(rel <= 1e-5).flatten().tolist()]
    print("1e-7 input noise: worst gradient change %.2e (%s); stem %.2e; %d of %d parameters above 1e-4; within 1e-5: %s"
          % (float(rel.max()), names[worst], float(rel[stem].max()), int((rel > 1e-4).sum()), rel.numel(), good))
    assert float(rel.max()) > 1e-3 and ".low" in names[worst], (float(rel.max()), names[worst])
    assert int((rel > 1e-4).sum()) > rel.numel() // 2
    assert len(good) >= 10 and all(".hgs." not in n for n in good), good


def test_dcn_backward_streams_equal_one_stream(tmp_path):
    """config 4 heads (six DCN layers, bf16 operands): wgrad beside dgrad on two streams vs one after the other."""
    extra = ["--dcn", "--backbone", "hourglass_tiny"]
    one = _run(tmp_path, "dcn_one", dict(DET, **ONE, RR_DCN_BWD_STREAMS="0"), 256, 2, 2, extra)
    noise = max(r["grad"][0] for r in one[1]["repeat_vs_first"])
    two = _run(tmp_path, "dcn_two", dict(DET, **MANY, RR_DCN_BWD_STREAMS="1", RR_WGRAD_STRESS="1"), 256, 2, 3, extra)
    # the DCN data gradient pre-sums d input in fixed point on chip and adds with float atomics: same spread in both arms
    bound = max(BOUND, 4 * noise)
    print("DCN heads, one stream run to run: %.2e" % noise)
    _compare(one, two, "DCN heads: two streams under stress vs one stream", bound)
    for r in two[1]["repeat_vs_first"]:
        assert r["grad"][0] <= bound, r


def test_config4_bf16_model_with_dcn_heads_streams_equal_one_stream(tmp_path):
    """BASELINE configs[3] as bench.py times it — cfg.Model.bf16 AND the DCN heads together (stream_step_worker.py --dcn
    --bf16): the weight-gradient side stream, the DCN backward's two streams, bf16 images crossing both (the DCN data gradient
    adding into a fan-in buffer other streams' kernels wrote; dY's image read by the weight gradient on the side stream), under
    stress, against the one-stream step.  Heads at 2 x 64 x 64 x 256: the size from which the conv16 kernels take the launches."""
    extra = ["--dcn", "--bf16", "--backbone", "hourglass_tiny"]
    one = _run(tmp_path, "c4_one", dict(DET, **ONE, RR_DCN_BWD_STREAMS="0"), 256, 2, 2, extra)
    noise = max(r["grad"][0] for r in one[1]["repeat_vs_first"])
    print("config 4 (bf16 + DCN heads), one stream run to run: %.2e" % noise)
    bound = max(BOUND, 4 * noise)          # (the DCN data gradient's fixed-point pre-sum + float atomics: same spread in both arms)
    two = _run(tmp_path, "c4_two", dict(DET, **MANY, RR_DCN_BWD_STREAMS="1", RR_WGRAD_STRESS="1"), 256, 2, 3, extra)
    _compare(one, two, "config 4 (bf16 + DCN heads): side streams under stress vs one stream", bound)
    for r in two[1]["repeat_vs_first"]:
        assert r["grad"][0] <= bound, r


def test_side_stream_bf16_model_equals_one_stream(tmp_path):
    """The same A/B for the config-4 precision (cfg.Model.bf16: every convolution on the bf16-operand kernels, incl. the
    parity-class stride-2 data gradients and the bf16 weight-gradient kernel on the side stream), under stress."""
    extra = ["--bf16"]
    one = _run(tmp_path, "bf16_one", dict(DET, **ONE), 256, 2, 2, extra)
    noise = max(r["grad"][0] for r in one[1]["repeat_vs_first"])
    print("bf16 model, one stream run to run: %.2e" % noise)
    assert noise <= BOUND, noise
    side = _run(tmp_path, "bf16_side", dict(DET, **MANY, RR_WGRAD_STRESS="1"), 256, 2, 3, extra)
    _compare(one, side, "bf16 model: side stream under stress vs one stream")
    for r in side[1]["repeat_vs_first"]:
        assert r["grad"][0] <= BOUND, r


def test_side_stream_f16x3_model_equals_one_stream(tmp_path):
    """The same A/B with cfg.Model.conv_math = "f16x3": the split-operand kernels read device words holding their operands'
    maxima (ops.amax_of: reduced on the stream that first needs them, remembered on the tensor, carried from the forward
    to the weight gradient on the side stream).  A maximum read before its reduction has finished gives a wrong operand
    scale — gradients off by powers of two or overflowing to inf — which this comparison and the finite check catch."""
    extra = ["--math", "f16x3"]
    one = _run(tmp_path, "split_one", dict(DET, **ONE), 256, 2, 2, extra)
    noise = max(r["grad"][0] for r in one[1]["repeat_vs_first"])
    print("f16x3 model, one stream run to run: %.2e" % noise)
    assert noise <= BOUND, noise
    side = _run(tmp_path, "split_side", dict(DET, **MANY, RR_WGRAD_STRESS="1"), 256, 2, 3, extra)
    _compare(one, side, "f16x3 model: side stream under stress vs one stream")
    for r in side[1]["repeat_vs_first"]:
        assert r["grad"][0] <= BOUND, r
