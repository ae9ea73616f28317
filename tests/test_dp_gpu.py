"""GPU: the data-parallel path (SyncBN statistics exchange + flat-gradient all-reduce + fused Adam) with
TWO ranks must reproduce the single-rank full-batch result.  Both ranks share cuda:0 and talk through
gloo (a 1-GPU box cannot host two RCCL ranks); the code path is the one `bench.py --gpus N` runs, only
the backend string differs."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    return SimpleNamespace(num_classes=10, Model=SimpleNamespace(num_stacks=1, backbone="hourglass_tiny",
                           nms_type_for_stage1="nms", nms_per_class_for_stage1=True))


def _build():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import det_fill
    from rrnet_amd.models.centernet import CenterNet
    m = CenterNet(_cfg())
    m.load_state_dict(det_fill({k: tuple(v.shape) for k, v in m.state_dict().items()}, 21))
    return m


def _loss(model, x):
    hms, whs, regs = model(x)
    return (hms[0] ** 2).mean() + (whs[0] ** 2).mean() + (regs[0] ** 2).mean()


def _data():
    g = torch.Generator().manual_seed(3)
    return torch.randn(4, 3, 64, 64, generator=g)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rrnet_amd.flat import FlatAdam, FlatParams
        model = _build().cuda().to(memory_format=torch.channels_last)
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model).train()
        fp = FlatParams(model, bucket_elems=4096)      # many buckets: their all-reduces start during backward
        fp.broadcast(0)
        opt = FlatAdam(fp, lr=1e-3)
        x = _data()[rank * 2:(rank + 1) * 2].cuda()
        opt.zero_grad()
        _loss(model, x).backward()
        launched = sum(w is not None for w in fp._works)
        # all but the buckets holding an autograd-accumulated parameter (the WH head tap weights) start during backward
        assert len(fp._bucket_range) > 4 and launched >= len(fp._bucket_range) - 3, (launched, len(fp._bucket_range))
        scale = fp.all_reduce_grads()
        g = (fp.grad * scale).cpu().numpy()
        rm = model.backbone.pre_layer[1].running_mean.cpu().numpy()
        opt.step()                       # (second all-reduce of an already reduced buffer is avoided below)
        if rank == 0:
            np.savez(out, grad=g, running_mean=rm)
    finally:
        dist.destroy_process_group()


def test_two_rank_syncbn_dp_matches_single_rank(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp.npz")
    mp.spawn(_worker, nprocs=2, args=(2, port, out), join=True)
    z = np.load(out)
    from rrnet_amd.flat import FlatParams
    model = _build().cuda().to(memory_format=torch.channels_last).train()
    fp = FlatParams(model)
    fp.zero_grad()
    _loss(model, _data().cuda()).backward()
    ref = fp.grad.cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(z["grad"] - ref).max() <= 2e-3 * scale, (np.abs(z["grad"] - ref).max(), scale)
    np.testing.assert_allclose(z["running_mean"], model.backbone.pre_layer[1].running_mean.cpu().numpy(), atol=1e-5)


def _run_bench(extra_env, launcher, port):
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    args = ["bench.py", "--backbone", "hourglass_tiny", "--size", "512", "--batch", "4", "--steps", "6", "--warmup", "2",
            "--no-cpu-baseline", "--no-extras", "--no-kernel-timing"]
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + args + ["--gpus", "2"]
    else:
        cmd = [sys.executable] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]                    # rank 0 prints exactly one JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_through_the_launcher():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank, env
    rendezvous on 127.0.0.1), with both ranks on this box's single GPU and gloo as the transport (RR_SINGLE_DEVICE /
    RR_DIST_BACKEND — a 1-GPU box cannot host two RCCL ranks).  Checks the contract fields and times the N>1 path:
    SyncBN exchanges, bucketed gradient all-reduce launched from backward, barrier + max-over-ranks timing.  Two
    ranks share one GPU, so the aggregate rate can at best equal the single-rank rate; gloo's host round trips
    (326 small exchanges per step here) take their toll, hence the loose floor."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    one = _run_bench({}, False, port)
    two = _run_bench({"RR_SINGLE_DEVICE": "1", "RR_DIST_BACKEND": "gloo"}, True, port)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "weak" and two["config"]["global_batch"] == 2 * one["config"]["global_batch"]
    assert two["config"]["parallelism"] == "dp2" and two["steps"] == 6 and two["warmup"] == 2
    assert two["value"] > 0 and abs(two["value"] - 8 * 1e3 / two["ms_per_step"]) < 1e-2 * two["value"]
    print("1 rank: %.1f img/s, 2 ranks on one GPU over gloo: %.1f img/s" % (one["value"], two["value"]))
    assert two["value"] >= 0.35 * one["value"], (one["value"], two["value"])
