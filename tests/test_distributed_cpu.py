"""N>1 path on CPU (gloo, world_size 2): flat-buffer broadcast + gradient all-reduce bookkeeping and
the SyncBN statistics exchange arithmetic.  The HIP kernels themselves are not involved (no GPU
here); what is checked is everything around them that only exists when world_size > 1."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rrnet_amd.flat import FlatParams
        torch.manual_seed(100 + rank)                      # ranks start with DIFFERENT weights
        net = nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.Conv2d(8, 5, 1))
        net = net.to(memory_format=torch.channels_last)
        fp = FlatParams(net)
        fp.broadcast(0)                                    # C2: one collective
        w0 = [p.detach().clone() for p in net.parameters()]
        fp.zero_grad()
        for p in net.parameters():                         # rank-dependent "gradients" written into the flat views
            p.grad.copy_(torch.full_like(p, float(rank + 1)))
        scale = fp.all_reduce_grads(chunk_elems=64)        # several slices
        g = [p.grad.detach().clone() * scale for p in net.parameters()]
        # SyncBN exchange arithmetic: [sum, sumsq, count] all-reduced == statistics of the concatenated batch
        gen = torch.Generator().manual_seed(7 + rank)
        n = 5 + 3 * rank                                   # ragged per-rank sample counts
        x = torch.randn(n, 8, generator=gen, dtype=torch.float64) * (rank + 1) + rank
        sums = torch.cat([x.sum(0), (x * x).sum(0), torch.tensor([float(n)], dtype=torch.float64)])
        dist.all_reduce(sums)
        cnt = sums[16]
        mean = sums[:8] / cnt
        var = sums[8:16] / cnt - mean * mean
        xs = [torch.zeros(5 + 3 * r, 8, dtype=torch.float64) for r in range(world)]
        dist.all_gather_object(obj := [None] * world, x.numpy().tolist())
        full = torch.tensor(sum(obj, []), dtype=torch.float64)
        q.put((rank, [t.numpy() for t in w0], [t.numpy() for t in g], mean.numpy(), var.numpy(),
               full.mean(0).numpy(), full.var(0, unbiased=False).numpy()))
    finally:
        dist.destroy_process_group()


def test_flat_broadcast_allreduce_and_syncbn_stats_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w_a, g_a, m_a, v_a, fm_a, fv_a), (_, w_b, g_b, m_b, v_b, fm_b, fv_b) = res
    for a, b in zip(w_a, w_b):
        assert np.array_equal(a, b)                        # broadcast made the replicas identical
    for a, b in zip(g_a, g_b):
        assert np.array_equal(a, b)
        assert np.allclose(a, 1.5)                         # mean of the rank gradients 1 and 2 (DDP averages)
    np.testing.assert_allclose(m_a, fm_a, rtol=1e-12)
    np.testing.assert_allclose(v_a, fv_a, rtol=1e-10)
    np.testing.assert_allclose(m_a, m_b)


def test_flat_params_views_and_state_dict_cpu():
    sys.path.insert(0, ROOT)
    from rrnet_amd.flat import FlatParams
    net = nn.Sequential(nn.Conv2d(4, 8, 3, bias=True), nn.BatchNorm2d(8)).to(memory_format=torch.channels_last)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    fp = FlatParams(net)
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd0[k])
    w = net[0].weight
    assert w.shape == (8, 4, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()     # OHWI in memory
    assert fp.numel % 4 == 0 and w.grad.data_ptr() == w._rr_grad.data_ptr()
    w.grad.fill_(2.0)
    fp.zero_grad()
    assert float(fp.grad.abs().sum()) == 0.0
    net.load_state_dict({k: v + 1 if v.is_floating_point() else v for k, v in sd0.items()})
    assert torch.equal(net[0].weight, sd0["0.weight"] + 1)


def _overlap_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rrnet_amd.flat import FlatParams
        torch.manual_seed(5)
        net = nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.Conv2d(8, 16, 3), nn.BatchNorm2d(16),
                            nn.Conv2d(16, 4, 1)).to(memory_format=torch.channels_last)
        fp = FlatParams(net, bucket_elems=64)              # several buckets
        params = list(net.parameters())
        nb = len(fp._bucket_range)
        out = {"nb": nb}
        for trial, skip_last in (("all", False), ("partial", True)):
            fp.zero_grad()
            for i, p in enumerate(params):
                p.grad.copy_(torch.full_like(p, float((rank + 1) * (i + 1))))
            launched = []
            order = list(reversed(params))                 # backward order
            if skip_last:
                order = order[:-2]                         # two parameters never report: the final flush covers them
            for p in order:
                fp.mark_ready(p)
                launched.append(sum(w is not None for w in fp._works))
            scale = fp.all_reduce_grads()
            ok = all(torch.allclose(p.grad * scale, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params))
            out[trial] = (ok, launched)
        # a parameter that reports twice BEFORE its bucket was launched: the bucket is deferred to the end-of-step
        # exchange (no error, correct sums); AFTER the launch it is an error (the bucket already holds the rank sum)
        fp.zero_grad()
        for i, p in enumerate(params):
            p.grad.copy_(torch.full_like(p, float((rank + 1) * (i + 1))))
        b0 = fp._bucket_of[id(params[0])]
        fp.mark_ready(params[0])
        early = fp._works[b0] is not None
        try:
            fp.mark_ready(params[0])
            out["double"] = "raised" if early else "deferred"
            if early:
                out["double"] = "missed"
        except RuntimeError:
            out["double"] = "raised"
        for p in params[1:]:
            fp.mark_ready(p)
        out["deferred_not_launched"] = early or fp._works[b0] is None
        scale = fp.all_reduce_grads()
        again = fp.all_reduce_grads()                      # idempotent within a step: no second sum over the ranks
        out["idempotent"] = (again == scale) and all(
            torch.allclose(p.grad * scale, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params))
        # gradient accumulation: backward passes under no_sync() report nothing
        fp.zero_grad()
        with fp.no_sync():
            for p in params:
                fp.mark_ready(p)
        out["no_sync"] = all(w is None for w in fp._works) and not fp._marked
        for i, p in enumerate(params):
            p.grad.copy_(torch.full_like(p, float((rank + 1) * (i + 1))))
        for p in reversed(params):
            fp.mark_ready(p)
        scale = fp.all_reduce_grads()
        out["no_sync"] = out["no_sync"] and all(
            torch.allclose(p.grad * scale, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_bucketed_overlap_bookkeeping_gloo():
    """FlatParams.mark_ready: buckets are all-reduced as soon as their last parameter reports (backward order),
    unreported parameters are covered by the final flush, a double report defers its bucket (or is an error once
    the bucket was launched), all_reduce_grads is idempotent within a step, no_sync() suppresses the reports;
    result = plain all-reduce."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        out = res[r]
        assert out["nb"] >= 3
        ok, launched = out["all"]
        assert ok and launched[-1] == out["nb"] and launched[0] <= 1 and launched == sorted(launched)
        assert any(a < b for a, b in zip(launched, launched[1:]))          # launched progressively, not at the end
        ok, launched = out["partial"]
        assert ok and launched[-1] < out["nb"]
        assert out["double"] in ("deferred", "raised") and out["deferred_not_launched"]
        assert out["idempotent"] and out["no_sync"]


def _buf_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rrnet_amd import dptrace
        from rrnet_amd.operators.base_operator import broadcast_buffers
        torch.manual_seed(rank)
        net = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 6, 1), nn.BatchNorm2d(6))
        for m in net:
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(); m.running_var.uniform_(0.5, 2.0); m.num_batches_tracked.fill_(1000003 + rank)
        dptrace.reset()
        n = broadcast_buffers(net, 0)
        q.put((rank, n, dptrace.counts(), {k: v.numpy().copy() for k, v in net.state_dict().items() if "running" in k or "tracked" in k}))
    finally:
        dist.destroy_process_group()


def test_buffers_broadcast_in_one_collective():
    """RCCLDataParallel's buffer broadcast (operators/base_operator.py:24 in the reference = DDP's constructor): all BN
    running statistics and the int64 counters arrive exactly, through ONE collective."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_buf_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda r: r[0])
    for p in ps:
        p.join(60)
    assert res[0][1] == res[1][1] == 6
    assert res[0][2] == res[1][2] == {"default": 1}
    for k, v in res[0][3].items():
        np.testing.assert_array_equal(v, res[1][3][k])
        if "tracked" in k:
            assert int(v) == 1000003
