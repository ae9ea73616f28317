"""Host-side logic that needs no GPU: the collective trace / counters (rrnet_amd/dptrace.py), the launch-sampling rule of
bench.py's kernel timer, the zero pool's slice independence, the defaults of the gradient-link bookkeeping."""
import torch


def test_dptrace_counts_and_sequences():
    from rrnet_amd import dptrace
    dptrace.reset()
    was = dptrace.ENABLED
    try:
        dptrace.ENABLED = True
        dptrace.record("default", "all_reduce", 513, "syncbn_fwd")
        dptrace.mark("param 3 bucket 0")
        dptrace.record("grads", "all_reduce", 4096, "bucket 0")
        dptrace.record("default", "all_reduce", 1024, "syncbn_bwd")
        assert dptrace.counts() == {"default": 2, "grads": 1}
        assert [e[:3] for e in dptrace.sequence("default")] == [("default", "all_reduce", 513), ("default", "all_reduce", 1024)]
        assert len(dptrace.sequence()) == 3 and len(dptrace.EVENTS) == 4          # the mark is kept, but is no collective
        dptrace.ENABLED = False
        dptrace.record("default", "broadcast", 7)
        assert dptrace.counts()["default"] == 3 and len(dptrace.EVENTS) == 4       # counters always on, trace only on request
    finally:
        dptrace.ENABLED = was
        dptrace.reset()


def test_kernel_timer_samples_every_nth_launch_of_a_name(monkeypatch):
    from rrnet_amd import ops

    class FakeEvent:
        def __init__(self, enable_timing=True):
            pass

        def record(self):
            pass

    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    t = ops.KernelTimer(only={"a"}, every=4)
    ran = []
    for i in range(10):
        t.launch("a", 1.0, lambda i=i: ran.append(("a", i)))
        t.launch("b", 1.0, lambda i=i: ran.append(("b", i)))
    assert len(ran) == 20                                   # every launch runs
    assert len(t.records) == 3                              # launches 0, 4, 8 of "a" carry events; "b" is not timed at all
    assert all(r[0] == "a" for r in t.records)


def test_zero_pool_slices_are_independent_tensors():
    from rrnet_amd.ops import _ZeroPool
    p = _ZeroPool()
    a, b = p.take(5, torch.device("cpu")), p.take(6, torch.device("cpu"))
    va, vb = a._version, b._version
    a[0] = 3.0
    assert a._version == va + 1 and b._version == vb        # an in-place write does not touch the sibling's version counter
    assert float(b.abs().sum()) == 0.0 and b.data_ptr() - a.data_ptr() == 48       # 16-byte aligned, disjoint


def test_gradient_link_defaults():
    from rrnet_amd import functional as RF, ops
    link = ops.BnLink()
    assert link.sums is None and link.dz is None and link.consumers == 0 and not link.use_z and not link.relu_bias
    acc = RF.GradAcc()
    assert acc.buf is None and acc.pending == 0 and acc.link is None


def test_conv_math_switch_and_split_layer_policy():
    """cfg.Model.conv_math -> ops.math_mode, and the layer policy of the split-operand arithmetic (ops._bf16_ok): host logic
    only — no kernel is launched."""
    from types import SimpleNamespace
    import pytest
    import torch
    from rrnet_amd import ops
    assert ops.math_mode(SimpleNamespace()) == ops.MATH_F32
    assert ops.math_mode(SimpleNamespace(bf16=True)) == ops.MATH_BF16
    assert ops.math_mode(SimpleNamespace(bf16=True, conv_math="f16x3")) == ops.MATH_F16X3      # conv_math wins
    assert ops.math_mode(SimpleNamespace(conv_math="f32")) == ops.MATH_F32
    assert ops.math_mode(SimpleNamespace(conv_math=None, bf16=False)) == ops.MATH_F32
    with pytest.raises(ValueError):
        ops.math_mode(SimpleNamespace(conv_math="fp8"))
    t = torch.empty(4)
    saved = ops.BF16
    try:
        ops.BF16 = ops.MATH_F32
        assert ops._bf16_ok(256, 256, 3, 3, t, pixels=1 << 20) == 0
        ops.BF16 = ops.MATH_BF16
        assert ops._bf16_ok(256, 256, 3, 3, t) == ops.MATH_BF16
        assert ops._bf16_ok(254, 256, 3, 3, t) == 0                        # C not a multiple of 4: fp32 kernels
        ops.BF16 = ops.MATH_F16X3
        assert ops._bf16_ok(256, 256, 3, 3, t, pixels=8 * 256 * 256) == ops.MATH_F16X3
        assert ops._bf16_ok(384, 384, 3, 3, t, pixels=8 * 16 * 16) == ops.MATH_F16X3
        assert ops._bf16_ok(512, 512, 3, 3, t, pixels=8 * 8 * 8) == 0      # 512 output pixels: stays on the fp32-MFMA kernel
        assert ops._bf16_ok(256, 256, 1, 1, t, pixels=8 * 256 * 256) == 0  # reduction length 256 < 1024
        assert ops._bf16_ok(256, 36, 3, 3, t, pixels=8 * 256 * 256) == 0   # narrow output (WH head)
        assert ops._bf16_ok(256, 256, 3, 3, t) == 0                        # a caller that does not state its size
        with ops.bf16_scope(ops.MATH_BF16):
            assert ops.BF16 == ops.MATH_BF16
        assert ops.BF16 == ops.MATH_F16X3
    finally:
        ops.BF16 = saved
