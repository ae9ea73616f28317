"""Flat parameter / gradient storage + fused Adam + RCCL gradient exchange.

MI355X-first replacement for what the reference gets from DistributedDataParallel + optim.Adam
(operators/base_operator.py:24, operators/rrnet_operator.py:29,137-138): all parameters live in
ONE fp32 buffer and all gradients in another (each tensor 16-byte aligned, conv weights kept in
their OHWI physical layout), so that
  * wgrad / BN-backward kernels accumulate straight into the gradient buffer (no per-tensor
    gradient allocation, no autograd accumulation pass),
  * the data-parallel exchange is a handful of large all-reduces over slices of that buffer
    (RCCL over xGMI; 765 MB for hourglass-104) instead of DDP's 25 MB buckets + graph walk,
  * Adam is a single kernel launch over 191 M elements instead of 519 small ones.
`module.state_dict()` / `load_state_dict()` keep working: parameters stay nn.Parameters whose
`.data` are views into the flat buffer.
"""
import os

import torch
import torch.distributed as dist

from rrnet_amd import dptrace, ops


class FlatParams:
    """One flat fp32 buffer for the parameters and one for the gradients.  Data-parallel exchange: the gradient
    buffer is cut into buckets (contiguous parameter ranges); the backward kernels report every parameter whose
    gradient is complete (`mark_ready`, called by the autograd nodes of rrnet_amd.functional right after their
    wgrad / bias / BN-affine kernels), and a bucket's all-reduce is launched — asynchronously, on RCCL's stream — the
    moment its last parameter reports, i.e. while backward is still running on the earlier layers.
    `all_reduce_grads` (called from the optimizer step) launches whatever is left — buckets holding one of the few
    parameters whose gradient autograd accumulates itself (the WH head's 17-tap weights) — and waits."""

    def __init__(self, module, bucket_elems=16 * 1024 * 1024):
        params = [p for p in module.parameters() if p.requires_grad]
        assert params, "no trainable parameters"
        dev = params[0].device      # HBM in production; CPU tensors are accepted so that the data-parallel
                                    # bookkeeping can be exercised with the gloo backend in the CPU test-suite
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params = params
        self.numel = total
        for p, o in zip(params, offs):
            n = p.numel()
            if p.dim() == 4:          # physical OHWI, logical [K,C,R,S]
                k, c, r, s = p.shape
                view = self.flat[o:o + n].view(k, r, s, c).permute(0, 3, 1, 2)
                gview = self.grad[o:o + n].view(k, r, s, c).permute(0, 3, 1, 2)
            else:
                view = self.flat[o:o + n].view(p.shape)
                gview = self.grad[o:o + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = gview
            p._rr_grad = gview
            p._rr_flat = self
        # buckets: consecutive parameters, closed once they hold >= bucket_elems elements
        self._bucket_of, self._bucket_range, self._bucket_need = {}, [], []
        start, need = 0, 0
        for i, (p, o) in enumerate(zip(params, offs)):
            self._bucket_of[id(p)] = len(self._bucket_range)
            need += 1
            end = offs[i + 1] if i + 1 < len(params) else total
            if end - start >= bucket_elems or i + 1 == len(params):
                self._bucket_range.append((start, end))
                self._bucket_need.append(need)
                start, need = end, 0
        self._index_of = {id(p): i for i, p in enumerate(params)}
        self._offs = {id(p): o for p, o in zip(params, offs)}
        # flipped / transposed copies of the 4-D filters for the stride-1 data gradients (rr_conv_dgrad_s1*): one flat
        # buffer with the parameters' offsets, filled by ONE launch per optimizer step instead of one launch per layer
        # inside backward (lazily allocated at the first data gradient; RR_WT_CACHE=0: per-layer launches as before)
        self.wt_flat = None
        self.w16_flat = self.wt16_flat = None       # bf16 copies (plain / flipped) for the bf16-operand kernels, on demand
        self._wt_table = None
        self._wt_tabled = set()
        self.epoch = 0                              # optimizer steps taken through the flat buffer (FlatAdam.step): cache key for
                                                    # per-step quantities derived from the parameters (functional._wamax_attach)
        self._wt_version = None
        self._wt_pver = {}                          # id(parameter) -> its own version counter at the last fill
        self._wt_enabled = True and dev.type == "cuda"
        self.overlap = os.environ.get("RR_DP_OVERLAP", "1") != "0"
        self._pg = None
        self._begin_step()

    def _build_wt_table(self):
        import numpy as np
        rows = []
        self._wt_tabled = set()             # ids of the filters the table covers; the rest keep the per-layer launch
        for p in self.params:
            if p.dim() != 4:
                continue
            k, c, r, s = p.shape
            if k > 65535 or r * s >= 2048 or (k + 31) // 32 > 1023 or (c + 31) // 32 > 1023:
                continue
            self._wt_tabled.add(id(p))
            o = self._offs[id(p)]
            kt, ct = (k + 31) // 32, (c + 31) // 32
            tap, ki, ci = np.meshgrid(np.arange(r * s), np.arange(kt), np.arange(ct), indexing="ij")
            n = tap.size
            t = np.empty((n, 4), dtype=np.int32)
            t[:, 0] = o
            t[:, 1] = ((r * s) << 16) | k
            t[:, 2] = c
            t[:, 3] = (tap.reshape(-1) << 20) | (ki.reshape(-1) << 10) | ci.reshape(-1)
            rows.append(t)
        tab = np.concatenate(rows) if rows else np.zeros((0, 4), dtype=np.int32)
        self._wt_table = torch.from_numpy(tab).to(self.flat.device)

    def refresh_wt(self):
        """Refill the flipped-filter cache from the current parameters (one launch).  Called by FlatAdam.step right after
        the update kernel, and by wt_view when the parameters were written some other way: an in-place write to the flat
        buffer (broadcast) or to a parameter (load_state_dict's param.copy_) moves that tensor's version counter, which is
        compared at every use.  A write through `p.data` is invisible to both counters: call invalidate_wt() after one."""
        if not self._wt_enabled or self.wt_flat is None:
            return
        from rrnet_amd import _C
        if self.w16_flat is not None:
            _C.check(_C.fn("rr_weight_flip_transpose_batch_bf16")(_C.ptr(self.flat), _C.ptr(self.wt_flat), _C.ptr(self.w16_flat),
                                                                  _C.ptr(self.wt16_flat), _C.ptr(self._wt_table),
                                                                  self._wt_table.shape[0], _C.stream()),
                     "rr_weight_flip_transpose_batch_bf16")
        else:
            _C.check(_C.fn("rr_weight_flip_transpose_batch")(_C.ptr(self.flat), _C.ptr(self.wt_flat), _C.ptr(self._wt_table),
                                                             self._wt_table.shape[0], _C.stream()), "rr_weight_flip_transpose_batch")
        self._wt_version = self.flat._version
        self._wt_pver = {id(p): p._version for p in self.params if p.dim() == 4}

    def invalidate_wt(self):
        """Mark every per-step quantity derived from the parameters stale: the flipped-filter cache (the next data gradient
        refills it) and, through `epoch`, what the autograd nodes remember per optimizer step (the split-operand kernels'
        filter maxima, functional._wamax_attach).  The one call to make after writing parameters through `p.data`."""
        self._wt_version = None
        self.epoch += 1

    def _wt_ready(self, p, bf16):
        if not self._wt_enabled or p.dim() != 4 or id(p) not in self._offs:
            return False
        if self.wt_flat is None:
            self._build_wt_table()
            self.wt_flat = torch.empty_like(self.flat)
            self._wt_version = None
        if bf16 and self.w16_flat is None:
            self.w16_flat = torch.empty(self.numel, dtype=torch.bfloat16, device=self.flat.device)
            self.wt16_flat = torch.empty(self.numel, dtype=torch.bfloat16, device=self.flat.device)
            self._wt_version = None
        if id(p) not in self._wt_tabled:      # a shape the batch kernel's table cannot encode: nothing ever fills its slice
            return False
        if self._wt_version != self.flat._version or self._wt_pver.get(id(p)) != p._version:   # written since the last fill
            self.refresh_wt()
        return True

    def wt_view(self, p):
        """Flat fp32 tensor [k*c*r*s] holding wt[c][R-1-r][S-1-s][k] of the 4-D parameter p, or None (cache off, p not a
        filter of this buffer)."""
        if not self._wt_ready(p, False):
            return None
        o = self._offs[id(p)]
        return self.wt_flat[o:o + p.numel()]

    def w16_views(self, p):
        """(bf16 copy of the filter [k][r][s][c], bf16 copy of its flipped / transposed form) for the bf16-operand kernels,
        or (None, None)."""
        if not self._wt_ready(p, True):
            return None, None
        o = self._offs[id(p)]
        return self.w16_flat[o:o + p.numel()], self.wt16_flat[o:o + p.numel()]

    def _begin_step(self):
        nb = len(self._bucket_range)
        self._ready = [0] * nb
        self._works = [None] * nb
        self._marked = set()
        self._late = set()         # buckets that must wait for the end-of-step exchange (a parameter reported twice)
        self._reduced = False      # this step's gradient exchange has completed

    def _distributed(self):
        return dptrace.dp_active()

    def _group(self):
        """Own communicator (= own RCCL stream) for the gradient buckets: on the default one a 64 MB bucket would sit in
        front of the next SyncBN statistics all-reduce and stall the compute stream for its whole duration.  Created at
        the first use; every rank gets here at the same point of the program."""
        if self._pg is None:
            self._pg = dist.new_group()
            # RCCL builds a communicator lazily at its first collective: run that first collective HERE, on the main
            # thread and outside backward, so that the bucket launched from an autograd worker thread in the middle of
            # backward (interleaved with SyncBN exchanges on the default communicator) finds it ready
            warm = torch.zeros(1, dtype=torch.float32, device=self.grad.device)
            dptrace.record("grads", "all_reduce", 1, "communicator warm-up")
            dist.all_reduce(warm, group=self._pg)
            if warm.is_cuda:
                torch.cuda.current_stream(warm.device).synchronize()
        return self._pg

    def no_sync(self):
        """Context manager for gradient accumulation (DistributedDataParallel.no_sync): backward passes inside it
        only accumulate locally; the exchange belongs to the last backward of the step, run outside."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._accumulating = getattr(self, "_accumulating", False), True
            try:
                yield
            finally:
                self._accumulating = prev
        return ctx()

    def mark_ready(self, p):
        """The gradient of `p` is complete for this step (every kernel that adds into it has been enqueued)."""
        if not self.overlap or not self._distributed() or getattr(self, "_accumulating", False):
            return
        key = id(p)
        b = self._bucket_of[key]
        if key in self._marked:
            # a second report in one step (a second backward without no_sync(), shared weights).  If the bucket's
            # all-reduce has not started it is simply left to the end-of-step exchange; if it has, the bucket holds
            # the sum over ranks already and adding a local gradient on top cannot be undone
            if self._works[b] is not None:
                raise RuntimeError("FlatParams.mark_ready: parameter #%d %s reported its gradient again after its "
                                   "bucket's all-reduce was launched; wrap all but the last backward of a step in "
                                   "`with flat.no_sync():` (gradient accumulation) or set RR_DP_OVERLAP=0"
                                   % ([id(q) for q in self.params].index(key), tuple(p.shape)))
            self._late.add(b)
            return
        self._marked.add(key)
        self._ready[b] += 1
        dptrace.mark("param %d bucket %d" % (self._index_of[key], b))
        if self._ready[b] == self._bucket_need[b] and self._works[b] is None and b not in self._late:
            o0, o1 = self._bucket_range[b]
            dptrace.record("grads", "all_reduce", o1 - o0, "bucket %d" % b)
            # the collective is ordered behind the stream it is launched from; parameters of this bucket may have been
            # written on another one (weight gradients run on a side stream): wait for those first
            ops.join_aux_streams(self.grad.device)
            self._works[b] = dist.all_reduce(self.grad[o0:o1], group=self._group(), async_op=True)

    def zero_grad(self):
        self._begin_step()
        self.grad.zero_()
        for p in self.params:        # keep .grad pointing at the flat views (set_to_none would drop them)
            if p.grad is None or p.grad.data_ptr() != p._rr_grad.data_ptr():
                p.grad = p._rr_grad

    def broadcast(self, src=0):
        """Initial parameter broadcast (C2 in SURVEY §2.2): one collective."""
        if self._distributed():
            dptrace.record("default", "broadcast", self.flat.numel(), "parameters")
            dist.broadcast(self.flat, src)
            self._group()          # the buckets' communicator is set up here, on the main thread, before any backward

    def all_reduce_grads(self, chunk_elems=64 * 1024 * 1024):
        """Sum-all-reduce of the flat gradient in a few large slices (C3).  Returns the factor the
        optimizer must scale the gradient with (DDP averages)."""
        if not self._distributed():
            return 1.0
        if self._reduced:          # idempotent within a step: a second call must not sum the ranks twice
            return 1.0 / dist.get_world_size()
        ops.join_aux_streams(self.grad.device)
        for b, (o0, o1) in enumerate(self._bucket_range):      # buckets that were not complete during backward
            if self._works[b] is None:
                dptrace.record("grads", "all_reduce", o1 - o0, "bucket %d (end of step)" % b)
                self._works[b] = dist.all_reduce(self.grad[o0:o1], group=self._group(), async_op=True)
        for w in self._works:
            w.wait()
        self._works = [None] * len(self._bucket_range)
        self._reduced = True
        return 1.0 / dist.get_world_size()


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (defaults of operators/rrnet_operator.py:29: betas (0.9,0.999),
    eps 1e-8, no weight decay, no amsgrad) as one fused kernel over FlatParams, preceded by the
    data-parallel gradient all-reduce.  Subclasses Optimizer so lr schedulers drive `param_groups`."""

    def __init__(self, module_or_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.fp = module_or_flat if isinstance(module_or_flat, FlatParams) else FlatParams(module_or_flat)
        super().__init__(self.fp.params, dict(lr=lr, betas=betas, eps=eps))
        self.exp_avg = torch.zeros_like(self.fp.flat)
        self.exp_avg_sq = torch.zeros_like(self.fp.flat)
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        self.fp.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        ops.join_aux_streams(self.fp.grad.device)     # weight gradients still in flight on their side stream
        scale = self.fp.all_reduce_grads()
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'][0], g['betas'][1],
                      g['eps'], self.step_count, scale)
        self.fp.epoch += 1
        self.fp.refresh_wt()          # the update kernel wrote through raw pointers: refill the flipped-filter cache now
