"""operators/base_operator.py:11-51 of the reference.  `DistributedDataParallel(model, ...)` is
replaced by RCCLDataParallel: parameters and gradients live in flat HBM buffers
(rrnet_amd.flat), the initial broadcast is one collective and the gradient exchange is a few
large RCCL all-reduces over xGMI issued from the optimizer step."""
import os
import random

import torch
import torch.nn as nn

from rrnet_amd.flat import FlatParams


def broadcast_buffers(module, src=0):
    """All buffers of `module` from rank `src` in one collective: packed into one float64 tensor (exact for float32
    statistics and for the int64 `num_batches_tracked` counters below 2^53), broadcast, unpacked in place."""
    from rrnet_amd import dptrace
    bufs = [b for b in module.buffers() if b.numel() > 0]
    if not bufs:
        return 0
    flat = torch.cat([b.detach().reshape(-1).to(torch.float64) for b in bufs])
    dptrace.record("default", "broadcast", flat.numel(), "buffers")
    torch.distributed.broadcast(flat, src)
    off = 0
    with torch.no_grad():
        for b in bufs:
            n = b.numel()
            b.copy_(flat[off:off + n].view(b.shape).to(b.dtype))
            off += n
    return len(bufs)


class RCCLDataParallel(nn.Module):
    """Keeps DDP's surface used by the reference (`self.model(x)`, `self.model.module`,
    state_dict of `.module`)."""

    def __init__(self, module, flat=None):
        super().__init__()
        self.module = module
        self.flat = flat if flat is not None else FlatParams(module)
        self.flat.broadcast(0)
        # buffers (BN running statistics and counters) start identical on all ranks — rank 0's — through ONE broadcast
        # of a packed fp64 image (490 single-buffer broadcasts in round 2; fp64 holds the int64 counters exactly)
        from rrnet_amd import dptrace
        if dptrace.dp_active():
            broadcast_buffers(module, 0)

    def forward(self, *a, **kw):
        return self.module(*a, **kw)


class BaseOperator(object):
    def __init__(self, cfg, model, lr_sch=None, flat=None):
        self.cfg = cfg
        random.seed(cfg.seed)
        torch.manual_seed(cfg.seed)
        torch.cuda.manual_seed(cfg.seed)
        self.model = RCCLDataParallel(model, flat)
        self.lr_sch = lr_sch

    def criterion(self, outs, labels):
        raise NotImplementedError

    def training_process(self):
        raise NotImplementedError

    def evaluation_process(self):
        raise NotImplementedError

    @staticmethod
    def save_ckp(models, step, path):
        """Same file format as the reference: state_dict of the bare module, reference key names."""
        sd = {k: v.detach().cpu().contiguous() for k, v in models.state_dict().items()}
        torch.save(sd, os.path.join(path, 'ckp-{}.pth'.format(step)))
