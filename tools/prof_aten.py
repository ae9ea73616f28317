"""Builder tool: which torch (aten) ops a train step launches besides our kernels — small fills / copies / adds and
where they come from (python stack).  python tools/prof_aten.py [--size 1024] [--batch 8] [--bf16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=1024); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--bf16", action="store_true")
a = ap.parse_args()
from rrnet_amd.configs.rrnet_config import Config as cfg
from rrnet_amd.operators.rrnet_operator import RRNetOperator
cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = a.batch, (a.size, a.size), "hourglass"
cfg.Model.bf16 = a.bf16
cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
torch.manual_seed(cfg.seed)
op = RRNetOperator(cfg); op.model.train()
for i in range(2): op.train_step(i, op.training_loader.get_batch())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    op.train_step(2, op.training_loader.get_batch())
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=40, max_name_column_width=60, max_src_column_width=110))
# the copies, one by one: device time, shapes and the chain of enclosing ops (autograd node / function) that asked for them
ev = [e for e in prof.events() if e.name in ("aten::copy_", "aten::clone", "aten::contiguous") and e.device_time_total > 20]
import collections
small = collections.Counter()
for e in prof.events():
    if e.name == "aten::copy_" and 0 < e.device_time_total <= 20:
        chain, q = [], e.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name[:50]); q = q.cpu_parent
        small[(str(e.input_shapes)[:60], " <- ".join(chain))] += 1
for k, v in small.most_common(15):
    print("small copy x%4d  %s  <- %s" % (v, k[0], k[1]))
for e in sorted(ev, key=lambda e: -e.device_time_total)[:12]:
    chain, q = [], e.cpu_parent
    while q is not None and len(chain) < 6:
        chain.append(q.name[:60]); q = q.cpu_parent
    print("%-14s %8.1f us  %s  <- %s" % (e.name, e.device_time_total, e.input_shapes, " <- ".join(chain)))
# device-side memcpys of the step (hipMemcpyAsync: recorded as their own events) and the op that issued each
mc = collections.Counter()
for e in prof.events():
    if "emcpy" in e.name or "emset" in e.name:
        chain, q = [], e.cpu_parent
        while q is not None and len(chain) < 5:
            chain.append(q.name[:48]); q = q.cpu_parent
        mc[(e.name[:30], " <- ".join(chain))] += 1
for k, v in mc.most_common(20):
    print("memcpy/memset x%4d  %s  <- %s" % (v, k[0], k[1]))
