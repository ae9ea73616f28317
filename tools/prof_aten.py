"""Builder tool: which torch (aten) ops a train step launches besides our kernels — small fills / copies / adds and
where they come from (python stack).  python tools/prof_aten.py [--size 1024] [--batch 8]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=1024); ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
from rrnet_amd.configs.rrnet_config import Config as cfg
from rrnet_amd.operators.rrnet_operator import RRNetOperator
cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = a.batch, (a.size, a.size), "hourglass"
cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
torch.manual_seed(cfg.seed)
op = RRNetOperator(cfg); op.model.train()
for i in range(2): op.train_step(i, op.training_loader.get_batch())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    op.train_step(2, op.training_loader.get_batch())
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=40, max_name_column_width=60, max_src_column_width=110))
