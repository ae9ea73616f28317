"""Builder tool: which BatchNorm layers of one full-size train step still run rr_bn_bwd_reduce (their sums did not come out of
a data gradient's epilogue): shape, mask source, count.  python tools/list_unfused_bn.py"""
import sys, collections, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrnet_amd import ops
from rrnet_amd.configs.rrnet_config import Config as cfg
from rrnet_amd.operators.rrnet_operator import RRNetOperator
cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = 8, (1024, 1024), "hourglass"
cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
torch.manual_seed(cfg.seed)
op = RRNetOperator(cfg); op.model.train()
cnt = collections.Counter()
orig = ops.bn_bwd_reduce
def wrap(dz, z, y, *a, **k):
    cnt[(tuple(y.shape), z is not None, k.get("mask_scale") is not None)] += 1
    return orig(dz, z, y, *a, **k)
ops.bn_bwd_reduce = wrap
op.train_step(0, op.training_loader.get_batch()); torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[0][0][2] * kv[0][0][3] * kv[0][0][1] * kv[1]):
    print(v, k)
