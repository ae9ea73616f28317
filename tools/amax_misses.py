"""Which split-operand launches of one f16x3 train step still reduce an operand's maximum themselves (ops.amax_of misses:
no remembered word / other version / other stream), by call site — the list the fused producers are aimed with.
  python tools/amax_misses.py"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rrnet_amd import ops
from rrnet_amd.configs.rrnet_config import Config as cfg
from rrnet_amd.operators.rrnet_operator import RRNetOperator
cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = 8, (1024, 1024), "hourglass"
cfg.Model.conv_math = "f16x3"
cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
torch.manual_seed(1)
op = RRNetOperator(cfg); op.model.train()
b = op.training_loader.get_batch()
op.train_step(2000, (b[0], b[1].clone()) + tuple(b[2:]))
torch.cuda.synchronize()
orig = ops.amax_of
cnt = collections.Counter()
def spy(t):
    hit = getattr(t, "_rr_amax", None)
    sid = torch.cuda.current_stream(t.device).cuda_stream
    miss = not (hit is not None and hit[0] == t._version and (hit[1] == sid or hit[1] is None))
    if miss:
        st = traceback.extract_stack(limit=6)
        why = "none" if hit is None else ("version" if hit[0] != t._version else "stream")
        cnt[(why, tuple(t.shape), " <- ".join("%s:%d" % (f.name, f.lineno) for f in st[:-1][-4:]))] += 1
    return orig(t)
ops.amax_of = spy
op.train_step(2001, (b[0], b[1].clone()) + tuple(b[2:]))
torch.cuda.synchronize()
for k, v in cnt.most_common(25):
    print(v, k)
print("total misses", sum(cnt.values()))
