"""Host-side cost of one train step (builder tool): wall time of the enqueue alone (no synchronisation inside) and a
cProfile of it.  python tools/prof_host.py [--size 1024] [--batch 8]"""
import argparse, cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=1024); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--top", type=int, default=35)
ap.add_argument("--math", default=None, help="cfg.Model.conv_math")
a = ap.parse_args()
from rrnet_amd.configs.rrnet_config import Config as cfg
from rrnet_amd.operators.rrnet_operator import RRNetOperator
cfg.Train.batch_size, cfg.Train.crop_size, cfg.Model.backbone = a.batch, (a.size, a.size), "hourglass"
cfg.Model.conv_math = a.math
cfg.Distributed.gpu_id, cfg.Distributed.rank, cfg.Distributed.world_size = 0, 0, 1
torch.manual_seed(cfg.seed)
op = RRNetOperator(cfg); op.model.train()
b = op.training_loader.get_batch()
fresh = lambda: (b[0], b[1].clone()) + tuple(b[2:])
for i in range(2): op.train_step(i, fresh())
torch.cuda.synchronize()
ts = []
for i in range(3):
    t0 = time.perf_counter(); op.train_step(2 + i, fresh()); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("enqueue ms / step ms:", ["%.1f / %.1f" % (x * 1e3, y * 1e3) for x, y in ts])
pr = cProfile.Profile(); pr.enable(); op.train_step(5, fresh()); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(a.top)
