"""DistributedWrapper(cfg, operator_class).train() / .eval() — the launch surface of the reference's
operators/distributed_wrapper.py:6-69 (scripts/RRNet/train.py:8-9), one process per GPU.

Two ways in:
  * plain `python train.py`: one worker per visible GPU is spawned here and they meet over TCP at
    `cfg.Distributed.dist_url`, as in the reference;
  * under `python -m torch.distributed.run` (RANK / WORLD_SIZE / LOCAL_RANK in the environment) the launcher
    already made the processes: this one takes its place and nothing is spawned.
Backend 'nccl' is RCCL on ROCm: the collectives run over xGMI.  The worker pins its GPU BEFORE the process group
exists (RCCL binds the communicator to the current device)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

_PHASES = {"train": "training_process", "eval": "evaluation_process"}


def _worker(local_rank, wrapper, phase, gpus_here):
    operator = wrapper.init_operator(local_rank, gpus_here, wrapper.cfg)
    getattr(operator, _PHASES[phase])()


class DistributedWrapper(object):
    def __init__(self, cfg, operator_class):
        self.cfg = cfg
        self.operator_class = operator_class

    # ---- reference surface ---------------------------------------------------------------------------------
    def setup_distributed_params(self):
        """Nodes x GPUs-per-node -> world size (cfg.Distributed.world_size holds the node count on entry)."""
        gpus_here = torch.cuda.device_count()
        if gpus_here < 1:
            raise ValueError('[x] Can not get gpu numbers!')
        d = self.cfg.Distributed
        d.ngpus_per_node = gpus_here
        d.world_size = gpus_here * d.world_size

    def init_operator(self, gpu, ngpus_per_node, cfg):
        """Pin the GPU, join the process group, build the operator (model broadcast happens in its constructor)."""
        d = cfg.Distributed
        d.gpu_id = gpu
        print("=> Use GPU: {}".format(gpu))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(gpu)
        if "RANK" in os.environ and d.dist_url == "env://":
            d.rank = int(os.environ["RANK"])
        else:
            d.rank = d.rank * ngpus_per_node + gpu          # node index * GPUs per node + local GPU
        if not dist.is_initialized():
            dist.init_process_group(backend='nccl', init_method=d.dist_url, world_size=d.world_size, rank=d.rank)
        return self.operator_class(cfg)

    def train(self):
        self._run("train")

    def eval(self):
        self._run("eval")

    def dist_training_process(self, gpu, ngpus_per_node, cfg):
        self.init_operator(gpu, ngpus_per_node, cfg).training_process()

    def dist_evaluation_process(self, gpu, ngpus_per_node, cfg):
        self.init_operator(gpu, ngpus_per_node, cfg).evaluation_process()

    # ---- launch --------------------------------------------------------------------------------------------
    def _run(self, phase):
        env = os.environ
        if "RANK" in env and "WORLD_SIZE" in env:            # started by torch.distributed.run
            d = self.cfg.Distributed
            d.world_size = int(env["WORLD_SIZE"])
            d.ngpus_per_node = int(env.get("LOCAL_WORLD_SIZE", 1))
            d.dist_url = "env://"
            _worker(int(env.get("LOCAL_RANK", 0)), self, phase, d.ngpus_per_node)
            return
        self.setup_distributed_params()
        n = self.cfg.Distributed.ngpus_per_node
        mp.spawn(_worker, nprocs=n, args=(self, phase, n))
