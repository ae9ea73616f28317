"""operators/distributed_wrapper.py:6-69 of the reference: one process per GPU, rendezvous over
TCP on localhost.  backend 'nccl' IS RCCL on ROCm (collectives run over xGMI).  When the process
was started by torch.distributed.run (RANK / WORLD_SIZE in the environment) no further processes
are spawned — the launcher already made one per GPU."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class DistributedWrapper(object):
    def __init__(self, cfg, operator_class):
        self.cfg = cfg
        self.operator_class = operator_class

    def setup_distributed_params(self):
        ngpus_per_node = torch.cuda.device_count()
        if ngpus_per_node < 1:
            raise ValueError('[x] Can not get gpu numbers!')
        self.cfg.Distributed.ngpus_per_node = ngpus_per_node
        self.cfg.Distributed.world_size = ngpus_per_node * self.cfg.Distributed.world_size

    def init_operator(self, gpu, ngpus_per_node, cfg):
        cfg.Distributed.gpu_id = gpu
        print("=> Use GPU: {}".format(gpu))
        cfg.Distributed.rank = cfg.Distributed.rank * ngpus_per_node + gpu
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(gpu)
        if not dist.is_initialized():
            dist.init_process_group(backend='nccl', init_method=cfg.Distributed.dist_url,
                                    world_size=cfg.Distributed.world_size, rank=cfg.Distributed.rank)
        return self.operator_class(cfg)

    def _launch(self, fn):
        if "RANK" in os.environ and "WORLD_SIZE" in os.environ:        # under torch.distributed.run
            local = int(os.environ.get("LOCAL_RANK", 0))
            self.cfg.Distributed.world_size = int(os.environ["WORLD_SIZE"])
            self.cfg.Distributed.ngpus_per_node = int(os.environ.get("LOCAL_WORLD_SIZE", 1))
            self.cfg.Distributed.rank = 0
            self.cfg.Distributed.dist_url = "env://"
            fn(local, int(os.environ["WORLD_SIZE"]) if False else self.cfg.Distributed.ngpus_per_node, self.cfg)
            return
        self.setup_distributed_params()
        mp.spawn(fn, nprocs=self.cfg.Distributed.ngpus_per_node, args=(self.cfg.Distributed.ngpus_per_node, self.cfg))

    def train(self):
        self._launch(self.dist_training_process)

    def eval(self):
        self._launch(self.dist_evaluation_process)

    def dist_training_process(self, gpu, ngpus_per_node, cfg):
        self.init_operator(gpu, ngpus_per_node, cfg).training_process()

    def dist_evaluation_process(self, gpu, ngpus_per_node, cfg):
        self.init_operator(gpu, ngpus_per_node, cfg).evaluation_process()
