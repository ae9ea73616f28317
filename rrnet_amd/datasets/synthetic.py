"""Synthetic VisDrone-DET-shaped batches (there is no dataset in either repository).

Recipe of SURVEY §8(d): seed 219 + rank; uint8-uniform frames normalised with the ImageNet
mean/std of configs/rrnet_config.py:36-37; `boxes_per_image` annotations with log-uniform sizes in
[5,166] px and classes 1..10.  The training targets of the collate_fn_ctnet contract
(datasets/drones_det.py:70-94: imgs [B,3,H,W], annos [B,M,8], hm [B,10,H/4,W/4], wh/offset
[B,M,2], ind/reg_mask [B,M,1]) are built ON THE DEVICE by rr_ctnet_targets (csrc/targets.hip) — one launch
per batch instead of the reference's host loop over boxes (datasets/transforms/functional.py:230-262).
The host restatement of that loop lives in oracle/targets.py (checker only)."""
import numpy as np
import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def synth_annotations(rng, n, height, width, min_wh=5.0, max_wh=166.0):
    max_wh = min(max_wh, height / 2.0, width / 2.0)
    w = np.exp(rng.uniform(np.log(min_wh), np.log(max_wh), n))
    h = np.exp(rng.uniform(np.log(min_wh), np.log(max_wh), n))
    x = rng.uniform(0, width - w)
    y = rng.uniform(0, height - h)
    cls = rng.integers(1, 11, n)
    return np.stack([x, y, w, h, np.ones(n), cls, np.zeros(n), np.zeros(n)], 1).astype(np.float32)


def collate_ctnet_device(annos_list, height, width, scale_factor=4, num_classes=10, device="cuda"):
    """collate_fn_ctnet (datasets/drones_det.py:70-94) with the targets built on the device by rr_ctnet_targets:
    list of per-image annotation tensors [n_i, 8] -> (annos [B,M,8], hm, whs, inds, offsets, reg_masks) on `device`."""
    from rrnet_amd import ops
    bs = len(annos_list)
    m = max(int(a.size(0)) for a in annos_list)
    annos = torch.zeros(bs, m, 8)
    for i, a in enumerate(annos_list):
        annos[i, :a.size(0)] = a[:, :8]
    counts = torch.tensor([int(a.size(0)) for a in annos_list], dtype=torch.int32)
    annos_d = annos.to(device)
    hm, wh, ind, off, mask = ops.ctnet_targets(annos_d, counts.to(device), height, width, scale_factor, num_classes)
    return annos_d, hm, wh, ind, off, mask


def synth_frames(batch_size, height, width, boxes_per_image=100, seed=219, rank=0):
    """Host side of the recipe (what a real loader would hand over): normalised frames [B,3,H,W] float32 (CPU) and
    the per-image annotation tensors [n,8]."""
    rng = np.random.default_rng(seed + rank)
    mean = torch.tensor(MEAN).view(3, 1, 1)
    std = torch.tensor(STD).view(3, 1, 1)
    imgs, annos_list = [], []
    for _ in range(batch_size):
        frame = torch.from_numpy(rng.integers(0, 256, (height, width, 3), dtype=np.uint8))
        imgs.append((frame.permute(2, 0, 1).float() / 255. - mean) / std)
        annos_list.append(torch.from_numpy(synth_annotations(rng, boxes_per_image, height, width)))
    return torch.stack(imgs), annos_list


def synth_batch(batch_size, height, width, boxes_per_image=100, seed=219, rank=0, scale_factor=4, num_classes=10,
                device="cuda"):
    """One batch in collate_fn_ctnet's order, resident on `device`:
    (imgs, annos, hms, whs, inds, offsets, reg_masks, names); the targets come from rr_ctnet_targets."""
    imgs, annos_list = synth_frames(batch_size, height, width, boxes_per_image, seed, rank)
    annos, hm, wh, ind, off, mask = collate_ctnet_device(annos_list, height, width, scale_factor, num_classes, device)
    imgs = imgs.to(device).contiguous(memory_format=torch.channels_last)
    return imgs, annos, hm, wh, ind, off, mask, ["synthetic_%06d" % i for i in range(batch_size)]


class SyntheticDronesDET:
    """Iteration-based loader with the `get_batch()` surface of datasets/dataloader.py:27-37:
    returns CUDA tensors (imgs, annos, hms, whs, inds, offsets, reg_masks, names)."""

    def __init__(self, cfg, batch_size, height, width, boxes_per_image=100, rank=0, device="cuda", pool=2):
        self.pool = [synth_batch(batch_size, height, width, boxes_per_image, cfg.seed + 1000 * i, rank,
                                 cfg.Train.scale_factor, cfg.num_classes, device) for i in range(pool)]
        self.device = device
        self.i = 0

    def get_batch(self):
        """The batches are resident on the device.  The annotations are handed out as a copy: the criterion converts
        them to xyxy in place (rrnet_operator.py:67), which the reference's per-step H2D copy absorbed."""
        b = self.pool[self.i % len(self.pool)]
        self.i += 1
        return (b[0], b[1].clone()) + tuple(b[2:])

    def __len__(self):
        return len(self.pool)


class HostFedDronesDET:
    """The reference's hand-over (operators/rrnet_operator.py:121: `batch = self.training_loader.get_batch()` returns HOST
    tensors that the step moves to the GPU; datasets/drones_det.py:70-94 builds them): the frames and annotations of
    every batch live in PINNED host memory and cross PCIe once per step — on a copy stream of their own, one batch ahead of
    the compute stream, into one of two device slots; `get_batch()` makes the compute stream wait for the batch's copy
    event, builds the targets on the device (rr_ctnet_targets, as the device-resident loader does once at start-up) and
    starts the copy of the NEXT batch, which may overwrite the other slot only after the step that used it has been
    enqueued (the copy stream waits for the compute stream at that point).  Same `get_batch()` surface and the same
    batches, bit for bit, as SyntheticDronesDET (tests/test_train_gpu.py).  ~100 MB per step at B=8, 1024x1024: 2 ms of
    PCIe Gen5 time under a 458 ms step."""

    def __init__(self, cfg, batch_size, height, width, boxes_per_image=100, rank=0, device="cuda", pool=2):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.hw = (height, width)
        self.cfg = (cfg.Train.scale_factor, cfg.num_classes)
        self.host = []
        for i in range(pool):
            imgs, annos_list = synth_frames(batch_size, height, width, boxes_per_image, cfg.seed + 1000 * i, rank)
            m = max(int(a.size(0)) for a in annos_list)
            annos = torch.zeros(batch_size, m, 8)
            for j, a in enumerate(annos_list):
                annos[j, :a.size(0)] = a[:, :8]
            counts = torch.tensor([int(a.size(0)) for a in annos_list], dtype=torch.int32)
            self.host.append((imgs.contiguous(memory_format=torch.channels_last).pin_memory(), annos.pin_memory(),
                              counts.pin_memory(), ["synthetic_%06d" % j for j in range(batch_size)]))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [None, None]           # device buffers (imgs, annos, counts) of the two batches in flight
        self.events = [None, None]
        self.i = 0
        self._prefetch(0)

    def _prefetch(self, i):
        slot = i % 2
        h = self.host[i % len(self.host)]
        cur = torch.cuda.current_stream(self.device)
        self.copy_stream.wait_stream(cur)          # the step that last read this slot has been enqueued on `cur`
        with torch.cuda.stream(self.copy_stream):
            if self.slots[slot] is None or self.slots[slot][1].shape != h[1].shape:
                self.slots[slot] = (torch.empty_like(h[0], device=self.device), torch.empty_like(h[1], device=self.device),
                                    torch.empty_like(h[2], device=self.device))
            d = self.slots[slot]
            for dst, src in zip(d, h[:3]):
                dst.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.events[slot] = ev

    def get_batch(self):
        from rrnet_amd import ops
        i = self.i
        self.i += 1
        slot = i % 2
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.events[slot])
        imgs, annos, counts = self.slots[slot]
        for t in (imgs, annos, counts):
            t.record_stream(cur)                   # allocated on the copy stream, read on the compute stream
        hm, wh, ind, off, mask = ops.ctnet_targets(annos, counts, self.hw[0], self.hw[1], *self.cfg)
        batch = (imgs, annos.clone(), hm, wh, ind, off, mask, self.host[i % len(self.host)][3])
        self._prefetch(i + 1)
        return batch

    def __len__(self):
        return len(self.host)


_LOADERS = {}


def make_dataloader(cfg, collate_fn='rrnet'):
    """datasets/__init__.py make_dataloader surface: (training_loader, validation_loader).  Real
    VisDrone loading / augmentation is out of scope; synthetic frames of the configured crop size.  The generated
    pool is cached per (batch, size, seed, rank): a second operator in the same process (bench.py's secondary
    workloads) reuses the resident batches."""
    rank = getattr(cfg.Distributed, "rank", 0)
    h, w = cfg.Train.crop_size
    key = (cfg.Train.batch_size, h, w, cfg.seed, rank, cfg.Train.scale_factor, cfg.num_classes)
    if key not in _LOADERS:
        _LOADERS[key] = SyntheticDronesDET(cfg, cfg.Train.batch_size, h, w, rank=rank)
    return _LOADERS[key], None


def synth_head_outputs(n_frames, hf, wf, num_classes=10, seed=219, device="cpu", planted=200, cluster=5):
    """Stage-1 head outputs + backbone feature of `n_frames` frames for the inference-only workload
    (BASELINE.json configs[4]; recipe SURVEY §8(d)): hm logits -2.19 + N(0,1) with `planted` peaks (+U[3,8])
    placed in clusters of `cluster` neighbouring pixels of one class so that NMS has work to do,
    wh = 2 + 8|N(0,1)|, offset U[0,1), feature relu(N(0,1)) [256,hf,wf].  Generated on `device` from a
    seeded torch generator (the CUDA and CPU streams differ; a test moves the CPU tensors over)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    kw = dict(generator=g, device=device)
    hm = torch.randn((n_frames, num_classes, hf, wf), **kw) - 2.19
    n_cl = max(planted // cluster, 1)
    for f in range(n_frames):
        cy = torch.randint(2, hf - 2, (n_cl,), **kw)
        cx = torch.randint(2, wf - 2, (n_cl,), **kw)
        cc = torch.randint(0, num_classes, (n_cl,), **kw)
        dy = torch.randint(-2, 3, (n_cl, cluster), **kw)
        dx = torch.randint(-2, 3, (n_cl, cluster), **kw)
        amp = 3.0 + 5.0 * torch.rand((n_cl, cluster), **kw)
        hm[f].index_put_((cc[:, None].expand(-1, cluster), cy[:, None] + dy, cx[:, None] + dx), amp, accumulate=True)
    wh = 2.0 + 8.0 * torch.randn((n_frames, 2, hf, wf), **kw).abs()
    off = torch.rand((n_frames, 2, hf, wf), **kw)
    feat = torch.relu(torch.randn((n_frames, 256, hf, wf), **kw))
    return hm, wh, off, feat
