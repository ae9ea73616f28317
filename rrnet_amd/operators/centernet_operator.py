"""CenterNetOperator — operators/centernet_operator.py:20-297 of the reference on the MI355X path (BASELINE
configs[0]: CenterNet + hourglass through the operator).  Same constructor / criterion / training_process /
transform_bbox / _topk / _ctnet_nms / _ext_nms / save_result / evaluation_process surface; per-GPU BatchNorm like
the reference (it does not convert to SyncBN, :24).  Underneath: the HIP kernels of librrnet_hip.so, the fused flat
Adam, the decode kernel in its CenterNet box mode (x,y,w,h * scale, cls+1, no clamp), the batched Soft-NMS."""
import os

import numpy as np
import torch
import torch.optim as optim

from rrnet_amd import functional as RF
from rrnet_amd import ops
from rrnet_amd.datasets import make_dataloader
from rrnet_amd.datasets.transforms.functional import flip_annos, flip_img
from rrnet_amd.flat import FlatAdam, FlatParams
from rrnet_amd.models.centernet import CenterNet
from rrnet_amd.models.rrnet import RRNet
from rrnet_amd.modules.loss.focalloss import FocalLossHM
from rrnet_amd.modules.loss.regl1loss import RegL1Loss
from .base_operator import BaseOperator
from .rrnet_operator import RRNetOperator



def _released(v):
    """The same values without the autograd graph (recursively through tuples / lists).  backward() has consumed the graph;
    handing its roots out would keep every node of the step alive until the caller drops them — and with the nodes what they
    attach to their ctx by hand (shared fan-in buffers, bf16 images, parameter references): 20 GiB at the bench configuration
    that a training loop holding `losses` until the next step returns would carry through that step's forward."""
    if torch.is_tensor(v):
        return v.detach()
    if isinstance(v, (tuple, list)):
        return type(v)(_released(x) for x in v)
    return v

class CenterNetOperator(BaseOperator):
    def __init__(self, cfg):
        self.cfg = cfg
        model = CenterNet(cfg).cuda(cfg.Distributed.gpu_id).to(memory_format=torch.channels_last)
        flat = FlatParams(model)
        self.optimizer = FlatAdam(flat, lr=cfg.Train.lr)
        self.lr_sch = optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=cfg.Train.lr_milestones, gamma=0.1)
        self.training_loader, self.validation_loader = make_dataloader(cfg, collate_fn='ctnet')
        super(CenterNetOperator, self).__init__(cfg=self.cfg, model=model, lr_sch=self.lr_sch, flat=flat)
        self.focal_loss = FocalLossHM()
        self.l1_loss = RegL1Loss()
        self.main_proc_flag = cfg.Distributed.gpu_id == 0

    def criterion(self, outs, annos):
        """centernet_operator.py:42-59 -> (hm_loss, wh_loss, off_loss); sigmoid + clamp live inside the fused loss."""
        hms, whs, offsets = outs
        t_hms, t_whs, t_inds, t_offsets, t_reg_masks = annos
        ns = self.cfg.Model.num_stacks
        gt = ops.to_nhwc(t_hms)
        hm_loss = wh_loss = off_loss = 0
        for s in range(ns):
            hm_loss = hm_loss + self.focal_loss.from_logits(hms[s], gt) / ns
            wh_loss = wh_loss + self.l1_loss(whs[s], t_reg_masks, t_inds, t_whs) / ns
            off_loss = off_loss + self.l1_loss(offsets[s], t_reg_masks, t_inds, t_offsets) / ns
        return hm_loss, wh_loss, off_loss

    def train_step(self, step, batch):
        """One iteration of training_process (centernet_operator.py:76-108) without the logging."""
        imgs, _annos, hms, whs, inds, offsets, reg_masks, _names = batch
        self.lr_sch.step()
        self.optimizer.zero_grad()
        outs = self.model(imgs)
        hm_loss, wh_loss, off_loss = self.criterion(outs, (hms, whs, inds, offsets, reg_masks))
        loss = hm_loss + (0.1 * wh_loss) + off_loss
        loss.backward()
        self.optimizer.step()
        return _released(outs), _released((loss, hm_loss, wh_loss, off_loss))

    def training_process(self):
        self.model.train()
        totals = np.zeros(4)
        log_dir = os.path.join('./log', self.cfg.log_prefix)
        for step in range(self.cfg.Train.iter_num):
            batch = self.training_loader.get_batch()
            outs, losses = self.train_step(step, batch)
            totals += np.array([float(l.detach()) for l in losses])
            pi = self.cfg.Train.print_interval
            if self.main_proc_flag:
                if step % pi == pi - 1:
                    print("step %d  loss %.4f hm %.4f wh %.4f off %.4f  lr %.3g" %
                          ((step,) + tuple(totals / pi) + (self.optimizer.param_groups[0]['lr'],)), flush=True)
                    pred = self.transform_bbox(outs[0][1], outs[1][1], outs[2][1],
                                               scale_factor=self.cfg.Train.scale_factor)
                    self._ext_nms(pred)
                    totals[:] = 0
                ci = self.cfg.Train.checkpoint_interval
                if step % ci == ci - 1 or step == self.cfg.Train.iter_num - 1:
                    os.makedirs(log_dir, exist_ok=True)
                    self.save_ckp(self.model.module, step, log_dir)

    def transform_bbox(self, hm, wh, offset, k=250, scale_factor=4):
        """centernet_operator.py:152-178: rows [x,y,w,h,score,cls+1] of image 0 in image coordinates, score > 0.01.
        offset=None: the reference's +0.5 centre."""
        hm, wh = ops.to_nhwc(hm.detach()), ops.to_nhwc(wh.detach())
        if offset is None:
            offset = torch.full_like(wh, 0.5)
        rows = ops.decode_topk(hm[0:1], wh[0:1], ops.to_nhwc(offset.detach())[0:1], k, is_logits=True, box_mode=1,
                               scale=float(scale_factor))[0]
        return rows[rows[:, 4] > 0.01, :]

    _gather_feat = staticmethod(RRNet._gather_feat)

    def _transpose_and_gather_feat(self, feat, ind):
        feat = ops.to_nhwc(feat).permute(0, 2, 3, 1)
        return self._gather_feat(feat.reshape(feat.size(0), -1, feat.size(3)), ind)

    def _topk(self, scores, k=40):
        """centernet_operator.py:186-202 (same contract as RRNet._topk)."""
        b, c, h, w = scores.shape
        zero = ops.zeros_nhwc(b, 2, h, w, scores.device)
        rows, pix = ops.decode_topk(ops.to_nhwc(scores.detach().float()), zero, zero, k, is_logits=False, want_pix=True)
        pix = pix.long()
        return rows[..., 4], pix, rows[..., 5].int(), (pix // w).float(), (pix % w).float()

    def _ctnet_nms(self, heat, kernel=3):
        """centernet_operator.py:204-210: heat (a score map, e.g. sigmoid of the heat-map) * (heat == its 3x3 max)."""
        if kernel != 3:
            raise NotImplementedError("_ctnet_nms: the kernel implements the 3x3 window the reference uses")
        return ops.peak3x3(ops.to_nhwc(heat.detach().float()), is_logits=False)

    @staticmethod
    def _ext_nms(pred_bbox):
        """centernet_operator.py:222-236: per-class gaussian Soft-NMS; boxes come back as XYXY rows (the reference
        does not convert them back), on the host."""
        if pred_bbox.size(0) == 0:
            return pred_bbox
        kept = RRNetOperator._ext_nms(pred_bbox)
        kept[:, 2:4] += kept[:, 0:2]
        return kept

    @staticmethod
    def save_result(file_path, pred_bbox):
        """centernet_operator.py:238-249: rounded integer boxes, width / height from the rounded corners."""
        pred_bbox = torch.clamp(pred_bbox, min=0.)
        with open(file_path, 'w') as f:
            for i in range(pred_bbox.size()[0]):
                bbox = pred_bbox[i]
                bbox[:4] = torch.round(bbox[:4])
                f.write('%d,%d,%d,%d,%.4f,%d,-1,-1\n' % (int(bbox[0]), int(bbox[1]), int(bbox[2]) - int(bbox[0]),
                                                         int(bbox[3]) - int(bbox[1]), float(bbox[4]), int(bbox[5])))

    def evaluate_images(self, imgs):
        """Body of evaluation_process (:262-285) for one image batch (bs=1): every scale twice (flipped, plain)."""
        boxes = []
        sf = self.cfg.Train.scale_factor
        for scale in self.cfg.Val.scales:
            img = ops.resize_bilinear_ac(imgs, scale)
            w = img.size(3)
            flipped = flip_img(img.squeeze(0)).unsqueeze(0).contiguous(memory_format=torch.channels_last)
            outs = self.model(flipped)
            pred = self.transform_bbox(outs[0][1], outs[1][1], outs[2][1], scale_factor=sf).cpu()
            pred = flip_annos(pred, w)
            pred[:, :4] = pred[:, :4] / scale
            boxes.append(pred)
            outs = self.model(img)
            pred = self.transform_bbox(outs[0][1], outs[1][1], outs[2][1], scale_factor=sf).cpu()
            pred[:, :4] = pred[:, :4] / scale
            boxes.append(pred)
        pred = torch.cat(boxes, dim=0)
        _, idx = torch.sort(pred[:, 4], descending=True)
        pred = pred[idx]
        if not self.cfg.Val.auto_test:
            pred = self._ext_nms(pred)
        return pred

    def evaluation_process(self):
        self.model.eval()
        state_dict = torch.load(self.cfg.Val.model_path, map_location='cpu')
        self.model.module.load_state_dict(state_dict)
        if self.validation_loader is None:
            raise RuntimeError("no validation data: the VisDrone loader is outside the accelerated path")
        os.makedirs(self.cfg.Val.result_dir, exist_ok=True)
        with torch.no_grad():
            for data in self.validation_loader:
                imgs, _annos, names = data
                pred = self.evaluate_images(imgs.cuda())
                self.save_result(os.path.join(self.cfg.Val.result_dir, names[0] + '.txt'), pred)
            print('=> Evaluation Done!')
