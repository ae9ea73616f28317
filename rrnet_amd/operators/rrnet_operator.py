"""RRNetOperator — operators/rrnet_operator.py:22-284 of the reference on the MI355X path.

Same constructor / criterion / training_process / generate_bbox / _ext_nms / save_result /
evaluation_process surface.  What changed underneath:
  * model, losses, decode, NMS, RoIAlign, optimizer are the HIP kernels of librrnet_hip.so;
  * Adam runs fused over one flat buffer and carries the RCCL gradient all-reduce (FlatAdam);
  * the stage-2 loss has no per-image host loop and no `.sum() == 0` synchronisation;
  * `_ext_nms` runs the bit-exact wavefront-parallel Soft-NMS over all classes in one launch.
Out of scope and reduced to stdout: tensorboard / cv2 visualisation (utils/vis)."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from rrnet_amd import functional as RF
from rrnet_amd import ops
from rrnet_amd.datasets import make_dataloader
from rrnet_amd.ext.nms.nms_wrapper import soft_nms_segments
from rrnet_amd.flat import FlatAdam, FlatParams
from rrnet_amd.models.rrnet import RRNet
from rrnet_amd.modules.loss.focalloss import FocalLossHM
from rrnet_amd.modules.loss.regl1loss import RegL1Loss
from .base_operator import BaseOperator



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

class RRNetOperator(BaseOperator):
    def __init__(self, cfg):
        self.cfg = cfg
        model = RRNet(cfg).cuda(cfg.Distributed.gpu_id).to(memory_format=torch.channels_last)
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
        flat = FlatParams(model)
        self.optimizer = FlatAdam(flat, lr=cfg.Train.lr)
        self.lr_sch = optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=cfg.Train.lr_milestones, gamma=0.1)
        self.training_loader, self.validation_loader = make_dataloader(cfg, collate_fn='rrnet')
        super(RRNetOperator, self).__init__(cfg=self.cfg, model=model, lr_sch=self.lr_sch, flat=flat)
        self.hm_focal_loss = FocalLossHM()
        self.l1_loss = RegL1Loss()
        self.main_proc_flag = cfg.Distributed.gpu_id == 0

    def criterion(self, outs, targets):
        """rrnet_operator.py:42-84 -> (hm_loss, wh_loss, off_loss, s2_reg_loss)."""
        s1_hms, s1_whs, s1_offsets, s2_reg, bxyxy, scores, _ = outs
        gt_hms, gt_whs, gt_inds, gt_offsets, gt_reg_masks, gt_annos = targets
        ns = self.cfg.Model.num_stacks
        gt_hm_nhwc = ops.to_nhwc(gt_hms)                 # one layout for logits and targets
        hm_loss = wh_loss = off_loss = 0
        for s in range(ns):
            hm_loss = hm_loss + self.hm_focal_loss.from_logits(s1_hms[s], gt_hm_nhwc) / ns
            wh_loss = wh_loss + self.l1_loss(s1_whs[s], gt_reg_masks, gt_inds, gt_whs) / ns
            off_loss = off_loss + self.l1_loss(s1_offsets[s], gt_reg_masks, gt_inds, gt_offsets) / ns
        gt_annos[:, :, 2:4] += gt_annos[:, :, 0:2]       # in place, as the reference (:67)
        s2_reg_loss = RF.stage2_reg_loss(s2_reg, bxyxy, gt_annos, self.cfg.Train.scale_factor)
        return hm_loss, wh_loss, off_loss, s2_reg_loss

    @staticmethod
    def generate_bbox_target(ex_rois, gt_rois):
        """rrnet_operator.py:86-102 (kept for API parity; the fused loss kernel computes the same)."""
        ew = ex_rois[:, 2] - ex_rois[:, 0] + 1.0
        eh = ex_rois[:, 3] - ex_rois[:, 1] + 1.0
        ecx, ecy = ex_rois[:, 0] + 0.5 * ew, ex_rois[:, 1] + 0.5 * eh
        gw = gt_rois[:, 2] - gt_rois[:, 0] + 1.0
        gh = gt_rois[:, 3] - gt_rois[:, 1] + 1.0
        gcx, gcy = gt_rois[:, 0] + 0.5 * gw, gt_rois[:, 1] + 0.5 * gh
        return torch.stack(((gcx - ecx) / ew, (gcy - ecy) / eh, torch.log(gw / ew), torch.log(gh / eh)), dim=1)

    def train_step(self, step, batch):
        """One iteration of training_process (rrnet_operator.py:116-144) without the logging."""
        imgs, annos, gt_hms, gt_whs, gt_inds, gt_offsets, gt_reg_masks, _names = batch
        self.lr_sch.step()
        self.optimizer.zero_grad()
        outs = self.model(imgs)
        targets = gt_hms, gt_whs, gt_inds, gt_offsets, gt_reg_masks, annos
        hm_loss, wh_loss, offset_loss, s2_reg_loss = self.criterion(outs, targets)
        s2_factor = 0 if step < 2000 else 1
        loss = hm_loss + (0.1 * wh_loss) + offset_loss + s2_reg_loss * s2_factor
        loss.backward()
        self.optimizer.step()
        return _released(outs), _released((loss, hm_loss, wh_loss, offset_loss, s2_reg_loss))

    def training_process(self):
        self.model.train()
        totals = np.zeros(5)
        log_dir = os.path.join('./log', self.cfg.log_prefix)
        for step in range(self.cfg.Train.iter_num):
            batch = self.training_loader.get_batch()
            outs, losses = self.train_step(step, batch)
            totals += np.array([float(l.detach()) for l in losses])
            pi = self.cfg.Train.print_interval
            if self.main_proc_flag:
                if step % pi == pi - 1:
                    lr = self.optimizer.param_groups[0]['lr']
                    print("step %d  loss %.4f hm %.4f wh %.4f off %.4f s2 %.4f  lr %.3g" %
                          ((step,) + tuple(totals / pi) + (lr,)), flush=True)
                    s1_pred_bbox, s2_pred_bbox = self.generate_bbox(outs, batch_idx=0)
                    s2_pred_bbox = self._ext_nms(s2_pred_bbox)
                    totals[:] = 0
                ci = self.cfg.Train.checkpoint_interval
                if step % ci == ci - 1 or step == self.cfg.Train.iter_num - 1:
                    os.makedirs(log_dir, exist_ok=True)
                    self.save_ckp(self.model.module, step, log_dir)

    def generate_bbox(self, outs, batch_idx=0):
        """rrnet_operator.py:188-209 -> (stage-1 boxes [n,6] xywh cls=0, stage-2 boxes [n,6] xywh cls+1)."""
        _, _, _, s2_reg, bxyxy, scores, clses = outs
        flag = bxyxy[:, 0] == batch_idx
        reg = s2_reg[flag].detach()
        box = bxyxy[flag, 1:] * self.cfg.Train.scale_factor
        score, cls = scores[flag], clses[flag]
        xy, wh = box[:, 0:2], box[:, 2:4] - box[:, 0:2]
        s1 = torch.cat((xy, wh, score.view(-1, 1), torch.zeros((box.size(0), 1), device=box.device)), dim=1)
        wh1 = wh + 1
        cx = reg[:, 0] * wh1[:, 0] + xy[:, 0] + wh1[:, 0] / 2
        cy = reg[:, 1] * wh1[:, 1] + xy[:, 1] + wh1[:, 1] / 2
        w = reg[:, 2].exp() * wh1[:, 0]
        h = reg[:, 3].exp() * wh1[:, 1]
        s2 = torch.stack((cx - w / 2., cy - h / 2., w, h, score, cls.float() + 1), dim=1)
        return s1, s2

    @staticmethod
    def _ext_nms_device(pred_bbox, per_cls=True):
        """rrnet_operator.py:211-232 on the device: per-class gaussian Soft-NMS (Nt 0.7, threshold 0.1) on xywh boxes,
        all classes in one launch; returns a device tensor."""
        dev = pred_bbox.device if pred_bbox.is_cuda else torch.device("cuda", torch.cuda.current_device())
        b = pred_bbox.detach().to(dev, torch.float32).clone()
        b[:, 2] = b[:, 0] + b[:, 2]
        b[:, 3] = b[:, 1] + b[:, 3]
        n = b.size(0)
        if per_cls:
            # one segment per class id present, like the reference's unique() loop (ids are cls+1, any dataset)
            lo, hi = int(b[:, 5].min()), int(b[:, 5].max())
            nc = hi - lo + 1
            if nc > 1023:
                raise ValueError("_ext_nms: class ids span %d values (limit 1023)" % nc)
            grouped, seg_off, seg_len = ops.group_by_class(b.view(1, n, 6), nc, cls_base=lo)
        else:
            nc = 1
            grouped = b.view(1, n, 6).clone()
            seg_off = torch.tensor([0, n], dtype=torch.int32, device=dev)
            seg_len = None
        rows = grouped.view(-1, 6)
        n_out = soft_nms_segments(rows, seg_off, n, sigma=0.5, Nt=0.7, threshold=0.1, method=2, seg_len=seg_len)
        _, _, _, kept = ops.pack_segments(rows, seg_off, n_out, nc, want_rois=False, want_rows=True)
        kept[:, 2:4] -= kept[:, 0:2]
        return kept

    @staticmethod
    def _ext_nms(pred_bbox, per_cls=True):
        """rrnet_operator.py:211-232 -> CPU tensor, like the reference."""
        if pred_bbox.size(0) == 0:
            return pred_bbox
        return RRNetOperator._ext_nms_device(pred_bbox, per_cls).cpu()

    @staticmethod
    def save_result(file_path, pred_bbox):
        """rrnet_operator.py:234-244 text format."""
        pred_bbox = torch.clamp(pred_bbox, min=0.)
        with open(file_path, 'w') as f:
            for i in range(pred_bbox.size()[0]):
                bbox = pred_bbox[i]
                f.write('%f,%f,%f,%f,%.4f,%d,-1,-1\n' % (float(bbox[0]), float(bbox[1]), float(bbox[2]),
                                                        float(bbox[3]), float(bbox[4]), int(bbox[5])))

    def evaluate_images(self, imgs):
        """Multi-scale inference of rrnet_operator.py:256-276 for one image batch (bs=1) -> boxes [n,6] on the host.
        Everything between the resize and the final result stays on the device: per-scale boxes, the cross-scale
        concatenation, both score sorts (rr_sort_rows_by_score) and the Soft-NMS; one D2H copy at the end."""
        multi_scale_bboxes = []
        for scale in self.cfg.Val.scales:
            img = ops.resize_bilinear_ac(imgs, scale)
            outs = self.model(img)
            _, pred_bbox = self.generate_bbox(outs)
            if not self.cfg.Val.auto_test:
                pred_bbox = pred_bbox[pred_bbox[:, 4] > 0.01]
            pred_bbox = pred_bbox.clone()
            pred_bbox[:, :4] = pred_bbox[:, :4] / scale
            multi_scale_bboxes.append(pred_bbox)
        pred_bbox = torch.cat(multi_scale_bboxes, dim=0)
        if pred_bbox.size(0) == 0:
            return pred_bbox.cpu()
        if pred_bbox.size(0) > 16384:          # beyond the LDS sort (6 scales x 1500 boxes = 9000 in the reference config)
            pred_bbox = pred_bbox[torch.sort(pred_bbox[:, 4], descending=True, stable=True)[1]]
        else:
            pred_bbox = ops.sort_rows_by_score(pred_bbox)
        if not self.cfg.Val.auto_test:
            pred_bbox = self._ext_nms_device(pred_bbox)
        if 0 < pred_bbox.size(0) <= 16384:
            pred_bbox = ops.sort_rows_by_score(pred_bbox)
        elif pred_bbox.size(0) > 16384:
            pred_bbox = pred_bbox[torch.sort(pred_bbox[:, 4], descending=True, stable=True)[1]]
        return pred_bbox.cpu()

    def evaluation_process(self):
        self.model.eval()
        state_dict = torch.load(self.cfg.Val.model_path, map_location='cpu')
        self.model.module.load_state_dict(state_dict)
        if self.validation_loader is None:
            raise RuntimeError("no validation data: the VisDrone loader is outside the accelerated path")
        os.makedirs(self.cfg.Val.result_dir, exist_ok=True)
        with torch.no_grad():
            for step, data in enumerate(self.validation_loader):
                imgs, annos, names = data
                pred_bbox = self.evaluate_images(imgs.cuda())
                self.save_result(os.path.join(self.cfg.Val.result_dir, names[0] + '.txt'), pred_bbox)
            print('=> Evaluation Done!')
