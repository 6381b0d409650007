"""Anchor target assignment on the device, no host round trips.

Restates AxisAlignedTargetAssigner (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-213) with
boxes3d_nearest_bev_iou (pcdet/utils/box_utils.py:252-323).  The reference trims padded ground-truth rows with a python
loop over `.sum()` and takes arg-maxes through `.cpu().numpy()` (:53-57,:148,:153); here padding is a mask and everything
stays vectorised on the GPU.  POS_FRACTION < 0 (no sampling), as hvpr.yaml:122 sets."""
import numpy as np
import torch

from .common_utils import limit_period


def nearest_bev_boxes(boxes):
    """(N,7) -> axis-aligned (x1,y1,x2,y2) after snapping the heading to the nearest axis (box_utils.py:297-308)."""
    rot = limit_period(boxes[:, 6], 0.5, np.pi).abs()
    dims = torch.where(rot[:, None] < np.pi / 4, boxes[:, 3:5], boxes[:, 3:5].flip(1))    # (slices: a python index list is a synchronising host copy)
    return torch.cat((boxes[:, 0:2] - dims / 2, boxes[:, 0:2] + dims / 2), dim=1)


def iou_axis_aligned(a, b):
    """(N,4) x (M,4) -> (N,M) (box_utils.py:252-272)."""
    xl = torch.max(a[:, 0, None], b[None, :, 0])
    xr = torch.min(a[:, 2, None], b[None, :, 2])
    yl = torch.max(a[:, 1, None], b[None, :, 1])
    yr = torch.min(a[:, 3, None], b[None, :, 3])
    inter = torch.clamp_min(xr - xl, 0) * torch.clamp_min(yr - yl, 0)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / torch.clamp_min(area_a[:, None] + area_b[None, :] - inter, 1e-6)


def boxes3d_nearest_bev_iou(boxes_a, boxes_b):
    return iou_axis_aligned(nearest_bev_boxes(boxes_a), nearest_bev_boxes(boxes_b))


class AxisAlignedTargetAssigner:
    def __init__(self, model_cfg, class_names, box_coder, match_height=False):
        assert not match_height, "hvpr path: MATCH_HEIGHT False (hvpr.yaml:124)"
        acfg = model_cfg.ANCHOR_GENERATOR_CONFIG
        tcfg = model_cfg.TARGET_ASSIGNER_CONFIG
        assert tcfg.POS_FRACTION < 0, "hvpr path: no fg/bg sampling (hvpr.yaml:122)"
        self.box_coder = box_coder
        self.class_names = list(class_names)
        self.anchor_class_names = [c["class_name"] for c in acfg]
        self.matched = {c["class_name"]: c["matched_threshold"] for c in acfg}
        self.unmatched = {c["class_name"]: c["unmatched_threshold"] for c in acfg}
        self.norm_by_num_examples = tcfg.NORM_BY_NUM_EXAMPLES

    def assign_targets(self, all_anchors, gt_boxes_with_classes):
        """all_anchors: list of (nz,ny,nx,1,R,7); gt (B,G,8).  Returns box_cls_labels (B,A) i32, box_reg_targets (B,A,7),
        reg_weights (B,A), anchors ordered (z,y,x,class,rot) as the single head predicts them.  The reference loops over the
        frames (:45-111); here a frame is a leading dimension of every tensor — the same arithmetic per element, one launch per
        operation for the whole batch instead of one per frame (16 frames: ~1000 small launches less per training step).  The
        (B, anchors, G) IoU intermediates are 4·B·A·G bytes each (batch 16, 147 k anchors, 50 padded boxes: 470 MB)."""
        gt_all = gt_boxes_with_classes
        B, G = gt_all.shape[0], gt_all.shape[1]
        nz = gt_all.abs().sum(dim=2) != 0
        ar = torch.arange(G, device=gt_all.device)
        last = torch.where(nz, ar[None, :], torch.zeros_like(ar)[None, :]).max(dim=1)[0]      # trailing zero rows are padding (:53-57)
        valid = ar[None, :] <= last[:, None]
        gcls = gt_all[:, :, -1].int()
        per_class = []
        for cname, anchors in zip(self.anchor_class_names, all_anchors):
            fms = anchors.shape[:3]
            a = anchors.reshape(-1, anchors.shape[-1])
            name_idx = self.class_names.index(cname)
            # python-style class_names[c - 1]: class 0 (a padded row inside the valid range) wraps to the last class
            same = torch.remainder(gcls - 1, len(self.class_names)) == name_idx
            lab, tgt, w = self._assign_batch(a, gt_all[:, :, :-1], gcls, valid & same, self.matched[cname], self.unmatched[cname])
            per_class.append((lab.view(B, *fms, -1), tgt.view(B, *fms, -1, self.box_coder.code_size), w.view(B, *fms, -1)))
        return {"box_cls_labels": torch.cat([p[0] for p in per_class], dim=-1).reshape(B, -1),
                "box_reg_targets": torch.cat([p[1] for p in per_class], dim=-2).reshape(B, -1, self.box_coder.code_size),
                "reg_weights": torch.cat([p[2] for p in per_class], dim=-1).reshape(B, -1)}

    FRAMES_PER_PASS = 4      # the (frames, anchors, ground truths) intermediates of one pass: ~120 MB each at 147 k anchors x 50 boxes

    def _assign_batch(self, anchors, gt, gt_classes, use, matched_thr, unmatched_thr):
        """anchors (A,7), gt (B,G,7), gt_classes / use (B,G) -> labels (B,A) i32, targets (B,A,7), weights (B,A) — assign_targets_single
        (:113-213) for FRAMES_PER_PASS frames at once: the launch count of a whole-batch pass without its peak memory (a batch of 16
        held eight (B,A,G) intermediates of 470 MB each)."""
        B = gt.shape[0]
        if B <= self.FRAMES_PER_PASS:
            return self._assign_frames(anchors, gt, gt_classes, use, matched_thr, unmatched_thr)
        parts = [self._assign_frames(anchors, gt[b:b + self.FRAMES_PER_PASS], gt_classes[b:b + self.FRAMES_PER_PASS],
                                     use[b:b + self.FRAMES_PER_PASS], matched_thr, unmatched_thr)
                 for b in range(0, B, self.FRAMES_PER_PASS)]
        return tuple(torch.cat([p[i] for p in parts], dim=0) for i in range(3))

    def _assign_frames(self, anchors, gt, gt_classes, use, matched_thr, unmatched_thr):
        B, G = gt.shape[0], gt.shape[1]
        A = anchors.shape[0]
        ab = nearest_bev_boxes(anchors[:, 0:7])                                     # (A,4)
        gb = nearest_bev_boxes(gt.reshape(B * G, -1)[:, 0:7]).view(B, G, 4)
        # (B,G,A) with the anchors innermost: the maximum over the G ground truths then runs over a strided outer dimension and the one
        # over the anchors over contiguous rows — with G innermost torch reduced 8-element rows at 46 GB/s (0.4 ms per pass of four frames)
        xl = torch.max(ab[None, None, :, 0], gb[:, :, None, 0])
        xr = torch.min(ab[None, None, :, 2], gb[:, :, None, 2])
        yl = torch.max(ab[None, None, :, 1], gb[:, :, None, 1])
        yr = torch.min(ab[None, None, :, 3], gb[:, :, None, 3])
        inter = xr.sub_(xl).clamp_min_(0).mul_(yr.sub_(yl).clamp_min_(0))          # (in place: xr becomes the intersection)
        del xl, yl, yr
        area_a = (ab[:, 2] - ab[:, 0]) * (ab[:, 3] - ab[:, 1])
        area_b = (gb[:, :, 2] - gb[:, :, 0]) * (gb[:, :, 3] - gb[:, :, 1])
        iou = inter / torch.clamp_min(area_a[None, None, :] + area_b[:, :, None] - inter, 1e-6)      # (B,G,A), box_utils.py:252-272
        del inter
        iou.masked_fill_(~use[:, :, None], -2.0)                                   # masked ground truths never match
        a2g_max, a2g_arg = iou.max(dim=1)
        g2a_max = iou.max(dim=2)[0]
        g2a_max = torch.where(g2a_max <= 0, torch.full_like(g2a_max, -1.0), g2a_max)   # no overlap at all: no forced match (:155-156)
        force = (iou == g2a_max[:, :, None]).any(dim=1)                            # best anchor(s) of every ground truth (:158-161)
        cls_of = torch.gather(gt_classes, 1, a2g_arg)
        labels = torch.full((B, A), -1, dtype=torch.int32, device=anchors.device)
        labels = torch.where(force, cls_of, labels)
        labels = torch.where(a2g_max >= matched_thr, cls_of, labels)
        labels = torch.where(a2g_max < unmatched_thr, torch.zeros_like(labels), labels)   # background ...
        labels = torch.where(force, cls_of, labels)                                       # ... but forced matches win (:186-190)
        fg = labels > 0
        matched = torch.gather(gt[:, :, :7], 1, a2g_arg[:, :, None].expand(-1, -1, 7)).reshape(B * A, 7)
        enc = self.box_coder.encode_torch(matched, anchors[None, :, :7].expand(B, -1, -1).reshape(B * A, 7)).view(B, A, -1)
        targets = torch.where(fg[:, :, None], enc, torch.zeros_like(enc))
        if self.norm_by_num_examples:
            n = torch.clamp((labels >= 0).sum(dim=1).float(), min=1.0)
            w = fg.float() / n[:, None]
        else:
            w = fg.float()
        return labels, targets, w
