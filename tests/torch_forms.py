"""TEST INFRASTRUCTURE ONLY — plain-torch forms of the training-mode modules of hvpr_amd.

The product modules (hvpr_amd/bev_backbone.py, vfe.py, map_to_bev.py, anchor_head.py, pointnet2.py) run their training forward
on the library's HIP kernels and raise on anything else; they hold no torch fallback.  The functions here restate the same
forwards with torch ops + torch autograd on the modules' OWN parameters and buffers (same state-dict, same running-statistics
updates), following the reference line by line:

  backbone_train   BaseBEVBackbone_Scale.forward, training branch      pcdet/models/backbones_2d/base_bev_backbone.py:228-279
  vfe_train        PillarVFE_Scale.forward                              pcdet/models/backbones_3d/vfe/pillar_vfe.py:184-221
  memory_train     MemoryUnit_Agg.forward, training branch              pcdet/models/backbones_2d/map_to_bev/memory_module.py:31-59
  topk_points      the top-k of PointPillarScatter...get_score          .../map_to_bev/pointpillar_scatter.py:67-83
  scatter_train    PointPillarScatter_Agg_Memory_1_scale.forward, training branch   .../pointpillar_scatter.py:87-167
  head_train       AnchorHeadSingle.forward, training branch            pcdet/models/dense_heads/anchor_head_single.py:41-108
  assign_targets   AxisAlignedTargetAssigner.assign_targets             .../target_assigner/axis_aligned_target_assigner.py:36-213
  rpn_losses, memory_loss, head_get_loss   the head's losses            .../anchor_head_template.py:101-291, pcdet/utils/loss_utils.py
  sa_forward / fp_forward   PointnetSAModuleMSG / PointnetFPModule with torch shared MLPs (pointnet2_backbone.py:27-47)

They serve (a) as the comparator of the GPU parity tests (whole-module and whole-train-step: tests/test_gpu_train_step_parity.py,
test_gpu_conv_train.py, test_gpu_train_ops.py) and (b) to run the device-agnostic host logic (target assigner, losses, DDP
gradient averaging) on CPU in the `-m "not gpu"` suite.  `patched(model)` swaps them in on a module tree and restores afterwards."""
import contextlib
import math
import types

import numpy as np

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ a11 two-stream backbone
def backbone_train(self, data_dict):
    x, xp, y = data_dict["spatial_features"], data_dict["spatial_features_point"], data_dict["spatial_scale_features"]

    def attention(t, w):                       # spatial_attention.py:57-63
        sp = self.attention.spatial
        pooled = torch.cat((w.max(dim=1, keepdim=True)[0], w.mean(dim=1, keepdim=True)), dim=1)
        return torch.sigmoid(sp.norm(sp.conv(pooled))) * t
    ups, ups_p = [], []
    for i in range(len(self.blocks)):
        x, xp, y = self.blocks[i](x), self.blocks[i](xp), self.scale_layers[i](y)
        xa, xpa = x, xp
        for _ in range(self.sfm_layer_nums[i]):
            xa = attention(self.sfmblocks_down[i](xa), y) + xa
            xpa = attention(self.sfmblocks_down[i](xpa), y) + xpa
        ups.append(self.deblocks[i](xa))
        ups_p.append(self.deblocks[i](xpa))
    data_dict["spatial_features_2d"] = torch.cat(ups, dim=1)
    data_dict["spatial_features_point_2d"] = torch.cat(ups_p, dim=1)
    return data_dict


# ------------------------------------------------------------------------------------------------ a2 pillar VFE (training)
def vfe_train(self, batch_dict, voxels, num, coords):
    md = batch_dict.get("voxel_count_device")
    if md is not None:
        m = int(md.item())
        voxels, num, coords = voxels[:m], num[:m], coords[:m]
        batch_dict["voxels"], batch_dict["voxel_num_points"], batch_dict["voxel_coords"] = voxels, num, coords
        batch_dict["voxel_count_device"] = None
    n = num.to(voxels.dtype)
    c = coords.to(voxels.dtype)
    M, P, _ = voxels.shape
    xyz = voxels[:, :, :3]
    mean = xyz.sum(dim=1, keepdim=True) / n.view(-1, 1, 1)
    centre = c[:, [3, 2, 1]] * voxels.new_tensor(self.voxel_size) + voxels.new_tensor(self.offsets)
    mask = (torch.arange(P, device=voxels.device).view(1, -1) < num.view(-1, 1)).unsqueeze(-1).to(voxels.dtype)
    x = torch.cat([voxels, xyz - mean, xyz - centre.unsqueeze(1)], dim=-1) * mask
    for layer in self.pfn_layers:              # pillar_vfe.py:14-27: BatchNorm1d over all M*P slots, padded ones included
        y = layer.linear(x)
        y = torch.relu(layer.norm(y.permute(0, 2, 1)).permute(0, 2, 1))
        ymax = y.max(dim=1, keepdim=True)[0]
        x = ymax if layer.last_vfe else torch.cat([y, ymax.expand(-1, P, -1)], dim=2)
    s = torch.cat([n.unsqueeze(1), torch.norm(mean, 2, 2), mean.squeeze(1)], dim=-1)
    for seq in self.pfn_scale_layers:          # pillar_vfe.py:213-216
        s = seq(s)
    batch_dict["pillar_features"] = x.reshape(M, -1)
    batch_dict["pillar_scale_features"] = s
    batch_dict["pillar_mask"] = mask
    return batch_dict


# ------------------------------------------------------------------------------------------------ a10 memory + get_score
def hard_shrink_relu(x, lambd=0.0, epsilon=1e-12):
    """memory_module.py:85-87."""
    return (F.relu(x - lambd) * x) / (torch.abs(x - lambd) + epsilon)


def memory_train(self, pillars, k, positives):
    nv, _, d = positives.shape
    att = torch.softmax(F.linear(positives.reshape(-1, d), self.weight), dim=1)          # (nv*k, items), materialised
    if self.shrink_thres > 0:
        att = hard_shrink_relu(att, self.shrink_thres)
        att = F.normalize(att, p=1, dim=1)
    mem = F.linear(att, self.weight.t()).reshape(nv, k, d)
    agg = torch.softmax((mem * pillars.unsqueeze(1)).sum(dim=2), dim=1)
    return {"output": (agg.detach().unsqueeze(2) * mem).sum(dim=1), "att": att}


def topk_points(self, pillars, points):
    """pointpillar_scatter.py:70-73: indices of the top-k over points of softmax(points @ pillars^T, dim=0) = of the raw logits."""
    return torch.topk(pillars @ points.t(), self.k, dim=1)[1]


def get_score(self, points, pillars):
    with torch.no_grad():
        idx = topk_points(self, pillars.detach(), points.detach())
    positives = points[idx]
    w = torch.softmax(torch.bmm(pillars.unsqueeze(1), positives.transpose(1, 2)).squeeze(1), dim=1)
    return (w.detach().unsqueeze(2) * positives).sum(dim=1), positives


def scatter_train(self, batch_dict):
    """pointpillar_scatter.py:87-167 as written there: a loop over the frames with boolean masks, get_score, the memory with the
    positives as its second input (T1), dense canvases filled by column assignment; the pillar half of the memory-fed canvas is
    detached (:140)."""
    pf, sf, coords = batch_dict["pillar_features"], batch_dict["pillar_scale_features"], batch_dict["voxel_coords"]
    point_f, point_c = batch_dict["point_features"], batch_dict["point_coords"]
    B = int(batch_dict["batch_size"]) if "batch_size" in batch_dict else int(coords[:, 0].max().item()) + 1
    cells = self.nx * self.ny
    sp, spp, sc, pos_point, pos_mem = [], [], [], [], []
    for b in range(B):
        vm, pm = coords[:, 0] == b, point_c[:, 0] == b
        c = coords[vm]
        idx = (c[:, 1] + c[:, 2] * self.nx + c[:, 3]).long()
        pillars = pf[vm]
        agg, positives = get_score(self, point_f[pm], pillars)
        mem = self.memory(pillars, self.k, positives)["output"]
        canv = [pf.new_zeros(self.num_bev_features, cells), pf.new_zeros(self.num_bev_features, cells), sf.new_zeros(sf.shape[1], cells)]
        canv[0][:, idx] = torch.cat((pillars.detach(), mem), dim=1).t()
        canv[1][:, idx] = torch.cat((pillars, agg), dim=1).t()
        canv[2][:, idx] = sf[vm].t()
        sp.append(canv[0]); spp.append(canv[1]); sc.append(canv[2])
        pos_point.append(agg); pos_mem.append(mem)
    batch_dict["spatial_features"] = torch.stack(sp, 0).view(B, -1, self.ny, self.nx)
    batch_dict["spatial_features_point"] = torch.stack(spp, 0).view(B, -1, self.ny, self.nx)
    batch_dict["spatial_scale_features"] = torch.stack(sc, 0).view(B, -1, self.ny, self.nx)
    batch_dict["point_positive_features"] = torch.cat(pos_point, 0)
    batch_dict["memory_positive_features"] = torch.cat(pos_mem, 0)
    batch_dict["memory_items"] = self.memory.weight
    return batch_dict


# ------------------------------------------------------------------------------------------------ a6 head (training)
def head_train(self, data_dict):
    fr = self.forward_ret_dict
    heads = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
    for key, suffix in (("spatial_features_2d", ""), ("spatial_features_point_2d", "_point")):
        parts = [h(data_dict[key]).permute(0, 2, 3, 1).contiguous() for h in heads]
        fr["cls_preds" + suffix], fr["box_preds" + suffix] = parts[0], parts[1]
        if self.conv_dir_cls is not None:
            fr["dir_cls_preds" + suffix] = parts[2]
    return self._finish_train(data_dict)



# ------------------------------------------------------------------------------------------------ a12 target assigner
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



def _assign_frames(assigner, anchors, gt, gt_classes, use, matched_thr, unmatched_thr):
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
    enc = assigner.box_coder.encode_torch(matched, anchors[None, :, :7].expand(B, -1, -1).reshape(B * A, 7)).view(B, A, -1)
    targets = torch.where(fg[:, :, None], enc, torch.zeros_like(enc))
    if assigner.norm_by_num_examples:
        n = torch.clamp((labels >= 0).sum(dim=1).float(), min=1.0)
        w = fg.float() / n[:, None]
    else:
        w = fg.float()
    return labels, targets, w


def assign_targets(self, all_anchors, gt_boxes_with_classes):
    """AxisAlignedTargetAssigner.assign_targets (axis_aligned_target_assigner.py:36-111) in torch: a frame is a leading dimension of
    every tensor, four frames per pass (the (frames, ground truths, anchors) IoU intermediates are ~120 MB each at 147 k anchors)."""
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
        same = torch.remainder(gcls - 1, len(self.class_names)) == name_idx      # python-style class_names[c - 1]
        parts = [_assign_frames(self, a, gt_all[b:b + 4, :, :-1], gcls[b:b + 4], (valid & same)[b:b + 4], self.matched[cname],
                                self.unmatched[cname]) for b in range(0, B, 4)]
        lab, tgt, w = (torch.cat([p[i] for p in parts], dim=0) for i in range(3))
        per_class.append((lab.view(B, *fms, -1), tgt.view(B, *fms, -1, self.box_coder.code_size), w.view(B, *fms, -1)))
    return {"box_cls_labels": torch.cat([p[0] for p in per_class], dim=-1).reshape(B, -1),
            "box_reg_targets": torch.cat([p[1] for p in per_class], dim=-2).reshape(B, -1, self.box_coder.code_size),
            "reg_weights": torch.cat([p[2] for p in per_class], dim=-1).reshape(B, -1)}


def head_assign_targets(self, gt_boxes):
    """AnchorHeadTemplate.assign_targets (anchor_head_template.py:89-99) over the torch assigner above (the product's assigner object
    only parses the configuration: class names, thresholds, the box coder)."""
    if self.target_assigner is None:
        from hvpr_amd.target_assigner import AxisAlignedTargetAssigner
        self.target_assigner = AxisAlignedTargetAssigner(self.model_cfg, self.class_names, self.box_coder,
                                                         self.model_cfg.TARGET_ASSIGNER_CONFIG.MATCH_HEIGHT)
    return assign_targets(self.target_assigner, [a.to(gt_boxes.device) for a in self.anchors], gt_boxes)


# ------------------------------------------------------------------------------------------------ a13 losses
def limit_period(val, offset=0.5, period=math.pi):
    return val - torch.floor(val / period + offset) * period


def sigmoid_focal_loss(logits, one_hot, weights, alpha=0.25, gamma=2.0):
    """(B,A,C) logits / one-hot targets, (B,A) weights -> (B,A,C) weighted focal BCE (loss_utils.py:51-72)."""
    p = torch.sigmoid(logits)
    alpha_w = one_hot * alpha + (1 - one_hot) * (1 - alpha)
    pt = one_hot * (1.0 - p) + (1.0 - one_hot) * p
    bce = torch.clamp(logits, min=0) - logits * one_hot + torch.log1p(torch.exp(-torch.abs(logits)))
    return alpha_w * torch.pow(pt, gamma) * bce * weights.unsqueeze(-1)


def weighted_smooth_l1(pred, target, weights, code_weights, beta=1.0 / 9.0):
    """(B,A,7) -> (B,A,7); NaN targets are ignored (loss_utils.py:117-136)."""
    target = torch.where(torch.isnan(target), pred, target)
    d = torch.abs((pred - target) * code_weights.view(1, 1, -1))
    loss = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta) if beta >= 1e-5 else d
    return loss * weights.unsqueeze(-1)


def weighted_cross_entropy(logits, one_hot, weights):
    """(B,A,C) logits, (B,A,C) one-hot, (B,A) weights -> (B,A) (loss_utils.py:188-206)."""
    return F.cross_entropy(logits.permute(0, 2, 1), one_hot.argmax(dim=-1), reduction="none") * weights


def add_sin_difference(a, b, dim=6):
    """sin(a-b) = sin a cos b - cos a sin b, applied to the heading slot (anchor_head_template.py:153-160)."""
    ra = torch.sin(a[..., dim:dim + 1]) * torch.cos(b[..., dim:dim + 1])
    rb = torch.cos(a[..., dim:dim + 1]) * torch.sin(b[..., dim:dim + 1])
    return (torch.cat([a[..., :dim], ra, a[..., dim + 1:]], dim=-1), torch.cat([b[..., :dim], rb, b[..., dim + 1:]], dim=-1))


def direction_targets(anchors, reg_targets, dir_offset, num_bins):
    """One-hot direction bin of the ground-truth heading (anchor_head_template.py:162-176)."""
    rot_gt = reg_targets[..., 6] + anchors[..., 6]
    off = limit_period(rot_gt - dir_offset, 0, 2 * np.pi)
    bins = torch.clamp(torch.floor(off / (2 * np.pi / num_bins)).long(), min=0, max=num_bins - 1)
    return F.one_hot(bins, num_bins).to(anchors.dtype)


_code_w_cache = {}


def _code_weights(values, dtype, device):
    """The code weights as a device tensor, built once per (values, dtype, device): torch.tensor(list, device=...) is a synchronising
    host copy, and this sits in every training step."""
    key = (tuple(float(v) for v in values), dtype, device)
    if key not in _code_w_cache:
        _code_w_cache[key] = torch.tensor(list(key[0]), dtype=dtype, device=device)
    return _code_w_cache[key]


def rpn_losses(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, num_class, num_anchors_per_loc, cfg_weights,
               dir_offset, num_dir_bins):
    """Losses of ONE prediction stream.  cls/box/dir preds are NHWC head outputs; labels (B,A) i32, reg_targets (B,A,7).
    Returns (cls_loss, box_loss (loc + dir), parts dict)."""
    B = cls_preds.shape[0]
    positives = labels > 0
    negatives = labels == 0
    cared = labels >= 0
    pos_norm = torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
    cls_w = (negatives.float() + positives.float()) / pos_norm
    reg_w = positives.float() / pos_norm
    lab = torch.where(positives, torch.ones_like(labels), labels) if num_class == 1 else labels
    tgt = (lab * cared.to(lab.dtype)).long()
    one_hot = F.one_hot(tgt, num_class + 1)[..., 1:].to(cls_preds.dtype)
    cls_loss = sigmoid_focal_loss(cls_preds.reshape(B, -1, num_class), one_hot, cls_w).sum() / B * cfg_weights["cls_weight"]

    bp = box_preds.reshape(B, -1, box_preds.shape[-1] // num_anchors_per_loc)
    code_w = _code_weights(cfg_weights["code_weights"], bp.dtype, bp.device)
    bp_sin, tg_sin = add_sin_difference(bp, reg_targets)
    loc_loss = weighted_smooth_l1(bp_sin, tg_sin, reg_w, code_w).sum() / B * cfg_weights["loc_weight"]
    parts = {"cls": cls_loss, "loc": loc_loss}
    box_loss = loc_loss
    if dir_preds is not None:
        dt = direction_targets(anchors.reshape(1, -1, anchors.shape[-1]).expand(B, -1, -1), reg_targets, dir_offset, num_dir_bins)
        w = positives.to(bp.dtype)
        w = w / torch.clamp(w.sum(-1, keepdim=True), min=1.0)
        dir_loss = weighted_cross_entropy(dir_preds.reshape(B, -1, num_dir_bins), dt, w).sum() / B * cfg_weights["dir_weight"]
        box_loss = box_loss + dir_loss
        parts["dir"] = dir_loss
    return cls_loss, box_loss, parts


def memory_loss(memory_pos, point_pos, mem_weight):
    """MSE(memory features, detached point features) / #pillars (anchor_head_template.py:262-275; the divisor is the
    number of pillars of the batch, kept as the reference wrote it)."""
    return F.mse_loss(memory_pos, point_pos.detach()) / point_pos.shape[0] * mem_weight


def head_get_loss(self):
    """AnchorHeadTemplate.get_loss (anchor_head_template.py:277-291) over the torch forms above."""
    fr = self.forward_ret_dict
    w = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
    anchors = torch.cat(self.anchors, dim=-3).reshape(-1, 7).to(fr["box_preds"].device)
    common = dict(labels=fr["box_cls_labels"], reg_targets=fr["box_reg_targets"], anchors=anchors, num_class=self.num_class,
                  num_anchors_per_loc=self.num_anchors_per_location, cfg_weights=w, dir_offset=self.model_cfg.DIR_OFFSET,
                  num_dir_bins=self.model_cfg.NUM_DIR_BINS)
    cls, box, parts = rpn_losses(fr["cls_preds"], fr["box_preds"], fr.get("dir_cls_preds"), **common)
    cls_p, box_p, parts_p = rpn_losses(fr["cls_preds_point"], fr["box_preds_point"], fr.get("dir_cls_preds_point"), **common)
    mem = memory_loss(fr["pos_memory_feas"], fr["pos_point_feas"], w["mem_weight"])
    tb = {"rpn_loss_cls": parts["cls"], "rpn_loss_loc": parts["loc"], "rpn_loss_cls_pt": parts_p["cls"],
          "rpn_loss_loc_pt": parts_p["loc"], "mem_loss": mem, "rpn_loss": cls + box, "rpn_loss_point": cls_p + box_p}
    if "dir" in parts:
        tb["rpn_loss_dir"], tb["rpn_loss_dir_pt"] = parts["dir"], parts_p["dir"]
    return cls + box, cls_p + box_p, mem, {k: v.detach() for k, v in tb.items()}, fr["memory_items"]

# ------------------------------------------------------------------------------------------------ a9 point-stream modules
def sa_forward_rows(self, xyz, features=None, pre=None):
    """PointnetSAModuleMSG (pointnet2_backbone.py:27-34) with torch gathers, torch 1x1 Conv2d + BatchNorm2d + ReLU and torch max;
    the index tensors (FPS, ball query) come from `pre` or the module's own index kernels."""
    idx, new_xyz, balls = (pre if pre is not None else self.indices(xyz))[:3]
    B = xyz.shape[0]
    ar = torch.arange(B, device=xyz.device)[:, None, None]
    outs = []
    for mlp, bidx in zip(self.mlps, balls):
        bi = bidx.long()                                               # (B, npoint, nsample)
        g = xyz[ar, bi] - new_xyz.unsqueeze(2)                         # (B, npoint, nsample, 3): xyz channels first
        if features is not None:
            g = torch.cat([g, features[ar, bi]], dim=-1)
        f = mlp(g.permute(0, 3, 1, 2))                                 # (B, C', npoint, nsample)
        outs.append(f.max(dim=-1)[0])
    return new_xyz, torch.cat(outs, dim=1).transpose(1, 2).contiguous()


def fp_forward_rows(self, unknown, known, unknow_feats, known_feats, pre=None):
    """PointnetFPModule (pointnet2_backbone.py:40-47, 86-89) with torch gathers and the torch shared MLP."""
    from hvpr_amd import pointnet2
    dist, idx = (pre if pre is not None else pointnet2.three_nn(unknown, known))[:2]
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(dim=2, keepdim=True)
    B = idx.shape[0]
    ar = torch.arange(B, device=idx.device)[:, None, None]
    g = known_feats[ar, idx.long()]                                    # (B, n, 3, C1)
    f = (g[:, :, 0] * w[:, :, 0:1] + g[:, :, 1] * w[:, :, 1:2]) + g[:, :, 2] * w[:, :, 2:3]
    if unknow_feats is not None:
        f = torch.cat([f, unknow_feats], dim=-1)
    return self.mlp(f.transpose(1, 2).unsqueeze(-1)).squeeze(-1).transpose(1, 2).contiguous()


# ------------------------------------------------------------------------------------------------ swapping them in
def _table():
    from hvpr_amd import anchor_head, bev_backbone, map_to_bev, pointnet2, vfe
    return {
        pointnet2.PointnetSAModuleMSG: {"forward_rows": sa_forward_rows},
        pointnet2.PointnetFPModule: {"forward_rows": fp_forward_rows},
        bev_backbone.BaseBEVBackbone_Scale: {"_forward_train": backbone_train},
        vfe.PillarVFE_Scale: {"_forward_train": vfe_train},
        map_to_bev.MemoryUnit_Agg: {"_forward_train": memory_train},
        map_to_bev.PointPillarScatter_Agg_Memory_1_scale: {"get_score": get_score, "_topk_points": topk_points,
                                                           "_forward_train": scatter_train},
        anchor_head.AnchorHeadSingle: {"_forward_train": head_train, "get_loss": head_get_loss, "assign_targets": head_assign_targets},
    }


def patch(root, only=None):
    """Swap the torch forms in on every matching module under `root` (instance attributes; the classes stay untouched).
    only: optional iterable of class names to restrict to.  Returns the list of (module, attribute) that were set."""
    done = []
    table = _table()
    for m in root.modules():
        for cls, fns in table.items():
            if type(m) is cls and (only is None or cls.__name__ in only):
                for name, fn in fns.items():
                    setattr(m, name, types.MethodType(fn, m))
                    done.append((m, name))
    return done


def unpatch(done):
    for m, name in done:
        if name in m.__dict__:
            delattr(m, name)


@contextlib.contextmanager
def patched(root, only=None):
    done = patch(root, only)
    try:
        yield root
    finally:
        unpatch(done)
