"""Training losses of the anchor head, functional form.

Restates pcdet/utils/loss_utils.py — SigmoidFocalClassificationLoss (:9-72), WeightedSmoothL1Loss (:75-136),
WeightedCrossEntropyLoss (:181-206) — and the head's loss assembly, pcdet/models/dense_heads/anchor_head_template.py:101-291.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .common_utils import limit_period


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
