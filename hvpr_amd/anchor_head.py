"""Dense anchor head with the reference's plugin interface (pcdet/models/dense_heads/anchor_head_single.py,
anchor_head_template.py, target_assigner/anchor_generator.py; utils/box_coder_utils.py ResidualCoder).

Eval forward = one fp32-MFMA 1x1 convolution for the three heads (cls | box | dir concatenated on the channel axis)
plus one decode launch (anchors, residual decode, direction fix, class sigmoid/max) instead of ~20 PyTorch kernels."""
import os

import numpy as np
import torch
import torch.nn as nn

from . import kernels
from .folding import FoldCache


class ResidualCoder:
    """utils/box_coder_utils.py:5-77 (plain residual form)."""

    def __init__(self, code_size=7, encode_angle_by_sincos=False, **kwargs):
        assert not encode_angle_by_sincos, "HIP decode: plain angle residual"
        self.code_size = code_size
        self.encode_angle_by_sincos = False

    @staticmethod
    def encode_torch(boxes, anchors):
        anchors[:, 3:6] = torch.clamp_min(anchors[:, 3:6], min=1e-5)     # in place, like the reference (:22-23)
        boxes[:, 3:6] = torch.clamp_min(boxes[:, 3:6], min=1e-5)
        xa, ya, za, dxa, dya, dza, ra, *cas = torch.split(anchors, 1, dim=-1)
        xg, yg, zg, dxg, dyg, dzg, rg, *cgs = torch.split(boxes, 1, dim=-1)
        diag = torch.sqrt(dxa ** 2 + dya ** 2)
        parts = [(xg - xa) / diag, (yg - ya) / diag, (zg - za) / dza, torch.log(dxg / dxa), torch.log(dyg / dya),
                 torch.log(dzg / dza), rg - ra] + [g - a for g, a in zip(cgs, cas)]
        return torch.cat(parts, dim=-1)

    @staticmethod
    def decode_torch(enc, anchors):
        xa, ya, za, dxa, dya, dza, ra, *cas = torch.split(anchors, 1, dim=-1)
        xt, yt, zt, dxt, dyt, dzt, rt, *cts = torch.split(enc, 1, dim=-1)
        diag = torch.sqrt(dxa ** 2 + dya ** 2)
        parts = [xt * diag + xa, yt * diag + ya, zt * dza + za, torch.exp(dxt) * dxa, torch.exp(dyt) * dya,
                 torch.exp(dzt) * dza, rt + ra] + [t + a for t, a in zip(cts, cas)]
        return torch.cat(parts, dim=-1)


class AnchorGenerator:
    """target_assigner/anchor_generator.py:4-60.  Anchor centres come from torch.arange over a float32 range with the
    same numpy scalar promotion as the reference, so that the centres are bit-identical."""

    def __init__(self, anchor_range, anchor_generator_config):
        self.anchor_range = np.asarray(anchor_range, dtype=np.float32)
        self.cfg = anchor_generator_config
        self.anchor_sizes = [c["anchor_sizes"] for c in anchor_generator_config]
        self.anchor_rotations = [c["anchor_rotations"] for c in anchor_generator_config]
        self.anchor_heights = [c["anchor_bottom_heights"] for c in anchor_generator_config]
        self.align_center = [c.get("align_center", False) for c in anchor_generator_config]
        self.num_of_anchor_sets = len(self.anchor_sizes)

    def shifts(self, grid_size, align_center):
        r = self.anchor_range
        gx, gy = np.int64(grid_size[0]), np.int64(grid_size[1])
        if align_center:
            xs, ys = (r[3] - r[0]) / gx, (r[4] - r[1]) / gy
            xo, yo = xs / 2, ys / 2
        else:
            xs, ys = (r[3] - r[0]) / (gx - 1), (r[4] - r[1]) / (gy - 1)
            xo, yo = 0, 0
        x = torch.arange(r[0] + xo, r[3] + 1e-5, step=xs, dtype=torch.float32)
        y = torch.arange(r[1] + yo, r[4] + 1e-5, step=ys, dtype=torch.float32)
        return x, y

    def generate_anchors(self, grid_sizes):
        assert len(grid_sizes) == self.num_of_anchor_sets
        all_anchors, per_loc = [], []
        for gs, sizes, rots, heights, ac in zip(grid_sizes, self.anchor_sizes, self.anchor_rotations, self.anchor_heights,
                                                self.align_center):
            per_loc.append(len(rots) * len(sizes) * len(heights))
            x, y = self.shifts(gs, ac)
            z = torch.tensor(heights, dtype=torch.float32)
            sz = torch.tensor(sizes, dtype=torch.float32)
            rt = torch.tensor(rots, dtype=torch.float32)
            a = torch.zeros(len(z), len(y), len(x), sz.shape[0], rt.shape[0], 7)
            a[..., 0] = x.view(1, 1, -1, 1, 1)
            a[..., 1] = y.view(1, -1, 1, 1, 1)
            a[..., 2] = z.view(-1, 1, 1, 1, 1)
            a[..., 3:6] = sz.view(1, 1, 1, -1, 1, 3)
            a[..., 6] = rt.view(1, 1, 1, 1, -1)
            a[..., 2] += a[..., 5] / 2        # bottom height -> box centre (:58)
            all_anchors.append(a)
        return all_anchors, per_loc


class AnchorHeadTemplate(nn.Module):
    """anchor_head_template.py:11-99 (construction, anchors) and :293-340 (box generation)."""

    def __init__(self, model_cfg, num_class, class_names, grid_size, point_cloud_range, predict_boxes_when_training):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.class_names = class_names
        self.predict_boxes_when_training = predict_boxes_when_training
        self.use_multihead = model_cfg.get("USE_MULTIHEAD", False)
        assert not self.use_multihead, "hvpr path: single head"
        tcfg = model_cfg.TARGET_ASSIGNER_CONFIG
        assert tcfg.BOX_CODER == "ResidualCoder"
        self.box_coder = ResidualCoder(num_dir_bins=tcfg.get("NUM_DIR_BINS", 6), **tcfg.get("BOX_CODER_CONFIG", {}))
        agc = model_cfg.ANCHOR_GENERATOR_CONFIG
        grid_size = np.asarray(grid_size)
        self.anchor_generator = AnchorGenerator(point_cloud_range, agc)
        self.feature_map_sizes = [grid_size[:2] // c["feature_map_stride"] for c in agc]
        anchors, self.num_anchors_per_location = self.anchor_generator.generate_anchors(self.feature_map_sizes)
        dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")   # reference: `.cuda()` at build
        self.anchors = [a.to(dev) for a in anchors]
        self.target_assigner = None
        self.forward_ret_dict = {}

    def _decode_tables(self, device):
        """x/y centre arrays and the per-location anchor table [na,5] for hvpr_head_decode_f32."""
        fms = self.feature_map_sizes[0]
        assert all((f == fms).all() for f in self.feature_map_sizes), "single head: one feature-map size"
        x, y = self.anchor_generator.shifts(fms, self.anchor_generator.align_center[0])
        rows = []
        for sizes, rots, heights in zip(self.anchor_generator.anchor_sizes, self.anchor_generator.anchor_rotations,
                                        self.anchor_generator.anchor_heights):
            assert len(heights) == 1
            for s in sizes:
                for r in rots:
                    zc = torch.tensor(heights[0], dtype=torch.float32) + torch.tensor(s[2], dtype=torch.float32) / 2
                    rows.append([float(zc), s[0], s[1], s[2], r])
        return x.to(device), y.to(device), torch.tensor(rows, dtype=torch.float32, device=device)

    def assign_targets(self, gt_boxes):
        """anchor_head_template.py:89-99 -> AxisAlignedTargetAssigner (device, no host syncs)."""
        if self.target_assigner is None:
            from .target_assigner import AxisAlignedTargetAssigner
            assert self.model_cfg.TARGET_ASSIGNER_CONFIG.NAME == "AxisAlignedTargetAssigner"
            self.target_assigner = AxisAlignedTargetAssigner(self.model_cfg, self.class_names, self.box_coder,
                                                             self.model_cfg.TARGET_ASSIGNER_CONFIG.MATCH_HEIGHT)
        return self.target_assigner.assign_targets([a.to(gt_boxes.device) for a in self.anchors], gt_boxes)

    def _anchor_rot(self, device):
        """Heading of every anchor in the head's order (z, y, x, class, size, rot): the direction targets need it."""
        if getattr(self, "_rot_cache", None) is None or self._rot_cache.device != device:
            self._rot_cache = torch.cat(self.anchors, dim=-3)[..., 6].reshape(-1).to(device=device, dtype=torch.float32).contiguous()
        return self._rot_cache

    def get_loss(self):
        """anchor_head_template.py:277-291: (rpn_loss, rpn_loss_point, mem_loss, tb_dict, memory items) on the library's loss kernels
        (losses.py: one launch per stream for the three losses and their gradients).  tb_dict holds device scalars (the reference
        calls .item() on each: ~10 host syncs per step)."""
        from . import losses
        fr = self.forward_ret_dict
        w = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        common = dict(labels=fr["box_cls_labels"], reg_targets=fr["box_reg_targets"], anchor_rot=self._anchor_rot(fr["box_preds"].device),
                      pos_count=fr["positives_per_frame"], num_class=self.num_class, cfg_weights=w, dir_offset=self.model_cfg.DIR_OFFSET,
                      num_dir_bins=self.model_cfg.NUM_DIR_BINS)
        cls, box, parts = losses.rpn_losses(fr["cls_preds"], fr["box_preds"], fr.get("dir_cls_preds"), **common)
        cls_p, box_p, parts_p = losses.rpn_losses(fr["cls_preds_point"], fr["box_preds_point"], fr.get("dir_cls_preds_point"), **common)
        mem = losses.memory_loss(fr["pos_memory_feas"], fr["pos_point_feas"], w["mem_weight"])
        tb = {"rpn_loss_cls": parts["cls"], "rpn_loss_loc": parts["loc"], "rpn_loss_cls_pt": parts_p["cls"],
              "rpn_loss_loc_pt": parts_p["loc"], "mem_loss": mem, "rpn_loss": cls + box, "rpn_loss_point": cls_p + box_p}
        if "dir" in parts:
            tb["rpn_loss_dir"], tb["rpn_loss_dir_pt"] = parts["dir"], parts_p["dir"]
        return cls + box, cls_p + box_p, mem, {k: v.detach() for k, v in tb.items()}, fr["memory_items"]

    def generate_predicted_boxes(self, batch_size, cls_preds, box_preds, dir_cls_preds=None):
        """torch form of anchor_head_template.py:293-340, kept for callers that hold separate NHWC head outputs."""
        from .common_utils import limit_period
        anchors = torch.cat(self.anchors, dim=-3)
        A = anchors.view(-1, anchors.shape[-1]).shape[0]
        batch_anchors = anchors.view(1, -1, anchors.shape[-1]).repeat(batch_size, 1, 1).to(box_preds.device)
        batch_cls = cls_preds.reshape(batch_size, A, -1).float()
        boxes = self.box_coder.decode_torch(box_preds.reshape(batch_size, A, -1), batch_anchors)
        if dir_cls_preds is not None:
            labels = torch.max(dir_cls_preds.reshape(batch_size, A, -1), dim=-1)[1]
            period = 2 * np.pi / self.model_cfg.NUM_DIR_BINS
            rot = limit_period(boxes[..., 6] - self.model_cfg.DIR_OFFSET, self.model_cfg.DIR_LIMIT_OFFSET, period)
            boxes[..., 6] = rot + self.model_cfg.DIR_OFFSET + period * labels.to(boxes.dtype)
        return batch_cls, boxes


class AnchorHeadSingle(AnchorHeadTemplate):
    """anchor_head_single.py:6-145."""

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range,
                 predict_boxes_when_training=True):
        super().__init__(model_cfg=model_cfg, num_class=num_class, class_names=class_names, grid_size=grid_size,
                         point_cloud_range=point_cloud_range, predict_boxes_when_training=predict_boxes_when_training)
        self.num_anchors_per_location = sum(self.num_anchors_per_location)
        na = self.num_anchors_per_location
        self.conv_cls = nn.Conv2d(input_channels, na * self.num_class, kernel_size=1)
        self.conv_box = nn.Conv2d(input_channels, na * self.box_coder.code_size, kernel_size=1)
        self.num_dir_bins = model_cfg.NUM_DIR_BINS if model_cfg.get("USE_DIRECTION_CLASSIFIER", None) is not None else 0
        self.conv_dir_cls = nn.Conv2d(input_channels, na * self.num_dir_bins, kernel_size=1) if self.num_dir_bins else None
        self.init_weights()
        self._fold = FoldCache()

    def init_weights(self):
        pi = 0.01
        nn.init.constant_(self.conv_cls.bias, -np.log((1 - pi) / pi))
        nn.init.normal_(self.conv_box.weight, mean=0, std=0.001)

    def train(self, mode=True):
        self._fold.invalidate()
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self._fold.invalidate()
        return super()._load_from_state_dict(*a, **k)

    def _build_packed(self, device):
        convs = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
        w = torch.cat([c.weight.detach().float() for c in convs], dim=0)
        b = torch.cat([c.bias.detach().float() for c in convs], dim=0)
        x, y, table = self._decode_tables(device)
        return {"pc": kernels.pack_conv(w, None, b, relu=False, tile_cfg=1), "xs": x, "ys": y, "table": table}

    def _forward_train(self, data_dict):
        """Training forward, anchor_head_single.py:41-108: both streams through the same three 1x1 convolutions, targets assigned
        once.  The three convolutions run as ONE convolution on the library's kernels (forward, data and weight gradient:
        hvpr_amd/conv_train.py), output channels padded to a multiple of 8 with zero rows (the data gradient is a convolution
        with that many input channels); the biases are added by torch.  CPU tensors raise (torch form: tests/torch_forms.py)."""
        from . import conv_train as ct
        fr = self.forward_ret_dict
        heads = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
        f0 = data_dict["spatial_features_2d"]
        if not f0.is_cuda or f0.dtype != torch.float32:
            raise RuntimeError("hvpr_amd: AnchorHeadSingle's training forward needs fp32 GPU tensors (the HIP path has no CPU fallback)")
        w = torch.cat([h.weight for h in heads], dim=0)
        b = torch.cat([h.bias for h in heads], dim=0)
        n_out = w.shape[0]
        pad = (-n_out) % 8
        if pad:
            w = torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], dim=0)
        cuts = np.cumsum([0] + [h.weight.shape[0] for h in heads])
        for key, suffix in (("spatial_features_2d", ""), ("spatial_features_point_2d", "_point")):
            out = ct.conv(data_dict[key].permute(0, 2, 3, 1), w)[..., :n_out] + b
            parts = [out[..., cuts[i]:cuts[i + 1]].contiguous() for i in range(len(heads))]
            fr["cls_preds" + suffix], fr["box_preds" + suffix] = parts[0], parts[1]
            if self.conv_dir_cls is not None:
                fr["dir_cls_preds" + suffix] = parts[2]
        return self._finish_train(data_dict)

    def _finish_train(self, data_dict):
        """The device-agnostic rest of the training forward (anchor_head_single.py:86-108): loss inputs, target assignment, boxes."""
        fr = self.forward_ret_dict
        fr["pos_point_feas"] = data_dict["point_positive_features"]
        fr["pos_memory_feas"] = data_dict["memory_positive_features"]
        fr["memory_items"] = data_dict["memory_items"]
        fr.update(self.assign_targets(gt_boxes=data_dict["gt_boxes"]))
        if self.predict_boxes_when_training:
            data_dict["batch_cls_preds"], data_dict["batch_box_preds"] = self.generate_predicted_boxes(
                data_dict["batch_size"], fr["cls_preds"], fr["box_preds"], fr.get("dir_cls_preds"))
            data_dict["cls_preds_normalized"] = False
        return data_dict

    def forward(self, data_dict):
        if self.training:
            return self._forward_train(data_dict)
        f2d = data_dict["spatial_features_2d"]
        x = f2d.permute(0, 2, 3, 1).contiguous()
        P = self._fold.get(x.device, lambda: self._build_packed(x.device))
        head = kernels.conv2d_nhwc(x, P["pc"])                                   # (B, H, W, na*(nc+7+bins))
        na, nc, nb = self.num_anchors_per_location, self.num_class, self.num_dir_bins
        self.forward_ret_dict["cls_preds"] = head[..., : na * nc]                 # NHWC views, as the reference keeps
        self.forward_ret_dict["box_preds"] = head[..., na * nc: na * (nc + 7)]
        if nb:
            self.forward_ret_dict["dir_cls_preds"] = head[..., na * (nc + 7):]
        period = 2 * np.pi / nb if nb else 1.0
        cls, box, scores, labels = kernels.head_decode(head, na, nc, nb, P["xs"], P["ys"], P["table"],
                                                       self.model_cfg.DIR_OFFSET, self.model_cfg.DIR_LIMIT_OFFSET, period,
                                                       out=data_dict.get("_out_head"))
        data_dict["batch_cls_preds"] = cls
        data_dict["batch_box_preds"] = box
        data_dict["cls_preds_normalized"] = False
        data_dict["batch_max_scores"] = scores       # sigmoid + class max, fused (detector3d_template.py:206-207,241)
        data_dict["batch_max_labels"] = labels
        return data_dict


__all__ = {
    "AnchorHeadTemplate": AnchorHeadTemplate,
    "AnchorHeadSingle": AnchorHeadSingle,
}
