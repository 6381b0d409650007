"""Detector layer with the reference's model API (pcdet/models/detectors/, pcdet/models/__init__.py [absent upstream file],
pcdet/models/model_utils/model_nms_utils.py): module registries keyed by the yaml NAME, the constructor-kwarg contract of
Detector3DTemplate.build_*, ordered module_list, forward(batch_dict) -> batch_dict, post_processing -> pred_dicts."""
import os

import numpy as np
import torch
import torch.nn as nn

from . import anchor_head, bev_backbone, iou3d_nms_utils, kernels, map_to_bev, pointnet2, vfe
from .voxel_generator import VoxelGenerator

backbones_3d_all = pointnet2.__all__   # PointNet2MSG: the training-only point stream
REGISTRY = {
    "backbone_3d": backbones_3d_all,
    "vfe": vfe.__all__,
    "map_to_bev": map_to_bev.__all__,
    "backbone_2d": bev_backbone.__all__,
    "dense_head": anchor_head.__all__,
}


# ---------------------------------------------------------------------------------------------- model_nms_utils
def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.py:6-25, same arguments and return value (selected ids into box_scores, their scores)."""
    n = box_scores.shape[0]
    dev = box_scores.device
    if n == 0:
        return torch.zeros((0,), dtype=torch.long, device=dev), box_scores[:0]
    pre = min(int(nms_config.NMS_PRE_MAXSIZE), n)
    ws = kernels.PostWorkspace(1, n, pre, dev)
    order, _, cnt = kernels.score_topk(box_scores.float().reshape(1, n).contiguous(), score_thresh, pre, ws, want_scores=False)
    keep, kc = kernels.nms_bev(box_preds.float().contiguous(), order[0].contiguous(), cnt, pre, float(nms_config.NMS_THRESH),
                               int(nms_config.NMS_POST_MAXSIZE), ws.nms)
    selected = keep[: int(kc.item())].long()
    return selected, box_scores[selected]


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.py:28-65: per class k — score filter, top NMS_PRE_MAXSIZE, rotated NMS, first NMS_POST_MAXSIZE — on the
    same kernels as the class-agnostic path.  cls_scores (N, num_class), box_preds (N, 7+).  Returns (scores, labels = class
    index k as int64, boxes), classes concatenated in order."""
    pred_scores, pred_labels, pred_boxes = [], [], []
    for k in range(cls_scores.shape[1]):
        col = cls_scores[:, k].contiguous()
        selected, sc = class_agnostic_nms(col, box_preds, nms_config, score_thresh=score_thresh)
        pred_scores.append(sc)
        pred_labels.append(torch.full((selected.shape[0],), k, dtype=torch.long, device=col.device))
        pred_boxes.append(box_preds[selected])
    return torch.cat(pred_scores, dim=0), torch.cat(pred_labels, dim=0), torch.cat(pred_boxes, dim=0)


# ---------------------------------------------------------------------------------------------- detectors
class Detector3DTemplate(nn.Module):
    """detectors/detector3d_template.py:13-132 (construction), :168-318 (post-processing, recall)."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.dataset = dataset
        self.class_names = dataset.class_names
        self.register_buffer("global_step", torch.LongTensor(1).zero_())
        self.module_topology = ["backbone_3d", "vfe", "map_to_bev_module", "pfe", "backbone_2d", "dense_head",
                                "point_head", "roi_head"]
        self._post_ws = None

    @property
    def mode(self):
        return "TRAIN" if self.training else "TEST"

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        info = {
            "module_list": [],
            "num_rawpoint_features": self.dataset.point_feature_encoder.num_point_features,
            "num_point_features": self.dataset.point_feature_encoder.num_point_features,
            "grid_size": self.dataset.grid_size,
            "point_cloud_range": self.dataset.point_cloud_range,
            "voxel_size": self.dataset.voxel_size,
        }
        for name in self.module_topology:
            module, info = getattr(self, "build_%s" % name)(model_info_dict=info)
            self.add_module(name, module)
        return info["module_list"]

    def build_backbone_3d(self, model_info_dict):
        cfg = self.model_cfg.get("BACKBONE_3D", None)
        if cfg is None or cfg.NAME not in backbones_3d_all:
            # the point stream only exists in training (MixAnchor_Memory skips it in eval, pointpillar.py:54)
            return None, model_info_dict
        m = backbones_3d_all[cfg.NAME](model_cfg=cfg, input_channels=model_info_dict["num_point_features"],
                                       grid_size=model_info_dict["grid_size"], voxel_size=model_info_dict["voxel_size"],
                                       point_cloud_range=model_info_dict["point_cloud_range"])
        model_info_dict["module_list"].append(m)
        model_info_dict["num_point_features"] = m.num_point_features
        return m, model_info_dict

    def build_vfe(self, model_info_dict):
        cfg = self.model_cfg.get("VFE", None)
        if cfg is None:
            return None, model_info_dict
        m = vfe.__all__[cfg.NAME](model_cfg=cfg, num_point_features=model_info_dict["num_rawpoint_features"],
                                  point_cloud_range=model_info_dict["point_cloud_range"], voxel_size=model_info_dict["voxel_size"])
        model_info_dict["num_point_features"] = m.get_output_feature_dim()
        model_info_dict["module_list"].append(m)
        return m, model_info_dict

    def build_map_to_bev_module(self, model_info_dict):
        cfg = self.model_cfg.get("MAP_TO_BEV", None)
        if cfg is None:
            return None, model_info_dict
        m = map_to_bev.__all__[cfg.NAME](model_cfg=cfg, grid_size=model_info_dict["grid_size"])
        model_info_dict["module_list"].append(m)
        model_info_dict["num_bev_features"] = m.num_bev_features
        return m, model_info_dict

    def build_backbone_2d(self, model_info_dict):
        cfg = self.model_cfg.get("BACKBONE_2D", None)
        if cfg is None:
            return None, model_info_dict
        m = bev_backbone.__all__[cfg.NAME](model_cfg=cfg, input_channels=model_info_dict["num_bev_features"])
        model_info_dict["module_list"].append(m)
        model_info_dict["num_bev_features"] = m.num_bev_features
        return m, model_info_dict

    def build_dense_head(self, model_info_dict):
        cfg = self.model_cfg.get("DENSE_HEAD", None)
        if cfg is None:
            return None, model_info_dict
        m = anchor_head.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=model_info_dict["num_bev_features"],
            num_class=self.num_class if not cfg.CLASS_AGNOSTIC else 1, class_names=self.class_names,
            grid_size=model_info_dict["grid_size"], point_cloud_range=model_info_dict["point_cloud_range"],
            predict_boxes_when_training=self.model_cfg.get("ROI_HEAD", False))
        model_info_dict["module_list"].append(m)
        return m, model_info_dict

    def _absent(self, key, model_info_dict):
        assert self.model_cfg.get(key, None) is None, f"{key} is outside the hvpr hot path"
        return None, model_info_dict

    def build_pfe(self, model_info_dict):
        return self._absent("PFE", model_info_dict)

    def build_point_head(self, model_info_dict):
        return self._absent("POINT_HEAD", model_info_dict)

    def build_roi_head(self, model_info_dict):
        return self._absent("ROI_HEAD", model_info_dict)

    def forward(self, **kwargs):
        raise NotImplementedError

    # ------------------------------------------------------------------------------------------ post-processing
    def post_processing(self, batch_dict, sync=True):
        """detector3d_template.py:168-274, class-agnostic single-head branch, all on device.

        sync=True  -> the reference's return: pred_dicts[i] = {pred_boxes (n,7), pred_scores (n,), pred_labels (n,)}.
        sync=False -> no host read-back: every tensor is padded to NMS_POST_MAXSIZE rows and pred_count (device i32)
                      says how many are live (rows past it are copies of anchor 0 and must be ignored).
        """
        cfg = self.model_cfg.POST_PROCESSING
        ncfg = cfg.NMS_CONFIG
        if ncfg.MULTI_CLASSES_NMS:
            return self._post_processing_multi_class(batch_dict)
        B = batch_dict["batch_size"]
        boxes_all = batch_dict["batch_box_preds"]
        assert boxes_all.dim() == 3 and boxes_all.shape[0] == B
        cls_all = batch_dict["batch_cls_preds"]
        assert cls_all.shape[2] in (1, self.num_class)
        if "batch_max_scores" in batch_dict:
            scores, labels = batch_dict["batch_max_scores"], batch_dict["batch_max_labels"]
        else:
            c = cls_all if batch_dict["cls_preds_normalized"] else torch.sigmoid(cls_all)
            scores, lab = torch.max(c, dim=-1)
            labels = (lab + 1).to(torch.int32)
        A = scores.shape[1]
        pre, post = min(int(ncfg.NMS_PRE_MAXSIZE), A), int(ncfg.NMS_POST_MAXSIZE)
        dev = scores.device
        if self._post_ws is None or (self._post_ws.batch, self._post_ws.n_scores, self._post_ws.pre_max) != (B, A, pre) \
                or self._post_ws.topk.device != dev:
            self._post_ws = kernels.PostWorkspace(B, A, pre, dev)
        order, _, counts = kernels.score_topk(scores.contiguous(), cfg.SCORE_THRESH, pre, self._post_ws, want_scores=False)
        pred_dicts, recall_dict = [], {}
        for b in range(B):
            keep, kc = kernels.nms_bev(boxes_all[b], order[b], counts[b:b + 1], pre, float(ncfg.NMS_THRESH), post,
                                       self._post_ws.nms)
            if labels.dtype == torch.int32 and not cfg.OUTPUT_RAW_SCORE and boxes_all[b].is_contiguous():
                pb, ps, pl, sel = kernels.gather_predictions(boxes_all[b], scores[b].contiguous(), labels[b].contiguous(), keep)
                rec = {"pred_boxes": pb, "pred_labels": pl, "pred_scores": ps, "selected": sel, "pred_count": kc}
            else:
                sel = keep.long()
                rec = {"pred_boxes": boxes_all[b].index_select(0, sel), "pred_labels": labels[b].index_select(0, sel).long(),
                       "pred_scores": (cls_all[b].max(dim=-1)[0] if cfg.OUTPUT_RAW_SCORE else scores[b]).index_select(0, sel),
                       "selected": sel, "pred_count": kc}
            if sync:
                n = int(kc.item())
                rec = {k: (v[:n] if k != "pred_count" else v) for k, v in rec.items()}
                recall_dict = self.generate_recall_record(rec["pred_boxes"], recall_dict, b, batch_dict, cfg.RECALL_THRESH_LIST)
            pred_dicts.append(rec)
        return pred_dicts, recall_dict, batch_dict

    def _post_processing_multi_class(self, batch_dict):
        """MULTI_CLASSES_NMS branch, detector3d_template.py:214-239 (single head): one NMS per class on the sigmoid scores.
        Labels are 1-based class ids (the reference's single-head label mapping `arange(1, num_class)` is one short for its own
        assert at :224; `arange(1, num_class + 1)` is what upstream OpenPCDet has).  Host-synchronising, like the reference."""
        cfg = self.model_cfg.POST_PROCESSING
        pred_dicts, recall_dict = [], {}
        for b in range(batch_dict["batch_size"]):
            cls = batch_dict["batch_cls_preds"][b]
            cls = cls if batch_dict["cls_preds_normalized"] else torch.sigmoid(cls)
            boxes = batch_dict["batch_box_preds"][b]
            mapping = torch.arange(1, cls.shape[1] + 1, device=cls.device)
            sc, lab, bx = multi_classes_nms(cls.float(), boxes, cfg.NMS_CONFIG, score_thresh=cfg.SCORE_THRESH)
            rec = {"pred_boxes": bx, "pred_scores": sc, "pred_labels": mapping[lab]}
            recall_dict = self.generate_recall_record(bx, recall_dict, b, batch_dict, cfg.RECALL_THRESH_LIST)
            pred_dicts.append(rec)
        return pred_dicts, recall_dict, batch_dict

    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
        """detector3d_template.py:276-318 (no ROI head on this path, so the roi_* counters stay 0).  Trailing all-zero gt rows
        are cut with the reference's `while k > 0`, which never cuts row 0: a frame without ground truth counts one zero box.
        Two host reads per frame (the gt rows' sums, the recalled counts of all thresholds) instead of one per row / threshold."""
        if "gt_boxes" not in data_dict:
            return recall_dict
        gt = data_dict["gt_boxes"][batch_index]
        if len(recall_dict) == 0:
            recall_dict = {"gt": 0}
            for t in thresh_list:
                recall_dict["roi_%s" % str(t)] = 0
                recall_dict["rcnn_%s" % str(t)] = 0
        row_is_zero = (gt.sum(dim=1) == 0).tolist()
        k = len(row_is_zero) - 1
        while k > 0 and row_is_zero[k]:
            k -= 1
        gt = gt[:k + 1]
        if gt.shape[0] > 0:
            if box_preds.shape[0] > 0:
                best = iou3d_nms_utils.boxes_iou3d_gpu(box_preds[:, 0:7], gt[:, 0:7]).max(dim=0)[0]
                thr = torch.tensor([float(t) for t in thresh_list], dtype=best.dtype).to(best.device, non_blocking=True)
                recalled = (best.unsqueeze(0) > thr.unsqueeze(1)).sum(dim=1).tolist()
                for t, n in zip(thresh_list, recalled):
                    recall_dict["rcnn_%s" % str(t)] += int(n)
            recall_dict["gt"] += gt.shape[0]
        return recall_dict

    # ------------------------------------------------------------------------------------------ checkpoints
    def load_params_from_file(self, filename, logger=None, to_cpu=False):
        """detector3d_template.py:320-346: load by key + shape match."""
        ckpt = torch.load(filename, map_location=torch.device("cpu") if to_cpu else None)
        disk = ckpt["model_state"]
        state = self.state_dict()
        update = {k: v for k, v in disk.items() if k in state and state[k].shape == v.shape}
        state.update(update)
        self.load_state_dict(state)
        from . import conv_train
        conv_train.weights_changed()         # (load_state_dict bumps the versions itself; a restore is rare enough to be explicit)
        if logger is not None:
            for k in state:
                if k not in update:
                    logger.info("Not updated weight %s: %s" % (k, str(state[k].shape)))
            logger.info("==> Done (loaded %d/%d)" % (len(update), len(state)))
        return len(update), len(state)


    def load_params_with_optimizer(self, filename, to_cpu=False, optimizer=None, logger=None):
        """detector3d_template.py:348-375: resume — strict model load, optimizer state from the same file (or the side file
        `<name>_optim.<ext>` older checkpoints used).  Returns (it, epoch) like the reference."""
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        ckpt = torch.load(filename, map_location=torch.device("cpu") if to_cpu else None)
        self.load_state_dict(ckpt["model_state"])
        if optimizer is not None:
            ost = ckpt.get("optimizer_state")
            if ost is None:
                stem, ext = os.path.splitext(filename)
                side = "%s_optim%s" % (stem, ext)
                if os.path.exists(side):
                    ost = torch.load(side, map_location=torch.device("cpu") if to_cpu else None)["optimizer_state"]
            if ost is not None:
                optimizer.load_state_dict(ost)
        if logger is not None:
            logger.info("==> Resumed from %s (epoch %s, it %s, version %s)" % (filename, ckpt.get("epoch", -1), ckpt.get("it", 0.0),
                                                                              ckpt.get("version", "none")))
        return ckpt.get("it", 0.0), ckpt.get("epoch", -1)


class _VoxelizingDetector(Detector3DTemplate):
    """Adds the step the north_star moves on-device: when batch_dict carries raw `points` (N,5) [b,x,y,z,r] and no
    `voxels`, voxelize on the GPU (replaces the CPU dataloader step data_processor.py:43-75)."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self._voxgen = {}            # one generator per mode: MAX_NUMBER_OF_VOXELS differs between train and test

    def _voxel_cfg(self):
        for p in self.dataset.dataset_cfg.DATA_PROCESSOR:
            if p.NAME == "transform_points_to_voxels":
                return p
        raise KeyError("transform_points_to_voxels")

    def _voxel_generator(self, device):
        mode = "train" if self.training else "test"
        key = (mode, torch.device(device))
        if key not in self._voxgen:
            vc = self._voxel_cfg()
            self._voxgen[key] = VoxelGenerator(vc.VOXEL_SIZE, self.dataset.point_cloud_range, vc.MAX_POINTS_PER_VOXEL,
                                               vc.MAX_NUMBER_OF_VOXELS[mode], device=device)
        return self._voxgen[key]

    @staticmethod
    def _frame_offsets(batch_dict):
        offs = batch_dict.get("point_frame_offsets")
        if offs is None:   # frames are contiguous and ordered (dataset.py:161-166); count per frame on device
            pts, B = batch_dict["points"], batch_dict["batch_size"]
            if pts.is_cuda and pts.dtype == torch.float32 and pts.is_contiguous():
                offs = kernels.frame_offsets(pts, B)
            else:
                offs = torch.searchsorted(pts[:, 0].contiguous(), torch.arange(B + 1, device=pts.device, dtype=pts.dtype)).to(torch.int32)
        return offs

    def voxelize_on_device(self, batch_dict):
        pts = batch_dict["points"]
        B = batch_dict["batch_size"]
        offs = self._frame_offsets(batch_dict)
        v, c, n, vo = self._voxel_generator(pts.device).generate_batch(pts, offs, B, xyz_col=1, n_feat=pts.shape[1] - 1)
        batch_dict["voxels"], batch_dict["voxel_coords"], batch_dict["voxel_num_points"] = v, c, n
        batch_dict["voxel_offsets"] = vo
        batch_dict["voxel_count_device"] = vo[B:B + 1]     # live row count, read on device by the next kernels
        return batch_dict


class MixAnchor_Memory(_VoxelizingDetector):
    """detectors/pointpillar.py:36-68."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()
        # The fused eval forward does not need the padded `voxels` (ΣM, 32, 4) tensor and `pillar_mask` the reference's data
        # loader / VFE leave in batch_dict (nothing downstream reads them: data_processor.py:43-75 -> pillar_vfe.py:184-221 are
        # one entry point here).  Set to True to have hvpr_encode_fwd_f32 materialise them as well (tests do).
        self.export_voxels = bool(model_cfg.get("EXPORT_VOXELS", False))
        self.index_mode = int(model_cfg.get("ENCODE_INDEX_MODE", 1))      # hvpr_encode_fwd_f32's index_mode (see encode_fused)

    def get_training_loss(self):
        """detectors/pointpillar.py:59-68 with the arity decision of SURVEY.md T2: loss = rpn + rpn_point + mem."""
        rpn, rpn_point, mem, tb_dict, items = self.dense_head.get_loss()
        loss = rpn + rpn_point + mem
        tb_dict = {"loss_rpn": rpn.detach(), **tb_dict}
        return loss, tb_dict, {"items": items}

    def _can_fuse_encode(self, batch_dict):
        """The fused a1..a4 entry point covers the hvpr.yaml shapes: raw (N,5) points in, PillarVFE_Scale 10->16, 32->64 with the
        5->16->32 scale stream, a 64-wide memory bank, nz == 1, <= 32 points per voxel.  Anything else takes the module
        chain (three C-ABI calls) — same kernels' arithmetic, same results."""
        if self.training or "voxels" in batch_dict:
            return False
        v, m = getattr(self, "vfe", None), getattr(self, "map_to_bev_module", None)
        if not isinstance(v, vfe.PillarVFE_Scale) or not isinstance(m, map_to_bev.PointPillarScatter_Agg_Memory_1_scale):
            return False
        pts = batch_dict["points"]
        return (pts.is_cuda and pts.dim() == 2 and pts.shape[1] == 5 and v.num_filters == [32, 64] and
                v.num_scale_features == [16, 32] and tuple(m.memory.weight.shape[1:]) == (64,) and m.nz == 1 and
                int(self._voxel_cfg().MAX_POINTS_PER_VOXEL) <= 32 and m.memory.weight.shape[0] <= 2048 and m.k <= 32)

    def encode_fused(self, batch_dict):
        """a1-a4 through hvpr_encode_fwd_f32: fills every batch_dict key the module chain voxelize_on_device -> vfe ->
        map_to_bev_module would.  The detector has ONE encode lane per device (the eager forward on the current stream; one encode
        per step of the frame pipeline), which is what index_mode 1 asks of its caller (include/hvpr_amd.h): three launches
        instead of five.  `self.index_mode = 0` switches to the form that is safe under any concurrency."""
        pts, B = batch_dict["points"].contiguous(), batch_dict["batch_size"]
        vg = self._voxel_generator(pts.device)
        m = self.map_to_bev_module
        r = kernels.encode_fwd(pts, self._frame_offsets(batch_dict), B, vg.point_cloud_range, vg.voxel_size, vg.grid_size,
                               vg.max_num_points, vg.max_voxels, vg._workspace(B, pts.shape[0]),
                               self.vfe._fold.get(pts.device, self.vfe._build_folded), self.vfe.offsets,
                               m.memory.packed_bank(), m.k, xyz_col=1, cap_mode=vg.cap_mode, want_voxels=self.export_voxels,
                               want_mask=self.export_voxels, out=batch_dict.get("_out_spatial"), state=batch_dict.get("_canvas_state"),
                               index_mode=self.index_mode)
        vo = r["voxel_offsets"]
        batch_dict.update(voxels=r["voxels"], voxel_coords=r["coords"], voxel_num_points=r["num_points"], voxel_offsets=vo,
                          voxel_count_device=vo[B:B + 1], pillar_features=r["pillar_features"],
                          pillar_scale_features=r["pillar_scale_features"], pillar_mask=r["pillar_mask"],
                          spatial_features=r["spatial"], spatial_scale_features=r["spatial_scale"])
        return batch_dict

    def persistent_canvases(self, example_batch):
        """{"_out_spatial", "_canvas_state"} for callers that replay the eval forward on fixed buffers (hipGraphs, the frame
        pipeline): a zeroed canvas pair that stays theirs plus its occupancy state, so that encode_fused clears only the cells
        the previous frame left behind (~2.4 MB instead of 47 MB per hvpr_car frame).  {} when the fused path does not apply."""
        if not self._can_fuse_encode(example_batch):
            return {}
        m = self.map_to_bev_module
        canv, state = kernels.canvas_buffers(example_batch["batch_size"], m.nx, m.ny, example_batch["points"].device)
        return {"_out_spatial": canv, "_canvas_state": state}

    # the eval forward in the three stages a frame pipeline overlaps (PipelinedForward)
    def stage_encode(self, batch_dict):
        """a1-a4: points -> BEV canvases."""
        if self._can_fuse_encode(batch_dict):
            return self.encode_fused(batch_dict)
        if "voxels" not in batch_dict:
            batch_dict = self.voxelize_on_device(batch_dict)
        return self.map_to_bev_module(self.vfe(batch_dict))

    def stage_dense(self, batch_dict):
        """a5-a7: canvases -> decoded boxes and scores of every anchor."""
        return self.dense_head(self.backbone_2d(batch_dict))

    def stage_dense_a(self, batch_dict, split):
        """First half of stage_dense on caller-owned boundary buffers (BaseBEVBackbone_Scale.split_buffers): the trunk and every
        branch but the last."""
        return self.backbone_2d({**batch_dict, "_bev_split": {**split, "phase": "a"}})

    def stage_dense_b(self, batch_dict, split):
        """Second half: the last level's branch, head and decode."""
        bd = self.backbone_2d({**batch_dict, "_bev_split": {**split, "phase": "b"}})
        bd.pop("_bev_split", None)
        return self.dense_head(bd)

    def prefetch_point_indices(self, batch_dict):
        """Training: compute the point stream's index tensors (FPS, ball query, three-NN: they depend on the coordinates only)
        for `batch_dict` on a side stream and park them in it; the forward of that batch then skips them.  Called for the NEXT
        batch before the current step is enqueued, the ~16 ms of latency-bound index kernels (FPS alone is 4096 dependent
        iterations on 16 CUs) run beside the current step instead of in front of the next one.  Results are identical."""
        pn = getattr(self, "backbone_3d", None)
        if pn is None or not hasattr(pn, "index_plan") or "points" not in batch_dict or not batch_dict["points"].is_cuda:
            return batch_dict
        if getattr(self, "_pn2_stream", None) is None:
            self._pn2_stream = torch.cuda.Stream()
        s = self._pn2_stream
        s.wait_stream(torch.cuda.current_stream())      # the batch (and whatever produced it) is in front of us
        with torch.cuda.stream(s):
            plan = pn.index_plan(batch_dict["points"], batch_dict["batch_size"])
            ready = torch.cuda.Event()
            ready.record(s)
        batch_dict["points"].record_stream(s)
        batch_dict["_pn2_plan"] = (plan, ready)
        return batch_dict

    def forward(self, batch_dict, sync=True):
        fused = self._can_fuse_encode(batch_dict)
        if fused:
            batch_dict = self.encode_fused(batch_dict)
        elif "voxels" not in batch_dict:
            batch_dict = self.voxelize_on_device(batch_dict)
        if self.training:
            for m in self.module_list:            # point stream first (module_topology), then the pillar stream
                batch_dict = m(batch_dict)
            loss, tb_dict, disp_dict = self.get_training_loss()
            return {"loss": loss}, tb_dict, disp_dict
        # eval skips the point stream: module_list[1:] in the reference (pointpillar.py:54); here the point stream is
        # only put in module_list when it exists, so skip it by type
        for m in self.module_list:
            if m is getattr(self, "backbone_3d", None) or (fused and (m is self.vfe or m is self.map_to_bev_module)):
                continue
            batch_dict = m(batch_dict)
        out = self.post_processing(batch_dict, sync=sync)
        if fused and sync and self.index_mode == 1:
            # the results have just been read back (host sync): also look at the voxelizer workspace's error word — a one-launch
            # index kernel that had to give up a wait reported zero pillars; say so instead of returning an empty frame
            self._voxel_generator(batch_dict["points"].device)._ws.status()
        return out


class _CapturedState:
    """What a captured hipGraph bakes in as raw pointers without owning it: folded / packed weights (FoldCache values, the
    packed memory bank), the post-processing and voxelizer workspaces.  The wrapper (a) keeps a reference to every such object
    as it was at capture time, so the memory a replay touches can never be freed under it, and (b) refuses to replay once the
    model has replaced any of them (train()/eval() toggles, load_state_dict, set_conv_precision, a bigger batch elsewhere):
    the replay would silently use stale weights or a workspace the model no longer looks at."""

    def __init__(self, model):
        self.slots = []
        for m in model.modules():
            if hasattr(m, "_fold") and hasattr(m._fold, "value"):
                self.slots.append((m._fold, "value"))
            if hasattr(m, "_packed"):
                self.slots.append((m, "_packed"))
        self.slots.append((model, "_post_ws"))
        for vg in getattr(model, "_voxgen", {}).values():
            self.slots.append((vg, "_ws"))
        self.keep = [getattr(o, a) for o, a in self.slots]        # strong references: ids below cannot be recycled

    def check(self):
        for (o, a), was in zip(self.slots, self.keep):
            if getattr(o, a) is not was:
                raise RuntimeError(f"hvpr_amd: {type(o).__name__}.{a} was rebuilt after this hipGraph was captured (train()/eval(), "
                                   "load_state_dict, set_conv_precision or a workspace re-allocation): capture a new graph")


class GraphedForward:
    """Whole-frame hipGraph of the eval forward (voxelize -> ... -> NMS) for a fixed input shape.

    Everything data dependent on this path (pillar count, candidate count, keep count) lives in device words that the
    kernels read, so the launch sequence is static and can be captured once and replayed: ~45 launches per frame become
    one graph launch and the host is out of the loop.  Inputs are copied into static buffers; outputs are the padded
    `sync=False` tensors of post_processing (valid until the next replay).
    """

    def __init__(self, model, example_batch, warmup=3):
        assert not model.training
        self.model = model
        self.static_in = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        if hasattr(model, "persistent_canvases"):
            self.static_in.update(model.persistent_canvases(example_batch))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):            # lazy init (workspaces, folded weights, occupancy queries) outside capture
                model(dict(self.static_in), sync=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = model(dict(self.static_in), sync=False)
        self.captured = _CapturedState(model)

    def __call__(self, batch):
        self.captured.check()
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static_in[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.static_out

    check_status = None      # (assigned below: the same method as PipelinedForward's)


class PipelinedForward:
    """Frame pipeline of the eval forward for throughput: step k runs, concurrently on HIP streams inside ONE graph replay,
    stage_encode of frame k, the convolutions of earlier frames and post_processing of the oldest frame in flight.

    depth 3: encode(k) | stage_dense(k-1) | post(k-2).
    depth 4 (default for the fp32 kernels): encode(k) | stage_dense_a(k-1) | stage_dense_b(k-2) | post(k-3) — the last level's
    branch (three SFM convolutions, deconvolution: 0.6 tiles per workgroup slot at batch 1), head and decode of one frame run
    beside the trunk of the next one instead of alone at the end of the stage.

    Why: at batch 1 the encode group and top-k + NMS are latency-bound chains of small kernels that leave most of the 256 CUs
    idle; the convolutions are throughput-bound but their upper levels do not fill the chip.  Results are bit-identical to the
    serial forward (same kernels, same inputs; tests/test_gpu_e2e.py); the latency of one frame becomes `depth` steps.

    Two lanes of boundary buffers (canvases, backbone boundary, head outputs) alternate, so there are two graphs (even / odd
    steps).  __call__(batch) enqueues frame k and returns the `sync=False` result of frame k - depth + 1 (None for the first
    depth - 1 calls); flush() drains the frames still in flight."""

    _HEAD_KEYS = ("batch_cls_preds", "batch_box_preds", "batch_max_scores", "batch_max_labels")

    def __init__(self, model, example_batch, warmup=2, depth=None):
        assert not model.training and hasattr(model, "stage_encode")
        if depth is None:
            fp32 = getattr(model.backbone_2d, "conv_precision", "fp32") == "fp32" and hasattr(model.backbone_2d, "split_buffers")
            depth = 4 if fp32 else 3
        assert depth in (3, 4)
        self.model, self.step, self.depth = model, 0, depth
        self.B = example_batch["batch_size"]
        self.inp = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()} for _ in range(2)]
        self.canvas, self.head, self.aux, self.split = [None, None], [None, None], [None, None], [None, None]
        self.enc, self.f2d = [None, None], [None, None]
        # each lane owns its canvases for good, so the encode stage only clears what the lane's previous frame left in them
        self.own = [model.persistent_canvases(example_batch) if hasattr(model, "persistent_canvases") else {} for _ in range(2)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for p in range(2):                       # eager runs: lazy init + the persistent boundary buffers of each lane
                for _ in range(warmup):
                    bd = model.stage_encode({**self.inp[p], **self.own[p]})
                    self.canvas[p] = (bd["spatial_features"], bd["spatial_scale_features"])
                    if depth == 4:
                        if self.split[p] is None:
                            sp = bd["spatial_features"]
                            self.split[p] = model.backbone_2d.split_buffers(self.B, sp.shape[2], sp.shape[3], sp.device)
                        model.stage_dense_a(bd, self.split[p])
                        bd = model.stage_dense_b(bd, self.split[p])
                    else:
                        bd = model.stage_dense(bd)
                    self.head[p] = tuple(bd[k] for k in self._HEAD_KEYS)
                    model.post_processing(bd, sync=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.s_enc, self.s_post, self.s_b = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        self.graphs, self.out = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()], [None, None]
        pool = None
        for p in range(2):
            q = 1 - p
            # step k has parity p: frame k uses lane p, frame k-1 lane q, frame k-2 lane p, frame k-3 lane q
            post_lane = p if depth == 3 else q
            with torch.cuda.graph(self.graphs[p], pool=pool), torch.no_grad():
                main = torch.cuda.current_stream()
                self.s_enc.wait_stream(main)
                self.s_post.wait_stream(main)
                with torch.cuda.stream(self.s_post):      # oldest frame in flight: top-k + NMS
                    self.out[p] = model.post_processing(self._post_dict(post_lane), sync=False)[0]
                with torch.cuda.stream(self.s_enc):       # frame k (lane p): points -> canvases of lane p
                    bd = model.stage_encode({**self.inp[p], "_out_spatial": self.canvas[p], **self.own[p]})
                    self.aux[p] = bd["voxel_offsets"]
                    self.enc[p] = {k: bd[k] for k in ("voxel_coords", "voxel_num_points", "voxel_offsets", "pillar_features",
                                                      "pillar_scale_features") if bd.get(k) is not None}
                dense_in = {"batch_size": self.B, "spatial_features": self.canvas[q][0], "spatial_scale_features": self.canvas[q][1]}
                if depth == 3:
                    # frame k-1 (lane q): convolutions on the capture stream (+ the backbone's own branch streams)
                    self.f2d[q] = model.stage_dense({**dense_in, "_out_head": self.head[q]})["spatial_features_2d"]
                else:
                    self.s_b.wait_stream(main)
                    with torch.cuda.stream(self.s_b):     # frame k-2 (lane p): last branch + head + decode
                        model.stage_dense_b({"batch_size": self.B, "_out_head": self.head[p]}, self.split[p])
                    model.stage_dense_a(dense_in, self.split[q])          # frame k-1 (lane q): trunk + the other branches
                    main.wait_stream(self.s_b)
                main.wait_stream(self.s_enc)
                main.wait_stream(self.s_post)
            pool = self.graphs[p].pool()
        self.captured = _CapturedState(model)

    def _post_dict(self, p):
        d = dict(zip(self._HEAD_KEYS, self.head[p]))
        d.update(batch_size=self.B, cls_preds_normalized=False)
        return d

    def __call__(self, batch):
        self.captured.check()
        p = self.step & 1
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.inp[p][k].copy_(v, non_blocking=True)
        self.graphs[p].replay()
        self.step += 1
        return self.out[p] if self.step >= self.depth else None

    def inspect(self, batch):
        """Parity hook (bench.py's parity gates, tests): push ONE frame through every stage of THIS pipeline — `depth` steps with the
        same input, after which every lane's boundary buffers hold that frame — synchronise, and return copies of the encode
        stage's outputs, the canvases, the 384-channel feature map, the head's decoded outputs and the frame's post-processing
        result.  Nothing is recomputed outside the captured graphs: these are the tensors the timed replays produce."""
        for _ in range(self.depth):
            out = self(batch)
        torch.cuda.synchronize()
        self.check_status()
        p = (self.step - 1) & 1                     # lane the last step encoded into; its head buffers were written in the same step
        q = 1 - p
        d = {k: v.clone() for k, v in self.enc[p].items()}
        d["spatial_features"], d["spatial_scale_features"] = self.canvas[p][0].clone(), self.canvas[p][1].clone()
        if self.depth == 4:
            d["spatial_features_2d"] = self.split[p]["out"].permute(0, 3, 1, 2).clone()
            head_now, head_post = self.head[p], self.head[q]
        else:
            d["spatial_features_2d"] = self.f2d[q].clone()
            head_now, head_post = self.head[q], self.head[p]
        d.update({k: v.clone() for k, v in zip(self._HEAD_KEYS, head_now)})
        # the post-processing of the last step read the OTHER lane's head buffers (the same frame, one step earlier)
        d["post_input"] = {k: v.clone() for k, v in zip(self._HEAD_KEYS, head_post)}
        d["post"] = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in rec.items()} for rec in out]
        return d

    def check_status(self):
        """SYNCHRONISES: raises when the voxelizer workspace's error word is up (a one-launch index kernel gave up a wait:
        include/hvpr_amd.h, hvpr_voxelize_workspace_status).  Call it wherever the results of a run are read back."""
        for vg in getattr(self.model, "_voxgen", {}).values():
            if vg._ws is not None:
                vg._ws.status()

    def flush(self):
        """depth - 1 more steps (re-encoding the last inputs, whose results are dropped): yields the results of the frames still in
        flight, oldest first.  Each result is only valid until the next step on its lane — consume it before the next."""
        for _ in range(self.depth - 1):
            p = self.step & 1
            self.graphs[p].replay()
            self.step += 1
            if self.step >= self.depth:
                yield self.out[p]


GraphedForward.check_status = PipelinedForward.check_status


class PointPillar(_VoxelizingDetector):
    """detectors/pointpillar.py:4-34."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()

    def forward(self, batch_dict, sync=True):
        if self.training:
            raise NotImplementedError("hvpr_amd: plain PointPillar has no training path (the reference's own detectors/pointpillar.py:24-32 unpacks "
                                      "two values from a get_loss that returns five, anchor_head_template.py:291); train MixAnchor_Memory")
        if "voxels" not in batch_dict:
            batch_dict = self.voxelize_on_device(batch_dict)
        for m in self.module_list:
            batch_dict = m(batch_dict)
        return self.post_processing(batch_dict, sync=sync)


__all__ = {
    "Detector3DTemplate": Detector3DTemplate,
    "PointPillar": PointPillar,
    "MixAnchor_Memory": MixAnchor_Memory,
}


def build_detector(model_cfg, num_class, dataset):
    """detectors/__init__.py:11-16."""
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)


# ---------------------------------------------------------------------------------------------- pcdet/models/__init__.py
def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict):
    """Every ndarray except the host-object keys becomes a float32 device tensor."""
    for k, v in batch_dict.items():
        if not isinstance(v, np.ndarray) or k in ("frame_id", "metadata", "calib", "image_shape"):
            continue
        batch_dict[k] = torch.from_numpy(v).float().cuda()


class SyntheticDataset:
    """The slice of DatasetTemplate the model constructors read (dataset.py:17-40): class names, range, voxel/grid size,
    point feature count.  Used by tests and bench.py, which have no KITTI files."""

    class _Encoder:
        def __init__(self, n):
            self.num_point_features = n

    def __init__(self, cfg, training=False):
        from .voxel_generator import grid_size_of
        self.dataset_cfg = cfg.DATA_CONFIG
        self.class_names = list(cfg.CLASS_NAMES)
        self.training = training
        self.point_cloud_range = np.array(self.dataset_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        self.point_feature_encoder = self._Encoder(len(self.dataset_cfg.POINT_FEATURE_ENCODING.used_feature_list))
        vc = [p for p in self.dataset_cfg.DATA_PROCESSOR if p.NAME == "transform_points_to_voxels"][0]
        self.voxel_size = vc.VOXEL_SIZE
        self.grid_size = grid_size_of(self.point_cloud_range, self.voxel_size)
