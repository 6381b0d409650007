"""Anchor target assignment on the device through the library's own kernels.

AxisAlignedTargetAssigner (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-213) with
boxes3d_nearest_bev_iou (pcdet/utils/box_utils.py:252-323) and ResidualCoder.encode_torch (pcdet/utils/box_coder_utils.py:13-43):
hvpr_assign_targets_f32, two launches per anchor set for the whole batch, results written in the head's anchor order.  The
reference trims padded ground-truth rows with a python loop over `.sum()` and takes its arg-maxes through `.cpu().numpy()`
(:53-57,:148,:153): two host syncs per frame; here there is none.  POS_FRACTION < 0 (no sampling) and NORM_BY_NUM_EXAMPLES
false, as hvpr.yaml:114-124 sets.  CPU tensors raise (torch form: tests/torch_forms.py)."""
import torch

from . import kernels


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
        self._flat = None

    def _anchor_sets(self, all_anchors, device):
        """Every anchor set as a contiguous (n, 7) device tensor + its per-location count, kept between calls."""
        key = (tuple(a.data_ptr() for a in all_anchors), device)
        if self._flat is None or self._flat[0] != key:
            sets = []
            for a in all_anchors:
                assert a.shape[-1] == 7 and a.shape[0] == 1, "hvpr path: one anchor height, 7 box parameters"
                per_loc = int(a.shape[3] * a.shape[4])
                sets.append((a.reshape(-1, 7).to(device=device, dtype=torch.float32).contiguous(), per_loc))
            self._flat = (key, sets)
        return self._flat[1]

    def assign_targets(self, all_anchors, gt_boxes_with_classes):
        """all_anchors: list of (nz,ny,nx,S,R,7); gt (B,G,8).  Returns box_cls_labels (B,A) i32, box_reg_targets (B,A,7),
        reg_weights (B,A) and positives_per_frame (B,) i32, anchors ordered (z,y,x,class,size,rot) as the single head predicts them."""
        gt = gt_boxes_with_classes
        if not gt.is_cuda or gt.dtype != torch.float32:
            raise RuntimeError("hvpr_amd: the target assigner needs fp32 GPU tensors (the HIP path has no CPU fallback)")
        if self.norm_by_num_examples:
            raise ValueError("hvpr_amd: NORM_BY_NUM_EXAMPLES is not built (hvpr.yaml:123 sets it False)")
        gt = gt.contiguous()
        B, G = int(gt.shape[0]), int(gt.shape[1])
        sets = self._anchor_sets(all_anchors, gt.device)
        stride = sum(p for _, p in sets)
        n_loc = sets[0][0].shape[0] // sets[0][1]
        assert all(a.shape[0] // p == n_loc for a, p in sets), "every anchor set covers the same feature map"
        A = n_loc * stride
        labels = torch.empty((B, A), dtype=torch.int32, device=gt.device)
        targets = torch.empty((B, A, self.box_coder.code_size), dtype=torch.float32, device=gt.device)
        weights = torch.empty((B, A), dtype=torch.float32, device=gt.device)
        pos = torch.zeros((B,), dtype=torch.int32, device=gt.device)
        L = kernels.lib()
        ws = torch.empty(max(int(L.hvpr_assign_targets_workspace_bytes(B, G)), 256), dtype=torch.uint8, device=gt.device)
        off = 0
        for cname, (a, per_loc) in zip(self.anchor_class_names, sets):
            kernels.check(L.hvpr_assign_targets_f32(a.data_ptr(), a.shape[0], gt.data_ptr(), B, G, self.class_names.index(cname),
                                                    len(self.class_names), float(self.matched[cname]), float(self.unmatched[cname]),
                                                    per_loc, stride, off, A, labels.data_ptr(), targets.data_ptr(), weights.data_ptr(),
                                                    pos.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()), "hvpr_assign_targets_f32")
            off += per_loc
        return {"box_cls_labels": labels, "box_reg_targets": targets, "reg_weights": weights, "positives_per_frame": pos}
