"""Training losses of the anchor head on the library's own kernels.

hvpr_rpn_losses_f32: the three losses of one prediction stream — SigmoidFocalClassificationLoss (pcdet/utils/loss_utils.py:9-72),
WeightedSmoothL1Loss on the sin-difference-encoded residuals (:75-136), WeightedCrossEntropyLoss on the direction bin (:181-206),
assembled as pcdet/models/dense_heads/anchor_head_template.py:101-260 does — AND their gradients w.r.t. the predictions in ONE
launch + a fixed-order sum (the reference: ~40 elementwise torch kernels per stream and as many again in autograd's backward).
hvpr_mse_loss_f32: get_mem_loss (:262-275).  CPU tensors raise (torch forms: tests/torch_forms.py)."""
import ctypes

import torch

from . import kernels


class _RpnLosses(torch.autograd.Function):
    """(cls, box, dir predictions) -> the stream's (cls, loc, dir) losses as one (3,) tensor; backward scales the gradients the
    forward launch already produced by the upstream gradient of each scalar."""

    @staticmethod
    def forward(ctx, cls_preds, box_preds, dir_preds, labels, reg_targets, anchor_rot, pos_count, num_class, num_dir_bins, weights,
                dir_offset, alpha, gamma, beta):
        if not cls_preds.is_cuda or cls_preds.dtype != torch.float32:
            raise RuntimeError("hvpr_amd: the head's losses need fp32 GPU tensors (the HIP path has no CPU fallback)")
        B, A = labels.shape
        cls_c, box_c = cls_preds.contiguous(), box_preds.contiguous()
        dir_c = None if dir_preds is None else dir_preds.contiguous()
        assert cls_c.numel() == B * A * num_class and box_c.numel() == B * A * 7
        L = kernels.lib()
        out = torch.empty((3,), dtype=torch.float32, device=cls_c.device)
        g_cls, g_box = torch.empty_like(cls_c), torch.empty_like(box_c)
        g_dir = None if dir_c is None else torch.empty_like(dir_c)
        ws = torch.empty(int(L.hvpr_rpn_losses_workspace_bytes(B, A)), dtype=torch.uint8, device=cls_c.device)
        cw = (ctypes.c_float * 7)(*[float(v) for v in weights["code_weights"]])
        kernels.check(L.hvpr_rpn_losses_f32(
            cls_c.data_ptr(), box_c.data_ptr(), None if dir_c is None else dir_c.data_ptr(), kernels._ptr(labels, torch.int32, "labels"),
            kernels._ptr(reg_targets, torch.float32, "reg_targets"), None if dir_c is None else kernels._ptr(anchor_rot, torch.float32, "anchor_rot"),
            kernels._ptr(pos_count, torch.int32, "positives_per_frame"), B, A, int(num_class), int(num_dir_bins if dir_c is not None else 0),
            float(alpha), float(gamma), float(beta), ctypes.cast(cw, ctypes.c_void_p), float(weights["cls_weight"]), float(weights["loc_weight"]),
            float(weights["dir_weight"]), float(dir_offset), out.data_ptr(), g_cls.data_ptr(), g_box.data_ptr(),
            None if g_dir is None else g_dir.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()), "hvpr_rpn_losses_f32")
        ctx.save_for_backward(g_cls, g_box, g_dir)
        ctx.shapes = (cls_preds.shape, box_preds.shape, None if dir_preds is None else dir_preds.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        g_cls, g_box, g_dir = ctx.saved_tensors
        s = ctx.shapes
        return ((g_cls * g[0]).view(s[0]), (g_box * g[1]).view(s[1]), None if g_dir is None else (g_dir * g[2]).view(s[2]),
                None, None, None, None, None, None, None, None, None, None, None)


def rpn_losses(cls_preds, box_preds, dir_preds, labels, reg_targets, anchor_rot, pos_count, num_class, cfg_weights, dir_offset,
               num_dir_bins, alpha=0.25, gamma=2.0, beta=1.0 / 9.0):
    """Losses of ONE prediction stream.  cls/box/dir preds are NHWC head outputs; labels (B,A) i32, reg_targets (B,A,7),
    anchor_rot (A,) the anchors' headings, pos_count (B,) i32 positives per frame (the target assigner's).
    Returns (cls_loss, box_loss (loc + dir), parts dict)."""
    out = _RpnLosses.apply(cls_preds, box_preds, dir_preds, labels, reg_targets, anchor_rot, pos_count, num_class, num_dir_bins,
                           cfg_weights, dir_offset, alpha, gamma, beta)
    parts = {"cls": out[0], "loc": out[1]}
    box_loss = out[1]
    if dir_preds is not None:
        parts["dir"] = out[2]
        box_loss = box_loss + out[2]
    return out[0], box_loss, parts


class _MseLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, weight):
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError("hvpr_amd: the memory loss needs fp32 GPU tensors (the HIP path has no CPU fallback)")
        xc, tc = x.contiguous(), target.detach().contiguous()
        assert xc.dim() == 2 and xc.shape == tc.shape
        L = kernels.lib()
        out = torch.empty((), dtype=torch.float32, device=xc.device)
        gx = torch.empty_like(xc)
        ws = torch.empty(int(L.hvpr_mse_loss_workspace_bytes()), dtype=torch.uint8, device=xc.device)
        kernels.check(L.hvpr_mse_loss_f32(xc.data_ptr(), tc.data_ptr(), xc.shape[0], xc.shape[1], float(weight), out.data_ptr(), gx.data_ptr(),
                                          ws.data_ptr(), ws.numel(), kernels._stream()), "hvpr_mse_loss_f32")
        ctx.save_for_backward(gx)
        return out

    @staticmethod
    def backward(ctx, g):
        (gx,) = ctx.saved_tensors
        return gx * g, None, None


def memory_loss(memory_pos, point_pos, mem_weight):
    """MSE(memory features, detached point features) / #pillars (anchor_head_template.py:262-275; the divisor is the
    number of pillars of the batch, kept as the reference wrote it)."""
    if memory_pos.shape[0] == 0:
        return memory_pos.sum() * 0.0
    return _MseLoss.apply(memory_pos, point_pos, mem_weight)
