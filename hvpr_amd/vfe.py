"""Voxel feature encoders with the reference's plugin interface (pcdet/models/backbones_3d/vfe/):
same registry keys, constructor kwargs, state-dict names and batch_dict keys; the eval forward is ONE HIP launch
(hvpr_pillar_vfe_fwd_f32) instead of the reference's chain of PyTorch ops."""
import torch
import torch.nn as nn

from . import kernels
from ._lib import check, lib
from .folding import FoldCache, bn_scale_shift


class VFETemplate(nn.Module):
    """pcdet/models/backbones_3d/vfe/vfe_template.py:4-22."""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg

    def get_output_feature_dim(self):
        raise NotImplementedError

    def forward(self, **kwargs):
        raise NotImplementedError


class PFNLayer(nn.Module):
    """Parameter holder of one pillar-feature layer (pillar_vfe.py:8-27): `linear` (no bias) + `norm` (eps 1e-3, mom 0.01)."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        assert use_norm, "the HIP path is built for USE_NORM: True"
        self.last_vfe = last_layer
        width = out_channels if last_layer else out_channels // 2
        self.linear = nn.Linear(in_channels, width, bias=False)
        self.norm = nn.BatchNorm1d(width, eps=1e-3, momentum=0.01)


def _as_i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


class PillarVFE_Scale(VFETemplate):
    """Pillar encoder + scale stream — pillar_vfe.py:127-221."""

    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range):
        super().__init__(model_cfg=model_cfg)
        assert model_cfg.USE_ABSLOTE_XYZ and not model_cfg.WITH_DISTANCE, "HIP VFE: USE_ABSLOTE_XYZ=True, WITH_DISTANCE=False"
        self.num_filters = list(model_cfg.NUM_FILTERS)
        self.num_scale_features = list(model_cfg.NUM_SCALE_FEATURES)
        widths = [num_point_features + 6] + self.num_filters
        self.pfn_layers = nn.ModuleList(
            PFNLayer(widths[i], widths[i + 1], model_cfg.USE_NORM, last_layer=(i >= len(widths) - 2))
            for i in range(len(widths) - 1))
        sw = [5] + self.num_scale_features
        self.pfn_scale_layers = nn.ModuleList(
            nn.Sequential(nn.Linear(sw[i], sw[i + 1], bias=False), nn.BatchNorm1d(sw[i + 1], eps=1e-3, momentum=0.01), nn.ReLU())
            for i in range(len(sw) - 1))
        self.voxel_size = [float(v) for v in voxel_size]
        self.offsets = [self.voxel_size[i] / 2 + float(point_cloud_range[i]) for i in range(3)]
        self._fold = FoldCache()

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def train(self, mode=True):
        self._fold.invalidate()
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self._fold.invalidate()
        return super()._load_from_state_dict(*a, **k)

    def _build_folded(self):
        f = {}
        for key, lin, bn in (("0", self.pfn_layers[0].linear, self.pfn_layers[0].norm),
                             ("1", self.pfn_layers[1].linear, self.pfn_layers[1].norm),
                             ("s0", self.pfn_scale_layers[0][0], self.pfn_scale_layers[0][1]),
                             ("s1", self.pfn_scale_layers[1][0], self.pfn_scale_layers[1][1])):
            s, t = bn_scale_shift(bn)
            f["w" + key] = (lin.weight.detach().float() * s[:, None]).contiguous()
            f["b" + key] = t.contiguous()
        return f

    def forward(self, batch_dict, **kwargs):
        voxels, num, coords = batch_dict["voxels"], batch_dict["voxel_num_points"], batch_dict["voxel_coords"]
        if self.training:
            return self._forward_train(batch_dict, voxels, num, coords)
        folded = self._fold.get(voxels.device, self._build_folded)
        pf, sf, mask = kernels.pillar_vfe_fwd(voxels.contiguous(), _as_i32(num), _as_i32(coords).contiguous(), folded,
                                              self.voxel_size, self.offsets, m_device=batch_dict.get("voxel_count_device"))
        batch_dict["pillar_features"] = pf
        batch_dict["pillar_scale_features"] = sf
        batch_dict["pillar_mask"] = mask
        return batch_dict


_ws_cache = {}


def _train_ws(device):
    if device not in _ws_cache:
        _ws_cache[device] = torch.empty((lib().hvpr_pillar_vfe_train_workspace_bytes(),), dtype=torch.uint8, device=device)
    return _ws_cache[device]


class _PfnTrain(torch.autograd.Function):
    """The two PFN layers with batch-statistics BatchNorm on hvpr_pillar_vfe_train_fwd_f32 / hvpr_pillar_vfe_bwd_f32
    (pillar_vfe.py:184-221; nothing of size (M, 32, C) is kept or materialised)."""

    @staticmethod
    def forward(ctx, voxels, num, coords, w0, g0, b0, w1, g1, b1, eps, vs, off):
        M, P, _ = voxels.shape
        dev = voxels.device
        args = [t.detach().contiguous() for t in (w0, g0, b0, w1, g1, b1)]
        out = torch.empty((M, 64), dtype=torch.float32, device=dev)
        m0, v0 = torch.empty(16, device=dev), torch.empty(16, device=dev)
        m1, v1 = torch.empty(64, device=dev), torch.empty(64, device=dev)
        ws = _train_ws(dev)
        check(lib().hvpr_pillar_vfe_train_fwd_f32(kernels._ptr(voxels, torch.float32, "voxels"), kernels._ptr(num, torch.int32, "voxel_num_points"),
                                                  kernels._ptr(coords, torch.int32, "voxel_coords"), M, P,
                                                  *[kernels._ptr(a, torch.float32, "PFN parameter") for a in args], float(eps),
                                                  vs[0], vs[1], vs[2], off[0], off[1], off[2], out.data_ptr(), m0.data_ptr(), v0.data_ptr(),
                                                  m1.data_ptr(), v1.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()),
              "hvpr_pillar_vfe_train_fwd_f32")
        ctx.save_for_backward(voxels, num, coords, *args)
        ctx.geom = (float(eps), tuple(vs), tuple(off))
        ctx.mark_non_differentiable(m0, v0, m1, v1)
        ctx.set_materialize_grads(False)     # (their gradients would arrive as zero tensors: one allocation + fill each, per call)
        return out, m0, v0, m1, v1

    @staticmethod
    def backward(ctx, d_out, *_):
        voxels, num, coords, w0, g0, b0, w1, g1, b1 = ctx.saved_tensors
        if d_out is None:
            d_out = torch.zeros((voxels.shape[0], w1.shape[0]), dtype=torch.float32, device=voxels.device)
        eps, vs, off = ctx.geom
        M, P, _ = voxels.shape
        grads = [torch.empty_like(t) for t in (w0, g0, b0, w1, g1, b1)]
        ws = _train_ws(voxels.device)
        check(lib().hvpr_pillar_vfe_bwd_f32(voxels.data_ptr(), num.data_ptr(), coords.data_ptr(), M, P, w0.data_ptr(), g0.data_ptr(),
                                            b0.data_ptr(), w1.data_ptr(), g1.data_ptr(), b1.data_ptr(), eps, vs[0], vs[1], vs[2],
                                            off[0], off[1], off[2], kernels._ptr(d_out.contiguous(), torch.float32, "d pillar_features"),
                                            *[g.data_ptr() for g in grads], ws.data_ptr(), ws.numel(), kernels._stream()),
              "hvpr_pillar_vfe_bwd_f32")
        return (None, None, None, *grads, None, None, None)


def _update_running(bn, mean, var, n):
    """nn.BatchNorm1d's running statistics from the batch mean / biased variance (momentum, unbiased variance)."""
    if not bn.track_running_stats:
        return
    with torch.no_grad():
        bn.num_batches_tracked += 1
        mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
        if torch.is_tensor(n):              # SyncBatchNorm: the global count, a device scalar
            bn.running_var.mul_(1 - mom).add_(var * (n / (n - 1).clamp_min(1.0)).to(var.dtype) * mom)
        else:
            bn.running_var.mul_(1 - mom).add_(var, alpha=mom * n / max(n - 1, 1))


def _train_forward(self, batch_dict, voxels, num, coords):
    """Training forward with batch-statistics BatchNorm (pillar_vfe.py:184-221; BN over all M*P slots, padded slots included —
    SURVEY.md B.5): the two PFN layers, forward and backward, on hvpr_pillar_vfe_train_fwd_f32 / hvpr_pillar_vfe_bwd_f32.
    Unsupported widths / CPU tensors raise (torch form: tests/torch_forms.py)."""
    md = batch_dict.get("voxel_count_device")
    if md is not None:                      # rows past the live count are unspecified: drop them (one host read per step)
        m = int(md.item())
        voxels, num, coords = voxels[:m], num[:m], coords[:m]
        batch_dict["voxels"], batch_dict["voxel_num_points"], batch_dict["voxel_coords"] = voxels, num, coords
        batch_dict["voxel_count_device"] = None
    n = num.to(voxels.dtype)
    M, P, _ = voxels.shape
    xyz = voxels[:, :, :3]
    mean = xyz.sum(dim=1, keepdim=True) / n.view(-1, 1, 1)
    mask = (torch.arange(P, device=voxels.device).view(1, -1) < num.view(-1, 1)).unsqueeze(-1).to(voxels.dtype)
    if not (voxels.is_cuda and voxels.dtype == torch.float32):
        raise RuntimeError("hvpr_amd: PillarVFE_Scale's training forward needs fp32 GPU tensors (the HIP path has no CPU fallback)")
    from . import conv_train as ct
    if ct.any_rank_true(M == 0, voxels.device):    # under SyncBatchNorm all ranks raise together: a lone raiser would leave the
        #                                            others waiting in the statistics' all-reduce
        raise ValueError("hvpr_amd: PillarVFE_Scale's training forward got a batch without a single pillar (train-mode BatchNorm has "
                         "no statistics to take; the reference divides by zero there)" if M == 0 else
                         "hvpr_amd: another rank's batch has no pillar — SyncBatchNorm's all-reduce cannot be joined by all ranks")
    if len(self.pfn_layers) != 2:
        raise ValueError("hvpr_amd: the VFE training kernels are built for two PFN layers (hvpr.yaml NUM_FILTERS: [32, 64])")
    l0, l1 = self.pfn_layers[0], self.pfn_layers[1]
    if tuple(l0.linear.weight.shape) != (16, 10) or tuple(l1.linear.weight.shape) != (64, 32) or l0.norm.eps != l1.norm.eps or P > 32:
        raise ValueError("hvpr_amd: the VFE training kernels are built for 4 point features, NUM_FILTERS [32, 64], one eps and <= 32 points per pillar")
    # both PFN layers, forward and backward, on the library's kernels
    x, m0, v0, m1, v1 = _PfnTrain.apply(voxels.contiguous(), _as_i32(num).contiguous(), _as_i32(coords).contiguous(),
                                        l0.linear.weight, l0.norm.weight, l0.norm.bias, l1.linear.weight, l1.norm.weight,
                                        l1.norm.bias, l0.norm.eps, self.voxel_size, self.offsets)
    cnt = ct.global_count(M * P, voxels.device)
    _update_running(l0.norm, m0, v0, cnt)
    _update_running(l1.norm, m1, v1, cnt)
    # scale stream (pillar_vfe.py:213-216): [n, |mean|, mean] (5 columns, zero-padded to the convolution kernel's 8) through
    # Linear (no bias) + train-mode BatchNorm1d + ReLU twice, as 1x1 convolutions over the M rows on the library's kernels
    s = torch.cat([n.unsqueeze(1), torch.norm(mean, 2, 2), mean.squeeze(1), mean.new_zeros((M, 3))], dim=-1)
    t = s.view(1, 1, M, 8)
    for seq in self.pfn_scale_layers:
        lin, bn = seq[0], seq[1]
        w = lin.weight
        if lin.bias is not None or w.shape[0] % 8 != 0 or w.shape[1] > t.shape[-1]:
            raise ValueError("hvpr_amd: the VFE scale stream kernels take bias-free layers with widths that are multiples of 8")
        pad = t.shape[-1] - w.shape[1]
        w = w if pad == 0 else torch.cat([w, w.new_zeros((w.shape[0], pad))], dim=1)
        t = ct.bn_relu(ct.conv(t, w.view(w.shape[0], w.shape[1], 1, 1)), bn)
    s = t.view(M, -1)
    batch_dict["pillar_features"] = x.reshape(M, -1)
    batch_dict["pillar_scale_features"] = s
    batch_dict["pillar_mask"] = mask
    return batch_dict


PillarVFE_Scale._forward_train = _train_forward


__all__ = {
    "VFETemplate": VFETemplate,
    "PillarVFE_Scale": PillarVFE_Scale,
}
