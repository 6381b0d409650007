"""Pillar -> BEV modules with the reference's plugin interface (pcdet/models/backbones_2d/map_to_bev/).

Eval forward = memory read-out (hvpr_memory_readout_fwd_f32) + gather-form scatter (hvpr_scatter_bev_fwd_f32) for the
whole batch: no python loop over frames, no `.item()` host sync (the reference has both: pointpillar_scatter.py:176-178).
The canvases come back as (B, C, ny, nx) tensors in channels_last memory format."""
import math

import torch
import torch.nn as nn

from . import kernels


class MemoryUnit_Agg(nn.Module):
    """Memory bank — map_to_bev/memory_module.py:11-27 (weight (mem_dim, fea_dim), U(+-1/sqrt(fea_dim)))."""

    def __init__(self, mem_dim, fea_dim, shrink_thres=0.0025):
        super().__init__()
        self.mem_dim, self.fea_dim, self.shrink_thres = mem_dim, fea_dim, shrink_thres
        self.weight = nn.Parameter(torch.empty(mem_dim, fea_dim))
        stdv = 1.0 / math.sqrt(fea_dim)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, input1, k, input2=None):
        """Eval branch (memory_module.py:60-77): returns {'output': (nv, C)}; 'att' is never consumed in eval."""
        if self.training:
            raise NotImplementedError("hvpr_amd: the training branch of MemoryUnit_Agg is not built yet")
        return {"output": kernels.memory_readout_fwd(input1.contiguous(), self.weight.detach().contiguous(), k)}

    def extra_repr(self):
        return f"mem_dim={self.mem_dim}, fea_dim={self.fea_dim}"


class _ScatterBase(nn.Module):
    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg.NUM_BEV_FEATURES
        self.nx, self.ny, self.nz = [int(g) for g in grid_size]
        assert self.nz == 1
        self._ws = None

    def _workspace(self, batch, device):
        if self._ws is None or self._ws[0] != (batch, device):
            self._ws = ((batch, device), kernels.scatter_workspace(batch, self.nx, self.ny, device))
        return self._ws[1]


def _coords_i32(batch_dict):
    c = batch_dict["voxel_coords"]
    return (c if c.dtype == torch.int32 else c.to(torch.int32)).contiguous()


def _batch_size(batch_dict):
    # the reference derives it with a host sync, coords[:, 0].max().item() + 1 (pointpillar_scatter.py:176); every
    # caller of the path already knows it (dataset.py:178), so it is read from the dict when present.
    if "batch_size" in batch_dict:
        return int(batch_dict["batch_size"])
    return int(batch_dict["voxel_coords"][:, 0].max().item()) + 1


class PointPillarScatter(_ScatterBase):
    """Plain scatter — pointpillar_scatter.py:5-37."""

    def forward(self, batch_dict, **kwargs):
        pf = batch_dict["pillar_features"]
        B = _batch_size(batch_dict)
        sp, _ = kernels.scatter_bev_fwd(pf.contiguous(), None, None, _coords_i32(batch_dict), B, self.nx, self.ny,
                                        self._workspace(B, pf.device), m_device=batch_dict.get("voxel_count_device"))
        batch_dict["spatial_features"] = sp
        return batch_dict


class PointPillarScatter_Agg_Memory_1_scale(_ScatterBase):
    """Scatter with memory read-out and scale canvas — pointpillar_scatter.py:39-222 (eval branch :169-222)."""

    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__(model_cfg, grid_size)
        self.num_coord_points = model_cfg.NUM_COORD_POINTS
        self.num_pt_features = model_cfg.NUM_PT_FEATURES
        self.num_scale_features = model_cfg.NUM_SCALE_FEATURES
        self.k = model_cfg.NUM_K
        self.mem_size = model_cfg.NUM_M
        self.shrink_thres = model_cfg.SHRINK_TH
        self.memory = MemoryUnit_Agg(self.mem_size, self.num_pt_features, self.shrink_thres)

    def forward(self, batch_dict, **kwargs):
        if self.training:
            raise NotImplementedError("hvpr_amd: the training branch of the scatter module is not built yet")
        pf, sf = batch_dict["pillar_features"], batch_dict["pillar_scale_features"]
        md = batch_dict.get("voxel_count_device")
        B = _batch_size(batch_dict)
        mem = kernels.memory_readout_fwd(pf.contiguous(), self.memory.weight.detach().contiguous(), self.k, m_device=md)
        sp, sc = kernels.scatter_bev_fwd(pf.contiguous(), mem, sf.contiguous(), _coords_i32(batch_dict), B, self.nx,
                                         self.ny, self._workspace(B, pf.device), m_device=md)
        batch_dict["spatial_features"] = sp           # (B, 128, ny, nx): ch 0-63 pillar (detached), 64-127 memory
        batch_dict["spatial_scale_features"] = sc     # (B, 32, ny, nx)
        return batch_dict


__all__ = {
    "PointPillarScatter": PointPillarScatter,
    "PointPillarScatter_Agg_Memory_1_scale": PointPillarScatter_Agg_Memory_1_scale,
}
