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
        self._packed = None

    def train(self, mode=True):
        self._packed = None
        return super().train(mode)

    def _load_from_state_dict(self, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(*a, **k)

    def packed_bank(self):
        """The bank in the read-out kernel's streaming layout (kernels.PackedBank), re-packed when the weight changed:
        tracked in-place updates bump the tensor version, .to() changes the storage, train() / load_state_dict drop the
        copy.  (Writing through `weight.data` while in eval mode is invisible to all of these, as for the folded BatchNorms:
        call train(False) again afterwards.)"""
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, kernels.PackedBank(w) if (w.is_cuda and w.shape[1] == 64) else w.detach().contiguous())
        return self._packed[1]

    def forward(self, input1, k, input2=None):
        """Eval branch (memory_module.py:60-77): returns {'output': (nv, C)}; 'att' is never consumed in eval.
        Training branch (:31-59): input2 = the k positive point features of every pillar, (nv, k, C)."""
        if self.training:
            return self._forward_train(input1, k, input2)
        return {"output": kernels.memory_readout_fwd(input1.contiguous(), self.packed_bank(), k)}

    def _check_train(self, t, d):
        if not t.is_cuda:
            raise RuntimeError("hvpr_amd: the memory training branch needs GPU tensors (the HIP path has no CPU fallback)")
        if d != 64 or self.mem_dim > 2048 or not self.shrink_thres > 0:
            raise ValueError("hvpr_amd: the memory training branch is built for 64 channels, <= 2048 items and SHRINK_TH > 0 (hvpr.yaml:83-85)")

    def _forward_train(self, pillars, k, positives):
        """Hard-shrink addressing through hvpr_memory_train_fwd/bwd_f32: the (nv*k, items) attention is never materialised ('att'
        is not returned: nothing on the path consumes it, pointpillar_scatter.py:133-138).  Torch form: tests/torch_forms.py."""
        nv, _, d = positives.shape
        self._check_train(positives, d)
        mem = _MemoryTrain.apply(positives.reshape(-1, d), self.weight, float(self.shrink_thres)).reshape(nv, k, d)
        agg = torch.softmax((mem * pillars.unsqueeze(1)).sum(dim=2), dim=1)
        return {"output": (agg.detach().unsqueeze(2) * mem).sum(dim=1)}

    def forward_train_indexed(self, pillars, k, points, idx, plan=None):
        """The training branch when the k positives of every pillar are ROWS OF ONE POINT TENSOR, positives = points[idx] (what
        the scatter module's training branch has, pointpillar_scatter.py:75-76,133): the addressing of memory_module.py:36-50 is a
        function of the row alone — softmax(x W^T) -> hard shrink -> L1 normalise -> . W — so it is evaluated ONCE PER POINT and
        gathered, instead of once per (pillar, k) pair: a point is picked by 4.8 pillars on average at hvpr.yaml's sizes (16 384
        points against ~3 900 pillars x 20 per frame), some by hundreds.  Same values per row (a row's result does not depend on its
        neighbours in the launch); the gradients reach the points as J^T (sum of the picks' dy) instead of sum of J^T dy —
        round-off apart the same thing.  idx (nv, k) int64 rows of `points`; plan: the shared _EdgePlan of idx (optional)."""
        nv, d = idx.shape[0], points.shape[1]
        self._check_train(points, d)
        mem_points = _MemoryTrain.apply(points, self.weight, float(self.shrink_thres))      # (N, d)
        mem = _GatherRows.apply(mem_points, idx, plan)                                       # (nv, k, d)
        agg = torch.softmax((mem * pillars.unsqueeze(1)).sum(dim=2), dim=1)
        return {"output": (agg.detach().unsqueeze(2) * mem).sum(dim=1)}

    def extra_repr(self):
        return f"mem_dim={self.mem_dim}, fea_dim={self.fea_dim}"


class _MemoryTrain(torch.autograd.Function):
    """y = normalize_1(hard_shrink(softmax(x W^T))) W  (memory_module.py:36-48) on hvpr_memory_train_fwd_f32 / _bwd_f32."""

    @staticmethod
    def _ws(n_items, device):
        return torch.empty(kernels.lib().hvpr_memory_train_workspace_bytes(n_items), dtype=torch.uint8, device=device)

    @staticmethod
    def forward(ctx, x, weight, lambd):
        x, w = x.contiguous(), weight.detach().contiguous()
        R, n_items = x.shape[0], w.shape[0]
        y = torch.empty_like(x)
        stats = torch.empty((R, 4), dtype=torch.float32, device=x.device)
        ws = _MemoryTrain._ws(n_items, x.device)
        kernels.check(kernels.lib().hvpr_memory_train_fwd_f32(kernels._ptr(x, torch.float32, "x"), R, kernels._ptr(w, torch.float32, "memory.weight"),
                                                              n_items, lambd, y.data_ptr(), stats.data_ptr(), ws.data_ptr(), ws.numel(),
                                                              kernels._stream()), "hvpr_memory_train_fwd_f32")
        ctx.save_for_backward(x, w, stats)
        ctx.lambd = lambd
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, stats = ctx.saved_tensors
        dy = dy.contiguous()
        R, n_items = x.shape[0], w.shape[0]
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        scratch = torch.empty((2 * max(R, 1),), dtype=torch.float32, device=x.device)
        ws = _MemoryTrain._ws(n_items, x.device)
        kernels.check(kernels.lib().hvpr_memory_train_bwd_f32(x.data_ptr(), kernels._ptr(dy, torch.float32, "dy"), R, w.data_ptr(), n_items, ctx.lambd,
                                                              stats.data_ptr(), dx.data_ptr(), dw.data_ptr(), scratch.data_ptr(), ws.data_ptr(),
                                                              ws.numel(), kernels._stream()), "hvpr_memory_train_bwd_f32")
        return dx, dw, None


class _GatherRows(torch.autograd.Function):
    """rows[idx] for a (N, C) feature matrix and an integer index tensor of any shape — `points[idx]` of get_score
    (pointpillar_scatter.py:76): hvpr_gather_rows_f32 forward; backward = hvpr_segment_sum_rows_f32 over the picks sorted by point
    (one stable argsort of the frame's picks; torch's own index backward took 26 ms of a training step, and float atomics,
    hvpr_scatter_add_rows_f32, are not reproducible run to run)."""

    @staticmethod
    def forward(ctx, rows, idx, plan=None):
        rows = rows.contiguous()
        flat = idx.reshape(-1).to(torch.int32).contiguous()
        out = torch.empty((flat.numel(), rows.shape[1]), dtype=torch.float32, device=rows.device)
        kernels.check(kernels.lib().hvpr_gather_rows_f32(kernels._ptr(rows, torch.float32, "rows"), rows.shape[0], rows.shape[1], flat.data_ptr(),
                                                         flat.numel(), out.data_ptr(), kernels._stream()), "hvpr_gather_rows_f32")
        ctx.save_for_backward(flat)
        ctx.n = rows.shape[0]
        ctx.plan = plan
        return out.view(*idx.shape, rows.shape[1])

    @staticmethod
    def backward(ctx, grad):
        (flat,) = ctx.saved_tensors
        grad = grad.contiguous()
        c = grad.shape[-1]
        # a point may be picked by many pillars: its gradient is summed pick by pick in ascending order (no float atomics)
        order, chunk_ptr, dest_ptr = ctx.plan.get() if ctx.plan is not None else kernels.edges_by_destination(flat.to(torch.int64), ctx.n)
        g = kernels.segment_sum_rows(grad.reshape(-1, c), 0, c, order, None, chunk_ptr, dest_ptr, ctx.n)
        return g, None, None


class _EdgePlan:
    """The picks idx -> rows of an (n, C) tensor grouped by destination (kernels.edges_by_destination: one stable argsort + a
    handful of scans), built on first use and shared by every gather of the SAME index tensor — the training branch gathers the point
    features and the per-point memory read-out with one idx."""

    def __init__(self, idx, n):
        self.idx, self.n, self._plan = idx, n, None

    def get(self):
        if self._plan is None:
            self._plan = kernels.edges_by_destination(self.idx.reshape(-1).to(torch.int64), self.n)
        return self._plan


class _ScatterCanvas(torch.autograd.Function):
    """Scatter (M,C) rows to a dense NHWC canvas with the HIP gather-form kernel; backward gathers the rows back."""

    @staticmethod
    def forward(ctx, feats, coords_i32, batch, nx, ny, ws):
        c = feats.shape[1]
        if c == 128:
            sp, _ = kernels.scatter_bev_fwd(feats[:, :64].contiguous(), feats[:, 64:].contiguous(), None, coords_i32, batch, nx, ny, ws)
        elif c == 64:
            sp, _ = kernels.scatter_bev_fwd(feats.contiguous(), None, None, coords_i32, batch, nx, ny, ws)
        else:           # narrower canvas (the 32-channel scale stream): pad to the 64-channel kernel and slice
            assert c < 64
            pad = torch.cat([feats, feats.new_zeros(feats.shape[0], 64 - c)], dim=1)
            sp = kernels.scatter_bev_fwd(pad, None, None, coords_i32, batch, nx, ny, ws)[0][:, :c]
        rows = ((coords_i32[:, 0] * ny + coords_i32[:, 2]) * nx + coords_i32[:, 3]).contiguous()      # int32 cell of every pillar
        ctx.save_for_backward(rows)
        ctx.c = c
        return sp

    @staticmethod
    def backward(ctx, grad):
        """Every pillar reads its cell back: hvpr_gather_rows_f32 over the NHWC gradient canvas (one row = one BEV cell)."""
        (rows,) = ctx.saved_tensors
        g = grad.permute(0, 2, 3, 1).contiguous()                     # NHWC (a view when grad is channels_last)
        cells, ch = g.shape[0] * g.shape[1] * g.shape[2], g.shape[3]
        out = torch.empty((rows.shape[0], ch), dtype=torch.float32, device=g.device)
        kernels.check(kernels.lib().hvpr_gather_rows_f32(kernels._ptr(g, torch.float32, "grad canvas"), cells, ch, rows.data_ptr(), rows.shape[0],
                                                         out.data_ptr(), kernels._stream()), "hvpr_gather_rows_f32")
        return (out if ch == ctx.c else out[:, :ctx.c].contiguous()), None, None, None, None, None


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


def _frame_ranges(cols, B):
    """For every batch-index column of `cols`: [(lo, hi)] * B if its rows are grouped by frame in ascending order (frame b = rows
    lo .. hi), else None.  One device -> host read for all columns."""
    parts = []
    for c in cols:
        c = c.long()
        ok = ((c[1:] >= c[:-1]).sum() == max(c.numel() - 1, 0)) & ((c >= 0) & (c < B)).sum().eq(c.numel())
        cnt = (c.view(-1, 1) == torch.arange(B, device=c.device).view(1, -1)).sum(dim=0)      # (torch.bincount reads its size on the host)
        parts += [cnt, ok.view(1).long()]
    host = torch.cat(parts).tolist()
    out = []
    for i in range(len(cols)):
        cnt, ok = host[i * (B + 1):i * (B + 1) + B], host[i * (B + 1) + B]
        if not ok:
            out.append(None)
            continue
        lo, r = 0, []
        for n in cnt:
            r.append((lo, lo + n))
            lo += n
        out.append(r)
    return out


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

    def get_score(self, points, pillars):
        """Point <-> pillar attention of one sample (pointpillar_scatter.py:67-83): points (N,C), pillars (M,C).
        Returns the aggregate (M,C) AND the k positive point features (M,k,C) — the training branch of the memory needs the
        latter."""
        # The reference takes the top-k over points of softmax(points @ pillars^T, dim=0) and uses ONLY the indices
        # (:70-73).  softmax is increasing within a column, so the indices are those of the raw logits: the (N, M)
        # softmax (60 M elements per sample, 26 % of a training step when materialised) is skipped, and the logits are
        # formed transposed so that the top-k runs along the contiguous dimension.
        with torch.no_grad():
            idx = self._topk_points(pillars.detach(), points.detach())            # (M, k), descending
        positives = _GatherRows.apply(points, idx)                                          # (M, k, C)
        w = torch.softmax((pillars.unsqueeze(1) * positives).sum(dim=2), dim=1)            # (M, k) logits of the k positives
        return (w.detach().unsqueeze(2) * positives).sum(dim=1), positives

    def _topk_points(self, pillars, points):
        """Indices (M, k) of the k points with the largest pillar . point logits, descending — hvpr_point_pillar_topk_f32: the
        (M, N) logits are never materialised (half-precision matrix-core pre-filter over blocks of 2048 points with a rising
        bound, exact fp32 re-check of the surviving candidates)."""
        k, N = self.k, points.shape[0]
        nb = 1
        if not pillars.is_cuda:
            raise RuntimeError("hvpr_amd: get_score needs GPU tensors (the HIP path has no CPU fallback)")
        if pillars.shape[1] != 64:
            raise ValueError("hvpr_amd: get_score is built for 64-channel point / pillar features (hvpr.yaml:81)")
        if pillars.shape[0] == 0:
            return torch.zeros((0, k), dtype=torch.long, device=pillars.device)
        if N < k * max(nb, 1):
            raise ValueError(f"hvpr_amd: get_score needs at least k = {k} points per sample, got {N}")
        pf, pts = pillars.contiguous(), points.contiguous()
        packed = kernels.PackedBank(pts)                               # fp16 operand tiles + channel maxima of this sample's points
        idx = torch.empty((pf.shape[0], k), dtype=torch.int32, device=pf.device)
        kernels.check(kernels.lib().hvpr_point_pillar_topk_f32(kernels._ptr(pf, torch.float32, "pillars"), pf.shape[0], packed.rows.data_ptr(),
                                                               packed.data.data_ptr(), N, k, idx.data_ptr(), kernels._stream()),
                      "hvpr_point_pillar_topk_f32")
        return idx.long()

    def _forward_train(self, batch_dict):
        """Training branch, pointpillar_scatter.py:87-167: three canvases (memory-fed, point-fed, scale)."""
        pf, sf = batch_dict["pillar_features"], batch_dict["pillar_scale_features"]
        coords = _coords_i32(batch_dict)
        point_f, point_c = batch_dict["point_features"], batch_dict["point_coords"]
        B = _batch_size(batch_dict)
        pos_point, pos_mem = [], []
        # rows of one frame: a slice when the rows are grouped by frame (what the voxelizer, the point stream and the reference's
        # collate produce) — ONE host read for all frames and both tensors; a boolean mask per frame is a host sync each, forward
        # and backward, and a host that cannot run ahead of the device leaves it idle in every launch-bound stretch of the step
        vr, pr = _frame_ranges([coords[:, 0], point_c[:, 0]], B)
        if vr is not None and pr is not None:
            # Only the top-k itself is per frame (a pillar looks at the points of ITS frame, :101-104): with the picks as indices into
            # the whole point tensor, the gather, the attention weights and the memory addressing are row-wise and run ONCE for the
            # batch — the same arithmetic per row as the reference's loop over the frames, ~900 small launches less per step.
            with torch.no_grad():
                picks = [self._topk_points(pf[v0:v1].detach(), point_f[p0:p1].detach()) + p0
                         for (v0, v1), (p0, p1) in zip(vr, pr) if v1 > v0]
            idx = torch.cat(picks, 0) if picks else torch.zeros((0, self.k), dtype=torch.long, device=pf.device)
            plan = _EdgePlan(idx, point_f.shape[0])
            positives = _GatherRows.apply(point_f, idx, plan)                                  # (M, k, C)
            wgt = torch.softmax((pf.unsqueeze(1) * positives).sum(dim=2), dim=1)               # get_score, :76-83
            pos_point = (wgt.detach().unsqueeze(2) * positives).sum(dim=1)
            # the memory's second input IS positives = point_f[idx] (T1): addressed once per point, then gathered (MemoryUnit_Agg)
            pos_mem = self.memory.forward_train_indexed(pf, self.k, point_f, idx, plan)["output"]
        else:       # rows not grouped by frame: the reference's boolean masks, frame by frame
            for b in range(B):
                agg, positives = self.get_score(point_f[point_c[:, 0] == b], pf[coords[:, 0] == b])
                pos_point.append(agg)
                pos_mem.append(self.memory(pf[coords[:, 0] == b], self.k, positives)["output"])
            pos_point, pos_mem = torch.cat(pos_point, 0), torch.cat(pos_mem, 0)
        ws = self._workspace(B, pf.device)
        args = (coords, B, self.nx, self.ny, ws)
        batch_dict["spatial_features"] = _ScatterCanvas.apply(torch.cat([pf.detach(), pos_mem], dim=1), *args)
        batch_dict["spatial_features_point"] = _ScatterCanvas.apply(torch.cat([pf, pos_point], dim=1), *args)
        batch_dict["spatial_scale_features"] = _ScatterCanvas.apply(sf, *args)
        batch_dict["point_positive_features"] = pos_point
        batch_dict["memory_positive_features"] = pos_mem
        batch_dict["memory_items"] = self.memory.weight
        return batch_dict

    def forward(self, batch_dict, **kwargs):
        if self.training:
            return self._forward_train(batch_dict)
        pf, sf = batch_dict["pillar_features"], batch_dict["pillar_scale_features"]
        md = batch_dict.get("voxel_count_device")
        B = _batch_size(batch_dict)
        bank = self.memory.packed_bank()
        if pf.shape[1] == 64 and sf.shape[1] == 32 and bank.shape[1] == 64:   # hvpr.yaml widths: fused read-out + scatter
            _, sp, sc = kernels.memory_scatter_fwd(pf.contiguous(), sf.contiguous(), _coords_i32(batch_dict), bank, self.k, B,
                                                   self.nx, self.ny, self._workspace(B, pf.device), m_device=md,
                                                   out=batch_dict.get("_out_spatial"))
        else:
            mem = kernels.memory_readout_fwd(pf.contiguous(), bank, self.k, m_device=md)
            sp, sc = kernels.scatter_bev_fwd(pf.contiguous(), mem, sf.contiguous(), _coords_i32(batch_dict), B, self.nx,
                                             self.ny, self._workspace(B, pf.device), m_device=md)
        batch_dict["spatial_features"] = sp           # (B, 128, ny, nx): ch 0-63 pillar (detached), 64-127 memory
        batch_dict["spatial_scale_features"] = sc     # (B, 32, ny, nx)
        return batch_dict


__all__ = {
    "PointPillarScatter": PointPillarScatter,
    "PointPillarScatter_Agg_Memory_1_scale": PointPillarScatter_Agg_Memory_1_scale,
}
