"""PointNet++ point stream (training only) with the reference's interfaces:

  * the ops of the absent native package pcdet/ops/pointnet2/pointnet2_batch (setup.py:94-109), all HIP kernels behind the
    C-ABI: furthest point sampling, ball query, three-NN (indices) and grouping / gather / three-interpolate with their
    backward (autograd.Function over hvpr_group_points_f32 / hvpr_three_interpolate_f32 and their _grad twins);
  * PointnetSAModuleMSG / PointnetFPModule with the constructor kwargs used at
    pcdet/models/backbones_3d/pointnet2_backbone.py:27-34,43-47 and OpenPCDet's parameter names (mlps.{i}.{j}, mlp.{j}); their
    shared MLPs (1x1 convolution + train-mode BatchNorm + ReLU) run on the library's matrix-core convolution and BatchNorm kernels
    over a ROW layout (points x channels), with hvpr_group_rows / hvpr_max_samples / hvpr_fp_rows around them (csrc/point_mlp.hip);
  * PointNet2MSG, pcdet/models/backbones_3d/pointnet2_backbone.py:9-95 (registry key of backbones_3d).
"""
import torch
import torch.nn as nn

from . import kernels
from ._lib import check, lib


# ------------------------------------------------------------------------------------------------ index ops (HIP)
def furthest_point_sample(xyz, npoint):
    """xyz (B,N,3) f32 cuda -> idx (B,npoint) i32; first pick is index 0, ties -> lowest index."""
    B, N, _ = xyz.shape
    idx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    check(lib().hvpr_furthest_point_sample_f32(kernels._ptr(xyz.contiguous(), torch.float32, "xyz"), B, N, int(npoint),
                                               idx.data_ptr(), kernels._stream()), "hvpr_furthest_point_sample_f32")
    return idx


def ball_query(radius, nsample, xyz, new_xyz):
    """xyz (B,N,3), new_xyz (B,M,3) -> idx (B,M,nsample) i32 (first nsample within radius, first hit pre-fills)."""
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=xyz.device)
    check(lib().hvpr_ball_query_f32(kernels._ptr(xyz.contiguous(), torch.float32, "xyz"),
                                    kernels._ptr(new_xyz.contiguous(), torch.float32, "new_xyz"), B, N, M, float(radius),
                                    int(nsample), idx.data_ptr(), kernels._stream()), "hvpr_ball_query_f32")
    return idx


def three_nn(unknown, known):
    """unknown (B,n,3), known (B,m,3) -> dist (B,n,3) ascending, idx (B,n,3) i32."""
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
    check(lib().hvpr_three_nn_f32(kernels._ptr(unknown.contiguous(), torch.float32, "unknown"),
                                  kernels._ptr(known.contiguous(), torch.float32, "known"), B, n, m, dist.data_ptr(),
                                  idx.data_ptr(), kernels._stream()), "hvpr_three_nn_f32")
    return dist, idx


# ------------------------------------------------------------------------------------------------ differentiable gathers (HIP)
class _GroupPoints(torch.autograd.Function):
    """grouping_operation of pointnet2_batch: features (B,C,N), idx (B,np,ns) i32 -> (B,C,np,ns); backward = scatter-add."""

    @staticmethod
    def forward(ctx, features, idx):
        features = features.contiguous()
        B, C, N = features.shape
        _, np_, ns = idx.shape
        out = torch.empty((B, C, np_, ns), dtype=torch.float32, device=features.device)
        check(lib().hvpr_group_points_f32(kernels._ptr(features, torch.float32, "features"), kernels._ptr(idx, torch.int32, "idx"),
                                          B, C, N, np_, ns, out.data_ptr(), kernels._stream()), "hvpr_group_points_f32")
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        grad = grad.contiguous()
        B, C, np_, ns = grad.shape
        gf = torch.empty((B, C, ctx.N), dtype=torch.float32, device=grad.device)
        check(lib().hvpr_group_points_grad_f32(kernels._ptr(grad, torch.float32, "grad_out"), idx.data_ptr(), B, C, ctx.N, np_, ns,
                                               gf.data_ptr(), kernels._stream()), "hvpr_group_points_grad_f32")
        return gf, None


class _ThreeInterpolate(torch.autograd.Function):
    """three_interpolate of pointnet2_batch: features (B,C,m), idx / weight (B,n,3) -> (B,C,n); the weights carry no gradient
    (they come from three_nn distances, as in the reference's op)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        features, weight = features.contiguous(), weight.contiguous()
        B, C, m = features.shape
        n = idx.shape[1]
        out = torch.empty((B, C, n), dtype=torch.float32, device=features.device)
        check(lib().hvpr_three_interpolate_f32(kernels._ptr(features, torch.float32, "features"), kernels._ptr(idx, torch.int32, "idx"),
                                               kernels._ptr(weight, torch.float32, "weight"), B, C, m, n, out.data_ptr(),
                                               kernels._stream()), "hvpr_three_interpolate_f32")
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad):
        idx, weight = ctx.saved_tensors
        grad = grad.contiguous()
        B, C, n = grad.shape
        gf = torch.empty((B, C, ctx.m), dtype=torch.float32, device=grad.device)
        check(lib().hvpr_three_interpolate_grad_f32(kernels._ptr(grad, torch.float32, "grad_out"), idx.data_ptr(), weight.data_ptr(),
                                                    B, C, ctx.m, n, gf.data_ptr(), kernels._stream()), "hvpr_three_interpolate_grad_f32")
        return gf, None, None


def _i32(idx):
    return (idx if idx.dtype == torch.int32 else idx.to(torch.int32)).contiguous()


def gather_operation(features, idx):
    """features (B,C,N), idx (B,np) -> (B,C,np) (hvpr_group_points_f32 with one sample per group)."""
    return _GroupPoints.apply(features, _i32(idx).unsqueeze(-1)).squeeze(-1)


def grouping_operation(features, idx):
    """features (B,C,N), idx (B,np,ns) -> (B,C,np,ns); backward scatter-adds into (B,C,N)."""
    return _GroupPoints.apply(features, _i32(idx))


def three_interpolate(features, idx, weight):
    """features (B,C,m), idx (B,n,3), weight (B,n,3) -> (B,C,n)."""
    return _ThreeInterpolate.apply(features, _i32(idx), weight.detach())


# ------------------------------------------------------------------------------------------------ row-layout pieces (HIP)
def _cpad(c):
    return (c + 7) // 8 * 8


@torch.no_grad()
def group_edges(idx, n_per_batch):
    """(order, chunk_ptr, dest_ptr) of kernels.edges_by_destination for an index tensor idx (B, ...) into n_per_batch rows per sample: the
    edges of the backward of a gather, grouped by the row they scatter to.  Depends on idx only, so the index plan of a batch
    (PointNet2MSG.index_plan, computed ahead on a side stream) carries it and the backward does not have to sort."""
    B = idx.shape[0]
    dst = idx.reshape(B, -1).to(torch.int64) + torch.arange(B, device=idx.device, dtype=torch.int64).view(B, 1) * n_per_batch
    return kernels.edges_by_destination(dst, B * n_per_batch)


class _GroupRows(torch.autograd.Function):
    """QueryAndGroup (use_xyz) in ROW layout: xyz (B,N,3), features (B,N,C) or None, new_xyz (B,np,3), idx (B,np,ns) i32 ->
    (B*np*ns, cpad) rows [xyz[idx] - new_xyz | features[idx] | 0]; backward scatter-adds the feature columns (hvpr_group_rows_*)."""

    @staticmethod
    def forward(ctx, xyz, features, new_xyz, idx, cpad, csr=None):
        B, N, _ = xyz.shape
        _, np_, ns = idx.shape
        ctx.csr = csr
        C = 0 if features is None else features.shape[-1]
        xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
        features = None if features is None else features.contiguous()
        out = torch.empty((B * np_ * ns, cpad), dtype=torch.float32, device=xyz.device)
        check(lib().hvpr_group_rows_f32(kernels._ptr(xyz, torch.float32, "xyz"), kernels._ptr(features, torch.float32, "features"),
                                        kernels._ptr(new_xyz, torch.float32, "new_xyz"), kernels._ptr(idx, torch.int32, "idx"), B, N, C, np_, ns,
                                        cpad, out.data_ptr(), kernels._stream()), "hvpr_group_rows_f32")
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, C, np_, ns, cpad)
        return out

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        B, N, C, np_, ns, cpad = ctx.dims
        if C == 0 or not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None
        grad = grad.contiguous()
        # a point belongs to many groups: its gradient is the sum over the (group, sample) rows that picked it, taken row by row in
        # ascending row order (kernels.edges_by_destination): no float atomics, the same bits every run
        order, chunk_ptr, dest_ptr = ctx.csr if ctx.csr is not None else group_edges(idx, N)
        gf = kernels.segment_sum_rows(grad, 3, C, order, None, chunk_ptr, dest_ptr, B * N).view(B, N, C)
        return None, gf, None, None, None, None


class _MaxSamples(torch.autograd.Function):
    """y (G*ns, C) rows -> (G, C): max over the ns samples of every group (the max-pool over the sample axis of
    PointnetSAModuleMSG); the gradient goes to the arg-max sample."""

    @staticmethod
    def forward(ctx, y, ns):
        y = y.contiguous()
        C = y.shape[1]
        G = y.shape[0] // ns
        out = torch.empty((G, C), dtype=torch.float32, device=y.device)
        arg = torch.empty((G, C), dtype=torch.uint8, device=y.device)
        check(lib().hvpr_max_samples_f32(kernels._ptr(y, torch.float32, "y"), G, ns, C, out.data_ptr(), arg.data_ptr(), kernels._stream()),
              "hvpr_max_samples_f32")
        ctx.save_for_backward(arg)
        ctx.ns = ns
        return out

    @staticmethod
    def backward(ctx, grad):
        (arg,) = ctx.saved_tensors
        grad = grad.contiguous()
        G, C = grad.shape
        gy = torch.empty((G * ctx.ns, C), dtype=torch.float32, device=grad.device)
        check(lib().hvpr_max_samples_grad_f32(kernels._ptr(grad, torch.float32, "grad_out"), arg.data_ptr(), G, ctx.ns, C, gy.data_ptr(),
                                              kernels._stream()), "hvpr_max_samples_grad_f32")
        return gy, None


class _FpRows(torch.autograd.Function):
    """PointnetFPModule's input in ROW layout: known (B,m,C1), idx / weight (B,n,3), skip (B,n,C2) or None -> (B*n, cpad) rows
    [three_interpolate(known) | skip | 0]  (hvpr_fp_rows_*; the weights carry no gradient, as in the reference's op)."""

    @staticmethod
    def forward(ctx, known, idx, weight, skip, cpad, csr=None):
        ctx.csr = csr
        known, weight = known.contiguous(), weight.contiguous()
        skip = None if skip is None else skip.contiguous()
        B, m, C1 = known.shape
        n = idx.shape[1]
        C2 = 0 if skip is None else skip.shape[-1]
        out = torch.empty((B * n, cpad), dtype=torch.float32, device=known.device)
        check(lib().hvpr_fp_rows_f32(kernels._ptr(known, torch.float32, "known"), kernels._ptr(idx, torch.int32, "idx"),
                                     kernels._ptr(weight, torch.float32, "weight"), kernels._ptr(skip, torch.float32, "skip"), B, m, n, C1, C2, cpad,
                                     out.data_ptr(), kernels._stream()), "hvpr_fp_rows_f32")
        ctx.save_for_backward(idx, weight)
        ctx.dims = (B, m, n, C1, C2, cpad)
        return out

    @staticmethod
    def backward(ctx, grad):
        idx, weight = ctx.saved_tensors
        B, m, n, C1, C2, cpad = ctx.dims
        grad = grad.contiguous()
        # a known point feeds many unknown ones: edge (row, k) carries weight[row, k] times the row's gradient to known point idx[row, k];
        # summed per known point in ascending edge order (no float atomics)
        order, chunk_ptr, dest_ptr = ctx.csr if ctx.csr is not None else group_edges(idx, m)
        edge_row = torch.div(order, 3, rounding_mode="floor").to(torch.int32)
        edge_w = weight.reshape(-1)[order.long()].contiguous()
        gk = kernels.segment_sum_rows(grad, 0, C1, edge_row, edge_w, chunk_ptr, dest_ptr, B * m).view(B, m, C1)
        gs = grad[:, C1:C1 + C2].reshape(B, n, C2).contiguous() if (C2 > 0 and ctx.needs_input_grad[3]) else None
        return gk, None, None, gs, None, None


def _rows_image(x):
    """(S, C) rows as the (1, H, W, C) image the convolution / BatchNorm kernels take (any factorisation is the same 1x1 conv)."""
    S = x.shape[0]
    w = 64
    while w > 1 and S % w:
        w //= 2
    return x.view(1, S // w, w, x.shape[1])


def shared_mlp_rows(seq, x):
    """The shared MLP `seq` = [Conv2d 1x1 (no bias), BatchNorm2d, ReLU] x L of pointnet2_backbone.py:27-47 on rows x (S, cpad):
    every layer = hvpr_conv2d_nhwc_f32 (forward / data gradient) + hvpr_conv2d_wgrad_nhwc_f32 + the train-mode hvpr_bn_* kernels
    (batch statistics over all S rows, differentiated through; running statistics updated like nn.BatchNorm2d)."""
    from . import conv_train as ct
    mods = list(seq)
    assert len(mods) % 3 == 0
    t = _rows_image(x)
    for k in range(0, len(mods), 3):
        conv, bn = mods[k], mods[k + 1]
        w = conv.weight
        if conv.bias is not None or tuple(conv.kernel_size) != (1, 1) or w.shape[0] % 8 != 0:
            raise ValueError("hvpr_amd: the point-stream MLP kernels take 1x1 convolutions without bias and widths that are multiples of 8")
        pad = t.shape[-1] - w.shape[1]
        assert pad >= 0
        if pad:
            w = torch.cat([w, w.new_zeros((w.shape[0], pad, 1, 1))], dim=1)
        t = ct.bn_relu(ct.conv(t, w), bn)
    return t.reshape(x.shape[0], -1)


# ------------------------------------------------------------------------------------------------ modules
class QueryAndGroup(nn.Module):
    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None, idx=None):
        """The reference's channel-major form: (B, 3 + C, npoint, nsample), xyz channels first."""
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped   # xyz channels first


def _shared_mlp(widths):
    layers = []
    for cin, cout in zip(widths[:-1], widths[1:]):
        layers += [nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]
    return nn.Sequential(*layers)


class PointnetSAModuleMSG(nn.Module):
    """Set abstraction with multi-scale grouping: FPS -> per scale (ball query, group, shared MLP, max over samples) -> concat.
    Everything runs on the library's kernels in ROW layout (points x channels); forward() keeps the reference's channel-major
    signature, forward_rows() is what PointNet2MSG chains."""

    def __init__(self, *, npoint, radii, nsamples, mlps, use_xyz=True, bn=True):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        if not use_xyz or not bn:
            raise NotImplementedError("hvpr_amd: PointnetSAModuleMSG is built for use_xyz=True, bn=True (pointnet2_backbone.py:27-34)")
        self.npoint = npoint
        self.groupers = nn.ModuleList(QueryAndGroup(r, n, use_xyz) for r, n in zip(radii, nsamples))
        self.mlps = nn.ModuleList()
        for spec in mlps:
            spec = list(spec)
            spec[0] += 3
            self.mlps.append(_shared_mlp(spec))

    def indices(self, xyz):
        """The index half of forward — it depends on the coordinates only: (FPS idx, new_xyz, [ball-query idx per scale], [the
        edges of each scale's backward grouped by point, group_edges])."""
        idx = furthest_point_sample(xyz, self.npoint)
        new_xyz = gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
        balls = [ball_query(g.radius, g.nsample, xyz, new_xyz) for g in self.groupers]
        return idx, new_xyz, balls, [group_edges(b, xyz.shape[1]) for b in balls]

    def forward_rows(self, xyz, features=None, pre=None):
        """xyz (B,N,3), features (B,N,C) rows or None -> new_xyz (B,npoint,3), features (B,npoint,C') rows."""
        idx, new_xyz, balls, csrs = pre if pre is not None else self.indices(xyz)
        B = xyz.shape[0]
        C = 0 if features is None else features.shape[-1]
        outs = []
        for grouper, mlp, bidx, csr in zip(self.groupers, self.mlps, balls, csrs):
            x = _GroupRows.apply(xyz, features, new_xyz, bidx, _cpad(3 + C), csr)     # (B*npoint*nsample, cpad)
            outs.append(_MaxSamples.apply(shared_mlp_rows(mlp, x), grouper.nsample))  # (B*npoint, C')
        return new_xyz, torch.cat(outs, dim=1).view(B, self.npoint, -1)

    def forward(self, xyz, features=None, pre=None):
        """Reference signature: features (B,C,N) -> (B,C',npoint)."""
        rows = None if features is None else features.transpose(1, 2).contiguous()
        new_xyz, out = self.forward_rows(xyz, rows, pre=pre)
        return new_xyz, out.transpose(1, 2).contiguous()


def three_nn_plan(unknown, known):
    """(dist, idx) of three_nn + the edges of the interpolation's backward grouped by known point."""
    dist, idx = three_nn(unknown, known)
    return dist, idx, group_edges(idx, known.shape[1])


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance interpolation from the 3 nearest known points, concat skip, shared MLP — rows."""

    def __init__(self, *, mlp, bn=True):
        super().__init__()
        self.mlp = _shared_mlp(list(mlp))

    def forward_rows(self, unknown, known, unknow_feats, known_feats, pre=None):
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,n,C2) rows or None, known_feats (B,m,C1) rows -> (B,n,C') rows."""
        dist, idx, csr = pre if pre is not None else three_nn_plan(unknown, known)
        w = 1.0 / (dist + 1e-8)
        w = w / w.sum(dim=2, keepdim=True)
        B, n = idx.shape[0], idx.shape[1]
        c = known_feats.shape[-1] + (0 if unknow_feats is None else unknow_feats.shape[-1])
        x = _FpRows.apply(known_feats, _i32(idx), w.detach(), unknow_feats, _cpad(c), csr)
        return shared_mlp_rows(self.mlp, x).view(B, n, -1)

    def forward(self, unknown, known, unknow_feats, known_feats, pre=None):
        """Reference signature: channel-major features (B,C,n) / (B,C,m) -> (B,C',n)."""
        uf = None if unknow_feats is None else unknow_feats.transpose(1, 2).contiguous()
        out = self.forward_rows(unknown, known, uf, known_feats.transpose(1, 2).contiguous(), pre=pre)
        return out.transpose(1, 2).contiguous()


def _tensors_of(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)


class PointNet2MSG(nn.Module):
    """Encoder/decoder producing one feature vector per input point — pointnet2_backbone.py:9-95."""

    def __init__(self, model_cfg, input_channels, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        sa = model_cfg.SA_CONFIG
        self.SA_modules = nn.ModuleList()
        cin = input_channels - 3
        skips = [cin]
        for k in range(len(sa.NPOINTS)):
            specs = [[cin] + list(m) for m in sa.MLPS[k]]
            self.SA_modules.append(PointnetSAModuleMSG(npoint=sa.NPOINTS[k], radii=list(sa.RADIUS[k]), nsamples=list(sa.NSAMPLE[k]),
                                                       mlps=specs, use_xyz=sa.get("USE_XYZ", True)))
            cin = sum(s[-1] for s in specs)
            skips.append(cin)
        fp = [list(m) for m in model_cfg.FP_MLPS]
        self.FP_modules = nn.ModuleList()
        for k in range(len(fp)):
            pre = fp[k + 1][-1] if k + 1 < len(fp) else cin
            self.FP_modules.append(PointnetFPModule(mlp=[pre + skips[k]] + fp[k]))
        self.num_point_features = fp[0][-1]

    @torch.no_grad()
    def index_plan(self, points, batch_size):
        """Every index tensor of the forward — furthest-point samples, ball-query groups, three nearest neighbours — from the
        point coordinates alone (none of them depends on a weight): {"sa": [(fps idx, new_xyz, [ball idx], [backward edges])],
        "fp": {i: (dist, idx, backward edges)}}.  A training loop can compute the plan of the NEXT batch on a side stream while the current step runs
        (detector.prefetch_point_indices); forward() takes it from batch_dict["_pn2_plan"]."""
        xyz = points[:, 1:4].contiguous().view(batch_size, -1, 3)
        l_xyz, sa_plan = [xyz], []
        for sa in self.SA_modules:
            pre = sa.indices(l_xyz[-1])
            sa_plan.append(pre)
            l_xyz.append(pre[1])
        fp_plan = {i: three_nn_plan(l_xyz[i - 1], l_xyz[i]) for i in range(-1, -(len(self.FP_modules) + 1), -1)}
        return {"sa": sa_plan, "fp": fp_plan}

    def forward(self, batch_dict):
        B = batch_dict["batch_size"]
        pts = batch_dict["points"]
        bidx, xyz, feats = pts[:, 0], pts[:, 1:4].contiguous(), (pts[:, 4:].contiguous() if pts.shape[1] > 4 else None)
        assert pts.shape[0] % B == 0, "PointNet2MSG needs the same number of points in every sample (pointnet2_backbone.py:76)"
        xyz = xyz.view(B, -1, 3)
        feats = feats.view(B, -1, feats.shape[-1]) if feats is not None else None          # ROW layout (B, N, C) end to end
        plan = batch_dict.pop("_pn2_plan", None)
        if plan is not None:                     # computed ahead on another stream: wait for it, keep its memory alive for us
            plan, ready = plan
            cur = torch.cuda.current_stream()
            cur.wait_event(ready)
            for t in _tensors_of(plan):
                t.record_stream(cur)
        l_xyz, l_feat = [xyz], [feats]
        for k, sa in enumerate(self.SA_modules):
            nx, nf = sa.forward_rows(l_xyz[-1], l_feat[-1], pre=plan["sa"][k] if plan is not None else None)
            l_xyz.append(nx)
            l_feat.append(nf)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            l_feat[i - 1] = self.FP_modules[i].forward_rows(l_xyz[i - 1], l_xyz[i], l_feat[i - 1], l_feat[i],
                                                            pre=plan["fp"][i] if plan is not None else None)
        pf = l_feat[0]
        batch_dict["point_features"] = pf.reshape(-1, pf.shape[-1])
        batch_dict["point_coords"] = torch.cat((bidx[:, None].float(), l_xyz[0].reshape(-1, 3)), dim=1)
        batch_dict["point_batch_idx"] = bidx
        return batch_dict


__all__ = {
    "PointNet2MSG": PointNet2MSG,
}
