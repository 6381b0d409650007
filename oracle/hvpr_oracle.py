"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy + CPU torch, fp32) of the reference's forward hot path, stage by
stage.  Nothing under ``hvpr_amd/`` may import this module: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and only as
the checker / the reported CPU baseline.

Pinning (see DESIGN.md "Oracle"):
  * PINNED against the importable reference modules (fixtures under tests/golden/, made by
    tests/golden/make_golden.py in the build container): pillar VFE, memory read-out,
    scatter, BEV backbone, anchors + box decode, limit_period, ResidualCoder.
  * PARITY UNPINNED (source absent from /root/reference, no reference test exists):
    voxelizer (third-party spconv v1.x), rotated-BEV NMS / IoU (pcdet/ops/iou3d_nms).
    Those live in the C files next to this one and follow SURVEY.md Appendix B.

Every function cites the reference file:line it restates (paths relative to /root/reference).
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """The C half of the oracle (voxelize_ref.c, iou3d_nms_ref.c -> liboracle.so)."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        lib = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        lib.hvpr_oracle_voxelize.restype = ctypes.c_int
        lib.hvpr_oracle_voxelize.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp, ip, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, fp, ip, ip, ip]
        lib.hvpr_oracle_box_overlap.restype = ctypes.c_float
        lib.hvpr_oracle_box_overlap.argtypes = [fp, fp]
        for name in ("hvpr_oracle_boxes_overlap_bev", "hvpr_oracle_boxes_iou_bev", "hvpr_oracle_boxes_iou3d"):
            fn = getattr(lib, name)
            fn.restype = None
            fn.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, fp]
        lib.hvpr_oracle_nms_sorted.restype = ctypes.c_int
        lib.hvpr_oracle_nms_sorted.argtypes = [fp, ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_int64)]
        _LIB = lib
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


# ----------------------------------------------------------------------------------------
# a1  voxelizer — pcdet/datasets/processor/data_processor.py:43-75 -> spconv VoxelGenerator
# ----------------------------------------------------------------------------------------
def grid_size_of(point_cloud_range, voxel_size):
    """data_processor.py:56-57 — round((hi-lo)/vs) with the range held in float32 (dataset.py:25)."""
    r = np.asarray(point_cloud_range, dtype=np.float32)
    g = (r[3:6] - r[0:3]) / np.asarray(voxel_size, dtype=np.float32)
    return np.round(g).astype(np.int64)


def voxelize(points, voxel_size, point_cloud_range, max_points, max_voxels, mode="v2"):
    """Sequential first-touch voxel hash (C).  Returns voxels (M,P,C) f32, coords (M,3) i32 zyx, num (M,) i32."""
    pts = _f32(points)
    n, c = pts.shape
    rng = _f32(point_cloud_range)
    vs = _f32(voxel_size)
    grid = grid_size_of(point_cloud_range, voxel_size).astype(np.int32)
    voxels = np.zeros((max_voxels, max_points, c), dtype=np.float32)
    coords = np.zeros((max_voxels, 3), dtype=np.int32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    cmap = np.full((int(grid[2]) * int(grid[1]) * int(grid[0]),), -1, dtype=np.int32)
    lo = np.ascontiguousarray(rng[:3])
    m = _lib().hvpr_oracle_voxelize(_fp(pts), n, c, _fp(lo), _fp(vs), _ip(grid), int(max_points), int(max_voxels),
                                    1 if mode == "v1" else 0, _fp(voxels), _ip(coords), _ip(num), _ip(cmap))
    assert (cmap == -1).all()
    return voxels[:m].copy(), coords[:m].copy(), num[:m].copy()


def voxelize_py(points, voxel_size, point_cloud_range, max_points, max_voxels, mode="v2"):
    """Pure-python twin of the C loop (small inputs only) — cross-checks the C build."""
    pts = _f32(points)
    rng = _f32(point_cloud_range)
    vs = _f32(voxel_size)
    grid = grid_size_of(point_cloud_range, voxel_size)
    table = {}
    voxels, coords, num = [], [], []
    for i in range(pts.shape[0]):
        c = np.floor((pts[i, :3] - rng[:3]) / vs)          # fp32 sub, div, floor
        if (c < 0).any() or (c >= grid.astype(np.float32)).any():
            continue
        key = (int(c[2]), int(c[1]), int(c[0]))
        vid = table.get(key)
        if vid is None:
            if len(coords) >= max_voxels:
                if mode == "v1":
                    break
                continue
            vid = len(coords)
            table[key] = vid
            coords.append(key)
            voxels.append(np.zeros((max_points, pts.shape[1]), np.float32))
            num.append(0)
        if num[vid] < max_points:
            voxels[vid][num[vid]] = pts[i]
            num[vid] += 1
    if not coords:
        return (np.zeros((0, max_points, pts.shape[1]), np.float32), np.zeros((0, 3), np.int32),
                np.zeros((0,), np.int32))
    return np.stack(voxels), np.asarray(coords, np.int32), np.asarray(num, np.int32)


# ----------------------------------------------------------------------------------------
# a2  pillar VFE — pcdet/models/backbones_3d/vfe/pillar_vfe.py:184-221, PFNLayer :29-49
# ----------------------------------------------------------------------------------------
def _bn_eval(x, bn, eps=1e-3):
    """Eval-mode BatchNorm over the last (channel) dim with running stats."""
    w, b, mean, var = bn
    return (x - mean) / torch.sqrt(var + eps) * w + b


def _bn_train(x2d, w, b, eps=1e-3):
    """Train-mode BatchNorm: biased batch variance over dim 0."""
    mean = x2d.mean(dim=0)
    var = x2d.var(dim=0, unbiased=False)
    return (x2d - mean) / torch.sqrt(var + eps) * w + b, mean, var


def pillar_vfe_scale(voxels, num_points, coords, params, voxel_size, point_cloud_range, training=False):
    """Restates PillarVFE_Scale.forward with USE_ABSLOTE_XYZ=True, WITH_DISTANCE=False.

    voxels (M,P,4) f32, num_points (M,) f32, coords (M,4) f32 [b,z,y,x].
    params: dict of torch tensors keyed like the reference state_dict
        pfn_layers.{0,1}.linear.weight / .norm.{weight,bias,running_mean,running_var}
        pfn_scale_layers.{0,1}.0.weight / .1.{weight,bias,running_mean,running_var}
    Returns pillar_features (M,C1), pillar_scale_features (M,Cs), pillar_mask (M,P,1)
    (+ list of batch (mean,var) per BN when training=True).
    """
    v = torch.as_tensor(voxels, dtype=torch.float32)
    n = torch.as_tensor(num_points, dtype=torch.float32)
    c = torch.as_tensor(coords, dtype=torch.float32)
    M, P, _ = v.shape
    vx, vy, vz = [float(s) for s in voxel_size]
    x_off = vx / 2 + point_cloud_range[0]                    # pillar_vfe.py:166-171
    y_off = vy / 2 + point_cloud_range[1]
    z_off = vz / 2 + point_cloud_range[2]
    mean = v[:, :, :3].sum(dim=1, keepdim=True) / n.view(-1, 1, 1)       # :187
    f_cluster = v[:, :, :3] - mean                                       # :188
    f_center = torch.zeros_like(v[:, :, :3])
    f_center[:, :, 0] = v[:, :, 0] - (c[:, 3].unsqueeze(1) * vx + x_off)   # :190-193
    f_center[:, :, 1] = v[:, :, 1] - (c[:, 2].unsqueeze(1) * vy + y_off)
    f_center[:, :, 2] = v[:, :, 2] - (c[:, 1].unsqueeze(1) * vz + z_off)
    feats = torch.cat([v, f_cluster, f_center], dim=-1)                  # :195-203  (M,P,10)
    slot = torch.arange(P, dtype=torch.int32).view(1, -1)
    mask = (n.int().view(-1, 1) > slot).unsqueeze(-1).float()            # :205-207
    feats = feats * mask                                                 # :208
    stats = []
    x = feats
    for li in (0, 1):
        w = params[f"pfn_layers.{li}.linear.weight"]
        bn = [params[f"pfn_layers.{li}.norm.{k}"] for k in ("weight", "bias", "running_mean", "running_var")]
        y = x @ w.t()                                                    # Linear, no bias (:22)
        if training:
            flat, bm, bv = _bn_train(y.reshape(M * P, -1), bn[0], bn[1])   # stats over all M*P slots
            y = flat.reshape(M, P, -1)
            stats.append((bm, bv))
        else:
            y = _bn_eval(y, bn)
        y = torch.relu(y)
        y_max = y.max(dim=1, keepdim=True)[0]                            # :42
        x = y_max if li == 1 else torch.cat([y, y_max.expand(-1, P, -1)], dim=2)   # :44-49
    # NB the reference does .squeeze(), which collapses M==1 to (C,) (:211); the oracle keeps (M,C).
    pillar = x.reshape(M, -1)
    d_mean = torch.norm(mean, 2, 2, keepdim=True)                        # :213
    s = torch.cat([n.unsqueeze(1), d_mean.squeeze(1), mean.squeeze(1)], dim=-1)   # :214  (M,5)
    for li in (0, 1):
        w = params[f"pfn_scale_layers.{li}.0.weight"]
        bn = [params[f"pfn_scale_layers.{li}.1.{k}"] for k in ("weight", "bias", "running_mean", "running_var")]
        s = s @ w.t()
        if training:
            s, bm, bv = _bn_train(s, bn[0], bn[1])
            stats.append((bm, bv))
        else:
            s = _bn_eval(s, bn)
        s = torch.relu(s)
    if training:
        return pillar, s, mask, stats
    return pillar, s, mask


# ----------------------------------------------------------------------------------------
# a3  memory read-out (eval) — map_to_bev/memory_module.py:60-77 (line 75 is a stray fragment)
# ----------------------------------------------------------------------------------------
def memory_readout_eval(f, W, k):
    """f (M,C), W (items,C) -> output (M,C), indices (M,k) (descending score), logits (M,items)."""
    f = torch.as_tensor(f, dtype=torch.float32)
    W = torch.as_tensor(W, dtype=torch.float32)
    logits = f @ W.t()                                                   # :64
    score = torch.softmax(logits, dim=1)                                 # :65
    _, idx = torch.topk(score, k, dim=1)                                 # :66
    mp = W[idx]                                                          # :67  (M,k,C)
    agg = (mp * f.unsqueeze(1)).sum(dim=2)                               # :70-71
    agg = torch.softmax(agg, dim=1)                                      # :72
    out = (agg.unsqueeze(2) * mp).sum(dim=1)                             # :73-74
    return out, idx, logits


# ----------------------------------------------------------------------------------------
# a4  scatter (eval) — map_to_bev/pointpillar_scatter.py:169-222
# ----------------------------------------------------------------------------------------
def scatter_eval(pillar_features, memory_out, pillar_scale_features, coords, batch_size, nx, ny):
    """-> spatial_features (B,2C,ny,nx) [pillar | memory], spatial_scale_features (B,Cs,ny,nx)."""
    pf = torch.as_tensor(pillar_features, dtype=torch.float32)
    mo = torch.as_tensor(memory_out, dtype=torch.float32)
    sf = torch.as_tensor(pillar_scale_features, dtype=torch.float32)
    c = torch.as_tensor(coords, dtype=torch.float32)
    C, Cs = pf.shape[1], sf.shape[1]
    canv = torch.zeros(batch_size, 2 * C, ny * nx)
    canv_s = torch.zeros(batch_size, Cs, ny * nx)
    for b in range(batch_size):
        m = c[:, 0] == b
        idx = (c[m, 1] + c[m, 2] * nx + c[m, 3]).long()                  # :192
        canv[b][:, idx] = torch.cat([pf[m], mo[m]], dim=1).t()           # :204,207
        canv_s[b][:, idx] = sf[m].t()                                    # :208
    return canv.view(batch_size, 2 * C, ny, nx), canv_s.view(batch_size, Cs, ny, nx)


# ----------------------------------------------------------------------------------------
# a5  BEV backbone (eval) — backbones_2d/base_bev_backbone.py:280-315, spatial_attention.py:47-63
# ----------------------------------------------------------------------------------------
def _conv_bn_relu(x, w, bn, stride=1, pad=1, eps=1e-3, bias=None, relu=True):
    y = F.conv2d(x, w, bias=bias, stride=stride, padding=pad)
    g, b, mean, var = bn
    y = (y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + eps) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    return torch.relu(y) if relu else y


def _bn_of(params, prefix):
    return [params[f"{prefix}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")]


def spatial_gate(y, params):
    """sigmoid(BN(conv3x3_{2->1,bias}(cat[max_c y, mean_c y])))  — spatial_attention.py:47-62."""
    pooled = torch.cat([y.max(dim=1, keepdim=True)[0], y.mean(dim=1, keepdim=True)], dim=1)
    a = _conv_bn_relu(pooled, params["attention.spatial.conv.weight"], _bn_of(params, "attention.spatial.norm"),
                      bias=params["attention.spatial.conv.bias"], relu=False)
    return torch.sigmoid(a)


def bev_backbone_eval(spatial_features, spatial_scale_features, params, layer_nums, layer_strides, sfm_layer_nums,
                      upsample_strides):
    """params keyed like the reference state_dict of BaseBEVBackbone_Scale."""
    x = torch.as_tensor(spatial_features, dtype=torch.float32)
    y = torch.as_tensor(spatial_scale_features, dtype=torch.float32)
    ups = []
    for i in range(len(layer_nums)):
        # blocks[i]: ZeroPad(1)+Conv3x3(stride)+BN+ReLU, then layer_nums[i] x (Conv3x3 pad1+BN+ReLU)  (:154-169)
        x = _conv_bn_relu(x, params[f"blocks.{i}.1.weight"], _bn_of(params, f"blocks.{i}.2"), stride=layer_strides[i])
        for k in range(layer_nums[i]):
            x = _conv_bn_relu(x, params[f"blocks.{i}.{4 + 3 * k}.weight"], _bn_of(params, f"blocks.{i}.{5 + 3 * k}"))
        y = _conv_bn_relu(y, params[f"scale_layers.{i}.1.weight"], _bn_of(params, f"scale_layers.{i}.2"),
                          stride=layer_strides[i])                        # :200-209
        x_att = x
        for _ in range(sfm_layer_nums[i]):                                # :291-295, shared weights
            t = _conv_bn_relu(x_att, params[f"sfmblocks_down.{i}.0.weight"], _bn_of(params, f"sfmblocks_down.{i}.1"))
            x_att = spatial_gate(y, params) * t + x_att
        s = upsample_strides[i]
        u = F.conv_transpose2d(x_att, params[f"deblocks.{i}.0.weight"], stride=s)   # :177-188
        g, b, mean, var = _bn_of(params, f"deblocks.{i}.1")
        u = (u - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-3) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
        ups.append(torch.relu(u))
    return torch.cat(ups, dim=1)                                          # :303-304


# ----------------------------------------------------------------------------------------
# a11  BEV backbone, TRAINING forward (two streams through shared weights) — backbones_2d/base_bev_backbone.py:228-279,
#      spatial_attention.py:47-63; train-mode BatchNorm2d(eps 1e-3, momentum 0.01) as torch defines it
# ----------------------------------------------------------------------------------------
def _bn2d_train(x, params, prefix, running, eps=1e-3, momentum=0.01):
    """Batch statistics over (N,H,W), biased variance in the normalisation, UNBIASED variance into the running estimate; one
    running-statistics update per CALL (the shared SFM / gate BatchNorms are called 6 / 18 times per forward at hvpr.yaml)."""
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    n = x.numel() // x.shape[1]
    with torch.no_grad():
        rm, rv = running[prefix + ".running_mean"], running[prefix + ".running_var"]
        running[prefix + ".running_mean"] = (1 - momentum) * rm + momentum * mean.detach().to(rm.dtype)
        running[prefix + ".running_var"] = (1 - momentum) * rv + momentum * (var.detach() * (n / max(n - 1, 1))).to(rv.dtype)
    xh = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + eps)
    return xh * params[prefix + ".weight"].view(1, -1, 1, 1) + params[prefix + ".bias"].view(1, -1, 1, 1)


def bev_backbone_train(spatial_features, spatial_features_point, spatial_scale_features, params, layer_nums, layer_strides,
                       sfm_layer_nums, upsample_strides, trace=None, relu_masks=None):
    """Training forward of BaseBEVBackbone_Scale (:228-279).  params: torch tensors keyed like the reference's state_dict
    (leaves may require grad: the function is differentiable through torch autograd, dtype follows the inputs).
    Returns (spatial_features_2d, spatial_features_point_2d, running) with running = the running statistics after the call.
    trace: optional list that receives (name, activation) for every block / scale / SFM step / deblock output, and
    (name + ".pre", pre-activation) for the ReLU inside it.  relu_masks: optional {name: bool tensor} — the ReLU of that op is
    replaced by a multiplication with the given mask: lets a test differentiate the function on the branch (the set of ReLU
    decisions) another fp32 implementation took, where a pre-activation within round-off of zero went the other way."""
    x, xp, y = spatial_features, spatial_features_point, spatial_scale_features
    running = {k: v.detach().clone() for k, v in params.items() if "running_" in k}

    def tr(name, t):
        if trace is not None:
            if t.requires_grad:
                t.retain_grad()
            trace.append((name, t))
        return t

    def relu(v, name):
        if trace is not None:
            trace.append((name + ".pre", v))
        if relu_masks is not None and name in relu_masks:
            return v * relu_masks[name].to(v.dtype)
        return torch.relu(v)

    def cbr(t, conv, bn, name, stride=1):
        return relu(_bn2d_train(F.conv2d(t, params[conv + ".weight"], stride=stride, padding=1), params, bn, running), name)

    def block(t, i, who):                                                # :154-169; ZeroPad2d(1) + pad-0 conv = pad-1 conv
        t = tr(f"L{i}.{who}.block0", cbr(t, f"blocks.{i}.1", f"blocks.{i}.2", f"L{i}.{who}.block0", layer_strides[i]))
        for k in range(layer_nums[i]):
            t = tr(f"L{i}.{who}.block{k + 1}", cbr(t, f"blocks.{i}.{4 + 3 * k}", f"blocks.{i}.{5 + 3 * k}", f"L{i}.{who}.block{k + 1}"))
        return t

    def attention(t, w):                                                 # spatial_attention.py:57-63
        pooled = torch.cat([w.max(dim=1, keepdim=True)[0], w.mean(dim=1, keepdim=True)], dim=1)
        a = F.conv2d(pooled, params["attention.spatial.conv.weight"], params["attention.spatial.conv.bias"], padding=1)
        return torch.sigmoid(_bn2d_train(a, params, "attention.spatial.norm", running)) * t

    def deblock(t, i, name):                                             # :177-188
        u = F.conv_transpose2d(t, params[f"deblocks.{i}.0.weight"], stride=upsample_strides[i])
        return relu(_bn2d_train(u, params, f"deblocks.{i}.1", running), name)

    ups, ups_p = [], []
    for i in range(len(layer_nums)):
        x = block(x, i, "x")                                             # :244-246 — the call ORDER fixes the running statistics
        xp = block(xp, i, "xp")
        y = tr(f"L{i}.y", cbr(y, f"scale_layers.{i}.1", f"scale_layers.{i}.2", f"L{i}.y", layer_strides[i]))
        xa, xpa = x, xp
        for j in range(sfm_layer_nums[i]):                               # :251-257
            xa = tr(f"L{i}.x.sfm{j}", attention(cbr(xa, f"sfmblocks_down.{i}.0", f"sfmblocks_down.{i}.1", f"L{i}.x.sfm{j}"), y) + xa)
            xpa = tr(f"L{i}.xp.sfm{j}", attention(cbr(xpa, f"sfmblocks_down.{i}.0", f"sfmblocks_down.{i}.1", f"L{i}.xp.sfm{j}"), y) + xpa)
        ups.append(tr(f"L{i}.x.up", deblock(xa, i, f"L{i}.x.up")))       # :260-262
        ups_p.append(tr(f"L{i}.xp.up", deblock(xpa, i, f"L{i}.xp.up")))
    return torch.cat(ups, dim=1), torch.cat(ups_p, dim=1), running


# ----------------------------------------------------------------------------------------
# a6/a7  head + anchors + decode — dense_heads/anchor_head_single.py:109-145,
#        anchor_head_template.py:293-340, target_assigner/anchor_generator.py:17-60,
#        utils/box_coder_utils.py:45-77, utils/common_utils.py:20-23
# ----------------------------------------------------------------------------------------
def limit_period(val, offset=0.5, period=math.pi):
    val = torch.as_tensor(val, dtype=torch.float32)
    return val - torch.floor(val / period + offset) * period


def generate_anchors(point_cloud_range, feature_map_size_xy, anchor_sizes, anchor_rotations, anchor_bottom_heights,
                     align_center=False):
    """-> (nz, ny, nx, n_size, n_rot, 7) f32, as anchor_generator.py:56-58."""
    # the reference holds the range as a float32 ndarray (dataset.py:25) and the feature-map size as
    # int64 (anchor_head_template.py:43); numpy scalar promotion decides the bits of the strides.
    r = np.asarray(point_cloud_range, dtype=np.float32)
    gx, gy = np.int64(feature_map_size_xy[0]), np.int64(feature_map_size_xy[1])
    if align_center:
        xs, ys = (r[3] - r[0]) / gx, (r[4] - r[1]) / gy
        xo, yo = xs / 2, ys / 2
    else:
        xs, ys = (r[3] - r[0]) / (gx - 1), (r[4] - r[1]) / (gy - 1)
        xo, yo = 0, 0
    x_shifts = torch.arange(r[0] + xo, r[3] + 1e-5, step=xs, dtype=torch.float32)
    y_shifts = torch.arange(r[1] + yo, r[4] + 1e-5, step=ys, dtype=torch.float32)
    z_shifts = torch.tensor(anchor_bottom_heights, dtype=torch.float32)
    sizes = torch.tensor(anchor_sizes, dtype=torch.float32)              # (S,3)
    rots = torch.tensor(anchor_rotations, dtype=torch.float32)           # (R,)
    nxx, nyy, nzz, S, R = len(x_shifts), len(y_shifts), len(z_shifts), sizes.shape[0], rots.shape[0]
    a = torch.zeros(nzz, nyy, nxx, S, R, 7)
    a[..., 0] = x_shifts.view(1, 1, -1, 1, 1)
    a[..., 1] = y_shifts.view(1, -1, 1, 1, 1)
    a[..., 2] = z_shifts.view(-1, 1, 1, 1, 1)
    a[..., 3:6] = sizes.view(1, 1, 1, S, 1, 3)
    a[..., 6] = rots.view(1, 1, 1, 1, R)
    a[..., 2] += a[..., 5] / 2                                           # :58
    return a


def residual_decode(enc, anchors):
    """ResidualCoder.decode_torch — box_coder_utils.py:45-77."""
    xa, ya, za, dxa, dya, dza, ra = torch.split(anchors, 1, dim=-1)
    xt, yt, zt, dxt, dyt, dzt, rt = torch.split(enc, 1, dim=-1)
    diag = torch.sqrt(dxa ** 2 + dya ** 2)
    return torch.cat([xt * diag + xa, yt * diag + ya, zt * dza + za,
                      torch.exp(dxt) * dxa, torch.exp(dyt) * dya, torch.exp(dzt) * dza, rt + ra], dim=-1)


def residual_encode(boxes, anchors):
    """ResidualCoder.encode_torch — box_coder_utils.py:13-43 (without the in-place clamp side effect)."""
    anchors = anchors.clone()
    boxes = boxes.clone()
    anchors[:, 3:6] = torch.clamp_min(anchors[:, 3:6], 1e-5)
    boxes[:, 3:6] = torch.clamp_min(boxes[:, 3:6], 1e-5)
    xa, ya, za, dxa, dya, dza, ra = torch.split(anchors[:, :7], 1, dim=-1)
    xg, yg, zg, dxg, dyg, dzg, rg = torch.split(boxes[:, :7], 1, dim=-1)
    diag = torch.sqrt(dxa ** 2 + dya ** 2)
    return torch.cat([(xg - xa) / diag, (yg - ya) / diag, (zg - za) / dza,
                      torch.log(dxg / dxa), torch.log(dyg / dya), torch.log(dzg / dza), rg - ra], dim=-1)


def head_forward(spatial_features_2d, params):
    """Three 1x1 convs + NHWC permute — anchor_head_single.py:109-127."""
    x = torch.as_tensor(spatial_features_2d, dtype=torch.float32)
    out = []
    for name in ("conv_cls", "conv_box", "conv_dir_cls"):
        y = F.conv2d(x, params[f"{name}.weight"], params[f"{name}.bias"])
        out.append(y.permute(0, 2, 3, 1).contiguous())
    return out


def generate_predicted_boxes(cls_preds, box_preds, dir_cls_preds, anchors, dir_offset, dir_limit_offset, num_dir_bins):
    """anchor_head_template.py:293-340 (single head, ResidualCoder)."""
    B = cls_preds.shape[0]
    anc = anchors.reshape(-1, 7)
    A = anc.shape[0]
    batch_cls = cls_preds.reshape(B, A, -1).float()
    enc = box_preds.reshape(B, A, -1)
    boxes = residual_decode(enc, anc.unsqueeze(0).expand(B, -1, -1))
    dirp = dir_cls_preds.reshape(B, A, -1)
    labels = torch.max(dirp, dim=-1)[1]
    period = 2 * np.pi / num_dir_bins
    rot = limit_period(boxes[..., 6] - dir_offset, dir_limit_offset, period)
    boxes = boxes.clone()
    boxes[..., 6] = rot + dir_offset + period * labels.to(boxes.dtype)
    return batch_cls, boxes


# ----------------------------------------------------------------------------------------
# a8  post-processing — detectors/detector3d_template.py:168-274, model_utils/model_nms_utils.py:6-25
# ----------------------------------------------------------------------------------------
def stable_order_desc(scores):
    """The build's defined tie rule: descending score, ascending index on ties."""
    s = np.asarray(scores, dtype=np.float32)
    return np.lexsort((np.arange(s.shape[0]), -s.astype(np.float64))).astype(np.int64)


def nms_sorted(sorted_boxes, thresh):
    """Mask + greedy sweep over boxes ALREADY in score order.  Returns the kept positions, ascending."""
    sb = np.ascontiguousarray(_f32(sorted_boxes)[:, :7])
    keep = np.zeros((sb.shape[0],), dtype=np.int64)
    nk = _lib().hvpr_oracle_nms_sorted(_fp(sb), sb.shape[0], float(thresh),
                                       keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return keep[:nk].copy()


def nms_bev(boxes, scores, thresh):
    """nms_gpu restatement: sort desc (stable), bit-mask, sweep.  Returns indices into `boxes`."""
    b = _f32(boxes)[:, :7]
    order = stable_order_desc(scores)
    return order[nms_sorted(b[order], thresh)]


def boxes_iou_bev(a, b):
    a, b = _f32(a)[:, :7].copy(), _f32(b)[:, :7].copy()
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    _lib().hvpr_oracle_boxes_iou_bev(_fp(a), a.shape[0], _fp(b), b.shape[0], _fp(out))
    return out


def boxes_overlap_bev(a, b):
    a, b = _f32(a)[:, :7].copy(), _f32(b)[:, :7].copy()
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    _lib().hvpr_oracle_boxes_overlap_bev(_fp(a), a.shape[0], _fp(b), b.shape[0], _fp(out))
    return out


def boxes_iou3d(a, b):
    a, b = _f32(a)[:, :7].copy(), _f32(b)[:, :7].copy()
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    _lib().hvpr_oracle_boxes_iou3d(_fp(a), a.shape[0], _fp(b), b.shape[0], _fp(out))
    return out


def class_agnostic_nms(box_scores, box_preds, score_thresh, nms_thresh, pre_maxsize, post_maxsize):
    """model_nms_utils.py:6-25 with the stable-descending tie rule.  Returns selected anchor ids + scores."""
    s = np.asarray(box_scores, dtype=np.float32)
    bx = _f32(box_preds)
    passing = np.nonzero(s >= np.float32(score_thresh))[0]
    if passing.size == 0:
        return np.zeros((0,), np.int64), np.zeros((0,), np.float32)
    ps = s[passing]
    top = stable_order_desc(ps)[: min(pre_maxsize, ps.shape[0])]          # torch.topk (:15)
    keep = nms_bev(bx[passing][top], ps[top], nms_thresh)[:post_maxsize]   # :17-20
    sel = passing[top[keep]]
    return sel.astype(np.int64), s[sel]


def multi_classes_nms(cls_scores, box_preds, score_thresh, nms_thresh, pre_maxsize, post_maxsize):
    """model_nms_utils.py:28-65: the class-agnostic chain once per class column, results concatenated in class order.
    Returns (scores, labels = class index k, boxes, selected anchor ids)."""
    s = np.asarray(cls_scores, dtype=np.float32)
    bx = _f32(box_preds)
    sc, lab, sel = [], [], []
    for k in range(s.shape[1]):
        ids, sk = class_agnostic_nms(s[:, k], bx, score_thresh, nms_thresh, pre_maxsize, post_maxsize)
        sc.append(sk); sel.append(ids); lab.append(np.full((len(ids),), k, np.int64))
    sel = np.concatenate(sel)
    return np.concatenate(sc), np.concatenate(lab), bx[sel], sel


def post_process_frame(cls_logits, boxes, score_thresh, nms_thresh, pre_maxsize, post_maxsize, normalized=False,
                       raw_score=False, multi_classes=False):
    """One frame of post_processing: the class-agnostic branch (:241-259; label = arg-max class + 1, OUTPUT_RAW_SCORE puts the
    un-normalised maximum out, :254-256) or the MULTI_CLASSES_NMS branch (:214-239; label mapping k -> k + 1)."""
    src = torch.as_tensor(cls_logits, dtype=torch.float32)
    cls = src if normalized else torch.sigmoid(src)
    bx = np.asarray(boxes)
    if multi_classes:
        sc, lab, pb, sel = multi_classes_nms(cls.numpy(), bx, score_thresh, nms_thresh, pre_maxsize, post_maxsize)
        return {"pred_boxes": pb, "pred_scores": sc, "pred_labels": lab + 1, "selected": sel}
    score, label = torch.max(cls, dim=-1)
    sel, sc = class_agnostic_nms(score.numpy(), bx, score_thresh, nms_thresh, pre_maxsize, post_maxsize)
    if raw_score:
        sc = torch.max(src, dim=-1)[0].numpy()[sel]
    return {"pred_boxes": bx[sel], "pred_scores": sc, "pred_labels": (label.numpy()[sel] + 1), "selected": sel}


def recall_record(box_preds, recall_dict, gt_boxes, thresh_list):
    """generate_recall_record without an ROI head (detector3d_template.py:276-318): trailing all-zero gt rows are cut — but
    `while k > 0` never cuts row 0, so a frame without ground truth counts ONE (zero) box —, a gt box is recalled at threshold
    t when its best 3D IoU over the predictions is > t."""
    gt = _f32(gt_boxes)
    if len(recall_dict) == 0:
        recall_dict = {"gt": 0}
        for t in thresh_list:
            recall_dict["roi_%s" % str(t)] = 0
            recall_dict["rcnn_%s" % str(t)] = 0
    k = gt.shape[0] - 1
    while k > 0 and gt[k].sum() == 0:
        k -= 1
    gt = gt[:k + 1]
    if gt.shape[0] > 0:
        bp = _f32(box_preds)
        if bp.shape[0] > 0:
            best = boxes_iou3d(bp[:, :7], gt[:, :7]).max(axis=0)
            for t in thresh_list:
                recall_dict["rcnn_%s" % str(t)] += int((best > np.float32(t)).sum())
        recall_dict["gt"] += gt.shape[0]
    return recall_dict


def post_processing(batch_cls_preds, batch_box_preds, gt_boxes, score_thresh, nms_thresh, pre_maxsize, post_maxsize,
                    thresh_list, **kw):
    """post_processing over a batch (detector3d_template.py:168-274): per-frame records + the recall counters."""
    preds, recall = [], {}
    for b in range(len(batch_cls_preds)):
        rec = post_process_frame(batch_cls_preds[b], batch_box_preds[b], score_thresh, nms_thresh, pre_maxsize, post_maxsize, **kw)
        if gt_boxes is not None:
            recall = recall_record(rec["pred_boxes"], recall, gt_boxes[b], thresh_list)
        preds.append(rec)
    return preds, recall


# ----------------------------------------------------------------------------------------
# whole forward path a1 -> a8 for one batch (CPU).  `params` uses the detector's state-dict names
# (vfe.*, map_to_bev_module.memory.weight, backbone_2d.*, dense_head.*).
# ----------------------------------------------------------------------------------------
def _sub(params, prefix):
    n = len(prefix)
    return {k[n:]: torch.as_tensor(v, dtype=torch.float32) for k, v in params.items() if k.startswith(prefix)}


def forward_frames(frames, params, cfg, stages=None, timings=None):
    """frames: list of (N_i, 4) float32 arrays.  cfg: dict with point_cloud_range, voxel_size, max_points, max_voxels,
    k, layer_nums, layer_strides, sfm_layer_nums, upsample_strides, anchor_sizes, anchor_rotations,
    anchor_bottom_heights, feature_map_stride, dir_offset, dir_limit_offset, num_dir_bins, score_thresh, nms_thresh,
    nms_pre, nms_post.  Returns (pred_dicts, intermediates)."""
    import time
    t0 = time.perf_counter()

    def lap(name):
        nonlocal t0
        if timings is not None:
            t1 = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (t1 - t0)
            t0 = t1

    rng, vs = cfg["point_cloud_range"], cfg["voxel_size"]
    grid = grid_size_of(rng, vs)
    nx, ny = int(grid[0]), int(grid[1])
    vox, coords, num = [], [], []
    for b, f in enumerate(frames):
        v, c, n = voxelize(f, vs, rng, cfg["max_points"], cfg["max_voxels"])
        vox.append(v); num.append(n)
        coords.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], axis=1))
    vox, coords, num = np.concatenate(vox), np.concatenate(coords), np.concatenate(num)
    lap("voxelize")
    inter = {"voxels": vox, "voxel_coords": coords, "voxel_num_points": num}
    pf, sf, _ = pillar_vfe_scale(vox, num.astype(np.float32), coords.astype(np.float32), _sub(params, "vfe."), vs, rng)
    lap("vfe")
    inter.update(pillar_features=pf, pillar_scale_features=sf)
    mem, _, _ = memory_readout_eval(pf, params["map_to_bev_module.memory.weight"], cfg["k"])
    sp, sc = scatter_eval(pf, mem, sf, coords.astype(np.float32), len(frames), nx, ny)
    lap("memory_scatter")
    inter.update(memory_features=mem, spatial_features=sp, spatial_scale_features=sc)
    f2d = bev_backbone_eval(sp, sc, _sub(params, "backbone_2d."), cfg["layer_nums"], cfg["layer_strides"],
                            cfg["sfm_layer_nums"], cfg["upsample_strides"])
    lap("backbone")
    inter["spatial_features_2d"] = f2d
    cls, box, dirp = head_forward(f2d, _sub(params, "dense_head."))
    stride = cfg["feature_map_stride"]
    # one anchor block per class config, concatenated along the size axis (anchor_head_template.py:296 cats dim=-3)
    acfgs = cfg.get("anchor_configs") or [dict(anchor_sizes=cfg["anchor_sizes"], anchor_rotations=cfg["anchor_rotations"],
                                                 anchor_bottom_heights=cfg["anchor_bottom_heights"])]
    anchors = torch.cat([generate_anchors(rng, (nx // stride, ny // stride), a["anchor_sizes"], a["anchor_rotations"],
                                          a["anchor_bottom_heights"]) for a in acfgs], dim=-3)
    bc, bb = generate_predicted_boxes(cls, box, dirp, anchors, cfg["dir_offset"], cfg["dir_limit_offset"], cfg["num_dir_bins"])
    lap("head_decode")
    inter.update(batch_cls_preds=bc, batch_box_preds=bb)
    preds = [post_process_frame(bc[b].numpy(), bb[b].numpy(), cfg["score_thresh"], cfg["nms_thresh"], cfg["nms_pre"],
                                cfg["nms_post"]) for b in range(len(frames))]
    lap("post")
    return preds, inter


def cfg_from_model_cfg(cfg):
    """Flatten an hvpr yaml (hvpr_amd.config AttrDict) into the dict forward_frames takes."""
    dp = [p for p in cfg.DATA_CONFIG.DATA_PROCESSOR if p.NAME == "transform_points_to_voxels"][0]
    m = cfg.MODEL
    ag = m.DENSE_HEAD.ANCHOR_GENERATOR_CONFIG[0]
    return dict(point_cloud_range=list(cfg.DATA_CONFIG.POINT_CLOUD_RANGE), voxel_size=list(dp.VOXEL_SIZE),
                max_points=dp.MAX_POINTS_PER_VOXEL, max_voxels=dp.MAX_NUMBER_OF_VOXELS["test"], k=m.MAP_TO_BEV.NUM_K,
                layer_nums=list(m.BACKBONE_2D.LAYER_NUMS), layer_strides=list(m.BACKBONE_2D.LAYER_STRIDES),
                sfm_layer_nums=list(m.BACKBONE_2D.SFM_LAYER_NUMS), upsample_strides=list(m.BACKBONE_2D.UPSAMPLE_STRIDES),
                anchor_sizes=ag["anchor_sizes"], anchor_rotations=ag["anchor_rotations"],
                anchor_bottom_heights=ag["anchor_bottom_heights"], feature_map_stride=ag["feature_map_stride"],
                anchor_configs=[dict(anchor_sizes=a["anchor_sizes"], anchor_rotations=a["anchor_rotations"],
                                     anchor_bottom_heights=a["anchor_bottom_heights"]) for a in m.DENSE_HEAD.ANCHOR_GENERATOR_CONFIG],
                dir_offset=m.DENSE_HEAD.DIR_OFFSET, dir_limit_offset=m.DENSE_HEAD.DIR_LIMIT_OFFSET,
                num_dir_bins=m.DENSE_HEAD.NUM_DIR_BINS, score_thresh=m.POST_PROCESSING.SCORE_THRESH,
                nms_thresh=m.POST_PROCESSING.NMS_CONFIG.NMS_THRESH, nms_pre=m.POST_PROCESSING.NMS_CONFIG.NMS_PRE_MAXSIZE,
                nms_post=m.POST_PROCESSING.NMS_CONFIG.NMS_POST_MAXSIZE)


# ----------------------------------------------------------------------------------------
# a9 (training)  PointNet++ index ops — pcdet/ops/pointnet2/pointnet2_batch (ABSENT from the reference,
# setup.py:94-109; restated from SURVEY.md Appendix B.4).  PARITY UNPINNED.  fp32 distances computed as
# (dx*dx + dy*dy) + dz*dz; ties resolve to the lowest index (the build's defined rule).
# NOTE on the tie rule: upstream's furthest_point_sampling kernel takes a strided per-thread arg-max followed by a tree reduction
# that keeps the FIRST operand on equal distances — on exact ties its winner depends on the block size and need not be the lowest
# index.  Exact ties do occur: sample_points pads short clouds with DUPLICATES of existing points (data_processor.py:100-104).  A
# duplicate has distance 0 to the set as soon as its twin is picked, so a tie can only move the pick between two copies of the same
# coordinates: the sampled COORDINATES, hence every feature, are the same under either rule; only the index bookkeeping may differ.
# This rule is therefore the build's own, unpinnable (source absent), and harmless to features.
# ----------------------------------------------------------------------------------------
def _d2(a, b):
    d = (a - b).astype(np.float32)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32) + d[..., 2] * d[..., 2]


def furthest_point_sample(xyz, npoint):
    """xyz (B,N,3) -> idx (B,npoint) int32; first pick index 0, running min distance initialised to 1e10."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    out = np.zeros((B, npoint), np.int32)
    for b in range(B):
        md = np.full((N,), 1e10, np.float32)
        last = 0
        for j in range(1, npoint):
            md = np.minimum(md, _d2(xyz[b], xyz[b, last]))
            last = int(np.argmax(md))          # numpy argmax returns the first (lowest-index) maximum
            out[b, j] = last
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, M, _ = new_xyz.shape
    r2 = np.float32(radius) * np.float32(radius)
    out = np.zeros((B, M, nsample), np.int32)
    for b in range(B):
        for m in range(M):
            hits = np.nonzero(_d2(xyz[b], new_xyz[b, m]) < r2)[0][:nsample]
            if hits.size:
                out[b, m, :] = hits[0]
                out[b, m, :hits.size] = hits
    return out


def three_nn(unknown, known):
    unknown, known = _f32(unknown), _f32(known)
    B, n, _ = unknown.shape
    dist = np.zeros((B, n, 3), np.float32)
    idx = np.zeros((B, n, 3), np.int32)
    for b in range(B):
        for i in range(n):
            d = _d2(known[b], unknown[b, i])
            order = np.argsort(d, kind="stable")[:3]
            idx[b, i] = order
            dist[b, i] = np.sqrt(d[order])
    return dist, idx


# ----------------------------------------------------------------------------------------
# f2  KITTI point pre-processing — numpy restatement, pinned by fixture G11 to the reference's
# DataProcessor / Calibration / get_fov_flag run in the build container.
# ----------------------------------------------------------------------------------------
def mask_points_by_range(points, limit_range):
    """pcdet/utils/common_utils.py:59-62: x and y only, both ends inclusive."""
    p = _f32(points)
    r = _f32(limit_range)
    return (p[:, 0] >= r[0]) & (p[:, 0] <= r[3]) & (p[:, 1] >= r[1]) & (p[:, 1] <= r[4])


def _dot4(xyz, m):
    """[x,y,z,1] @ m (4,3) in fp32, left to right, no FMA (the order the HIP kernel uses)."""
    x, y, z = xyz[:, 0:1], xyz[:, 1:2], xyz[:, 2:3]
    return (((x * m[0] + y * m[1]).astype(np.float32) + z * m[2]).astype(np.float32) + m[3]).astype(np.float32)


def fov_flag(points_xyz, V2C, R0, P2, img_shape):
    """Calibration.lidar_to_rect + rect_to_img (pcdet/utils/calibration_kitti.py:65-84) + KittiDataset.get_fov_flag
    (pcdet/datasets/kitti/kitti_dataset.py:100-116).  All float32."""
    A = np.dot(_f32(V2C).T, _f32(R0).T).astype(np.float32)      # (4,3), formed exactly as calibration_kitti.py:71 does
    Pt = _f32(P2).T                                              # (4,3)
    rect = _dot4(_f32(points_xyz), A)
    hom = _dot4(rect, Pt)
    u, v = hom[:, 0] / rect[:, 2], hom[:, 1] / rect[:, 2]
    depth = hom[:, 2] - Pt[3, 2]
    return (u >= 0) & (u < img_shape[1]) & (v >= 0) & (v < img_shape[0]) & (depth >= 0)


def near_flag(points, thresh=40.0):
    p = _f32(points)
    return np.sqrt(((p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]).astype(np.float32) + p[:, 2] * p[:, 2]).astype(np.float32)) < np.float32(thresh)


def sample_points_choice(near, num_points, rng):
    """The index selection of DataProcessor.sample_points (pcdet/datasets/processor/data_processor.py:77-108) with the
    same sequence of RNG calls (`rng` offers choice / shuffle like numpy.random), including the redundant first draw."""
    n = len(near)
    if num_points == -1:
        return np.arange(n)
    if num_points < n:
        far_idx = np.where(near == 0)[0]
        near_idx = np.where(near == 1)[0]
        picked = rng.choice(near_idx, num_points - len(far_idx), replace=False)      # drawn unconditionally (:93)
        if num_points > len(far_idx):
            picked = rng.choice(near_idx, num_points - len(far_idx), replace=False)
            choice = np.concatenate((picked, far_idx), axis=0) if len(far_idx) > 0 else picked
        else:
            choice = rng.choice(np.arange(0, n, dtype=np.int32), num_points, replace=False)
        rng.shuffle(choice)
    else:
        choice = np.arange(0, n, dtype=np.int32)
        if num_points > n:
            extra = rng.choice(choice, num_points - n, replace=False)
            choice = np.concatenate((choice, extra), axis=0)
        rng.shuffle(choice)
    return choice
