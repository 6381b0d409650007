"""TEST INFRASTRUCTURE ONLY — plain-torch forms of the training-mode modules of hvpr_amd.

The product modules (hvpr_amd/bev_backbone.py, vfe.py, map_to_bev.py, anchor_head.py, pointnet2.py) run their training forward
on the library's HIP kernels and raise on anything else; they hold no torch fallback.  The functions here restate the same
forwards with torch ops + torch autograd on the modules' OWN parameters and buffers (same state-dict, same running-statistics
updates), following the reference line by line:

  backbone_train   BaseBEVBackbone_Scale.forward, training branch      pcdet/models/backbones_2d/base_bev_backbone.py:228-279
  vfe_train        PillarVFE_Scale.forward                              pcdet/models/backbones_3d/vfe/pillar_vfe.py:184-221
  memory_train     MemoryUnit_Agg.forward, training branch              pcdet/models/backbones_2d/map_to_bev/memory_module.py:31-59
  topk_points      the top-k of PointPillarScatter...get_score          .../map_to_bev/pointpillar_scatter.py:67-83
  scatter_train    PointPillarScatter_Agg_Memory_1_scale.forward, training branch   .../pointpillar_scatter.py:87-167
  head_train       AnchorHeadSingle.forward, training branch            pcdet/models/dense_heads/anchor_head_single.py:41-108
  sa_forward / fp_forward   PointnetSAModuleMSG / PointnetFPModule with torch shared MLPs (pointnet2_backbone.py:27-47)

They serve (a) as the comparator of the GPU parity tests (whole-module and whole-train-step: tests/test_gpu_train_step_parity.py,
test_gpu_conv_train.py, test_gpu_train_ops.py) and (b) to run the device-agnostic host logic (target assigner, losses, DDP
gradient averaging) on CPU in the `-m "not gpu"` suite.  `patched(model)` swaps them in on a module tree and restores afterwards."""
import contextlib
import types

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ a11 two-stream backbone
def backbone_train(self, data_dict):
    x, xp, y = data_dict["spatial_features"], data_dict["spatial_features_point"], data_dict["spatial_scale_features"]

    def attention(t, w):                       # spatial_attention.py:57-63
        sp = self.attention.spatial
        pooled = torch.cat((w.max(dim=1, keepdim=True)[0], w.mean(dim=1, keepdim=True)), dim=1)
        return torch.sigmoid(sp.norm(sp.conv(pooled))) * t
    ups, ups_p = [], []
    for i in range(len(self.blocks)):
        x, xp, y = self.blocks[i](x), self.blocks[i](xp), self.scale_layers[i](y)
        xa, xpa = x, xp
        for _ in range(self.sfm_layer_nums[i]):
            xa = attention(self.sfmblocks_down[i](xa), y) + xa
            xpa = attention(self.sfmblocks_down[i](xpa), y) + xpa
        ups.append(self.deblocks[i](xa))
        ups_p.append(self.deblocks[i](xpa))
    data_dict["spatial_features_2d"] = torch.cat(ups, dim=1)
    data_dict["spatial_features_point_2d"] = torch.cat(ups_p, dim=1)
    return data_dict


# ------------------------------------------------------------------------------------------------ a2 pillar VFE (training)
def vfe_train(self, batch_dict, voxels, num, coords):
    md = batch_dict.get("voxel_count_device")
    if md is not None:
        m = int(md.item())
        voxels, num, coords = voxels[:m], num[:m], coords[:m]
        batch_dict["voxels"], batch_dict["voxel_num_points"], batch_dict["voxel_coords"] = voxels, num, coords
        batch_dict["voxel_count_device"] = None
    n = num.to(voxels.dtype)
    c = coords.to(voxels.dtype)
    M, P, _ = voxels.shape
    xyz = voxels[:, :, :3]
    mean = xyz.sum(dim=1, keepdim=True) / n.view(-1, 1, 1)
    centre = c[:, [3, 2, 1]] * voxels.new_tensor(self.voxel_size) + voxels.new_tensor(self.offsets)
    mask = (torch.arange(P, device=voxels.device).view(1, -1) < num.view(-1, 1)).unsqueeze(-1).to(voxels.dtype)
    x = torch.cat([voxels, xyz - mean, xyz - centre.unsqueeze(1)], dim=-1) * mask
    for layer in self.pfn_layers:              # pillar_vfe.py:14-27: BatchNorm1d over all M*P slots, padded ones included
        y = layer.linear(x)
        y = torch.relu(layer.norm(y.permute(0, 2, 1)).permute(0, 2, 1))
        ymax = y.max(dim=1, keepdim=True)[0]
        x = ymax if layer.last_vfe else torch.cat([y, ymax.expand(-1, P, -1)], dim=2)
    s = torch.cat([n.unsqueeze(1), torch.norm(mean, 2, 2), mean.squeeze(1)], dim=-1)
    for seq in self.pfn_scale_layers:          # pillar_vfe.py:213-216
        s = seq(s)
    batch_dict["pillar_features"] = x.reshape(M, -1)
    batch_dict["pillar_scale_features"] = s
    batch_dict["pillar_mask"] = mask
    return batch_dict


# ------------------------------------------------------------------------------------------------ a10 memory + get_score
def hard_shrink_relu(x, lambd=0.0, epsilon=1e-12):
    """memory_module.py:85-87."""
    return (F.relu(x - lambd) * x) / (torch.abs(x - lambd) + epsilon)


def memory_train(self, pillars, k, positives):
    nv, _, d = positives.shape
    att = torch.softmax(F.linear(positives.reshape(-1, d), self.weight), dim=1)          # (nv*k, items), materialised
    if self.shrink_thres > 0:
        att = hard_shrink_relu(att, self.shrink_thres)
        att = F.normalize(att, p=1, dim=1)
    mem = F.linear(att, self.weight.t()).reshape(nv, k, d)
    agg = torch.softmax((mem * pillars.unsqueeze(1)).sum(dim=2), dim=1)
    return {"output": (agg.detach().unsqueeze(2) * mem).sum(dim=1), "att": att}


def topk_points(self, pillars, points):
    """pointpillar_scatter.py:70-73: indices of the top-k over points of softmax(points @ pillars^T, dim=0) = of the raw logits."""
    return torch.topk(pillars @ points.t(), self.k, dim=1)[1]


def get_score(self, points, pillars):
    with torch.no_grad():
        idx = topk_points(self, pillars.detach(), points.detach())
    positives = points[idx]
    w = torch.softmax(torch.bmm(pillars.unsqueeze(1), positives.transpose(1, 2)).squeeze(1), dim=1)
    return (w.detach().unsqueeze(2) * positives).sum(dim=1), positives


def scatter_train(self, batch_dict):
    """pointpillar_scatter.py:87-167 as written there: a loop over the frames with boolean masks, get_score, the memory with the
    positives as its second input (T1), dense canvases filled by column assignment; the pillar half of the memory-fed canvas is
    detached (:140)."""
    pf, sf, coords = batch_dict["pillar_features"], batch_dict["pillar_scale_features"], batch_dict["voxel_coords"]
    point_f, point_c = batch_dict["point_features"], batch_dict["point_coords"]
    B = int(batch_dict["batch_size"]) if "batch_size" in batch_dict else int(coords[:, 0].max().item()) + 1
    cells = self.nx * self.ny
    sp, spp, sc, pos_point, pos_mem = [], [], [], [], []
    for b in range(B):
        vm, pm = coords[:, 0] == b, point_c[:, 0] == b
        c = coords[vm]
        idx = (c[:, 1] + c[:, 2] * self.nx + c[:, 3]).long()
        pillars = pf[vm]
        agg, positives = get_score(self, point_f[pm], pillars)
        mem = self.memory(pillars, self.k, positives)["output"]
        canv = [pf.new_zeros(self.num_bev_features, cells), pf.new_zeros(self.num_bev_features, cells), sf.new_zeros(sf.shape[1], cells)]
        canv[0][:, idx] = torch.cat((pillars.detach(), mem), dim=1).t()
        canv[1][:, idx] = torch.cat((pillars, agg), dim=1).t()
        canv[2][:, idx] = sf[vm].t()
        sp.append(canv[0]); spp.append(canv[1]); sc.append(canv[2])
        pos_point.append(agg); pos_mem.append(mem)
    batch_dict["spatial_features"] = torch.stack(sp, 0).view(B, -1, self.ny, self.nx)
    batch_dict["spatial_features_point"] = torch.stack(spp, 0).view(B, -1, self.ny, self.nx)
    batch_dict["spatial_scale_features"] = torch.stack(sc, 0).view(B, -1, self.ny, self.nx)
    batch_dict["point_positive_features"] = torch.cat(pos_point, 0)
    batch_dict["memory_positive_features"] = torch.cat(pos_mem, 0)
    batch_dict["memory_items"] = self.memory.weight
    return batch_dict


# ------------------------------------------------------------------------------------------------ a6 head (training)
def head_train(self, data_dict):
    fr = self.forward_ret_dict
    heads = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
    for key, suffix in (("spatial_features_2d", ""), ("spatial_features_point_2d", "_point")):
        parts = [h(data_dict[key]).permute(0, 2, 3, 1).contiguous() for h in heads]
        fr["cls_preds" + suffix], fr["box_preds" + suffix] = parts[0], parts[1]
        if self.conv_dir_cls is not None:
            fr["dir_cls_preds" + suffix] = parts[2]
    return self._finish_train(data_dict)


# ------------------------------------------------------------------------------------------------ a9 point-stream modules
def sa_forward_rows(self, xyz, features=None, pre=None):
    """PointnetSAModuleMSG (pointnet2_backbone.py:27-34) with torch gathers, torch 1x1 Conv2d + BatchNorm2d + ReLU and torch max;
    the index tensors (FPS, ball query) come from `pre` or the module's own index kernels."""
    idx, new_xyz, balls = (pre if pre is not None else self.indices(xyz))[:3]
    B = xyz.shape[0]
    ar = torch.arange(B, device=xyz.device)[:, None, None]
    outs = []
    for mlp, bidx in zip(self.mlps, balls):
        bi = bidx.long()                                               # (B, npoint, nsample)
        g = xyz[ar, bi] - new_xyz.unsqueeze(2)                         # (B, npoint, nsample, 3): xyz channels first
        if features is not None:
            g = torch.cat([g, features[ar, bi]], dim=-1)
        f = mlp(g.permute(0, 3, 1, 2))                                 # (B, C', npoint, nsample)
        outs.append(f.max(dim=-1)[0])
    return new_xyz, torch.cat(outs, dim=1).transpose(1, 2).contiguous()


def fp_forward_rows(self, unknown, known, unknow_feats, known_feats, pre=None):
    """PointnetFPModule (pointnet2_backbone.py:40-47, 86-89) with torch gathers and the torch shared MLP."""
    from hvpr_amd import pointnet2
    dist, idx = (pre if pre is not None else pointnet2.three_nn(unknown, known))[:2]
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(dim=2, keepdim=True)
    B = idx.shape[0]
    ar = torch.arange(B, device=idx.device)[:, None, None]
    g = known_feats[ar, idx.long()]                                    # (B, n, 3, C1)
    f = (g[:, :, 0] * w[:, :, 0:1] + g[:, :, 1] * w[:, :, 1:2]) + g[:, :, 2] * w[:, :, 2:3]
    if unknow_feats is not None:
        f = torch.cat([f, unknow_feats], dim=-1)
    return self.mlp(f.transpose(1, 2).unsqueeze(-1)).squeeze(-1).transpose(1, 2).contiguous()


# ------------------------------------------------------------------------------------------------ swapping them in
def _table():
    from hvpr_amd import anchor_head, bev_backbone, map_to_bev, pointnet2, vfe
    return {
        pointnet2.PointnetSAModuleMSG: {"forward_rows": sa_forward_rows},
        pointnet2.PointnetFPModule: {"forward_rows": fp_forward_rows},
        bev_backbone.BaseBEVBackbone_Scale: {"_forward_train": backbone_train},
        vfe.PillarVFE_Scale: {"_forward_train": vfe_train},
        map_to_bev.MemoryUnit_Agg: {"_forward_train": memory_train},
        map_to_bev.PointPillarScatter_Agg_Memory_1_scale: {"get_score": get_score, "_topk_points": topk_points,
                                                           "_forward_train": scatter_train},
        anchor_head.AnchorHeadSingle: {"_forward_train": head_train},
    }


def patch(root, only=None):
    """Swap the torch forms in on every matching module under `root` (instance attributes; the classes stay untouched).
    only: optional iterable of class names to restrict to.  Returns the list of (module, attribute) that were set."""
    done = []
    table = _table()
    for m in root.modules():
        for cls, fns in table.items():
            if type(m) is cls and (only is None or cls.__name__ in only):
                for name, fn in fns.items():
                    setattr(m, name, types.MethodType(fn, m))
                    done.append((m, name))
    return done


def unpatch(done):
    for m, name in done:
        if name in m.__dict__:
            delattr(m, name)


@contextlib.contextmanager
def patched(root, only=None):
    done = patch(root, only)
    try:
        yield root
    finally:
        unpatch(done)
