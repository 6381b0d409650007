"""GPU: rows a12 + a13 as the library's own kernels (hvpr_assign_targets_f32, hvpr_rpn_losses_f32, hvpr_mse_loss_f32) against the
torch forms of tests/torch_forms.py — which the CPU suite pins to the reference's own AxisAlignedTargetAssigner / loss_utils /
get_loss through fixtures G8 and G16 (tests/test_train_host_logic.py); the kernels themselves also run against G8 and G16 directly
(tests/test_gpu_train_fixtures.py).  Here: full-size anchor sets, three classes, padded / foreign-class / degenerate ground truths,
and the gradients of every loss against torch autograd."""
import copy

import numpy as np
import pytest
import torch

import torch_forms
from hvpr_amd import anchor_head
from hvpr_amd.config import hvpr_3class_cfg, hvpr_car_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _heads(which, nx=296, ny=248):
    cfg = hvpr_car_cfg() if which == "car" else hvpr_3class_cfg()
    rng = np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, np.float32)
    rng[3] = rng[0] + nx * 0.16
    rng[4] = rng[1] + ny * 0.16
    mk = lambda: anchor_head.AnchorHeadSingle(model_cfg=cfg.MODEL.DENSE_HEAD, input_channels=64, num_class=len(cfg.CLASS_NAMES),
                                              class_names=cfg.CLASS_NAMES, grid_size=np.array([nx, ny, 1]), point_cloud_range=rng)
    hip = mk().to(DEV).train()
    ref = copy.deepcopy(hip)
    hip.anchors = [a.to(DEV) for a in hip.anchors]
    ref.anchors = [a.to(DEV) for a in ref.anchors]
    torch_forms.patch(ref)
    return cfg, hip, ref, rng


def _gt(rng_np, B, G, n_class, pcr, seed):
    r = np.random.default_rng(seed)
    sizes = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], np.float32)
    g = np.zeros((B, G, 8), np.float32)
    for b in range(B):
        k = int(r.integers(0, G + 1)) if b else G // 2           # frame 0: half the rows; another frame may have none or all
        cls = r.integers(0, n_class, k)
        g[b, :k, 0] = r.uniform(pcr[0] + 2, pcr[3] - 2, k)
        g[b, :k, 1] = r.uniform(pcr[1] + 2, pcr[4] - 2, k)
        g[b, :k, 2] = r.uniform(-1.2, -0.6, k)
        g[b, :k, 3:6] = sizes[cls] * r.uniform(0.85, 1.15, (k, 3))
        g[b, :k, 6] = r.uniform(-np.pi, np.pi, k)
        g[b, :k, 7] = cls + 1
        if k > 3:
            g[b, 1] = 0.0                                         # an all-zero row INSIDE the valid range: class 0 wraps to the last class
            g[b, 2, 6] = np.pi / 4                                # heading exactly on the axis-snapping border
    return torch.from_numpy(g).to(DEV)


@pytest.mark.parametrize("which,B,G", [("car", 3, 12), ("3class", 4, 30), ("car", 1, 1), ("3class", 2, 0)])
def test_target_assigner_kernels_equal_the_torch_form(which, B, G):
    cfg, hip, ref, pcr = _heads(which)
    if G == 0:
        gt = torch.zeros((B, 1, 8), device=DEV)                   # only padding
    else:
        gt = _gt(None, B, G, len(cfg.CLASS_NAMES), pcr, seed=B * 100 + G)
    got = hip.assign_targets(gt.clone())
    want = ref.assign_targets(gt.clone())
    assert got["box_cls_labels"].dtype == torch.int32
    assert torch.equal(got["box_cls_labels"], want["box_cls_labels"].to(torch.int32))            # labels: exact
    assert torch.equal(got["reg_weights"], want["reg_weights"])
    torch.testing.assert_close(got["box_reg_targets"], want["box_reg_targets"], rtol=1e-5, atol=1e-6)
    assert torch.equal(got["positives_per_frame"].long(), (want["box_cls_labels"] > 0).sum(dim=1))
    if G > 1:
        lab = want["box_cls_labels"]
        assert int((lab > 0).sum()) >= G // 2 and int((lab == -1).sum()) > 0 and int((lab == 0).sum()) > 0


@pytest.mark.parametrize("which", ["car", "3class"])
def test_loss_kernels_values_and_gradients_equal_torch_autograd(which):
    cfg, hip, ref, pcr = _heads(which, nx=96, ny=80)
    B, n_class = 3, len(cfg.CLASS_NAMES)
    gt = _gt(None, B, 10, n_class, pcr, seed=5)
    gen = torch.Generator(device="cpu").manual_seed(7)
    na = hip.num_anchors_per_location
    shapes = {"cls_preds": na * n_class, "box_preds": na * 7, "dir_cls_preds": na * 2}
    preds = {}
    for sfx in ("", "_point"):
        for k, c in shapes.items():
            scale = 3.0 if k == "cls_preds" else (0.4 if k == "box_preds" else 1.5)
            preds[k + sfx] = (torch.randn(B, 80, 96, c, generator=gen) * scale).to(DEV)
    pos_p = torch.randn(500, 64, generator=gen).to(DEV)
    pos_m = torch.randn(500, 64, generator=gen).to(DEV)
    res = {}
    for name, head in (("hip", hip), ("ref", ref)):
        leaves = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
        pm = pos_m.clone().requires_grad_(True)
        fr = head.forward_ret_dict
        fr.clear()
        fr.update(leaves)
        fr.update(pos_point_feas=pos_p, pos_memory_feas=pm, memory_items=None)
        fr.update(head.assign_targets(gt.clone()))
        rpn, rpn_pt, mem, tb, _ = head.get_loss()
        # NaN targets are ignored by the smooth-L1 (loss_utils.py:117-119): poison one positive's target and run again below
        cot = torch.tensor([0.7, 1.3, 2.1], device=DEV)
        (rpn * cot[0] + rpn_pt * cot[1] + mem * cot[2]).backward()
        res[name] = (dict(rpn=rpn.detach(), rpn_pt=rpn_pt.detach(), mem=mem.detach(), **tb), {k: v.grad for k, v in leaves.items()}, pm.grad)
    (vh, gh, mh), (vr, gr, mr) = res["hip"], res["ref"]
    for k in vr:
        torch.testing.assert_close(vh[k], vr[k], rtol=2e-5, atol=1e-7, msg=k)
    for k in gr:
        err = float((gh[k] - gr[k]).norm() / gr[k].norm())
        assert err < 2e-5, (k, err)
        # element-wise too: no gradient where the reference has none (don't-care anchors, negatives of the box / direction losses)
        assert bool(((gr[k] == 0) == (gh[k] == 0)).float().mean() > 0.9999), k
    assert float((mh - mr).norm() / mr.norm()) < 1e-6


def test_assigner_and_losses_refuse_cpu_tensors():
    cfg = hvpr_car_cfg()
    head = anchor_head.AnchorHeadSingle(model_cfg=cfg.MODEL.DENSE_HEAD, input_channels=64, num_class=1, class_names=cfg.CLASS_NAMES,
                                        grid_size=np.array([16, 16, 1]), point_cloud_range=np.array([0, -1.28, -3, 2.56, 1.28, 1], np.float32))
    head.anchors = [a.cpu() for a in head.anchors]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        head.assign_targets(torch.zeros(1, 2, 8))
    from hvpr_amd import losses
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        losses.memory_loss(torch.zeros(4, 64), torch.zeros(4, 64), 1.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        losses.rpn_losses(torch.zeros(1, 4, 1), torch.zeros(1, 4, 7), None, torch.zeros(1, 4, dtype=torch.int32), torch.zeros(1, 4, 7),
                          torch.zeros(4), torch.zeros(1, dtype=torch.int32), 1, {"code_weights": [1.0] * 7, "cls_weight": 1.0,
                                                                                 "loc_weight": 2.0, "dir_weight": 0.2}, 0.78539, 2)
