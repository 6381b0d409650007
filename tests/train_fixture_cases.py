"""Shared bodies of the training-row fixture checks (G8 target assigner + losses, G9 Adam-onecycle, G10 train-branch memory +
get_score): run on CPU by tests/test_train_host_logic.py and on cuda:0 by tests/test_gpu_train_fixtures.py.  The fixtures come
from the reference's own classes (tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

import torch_forms
from detparams import det_tensor
from hvpr_amd import anchor_head, map_to_bev, optim
from hvpr_amd.config import AttrDict


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _head_cfg():
    return AttrDict(
        CLASS_AGNOSTIC=False, USE_DIRECTION_CLASSIFIER=True, DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0.0, NUM_DIR_BINS=2,
        ANCHOR_GENERATOR_CONFIG=[dict(class_name="Car", anchor_sizes=[[3.9, 1.6, 1.56]], anchor_rotations=[0, 1.57],
                                      anchor_bottom_heights=[-1.78], align_center=False, feature_map_stride=1,
                                      matched_threshold=0.6, unmatched_threshold=0.45)],
        TARGET_ASSIGNER_CONFIG=dict(NAME="AxisAlignedTargetAssigner", POS_FRACTION=-1.0, SAMPLE_SIZE=512,
                                    NORM_BY_NUM_EXAMPLES=False, MATCH_HEIGHT=False, BOX_CODER="ResidualCoder"),
        LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, mem_weight=1.0, code_weights=[1.0] * 7)))


def _np(t):
    return t.detach().cpu().numpy()


def _head(z, dev):
    head = anchor_head.AnchorHeadSingle(model_cfg=_head_cfg(), input_channels=24, num_class=1, class_names=["Car"],
                                        grid_size=np.array([int(z["nx"]), int(z["ny"]), 1]), point_cloud_range=z["point_cloud_range"])
    head.anchors = [a.to(dev) for a in head.anchors]
    head = head.to(dev)
    if torch.device(dev).type == "cpu":         # the product's head, assigner and losses run on the library's kernels and raise on CPU
        torch_forms.patch(head)                 # tensors: the host suite checks the torch forms of tests/ against the same fixtures
    return head


def run_g8(golden_dir, dev="cpu", rtol=1e-5):
    z = _load(golden_dir, "g8_assigner_losses.npz")
    head = _head(z, dev)
    head.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")})
    head.train()
    t = lambda k: torch.from_numpy(z[k]).to(dev)
    head({"spatial_features_2d": t("spatial_features_2d"), "spatial_features_point_2d": t("spatial_features_point_2d"),
          "point_positive_features": t("pos_point"), "memory_positive_features": t("pos_memory"),
          "memory_items": torch.zeros(1, device=dev), "gt_boxes": t("gt_boxes"), "batch_size": 2})
    fr = head.forward_ret_dict
    assert fr["box_cls_labels"].device.type == torch.device(dev).type      # the assigner ran where the anchors live
    ref_labels = z["target.box_cls_labels"]
    np.testing.assert_array_equal(_np(fr["box_cls_labels"]), ref_labels)                    # integer labels: exact
    assert (ref_labels > 0).sum() >= 4 and (ref_labels == -1).sum() > 0                    # the case has positives and don't-cares
    np.testing.assert_array_equal(_np(fr["reg_weights"]), z["target.reg_weights"])
    np.testing.assert_allclose(_np(fr["box_reg_targets"]), z["target.box_reg_targets"], rtol=1e-5, atol=1e-6)
    rpn, rpn_pt, mem, tb, _ = head.get_loss()
    np.testing.assert_allclose(rpn.item(), float(z["rpn_loss"]), rtol=rtol)
    np.testing.assert_allclose(rpn_pt.item(), float(z["rpn_loss_point"]), rtol=rtol)
    np.testing.assert_allclose(mem.item(), float(z["mem_loss"]), rtol=rtol)
    for k in ("rpn_loss_cls", "rpn_loss_loc", "rpn_loss_dir", "rpn_loss_cls_pt", "rpn_loss_loc_pt", "rpn_loss_dir_pt", "mem_loss"):
        np.testing.assert_allclose(tb[k].item(), float(z["tb." + k]), rtol=rtol, err_msg=k)


def run_g8_no_gt(golden_dir, dev="cpu"):
    z = _load(golden_dir, "g8_assigner_losses.npz")
    head = _head(z, dev)
    t = head.assign_targets(torch.zeros(1, 3, 8, device=dev))
    assert (t["box_cls_labels"] == 0).all() and (t["reg_weights"] == 0).all() and (t["box_reg_targets"] == 0).all()


def run_g8_batch_passes(golden_dir, dev="cpu"):
    """Frames are independent in the assigner (the torch form works through the batch four frames at a time, the kernels take a frame
    per grid row): a batch of 6 (the two fixture frames three times over, the middle pair with a ground truth removed) must give,
    frame by frame, what batches of 2 give — which G8 pins."""
    z = _load(golden_dir, "g8_assigner_losses.npz")
    head = _head(z, dev)
    gt = torch.from_numpy(z["gt_boxes"]).to(dev)
    gt2 = gt.clone(); gt2[:, 0] = 0
    six = head.assign_targets(torch.cat([gt, gt2, gt], dim=0))
    for lo, g in ((0, gt), (2, gt2), (4, gt)):
        two = head.assign_targets(g)
        for k in ("box_cls_labels", "box_reg_targets", "reg_weights"):
            assert torch.equal(six[k][lo:lo + 2], two[k]), (lo, k)
    np.testing.assert_array_equal(_np(six["box_cls_labels"][:2]), z["target.box_cls_labels"])


def run_g9(golden_dir, dev="cpu", rtol=2e-5, make_optimizer=None):
    """100 schedule steps + 3 optimiser steps; make_optimizer(net, wd) lets the GPU test run the same case through the
    fused flat-buffer kernel."""
    z = _load(golden_dir, "g9_onecycle.npz")
    net = torch.nn.Sequential(torch.nn.Linear(6, 5, bias=False), torch.nn.BatchNorm1d(5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    net.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init.")}, strict=False)
    net = net.to(dev)
    opt = (make_optimizer or (lambda n, wd: optim.AdamOneCycle(n, wd=wd)))(net, 0.01)
    sched = optim.OneCycle(opt, 100, 0.003, [0.95, 0.85], 10, 0.4)
    xs = torch.from_numpy(z["x"]).to(dev)
    lrs, moms = [], []
    for it in range(100):
        sched.step(it)
        lrs.append(opt.lr); moms.append(opt.mom)
        if it < 3:
            net.train(); opt.zero_grad()
            net(xs[it]).pow(2).mean().backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 10)
            opt.step()
            for k, v in net.state_dict().items():
                if "num_batches" not in k:
                    np.testing.assert_allclose(_np(v), z[f"after{it}.{k}"], rtol=rtol, atol=1e-7, err_msg=f"step {it} {k}")
    np.testing.assert_allclose(lrs, z["lr"], rtol=1e-12)
    np.testing.assert_allclose(moms, z["mom"], rtol=1e-12)


def run_g10(golden_dir, dev="cpu", rtol=1e-5):
    z = _load(golden_dir, "g10_train_memory.npz")
    cfg = AttrDict(NUM_BEV_FEATURES=128, NUM_COORD_POINTS=3, NUM_PT_FEATURES=64, NUM_SCALE_FEATURES=32, NUM_K=20, NUM_M=2000, SHRINK_TH=0.0025)
    m = map_to_bev.PointPillarScatter_Agg_Memory_1_scale(cfg, np.array([12, 10, 1]))
    m.memory.weight.data = torch.from_numpy(det_tensor(str(z["W_name"]), (2000, 64), int(z["W_seed"])))
    m = m.to(dev).train()
    if torch.device(dev).type == "cpu":         # the product raises on CPU tensors: the host suite runs the torch forms of tests/
        torch_forms.patch(m)
    pillars, points = torch.from_numpy(z["pillars"]).to(dev), torch.from_numpy(z["points"]).to(dev)
    agg, positives = m.get_score(points, pillars)
    np.testing.assert_allclose(_np(agg), z["get_score_output"], rtol=rtol, atol=1e-6)
    np.testing.assert_array_equal(_np(positives), z["positives"])          # the same k points per pillar, in the same order
    out = m.memory(pillars, 20, positives)
    np.testing.assert_allclose(_np(out["output"]), z["memory_output"], rtol=10 * rtol, atol=1e-6)
    if "att" in out:
        np.testing.assert_allclose(_np(out["att"].sum(1)), z["memory_att_rowsum"], rtol=rtol)
        nnz = _np((out["att"] > 0).sum(1))
        if torch.device(dev).type == "cpu":
            np.testing.assert_array_equal(nnz, z["memory_att_nnz"])
        else:   # a softmax value within one ulp of the shrink threshold may flip on another device
            assert np.abs(nnz - z["memory_att_nnz"]).max() <= 1
    # gradients flow to the bank through the read-out and to the pillars through the point stream weights only via detach rules
    out["output"].sum().backward()
    assert m.memory.weight.grad is not None and torch.isfinite(m.memory.weight.grad).all()


def _g16_modules(z, dev, dtype):
    cfg = AttrDict(NUM_BEV_FEATURES=128, NUM_COORD_POINTS=3, NUM_PT_FEATURES=64, NUM_SCALE_FEATURES=32, NUM_K=20, NUM_M=2000, SHRINK_TH=0.0025)
    nx, ny = int(z["nx"]), int(z["ny"])
    scat = map_to_bev.PointPillarScatter_Agg_Memory_1_scale(cfg, np.array([nx, ny, 1]))
    scat.memory.weight.data = torch.from_numpy(det_tensor("memory.weight", (2000, 64), int(z["seed"])) * np.float32(z["bank_scale"]))
    head = anchor_head.AnchorHeadSingle(model_cfg=_head_cfg(), input_channels=128, num_class=1, class_names=["Car"],
                                        grid_size=np.array([nx, ny, 1]), point_cloud_range=z["point_cloud_range"])
    head.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")})
    scat, head = scat.to(dev).to(dtype).train(), head.to(dev).to(dtype).train()
    head.anchors = [a.to(dev).to(dtype) for a in head.anchors]
    return scat, head


def run_g16(golden_dir, dev="cpu", dtype=torch.float32, value_rtol=1e-5, grad_tol=1e-6):
    """Fixture G16 — the reference's scatter training branch + memory addressing + head + get_loss, and WHICH tensor receives WHICH
    gradient (every `.detach()` of pointpillar_scatter.py:75-80,140, memory_module.py:56, anchor_head_template.py:268 shows as an
    exactly-zero block).  On the CPU the torch forms of tests/torch_forms.py run (float64: gradients to grad_tol norm-wise); on
    cuda:0 the product's kernels (fp32: tolerance = max(grad_tol, 3 x the distance of the reference's OWN fp32 gradients from its
    float64 ones))."""
    z = _load(golden_dir, "g16_train_branch_gradients.npz")
    scat, head = _g16_modules(z, dev, dtype)
    on_cpu = torch.device(dev).type == "cpu"
    if on_cpu:
        torch_forms.patch(scat)
        torch_forms.patch(head)
    t = lambda k: torch.from_numpy(z[k]).to(dev).to(dtype)
    ins = {k: t(k).requires_grad_(True) for k in ("pillar_features", "point_features", "pillar_scale_features")}
    d = scat({**ins, "voxel_coords": t("voxel_coords"), "point_coords": t("point_coords"), "batch_size": 2})
    canv = {k: d[k] for k in ("spatial_features", "spatial_features_point", "spatial_scale_features")}
    d.update(spatial_features_2d=d["spatial_features"], spatial_features_point_2d=d["spatial_features_point"], gt_boxes=t("gt_boxes"))
    head(d)
    np.testing.assert_array_equal(_np(head.forward_ret_dict["box_cls_labels"]), z["box_cls_labels"])
    rpn, rpn_pt, mem, tb, items = head.get_loss()
    assert items is scat.memory.weight

    # ---- values (fp32 fixture values; the float64 run is closer to them than the fp32 reference's own round-off)
    vt = max(value_rtol, 1e-5)
    for k, v in canv.items():
        ref = z[k]
        assert np.abs(_np(v).astype(np.float64) - ref).max() <= vt * np.abs(ref).max(), k
        occ = slice(0, 64) if k == "spatial_features" else slice(None)   # (a memory row is all zero when no item passes SHRINK_TH)
        assert ((_np(v)[:, occ] != 0) == (ref[:, occ] != 0)).all(), k   # the same cells / channels are occupied
    np.testing.assert_allclose(_np(d["point_positive_features"]), z["point_positive_features"], rtol=0, atol=vt * np.abs(z["point_positive_features"]).max())
    ref_mem = z["memory_positive_features_f64"] if dtype == torch.float64 else z["memory_positive_features"]
    # a softmax value within round-off of SHRINK_TH may pass on one side and not on the other: one item of one row, norm-wise small
    err = np.linalg.norm(_np(d["memory_positive_features"]).astype(np.float64) - ref_mem) / np.linalg.norm(ref_mem)
    assert err < 10 * vt, err
    ref_losses = z["losses_f64"] if dtype == torch.float64 else z["losses"]
    np.testing.assert_allclose([rpn.item(), rpn_pt.item(), mem.item()], ref_losses, rtol=value_rtol if dtype == torch.float32 else 1e-9)
    if dtype == torch.float32:
        for k in ("rpn_loss_cls", "rpn_loss_loc", "rpn_loss_dir", "rpn_loss_cls_pt", "rpn_loss_loc_pt", "rpn_loss_dir_pt", "mem_loss"):
            np.testing.assert_allclose(float(tb[k]), float(z["tb." + k]), rtol=value_rtol, err_msg=k)

    # ---- gradients, loss by loss: who gets one and who does not
    leaves = {"pillar_features": ins["pillar_features"], "point_features": ins["point_features"], "memory.weight": scat.memory.weight}
    report = {}
    for lname, val in (("rpn_loss", rpn), ("rpn_loss_point", rpn_pt), ("mem_loss", mem)):
        g = torch.autograd.grad(val, list(leaves.values()), retain_graph=True, allow_unused=True)
        for (name, leaf), gv in zip(leaves.items(), g):
            key = f"{lname}.{name}"
            if bool(z["zero." + key]):
                assert gv is None or float(gv.abs().max()) == 0.0, f"{key}: the reference detaches here, got a gradient"
                continue
            assert gv is not None, f"{key}: the reference has a gradient here"
            ref = z["grad." + key].astype(np.float64)
            e = np.linalg.norm(_np(gv).astype(np.float64) - ref) / np.linalg.norm(ref)
            tol = grad_tol if dtype == torch.float64 else max(grad_tol, 3 * float(z["ref32_err." + key]))
            report[key] = (e, tol)
            assert e < tol, (key, e, tol)
    total = rpn + rpn_pt + mem + (canv["spatial_scale_features"] * t("cot_scale")).sum()
    hp = dict(head.named_parameters())
    g = torch.autograd.grad(total, [ins["pillar_scale_features"]] + list(hp.values()))
    for key, gv in zip(["pillar_scale_features"] + ["head." + k for k in hp], g):
        ref = z["grad.total." + key].astype(np.float64)
        e = np.linalg.norm(_np(gv).astype(np.float64) - ref) / np.linalg.norm(ref)
        tol = grad_tol if dtype == torch.float64 else max(grad_tol, 3 * float(z["ref32_err.total." + key]))
        report["total." + key] = (e, tol)
        assert e < tol, (key, e, tol)
    return report


def drive_like_g17(model, optimizer, loader, model_func, lr_scheduler, accumulated_iter, grad_norm_clip, tbar, clip=None):
    """A driver loop that makes the calls fixture G17 records from the reference's train_one_epoch (tools/train_utils/train_utils.py:
    25-52), in that order, one iteration per batch of `loader`: scheduler, a read of optimizer.lr, model.train(), zero_grad,
    model_func -> 4-tuple, backward, the EXTERNAL clip over model.parameters(), optimizer.step(), then the progress-bar update with
    disp_dict + loss + lr.  Returns (accumulated_iter, items of the last iteration, [loss per iteration])."""
    clip = clip or torch.nn.utils.clip_grad_norm_
    items, losses = None, []
    for batch in loader:
        lr_scheduler.step(accumulated_iter)
        shown_lr = float(optimizer.lr)
        model.train()
        optimizer.zero_grad()
        loss, tb_dict, disp_dict, items = model_func(model, batch)
        loss.backward()
        clip(model.parameters(), grad_norm_clip)
        optimizer.step()
        accumulated_iter += 1
        disp_dict.update(loss=loss.item(), lr=shown_lr)
        tbar.set_postfix(disp_dict)
        tbar.refresh()
        losses.append(float(disp_dict["loss"]))
    return accumulated_iter, items, losses


def run_g17_protocol(golden_dir):
    """drive_like_g17 on the recording stand-ins of make_golden.g17_stubs makes exactly the calls the reference's loop made."""
    from make_golden import g17_stubs
    z = _load(golden_dir, "g17_train_loop_protocol.npz")
    log = []
    model, opt, sched, bar, model_func, clip, items, loader = g17_stubs(log)
    it, got_items, _ = drive_like_g17(model, opt, loader, model_func, sched, int(z["start_iter"]), int(z["grad_norm_clip"]), bar, clip)
    assert log == [str(c) for c in z["calls"]], "\n".join(f"{a!s:60s} | {b!s}" for a, b in zip(log, z["calls"]))
    assert it == int(z["returned_iter"]) and got_items is items
