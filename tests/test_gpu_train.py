"""GPU: the training rows a9-a15 — PointNet++ index ops bit-exact vs the oracle, train-mode VFE vs the oracle, scatter
autograd, and whole training steps (forward + backward + Adam-onecycle) of the hvpr_car detector."""
import numpy as np
import pytest
import torch

from hvpr_amd import detector, optim, pointnet2, synthetic, synthetic_weights
from hvpr_amd.config import hvpr_car_cfg
from oracle import hvpr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cloud(seed, B, N):
    return np.stack([synthetic.hvpr_frame(seed + b, num_points=N)[:, :3] for b in range(B)])


def test_fps_ball_query_three_nn_bit_exact():
    xyz = _cloud(0, 2, 2048)
    t = torch.from_numpy(xyz).to(DEV)
    idx = pointnet2.furthest_point_sample(t, 256)
    ref = O.furthest_point_sample(xyz, 256)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    new_xyz = np.take_along_axis(xyz, ref[..., None].astype(np.int64), axis=1)
    for r, ns in ((0.5, 16), (1.0, 32)):
        bq = pointnet2.ball_query(r, ns, t, torch.from_numpy(new_xyz).to(DEV))
        np.testing.assert_array_equal(bq.cpu().numpy(), O.ball_query(r, ns, xyz, new_xyz))
    d, i = pointnet2.three_nn(t[:, :700].contiguous(), torch.from_numpy(new_xyz).to(DEV))
    rd, ri = O.three_nn(xyz[:, :700], new_xyz)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_allclose(d.cpu().numpy(), rd, rtol=1e-6, atol=1e-7)


def test_fps_full_size_is_a_permutation_prefix():
    xyz = _cloud(3, 1, 16384)
    idx = pointnet2.furthest_point_sample(torch.from_numpy(xyz).to(DEV), 4096).cpu().numpy()[0]
    assert idx[0] == 0 and len(set(idx.tolist())) >= 4000      # duplicates only among padded duplicate points
    # spot check the first 64 picks against the oracle at full size
    np.testing.assert_array_equal(idx[:64], O.furthest_point_sample(xyz, 64)[0])


def _gt_boxes(B, rng):
    g = np.zeros((B, 10, 8), np.float32)
    for b in range(B):
        k = rng.integers(3, 9)
        g[b, :k, 0] = rng.uniform(5, 42, k); g[b, :k, 1] = rng.uniform(-15, 15, k); g[b, :k, 2] = rng.uniform(-1.2, -0.8, k)
        g[b, :k, 3:6] = np.array([3.9, 1.6, 1.56]) * rng.uniform(0.9, 1.1, (k, 3))
        g[b, :k, 6] = rng.uniform(-np.pi, np.pi, k); g[b, :k, 7] = 1
    return g


def _train_batch(seeds, rng):
    frames = [synthetic.hvpr_frame(s, shuffle=True) for s in seeds]
    pts = np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)])
    return {"points": torch.from_numpy(pts).to(DEV), "gt_boxes": torch.from_numpy(_gt_boxes(len(seeds), rng)).to(DEV),
            "batch_size": len(seeds)}


def test_vfe_train_mode_matches_oracle():
    cfg = hvpr_car_cfg()
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
    params = synthetic_weights.load_synthetic(model, seed=1)
    vfe = model.vfe.to(DEV).train()
    v, c, n = O.voxelize(synthetic.hvpr_frame(5), [0.16, 0.16, 3], list(cfg.DATA_CONFIG.POINT_CLOUD_RANGE), 32, 16000)
    coords = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    bd = vfe({"voxels": torch.from_numpy(v).to(DEV), "voxel_num_points": torch.from_numpy(n).to(DEV),
              "voxel_coords": torch.from_numpy(coords).to(DEV)})
    rpf, rsf, rmask, _ = O.pillar_vfe_scale(v, n.astype(np.float32), coords.astype(np.float32), O._sub(params, "vfe."), [0.16, 0.16, 3],
                                            list(cfg.DATA_CONFIG.POINT_CLOUD_RANGE), training=True)
    np.testing.assert_allclose(bd["pillar_features"].detach().cpu().numpy(), rpf.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(bd["pillar_scale_features"].detach().cpu().numpy(), rsf.numpy(), rtol=1e-3, atol=1e-3)


def test_training_steps_run_and_reduce_the_loss():
    cfg = hvpr_car_cfg()
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
    synthetic_weights.load_synthetic(model, seed=2, cls_bias=-4.595)
    model = model.to(DEV)
    opt = optim.build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = optim.build_scheduler(opt, total_iters_each_epoch=10, total_epochs=1, last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
    rng = np.random.default_rng(0)
    batch = _train_batch([10, 11], rng)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    losses = []
    for it in range(4):
        loss, tb = optim.train_step(model, opt, sched, dict(batch), it, cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        losses.append(float(loss))
        assert np.isfinite(losses[-1])
    assert losses[-1] < losses[0], losses
    changed = [k for k, v in model.named_parameters() if not torch.equal(v, before[k])]
    for prefix in ("backbone_3d.", "vfe.", "map_to_bev_module.memory", "backbone_2d.", "dense_head."):
        assert any(k.startswith(prefix) for k in changed), prefix            # every sub-module receives gradients
    assert int(model.global_step) == 4
    assert {"rpn_loss_cls", "rpn_loss_loc", "rpn_loss_dir", "mem_loss", "rpn_loss_point"} <= set(tb)
    # back to eval: the HIP inference path still works with the trained weights (folded BN is rebuilt)
    model.eval()
    with torch.no_grad():
        preds, _, _ = model({"points": batch["points"][batch["points"][:, 0] == 0].contiguous(), "batch_size": 1})
    assert preds[0]["pred_boxes"].shape[1] == 7


def test_get_score_topk_through_the_readout_kernel_equals_torch_topk():
    from hvpr_amd.map_to_bev import PointPillarScatter_Agg_Memory_1_scale
    cfg = hvpr_car_cfg()
    mod = PointPillarScatter_Agg_Memory_1_scale(cfg.MODEL.MAP_TO_BEV, grid_size=np.array([296, 248, 1])).to(DEV)
    g = torch.Generator().manual_seed(5)
    for M, N in ((3000, 16384), (257, 5000), (64, 2048 + 7), (10, 25)):
        pillars, points = torch.randn(M, 64, generator=g).to(DEV), torch.randn(N, 64, generator=g).to(DEV)
        got = mod._topk_points(pillars, points)
        logits = pillars @ points.t()
        want = torch.topk(logits, mod.k, dim=1)
        assert got.shape == (M, mod.k)
        # same top-k VALUES in the same (descending) order; indices equal wherever the values are distinct
        np.testing.assert_allclose(logits.gather(1, got).cpu().numpy(), want[0].cpu().numpy(), rtol=1e-5, atol=1e-5)
        assert (got == want[1]).float().mean() > 0.999


def test_config3_full_train_step_batch16():
    """BASELINE.json configs[2]: hvpr_car full train step (a1..a15: on-GPU voxelize, point stream, VFE, get_score + memory
    train branch, three canvases, two-stream backbone, head, target assigner, losses, backward, clip, Adam-onecycle) at the FULL
    size — 16 frames of 16384 points per step.  Finite loss, EVERY trainable parameter receives a finite gradient, the step
    changes the weights, peak memory is recorded (and bounded well below the 288 GB of the part)."""
    import json
    import os
    cfg = hvpr_car_cfg()
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
    synthetic_weights.load_synthetic(model, seed=3, cls_bias=-4.595)
    model = model.to(DEV).train()
    opt = optim.build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = optim.build_scheduler(opt, total_iters_each_epoch=10, total_epochs=1, last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
    rng = np.random.default_rng(16)
    batch = _train_batch(list(range(100, 116)), rng)
    assert batch["batch_size"] == 16 and batch["points"].shape == (16 * 16384, 5)
    torch.cuda.reset_peak_memory_stats()
    sched.step(0)
    opt.zero_grad()
    ret, tb, _ = model(dict(batch))
    loss = ret["loss"].mean()
    assert torch.isfinite(loss)
    loss.backward()
    missing = [k for k, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing
    bad = [k for k, p in model.named_parameters() if p.requires_grad and not torch.isfinite(p.grad).all()]
    assert not bad, bad
    nonzero = [k for k, p in model.named_parameters() if p.requires_grad and float(p.grad.abs().max()) > 0]
    assert len(nonzero) >= 0.95 * sum(1 for p in model.parameters() if p.requires_grad)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.OPTIMIZATION.GRAD_NORM_CLIP)
    opt.step()
    model.update_global_step()
    assert all(not torch.equal(v, before[k]) for k, v in model.named_parameters() if k in nonzero[:50])
    loss2, _ = optim.train_step(model, opt, sched, dict(batch), 1, cfg.OPTIMIZATION.GRAD_NORM_CLIP)     # a second whole step
    assert np.isfinite(float(loss2))
    peak = torch.cuda.max_memory_allocated() / 2**30
    print(f"config 3 (batch 16): loss {float(loss):.4f} -> {float(loss2):.4f}, peak memory {peak:.1f} GiB")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"config": "hvpr_car full train step, batch 16", "loss": [float(loss), float(loss2)], "peak_mem_GiB": round(peak, 1)},
              open(os.path.join("gpurun_out", "config3_train_step_test.json"), "w"))
    assert peak < 200.0


def test_pointnet2_msg_whole_module_matches_a_torch_plus_oracle_reference():
    """Row a9 as a whole: PointNet2MSG (4 SA-MSG scales + 2 FP levels) with EVERYTHING on the library's kernels — FPS, ball query,
    three-NN, the row-layout grouping / max-over-samples / interpolation (+ backward) and the shared MLPs (1x1 convolution +
    train-mode BatchNorm + ReLU on the matrix-core convolution and BatchNorm kernels) — against the torch form of the same module
    (tests/torch_forms.py: torch gathers, torch Conv2d / BatchNorm2d / max) with the index ops taken from the CPU oracle: point
    features, every parameter gradient, every BatchNorm's running statistics."""
    import copy
    import torch_forms
    cfg = hvpr_car_cfg()
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
    synthetic_weights.load_synthetic(model, seed=21)
    a = model.backbone_3d.to(DEV).train()
    b = copy.deepcopy(a)
    B, N = 2, 4096
    frames = [synthetic.hvpr_frame(70 + i, num_points=N, shuffle=True) for i in range(B)]
    pts = np.concatenate([np.concatenate([np.full((N, 1), i, np.float32), f], 1) for i, f in enumerate(frames)])
    # the SA levels sample 4096 and 1024 points (hvpr.yaml:61-66): with N = 4096 the first level keeps every point

    def run(mod):
        p = torch.from_numpy(pts).to(DEV)
        out = mod({"points": p, "batch_size": B})["point_features"]
        w = torch.linspace(0.5, 1.5, out.shape[1], device=DEV)
        (out * w).pow(2).mean().backward()
        return (out.detach(), {k: v.grad.detach().clone() for k, v in mod.named_parameters()},
                {k: v.detach().clone() for k, v in mod.named_buffers()})
    got, ggot, bgot = run(a)
    ref_ops = {
        "furthest_point_sample": lambda xyz, n: torch.from_numpy(O.furthest_point_sample(xyz.cpu().numpy(), n)).to(DEV),
        "ball_query": lambda r, ns, xyz, new: torch.from_numpy(O.ball_query(r, ns, xyz.cpu().numpy(), new.cpu().numpy())).to(DEV),
        "three_nn": lambda u, k: tuple(torch.from_numpy(t).to(DEV) for t in O.three_nn(u.cpu().numpy(), k.cpu().numpy())),
        "gather_operation": lambda f, idx: f.gather(2, idx.long().unsqueeze(1).expand(-1, f.shape[1], -1)),
    }
    saved = {k: getattr(pointnet2, k) for k in ref_ops}
    try:
        for k, v in ref_ops.items():
            setattr(pointnet2, k, v)
        with torch_forms.patched(b):
            want, gwant, bwant = run(b)
    finally:
        for k, v in saved.items():
            setattr(pointnet2, k, v)
    assert got.shape == (B * N, 64)
    rms = float(want.pow(2).mean().sqrt())
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-3, atol=1e-3 * rms)
    assert set(ggot) == set(gwant)
    # gradients of a ReLU network: the two runs may disagree on a ReLU decision whose pre-activation sits within round-off of zero,
    # and a flipped decision moves the gradients behind it by a finite amount: typical tensor within 2e-3, none beyond 1e-1 (a
    # wrong index or a wrong scatter-add shows up as O(1))
    errs = {k: float((ggot[k] - gwant[k]).norm() / gwant[k].norm().clamp_min(1e-30)) for k in gwant}
    print("PointNet2MSG gradients, own kernels vs torch form: median %.2e max %.2e" % (np.median(list(errs.values())), max(errs.values())))
    assert np.median(list(errs.values())) < 2e-3, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    assert max(errs.values()) < 1e-1, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    for k in bwant:
        if bwant[k].dtype.is_floating_point:
            torch.testing.assert_close(bgot[k], bwant[k], rtol=1e-4, atol=1e-6, msg=k)
        else:
            assert torch.equal(bgot[k], bwant[k]), k


def test_point_pillar_topk_ties_duplicates_and_order():
    """hvpr_point_pillar_topk_f32 beyond random data: 200 copies of one point (more near-ties than the candidate list holds: the
    exact pass over all items), an all-zero pillar row (k lowest indices), an item count that is not a multiple of 16 / 2048,
    and the output order (logit descending, index ascending on exact ties)."""
    from hvpr_amd.map_to_bev import PointPillarScatter_Agg_Memory_1_scale
    cfg = hvpr_car_cfg()
    mod = PointPillarScatter_Agg_Memory_1_scale(cfg.MODEL.MAP_TO_BEV, grid_size=np.array([296, 248, 1])).to(DEV)
    g = torch.Generator().manual_seed(12)
    N, M = 5003, 70
    points = torch.randn(N, 64, generator=g)
    pillars = torch.randn(M, 64, generator=g)
    dup = torch.randperm(N, generator=g)[:200]
    points[dup] = points[int(dup[0])].clone()                      # 200 identical rows
    pillars[3] = 2.0 * points[int(dup[0])]                 # ... which are this pillar's best match: a 200-way exact tie at the top
    pillars[5] = 0.0
    def check(pillars, points):
        got = mod._topk_points(pillars.to(DEV), points.to(DEV)).cpu().numpy()
        logits = (pillars.double() @ points.double().t()).numpy()
        for m in range(pillars.shape[0]):
            row = logits[m]
            vals = row[got[m]]
            assert np.all(np.diff(vals) <= 1e-6 * np.abs(row).max() + 1e-12), m            # descending (up to fp32 round-off)
            kth = np.sort(row)[-mod.k]
            assert vals.min() >= kth - 1e-5 * max(np.abs(row).max(), 1.0), m               # the k largest values (fp32 round-off)
            assert len(set(got[m].tolist())) == mod.k
        return got

    got = check(pillars, points)
    np.testing.assert_array_equal(got[5], np.arange(mod.k))                             # zero row: the k lowest indices
    np.testing.assert_array_equal(got[3], np.sort(dup.numpy())[:mod.k])                 # exact ties: lowest indices first
    # a frame padded by repetition (sample_points, data_processor.py:77-108): 16384 rows drawn from 3000 distinct points, and
    # non-negative, correlated features (post-ReLU) whose logits crowd together
    base = torch.relu(torch.randn(3000, 64, generator=g) + 1.0) * torch.rand(1, 64, generator=g)
    rep = base[torch.randint(0, 3000, (16384,), generator=g)]
    got = check(torch.relu(torch.randn(M, 64, generator=g) + 1.0), rep)
    for m in range(0, M, 9):                  # exact ties among copies: the lower index comes first
        rows = rep[got[m]]
        for a in range(mod.k - 1):
            if torch.equal(rows[a], rows[a + 1]):
                assert got[m][a] < got[m][a + 1], (m, a)
    # features outside the fp16 range of the pre-filter: every item is a candidate
    check(pillars[:20] * 1e5, points)


def test_point_index_prefetch_gives_the_same_step():
    """detector.prefetch_point_indices (the point stream's FPS / ball-query / three-NN indices computed ahead on a side stream)
    changes nothing: the loss of a training step is bit-identical with and without it, the point stream's gradients equal up to the
    order of its atomic scatter-adds."""
    import copy
    from hvpr_amd import detector, optim
    cfg = hvpr_car_cfg()
    torch.manual_seed(0)
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg, training=True))
    synthetic_weights.load_synthetic(model, seed=3, cls_bias=-4.595)
    model = model.to(DEV).train()
    batch = _train_batch([20, 21], np.random.default_rng(4))
    ref = copy.deepcopy(model)

    def step(m, b):
        m.zero_grad()
        ret, _, _ = m(b)
        ret["loss"].mean().backward()
        return ret["loss"].detach().clone()

    l0 = step(ref, dict(batch))
    l1 = step(model, model.prefetch_point_indices(dict(batch)))
    torch.cuda.synchronize()
    assert torch.equal(l0, l1)
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        if p.grad is None:
            assert q.grad is None
            continue
        if "backbone_3d" in n:                    # the point stream itself: same indices -> same gradients
            torch.testing.assert_close(p.grad, q.grad, rtol=1e-3, atol=1e-4 * float(q.grad.abs().max()) + 1e-12)
    got = [b for b in optim.prefetching(model, [dict(batch), dict(batch), dict(batch)])]
    assert len(got) == 3 and all("_pn2_plan" in b for b in got)
