"""GPU: the training exports of SURVEY.md §8b that replace torch ops — grouping / gather / three-interpolate with their
backward (vs differentiable torch gathers of the same definition) and the fused flat Adam-onecycle step (vs fixture G9 from
the reference's OptimWrapper + OneCycle, and vs the per-tensor AdamOneCycle)."""
import numpy as np
import pytest
import torch

import train_fixture_cases as C
from hvpr_amd import optim, pointnet2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_group(features, idx):
    B, Cc, _ = features.shape
    _, np_, ns = idx.shape
    return features.gather(2, idx.long().reshape(B, 1, np_ * ns).expand(-1, Cc, -1)).reshape(B, Cc, np_, ns)


def _torch_interp(features, idx, weight):
    B, Cc, _ = features.shape
    n = idx.shape[1]
    g = features.gather(2, idx.long().reshape(B, 1, n * 3).expand(-1, Cc, -1)).reshape(B, Cc, n, 3)
    return (g * weight.unsqueeze(1)).sum(dim=-1)


@pytest.mark.parametrize("B,Cc,N,np_,ns", [(2, 4, 500, 64, 16), (1, 67, 4096, 1024, 32), (3, 1, 33, 7, 1), (2, 131, 100, 1, 5)])
def test_group_points_forward_and_backward(B, Cc, N, np_, ns):
    g = torch.Generator().manual_seed(B * 1000 + Cc)
    f = torch.randn(B, Cc, N, generator=g).to(DEV).requires_grad_(True)
    idx = torch.randint(0, N, (B, np_, ns), generator=g, dtype=torch.int32).to(DEV)
    idx[:, 0, :] = 5 % N                                   # heavy duplicates: the backward accumulates
    out = pointnet2.grouping_operation(f, idx)
    ref = _torch_group(f.detach().clone().requires_grad_(True), idx)
    assert torch.equal(out, ref)                           # a gather: bit-exact
    go = torch.randn(out.shape, generator=g).to(DEV)
    (gf,) = torch.autograd.grad(out, f, go)
    f2 = f.detach().clone().requires_grad_(True)
    (gr,) = torch.autograd.grad(_torch_group(f2, idx), f2, go)
    np.testing.assert_allclose(gf.cpu().numpy(), gr.cpu().numpy(), rtol=1e-5, atol=1e-5)      # atomics: order differs
    # gather_operation = one sample per group
    i1 = idx[:, :, 0].contiguous()
    assert torch.equal(pointnet2.gather_operation(f, i1), f.gather(2, i1.long().unsqueeze(1).expand(-1, Cc, -1)))


@pytest.mark.parametrize("B,Cc,m,n", [(2, 128, 1024, 4096), (1, 5, 10, 300), (2, 64, 4096, 16384)])
def test_three_interpolate_forward_and_backward(B, Cc, m, n):
    g = torch.Generator().manual_seed(m + n)
    f = torch.randn(B, Cc, m, generator=g).to(DEV).requires_grad_(True)
    idx = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32).to(DEV)
    w = torch.rand(B, n, 3, generator=g).to(DEV)
    w = w / w.sum(-1, keepdim=True)
    out = pointnet2.three_interpolate(f, idx, w)
    f2 = f.detach().clone().requires_grad_(True)
    ref = _torch_interp(f2, idx, w)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    go = torch.randn(out.shape, generator=g).to(DEV)
    (gf,) = torch.autograd.grad(out, f, go)
    (gr,) = torch.autograd.grad(ref, f2, go)
    np.testing.assert_allclose(gf.cpu().numpy(), gr.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_fused_adam_reproduces_fixture_g9(golden_dir):
    """Fixture G9 = 100 one-cycle steps + 3 optimiser steps of the reference's OptimWrapper(Adam, true_wd) + OneCycle, here
    through hvpr_fused_adam_truewd_f32 on one flat buffer."""
    made = []

    def make(net, wd):
        made.append(optim.FusedAdamOneCycle(net, wd=wd))
        return made[-1]
    C.run_g9(golden_dir, DEV, rtol=1e-4, make_optimizer=make)
    assert made and made[0].steps == 3


def test_fused_adam_equals_per_tensor_adam_with_clipping_and_checkpoint_interop():
    torch.manual_seed(0)

    def net():
        torch.manual_seed(1)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 17, 3, bias=False), torch.nn.BatchNorm2d(17), torch.nn.ReLU(),
                                   torch.nn.Conv2d(17, 5, 1), torch.nn.Flatten(), torch.nn.Linear(5 * 6 * 6, 3)).to(DEV)
    a, b = net(), net()
    oa, ob = optim.AdamOneCycle(a, wd=0.01), optim.FusedAdamOneCycle(b, wd=0.01)
    sa, sb = optim.OneCycle(oa, 20, 0.003, [0.95, 0.85], 10, 0.4), optim.OneCycle(ob, 20, 0.003, [0.95, 0.85], 10, 0.4)
    xs = torch.randn(8, 4, 3, 8, 8, device=DEV) * 30       # large inputs: the 0.5 clip is active

    def step(m, o, s, it, x):
        s.step(it)
        m.train(); o.zero_grad()
        m(x).pow(2).mean().backward()
        if hasattr(o, "clip_grad_norm"):
            o.clip_grad_norm(0.5)
        else:
            torch.nn.utils.clip_grad_norm_(m.parameters(), 0.5)
        o.step()
    for it in range(5):
        step(a, oa, sa, it, xs[it]); step(b, ob, sb, it, xs[it])
        for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
            np.testing.assert_allclose(v.float().cpu().numpy(), w.float().cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg=f"step {it} {k}")
    # checkpoints move between the two forms (torch.optim.Adam's state-dict layout in the same parameter order)
    c, d = net(), net()
    c.load_state_dict(b.state_dict()); d.load_state_dict(b.state_dict())
    oc, od = optim.AdamOneCycle(c, wd=0.01), optim.FusedAdamOneCycle(d, wd=0.01)
    oc.load_state_dict(ob.state_dict()); od.load_state_dict(oa.state_dict())
    sc, sd = optim.OneCycle(oc, 20, 0.003, [0.95, 0.85], 10, 0.4), optim.OneCycle(od, 20, 0.003, [0.95, 0.85], 10, 0.4)
    step(a, oa, sa, 5, xs[5]); step(c, oc, sc, 5, xs[5]); step(d, od, sd, 5, xs[5])
    for (k, v), (_, w), (_, u) in zip(a.state_dict().items(), c.state_dict().items(), d.state_dict().items()):
        np.testing.assert_allclose(v.float().cpu().numpy(), w.float().cpu().numpy(), rtol=5e-5, atol=1e-7, err_msg="fused -> torch " + k)
        np.testing.assert_allclose(v.float().cpu().numpy(), u.float().cpu().numpy(), rtol=5e-5, atol=1e-7, err_msg="torch -> fused " + k)


def test_gather_rows_forward_and_scatter_add_backward():
    from hvpr_amd.map_to_bev import _GatherRows
    g = torch.Generator().manual_seed(9)
    rows = torch.randn(5000, 64, generator=g).to(DEV).requires_grad_(True)
    idx = torch.randint(0, 5000, (700, 20), generator=g).to(DEV)
    idx[:50] = 7                                            # one hot row: the backward accumulates 1000 times into it
    out = _GatherRows.apply(rows, idx)
    assert out.shape == (700, 20, 64) and torch.equal(out, rows[idx])
    go = torch.randn(out.shape, generator=g).to(DEV)
    (gr,) = torch.autograd.grad(out, rows, go)
    r2 = rows.detach().clone().requires_grad_(True)
    (gt,) = torch.autograd.grad(r2[idx], r2, go)
    np.testing.assert_allclose(gr.cpu().numpy(), gt.cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("R,n_items,scale", [(700, 2000, 6.0), (33, 2000, 9.0), (1200, 777, 5.0), (4096, 2000, 1.0), (16, 2048, 12.0)])
def test_memory_train_forward_backward_match_the_reference_formula(R, n_items, scale):
    """hvpr_memory_train_fwd/bwd vs torch autograd (float64) of memory_module.py:36-48 + hard_shrink_relu :85-87.  `scale` sets
    the feature norm: small -> every softmax value is below the shrink threshold (empty support: zero output, zero gradients),
    large -> a few items per row survive."""
    from hvpr_amd.map_to_bev import _MemoryTrain
    from torch_forms import hard_shrink_relu
    g = torch.Generator().manual_seed(R + n_items)
    x = (torch.relu(torch.randn(R, 64, generator=g)) * scale / 4).to(DEV).requires_grad_(True)
    w = ((torch.rand(n_items, 64, generator=g) * 2 - 1) / 8).to(DEV).requires_grad_(True)
    lam = 0.0025
    y = _MemoryTrain.apply(x, w, lam)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    att = torch.softmax(xd @ wd.t(), dim=1)
    sh = torch.nn.functional.normalize(hard_shrink_relu(att, lam), p=1, dim=1)
    yr = sh @ wd
    nnz = (att > lam).sum(1)
    if scale >= 5:
        assert int((nnz > 0).sum()) > R // 4                 # the case really exercises non-empty supports
    # rows whose support decision sits within fp32 noise of the threshold may legitimately differ: compare the others
    margin = ((att - lam).abs() / lam).min(dim=1)[0]
    ok = (margin > 1e-4).cpu().numpy()
    assert ok.mean() > 0.9
    rms = float(yr.detach().pow(2).mean().sqrt()) + 1e-30
    np.testing.assert_allclose(y.detach().cpu().numpy()[ok], yr.detach().cpu().numpy()[ok], rtol=1e-3, atol=1e-3 * rms)
    dy = torch.randn(R, 64, generator=g).to(DEV) * torch.from_numpy(ok).to(DEV).float()[:, None]      # no gradient into the excluded rows
    dx, dw = torch.autograd.grad(y, (x, w), dy)
    dxr, dwr = torch.autograd.grad(yr, (xd, wd), dy.double())
    for got, ref, what in ((dx, dxr, "dx"), (dw, dwr, "dW")):
        err = float((got.double() - ref).norm() / ref.norm().clamp_min(1e-30))
        assert err < 1e-3 or float(ref.norm()) < 1e-12, (what, err)
        if float(ref.norm()) < 1e-12:
            assert float(got.abs().max()) < 1e-12


@pytest.mark.parametrize("c", [128, 64, 32])
def test_scatter_canvas_forward_and_gather_backward(c):
    """The differentiable scatter of the training branch (pointpillar_scatter.py:87-167: rows -> dense canvases): forward through
    hvpr_scatter_bev_fwd_f32, backward = every pillar reads its cell of the gradient canvas through hvpr_gather_rows_f32."""
    from hvpr_amd import kernels
    from hvpr_amd.map_to_bev import _ScatterCanvas
    g = torch.Generator().manual_seed(c)
    B, nx, ny, M = 2, 40, 24, 300
    cells = torch.randperm(B * nx * ny, generator=g)[:M]
    coords = torch.stack([cells // (nx * ny), torch.zeros(M, dtype=torch.long), (cells // nx) % ny, cells % nx], 1).to(torch.int32).to(DEV)
    feats = torch.randn(M, c, generator=g).to(DEV).requires_grad_(True)
    ws = kernels.scatter_workspace(B, nx, ny, DEV)
    canvas = _ScatterCanvas.apply(feats, coords, B, nx, ny, ws)
    assert canvas.shape == (B, c, ny, nx)
    ref = torch.zeros(B, ny, nx, c, device=DEV)
    ref[coords[:, 0].long(), coords[:, 2].long(), coords[:, 3].long()] = feats.detach()
    assert torch.equal(canvas.permute(0, 2, 3, 1), ref)
    go = torch.randn(canvas.shape, generator=g).to(DEV)
    (gf,) = torch.autograd.grad(canvas, feats, go)
    want = go.permute(0, 2, 3, 1)[coords[:, 0].long(), coords[:, 2].long(), coords[:, 3].long()]
    assert torch.equal(gf, want)


@pytest.mark.parametrize("M,P", [(2, 32), (3, 32), (7, 32), (1001, 32), (3, 20), (777, 20), (50, 5)])
def test_pillar_vfe_train_fwd_bwd_match_float64_autograd(M, P):
    """hvpr_pillar_vfe_train_fwd_f32 / hvpr_pillar_vfe_bwd_f32 (the two PFN layers with batch-statistics BatchNorm,
    pillar_vfe.py:184-221) against torch autograd of the same module in float64: features, running statistics and the
    gradients of both Linear weights and both BatchNorm affines."""
    import copy
    from hvpr_amd import vfe as V
    from hvpr_amd.config import hvpr_car_cfg
    cfg = hvpr_car_cfg()
    g = torch.Generator().manual_seed(M + P)
    mod = V.PillarVFE_Scale(cfg.MODEL.VFE, 4, [0.16, 0.16, 3], [0, -19.84, -2.5, 47.36, 19.84, 0.5]).train()
    with torch.no_grad():
        for layer in mod.pfn_layers:
            layer.norm.weight.copy_(torch.rand(layer.norm.weight.shape, generator=g) + 0.5)
            layer.norm.bias.copy_(torch.randn(layer.norm.bias.shape, generator=g) * 0.3)
    import torch_forms
    ref = copy.deepcopy(mod).double().to(DEV)
    torch_forms.patch(ref)                                               # the torch form of the same module (tests/torch_forms.py)
    mod = mod.to(DEV)
    num = torch.randint(1, min(9, P + 1), (M,), generator=g)       # P < 32 (config 5 uses 20): BatchNorm counts M * P slots
    num[0] = P                                                           # a full pillar: no padded slot
    if M > 2:
        num[2] = 1
    coords = torch.stack([torch.zeros(M, dtype=torch.long), torch.zeros(M, dtype=torch.long),
                          torch.randint(0, 248, (M,), generator=g), torch.randint(0, 296, (M,), generator=g)], dim=1).int()
    vox = torch.rand(M, P, 4, generator=g) * torch.tensor([47.0, 39.0, 3.0, 1.0]) + torch.tensor([0.0, -19.5, -2.5, 0.0])
    vox = vox * (torch.arange(P).view(1, -1, 1) < num.view(-1, 1, 1))
    dfeat = torch.randn(M, 64, generator=g)

    def run(m, dtype):
        bd = {"voxels": vox.to(DEV, dtype), "voxel_num_points": num.to(DEV).int(), "voxel_coords": coords.to(DEV)}
        out = m(bd)["pillar_features"]
        (out * dfeat.to(DEV, dtype)).sum().backward()
        return out

    o_ref = run(ref, torch.float64)
    o_hip = run(mod, torch.float32)
    scale = float(o_ref.detach().abs().max())
    assert float((o_hip.double() - o_ref).abs().max()) < 2e-5 * scale
    for (name, p_h), (_, p_r) in zip(mod.named_parameters(), ref.named_parameters()):
        if "pfn_layers" not in name:
            continue
        gr, gh = p_r.grad, p_h.grad.double()
        err = float((gh - gr).abs().max()) / max(float(gr.abs().max()), 1e-6)
        assert err < 2e-4, (name, err, float(gr.abs().max()))
    for l_h, l_r in zip(mod.pfn_layers, ref.pfn_layers):
        assert float((l_h.norm.running_mean.double() - l_r.norm.running_mean).abs().max()) < 1e-6
        assert float((l_h.norm.running_var.double() - l_r.norm.running_var).abs().max()) < 1e-5 * float(l_r.norm.running_var.abs().max())
        assert int(l_h.norm.num_batches_tracked) == 1
    # the torch form in fp32 on the same module sits in the same band
    with torch_forms.patched(mod):
        mod.zero_grad()
        o_t = run(mod, torch.float32)
    assert float((o_t.double() - o_ref).abs().max()) < 2e-5 * scale


@pytest.mark.parametrize("N,H,W,C", [(2, 40, 48, 32), (1, 31, 45, 64), (2, 9, 7, 128), (3, 5, 6, 8)])
def test_spatial_gate_train_fwd_bwd_match_torch_autograd(N, H, W, C):
    """hvpr_spatial_gate_train_fwd/bwd_f32 (ChannelPool -> conv3x3 2->1 + bias -> BatchNorm2d(1) with batch statistics -> sigmoid,
    spatial_attention.py:47-63) against torch autograd of the same formula in float64: gate, batch statistics, and the gradients
    of y, the convolution weight / bias and the BatchNorm affine."""
    from hvpr_amd import conv_train as ct
    g = torch.Generator().manual_seed(C + H)
    y = torch.relu(torch.randn(N, H, W, C, generator=g)).to(DEV).requires_grad_(True)       # post-ReLU scale stream
    w = (torch.randn(1, 2, 3, 3, generator=g) * 0.4).to(DEV).requires_grad_(True)
    b = torch.tensor([0.3], device=DEV, requires_grad=True)
    gamma = torch.tensor([1.3], device=DEV, requires_grad=True)
    beta = torch.tensor([-0.2], device=DEV, requires_grad=True)
    eps = 1e-3
    gate, mean, var = ct.spatial_gate_train(y, w, b, gamma, beta, eps)
    dg = torch.randn(N, H, W, 1, generator=g).to(DEV)
    got = torch.autograd.grad(gate, (y, w, b, gamma, beta), dg)
    y64, w64, b64, g64, be64 = (t.detach().double().requires_grad_(True) for t in (y, w, b, gamma, beta))
    yc = y64.permute(0, 3, 1, 2)
    pooled = torch.cat((yc.max(dim=1, keepdim=True)[0], yc.mean(dim=1, keepdim=True)), dim=1)
    a = torch.nn.functional.conv2d(pooled, w64, b64, padding=1)
    m, v = a.mean(), a.var(unbiased=False)
    ref = torch.sigmoid((a - m) / torch.sqrt(v + eps) * g64 + be64).permute(0, 2, 3, 1)
    want = torch.autograd.grad(ref, (y64, w64, b64, g64, be64), dg.double())
    np.testing.assert_allclose(gate.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(mean), float(m), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(float(var), float(v), rtol=1e-4)
    for a_, b_, what in zip(got, want, ("dy", "dw", "dbias", "dgamma", "dbeta")):
        scale = float(b_.abs().max())
        if what == "dbias":        # exact gradient of a bias in front of a batch-statistics BatchNorm is zero
            assert float(a_.abs().max()) < 1e-4 * float(want[1].abs().max()), what
            continue
        np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=1e-3, atol=2e-5 * scale, err_msg=what)


@pytest.mark.parametrize("B,N,C,npoint,ns", [(2, 500, 1, 64, 16), (1, 300, 67 - 3, 40, 32), (3, 64, 5, 7, 4)])
def test_group_rows_max_samples_and_fp_rows_match_torch(B, N, C, npoint, ns):
    """The row-layout data movement of the point stream (csrc/point_mlp.hip) against torch indexing: grouping (+ scatter-add
    gradient), max over the samples (+ arg-max gradient), feature-propagation rows (+ both gradients)."""
    from hvpr_amd import pointnet2 as P
    g = torch.Generator().manual_seed(B * 100 + C)
    xyz = torch.randn(B, N, 3, generator=g).to(DEV)
    feat = torch.randn(B, N, C, generator=g).to(DEV).requires_grad_(True)
    new_xyz = torch.randn(B, npoint, 3, generator=g).to(DEV)
    idx = torch.randint(0, N, (B, npoint, ns), generator=g).int().to(DEV)
    cpad = P._cpad(3 + C)
    rows = P._GroupRows.apply(xyz, feat, new_xyz, idx, cpad)
    ar = torch.arange(B, device=DEV)[:, None, None]
    fr = feat.detach().clone().requires_grad_(True)
    want = torch.cat([xyz[ar, idx.long()] - new_xyz.unsqueeze(2), fr[ar, idx.long()],
                      torch.zeros(B, npoint, ns, cpad - 3 - C, device=DEV)], dim=-1).reshape(-1, cpad)
    assert torch.equal(rows, want)
    go = torch.randn(rows.shape, generator=g).to(DEV)
    (gf,) = torch.autograd.grad(rows, feat, go)
    (gw,) = torch.autograd.grad(want, fr, go)
    torch.testing.assert_close(gf, gw, rtol=1e-4, atol=1e-5)
    # max over samples
    y = torch.randn(B * npoint * ns, 24, generator=g).to(DEV)
    y[: ns] = 0.0                                                       # a group of exact ties: the lowest sample takes the gradient
    y = y.requires_grad_(True)
    out = P._MaxSamples.apply(y, ns)
    yr = y.detach().clone().requires_grad_(True)
    wout = yr.view(-1, ns, 24).max(dim=1)[0]
    assert torch.equal(out, wout)
    go = torch.randn(out.shape, generator=g).to(DEV)
    (gy,) = torch.autograd.grad(out, y, go)
    assert torch.equal(gy[0], go[0]) and float(gy[1:ns].abs().max()) == 0.0
    (gyr,) = torch.autograd.grad(wout, yr, go)
    assert torch.equal(gy[ns:], gyr[ns:])
    # feature propagation rows
    m, n, C1, C2 = N, npoint * 3, 16, C
    known = torch.randn(B, m, C1, generator=g).to(DEV).requires_grad_(True)
    skip = torch.randn(B, n, C2, generator=g).to(DEV).requires_grad_(True)
    i3 = torch.randint(0, m, (B, n, 3), generator=g).int().to(DEV)
    w3 = torch.rand(B, n, 3, generator=g).to(DEV)
    cp = P._cpad(C1 + C2)
    rows = P._FpRows.apply(known, i3, w3, skip, cp)
    kr, sr = known.detach().clone().requires_grad_(True), skip.detach().clone().requires_grad_(True)
    gk = kr[ar, i3.long()]
    want = torch.cat([(gk[:, :, 0] * w3[:, :, 0:1] + gk[:, :, 1] * w3[:, :, 1:2]) + gk[:, :, 2] * w3[:, :, 2:3], sr,
                      torch.zeros(B, n, cp - C1 - C2, device=DEV)], dim=-1).reshape(-1, cp)
    torch.testing.assert_close(rows, want, rtol=1e-6, atol=1e-6)
    go = torch.randn(rows.shape, generator=g).to(DEV)
    got = torch.autograd.grad(rows, (known, skip), go)
    wantg = torch.autograd.grad(want, (kr, sr), go)
    torch.testing.assert_close(got[0], wantg[0], rtol=1e-4, atol=1e-5)
    assert torch.equal(got[1], wantg[1])


def test_row_kernels_without_features_and_without_skip():
    """The first SA level of a 3-channel cloud has no point features (rows = [xyz offsets | 0]) and the last FP level of such a
    cloud has no skip features: both forms of hvpr_group_rows_f32 / hvpr_fp_rows_f32."""
    from hvpr_amd import pointnet2 as P
    g = torch.Generator().manual_seed(9)
    B, N, npoint, ns = 2, 200, 30, 8
    xyz = torch.randn(B, N, 3, generator=g).to(DEV)
    new_xyz = torch.randn(B, npoint, 3, generator=g).to(DEV)
    idx = torch.randint(0, N, (B, npoint, ns), generator=g).int().to(DEV)
    rows = P._GroupRows.apply(xyz, None, new_xyz, idx, 8)
    ar = torch.arange(B, device=DEV)[:, None, None]
    want = torch.cat([xyz[ar, idx.long()] - new_xyz.unsqueeze(2), torch.zeros(B, npoint, ns, 5, device=DEV)], dim=-1).reshape(-1, 8)
    assert torch.equal(rows, want)
    known = torch.randn(B, N, 16, generator=g).to(DEV).requires_grad_(True)
    i3 = torch.randint(0, N, (B, 50, 3), generator=g).int().to(DEV)
    w3 = torch.rand(B, 50, 3, generator=g).to(DEV)
    rows = P._FpRows.apply(known, i3, w3, None, 16)
    gk = known[ar, i3.long()]
    want = ((gk[:, :, 0] * w3[:, :, 0:1] + gk[:, :, 1] * w3[:, :, 1:2]) + gk[:, :, 2] * w3[:, :, 2:3]).reshape(-1, 16)
    torch.testing.assert_close(rows, want, rtol=1e-6, atol=1e-6)
    (gkn,) = torch.autograd.grad(rows, known, torch.ones_like(rows))
    (gw,) = torch.autograd.grad(want, known, torch.ones_like(want))
    torch.testing.assert_close(gkn, gw, rtol=1e-4, atol=1e-5)


def test_scattering_gradients_are_reproducible_and_match_float64():
    """The three gradients that scatter (QueryAndGroup rows, three-interpolate rows, the row gather of get_score) are sums per
    destination in a fixed order (hvpr_segment_sum_rows_f32 over edges sorted by destination, stable) — no float atomics: two
    backward passes give the SAME BITS, and the values match a float64 index_add."""
    from hvpr_amd import kernels, map_to_bev
    g = torch.Generator().manual_seed(5)
    B, N, C, np_, ns = 2, 300, 24, 64, 16
    xyz = torch.rand(B, N, 3, generator=g).to(DEV)
    new_xyz = xyz[:, :np_].contiguous()
    idx = torch.randint(0, 40, (B, np_, ns), generator=g, dtype=torch.int32).to(DEV)       # heavy duplication: 40 hot points
    gout = torch.randn(B * np_ * ns, 32, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        feat = torch.randn(B, N, C, generator=torch.Generator().manual_seed(6)).to(DEV).requires_grad_(True)
        rows = pointnet2._GroupRows.apply(xyz, feat, new_xyz, idx, 32)
        (rows * gout).sum().backward()
        outs.append(feat.grad.clone())
    assert torch.equal(outs[0], outs[1])
    ref = torch.zeros(B * N, C, dtype=torch.float64, device=DEV)
    dst = (idx.long().view(B, -1) + torch.arange(B, device=DEV).view(B, 1) * N).view(-1)
    ref.index_add_(0, dst, gout[:, 3:3 + C].double())
    torch.testing.assert_close(outs[0].double().view(B * N, C), ref, rtol=1e-5, atol=1e-5)
    # three-interpolate rows
    m, n, C1 = 50, 400, 16
    known0 = torch.randn(B, m, C1, generator=g).to(DEV)
    idx3 = torch.randint(0, 12, (B, n, 3), generator=g, dtype=torch.int32).to(DEV)
    w3 = torch.rand(B, n, 3, generator=g).to(DEV)
    gout = torch.randn(B * n, 16, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        known = known0.clone().requires_grad_(True)
        rows = pointnet2._FpRows.apply(known, idx3, w3, None, 16)
        (rows * gout).sum().backward()
        outs.append(known.grad.clone())
    assert torch.equal(outs[0], outs[1])
    ref = torch.zeros(B * m, C1, dtype=torch.float64, device=DEV)
    dst = (idx3.long().view(B, -1) + torch.arange(B, device=DEV).view(B, 1) * m).view(-1)
    ref.index_add_(0, dst, (gout.double().view(B * n, 1, 16) * w3.double().view(B * n, 3, 1)).view(-1, 16))
    torch.testing.assert_close(outs[0].double().view(B * m, C1), ref, rtol=1e-5, atol=1e-5)
    # row gather
    rows0 = torch.randn(500, 64, generator=g).to(DEV)
    pick = torch.randint(0, 30, (200, 20), generator=g).to(DEV)
    gout = torch.randn(200, 20, 64, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        r = rows0.clone().requires_grad_(True)
        (map_to_bev._GatherRows.apply(r, pick) * gout).sum().backward()
        outs.append(r.grad.clone())
    assert torch.equal(outs[0], outs[1])
    ref = torch.zeros(500, 64, dtype=torch.float64, device=DEV).index_add_(0, pick.view(-1), gout.double().view(-1, 64))
    torch.testing.assert_close(outs[0].double(), ref, rtol=1e-5, atol=1e-5)


class _TwoRankReplay:
    """SyncBatchNorm hook for ONE GPU: the two 'ranks' run one after the other, round after round; every hook call records the
    rank's own buffer and adds what the OTHER rank recorded at the same call index in the previous round.  Call i is right once
    calls 0..i-1 were, so after (number of hook calls per step) + 1 rounds both ranks see exactly what a lock-step all-reduce
    would have given them."""

    def __init__(self):
        from hvpr_amd._lib import ALLREDUCE_FN, lib
        self.prev, self.cur, self.rank, self.calls = {0: [], 1: []}, {0: [], 1: []}, 0, 0
        self.fn = ALLREDUCE_FN(self)
        lib().hvpr_set_batchnorm_allreduce(self.fn, None)

    def __call__(self, buf, n, stream, ctx):
        from hvpr_amd import conv_train as ct
        t = ct.device_doubles(buf, n)
        i = len(self.cur[self.rank])
        self.cur[self.rank].append(t.clone())
        other = self.prev[1 - self.rank]
        if i < len(other):
            assert other[i].numel() == n
            t += other[i]
        self.calls += 1
        return 0

    def next_round(self):
        self.prev, self.cur = self.cur, {0: [], 1: []}

    def close(self):
        from hvpr_amd._lib import lib
        lib().hvpr_set_batchnorm_allreduce(None, None)


def test_sync_batchnorm_hook_pfn_two_emulated_ranks_equal_the_joint_batch():
    """hvpr_set_batchnorm_allreduce on the fused PFN kernels (two BatchNorm1d inside one call, forward and backward): two ranks
    with 700 / 301 pillars, emulated on one GPU (_TwoRankReplay), against ONE process holding all 1001 pillars — features equal,
    parameter gradients add up to the joint ones (what DDP's mean x world size gives), batch means equal."""
    import copy
    from hvpr_amd import vfe as V
    from hvpr_amd.config import hvpr_car_cfg
    cfg = hvpr_car_cfg()
    g = torch.Generator().manual_seed(5)
    M, MA, P = 1001, 700, 32
    base = V.PillarVFE_Scale(cfg.MODEL.VFE, 4, [0.16, 0.16, 3], [0, -19.84, -2.5, 47.36, 19.84, 0.5]).train()
    with torch.no_grad():
        for layer in base.pfn_layers:
            layer.norm.weight.copy_(torch.rand(layer.norm.weight.shape, generator=g) + 0.5)
            layer.norm.bias.copy_(torch.randn(layer.norm.bias.shape, generator=g) * 0.3)
    num = torch.randint(1, 9, (M,), generator=g)
    num[0] = P
    coords = torch.stack([torch.zeros(M, dtype=torch.long), torch.zeros(M, dtype=torch.long),
                          torch.randint(0, 248, (M,), generator=g), torch.randint(0, 296, (M,), generator=g)], dim=1).int()
    vox = torch.rand(M, P, 4, generator=g) * torch.tensor([47.0, 39.0, 3.0, 1.0]) + torch.tensor([0.0, -19.5, -2.5, 0.0])
    vox = vox * (torch.arange(P).view(1, -1, 1) < num.view(-1, 1, 1))
    dfeat = torch.randn(M, 64, generator=g)

    def run(m, sl):
        m.zero_grad()
        bd = {"voxels": vox[sl].to(DEV), "voxel_num_points": num[sl].to(DEV).int(), "voxel_coords": coords[sl].to(DEV)}
        out = m(bd)["pillar_features"]
        (out * dfeat[sl].to(DEV)).sum().backward()
        return out.detach()

    joint = copy.deepcopy(base).to(DEV)
    o_joint = run(joint, slice(0, M))
    parts = (slice(0, MA), slice(MA, M))
    hook = _TwoRankReplay()
    try:
        for rnd in range(7):                      # 2 hook calls forward + 4 backward (the backward recomputes the statistics)
            mods = [copy.deepcopy(base).to(DEV) for _ in range(2)]
            outs = []
            for r in range(2):
                hook.rank = r
                outs.append(run(mods[r], parts[r]))
            assert len(hook.cur[0]) == 6 and len(hook.cur[1]) == 6
            hook.next_round()
    finally:
        hook.close()
    scale = float(o_joint.abs().max())
    assert float((torch.cat(outs) - o_joint).abs().max()) < 2e-6 * scale
    for (name, pj), (_, pa), (_, pb) in zip(joint.named_parameters(), mods[0].named_parameters(), mods[1].named_parameters()):
        if "pfn_layers" not in name:
            continue
        gj, gs = pj.grad.double(), pa.grad.double() + pb.grad.double()
        assert float((gs - gj).abs().max()) < 2e-5 * max(float(gj.abs().max()), 1e-6), name
    for lj, la, lb in zip(joint.pfn_layers, mods[0].pfn_layers, mods[1].pfn_layers):
        assert torch.equal(la.norm.running_mean, lb.norm.running_mean)
        assert float((la.norm.running_mean - lj.norm.running_mean).abs().max()) < 1e-7
    # and per-rank statistics without the hook are NOT the joint ones (the test would pass vacuously otherwise)
    alone = run(copy.deepcopy(base).to(DEV), parts[0])
    assert float((alone - o_joint[:MA]).abs().max()) > 1e-4 * scale


def test_sync_batchnorm_hook_spatial_gate_two_emulated_ranks_equal_the_joint_batch():
    """The same for SpatialAttention's one-channel BatchNorm (hvpr_spatial_gate_train_fwd/bwd_f32): frames 0 and 1 on two emulated
    ranks against both frames in one process — gate and dy equal, parameter gradients add up."""
    from hvpr_amd import conv_train as ct
    g = torch.Generator().manual_seed(11)
    N, H, W, C = 2, 40, 48, 32
    y = torch.relu(torch.randn(N, H, W, C, generator=g) * torch.tensor([1.0, 2.5]).view(2, 1, 1, 1)).to(DEV)
    w = (torch.randn(1, 2, 3, 3, generator=g) * 0.4).to(DEV)
    b, gamma, beta = torch.tensor([0.3], device=DEV), torch.tensor([1.3], device=DEV), torch.tensor([-0.2], device=DEV)
    dg = torch.randn(N, H, W, 1, generator=g).to(DEV)

    def run(sl):
        leaves = [t.clone().requires_grad_(True) for t in (y[sl], w, b, gamma, beta)]
        gate, mean, var = ct.spatial_gate_train(*leaves, 1e-3)
        grads = torch.autograd.grad(gate, leaves, dg[sl])
        return gate.detach(), mean, var, grads

    gate_j, mean_j, var_j, grads_j = run(slice(0, 2))
    hook = _TwoRankReplay()
    try:
        for rnd in range(3):
            res = []
            for r in range(2):
                hook.rank = r
                res.append(run(slice(r, r + 1)))
            hook.next_round()
    finally:
        hook.close()
    assert float((torch.cat([res[0][0], res[1][0]]) - gate_j).abs().max()) < 1e-6
    for r in range(2):
        np.testing.assert_allclose(float(res[r][1]), float(mean_j), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(float(res[r][2]), float(var_j), rtol=1e-6)
    dy = torch.cat([res[0][3][0], res[1][3][0]])
    assert float((dy - grads_j[0]).abs().max()) < 2e-6 * float(grads_j[0].abs().max())
    for k, what in ((1, "dw"), (3, "dgamma"), (4, "dbeta")):
        s = res[0][3][k].double() + res[1][3][k].double()
        assert float((s - grads_j[k].double()).abs().max()) < 2e-5 * float(grads_j[k].abs().max()), what
    alone = run(slice(0, 1))
    assert float((alone[0] - gate_j[:1]).abs().max()) > 1e-3


def test_sync_batchnorm_hook_failure_is_reported_not_swallowed():
    """A hook that fails (returns non-zero — e.g. the python side caught an exception from the collective) makes the entry point
    return HVPR_ERR_LAUNCH, which the wrapper raises; removing the hook restores the per-rank path."""
    from hvpr_amd import conv_train as ct
    from hvpr_amd._lib import ALLREDUCE_FN, lib
    y = torch.rand(1, 8, 8, 8, device=DEV)
    w = torch.randn(1, 2, 3, 3, device=DEV)
    one = torch.ones(1, device=DEV)
    bad = ALLREDUCE_FN(lambda buf, n, stream, ctx: 1)
    lib().hvpr_set_batchnorm_allreduce(bad, None)
    try:
        with pytest.raises(RuntimeError, match="hvpr_spatial_gate_train_fwd_f32"):
            ct.spatial_gate_train(y, w, one, one, one, 1e-3)
    finally:
        lib().hvpr_set_batchnorm_allreduce(None, None)
    gate, _, _ = ct.spatial_gate_train(y, w, one, one, one, 1e-3)
    assert bool(torch.isfinite(gate).all())


def test_memory_addressing_once_per_point_equals_once_per_pick():
    """MemoryUnit_Agg.forward_train_indexed (the addressing evaluated once per point, then gathered: what the scatter module's training
    branch runs) against _forward_train on the materialised positives points[idx] (memory_module.py:31-59 as written, one row per
    (pillar, k) pair): the SAME bits forward (a row's result does not depend on its neighbours in the launch), gradients w.r.t. the
    points, the pillars and the bank equal to round-off (J^T applied to the summed dy of a point's picks instead of summed J^T dy)."""
    from hvpr_amd import map_to_bev
    g = torch.Generator().manual_seed(77)
    N, M, k = 5000, 700, 20
    mem = map_to_bev.MemoryUnit_Agg(2000, 64, 0.0025).to(DEV).train()
    mem.weight.data.mul_(4.0)                                   # supports of 10-80 items per row (G16's regime)
    points0 = (torch.relu(torch.randn(N, 64, generator=g)) * 0.5).to(DEV)
    pillars0 = torch.relu(torch.randn(M, 64, generator=g)).to(DEV)
    # picks concentrated on a tenth of the points: every picked point serves ~30 (pillar, k) pairs
    idx = (torch.randint(0, N // 10, (M, k), generator=g) * 10).to(DEV)
    cot = torch.randn(M, 64, generator=g).to(DEV)
    res = []
    for indexed in (False, True):
        pts, pil = points0.clone().requires_grad_(True), pillars0.clone().requires_grad_(True)
        mem.weight.grad = None
        if indexed:
            out = mem.forward_train_indexed(pil, k, pts, idx, map_to_bev._EdgePlan(idx, N))["output"]
        else:
            out = mem(pil, k, map_to_bev._GatherRows.apply(pts, idx))["output"]
        (out * cot).sum().backward()
        res.append((out.detach(), pts.grad.clone(), mem.weight.grad.clone(), pil.grad))
    (o0, gp0, gw0, gq0), (o1, gp1, gw1, gq1) = res
    assert torch.equal(o0, o1)
    assert gq0 is None and gq1 is None                         # the pillars only enter through detached aggregation weights (:53-57)
    assert float((gp1 - gp0).norm() / gp0.norm()) < 1e-5 and float((gw1 - gw0).norm() / gw0.norm()) < 1e-5
    assert float(gp0.norm()) > 0 and float(gw0.norm()) > 0
