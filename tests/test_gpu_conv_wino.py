"""hvpr_conv2d_wino_nhwc_f32 (Winograd F(2x2,3x3), fp32 matrix cores) against torch's conv2d in float64 and against the direct
kernel hvpr_conv2d_nhwc_f32: BaseBEVBackbone_Scale's Conv3x3 + BN + ReLU and SFM step (base_bev_backbone.py:228-315)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, w, scale, shift, relu, gate=None, resid=None):
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double() * scale.double().view(-1, 1, 1, 1), padding=1).permute(0, 2, 3, 1)
    y = y + shift.double()
    if relu:
        y = torch.relu(y)
    if gate is not None:
        y = gate.double().unsqueeze(-1) * y + resid.double()
    return y


@pytest.mark.parametrize("N,H,W,cin,cout,groups,relu,gated", [
    (1, 8, 16, 8, 64, 1, False, False),          # one tile, one chunk
    (1, 16, 16, 16, 64, 2, False, False),
    (2, 13, 21, 24, 68, 1, True, False),         # ragged image, cout not a multiple of 64, odd chunk count
    (2, 13, 21, 24, 68, 2, True, True),
    (1, 62, 74, 64, 128, 1, True, True),         # level-2 geometry, SFM epilogue
    (3, 31, 37, 128, 128, 2, True, False),
    (1, 8, 16, 8, 64, 4, False, False),          # px_groups 4 = 32 output channels per workgroup
    (2, 13, 21, 24, 68, 4, True, True),
    (1, 62, 74, 64, 128, 4, True, True),
    (3, 31, 37, 128, 160, 4, True, False),
])
def test_wino_conv_matches_float64(N, H, W, cin, cout, groups, relu, gated):
    from hvpr_amd import kernels
    g = torch.Generator().manual_seed(H * 131 + cin)
    x = torch.randn(N, H, W, cin, generator=g).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)).to(DEV)
    scale = (torch.rand(cout, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    gate = torch.rand(N, H, W, generator=g).to(DEV) if gated else None
    resid = torch.randn(N, H, W, cout, generator=g).to(DEV) if gated else None
    pc = kernels.pack_conv_wino(w, scale, shift, relu=relu, px_groups=groups)
    y = kernels.conv2d_wino_nhwc(x, pc, gate=gate, resid=resid)
    ref = _ref(x, w, scale, shift, relu, gate, resid)
    pd = kernels.pack_conv(w, scale, shift, stride=1, relu=relu, tile_cfg=1)
    yd = kernels.conv2d_nhwc(x, pd, gate=gate, resid=resid)
    s = float(ref.abs().max())
    err_w, err_d = float((y.double() - ref).abs().max()) / s, float((yd.double() - ref).abs().max()) / s
    print(f"winograd {err_w:.2e}  direct {err_d:.2e} (max abs error / output scale)")
    assert err_w < 5e-6, (err_w, err_d)
    # written into a channel slice of a wider tensor (the concat of the 2D backbone)
    wide = torch.full((N, H, W, cout + 24), 7.0, device=DEV)
    kernels.conv2d_wino_nhwc(x, pc, out=wide, out_coff=12, gate=gate, resid=resid)
    assert torch.equal(wide[..., 12:12 + cout], y) and bool((wide[..., :12] == 7).all()) and bool((wide[..., 12 + cout:] == 7).all())


def test_wino_adjoint_pack_is_the_data_gradient():
    """pack(adjoint) + the same kernel = d(conv)/d(input) (a11: base_bev_backbone.py:228-279 in training)."""
    from hvpr_amd import kernels
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 20, 28, 32, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(64, 32, 3, 3, generator=g) / 17.0).to(DEV)
    dy = torch.randn(2, 20, 28, 64, generator=g).to(DEV)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    (y * dy.double()).sum().backward()
    pc = kernels.pack_conv_wino(w, relu=False, adjoint=True)
    assert (pc.cin, pc.cout) == (64, 32)
    dx = kernels.conv2d_wino_nhwc(dy, pc)
    err = float((dx.double() - x.grad.double()).abs().max() / x.grad.abs().max())
    assert err < 5e-6, err


@pytest.mark.parametrize("N,H,W,cin,cout", [(1, 8, 8, 64, 64), (2, 13, 21, 24, 68), (3, 31, 37, 128, 192), (2, 62, 74, 64, 128)])
def test_wino_wgrad_matches_float64(N, H, W, cin, cout):
    """hvpr_conv2d_wino_wgrad_nhwc_f32 against autograd in float64 and the direct weight-gradient kernel."""
    import os
    from hvpr_amd import conv_train
    g = torch.Generator().manual_seed(W + cin)
    x = torch.randn(N, H, W, cin, generator=g).to(DEV)
    dz = torch.randn(N, H, W, cout, generator=g).to(DEV)
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    (y * dz.double()).sum().backward()
    dw = conv_train.conv_wgrad(x, dz, 9, 1, cout, cin)
    os.environ["HVPR_CONV_ALGO"] = "direct"           # the direct weight-gradient kernel on the same tensors
    try:
        dwd = conv_train.conv_wgrad(x, dz, 9, 1, cout, cin)
    finally:
        del os.environ["HVPR_CONV_ALGO"]
    s = float(w.grad.abs().max())
    ew, ed = float((dw.double() - w.grad).abs().max()) / s, float((dwd.double() - w.grad).abs().max()) / s
    print(f"winograd wgrad {ew:.2e}  direct {ed:.2e}")
    assert ew < 2e-5 and ew < 4 * ed + 1e-6, (ew, ed)
    assert torch.equal(dw, conv_train.conv_wgrad(x, dz, 9, 1, cout, cin))          # deterministic


def test_wino_full_size_properties():
    """BASELINE.json sizes (level 0 of hvpr_car: 248 x 296 x 128, batch 2) through size-independent properties: the direct kernel
    on the same input (1e-5 of the output scale), linearity in the input, and determinism of forward and weight gradient."""
    from hvpr_amd import conv_train, kernels
    g = torch.Generator().manual_seed(77)
    N, H, W, C = 2, 248, 296, 128
    x1 = torch.randn(N, H, W, C, generator=g).to(DEV)
    x2 = torch.randn(N, H, W, C, generator=g).to(DEV)
    w = (torch.randn(C, C, 3, 3, generator=g) / np.sqrt(9 * C)).to(DEV)
    pw = kernels.pack_conv_wino(w, relu=False)
    y1, y2, y12 = kernels.conv2d_wino_nhwc(x1, pw), kernels.conv2d_wino_nhwc(x2, pw), kernels.conv2d_wino_nhwc(x1 + x2, pw)
    s = float(y12.abs().max())
    assert float((y12 - (y1 + y2)).abs().max()) < 2e-5 * s                       # linearity (fp32 round-off only)
    yd = kernels.conv2d_nhwc(x1, kernels.pack_conv(w, None, None, stride=1, relu=False, tile_cfg=1))
    assert float((y1 - yd).abs().max()) < 1e-5 * float(yd.abs().max())
    assert torch.equal(y1, kernels.conv2d_wino_nhwc(x1, pw))
    dw = conv_train.conv_wgrad(x1, y2, 9, 1, C, C)
    assert torch.equal(dw, conv_train.conv_wgrad(x1, y2, 9, 1, C, C))
    # <dz, conv(x, w)> = <dw, w>: the weight gradient is the adjoint of the convolution in w
    lhs = float((y2.double() * y1.double()).sum())
    rhs = float((dw.double() * w.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * abs(lhs), (lhs, rhs)


@pytest.mark.parametrize("N,H,W,cin,cout", [(2, 13, 21, 24, 68), (1, 62, 74, 64, 128), (3, 31, 37, 128, 192)])
def test_wino_fused_bn_statistics(N, H, W, cin, cout):
    """The per-tile sums the Winograd kernel leaves behind (bn_partials) give the same batch statistics as a pass over the
    written output (hvpr_bn_stats_nhwc_f32) and as float64 — ragged images, channel counts that are not a multiple of 64."""
    from hvpr_amd import conv_train
    g = torch.Generator().manual_seed(H + cout)
    x = torch.randn(N, H, W, cin, generator=g).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)).to(DEV)
    z, partials = conv_train.conv_fwd_raw(x, w, 1, stats=True)
    assert partials is not None and partials.shape[1:] == (2, cout)
    mean, var, invstd, _ = conv_train.bn_statistics(z, 1e-3, partials)
    mean2, var2, invstd2, _ = conv_train.bn_statistics(z, 1e-3)               # the separate pass over z
    zd = z.double().reshape(-1, cout)
    np.testing.assert_allclose(mean.cpu().numpy(), zd.mean(0).cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(var.cpu().numpy(), zd.var(0, unbiased=False).cpu().numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(mean.cpu().numpy(), mean2.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(invstd.cpu().numpy(), invstd2.cpu().numpy(), rtol=2e-5)
