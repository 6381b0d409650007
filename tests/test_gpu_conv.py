"""GPU parity: implicit-GEMM fp32-MFMA convolution vs torch CPU conv (the oracle's F.conv2d) — 1e-3 rel (north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hvpr_amd import kernels

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(gen, *shape):
    return torch.randn(*shape, generator=gen)


def _close(got, ref, tol=1e-3):
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("cin,cout,stride,H,W", [(16, 32, 1, 11, 19), (32, 128, 1, 16, 24), (64, 96, 2, 17, 23),
                                                 (128, 256, 2, 24, 36), (8, 64, 1, 8, 16)])
def test_conv3x3_bn_relu(cfg, cin, cout, stride, H, W):
    g = torch.Generator().manual_seed(cin * 1000 + cout + cfg)
    x = _rand(g, 2, cin, H, W)
    w = _rand(g, cout, cin, 3, 3) / np.sqrt(cin * 9)
    scale, shift = torch.rand(cout, generator=g) + 0.5, _rand(g, cout) * 0.3
    ref = F.relu(F.conv2d(x, w, stride=stride, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    pc = kernels.pack_conv(w.to(DEV), scale.to(DEV), shift.to(DEV), stride=stride, relu=True, tile_cfg=cfg)
    y = kernels.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), pc)
    _close(y.permute(0, 3, 1, 2).cpu(), ref)


@pytest.mark.parametrize("cfg", [0, 1])
def test_sfm_step_gate_residual(cfg):
    g = torch.Generator().manual_seed(77 + cfg)
    x = _rand(g, 1, 64, 13, 21)
    w = _rand(g, 64, 64, 3, 3) / np.sqrt(64 * 9)
    scale, shift = torch.rand(64, generator=g) + 0.5, _rand(g, 64) * 0.3
    gate = torch.rand(1, 1, 13, 21, generator=g)
    ref = gate * F.relu(F.conv2d(x, w, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) + x
    pc = kernels.pack_conv(w.to(DEV), scale.to(DEV), shift.to(DEV), tile_cfg=cfg)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = kernels.conv2d_nhwc(xn, pc, gate=gate.reshape(1, 13, 21).contiguous().to(DEV), resid=xn)
    _close(y.permute(0, 3, 1, 2).cpu(), ref)


@pytest.mark.parametrize("s,cin,cout", [(1, 32, 16), (2, 64, 24), (4, 128, 128)])
def test_deconv_into_concat_slice(s, cin, cout):
    g = torch.Generator().manual_seed(900 + s)
    x = _rand(g, 2, cin, 7, 9)
    w = _rand(g, cin, cout, s, s) / np.sqrt(cin)
    scale, shift = torch.rand(cout, generator=g) + 0.5, _rand(g, cout) * 0.3
    ref = F.relu(F.conv_transpose2d(x, w, stride=s) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    pc = kernels.pack_deconv(w.to(DEV), scale.to(DEV), shift.to(DEV), tile_cfg=1)
    out = torch.full((2, 7 * s, 9 * s, cout + 40), -7.0, device=DEV)
    kernels.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), pc, out=out, out_coff=24)
    o = out.cpu()
    _close(o[..., 24:24 + cout].permute(0, 3, 1, 2), ref)
    assert (o[..., :24] == -7).all() and (o[..., 24 + cout:] == -7).all()


def test_head_1x1_with_bias():
    g = torch.Generator().manual_seed(5)
    x = _rand(g, 1, 384, 9, 14)
    w = _rand(g, 20, 384, 1, 1) / np.sqrt(384)
    b = _rand(g, 20)
    ref = F.conv2d(x, w, b)
    pc = kernels.pack_conv(w.to(DEV), None, b.to(DEV), relu=False, tile_cfg=2)
    y = kernels.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), pc)
    _close(y.permute(0, 3, 1, 2).cpu(), ref)


# ------------------------------------------------------------------------------------------------ split-bf16 precision modes
# planes = 2: "bf16x3" (3 products, ~2^-16 per product); planes = 3: "bf16x6" (6 products, fp32-grade)
TOL = {2: 1e-4, 3: 4e-6}


@pytest.mark.parametrize("planes", [2, 3])
def test_split_roundtrip(planes):
    g = torch.Generator().manual_seed(3)
    x = (_rand(g, 2, 5, 7, 32) * 10).to(DEV)
    back = kernels.unsplit_bf16(kernels.split_bf16(x, planes))
    rel = float(((back - x).abs() / x.abs().clamp_min(1e-30)).max())
    assert rel < 2.0 ** -16 if planes == 2 else rel == 0.0            # three planes hold all 24 mantissa bits


@pytest.mark.parametrize("planes", [2, 3])
@pytest.mark.parametrize("cfg", [0, 1, 3, 4, 5])
@pytest.mark.parametrize("cin,cout,stride,H,W", [(16, 32, 1, 11, 19), (32, 128, 1, 16, 24), (64, 96, 2, 17, 23),
                                                 (128, 256, 2, 24, 36), (256, 64, 1, 9, 33), (48, 64, 1, 20, 20)])
def test_split_bf16_conv_vs_float64(planes, cfg, cin, cout, stride, H, W):
    """bf16x3 stays two orders of magnitude inside the 1e-3 tolerance of north_star; bf16x6 is as close to float64 as the
    exact fp32 matrix-core kernel is (a few 1e-6)."""
    if stride == 2 and (cfg in (3, 5) or (planes == 3 and cfg != 4)):
        pytest.skip("stride 2: tile configurations 0, 1, 4 (three planes: 4 only)")
    g = torch.Generator().manual_seed(cin * 1000 + cout + cfg)
    x = _rand(g, 2, cin, H, W)
    w = _rand(g, cout, cin, 3, 3) / np.sqrt(cin * 9)
    scale, shift = torch.rand(cout, generator=g) + 0.5, _rand(g, cout) * 0.3
    ref = F.relu(F.conv2d(x.double(), w.double(), stride=stride, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    pc = kernels.pack_conv_bf3(w.to(DEV), scale.to(DEV), shift.to(DEV), stride=stride, relu=True, tile_cfg=cfg, planes=planes)
    xs = kernels.split_bf16(x.permute(0, 2, 3, 1).contiguous().to(DEV), planes)
    y = kernels.conv2d_nhwc_bf3(xs, pc, out_split=False)
    _close(y.permute(0, 3, 1, 2).cpu().double(), ref, tol=TOL[planes])
    if cout % 8 == 0:                                        # split output = the same numbers in split form
        ys = kernels.unsplit_bf16(kernels.conv2d_nhwc_bf3(xs, pc, out_split=True))
        _close(ys.permute(0, 3, 1, 2).cpu().double(), ref, tol=TOL[planes])


@pytest.mark.parametrize("planes", [2, 3])
@pytest.mark.parametrize("cfg", [0, 1, 3, 4, 5])
def test_split_bf16_sfm_step_gate_residual(planes, cfg):
    g = torch.Generator().manual_seed(177 + cfg)
    x = _rand(g, 1, 64, 13, 21)
    w = _rand(g, 64, 64, 3, 3) / np.sqrt(64 * 9)
    scale, shift = torch.rand(64, generator=g) + 0.5, _rand(g, 64) * 0.3
    gate = torch.rand(1, 1, 13, 21, generator=g)
    ref = gate.double() * F.relu(F.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)) + x.double()
    pc = kernels.pack_conv_bf3(w.to(DEV), scale.to(DEV), shift.to(DEV), tile_cfg=cfg, planes=planes)
    xs = kernels.split_bf16(x.permute(0, 2, 3, 1).contiguous().to(DEV), planes)
    gd = gate.reshape(1, 13, 21).contiguous().to(DEV)
    for out_split in (False, True):
        y = kernels.conv2d_nhwc_bf3(xs, pc, out_split=out_split, gate=gd, resid=xs)
        y = kernels.unsplit_bf16(y) if out_split else y
        _close(y.permute(0, 3, 1, 2).cpu().double(), ref, tol=TOL[planes])


@pytest.mark.parametrize("planes", [2, 3])
@pytest.mark.parametrize("s,cin,cout", [(1, 64, 16), (2, 128, 24), (4, 256, 128)])
def test_split_bf16_deconv_into_concat_slice(planes, s, cin, cout):
    g = torch.Generator().manual_seed(900 + s)
    x = _rand(g, 2, cin, 7, 9)
    w = _rand(g, cin, cout, s, s) / np.sqrt(cin)
    scale, shift = torch.rand(cout, generator=g) + 0.5, _rand(g, cout) * 0.3
    ref = F.relu(F.conv_transpose2d(x.double(), w.double(), stride=s) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    pc = kernels.pack_deconv_bf3(w.to(DEV), scale.to(DEV), shift.to(DEV), planes=planes)
    out = torch.full((2, 7 * s, 9 * s, cout + 40), -7.0, device=DEV)
    xs = kernels.split_bf16(x.permute(0, 2, 3, 1).contiguous().to(DEV), planes)
    o = kernels.deconv_nhwc_bf3(xs, pc, out, out_coff=24)
    _close(o[..., 24:24 + cout].permute(0, 3, 1, 2).cpu().double(), ref, tol=TOL[planes])
    assert (o[..., :24] == -7).all() and (o[..., 24 + cout:] == -7).all()
