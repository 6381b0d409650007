"""Norm-wise error of every training operator of hvpr_amd.conv_train against float64 torch (one-off precision probe)."""
import sys, os
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import conv_train as ct
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)


def nerr(a, b):
    return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))


for (N, H, W, cin, cout, s) in [(2, 64, 80, 128, 128, 1), (2, 64, 80, 128, 256, 2), (2, 32, 40, 256, 256, 1), (2, 64, 80, 32, 32, 1)]:
    x = torch.randn(N, H, W, cin, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(DEV).requires_grad_(True)
    z = ct.conv(x, w, s)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    zd = F.conv2d(xd.permute(0, 3, 1, 2), wd, stride=s, padding=1).permute(0, 2, 3, 1)
    dz = torch.randn(z.shape, generator=g).to(DEV)
    dx, dw = torch.autograd.grad(z, (x, w), dz)
    dxd, dwd = torch.autograd.grad(zd, (xd, wd), dz.double())
    xt, wt = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    zt = F.conv2d(xt.permute(0, 3, 1, 2), wt, stride=s, padding=1).permute(0, 2, 3, 1)
    dxt, dwt = torch.autograd.grad(zt, (xt, wt), dz)
    print(f"conv {cin}->{cout} s{s}: fwd {nerr(z, zd):.1e} ({nerr(zt, zd):.1e})  dgrad {nerr(dx, dxd):.1e} ({nerr(dxt, dxd):.1e})  wgrad {nerr(dw, dwd):.1e} ({nerr(dwt, dwd):.1e})   [own (torch fp32)]")
for (N, H, W, cin, cout, s) in [(2, 64, 80, 128, 128, 1), (2, 32, 40, 256, 128, 2), (2, 16, 20, 512, 128, 4)]:
    x = torch.randn(N, H, W, cin, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(cin, cout, s, s, generator=g) / cin ** 0.5).to(DEV).requires_grad_(True)
    z = ct.deconv(x, w)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    zd = F.conv_transpose2d(xd.permute(0, 3, 1, 2), wd, stride=s).permute(0, 2, 3, 1)
    dz = torch.randn(z.shape, generator=g).to(DEV)
    dx, dw = torch.autograd.grad(z, (x, w), dz)
    dxd, dwd = torch.autograd.grad(zd, (xd, wd), dz.double())
    print(f"deconv {cin}->{cout} s{s}: fwd {nerr(z, zd):.1e}  dgrad {nerr(dx, dxd):.1e}  wgrad {nerr(dw, dwd):.1e}")
for shape, sparse in [((2, 64, 80, 128), False), ((2, 64, 80, 128), True), ((2, 16, 20, 512), True)]:
    C = shape[-1]
    z = (torch.randn(shape, generator=g) * 2 + 0.5)
    if sparse:
        z = z * (torch.rand(shape[:3] + (1,), generator=g) < 0.2) + 3.0
    z = z.to(DEV).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    import copy
    ref = copy.deepcopy(bn).double()
    ref32 = copy.deepcopy(bn)
    y = ct.bn_relu(z, bn)
    zd = z.detach().double().requires_grad_(True)
    yd = torch.relu(ref(zd.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    zt = z.detach().clone().requires_grad_(True)
    yt = torch.relu(ref32(zt.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    dy = torch.randn(shape, generator=g).to(DEV)
    dz, dg, db = torch.autograd.grad(y, (z, bn.weight, bn.bias), dy)
    dzd, dgd, dbd = torch.autograd.grad(yd, (zd, ref.weight, ref.bias), dy.double())
    dzt, dgt, dbt = torch.autograd.grad(yt, (zt, ref32.weight, ref32.bias), dy)
    print(f"bn_relu {shape} sparse={sparse}: y {nerr(y, yd):.1e} ({nerr(yt, yd):.1e}) dz {nerr(dz, dzd):.1e} ({nerr(dzt, dzd):.1e}) dgamma {nerr(dg, dgd):.1e} ({nerr(dgt, dgd):.1e}) dbeta {nerr(db, dbd):.1e} ({nerr(dbt, dbd):.1e})")
