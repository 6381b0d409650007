"""Random small shapes through the training convolutions (forward, data gradient, weight gradient; 3x3 stride 1 / 2 and 1x1) against
torch autograd in float64: python tools/fuzz_conv_train.py [cases] [seed]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from hvpr_amd import conv_train as ct

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda:0"
worst = {}
for it in range(cases):
    k = int(rng.choice([1, 3, 3, 3]))
    stride = int(rng.choice([1, 2])) if k == 3 else 1
    N, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 41)), int(rng.integers(1, 41))
    cin, cout = 8 * int(rng.integers(1, 18)), 8 * int(rng.integers(1, 18))
    g = torch.Generator().manual_seed(it)
    x = torch.randn(N, H, W, cin, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, generator=g) / (k * cin ** 0.5)).to(dev).requires_grad_(True)
    z = ct.conv(x, w, stride)
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    zr = F.conv2d(xr.permute(0, 3, 1, 2), wr, stride=stride, padding=k // 2).permute(0, 2, 3, 1)
    assert z.shape == zr.shape, (z.shape, zr.shape)
    dz = torch.randn(z.shape, generator=g).to(dev)
    dx, dw = torch.autograd.grad(z, (x, w), dz)
    dxr, dwr = torch.autograd.grad(zr, (xr, wr), dz.double())
    for name, got, ref in (("fwd", z, zr), ("dgrad", dx, dxr), ("wgrad", dw, dwr)):
        err = float((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        key = (name, k, stride)
        if err > worst.get(key, (0.0,))[0]:
            worst[key] = (err, (N, H, W, cin, cout))
        assert err < 2e-5, (name, k, stride, N, H, W, cin, cout, err)
for key in sorted(worst):
    print(key, "worst rel-to-max error %.2e at %s" % worst[key])
print("ok:", cases, "cases")
