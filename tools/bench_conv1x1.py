"""Micro-benchmark of the 1x1-type layers of hvpr_car: the three deconvs (ConvTranspose k = s) into the 384-channel concat and the head."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import kernels

DEV = "cuda:0"


def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


CFG = int(sys.argv[1]) if len(sys.argv) > 1 else 1     # tile: 0 = 128 px x 128 ch, 1 = 64 x 64, 2 = 128 px x 64 ch
g = torch.Generator().manual_seed(0)
out = torch.empty(1, 248, 296, 384, device=DEV)
ones, zeros = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
for name, cin, H, W, s, coff in (("deconv L0 k1 128->128", 128, 248, 296, 1, 0), ("deconv L1 k2 256->128", 256, 124, 148, 2, 128),
                                 ("deconv L2 k4 512->128", 512, 62, 74, 4, 256)):
    x = torch.randn(1, H, W, cin, generator=g).to(DEV)
    w = torch.randn(cin, 128, s, s, generator=g).to(DEV) / cin ** 0.5
    pc = kernels.pack_deconv(w, ones, zeros, tile_cfg=CFG)
    us = timeit(lambda: kernels.conv2d_nhwc(x, pc, out=out, out_coff=coff))
    fl = 2 * cin * 128 * s * s * H * W
    mb = (x.numel() + 248 * 296 * 128) * 4 / 1e6
    print(f"{name:26s} {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s  {mb / us * 1e-3 * 1e3:7.1f} GB/s (in+out {mb:.0f} MB)", flush=True)
x = torch.randn(1, 248, 296, 384, generator=g).to(DEV)
w = torch.randn(20, 384, 1, 1, generator=g).to(DEV) / 384 ** 0.5
pc = kernels.pack_conv(w, None, torch.zeros(20, device=DEV), relu=False, tile_cfg=CFG)
us = timeit(lambda: kernels.conv2d_nhwc(x, pc))
mb = (x.numel() + 248 * 296 * 20) * 4 / 1e6
print(f"{'head 1x1 384->20':26s} {us:7.1f} us  {2 * 384 * 20 * 248 * 296 / us / 1e6:6.1f} TF/s  {mb / us:7.1f} GB/s (in+out {mb:.0f} MB)", flush=True)
