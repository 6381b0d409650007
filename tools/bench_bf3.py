"""bf16x3 convolution experiment: accuracy vs the exact fp32 kernel and time, on the hvpr_car 3x3 layer shapes."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import kernels

DEV = "cuda:0"


def conv3(xs, pc):
    return kernels.conv2d_nhwc_bf3(xs, pc, out_split=False)


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


SHAPES = [("L0 3x3 128->128 @248x296", 128, 128, 248, 296, 1), ("L1 3x3 256->256 @124x148", 256, 256, 124, 148, 1),
          ("L2 3x3 512->512 @62x74", 512, 512, 62, 74, 1), ("L1 3x3s2 128->256 @248x296", 128, 256, 248, 296, 2),
          ("small 16->32 @11x19", 16, 32, 11, 19, 1)]
for name, cin, cout, H, W, stride in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(1, H, W, cin, generator=g)).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(DEV)
    scale, shift = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    pc = kernels.pack_conv(w, scale, shift, stride=stride, tile_cfg=1)
    ref = kernels.conv2d_nhwc(x, pc)
    t_ref = timeit(lambda: kernels.conv2d_nhwc(x, pc, out=ref))
    ref64 = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), stride=stride, padding=1).relu().permute(0, 2, 3, 1)
    line = f"{name:28s} fp32 {t_ref:7.1f} us (err {float((ref.cpu().double() - ref64).abs().max() / ref64.abs().max()):.1e})"
    for planes in (2, 3):
        xs = kernels.split_bf16(x, planes)
        for cfg in ((1, 3, 4, 5) if stride == 1 else (1, 4)):
            pc3 = kernels.pack_conv_bf3(w, scale, shift, stride=stride, tile_cfg=cfg, planes=planes)
            try:
                y = conv3(xs, pc3)
            except Exception:
                continue
            err = float((y.cpu().double() - ref64).abs().max() / ref64.abs().max())
            t = timeit(lambda: conv3(xs, pc3))
            line += f" | x{3 if planes == 2 else 6} cfg{cfg}: {t:6.1f} us err {err:.1e}"
    print(line, flush=True)
