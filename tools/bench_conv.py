"""Micro-benchmark of hvpr_conv2d_nhwc_f32 on the hvpr_car layer shapes: TFLOP/s per tile configuration."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import kernels

DEV = "cuda:0"
SHAPES = [  # name, Cin, Cout, H, W, stride, taps
    ("L0 3x3 128->128 @248x296", 128, 128, 248, 296, 1, 9),
    ("L1 3x3 256->256 @124x148", 256, 256, 124, 148, 1, 9),
    ("L2 3x3 512->512 @62x74", 512, 512, 62, 74, 1, 9),
    ("L1 3x3s2 128->256 @248x296", 128, 256, 248, 296, 2, 9),
    ("L2 3x3s2 256->512 @124x148", 256, 512, 124, 148, 2, 9),
    ("SYN 3x3 512->128 @248x296", 512, 128, 248, 296, 1, 9),
    ("SYN 3x3 32->128 @248x296", 32, 128, 248, 296, 1, 9),
]


def run(name, cin, cout, H, W, stride, taps, cfg, iters=20):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, H, W, cin, generator=g).to(DEV)
    w = torch.randn(cout, cin, 3, 3, generator=g).to(DEV) / (cin * 9) ** 0.5
    pc = kernels.pack_conv(w, torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV), stride=stride, tile_cfg=cfg)
    y = kernels.conv2d_nhwc(x, pc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        kernels.conv2d_nhwc(x, pc, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    OH, OW = y.shape[1], y.shape[2]
    fl = 2 * cin * cout * taps * OH * OW
    return us, fl / us / 1e6


if __name__ == "__main__":
    cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
    only = os.environ.get("CONV_SHAPE")
    for si, s in enumerate(SHAPES):
        if only is not None and int(only) != si:
            continue
        line = f"{s[0]:32s}"
        for cfg in cfgs:
            us, tf = run(*s, cfg)
            line += f"  cfg{cfg}: {us:7.1f} us {tf:6.1f} TF/s"
        print(line, flush=True)
