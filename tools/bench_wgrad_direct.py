"""Stand-alone timing of the direct weight gradient (hvpr_conv2d_wgrad_nhwc_f32) on the 1x1 (deconvolution / head) and 3x3 stride-2
shapes of the training step at batch 16."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvpr_amd import conv_train

dev = "cuda:0"
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
# (H, W, Cin, Cout, taps, stride)
SHAPES = [(248, 296, 128, 128, 1, 1), (124, 148, 256, 512, 1, 1), (62, 74, 512, 2048, 1, 1), (248, 296, 384, 20, 1, 1),
          (248, 296, 128, 256, 9, 2), (124, 148, 256, 512, 9, 2)]
for (H, W, Cin, Cout, taps, stride) in SHAPES:
    OH, OW = (H, W) if taps == 1 else ((H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1)
    x = torch.randn(batch, H, W, Cin, device=dev)
    dz = torch.randn(batch, OH, OW, Cout, device=dev)
    for _ in range(3):
        conv_train.conv_wgrad(x, dz, taps, stride, Cout, Cin)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        conv_train.conv_wgrad(x, dz, taps, stride, Cout, Cin)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    flops = 2 * taps * Cin * Cout * OH * OW * batch
    gb = (x.numel() + dz.numel()) * 4 / 1e9
    print(json.dumps({"shape": f"{H}x{W} {Cin}->{Cout} taps {taps} stride {stride}", "us": round(us, 1), "TFLOPs": round(flops / us / 1e6, 1),
                      "min_GB": round(gb, 2), "GBps_at_min": round(gb / us * 1e6)}), flush=True)
