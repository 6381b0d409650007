"""Stand-alone timing of the Winograd-domain weight gradient (hvpr_conv2d_wino_wgrad_f32) on the three backbone levels at batch 16."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvpr_amd import conv_train

dev = "cuda:0"
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for (H, W, C) in [(248, 296, 128), (124, 148, 256), (62, 74, 512)]:
    x = torch.randn(batch, H, W, C, device=dev)
    dz = torch.randn(batch, H, W, C, device=dev)
    for _ in range(3):
        dw = conv_train.conv_wgrad(x, dz, 9, 1, C, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        conv_train.conv_wgrad(x, dz, 9, 1, C, C)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    flops = 2 * 9 * C * C * H * W * batch
    print(json.dumps({"level": f"{H}x{W}x{C}", "batch": batch, "us": round(us, 1), "direct_TFLOPs": round(flops / us / 1e6, 1),
                      "executed_TFLOPs": round(flops * 16 / 36 / us / 1e6, 1)}), flush=True)
