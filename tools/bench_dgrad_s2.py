"""Timing of the data gradient of the two stride-2 3x3 level entries at batch 16 as the training step runs it (hvpr_amd/conv_train.py
_Conv.backward): zero-upsampled gradient -> stride-1 adjoint convolution."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvpr_amd import conv_train as ct

dev = "cuda:0"
for (H, W, cin, cout) in [(248, 296, 128, 256), (124, 148, 256, 512)]:
    x = torch.randn(16, H, W, cin, device=dev, requires_grad=True)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)).requires_grad_(True)
    z = ct.conv(x, w, 2)
    dz = torch.randn_like(z)
    for _ in range(2):
        torch.autograd.grad(z, x, dz, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        torch.autograd.grad(z, x, dz, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    useful = 2 * 9 * cin * cout * z.shape[1] * z.shape[2] * 16
    print(json.dumps({"layer": f"{H}x{W} {cin}->{cout} stride 2", "dgrad_us": round(us, 1), "useful_TFLOPs": round(useful / us / 1e6, 1)}), flush=True)
