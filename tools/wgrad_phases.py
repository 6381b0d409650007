"""One call of the Winograd weight gradient per backbone level through the timing build (HVPR_AMD_LIB=hvpr_amd/libhvpr_amd_timing.so):
prints the per-tile phase cycles of two workgroups."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvpr_amd import conv_train

for (H, W, C) in [(248, 296, 128), (124, 148, 256), (62, 74, 512)]:
    x = torch.randn(16, H, W, C, device="cuda:0")
    dz = torch.randn(16, H, W, C, device="cuda:0")
    print("level", H, W, C, flush=True)
    conv_train.conv_wgrad(x, dz, 9, 1, C, C)
    torch.cuda.synchronize()
