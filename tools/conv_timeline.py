"""Kernel experiment (timing build only): cycle stamps of the chunk phases inside hvpr_conv2d_nhwc_f32 (second tile of 8 workgroups).
    python -m hvpr_amd.build --timing && HVPR_AMD_LIB=$PWD/hvpr_amd/libhvpr_amd_timing.so python tools/conv_timeline.py [shape]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import kernels, _lib
from bench_conv import SHAPES

si = int(sys.argv[1]) if len(sys.argv) > 1 else 0
name, cin, cout, H, W, stride, taps = SHAPES[si]
g = torch.Generator().manual_seed(0)
x = torch.randn(1, H, W, cin, generator=g).cuda()
w = torch.randn(cout, cin, 3, 3, generator=g).cuda() / (cin * 9) ** 0.5
pc = kernels.pack_conv(w, torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda"), stride=stride, tile_cfg=1)
y = kernels.conv2d_nhwc(x, pc)
for _ in range(3):
    kernels.conv2d_nhwc(x, pc, out=y)
torch.cuda.synchronize()
L = _lib.lib()
n = 8 * 4 * 64 * 4
buf = (ctypes.c_longlong * n)()
L.hvpr_exp_conv_dbg.restype = ctypes.c_int
assert L.hvpr_exp_conv_dbg(buf, n) == 0
d = np.frombuffer(buf, dtype=np.int64).reshape(8, 4, 64, 4)
nch = cin // 8
print(name, "chunks", nch)
for s in range(8):
    dd = d[s, :, :nch]
    if dd[0, 0, 0] == 0:
        continue
    wait = (dd[:, :, 1] - dd[:, :, 0]).mean(); issue = (dd[:, :, 2] - dd[:, :, 1]).mean(); mul = (dd[:, :, 3] - dd[:, :, 2]).mean()
    period = (dd[:, 1:, 0] - dd[:, :-1, 0]).mean()
    print("blk slot %d (block %d): per chunk wait+barrier %.0f, DMA issue %.0f, multiply %.0f cycles; chunk period %.0f" % (s, s * 97 + 5, wait, issue, mul, period))
s = 0
print("wave 0 of slot 0, chunks 4..8:", (d[s, 0, 4:9] - d[s, 0, 4, 0]).tolist())
