"""Kernel experiment (timing build only): per-workgroup, per-tile time stamps of one hvpr_conv2d_nhwc_f32 launch.
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
buf = (ctypes.c_longlong * (1024 * 4 * 8))()
L.hvpr_exp_conv_dbg.restype = ctypes.c_int
assert L.hvpr_exp_conv_dbg(buf, 1024 * 4 * 8) == 0
d = np.frombuffer(buf, dtype=np.int64).reshape(1024, 4, 8)
live = d[:, :, 5] > 0
t0 = d[:, :, 2][live].min()
print(name, "records", int(live.sum()), "kernel span us", (d[:, :, 5][live].max() - t0) / 100.0)
tile_us = (d[:, :, 5] - d[:, :, 2])[live] / 100.0
print("tile us: mean %.1f min %.1f max %.1f" % (tile_us.mean(), tile_us.min(), tile_us.max()))
print("prologue us mean %.2f, k loop us mean %.2f (cycles mean %.0f), epilogue us mean %.2f" % (
    ((d[:, :, 3] - d[:, :, 2])[live] / 100.0).mean(), ((d[:, :, 4] - d[:, :, 3])[live] / 100.0).mean(), d[:, :, 6][live].mean(),
    ((d[:, :, 5] - d[:, :, 4])[live] / 100.0).mean()))
# co-residency: workgroups per (se, sh, cu) from HW_ID bits: cu [11:8], sh [12], se [15:13], xcc?/queue higher
hw = d[:, 0, 0]
cu = (hw >> 8) & 0xfffff0ff   # everything above the wave/simd bits
groups = {}
for b in range(1024):
    if live[b, 0]:
        groups.setdefault(int(hw[b]) >> 8, []).append(b)
sizes = sorted(len(v) for v in groups.values())
print("distinct hw ids (>>8):", len(groups), "workgroups per id: min %d max %d" % (sizes[0], sizes[-1]))
k = next(iter(sorted(groups, key=lambda k: -len(groups[k]))))
print("example CU", hex(k), "blocks", groups[k][:6])
for b in groups[k][:6]:
    print("   blk", b, [(int(d[b, i, 1]), round((d[b, i, 2] - t0) / 100.0, 1), round((d[b, i, 5] - t0) / 100.0, 1)) for i in range(4) if live[b, i]])
# first-start spread
st = d[:, 0, 2][live[:, 0]]
print("first tile start us: min %.1f max %.1f" % ((st.min() - t0) / 100.0, (st.max() - t0) / 100.0))
end = np.where(live, d[:, :, 5], 0).max(axis=1)[:768]
end_us = (end - t0) / 100.0
print("workgroup end us: percentiles 10/50/90/99/100:", np.percentile(end_us, [10, 50, 90, 99, 100]).round(1))
tl = (d[:, :, 5] - d[:, :, 2]) / 100.0
slow = np.argwhere(live & (tl > 80))
print("tiles slower than 80 us:", len(slow))
tx_n = (W + 7) // 8
for b, i in slow[:12]:
    it = int(d[b, i, 1]); xcd = it & 7; j = it >> 3; ct = j % 2; pt = (j // 2) * 8 + xcd
    print("   blk %d k %d it %d xcd %d ct %d tile (%d,%d) hw %x: start %.1f prologue %.1f kloop %.1f epi %.1f" % (
        b, i, it, xcd, ct, pt % tx_n, pt // tx_n, d[b, i, 0] >> 8, (d[b, i, 2] - t0) / 100.0, (d[b, i, 3] - d[b, i, 2]) / 100.0,
        (d[b, i, 4] - d[b, i, 3]) / 100.0, (d[b, i, 5] - d[b, i, 4]) / 100.0))
# per xcd mean tile time
for xc in range(8):
    m = live & ((d[:, :, 1] & 7) == xc)
    print("   xcd %d: mean tile %.1f us, max end %.1f" % (xc, tl[m].mean(), ((d[:, :, 5][m]).max() - t0) / 100.0))
