"""Per-layer timing of the Winograd kernel against the direct fp32 kernel on the three backbone levels (stand-alone launches)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hvpr_amd import kernels

dev = "cuda:0"
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = []
for (H, W, C) in [(248, 296, 128), (124, 148, 256), (62, 74, 512)]:
    x = torch.randn(batch, H, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / (3 * C ** 0.5)
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    row = {"level": f"{H}x{W}x{C}", "batch": batch}
    flops = 2 * 9 * C * C * H * W * batch
    cands = [("direct64", lambda: kernels.conv2d_nhwc(x, kernels_pd[1]))]
    kernels_pd = {t: kernels.pack_conv(w, sc, sh, stride=1, relu=True, tile_cfg=t) for t in (0, 1, 2)}
    pw = {g: kernels.pack_conv_wino(w, sc, sh, relu=True, px_groups=g) for g in (1, 2, 4)}
    outs = {}
    for name, fn in [("direct_64x64", lambda: kernels.conv2d_nhwc(x, kernels_pd[1])),
                     ("direct_128x64", lambda: kernels.conv2d_nhwc(x, kernels_pd[2])),
                     ("direct_128x128", lambda: kernels.conv2d_nhwc(x, kernels_pd[0])),
                     ("wino_g1", lambda: kernels.conv2d_wino_nhwc(x, pw[1])),
                     ("wino_g2", lambda: kernels.conv2d_wino_nhwc(x, pw[2])),
                     ("wino_c32", lambda: kernels.conv2d_wino_nhwc(x, pw[4]))]:
        for _ in range(3):
            outs[name] = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row[name] = {"us": round(us, 1), "algorithmic_TFLOPs": round(flops / us / 1e6, 1)}
    row["max_diff_wino_vs_direct"] = float((outs["wino_g1"] - outs["direct_64x64"]).abs().max() / outs["direct_64x64"].abs().max())
    res.append(row)
    print(json.dumps(row), flush=True)
