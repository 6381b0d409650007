"""Encode group (voxelize + VFE + memory read-out + scatter) alone, as one hipGraph, on two workloads: hvpr_car batch 1 and the
dense scene of SURVEY.md §8d config 5 (200 k points / frame, 512 x 512 grid, batch 4).  Prints achieved algorithmic GB/s."""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import detector, synthetic, synthetic_weights  # noqa: E402
from hvpr_amd.config import hvpr_car_cfg  # noqa: E402

DEV = "cuda:0"


def batch_of(frames):
    pts = np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)])
    offs = np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)
    return {"points": torch.from_numpy(pts).to(DEV), "point_frame_offsets": torch.from_numpy(offs).to(DEV), "batch_size": len(frames)}


def run(name, cfg, frames, iters=50):
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
    synthetic_weights.load_synthetic(model, seed=0)
    model = model.to(DEV).eval()
    b = batch_of(frames)
    if "--dense-clear" not in sys.argv:
        b.update(model.persistent_canvases(b))     # the canvases stay ours: only the previous frame's cells are cleared
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(2):
            bd = model.stage_encode(dict(b))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        bd = model.stage_encode(dict(b))
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    B, _, ny, nx = bd["spatial_features"].shape
    n_pts = sum(len(f) for f in frames)
    w_bytes = 4 * (16 * 10 + 16 + 64 * 32 + 64 + 16 * 5 + 16 + 32 * 16 + 32) + 2000 * 64 * 4
    alg = 16 * n_pts + 4 * 160 * nx * ny * B + w_bytes
    print(json.dumps({"workload": name, "points": n_pts, "pillars": int(bd["voxel_offsets"][-1]), "grid": [nx, ny], "batch": B,
                      "group_us": round(us, 1), "algorithmic_bytes": alg, "achieved_GBps": round(alg / us / 1e3, 1),
                      "frac_of_8TBps": round(alg / us / 1e3 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    car = hvpr_car_cfg()
    if "--only16" not in sys.argv and "--only-dense" not in sys.argv:
        run("hvpr_car batch 1", car, [synthetic.hvpr_frame(0)])
    if "--car1" in sys.argv:
        sys.exit(0)
    if "--only-dense" not in sys.argv:
        run("hvpr_car batch 16", car, [synthetic.hvpr_frame(i) for i in range(16)])
    if "--only16" in sys.argv:
        sys.exit(0)
    dense = copy.deepcopy(car)
    rng = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    dense.DATA_CONFIG.POINT_CLOUD_RANGE = rng
    for p in dense.DATA_CONFIG.DATA_PROCESSOR:
        if p.NAME == "transform_points_to_voxels":
            p.VOXEL_SIZE, p.MAX_POINTS_PER_VOXEL, p.MAX_NUMBER_OF_VOXELS = [0.2, 0.2, 8.0], 20, {"train": 60000, "test": 60000}
    run("dense scene 200k pts, 512x512, batch 4 (config 5)", dense, [synthetic.uniform_frame(70 + b, 200000, rng) for b in range(4)])
