"""Frames/s of the frame pipeline (detector.PipelinedForward) when a step carries B frames through every stage (B = 1: bench.py's
`value` configuration).  python tools/bench_pipe_batch.py [B ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import detector, synthetic, synthetic_weights  # noqa: E402
from hvpr_amd.config import hvpr_car_cfg  # noqa: E402

DEV = torch.device("cuda", 0)


def batch_of(frames):
    pts = np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)])
    offs = np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)
    return {"points": torch.from_numpy(pts).to(DEV), "point_frame_offsets": torch.from_numpy(offs).to(DEV), "batch_size": len(frames)}


def main():
    cfg = hvpr_car_cfg()
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
    synthetic_weights.load_synthetic(model, seed=0, cls_bias=-4.59511985013459)
    model = model.to(DEV).eval()
    frames = [synthetic.hvpr_frame(i) for i in range(8)]
    for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
        batches = [batch_of([frames[(i * B + j) % 8] for j in range(B)]) for i in range(8)]
        with torch.no_grad():
            pipe = detector.PipelinedForward(model, batches[0])
            for i in range(20):
                pipe(batches[i % 8])
            torch.cuda.synchronize()
            steps = 200 // B
            t0 = time.perf_counter()
            for i in range(steps):
                pipe(batches[i % 8])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            for _ in pipe.flush():
                pass
        print(f"B={B}: {steps * B / dt:.1f} frames/s, {1e3 * dt / steps:.3f} ms per step of {B} frame(s)", flush=True)
        del pipe
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
