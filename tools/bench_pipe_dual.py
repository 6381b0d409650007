"""Frames/s of R independent frame pipelines (detector.PipelinedForward, one model replica and one HIP stream each) fed in turn with
batch-1 frames: python tools/bench_pipe_dual.py [R ...].  R = 1 is bench.py's `value` configuration."""
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import detector, synthetic, synthetic_weights  # noqa: E402
from hvpr_amd.config import hvpr_car_cfg  # noqa: E402

DEV = torch.device("cuda", 0)


def batch_of(f):
    pts = np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)
    return {"points": torch.from_numpy(pts).to(DEV), "point_frame_offsets": torch.tensor([0, len(f)], dtype=torch.int32, device=DEV), "batch_size": 1}


def main():
    cfg = hvpr_car_cfg()

    def replica():
        m = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
        synthetic_weights.load_synthetic(m, seed=0, cls_bias=-4.59511985013459)
        return m.to(DEV).eval()
    batches = [batch_of(synthetic.hvpr_frame(i)) for i in range(8)]
    for R in [int(a) for a in sys.argv[1:]] or [1, 2]:
        with torch.no_grad():
            models = [replica() for _ in range(R)]
            streams = [torch.cuda.Stream() for _ in range(R)]
            pipes = []
            for m, s in zip(models, streams):
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    pipes.append(detector.PipelinedForward(m, batches[0]))
            torch.cuda.synchronize()

            def step(i):
                with torch.cuda.stream(streams[i % R]):
                    return pipes[i % R](batches[i % 8])
            for i in range(24):
                step(i)
            torch.cuda.synchronize()
            steps = 240
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"R={R}: {steps / dt:.1f} frames/s, {1e3 * dt / steps:.3f} ms per frame", flush=True)
        del pipes, models
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
