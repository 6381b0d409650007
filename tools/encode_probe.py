"""Kernel experiment: one eager fused-encode call (for the HVPR_EXP_TIMING build: HVPR_AMD_LIB=hvpr_amd/libhvpr_amd_timing.so)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import detector, synthetic, synthetic_weights  # noqa: E402
from hvpr_amd.config import hvpr_car_cfg  # noqa: E402

cfg = hvpr_car_cfg()
model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
synthetic_weights.load_synthetic(model, seed=0)
model = model.to("cuda:0").eval()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # frames per call (> 4: the 8-wave read-out form)
fr = [synthetic.hvpr_frame(i) for i in range(B)]
pts = np.concatenate([np.concatenate([np.full((len(f), 1), i, np.float32), f], 1) for i, f in enumerate(fr)], 0)
off = np.concatenate([[0], np.cumsum([len(f) for f in fr])]).astype(np.int32)
b = {"points": torch.from_numpy(pts).cuda(), "point_frame_offsets": torch.from_numpy(off).cuda(), "batch_size": B}
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        model.stage_encode(dict(b))
        torch.cuda.synchronize()
