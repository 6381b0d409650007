"""Calibration run for oracle/survivor_flips deltas: GPU pipeline vs CPU oracle on pool frames, both cls biases; prints score / box
differences and every flip root.  python tools/calib_flips.py [n_frames]"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hvpr_amd import detector, synthetic, synthetic_weights
from hvpr_amd.config import hvpr_car_cfg
from oracle import hvpr_oracle as O, survivor_flips as SF

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = hvpr_car_cfg()
for bias in (-2.0, -4.59511985013459):
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
    params = synthetic_weights.load_synthetic(model, seed=0, cls_bias=bias)
    model = model.to("cuda:0").eval()
    for i in range(n):
        f = synthetic.hvpr_frame(i)
        pts = np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)
        with torch.no_grad():
            preds, _, bd = model({"points": torch.from_numpy(pts).cuda(), "batch_size": 1})
        ref, inter = O.forward_frames([f], params, O.cfg_from_model_cfg(cfg))
        sg, bg = bd["batch_max_scores"][0].cpu().numpy(), bd["batch_box_preds"][0].cpu().numpy()
        sc = torch.sigmoid(inter["batch_cls_preds"][0]).max(-1)[0].numpy()
        bc = inter["batch_box_preds"][0].numpy()
        r = SF.explain(sg, bg, sc, bc, 0.1, 0.1, 4096, 500, 1e-4, 1e-2)
        cand = sg >= 0.09
        print(json.dumps({"bias": bias, "frame": i, "score_absdiff": float(np.abs(sg - sc).max()), "box_absdiff_xyzlwh": float(np.abs(bg[cand, :6] - bc[cand, :6]).max()),
                          "surv": [r["survivors_a"], r["survivors_b"], r["common"]], "flips": len(r["flips"]), "unexpl": len(r["unexplained"]),
                          "roots": r["roots"]}), flush=True)
