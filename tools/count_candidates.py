import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import detector, synthetic, synthetic_weights
from hvpr_amd.config import hvpr_car_cfg
cfg = hvpr_car_cfg()
model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
for bias in (-4.595, -3.5, -3.0, -2.5, -2.0):
    synthetic_weights.load_synthetic(model, seed=0, cls_bias=bias)
    model = model.to("cuda:0").eval()
    out = []
    for seed in range(3):
        f = synthetic.hvpr_frame(seed)
        pts = np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)
        with torch.no_grad():
            preds, _, bd = model({"points": torch.from_numpy(pts).cuda(), "batch_size": 1})
        out.append((int((bd["batch_max_scores"] >= 0.1).sum()), len(preds[0]["pred_boxes"])))
    print(bias, out)
