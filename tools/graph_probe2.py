import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import detector, synthetic, synthetic_weights
from hvpr_amd.config import hvpr_car_cfg
cfg = hvpr_car_cfg()
model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
synthetic_weights.load_synthetic(model, seed=0, cls_bias=-4.595)
model = model.cuda().eval()
bs = []
for s in range(4):
    f = synthetic.hvpr_frame(s)
    pts = np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)
    bs.append({"points": torch.from_numpy(pts).cuda(), "point_frame_offsets": torch.tensor([0, len(f)], dtype=torch.int32, device="cuda"), "batch_size": 1})
g = detector.GraphedForward(model, bs[0])
torch.cuda.synchronize()
bad = 0
for i in range(40):
    t0 = time.perf_counter(); out = g(bs[i % 4]); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    kept = int(out[0][0]["pred_count"]); vo = out[2]["voxel_offsets"].tolist()
    if kept != 500 or ms > 12:
        bad += 1
        bd = out[2]
        print("BAD iter", i, "ms", round(ms, 1), "kept", kept, "vo", vo, "num max", int(bd["voxel_num_points"].max()),
              "pf nan", int(torch.isnan(bd["pillar_features"]).sum()), "sf2d max", float(bd["spatial_features_2d"].max()),
              "scores>=.1", int((bd["batch_max_scores"] >= 0.1).sum()))
print("bad", bad, "of 40")
