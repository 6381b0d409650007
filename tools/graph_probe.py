import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hvpr_amd import detector, synthetic, synthetic_weights
from hvpr_amd.config import hvpr_car_cfg
cfg = hvpr_car_cfg()
model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
synthetic_weights.load_synthetic(model, seed=0, cls_bias=-4.595)
model = model.cuda().eval()
f = synthetic.hvpr_frame(0)
pts = np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)
b = {"points": torch.from_numpy(pts).cuda(), "point_frame_offsets": torch.tensor([0, len(f)], dtype=torch.int32, device="cuda"), "batch_size": 1}
g = detector.GraphedForward(model, b)
torch.cuda.synchronize()
for name, fn in (("replay only", lambda: g.graph.replay()), ("call", lambda: g(b))):
    t0 = time.perf_counter()
    for _ in range(20): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, "host per iter ms", (t1 - t0) / 20 * 1e3, "total per iter ms", (t2 - t0) / 20 * 1e3)
bs = []
for s in range(4):
    f = synthetic.hvpr_frame(s)
    pts = np.concatenate([np.zeros((len(f), 1), np.float32), f], 1)
    bs.append({"points": torch.from_numpy(pts).cuda(), "point_frame_offsets": torch.tensor([0, len(f)], dtype=torch.int32, device="cuda"), "batch_size": 1})
torch.cuda.synchronize()
for i in range(8):
    t0 = time.perf_counter(); out = g(bs[i % 4]); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("frame", i % 4, "ms", (t1 - t0) * 1e3, "pillars", int(out[2]["voxel_offsets"][-1]), "kept", int(out[0][0]["pred_count"]))
bd = out[2]
for k in ("voxel_num_points", "pillar_features", "spatial_features", "spatial_features_2d", "batch_max_scores", "batch_box_preds"):
    t = bd[k].float()
    print(k, tuple(t.shape), "nan", int(torch.isnan(t).sum()), "min", float(t.min()), "max", float(t.max()))
print("voxel_offsets", bd["voxel_offsets"].tolist())
# back to the capture-time frame
t0 = time.perf_counter(); out = g(b); torch.cuda.synchronize(); print("orig frame ms", (time.perf_counter() - t0) * 1e3, "kept", int(out[0][0]["pred_count"]))
