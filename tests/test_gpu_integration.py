"""GPU: the rows around the hot path composed — host frame -> pinned upload (f1) -> range mask + sample_points on the device (f2)
-> detector (a1..a8) -> KITTI prediction dicts -> AP evaluator (f3)."""
import numpy as np
import pytest
import torch

from hvpr_amd import detector, kitti_eval, preprocess, synthetic, synthetic_weights
from hvpr_amd.config import hvpr_car_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CALIB = {"P2": np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]], np.float32),
         "R0": np.eye(3, dtype=np.float32),
         "Tr_velo2cam": np.array([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27]], np.float32)}


class _Seeded:
    def __init__(self, seed):
        self.r = np.random.RandomState(seed)

    def choice(self, *a, **k):
        return self.r.choice(*a, **k)

    def shuffle(self, x):
        return self.r.shuffle(x)


def test_raw_frames_to_average_precision():
    cfg = hvpr_car_cfg()
    model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
    synthetic_weights.load_synthetic(model, seed=0, cls_bias=-2.0)
    model = model.to(DEV).eval()
    pre = preprocess.PointPreprocessor(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, num_points=16384, fov_points_only=False)
    pipe = preprocess.InputPipeline(max_points=40000, n_feat=4, max_batch=1, device=DEV)
    raws = [synthetic.kitti_like_frame(80 + i) for i in range(3)]             # ~20 k raw points, some outside the range
    pipe.put([raws[0]])
    gts, dts = [], []
    for i, raw in enumerate(raws):
        if i + 1 < len(raws):
            pipe.put([raws[i + 1]])                                            # next upload overlaps this frame's compute
        bd, slot = pipe.get()
        pts = pre(bd["points"][:, 1:].contiguous(), rng=_Seeded(i))            # range mask + sample_points (16384 rows)
        pipe.release(slot)
        assert pts.shape == (16384, 4)
        # the device pre-processing equals the host recipe
        m = (raw[:, 0] >= 0) & (raw[:, 0] <= 47.36) & (raw[:, 1] >= -19.84) & (raw[:, 1] <= 19.84)
        near = (np.sqrt((raw[m][:, 0] ** 2 + raw[m][:, 1] ** 2).astype(np.float32) + raw[m][:, 2] ** 2) < 40).astype(np.uint8)
        choice = preprocess.sample_points_choice(near, 16384, _Seeded(i))
        np.testing.assert_array_equal(pts.cpu().numpy(), raw[m][choice])
        batch = {"points": torch.cat([torch.zeros((16384, 1), device=DEV), pts], 1), "batch_size": 1}
        with torch.no_grad():
            preds, _, _ = model(batch)
        annos = kitti_eval.generate_prediction_dicts({"calib": [CALIB], "image_shape": [np.array([375, 1242])], "frame_id": ["%06d" % i]},
                                                     preds, cfg.CLASS_NAMES)
        dt = annos[0]
        assert len(dt["name"]) > 10 and set(dt["name"]) == {"Car"}
        # ground truth = the confident half of the detections (easy: not occluded, not truncated) -> AP must be 100 for them
        keep = dt["score"] >= np.median(dt["score"])
        big = (dt["bbox"][:, 3] - dt["bbox"][:, 1]) > 45
        sel = keep & big
        gts.append({"name": dt["name"][sel], "truncated": np.zeros(sel.sum()), "occluded": np.zeros(sel.sum(), np.int64), "alpha": dt["alpha"][sel],
                    "bbox": dt["bbox"][sel], "dimensions": dt["dimensions"][sel], "location": dt["location"][sel],
                    "rotation_y": dt["rotation_y"][sel]})
        dts.append({k: (v[sel] if k != "frame_id" else v) for k, v in dt.items()})
    assert sum(len(g["name"]) for g in gts) > 5
    text, ret = kitti_eval.get_official_eval_result(gts, dts, ["Car"])
    # BEV NMS keeps the detections apart in the ground plane, so every ground-truth box has exactly one twin in BEV and 3-D
    # (their 2-D image boxes may overlap: the image AP is not asserted)
    for k in ("Car_3d/easy_R40", "Car_3d/moderate_R40", "Car_3d/hard_R40", "Car_bev/easy_R40", "Car_bev/hard_R40"):
        assert ret[k] > 99.9, (k, ret[k], text)
    assert 0 < ret["Car_image/easy_R40"] <= 100
