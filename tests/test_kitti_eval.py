"""f3 of SURVEY.md §8f: the KITTI AP evaluator against fixture G12 — the reference's own eval.py run in the build container
(numba.jit replaced by the identity, the absent rotate_iou.py by the CPU oracle's rotated intersection)."""
import os

import numpy as np
import pytest

from hvpr_amd import kitti_eval
from make_golden import synthetic_kitti_annos
from oracle import hvpr_oracle as O


def _oracle_intersection(a5, b5):
    def as7(b):
        t = np.zeros((len(b), 7), np.float32)
        t[:, 0:2], t[:, 3:5], t[:, 5], t[:, 6] = b[:, 0:2], b[:, 2:4], 1.0, -b[:, 4]
        return t
    if len(a5) == 0 or len(b5) == 0:
        return np.zeros((len(a5), len(b5)))
    return O.boxes_overlap_bev(as7(a5), as7(b5)).astype(np.float32).astype(np.float64)


@pytest.fixture(scope="module")
def g12(golden_dir):
    return np.load(os.path.join(golden_dir, "g12_kitti_eval.npz"))


def test_official_result_matches_the_reference_eval(g12):
    gts, dts = synthetic_kitti_annos()
    text, ret = kitti_eval.get_official_eval_result(gts, dts, ["Car", "Pedestrian", "Cyclist"], rotated_intersection=_oracle_intersection)
    assert sorted(ret) == [str(k) for k in g12["keys"]]
    for k, v in zip(g12["keys"], g12["values"]):
        assert abs(ret[str(k)] - v) < 1e-6, (k, ret[str(k)], v)
    assert text.splitlines()[:5] == str(g12["text"]).splitlines()[:5]


def test_precision_curves_match_the_reference_eval(g12):
    gts, dts = synthetic_kitti_annos()
    mo = kitti_eval._OVERLAPS[:, :, [0, 1, 2]]
    for metric, key in ((0, "prec_bbox"), (1, "prec_bev"), (2, "prec_3d")):
        r = kitti_eval.eval_class(gts, dts, [0, 1, 2], metric, mo, compute_aos=(metric == 0), rotated_intersection=_oracle_intersection)
        np.testing.assert_allclose(r["precision"], g12[key], rtol=0, atol=1e-9)
        if metric == 0:
            np.testing.assert_allclose(r["orientation"], g12["prec_aos"], rtol=0, atol=1e-9)


def test_thresholds_and_map_helpers():
    th = kitti_eval.get_thresholds(np.array([0.9, 0.8, 0.7, 0.6, 0.5]), 5, num_sample_pts=3)
    assert list(th) == [0.9, 0.8, 0.5]      # sample points 0, 0.5, 1.0: the score whose recall is nearest from below, ties to the earlier one
    p = np.linspace(1, 0, 41)[None]
    assert abs(float(kitti_eval.get_mAP(p)[0]) - 100 * np.mean(np.linspace(1, 0, 11))) < 1e-9
    assert abs(float(kitti_eval.get_mAP_R40(p)[0]) - 100 * np.mean(np.linspace(1, 0, 41)[1:])) < 1e-9


def test_prediction_formatting_round_trip():
    calib = {"P2": np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]], np.float32),
             "R0": np.eye(3, dtype=np.float32),
             "Tr_velo2cam": np.array([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27]], np.float32)}
    boxes = np.array([[20.0, 2.0, -0.9, 3.9, 1.6, 1.56, 0.3], [35.0, -6.0, -0.8, 0.8, 0.6, 1.73, -1.2]], np.float32)
    pred = [{"pred_boxes": boxes, "pred_scores": np.array([0.9, 0.4], np.float32), "pred_labels": np.array([1, 2])}]
    annos = kitti_eval.generate_prediction_dicts({"calib": [calib], "image_shape": [np.array([375, 1242])], "frame_id": ["000007"]}, pred,
                                                 ["Car", "Pedestrian", "Cyclist"])
    a = annos[0]
    assert list(a["name"]) == ["Car", "Pedestrian"] and a["frame_id"] == "000007"
    # lidar x forward / y left / z up -> camera x right / y down / z forward; location is the bottom centre
    np.testing.assert_allclose(a["location"][0], [-2.0, 0.9 + 0.78 - 0.08, 20.0 - 0.27], atol=1e-5)
    np.testing.assert_allclose(a["dimensions"][0], [3.9, 1.56, 1.6], atol=1e-6)          # l, h, w
    np.testing.assert_allclose(a["rotation_y"][0], -0.3 - np.pi / 2, atol=1e-6)
    assert (a["bbox"][:, 0] < a["bbox"][:, 2]).all() and (a["bbox"][:, 1] < a["bbox"][:, 3]).all()
    assert (a["bbox"] >= 0).all() and (a["bbox"][:, 2] <= 1241).all() and (a["bbox"][:, 3] <= 374).all()


@pytest.mark.gpu
def test_hip_intersection_gives_the_same_ap(g12):
    gts, dts = synthetic_kitti_annos()
    _, ret = kitti_eval.get_official_eval_result(gts, dts, ["Car", "Pedestrian", "Cyclist"])
    for k, v in zip(g12["keys"], g12["values"]):
        assert abs(ret[str(k)] - v) < 1e-3, (k, ret[str(k)], v)
