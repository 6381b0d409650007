"""f4 of SURVEY.md §8f: CPU natives of the GT-sampling augmentation (libhvpr_cpu.so) against independent restatements."""
import ctypes
import os
import re

import numpy as np

from hvpr_amd import gt_sampling as G
from oracle import hvpr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _boxes(rng, n):
    return np.concatenate([rng.uniform(-10, 10, (n, 3)), rng.uniform(0.5, 5, (n, 3)), rng.uniform(-4, 4, (n, 1))], 1).astype(np.float32)


def test_library_exports_the_header_symbols():
    hdr = open(os.path.join(ROOT, "include", "hvpr_cpu.h")).read()
    names = set(re.findall(r"\b(hvpr_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(os.path.join(ROOT, "hvpr_amd", "libhvpr_cpu.so"))
    assert names == {"hvpr_points_in_boxes_cpu", "hvpr_boxes_bev_iou_cpu"}
    for n in names:
        assert hasattr(lib, n)


def test_boxes_bev_iou_cpu_matches_the_oracle_geometry():
    rng = np.random.default_rng(0)
    a, b = _boxes(rng, 60), _boxes(rng, 45)
    got = G.boxes_bev_iou_cpu(a, b)
    np.testing.assert_allclose(got, O.boxes_iou_bev(a, b), atol=2e-6)            # same published algorithm, fp32
    np.testing.assert_allclose(np.diag(G.boxes_bev_iou_cpu(a, a)), 1.0, atol=1e-5)
    np.testing.assert_allclose(got, G.boxes_bev_iou_cpu(b, a).T, atol=1e-5)
    # axis-aligned known answer: 2x2 squares offset by 1 -> inter 1, union 7
    sq = np.array([[0, 0, 0, 2, 2, 1, 0], [1, 1, 0, 2, 2, 1, 0], [1, 1, 0, 2, 2, 1, np.pi / 2]], np.float32)
    np.testing.assert_allclose(G.boxes_bev_iou_cpu(sq[:1], sq[1:]), [[1 / 7, 1 / 7]], atol=1e-6)
    assert G.boxes_bev_iou_cpu(a[:0], b).shape == (0, 45)


def test_points_in_boxes_cpu_and_remove_points():
    rng = np.random.default_rng(1)
    boxes = _boxes(rng, 12)
    pts = rng.uniform(-12, 12, (5000, 4)).astype(np.float32)
    got = G.points_in_boxes_cpu(pts[:, :3], boxes)
    # independent numpy restatement
    d = pts[None, :, :3] - boxes[:, None, :3]
    c, s = np.cos(-boxes[:, 6])[:, None], np.sin(-boxes[:, 6])[:, None]
    lx, ly = d[..., 0] * c - d[..., 1] * s, d[..., 0] * s + d[..., 1] * c
    want = (np.abs(d[..., 2]) <= boxes[:, None, 5] / 2) & (np.abs(lx) < boxes[:, None, 3] / 2) & (np.abs(ly) < boxes[:, None, 4] / 2)
    assert got.shape == (12, 5000) and got.dtype == np.int32
    assert (got.astype(bool) != want).sum() <= 2                                  # fp32 rounding of points on a face
    assert got.sum() > 50
    kept = G.remove_points_in_boxes3d(pts, boxes)
    assert len(kept) == int((got.sum(axis=0) == 0).sum()) and G.points_in_boxes_cpu(kept[:, :3], boxes).sum() == 0
    # centre in, far point out, point on the z face in (<=)
    one = np.array([[0, 0, 0, 2, 1, 1, 0.3]], np.float32)
    p = np.array([[0, 0, 0], [5, 0, 0], [0, 0, 0.5], [0, 0, 0.5001]], np.float32)
    assert G.points_in_boxes_cpu(p, one).tolist() == [[1, 0, 1, 0]]
