"""Seeded synthetic inputs (numpy only): KITTI-like LiDAR frames and stress clouds.

No dataset can be downloaded here, so tests and bench.py use these generators (recipe: SURVEY.md §8d).
The pre-processing mirrors the reference data path: mask_points_by_range (pcdet/utils/common_utils.py:59-62)
then sample_points to 16384 (pcdet/datasets/processor/data_processor.py:77-108).
"""
import numpy as np

HVPR_RANGE = [0.0, -19.84, -2.5, 47.36, 19.84, 0.5]
HVPR_VOXEL = [0.16, 0.16, 3.0]


def kitti_like_frame(seed, n_az=330, n_beams=64):
    """64 beams x n_az azimuth steps over +-45 deg; ground plane 1.73 m below the sensor clipped by 25 random
    obstacle sectors; 2 cm noise; intensity U[0,1).  Returns (n, 4) float32 [x, y, z, r]."""
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    az = np.deg2rad(np.linspace(-45.0, 45.0, n_az))
    h = 1.73
    with np.errstate(divide="ignore"):
        ground = np.where(el < 0, h / -np.sin(el), np.inf)          # range of the ground hit per beam
    rng_max = np.full((n_az,), np.inf)
    for _ in range(25):
        a0 = rng.integers(0, n_az)
        wdt = rng.integers(5, 41)
        rng_max[a0:a0 + wdt] = np.minimum(rng_max[a0:a0 + wdt], rng.uniform(5.0, 45.0))
    r = np.minimum(ground[:, None], rng_max[None, :])                 # (beams, az)
    ok = np.isfinite(r)
    E, A = np.meshgrid(el, az, indexing="ij")
    x = r * np.cos(E) * np.cos(A)
    y = r * np.cos(E) * np.sin(A)
    z = r * np.sin(E)
    pts = np.stack([x[ok], y[ok], z[ok]], axis=1)
    pts += rng.normal(0.0, 0.02, pts.shape)
    inten = rng.uniform(0.0, 1.0, (pts.shape[0], 1))
    return np.concatenate([pts, inten], axis=1).astype(np.float32)


def mask_points_by_range(points, limit_range):
    """common_utils.py:59-62 — inclusive x/y mask, z untouched."""
    m = (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) & \
        (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])
    return points[m]


def sample_points(points, num_points, rng):
    """data_processor.py:77-108 restated with a seeded Generator (the reference uses the global np.random)."""
    n = len(points)
    if num_points == -1:
        return points
    if num_points < n:
        depth = np.linalg.norm(points[:, :3], axis=1)
        near = np.where(depth < 40.0)[0]
        far = np.where(depth >= 40.0)[0]
        if num_points > len(far):
            choice = np.concatenate([rng.choice(near, num_points - len(far), replace=False), far]) if len(far) else \
                rng.choice(near, num_points, replace=False)
        else:
            choice = rng.choice(np.arange(n), num_points, replace=False)
        rng.shuffle(choice)
    else:
        choice = np.arange(n)
        if num_points > n:
            extra = rng.choice(choice, num_points - n, replace=(num_points - n > n))
            choice = np.concatenate([choice, extra])
        rng.shuffle(choice)
    return points[choice]


def hvpr_frame(seed, num_points=16384, shuffle=False, n_az=330):
    """A frame as the model sees it: generate -> range mask -> sample/pad to num_points [-> shuffle]."""
    rng = np.random.default_rng(seed + 7919)
    pts = mask_points_by_range(kitti_like_frame(seed, n_az=n_az), HVPR_RANGE)
    pts = sample_points(pts, num_points, rng)
    if shuffle:
        pts = pts[rng.permutation(len(pts))]
    return np.ascontiguousarray(pts, dtype=np.float32)


def uniform_frame(seed, n, point_cloud_range, n_feat=4):
    rng = np.random.default_rng(seed)
    lo = np.asarray(point_cloud_range[:3], dtype=np.float64)
    hi = np.asarray(point_cloud_range[3:6], dtype=np.float64)
    xyz = rng.uniform(lo, hi, (n, 3))
    extra = rng.uniform(0.0, 1.0, (n, n_feat - 3))
    return np.concatenate([xyz, extra], axis=1).astype(np.float32)
