"""An independent float64 check of the rotated-BEV geometry: exact convex-polygon intersection by Sutherland-Hodgman clipping.
Nothing here is shared with oracle/iou3d_nms_ref.c or csrc/iou3d_nms.hip (those restate the published routine of OpenPCDet's
iou3d_nms, absent from the reference snapshot: crossing tests + in-box tests with a 1e-2 margin + a fan area around the centroid)."""
import numpy as np


def corners(box):
    """box [x, y, z, dx, dy, dz, heading] -> (4, 2) float64 corners, counter-clockwise."""
    x, y, dx, dy, r = float(box[0]), float(box[1]), float(box[3]), float(box[4]), float(box[6])
    c, s = np.cos(r), np.sin(r)
    loc = np.array([[dx / 2, dy / 2], [-dx / 2, dy / 2], [-dx / 2, -dy / 2], [dx / 2, -dy / 2]], np.float64)
    rot = np.array([[c, -s], [s, c]], np.float64)
    return loc @ rot.T + np.array([x, y], np.float64)


def _clip(poly, a, b):
    """Keep the part of `poly` on the left of the directed edge a -> b."""
    out = []
    n = len(poly)
    for i in range(n):
        p, q = poly[i], poly[(i + 1) % n]
        sp = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        sq = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0):
            t = sp / (sp - sq)
            out.append(p + t * (q - p))
    return out


def intersection_area(box_a, box_b):
    poly = list(corners(box_a))
    cb = corners(box_b)
    for i in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, cb[i], cb[(i + 1) % 4])
    if len(poly) < 3:
        return 0.0
    p = np.array(poly)
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def iou_bev(box_a, box_b):
    inter = intersection_area(box_a, box_b)
    sa, sb = float(box_a[3]) * float(box_a[4]), float(box_b[3]) * float(box_b[4])
    return inter / max(sa + sb - inter, 1e-8)


def random_pairs(n, seed):
    """n pairs of car / pedestrian / cyclist sized boxes at general headings, centres close enough that about half overlap."""
    rng = np.random.default_rng(seed)
    sizes = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], np.float64)

    def boxes(centre):
        k = rng.integers(0, 3, n)
        b = np.zeros((n, 7), np.float64)
        b[:, :2] = centre
        b[:, 2] = rng.uniform(-1.5, -0.5, n)
        b[:, 3:6] = sizes[k] * rng.uniform(0.8, 1.25, (n, 3))
        b[:, 6] = rng.uniform(-np.pi, np.pi, n)
        return b
    ca = rng.uniform(-20, 20, (n, 2))
    a = boxes(ca)
    b = boxes(ca + rng.normal(0, 1.2, (n, 2)))
    return a.astype(np.float32), b.astype(np.float32)
