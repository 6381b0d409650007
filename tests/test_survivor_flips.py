"""CPU: oracle/survivor_flips.explain — the tool the end-to-end tests and bench.py's parity gates use to turn "the two survivor
sets differ in n ids" into an exact statement (every difference hangs on a root decision whose quantity is within delta of its
threshold).  Here: it finds the root of a round-off-sized cascade, and it does NOT explain a real difference."""
import numpy as np

from oracle import survivor_flips as SF


def _crowd(seed=5, n=30000):
    rng = np.random.default_rng(seed)
    xy = rng.uniform([0, -19], [47, 19], (n, 2))
    boxes = np.concatenate([xy, np.full((n, 1), -1.0), np.tile([3.9, 1.6, 1.56], (n, 1)) * rng.uniform(0.9, 1.1, (n, 3)),
                            rng.uniform(-3, 3, (n, 1))], 1).astype(np.float32)
    scores = rng.uniform(0.0995, 0.12, n).astype(np.float32)
    return rng, boxes, scores


def test_roundoff_sized_flips_are_traced_to_their_root():
    rng, boxes, scores = _crowd()
    eps = 1e-4
    b2 = (boxes + rng.normal(0, eps, boxes.shape)).astype(np.float32)
    s2 = (scores + rng.normal(0, eps * 0.01, len(scores))).astype(np.float32)
    r = SF.explain(scores, boxes, s2, b2, 0.1, 0.1, 4096, 500, delta_score=eps * 0.05, delta_iou=20 * eps)
    assert r["survivors_a"] > 200 and len(r["flips"]) >= 2 and r["unexplained"] == []
    assert r["common"] + sum(f["kept_by"] == "a" for f in r["flips"]) == r["survivors_a"]
    assert r["common"] + sum(f["kept_by"] == "b" for f in r["flips"]) == r["survivors_b"]
    for root in r["roots"]:
        assert root["within_delta"]
        if root["kind"] == "iou_thresh":                       # the pair's IoU is on different sides of the threshold, by < delta
            assert min(root["a"], root["b"]) <= 0.1 < max(root["a"], root["b"]) and abs(root["a"] - root["b"]) <= 20 * eps
    # identical inputs: nothing to explain
    r0 = SF.explain(scores, boxes, scores, boxes, 0.1, 0.1, 4096, 500, 1e-6, 1e-6)
    assert r0["flips"] == [] and r0["common"] == r0["survivors_a"] == r0["survivors_b"]


def test_a_real_difference_is_not_explained():
    rng, boxes, scores = _crowd(seed=6)
    r0 = SF.explain(scores, boxes, scores, boxes, 0.1, 0.1, 4096, 500, 1e-6, 1e-6)
    from oracle import hvpr_oracle as O
    sel, _ = O.class_agnostic_nms(scores, boxes, 0.1, 0.1, 4096, 500)
    b2, s2 = boxes.copy(), scores.copy()
    b2[sel[3], 0] += 1.5                                       # a survivor moved by 1.5 m: what it suppresses changes for real
    s2[sel[7]] = 0.05                                          # and one dropped below the score threshold by far
    r = SF.explain(scores, boxes, s2, b2, 0.1, 0.1, 4096, 500, delta_score=1e-6, delta_iou=1e-4)
    assert len(r["flips"]) > 0 and len(r["unexplained"]) > 0
    kinds = {f["root"]["kind"] for f in r["unexplained"] if f["root"]}
    assert "score_thresh" in kinds or "iou_thresh" in kinds


def test_pre_max_and_post_max_cuts():
    rng, boxes, scores = _crowd(seed=7, n=12000)
    s2 = scores.copy()
    from oracle import hvpr_oracle as O
    order = O.stable_order_desc(scores)
    a, b = order[199], order[200]                              # straddle a pre-max cut of 200 and swap them by a hair
    s2[a], s2[b] = scores[b], scores[a]
    d = float(abs(scores[a] - scores[b])) + 1e-9
    r = SF.explain(scores, boxes, s2, boxes, 0.0995, 0.1, 200, 50, delta_score=2 * d, delta_iou=1e-6)
    for f in r["flips"]:
        assert f["root"] is not None
    assert r["unexplained"] == []
