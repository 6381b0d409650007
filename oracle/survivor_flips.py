"""ORACLE — TEST INFRASTRUCTURE ONLY: why do two runs of the post-processing keep different boxes?

Post-processing (detector3d_template.py:168-274, model_nms_utils.py:6-25) is a chain of hard decisions — score >= SCORE_THRESH,
the NMS_PRE_MAXSIZE cut of the score order, greedy suppression at IoU > NMS_THRESH, the NMS_POST_MAXSIZE cut — so two fp32
pipelines whose scores and boxes agree to round-off may still return different survivor sets: ONE decision whose quantity lies
within round-off of its threshold flips, and the greedy sweep carries the flip on (a box that survives on one side suppresses
others there).  `explain` takes the two sides' per-anchor scores and boxes of one frame and traces EVERY id of the symmetric
difference of the survivor sets back, through that cascade, to a root decision, and reports for each root the quantity on both
sides.  A flip is *explained* when its root quantity differs between the sides by no more than the given delta AND the two
values lie on different sides of the threshold (or, for an order swap, the two scores are within delta of each other).

Used by tests/test_gpu_e2e.py and bench.py's parity gates; nothing under hvpr_amd/ imports it.
"""
import numpy as np

from . import hvpr_oracle as O


# The deltas the end-to-end tests and bench.py's parity gates use (fp32 pipelines, hvpr_car, synthetic weights).  Calibrated on
# MI355X against the CPU oracle, 8 pool frames x 2 cls biases (tools/calib_flips.py, round 6): every one of the 243 flips observed was
# an ORDER SWAP of two overlapping candidates whose scores differ by <= 5.4e-7 on either side (the synthetic head gives anchors in
# featureless regions scores that agree to a few ulp); no IoU-threshold or score-threshold root occurred.  Scores differ by <= 1.3e-5
# over all anchors, box parameters by <= 7.7e-6 m, which bounds the IoU of a pair to ~8 * 7.7e-6 / 1.6 = 4e-5.
DELTA_SCORE = 2e-6
DELTA_IOU = 1e-4


class _Side:
    """One pipeline's post-processing of one frame, with the bookkeeping the explanation needs."""

    def __init__(self, scores, boxes, score_thresh, nms_thresh, pre_max, post_max):
        self.s = np.asarray(scores, dtype=np.float32).reshape(-1)
        self.b = np.ascontiguousarray(np.asarray(boxes, dtype=np.float32)[:, :7])
        self.score_thresh, self.nms_thresh, self.pre_max, self.post_max = np.float32(score_thresh), float(nms_thresh), pre_max, post_max
        passing = np.nonzero(self.s >= self.score_thresh)[0]
        order = passing[O.stable_order_desc(self.s[passing])]               # anchor ids, descending score, ascending id on ties
        self.cand = order[:pre_max]
        self.cut_score = float(self.s[order[pre_max]]) if len(order) > pre_max else None   # best score that the pre-max cut dropped
        self.last_in = float(self.s[self.cand[-1]]) if len(self.cand) else None
        self.pos = {int(a): i for i, a in enumerate(self.cand)}             # anchor id -> position in the sweep order
        kept_pos = O.nms_sorted(self.b[self.cand], nms_thresh) if len(self.cand) else np.zeros((0,), np.int64)
        self.kept = self.cand[kept_pos]                                       # every box the sweep keeps (before the post-max cut)
        self.kept_rank = {int(a): r for r, a in enumerate(self.kept)}
        self.survivors = self.kept[:post_max]
        self._iou_to_kept = None

    def iou_to_kept(self):
        if self._iou_to_kept is None:
            self._iou_to_kept = O.boxes_iou_bev(self.b[self.cand], self.b[self.kept]) if len(self.kept) else np.zeros((len(self.cand), 0), np.float32)
        return self._iou_to_kept

    def suppressor(self, a):
        """The kept box that suppresses candidate `a` in the greedy sweep: the first kept box in front of it with IoU > thresh."""
        i = self.pos[a]
        row = self.iou_to_kept()[i]
        for r, k in enumerate(self.kept):
            if self.pos[int(k)] >= i:
                break
            if row[r] > self.nms_thresh:
                return int(k), float(row[r])
        raise AssertionError(f"candidate {a} is neither kept nor suppressed")

    def status(self, a):
        if a in self.kept_rank:
            return "survivor" if self.kept_rank[a] < self.post_max else "beyond_post_max"
        if a in self.pos:
            return "suppressed"
        return "below_score_thresh" if self.s[a] < self.score_thresh else "beyond_pre_max"

    def iou(self, a, b):
        return float(O.boxes_iou_bev(self.b[a:a + 1], self.b[b:b + 1])[0, 0])


def explain(scores_a, boxes_a, scores_b, boxes_b, score_thresh, nms_thresh, pre_max, post_max, delta_score, delta_iou):
    """Returns {"survivors_a", "survivors_b", "common", "flips": [...], "roots": [...], "unexplained": [...]}: every id kept by
    exactly one side, the chain it hangs on and the root decision (kind, ids, the quantity on both sides, threshold)."""
    A = _Side(scores_a, boxes_a, score_thresh, nms_thresh, pre_max, post_max)
    B = _Side(scores_b, boxes_b, score_thresh, nms_thresh, pre_max, post_max)
    sa, sb = set(int(x) for x in A.survivors), set(int(x) for x in B.survivors)
    memo, roots = {}, {}

    def root(kind, ids, qa, qb, thr, ok):
        key = (kind,) + tuple(ids)
        roots[key] = {"kind": kind, "ids": list(ids), "a": qa, "b": qb, "threshold": thr, "within_delta": bool(ok)}
        return key

    def why(x, visiting):
        """Root key of the reason the two sides treat x differently in the sweep (kept vs not), or None when unexplained."""
        if x in memo:
            return memo[x]
        if x in visiting:
            return None
        visiting = visiting | {x}
        ka, kb = x in A.kept_rank, x in B.kept_rank
        assert ka != kb
        K, N = (A, B) if ka else (B, A)             # K keeps x in its sweep, N does not
        st = N.status(x)
        if st == "below_score_thresh":
            qa, qb = float(A.s[x]), float(B.s[x])
            r = root("score_thresh", (x,), qa, qb, float(K.score_thresh), abs(qa - qb) <= delta_score)
        elif st == "beyond_pre_max":
            qa, qb = float(A.s[x]), float(B.s[x])
            r = root("pre_max_cut", (x,), qa, qb, N.last_in, abs(float(N.s[x]) - N.last_in) <= delta_score)
        else:                                        # suppressed on N by y
            y, iou_n = N.suppressor(x)
            if y in K.pos and K.pos[y] > K.pos[x]:   # on K, y comes AFTER x (there x may even suppress y): two near-equal scores in swapped order
                d = max(abs(float(A.s[x]) - float(A.s[y])), abs(float(B.s[x]) - float(B.s[y])))
                r = root("order_swap", (y, x), float(A.s[y]) - float(A.s[x]), float(B.s[y]) - float(B.s[x]), 0.0, d <= delta_score)
            elif y not in K.kept_rank:               # y itself is treated differently: the flip of x hangs on the flip of y
                r = why(y, visiting)
            else:                                    # same order, the IoU of the pair is on different sides of the threshold
                iou_k = K.iou(y, x)
                qa, qb = (iou_k, iou_n) if K is A else (iou_n, iou_k)
                r = root("iou_thresh", (y, x), qa, qb, float(N.nms_thresh), abs(qa - qb) <= delta_iou and iou_k <= N.nms_thresh < iou_n)
        memo[x] = r
        return r

    flips, unexplained = [], []
    sweep_diff = sorted(set(A.kept_rank) ^ set(B.kept_rank), key=lambda a: min(A.pos.get(a, 1 << 30), B.pos.get(a, 1 << 30)))
    for x in sorted(sa ^ sb):
        in_a = x in sa
        if (x in A.kept_rank) != (x in B.kept_rank):
            r = why(x, frozenset())
            how = "sweep"
        else:
            # kept by both sweeps, but its rank among the kept boxes straddles the post-max cut: an earlier sweep difference
            # (or an order swap among kept boxes) moved it; the earliest sweep difference in front of it is the cause
            earlier = [d for d in sweep_diff if min(A.pos.get(d, 1 << 30), B.pos.get(d, 1 << 30)) < max(A.pos[x], B.pos[x])]
            r = why(earlier[0], frozenset()) if earlier else None
            how = "post_max_cut"
        rec = {"id": x, "kept_by": "a" if in_a else "b", "via": how, "root": None if r is None else roots[r]}
        flips.append(rec)
        if r is None or not roots[r]["within_delta"]:
            unexplained.append(rec)
    used = {}
    for f in flips:
        if f["root"] is not None:
            used[(f["root"]["kind"],) + tuple(f["root"]["ids"])] = f["root"]
    return {"survivors_a": len(sa), "survivors_b": len(sb), "common": len(sa & sb), "flips": flips, "roots": list(used.values()),
            "unexplained": unexplained, "delta_score": delta_score, "delta_iou": delta_iou}
