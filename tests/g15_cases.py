"""Reader of fixture G15 (tests/golden/g15_post_processing.npz; generator: make_golden.g15_post_processing) shared by the CPU
test (oracle) and the GPU test (product).  Equal scores are put into this build's tie order — descending score, ascending id
 — before anything is compared; torch leaves that order unspecified and the fixture's tie frames are built so that
the reference's result does not depend on it as a set."""
import os

import numpy as np

TAGS = ("car", "car_normalized_raw", "car_yaml_sizes", "three_agnostic", "three_agnostic_raw", "three_multi")
SCORE_THRESH = 0.1
RECALL_THRESH_LIST = [0.3, 0.5, 0.7]


class Case:
    def __init__(self, z, tag):
        self.tag = tag
        self.cls, self.boxes, self.gt_boxes = z[tag + ".cls"], z[tag + ".boxes"], z[tag + ".gt_boxes"]
        self.num_class, multi, raw, norm, self.pre, self.post = (int(v) for v in z[tag + ".cfg"])
        self.multi, self.raw, self.normalized = bool(multi), bool(raw), bool(norm)
        self.nms_thresh = float(z[tag + ".nms_thresh"])
        self.recall = {str(k): int(v) for k, v in zip(z[tag + ".recall_keys"], z[tag + ".recall_values"])}
        self.frames = []
        for b in range(self.cls.shape[0]):
            f = {k: z[f"{tag}.f{b}.{k}"] for k in ("pred_boxes", "pred_scores", "pred_labels")}
            if self.multi:
                f.update({k: z[f"{tag}.f{b}.{k}"] for k in ("mc_scores", "mc_labels", "mc_boxes")})
            else:
                sel, ss = z[f"{tag}.f{b}.selected"], z[f"{tag}.f{b}.selected_scores"]
                perm = np.lexsort((sel, -ss.astype(np.float64)))          # the build's tie order
                # post_processing ran the same chain: its rows are in `selected`'s order (scores may be the raw ones)
                np.testing.assert_array_equal(f["pred_boxes"], self.boxes[b][sel])
                f = {k: v[perm] for k, v in f.items()}
                f["selected"], f["selected_scores"] = sel[perm], ss[perm]
            self.frames.append(f)

    def kwargs(self):
        return dict(normalized=self.normalized, raw_score=self.raw, multi_classes=self.multi)


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "g15_post_processing.npz"), allow_pickle=False)
    return [Case(z, t) for t in TAGS]
