"""KITTI object AP evaluation and prediction formatting ("next" row f3 of SURVEY.md §8f).

Restates, as vectorised numpy over per-frame arrays:
  * pcdet/datasets/kitti/kitti_object_eval_python/eval.py — clean_data (:30-83), image_box_overlap (:86-113),
    bev / d3 overlaps (:116-154, over the ABSENT rotate_iou.py), compute_statistics_jit (:157-275), get_thresholds (:9-27),
    eval_class (:448-553), get_mAP / get_mAP_R40 (:556-567), get_official_eval_result (:639-746);
  * KittiDataset.generate_prediction_dicts (pcdet/datasets/kitti/kitti_dataset.py:246-320) with
    box_utils.boxes3d_lidar_to_kitti_camera / boxes3d_kitti_camera_to_imageboxes (pcdet/utils/box_utils.py:152-235).

Metric code, not on the timed path.  The rotated-BEV intersection comes from the same HIP geometry kernel as NMS
(hvpr_boxes_pairwise_f32) unless another `rotated_intersection` callable is passed (tests pass the CPU oracle).
"""
import numpy as np

CLASS_TO_NAME = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting", 5: "Truck"}
_MIN_HEIGHT, _MAX_OCCLUSION, _MAX_TRUNCATION = (40, 25, 25), (0, 1, 2), (0.15, 0.3, 0.5)
_N_SAMPLE = 41
# [overlap set][metric bbox|bev|3d][class]: eval.py:640-645
_OVERLAPS = np.array([[[0.7, 0.5, 0.5, 0.7, 0.5, 0.7]] * 3,
                      [[0.7, 0.5, 0.5, 0.7, 0.5, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5]]])


# ------------------------------------------------------------------------------------------------ overlaps
def hip_rotated_intersection(a5, b5):
    """Intersection areas of rotated rectangles (x, z, l, w, rotation_y) of the camera ground plane, on the device (the NMS
    geometry kernel).  rotation_y turns clockwise in the (x, z) plane (KITTI: about the downward camera y axis), the
    kernel's heading counter-clockwise: the angle is negated.  The reference's own rotate_iou.py is absent (PARITY UNPINNED)."""
    import torch
    from . import kernels

    def as7(b):
        t = np.zeros((len(b), 7), np.float32)
        t[:, 0:2], t[:, 3:5], t[:, 5], t[:, 6] = b[:, 0:2], b[:, 2:4], 1.0, -b[:, 4]
        return torch.from_numpy(t).cuda()
    if len(a5) == 0 or len(b5) == 0:
        return np.zeros((len(a5), len(b5)), np.float64)
    return kernels.boxes_pairwise(as7(np.asarray(a5, np.float32)), as7(np.asarray(b5, np.float32)), 0).cpu().numpy().astype(np.float64)


def image_box_overlap(a, b, criterion=-1):
    """Axis-aligned (x1,y1,x2,y2) overlap matrix; criterion -1 IoU, 0 / area(a), 1 / area(b)."""
    a, b = np.asarray(a, np.float64).reshape(-1, 4), np.asarray(b, np.float64).reshape(-1, 4)
    iw = np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0])
    ih = np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1])
    inter = np.where((iw > 0) & (ih > 0), iw * ih, 0.0)
    area_a = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None]
    area_b = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None, :]
    ua = {-1: area_a + area_b - inter, 0: area_a + 0 * area_b, 1: area_b + 0 * area_a}.get(criterion, np.ones_like(inter))
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(inter > 0, inter / ua, 0.0)


def _bev5(anno):
    return np.concatenate([anno["location"][:, [0, 2]], anno["dimensions"][:, [0, 2]], anno["rotation_y"][:, None]], axis=1)


def frame_overlap(gt, dt, metric, rotated_intersection):
    """(num_dt, num_gt) overlap of one frame — the orientation eval_class indexes (`overlaps[j, i]`, j over detections)."""
    if metric == 0:
        return image_box_overlap(dt["bbox"], gt["bbox"])
    d5, g5 = _bev5(dt), _bev5(gt)
    inter = rotated_intersection(d5, g5)
    area_d, area_g = (d5[:, 2] * d5[:, 3])[:, None], (g5[:, 2] * g5[:, 3])[None, :]
    if metric == 1:
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(inter > 0, inter / (area_d + area_g - inter), 0.0)
    # 3-D: camera y points down, `location` is the bottom centre, dimensions[:, 1] the height (eval.py:128-147)
    dy, dh, gy, gh = dt["location"][:, 1][:, None], dt["dimensions"][:, 1][:, None], gt["location"][:, 1][None, :], gt["dimensions"][:, 1][None, :]
    ih = np.minimum(dy, gy) - np.maximum(dy - dh, gy - gh)
    vol = np.where((inter > 0) & (ih > 0), inter * ih, 0.0)
    vd, vg = np.prod(dt["dimensions"], axis=1)[:, None], np.prod(gt["dimensions"], axis=1)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(vol > 0, vol / (vd + vg - vol), 0.0)


# ------------------------------------------------------------------------------------------------ filtering
def clean_data(gt, dt, cls, difficulty):
    """-> (num_valid_gt, gt_flag, dt_flag, dontcare boxes); flags: 0 evaluate, 1 ignore, -1 other class."""
    name = CLASS_TO_NAME[cls].lower()
    gnames = np.array([n.lower() for n in gt["name"]])
    height = gt["bbox"][:, 3] - gt["bbox"][:, 1] if len(gnames) else np.zeros(0)
    same = gnames == name
    neighbour = ((name == "pedestrian") & (gnames == "person_sitting")) | ((name == "car") & (gnames == "van"))
    hard = (gt["occluded"] > _MAX_OCCLUSION[difficulty]) | (gt["truncated"] > _MAX_TRUNCATION[difficulty]) | (height <= _MIN_HEIGHT[difficulty])
    gflag = np.full(len(gnames), -1, np.int64)
    gflag[neighbour | (same & hard)] = 1
    gflag[same & ~hard] = 0
    dc = gt["bbox"][np.array([n == "DontCare" for n in gt["name"]], bool)] if len(gnames) else np.zeros((0, 4))
    dnames = np.array([n.lower() for n in dt["name"]])
    dheight = np.abs(dt["bbox"][:, 3] - dt["bbox"][:, 1]) if len(dnames) else np.zeros(0)
    dflag = np.where(dheight < _MIN_HEIGHT[difficulty], 1, np.where(dnames == name, 0, -1)).astype(np.int64) if len(dnames) else np.zeros(0, np.int64)
    return int((gflag == 0).sum()), gflag, dflag, np.asarray(dc, np.float64).reshape(-1, 4)


# ------------------------------------------------------------------------------------------------ matching
def _pick(col, cand, dflag, dscore, with_fp):
    """The detection compute_statistics_jit's scan over j ends with for one ground truth (eval.py:199-223), or -1."""
    js = np.nonzero(cand)[0]
    if len(js) == 0:
        return -1
    if not with_fp:                       # threshold pass: highest score, first one on ties (strict '>')
        return int(js[np.argmax(dscore[js])])
    det, max_ov, on_ignored = -1, 0.0, False
    for j in js:                          # counting pass: evaluated detections by strictly larger overlap; an ignored one
        if dflag[j] == 0 and (col[j] > max_ov or on_ignored):        # only while nothing else was found, and any evaluated
            det, max_ov, on_ignored = int(j), col[j], False          # detection after it replaces it
        elif dflag[j] == 1 and det < 0:
            det, on_ignored = int(j), True
    return det


def _match(overlap, gflag, dflag, dscore, min_overlap, thresh, with_fp):
    """One frame: greedy assignment in ground-truth order.  -> (tp, fn, [(gt, det) of every true positive], assigned, low)."""
    nd = len(dflag)
    assigned = np.zeros(nd, bool)
    low = (dscore < thresh) if with_fp else np.zeros(nd, bool)
    tp = fn = 0
    hits = []
    for i in np.nonzero(gflag != -1)[0]:
        cand = (dflag != -1) & ~assigned & ~low & (overlap[:, i] > min_overlap)
        best = _pick(overlap[:, i], cand, dflag, dscore, with_fp)
        if best < 0:
            fn += int(gflag[i] == 0)
            continue
        assigned[best] = True
        if not (gflag[i] == 1 or dflag[best] == 1):
            tp += 1
            hits.append((int(i), best))
    return tp, fn, hits, assigned, low


def _frame_stats(overlap, gt_alpha, dt_alpha, dt_bbox, gflag, dflag, dscore, dc, metric, min_overlap, thresh, compute_aos):
    tp, fn, hits, assigned, low = _match(overlap, gflag, dflag, dscore, min_overlap, thresh, True)
    fp = int((~(assigned | (dflag == -1) | (dflag == 1) | low)).sum())
    if metric == 0 and len(dc):           # detections inside DontCare regions are not false positives (eval.py:247-259)
        ov = image_box_overlap(dt_bbox, dc, 0)
        for i in range(len(dc)):
            free = ~assigned & (dflag == 0) & ~low & (ov[:, i] > min_overlap)
            assigned |= free
            fp -= int(free.sum())
    sim = 0.0
    if compute_aos:
        sim = float(sum((1.0 + np.cos(gt_alpha[i] - dt_alpha[j])) / 2.0 for i, j in hits)) if (tp > 0 or fp > 0) else -1.0
    return tp, fp, fn, sim


def get_thresholds(scores, num_gt, num_sample_pts=_N_SAMPLE):
    scores = np.sort(np.asarray(scores, np.float64))[::-1]
    cur, out = 0.0, []
    for i, s in enumerate(scores):
        l = (i + 1) / num_gt
        r = (i + 2) / num_gt if i < len(scores) - 1 else l
        if (r - cur) < (cur - l) and i < len(scores) - 1:
            continue
        out.append(s)
        cur += 1 / (num_sample_pts - 1.0)
    return np.array(out)


def eval_class(gt_annos, dt_annos, classes, metric, min_overlaps, compute_aos=False, rotated_intersection=hip_rotated_intersection):
    """-> dict(precision, recall, orientation), each [class, difficulty, overlap set, 41]."""
    assert len(gt_annos) == len(dt_annos)
    overlaps = [frame_overlap(g, d, metric, rotated_intersection) for g, d in zip(gt_annos, dt_annos)]
    shape = (len(classes), 3, len(min_overlaps), _N_SAMPLE)
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, cls in enumerate(classes):
        for l in range(3):
            cleaned = [clean_data(g, d, cls, l) for g, d in zip(gt_annos, dt_annos)]
            n_valid = sum(c[0] for c in cleaned)
            for k, mo in enumerate(min_overlaps[:, metric, m]):
                scores = []
                for ov, d, (_, gflag, dflag, _) in zip(overlaps, dt_annos, cleaned):
                    _, _, hits, _, _ = _match(ov, gflag, dflag, d["score"], mo, 0.0, False)
                    scores += [d["score"][j] for _, j in hits]
                th = get_thresholds(np.array(scores), n_valid)
                pr = np.zeros((len(th), 4))
                for ov, g, d, (_, gflag, dflag, dc) in zip(overlaps, gt_annos, dt_annos, cleaned):
                    for t, thr in enumerate(th):
                        tp, fp, fn, sim = _frame_stats(ov, g["alpha"], d["alpha"], d["bbox"], gflag, dflag, d["score"], dc, metric, mo, thr,
                                                       compute_aos)
                        pr[t, :3] += (tp, fp, fn)
                        if sim != -1:
                            pr[t, 3] += sim
                n = len(th)
                with np.errstate(divide="ignore", invalid="ignore"):
                    recall[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, l, k, :n] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                for arr in (precision, recall, aos):       # monotone envelope over the sampled thresholds (eval.py:540-545)
                    for i in range(n):
                        arr[m, l, k, i] = np.max(arr[m, l, k, i:])
    return {"precision": precision, "recall": recall, "orientation": aos}


def get_mAP(prec):
    return prec[..., ::4].sum(axis=-1) / 11 * 100


def get_mAP_R40(prec):
    return prec[..., 1:].sum(axis=-1) / 40 * 100


def get_official_eval_result(gt_annos, dt_annos, current_classes, rotated_intersection=hip_rotated_intersection):
    """-> (text, ret_dict) with the reference's keys ('Car_3d/moderate_R40', ...)."""
    names = {v: k for k, v in CLASS_TO_NAME.items()}
    classes = [names[c] if isinstance(c, str) else c for c in (current_classes if isinstance(current_classes, (list, tuple)) else [current_classes])]
    mo = _OVERLAPS[:, :, classes]
    compute_aos = any(len(a["alpha"]) and a["alpha"][0] != -10 for a in dt_annos[:next((i + 1 for i, a in enumerate(dt_annos) if len(a["alpha"])), 0)])
    res, tables = {}, {}
    for metric, tag in ((0, "bbox"), (1, "bev"), (2, "3d")):
        r = eval_class(gt_annos, dt_annos, classes, metric, mo, compute_aos and metric == 0, rotated_intersection)
        tables[tag] = (get_mAP(r["precision"]), get_mAP_R40(r["precision"]))
        if metric == 0 and compute_aos:
            tables["aos"] = (get_mAP(r["orientation"]), get_mAP_R40(r["orientation"]))
    text, ret = "", {}
    for j, c in enumerate(classes):
        cname = CLASS_TO_NAME[c]
        for i in range(mo.shape[0]):
            for suffix, pick in (("AP", 0), ("AP_R40", 1)):
                text += "%s %s@%.2f, %.2f, %.2f:\n" % ((cname, suffix) + tuple(mo[i, :, j]))
                for tag, label in (("bbox", "bbox"), ("bev", "bev "), ("3d", "3d  ")):
                    v = tables[tag][pick][j, :, i]
                    text += "%s AP:%.4f, %.4f, %.4f\n" % (label, v[0], v[1], v[2])
                if compute_aos:
                    v = tables["aos"][pick][j, :, i]
                    text += "aos  AP:%.2f, %.2f, %.2f\n" % (v[0], v[1], v[2])
        for tag, key in (("3d", "3d"), ("bev", "bev"), ("bbox", "image")) + ((("aos", "aos"),) if compute_aos else ()):
            for d, dn in enumerate(("easy", "moderate", "hard")):
                ret["%s_%s/%s_R40" % (cname, key, dn)] = tables[tag][1][j, d, 0]
    return text, ret


# ------------------------------------------------------------------------------------------------ prediction formatting
def lidar_to_rect(xyz, calib):
    hom = np.hstack((xyz, np.ones((xyz.shape[0], 1), dtype=np.float32)))
    return np.dot(hom, np.dot(np.asarray(calib["Tr_velo2cam"], np.float32).T, np.asarray(calib["R0"], np.float32).T))


def boxes3d_lidar_to_kitti_camera(boxes, calib):
    """(N,7) lidar [x,y,z,dx,dy,dz,heading], centre -> camera [x,y,z,l,h,w,ry], bottom centre (box_utils.py:152-166)."""
    b = np.array(boxes, dtype=np.float32, copy=True)
    b[:, 2] -= b[:, 5] / 2
    return np.concatenate([lidar_to_rect(b[:, 0:3], calib), b[:, 3:4], b[:, 5:6], b[:, 4:5], -b[:, 6:7] - np.pi / 2], axis=-1)


def boxes3d_kitti_camera_to_imageboxes(boxes_cam, calib, image_shape=None):
    """Image-plane bounding boxes of camera-frame 3-D boxes (box_utils.py:169-235)."""
    n = boxes_cam.shape[0]
    l, h, w, ry = boxes_cam[:, 3], boxes_cam[:, 4], boxes_cam[:, 5], boxes_cam[:, 6]
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1], np.float32) / 2
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1], np.float32) / 2
    x, z = l[:, None] * sx, w[:, None] * sz
    y = np.zeros((n, 8), np.float32)
    y[:, 4:] = -h[:, None]
    c, s = np.cos(ry)[:, None], np.sin(ry)[:, None]
    corners = np.stack([x * c + z * s + boxes_cam[:, 0:1], y + boxes_cam[:, 1:2], -x * s + z * c + boxes_cam[:, 2:3]], axis=2)   # (N,8,3)
    P2 = np.asarray(calib["P2"], np.float32)
    rect = corners.reshape(-1, 3)
    hom = np.dot(np.hstack((rect, np.ones((rect.shape[0], 1), np.float32))), P2.T)
    uv = (hom[:, 0:2].T / rect[:, 2]).T.reshape(-1, 8, 2)
    out = np.concatenate([uv.min(axis=1), uv.max(axis=1)], axis=1)
    if image_shape is not None:
        out[:, [0, 2]] = np.clip(out[:, [0, 2]], 0, image_shape[1] - 1)
        out[:, [1, 3]] = np.clip(out[:, [1, 3]], 0, image_shape[0] - 1)
    return out


def generate_prediction_dicts(batch_dict, pred_dicts, class_names):
    """pred_dicts (the detector's sync=True output) -> KITTI annotation dicts (kitti_dataset.py:246-320)."""
    annos = []
    for idx, box_dict in enumerate(pred_dicts):
        scores = np.asarray(box_dict["pred_scores"].detach().cpu() if hasattr(box_dict["pred_scores"], "detach") else box_dict["pred_scores"])
        boxes = np.asarray(box_dict["pred_boxes"].detach().cpu() if hasattr(box_dict["pred_boxes"], "detach") else box_dict["pred_boxes"])
        labels = np.asarray(box_dict["pred_labels"].detach().cpu() if hasattr(box_dict["pred_labels"], "detach") else box_dict["pred_labels"])
        n = scores.shape[0]
        d = {"name": np.zeros(n), "truncated": np.zeros(n), "occluded": np.zeros(n), "alpha": np.zeros(n), "bbox": np.zeros((n, 4)),
             "dimensions": np.zeros((n, 3)), "location": np.zeros((n, 3)), "rotation_y": np.zeros(n), "score": np.zeros(n),
             "boxes_lidar": np.zeros((n, 7))}
        if n:
            calib, shape = batch_dict["calib"][idx], batch_dict["image_shape"][idx]
            cam = boxes3d_lidar_to_kitti_camera(boxes, calib)
            d.update(name=np.array(class_names)[labels - 1], alpha=-np.arctan2(-boxes[:, 1], boxes[:, 0]) + cam[:, 6],
                     bbox=boxes3d_kitti_camera_to_imageboxes(cam, calib, image_shape=shape), dimensions=cam[:, 3:6], location=cam[:, 0:3],
                     rotation_y=cam[:, 6], score=scores, boxes_lidar=boxes)
        d["frame_id"] = batch_dict["frame_id"][idx]
        annos.append(d)
    return annos
