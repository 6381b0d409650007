"""GPU parity: gate, head decode, score top-k (bit-exact order), rotated-BEV NMS (bit-exact survivors), pairwise IoU."""
import os

import numpy as np
import pytest
import torch

from hvpr_amd import kernels
from oracle import hvpr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _boxes(rng, n, spread=40.0, clustered=True):
    """Car-sized boxes; clustered so that many pairs overlap."""
    if clustered:
        centres = rng.uniform([0, -20], [spread, 20], (max(n // 6, 1), 2))
        xy = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 0.8, (n, 2))
    else:
        xy = rng.uniform([0, -20], [spread, 20], (n, 2))
    z = rng.normal(-1.0, 0.2, (n, 1))
    size = np.array([3.9, 1.6, 1.56]) * rng.uniform(0.8, 1.2, (n, 3))
    yaw = rng.uniform(-np.pi, np.pi, (n, 1))
    return np.concatenate([xy, z, size, yaw], 1).astype(np.float32)


def test_spatial_gate_vs_oracle():
    g = torch.Generator().manual_seed(3)
    for C, H, W in ((32, 19, 37), (64, 16, 16), (128, 5, 50)):
        y = torch.randn(2, C, H, W, generator=g)
        params = {"attention.spatial.conv.weight": torch.randn(1, 2, 3, 3, generator=g) * 0.5,
                  "attention.spatial.conv.bias": torch.randn(1, generator=g),
                  "attention.spatial.norm.weight": torch.rand(1, generator=g) + 0.5,
                  "attention.spatial.norm.bias": torch.randn(1, generator=g) * 0.2,
                  "attention.spatial.norm.running_mean": torch.randn(1, generator=g) * 0.2,
                  "attention.spatial.norm.running_var": torch.rand(1, generator=g) + 0.5}
        ref = O.spatial_gate(y, params)
        s = params["attention.spatial.norm.weight"] / torch.sqrt(params["attention.spatial.norm.running_var"] + 1e-3)
        t = params["attention.spatial.norm.bias"] - params["attention.spatial.norm.running_mean"] * s
        gate = kernels.spatial_gate(y.permute(0, 2, 3, 1).contiguous().to(DEV),
                                    params["attention.spatial.conv.weight"].reshape(18).to(DEV),
                                    params["attention.spatial.conv.bias"].item(), s.item(), t.item())
        np.testing.assert_allclose(gate.cpu().numpy(), ref[:, 0].numpy(), rtol=1e-3, atol=1e-5)


def test_head_decode_golden(golden_dir):
    for stride in (1, 2):
        z = np.load(os.path.join(golden_dir, f"g5_head_stride{stride}.npz"))
        cls, box, dirp = (torch.from_numpy(z[k]) for k in ("cls_preds", "box_preds", "dir_cls_preds"))
        head = torch.cat([cls, box, dirp], dim=-1).contiguous().to(DEV)
        anc = torch.from_numpy(z["anchors"])                     # (1, H, W, 1, 2, 7)
        xs, ys = anc[0, 0, :, 0, 0, 0].contiguous(), anc[0, :, 0, 0, 0, 1].contiguous()
        table = anc[0, 0, 0, 0, :, [2, 3, 4, 5, 6]].contiguous()
        c, b, s, lab = kernels.head_decode(head, 2, 1, 2, xs.to(DEV), ys.to(DEV), table.to(DEV), 0.78539, 0.0, np.pi)
        np.testing.assert_array_equal(c.cpu().numpy(), z["batch_cls_preds"])
        np.testing.assert_allclose(b.cpu().numpy(), z["batch_box_preds"], rtol=1e-3, atol=1e-5)
        np.testing.assert_allclose(s.cpu().numpy(), 1 / (1 + np.exp(-z["batch_cls_preds"][..., 0].astype(np.float64))), rtol=1e-5)
        assert (lab.cpu().numpy() == 1).all()


@pytest.mark.parametrize("A,thresh,pre", [(1000, 0.5, 4096), (146816, 0.9, 4096), (146816, 0.5, 4096), (146816, None, 4096),
                                          (30000, 0.0, 512), (5, 2.0, 16)])
def test_score_topk_exact_order(A, thresh, pre):
    rng = np.random.default_rng(A + pre)
    s = rng.uniform(0, 1, (2, A)).astype(np.float32)
    s[0, : A // 3] = np.round(s[0, : A // 3], 2)     # heavy ties -> the id tie-break must be honoured
    ws = kernels.PostWorkspace(2, A, pre, DEV)
    order, ss, counts = kernels.score_topk(torch.from_numpy(s).to(DEV), thresh, pre, ws)
    order, ss, counts = order.cpu().numpy(), ss.cpu().numpy(), counts.cpu().numpy()
    for b in range(2):
        passing = np.nonzero(s[b] >= np.float32(thresh))[0] if thresh is not None else np.arange(A)
        ref = passing[O.stable_order_desc(s[b][passing])][:pre]
        assert counts[b] == len(ref)
        np.testing.assert_array_equal(order[b, : counts[b]], ref)
        np.testing.assert_array_equal(ss[b, : counts[b]], s[b][ref])


@pytest.mark.parametrize("n,seed", [(1, 0), (63, 1), (64, 2), (65, 3), (700, 4), (4096, 5)])
def test_nms_survivors_bit_exact(n, seed):
    rng = np.random.default_rng(seed)
    boxes = _boxes(rng, n)
    scores = rng.uniform(0.1, 1, n).astype(np.float32)
    ref = O.nms_bev(boxes, scores, 0.1)
    ws = kernels.PostWorkspace(1, n, max(n, 1), DEV)
    order, _, cnt = kernels.score_topk(torch.from_numpy(scores[None]).to(DEV), None, n, ws)
    keep, kc = kernels.nms_bev(torch.from_numpy(boxes).to(DEV), order[0].contiguous(), cnt, n, 0.1, n, ws.nms)
    got = keep.cpu().numpy()[: int(kc.item())]
    np.testing.assert_array_equal(got, ref)
    # properties: survivors mutually below the threshold, score-sorted, idempotent
    iou = O.boxes_iou_bev(boxes[got], boxes[got])
    assert (iou[np.triu_indices(len(got), 1)] <= 0.1).all()
    assert (np.diff(scores[got]) <= 0).all()
    keep2, kc2 = kernels.nms_bev(torch.from_numpy(boxes[got]).to(DEV), None, None, len(got), 0.1, len(got), ws.nms, False)
    np.testing.assert_array_equal(keep2.cpu().numpy()[: int(kc2.item())], np.arange(len(got)))


def test_nms_post_max_and_thresholds():
    rng = np.random.default_rng(11)
    boxes = _boxes(rng, 2000, clustered=False)
    scores = rng.uniform(0, 1, 2000).astype(np.float32)
    ws = kernels.PostWorkspace(1, 2000, 2000, DEV)
    order, _, cnt = kernels.score_topk(torch.from_numpy(scores[None]).to(DEV), None, 2000, ws)
    for thr in (0.01, 0.1, 0.7):
        ref = O.nms_bev(boxes, scores, thr)[:500]
        keep, kc = kernels.nms_bev(torch.from_numpy(boxes).to(DEV), order[0].contiguous(), cnt, 2000, thr, 500, ws.nms)
        np.testing.assert_array_equal(keep.cpu().numpy()[: int(kc.item())], ref)


def test_pairwise_overlap_and_iou():
    rng = np.random.default_rng(5)
    a, b = _boxes(rng, 150), _boxes(rng, 90)
    ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
    for mode, fn in ((0, O.boxes_overlap_bev), (1, O.boxes_iou_bev), (2, O.boxes_iou3d)):
        got = kernels.boxes_pairwise(ta, tb, mode).cpu().numpy()
        ref = fn(a, b)
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)
    # identical boxes -> IoU 1 (within polygon round-off); disjoint -> exactly 0
    same = kernels.boxes_pairwise(ta[:10], ta[:10], 1).cpu().numpy()
    np.testing.assert_allclose(np.diag(same), 1.0, atol=1e-3)
    far = a.copy(); far[:, 0] += 500
    assert (kernels.boxes_pairwise(ta, torch.from_numpy(far).to(DEV), 0).cpu().numpy() == 0).all()


def test_multi_classes_nms_and_nms_normal_gpu():
    """model_nms_utils.multi_classes_nms (model_nms_utils.py:28-65) = one class-agnostic pass per class column, and the axis-aligned
    nms_normal_gpu wrapper, both against greedy NMS on the CPU oracle's IoU."""
    from hvpr_amd import detector, iou3d_nms_utils
    from hvpr_amd.config import AttrDict
    rng = np.random.default_rng(5)
    n = 900
    xy = rng.uniform([0, -20], [46, 20], (n, 2))
    boxes = np.concatenate([xy, np.full((n, 1), -1.0), rng.uniform(0.9, 1.1, (n, 3)) * [3.9, 1.6, 1.56], rng.uniform(-3, 3, (n, 1))], 1).astype(np.float32)
    cls = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    ncfg = AttrDict(NMS_TYPE="nms_gpu", NMS_THRESH=0.2, NMS_PRE_MAXSIZE=300, NMS_POST_MAXSIZE=40, MULTI_CLASSES_NMS=True)
    sc, lab, bx = detector.multi_classes_nms(torch.from_numpy(cls).to(DEV), torch.from_numpy(boxes).to(DEV), ncfg, score_thresh=0.3)
    want_s, want_l, want_b = [], [], []
    for k in range(3):
        sel, s = O.class_agnostic_nms(cls[:, k], boxes, 0.3, 0.2, 300, 40)
        want_s.append(s); want_l.append(np.full(len(sel), k)); want_b.append(boxes[sel])
    np.testing.assert_array_equal(sc.cpu().numpy(), np.concatenate(want_s))
    np.testing.assert_array_equal(lab.cpu().numpy(), np.concatenate(want_l))
    np.testing.assert_array_equal(bx.cpu().numpy(), np.concatenate(want_b))
    # axis-aligned NMS: closed-form IoU of the heading-less boxes, greedy in descending score
    b = boxes[:400]
    s = cls[:400, 0]
    keep, none = iou3d_nms_utils.nms_normal_gpu(torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV), 0.1)
    assert none is None
    order = np.lexsort((np.arange(len(s)), -s))
    x1, x2, y1, y2 = b[:, 0] - b[:, 3] / 2, b[:, 0] + b[:, 3] / 2, b[:, 1] - b[:, 4] / 2, b[:, 1] + b[:, 4] / 2
    area = b[:, 3] * b[:, 4]
    kept, near = [], 0
    for i in order:
        ok = True
        for j in kept:
            inter = max(min(x2[i], x2[j]) - max(x1[i], x1[j]), 0) * max(min(y2[i], y2[j]) - max(y1[i], y1[j]), 0)
            iou = inter / max(area[i] + area[j] - inter, 1e-8)
            near += abs(iou - 0.1) < 1e-4
            if iou > 0.1:
                ok = False
                break
        if ok:
            kept.append(i)
    if near == 0:          # no decision within rounding of the threshold: the survivor list is exact
        np.testing.assert_array_equal(keep.cpu().numpy(), np.array(kept))
    else:
        assert len(set(keep.cpu().numpy().tolist()) ^ set(kept)) <= 2 * near


def test_g14_rotated_iou_and_nms_on_axis_aligned_boxes_vs_reference(golden_dir):
    """Partial pin of row a8 on the GPU (fixture G14 = the reference's box_utils.boxes3d_nearest_bev_iou, box_utils.py:297-323, on
    boxes whose heading is a multiple of pi/2, where the rotated BEV IoU IS the axis-aligned one): hvpr_boxes_pairwise_f32 within
    the published routine's in-box margin of the reference IoU, exact zeros stay zero, and the NMS kernel's survivors = greedy NMS
    on the REFERENCE's IoU matrix at nine thresholds (see tests/test_oracle_golden.py for what is statement and what observation)."""
    from test_oracle_golden import G14_IOU_ATOL, _greedy_nms_from_iou
    z = np.load(f"{golden_dir}/g14_axis_aligned_iou.npz")
    a, b = torch.from_numpy(z["boxes_a"]).to(DEV), torch.from_numpy(z["boxes_b"]).to(DEV)
    got = kernels.boxes_pairwise(a, b, 1).cpu().numpy()
    np.testing.assert_allclose(got, z["iou_ab"], rtol=0, atol=G14_IOU_ATOL)
    assert (got[z["iou_ab"] == 0] == 0).all()
    assert abs(got[0, 0] - 1) < 2e-3 and abs(got[1, 1] - 1) < 2e-3
    boxes, scores = z["nms_boxes"], z["nms_scores"]
    n = len(scores)
    tb = torch.from_numpy(boxes).to(DEV)
    np.testing.assert_allclose(kernels.boxes_pairwise(tb, tb, 1).cpu().numpy(), z["iou_nms"], rtol=0, atol=G14_IOU_ATOL)
    ws = kernels.PostWorkspace(1, n, n, DEV)
    order, _, cnt = kernels.score_topk(torch.from_numpy(scores[None]).to(DEV), None, n, ws)
    for thr in (0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8):
        keep, kc = kernels.nms_bev(tb, order[0].contiguous(), cnt, n, thr, n, ws.nms)
        np.testing.assert_array_equal(keep.cpu().numpy()[: int(kc.item())], _greedy_nms_from_iou(z["iou_nms"], scores, thr)[0])


def test_rotated_iou_kernel_against_exact_float64_clipping():
    """hvpr_boxes_pairwise_f32 at GENERAL headings against an exact float64 Sutherland-Hodgman intersection (tests/exact_geometry.py,
    independent of the published routine both the oracle and the kernel restate): 4000 random pairs, deviation within the reach of
    the routine's 1e-2 in-box margin, no systematic slip — and the kernel equals the oracle on the same pairs."""
    from exact_geometry import iou_bev, random_pairs
    a, b = random_pairs(4000, 2024)
    got = np.concatenate([np.diagonal(kernels.boxes_pairwise(torch.from_numpy(a[i:i + 500]).to(DEV), torch.from_numpy(b[i:i + 500]).to(DEV), 1)
                                      .cpu().numpy()) for i in range(0, len(a), 500)]).astype(np.float64)
    exact = np.array([iou_bev(a[i], b[i]) for i in range(len(a))])
    d = got - exact
    print(f"rotated IoU, kernel vs exact float64 clipping: max |d| {np.abs(d).max():.2e}, mean d {d.mean():.2e}")
    assert np.abs(d).max() <= 1e-2 and abs(d.mean()) <= 1e-4
    ora = np.array([O.boxes_iou_bev(a[i:i + 1], b[i:i + 1])[0, 0] for i in range(len(a))])
    np.testing.assert_allclose(got, ora, rtol=1e-5, atol=1e-6)


def _g15_detector(c):
    from hvpr_amd import detector
    from hvpr_amd.config import AttrDict
    import g15_cases
    cfg = AttrDict(POST_PROCESSING=dict(RECALL_THRESH_LIST=g15_cases.RECALL_THRESH_LIST, SCORE_THRESH=g15_cases.SCORE_THRESH,
                                        OUTPUT_RAW_SCORE=c.raw, EVAL_METRIC="kitti",
                                        NMS_CONFIG=dict(MULTI_CLASSES_NMS=c.multi, NMS_TYPE="nms_gpu", NMS_THRESH=c.nms_thresh,
                                                        NMS_PRE_MAXSIZE=c.pre, NMS_POST_MAXSIZE=c.post)))

    class DS:
        class_names = ["Car", "Pedestrian", "Cyclist"][: c.num_class]
    return detector.Detector3DTemplate(cfg, c.num_class, DS()), cfg, detector


def test_post_processing_wrapper_against_reference_fixture_g15(golden_dir):
    """Row a8's wrapper on the product path against the REFERENCE's own functions (fixture G15 = model_nms_utils.py:6-65 and
    Detector3DTemplate.post_processing / generate_recall_record, detector3d_template.py:168-318, run in the build container):
    selected anchor ids, boxes and labels exact, scores to the sigmoid's last bit, recall counters exact — class-agnostic branch
    (one / three classes, raw-score output, normalised input), MULTI_CLASSES_NMS branch, the two model_nms_utils functions
    called directly, and the no-read-back form (sync=False)."""
    import g15_cases
    for c in g15_cases.load(golden_dir):
        det, cfg, detector = _g15_detector(c)
        bd = {"batch_size": c.cls.shape[0], "batch_cls_preds": torch.from_numpy(c.cls).to(DEV),
              "batch_box_preds": torch.from_numpy(c.boxes).to(DEV), "cls_preds_normalized": c.normalized,
              "gt_boxes": torch.from_numpy(c.gt_boxes).to(DEV)}
        preds, recall, _ = det.post_processing(dict(bd))
        assert recall == c.recall, (c.tag, recall, c.recall)
        exact_scores = c.normalized or c.raw
        for b, (p, f) in enumerate(zip(preds, c.frames)):
            msg = f"{c.tag} f{b}"
            np.testing.assert_array_equal(p["pred_boxes"].cpu().numpy(), f["pred_boxes"], err_msg=msg)
            np.testing.assert_array_equal(p["pred_labels"].cpu().numpy(), f["pred_labels"], err_msg=msg)
            if exact_scores:
                np.testing.assert_array_equal(p["pred_scores"].cpu().numpy(), f["pred_scores"], err_msg=msg)
            else:
                np.testing.assert_allclose(p["pred_scores"].cpu().numpy(), f["pred_scores"], rtol=1e-6, atol=0, err_msg=msg)
            ncfg = cfg.POST_PROCESSING.NMS_CONFIG
            sc = bd["batch_cls_preds"][b] if c.normalized else torch.sigmoid(bd["batch_cls_preds"][b])
            if c.multi:
                s, l, bx = detector.multi_classes_nms(sc, bd["batch_box_preds"][b], ncfg, score_thresh=g15_cases.SCORE_THRESH)
                np.testing.assert_allclose(s.cpu().numpy(), f["mc_scores"], rtol=1e-6, atol=0, err_msg=msg)
                np.testing.assert_array_equal(l.cpu().numpy(), f["mc_labels"], err_msg=msg)
                np.testing.assert_array_equal(bx.cpu().numpy(), f["mc_boxes"], err_msg=msg)
            else:
                np.testing.assert_array_equal(p["selected"].cpu().numpy(), f["selected"], err_msg=msg)
                sel, ss = detector.class_agnostic_nms(sc.max(-1)[0], bd["batch_box_preds"][b], ncfg, score_thresh=g15_cases.SCORE_THRESH)
                np.testing.assert_array_equal(sel.cpu().numpy(), f["selected"], err_msg=msg)
                np.testing.assert_allclose(ss.cpu().numpy(), f["selected_scores"], rtol=1e-6, atol=0, err_msg=msg)
        if not c.multi:     # the form the frame pipeline uses: padded rows + a device count, no host read
            preds, recall, _ = det.post_processing(dict(bd), sync=False)
            assert recall == {}
            for b, (p, f) in enumerate(zip(preds, c.frames)):
                n = int(p["pred_count"].item())
                assert n == len(f["selected"]) and p["selected"].shape[0] == min(c.post, p["selected"].shape[0])
                np.testing.assert_array_equal(p["selected"][:n].cpu().numpy(), f["selected"])
                np.testing.assert_array_equal(p["pred_boxes"][:n].cpu().numpy(), f["pred_boxes"])
                np.testing.assert_array_equal(p["pred_labels"][:n].cpu().numpy(), f["pred_labels"])


def test_generate_recall_record_branches():
    """generate_recall_record (detector3d_template.py:276-318) beyond G15: no gt_boxes key -> the dict is returned untouched;
    counters accumulate over frames and over calls; no predictions -> only `gt` moves; a prediction equal to a gt box is recalled
    at every threshold, one 10 m away at none."""
    from hvpr_amd import detector
    rng = np.random.default_rng(7)
    gt = torch.zeros(2, 4, 8, device=DEV)
    b = torch.from_numpy(_boxes(rng, 3, clustered=False)).to(DEV)
    gt[0, :3, :7], gt[0, :3, 7] = b, 1.0
    gt[1, :1, :7], gt[1, :1, 7] = b[:1], 1.0
    rec = detector.Detector3DTemplate.generate_recall_record
    assert rec(b, {"x": 1}, 0, {}, [0.3]) == {"x": 1}
    far = b.clone(); far[:, 0] += 10.0
    d = rec(b, {}, 0, {"gt_boxes": gt}, [0.3, 0.5, 0.7])
    assert d == {"gt": 3, "roi_0.3": 0, "rcnn_0.3": 3, "roi_0.5": 0, "rcnn_0.5": 3, "roi_0.7": 0, "rcnn_0.7": 3}
    d = rec(far, d, 1, {"gt_boxes": gt}, [0.3, 0.5, 0.7])
    assert d == {"gt": 4, "roi_0.3": 0, "rcnn_0.3": 3, "roi_0.5": 0, "rcnn_0.5": 3, "roi_0.7": 0, "rcnn_0.7": 3}
    d = rec(b[:0], d, 0, {"gt_boxes": gt}, [0.3, 0.5, 0.7])
    assert d["gt"] == 7 and d["rcnn_0.3"] == 3
    d = rec(b[:1], d, 1, {"gt_boxes": torch.zeros(2, 4, 8, device=DEV)}, [0.3, 0.5, 0.7])     # `while k > 0`: one zero box stays
    assert d["gt"] == 8 and d["rcnn_0.3"] == 3
