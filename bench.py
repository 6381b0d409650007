#!/usr/bin/env python3
"""Benchmark of the HVPR forward hot path on MI355X (BASELINE.json metric: KITTI frames/sec/GPU, fwd, ~20k pts;
VFE+scatter achieved HBM GB/s vs peak).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the path a1..a8 (voxelize -> pillar VFE -> memory read-out + scatter -> BEV backbone -> head +
decode -> score top-k + rotated NMS) over one batch of ONE synthetic frame (tools/cfgs hvpr_car.yaml, batch=1 forward —
BASELINE.json configs[1]) whose points are already resident in HBM.  Multi-GPU = replicas only: frames are sharded over
ranks, no data-path collective; value = frames of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 with `roofline` (VFE+scatter kernel group, HBM bound — the group BASELINE.json's metric
names), `roofline_mfma` (BEV backbone + head, fp32 matrix-core bound) and `cpu_baseline` (the CPU oracle timed on this
box's host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hvpr_amd import detector, distributed, synthetic, synthetic_weights  # noqa: E402
from hvpr_amd.config import hvpr_car_cfg  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32, exact fp32
N_POOL = 8                      # distinct frames per rank, cycled


def make_batch(frame, device):
    pts = np.concatenate([np.zeros((len(frame), 1), np.float32), frame], axis=1)
    return {"points": torch.from_numpy(pts).to(device),
            "point_frame_offsets": torch.tensor([0, len(frame)], dtype=torch.int32, device=device),
            "batch_size": 1}


def conv_flops(model, H, W):
    """2*MACs of BaseBEVBackbone_Scale (eval) + the three 1x1 heads (BN / ReLU / gate not counted)."""
    bb = model.backbone_2d
    fl = 0
    h, w = H, W
    for i, blk in enumerate(bb.blocks):
        s = bb.layer_strides[i]
        h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
        convs = [m for m in blk if isinstance(m, torch.nn.Conv2d)]
        for c in convs:
            fl += 2 * c.in_channels * c.out_channels * 9 * h * w
        sf = bb.sfmblocks_down[i][0]
        fl += bb.sfm_layer_nums[i] * 2 * sf.in_channels * sf.out_channels * 9 * h * w
        sc = bb.scale_layers[i][1]
        fl += 2 * sc.in_channels * sc.out_channels * 9 * h * w
        de = bb.deblocks[i][0]
        fl += 2 * de.in_channels * de.out_channels * (h * w) * int(bb.upsample_strides[i]) ** 2
    head = model.dense_head
    for c in (head.conv_cls, head.conv_box, head.conv_dir_cls):
        fl += 2 * c.in_channels * c.out_channels * H * W
    return fl


def wino_conv_flops(model, H, W):
    """2*MACs (direct-convolution count) of the stride-1 3x3 layers that run on the Winograd F(2x2,3x3) kernel, which executes
    16/36 of them (kernels.pack_conv_auto: every 3x3 / stride-1 layer unless HVPR_CONV_ALGO=direct)."""
    from hvpr_amd import kernels
    if kernels.conv_algo() != "winograd":
        return 0
    bb = model.backbone_2d
    fl = 0
    h, w = H, W
    for i, blk in enumerate(bb.blocks):
        s = bb.layer_strides[i]
        h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
        for c in [m for m in blk if isinstance(m, torch.nn.Conv2d)]:
            if c.stride[0] == 1:
                fl += 2 * c.in_channels * c.out_channels * 9 * h * w
        sf = bb.sfmblocks_down[i][0]
        fl += bb.sfm_layer_nums[i] * 2 * sf.in_channels * sf.out_channels * 9 * h * w
        sc = bb.scale_layers[i][1]
        if sc.stride[0] == 1:
            fl += 2 * sc.in_channels * sc.out_channels * 9 * h * w
    return fl


def split_conv_flops(model, H, W, precision):
    """2*MACs of the layers the split-bf16 modes run on the bf16 matrix cores: trunk + SFM 3x3 (and the deconvolutions in bf16x3)."""
    bb = model.backbone_2d
    fl = 0
    h, w = H, W
    for i, blk in enumerate(bb.blocks):
        s = bb.layer_strides[i]
        h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
        for c in [m for m in blk if isinstance(m, torch.nn.Conv2d)]:
            fl += 2 * c.in_channels * c.out_channels * 9 * h * w
        sf = bb.sfmblocks_down[i][0]
        fl += bb.sfm_layer_nums[i] * 2 * sf.in_channels * sf.out_channels * 9 * h * w
        if precision == "bf16x3":
            de = bb.deblocks[i][0]
            fl += 2 * de.in_channels * de.out_channels * (h * w) * int(bb.upsample_strides[i]) ** 2
    return fl


class StagedGraphs:
    """The same forward as MixAnchor_Memory.forward(sync=False), captured as THREE hipGraphs (VFE+scatter group,
    backbone+head+decode, top-k+NMS) so that each group can be bracketed by HIP events on the launch stream without
    host launch gaps inside a group."""

    def __init__(self, model, example_batch):
        self.static_in = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        self.static_in.update(model.persistent_canvases(example_batch))   # as GraphedForward / PipelinedForward do
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                model(dict(self.static_in), sync=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
        pool = None
        with torch.no_grad():
            with torch.cuda.graph(self.graphs[0]):
                bd = model.stage_encode(dict(self.static_in))
            pool = self.graphs[0].pool()
            with torch.cuda.graph(self.graphs[1], pool=pool):
                bd = model.backbone_2d(bd)
                bd = model.dense_head.forward(bd)
            with torch.cuda.graph(self.graphs[2], pool=pool):
                self.out = model.post_processing(bd, sync=False)
        self.captured = detector._CapturedState(model)      # keeps the captured workspaces / packed weights alive

    def run(self, batch, ev):
        self.captured.check()
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static_in[k].copy_(v, non_blocking=True)
        ev[0].record()
        for i, g in enumerate(self.graphs):
            g.replay()
            ev[i + 1].record()
        return self.out


def _no_grad(fn):
    if fn is None:
        return None

    def wrapped(*a, **k):
        with torch.no_grad():
            return fn(*a, **k)
    return wrapped


def host_cores():
    """Cores this process may really use: the cgroup CPU quota when there is one (the GPU box shows 256 logical CPUs
    behind a 16-CPU quota; running 256 torch threads against it is 60x slower than 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg, params, n_timed=10, n_warm=3, check=None):
    """the CPU restatement of a1..a8 on this box's host cores, 3 warm-up + 10 timed frames.  check(frame, preds,
    intermediates) is called for every frame OUTSIDE the timed intervals: the parity gates (the GPU pipeline that produced `value`
    is handed the same frame and compared stage by stage)."""
    from oracle import hvpr_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    ocfg = O.cfg_from_model_cfg(cfg)
    frames = [synthetic.hvpr_frame(1000 + i) for i in range(n_timed + n_warm)]
    for f in frames[:n_warm]:
        preds, inter = O.forward_frames([f], params, ocfg)             # warm-up (allocators, oneDNN primitives)
        if check is not None:
            check(f, preds, inter)
        del preds, inter
    timings = {}
    dt = 0.0
    for f in frames[n_warm:]:
        t0 = time.perf_counter()
        preds, inter = O.forward_frames([f], params, ocfg, timings=timings)
        dt += time.perf_counter() - t0
        if check is not None:
            check(f, preds, inter)
        del preds, inter
    return {"value": n_timed / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_timed} synthetic hvpr_car frames (batch=1, a1..a8) after {n_warm} warm-up frames",
            "threads": {"dense stages (VFE, memory, scatter, backbone, head, decode)": f"torch CPU fp32, {cores} threads",
                        "voxelize, top-k + rotated NMS": "C restatement, 1 thread (sequential algorithms)"},
            "stage_ms": {k: round(1e3 * v / n_timed, 2) for k, v in timings.items()}}


# ---------------------------------------------------------------------------------------------------------------- parity gates
PARITY_RTOL = 1e-3            # north_star: feature tensors and box regressions within 1e-3 relative (fp32)
# A survivor-set difference between the GPU pipeline and the CPU oracle must hang on a root decision whose quantity differs between
# the two sides by no more than oracle/survivor_flips.DELTA_SCORE / DELTA_IOU (calibration: that file's header)


def _relerr(got, ref):
    """max over elements of |got - ref| / (|ref| + rms(ref)): <= rtol is the element-wise statement |got - ref| <= rtol |ref| +
    rtol rms(ref) of tests/test_gpu_e2e.py (the absolute term is tied to the tensor's own scale)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    rms = float(np.sqrt(np.mean(np.square(ref)))) or 1e-30
    return float((np.abs(got - ref) / (np.abs(ref) + rms)).max()) if ref.size else 0.0


def eager_inspect(model, batch):
    """The dictionary PipelinedForward.inspect returns, from one eager forward (modes without the frame pipeline)."""
    out, _, bd = model(dict(batch), sync=False)
    torch.cuda.synchronize()
    d = {k: bd[k] for k in ("voxel_coords", "voxel_num_points", "voxel_offsets", "pillar_features", "pillar_scale_features",
                            "spatial_features", "spatial_scale_features", "spatial_features_2d", "batch_cls_preds", "batch_box_preds",
                            "batch_max_scores", "batch_max_labels")}
    d["post_input"] = {k: bd[k] for k in ("batch_cls_preds", "batch_box_preds", "batch_max_scores", "batch_max_labels")}
    d["post"] = out
    return d


class ParityGates:
    """SURVEY.md §8d "parity gates run with every benchmark": the object that was TIMED (the frame pipeline / graph / eager model of
    this run, its persistent canvases, packed weights and workspaces as the timed replays left them) is handed the frames the CPU
    oracle has just been timed on, and compared with the oracle's intermediates:
      voxel coords + counts + order bit-exact; pillar features, canvases, the 384-channel feature map and box regressions within
      1e-3 relative; NMS survivor ids bit-exact when the oracle's post-processing is fed the GPU's own scores and boxes; and the two
      pipelines' OWN survivor sets differ only where oracle/survivor_flips.explain can exhibit a root decision within round-off of
      its threshold."""

    def __init__(self, inspect, model_cfg, device):
        pp = model_cfg.POST_PROCESSING
        self.inspect, self.device = inspect, device
        self.post = (float(pp.SCORE_THRESH), float(pp.NMS_CONFIG.NMS_THRESH), int(pp.NMS_CONFIG.NMS_PRE_MAXSIZE), int(pp.NMS_CONFIG.NMS_POST_MAXSIZE))
        self.n = 0
        self.worst = {"pillar_rel": 0.0, "pillar_scale_rel": 0.0, "canvas_rel": 0.0, "feat2d_rel": 0.0, "cls_rel": 0.0, "box_rel": 0.0,
                      "heading_outside_frac": 0.0, "score_absdiff": 0.0}
        self.voxel_exact = self.nms_exact = True
        self.common = self.total = self.flips = self.unexplained = 0
        self.min_common_frac = 1.0
        self.roots, self.kept, self.same_order = [], [], [0, 0]

    def __call__(self, frame, ref_preds, inter):
        from oracle import hvpr_oracle as O
        from oracle import survivor_flips
        got = self.inspect(make_batch(frame, self.device))
        cpu = lambda t: t.detach().cpu().numpy()
        m = len(inter["voxel_coords"])
        ok = int(cpu(got["voxel_offsets"])[-1]) == m
        ok = ok and np.array_equal(cpu(got["voxel_coords"][:m]), inter["voxel_coords"])
        ok = ok and np.array_equal(cpu(got["voxel_num_points"][:m]), inter["voxel_num_points"])
        self.voxel_exact &= bool(ok)
        w = self.worst
        if ok:
            w["pillar_rel"] = max(w["pillar_rel"], _relerr(cpu(got["pillar_features"][:m]), inter["pillar_features"].numpy()))
            w["pillar_scale_rel"] = max(w["pillar_scale_rel"], _relerr(cpu(got["pillar_scale_features"][:m]), inter["pillar_scale_features"].numpy()))
        w["canvas_rel"] = max(w["canvas_rel"], _relerr(cpu(got["spatial_features"]), inter["spatial_features"].numpy()),
                              _relerr(cpu(got["spatial_scale_features"]), inter["spatial_scale_features"].numpy()))
        w["feat2d_rel"] = max(w["feat2d_rel"], _relerr(cpu(got["spatial_features_2d"]), inter["spatial_features_2d"].numpy()))
        w["cls_rel"] = max(w["cls_rel"], _relerr(cpu(got["batch_cls_preds"]), inter["batch_cls_preds"].numpy()))
        gb, rb = cpu(got["batch_box_preds"]), inter["batch_box_preds"].numpy()
        w["box_rel"] = max([w["box_rel"]] + [_relerr(gb[..., c], rb[..., c]) for c in range(6)])
        # heading: the direction bin is an argmax of two logits — a near-tie may land in the other bin (anchor_head_template.py:327-333)
        w["heading_outside_frac"] = max(w["heading_outside_frac"],
                                        float((np.abs(gb[..., 6] - rb[..., 6]) >= PARITY_RTOL * np.abs(rb[..., 6]).max()).mean()))
        # a8, strict: the oracle's post-processing on the GPU's own scores and boxes (the tensors the captured top-k + NMS read)
        pin = got["post_input"]
        s_gpu, b_gpu = cpu(pin["batch_max_scores"])[0], cpu(pin["batch_box_preds"])[0]
        sel_ref, sc_ref = O.class_agnostic_nms(s_gpu, b_gpu, *self.post)
        rec = got["post"][0]
        n = int(rec["pred_count"].item())
        sel = cpu(rec["selected"][:n])
        self.nms_exact &= bool(np.array_equal(sel, sel_ref) and np.array_equal(cpu(rec["pred_scores"][:n]), sc_ref)
                               and np.array_equal(cpu(rec["pred_boxes"][:n]), b_gpu[sel_ref]))
        self.kept.append(n)
        # a8, end to end: each pipeline on its own logits; every difference traced to its root
        s_cpu = torch.sigmoid(inter["batch_cls_preds"][0]).max(dim=-1)[0].numpy()
        w["score_absdiff"] = max(w["score_absdiff"], float(np.abs(s_gpu - s_cpu).max()))
        r = survivor_flips.explain(s_gpu, b_gpu, s_cpu, rb[0], *self.post, delta_score=survivor_flips.DELTA_SCORE, delta_iou=survivor_flips.DELTA_IOU)
        # the same with ONE score order (the GPU's scores on both sides, each side its own boxes): what is left are IoU decisions
        r1 = survivor_flips.explain(s_gpu, b_gpu, s_gpu, rb[0], *self.post, delta_score=0.0, delta_iou=survivor_flips.DELTA_IOU)
        self.same_order[0] += r1["common"]
        self.same_order[1] += max(r1["survivors_a"], r1["survivors_b"])
        self.unexplained += len(r1["unexplained"])
        assert r["survivors_b"] == len(ref_preds[0]["selected"])          # side b IS the oracle's own post-processing
        self.common += r["common"]
        self.total += max(r["survivors_a"], r["survivors_b"])
        self.min_common_frac = min(self.min_common_frac, r["common"] / max(r["survivors_a"], r["survivors_b"], 1))
        self.flips += len(r["flips"])
        self.unexplained += len(r["unexplained"])
        for root in r["roots"]:
            if len(self.roots) < 12:
                self.roots.append({"frame": self.n, **{k: (round(v, 8) if isinstance(v, float) else v) for k, v in root.items()}})
        self.n += 1

    def result(self):
        from oracle import survivor_flips
        w = {k: float(f"{v:.3e}") for k, v in self.worst.items()}
        ok = (self.n > 0 and self.voxel_exact and self.nms_exact and self.unexplained == 0 and w["heading_outside_frac"] < PARITY_RTOL
              and self.same_order[0] >= 0.99 * self.same_order[1]
              and all(w[k] <= PARITY_RTOL for k in ("pillar_rel", "pillar_scale_rel", "canvas_rel", "feat2d_rel", "cls_rel", "box_rel")))
        return {"ok": bool(ok), "frames": self.n, "voxel_exact": bool(self.voxel_exact), **w, "rtol": PARITY_RTOL,
                "nms_exact_on_gpu_logits": bool(self.nms_exact), "boxes_kept_per_frame": self.kept,
                "survivors_common": [self.common, self.total], "survivors_common_min_frac": round(self.min_common_frac, 4),
                "survivors_common_same_score_order": self.same_order,
                "survivor_flips": self.flips, "survivor_flips_unexplained": self.unexplained,
                "flip_delta": {"score": survivor_flips.DELTA_SCORE, "iou": survivor_flips.DELTA_IOU}, "flip_roots": self.roots,
                "what": "the timed object (same pipeline / graphs, persistent canvases, packed weights) run on the cpu_baseline frames and "
                        "compared with the CPU oracle stage by stage: voxel coords + counts + order exact; *_rel = max |gpu - cpu| / (|cpu| + "
                        "rms(cpu)) <= rtol; heading_outside_frac = direction-bin near-ties; nms_exact_on_gpu_logits = oracle NMS on the GPU's "
                        "scores and boxes gives the GPU's ids, scores and boxes bit for bit; survivors_common = [common, max(|gpu|, |cpu|)] "
                        "summed over frames, each pipeline on its OWN logits; every id in the symmetric difference is traced to a root decision "
                        "(flip_roots: quantity on side a = GPU and b = CPU oracle, threshold; order_swap = two overlapping candidates whose scores "
                        "agree to a few ulp come in the other order) whose two values differ by <= flip_delta; "
                        "survivors_common_same_score_order = the same with the GPU's scores on both sides (each side its own boxes): only IoU "
                        "decisions can differ then"}


class GroupGraphs:
    """The VFE+scatter group (hvpr_encode_fwd_f32) captured once per pool frame — every graph reads its own static copy of a
    DIFFERENT frame and all of them write the same persistent canvas pair — so that back-to-back replays are what a stream of
    frames costs: the stale-cell clear of the previous frame's pillars is inside the timed interval."""

    def __init__(self, model, batches):
        own = model.persistent_canvases(batches[0])
        self.inputs = [{**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}, **own} for b in batches]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for inp in self.inputs:
                model.stage_encode(dict(inp))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # ONE hipGraph holding the group of every pool frame back to back (a different frame each time, all on the same persistent
        # canvas pair): the interval between two events around a replay is then the group's kernels and the gaps between them,
        # as inside a frame graph — with one graph per frame every group also paid the start-up of a graph launch
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            for inp in self.inputs:
                self.bd = model.stage_encode(dict(inp))
        self.n = len(self.inputs)
        self.captured = detector._CapturedState(model)

    def time_us(self, rounds=5):
        self.captured.check()
        self.graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rounds):
            self.graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (rounds * self.n)


def group_bytes(n_pts, nx, ny, batch):
    """Algorithmic bytes of the VFE+scatter group: raw points read once + dense canvases written once (zeros
    included) + VFE weights and memory bank read once."""
    w_bytes = 4 * (16 * 10 + 16 + 64 * 32 + 64 + 16 * 5 + 16 + 32 * 16 + 32) + 2000 * 64 * 4
    return 16 * n_pts + 4 * (128 + 32) * nx * ny * batch + w_bytes


def batch_of(frames, device):
    pts = np.concatenate([np.concatenate([np.full((len(f), 1), b, np.float32), f], 1) for b, f in enumerate(frames)])
    offs = np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)
    return {"points": torch.from_numpy(pts).to(device), "point_frame_offsets": torch.from_numpy(offs).to(device), "batch_size": len(frames)}


def extra_group_lines(device):
    """The same group at hvpr_car batch 16 and on BASELINE.json configs[4] (dense scene: 200 k uniform points per frame,
    512 x 512 grid, 20 points / pillar, 60 000-pillar cap, batch 4), each as hipGraphs over two alternating batches."""
    import copy
    out = {}
    car = hvpr_car_cfg()
    dense = copy.deepcopy(car)
    rng = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    dense.DATA_CONFIG.POINT_CLOUD_RANGE = rng
    for p in dense.DATA_CONFIG.DATA_PROCESSOR:
        if p.NAME == "transform_points_to_voxels":
            p.VOXEL_SIZE, p.MAX_POINTS_PER_VOXEL, p.MAX_NUMBER_OF_VOXELS = [0.2, 0.2, 8.0], 20, {"train": 60000, "test": 60000}
    for key, cfg, make in (("hvpr_car_batch16", car, lambda k: [synthetic.hvpr_frame(500 + 16 * k + i) for i in range(16)]),
                           ("dense_scene", dense, lambda k: [synthetic.uniform_frame(70 + 4 * k + b, 200000, rng) for b in range(4)])):
        model = detector.build_network(cfg.MODEL, 1, detector.SyntheticDataset(cfg))
        synthetic_weights.load_synthetic(model, seed=0)
        model = model.to(device).eval()
        batches = [batch_of(make(k), device) for k in range(2)]
        gg = GroupGraphs(model, batches)
        us = gg.time_us(rounds=10)
        B, _, ny, nx = gg.bd["spatial_features"].shape
        n_pts = int(batches[0]["points"].shape[0])
        alg = group_bytes(n_pts, nx, ny, B)
        out[key] = {"workload": f"{B} frames x {n_pts // B} points, grid {nx}x{ny}, {int(gg.bd['voxel_offsets'][-1])} pillars",
                    "group_ms": round(us / 1e3, 4), "algorithmic_bytes": alg, "algorithmic_GBps": round(alg / us / 1e3, 1),
                    "frac_of_hbm_peak": round(alg / us / 1e3 / HBM_PEAK_GBPS, 4)}
        del gg, model, batches
        torch.cuda.empty_cache()
    return out


def train_conv_flops(model, H, W):
    """Direct-convolution FLOPs (2 * MACs) of ONE frame's training step through the backbone and head — forward, data gradient and
    weight gradient of BOTH streams (the scale stream once) — and the part of them that is EXECUTED: the stride-1 3x3 layers run
    forward, data gradient and weight gradient in the Winograd F(2x2,3x3) domain (16 of 36 multiplies); the stride-2 layers run all three
    directly (the data gradient as a gather per output-pixel parity class: its direct count)."""
    bb = model.backbone_2d
    direct = executed = 0.0
    h, w = H, W
    for i, blk in enumerate(bb.blocks):
        s = bb.layer_strides[i]
        h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
        px = h * w

        def add(cin, cout, stride, streams):
            nonlocal direct, executed
            f = 2.0 * cin * cout * 9 * px * streams
            direct += 3 * f                                   # fwd + dgrad + wgrad
            if stride == 1:
                executed += 3 * f * 16.0 / 36.0
            else:
                executed += 3 * f                                 # fwd, dgrad (hvpr_conv2d_s2_dgrad_nhwc_f32), wgrad: direct
        convs = [m for m in blk if isinstance(m, torch.nn.Conv2d)]
        for c in convs:
            add(c.in_channels, c.out_channels, c.stride[0], 2)
        sf = bb.sfmblocks_down[i][0]
        for _ in range(bb.sfm_layer_nums[i]):
            add(sf.in_channels, sf.out_channels, 1, 2)
        sc = bb.scale_layers[i][1]
        add(sc.in_channels, sc.out_channels, sc.stride[0], 1)
        de = bb.deblocks[i][0]
        f = 2.0 * de.in_channels * de.out_channels * px * int(bb.upsample_strides[i]) ** 2 * 2
        direct += 3 * f
        executed += 3 * f
    head = model.dense_head
    for c in (head.conv_cls, head.conv_box, head.conv_dir_cls):
        f = 2.0 * c.in_channels * c.out_channels * H * W * 2
        direct += 3 * f
        executed += 3 * f
    return direct, executed


def train_step_line(device, which="car", batch=16, steps=3, warmup=2, ddp=False):
    """BASELINE.json configs[2] (which="car", batch 16) / configs[3] per GPU (which="3class", batch 8): the full train step
    (a1..a15: fwd + bwd + Adam-onecycle), a bounded number of steps; matrix-core fraction from the executed convolution FLOPs.
    ddp=True (every rank calls it): the model is wrapped in DistributedDataParallel over the RCCL group — the gradient all-reduce
    of row a15 (reference tools/train.py:143-145) is inside the timed steps; time = max over ranks between two barriers."""
    from hvpr_amd import optim
    assert warmup >= 1, "the clock starts after warm-up step `warmup - 1`"
    from hvpr_amd.config import hvpr_3class_cfg
    cfg = hvpr_car_cfg() if which == "car" else hvpr_3class_cfg()
    n_class = len(cfg.CLASS_NAMES)
    ds = detector.SyntheticDataset(cfg, training=True)
    model = detector.build_network(cfg.MODEL, n_class, ds)
    synthetic_weights.load_synthetic(model, seed=0, cls_bias=-4.59511985013459)
    model = model.to(device)
    rank, _, world = distributed.env_rank()
    grad_bytes = 4 * sum(p.numel() for p in model.parameters() if p.requires_grad)
    if ddp:
        model = distributed.wrap_ddp(model, device)
    opt = optim.build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = optim.build_scheduler(opt, total_iters_each_epoch=steps + warmup, total_epochs=1, last_epoch=-1, optim_cfg=cfg.OPTIMIZATION)
    rng = np.random.default_rng(rank if ddp else 0)
    sizes = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], np.float32)

    def gt(B, per=8):
        g = np.zeros((B, per, 8), np.float32)
        cls = rng.integers(0, n_class, (B, per))
        g[..., 0] = rng.uniform(3, 44, (B, per)); g[..., 1] = rng.uniform(-17, 17, (B, per)); g[..., 2] = rng.uniform(-1.2, -0.8, (B, per))
        g[..., 3:6] = sizes[cls] * rng.uniform(0.9, 1.1, (B, per, 3))
        g[..., 6] = rng.uniform(-np.pi, np.pi, (B, per)); g[..., 7] = cls + 1
        return g
    pool = []
    for k in range(2):
        b = batch_of([synthetic.hvpr_frame(2000 + (rank * 1000 if ddp else 0) + batch * k + i, shuffle=True) for i in range(batch)], device)
        b.pop("point_frame_offsets")
        b["gt_boxes"] = torch.from_numpy(gt(batch)).to(device)
        pool.append(b)
    torch.cuda.reset_peak_memory_stats()
    losses = []
    # batches one ahead (optim.prefetching): the point-stream index kernels (FPS, ball query, three-NN) of batch i + 1 run on a side
    # stream beside step i.  The generator enqueues the index work of batch i + 1 BEFORE it yields batch i, so the clock starts
    # when batch `warmup - 1` has been stepped (its successor's index work is then enqueued inside the timed region) and stops
    # after step warmup + steps - 1, whose own successor's index work was enqueued inside as well: `steps` index plans, `steps`
    # steps.
    t0 = 0.0
    for it, b in enumerate(optim.prefetching(model, (dict(pool[i % 2]) for i in range(warmup + steps + 1)))):
        if it == warmup + steps:
            break
        loss, _ = optim.train_step(model, opt, sched, b, it, cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        if it == warmup - 1:
            if ddp:
                distributed.barrier(device)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if it >= warmup:
            losses.append(loss)
    if ddp:
        distributed.barrier(device)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ddp:
        dt = distributed.max_over_ranks(dt, device)
    dt /= steps
    inner = model.module if hasattr(model, "module") else model
    direct, executed = train_conv_flops(inner, int(ds.grid_size[1]), int(ds.grid_size[0]))
    if ddp:
        res = {"workload": f"hvpr_{which}.yaml full train step a1..a15 inside DistributedDataParallel over RCCL, batch={batch}/GPU x {world} GPUs, "
                           f"8 GT boxes/frame, {warmup} warm-up + {steps} timed steps, max over ranks between two barriers",
               "ms_per_step": round(1e3 * dt, 1), "frames_per_s_global": round(world * batch / dt, 2), "ranks": world,
               "allreduce_MB": round(grad_bytes / 1e6, 1), "parallelism": f"dp{world}", "scaling": "weak",
               "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)],
               "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
        del model, opt, pool
        torch.cuda.empty_cache()
        return res
    res = {"workload": f"hvpr_{which}.yaml full train step a1..a15, batch={batch}, 8 GT boxes/frame, {warmup} warm-up + {steps} timed steps",
           "steps_per_s": round(1.0 / dt, 3), "frames_per_s": round(batch / dt, 2), "ms_per_step": round(1e3 * dt, 1),
           "conv_direct_TFLOP_per_step": round(direct * batch / 1e12, 2), "conv_executed_TFLOP_per_step": round(executed * batch / 1e12, 2),
           "mfma_executed_TFLOPs": round(executed * batch / dt / 1e12, 1),
           "mfma_frac_of_f32_peak": round(executed * batch / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
           "mfma_frac_direct_count": round(direct * batch / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
           "flops_note": "mfma_frac_of_f32_peak = convolution FLOPs actually EXECUTED on the fp32 matrix cores (Winograd layers: 16/36 of "
                         "their direct count, forward, data gradient and weight gradient) / step time / 157.3; mfma_frac_direct_count = the "
                         "direct-convolution count over the same time (may exceed what is executed by 2.25x on the Winograd layers)",
           "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)],
           "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
           "kernels": "every training module on the library's HIP kernels: voxelizer, point-stream index ops + gathers, VFE forward/backward, "
                      "get_score top-k, memory addressing (once per point, then gathered), scatter, backbone + head convolutions fwd/dgrad/wgrad (Winograd F(2x2,3x3) where 3x3 "
                      "stride 1), train-mode BatchNorm, target assigner, the head's losses with their gradients, flat fused Adam"}
    del model, opt, pool
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gates (profiling runs); the JSON line then carries parity = null")
    ap.add_argument("--no-extras", action="store_true", help="skip the batch-16 / dense-scene group lines and the train-step line")
    ap.add_argument("--probe-steps", type=int, default=30, help="extra untimed steps with per-stage HIP events")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one whole-frame hipGraph per step (frame latency = step) instead of the 3-stage frame pipeline")
    ap.add_argument("--conv-precision", choices=["fp32", "bf16x3", "bf16x6"], default="fp32",
                    help="fp32: exact fp32 matrix-core convolutions; bf16x3: 3-term split-bf16 trunk/SFM convolutions")
    ap.add_argument("--skip-single", action="store_true", help="profiling: do not time the single-graph latency mode")
    ap.add_argument("--alt", action="store_true", help="also time the opt-in modes (bf16x6 / bf16x3 split-precision convolutions, direct "
                    "fp32 convolutions) and report them as `alt_precision` (never `value`); off by default: three more pipelines")
    ap.add_argument("--cls-bias", type=float, default=-4.59511985013459, help="conv_cls.bias of the synthetic weights")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: this process becomes the launcher — N fresh child processes, one rank per GPU (reference:
        # tools/train.py:61-70 / common_utils.py:141-154).  Nothing here has touched the GPU (device_count() does not).
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible")
        # a rank that hangs (a collective nobody else joins) must not hold the node: everything is ended after HVPR_BENCH_TIMEOUT
        # seconds (default 40 min: N = 8 needs ~6 min including the DDP train-step line)
        sys.exit(distributed.launch_local(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:],
                                          timeout=float(os.environ.get("HVPR_BENCH_TIMEOUT", "2400"))))
    rank, local_rank, world = distributed.env_rank()
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if torch.cuda.device_count() < max(world, 1):
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed.init("nccl", device)   # "nccl" IS RCCL on ROCm; a no-op at world size 1

    cfg = hvpr_car_cfg()
    ds = detector.SyntheticDataset(cfg)
    model = detector.build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    # conv_cls.bias keeps the reference's own initial value -log(99) (anchor_head_single.py:35-37); with these random
    # weights ~6-7 k anchors still pass SCORE_THRESH, so top-k always delivers NMS_PRE_MAXSIZE = 4096 candidates and NMS
    # keeps NMS_POST_MAXSIZE = 500: the post-processing runs at its maximum size
    params = synthetic_weights.load_synthetic(model, seed=0, cls_bias=args.cls_bias)
    model = model.to(device).eval()
    model.backbone_2d.set_conv_precision(args.conv_precision)

    frames = [synthetic.hvpr_frame(rank * 1000 + i) for i in range(N_POOL)]
    raw_pts = int(np.mean([len(synthetic.kitti_like_frame(rank * 1000 + i)) for i in range(2)]))
    batches = [make_batch(f, device) for f in frames]
    n_pts = len(frames[0])
    nx, ny = int(ds.grid_size[0]), int(ds.grid_size[1])

    dt_local = [0.0]          # this rank's own time of the last timed() call

    def barrier():
        distributed.barrier(device)

    # one hipGraph per frame: every data-dependent size on this path is a device word, so the launch sequence is static
    def timed(step):
        for i in range(args.warmup):
            step(batches[i % N_POOL])
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(batches[(args.warmup + i) % N_POOL])
        barrier()
        dt_local[0] = time.perf_counter() - t0
        return distributed.max_over_ranks(dt_local[0], device)

    alt = None
    timed_object = None
    with torch.no_grad():
        if args.no_graph:
            mode = "eager launches"
            dt = timed(lambda b: model(dict(b), sync=False))
            dt_single = None
        else:
            # latency mode: one whole-frame hipGraph per step (a1..a8 of ONE frame, serial)
            dt_single = None if (args.skip_single and not args.no_pipeline) else timed(detector.GraphedForward(model, batches[0]))
            mode, dt = "one whole-frame hipGraph replay per step (frame latency = 1 step)", dt_single
            if not args.no_pipeline:
                # throughput mode: every step is one graph replay that encodes frame k, convolves frames k-1 / k-2 (two halves of
                # the backbone) and runs top-k + NMS of frame k-3 on separate streams; K timed steps retire exactly K frames (the
                # pipeline is full before and after the timed region; the warm-up fills it), nothing is skipped
                pipe = detector.PipelinedForward(model, batches[0])
                dt = timed(pipe)
                for _ in pipe.flush():
                    pass
                pipe.check_status()                       # (synchronising) no index kernel gave up a wait during the timed replays
                timed_object = pipe                       # the parity gates below run on THIS object
                mode = ("4-stage frame pipeline: one hipGraph replay per step = encode(k) | trunk + branches 0,1 (k-1) | last branch + head + "
                        "decode (k-2) | top-k+NMS (k-3) on separate HIP streams (frame latency = 4 steps)") if pipe.depth == 4 else \
                       ("3-stage frame pipeline: one hipGraph replay per step = encode(k) | convolutions(k-1) | top-k+NMS(k-2) "
                        "on three HIP streams (frame latency = 3 steps)")

        dt_rank = dt_local[0]      # this rank's own time of the headline run (`dt` is the max over ranks)
        # reported next to the headline, never as `value` (which stays one frame per step, hvpr_car.yaml batch = 1): the same pipeline
        # with TWO frames per graph replay — every convolution launch sees twice the tiles (4.6 / 2.5 / 1.25 rounds of the resident
        # workgroups instead of 2.3 / 1.25 / 0.63), frame latency doubles, per-frame results are those of batch 1 bit for bit
        # (tests/test_gpu_e2e.py: a frame alone == the frame inside a batch)
        def run_two_frames_per_replay():
            """Runs AFTER the parity gates: a batch-2 pipeline re-allocates the model's post-processing workspace, which the batch-1
            graphs that were timed (and that the gates re-use) hold by address."""
            two = None
            if world == 1 and not args.no_extras and not args.no_graph and not args.no_pipeline and args.conv_precision == "fp32":
                def pair(i):
                    a, b = frames[i % N_POOL], frames[(i + 1) % N_POOL]
                    pts = np.concatenate([np.concatenate([np.full((len(f), 1), j, np.float32), f], 1) for j, f in enumerate((a, b))])
                    return {"points": torch.from_numpy(pts).to(device), "batch_size": 2,
                            "point_frame_offsets": torch.tensor([0, len(a), len(a) + len(b)], dtype=torch.int32, device=device)}
                pairs = [pair(2 * i) for i in range(N_POOL // 2)]
                p2 = detector.PipelinedForward(model, pairs[0])
                n2 = max(args.steps // 2, 1)
                for i in range(max(args.warmup // 2, 4) + 60):   # (the GPU has idled through the CPU baseline: let the clocks come back up)
                    p2(pairs[i % len(pairs)])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n2):
                    p2(pairs[i % len(pairs)])
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
                for _ in p2.flush():
                    pass
                del p2, pairs
                two = {"value": round(2 * n2 / dt2, 2), "unit": "frames/s", "ms_per_replay_of_two_frames": round(1e3 * dt2 / n2, 4),
                       "what": "PipelinedForward with two frames per graph replay (batch 2 through every stage); NOT `value`"}
            return two
        def run_alt_modes():
            """The opt-in precision / algorithm modes (--alt), AFTER the parity gates: switching the convolution mode re-packs the
            weights the timed graphs hold by address."""
            alt = None
            if args.alt and world == 1 and not args.no_graph and not args.no_pipeline and args.conv_precision == "fp32":
                # reported next to the headline, never as `value`: the same pipeline with the trunk/SFM 3x3 convolutions on the bf16
                # matrix cores with split operands
                alt = []
                for prec, what, tol in (
                        ("bf16x6", "operands split into 3 bf16 planes (exact), 6 products, fp32 accumulate: fp32 emulation — per-layer error vs "
                                   "float64 1.2e-6..1.8e-6, the same as the exact fp32 kernel (1.3e-6..1.8e-6)", "fp32-grade (tests/test_gpu_conv.py: 4e-6)"),
                        ("bf16x3", "operands split into 2 bf16 planes, 3 products, fp32 accumulate: ~2^-16 per product",
                         "features / boxes within 1e-3 relative of the fp32 path (tests/test_gpu_e2e.py), observed ~1e-5")):
                    model.backbone_2d.set_conv_precision(prec)
                    p3 = detector.PipelinedForward(model, batches[0])
                    dt3 = timed(p3)
                    for _ in p3.flush():
                        pass
                    del p3
                    sg3, bb_ms = StagedGraphs(model, batches[0]), 0.0       # backbone + head + decode alone, HIP events
                    for i in range(10):
                        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                        sg3.run(batches[i % N_POOL], ev)
                        torch.cuda.synchronize()
                        bb_ms += ev[1].elapsed_time(ev[2]) / 10
                    del sg3
                    nprod = {"bf16x6": 6, "bf16x3": 3}[prec]
                    fl_all, fl_split = conv_flops(model, int(ds.grid_size[1]), int(ds.grid_size[0])), split_conv_flops(model, int(ds.grid_size[1]), int(ds.grid_size[0]), prec)
                    executed = nprod * fl_split          # bf16 products executed on the bf16 matrix cores (the rest runs on the fp32 kernel)
                    alt.append({"mode": prec, "what": what + "; v_mfma_f32_32x32x16_bf16; everything else as in `value` (opt-in: "
                                "HVPR_CONV_PRECISION=" + prec + ")", "value": round(world * args.steps / dt3, 2), "unit": "frames/s",
                                "ms_per_step": round(1e3 * dt3 / args.steps, 4), "tolerance": tol,
                                "mfma": {"backbone_head_decode_ms": round(bb_ms, 4), "algorithmic_TFLOPs": round(fl_all / (bb_ms * 1e-3) / 1e12, 1),
                                         "executed_bf16_TFLOPs": round(executed / (bb_ms * 1e-3) / 1e12, 1), "bf16_dense_peak_TFLOPs": 2500.0,
                                         "frac_of_bf16_peak": round(executed / (bb_ms * 1e-3) / 1e12 / 2500.0, 4),
                                         "note": "executed = products x FLOPs of the layers that run split; fp32 peak for comparison 157.3"}})
                model.backbone_2d.set_conv_precision("fp32")
                # the same pipeline with every convolution on the direct fp32 kernel (round 1's configuration)
                os.environ["HVPR_CONV_ALGO"] = "direct"
                model.backbone_2d._fold.invalidate()
                pd = detector.PipelinedForward(model, batches[0])
                dtd = timed(pd)
                for _ in pd.flush():
                    pass
                del pd
                del os.environ["HVPR_CONV_ALGO"]
                model.backbone_2d._fold.invalidate()
                alt.append({"mode": "conv_algo_direct", "what": "HVPR_CONV_ALGO=direct: the stride-1 3x3 layers on the direct implicit-GEMM kernel "
                            "(hvpr_conv2d_nhwc_f32) instead of Winograd F(2x2,3x3); same fp32 arithmetic, 2.25x the multiplies",
                            "value": round(world * args.steps / dtd, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dtd / args.steps, 4)})

            return alt

        # ---- per-stage probe (untimed): HIP events on the launch stream ----
        stage = np.zeros(3)
        n_pillars = 0
        out = None
        staged = StagedGraphs(model, batches[0]) if args.probe_steps > 0 else None
        # The probe steps are enqueued back to back and read after ONE synchronize at the end: while the GPU runs step i the
        # host has long enqueued step i + 1, so the interval between two events holds GPU time only (with a synchronize
        # per step the first interval also counted the host's graph-launch latency, ~10-15 us of idle GPU).
        evs, pillars_dev = [], torch.zeros((), dtype=torch.int64, device=device)
        for i in range(args.probe_steps):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            out = staged.run(batches[i % N_POOL], ev)
            pillars_dev += out[2]["voxel_offsets"][-1]
            evs.append(ev)
        torch.cuda.synchronize()
        for ev in evs:
            stage += [ev[k].elapsed_time(ev[k + 1]) for k in range(3)]
        n_pillars = float(pillars_dev.item()) / max(args.probe_steps, 1)
        stage /= max(args.probe_steps, 1)
        # The VFE+scatter group alone: R consecutive replays of its captured graph between two HIP events (a different
        # frame per repetition), so that the interval is the group's kernels and the gaps between them — a single replay between two
        # events also counts the ~15 us the command processor needs to start a graph launch, which a frame pays once for
        # its ~50 kernels, not per group (rocprofv3 kernel durations of the same command: profiles/).
        group_us = None
        if staged is not None:
            # eight captured copies of the group, one per pool frame, on ONE persistent canvas pair: consecutive replays see
            # consecutive DIFFERENT frames (the stale-cell clear of the previous frame's pillars is inside the interval)
            group_us = GroupGraphs(model, batches).time_us(rounds=5)
        kept = int(out[0][0]["pred_count"].item()) if args.probe_steps > 0 else -1

    rccl_ranks = distributed.ranks_seen(device)                       # all-reduce of ones over RCCL
    per_rank_fps = [round(args.steps / t, 2) for t in distributed.gather_floats(dt_rank, device)]
    # N > 1: the exchange step the path really has — BASELINE.json configs[3]: hvpr 3-class, batch 8 per GPU, data parallel over
    # RCCL (one gradient all-reduce of ~62 MB per step, reference tools/train.py:143-145) — timed on every rank
    train_ddp = None
    if world > 1 and not args.no_extras:
        staged = None
        torch.cuda.empty_cache()
        train_ddp = train_step_line(device, "3class", 8, ddp=True)
    if rank != 0:
        distributed.finalize()
        return

    fps = world * args.steps / dt
    # algorithmic bytes of the VFE+scatter group per frame (SURVEY.md §8d / BASELINE.md §6): raw points read once +
    # dense canvases written once (zeros included) + VFE weights and memory bank read once
    group_bytes_ = group_bytes(n_pts, nx, ny, 1)
    group_s = (group_us * 1e-6) if group_us else stage[0] * 1e-3
    flops = conv_flops(model, ny, nx)
    executed = flops - wino_conv_flops(model, ny, nx) * (1.0 - 16.0 / 36.0)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))
    # per-kernel averages of the committed rocprofv3 --kernel-trace --stats run of this command (serial single-stream frame graph,
    # HVPR_BEV_STREAMS=1 --no-pipeline): the group above is timed live; its members are quoted from the profile
    members = {}
    import glob
    import re

    def _round_of(path):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        return int(m.group(1)) if m else -1
    spaths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_serial.csv")), key=_round_of)   # newest ROUND, numerically
    spath = spaths[-1] if spaths else ""
    if os.path.exists(spath):
        import csv
        for r in csv.DictReader(open(spath)):
            for k in ("k_index", "k1_keys", "k2_scan", "k3_fill", "k_vfe", "k_memory_readout"):
                if k in r["Name"]:
                    members[k] = round(float(r["AverageNs"]) / 1e3, 2)
    # `achieved` / `frac` / `avg_duration_us` are the LIVE measurement of this run (HIP events, the group replayed alone).  The group
    # INSIDE the frame (caches as the convolution stage leaves them) = sum of its members' rocprofv3 durations in the committed serial
    # single-stream frame profile is reported beside it, labelled with its file — and flagged when it no longer describes the code
    # that is being timed (a profile that was not regenerated after a kernel change): more than 25 % away from the live figure.
    in_frame_us = round(sum(members.values()), 2) if len(members) >= 3 else None
    isolated_us = group_s * 1e6
    profile_consistent = None if in_frame_us is None else bool(abs(in_frame_us - isolated_us) <= 0.25 * isolated_us)
    res = {
        "metric": "KITTI frames/sec/GPU (fwd, ~20k pts); VFE+scatter achieved HBM GB/s vs peak",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "rccl_ranks_seen": rccl_ranks, "per_rank_frames_per_s": per_rank_fps,
        "config": {"workload": "hvpr_car.yaml batch=1 forward-only, a1..a8 (on-GPU voxelize, pillar VFE, memory read-out + "
                               "scatter, BEV backbone, head + decode, score top-k + rotated-BEV NMS); frames sharded "
                               "over ranks, replicas only",
                   "frame": f"synthetic 64-beam LiDAR, ~{raw_pts} raw points -> range mask -> {n_pts} sampled points, "
                            f"~{int(n_pillars)} pillars, grid {nx}x{ny}x1", "global_batch": world,
                   "weights": f"deterministic synthetic (seed 0), BN stats randomised, cls bias {args.cls_bias:.3f}",
                   "nms_candidates_kept": kept, "launch": mode},
        "single_graph_latency_mode": None if dt_single is None else {"value": round(world * args.steps / dt_single, 2),
                                                                     "unit": "frames/s", "ms_per_frame": round(1e3 * dt_single / args.steps, 4)},
        "stage_ms": {"voxelize+vfe+memory+scatter": round(float(stage[0]), 4), "backbone+head+decode": round(float(stage[1]), 4),
                     "topk+nms": round(float(stage[2]), 4)},
        "roofline": {"kernel": "VFE+scatter group = hvpr_encode_fwd_f32 (index_mode 1): k_index (cell keys, rank scan, arena fill as the phases of one launch), k_vfe_gather (voxel gather + pillar VFE + canvas clear), k_memory_readout (+ canvas cells): 3 launches (5 with index_mode 0 or beyond 32 768 points)",
                     "bound": "hbm", "achieved": round(group_bytes_ / group_s / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(group_bytes_ / group_s / 1e9 / HBM_PEAK_GBPS, 5), "algorithmic_bytes": group_bytes_,
                     "avg_duration_us": round(group_s * 1e6, 2),
                     "avg_duration_is": "live: HIP events around replays of the group alone in this run, caches warm",
                     "in_frame_profile": {"file": os.path.basename(spath) if spath else None, "sum_of_members_us": in_frame_us,
                                          "frac_at_that_duration": None if not in_frame_us else round(group_bytes_ / (in_frame_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
                                          "consistent_with_live_within_25pct": profile_consistent,
                                          "what": "sum of the group's member kernels in the committed rocprofv3 --kernel-trace --stats summary of the serial "
                                                  "single-stream frame (tools/prof_serial.sh): the group as it runs inside a frame, its inputs and weights "
                                                  "evicted by the convolution stage; a stale file shows up as consistent_with_live_within_25pct = false"},
                     "isolated_warm_us_live": round(isolated_us, 2),
                     "single_replay_between_events_us": round(float(stage[0]) * 1e3, 2),
                     "timing": "HIP events around 5 replays of ONE captured hipGraph that holds the group of the 8 pool frames back to back (on ONE "
                               "persistent canvas pair: every group encodes a DIFFERENT frame, so the stale-cell clear of the previous frame is inside "
                               "the interval), / 40; single_replay_between_events_us additionally holds the start-up of one graph launch",
                     "traffic": None if traffic is None else traffic.get("vfe_scatter_group_bytes"),
                     "achieved_physical": None if traffic is None else round(traffic.get("vfe_scatter_group_bytes") / group_s / 1e9, 2),
                     "frac_physical": None if traffic is None else round(traffic.get("vfe_scatter_group_bytes") / group_s / 1e9 / HBM_PEAK_GBPS, 5),
                     "note": "`achieved` / `frac` = ALGORITHMIC bytes (SURVEY.md §8d: points read once + dense canvases written once + weights) / "
                             "measured group time; `achieved_physical` = HBM bytes the counters saw (`traffic`, rocprofv3 --pmc of the committed "
                             "profile, FETCH_SIZE doubled per the gfx950 correction) / the same time — lower, because persistent canvases are not "
                             "re-written where nothing changed",
                     "member_kernels_avg_us_from_profile": members,
                     "canvas": "persistent canvases + occupancy state (hvpr_encode_fwd_f32 canvas_state): the dense result is the same, "
                               "but only stale cells are cleared — `traffic` (PMC) is therefore BELOW the algorithmic bytes, which still "
                               "count the dense canvases written once; dense-clear variant: tools/bench_group.py --dense-clear"},
        "roofline_mfma": {"kernel": "BEV backbone + head convolutions on v_mfma_f32_32x32x2_f32: hvpr_conv2d_wino_nhwc_f32 (Winograd F(2x2,3x3), "
                                    "the stride-1 3x3 layers) + hvpr_conv2d_nhwc_f32 (stride-2 3x3, 1x1, ConvTranspose)",
                          "bound": "mfma", "achieved": round(executed / (stage[1] * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": round(executed / (stage[1] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                          "executed_flops": executed, "algorithmic_flops": flops,
                          "algorithmic_TFLOPs": round(flops / (stage[1] * 1e-3) / 1e12, 2),
                          "avg_duration_us": round(float(stage[1]) * 1e3, 1),
                          "traffic": None if traffic is None else traffic.get("conv_stack_bytes"),
                          "note": "`achieved` / `frac` = matrix-core FLOPs actually EXECUTED (the Winograd layers issue 16/36 of their "
                                  "direct-convolution count) / stage time; `algorithmic_TFLOPs` = the direct-convolution count of the same "
                                  "layers (SURVEY.md §8a: 453.65 GFLOP/frame) / the same time — it may exceed the peak, which is the point of "
                                  "the algorithm.  peak = 157.3 TFLOP/s at 2.4 GHz; under this load the shader clock measured 2.02 GHz "
                                  "(DESIGN.md §4.2), i.e. ~132 TFLOP/s attainable.  alt_precision[mode=conv_algo_direct] is the same frame "
                                  "with every layer on the direct kernel"},
    }
    res["train_step_ddp"] = train_ddp
    # cpu_baseline + parity gates while the timed object is still alive: the oracle's frames go through it as well
    gates = None
    if not args.no_parity:
        with torch.no_grad():
            gates = ParityGates(timed_object.inspect if timed_object is not None else (lambda b: eager_inspect(model, b)), cfg.MODEL, device)
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(cfg, params, check=_no_grad(gates))
    else:
        res["cpu_baseline"] = None
        if gates is not None:                     # N > 1 (or --no-cpu-baseline): the gates alone, on two frames
            from oracle import hvpr_oracle as O
            for i in range(2):
                f = synthetic.hvpr_frame(1000 + i)
                _no_grad(gates)(f, *O.forward_frames([f], params, O.cfg_from_model_cfg(cfg)))
    res["parity"] = None if gates is None else gates.result()
    del timed_object
    with torch.no_grad():
        res["pipeline_two_frames_per_replay"] = run_two_frames_per_replay()
        res["alt_precision"] = run_alt_modes()
    if world == 1 and not args.no_extras:
        # driver-visible numbers for the other BASELINE.json configs (bounded step counts): the same group at batch 16 and on
        # the dense scene (configs[4]); the full train step (configs[2])
        del model, batches, staged
        torch.cuda.empty_cache()
        res["group_other_configs"] = extra_group_lines(device)
        res["train_step"] = train_step_line(device, "car", 16)
        res["train_step_3class_b8"] = train_step_line(device, "3class", 8)          # BASELINE.json configs[3], one GPU's share
    print(json.dumps(res), flush=True)
    distributed.finalize()
    if res["parity"] is not None and not res["parity"]["ok"]:
        print("bench.py: PARITY GATE FAILED: " + json.dumps({k: v for k, v in res["parity"].items() if k != "what"}), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
