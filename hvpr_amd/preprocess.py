"""Point pre-processing in front of the voxelizer, on the device ("next" row f2 of SURVEY.md §8f) plus the host->device
input path (row f1).  Mirrors, with the reference's names:

  * mask_points_by_range               pcdet/utils/common_utils.py:59-62
  * DataProcessor.mask_points_and_boxes_outside_range / sample_points
                                       pcdet/datasets/processor/data_processor.py:20-30,77-108
  * KittiDataset.get_fov_flag          pcdet/datasets/kitti/kitti_dataset.py:100-116 (over Calibration.lidar_to_rect /
                                       rect_to_img, pcdet/utils/calibration_kitti.py:65-84)
  * collate_batch's `points` branch + load_data_to_gpu
                                       pcdet/datasets/dataset.py:148-180 and the absent pcdet/models/__init__.py

Flags, stable compaction and the row gather are HIP kernels (csrc/preprocess.hip).  `sample_points` draws its indices on
the host with the SAME sequence of numpy RNG calls as the reference (test.py:47 seeds numpy even at test time), from near
flags computed on the device — bit-identical post-sampling points for a given seed.
"""
import numpy as np
import torch

from . import kernels
from ._lib import check, lib


def _flags(points, mode, range_xy=None, near=40.0, fov=None):
    n = points.shape[0]
    flags = torch.empty((n,), dtype=torch.uint8, device=points.device)
    a = p = None
    h = w = 0
    if fov is not None:
        a, p, (h, w) = fov
    check(lib().hvpr_point_flags_f32(kernels._ptr(points, torch.float32, "points"), n, points.shape[1], mode,
                                     kernels._ptr(range_xy, torch.float32), float(near), kernels._ptr(a, torch.float32),
                                     kernels._ptr(p, torch.float32), int(h), int(w), flags.data_ptr(), kernels._stream()),
          "hvpr_point_flags_f32")
    return flags


def mask_points_by_range(points, limit_range):
    """points (N, >=3) f32 cuda -> uint8 mask (x and y inside [lo, hi], inclusive)."""
    r = torch.as_tensor(np.asarray(limit_range, np.float32)[[0, 1, 3, 4]], device=points.device)
    return _flags(points.contiguous(), 0, r)


def fov_matrices(calib, device):
    """(V2C^T R0^T) and P2^T as the reference forms them (float32 numpy products), on the device."""
    V2C, R0, P2 = (np.asarray(calib[k], np.float32) for k in ("Tr_velo2cam", "R0", "P2"))
    a = np.ascontiguousarray(np.dot(V2C.T, R0.T).astype(np.float32))
    p = np.ascontiguousarray(P2.T.astype(np.float32))
    return torch.from_numpy(a).to(device), torch.from_numpy(p).to(device)


def get_fov_flag(points, calib, img_shape):
    """uint8 mask of the points that project into the image (and lie in front of the camera)."""
    a, p = fov_matrices(calib, points.device)
    every = torch.tensor([-3.0e38, -3.0e38, 3.0e38, 3.0e38], device=points.device)
    return _flags(points.contiguous(), 0, every, fov=(a, p, (int(img_shape[0]), int(img_shape[1]))))


def compact_rows(rows, flags, count_only=False):
    """rows[flags != 0], stable.  Returns (out, count): `out` has rows.shape[0] rows of which the first `count` (device
    int32) are live — no host synchronisation; slice with int(count) when a host size is needed."""
    rows = rows.contiguous()
    n, w = rows.shape
    out = torch.empty_like(rows)
    count = torch.zeros((1,), dtype=torch.int32, device=rows.device)
    ws = torch.empty((lib().hvpr_compact_workspace_bytes(n),), dtype=torch.uint8, device=rows.device)
    check(lib().hvpr_compact_rows_f32(kernels._ptr(rows, torch.float32, "rows"), n, w, kernels._ptr(flags, torch.uint8, "flags"),
                                      out.data_ptr(), n, count.data_ptr(), ws.data_ptr(), ws.numel(), kernels._stream()),
          "hvpr_compact_rows_f32")
    return out, count


def gather_rows(rows, idx):
    rows = rows.contiguous()
    idx = idx.to(torch.int32).contiguous()
    out = torch.empty((idx.shape[0], rows.shape[1]), dtype=torch.float32, device=rows.device)
    check(lib().hvpr_gather_rows_f32(kernels._ptr(rows, torch.float32, "rows"), rows.shape[0], rows.shape[1],
                                     kernels._ptr(idx, torch.int32, "idx"), idx.shape[0], out.data_ptr(), kernels._stream()),
          "hvpr_gather_rows_f32")
    return out


def sample_points_choice(near, num_points, rng=np.random):
    """Index selection of DataProcessor.sample_points (data_processor.py:77-108): same RNG call sequence, including the
    draw at :91 that happens before the size test (so, like the reference, this raises ValueError when more than
    num_points points lie beyond 40 m)."""
    n = len(near)
    if num_points == -1:
        return np.arange(n)
    every = np.arange(0, n, dtype=np.int32)
    if num_points < n:
        far_idx, near_idx = np.where(near == 0)[0], np.where(near == 1)[0]
        want = num_points - len(far_idx)
        drawn = rng.choice(near_idx, want, replace=False)
        if want > 0:
            drawn = rng.choice(near_idx, want, replace=False)
            choice = np.concatenate((drawn, far_idx), axis=0) if len(far_idx) > 0 else drawn
        else:
            choice = rng.choice(every, num_points, replace=False)
    else:
        choice = every
        if num_points > n:
            choice = np.concatenate((choice, rng.choice(choice, num_points - n, replace=False)), axis=0)
    rng.shuffle(choice)
    return choice


class PointPreprocessor:
    """The device-side DataProcessor steps of kitti_dataset.yaml that precede `transform_points_to_voxels`
    (which MixAnchor_Memory.forward performs itself): FOV filter, range mask, sample_points."""

    def __init__(self, point_cloud_range, num_points=16384, training=False, fov_points_only=True):
        self.point_cloud_range = np.asarray(point_cloud_range, np.float32)
        self.num_points, self.training, self.fov_points_only = num_points, training, fov_points_only

    def mask_points_and_boxes_outside_range(self, points):
        out, count = compact_rows(points, mask_points_by_range(points, self.point_cloud_range))
        return out[: int(count.item())]

    def fov_filter(self, points, calib, img_shape):
        out, count = compact_rows(points, get_fov_flag(points, calib, img_shape))
        return out[: int(count.item())]

    def sample_points(self, points, rng=np.random):
        if self.num_points == -1:
            return points
        near = _flags(points.contiguous(), 1, near=40.0).cpu().numpy() if self.num_points < points.shape[0] else \
            np.ones((points.shape[0],), np.uint8)
        choice = sample_points_choice(near, self.num_points, rng)
        return gather_rows(points, torch.from_numpy(np.ascontiguousarray(choice).astype(np.int32)).to(points.device))

    def __call__(self, points, calib=None, img_shape=None, rng=np.random):
        if self.fov_points_only and calib is not None:
            points = self.fov_filter(points, calib, img_shape)
        points = self.mask_points_and_boxes_outside_range(points)
        return self.sample_points(points, rng)


# ------------------------------------------------------------------------------------------------ f1: host -> device
class InputPipeline:
    """collate_batch's `points` branch (dataset.py:161-166: a batch-index column in front of every frame's rows) and
    load_data_to_gpu, as a double-buffered pinned-memory upload on its own HIP stream: frame k+1 is copied while frame k
    computes.  `put(frames)` stages a list of (N_i, C) float32 arrays; `get()` returns the batch_dict of the OLDEST staged
    batch with device tensors `points` (sum N_i, C+1) and `point_frame_offsets` (B+1,) int32."""

    def __init__(self, max_points, n_feat=4, max_batch=1, device="cuda:0", depth=2):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.slots = [{"host": torch.empty((max_points, n_feat + 1), dtype=torch.float32).pin_memory(),
                       "dev": torch.empty((max_points, n_feat + 1), dtype=torch.float32, device=self.device),
                       "off_host": torch.zeros((max_batch + 1,), dtype=torch.int32).pin_memory(),
                       "off_dev": torch.zeros((max_batch + 1,), dtype=torch.int32, device=self.device),
                       "ready": torch.cuda.Event(), "free": torch.cuda.Event(), "n": 0, "b": 0} for _ in range(depth)]
        self.head = self.tail = 0

    def put(self, frames):
        s = self.slots[self.head % len(self.slots)]
        s["free"].synchronize()                    # the batch that used this slot has been consumed by the GPU
        n = 0
        for b, f in enumerate(frames):
            m = len(f)
            s["host"][n:n + m, 0] = float(b)
            s["host"][n:n + m, 1:] = torch.from_numpy(np.ascontiguousarray(f, np.float32))
            s["off_host"][b] = n
            n += m
        s["off_host"][len(frames)] = n
        s["n"], s["b"] = n, len(frames)
        with torch.cuda.stream(self.stream):
            s["dev"][:n].copy_(s["host"][:n], non_blocking=True)
            s["off_dev"][: len(frames) + 1].copy_(s["off_host"][: len(frames) + 1], non_blocking=True)
            s["ready"].record(self.stream)
        self.head += 1

    def get(self):
        assert self.tail < self.head, "InputPipeline.get() without a staged batch"
        s = self.slots[self.tail % len(self.slots)]
        self.tail += 1
        torch.cuda.current_stream(self.device).wait_event(s["ready"])
        return {"points": s["dev"][: s["n"]], "point_frame_offsets": s["off_dev"][: s["b"] + 1], "batch_size": s["b"]}, s

    @staticmethod
    def release(slot):
        """Call after the consumer's kernels that read the slot are enqueued."""
        slot["free"].record(torch.cuda.current_stream())
