"""The CPU natives of the reference's GT-sampling augmentation ("next" row f4 of SURVEY.md §8f), with the names and
signatures their call sites use:

  roiaware_pool3d_utils.points_in_boxes_cpu(points (N,3), boxes (M,7)) -> (M,N) int   (box_utils.py:85, kitti_dataset.py:217)
  iou3d_nms_utils.boxes_bev_iou_cpu(boxes_a (N,7), boxes_b (M,7)) -> (N,M) float32     (database_sampler.py:184-185)
  box_utils.remove_points_in_boxes3d(points, boxes3d)                                   (box_utils.py:74-88)

Host code (libhvpr_cpu.so, C++ through ctypes) for data-loader workers; numpy in, numpy out."""
import ctypes
import os

import numpy as np

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhvpr_cpu.so")
        if not os.path.exists(path):
            raise RuntimeError("hvpr_amd: libhvpr_cpu.so is missing — run `python -m hvpr_amd.build`")
        _LIB = ctypes.CDLL(path)
        _LIB.hvpr_points_in_boxes_cpu.restype = ctypes.c_int
        _LIB.hvpr_points_in_boxes_cpu.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _LIB.hvpr_boxes_bev_iou_cpu.restype = ctypes.c_int
        _LIB.hvpr_boxes_bev_iou_cpu.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return _LIB


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def points_in_boxes_cpu(points, boxes):
    p = np.ascontiguousarray(_np(points), np.float32)
    b = np.ascontiguousarray(_np(boxes)[:, :7], np.float32)
    out = np.zeros((b.shape[0], p.shape[0]), np.int32)
    if _lib().hvpr_points_in_boxes_cpu(p.ctypes.data, p.shape[0], p.shape[1], b.ctypes.data, b.shape[0], out.ctypes.data) != 0:
        raise ValueError("points_in_boxes_cpu: invalid arguments")
    return out


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    a = np.ascontiguousarray(_np(boxes_a)[:, :7], np.float32)
    b = np.ascontiguousarray(_np(boxes_b)[:, :7], np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    if _lib().hvpr_boxes_bev_iou_cpu(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], out.ctypes.data) != 0:
        raise ValueError("boxes_bev_iou_cpu: invalid arguments")
    return out


def remove_points_in_boxes3d(points, boxes3d):
    """box_utils.py:74-88: drop every point that lies in any of the boxes."""
    masks = points_in_boxes_cpu(_np(points)[:, 0:3], boxes3d)
    return _np(points)[masks.sum(axis=0) == 0]
