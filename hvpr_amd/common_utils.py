"""Leaf helpers of pcdet/utils/common_utils.py that the hot path uses."""
import numpy as np
import torch


def limit_period(val, offset=0.5, period=np.pi):
    """common_utils.py:20-23."""
    is_np = isinstance(val, np.ndarray)
    t = torch.from_numpy(val).float() if is_np else val
    out = t - torch.floor(t / period + offset) * period
    return out.numpy() if is_np else out


def mask_points_by_range(points, limit_range):
    """common_utils.py:59-62."""
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) & \
           (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])
