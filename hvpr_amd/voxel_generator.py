"""GPU voxel generator with the spconv interface the reference's dataloader expects
(pcdet/datasets/processor/data_processor.py:45-67):

    VoxelGenerator(voxel_size=, point_cloud_range=, max_num_points=, max_voxels=).generate(points)
        -> {'voxels', 'coordinates' (zyx int32), 'num_points_per_voxel'}        (V2 form, a dict)

`generate` accepts a numpy array (drop-in for the CPU dataloader: H2D, kernels, D2H) or a device tensor (stays on the
device).  `generate_batch` is the on-device form the detector uses: no host round trip at all."""
import numpy as np
import torch

from . import kernels


def grid_size_of(point_cloud_range, voxel_size):
    """data_processor.py:56-57 with the range held in float32 (dataset.py:25)."""
    r = np.asarray(point_cloud_range, dtype=np.float32)
    return np.round((r[3:6] - r[0:3]) / np.asarray(voxel_size, dtype=np.float32)).astype(np.int64)


class VoxelGenerator:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels, full_mean=False, block_filtering=False,
                 device="cuda:0", v1_break=False):
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.max_num_points = int(max_num_points)
        self.max_voxels = int(max_voxels)
        self.grid_size = grid_size_of(point_cloud_range, voxel_size)
        self.device = torch.device(device)
        self.cap_mode = 1 if v1_break else 0
        self._ws = None

    def _workspace(self, batch, n):
        if self._ws is None or self._ws.key[0] < batch or self._ws.key[1] < n:
            old = self._ws.key if self._ws is not None else (1, 1)
            self._ws = kernels.VoxelizeWorkspace(max(batch, old[0]), max(n, old[1], 1), self.grid_size, self.device)
        return self._ws

    def generate_batch(self, points, frame_offsets, batch, xyz_col=0, n_feat=None):
        """points (N, stride) device f32, frames contiguous; frame_offsets (batch+1,) device i32.
        Returns voxels, coords [b,z,y,x] i32, num_points i32, voxel_offsets (batch+1,) i32 (all device)."""
        ws = self._workspace(batch, points.shape[0])
        return kernels.voxelize(points, frame_offsets, batch, self.point_cloud_range, self.voxel_size, self.grid_size,
                                self.max_num_points, self.max_voxels, ws, xyz_col=xyz_col, n_feat=n_feat,
                                cap_mode=self.cap_mode)

    def generate(self, points, max_voxels=None):
        is_np = isinstance(points, np.ndarray)
        p = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(self.device) if is_np else points.contiguous()
        offs = torch.tensor([0, p.shape[0]], dtype=torch.int32, device=p.device)
        v, c, n, vo = self.generate_batch(p, offs, 1)
        m = int(vo[1].item())
        out = {"voxels": v[:m], "coordinates": c[:m, 1:], "num_points_per_voxel": n[:m]}
        if is_np:
            out = {k: t.cpu().numpy() for k, t in out.items()}
        return out
