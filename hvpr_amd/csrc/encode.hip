// a1..a4 fused: raw points -> dense BEV canvases in five launches, three with index_mode 1 (hvpr_encode_fwd_f32, include/hvpr_amd.h).
//   K1 keys | K2 rank scan | K3 arena fill      voxelizer index kernels (voxelize.hip); index_mode 1: k_index, the three as phases of one launch
//   k_vfe<gather>                               voxel gather + pillar VFE + pillar/scale cells of the canvases (vfe.hip)
//   k_memory_readout                            memory read-out + memory cells of the main canvas (memory_scatter.hip)
//   Extra workgroups of the VFE launch clear every canvas cell that belongs to no pillar (47 MB at hvpr_car).
// Replaces the reference chain data_processor.py:43-75 -> pillar_vfe.py:184-221 -> memory_module.py:60-77 ->
// pointpillar_scatter.py:169-222.  Results are bit-identical to the three separate C-ABI calls (tests/test_gpu_stage1.py).
#include "common.h"
#include "internal.h"

extern "C" int hvpr_encode_fwd_f32(const float *points, int n_points, int point_stride, int xyz_col, int n_feat,
                                   const int32_t *frame_offsets, int batch, float lo_x, float lo_y, float lo_z, float vs_x,
                                   float vs_y, float vs_z, int nx, int ny, int nz, int max_points, int max_voxels,
                                   int cap_mode, float off_x, float off_y, float off_z, const float *w0, const float *b0,
                                   const float *w1, const float *b1, const float *ws0, const float *bs0, const float *ws1,
                                   const float *bs1, const float *bank, const float *bank_packed, int n_items, int k, float *voxels, int32_t *coords,
                                   int32_t *num_points, int32_t *voxel_offsets, int capacity, float *pillar_features,
                                   float *pillar_scale_features, float *pillar_mask, float *memory_features, float *spatial,
                                   float *spatial_scale, uint8_t *canvas_state, void *workspace, size_t workspace_bytes,
                                   int ws_max_batch, int ws_max_points, int index_mode, hvpr_stream_t stream) {
    if (!points || !frame_offsets || !coords || !num_points || !voxel_offsets || !workspace || !w0 || !b0 || !w1 || !b1 ||
        !ws0 || !bs0 || !ws1 || !bs1 || !bank || !pillar_features || !pillar_scale_features || !memory_features || !spatial ||
        !spatial_scale)
        return HVPR_ERR_INVALID_ARG;
    if (batch < 1 || n_points < 0 || xyz_col < 0 || point_stride < xyz_col + n_feat || nx < 1 || ny < 1 || nz < 1 ||
        max_points < 1 || max_voxels < 1 || capacity < 0 || (cap_mode != 0 && cap_mode != 1) || n_items < 1 || k < 1 ||
        (index_mode != 0 && index_mode != 1))
        return HVPR_ERR_INVALID_ARG;
    if (n_feat != 4 || nz != 1 || max_points > 32 || (long long)batch * nx * ny > 0x7ffffff0ll) return HVPR_ERR_UNSUPPORTED;
    const long long ncell = (long long)nx * ny * nz;
    if (ws_max_batch < batch || ws_max_points < n_points || ws_max_points < 1) return HVPR_ERR_WORKSPACE;
    if (workspace_bytes < hvpr_vox_ws_bytes(ws_max_batch, ws_max_points, ncell)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long long cells = (long long)batch * ncell;
    if (n_points == 0) {   // no points: empty canvases, zero pillars
        if (hipMemsetAsync(voxel_offsets, 0, sizeof(int) * (batch + 1), s) != hipSuccess) return HVPR_ERR_LAUNCH;
        if (hipMemsetAsync(spatial, 0, (size_t)cells * 128 * 4, s) != hipSuccess) return HVPR_ERR_LAUNCH;
        if (hipMemsetAsync(spatial_scale, 0, (size_t)cells * 32 * 4, s) != hipSuccess) return HVPR_ERR_LAUNCH;
        if (canvas_state && hipMemsetAsync(canvas_state, 0, (size_t)cells, s) != hipSuccess) return HVPR_ERR_LAUNCH;
        return HVPR_OK;
    }
    const VoxWs w = hvpr_vox_carve(workspace, ws_max_batch, ws_max_points, ncell);
    const VoxelizeArgs a{points, n_points, point_stride, xyz_col, n_feat, frame_offsets, batch, lo_x, lo_y, lo_z, vs_x, vs_y, vs_z,
                         nx, ny, nz, max_points, max_voxels, cap_mode};
    // (sizes of the pillar VFE's weights: csrc/vfe.hip — 10 -> 16 -> 64 per point, 5 -> 16 -> 32 for the scale stream)
    const WarmSmall small{{w0, b0, w1, b1, ws0, bs0, ws1, bs1}, {16 * 10, 16, 64 * 32, 64, 16 * 5, 16, 32 * 16, 32}};
    int st = hvpr_i_voxel_index(a, w, voxel_offsets, true, s, w1, b0, bank_packed, bank_packed ? hvpr_memory_bank_packed_floats(n_items) * 4 : 0,
                                bank, (size_t)n_items * 64 * 4, &small, index_mode);
    if (st != HVPR_OK) return st;
    const VfeWeights v{vs_x, vs_y, vs_z, off_x, off_y, off_z, w0, b0, w1, b1, ws0, bs0, ws1, bs1};
    st = hvpr_i_vfe_gather(a, w, voxel_offsets, capacity, v, voxels, coords, num_points, pillar_features, pillar_scale_features,
                           pillar_mask, spatial, 128, spatial_scale, canvas_state, s);
    if (st != HVPR_OK) return st;
    return hvpr_i_readout(pillar_features, capacity, voxel_offsets + batch, bank, bank_packed, n_items, k, memory_features, nullptr, coords,
                          batch, nx, ny, nullptr, spatial, 128, 64, s);
}
