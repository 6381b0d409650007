// Library-internal declarations shared between translation units (NOT part of the C-ABI: nothing here is exported).
//   * the voxelizer's workspace layout, read by the fused gather + VFE kernel (vfe.hip)
//   * the launchers the fused entry point hvpr_encode_fwd_f32 strings together
#pragma once
#include "common.h"

#define HVPR_INTERNAL __attribute__((visibility("hidden")))

constexpr int kScanThreads = 256;
constexpr int kScanItems = 2;
constexpr int kScanTile = kScanThreads * kScanItems;
constexpr int kIdle = 0x7fffffff;

struct VoxWs {
    int *cell_first;   // [B*ncell]  idle: kIdle
    int *cell_count;   // [B*ncell]  idle: 0
    int *cell_vid;     // [B*ncell]  scratch
    int *pt_cell;      // [N]
    int4 *vox_rec;     // [N] by global rank: {cell, point count, arena offset, first point index} — one 16-byte load per voxel
    int *arena;        // [N]  point indices, grouped by voxel (unordered inside a voxel)
    float4 *arena_pt;  // [N]  the points themselves next to their indices (fused encode path only: one load level less)
    int *frame_base;   // [B+1] rank of the first voxel of each frame (uncapped)
    unsigned long long *tile_state;   // [tiles]
    int *ticket;       // [1]
};

static inline VoxWs hvpr_vox_carve(void *ws, int batch, int n, long long ncell) {
    hvpr_carver c(ws);
    VoxWs w;
    w.cell_first = c.take<int>((size_t)batch * ncell);
    w.cell_count = c.take<int>((size_t)batch * ncell);
    w.cell_vid = c.take<int>((size_t)batch * ncell);
    w.pt_cell = c.take<int>(n);
    w.vox_rec = c.take<int4>(n);
    w.arena = c.take<int>(n);
    w.arena_pt = c.take<float4>(n);
    w.frame_base = c.take<int>(batch + 1);
    w.tile_state = c.take<unsigned long long>(hvpr_cdiv(n > 0 ? n : 1, kScanTile));
    w.ticket = c.take<int>(1);
    return w;
}

static inline size_t hvpr_vox_ws_bytes(int batch, int n, long long ncell) {
    VoxWs w = hvpr_vox_carve(nullptr, batch, n, ncell);
    return (size_t)((char *)(w.ticket) - (char *)nullptr) + 256;
}

struct VoxelizeArgs {
    const float *points;
    int n_points, point_stride, xyz_col, n_feat;
    const int32_t *frame_offsets;
    int batch;
    float lo_x, lo_y, lo_z, vs_x, vs_y, vs_z;
    int nx, ny, nz, max_points, max_voxels, cap_mode;
};

struct VfeWeights {
    float vs_x, vs_y, vs_z, off_x, off_y, off_z;
    const float *w0, *b0, *w1, *b1, *ws0, *bs0, *ws1, *bs1;
};

// K1-K3 of the voxelizer (cell keys, rank scan, arena fill) + voxel_offsets.  for_encode: K3 also copies each point (4
// floats) next to its index in the arena, and does not return the cell_first map to idle — the caller's next kernel reads
// the occupancy from it and resets it (hvpr_i_vfe_gather does both).
HVPR_INTERNAL int hvpr_i_voxel_index(const VoxelizeArgs &a, const VoxWs &w, int32_t *voxel_offsets, bool for_encode,
                                     hipStream_t s);
// K4 fused into the pillar VFE: selects each voxel's points straight from the arena, writes voxels (optional) / coords /
// num_points, the pillar and scale features and the pillar + scale cells of the NHWC canvases; extra workgroups of the same
// launch clear every canvas cell that belongs to no pillar and return cell_first to idle (pair with keep_cell_first above).
HVPR_INTERNAL int hvpr_i_vfe_gather(const VoxelizeArgs &a, const VoxWs &w, const int32_t *voxel_offsets, int capacity,
                                    const VfeWeights &v, float *voxels, int32_t *coords, int32_t *num_points,
                                    float *pillar_features, float *scale_features, float *pillar_mask, float *spatial,
                                    int spatial_channels, float *spatial_scale, hipStream_t s);
// memory read-out; optional cell map (gather-form scatter) or direct write of the memory cells of a pre-cleared canvas
HVPR_INTERNAL int hvpr_i_readout(const float *f, int M, const int32_t *m_device, const float *bank, int n_items, int k,
                                 float *out, int32_t *topk_idx, const int32_t *coords, int batch, int nx, int ny,
                                 int *cell_map, float *canvas, int canvas_channels, int canvas_offset, hipStream_t s);
