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
    int4 *arena_rec;   // [N]  fused encode path: {point index, voxel rank or -1 (voxel dropped by the cap), voxel point count, cell}
                       //      per arena position — a pillar wave reads a window of the arena and needs nothing else
    int *arena_total;  // [1]  number of arena positions in use (= in-range points), written by K2
    float *vfe_aux;    // [64] fused encode path: the pillar VFE's padded-slot column (hvpr_vfe_padded_slot below), computed by an
                       //      extra workgroup of K3 so that the pillar kernel's prologue does not have to
    int *frame_base;   // [B+1] rank of the first voxel of each frame (uncapped)
    unsigned long long *tile_state;   // [tiles]
    int *ticket;       // [1]
    unsigned long long *cell_pack;    // [B*ncell] one-launch index kernel only: {rank, point count, arena offset} of an occupied cell, 20 bits each
    int *sync;         // [8 + 64] one-launch index kernel: the 8-byte XCD census of barrier 1, the exit counter, barrier 2's flag per owner (idle: 0; the kernel returns them to 0)
};

static inline VoxWs hvpr_vox_carve(void *ws, int batch, int n, long long ncell, size_t *bytes = nullptr) {
    hvpr_carver c(ws);
    VoxWs w;
    w.cell_first = c.take<int>((size_t)batch * ncell);
    w.cell_count = c.take<int>((size_t)batch * ncell);
    w.cell_vid = c.take<int>((size_t)batch * ncell);
    w.pt_cell = c.take<int>(n);
    w.vox_rec = c.take<int4>(n);
    w.arena = c.take<int>(n);
    w.arena_pt = c.take<float4>(n);
    w.arena_rec = c.take<int4>(n);
    w.arena_total = c.take<int>(1);
    w.vfe_aux = c.take<float>(128);   // [64 .. 127]: scratch word of the L2 warmers
    w.frame_base = c.take<int>(batch + 1);
    w.tile_state = c.take<unsigned long long>(hvpr_cdiv(n > 0 ? n : 1, kScanTile));
    w.ticket = c.take<int>(1);
    w.cell_pack = c.take<unsigned long long>((size_t)batch * ncell);
    w.sync = c.take<int>(8 + 64);
    if (bytes) *bytes = c.off;        // everything that was carved: the size is never a guess about the last field
    return w;
}

static inline size_t hvpr_vox_ws_bytes(int batch, int n, long long ncell) {
    size_t bytes = 0;
    hvpr_vox_carve(nullptr, batch, n, ncell, &bytes);
    return bytes;
}

// The dense BEV canvases of the fused encode path are cleared by EXTRA workgroups of the pillar-VFE launch (47 MB at
// hvpr_car, next to the pillar waves' dependent loads; measured: hosting part of it on the ~22 CUs the memory read-out
// leaves idle slows the read-out by more than it saves — a CU streams only ~45 GB/s of stores).  No race with the cells
// the pillar waves and the read-out write: a cell whose voxel is emitted is skipped — the voxelizer's
// cell maps still say which cells are occupied (K3 leaves cell_first alone on this path) and this pass returns them to
// idle while it is there.
struct ClearJob {
    int *cell_first;              // [B * ny * nx], kIdle = empty; reset here
    int *cell_count;              // [B * ny * nx]: returned to its idle 0 here (K3 does not when K1 handed out the slots)
    const int *cell_vid;          // rank of an occupied cell's voxel (uncapped order)
    const int *frame_base;        // [B + 1]
    const int *voxel_offsets;     // [B + 1]
    int batch, nx, ny, max_voxels, capacity;
    float *spatial;               // [B, ny, nx, 128]
    float *spatial_scale;         // [B, ny, nx, 32]
    long long cell_lo, cell_hi;   // the cells this launch clears
    unsigned char *state;         // null, or [B * ny * nx]: 1 = the cell still holds a pillar of the PREVIOUS call on these
                                  // canvases (which are zero wherever it says 0): only those cells are cleared — persistent
                                  // canvases get ~2.4 MB of zeros per frame instead of 47 MB; updated to this call's occupancy
};

__device__ __forceinline__ void hvpr_canvas_clear(const ClearJob &c, int blk, int nblk) {
    constexpr int V = 32, VS = 8;   // float4 per cell of the main / scale canvas; a wave clears 64 cells (40 KB) per step
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blk * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)nblk * blockDim.x) >> 6;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long long c0 = c.cell_lo + wave * 64; c0 < c.cell_hi; c0 += n_waves * 64) {
        const long long cell = c0 + lane;
        bool emitted = false, stale = c.state == nullptr;     // without a state array every cell counts as stale
        if (cell < c.cell_hi) {
            if (c.state) stale = c.state[cell] != 0;
            if (c.cell_first[cell] != kIdle) {
                c.cell_first[cell] = kIdle;
                c.cell_count[cell] = 0;
                const int b = (int)(cell / ((long long)c.nx * c.ny));
                const int local = c.cell_vid[cell] - c.frame_base[b];
                emitted = local < c.max_voxels && c.voxel_offsets[b] + local < c.capacity;
            }
            if (c.state && stale != emitted) c.state[cell] = emitted ? 1 : 0;
        }
        const unsigned long long skip = __ballot(emitted || !stale);   // bit i: cell c0 + i belongs to a pillar or is zero already
        if (c.state && skip == ~0ull) continue;
        // streaming stores: the zeros must not push the arrays the neighbouring waves are reading out of the L2
        f32x4 *const main = reinterpret_cast<f32x4 *>(c.spatial) + c0 * V;
        f32x4 *const side = reinterpret_cast<f32x4 *>(c.spatial_scale) + c0 * VS;
        const int cells = (int)min(64ll, c.cell_hi - c0);
#pragma unroll 8
        for (int i = lane; i < cells * V; i += 64)
            if (!((skip >> (i / V)) & 1ull)) __builtin_nontemporal_store(zero, main + i);
#pragma unroll 8
        for (int i = lane; i < cells * VS; i += 64)
            if (!((skip >> (i / VS)) & 1ull)) __builtin_nontemporal_store(zero, side + i);
    }
}

// The pillar VFE's padded slot (pillar_vfe.py:205-208 masks the INPUT: a padded slot still yields ReLU(folded bias) in layer 0
// and goes through layer 1): its layer-1 column Z1 = W1a . ReLU(b0), computed ON THE MATRIX CORES with the operand layout of
// the pillar kernel (vfe.hip), so that it rounds exactly like a real column with zero input.  One wave; z1[64] by channel.
//   aw[mb][t]  = w1[(32 mb + lane % 32) * 32 + 8 (t / 4) + 4 (lane / 32) + t % 4]      (w1: 64 x 32 row-major, BatchNorm folded)
//   b0h[t]     = b0[8 (t / 4) + 4 (lane / 32) + t % 4]
typedef float hvpr_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void hvpr_vfe_padded_slot(const float (&aw)[2][8], const float (&b0h)[8], float *z1) {
    const int lane = threadIdx.x & 63, h = lane >> 5, slot = lane & 31;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        hvpr_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[mb][t], fmaxf(0.f + b0h[t], 0.f), acc, 0, 0, 0);
        if (slot == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *(float4 *)&z1[32 * mb + 8 * g + 4 * h] = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
    }
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
// vfe_w1 / vfe_b0 (fused path, optional): an extra workgroup of K3 leaves the VFE's padded-slot column in w.vfe_aux.
// warm / warm_bytes (fused path, optional): 256 more workgroups of K3 — 32 per XCD, a 32nd of the arrays each — read them once, so that the memory
// bank (768 KB with its packed copy) sits in every XCD's L2 when the read-out two launches later asks for it; inside a frame the
// convolution stage (3.9 GB through the L2s) has evicted it since the previous frame (read-out 14.6 us in the frame, 12.5 us warm).
// warm_small (optional): up to eight small arrays (the pillar VFE's weights: 12 KB in eight tensors) read by ONE of the warmers per XCD.
struct WarmSmall { const float *p[8]; int n[8]; };
HVPR_INTERNAL int hvpr_i_voxel_index(const VoxelizeArgs &a, const VoxWs &w, int32_t *voxel_offsets, bool for_encode,
                                     hipStream_t s, const float *vfe_w1 = nullptr, const float *vfe_b0 = nullptr,
                                     const void *warm0 = nullptr, size_t warm0_bytes = 0, const void *warm1 = nullptr, size_t warm1_bytes = 0,
                                     const WarmSmall *warm_small = nullptr, int index_mode = 0);
// K4 fused into the pillar VFE: selects each voxel's points straight from the arena, writes voxels (optional) / coords /
// num_points, the pillar and scale features and the pillar + scale cells of the NHWC canvases; extra workgroups of the same
// launch clear every canvas cell that belongs to no pillar and return cell_first to idle (pair with for_encode above).
HVPR_INTERNAL int hvpr_i_vfe_gather(const VoxelizeArgs &a, const VoxWs &w, const int32_t *voxel_offsets, int capacity,
                                    const VfeWeights &v, float *voxels, int32_t *coords, int32_t *num_points,
                                    float *pillar_features, float *scale_features, float *pillar_mask, float *spatial,
                                    int spatial_channels, float *spatial_scale, unsigned char *canvas_state, hipStream_t s);
// memory read-out; optional cell map (gather-form scatter) or direct write of the memory cells of a canvas
HVPR_INTERNAL int hvpr_i_readout(const float *f, int M, const int32_t *m_device, const float *bank, const float *bank_packed, int n_items, int k,
                                 float *out, int32_t *topk_idx, const int32_t *coords, int batch, int nx, int ny,
                                 int *cell_map, float *canvas, int canvas_channels, int canvas_offset, hipStream_t s);
