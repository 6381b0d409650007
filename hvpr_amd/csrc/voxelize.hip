// a1 — GPU voxelizer with the sequential first-touch semantics of spconv's VoxelGenerator
// (call site in the reference: pcdet/datasets/processor/data_processor.py:43-75; in-tree twin of the
// loop: tools/vis.py:9-60).  The sequential hash is re-expressed as order statistics so that it is
// bit-identical on a parallel machine:
//   voxel id of a cell   = rank of the cell among cells ordered by their smallest point index
//   slot of a point      = rank of its index among the points of the same cell (first max_points kept)
//   max_voxels (V2)      = cells of rank >= max_voxels are dropped, everything else unchanged
//   max_voxels (V1)      = additionally every point with index >= first index of the rank==max_voxels cell
//
// Four launches, no memsets: the two persistent cell maps are returned to their idle state by K3.
//   K1 cell keys (fp32 sub/div/floor, IEEE), atomicMin(first index) + atomicAdd(count) per cell
//   K2 single-pass decoupled-look-back scan over points of (is_first, count) -> voxel rank, arena offset
//   K3 each point appends its index to its voxel's arena segment (unordered)
//   K4 one wave per voxel: bitonic selection of the max_points smallest indices (ascending), gather.
#include "common.h"

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;
constexpr int kIdle = 0x7fffffff;

struct VoxWs {
    int *cell_first;   // [B*ncell]  idle: kIdle
    int *cell_count;   // [B*ncell]  idle: 0
    int *cell_vid;     // [B*ncell]  scratch
    int *pt_cell;      // [N]
    int *vox_cell;     // [N] by global rank
    int *vox_count;    // [N]
    int *vox_arena;    // [N]
    int *vox_first;    // [N]
    int *arena;        // [N]
    int *frame_base;   // [B+1] rank of the first voxel of each frame (uncapped)
    unsigned long long *tile_state;   // [tiles]
    int *ticket;       // [1]
};

__host__ VoxWs carve(void *ws, int batch, int n, long long ncell) {
    hvpr_carver c(ws);
    VoxWs w;
    w.cell_first = c.take<int>((size_t)batch * ncell);
    w.cell_count = c.take<int>((size_t)batch * ncell);
    w.cell_vid = c.take<int>((size_t)batch * ncell);
    w.pt_cell = c.take<int>(n);
    w.vox_cell = c.take<int>(n);
    w.vox_count = c.take<int>(n);
    w.vox_arena = c.take<int>(n);
    w.vox_first = c.take<int>(n);
    w.arena = c.take<int>(n);
    w.frame_base = c.take<int>(batch + 1);
    w.tile_state = c.take<unsigned long long>(hvpr_cdiv(n > 0 ? n : 1, kScanTile));
    w.ticket = c.take<int>(1);
    return w;
}

__host__ size_t ws_bytes(int batch, int n, long long ncell) {
    VoxWs w = carve(nullptr, batch, n, ncell);
    return (size_t)((char *)(w.ticket) - (char *)nullptr) + 256;
}

__device__ __forceinline__ int frame_of(const int *__restrict__ off, int batch, int i) {
    // largest b in [0,batch) with off[b] <= i
    int lo = 0, hi = batch;   // invariant: off[lo] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void k_reset(int *cell_first, int *cell_count, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        cell_first[i] = kIdle;
        cell_count[i] = 0;
    }
}

__global__ void __launch_bounds__(256) k1_keys(const float *__restrict__ pts, int n, int stride, int xyz_col,
                                               const int *__restrict__ foff, int batch, float lox, float loy, float loz,
                                               float vsx, float vsy, float vsz, int nx, int ny, int nz, VoxWs w, int tiles) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tiles) w.tile_state[i] = 0ull;
    if (i == 0) *w.ticket = 0;
    if (i >= n) return;
    const float *p = pts + (size_t)i * stride + xyz_col;
    // exact IEEE fp32: one subtract, one divide, floor (no fast-math; hipcc's default division is correctly rounded)
    const float cx = floorf(__fdiv_rn(__fsub_rn(p[0], lox), vsx));
    const float cy = floorf(__fdiv_rn(__fsub_rn(p[1], loy), vsy));
    const float cz = floorf(__fdiv_rn(__fsub_rn(p[2], loz), vsz));
    int g = -1;
    if (cx >= 0.f && cx < (float)nx && cy >= 0.f && cy < (float)ny && cz >= 0.f && cz < (float)nz) {
        const int b = frame_of(foff, batch, i);
        g = ((b * nz + (int)cz) * ny + (int)cy) * nx + (int)cx;
        atomicMin(&w.cell_first[g], i);
        atomicAdd(&w.cell_count[g], 1);
    }
    w.pt_cell[i] = g;
}

// status (2 bits) | first-touch sum (31 bits) | count sum (31 bits)
__device__ __forceinline__ unsigned long long pack(unsigned st, unsigned a, unsigned b) {
    return ((unsigned long long)st << 62) | ((unsigned long long)a << 31) | (unsigned long long)b;
}

__global__ void __launch_bounds__(kScanThreads) k2_scan(int n, const int *__restrict__ foff, int batch, VoxWs w) {
    __shared__ int s_tile;
    __shared__ unsigned s_wave_a[kScanThreads / 64], s_wave_b[kScanThreads / 64];
    __shared__ unsigned s_excl_a, s_excl_b;
    if (threadIdx.x == 0) s_tile = atomicAdd(w.ticket, 1);
    __syncthreads();
    const int tile = s_tile;
    const int base = tile * kScanTile + threadIdx.x * kScanItems;

    int gcell[kScanItems];
    unsigned fl[kScanItems], ct[kScanItems];
    unsigned ta = 0, tb = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        gcell[k] = -1; fl[k] = 0; ct[k] = 0;
        if (i < n) {
            const int g = w.pt_cell[i];
            if (g >= 0 && w.cell_first[g] == i) { gcell[k] = g; fl[k] = 1; ct[k] = (unsigned)w.cell_count[g]; }
        }
        ta += fl[k]; tb += ct[k];
    }
    // block exclusive scan of (ta, tb)
    unsigned ia = ta, ib = tb;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned ua = __shfl_up(ia, o, 64), ub = __shfl_up(ib, o, 64);
        if (lane >= o) { ia += ua; ib += ub; }
    }
    if (lane == 63) { s_wave_a[wid] = ia; s_wave_b[wid] = ib; }
    __syncthreads();
    unsigned wa = 0, wb = 0, tot_a = 0, tot_b = 0;
#pragma unroll
    for (int k = 0; k < kScanThreads / 64; ++k) {
        if (k < wid) { wa += s_wave_a[k]; wb += s_wave_b[k]; }
        tot_a += s_wave_a[k]; tot_b += s_wave_b[k];
    }
    // decoupled look-back (one lane; the tile word is both data and flag — single 8-byte agent-scope store)
    if (threadIdx.x == 0) {
        unsigned ea = 0, eb = 0;
        if (tile == 0) {
            __hip_atomic_store(&w.tile_state[0], pack(2u, tot_a, tot_b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(&w.tile_state[tile], pack(1u, tot_a, tot_b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int t = tile - 1; t >= 0; --t) {
                unsigned long long s;
                do {
                    s = __hip_atomic_load(&w.tile_state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((s >> 62) == 0) __builtin_amdgcn_s_sleep(1);
                } while ((s >> 62) == 0);
                ea += (unsigned)((s >> 31) & 0x7fffffffu);
                eb += (unsigned)(s & 0x7fffffffu);
                if ((s >> 62) == 2) break;
            }
            __hip_atomic_store(&w.tile_state[tile], pack(2u, ea + tot_a, eb + tot_b), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        s_excl_a = ea; s_excl_b = eb;
    }
    __syncthreads();
    unsigned ra = s_excl_a + wa + (ia - ta);   // exclusive prefix of this thread's first item
    unsigned rb = s_excl_b + wb + (ib - tb);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int i = base + k;
        if (i < n) {
            // frame bases: the thread owning the first point of a frame knows that frame's first voxel rank
            const int b = frame_of(foff, batch, i);
            if (foff[b] == i)
                for (int bb = b; bb >= 0 && foff[bb] == i; --bb) w.frame_base[bb] = (int)ra;
            if (fl[k]) {
                w.cell_vid[gcell[k]] = (int)ra;
                w.vox_cell[ra] = gcell[k];
                w.vox_count[ra] = (int)ct[k];
                w.vox_arena[ra] = (int)rb;
                w.vox_first[ra] = i;
            }
            if (i == n - 1) {
                const int total = (int)(ra + fl[k]);
                for (int bb = batch; bb >= 0 && foff[bb] == n; --bb) w.frame_base[bb] = total;
            }
        }
        ra += fl[k]; rb += ct[k];
    }
}

__global__ void __launch_bounds__(256) k3_fill(int n, const int *__restrict__ foff, int batch, int max_voxels, VoxWs w,
                                               int *__restrict__ voxel_offsets) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        int acc = 0;
        for (int b = 0; b < batch; ++b) {
            voxel_offsets[b] = acc;
            const int c = w.frame_base[b + 1] - w.frame_base[b];
            acc += c < max_voxels ? c : max_voxels;
        }
        voxel_offsets[batch] = acc;
    }
    if (i >= n) return;
    const int g = w.pt_cell[i];
    if (g < 0) return;
    const int r = w.cell_vid[g];
    const int b = frame_of(foff, batch, i);
    const int local = r - w.frame_base[b];
    const int slot = atomicSub(&w.cell_count[g], 1) - 1;   // returns the map to its idle 0
    w.cell_first[g] = kIdle;                               // idle again (benign same-value race)
    if (local < max_voxels) w.arena[w.vox_arena[r] + slot] = i;
}

__device__ __forceinline__ int bitonic64_asc(int v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = __shfl_xor(v, j, 64);
            const bool up = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            v = (lower == up) ? min(v, o) : max(v, o);
        }
    }
    return v;
}

__global__ void __launch_bounds__(256) k4_gather(const float *__restrict__ pts, int stride, int xyz_col, int n_feat,
                                                 int batch, int nx, int ny, int nz, int max_points, int max_voxels,
                                                 int cap_mode, VoxWs w, const int *__restrict__ voxel_offsets,
                                                 float *__restrict__ voxels, int *__restrict__ coords,
                                                 int *__restrict__ num_points, int capacity) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const int total = w.frame_base[batch];
    for (int r = wave; r < total; r += n_waves) {
        // frame of this rank: largest b with frame_base[b] <= r and a non-empty frame
        int b = 0;
        for (int bb = 1; bb < batch; ++bb) if (w.frame_base[bb] <= r) b = bb;
        const int local = r - w.frame_base[b];
        if (local >= max_voxels) continue;
        const int o = voxel_offsets[b] + local;
        if (o >= capacity) continue;
        int cutoff = kIdle;
        if (cap_mode == 1) {
            const int rc = w.frame_base[b] + max_voxels;
            if (rc < w.frame_base[b + 1]) cutoff = w.vox_first[rc];
        }
        const int cnt = w.vox_count[r];
        const int a0 = w.vox_arena[r];
        // the max_points smallest indices, ascending, in lanes [0, max_points)
        int v = lane < cnt ? w.arena[a0 + lane] : kIdle;
        v = bitonic64_asc(v, lane);
        const int chunk = 64 - max_points;
        for (int done = 64; done < cnt; done += chunk) {
            if (lane >= max_points) {
                const int j = done + (lane - max_points);
                v = j < cnt ? w.arena[a0 + j] : kIdle;
            }
            v = bitonic64_asc(v, lane);
        }
        const bool live = lane < max_points && v < cutoff;   // v == kIdle is never < cutoff
        const int num = __popcll(__ballot(live));
        if (lane < max_points) {
            float *dst = voxels + ((size_t)o * max_points + lane) * n_feat;
            if (live) {
                const float *src = pts + (size_t)v * stride + xyz_col;
                for (int f = 0; f < n_feat; ++f) dst[f] = src[f];
            } else {
                for (int f = 0; f < n_feat; ++f) dst[f] = 0.f;
            }
        }
        if (lane == 0) {
            const int g = w.vox_cell[r];
            const int cx = g % nx, cy = (g / nx) % ny, cz = (g / (nx * ny)) % nz;
            reinterpret_cast<int4 *>(coords)[o] = make_int4(b, cz, cy, cx);
            num_points[o] = num;
        }
    }
}

}  // namespace

extern "C" size_t hvpr_voxelize_workspace_bytes(int batch, int n_points, int nx, int ny, int nz) {
    if (batch < 1 || n_points < 0 || nx < 1 || ny < 1 || nz < 1) return 0;
    return ws_bytes(batch, n_points > 0 ? n_points : 1, (long long)nx * ny * nz);
}

extern "C" int hvpr_voxelize_workspace_reset(void *workspace, size_t workspace_bytes, int batch, int n_points, int nx,
                                             int ny, int nz, hvpr_stream_t stream) {
    if (!workspace || batch < 1 || n_points < 0 || nx < 1 || ny < 1 || nz < 1) return HVPR_ERR_INVALID_ARG;
    const long long ncell = (long long)nx * ny * nz;
    if (workspace_bytes < ws_bytes(batch, n_points > 0 ? n_points : 1, ncell)) return HVPR_ERR_WORKSPACE;
    VoxWs w = carve(workspace, batch, n_points > 0 ? n_points : 1, ncell);
    hipLaunchKernelGGL(k_reset, dim3(1024), dim3(256), 0, (hipStream_t)stream, w.cell_first, w.cell_count, batch * ncell);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_voxelize_f32(const float *points, int n_points, int point_stride, int xyz_col, int n_feat,
                                 const int32_t *frame_offsets, int batch, float lo_x, float lo_y, float lo_z, float vs_x,
                                 float vs_y, float vs_z, int nx, int ny, int nz, int max_points, int max_voxels,
                                 int cap_mode, float *voxels, int32_t *coords, int32_t *num_points,
                                 int32_t *voxel_offsets, int capacity, void *workspace, size_t workspace_bytes,
                                 int ws_max_batch, int ws_max_points, hvpr_stream_t stream) {
    if (!points || !frame_offsets || !voxels || !coords || !num_points || !voxel_offsets || !workspace)
        return HVPR_ERR_INVALID_ARG;
    if (batch < 1 || n_points < 0 || n_feat < 3 || xyz_col < 0 || point_stride < xyz_col + n_feat || nx < 1 || ny < 1 ||
        nz < 1 || max_points < 1 || max_voxels < 1 || capacity < 0 || (cap_mode != 0 && cap_mode != 1))
        return HVPR_ERR_INVALID_ARG;
    if (max_points > 63 || (long long)batch * nx * ny * nz > 0x7ffffff0ll) return HVPR_ERR_UNSUPPORTED;
    const long long ncell = (long long)nx * ny * nz;
    // the workspace is carved with the dimensions it was sized and reset for, not with this call's
    if (ws_max_batch < batch || ws_max_points < n_points || ws_max_points < 1) return HVPR_ERR_WORKSPACE;
    if (workspace_bytes < ws_bytes(ws_max_batch, ws_max_points, ncell)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (n_points == 0) {
        if (hipMemsetAsync(voxel_offsets, 0, sizeof(int) * (batch + 1), s) != hipSuccess) return HVPR_ERR_LAUNCH;
        return HVPR_OK;
    }
    VoxWs w = carve(workspace, ws_max_batch, ws_max_points, ncell);
    const int tiles = hvpr_cdiv(n_points, kScanTile);
    const int pblocks = hvpr_cdiv(n_points, 256);
    hipLaunchKernelGGL(k1_keys, dim3(pblocks), dim3(256), 0, s, points, n_points, point_stride, xyz_col, frame_offsets,
                       batch, lo_x, lo_y, lo_z, vs_x, vs_y, vs_z, nx, ny, nz, w, tiles);
    hipLaunchKernelGGL(k2_scan, dim3(tiles), dim3(kScanThreads), 0, s, n_points, frame_offsets, batch, w);
    hipLaunchKernelGGL(k3_fill, dim3(pblocks), dim3(256), 0, s, n_points, frame_offsets, batch, max_voxels, w,
                       voxel_offsets);
    int gblocks = hvpr_cdiv(n_points, 4);
    if (gblocks > 2048) gblocks = 2048;
    hipLaunchKernelGGL(k4_gather, dim3(gblocks), dim3(256), 0, s, points, point_stride, xyz_col, n_feat, batch, nx, ny,
                       nz, max_points, max_voxels, cap_mode, w, voxel_offsets, voxels, coords, num_points, capacity);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
