// f2 ("next" row of SURVEY.md §8f) — KITTI point pre-processing on the device, the steps in front of the voxelizer:
//   mask_points_by_range           pcdet/utils/common_utils.py:59-62 (x and y only, both ends inclusive), called from
//                                  DataProcessor.mask_points_and_boxes_outside_range, pcdet/datasets/processor/data_processor.py:20-30
//   FOV filter                     KittiDataset.get_fov_flag, pcdet/datasets/kitti/kitti_dataset.py:100-116 over
//                                  Calibration.lidar_to_rect / rect_to_img, pcdet/utils/calibration_kitti.py:65-84 (all float32)
//   near flag of sample_points     data_processor.py:89-90: ||xyz||_2 < 40
//   row compaction / row gather    points[mask], points[choice]
// Stable stream compaction in two launches without spin-waits: per-tile counts, then every tile sums the counts in front of it.
// fp32 arithmetic in a fixed order without FMA contraction (the reference's BLAS order for the K=4 products is
// unspecified: the defined order here is left to right).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kTile = kThreads * kItems;

struct FovArgs {
    float a[12];   // (V2C^T R0^T): rect_j = sum_i hom_i * a[i*3 + j], hom = [x,y,z,1]
    float p[12];   // P2^T:         img_j  = sum_i rhom_i * p[i*3 + j], rhom = [rect,1]
    float img_h, img_w;
    int enabled;
};

__device__ __forceinline__ float dot4(float x, float y, float z, const float *m, int j) {
    // ((x*m0 + y*m1) + z*m2) + 1*m3, no contraction (this file is compiled with -ffp-contract=off)
    return ((x * m[j] + y * m[3 + j]) + z * m[6 + j]) + m[9 + j];
}

__global__ void __launch_bounds__(kThreads) k_point_flags(const float *__restrict__ pts, int n, int stride, int mode, float x0,
                                                          float y0, float x1, float y1, float near_thresh, FovArgs fov,
                                                          unsigned char *__restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = pts + (size_t)i * stride;
    const float x = p[0], y = p[1], z = p[2];
    bool f;
    if (mode == 0) {
        f = x >= x0 && x <= x1 && y >= y0 && y <= y1;
        if (f && fov.enabled) {
            const float rx = dot4(x, y, z, fov.a, 0), ry = dot4(x, y, z, fov.a, 1), rz = dot4(x, y, z, fov.a, 2);
            const float hx = dot4(rx, ry, rz, fov.p, 0), hy = dot4(rx, ry, rz, fov.p, 1), hz = dot4(rx, ry, rz, fov.p, 2);
            const float u = __fdiv_rn(hx, rz), v = __fdiv_rn(hy, rz);       // divided by the rect depth (calibration_kitti.py:82)
            const float depth = hz - fov.p[11];                              // P2.T[3,2]
            f = u >= 0.f && u < fov.img_w && v >= 0.f && v < fov.img_h && depth >= 0.f;
        }
    } else {
        f = __fsqrt_rn((x * x + y * y) + z * z) < near_thresh;
    }
    flags[i] = f ? 1 : 0;
}

__global__ void __launch_bounds__(kThreads) k_tile_counts(const unsigned char *__restrict__ flags, int n, int *__restrict__ tile_count) {
    __shared__ int s_w[kThreads / 64];
    const int base = blockIdx.x * kTile + threadIdx.x * kItems;
    int c = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) c += (base + k < n && flags[base + k]) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void __launch_bounds__(kThreads) k_compact_rows(const float *__restrict__ src, int n, int row, const unsigned char *__restrict__ flags,
                                                           const int *__restrict__ tile_count, int tiles, float *__restrict__ dst,
                                                           int capacity, int *__restrict__ count) {
    __shared__ int s_w[kThreads / 64];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // exclusive prefix of the tile counts in front of this tile (a few hundred words at most)
    int part = 0;
    for (int t = threadIdx.x; t < blockIdx.x; t += kThreads) part += tile_count[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) s_w[wid] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (blockIdx.x == tiles - 1) *count = min(s_base + tile_count[blockIdx.x], capacity);
    }
    __syncthreads();
    const int base = blockIdx.x * kTile + threadIdx.x * kItems;
    int f[kItems], c = 0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) { f[k] = (base + k < n && flags[base + k]) ? 1 : 0; c += f[k]; }
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    __syncthreads();
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    int pos = s_base + inc - c;
    for (int w = 0; w < wid; ++w) pos += s_w[w];
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        if (f[k]) {
            if (pos < capacity)
                for (int j = 0; j < row; ++j) dst[(size_t)pos * row + j] = src[(size_t)(base + k) * row + j];
            ++pos;
        }
    }
}

__global__ void __launch_bounds__(256) k_gather_rows(const float *__restrict__ src, int n_src, int row, const int *__restrict__ idx, int m,
                                                     float *__restrict__ dst) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)m * row) return;
    const int r = (int)(t / row), j = (int)(t % row);
    const int s = idx[r];
    dst[t] = (s >= 0 && s < n_src) ? src[(size_t)s * row + j] : 0.f;
}

// frame_offsets of a collated point array (dataset.py:161-166: frames contiguous, ascending batch index in column 0):
// offsets[b] = first row of frame b, offsets[batch] = n.  One launch instead of searchsorted + arange + casts.
__global__ void __launch_bounds__(256) k_frame_offsets(const float *__restrict__ pts, int n, int stride, int batch,
                                                       int *__restrict__ offsets) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const int b_here = i < n ? min(max((int)pts[(size_t)i * stride], 0), batch) : batch;   // row n acts as "frame batch"
    const int b_prev = i > 0 ? min(max((int)pts[(size_t)(i - 1) * stride], 0), batch) : -1;
    for (int b = b_prev + 1; b <= b_here; ++b) offsets[b] = i;    // empty frames in between start here as well
}

}  // namespace

extern "C" int hvpr_point_flags_f32(const float *points, int n, int stride, int mode, const float *range_xy, float near_thresh,
                                    const float *fov_lidar_to_rect, const float *fov_rect_to_img, int img_h, int img_w,
                                    uint8_t *flags, hvpr_stream_t stream) {
    if (n < 0 || stride < 3 || (mode != 0 && mode != 1)) return HVPR_ERR_INVALID_ARG;
    if (n == 0) return HVPR_OK;
    if (!points || !flags || (mode == 0 && !range_xy)) return HVPR_ERR_INVALID_ARG;
    if ((fov_lidar_to_rect == nullptr) != (fov_rect_to_img == nullptr)) return HVPR_ERR_INVALID_ARG;
    FovArgs fov;
    fov.enabled = (mode == 0 && fov_lidar_to_rect) ? 1 : 0;
    for (int i = 0; i < 12; ++i) {
        fov.a[i] = fov.enabled ? fov_lidar_to_rect[i] : 0.f;
        fov.p[i] = fov.enabled ? fov_rect_to_img[i] : 0.f;
    }
    fov.img_h = (float)img_h; fov.img_w = (float)img_w;
    const float x0 = mode == 0 ? range_xy[0] : 0.f, y0 = mode == 0 ? range_xy[1] : 0.f;
    const float x1 = mode == 0 ? range_xy[2] : 0.f, y1 = mode == 0 ? range_xy[3] : 0.f;
    hipLaunchKernelGGL(k_point_flags, dim3(hvpr_cdiv(n, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, points, n, stride, mode, x0,
                       y0, x1, y1, near_thresh, fov, flags);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" size_t hvpr_compact_workspace_bytes(int n) { return n < 0 ? 0 : (size_t)(hvpr_cdiv(n > 0 ? n : 1, kTile) + 1) * sizeof(int) + 256; }

extern "C" int hvpr_compact_rows_f32(const float *src, int n, int row_floats, const uint8_t *flags, float *dst, int capacity,
                                     int32_t *count, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (n < 0 || row_floats < 1 || capacity < 0 || !count || !workspace) return HVPR_ERR_INVALID_ARG;
    if (workspace_bytes < hvpr_compact_workspace_bytes(n)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        hipLaunchKernelGGL(k_tile_counts, dim3(1), dim3(kThreads), 0, s, (const unsigned char *)nullptr, 0, (int *)workspace);
        hipLaunchKernelGGL(k_compact_rows, dim3(1), dim3(kThreads), 0, s, src, 0, row_floats, (const unsigned char *)nullptr,
                           (const int *)workspace, 1, dst, capacity, count);
        HVPR_CHECK_LAUNCH();
        return HVPR_OK;
    }
    if (!src || !flags || (!dst && capacity > 0)) return HVPR_ERR_INVALID_ARG;
    const int tiles = hvpr_cdiv(n, kTile);
    int *tile_count = (int *)workspace;
    hipLaunchKernelGGL(k_tile_counts, dim3(tiles), dim3(kThreads), 0, s, flags, n, tile_count);
    hipLaunchKernelGGL(k_compact_rows, dim3(tiles), dim3(kThreads), 0, s, src, n, row_floats, flags, tile_count, tiles, dst, capacity,
                       count);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_gather_rows_f32(const float *src, int n_src, int row_floats, const int32_t *idx, int m, float *dst,
                                    hvpr_stream_t stream) {
    if (n_src < 0 || row_floats < 1 || m < 0) return HVPR_ERR_INVALID_ARG;
    if (m == 0) return HVPR_OK;
    if (!src || !idx || !dst) return HVPR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_gather_rows, dim3(hvpr_cdiv((long long)m * row_floats, 256)), dim3(256), 0, (hipStream_t)stream, src, n_src,
                       row_floats, idx, m, dst);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_frame_offsets_f32(const float *points, int n_points, int point_stride, int batch, int32_t *frame_offsets,
                                      hvpr_stream_t stream) {
    if ((!points && n_points > 0) || !frame_offsets || n_points < 0 || point_stride < 1 || batch < 1) return HVPR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_frame_offsets, dim3(hvpr_cdiv(n_points + 1, 256)), dim3(256), 0, (hipStream_t)stream, points, n_points,
                       point_stride, batch, frame_offsets);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
