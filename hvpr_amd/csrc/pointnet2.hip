// a9 (training only) — index-producing ops of the PointNet++ point stream.  Replaces the reference's absent native module
// pcdet/ops/pointnet2/pointnet2_batch (sources named by setup.py:94-109; used through PointnetSAModuleMSG / PointnetFPModule
// at pcdet/models/backbones_3d/pointnet2_backbone.py:27-34,43-47,82,86-89): furthest point sampling, ball query, three-NN.
// They produce integer indices (no gradient); gathering / interpolation with gradients is done by the caller with
// differentiable gathers.  Distances are fp32 (dx*dx + dy*dy) + dz*dz without FMA contraction, the order the CPU oracle
// uses, so the indices are bit-exact.  Tie rule of this build: lowest index wins.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------- FPS
// One workgroup per sample; every thread keeps PPT points and their running min-distance in registers; each of the
// `npoint` dependent steps is: update with the last pick, thread-local arg-max, wave arg-max, 16-wave arg-max through LDS.
constexpr int FPS_THREADS = 1024;

__device__ __forceinline__ unsigned long long fps_key(float d, int idx) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(0xffffffffu - (unsigned)idx);   // d >= 0
}

template <int PPT>
__global__ void __launch_bounds__(FPS_THREADS) k_fps(const float *__restrict__ xyz, int N, int npoint, int *__restrict__ out) {
    __shared__ unsigned long long s_best[2][FPS_THREADS / 64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const float *p = xyz + (size_t)b * N * 3;
    float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = t + k * FPS_THREADS;
        const bool ok = i < N;
        px[k] = ok ? p[i * 3 + 0] : 0.f;
        py[k] = ok ? p[i * 3 + 1] : 0.f;
        pz[k] = ok ? p[i * 3 + 2] : 0.f;
        md[k] = ok ? 1e10f : -1.f;          // padding can never win
    }
    int last = 0;
    if (t == 0) out[(size_t)b * npoint] = 0;      // first pick is index 0
    for (int j = 1; j < npoint; ++j) {
        const float lx = p[last * 3 + 0], ly = p[last * 3 + 1], lz = p[last * 3 + 2];
        unsigned long long best = 0ull;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const float dx = px[k] - lx, dy = py[k] - ly, dz = pz[k] - lz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (md[k] >= 0.f) {
                md[k] = fminf(md[k], d);
                const unsigned long long key = fps_key(md[k], t + k * FPS_THREADS);
                best = key > best ? key : best;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffull), o, 64);
            const unsigned hi = __shfl_xor((unsigned)(best >> 32), o, 64);
            const unsigned long long ob = ((unsigned long long)hi << 32) | lo;
            best = ob > best ? ob : best;
        }
        const int buf = j & 1;
        if (lane == 0) s_best[buf][wid] = best;
        __syncthreads();
        unsigned long long g = s_best[buf][0];
#pragma unroll
        for (int w = 1; w < FPS_THREADS / 64; ++w) g = s_best[buf][w] > g ? s_best[buf][w] : g;
        last = (int)(0xffffffffu - (unsigned)(g & 0xffffffffull));
        if (t == 0) out[(size_t)b * npoint + j] = last;
    }
}

// ---------------------------------------------------------------------------------------------------- ball query
// idx[b,m,:] = the first nsample points (index order) with d2 < radius^2; the first hit pre-fills every slot; 0 if none.
// A workgroup takes 256 queries of one sample and walks the sample's points in tiles of 512 staged in LDS (structure of arrays:
// every thread reads the same point at the same time = a broadcast read); it stops when all of its queries are full.
constexpr int BQ_TILE = 512;
__global__ void __launch_bounds__(256) k_ball_query(const float *__restrict__ xyz, const float *__restrict__ new_xyz, int B, int N,
                                                    int M, float radius, int nsample, int *__restrict__ idx) {
    __shared__ float s_x[BQ_TILE], s_y[BQ_TILE], s_z[BQ_TILE];
    const int wg_per_b = (M + 255) / 256;
    const int b = blockIdx.x / wg_per_b, m = (blockIdx.x % wg_per_b) * 256 + threadIdx.x;
    const bool live = m < M;
    const long long t = (long long)b * M + (live ? m : 0);
    const float *q = new_xyz + (size_t)t * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float r2 = radius * radius;
    const float *p = xyz + (size_t)b * N * 3;
    int *o = idx + (size_t)t * nsample;
    int cnt = live ? 0 : nsample;
    for (int k0 = 0; k0 < N; k0 += BQ_TILE) {
        const int nk = min(BQ_TILE, N - k0);
        __syncthreads();                      // the previous tile has been read
        for (int i = threadIdx.x; i < nk; i += 256) {
            s_x[i] = p[(size_t)(k0 + i) * 3 + 0]; s_y[i] = p[(size_t)(k0 + i) * 3 + 1]; s_z[i] = p[(size_t)(k0 + i) * 3 + 2];
        }
        __syncthreads();
        for (int i = 0; i < nk && cnt < nsample; ++i) {
            const float dx = qx - s_x[i], dy = qy - s_y[i], dz = qz - s_z[i];
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < r2) {
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l) o[l] = k0 + i;
                o[cnt++] = k0 + i;
            }
        }
        if (__syncthreads_and(cnt >= nsample)) break;
    }
    if (live && cnt == 0)
        for (int l = 0; l < nsample; ++l) o[l] = 0;
}

// ---------------------------------------------------------------------------------------------------- three nearest
__global__ void __launch_bounds__(256) k_three_nn(const float *__restrict__ unknown, const float *__restrict__ known, int B, int n,
                                                  int m, float *__restrict__ dist, int *__restrict__ idx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * n) return;
    const int b = (int)(t / n);
    const float *u = unknown + (size_t)t * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float *kp = known + (size_t)b * m * 3;
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < m; ++k) {
        const float dx = ux - kp[k * 3 + 0], dy = uy - kp[k * 3 + 1], dz = uz - kp[k * 3 + 2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
        else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
        else if (d < b3) { b3 = d; i3 = k; }
    }
    dist[t * 3 + 0] = sqrtf(b1); dist[t * 3 + 1] = sqrtf(b2); dist[t * 3 + 2] = sqrtf(b3);
    idx[t * 3 + 0] = i1; idx[t * 3 + 1] = i2; idx[t * 3 + 2] = i3;
}

}  // namespace

extern "C" int hvpr_furthest_point_sample_f32(const float *xyz, int B, int N, int npoint, int32_t *idx, hvpr_stream_t stream) {
    if (!xyz || !idx || B < 1 || N < 1 || npoint < 1 || npoint > N) return HVPR_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int ppt = hvpr_cdiv(N, FPS_THREADS);
    if (ppt <= 1) hipLaunchKernelGGL(k_fps<1>, dim3(B), dim3(FPS_THREADS), 0, s, xyz, N, npoint, idx);
    else if (ppt <= 4) hipLaunchKernelGGL(k_fps<4>, dim3(B), dim3(FPS_THREADS), 0, s, xyz, N, npoint, idx);
    else if (ppt <= 16) hipLaunchKernelGGL(k_fps<16>, dim3(B), dim3(FPS_THREADS), 0, s, xyz, N, npoint, idx);
    else if (ppt <= 32) hipLaunchKernelGGL(k_fps<32>, dim3(B), dim3(FPS_THREADS), 0, s, xyz, N, npoint, idx);
    else return HVPR_ERR_UNSUPPORTED;   // more than 32768 points per sample
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_ball_query_f32(const float *xyz, const float *new_xyz, int B, int N, int M, float radius, int nsample,
                                   int32_t *idx, hvpr_stream_t stream) {
    if (!xyz || !new_xyz || !idx || B < 1 || N < 1 || M < 1 || nsample < 1 || !(radius > 0.f)) return HVPR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_ball_query, dim3(B * hvpr_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, B, N, M, radius,
                       nsample, idx);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_three_nn_f32(const float *unknown, const float *known, int B, int n, int m, float *dist, int32_t *idx,
                                 hvpr_stream_t stream) {
    if (!unknown || !known || !dist || !idx || B < 1 || n < 1 || m < 3) return HVPR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_three_nn, dim3(hvpr_cdiv((long long)B * n, 256)), dim3(256), 0, (hipStream_t)stream, unknown, known, B, n, m,
                       dist, idx);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
