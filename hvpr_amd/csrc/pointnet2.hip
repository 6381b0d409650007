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

// One workgroup per sample, the points and their running minimum distances in registers.  An iteration = distance update +
// arg-max (largest distance, lowest index on ties).  The arg-max is two DPP wave reductions (max of the distance, then min of the
// index among the lanes that hold it) and one LDS exchange between the waves; the winner's coordinates travel with it, so the next
// iteration does not start with a dependent global load (4096 iterations x (L2 round trip + 12 ds_bpermute) was 2 us each).
__device__ __forceinline__ int fps_dpp_min_i32(int v) {
    auto step = [](int x, int y) { return x < y ? x : y; };
    v = step(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));
    v = step(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));
    v = step(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
    v = step(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
    {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        v = step((int)r[0], (int)r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        v = step((int)r[0], (int)r[1]);
    }
    return v;
}

template <int PPT>
__global__ void __launch_bounds__(FPS_THREADS) k_fps(const float *__restrict__ xyz, int N, int npoint, int *__restrict__ out) {
    __shared__ float4 s_best[2][FPS_THREADS / 64];        // per wave: (distance, x, y, z) of its winner ...
    __shared__ int s_idx[2][FPS_THREADS / 64];            // ... and its index
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
    float lx = p[0], ly = p[1], lz = p[2];
    if (t == 0) out[(size_t)b * npoint] = 0;      // first pick is index 0
    for (int j = 1; j < npoint; ++j) {
        // this thread's best: largest distance, lowest index on ties (k ascending = index ascending: strict > keeps the first)
        float bd = -1.f, bx = 0.f, by = 0.f, bz = 0.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const float dx = px[k] - lx, dy = py[k] - ly, dz = pz[k] - lz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (md[k] >= 0.f) {
                md[k] = fminf(md[k], d);
                if (md[k] > bd) { bd = md[k]; bi = t + k * FPS_THREADS; bx = px[k]; by = py[k]; bz = pz[k]; }
            }
        }
        const float wd = hvpr_reduce_max<64>(bd);
        const int wi = fps_dpp_min_i32(bd == wd ? bi : 0x7fffffff);
        const int buf = j & 1;
        if (bi == wi && bd == wd) { s_best[buf][wid] = make_float4(bd, bx, by, bz); s_idx[buf][wid] = bi; }
        __syncthreads();
        float4 g = s_best[buf][0];
        int gi = s_idx[buf][0];
#pragma unroll
        for (int w = 1; w < FPS_THREADS / 64; ++w) {
            const float4 c = s_best[buf][w];
            const int ci = s_idx[buf][w];
            if (c.x > g.x || (c.x == g.x && ci < gi)) { g = c; gi = ci; }
        }
        lx = g.y; ly = g.z; lz = g.w;
        if (t == 0) out[(size_t)b * npoint + j] = gi;
    }
}

// ---------------------------------------------------------------------------------------------------- ball query
// idx[b,m,:] = the first nsample points (index order) with d2 < radius^2; the first hit pre-fills every slot; 0 if none.
__global__ void __launch_bounds__(256) k_ball_query(const float *__restrict__ xyz, const float *__restrict__ new_xyz, int B, int N,
                                                    int M, float radius, int nsample, int *__restrict__ idx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * M) return;
    const int b = (int)(t / M);
    const float *q = new_xyz + (size_t)t * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float r2 = radius * radius;
    const float *p = xyz + (size_t)b * N * 3;
    int *o = idx + (size_t)t * nsample;
    int cnt = 0;
    for (int k = 0; k < N && cnt < nsample; ++k) {
        const float dx = qx - p[k * 3 + 0], dy = qy - p[k * 3 + 1], dz = qz - p[k * 3 + 2];
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (d2 < r2) {
            if (cnt == 0)
                for (int l = 0; l < nsample; ++l) o[l] = k;
            o[cnt++] = k;
        }
    }
    if (cnt == 0)
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
    hipLaunchKernelGGL(k_ball_query, dim3(hvpr_cdiv((long long)B * M, 256)), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, B, N,
                       M, radius, nsample, idx);
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
