// SpatialAttention in TRAINING mode (reference: pcdet/models/backbones_2d/spatial_attention.py:47-63 —
// ChannelPool (max / mean over channels) -> 3x3 convolution 2 -> 1 with bias -> BatchNorm2d(1) with BATCH statistics ->
// sigmoid), forward and backward, on NHWC activations.  The gate depends on the scale stream only, so the backbone computes it
// once per level and uses it in all 2 x SFM_LAYER_NUMS steps (base_bev_backbone.py:250-257 calls the module once per step with
// the same input: the same values every time; the host side applies the running-statistics update once per such call).
//
//   forward   k_gt_pool      pooled[n,y,x] = (max_c y, mean_c y), argmax_c (lowest index on ties)
//             k_gt_conv      a = conv3x3(pooled) + bias, per-workgroup partial (sum a, sum a^2)
//             k_gt_sums      (count, sum, sum of squares) in double, fixed order: deterministic  [SyncBatchNorm hook here]
//             k_gt_stats     mean, biased variance, 1/sqrt(var + eps) of a over the batch
//             k_gt_apply     gate = sigmoid((a - mean) * invstd * gamma + beta)
//   backward  k_gt_bwd_red   ds = dgate * gate * (1 - gate); partial (sum ds, sum ds * xhat)
//             k_gt_bwd_fin   dbeta = sum ds, dgamma = sum ds * xhat                              [SyncBatchNorm hook here]
//             k_gt_bwd_da    da = gamma * invstd * (ds - dbeta / n - xhat * dgamma / n)      (batch statistics differentiated through)
//             k_gt_bwd_conv  d pooled = conv_transpose(da, w); dy = d mean / C + [c == argmax] d max; partial dW (18), dbias
//             k_gt_bwd_fin2  dW, dbias
#include "common.h"

namespace {

constexpr int GT = 16;   // tile edge of the convolution kernels (256 threads)

__global__ void __launch_bounds__(256) k_gt_pool(const float *__restrict__ y, long long P, int C, float2 *__restrict__ pooled,
                                                 int *__restrict__ argmax) {
    // 8 lanes per pixel, float4 each: 32 pixels per workgroup
    const int sub = threadIdx.x & 7;
    const long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    float mx = -INFINITY, sm = 0.f;
    int am = 0x7fffffff;
    if (p < P) {
        const float4 *src = (const float4 *)(y + (size_t)p * C);
        for (int c4 = sub; c4 < C / 4; c4 += 8) {
            const float4 v = src[c4];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e[k] > mx) { mx = e[k]; am = 4 * c4 + k; }
            sm += (v.x + v.y) + (v.z + v.w);
        }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
        sm += __shfl_xor(sm, o, 64);
    }
    if (p < P && sub == 0) {
        pooled[p] = make_float2(mx, sm / (float)C);
        argmax[p] = am;
    }
}

// block-wide sum of up to NV values per thread -> out[NV] (thread 0 writes); fixed order
template <int NV>
__device__ __forceinline__ void block_sums(float (&v)[NV], float *out) {
    __shared__ float s_red[4][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = hvpr_reduce_sum<64>(v[k]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) s_red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) out[k] = (s_red[0][k] + s_red[1][k]) + (s_red[2][k] + s_red[3][k]);
}

__global__ void __launch_bounds__(256) k_gt_conv(const float2 *__restrict__ pooled, int H, int W, const float *__restrict__ w18,
                                                 const float *__restrict__ bias, float *__restrict__ a, float *__restrict__ part) {
    __shared__ float2 s_p[(GT + 2) * (GT + 2)];
    const int tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
    for (int p = threadIdx.x; p < (GT + 2) * (GT + 2); p += 256) {
        const int iy = ty * GT + p / (GT + 2) - 1, ix = tx * GT + p % (GT + 2) - 1;
        s_p[p] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? pooled[((size_t)n * H + iy) * W + ix] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const int ly = threadIdx.x / GT, lx = threadIdx.x % GT;
    const int oy = ty * GT + ly, ox = tx * GT + lx;
    float v[2] = {0.f, 0.f};
    if (oy < H && ox < W) {
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float2 q = s_p[(ly + ky) * (GT + 2) + lx + kx];
                acc = fmaf(w18[ky * 3 + kx], q.x, acc);
                acc = fmaf(w18[9 + ky * 3 + kx], q.y, acc);
            }
        // statistics of the value BEFORE the bias is added: the variance does not depend on the shift, and a large conv bias
        // (|mean| >> std) would otherwise cancel in E[a^2] - mean^2 of fp32 partial sums
        v[0] = acc; v[1] = acc * acc;
        acc += bias[0];
        a[((size_t)n * H + oy) * W + ox] = acc;
    }
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_sums<2>(v, part + blk * 2);
}

// one workgroup: sums `blocks` partial rows of NV floats in double (lanes stride, then a fixed butterfly)
template <int NV>
__device__ __forceinline__ void final_sums(const float *__restrict__ part, int blocks, double (&tot)[NV]) {
    __shared__ double s_d[256];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < blocks; i += 256) acc += (double)part[(size_t)i * NV + k];
        s_d[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) s_d[threadIdx.x] += s_d[threadIdx.x + o];
            __syncthreads();
        }
        tot[k] = s_d[0];
        __syncthreads();
    }
}

// one workgroup: tot = (count, sum, sum of squares) of the partial rows, in double
__global__ void __launch_bounds__(256) k_gt_sums(const float *__restrict__ part, int blocks, double count, double *__restrict__ tot) {
    double t[2];
    final_sums<2>(part, blocks, t);
    if (threadIdx.x == 0) { tot[0] = count; tot[1] = t[0]; tot[2] = t[1]; }
}

// tot (after the SyncBatchNorm hook, if any: sums of (a - bias) and (a - bias)^2 over the global batch) -> mean, variance, 1/std
__global__ void __launch_bounds__(64) k_gt_stats(const double *__restrict__ tot, float eps, const float *__restrict__ bias,
                                                 float *__restrict__ stats) {
    if (threadIdx.x == 0) {
        const double count = tot[0];
        const double ms = tot[1] / count;
        double var = tot[2] / count - ms * ms;
        if (var < 0.0) var = 0.0;
        const double m = ms + (double)bias[0];
        stats[0] = (float)m; stats[1] = (float)var; stats[2] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ void __launch_bounds__(256) k_gt_apply(const float *__restrict__ a, long long P, const float *__restrict__ stats,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ gate) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float z = (a[p] - stats[0]) * stats[2] * gamma[0] + beta[0];
    gate[p] = 1.f / (1.f + expf(-z));
}

__global__ void __launch_bounds__(256) k_gt_bwd_red(const float *__restrict__ dgate, const float *__restrict__ gate, const float *__restrict__ a,
                                                    long long P, const float *__restrict__ stats, float *__restrict__ part) {
    float v[2] = {0.f, 0.f};
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
        const float g = gate[p];
        const float ds = dgate[p] * g * (1.f - g);
        v[0] += ds;
        v[1] = fmaf(ds, (a[p] - stats[0]) * stats[2], v[1]);
    }
    block_sums<2>(v, part + (size_t)blockIdx.x * 2);
}

// dbeta / dgamma of THIS rank (the parameter gradients), and tot = (count, sum ds, sum ds xhat) for the hook
__global__ void __launch_bounds__(256) k_gt_bwd_fin(const float *__restrict__ part, int blocks, double count, float *__restrict__ dgamma,
                                                    float *__restrict__ dbeta, double *__restrict__ tot) {
    double t[2];
    final_sums<2>(part, blocks, t);
    if (threadIdx.x == 0) { dbeta[0] = (float)t[0]; dgamma[0] = (float)t[1]; tot[0] = count; tot[1] = t[0]; tot[2] = t[1]; }
}

__global__ void __launch_bounds__(256) k_gt_bwd_da(const float *__restrict__ dgate, const float *__restrict__ gate, const float *__restrict__ a,
                                                   long long P, const float *__restrict__ stats, const float *__restrict__ gamma,
                                                   const double *__restrict__ tot, float *__restrict__ da) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float inv_n = (float)(1.0 / tot[0]), dbeta = (float)tot[1], dgamma = (float)tot[2];      // over the (global) batch
    const float g = gate[p];
    const float ds = dgate[p] * g * (1.f - g);
    const float xh = (a[p] - stats[0]) * stats[2];
    da[p] = gamma[0] * stats[2] * (ds - dbeta * inv_n - xh * (dgamma * inv_n));
}

__global__ void __launch_bounds__(256) k_gt_bwd_conv(const float *__restrict__ da, const float2 *__restrict__ pooled, const int *__restrict__ argmax,
                                                     int H, int W, int C, const float *__restrict__ w18, float *__restrict__ dy,
                                                     float *__restrict__ part) {
    __shared__ float s_da[(GT + 2) * (GT + 2)];
    __shared__ float2 s_p[(GT + 2) * (GT + 2)];
    __shared__ float2 s_dp[GT * GT];
    const int tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
    for (int p = threadIdx.x; p < (GT + 2) * (GT + 2); p += 256) {
        const int iy = ty * GT + p / (GT + 2) - 1, ix = tx * GT + p % (GT + 2) - 1;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const size_t q = ((size_t)n * H + iy) * W + ix;
        s_da[p] = in ? da[q] : 0.f;
        s_p[p] = in ? pooled[q] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const int ly = threadIdx.x / GT, lx = threadIdx.x % GT;
    const int oy = ty * GT + ly, ox = tx * GT + lx;
    const bool live = oy < H && ox < W;
    float v[19];
#pragma unroll
    for (int k = 0; k < 19; ++k) v[k] = 0.f;
    float dmax = 0.f, dmean = 0.f;
    if (live) {
        const float d0 = s_da[(ly + 1) * (GT + 2) + lx + 1];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // dW[ch][ky][kx] += da[o] * P[ch][o + (ky-1, kx-1)]
                const float2 q = s_p[(ly + ky) * (GT + 2) + lx + kx];
                v[ky * 3 + kx] = d0 * q.x;
                v[9 + ky * 3 + kx] = d0 * q.y;
                // dP[ch][o] = sum_k w[ch][k] * da[o - (ky-1, kx-1)]
                const float d = s_da[(ly + 2 - ky) * (GT + 2) + lx + 2 - kx];
                dmax = fmaf(w18[ky * 3 + kx], d, dmax);
                dmean = fmaf(w18[9 + ky * 3 + kx], d, dmean);
            }
        v[18] = d0;
    }
    s_dp[threadIdx.x] = make_float2(dmax, dmean / (float)C);
    __syncthreads();
    // dy rows of the tile: float4 per thread, 256 threads walk the GT*GT pixels x C/4 groups
    const int groups = C / 4;
    for (int t = threadIdx.x; t < GT * GT * groups; t += 256) {
        const int px = t / groups, g = t - px * groups;
        const int y_ = ty * GT + px / GT, x_ = tx * GT + px % GT;
        if (y_ >= H || x_ >= W) continue;
        const size_t q = ((size_t)n * H + y_) * W + x_;
        const float2 dp = s_dp[px];
        const int am = argmax[q] - 4 * g;
        float4 r = make_float4(dp.y, dp.y, dp.y, dp.y);
        if (am == 0) r.x += dp.x;
        else if (am == 1) r.y += dp.x;
        else if (am == 2) r.z += dp.x;
        else if (am == 3) r.w += dp.x;
        *(float4 *)(dy + q * C + 4 * g) = r;
    }
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_sums<19>(v, part + blk * 19);
}

__global__ void __launch_bounds__(256) k_gt_bwd_fin2(const float *__restrict__ part, int blocks, float *__restrict__ dw18, float *__restrict__ dbias) {
    double t[19];
    final_sums<19>(part, blocks, t);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 18; ++k) dw18[k] = (float)t[k];
        dbias[0] = (float)t[18];
    }
}

inline int red_blocks(long long P) {
    long long b = (P + 2047) / 2048;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" size_t hvpr_spatial_gate_train_workspace_bytes(int N, int H, int W) {
    if (N < 1 || H < 1 || W < 1) return 0;
    const size_t tiles = (size_t)N * hvpr_cdiv(H, GT) * hvpr_cdiv(W, GT);
    const size_t P = (size_t)N * H * W;
    return ((tiles * 19 + 1024 * 2) * sizeof(float) + 255) / 256 * 256 + (P * sizeof(float) + 255) / 256 * 256 + 256;
}

extern "C" int hvpr_spatial_gate_train_fwd_f32(const float *y, int N, int H, int W, int C, const float *w18, const float *conv_bias,
                                               const float *gamma, const float *beta, float eps, float *pooled, int32_t *argmax, float *a,
                                               float *stats, float *gate, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!y || !w18 || !conv_bias || !gamma || !beta || !pooled || !argmax || !a || !stats || !gate || !workspace || N < 1 || H < 1 || W < 1)
        return HVPR_ERR_INVALID_ARG;
    if (C < 4 || C % 4 != 0) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < hvpr_spatial_gate_train_workspace_bytes(N, H, W)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)N * H * W;
    float *part = (float *)workspace;
    const dim3 tiles(hvpr_cdiv(W, GT), hvpr_cdiv(H, GT), N);
    const int n_tiles = (int)(tiles.x * tiles.y * tiles.z);
    hipLaunchKernelGGL(k_gt_pool, dim3(hvpr_cdiv(P, 32)), dim3(256), 0, s, y, P, C, (float2 *)pooled, argmax);
    hipLaunchKernelGGL(k_gt_conv, tiles, dim3(256), 0, s, (const float2 *)pooled, H, W, w18, conv_bias, a, part);
    double *tot = (double *)((char *)workspace + hvpr_spatial_gate_train_workspace_bytes(N, H, W) - 256);
    hipLaunchKernelGGL(k_gt_sums, dim3(1), dim3(256), 0, s, (const float *)part, n_tiles, (double)P, tot);
    if (hvpr_i_bn_allreduce(tot, 3, s) != 0) return HVPR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_gt_stats, dim3(1), dim3(64), 0, s, (const double *)tot, eps, conv_bias, stats);
    hipLaunchKernelGGL(k_gt_apply, dim3(hvpr_cdiv(P, 256)), dim3(256), 0, s, (const float *)a, P, (const float *)stats, gamma, beta, gate);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_spatial_gate_train_bwd_f32(const float *dgate, const float *gate, const float *a, const float *stats, const float *pooled,
                                               const int32_t *argmax, const float *w18, const float *gamma, int N, int H, int W, int C,
                                               float *dy, float *dw18, float *dbias, float *dgamma, float *dbeta, void *workspace,
                                               size_t workspace_bytes, hvpr_stream_t stream) {
    if (!dgate || !gate || !a || !stats || !pooled || !argmax || !w18 || !gamma || !dy || !dw18 || !dbias || !dgamma || !dbeta || !workspace ||
        N < 1 || H < 1 || W < 1)
        return HVPR_ERR_INVALID_ARG;
    if (C < 4 || C % 4 != 0) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < hvpr_spatial_gate_train_workspace_bytes(N, H, W)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)N * H * W;
    const dim3 tiles(hvpr_cdiv(W, GT), hvpr_cdiv(H, GT), N);
    const int n_tiles = (int)(tiles.x * tiles.y * tiles.z);
    float *part = (float *)workspace;
    float *da = (float *)((char *)workspace + (((size_t)n_tiles * 19 + 1024 * 2) * sizeof(float) + 255) / 256 * 256);
    const int rb = red_blocks(P);
    hipLaunchKernelGGL(k_gt_bwd_red, dim3(rb), dim3(256), 0, s, dgate, gate, a, P, stats, part);
    double *tot = (double *)((char *)workspace + hvpr_spatial_gate_train_workspace_bytes(N, H, W) - 256);
    hipLaunchKernelGGL(k_gt_bwd_fin, dim3(1), dim3(256), 0, s, (const float *)part, rb, (double)P, dgamma, dbeta, tot);
    if (hvpr_i_bn_allreduce(tot, 3, s) != 0) return HVPR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_gt_bwd_da, dim3(hvpr_cdiv(P, 256)), dim3(256), 0, s, dgate, gate, a, P, stats, gamma, (const double *)tot, da);
    hipLaunchKernelGGL(k_gt_bwd_conv, tiles, dim3(256), 0, s, (const float *)da, (const float2 *)pooled, argmax, H, W, C, w18, dy, part);
    hipLaunchKernelGGL(k_gt_bwd_fin2, dim3(1), dim3(256), 0, s, (const float *)part, n_tiles, dw18, dbias);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
