// Training-only data movement of the point stream and the optimiser step (SURVEY.md §8a rows a9 and a14, §8b export list).
//
//   hvpr_group_points_f32 / _grad      grouping_operation of the absent pcdet/ops/pointnet2/pointnet2_batch natives (setup.py:
//                                      94-109; call sites pointnet2_backbone.py:27-34 through PointnetSAModuleMSG): features
//                                      (B, C, N), idx (B, np, ns) -> (B, C, np, ns); ns == 1 is gather_operation.
//   hvpr_three_interpolate_f32 / _grad three_interpolate of the same package (PointnetFPModule, pointnet2_backbone.py:43-47,
//                                      86-89): features (B, C, m), idx / weight (B, n, 3) -> (B, C, n).
//   hvpr_fused_adam_truewd_f32         one launch over a flat parameter buffer: decoupled ("true") weight decay
//                                      p *= 1 - wd * lr followed by the Adam step — OptimWrapper.step,
//                                      tools/train_utils/optimization/fastai_optim.py:132-149, with the gradient-norm clip of
//                                      tools/train_utils/train_utils.py:41 folded in as a device-side scale.
// The tensor layouts are the reference's (channel-major point features), so the wrappers drop into its modules.
#include "common.h"

namespace {

// out[b][c][j] = features[b][c][idx[b][j]]; one thread per (b, j) walks a strip of channels so that the index is read once and
// the writes of a wave are contiguous in j
constexpr int kStrip = 8;

__global__ void __launch_bounds__(256) k_group_points(const float *__restrict__ feat, const int *__restrict__ idx, int C, int N,
                                                      long long J, float *__restrict__ out) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z, c0 = blockIdx.y * kStrip;
    if (j >= J) return;
    const int i = idx[(size_t)b * J + j];
    const float *f = feat + ((size_t)b * C + c0) * N + i;
    float *o = out + ((size_t)b * C + c0) * J + j;
#pragma unroll
    for (int c = 0; c < kStrip; ++c)
        if (c0 + c < C) o[(size_t)c * J] = f[(size_t)c * N];
}

__global__ void __launch_bounds__(256) k_group_points_grad(const float *__restrict__ gout, const int *__restrict__ idx, int C, int N,
                                                           long long J, float *__restrict__ gfeat) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z, c0 = blockIdx.y * kStrip;
    if (j >= J) return;
    const int i = idx[(size_t)b * J + j];
    float *f = gfeat + ((size_t)b * C + c0) * N + i;
    const float *o = gout + ((size_t)b * C + c0) * J + j;
#pragma unroll
    for (int c = 0; c < kStrip; ++c)
        if (c0 + c < C) atomicAdd(f + (size_t)c * N, o[(size_t)c * J]);
}

// out[b][c][i] = sum_k features[b][c][idx[b][i][k]] * weight[b][i][k], k = 0..2 in that order
__global__ void __launch_bounds__(256) k_three_interpolate(const float *__restrict__ feat, const int *__restrict__ idx,
                                                           const float *__restrict__ w, int C, int m, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z, c0 = blockIdx.y * kStrip;
    if (i >= n) return;
    const size_t t = ((size_t)b * n + i) * 3;
    const int i0 = idx[t], i1 = idx[t + 1], i2 = idx[t + 2];
    const float w0 = w[t], w1 = w[t + 1], w2 = w[t + 2];
#pragma unroll
    for (int c = 0; c < kStrip; ++c) {
        if (c0 + c >= C) break;
        const float *f = feat + ((size_t)b * C + c0 + c) * m;
        out[((size_t)b * C + c0 + c) * n + i] = (f[i0] * w0 + f[i1] * w1) + f[i2] * w2;
    }
}

__global__ void __launch_bounds__(256) k_three_interpolate_grad(const float *__restrict__ gout, const int *__restrict__ idx,
                                                                const float *__restrict__ w, int C, int m, int n,
                                                                float *__restrict__ gfeat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z, c0 = blockIdx.y * kStrip;
    if (i >= n) return;
    const size_t t = ((size_t)b * n + i) * 3;
    const int i0 = idx[t], i1 = idx[t + 1], i2 = idx[t + 2];
    const float w0 = w[t], w1 = w[t + 1], w2 = w[t + 2];
#pragma unroll
    for (int c = 0; c < kStrip; ++c) {
        if (c0 + c >= C) break;
        const float g = gout[((size_t)b * C + c0 + c) * n + i];
        float *f = gfeat + ((size_t)b * C + c0 + c) * m;
        atomicAdd(f + i0, g * w0);
        atomicAdd(f + i1, g * w1);
        atomicAdd(f + i2, g * w2);
    }
}

// dst[idx[r]] += src[r]  (backward of the row gather dst[r] = src[idx[r]])
__global__ void __launch_bounds__(256) k_scatter_add_rows(const float *__restrict__ src, const int *__restrict__ idx, long long m, int row,
                                                          int n_dst, float *__restrict__ dst) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * row) return;
    const long long r = t / row;
    const int j = (int)(t % row), d = idx[r];
    if (d >= 0 && d < n_dst) atomicAdd(dst + (size_t)d * row + j, src[t]);
}

__global__ void __launch_bounds__(256) k_zero(float *__restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.f;
}

void zero_floats(float *p, long long n, hipStream_t s) {   // a kernel, not a memset node (captured memsets replayed wrongly on this stack)
    if (n <= 0) return;
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_zero, dim3((unsigned)blocks), dim3(256), 0, s, p, n);
}

// One thread = 4 consecutive parameters.  Same arithmetic as torch.optim.Adam (bias-corrected step size, eps added after the
// bias-corrected square root) preceded by the decoupled decay.  grad_scale (device word, may be null): the clip coefficient.
__global__ void __launch_bounds__(256) k_fused_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, long long n, float decay, float beta1, float beta2,
                                                    float eps, float step_size, float inv_sqrt_bc2,
                                                    const float *__restrict__ grad_scale) {
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    const float gs = grad_scale ? *grad_scale : 1.f;
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        gg *= gs;
        pp *= decay;
        mm = mm + (gg - mm) * (1.f - beta1);                 // lerp form, as torch's foreach path
        vv = vv * beta2 + (1.f - beta2) * gg * gg;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        pp -= step_size * (mm / denom);
    };
    if (i4 + 4 <= n) {
        float4 P = *(float4 *)(p + i4), M = *(float4 *)(m + i4), V = *(float4 *)(v + i4);
        const float4 G = *(const float4 *)(g + i4);
        upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
        *(float4 *)(p + i4) = P; *(float4 *)(m + i4) = M; *(float4 *)(v + i4) = V;
    } else {
        for (long long i = i4; i < n; ++i) upd(p[i], g[i], m[i], v[i]);
    }
}

}  // namespace

extern "C" int hvpr_group_points_f32(const float *features, const int32_t *idx, int B, int C, int N, int npoint, int nsample,
                                     float *out, hvpr_stream_t stream) {
    if (B < 0 || C < 1 || N < 1 || npoint < 0 || nsample < 1) return HVPR_ERR_INVALID_ARG;
    const long long J = (long long)npoint * nsample;
    if (B == 0 || J == 0) return HVPR_OK;
    if (!features || !idx || !out) return HVPR_ERR_INVALID_ARG;
    if (B > 65535 || hvpr_cdiv(C, kStrip) > 65535) return HVPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_group_points, dim3(hvpr_cdiv(J, 256), hvpr_cdiv(C, kStrip), B), dim3(256), 0, (hipStream_t)stream, features, idx, C,
                       N, J, out);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_group_points_grad_f32(const float *grad_out, const int32_t *idx, int B, int C, int N, int npoint, int nsample,
                                          float *grad_features, hvpr_stream_t stream) {
    if (B < 0 || C < 1 || N < 1 || npoint < 0 || nsample < 1) return HVPR_ERR_INVALID_ARG;
    const long long J = (long long)npoint * nsample;
    if (B == 0) return HVPR_OK;
    if (!grad_features) return HVPR_ERR_INVALID_ARG;
    if (B > 65535 || hvpr_cdiv(C, kStrip) > 65535) return HVPR_ERR_UNSUPPORTED;
    zero_floats(grad_features, (long long)B * C * N, (hipStream_t)stream);
    if (J > 0) {
        if (!grad_out || !idx) return HVPR_ERR_INVALID_ARG;
        hipLaunchKernelGGL(k_group_points_grad, dim3(hvpr_cdiv(J, 256), hvpr_cdiv(C, kStrip), B), dim3(256), 0, (hipStream_t)stream,
                           grad_out, idx, C, N, J, grad_features);
    }
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_three_interpolate_f32(const float *features, const int32_t *idx, const float *weight, int B, int C, int m, int n,
                                          float *out, hvpr_stream_t stream) {
    if (B < 0 || C < 1 || m < 1 || n < 0) return HVPR_ERR_INVALID_ARG;
    if (B == 0 || n == 0) return HVPR_OK;
    if (!features || !idx || !weight || !out) return HVPR_ERR_INVALID_ARG;
    if (B > 65535 || hvpr_cdiv(C, kStrip) > 65535) return HVPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_three_interpolate, dim3(hvpr_cdiv(n, 256), hvpr_cdiv(C, kStrip), B), dim3(256), 0, (hipStream_t)stream, features,
                       idx, weight, C, m, n, out);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_three_interpolate_grad_f32(const float *grad_out, const int32_t *idx, const float *weight, int B, int C, int m,
                                               int n, float *grad_features, hvpr_stream_t stream) {
    if (B < 0 || C < 1 || m < 1 || n < 0) return HVPR_ERR_INVALID_ARG;
    if (B == 0) return HVPR_OK;
    if (!grad_features) return HVPR_ERR_INVALID_ARG;
    if (B > 65535 || hvpr_cdiv(C, kStrip) > 65535) return HVPR_ERR_UNSUPPORTED;
    zero_floats(grad_features, (long long)B * C * m, (hipStream_t)stream);
    if (n > 0) {
        if (!grad_out || !idx || !weight) return HVPR_ERR_INVALID_ARG;
        hipLaunchKernelGGL(k_three_interpolate_grad, dim3(hvpr_cdiv(n, 256), hvpr_cdiv(C, kStrip), B), dim3(256), 0,
                           (hipStream_t)stream, grad_out, idx, weight, C, m, n, grad_features);
    }
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

// dst[d][c] = sum over the edges of destination d, in the ORDER they are listed, of w[e] * src[row[e]][off + c]: the atomic-free,
// run-to-run reproducible form of the scattering gradients (one sequential fp32 accumulation per output element)
__global__ void __launch_bounds__(256) k_segment_sum_rows(const float *__restrict__ src, long long src_stride, int src_off, int C,
                                                          const int *__restrict__ edge_row, const float *__restrict__ edge_w,
                                                          const int *__restrict__ rowptr, long long n_dst, float *__restrict__ dst,
                                                          long long dst_stride) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_dst * C) return;
    const long long d = t / C;
    const int c = (int)(t - d * C);
    const int e0 = rowptr[d], e1 = rowptr[d + 1];
    float acc = 0.f;
    int e = e0;
    for (; e + 8 <= e1; e += 8) {      // eight loads in flight, added in edge order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float x = src[(size_t)(edge_row ? edge_row[e + u] : e + u) * src_stride + src_off + c];
            v[u] = edge_w ? x * edge_w[e + u] : x;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; e < e1; ++e) {
        const float x = src[(size_t)(edge_row ? edge_row[e] : e) * src_stride + src_off + c];
        acc += edge_w ? x * edge_w[e] : x;
    }
    dst[(size_t)d * dst_stride + c] = acc;
}

extern "C" int hvpr_segment_sum_rows_f32(const float *src, long long src_stride, int src_off, int C, const int32_t *edge_row,
                                         const float *edge_w, const int32_t *rowptr, long long n_dst, float *dst, long long dst_stride,
                                         hvpr_stream_t stream) {
    if (C < 1 || src_off < 0 || src_stride < src_off + C || dst_stride < C || n_dst < 0) return HVPR_ERR_INVALID_ARG;
    if (n_dst == 0) return HVPR_OK;
    if (!src || !rowptr || !dst) return HVPR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_segment_sum_rows, dim3(hvpr_cdiv(n_dst * C, 256)), dim3(256), 0, (hipStream_t)stream, src, src_stride, src_off, C,
                       edge_row, edge_w, rowptr, n_dst, dst, dst_stride);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_scatter_add_rows_f32(const float *src, const int32_t *idx, long long m, int row_floats, int n_dst, float *dst,
                                         hvpr_stream_t stream) {
    if (m < 0 || row_floats < 1 || n_dst < 0) return HVPR_ERR_INVALID_ARG;
    if (n_dst == 0) return HVPR_OK;
    if (!dst) return HVPR_ERR_INVALID_ARG;
    zero_floats(dst, (long long)n_dst * row_floats, (hipStream_t)stream);
    if (m > 0) {
        if (!src || !idx) return HVPR_ERR_INVALID_ARG;
        hipLaunchKernelGGL(k_scatter_add_rows, dim3(hvpr_cdiv(m * row_floats, 256)), dim3(256), 0, (hipStream_t)stream, src, idx, m, row_floats,
                           n_dst, dst);
    }
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_fused_adam_truewd_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, long long n, float lr,
                                          float beta1, float beta2, float eps, float weight_decay, int step,
                                          const float *grad_scale_device, hvpr_stream_t stream) {
    if (n < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return HVPR_ERR_INVALID_ARG;
    if (n == 0) return HVPR_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return HVPR_ERR_INVALID_ARG;
    if (((size_t)params | (size_t)grads | (size_t)exp_avg | (size_t)exp_avg_sq) % 16 != 0) return HVPR_ERR_INVALID_ARG;
    // bias corrections in double, as torch does on the host (1 - beta ** step)
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const float decay = 1.f - weight_decay * lr;
    hipLaunchKernelGGL(k_fused_adam, dim3(hvpr_cdiv((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       n, decay, beta1, beta2, eps, step_size, inv_sqrt_bc2, grad_scale_device);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
