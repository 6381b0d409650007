// Row-major data movement around the shared MLPs of the PointNet++ point stream (reference:
// pcdet/models/backbones_3d/pointnet2_backbone.py:27-34 PointnetSAModuleMSG, :40-47 PointnetFPModule, whose natives
// pcdet/ops/pointnet2/pointnet2_batch are absent — setup.py:94-109).
//
// The shared MLPs themselves (1x1 convolution + train-mode BatchNorm + ReLU per layer) run on the library's own matrix-core
// convolution and BatchNorm kernels (hvpr_conv2d_nhwc_f32 / hvpr_conv2d_wgrad_nhwc_f32 / hvpr_bn_*: csrc/conv_igemm.hip,
// csrc/conv_train.hip) over a ROW layout: one row = one (group, sample) or one point, channels contiguous, padded with zero
// columns to a multiple of 8 (the convolution kernel's K chunk).  This file holds what surrounds them:
//
//   hvpr_group_rows_f32 / _grad    QueryAndGroup (use_xyz: xyz channels first): row (b, j, s) = [xyz[idx] - new_xyz[j] | feats[idx] | 0]
//   hvpr_max_samples_f32 / _grad   max over the nsample rows of a group (F.max_pool2d over the sample axis), arg-max kept
//   hvpr_fp_rows_f32 / _grad       PointnetFPModule's input: row (b, i) = [sum_k w_k known[idx_k] | skip[i] | 0]
//
// Gradients that scatter (a point belongs to many groups; a known point feeds many unknown ones) are accumulated with float
// atomics into a zeroed buffer.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) k_group_rows(const float *__restrict__ xyz, const float *__restrict__ feat,
                                                    const float *__restrict__ new_xyz, const int *__restrict__ idx, int N, int C,
                                                    int np, int ns, int cpad, long long total, float *__restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const long long row = t / cpad;                    // (b, j, s)
    const int col = (int)(t - row * cpad);
    const long long g = row / ns;                      // (b, j)
    const int b = (int)(g / np);
    float v = 0.f;
    if (col < 3 + C) {
        const int i = idx[row];
        if (col < 3) v = xyz[((size_t)b * N + i) * 3 + col] - new_xyz[(size_t)g * 3 + col];
        else v = feat[((size_t)b * N + i) * C + (col - 3)];
    }
    out[t] = v;
}

__global__ void __launch_bounds__(256) k_group_rows_grad(const float *__restrict__ gout, const int *__restrict__ idx, int N, int C,
                                                         int np, int ns, int cpad, long long total, float *__restrict__ gfeat) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // over rows x C
    if (t >= total) return;
    const long long row = t / C;
    const int c = (int)(t - row * C);
    const int b = (int)(row / ((long long)np * ns));
    atomicAdd(gfeat + ((size_t)b * N + idx[row]) * C + c, gout[(size_t)row * cpad + 3 + c]);
}

// out[g][c] = max_s y[g][s][c]; the LOWEST s among equal maxima is recorded (torch's max_pool / max(dim) rule on ties)
__global__ void __launch_bounds__(256) k_max_samples(const float *__restrict__ y, long long G, int ns, int C, float *__restrict__ out,
                                                     unsigned char *__restrict__ arg) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * C) return;
    const long long g = t / C;
    const int c = (int)(t - g * C);
    const float *p = y + (size_t)g * ns * C + c;
    float m = p[0];
    int a = 0;
    for (int s = 1; s < ns; ++s) {
        const float v = p[(size_t)s * C];
        if (v > m) { m = v; a = s; }
    }
    out[t] = m;
    arg[t] = (unsigned char)a;
}

__global__ void __launch_bounds__(256) k_max_samples_grad(const float *__restrict__ gout, const unsigned char *__restrict__ arg, long long G,
                                                          int ns, int C, float *__restrict__ gy) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // over G x ns x C
    if (t >= G * ns * C) return;
    const long long gs = t / C;
    const int c = (int)(t - gs * C);
    const long long g = gs / ns;
    const int s = (int)(gs - g * ns);
    gy[t] = arg[g * C + c] == s ? gout[g * C + c] : 0.f;
}

__global__ void __launch_bounds__(256) k_fp_rows(const float *__restrict__ known, const int *__restrict__ idx, const float *__restrict__ w,
                                                 const float *__restrict__ skip, int m, int n, int C1, int C2, int cpad, long long total,
                                                 float *__restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const long long row = t / cpad;                    // (b, i)
    const int col = (int)(t - row * cpad);
    const int b = (int)(row / n);
    float v = 0.f;
    if (col < C1) {
        const int *ii = idx + (size_t)row * 3;
        const float *ww = w + (size_t)row * 3;
        const float *f = known + (size_t)b * m * C1 + col;
        v = (f[(size_t)ii[0] * C1] * ww[0] + f[(size_t)ii[1] * C1] * ww[1]) + f[(size_t)ii[2] * C1] * ww[2];
    } else if (col < C1 + C2) {
        v = skip[(size_t)row * C2 + (col - C1)];
    }
    out[t] = v;
}

__global__ void __launch_bounds__(256) k_fp_rows_grad(const float *__restrict__ gout, const int *__restrict__ idx, const float *__restrict__ w,
                                                      int m, int n, int C1, int C2, int cpad, long long total, float *__restrict__ gknown,
                                                      float *__restrict__ gskip) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // over rows x (C1 + C2)
    if (t >= total) return;
    const int cc = C1 + C2;
    const long long row = t / cc;
    const int col = (int)(t - row * cc);
    const float g = gout[(size_t)row * cpad + col];
    if (col < C1) {
        const int b = (int)(row / n);
        const int *ii = idx + (size_t)row * 3;
        const float *ww = w + (size_t)row * 3;
        float *f = gknown + (size_t)b * m * C1 + col;
        atomicAdd(f + (size_t)ii[0] * C1, g * ww[0]);
        atomicAdd(f + (size_t)ii[1] * C1, g * ww[1]);
        atomicAdd(f + (size_t)ii[2] * C1, g * ww[2]);
    } else if (gskip) {
        gskip[(size_t)row * C2 + (col - C1)] = g;
    }
}

__global__ void __launch_bounds__(256) k_zero_f(float *__restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.f;
}

void zero_f(float *p, long long n, hipStream_t s) {
    if (n <= 0) return;
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_zero_f, dim3((unsigned)blocks), dim3(256), 0, s, p, n);
}

inline unsigned grid_of(long long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" int hvpr_group_rows_f32(const float *xyz, const float *features, const float *new_xyz, const int32_t *idx, int B, int N, int C,
                                   int npoint, int nsample, int cpad, float *out, hvpr_stream_t stream) {
    if (B < 0 || N < 1 || C < 0 || npoint < 0 || nsample < 1 || cpad < 3 + C) return HVPR_ERR_INVALID_ARG;
    const long long total = (long long)B * npoint * nsample * cpad;
    if (total == 0) return HVPR_OK;
    if (!xyz || !new_xyz || !idx || !out || (C > 0 && !features)) return HVPR_ERR_INVALID_ARG;
    if (total / 256 >= 0x7fffffffLL) return HVPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_group_rows, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, xyz, features, new_xyz, idx, N, C, npoint, nsample,
                       cpad, total, out);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_group_rows_grad_f32(const float *grad_out, const int32_t *idx, int B, int N, int C, int npoint, int nsample, int cpad,
                                        float *grad_features, hvpr_stream_t stream) {
    if (B < 0 || N < 1 || C < 1 || npoint < 0 || nsample < 1 || cpad < 3 + C) return HVPR_ERR_INVALID_ARG;
    if (B == 0) return HVPR_OK;
    if (!grad_features) return HVPR_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    zero_f(grad_features, (long long)B * N * C, s);
    const long long total = (long long)B * npoint * nsample * C;
    if (total > 0) {
        if (!grad_out || !idx) return HVPR_ERR_INVALID_ARG;
        if (total / 256 >= 0x7fffffffLL) return HVPR_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_group_rows_grad, dim3(grid_of(total)), dim3(256), 0, s, grad_out, idx, N, C, npoint, nsample, cpad, total, grad_features);
    }
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_max_samples_f32(const float *y, long long G, int nsample, int C, float *out, uint8_t *argmax, hvpr_stream_t stream) {
    if (G < 0 || nsample < 1 || nsample > 255 || C < 1) return HVPR_ERR_INVALID_ARG;
    if (G == 0) return HVPR_OK;
    if (!y || !out || !argmax) return HVPR_ERR_INVALID_ARG;
    if (G * C / 256 >= 0x7fffffffLL) return HVPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_max_samples, dim3(grid_of(G * C)), dim3(256), 0, (hipStream_t)stream, y, G, nsample, C, out, argmax);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_max_samples_grad_f32(const float *grad_out, const uint8_t *argmax, long long G, int nsample, int C, float *grad_y,
                                         hvpr_stream_t stream) {
    if (G < 0 || nsample < 1 || nsample > 255 || C < 1) return HVPR_ERR_INVALID_ARG;
    if (G == 0) return HVPR_OK;
    if (!grad_out || !argmax || !grad_y) return HVPR_ERR_INVALID_ARG;
    const long long total = G * nsample * C;
    if (total / 256 >= 0x7fffffffLL) return HVPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_max_samples_grad, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, grad_out, argmax, G, nsample, C, grad_y);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_fp_rows_f32(const float *known, const int32_t *idx, const float *weight, const float *skip, int B, int m, int n, int C1,
                                int C2, int cpad, float *out, hvpr_stream_t stream) {
    if (B < 0 || m < 1 || n < 0 || C1 < 1 || C2 < 0 || cpad < C1 + C2) return HVPR_ERR_INVALID_ARG;
    const long long total = (long long)B * n * cpad;
    if (total == 0) return HVPR_OK;
    if (!known || !idx || !weight || !out || (C2 > 0 && !skip)) return HVPR_ERR_INVALID_ARG;
    if (total / 256 >= 0x7fffffffLL) return HVPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_fp_rows, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, known, idx, weight, skip, m, n, C1, C2, cpad, total, out);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_fp_rows_grad_f32(const float *grad_out, const int32_t *idx, const float *weight, int B, int m, int n, int C1, int C2,
                                     int cpad, float *grad_known, float *grad_skip, hvpr_stream_t stream) {
    if (B < 0 || m < 1 || n < 0 || C1 < 1 || C2 < 0 || cpad < C1 + C2) return HVPR_ERR_INVALID_ARG;
    if (B == 0) return HVPR_OK;
    if (!grad_known) return HVPR_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    zero_f(grad_known, (long long)B * m * C1, s);
    const long long total = (long long)B * n * (C1 + C2);
    if (total > 0) {
        if (!grad_out || !idx || !weight) return HVPR_ERR_INVALID_ARG;
        if (total / 256 >= 0x7fffffffLL) return HVPR_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_fp_rows_grad, dim3(grid_of(total)), dim3(256), 0, s, grad_out, idx, weight, m, n, C1, C2, cpad, total, grad_known,
                           grad_skip);
    }
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
