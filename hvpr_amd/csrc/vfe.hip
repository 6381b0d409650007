// a2 — fused per-pillar PointNet VFE (eval mode, BatchNorm folded into weights/bias by the caller).
// Replaces the ~10 PyTorch ops of PillarVFE_Scale.forward (pcdet/models/backbones_3d/vfe/pillar_vfe.py:184-221)
// and the two PFNLayer.forward calls (:29-49) with one launch.
//
// Mapping (wave64): one wave works on two pillars at a time.
//   phase A  lane = slot (32 lanes per pillar): load the point, pillar mean by a 32-lane butterfly,
//            10-d decoration, masked; layer 0 (10->16) in registers; 32-lane max.
//   phase B  lane = output channel (64 lanes = 64 channels), one pillar after the other: layer 1 over the
//            DISTINCT slots only — the n valid points plus, when n < 32, ONE virtual zero-input slot
//            (pillar_vfe.py masks the input, not the output, so every padded slot yields the same
//            ReLU(folded bias) and takes part in both maxes).  The x_max half of
//            the concat is a per-pillar constant and is folded into the bias once.
//   scale    5 -> 16 -> 32 on lanes 0..15 / 0..31.
#include "common.h"

namespace {

constexpr int C0 = 16;    // layer-0 outputs (NUM_FILTERS[0] / 2)
constexpr int C1 = 64;    // layer-1 outputs
constexpr int CIN = 10;   // x y z r + cluster(3) + center(3)
constexpr int CS0 = 16, CS1 = 32;

__device__ __forceinline__ float rlf(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

__global__ void __launch_bounds__(256) k_vfe(const float4 *__restrict__ voxels, const int *__restrict__ num_points,
                                             const int4 *__restrict__ coords, int M, int P,
                                             const int *__restrict__ m_device, float vsx, float vsy, float vsz,
                                             float offx, float offy, float offz, const float *__restrict__ w0,
                                             const float *__restrict__ b0, const float *__restrict__ w1,
                                             const float *__restrict__ b1, const float *__restrict__ ws0,
                                             const float *__restrict__ bs0, const float *__restrict__ ws1,
                                             const float *__restrict__ bs1, float *__restrict__ pillar_features,
                                             float *__restrict__ scale_features, float *__restrict__ pillar_mask) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, slot = lane & 31;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    if (m_device) M = min(M, *m_device);

    // lane = channel weights, resident for the whole grid-stride loop
    float w1a[C0], w1b[C0];
#pragma unroll
    for (int j = 0; j < C0; ++j) { w1a[j] = w1[lane * 32 + j]; w1b[j] = w1[lane * 32 + C0 + j]; }
    const float bias1 = b1[lane];
    float wsa[5], wsb[CS0];
#pragma unroll
    for (int j = 0; j < 5; ++j) wsa[j] = ws0[(lane & 15) * 5 + j];
#pragma unroll
    for (int j = 0; j < CS0; ++j) wsb[j] = ws1[(lane & 31) * CS0 + j];
    const float bsa = bs0[lane & 15], bsb = bs1[lane & 31];

    for (int pair = wave; pair * 2 < M; pair += n_waves) {
        const int p = pair * 2 + half;
        const bool pv = p < M;
        const int n = pv ? num_points[p] : 0;
        const int4 cd = pv ? coords[p] : make_int4(0, 0, 0, 0);
        float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool valid = pv && slot < n && slot < P;
        if (pv && slot < P) pt = voxels[(size_t)p * P + slot];
        // ---- phase A: decoration + layer 0, lane = slot -------------------------------------------------
        const float fn = (float)n;
        const float mx = hvpr_reduce_sum<32>(pt.x) / fn;     // padded slots are zero (pillar_vfe.py:187)
        const float my = hvpr_reduce_sum<32>(pt.y) / fn;
        const float mz = hvpr_reduce_sum<32>(pt.z) / fn;
        float f[CIN];
        f[0] = pt.x; f[1] = pt.y; f[2] = pt.z; f[3] = pt.w;
        f[4] = pt.x - mx; f[5] = pt.y - my; f[6] = pt.z - mz;
        f[7] = pt.x - ((float)cd.w * vsx + offx);
        f[8] = pt.y - ((float)cd.z * vsy + offy);
        f[9] = pt.z - ((float)cd.y * vsz + offz);
        if (!valid) {
#pragma unroll
            for (int j = 0; j < CIN; ++j) f[j] = 0.f;
        }
        float y0[C0], xmax[C0];
#pragma unroll
        for (int c = 0; c < C0; ++c) {
            float a = b0[c];
#pragma unroll
            for (int j = 0; j < CIN; ++j) a = fmaf(w0[c * CIN + j], f[j], a);
            y0[c] = fmaxf(a, 0.f);
            xmax[c] = hvpr_reduce_max<32>(slot < P ? y0[c] : -INFINITY);   // all P slots take part, padded ones included
        }
        if (pillar_mask && pv && slot < P) pillar_mask[(size_t)p * P + slot] = valid ? 1.f : 0.f;

        // ---- phase B: layer 1, lane = channel, one pillar of the pair after the other ------------------
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ph = pair * 2 + h;
            if (ph >= M) break;
            const int src0 = h * 32;
            const int nh = __builtin_amdgcn_readlane(n, src0);
            float cst = bias1;   // bias + W1[:,16:32] . xmax  (per-pillar constant)
#pragma unroll
            for (int j = 0; j < C0; ++j) cst = fmaf(w1b[j], rlf(xmax[j], src0), cst);
            float best = -INFINITY;
            const int n_real = min(nh, P);
            const int n_eval = n_real < P ? n_real + 1 : n_real;   // +1 virtual zero-input slot
            for (int s = 0; s < n_eval; ++s) {
                // lane (src0 + s) holds y0 of slot s; slot n_real (if evaluated) is a padded slot -> ReLU(b0)
                float a = cst;
#pragma unroll
                for (int j = 0; j < C0; ++j)
                    a = fmaf(w1a[j], rlf(y0[j], src0 + s), a);
                best = fmaxf(best, fmaxf(a, 0.f));
            }
            pillar_features[(size_t)ph * C1 + lane] = best;

            // scale stream: [n, |mean|, mean_x, mean_y, mean_z] -> 16 -> 32   (pillar_vfe.py:213-216)
            const float smx = rlf(mx, src0);
            const float smy = rlf(my, src0);
            const float smz = rlf(mz, src0);
            const float nrm = sqrtf(smx * smx + smy * smy + smz * smz);
            float s1 = bsa;
            s1 = fmaf(wsa[0], (float)nh, s1);
            s1 = fmaf(wsa[1], nrm, s1);
            s1 = fmaf(wsa[2], smx, s1);
            s1 = fmaf(wsa[3], smy, s1);
            s1 = fmaf(wsa[4], smz, s1);
            s1 = fmaxf(s1, 0.f);   // lanes 0..15 hold channel (lane & 15)
            float s2 = bsb;
#pragma unroll
            for (int j = 0; j < CS0; ++j)
                s2 = fmaf(wsb[j], rlf(s1, j), s2);
            if (lane < CS1) scale_features[(size_t)ph * CS1 + lane] = fmaxf(s2, 0.f);
        }
    }
}

}  // namespace

extern "C" int hvpr_pillar_vfe_fwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coords, int M,
                                       int P, const int32_t *m_device, float vs_x, float vs_y, float vs_z, float off_x,
                                       float off_y, float off_z, const float *w0, const float *b0, const float *w1,
                                       const float *b1, const float *ws0, const float *bs0, const float *ws1,
                                       const float *bs1, float *pillar_features, float *pillar_scale_features,
                                       float *pillar_mask, hvpr_stream_t stream) {
    if (M < 0 || P < 1) return HVPR_ERR_INVALID_ARG;
    if (P > 32) return HVPR_ERR_UNSUPPORTED;
    if (M == 0) return HVPR_OK;
    if (!voxels || !num_points || !coords || !w0 || !b0 || !w1 || !b1 || !ws0 || !bs0 || !ws1 || !bs1 ||
        !pillar_features || !pillar_scale_features)
        return HVPR_ERR_INVALID_ARG;
    int blocks = hvpr_cdiv(hvpr_cdiv(M, 2), 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_vfe, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)voxels, num_points,
                       (const int4 *)coords, M, P, m_device, vs_x, vs_y, vs_z, off_x, off_y, off_z, w0, b0, w1, b1, ws0,
                       bs0, ws1, bs1, pillar_features, pillar_scale_features, pillar_mask);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
