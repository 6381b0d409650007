// a11 (training) — the pieces of the two-stream BEV backbone's training step that are not the forward implicit GEMM:
//   * weight gradient of the 3x3 (stride 1 | 2, pad 1) and 1x1 convolutions on the fp32 matrix cores, split-K over pixel
//     tiles with a deterministic two-stage reduction (hvpr_conv2d_wgrad_nhwc_f32),
//   * train-mode BatchNorm + ReLU over NHWC activations: batch statistics, normalise + ReLU, and the two backward passes
//     (per-channel reductions, then dz) — four bandwidth-bound kernels instead of torch's chain of element-wise launches.
// Together with hvpr_conv2d_nhwc_f32 (forward; data gradient = the same kernel on flipped / transposed weights) they carry
// BaseBEVBackbone_Scale.forward in training mode, pcdet/models/backbones_2d/base_bev_backbone.py:228-279 (conv + BN + ReLU
// blocks :154-169, SFM steps :171-175, deblocks :177-188, scale layers :200-209).
//
// Weight gradient as a GEMM: dW[tap][co][ci] = sum_pixels dz[p][co] * x[p shifted by tap][ci]  — M = co, N = ci, K = pixels.
// Both operands are pixel-major (NHWC), which is exactly what v_mfma_f32_32x32x2_f32 wants: a lane supplies ONE float per
// operand, A[i = lane & 31][k = lane >> 5] and B[k][j = lane & 31] with k = the pixel of a pair — 32 consecutive channels of
// one pixel per half-wave, a conflict-free ds_read_b32.  A workgroup (4 waves, 2 x 2) owns a (64 MB) x (64 NB) block of
// (co, ci) for ALL taps and walks its share of the pixel tiles: the dz tile and the input halo patch of a tile are staged
// once in LDS and feed TAPS x MB x NB accumulators per wave, so the arithmetic intensity is that of the forward kernel.
// The pixel loop is fully unrolled: every LDS address is lane base + immediate, no vector-ALU instruction between the MFMAs
// (fp32 MFMA issues on the vector ALU lanes — conv_igemm.hip).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
    const float *x;     // [N, H, W, Cin]
    const float *dz;    // [N, OH, OW, Cout]
    float *part;        // [n_chunks][TAPS][Cout][Cin] partial sums
    int N, H, W, Cin, OH, OW, Cout;
    int tiles_x, tiles_y, n_pt, n_chunks, n_ci_tiles;
};

// WM x WN = 4 waves over (co, ci): 2 x 2, or 1 x 4 for layers with few output channels (the detection head: 24 — a 128-row block
// multiplies four zero rows for every real one)
template <int TAPS, int S, int TH, int TW, int MB, int NB, int WM = 2>
__global__ void __launch_bounds__(256) k_wgrad(WgradArgs a) {
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * WM * MB, BN = 32 * WN * NB;       // co x ci block of the workgroup
    constexpr int HALO = TAPS == 9 ? 2 : 0;
    constexpr int PH = (TH - 1) * S + 1 + HALO, PW = (TW - 1) * S + 1 + HALO;
    constexpr int NPX = TH * TW, NPP = PH * PW;
    constexpr int DZ_V4 = NPX * BM / 4, X_V4 = NPP * BN / 4;
    constexpr int NLD_D = (DZ_V4 + 255) / 256, NLD_X = (X_V4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float s_dz[NPX * BM];
    __shared__ __attribute__((aligned(16))) float s_x[NPP * BN];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN, half = lane >> 5, l31 = lane & 31;
    const int ot = blockIdx.x, chunk = blockIdx.y;
    const int co0 = (ot / a.n_ci_tiles) * BM, ci0 = (ot % a.n_ci_tiles) * BN;

    f32x16 acc[TAPS][MB][NB];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][mb][nb][r] = 0.f;

    float4 rd[NLD_D], rx[NLD_X];
    auto fetch = [&](int pt) {          // global -> registers (zeros outside the image / past the channel counts)
        const int tx = pt % a.tiles_x, ty = (pt / a.tiles_x) % a.tiles_y, n = pt / (a.tiles_x * a.tiles_y);
        const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 * S - HALO / 2, ix0 = ox0 * S - HALO / 2;
#pragma unroll
        for (int i = 0; i < NLD_D; ++i) {
            const int v = tid + i * 256, px = v / (BM / 4), c4 = (v % (BM / 4)) * 4;
            const int oy = oy0 + px / TW, ox = ox0 + px % TW;
            const bool ok = v < DZ_V4 && oy < a.OH && ox < a.OW && co0 + c4 < a.Cout;
            rd[i] = ok ? *(const float4 *)(a.dz + (((size_t)n * a.OH + oy) * a.OW + ox) * a.Cout + co0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NLD_X; ++i) {
            const int v = tid + i * 256, px = v / (BN / 4), c4 = (v % (BN / 4)) * 4;
            const int iy = iy0 + px / PW, ix = ix0 + px % PW;
            const bool ok = v < X_V4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && ci0 + c4 < a.Cin;
            rx[i] = ok ? *(const float4 *)(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.Cin + ci0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {               // registers -> LDS
#pragma unroll
        for (int i = 0; i < NLD_D; ++i) {
            const int v = tid + i * 256;
            if (v < DZ_V4) *(float4 *)(s_dz + v * 4) = rd[i];
        }
#pragma unroll
        for (int i = 0; i < NLD_X; ++i) {
            const int v = tid + i * 256;
            if (v < X_V4) *(float4 *)(s_x + v * 4) = rx[i];
        }
    };

    // lane bases: everything else in the unrolled pixel loop is an immediate offset
    const float *pa = s_dz + half * BM + wm * (32 * MB) + l31;           // + (even pixel of the pair) * BM + mb * 32
    const float *pb = s_x + half * (S * BN) + wn * (32 * NB) + l31;      // + patch pixel of the even one * BN + nb * 32
    int pt = chunk;
    if (pt < a.n_pt) fetch(pt);
    for (; pt < a.n_pt; pt += a.n_chunks) {
        __syncthreads();                 // everybody is done reading the previous tile
        commit();
        __syncthreads();
        if (pt + a.n_chunks < a.n_pt) fetch(pt + a.n_chunks);   // the next tile travels while this one multiplies
#pragma unroll
        for (int m = 0; m < NPX / 2; ++m) {
            // pixel of this lane half: q = 2m + half.  TW is even, so both pixels of a pair sit in the same tile row.
            const int qy = (2 * m) / TW, qx0 = (2 * m) % TW;
            float av[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = pa[(2 * m) * BM + mb * 32];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int ky = TAPS == 9 ? t / 3 : 0, kx = TAPS == 9 ? t % 3 : 0;
                float bv[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    bv[nb] = pb[((qy * S + ky) * PW + qx0 * S + kx) * BN + nb * 32];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[t][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb], bv[nb], acc[t][mb][nb], 0, 0, 0);
            }
        }
    }
    // partial sums of this chunk: [chunk][tap][co][ci], ci contiguous across the 32 lanes of a half-wave
    // C/D map of 32x32: column (ci) = lane & 31, row (co) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float *out = a.part + (size_t)chunk * TAPS * a.Cout * a.Cin;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int ci = ci0 + wn * (32 * NB) + nb * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm * (32 * MB) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < a.Cout && ci < a.Cin) out[((size_t)t * a.Cout + co) * a.Cin + ci] = acc[t][mb][nb][r];
                }
            }
}

// dW[co][ci][tap] (torch's (Cout, Cin, k, k) layout) = sum over chunks of part[chunk][tap][co][ci].  A workgroup = 64 elements x 4 chunk
// slices: thread (e, sl) adds chunks sl, sl + 4, ... in four independent partial sums (one thread per element walking all chunks is a chain
// of up to 512 dependent loads on a few dozen workgroups: 88 us per call for the one-tile layers), the slices meet in LDS in a fixed order.
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float *__restrict__ part, int n_chunks, int taps, int Cout, int Cin,
                                                      float *__restrict__ dw) {
    __shared__ float s_sl[4][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;                       // over [tap][co][ci]
    const long long per = (long long)taps * Cout * Cin;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < per) {
        const float *src = part + i;
        int c = sl;
        for (; c + 12 < n_chunks; c += 16) {
            s0 += src[(size_t)c * per]; s1 += src[(size_t)(c + 4) * per]; s2 += src[(size_t)(c + 8) * per]; s3 += src[(size_t)(c + 12) * per];
        }
        for (; c < n_chunks; c += 4) s0 += src[(size_t)c * per];
    }
    s_sl[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl != 0 || i >= per) return;
    const float s = (s_sl[0][e] + s_sl[1][e]) + (s_sl[2][e] + s_sl[3][e]);
    const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), t = (int)(i / ((long long)Cin * Cout));
    dw[((size_t)co * Cin + ci) * taps + t] = s;
}

template <int TAPS, int S, int TH, int TW, int MB, int NB, int WM = 2>
int launch_wgrad(WgradArgs a, int target_wgs, hipStream_t s) {
    constexpr int BM = 32 * WM * MB, BN = 32 * (4 / WM) * NB;
    a.tiles_x = (a.OW + TW - 1) / TW;
    a.tiles_y = (a.OH + TH - 1) / TH;
    a.n_pt = a.N * a.tiles_x * a.tiles_y;
    a.n_ci_tiles = (a.Cin + BN - 1) / BN;
    const int n_ot = a.n_ci_tiles * ((a.Cout + BM - 1) / BM);
    hipLaunchKernelGGL((k_wgrad<TAPS, S, TH, TW, MB, NB, WM>), dim3(n_ot, a.n_chunks), dim3(256), 0, s, a);
    (void)target_wgs;
    return 0;
}

// 1x1 layers with at most 32 output channels and more than 256 input channels: one 32-row block, all four waves along ci (32 x 384)
bool wgrad_few_rows(int Cin, int Cout, int taps) { return taps == 1 && Cout <= 32 && Cin > 256; }

int wgrad_chunks(int N, int OH, int OW, int Cin, int Cout, int taps, int stride) {
    const int th = (taps == 9 && stride == 1) ? 8 : 4, tw = 8;
    const long long n_pt = (long long)N * ((OH + th - 1) / th) * ((OW + tw - 1) / tw);
    const bool few = wgrad_few_rows(Cin, Cout, taps);
    const int bm = few ? 32 : (taps == 9 ? 64 : 128), bn = few ? 384 : bm;
    const int n_ot = ((Cin + bn - 1) / bn) * ((Cout + bm - 1) / bm);
    long long chunks = (512 + n_ot - 1) / n_ot;                 // ~2 workgroups per CU
    if (chunks > n_pt) chunks = n_pt;
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}

// ------------------------------------------------------------------------------------------------ BatchNorm + ReLU (NHWC)
// z [P, C] with C % 4 == 0.  A workgroup of 256 threads covers C/4 channel groups x (256 / (C/4)) pixel lanes and walks a
// slab of pixels; per-thread fp32 partial sums over <= kSlab / lanes pixels, combined per workgroup in LDS in a fixed order,
// one row per workgroup in the workspace, final sum in double by the finalising kernel: deterministic.
// Pixels per workgroup: at most 2048, fewer while the pass has under ~2048 workgroups (eight per CU: a fixed 2048 gave the 73 k pixels
// of the 512-channel level at batch 16 — 4 MB per tensor and workgroup — 36 workgroups on 256 CUs, and the 1.17 M pixels of the first
// level 573: two on some CUs, three on others).
int bn_slab(long long P) {
    int slab = 2048;
    while (slab > 64 && P / slab < 2048) slab >>= 1;
    return slab;
}

template <bool BWD>
__global__ void __launch_bounds__(256) k_bn_reduce(const float *__restrict__ z, const float *__restrict__ dy, long long P, int C,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   const float *__restrict__ mean, const float *__restrict__ invstd, int relu,
                                                   const float *__restrict__ gate /* [P] or null: dy is multiplied by it */,
                                                   int slab, float *__restrict__ ws /* [blocks][2][C] */, int dy_cstride = 0 /* 0: C */) {
    __shared__ float4 s_a[256], s_b[256];
    const int groups = C / 4;                       // channel groups of 4
    const int DC = dy_cstride > 0 ? dy_cstride : C; // row pitch of dy (a channel slice of a wider tensor: the caller offsets the pointer)
    const int lanes = 256 / groups > 0 ? 256 / groups : 1;     // pixel lanes per workgroup (groups <= 256)
    const int g = threadIdx.x % groups, pl = threadIdx.x / groups;
    const long long p0 = (long long)blockIdx.x * slab;
    const long long p1 = p0 + slab < P ? p0 + slab : P;
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B = A;
    if (pl < lanes) {
        float4 sc = A, sh = A, mu = A, is = A;
        if (BWD) { sc = *(const float4 *)(scale + 4 * g); sh = *(const float4 *)(shift + 4 * g); mu = *(const float4 *)(mean + 4 * g); is = *(const float4 *)(invstd + 4 * g); }
        // four pixels per trip, all their loads issued before the sums: a workgroup walks 1 MB with 256 threads, and one
        // load in flight per thread (the dependent-accumulator form) kept the pass at 2.4 TB/s
        auto one = [&](const float4 v, float4 d, const float gp, float4 &A_, float4 &B_) {
            if (!BWD) {     // sum, sum of squares
                A_.x += v.x; A_.y += v.y; A_.z += v.z; A_.w += v.w;
                B_.x = fmaf(v.x, v.x, B_.x); B_.y = fmaf(v.y, v.y, B_.y); B_.z = fmaf(v.z, v.z, B_.z); B_.w = fmaf(v.w, v.w, B_.w);
            } else {        // s1 = sum dy * mask, s2 = sum dy * mask * xhat; mask = the ReLU passed (scale * z + shift > 0)
                if (gate) { d.x *= gp; d.y *= gp; d.z *= gp; d.w *= gp; }
                if (relu) {
                    if (!(fmaf(v.x, sc.x, sh.x) > 0.f)) d.x = 0.f;
                    if (!(fmaf(v.y, sc.y, sh.y) > 0.f)) d.y = 0.f;
                    if (!(fmaf(v.z, sc.z, sh.z) > 0.f)) d.z = 0.f;
                    if (!(fmaf(v.w, sc.w, sh.w) > 0.f)) d.w = 0.f;
                }
                A_.x += d.x; A_.y += d.y; A_.z += d.z; A_.w += d.w;
                B_.x = fmaf(d.x, (v.x - mu.x) * is.x, B_.x); B_.y = fmaf(d.y, (v.y - mu.y) * is.y, B_.y);
                B_.z = fmaf(d.z, (v.z - mu.z) * is.z, B_.z); B_.w = fmaf(d.w, (v.w - mu.w) * is.w, B_.w);
            }
        };
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 A1 = zero4, B1 = zero4, A2 = zero4, B2 = zero4, A3 = zero4, B3 = zero4;
        long long p = p0 + pl;
        for (; p + 3 * lanes < p1; p += 4 * lanes) {
            float4 v[4], d[4];
            float gp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *(const float4 *)(z + (p + u * lanes) * C + 4 * g);
                d[u] = BWD ? *(const float4 *)(dy + (p + u * lanes) * DC + 4 * g) : zero4;
                gp[u] = (BWD && gate) ? gate[p + u * lanes] : 1.f;
            }
            one(v[0], d[0], gp[0], A, B); one(v[1], d[1], gp[1], A1, B1); one(v[2], d[2], gp[2], A2, B2); one(v[3], d[3], gp[3], A3, B3);
        }
        for (; p < p1; p += lanes) {
            const float4 v = *(const float4 *)(z + p * C + 4 * g);
            const float4 d = BWD ? *(const float4 *)(dy + p * DC + 4 * g) : zero4;
            one(v, d, (BWD && gate) ? gate[p] : 1.f, A, B);
        }
        A.x += (A1.x + A2.x) + A3.x; A.y += (A1.y + A2.y) + A3.y; A.z += (A1.z + A2.z) + A3.z; A.w += (A1.w + A2.w) + A3.w;
        B.x += (B1.x + B2.x) + B3.x; B.y += (B1.y + B2.y) + B3.y; B.z += (B1.z + B2.z) + B3.z; B.w += (B1.w + B2.w) + B3.w;
    }
    s_a[threadIdx.x] = A; s_b[threadIdx.x] = B;
    __syncthreads();
    if (threadIdx.x < groups) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        for (int l = 0; l < lanes; ++l) {
            const float4 x = s_a[l * groups + g], y = s_b[l * groups + g];
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
            b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
        }
        float *row = ws + (size_t)blockIdx.x * 2 * C;
        *(float4 *)(row + 4 * g) = a;
        *(float4 *)(row + C + 4 * g) = b;
    }
}

// fwd: mean, biased variance and 1/sqrt(var + eps) per channel; bwd: s1, s2 per channel.  One wave per channel: lanes stride over
// the workgroup rows in double, then a fixed butterfly — deterministic.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ void __launch_bounds__(256) k_bn_finalize(const float *__restrict__ ws, int blocks, int C, double count, float eps, int bwd,
                                                     float *__restrict__ o0, float *__restrict__ o1, float *__restrict__ o2) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int i = lane; i < blocks; i += 64) { a += (double)ws[(size_t)i * 2 * C + c]; b += (double)ws[(size_t)i * 2 * C + C + c]; }
    a = wave_sum_f64(a); b = wave_sum_f64(b);
    if (lane != 0) return;
    if (bwd) { o0[c] = (float)a; o1[c] = (float)b; return; }
    const double m = a / count;
    double var = b / count - m * m;
    if (var < 0.0) var = 0.0;
    o0[c] = (float)m; o1[c] = (float)var; o2[c] = (float)(1.0 / sqrt(var + (double)eps));
}

// The same for C % 4 == 0 with four adjacent channels per WORKGROUP: thread t takes rows t, t + 256, ... as 16-byte reads (the one-channel
// form reads 4 of every 64-byte sector it touches and gives a channel one wave: with the 2 k rows of the finer slabs and the 9 k pixel
// tiles of a level-0 Winograd layer the pass was 24 us per call, 138 calls per training step), the four waves' sums meet in LDS in wave
// order: deterministic.
__global__ void __launch_bounds__(256) k_bn_finalize4(const float *__restrict__ ws, int blocks, int C, double count, float eps, int bwd,
                                                      float *__restrict__ o0, float *__restrict__ o1, float *__restrict__ o2) {
    __shared__ double s_part[4][8];
    const int c = blockIdx.x * 4, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < blocks; i += 256) {
        const float4 x = *(const float4 *)(ws + (size_t)i * 2 * C + c), y = *(const float4 *)(ws + (size_t)i * 2 * C + C + c);
        a[0] += (double)x.x; a[1] += (double)x.y; a[2] += (double)x.z; a[3] += (double)x.w;
        b[0] += (double)y.x; b[1] += (double)y.y; b[2] += (double)y.z; b[3] += (double)y.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = wave_sum_f64(a[k]); b[k] = wave_sum_f64(b[k]); }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_part[wid][k] = a[k]; s_part[wid][4 + k] = b[k]; }
    }
    __syncthreads();
    if (threadIdx.x >= 4) return;
    const int k = threadIdx.x;
    const double sa = ((s_part[0][k] + s_part[1][k]) + s_part[2][k]) + s_part[3][k];
    const double sb = ((s_part[0][4 + k] + s_part[1][4 + k]) + s_part[2][4 + k]) + s_part[3][4 + k];
    if (bwd) { o0[c + k] = (float)sa; o1[c + k] = (float)sb; return; }
    const double m = sa / count;
    double var = sb / count - m * m;
    if (var < 0.0) var = 0.0;
    o0[c + k] = (float)m; o1[c + k] = (float)var; o2[c + k] = (float)(1.0 / sqrt(var + (double)eps));
}

void launch_bn_finalize(const float *ws, int blocks, int C, double count, float eps, int bwd, float *o0, float *o1, float *o2, hipStream_t s) {
    if (C % 4 == 0) hipLaunchKernelGGL(k_bn_finalize4, dim3(C / 4), dim3(256), 0, s, ws, blocks, C, count, eps, bwd, o0, o1, o2);
    else hipLaunchKernelGGL(k_bn_finalize, dim3(hvpr_cdiv(C, 4)), dim3(256), 0, s, ws, blocks, C, count, eps, bwd, o0, o1, o2);
}

// What a train-mode nn.BatchNorm2d does with the batch moments besides normalising, in one launch instead of eight element-wise
// ones per layer: scale = gamma * invstd, shift = beta - mean * scale for the normalising kernel, and the running statistics
// running = (1 - momentum) * running + momentum * (mean | unbiased variance), num_batches_tracked += 1 (torch/nn/modules/batchnorm.py).
__global__ void __launch_bounds__(256) k_bn_affine(const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ invstd,
                                                   int C, const float *__restrict__ gamma, const float *__restrict__ beta, float momentum,
                                                   float momentum_unbiased, float *__restrict__ running_mean, float *__restrict__ running_var,
                                                   long long *__restrict__ tracked, float *__restrict__ scale, float *__restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && tracked) *tracked += 1;
    if (c >= C) return;
    const float sc = __fmul_rn(gamma[c], invstd[c]);
    scale[c] = sc;
    shift[c] = __fsub_rn(beta[c], __fmul_rn(mean[c], sc));
    if (running_mean) running_mean[c] = fmaf(momentum, mean[c], __fmul_rn(running_mean[c], 1.f - momentum));
    if (running_var) running_var[c] = fmaf(momentum_unbiased, var[c], __fmul_rn(running_var[c], 1.f - momentum));
}

// y = relu(z * scale + shift)   (scale = gamma * invstd, shift = beta - mean * scale);  with a gate: y = gate[p] * relu(..) + resid
// (the SFM step x_att = attention(sfm(x_att), y) + x_att, base_bev_backbone.py:250-255)
// (y_stride4 / y_off4: row pitch and channel offset of y in float4 — a channel slice of a wider tensor; y_stride4 == groups: y[i])
__global__ void __launch_bounds__(256) k_bn_apply(const float4 *__restrict__ z, long long n4, int groups, const float *__restrict__ scale,
                                                  const float *__restrict__ shift, int relu, const float *__restrict__ gate,
                                                  const float4 *__restrict__ resid, float4 *__restrict__ y, int y_stride4, int y_off4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        const float4 sc = *(const float4 *)(scale + 4 * g), sh = *(const float4 *)(shift + 4 * g);
        const float4 v = z[i];
        float4 r = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        if (gate) {
            const float gp = gate[i / groups];
            const float4 q = resid[i];
            r = make_float4(fmaf(gp, r.x, q.x), fmaf(gp, r.y, q.y), fmaf(gp, r.z, q.z), fmaf(gp, r.w, q.w));
        }
        y[y_stride4 == groups ? i : (i / groups) * y_stride4 + y_off4 + g] = r;
    }
}

// dz = scale * (dy_m - s1 / n - xhat * s2 / n),  dy_m = dy where the ReLU passed, else 0
__global__ void __launch_bounds__(256) k_bn_bwd_apply(const float4 *__restrict__ dy, const float4 *__restrict__ z, long long n4, int groups,
                                                      const float *__restrict__ scale, const float *__restrict__ shift,
                                                      const float *__restrict__ mean, const float *__restrict__ invstd,
                                                      const float *__restrict__ s1, const float *__restrict__ s2, float inv_n, int relu,
                                                      const float *__restrict__ gate, float *__restrict__ dgate /* zeroed, atomics */,
                                                      float4 *__restrict__ dz, int dy_stride4 /* row pitch of dy in float4; == groups: dy[i] */) {
    // grid-stride loop with trip counts that are uniform per wave (the stride is a multiple of 64 and of `groups`): the lanes of
    // one pixel stay together, so the per-pixel sum for dgate is a butterfly inside the wave
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n_iter = (n4 + stride - 1) / stride;
    for (long long it = 0; it < n_iter; ++it) {
        const long long i = it * stride + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = i < n4;
        float ga = 0.f;
        if (live) {
        const int g = (int)(i % groups);
        const float4 sc = *(const float4 *)(scale + 4 * g), sh = *(const float4 *)(shift + 4 * g);
        const float4 mu = *(const float4 *)(mean + 4 * g), is = *(const float4 *)(invstd + 4 * g);
        const float4 a1 = *(const float4 *)(s1 + 4 * g), a2 = *(const float4 *)(s2 + 4 * g);
        const float4 v = z[i];
        float4 d = dy[dy_stride4 == groups ? i : (i / groups) * dy_stride4 + g];
        if (gate) {          // y = gate * a + resid: d a = gate * dy, d gate = sum_c a * dy
            const float4 a = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
            ga = (relu ? fmaxf(a.x, 0.f) : a.x) * d.x + (relu ? fmaxf(a.y, 0.f) : a.y) * d.y + (relu ? fmaxf(a.z, 0.f) : a.z) * d.z +
                 (relu ? fmaxf(a.w, 0.f) : a.w) * d.w;
            const float gp = gate[i / groups];
            d.x *= gp; d.y *= gp; d.z *= gp; d.w *= gp;
        }
        if (relu) {
            if (!(fmaf(v.x, sc.x, sh.x) > 0.f)) d.x = 0.f;
            if (!(fmaf(v.y, sc.y, sh.y) > 0.f)) d.y = 0.f;
            if (!(fmaf(v.z, sc.z, sh.z) > 0.f)) d.z = 0.f;
            if (!(fmaf(v.w, sc.w, sh.w) > 0.f)) d.w = 0.f;
        }
        float4 r;
        r.x = sc.x * (d.x - a1.x * inv_n - (v.x - mu.x) * is.x * (a2.x * inv_n));
        r.y = sc.y * (d.y - a1.y * inv_n - (v.y - mu.y) * is.y * (a2.y * inv_n));
        r.z = sc.z * (d.z - a1.z * inv_n - (v.z - mu.z) * is.z * (a2.z * inv_n));
        r.w = sc.w * (d.w - a1.w * inv_n - (v.w - mu.w) * is.w * (a2.w * inv_n));
        dz[i] = r;
        }
        if (gate) {          // wave-uniform branch
            // sum over the lanes of one pixel: `groups` consecutive lanes (a power of two: <= 64 inside the wave; wider pixels span
            // several waves and meet in the atomic)
            if ((groups & (groups - 1)) == 0) {
                const int w = groups < 64 ? groups : 64;
                for (int o = w >> 1; o > 0; o >>= 1) ga += __shfl_xor(ga, o, 64);
                if (live && ((threadIdx.x & 63) & (w - 1)) == 0) atomicAdd(dgate + i / groups, ga);
            } else if (live) {       // C / 4 not a power of two (48, 96, ... channels): a pixel's lanes straddle waves unevenly
                atomicAdd(dgate + i / groups, ga);
            }
        }
    }
}

int bn_blocks(long long P) { const int slab = bn_slab(P); return (int)((P + slab - 1) / slab); }

__global__ void __launch_bounds__(256) k_zero_p(float *__restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.f;
}

}  // namespace

extern "C" size_t hvpr_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int taps, int stride) {
    if (N < 1 || OH < 1 || OW < 1 || Cin < 1 || Cout < 1 || (taps != 9 && taps != 1)) return 0;
    return (size_t)wgrad_chunks(N, OH, OW, Cin, Cout, taps, stride) * taps * Cout * Cin * sizeof(float);
}

extern "C" int hvpr_conv2d_wgrad_nhwc_f32(const float *x, int N, int H, int W, int Cin, const float *dz, int Cout, int taps, int stride,
                                          float *dw, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!x || !dz || !dw || !workspace || N < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return HVPR_ERR_INVALID_ARG;
    if ((taps != 9 && taps != 1) || (stride != 1 && stride != 2) || (taps == 1 && stride != 1)) return HVPR_ERR_UNSUPPORTED;
    if (Cin % 4 != 0 || Cout % 4 != 0) return HVPR_ERR_UNSUPPORTED;
    WgradArgs a;
    a.x = x; a.dz = dz; a.part = (float *)workspace;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.OH = taps == 9 ? (H + 2 - 3) / stride + 1 : H;
    a.OW = taps == 9 ? (W + 2 - 3) / stride + 1 : W;
    a.n_chunks = wgrad_chunks(N, a.OH, a.OW, Cin, Cout, taps, stride);
    if (workspace_bytes < hvpr_conv2d_wgrad_workspace_bytes(N, a.OH, a.OW, Cin, Cout, taps, stride)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (taps == 9 && stride == 1) launch_wgrad<9, 1, 8, 8, 1, 1>(a, 512, s);
    else if (taps == 9) launch_wgrad<9, 2, 4, 8, 1, 1>(a, 512, s);
    else if (wgrad_few_rows(Cin, Cout, taps)) launch_wgrad<1, 1, 4, 8, 1, 3, 1>(a, 512, s);
    else launch_wgrad<1, 1, 4, 8, 2, 2>(a, 512, s);
    const long long per = (long long)taps * Cout * Cin;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(hvpr_cdiv(per, 64)), dim3(256), 0, s, a.part, a.n_chunks, taps, Cout, Cin, dw);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" size_t hvpr_bn_workspace_bytes(long long P, int C) {
    if (P < 1 || C < 4) return 0;
    return (size_t)bn_blocks(P) * 2 * C * sizeof(float);
}

extern "C" int hvpr_bn_stats_nhwc_f32(const float *z, long long P, int C, float eps, float *mean, float *var, float *invstd,
                                      void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!z || !mean || !var || !invstd || !workspace || P < 1) return HVPR_ERR_INVALID_ARG;
    if (C < 4 || C % 4 != 0 || C > 1024) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < hvpr_bn_workspace_bytes(P, C)) return HVPR_ERR_WORKSPACE;
    const int blocks = bn_blocks(P);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_reduce<false>, dim3(blocks), dim3(256), 0, s, z, nullptr, P, C, nullptr, nullptr, nullptr, nullptr, 0, nullptr, bn_slab(P),
                       (float *)workspace);
    launch_bn_finalize((const float *)workspace, blocks, C, (double)P, eps, 0, mean, var, invstd, s);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_bn_finalize_partials_f32(const float *partials, int rows, int C, long long count, float eps, float *mean, float *var,
                                             float *invstd, hvpr_stream_t stream) {
    if (!partials || !mean || !var || !invstd || rows < 1 || count < 1) return HVPR_ERR_INVALID_ARG;
    if (C < 4 || C > 1024) return HVPR_ERR_UNSUPPORTED;
    launch_bn_finalize(partials, rows, C, (double)count, eps, 0, mean, var, invstd, (hipStream_t)stream);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_bn_train_affine_f32(const float *mean, const float *var, const float *invstd, int C, const float *gamma, const float *beta,
                                        float momentum, float momentum_unbiased, float *running_mean, float *running_var,
                                        long long *num_batches_tracked, float *scale, float *shift, hvpr_stream_t stream) {
    if (!mean || !var || !invstd || !gamma || !beta || !scale || !shift || C < 1) return HVPR_ERR_INVALID_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return HVPR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_bn_affine, dim3(hvpr_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, var, invstd, C, gamma, beta, momentum,
                       momentum_unbiased, running_mean, running_var, num_batches_tracked, scale, shift);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_bn_relu_fwd_nhwc_f32(const float *z, long long P, int C, const float *scale, const float *shift, int relu,
                                         const float *gate, const float *resid, float *y, hvpr_stream_t stream) {
    if (!z || !scale || !shift || !y || P < 1 || ((gate == nullptr) != (resid == nullptr))) return HVPR_ERR_INVALID_ARG;
    if (C < 4 || C % 4 != 0) return HVPR_ERR_UNSUPPORTED;
    const long long n4 = P * (C / 4);
    long long blocks = (n4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)z, n4, C / 4, scale, shift, relu,
                       gate, (const float4 *)resid, (float4 *)y, C / 4, 0);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

// The same into / out of a channel slice of a wider NHWC tensor (the deconvolution branches write their 128 channels straight into the
// 384-channel concatenation the head reads, and take their gradient out of its gradient: no torch.cat, no slice copies).
extern "C" int hvpr_bn_relu_fwd_slice_nhwc_f32(const float *z, long long P, int C, const float *scale, const float *shift, int relu, float *y,
                                               int y_cstride, int y_coff, hvpr_stream_t stream) {
    if (!z || !scale || !shift || !y || P < 1) return HVPR_ERR_INVALID_ARG;
    if (C < 4 || C % 4 != 0 || y_cstride % 4 != 0 || y_coff % 4 != 0 || y_coff < 0 || y_coff + C > y_cstride) return HVPR_ERR_UNSUPPORTED;
    const long long n4 = P * (C / 4);
    long long blocks = (n4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)z, n4, C / 4, scale, shift, relu,
                       (const float *)nullptr, (const float4 *)nullptr, (float4 *)y, y_cstride / 4, y_coff / 4);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

// The backward in its two halves, so that a caller can put an all-reduce between them (SyncBatchNorm: the sums and the count are
// those of the GLOBAL batch, tools/train.py:119-120): sums = local d beta / d gamma; apply = dz from whatever sums it is given.
extern "C" int hvpr_bn_relu_bwd_sums_nhwc_f32(const float *dy, const float *z, long long P, int C, const float *scale, const float *shift,
                                              const float *mean, const float *invstd, int relu, const float *gate, float *dgamma,
                                              float *dbeta, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!dy || !z || !scale || !shift || !mean || !invstd || !dgamma || !dbeta || !workspace || P < 1) return HVPR_ERR_INVALID_ARG;
    if (C < 4 || C % 4 != 0 || C > 1024) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < hvpr_bn_workspace_bytes(P, C)) return HVPR_ERR_WORKSPACE;
    const int blocks = bn_blocks(P);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_reduce<true>, dim3(blocks), dim3(256), 0, s, z, dy, P, C, scale, shift, mean, invstd, relu, gate, bn_slab(P), (float *)workspace);
    // s1 -> dbeta, s2 -> dgamma  (d beta = sum dy_m, d gamma = sum dy_m * xhat)
    launch_bn_finalize((const float *)workspace, blocks, C, (double)P, 0.f, 1, dbeta, dgamma, (float *)nullptr, s);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_bn_relu_bwd_apply_nhwc_f32(const float *dy, const float *z, long long P, int C, const float *scale, const float *shift,
                                               const float *mean, const float *invstd, int relu, const float *gate, float *dgate, float *dz,
                                               const float *dgamma_total, const float *dbeta_total, double inv_count, hvpr_stream_t stream) {
    if (!dy || !z || !scale || !shift || !mean || !invstd || !dz || !dgamma_total || !dbeta_total || P < 1 || !(inv_count > 0.0))
        return HVPR_ERR_INVALID_ARG;
    if ((gate == nullptr) != (dgate == nullptr)) return HVPR_ERR_INVALID_ARG;
    if (C < 4 || C % 4 != 0 || C > 1024) return HVPR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dgate) hipLaunchKernelGGL(k_zero_p, dim3(hvpr_cdiv(P, 256) > 8192 ? 8192 : hvpr_cdiv(P, 256)), dim3(256), 0, s, dgate, P);
    const long long n4 = P * (C / 4);
    long long g = (n4 + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)g), dim3(256), 0, s, (const float4 *)dy, (const float4 *)z, n4, C / 4, scale, shift, mean,
                       invstd, dbeta_total, dgamma_total, (float)inv_count, relu, gate, dgate, (float4 *)dz, C / 4);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_bn_relu_bwd_slice_nhwc_f32(const float *dy, int dy_cstride, int dy_coff, const float *z, long long P, int C,
                                               const float *scale, const float *shift, const float *mean, const float *invstd, int relu,
                                               float *dz, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes,
                                               hvpr_stream_t stream) {
    if (!dy || !z || !scale || !shift || !mean || !invstd || !dz || !dgamma || !dbeta || !workspace || P < 1) return HVPR_ERR_INVALID_ARG;
    if (C < 4 || C % 4 != 0 || C > 1024 || dy_cstride % 4 != 0 || dy_coff % 4 != 0 || dy_coff < 0 || dy_coff + C > dy_cstride) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < hvpr_bn_workspace_bytes(P, C)) return HVPR_ERR_WORKSPACE;
    const int blocks = bn_blocks(P);
    hipStream_t s = (hipStream_t)stream;
    const float *dys = dy + dy_coff;
    hipLaunchKernelGGL(k_bn_reduce<true>, dim3(blocks), dim3(256), 0, s, z, dys, P, C, scale, shift, mean, invstd, relu, (const float *)nullptr,
                       bn_slab(P), (float *)workspace, dy_cstride);
    launch_bn_finalize((const float *)workspace, blocks, C, (double)P, 0.f, 1, dbeta, dgamma, (float *)nullptr, s);
    const long long n4 = P * (C / 4);
    long long g = (n4 + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)g), dim3(256), 0, s, (const float4 *)dys, (const float4 *)z, n4, C / 4, scale, shift, mean,
                       invstd, dbeta, dgamma, (float)(1.0 / (double)P), relu, (const float *)nullptr, (float *)nullptr, (float4 *)dz,
                       dy_cstride / 4);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_bn_relu_bwd_nhwc_f32(const float *dy, const float *z, long long P, int C, const float *scale, const float *shift,
                                         const float *mean, const float *invstd, int relu, const float *gate, float *dgate, float *dz,
                                         float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!dz || (gate == nullptr) != (dgate == nullptr)) return HVPR_ERR_INVALID_ARG;
    const int st = hvpr_bn_relu_bwd_sums_nhwc_f32(dy, z, P, C, scale, shift, mean, invstd, relu, gate, dgamma, dbeta, workspace,
                                                  workspace_bytes, stream);
    if (st != HVPR_OK) return st;
    return hvpr_bn_relu_bwd_apply_nhwc_f32(dy, z, P, C, scale, shift, mean, invstd, relu, gate, dgate, dz, dgamma, dbeta, 1.0 / (double)P,
                                           stream);
}
