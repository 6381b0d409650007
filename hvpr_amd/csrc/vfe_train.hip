// a2 (training) — the two PFN layers of PillarVFE_Scale with batch-statistics BatchNorm, forward and backward, without
// materialising any (M, 32, C) tensor.  Replaces, in training mode, the PFNLayer chain of PillarVFE_Scale.forward,
// pcdet/models/backbones_3d/vfe/pillar_vfe.py:184-221 (PFNLayer.forward :29-49: linear -> BatchNorm1d over all M*P slots ->
// ReLU -> max over the slots -> concat), and autograd's backward of it (weights, gamma, beta of both layers; the voxels carry no
// gradient).  The scale stream (5 -> 16 -> 32 on M rows) stays with the caller.
//
//   x0[p]  = [voxel[p], xyz - mean, xyz - centre] * (p < n)                      (10)      padded slots: 0
//   y0     = x0 W0^T (16)   a0 = gamma0 (y0 - mu0) inv0 + beta0   z0 = relu(a0)   m0 = max_p z0
//   x1[p]  = [z0[p], m0]                                                          (32)
//   y1     = x1 W1^T (64)   a1 = ...                              z1 = relu(a1)   out = max_p z1
// mu / var are over ALL M * 32 slots (padded ones included, as the reference's BatchNorm1d sees them — SURVEY.md B.5).
//
// Everything is recomputed from the voxels in every pass (62 k pillars x 32 slots x 16 B = 127 MB per pass at batch 16): three
// forward passes (statistics of y0, statistics of y1, output) and two backward passes, each followed by a one-workgroup kernel
// that finishes the per-channel sums in double.  Mapping: ONE PILLAR PER HALF-WAVE, lane = slot; the weights are wave-uniform
// (scalar loads).  BatchNorm's backward is dense (every slot receives -(dbeta + xhat dgamma) / N) — it is carried by global
// moments instead of per-slot work:
//   g_y1 = c1 (g_a1 - dbeta1/N - xhat1 dgamma1/N),  c1 = gamma1 inv1,  g_a1 = dOut at the arg-max slot where out > 0, else 0
//   dW1  = c1 sum g_a1 x1^T  -  (c1/N) (dbeta1 s1^T + dgamma1 inv1 (W1 S1 - mu1 s1^T)),     s1 = sum x1, S1 = sum x1 x1^T
//   g_x1[p] = sum_c c1 g_a1[p][c] W1[c]  -  (Q x1[p] + w),   Q = W1^T diag(c1 inv1 dgamma1 / N) W1,
//                                                            w = W1^T (c1 (dbeta1 - mu1 inv1 dgamma1) / N)
// and the same one level down (dW0 from s0 = sum x0, S0 = sum x0 x0^T).  Arg-max ties take the lowest slot (torch's max).
#include "common.h"

namespace {

constexpr int P = 32, C0 = 16, CIN = 10, C1 = 64, K1 = 32;
constexpr int kWaves = 4, kThreads = kWaves * 64, kBlocks = 256;

// per-workgroup partial rows of the passes (floats): the eight half-waves of a workgroup add theirs in a fixed order
constexpr int kF1 = 2 * C0;                               // sum y0, sum y0^2
constexpr int kF2 = 2 * C1;                               // sum y1, sum y1^2
constexpr int kB1 = 2 * C1 + C1 * K1 + K1 + K1 * K1;      // dbeta1, dgamma1, sum g x1^T, s1, S1
constexpr int kB2 = 2 * C0 + C0 * CIN + CIN + CIN * CIN;  // dbeta0, dgamma0, sum g x0^T, s0, S0
constexpr int kMaxRow = kB1;

// device-side scratch after the partial rows (floats): folded BN of both layers, statistics, Q, w, finished layer-1 sums
struct Scratch {
    float sc0[C0], sh0[C0], mu0[C0], inv0[C0];
    float sc1[C1], sh1[C1], mu1[C1], inv1[C1];
    float q[K1 * K1], w[K1];
    float dbeta1[C1], dgamma1[C1];
};

struct Geom { float vsx, vsy, vsz, ox, oy, oz; };

struct Rows {           // what a lane (slot p of its half-wave's pillar) knows after the shared part of every pass
    float x0[CIN];
    float y0[C0];
    int n;
    bool act;           // the half-wave has a pillar
    bool ex;            // this lane's slot exists: p < PS, the configured points per pillar (<= P = 32 lanes)
};

__device__ __forceinline__ void load_rows(const float4 *__restrict__ voxels, const int *__restrict__ num, const int4 *__restrict__ coords,
                                          long long m, long long M, int p, int PS, const Geom &g, const float *__restrict__ w0, Rows &r) {
    r.act = m < M;
    r.ex = p < PS;
    const long long mm = r.act ? m : 0;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r.act && r.ex) v = voxels[mm * PS + p];
    r.n = r.act ? num[mm] : 0;
    const int4 c = coords[mm];                                  // [b, z, y, x]
    const float inv_n = 1.f / (float)(r.n > 0 ? r.n : 1);
    const float mx = hvpr_reduce_sum<32>(v.x) * inv_n, my = hvpr_reduce_sum<32>(v.y) * inv_n, mz = hvpr_reduce_sum<32>(v.z) * inv_n;
    const float cx = fmaf((float)c.w, g.vsx, g.ox), cy = fmaf((float)c.z, g.vsy, g.oy), cz = fmaf((float)c.y, g.vsz, g.oz);
    const float k = p < r.n ? 1.f : 0.f;
    r.x0[0] = v.x * k; r.x0[1] = v.y * k; r.x0[2] = v.z * k; r.x0[3] = v.w * k;
    r.x0[4] = (v.x - mx) * k; r.x0[5] = (v.y - my) * k; r.x0[6] = (v.z - mz) * k;
    r.x0[7] = (v.x - cx) * k; r.x0[8] = (v.y - cy) * k; r.x0[9] = (v.z - cz) * k;
#pragma unroll
    for (int c0 = 0; c0 < C0; ++c0) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CIN; ++j) s = fmaf(r.x0[j], w0[c0 * CIN + j], s);
        r.y0[c0] = s;
    }
}

// z0 (16) and m0 (16) of the lane's slot / pillar from y0 and the folded BatchNorm
__device__ __forceinline__ void layer0(const Rows &r, const Scratch *__restrict__ s, float z0[C0], float m0[C0]) {
#pragma unroll
    for (int c = 0; c < C0; ++c) {
        z0[c] = fmaxf(fmaf(r.y0[c], s->sc0[c], s->sh0[c]), 0.f);
        m0[c] = hvpr_reduce_max<32>(r.ex ? z0[c] : -INFINITY);      // lanes past PS are not slots (padded slots p >= n are)
    }
}

// y1[c] of the lane's slot for 16 consecutive channels
__device__ __forceinline__ void layer1_chunk(const float z0[C0], const float m0[C0], const float *__restrict__ w1, int c_base, float y1[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float *wr = w1 + (c_base + i) * K1;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < C0; ++j) s = fmaf(z0[j], wr[j], s);
#pragma unroll
        for (int j = 0; j < C0; ++j) s = fmaf(m0[j], wr[C0 + j], s);
        y1[i] = s;
    }
}

// lowest lane of this lane's half-wave for which `hit` holds (some lane of the half must hit)
__device__ __forceinline__ int first_in_half(bool hit, int half) {
    const unsigned long long b = __ballot(hit);
    const unsigned hb = (unsigned)(half ? (b >> 32) : (b & 0xffffffffull));
    return __ffs((int)hb) - 1;
}


// The eight half-waves of the workgroup add their partial sums into one LDS row, one after the other (fixed order), and the row
// goes to part[block].  `emit(f)` calls f(index, value) for every entry this lane holds.
template <int K, typename Emit>
__device__ __forceinline__ void wg_reduce_store(float *s_row, float *__restrict__ part, int wv, int half, Emit emit) {
    for (int i = threadIdx.x; i < K; i += kThreads) s_row[i] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < kWaves * 2; ++t) {
        if (wv * 2 + half == t) emit([&](int idx, float v) { s_row[idx] += v; });
        __syncthreads();
    }
    for (int i = threadIdx.x; i < K; i += kThreads) part[(size_t)blockIdx.x * kMaxRow + i] = s_row[i];
}

// sums[i] = sum over the workgroup rows, in row order, in double; sums[-1] = the element count; local[i] = a copy of the first
// n_local sums (what stays per-rank when the SyncBatchNorm hook replaces the front of `sums` by the sums over all ranks)
__global__ void __launch_bounds__(256) k_vfe_reduce_rows(const float *__restrict__ part, int K, double count, double *__restrict__ sums,
                                                         int n_local, double *__restrict__ local) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) sums[-1] = count;
    if (i >= K) return;
    double a = 0.0;
    for (int h = 0; h < kBlocks; ++h) a += (double)part[(size_t)h * kMaxRow + i];
    sums[i] = a;
    if (i < n_local) local[i] = a;
}

// MODE 0: statistics of y0.  MODE 1: statistics of y1.  MODE 2: out [M, 64].
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_vfe_train_fwd(const float4 *__restrict__ voxels, const int *__restrict__ num,
                                                            const int4 *__restrict__ coords, long long M, int PS, Geom g,
                                                            const float *__restrict__ w0, const float *__restrict__ w1,
                                                            const Scratch *__restrict__ sc, float *__restrict__ part, float *__restrict__ out) {
    __shared__ float s_row[MODE == 2 ? 1 : kF2];
    const int lane = threadIdx.x & 63, p = lane & 31, half = lane >> 5;
    float acc[MODE == 0 ? kF1 : (MODE == 1 ? kF2 : 1)];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(acc) / sizeof(float)); ++i) acc[i] = 0.f;
    const long long n_pairs = (M + 1) / 2;
    for (long long pair = (long long)blockIdx.x * kWaves + (threadIdx.x >> 6); pair < n_pairs; pair += (long long)kBlocks * kWaves) {
        const long long m = pair * 2 + half;
        Rows r;
        load_rows(voxels, num, coords, m, M, p, PS, g, w0, r);
        if (MODE == 0) {
            if (r.act) {
#pragma unroll
                for (int c = 0; c < C0; ++c) { acc[c] += r.y0[c]; acc[C0 + c] = fmaf(r.y0[c], r.y0[c], acc[C0 + c]); }
            }
            continue;
        }
        float z0[C0], m0[C0];
        layer0(r, sc, z0, m0);
#pragma unroll
        for (int cb = 0; cb < C1; cb += 16) {
            float y1[16];
            layer1_chunk(z0, m0, w1, cb, y1);
            if (MODE == 1) {
                if (r.act && r.ex) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { acc[cb + i] += y1[i]; acc[C1 + cb + i] = fmaf(y1[i], y1[i], acc[C1 + cb + i]); }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float z = fmaxf(fmaf(y1[i], sc->sc1[cb + i], sc->sh1[cb + i]), 0.f);
                    const float mx = hvpr_reduce_max<32>(r.ex ? z : -INFINITY);
                    if (r.act && p == ((cb + i) & 31)) out[m * C1 + cb + i] = mx;
                }
            }
        }
    }
    if (MODE != 2) {
        constexpr int K = MODE == 0 ? kF1 : kF2;
#pragma unroll
        for (int i = 0; i < K; ++i) acc[i] = hvpr_reduce_sum<32>(acc[i]);          // every lane of the half holds the sum
        wg_reduce_store<K>(s_row, part, threadIdx.x >> 6, half, [&](auto add) {
#pragma unroll
            for (int i = 0; i < K; ++i)
                if (p == (i & 31)) add(i, acc[i]);
        });
    }
}

// one workgroup: per-channel mean / biased variance / inv-std of a layer from the partial rows; folded scale / shift
__global__ void __launch_bounds__(256) k_vfe_fin_stats(const double *__restrict__ sums, int C, float eps,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, int layer,
                                                       Scratch *__restrict__ s, float *__restrict__ mean_out, float *__restrict__ var_out) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const double count = sums[-1];          // (of the global batch after the hook)
    const double a = sums[c], b = sums[C + c];
    const double mu = a / count;
    double var = b / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * inv, shift = beta[c] - (float)mu * scale;
    if (layer == 0) { s->sc0[c] = scale; s->sh0[c] = shift; s->mu0[c] = (float)mu; s->inv0[c] = inv; }
    else { s->sc1[c] = scale; s->sh1[c] = shift; s->mu1[c] = (float)mu; s->inv1[c] = inv; }
    mean_out[c] = (float)mu; var_out[c] = (float)var;
}

// backward, layer 1: sums for dbeta1 / dgamma1, the sparse part of dW1 and the moments s1, S1
__global__ void __launch_bounds__(kThreads) k_vfe_train_bwd1(const float4 *__restrict__ voxels, const int *__restrict__ num,
                                                             const int4 *__restrict__ coords, long long M, int PS, Geom g,
                                                             const float *__restrict__ w0, const float *__restrict__ w1,
                                                             const Scratch *__restrict__ sc, const float *__restrict__ d_out,
                                                             float *__restrict__ part) {
    __shared__ float s_x1[kWaves][2][P + 1][K1];            // the pillar's x1 rows; row P = the padded-slot row
    __shared__ float s_row[kB1];
    const int lane = threadIdx.x & 63, p = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
    float a_db[2] = {0.f, 0.f}, a_dg[2] = {0.f, 0.f};       // channel c lives in lane c & 31, slot c >> 5
    float a_w[C1];                                          // sum g x1[.][p] for column k = p
    float a_s = 0.f, a_S[K1];                               // s1[p], S1[p][.]
#pragma unroll
    for (int i = 0; i < C1; ++i) a_w[i] = 0.f;
#pragma unroll
    for (int i = 0; i < K1; ++i) a_S[i] = 0.f;
    float (*x1)[K1] = s_x1[wv][half];
    const long long n_pairs = (M + 1) / 2;
    for (long long pair = (long long)blockIdx.x * kWaves + wv; pair < n_pairs; pair += (long long)kBlocks * kWaves) {
        const long long m = pair * 2 + half;
        Rows r;
        load_rows(voxels, num, coords, m, M, p, PS, g, w0, r);
        float z0[C0], m0[C0];
        layer0(r, sc, z0, m0);
        // x1 rows of the pillar to LDS (a padded slot's row is the same for every padded slot: written once more as row P)
#pragma unroll
        for (int j = 0; j < C0; ++j) { x1[p][j] = z0[j]; x1[p][C0 + j] = m0[j]; }
        if (p == (r.n < PS ? r.n : 0)) {
#pragma unroll
            for (int j = 0; j < C0; ++j) { x1[P][j] = z0[j]; x1[P][C0 + j] = m0[j]; }
        }
        const float wpad = r.act ? (float)(PS - r.n) : 0.f;    // multiplicity of the padded row
        // moments over the live rows + the padded row (wave-uniform trip count: the larger of the two pillars)
        const int n_loop = max(__shfl(r.n, 0, 64), __shfl(r.n, 32, 64));
        for (int q = 0; q <= n_loop; ++q) {
            const bool padrow = q == n_loop;
            const int row = padrow ? P : q;
            const float wt = padrow ? wpad : ((r.act && q < r.n) ? 1.f : 0.f);
            const float xi = x1[row][p] * wt;
            a_s += xi;
#pragma unroll
            for (int j = 0; j < K1; ++j) a_S[j] = fmaf(xi, x1[row][j], a_S[j]);
        }
#pragma unroll
        for (int cb = 0; cb < C1; cb += 16) {
            float y1[16];
            layer1_chunk(z0, m0, w1, cb, y1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = cb + i;
                const float z = r.ex ? fmaxf(fmaf(y1[i], sc->sc1[c], sc->sh1[c]), 0.f) : -INFINITY;
                const float mx = hvpr_reduce_max<32>(z);
                const int am = first_in_half(z == mx, half);
                const float y_at = __shfl(y1[i], (half << 5) + am, 64);
                const float gc = (r.act && mx > 0.f) ? d_out[m * C1 + c] : 0.f;      // uniform in the half-wave
                if (p == (c & 31)) {
                    a_db[c >> 5] += gc;
                    a_dg[c >> 5] = fmaf(gc, (y_at - sc->mu1[c]) * sc->inv1[c], a_dg[c >> 5]);
                }
                a_w[c] = fmaf(gc, x1[am][p], a_w[c]);
            }
        }
    }
    wg_reduce_store<kB1>(s_row, part, wv, half, [&](auto add) {
        add(p, a_db[0]); add(32 + p, a_db[1]); add(C1 + p, a_dg[0]); add(C1 + 32 + p, a_dg[1]);
#pragma unroll
        for (int c = 0; c < C1; ++c) add(2 * C1 + c * K1 + p, a_w[c]);
        add(2 * C1 + C1 * K1 + p, a_s);
#pragma unroll
        for (int j = 0; j < K1; ++j) add(2 * C1 + C1 * K1 + K1 + p * K1 + j, a_S[j]);
    });
}

// one workgroup: finish layer 1 — dbeta1, dgamma1, dW1 and the Q, w of the dense part of g_x1
// (sh[-1] = count, db / dg: of the global batch when the SyncBatchNorm hook ran; gw, s1, S1 and `local` (db, dg): this rank's —
// the dense term of dW1 multiplies the GLOBAL dbeta / dgamma / N with the moments of the LOCAL slots, the parameter gradients
// dbeta1 / dgamma1 are the local sums)
__global__ void __launch_bounds__(256) k_vfe_fin_bwd1(const double *__restrict__ sh, const double *__restrict__ local, const float *__restrict__ w1,
                                                      const float *__restrict__ gamma1, Scratch *__restrict__ s,
                                                      float *__restrict__ dw1, float *__restrict__ dgamma1, float *__restrict__ dbeta1) {
    __shared__ double coef_q[C1], coef_w[C1];
    const double count = sh[-1];
    const double *db = sh, *dg = sh + C1, *gw = sh + 2 * C1, *s1 = sh + 2 * C1 + C1 * K1, *S1 = s1 + K1;
    if (threadIdx.x < C1) {
        const int c = threadIdx.x;
        const double c1 = (double)gamma1[c] * (double)s->inv1[c];
        dbeta1[c] = (float)local[c]; dgamma1[c] = (float)local[C1 + c];
        s->dbeta1[c] = (float)db[c]; s->dgamma1[c] = (float)dg[c];
        coef_q[c] = c1 * (double)s->inv1[c] * dg[c] / count;
        coef_w[c] = c1 * (db[c] - (double)s->mu1[c] * (double)s->inv1[c] * dg[c]) / count;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C1 * K1; i += 256) {                       // dW1[c][k]
        const int c = i / K1, k = i % K1;
        const double c1 = (double)gamma1[c] * (double)s->inv1[c];
        double ws = 0.0;                                                     // (W1 S1)[c][k]
        for (int j = 0; j < K1; ++j) ws += (double)w1[c * K1 + j] * S1[j * K1 + k];
        const double dense = (c1 / count) * (db[c] * s1[k] + dg[c] * (double)s->inv1[c] * (ws - (double)s->mu1[c] * s1[k]));
        dw1[i] = (float)(c1 * gw[i] - dense);
    }
    for (int i = threadIdx.x; i < K1 * K1; i += 256) {                       // Q[k][j]
        const int k = i / K1, j = i % K1;
        double a = 0.0;
        for (int c = 0; c < C1; ++c) a += (double)w1[c * K1 + k] * coef_q[c] * (double)w1[c * K1 + j];
        s->q[i] = (float)a;
    }
    if (threadIdx.x < K1) {
        double a = 0.0;
        for (int c = 0; c < C1; ++c) a += (double)w1[c * K1 + threadIdx.x] * coef_w[c];
        s->w[threadIdx.x] = (float)a;
    }
}

// backward, layer 0: g_x1 of every slot, through the max / ReLU of layer 0; sums for dbeta0 / dgamma0, sparse dW0, s0, S0
__global__ void __launch_bounds__(kThreads) k_vfe_train_bwd0(const float4 *__restrict__ voxels, const int *__restrict__ num,
                                                             const int4 *__restrict__ coords, long long M, int PS, Geom g,
                                                             const float *__restrict__ w0, const float *__restrict__ w1,
                                                             const float *__restrict__ gamma1, const Scratch *__restrict__ sc,
                                                             const float *__restrict__ d_out, float *__restrict__ part) {
    __shared__ float s_g[kWaves][2][P][C0];                 // g_a0 rows
    __shared__ float s_x[kWaves][2][P][CIN + 2];            // x0 rows (padded to 12)
    __shared__ float s_row[kB2];
    const int lane = threadIdx.x & 63, p = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
    float a_db[C0], a_dg[C0];                               // per-slot partial sums, reduced over the half-wave at the end
#pragma unroll
    for (int c = 0; c < C0; ++c) { a_db[c] = 0.f; a_dg[c] = 0.f; }
    // lane-distributed entries: e = p, p + 32, ... over [g x0^T (160) | s0 (10) | S0 (100)] = 270 -> 9 per lane of the half
    constexpr int NE = C0 * CIN + CIN + CIN * CIN, EPL = (NE + P - 1) / P;
    float a_e[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) a_e[i] = 0.f;
    const long long n_pairs = (M + 1) / 2;
    for (long long pair = (long long)blockIdx.x * kWaves + wv; pair < n_pairs; pair += (long long)kBlocks * kWaves) {
        const long long m = pair * 2 + half;
        Rows r;
        load_rows(voxels, num, coords, m, M, p, PS, g, w0, r);
        float z0[C0], m0[C0];
        layer0(r, sc, z0, m0);
        // dense part of g_x1: -(Q x1[p] + w)
        float gx[K1];
#pragma unroll
        for (int k = 0; k < K1; ++k) {
            float a = sc->w[k];
#pragma unroll
            for (int j = 0; j < C0; ++j) a = fmaf(sc->q[k * K1 + j], z0[j], a);
#pragma unroll
            for (int j = 0; j < C0; ++j) a = fmaf(sc->q[k * K1 + C0 + j], m0[j], a);
            gx[k] = r.ex ? -a : 0.f;              // lanes past PS are not slots: nothing flows through them
        }
        // sparse part: + c1 g_c W1[c] on the arg-max slot of every channel
#pragma unroll
        for (int cb = 0; cb < C1; cb += 16) {
            float y1[16];
            layer1_chunk(z0, m0, w1, cb, y1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = cb + i;
                const float z = r.ex ? fmaxf(fmaf(y1[i], sc->sc1[c], sc->sh1[c]), 0.f) : -INFINITY;
                const float mx = hvpr_reduce_max<32>(z);
                const int am = first_in_half(z == mx, half);
                const float gc = mx > 0.f ? d_out[(r.act ? m : 0) * C1 + c] * gamma1[c] * sc->inv1[c] : 0.f;
                if (p == am) {
#pragma unroll
                    for (int k = 0; k < K1; ++k) gx[k] = fmaf(gc, w1[c * K1 + k], gx[k]);
                }
            }
        }
        // through x1 = [z0, m0]: the m0 half sums over the slots and lands on the arg-max slot of z0; then the ReLU of layer 0
        float ga[C0];
#pragma unroll
        for (int c = 0; c < C0; ++c) {
            const float gm = hvpr_reduce_sum<32>(gx[C0 + c]);
            const int am0 = first_in_half(r.ex && z0[c] == m0[c], half);
            const float gz = gx[c] + (p == am0 ? gm : 0.f);
            ga[c] = (r.act && r.ex && z0[c] > 0.f) ? gz : 0.f;
            a_db[c] += ga[c];
            a_dg[c] = fmaf(ga[c], (r.y0[c] - sc->mu0[c]) * sc->inv0[c], a_dg[c]);
        }
        // sparse dW0 = sum ga x0^T and the moments of x0: over the live slots only (x0 = 0 on padded slots)
#pragma unroll
        for (int c = 0; c < C0; ++c) s_g[wv][half][p][c] = ga[c];
#pragma unroll
        for (int j = 0; j < CIN; ++j) s_x[wv][half][p][j] = r.x0[j];
        const int n_loop = max(__shfl(r.n, 0, 64), __shfl(r.n, 32, 64));
        for (int q = 0; q < n_loop; ++q) {
            const bool live = r.act && q < r.n;
#pragma unroll
            for (int i = 0; i < EPL; ++i) {
                const int e = p + i * P;
                float v = 0.f;
                if (e < C0 * CIN) v = s_g[wv][half][q][e / CIN] * s_x[wv][half][q][e % CIN];
                else if (e < C0 * CIN + CIN) v = s_x[wv][half][q][e - C0 * CIN];
                else if (e < NE) v = s_x[wv][half][q][(e - C0 * CIN - CIN) / CIN] * s_x[wv][half][q][(e - C0 * CIN - CIN) % CIN];
                if (live) a_e[i] += v;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C0; ++c) { a_db[c] = hvpr_reduce_sum<32>(a_db[c]); a_dg[c] = hvpr_reduce_sum<32>(a_dg[c]); }
    wg_reduce_store<kB2>(s_row, part, wv, half, [&](auto add) {
#pragma unroll
        for (int c = 0; c < C0; ++c)
            if (p == c) { add(c, a_db[c]); add(C0 + c, a_dg[c]); }
#pragma unroll
        for (int i = 0; i < EPL; ++i)
            if (p + i * P < NE) add(2 * C0 + p + i * P, a_e[i]);
    });
}

// one workgroup: finish layer 0 — dbeta0, dgamma0, dW0
__global__ void __launch_bounds__(256) k_vfe_fin_bwd0(const double *__restrict__ sh, const double *__restrict__ local, const float *__restrict__ w0,
                                                      const float *__restrict__ gamma0, const Scratch *__restrict__ s,
                                                      float *__restrict__ dw0, float *__restrict__ dgamma0, float *__restrict__ dbeta0) {
    const double count = sh[-1];
    const double *db = sh, *dg = sh + C0, *gw = sh + 2 * C0, *s0 = gw + C0 * CIN, *S0 = s0 + CIN;
    if (threadIdx.x < C0) { dbeta0[threadIdx.x] = (float)local[threadIdx.x]; dgamma0[threadIdx.x] = (float)local[C0 + threadIdx.x]; }
    if (threadIdx.x < C0 * CIN) {
        const int c = threadIdx.x / CIN, k = threadIdx.x % CIN;
        const double c0 = (double)gamma0[c] * (double)s->inv0[c];
        double ws = 0.0;
        for (int j = 0; j < CIN; ++j) ws += (double)w0[c * CIN + j] * S0[j * CIN + k];
        const double dense = (c0 / count) * (db[c] * s0[k] + dg[c] * (double)s->inv0[c] * (ws - (double)s->mu0[c] * s0[k]));
        dw0[threadIdx.x] = (float)(c0 * gw[threadIdx.x] - dense);
    }
}

// workspace: [kBlocks][kMaxRow] partial rows (f32) | count, [kMaxRow] reduced sums, [2 * C1] per-rank copy (f64) | Scratch
constexpr size_t kPartBytes = (size_t)kBlocks * kMaxRow * sizeof(float), kSumBytes = (size_t)(2 + kMaxRow + 2 * C1) * sizeof(double);
size_t ws_bytes() { return kPartBytes + kSumBytes + sizeof(Scratch) + 256; }

}  // namespace

extern "C" size_t hvpr_pillar_vfe_train_workspace_bytes(void) { return ws_bytes(); }

// statistics of both layers into the workspace scratch (and mean / biased variance out); MODE 2 pass when `out` is given
static int vfe_train_forward(const float *voxels, const int32_t *num_points, const int32_t *coords, long long M, int PS, const float *w0,
                             const float *gamma0, const float *beta0, const float *w1, const float *gamma1, const float *beta1, float eps,
                             Geom g, float *out, float *mean0, float *var0, float *mean1, float *var1, void *workspace, hipStream_t s) {
    float *part = (float *)workspace;
    double *sums = (double *)((char *)workspace + kPartBytes) + 2;          // sums[-1]: the count
    Scratch *sc = (Scratch *)((char *)workspace + kPartBytes + kSumBytes);
    const double count = (double)M * PS;
    hipLaunchKernelGGL(k_vfe_train_fwd<0>, dim3(kBlocks), dim3(kThreads), 0, s, (const float4 *)voxels, num_points, (const int4 *)coords, M, PS, g,
                       w0, w1, sc, part, nullptr);
    hipLaunchKernelGGL(k_vfe_reduce_rows, dim3(hvpr_cdiv(kF1, 256)), dim3(256), 0, s, part, kF1, count, sums, 0, nullptr);
    if (hvpr_i_bn_allreduce(sums - 1, 1 + kF1, s) != 0) return -1;          // SyncBatchNorm: count, sum y0, sum y0^2 over all ranks
    hipLaunchKernelGGL(k_vfe_fin_stats, dim3(1), dim3(256), 0, s, (const double *)sums, C0, eps, gamma0, beta0, 0, sc, mean0, var0);
    hipLaunchKernelGGL(k_vfe_train_fwd<1>, dim3(kBlocks), dim3(kThreads), 0, s, (const float4 *)voxels, num_points, (const int4 *)coords, M, PS, g,
                       w0, w1, sc, part, nullptr);
    hipLaunchKernelGGL(k_vfe_reduce_rows, dim3(hvpr_cdiv(kF2, 256)), dim3(256), 0, s, part, kF2, count, sums, 0, nullptr);
    if (hvpr_i_bn_allreduce(sums - 1, 1 + kF2, s) != 0) return -1;
    hipLaunchKernelGGL(k_vfe_fin_stats, dim3(1), dim3(256), 0, s, (const double *)sums, C1, eps, gamma1, beta1, 1, sc, mean1, var1);
    if (out)
        hipLaunchKernelGGL(k_vfe_train_fwd<2>, dim3(kBlocks), dim3(kThreads), 0, s, (const float4 *)voxels, num_points, (const int4 *)coords, M,
                           PS, g, w0, w1, sc, part, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int hvpr_pillar_vfe_train_fwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P_,
                                             const float *w0, const float *gamma0, const float *beta0, const float *w1,
                                             const float *gamma1, const float *beta1, float eps, float vs_x, float vs_y, float vs_z,
                                             float off_x, float off_y, float off_z, float *pillar_features, float *mean0, float *var0,
                                             float *mean1, float *var1, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!voxels || !num_points || !coords || !w0 || !gamma0 || !beta0 || !w1 || !gamma1 || !beta1 || !pillar_features || !mean0 || !var0 ||
        !mean1 || !var1 || !workspace || M < 1)
        return HVPR_ERR_INVALID_ARG;
    if (P_ < 1 || P_ > P) return HVPR_ERR_UNSUPPORTED;      // one lane per slot of a half-wave
    if (workspace_bytes < ws_bytes()) return HVPR_ERR_WORKSPACE;
    const Geom g = {vs_x, vs_y, vs_z, off_x, off_y, off_z};
    if (vfe_train_forward(voxels, num_points, coords, M, P_, w0, gamma0, beta0, w1, gamma1, beta1, eps, g, pillar_features, mean0, var0, mean1,
                          var1, workspace, (hipStream_t)stream) != 0)
        return HVPR_ERR_LAUNCH;
    return HVPR_OK;
}

extern "C" int hvpr_pillar_vfe_bwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P_, const float *w0,
                                       const float *gamma0, const float *beta0, const float *w1, const float *gamma1, const float *beta1,
                                       float eps, float vs_x, float vs_y, float vs_z, float off_x, float off_y, float off_z,
                                       const float *d_pillar_features, float *dw0, float *dgamma0, float *dbeta0, float *dw1, float *dgamma1,
                                       float *dbeta1, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!voxels || !num_points || !coords || !w0 || !gamma0 || !beta0 || !w1 || !gamma1 || !beta1 || !d_pillar_features || !dw0 || !dgamma0 ||
        !dbeta0 || !dw1 || !dgamma1 || !dbeta1 || !workspace || M < 1)
        return HVPR_ERR_INVALID_ARG;
    if (P_ < 1 || P_ > P) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < ws_bytes()) return HVPR_ERR_WORKSPACE;
    const Geom g = {vs_x, vs_y, vs_z, off_x, off_y, off_z};
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)workspace;
    double *sums = (double *)((char *)workspace + kPartBytes) + 2, *local = sums + kMaxRow;
    Scratch *sc = (Scratch *)((char *)workspace + kPartBytes + kSumBytes);
    // the batch statistics are recomputed (two passes) instead of being trusted from a caller: the workspace carries no state
    // between calls, and the dgamma1 / dbeta1 outputs serve as the throw-away mean / variance destinations until they are written
    if (vfe_train_forward(voxels, num_points, coords, M, P_, w0, gamma0, beta0, w1, gamma1, beta1, eps, g, nullptr, dgamma0, dbeta0, dgamma1,
                          dbeta1, workspace, s) != 0)
        return HVPR_ERR_LAUNCH;
    const double count = (double)M * P_;
    hipLaunchKernelGGL(k_vfe_train_bwd1, dim3(kBlocks), dim3(kThreads), 0, s, (const float4 *)voxels, num_points, (const int4 *)coords,
                       (long long)M, P_, g, w0, w1, sc, d_pillar_features, part);
    hipLaunchKernelGGL(k_vfe_reduce_rows, dim3(hvpr_cdiv(kB1, 256)), dim3(256), 0, s, part, kB1, count, sums, 2 * C1, local);
    if (hvpr_i_bn_allreduce(sums - 1, 1 + 2 * C1, s) != 0) return HVPR_ERR_LAUNCH;     // count, dbeta1, dgamma1 over all ranks
    hipLaunchKernelGGL(k_vfe_fin_bwd1, dim3(1), dim3(256), 0, s, (const double *)sums, (const double *)local, w1, gamma1, sc, dw1, dgamma1, dbeta1);
    hipLaunchKernelGGL(k_vfe_train_bwd0, dim3(kBlocks), dim3(kThreads), 0, s, (const float4 *)voxels, num_points, (const int4 *)coords,
                       (long long)M, P_, g, w0, w1, gamma1, sc, d_pillar_features, part);
    hipLaunchKernelGGL(k_vfe_reduce_rows, dim3(hvpr_cdiv(kB2, 256)), dim3(256), 0, s, part, kB2, count, sums, 2 * C0, local);
    if (hvpr_i_bn_allreduce(sums - 1, 1 + 2 * C0, s) != 0) return HVPR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_vfe_fin_bwd0, dim3(1), dim3(256), 0, s, (const double *)sums, (const double *)local, w0, gamma0, sc, dw0, dgamma0, dbeta0);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
