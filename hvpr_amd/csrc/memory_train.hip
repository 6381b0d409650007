// a10 (training) — MemAE memory addressing with hard shrinkage, forward and backward, WITHOUT materialising the
// (rows x items) attention matrix.  Replaces the training branch of MemoryUnit_Agg.forward,
// pcdet/models/backbones_2d/map_to_bev/memory_module.py:31-59 (hard_shrink_relu :85-87), up to the per-pillar aggregation of
// :50-57 (a (M, k) softmax the caller keeps in torch):
//     l = x W^T,  a = softmax(l),  s = relu(a - lambda) a / (|a - lambda| + 1e-12),  t = s / max(||s||_1, 1e-12),  y = t W
// for R = M * k rows x (R, 64) and a bank W (n_items <= 2048, 64).  The reference materialises a, s and t as (R, n_items)
// tensors (9.8 GB each at batch 16) and walks them with ~15 element-wise kernels forward and as many backward.
//
// Structure (the read-out kernel's, memory_scatter.hip): one workgroup = 16 rows; the 16 x n_items logits are computed on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 — softmax needs every logit, a reduced-precision pre-filter has nothing
// to filter) straight into LDS; one wave per row then does softmax statistics, the support S = {a_j > lambda} (at most
// 1 / lambda items, usually a handful), and the sparse sums over S.
//   forward  y, and per row (max, Z, ||s||_1) for the backward.
//   backward rows    recomputes the logits, then per row  q = sum_S t dt,  ds = (dt - q) / n,  da = ds hs'(a),  c = sum_S a da,
//                    dl_j = a_j (da_j - c)  for EVERY item (softmax couples them all),
//                    dx = sum_S a da w  -  c * (a . W)          (the dense product a . W: a second matrix-core pass over LDS)
//                    and the per-row scalars (c, q) for the bank gradient.
//   backward items   dW_j = sum_r [ a_rj (da_rj - c_r) x_r + t_rj dy_r ]: item-block-stationary; the logits tile x W_blk^T AND the
//                    tile dy W_blk^T (= dt) are recomputed on the matrix cores, turned into the two coefficient tiles in the
//                    accumulator registers and fed back as the A operands of the two second products without leaving the
//                    accumulator layout.  No atomics anywhere: a first version scatter-added the support terms from the row
//                    kernel (64 atomics per hit, 15 M per call on a few hot bank rows: 1.5 ms of its 2.3 ms).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kC = 64, kRows = 16, kItemsPad = 2048, kPitch = kItemsPad + 4, kThreads = 1024, kWaves = kThreads / 64;

// bank (n_items, 64) -> [tile of 16 items][4 channel groups][64 lanes] float4: lane (l15, q) of group g holds channels
// 16g + 4q .. + 3 of item 16 tile + l15 (rows past n_items zero) — 1 KB contiguous per load instruction
__global__ void __launch_bounds__(256) k_bank_pack_f32(const float *__restrict__ bank, int n_items, float4 *__restrict__ packed, int n_out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    const int t = o >> 8, g = (o >> 6) & 3, ln = o & 63, row = 16 * t + (ln & 15);
    packed[o] = row < n_items ? *(const float4 *)(bank + (size_t)row * kC + 16 * g + 4 * (ln >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// logits of the workgroup's 16 rows (features already in s_f) against the whole bank into s_logit; items past n_items = -inf
__device__ __forceinline__ void logits_to_lds(const float *s_f, float *s_logit, const float4 *__restrict__ bank_packed, int n_items) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    for (int i = tid; i < kRows * (kItemsPad - n_items); i += kThreads)
        s_logit[(i / (kItemsPad - n_items)) * kPitch + n_items + i % (kItemsPad - n_items)] = -INFINITY;
    float4 bf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bf[g] = *(const float4 *)(s_f + l15 * kC + 16 * g + 4 * q);
    const int n_tiles = (n_items + 15) >> 4;
    const int first = (int)((__builtin_amdgcn_readfirstlane((unsigned)wid) + blockIdx.x) % (unsigned)kWaves);
    float *const lrow = s_logit + l15 * kPitch + 4 * q;
    constexpr int kMaxTiles = kItemsPad / 16 / kWaves;      // 8
    // two tiles in flight per wave (4 float4 each): 16 waves x 8 KB per CU
    float4 a0[4], a1[4];
    auto load = [&](int t, float4 (&a)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = bank_packed[(size_t)t * 256 + g * 64 + lane];
    };
    auto mul = [&](int t, const float4 (&a)[4]) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].x, bf[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].y, bf[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].z, bf[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].w, bf[g].w, acc, 0, 0, 0);
        }
        // C/D map of 16x16: column (row of x) = lane & 15, row (item) = 4 * (lane >> 4) + reg
        if (16 * t + 16 <= n_items) {
            *(float4 *)(lrow + t * 16) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * t + 4 * q + r < n_items) lrow[t * 16 + r] = acc[r];
        }
    };
    if (first < n_tiles) load(first, a0);
#pragma unroll
    for (int i = 0; i < kMaxTiles; i += 2) {
        const int t0 = first + i * kWaves, t1 = t0 + kWaves, t2 = t1 + kWaves;
        if (t1 < n_tiles) load(t1, a1);
        if (t0 < n_tiles) mul(t0, a0);
        if (t2 < n_tiles && i + 2 < kMaxTiles) load(t2, a0);
        if (t1 < n_tiles) mul(t1, a1);
    }
}

// exp(x) for x <= 0 (a logit minus its row maximum; -inf for the padding past n_items) in six instructions: 2^(x L) with L = log2(e)
// split into a float and its remainder, the product's rounding error recovered by an fma, v_exp_f32 on the rounded product and a
// first-order correction for the rest — within ~1 ulp of expf(), whose library expansion (range checks, ldexp, denormal paths)
// measured as HALF of the bank-gradient kernel's time (13.4 -> 6.8 ms without it): 2.4 G exponentials per call.
__device__ __forceinline__ float exp_nonpos(float x) {
    x = fmaxf(x, -150.f);                                                 // exp2(-216) = 0: keeps -inf out of the fma below
    const float t = x * 1.44269504088896341f;
    const float r = fmaf(x, 1.44269504088896341f, -t) + x * 1.92596299112661746e-8f;
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e * r, 0.693147180559945309f, e);
}

__device__ __forceinline__ float hard_shrink(float a, float lambd) {      // memory_module.py:85-87, a > lambd
    const float u = a - lambd;
    return (u * a) / (u + 1e-12f);
}
__device__ __forceinline__ float hard_shrink_grad(float a, float lambd) {
    const float u = a - lambd, d = u + 1e-12f;
    return (u * u + 1e-12f * (a + u)) / (d * d);
}

// ------------------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(kThreads) k_memtrain_fwd(const float *__restrict__ x, long long R, const float *__restrict__ bank,
                                                           const float4 *__restrict__ bank_packed, int n_items, float lambd,
                                                           float *__restrict__ y, float *__restrict__ stats /* [R][4]: max, Z, n, - */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_logit = (float *)smem;
    float *s_f = s_logit + kRows * kPitch;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long r0 = (long long)blockIdx.x * kRows;
    const int nr = (int)min((long long)kRows, R - r0);
    for (int i = tid; i < kRows * kC; i += kThreads) s_f[i] = (i / kC) < nr ? x[(r0 + i / kC) * kC + (i % kC)] : 0.f;
    __syncthreads();
    logits_to_lds(s_f, s_logit, bank_packed, n_items);
    __syncthreads();
    for (int p = wid; p < nr; p += kWaves) {
        const float *row = s_logit + p * kPitch;
        float v[kItemsPad / 64];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < kItemsPad / 64; ++t) { v[t] = row[lane + 64 * t]; mx = fmaxf(mx, v[t]); }
        mx = hvpr_reduce_max<64>(mx);
        float z = 0.f;
#pragma unroll
        for (int t = 0; t < kItemsPad / 64; ++t) { v[t] = exp_nonpos(v[t] - mx); z += v[t]; }     // 0 for the padding (-inf)
        z = hvpr_reduce_sum<64>(z);
        const float inv_z = 1.f / z;
        // support: a_j > lambda.  One hit per lane per round, in (t, lane) order: deterministic sums.
        float n = 0.f, acc = 0.f;
#pragma unroll
        for (int t = 0; t < kItemsPad / 64; ++t) {
            const float a = v[t] * inv_z;
            const bool hit = a > lambd;
            unsigned long long m = __ballot(hit);
            if (m == 0ull) continue;                         // the common case: nothing of this slice is in the support
            const float s = hit ? hard_shrink(a, lambd) : 0.f;
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), src));
                n += sj;
                acc = fmaf(sj, bank[(size_t)(64 * t + src) * kC + lane], acc);
            }
        }
        y[(r0 + p) * kC + lane] = acc / fmaxf(n, 1e-12f);
        if (lane == 0) *(float4 *)(stats + (r0 + p) * 4) = make_float4(mx, z, n, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ backward, row-wise part
__global__ void __launch_bounds__(kThreads) k_memtrain_bwd_rows(const float *__restrict__ x, const float *__restrict__ dy, long long R,
                                                                const float *__restrict__ bank, const float4 *__restrict__ bank_packed,
                                                                int n_items, float lambd, const float *__restrict__ stats,
                                                                float *__restrict__ dx, float *__restrict__ crow /* [R][2]: c, q */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_logit = (float *)smem;                      // [16][kPitch]: logits, then a_j in place
    float *s_f = s_logit + kRows * kPitch;               // [16][64]
    float *s_abar = s_f + kRows * kC;                    // [16][64]  a . W
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long r0 = (long long)blockIdx.x * kRows;
    const int nr = (int)min((long long)kRows, R - r0);
    for (int i = tid; i < kRows * kC; i += kThreads) s_f[i] = (i / kC) < nr ? x[(r0 + i / kC) * kC + (i % kC)] : 0.f;
    __syncthreads();
    logits_to_lds(s_f, s_logit, bank_packed, n_items);
    __syncthreads();
    float c_row = 0.f, dxs = 0.f, q_row = 0.f;          // of the row this wave owns (p = wid)
    bool live = false;
    {
        const int p = wid;
        if (p < nr) {
            float *row = s_logit + p * kPitch;
            const float4 st = *(const float4 *)(stats + (r0 + p) * 4);
            const float mx = st.x, inv_z = 1.f / st.y, n = st.z;
            const float dyc = dy[(r0 + p) * kC + lane];
            live = n > 0.f;
            const float inv_n = 1.f / fmaxf(n, 1e-12f);
            float v[kItemsPad / 64];
#pragma unroll
            for (int t = 0; t < kItemsPad / 64; ++t) { v[t] = exp_nonpos(row[lane + 64 * t] - mx) * inv_z; row[lane + 64 * t] = v[t]; }   // a_j, in place
            if (live) {
                // pass 1 over the support: q = sum t_j dt_j
                float q = 0.f;
#pragma unroll
                for (int t = 0; t < kItemsPad / 64; ++t) {
                    const float a = v[t];
                    const bool hit = a > lambd;
                    unsigned long long m = __ballot(hit);
                    if (m == 0ull) continue;
                    const float tj = hit ? hard_shrink(a, lambd) * inv_n : 0.f;
                    while (m) {
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1ull;
                        const float dt = hvpr_reduce_sum<64>(dyc * bank[(size_t)(64 * t + src) * kC + lane]);
                        q = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(tj), src)), dt, q);
                    }
                }
                q_row = q;
                // pass 2: da, c, the sparse part of dx
#pragma unroll
                for (int t = 0; t < kItemsPad / 64; ++t) {
                    const float a = v[t];
                    const bool hit = a > lambd;
                    unsigned long long m = __ballot(hit);
                    if (m == 0ull) continue;
                    const float hg = hit ? hard_shrink_grad(a, lambd) : 0.f;
                    while (m) {
                        const int src = __ffsll((long long)m) - 1;
                        m &= m - 1ull;
                        const int j = 64 * t + src;
                        const float w = bank[(size_t)j * kC + lane];
                        const float dt = hvpr_reduce_sum<64>(dyc * w);
                        const float aj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), src));
                        const float da = (dt - q) * inv_n * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hg), src));
                        const float ada = aj * da;
                        c_row += ada;
                        dxs = fmaf(ada, w, dxs);
                    }
                }
            }
            if (lane == 0) *(float2 *)(crow + 2 * (r0 + p)) = make_float2(c_row, q_row);
        }
    }
    // every row of s_logit now holds a_j.  The dense product is only needed by rows with a non-empty support (c != 0 only there):
    // a workgroup without any (the state right after initialisation, where no softmax value reaches the threshold) is done.
    if (!__syncthreads_or(live ? 1 : 0)) {
        if (wid < nr) dx[(r0 + wid) * kC + lane] = 0.f;
        return;
    }
    // abar = a . W for the 16 rows on v_mfma_f32_16x16x4_f32: D (16 rows x 16 channels) += A (rows x 4 items) . B (4 items x 16
    // channels).  Lane (l15, q) of B supplies channels 4 l15 .. 4 l15 + 3 of item 4 s + q — ONE float4 of the row-major bank,
    // 256 contiguous bytes per item over the 16 lanes — to four MFMAs (channel = 4 l15 + r for accumulator r: which 16
    // channels an MFMA covers is a free choice).  Every wave covers all 64 channels for its sixteenth of the items; the sixteen
    // partial results are added through LDS (the logits area: nobody reads a_j any more after the barrier).
    {
        const int l15 = lane & 15, q = lane >> 4;
        const int n4 = (n_items + 3) >> 2, per = (n4 + kWaves - 1) / kWaves;
        const int k_lo = wid * per, k_hi = min(n4, k_lo + per);
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k4 = k_lo; k4 < k_hi; k4 += 8) {             // eight k steps per round: their bank loads are in flight together
            float av[8];
            float4 bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int item = 4 * (k4 + u) + q;
                const bool ok = k4 + u < k_hi && item < n_items;
                av[u] = ok ? s_logit[l15 * kPitch + item] : 0.f;                                                   // A[row = l15][k = q]
                bv[u] = ok ? *(const float4 *)(bank + (size_t)item * kC + 4 * l15) : make_float4(0.f, 0.f, 0.f, 0.f);   // B[k = q][channels 4 l15 ..]
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u].w, acc[3], 0, 0, 0);
            }
        }
        __syncthreads();          // all waves are done with a_j: the logits area becomes the partial-sum area [wave][row][channel]
        // C/D map: column j = l15 (channel 4 l15 + r of accumulator r), row (row of x) = 4 q + reg
        float *part = s_logit + wid * (kRows * kC);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) part[(4 * q + g) * kC + 4 * l15 + r] = acc[r][g];
    }
    __syncthreads();
    for (int i = tid; i < kRows * kC; i += kThreads) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s += s_logit[w * (kRows * kC) + i];
        s_abar[i] = s;
    }
    __syncthreads();
    if (wid < nr) dx[(r0 + wid) * kC + lane] = live ? dxs - c_row * s_abar[wid * kC + lane] : 0.f;
}

// ------------------------------------------------------------------------------------------------ backward, bank gradient
// dW[j] = sum_r [ cx_rj x_r + cy_rj dy_r ],  cx = a (da - c),  cy = t   (da, t = 0 outside the support a > lambda).
// Workgroup = 128 items (4 waves x 32) x a contiguous slice of the rows, walked 64 rows at a time:
//   P^T (rows x items) = x W_blk^T, D^T = dy W_blk^T      32x32x2 MFMA, M = rows, N = items, K = 64 channels
//   a = exp(P - max_r) / Z_r;  cx, cy from (a, D, q_r, 1/n_r, c_r)   in the accumulator registers
//   dWpart (items x channels) += CX (items x rows) . x + CY . dy: an accumulator register r of P^T holds (row rho(r) + 4 * half,
//   item lane & 31) — exactly the A operand A[i = item][k = half] of a 32x32x2 MFMA whose two k are rows rho(r) and rho(r) + 4,
//   so the coefficient tiles never leave their registers.
constexpr int kIB = 128, kRT = 64, kXP = kC + 1;      // LDS row pitch 65: a column of 32 rows hits 32 different banks

__global__ void __launch_bounds__(256) k_memtrain_bwd_items(const float *__restrict__ x, const float *__restrict__ dy, long long R,
                                                            const float *__restrict__ bank, int n_items, float lambd,
                                                            const float *__restrict__ stats, const float *__restrict__ crow, int n_splits,
                                                            float *__restrict__ part /* [n_splits][n_items_pad128][64] */) {
    __shared__ float s_x[kRT * kXP], s_dy[kRT * kXP];   // 2 x 16.6 KB
    __shared__ float s_w[kIB * kXP];                    // 33 KB, the item block (zero rows past n_items)
    __shared__ float s_mx[kRT], s_iz[kRT], s_c[kRT], s_q[kRT], s_in[kRT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int ib = blockIdx.x, split = blockIdx.y;
    const int j0 = ib * kIB;
    for (int i = tid; i < kIB * kC / 4; i += 256) {
        const int it = i / (kC / 4), c4 = (i % (kC / 4)) * 4;
        const float4 wv = j0 + it < n_items ? *(const float4 *)(bank + (size_t)(j0 + it) * kC + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float *d = s_w + it * kXP + c4;
        d[0] = wv.x; d[1] = wv.y; d[2] = wv.z; d[3] = wv.w;
    }
    const long long n_rt = (R + kRT - 1) / kRT;
    const long long per = (n_rt + n_splits - 1) / n_splits;
    const long long rt_lo = split * per, rt_hi = min(n_rt, rt_lo + per);
    f32x16 dw[2];        // items (32 of this wave) x channels (2 blocks of 32)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[b][r] = 0.f;
    const int item_l = wid * 32 + l31;          // this lane's item inside the block (B column / A row)
    const bool item_ok = j0 + item_l < n_items;
    // The next 64 rows of x / dy (and their row scalars) travel in registers while this tile multiplies: requested right after the
    // barrier that opens a tile, written to LDS between the two barriers of the next one.  Unconditional loads from a clamped row
    // (zeroed on the way to LDS): a select on a load result is a branch and a wait per load.
    constexpr int kPer = kRT * kC / 4 / 256;             // float4 per thread and tensor: 4
    float4 xr[kPer], dr[kPer], st_r = make_float4(0.f, 1.f, 0.f, 0.f);
    float2 cq_r = make_float2(0.f, 0.f);
    auto fetch = [&](long long rt) {
        const long long row0 = rt * kRT;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = tid + j * 256;
            const long long rc = min(row0 + i / (kC / 4), R - 1);
            const int c4 = (i % (kC / 4)) * 4;
            xr[j] = *(const float4 *)(x + rc * kC + c4);
            dr[j] = *(const float4 *)(dy + rc * kC + c4);
        }
        if (tid < kRT) {
            const long long rc = min(row0 + tid, R - 1);
            st_r = *(const float4 *)(stats + rc * 4);
            cq_r = *(const float2 *)(crow + 2 * rc);
        }
    };
    auto commit = [&](long long rt) {
        const long long row0 = rt * kRT;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = tid + j * 256;
            const bool ok = row0 + i / (kC / 4) < R;
            const int c4 = (i % (kC / 4)) * 4;
            float *d = s_x + (i / (kC / 4)) * kXP + c4, *e = s_dy + (i / (kC / 4)) * kXP + c4;
            d[0] = ok ? xr[j].x : 0.f; d[1] = ok ? xr[j].y : 0.f; d[2] = ok ? xr[j].z : 0.f; d[3] = ok ? xr[j].w : 0.f;
            e[0] = ok ? dr[j].x : 0.f; e[1] = ok ? dr[j].y : 0.f; e[2] = ok ? dr[j].z : 0.f; e[3] = ok ? dr[j].w : 0.f;
        }
        if (tid < kRT) {
            const bool ok = row0 + tid < R;
            s_mx[tid] = ok ? st_r.x : 0.f; s_iz[tid] = ok ? 1.f / st_r.y : 0.f; s_c[tid] = ok ? cq_r.x : 0.f; s_q[tid] = ok ? cq_r.y : 0.f;
            s_in[tid] = (ok && st_r.z > 0.f) ? 1.f / fmaxf(st_r.z, 1e-12f) : 0.f;          // rows without support: t = 0 and da = 0
        }
    };
    if (rt_lo < rt_hi) fetch(rt_lo);
    for (long long rt = rt_lo; rt < rt_hi; ++rt) {
        __syncthreads();
        commit(rt);
        __syncthreads();
        if (rt + 1 < rt_hi) fetch(rt + 1);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {        // 32 rows at a time
            // P^T[row][item] = sum_ch x[row][ch] W[item][ch]: A[i = row = l31][k] = x, B[k][j = item = l31] = W, k = channel pair
            f32x16 p, d;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = 0.f; d[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < kC / 2; ++kk)
                p = __builtin_amdgcn_mfma_f32_32x32x2f32(s_x[(rb * 32 + l31) * kXP + 2 * kk + half], s_w[item_l * kXP + 2 * kk + half], p, 0, 0, 0);
            // C/D map: column (item) = l31, row (row of x) = (r & 3) + 8 (r >> 2) + 4 half.   p -> a
            bool hit = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                p[r] = item_ok ? exp_nonpos(p[r] - s_mx[rr]) * s_iz[rr] : 0.f;
                hit |= p[r] > lambd;
            }
            // the support {a > lambda} is a handful of items per row, usually the same few: most 32 x 32 tiles have none, and
            // for those dt = dy W^T and the product CY . dy (cy = t = 0 outside the support) are not needed at all
            const bool any = __ballot(hit) != 0ull;             // wave-uniform
            if (any) {
#pragma unroll
                for (int kk = 0; kk < kC / 2; ++kk)
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(s_dy[(rb * 32 + l31) * kXP + 2 * kk + half], s_w[item_l * kXP + 2 * kk + half], d, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float a = p[r];
                float cx = -a * s_c[rr], cy = 0.f;
                if (a > lambd) {
                    const float in = s_in[rr];
                    cy = hard_shrink(a, lambd) * in;
                    cx = a * ((d[r] - s_q[rr]) * in * hard_shrink_grad(a, lambd) - s_c[rr]);
                }
                p[r] = cx; d[r] = cy;
            }
            // dw[item][ch] += sum_rows CX[item][row] x[row][ch] + CY[item][row] dy[row][ch]: A[i = item = l31][k = half] = the
            // coefficient register (rows rho(r), rho(r) + 4), B[k = half][j = ch = l31] = x / dy [row rho(r) + 4 half][ch].
            // (Adding the CY products of a tile without support would add exact zeros: skipping them changes no bit.)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
                for (int b = 0; b < 2; ++b) dw[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[r], s_x[rr * kXP + b * 32 + l31], dw[b], 0, 0, 0);
            }
            if (any) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
                    for (int b = 0; b < 2; ++b) dw[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[r], s_dy[rr * kXP + b * 32 + l31], dw[b], 0, 0, 0);
                }
            }
        }
    }
    const int ipad = (n_items + kIB - 1) / kIB * kIB;
    float *out = part + ((size_t)split * ipad + j0 + wid * 32) * kC;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int it = (r & 3) + 8 * (r >> 2) + 4 * half;       // item inside the wave's 32 (row of the C/D map)
            out[(size_t)it * kC + b * 32 + l31] = dw[b][r];
        }
}

// dW[j][c] = sum over splits of part[split][j][c], in split order
__global__ void __launch_bounds__(256) k_memtrain_items_reduce(const float *__restrict__ part, int n_splits, int n_items, int ipad,
                                                               float *__restrict__ dW) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items * kC) return;
    float s = 0.f;
    for (int k = 0; k < n_splits; ++k) s += part[(size_t)k * ipad * kC + i];
    dW[i] = s;
}

__global__ void __launch_bounds__(256) k_zero_f(float *__restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.f;
}

constexpr int kSplits = 32;
size_t fwd_lds() { return (size_t)kRows * kPitch * 4 + kRows * kC * 4; }
size_t bwd_lds() { return (size_t)kRows * kPitch * 4 + 2 * kRows * kC * 4; }

}  // namespace

extern "C" size_t hvpr_memory_train_workspace_bytes(int n_items) {
    if (n_items < 1) return 0;
    const size_t packed = (size_t)((n_items + 15) / 16) * 16 * kC * sizeof(float);
    const size_t part = (size_t)kSplits * ((n_items + kIB - 1) / kIB * kIB) * kC * sizeof(float);
    return packed + part + 256;
}

extern "C" int hvpr_memory_train_fwd_f32(const float *x, long long R, const float *bank, int n_items, float shrink_thres, float *y,
                                         float *row_stats, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (R < 0 || n_items < 1) return HVPR_ERR_INVALID_ARG;
    if (n_items > kItemsPad || !(shrink_thres > 0.f)) return HVPR_ERR_UNSUPPORTED;
    if (R == 0) return HVPR_OK;
    if (!x || !bank || !y || !row_stats || !workspace) return HVPR_ERR_INVALID_ARG;
    if (workspace_bytes < hvpr_memory_train_workspace_bytes(n_items)) return HVPR_ERR_WORKSPACE;
    if (R > 0x7fffffffll * kRows) return HVPR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    float4 *packed = (float4 *)workspace;
    const int n_out = ((n_items + 15) / 16) * 256;
    hipLaunchKernelGGL(k_bank_pack_f32, dim3(hvpr_cdiv(n_out, 256)), dim3(256), 0, s, bank, n_items, packed, n_out);
    static unsigned long long lds_set = 0ull;
    if (hvpr_ensure_dyn_lds((const void *)k_memtrain_fwd, (int)fwd_lds(), &lds_set) != 0) return HVPR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_memtrain_fwd, dim3((unsigned)hvpr_cdiv(R, kRows)), dim3(kThreads), fwd_lds(), s, x, R, bank, packed, n_items, shrink_thres,
                       y, row_stats);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_memory_train_bwd_f32(const float *x, const float *dy, long long R, const float *bank, int n_items, float shrink_thres,
                                         const float *row_stats, float *dx, float *dbank, float *row_scratch, void *workspace,
                                         size_t workspace_bytes, hvpr_stream_t stream) {
    if (R < 0 || n_items < 1) return HVPR_ERR_INVALID_ARG;
    if (n_items > kItemsPad || !(shrink_thres > 0.f)) return HVPR_ERR_UNSUPPORTED;
    if (!dbank || !bank) return HVPR_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (R == 0) {
        hipLaunchKernelGGL(k_zero_f, dim3(hvpr_cdiv((long long)n_items * kC, 256)), dim3(256), 0, s, dbank, (long long)n_items * kC);
        HVPR_CHECK_LAUNCH();
        return HVPR_OK;
    }
    if (!x || !dy || !row_stats || !dx || !row_scratch || !workspace) return HVPR_ERR_INVALID_ARG;
    if (workspace_bytes < hvpr_memory_train_workspace_bytes(n_items)) return HVPR_ERR_WORKSPACE;
    float4 *packed = (float4 *)workspace;
    const size_t packed_floats = (size_t)((n_items + 15) / 16) * 16 * kC;
    float *part = (float *)workspace + packed_floats;
    const int n_out = ((n_items + 15) / 16) * 256;
    hipLaunchKernelGGL(k_bank_pack_f32, dim3(hvpr_cdiv(n_out, 256)), dim3(256), 0, s, bank, n_items, packed, n_out);
    static unsigned long long lds_set = 0ull;
    if (hvpr_ensure_dyn_lds((const void *)k_memtrain_bwd_rows, (int)bwd_lds(), &lds_set) != 0) return HVPR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_memtrain_bwd_rows, dim3((unsigned)hvpr_cdiv(R, kRows)), dim3(kThreads), bwd_lds(), s, x, dy, R, bank, packed, n_items,
                       shrink_thres, row_stats, dx, row_scratch);
    const int ipad = (n_items + kIB - 1) / kIB * kIB;
    hipLaunchKernelGGL(k_memtrain_bwd_items, dim3(ipad / kIB, kSplits), dim3(256), 0, s, x, dy, R, bank, n_items, shrink_thres, row_stats,
                       row_scratch, kSplits, part);
    hipLaunchKernelGGL(k_memtrain_items_reduce, dim3(hvpr_cdiv((long long)n_items * kC, 256)), dim3(256), 0, s, part, kSplits, n_items, ipad, dbank);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
