// a10 (training) — the index half of the point <-> pillar attention: for every pillar the k points of its sample with the
// largest  pillar . point  logits, in descending order.  Replaces the top-k of get_score,
// pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:70-75 (softmax over the points is monotone per pillar, so the top-k
// of the raw logits gives the same indices), without the (N, M) logit matrix (60 M elements per sample).
//
// The read-out kernel's scheme (memory_scatter.hip) stretched over any number of items: 16 pillars per workgroup, the items in
// blocks of 2048.  Per block: fp16 matrix-core logits A into LDS, per pillar a lower bound tau_b of the block's k-th largest A;
// T = max_b tau_b is a lower bound of the k-th largest A over everything seen, and with the rounding bound eps of the pre-filter
// every member of the exact top-k has A >= T - 2 eps (memory_scatter.hip, step 2).  Candidates travel in a 64-entry list per
// pillar (LDS, two entries per lane) that is pruned with the rising T.  When the list is full, and once at the end, it is COMMITTED:
// its entries get their exact fp32 logits (butterfly-tree sum of the 64 products, the definition the read-out uses), and the k
// best of them and of the k committed so far — by (logit desc, index asc) — stay in registers, rank-ordered.  The k-th committed
// exact logit Lk then bounds the filter too (a member of the final top-k has L >= Lk, so A >= Lk - eps), which keeps frames that
// repeat points (sample_points pads by repetition: exact ties), or pillars whose logits crowd together, at a few commits instead
// of a fallback over every item.  Features outside the fp16 range make every item a candidate (a commit per 128 items).
#include "common.h"
#include "internal.h"
#include "select.h"

using namespace hvpr_sel;

namespace {

constexpr int kC = 64, kPillars = 16, kBlk = 2048, kPitch = kBlk + 4, kThreads = 1024, kWaves = kThreads / 64;
constexpr float kRelErr = 1.0e-3f, kAbsErr = 6.2e-5f, kHalfMax = 65000.f;      // as in memory_scatter.hip
constexpr int kCap = 128;          // candidate list entries per pillar (two per lane): frames padded by repeating points double the ties

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int src) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(v & 0xffffffffull), src);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// keep the entries with A >= thr, compacted to the front (one wave; entries e0 = cand[lane], e1 = cand[64 + lane])
__device__ __forceinline__ int prune(unsigned long long *cand, int lcnt, float thr, int lane) {
    const unsigned tb = ord_bits(thr);
    const unsigned long long e0 = lane < lcnt ? cand[lane] : 0ull, e1 = 64 + lane < lcnt ? cand[64 + lane] : 0ull;
    const bool k0 = lane < lcnt && tb <= (unsigned)(e0 >> 32), k1 = 64 + lane < lcnt && tb <= (unsigned)(e1 >> 32);
    const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
    const int n0 = __popcll(m0);
    // every read above is complete before the first write below (the values are consumed by the ballots)
    if (k0) cand[__popcll(m0 & ((1ull << lane) - 1ull))] = e0;
    if (k1) cand[n0 + __popcll(m1 & ((1ull << lane) - 1ull))] = e1;
    return n0 + __popcll(m1);
}

// Exact logits of the list's cnt <= kCap entries, merged with the committed keys ck (lane r < k: the r-th best so far, 0 = none):
// afterwards ck holds the k best of the union in rank order and Lk the k-th exact logit (once k keys exist).  Keys are
// (ordered bits of the exact logit) << 32 | ~item: unique, larger = better.  The list's storage is reused as the staging row.
__device__ __forceinline__ void commit(const float *__restrict__ items, float fc, unsigned long long *cand, int cnt, int k, int lane,
                                       unsigned long long &ck, float &Lk) {
    const int i0 = lane < cnt ? (int)(unsigned)(cand[lane] & 0xffffffffull) : 0;
    const int i1 = 64 + lane < cnt ? (int)(unsigned)(cand[64 + lane] & 0xffffffffull) : 0;
    float L0 = 0.f, L1 = 0.f;
#pragma unroll
    for (int part = 0; part < 4; ++part) {
        if (part * 32 < cnt) {             // wave-uniform
            float x[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int j = __builtin_amdgcn_readlane(part < 2 ? i0 : i1, (part & 1) * 32 + r);
                x[r] = items[(size_t)j * kC + lane];
            }
#pragma unroll
            for (int r = 0; r < 32; ++r) x[r] = __fmul_rn(x[r], fc);
            const float s = wave_sum32(x, lane);
            if ((lane >> 5) == (part & 1)) { if (part < 2) L0 = s; else L1 = s; }
        }
    }
    const unsigned long long key0 = lane < cnt ? (((unsigned long long)ord_bits(L0) << 32) | (unsigned)(0xffffffffu - (unsigned)i0)) : 0ull;
    const unsigned long long key1 = 64 + lane < cnt ? (((unsigned long long)ord_bits(L1) << 32) | (unsigned)(0xffffffffu - (unsigned)i1)) : 0ull;
    const int total = min(cnt, 64) + max(cnt - 64, 0) + __popcll(__ballot(ck != 0ull));
    const int kk = min(k, total);
    unsigned long long kth = 0ull;
#pragma unroll 1
    for (int bit = 63; bit >= 0; --bit) {
        const unsigned long long c = kth | (1ull << bit);
        if (__popcll(__ballot(key0 >= c)) + __popcll(__ballot(key1 >= c)) + __popcll(__ballot(ck >= c)) >= kk) kth = c;
    }
    const bool s0 = key0 != 0ull && key0 >= kth, s1 = key1 != 0ull && key1 >= kth, s2 = ck != 0ull && ck >= kth;
    int r0 = 0, r1 = 0, r2 = 0;
    for (unsigned long long m = __ballot(s0); m; m &= m - 1ull) {
        const unsigned long long o = readlane_u64(key0, __ffsll((long long)m) - 1);
        r0 += o > key0 ? 1 : 0; r1 += o > key1 ? 1 : 0; r2 += o > ck ? 1 : 0;
    }
    for (unsigned long long m = __ballot(s1); m; m &= m - 1ull) {
        const unsigned long long o = readlane_u64(key1, __ffsll((long long)m) - 1);
        r0 += o > key0 ? 1 : 0; r1 += o > key1 ? 1 : 0; r2 += o > ck ? 1 : 0;
    }
    for (unsigned long long m = __ballot(s2); m; m &= m - 1ull) {
        const unsigned long long o = readlane_u64(ck, __ffsll((long long)m) - 1);
        r0 += o > key0 ? 1 : 0; r1 += o > key1 ? 1 : 0; r2 += o > ck ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();           // the list was read into i0 / i1 above; LDS operations of one wave stay in order
    if (s0) cand[r0] = key0;
    if (s1) cand[r1] = key1;
    if (s2) cand[r2] = ck;
    __builtin_amdgcn_wave_barrier();
    ck = lane < kk ? cand[lane] : 0ull;
    __builtin_amdgcn_wave_barrier();
    if (kk == k) Lk = ord_to_float((unsigned)(readlane_u64(ck, k - 1) >> 32));
}

__global__ void __launch_bounds__(kThreads) k_point_topk(const float *__restrict__ f, int M, const float *__restrict__ items,
                                                         const uint4 *__restrict__ items_h, const float *__restrict__ wmax, int N,
                                                         int k, int *__restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_logit = (float *)smem;                              // [kPillars][kPitch]
    float *s_f = s_logit + kPillars * kPitch;                    // [kPillars][kC]
    unsigned long long *s_cand = (unsigned long long *)(s_f + kPillars * kC);   // [waves][kCap]: (ordered bits of A) << 32 | item
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p0 = blockIdx.x * kPillars;
    const int np = min(kPillars, M - p0);
    for (int i = tid; i < kPillars * kC; i += kThreads) s_f[i] = (i / kC) < np ? f[(size_t)(p0 + i / kC) * kC + (i % kC)] : 0.f;
    __syncthreads();
    const int l15 = lane & 15, q = lane >> 4;
    f16x8_t bfrag[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 lo = *(const float4 *)(s_f + l15 * kC + 32 * h + 8 * q), hi = *(const float4 *)(s_f + l15 * kC + 32 * h + 8 * q + 4);
        const uint4 u = make_uint4(f16_rne(lo.x) | (f16_rne(lo.y) << 16), f16_rne(lo.z) | (f16_rne(lo.w) << 16),
                                   f16_rne(hi.x) | (f16_rne(hi.y) << 16), f16_rne(hi.z) | (f16_rne(hi.w) << 16));
        bfrag[h] = __builtin_bit_cast(f16x8_t, u);
    }
    // this wave's pillar (selection phases): p = wid
    const bool has_p = wid < np;
    const float fc = s_f[wid * kC + lane];
    const float wm = wmax[lane], fa = fabsf(fc);
    float eps2 = 2.f * (hvpr_reduce_sum<64>(fmaf(kRelErr * wm, fa, kAbsErr * fa)) + kAbsErr * hvpr_reduce_sum<64>(wm) + 1e-30f);
    const float fmax_ = hvpr_reduce_max<64>(fa);
    const bool unfiltered = !(fmax_ <= kHalfMax) || !(hvpr_reduce_max<64>(wm) <= kHalfMax);     // outside the fp16 range
    float T = -3.4028235e38f;          // lower bound of the k-th largest A seen so far
    float Lk = -3.4028235e38f;         // k-th largest committed exact logit
    unsigned long long ck = 0ull;      // committed keys, lane r < k: the r-th best
    int lcnt = 0;                      // live entries of this pillar's candidate list
    unsigned long long *cand = s_cand + wid * kCap;
    const int n_tiles = (N + 15) >> 4, n_blocks = (N + kBlk - 1) / kBlk;
    float *const lrow = s_logit + l15 * kPitch + 4 * q;
    const int first = (int)((__builtin_amdgcn_readfirstlane((unsigned)wid) + blockIdx.x) % (unsigned)kWaves);

    for (int blk = 0; blk < n_blocks; ++blk) {
        const int n_here = min(kBlk, N - blk * kBlk);            // items of this block
        // ---- logits of the block into LDS (positions past n_here: -inf) ----
        if (n_here < kBlk)
            for (int i = tid; i < kPillars * (kBlk - n_here); i += kThreads)
                s_logit[(i / (kBlk - n_here)) * kPitch + n_here + i % (kBlk - n_here)] = -INFINITY;
        {
            constexpr int kMaxTiles = kBlk / 16 / kWaves;         // 8
            uint4 a[kMaxTiles][2];
#pragma unroll
            for (int i = 0; i < kMaxTiles; ++i) {
                const int t = blk * (kBlk / 16) + first + i * kWaves;
                if (t < n_tiles) {
                    a[i][0] = items_h[(size_t)t * 128 + lane];
                    a[i][1] = items_h[(size_t)t * 128 + 64 + lane];
                }
            }
#pragma unroll
            for (int i = 0; i < kMaxTiles; ++i) {
                const int tl = first + i * kWaves, t = blk * (kBlk / 16) + tl;
                if (t < n_tiles) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[i][0]), bfrag[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[i][1]), bfrag[1], acc, 0, 0, 0);
                    if (16 * tl + 16 <= n_here) {
                        *(float4 *)(lrow + tl * 16) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * tl + 4 * q + r < n_here) lrow[tl * 16 + r] = acc[r];
                    }
                }
            }
        }
        __syncthreads();
        // ---- this wave's pillar: candidates of the block ----
        if (has_p) {
            const float *row = s_logit + wid * kPitch;
            float v[kBlk / 64];
            float lmax = -INFINITY;
#pragma unroll
            for (int t = 0; t < kBlk / 64; ++t) { v[t] = row[lane + 64 * t]; lmax = fmaxf(lmax, v[t]); }
            unsigned hits = 0u;
            float thr = -3.4028235e38f;
            if (!unfiltered) {
                const float tau = ord_to_float(wave_kth_largest_hi16(ord_bits(lmax), k));     // NaN when the block has < k live lanes
                T = fmaxf(T, tau);                                                            // fmaxf drops a NaN
                thr = fmaxf(fmaxf(T - eps2, Lk - 0.5f * eps2), -3.4028235e38f);
#pragma unroll
                for (int t = 0; t < kBlk / 64; ++t) hits |= v[t] >= thr ? (1u << t) : 0u;
            } else {
#pragma unroll
                for (int t = 0; t < kBlk / 64; ++t) hits |= lane + 64 * t < n_here ? (1u << t) : 0u;
            }
            const int n_new = (int)hvpr_reduce_sum<64>((float)__popc(hits));               // exact: <= 2048
            // make room: drop what the risen bound left behind (unfiltered: the stored A may be NaN — nothing to prune by)
            if (lcnt + n_new > kCap && lcnt > 0 && !unfiltered) lcnt = prune(cand, lcnt, thr, lane);
            for (unsigned left = hits; __ballot(left != 0u) != 0ull;) {
                const bool has = left != 0u;
                const unsigned long long m = __ballot(has);
                if (lcnt + __popcll(m) > kCap) {       // full: settle the list's entries exactly
                    commit(items, fc, cand, lcnt, k, lane, ck, Lk);
                    lcnt = 0;
                }
                const int t = has ? __ffs((int)left) - 1 : 0;
                left &= left - 1u;
                if (has) cand[lcnt + __popcll(m & ((1ull << lane) - 1ull))] =
                    ((unsigned long long)ord_bits(row[lane + 64 * t]) << 32) | (unsigned)(blk * kBlk + lane + 64 * t);
                lcnt += __popcll(m);
            }
        }
        __syncthreads();            // the next block overwrites the logits
    }
    if (!has_p) return;
    int *out = idx_out + (size_t)(p0 + wid) * k;
    if (fmax_ == 0.f) {             // an all-zero pillar row: every logit is exactly 0, the rule gives the k lowest indices
        if (lane < k) out[lane] = lane;
        return;
    }
    if (!unfiltered) lcnt = prune(cand, lcnt, fmaxf(fmaxf(T - eps2, Lk - 0.5f * eps2), -3.4028235e38f), lane);
    commit(items, fc, cand, lcnt, k, lane, ck, Lk);
    if (lane < k) out[lane] = (int)(0xffffffffu - (unsigned)(ck & 0xffffffffull));       // k <= N: k keys exist
}

}  // namespace

extern "C" int hvpr_point_pillar_topk_f32(const float *pillars, int M, const float *points, const float *points_packed, int N, int k,
                                          int32_t *idx, hvpr_stream_t stream) {
    if (M < 0 || N < 1 || k < 1) return HVPR_ERR_INVALID_ARG;
    if (k > 32 || k > N) return HVPR_ERR_UNSUPPORTED;
    if (M == 0) return HVPR_OK;
    if (!pillars || !points || !points_packed || !idx) return HVPR_ERR_INVALID_ARG;
    const size_t lds = (size_t)kPillars * kPitch * 4 + kPillars * kC * 4 + kWaves * kCap * 8;
    static unsigned long long lds_set = 0ull;
    if (hvpr_ensure_dyn_lds((const void *)k_point_topk, (int)lds, &lds_set) != 0) return HVPR_ERR_LAUNCH;
    const int n_tiles = hvpr_cdiv(N, 16);
    hipLaunchKernelGGL(k_point_topk, dim3(hvpr_cdiv(M, kPillars)), dim3(kThreads), lds, (hipStream_t)stream, pillars, M, points,
                       (const uint4 *)points_packed, points_packed + (size_t)n_tiles * 512, N, k, idx);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
