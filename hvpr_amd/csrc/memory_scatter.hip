// a3 + a4 — MemAE memory read-out (eval) and the scatter of pillars to the dense BEV canvases.
//   a3 replaces MemoryUnit_Agg.forward eval branch, map_to_bev/memory_module.py:60-77
//   a4 replaces PointPillarScatter_Agg_Memory_1_scale.forward eval branch, map_to_bev/pointpillar_scatter.py:169-222
#include "common.h"
#include "internal.h"
#include "select.h"

namespace {

// ------------------------------------------------------------------------------------------------
// a3  memory read-out.  One workgroup = kPillars pillars x all n_items (<= 2048) memory items.
//
// The top-k of the EXACT fp32 logits is found with a half-precision pre-filter and an exact re-check, so that the matrix
// cores run at the fp16 rate (16x the fp32 MFMA rate) and the bank streams as 256 KB instead of 512 KB per workgroup:
//   1. A[p][j] ~ f[p] . bank[j] on v_mfma_f32_16x16x32_f16 (operands rounded to fp16, fp32 accumulate).  The logits never
//      leave the registers of the wave that computed them (round 4; rounds 1-3 kept 128 KB of rows in LDS and every pillar
//      wave read its row back): wave w owns tiles w, w + 16, ..., lane (p, q) the items 16 t + 4 q .. + 3 of pillar p.
//   2. Error bound per pillar: |A - L| <= eps_p for every item, L = the exact logit.  fp16 round-to-nearest errs by at most
//      2^-11 |x| for a normal result and by at most 2^-14 = 6.1e-5 when the result is subnormal (assumed flushed to zero by
//      the matrix cores: the weaker assumption), so with wmax_c = max_j |W_jc|
//         eps_p = 1.0e-3 * sum_c |f_c| wmax_c  +  6.2e-5 * (sum_c |f_c| + sum_c wmax_c)  (+ 1e-30)
//      ((1 + 2^-11)^2 - 1 = 0.00097680; the rest of 1.0e-3 covers the fp32 accumulation of both sides); a value beyond the
//      fp16 range makes eps_p infinite (everything becomes a candidate: exact slow path).
//      tau = a lower bound of the k-th largest A: the k-th largest (16 leading bits) of the 64 per-lane maxima of the pillar
//      (each the maximum of 32 items; they travel through 4 KB of LDS).  Every item of the exact top-k has
//      A >= tau - 2 eps_p:  k items have A >= tau, hence L >= tau - eps, so the k-th largest L is >= tau - eps, and an item
//      with L >= tau - eps has A >= tau - 2 eps.
//   3. Every wave compares its 32 accumulators per lane with its pillar's threshold (sign bits of A - threshold shifted into
//      one mask: two instructions per value) and appends its ~1.6 hits per pillar to the pillar's segment of a candidate
//      table in LDS; the order inside the table is a fixed function of the item ids (wave, quarter, tile), so a pillar's
//      result does not depend on its place in the batch.
//   4. One wave per pillar: the ~25 candidates get their exact logit, LANE = CANDIDATE: each lane reads its own bank row
//      (16 x 16 bytes) and runs a 64-term fma chain against the pillar's features (four partial sums) — no cross-lane
//      reduction, up to 64 candidates in one go.  top-k of (L desc, index asc) among the candidates = the exact top-k;
//      softmax over the selected L (memory_module.py:70-72 recomputes the same dot products); the selected (index, weight)
//      pairs are compacted to the first lanes and their rows summed with lane = channel.
// More than 64 candidates (mass ties, e.g. an all-zero feature row) takes an exact slow path over all items.
// ------------------------------------------------------------------------------------------------
constexpr int kC = 64;            // feature channels
constexpr int kPillars = 16;      // pillars per workgroup = the matrix-core columns of the logits step
constexpr int kItemsPad = 2048;   // most items
constexpr float kRelErr = 1.0e-3f, kAbsErr = 6.2e-5f, kHalfMax = 65000.f;    // see 2. above

using namespace hvpr_sel;

__device__ __forceinline__ float vmaxr(float a, float b) {   // v_max_f32 without the operand quieting fmaxf() adds
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
struct op_maxr { __device__ __forceinline__ float operator()(float a, float b) const { return vmaxr(a, b); } };

// WAVES = 16: one pillar per wave in the per-pillar steps (a 16th of the items per wave in the others); WAVES = 8: two pillars
// per wave in turn, 16 tiles per wave — half the workgroup, so that TWO independent workgroups share a CU and the memory phases of
// one fall into the compute phases of the other (the sixteen waves of one workgroup move in lockstep between its barriers)
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 4) k_memory_readout(const float *__restrict__ f, int M,
                                                             const int *__restrict__ m_device,
                                                             const float *__restrict__ bank,
                                                             const uint4 *__restrict__ bank_bf,
                                                             const float *__restrict__ wmax, int n_items, int k,
                                                             float *__restrict__ out, int *__restrict__ topk_idx,
                                                             const int4 *__restrict__ coords, int batch, int nx, int ny,
                                                             int *__restrict__ cell_map, float *__restrict__ canvas,
                                                             int canvas_channels, int canvas_offset) {
    // LDS pitches (round 6, SQ_LDS_BANK_CONFLICT 41 % -> see DESIGN.md §4.1): a pillar's feature row is kFP = 68 floats apart from the
    // next one (the 16 lanes of a b128 service group then land in distinct banks: with 64 they all hit the same four), the per-lane
    // maxima are stored [wave * 4 + quarter][pillar] with pitch 17, the hit counts [pillar][wave] with pitch 17, and a pillar's 16
    // candidate segments are kRowC = 196 ints apart from the next pillar's (16 x 12 + 4: the segments of the same wave of the 16
    // pillars start in 16 distinct banks instead of one)
    constexpr int kFP = kC + 4, kPmP = 17, kCntP = 17;
    __shared__ __attribute__((aligned(16))) float s_f[kPillars * kFP];
    __shared__ __attribute__((aligned(16))) float s_pm[64 * kPmP];           // per-lane maxima: [wave * 4 + quarter][pillar]
    __shared__ float s_tau[kPillars];
    // The items are dealt to 16 VIRTUAL waves (tile t belongs to virtual wave t mod 16, eight tiles each); a physical wave of the
    // 8-wave form plays two of them (v = wid, wid + 8).  Maxima, candidate segments and their order are those of the virtual waves,
    // so both forms select the same candidates in the same order: a pillar's result does not depend on the form that computed it.
    constexpr int kThreads = 64 * WAVES, kWaves = 16, kMaxTiles = 8, VPW = 16 / WAVES, PPW = kPillars / WAVES;
    constexpr int kSeg = 12;                        // candidate slots per (pillar, virtual wave); 1.6 expected, more than kSeg: slow path
    constexpr int kSlots = 4;                       // lanes per segment when the pillar's list is put together
    constexpr int kRowC = kWaves * kSeg + 4;        // ints between the candidate segments of consecutive pillars
    __shared__ int s_cnt[kPillars * kCntP];                                  // hits of wave w for pillar p: [p][w]
    // their item ids: [p][w][slot]; once every wave is through with them the same 12 KB hold the exact logits of ALL items of one
    // pillar (the rare exact path at the end)
    __shared__ union { int cand[kPillars * kRowC]; float logit[kItemsPad]; } s_u;
    int *const s_cand = s_u.cand;
    __shared__ int s_tot[kPillars], s_flag[kPillars];                        // candidates of a pillar over all waves; 1: no filtered path for it
    __shared__ int s_list[WAVES * 64];                                       // per physical wave: candidate ids of a round
    __shared__ __attribute__((aligned(16))) unsigned long long s_key[WAVES * (kPillars / WAVES) * 64];   // per wave and slot: their (order bits of the exact logit, ~id), zero padded
    __shared__ __attribute__((aligned(8))) int2 s_top[WAVES * 64];           // per physical wave: [slot][rank] (id, logit / weight) of the k selected
    if (m_device) M = min(M, *m_device);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p0 = blockIdx.x * kPillars;
    if (p0 >= M) return;
    const int np = min(kPillars, M - p0);

#ifdef HVPR_EXP_TIMING
#define RO_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ts[i] = __builtin_readcyclecounter(); } while (0)
    long long ts[12];
#else
#define RO_STAMP(i)
#endif
    RO_STAMP(0);
    // The bank tiles of the wave's first virtual wave do not depend on the features: in the 8-wave form both are requested before the
    // barrier — the features first (they return first and go to LDS while the tiles are still travelling) — so that the two round
    // trips are one (batch 16: 154 -> 150 us).
    constexpr int kFeat = kPillars * kC / kThreads;      // 1 (16 waves) or 2 (8 waves) feature values per thread
    float fv[kFeat];
#pragma unroll
    for (int j = 0; j < kFeat; ++j) {
        const int i = tid + j * kThreads, p = i / kC;
        fv[j] = p < np ? f[(size_t)(p0 + p) * kC + (i % kC)] : 0.f;
    }
    const int n_tiles = (n_items + 15) >> 4;
    uint4 a_first[kMaxTiles][2];
    constexpr bool kEarlyTiles = WAVES == 8;    // (the one-frame form measures 0.5 us SLOWER with the early request: 12.7 against 12.2 us)
    if (kEarlyTiles) {
#pragma unroll
        for (int i = 0; i < kMaxTiles; ++i) {
            const int t = wid + i * kWaves;
            if (t < n_tiles) {
                a_first[i][0] = bank_bf[(size_t)t * 128 + lane];
                a_first[i][1] = bank_bf[(size_t)t * 128 + 64 + lane];
            }
        }
    }
    const float wm = wmax[lane];
    const float wsum = wmax[kC], wtop = wmax[kC + 1];     // sum_c wmax_c, max_c wmax_c (k_wmax_stats)
#pragma unroll
    for (int j = 0; j < kFeat; ++j) s_f[((tid + j * kThreads) / kC) * kFP + (tid + j * kThreads) % kC] = fv[j];
#pragma unroll
    for (int h = 0; h < PPW; ++h) s_key[(wid * PPW + h) * 64 + lane] = 0ull;     // step 4's key lists end in zeros
    if (tid < kPillars) { s_tot[tid] = 0; s_flag[tid] = 0; }
    __syncthreads();
    RO_STAMP(1);

    // ---- step 1: A[p][j] on the fp16 matrix cores.  D (16 items x 16 pillars) = A (items x K) . B (K x pillars), K = 2 x 32
    // channels.  Lane (l15, q) holds channels 32h + 8q .. 32h + 8q + 7 of item / pillar l15 for both operands (the k index of
    // an MFMA is a free permutation as long as A and B agree).  B (the 16 pillars) stays in registers.
    const int l15 = lane & 15, q = lane >> 4;
    f32x4 acc[VPW][kMaxTiles];
    {
        f16x8_t bfrag[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float *fp = s_f + l15 * kFP + 32 * h + 8 * q;
            const float4 lo = *(const float4 *)fp, hi = *(const float4 *)(fp + 4);
            const unsigned w0 = f16_rne(lo.x) | (f16_rne(lo.y) << 16), w1 = f16_rne(lo.z) | (f16_rne(lo.w) << 16);
            const unsigned w2 = f16_rne(hi.x) | (f16_rne(hi.y) << 16), w3 = f16_rne(hi.z) | (f16_rne(hi.w) << 16);
            const uint4 u = make_uint4(w0, w1, w2, w3);
            bfrag[h] = __builtin_bit_cast(f16x8_t, u);
        }
        // the tiles of the packed bank ([tile][half][lane] 16 bytes: every load instruction reads 1 KB contiguous), the eight of a
        // virtual wave requested at once; one maximum per lane and virtual wave: 64 maxima per pillar
#pragma unroll
        for (int vw = 0; vw < VPW; ++vw) {
            const int v = wid + WAVES * vw;                                   // wave-uniform
            uint4 a[kMaxTiles][2];
#pragma unroll
            for (int i = 0; i < kMaxTiles; ++i) {
                const int t = v + i * kWaves;
                if (kEarlyTiles && vw == 0) {
                    a[i][0] = a_first[i][0];
                    a[i][1] = a_first[i][1];
                } else if (t < n_tiles) {
                    a[i][0] = bank_bf[(size_t)t * 128 + lane];
                    a[i][1] = bank_bf[(size_t)t * 128 + 64 + lane];
                }
            }
            float lmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < kMaxTiles; ++i) {
                const int t = v + i * kWaves;
                acc[vw][i] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};     // items past n_items never win
                if (t < n_tiles) {
                    f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[i][0]), bfrag[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[i][1]), bfrag[1], c, 0, 0, 0);
                    // C/D map of 16x16: column (pillar) = lane & 15, row (item) = 4 * (lane >> 4) + reg
                    if (16 * t + 16 > n_items) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * t + 4 * q + r >= n_items) c[r] = -INFINITY;
                    }
                    acc[vw][i] = c;
                    lmax = vmaxr(vmaxr(lmax, vmaxr(c[0], c[1])), vmaxr(c[2], c[3]));
                }
            }
            s_pm[(v * 4 + q) * kPmP + l15] = lmax;
        }
    }
    RO_STAMP(2);
    __syncthreads();
    RO_STAMP(3);

    // ---- step 2: wave w -> thresholds of its pillars w (, w + 8) ----
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = wid + WAVES * j;
        if (p < np) {
            const float fa = fabsf(s_f[p * kFP + lane]);                     // lane = channel
            // 2 eps_p: the bound on |A - L| of this pillar, doubled (see the header); infinite outside the fp16 range
            float term = fmaf(kRelErr * wm, fa, kAbsErr * fa);
            if (!(fa <= kHalfMax)) term = INFINITY;                          // (rides the sum: no second reduction)
            float eps2 = 2.f * (hvpr_reduce_sum<64>(term) + kAbsErr * wsum + 1e-30f);
            if (!(wtop <= kHalfMax)) eps2 = INFINITY;
            if (!(eps2 < INFINITY) && lane == 0) s_flag[p] = 1;      // outside the fp16 range: no pre-filter for this pillar
            // tau <= k-th largest of the 64 lane maxima (its 16 leading bits): a lower bound of the k-th largest A
            const float tau = ord_to_float(wave_kth_largest_hi16(ord_bits(s_pm[lane * kPmP + p]), k));
            // tau = -inf or eps2 = inf / NaN let every live item through; the -inf padding past n_items never passes
            if (lane == 0) s_tau[p] = fmaxf(tau - eps2, -3.4028235e38f);
        }
    }
    RO_STAMP(4);
    __syncthreads();
    RO_STAMP(5);

    // ---- step 3: every wave: which of its values per lane reach the threshold of the lane's pillar ----
#pragma unroll
    for (int vw = 0; vw < VPW; ++vw) {
        const int v = wid + WAVES * vw;
        const float tl = s_tau[l15];
        unsigned below = 0u;    // bit 31 - (4 i + r): A < threshold (the sign of the difference; -inf padding stays below)
#pragma unroll
        for (int i = 0; i < kMaxTiles; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) below = __builtin_amdgcn_alignbit(below, __float_as_uint(acc[vw][i][r] - tl), 31);
        unsigned hits = l15 < np ? ~below : 0u;
        const int mine = __popc(hits);
        // position inside the (pillar, virtual wave) segment: the lanes of the lower quarters first
        const int c1 = __shfl_up(mine, 16, 64), c2 = __shfl_up(mine, 32, 64), c3 = __shfl_up(mine, 48, 64);
        int pos = (q >= 1 ? c1 : 0) + (q >= 2 ? c2 : 0) + (q >= 3 ? c3 : 0);
        if (q == 3) {
            s_cnt[l15 * kCntP + v] = pos + mine;
            atomicAdd(&s_tot[l15], pos + mine);
            if (pos + mine > kSeg) s_flag[l15] = 1;
        }
        int *seg = s_cand + l15 * kRowC + v * kSeg;
        while (__ballot(hits != 0u) != 0ull) {
            if (hits != 0u) {
                const int b = __clz((int)hits);           // 4 i + r, ascending item order
                hits &= ~(0x80000000u >> b);
                if (pos < kSeg) seg[pos] = 16 * (v + (b >> 2) * kWaves) + 4 * q + (b & 3);
                ++pos;
            }
        }
    }
    RO_STAMP(6);
    __syncthreads();
    RO_STAMP(7);

    // ---- step 4: the per-pillar step, ONE instruction stream for the wave's pillars (WAVES = 8: two pillars together) ----
    // Pillar slot h of the wave is pillar wid + WAVES h.  The candidates of both slots share the wave's 64 candidate lanes (slot 0's
    // first) and get their exact logits from ONE batch of loads — four lanes per candidate, whose 16 bank-row values each stay in
    // registers; they are ranked inside their own pillar by counting (lane = candidate walks its pillar's zero-padded key list in
    // LDS, two keys per read: no per-pillar radix select on the scalar unit); the k best leave (id, logit) by rank in a table, the
    // softmax runs with lane = (slot, rank), and the weighted sum of the selected rows with lane = (slot, channel pair), in rank
    // order: a wave's chain has TWO memory round trips (candidate rows, selected rows) whatever PPW is.
    int *const list = s_list + wid * 64;                                     // candidate ids of the round, joint positions
    unsigned long long *const keys = s_key + wid * (PPW * 64);               // [slot][64] (order bits of L, ~id), zero padded
    int2 *const top = s_top + wid * 64;                                      // [slot][rank] -> (id, L bits, then weight bits)
    const int hrow = lane >> 4, hh = lane >> 5;
    // candidate counts: row h of the wave (lanes 16 h ..) looks at the 16 virtual-wave segments of slot h
    const int p_row = wid + WAVES * hrow;
    const int cw = (hrow < PPW && p_row < np) ? s_cnt[p_row * kCntP + (lane & 15)] : 0;
    // the pillars of the workgroup without a filtered path: the same word in every wave
    const unsigned slowm = (unsigned)__ballot(lane < np && (s_flag[lane & 15] != 0 || s_tot[lane & 15] > 64)) & 0xffffu;
    int pf = cw;                                          // inclusive prefix inside every 16-lane row
    pf += __builtin_amdgcn_update_dpp(0, pf, 0x111, 0xf, 0xf, true);
    pf += __builtin_amdgcn_update_dpp(0, pf, 0x112, 0xf, 0xf, true);
    pf += __builtin_amdgcn_update_dpp(0, pf, 0x114, 0xf, 0xf, true);
    pf += __builtin_amdgcn_update_dpp(0, pf, 0x118, 0xf, 0xf, true);
    int cnt_s[2];
    bool slow_s[2], live_s[2];
    int4 cpos_s[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        live_s[h] = h < PPW && wid + WAVES * h < np;
        cnt_s[h] = h < PPW ? __builtin_amdgcn_readlane(pf, 16 * h + 15) : 0;
        slow_s[h] = live_s[h] && ((slowm >> (wid + WAVES * h)) & 1u);
        // (the pillar's coordinates: requested here through the scalar unit, looked at only at the very end — off the chain)
        cpos_s[h] = make_int4(-1, 0, -1, -1);
        if ((cell_map || canvas) && live_s[h]) cpos_s[h] = coords[__builtin_amdgcn_readfirstlane(p0 + wid + WAVES * h)];
    }
    const bool joint = PPW == 2 && live_s[0] && live_s[1] && !slow_s[0] && !slow_s[1] && cnt_s[0] + cnt_s[1] <= 64;
    // softmax over the k selected exact logits of a round's slots, lane = (slot, rank): two DPP reductions inside the 32-lane
    // halves; the table then holds (id, weight), ranks past kk item 0 with weight 0
    auto softmax_ranks = [&](bool both, int sa, int kk_a, int kk_b) {
        const int r = lane & 31;
        const bool mine = both || hh == sa;                                // this half's slot belongs to the round
        const int kk = mine ? (hh == sa ? kk_a : kk_b) : 0;
        const int2 me = r < kk ? top[hh * 32 + r] : make_int2(0, 0);
        const float logit = r < kk ? __int_as_float(me.y) : -INFINITY;
        const float mx = hvpr_reduce<32>(logit, op_maxr());
        const float e = r < kk ? __expf(logit - mx) : 0.f;
        const float a = e / hvpr_reduce_sum<32>(e);
        if (mine) top[hh * 32 + r] = make_int2(me.x, __float_as_int(r < kk ? a : 0.f));
        if (topk_idx && r < kk) topk_idx[(size_t)(p0 + wid + WAVES * hh) * k + r] = me.x;
    };
    RO_STAMP(8);
#pragma unroll 1
    for (int round = 0; round < PPW; ++round) {            // ONE copy of the body; the slot's values are picked by scalar selects
        if (joint && round == 1) break;
        const int sa = round;                              // first (or only) slot of the round
        const bool live_a = sa ? live_s[1] : live_s[0], slow_a = sa ? slow_s[1] : slow_s[0];
        const int na = sa ? cnt_s[1] : cnt_s[0];
        if (!live_a || slow_a) continue;
        // (everything below is recomputed per round from this copy of the lane id: hoisted out of the loop, the addresses of
        // the body are spilled)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const bool both = joint;                           // the round also serves slot 1: its candidates follow slot 0's
        const int nb = both ? cnt_s[1] : 0, tot = na + nb;
        const int kk_a = min(k, na), kk_b = both ? min(k, nb) : 0;
        // the candidate lists: the 16 wave segments of a slot back to back, lane = (segment, slot in the segment)
        {
            const int w2 = ln / kSlots, s2 = ln % kSlots;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !both) break;
                const int h = sa + u, pp = wid + WAVES * h;
                const int cw2 = __shfl(cw, 16 * h + w2, 64), base2 = (u ? na : 0) + __shfl(pf - cw, 16 * h + w2, 64);
#pragma unroll
                for (int r = 0; r < kSeg / kSlots; ++r)
                    if (s2 + kSlots * r < cw2) list[base2 + s2 + kSlots * r] = s_cand[pp * kRowC + w2 * kSeg + s2 + kSlots * r];
            }
        }
        // lane c owns candidate c of the round
        const bool second = ln >= na;
        const int my_idx = ln < tot ? list[ln] : 0;
        // exact logits, FOUR LANES PER CANDIDATE: lane (c4, qq) reads channels 16 qq .. 16 qq + 15 of candidate 16 g + c4 (a lane
        // per candidate would touch 64 cache lines per load instruction), four fma chains each against the features of the
        // candidate's OWN pillar, the quad is summed by two DPP exchanges, then the sums move to lane = candidate: a candidate's
        // logit does not depend on where it sits.  Candidates 0..31 are requested at once, unconditionally; 32..63 only when
        // there are that many
        const int c4 = ln >> 2, qq = ln & 3;
        float4 rv[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 2 || tot > 32) {      // wave-uniform
                const int cc = 16 * g + c4;
                const int id = cc < tot ? list[cc] : 0;
                const float4 *rowp = (const float4 *)(bank + (size_t)id * kC + 16 * qq);
#pragma unroll
                for (int i = 0; i < 4; ++i) rv[g][i] = rowp[i];
            }
        }
        float L = -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 2 || tot > 32) {
                const int pcg = wid + WAVES * (sa + ((both && 16 * g + c4 >= na) ? 1 : 0));
                const float4 *fp = (const float4 *)(s_f + pcg * kFP + 16 * qq);
                float4 fq = fp[0];
                float a0 = rv[g][0].x * fq.x, a1 = rv[g][0].y * fq.y, a2 = rv[g][0].z * fq.z, a3 = rv[g][0].w * fq.w;
#pragma unroll
                for (int i = 1; i < 4; ++i) {
                    fq = fp[i];
                    a0 = fmaf(rv[g][i].x, fq.x, a0); a1 = fmaf(rv[g][i].y, fq.y, a1);
                    a2 = fmaf(rv[g][i].z, fq.z, a2); a3 = fmaf(rv[g][i].w, fq.w, a3);
                }
                float sq = (a0 + a1) + (a2 + a3);
                sq += hvpr_dpp<0xB1>(sq);    // quad_perm [1,0,3,2]
                sq += hvpr_dpp<0x4E>(sq);    // quad_perm [2,3,0,1]
                const float t = __shfl(sq, 4 * (ln & 15), 64);
                if ((ln >> 4) == g) L = t;
            }
        }
        // rank inside the pillar by (L desc, id asc): the number of its candidates that come before this one.  Every slot has
        // its own key list, zero beyond its end (zero is below every key): all lanes walk max(na, nb) entries, two per read
        const unsigned long long key = ((unsigned long long)ord_bits(L) << 32) | (unsigned)(0xffffffffu - (unsigned)my_idx);
        unsigned long long *const kp = keys + 64 * (sa + ((both && second) ? 1 : 0));
        if (ln < tot) kp[ln - (second ? na : 0)] = key;
        int rank = 0;
        const int lim = na > nb ? na : nb;                 // (lists are 64 long: walking on to a multiple of 8 reads zeros)
        for (int j = 0; j < lim; j += 8) {                 // four reads in flight per trip: a lone wave waits for LDS, not for issue
            const ulonglong2 k0 = *(const ulonglong2 *)(kp + j), k1 = *(const ulonglong2 *)(kp + j + 2);
            const ulonglong2 k2 = *(const ulonglong2 *)(kp + j + 4), k3 = *(const ulonglong2 *)(kp + j + 6);
            rank += (k0.x > key ? 1 : 0) + (k0.y > key ? 1 : 0) + (k1.x > key ? 1 : 0) + (k1.y > key ? 1 : 0);
            rank += (k2.x > key ? 1 : 0) + (k2.y > key ? 1 : 0) + (k3.x > key ? 1 : 0) + (k3.y > key ? 1 : 0);
        }
        if (ln < tot && rank < (second ? kk_b : kk_a)) top[(sa + (second ? 1 : 0)) * 32 + rank] = make_int2(my_idx, __float_as_int(L));
        softmax_ranks(both, sa, kk_a, kk_b);
    }
    // ---- the rare pillars that cannot take the filtered path (mass ties, values outside the fp16 range, more than 64 candidates or
    // more than kSeg in one segment), handled by the WHOLE workgroup one after the other: every wave knows the same set (flags
    // and totals in LDS since the last barrier), so the barriers below are taken by all waves or by none.  All threads put the
    // exact logit of EVERY item (four partial sums, one thread per item) into the 12 KB the candidate segments no longer need; the
    // pillar's own wave then takes k rounds of arg-max with (value desc, index asc) order, winner r to rank r of its table.
    // (Inline in the common path, this path's 32 registers of logits per lane cost the kernel ~200 B of scratch per lane — and
    // a read-out that uses scratch at all runs 25 % slower: 171 against 135 us at batch 16.) ----
    if (slowm != 0u) {
        __syncthreads();                                   // every wave is through with the candidate segments
        for (unsigned left = slowm; left != 0u; left &= left - 1u) {
            const int p = __ffs((int)left) - 1;
            for (int j = tid; j < kItemsPad; j += kThreads) {
                float sacc = -INFINITY;
                if (j < n_items) {
                    const float4 *rowp = (const float4 *)(bank + (size_t)j * kC);
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                    for (int i = 0; i < kC / 4; ++i) {
                        const float4 rw = rowp[i];
                        const float4 fv4 = *(const float4 *)(s_f + p * kFP + 4 * i);
                        a0 = fmaf(rw.x, fv4.x, a0); a1 = fmaf(rw.y, fv4.y, a1); a2 = fmaf(rw.z, fv4.z, a2); a3 = fmaf(rw.w, fv4.w, a3);
                    }
                    sacc = (a0 + a1) + (a2 + a3);
                }
                s_u.logit[j] = sacc;
            }
            __syncthreads();
            if (wid == p % WAVES) {
                const int sa = p / WAVES;
                unsigned long long prev = ~0ull;
                for (int r = 0; r < k; ++r) {
                    unsigned long long best = 0ull;
#pragma unroll 4
                    for (int t = 0; t < kItemsPad / 64; ++t) {
                        const unsigned long long c = ((unsigned long long)ord_bits(s_u.logit[64 * t + lane]) << 32) |
                                                     (unsigned)(0xffffffffu - (unsigned)(lane + 64 * t));
                        if (c < prev && c > best && lane + 64 * t < n_items) best = c;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffull), o, 64);
                        const unsigned hi = __shfl_xor((unsigned)(best >> 32), o, 64);
                        const unsigned long long ob = ((unsigned long long)hi << 32) | lo;
                        best = ob > best ? ob : best;
                    }
                    if (lane == 0)
                        top[sa * 32 + r] = make_int2((int)(0xffffffffu - (unsigned)(best & 0xffffffffull)),
                                                     __float_as_int(ord_to_float((unsigned)(best >> 32))));
                    prev = best;
                }
                softmax_ranks(false, sa, k, 0);
            }
            __syncthreads();                               // the buffer is free for the next such pillar
        }
    }
    RO_STAMP(9);
    // ---- weighted sum of the selected rows for all of the wave's slots at once, lane = (slot, channel pair), in rank order (a
    // fixed function of the pillar: not of its place in the wave or the batch); all loads in flight at once (ranks past k carry
    // item 0 with weight 0; rows past 16 / 24 are skipped as a block) ----
    if (PPW == 1) {
        // one pillar per wave: lane = channel; the (id, weight) pairs sit in lanes 0..31 and reach the loads / fmas through the
        // scalar unit.  Per channel the same sequence of fmas, in rank order, as the two-pillar form below: the same bits
        if (live_s[0]) {
            const int2 mine = top[lane & 31];
            float rows[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) rows[r] = bank[(size_t)__builtin_amdgcn_readlane(mine.x, r) * kC + lane];
            if (k > 16) {
#pragma unroll
                for (int r = 16; r < 24; ++r) rows[r] = bank[(size_t)__builtin_amdgcn_readlane(mine.x, r) * kC + lane];
            }
            if (k > 24) {
#pragma unroll
                for (int r = 24; r < 32; ++r) rows[r] = bank[(size_t)__builtin_amdgcn_readlane(mine.x, r) * kC + lane];
            }
            float o = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) o = fmaf(__int_as_float(__builtin_amdgcn_readlane(mine.y, r)), rows[r], o);
            if (k > 16) {
#pragma unroll
                for (int r = 16; r < 24; ++r) o = fmaf(__int_as_float(__builtin_amdgcn_readlane(mine.y, r)), rows[r], o);
            }
            if (k > 24) {
#pragma unroll
                for (int r = 24; r < 32; ++r) o = fmaf(__int_as_float(__builtin_amdgcn_readlane(mine.y, r)), rows[r], o);
            }
            out[(size_t)(p0 + wid) * kC + lane] = o;
            const int4 c = cpos_s[0];
            long long cell = -1;
            if ((unsigned)c.x < (unsigned)batch && (unsigned)c.z < (unsigned)ny && (unsigned)c.w < (unsigned)nx)
                cell = ((long long)c.x * ny + c.z) * nx + c.w;
            if (cell_map && lane == 0 && cell >= 0) cell_map[cell] = p0 + wid;
            if (canvas && cell >= 0) canvas[(size_t)cell * canvas_channels + canvas_offset + lane] = o;
        }
    } else
    {
        const bool half_live = hh < PPW && wid + WAVES * hh < np;
        const int cp2 = 2 * (lane & 31);
        const int2 *tp = top + hh * 32;
        float2 rows[32];
        if (half_live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rows[r] = *(const float2 *)(bank + (size_t)tp[r].x * kC + cp2);
            if (k > 16) {
#pragma unroll
                for (int r = 16; r < 24; ++r) rows[r] = *(const float2 *)(bank + (size_t)tp[r].x * kC + cp2);
            }
            if (k > 24) {
#pragma unroll
                for (int r = 24; r < 32; ++r) rows[r] = *(const float2 *)(bank + (size_t)tp[r].x * kC + cp2);
            }
            float ox = 0.f, oy = 0.f;     // (the weights are read again from the table: they would cost 32 registers next to the rows)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float a = __int_as_float(tp[r].y); ox = fmaf(a, rows[r].x, ox); oy = fmaf(a, rows[r].y, oy); }
            if (k > 16) {
#pragma unroll
                for (int r = 16; r < 24; ++r) { const float a = __int_as_float(tp[r].y); ox = fmaf(a, rows[r].x, ox); oy = fmaf(a, rows[r].y, oy); }
            }
            if (k > 24) {
#pragma unroll
                for (int r = 24; r < 32; ++r) { const float a = __int_as_float(tp[r].y); ox = fmaf(a, rows[r].x, ox); oy = fmaf(a, rows[r].y, oy); }
            }
            const int pp = wid + WAVES * hh;
            *(float2 *)(out + (size_t)(p0 + pp) * kC + cp2) = make_float2(ox, oy);
            // fused encode path: the memory channels of this pillar's cell, straight into the pre-cleared NHWC canvas
            // fused a3+a4: this slot's BEV cell (pointpillar_scatter.py:192, nz == 1)
            const int4 c = hh ? cpos_s[1] : cpos_s[0];
            long long cell = -1;
            if ((unsigned)c.x < (unsigned)batch && (unsigned)c.z < (unsigned)ny && (unsigned)c.w < (unsigned)nx)
                cell = ((long long)c.x * ny + c.z) * nx + c.w;
            if (cell_map && (lane & 31) == 0 && cell >= 0) cell_map[cell] = p0 + pp;   // gather-form scatter: k_cell_map's job
            if (canvas && cell >= 0) *(float2 *)(canvas + (size_t)cell * canvas_channels + canvas_offset + cp2) = make_float2(ox, oy);
        }
    }
#ifdef HVPR_EXP_TIMING
    RO_STAMP(10);
    if ((blockIdx.x == 3 || blockIdx.x == 100) && lane == 0 && (wid & 7) == 0)
        printf("readout wg %d wave %d: features %lld | logits %lld | bar %lld | tau %lld | bar %lld | hits %lld | bar %lld | counts %lld (cands %d + %d%s) | "
               "rounds %lld | rows %lld cycles\n", (int)blockIdx.x, wid, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3],
               ts[5] - ts[4], ts[6] - ts[5], ts[7] - ts[6], ts[8] - ts[7], cnt_s[0], cnt_s[1], joint ? " joint" : "", ts[9] - ts[8], ts[10] - ts[9]);
#endif
}

// ------------------------------------------------------------------------------------------------
// a4  scatter, gather formulation: every canvas element is written exactly once, coalesced, zeros
// included (no separate 47 MB memset, no strided column writes).  cell_map[b][y][x] = pillar row or -1;
// it is idle (-1 everywhere) between calls: k_scatter restores each entry after reading it.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cell_map(const int4 *__restrict__ coords, int M, const int *__restrict__ m_device,
                                                  int batch, int nx, int ny, int *__restrict__ cell_map) {
    if (m_device) M = min(M, *m_device);
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int4 c = coords[m];   // b, z, y, x
    if ((unsigned)c.x < (unsigned)batch && (unsigned)c.z < (unsigned)ny && (unsigned)c.w < (unsigned)nx)
        cell_map[((size_t)c.x * ny + c.z) * nx + c.w] = m;   // pointpillar_scatter.py:192 (nz == 1)
}

template <int CP, int CM, int CS>
__global__ void __launch_bounds__(256) k_scatter(const float *__restrict__ pillar, const float *__restrict__ memory,
                                                 const float *__restrict__ scale, long long n_cells,
                                                 int *__restrict__ cell_map, float *__restrict__ spatial,
                                                 float *__restrict__ spatial_scale) {
    constexpr int C = CP + CM;            // main canvas channels
    constexpr int V = C / 4;              // float4 per cell
    constexpr int CELLS = 8;              // cells per wave step
    static_assert(64 % V == 0 || V % 64 == 0, "channel count must tile a wave");
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long c0 = wave * CELLS; c0 < n_cells; c0 += n_waves * CELLS) {
        int mine = -1;
        if (lane < CELLS && c0 + lane < n_cells) mine = cell_map[c0 + lane];
        // main canvas: CELLS * V float4 in cell-major order
#pragma unroll
        for (int i = lane; i < CELLS * V; i += 64) {
            const int cell = i / V, q = i % V;
            const int m = __shfl(mine, cell, 64);
            if (c0 + cell < n_cells) {
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m >= 0) {
                    if (q * 4 < CP) val = *(const float4 *)(pillar + (size_t)m * CP + q * 4);
                    else if (CM > 0) val = *(const float4 *)(memory + (size_t)m * CM + (q * 4 - CP));
                }
                *(float4 *)(spatial + (size_t)(c0 + cell) * C + q * 4) = val;
            }
        }
        if (CS > 0) {
            constexpr int VS = CS / 4 > 0 ? CS / 4 : 1;
#pragma unroll
            for (int i = lane; i < CELLS * VS; i += 64) {
                const int cell = i / VS, q = i % VS;
                const int m = __shfl(mine, cell, 64);
                if (c0 + cell < n_cells) {
                    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m >= 0) val = *(const float4 *)(scale + (size_t)m * CS + q * 4);
                    *(float4 *)(spatial_scale + (size_t)(c0 + cell) * CS + q * 4) = val;
                }
            }
        }
        if (mine >= 0) cell_map[c0 + lane] = -1;   // back to idle
    }
}

template <int CP, int CM, int CS>
int launch_scatter(const float *pillar, const float *memory, const float *scale, long long n_cells, int *cell_map,
                   float *spatial, float *spatial_scale, hipStream_t s) {
    long long blocks = (n_cells + 8 * 4 - 1) / (8 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((k_scatter<CP, CM, CS>), dim3((unsigned)blocks), dim3(256), 0, s, pillar, memory, scale, n_cells,
                       cell_map, spatial, spatial_scale);
    return 0;
}

}  // namespace

namespace {
int launch_canvas(const float *, int, const float *, int, const float *, int, long long, int *, float *, float *, hipStream_t);
}  // namespace

int hvpr_i_readout(const float *f, int M, const int32_t *m_device, const float *bank, const float *bank_packed, int n_items, int k, float *out,
                   int32_t *topk_idx, const int32_t *coords, int batch, int nx, int ny, int *cell_map, float *canvas,
                   int canvas_channels, int canvas_offset, hipStream_t stream) {
    if (M < 0 || n_items < 1 || k < 1) return HVPR_ERR_INVALID_ARG;
    if (k > 32 || k > n_items || n_items > kItemsPad) return HVPR_ERR_UNSUPPORTED;
    if (M == 0) return HVPR_OK;
    if (!f || !bank || !bank_packed || !out) return HVPR_ERR_INVALID_ARG;   // the packed copy is required (fp16 tiles + channel maxima)
    if ((cell_map || canvas) && !coords) return HVPR_ERR_INVALID_ARG;
    if (canvas && canvas_offset + kC > canvas_channels) return HVPR_ERR_INVALID_ARG;
    const int n_tiles = hvpr_cdiv(n_items, 16);
    const uint4 *bank_bf = (const uint4 *)bank_packed;
    const float *wmax = bank_packed + (size_t)n_tiles * 512;      // 2 KB of fp16 per tile = 512 floats
    // one round of workgroups (a frame or two): 16 waves, one pillar per wave — the shortest chain (12.0 us at batch 1; the 8-wave form:
    // 18.6); beyond that the 8-wave form, two independent workgroups per CU: 157 / 575 us against 162 / 587 at batch 16 / dense scene
    if (M <= kPillars * 1024)      // (M is the capacity: one frame of 16 384 points at most)
        hipLaunchKernelGGL(k_memory_readout<16>, dim3(hvpr_cdiv(M, kPillars)), dim3(1024), 0, stream, f, M, m_device, bank, bank_bf, wmax,
                           n_items, k, out, topk_idx, (const int4 *)coords, batch, nx, ny, cell_map, canvas, canvas_channels, canvas_offset);
    else
        hipLaunchKernelGGL(k_memory_readout<8>, dim3(hvpr_cdiv(M, kPillars)), dim3(512), 0, stream, f, M, m_device, bank, bank_bf, wmax,
                           n_items, k, out, topk_idx, (const int4 *)coords, batch, nx, ny, cell_map, canvas, canvas_channels, canvas_offset);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

namespace {
int launch_readout(const float *f, int M, const int32_t *m_device, const float *bank, const float *bank_packed, int n_items, int k, float *out,
                   int32_t *topk_idx, const int32_t *coords, int batch, int nx, int ny, int *cell_map, hvpr_stream_t stream) {
    return hvpr_i_readout(f, M, m_device, bank, bank_packed, n_items, k, out, topk_idx, coords, batch, nx, ny, cell_map, nullptr, 0, 0,
                          (hipStream_t)stream);
}
}  // namespace

// bank (n_items, 64) row-major fp32 -> the read-out kernel's streaming copy:
//   [ceil(n_items / 16) tiles][2 channel halves][64 lanes] x 8 fp16 (round to nearest even; rows past n_items zero): lane
//   (l15, q) of half h holds channels 32h + 8q .. + 7 of item 16 * tile + l15 — the A operand of v_mfma_f32_16x16x32_f16,
//   1 KB contiguous per load instruction; followed by wmax[64] fp32 = max_j |bank[j][c]| (the pre-filter's error bound), their sum and their maximum.
__global__ void __launch_bounds__(256) k_bank_pack(const float *__restrict__ bank, int n_items, uint4 *__restrict__ packed, int n_out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    const int t = o >> 7, h = (o >> 6) & 1, ln = o & 63, row = 16 * t + (ln & 15), c0 = 32 * h + 8 * (ln >> 4);
    uint4 u = make_uint4(0u, 0u, 0u, 0u);
    if (row < n_items) {
        const float4 lo = *(const float4 *)(bank + (size_t)row * kC + c0), hi = *(const float4 *)(bank + (size_t)row * kC + c0 + 4);
        u = make_uint4(f16_rne(lo.x) | (f16_rne(lo.y) << 16), f16_rne(lo.z) | (f16_rne(lo.w) << 16),
                       f16_rne(hi.x) | (f16_rne(hi.y) << 16), f16_rne(hi.z) | (f16_rne(hi.w) << 16));
    }
    packed[o] = u;
}

// wmax[c] = max_j |bank[j][c]|.  Workgroup b covers items b*16 + part, stride 16 * gridDim.x; the workgroups combine through an
// atomic max on the bit patterns (non-negative floats order like unsigned integers; max is order-independent: deterministic).
// wmax must be zero on entry (k_wmax_zero).
__global__ void __launch_bounds__(64) k_wmax_zero(float *__restrict__ wmax) { wmax[threadIdx.x] = 0.f; }

__global__ void __launch_bounds__(1024) k_bank_wmax(const float *__restrict__ bank, int n_items, float *__restrict__ wmax) {
    __shared__ float s[16][kC];
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
    float m = 0.f;
    // eight independent loads per round (a one-load-per-iteration loop is one L2 round trip per item: 24 us for 2000 items)
    const int step = 16 * (int)gridDim.x;
    for (int j0 = (int)blockIdx.x * 16 + part; j0 < n_items; j0 += step * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = j0 + step * u < n_items ? fabsf(bank[(size_t)(j0 + step * u) * kC + c]) : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
    }
    s[part][c] = m;
    __syncthreads();
    if (part == 0) {
        for (int i = 1; i < 16; ++i) m = fmaxf(m, s[i][c]);
        // NaN rows propagate nothing here (fmaxf drops NaN): a NaN bank gives NaN logits either way
        atomicMax((unsigned *)wmax + c, __float_as_uint(m));
    }
}

// wmax[64] = sum_c wmax[c], wmax[65] = max_c wmax[c]: the same for every pillar, so not the read-out's work
__global__ void __launch_bounds__(64) k_wmax_stats(float *__restrict__ wmax) {
    const float wm = wmax[threadIdx.x];
    const float wsum = hvpr_reduce_sum<64>(wm), wtop = hvpr_reduce<64>(wm, op_maxr());
    if (threadIdx.x == 0) { wmax[kC] = wsum; wmax[kC + 1] = wtop; }
}

extern "C" size_t hvpr_memory_bank_packed_floats(int n_items) { return n_items < 1 ? 0 : (size_t)hvpr_cdiv(n_items, 16) * 512 + 2 * kC; }

extern "C" int hvpr_memory_bank_pack_f32(const float *bank, int n_items, float *packed, hvpr_stream_t stream) {
    if (!bank || !packed || n_items < 1) return HVPR_ERR_INVALID_ARG;
    const int n_tiles = hvpr_cdiv(n_items, 16), n_out = n_tiles * 128;
    hipLaunchKernelGGL(k_bank_pack, dim3(hvpr_cdiv(n_out, 256)), dim3(256), 0, (hipStream_t)stream, bank, n_items, (uint4 *)packed, n_out);
    float *wmax = packed + (size_t)n_tiles * 512;
    hipLaunchKernelGGL(k_wmax_zero, dim3(1), dim3(kC), 0, (hipStream_t)stream, wmax);
    const int wg = n_items <= 2048 ? 1 : (n_items < 128 * 128 ? hvpr_cdiv(n_items, 128) : 128);   // >= 8 items per wave slot
    hipLaunchKernelGGL(k_bank_wmax, dim3(wg), dim3(1024), 0, (hipStream_t)stream, bank, n_items, wmax);
    hipLaunchKernelGGL(k_wmax_stats, dim3(1), dim3(kC), 0, (hipStream_t)stream, wmax);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_memory_readout_fwd_f32(const float *f, int M, const int32_t *m_device, const float *bank, const float *bank_packed,
                                           int n_items, int k, float *out, int32_t *topk_idx, hvpr_stream_t stream) {
    return launch_readout(f, M, m_device, bank, bank_packed, n_items, k, out, topk_idx, nullptr, 0, 0, 0, nullptr, stream);
}

extern "C" size_t hvpr_scatter_workspace_bytes(int batch, int nx, int ny) {
    if (batch < 1 || nx < 1 || ny < 1) return 0;
    return (size_t)batch * nx * ny * sizeof(int);
}

extern "C" int hvpr_scatter_bev_fwd_f32(const float *pillar_features, int c_pillar, const float *memory_features,
                                        int c_mem, const float *scale_features, int c_scale, const int32_t *coords, int M,
                                        const int32_t *m_device, int batch, int nx, int ny, float *spatial,
                                        float *spatial_scale, void *workspace, size_t workspace_bytes,
                                        hvpr_stream_t stream) {
    if (M < 0 || batch < 1 || nx < 1 || ny < 1 || !spatial || !workspace) return HVPR_ERR_INVALID_ARG;
    if (M > 0 && (!pillar_features || !coords)) return HVPR_ERR_INVALID_ARG;
    if ((c_mem > 0 && M > 0 && !memory_features) || (c_scale > 0 && (!spatial_scale || (M > 0 && !scale_features))))
        return HVPR_ERR_INVALID_ARG;
    if (workspace_bytes < hvpr_scatter_workspace_bytes(batch, nx, ny)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int *cell_map = (int *)workspace;
    const long long n_cells = (long long)batch * nx * ny;
    if (M > 0) hipLaunchKernelGGL(k_cell_map, dim3(hvpr_cdiv(M, 256)), dim3(256), 0, s, (const int4 *)coords, M, m_device,
                                  batch, nx, ny, cell_map);
    return launch_canvas(pillar_features, c_pillar, memory_features, c_mem, scale_features, c_scale, n_cells, cell_map, spatial,
                         spatial_scale, s);
}

extern "C" int hvpr_memory_scatter_fwd_f32(const float *pillar_features, const float *scale_features, const int32_t *coords,
                                           int M, const int32_t *m_device, const float *bank, const float *bank_packed, int n_items, int k, int batch,
                                           int nx, int ny, float *memory_features, float *spatial, float *spatial_scale,
                                           void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (M < 0 || batch < 1 || nx < 1 || ny < 1 || !spatial || !spatial_scale || !workspace) return HVPR_ERR_INVALID_ARG;
    if (M > 0 && (!pillar_features || !scale_features || !coords || !memory_features)) return HVPR_ERR_INVALID_ARG;
    if (workspace_bytes < hvpr_scatter_workspace_bytes(batch, nx, ny)) return HVPR_ERR_WORKSPACE;
    int *cell_map = (int *)workspace;
    const int st = launch_readout(pillar_features, M, m_device, bank, bank_packed, n_items, k, memory_features, nullptr, coords, batch, nx,
                                  ny, cell_map, stream);
    if (st != HVPR_OK) return st;
    return launch_canvas(pillar_features, 64, memory_features, 64, scale_features, 32, (long long)batch * nx * ny, cell_map,
                         spatial, spatial_scale, (hipStream_t)stream);
}

namespace {
int launch_canvas(const float *pillar_features, int c_pillar, const float *memory_features, int c_mem,
                  const float *scale_features, int c_scale, long long n_cells, int *cell_map, float *spatial,
                  float *spatial_scale, hipStream_t s) {
    if (c_pillar == 64 && c_mem == 64 && c_scale == 32)
        launch_scatter<64, 64, 32>(pillar_features, memory_features, scale_features, n_cells, cell_map, spatial, spatial_scale, s);
    else if (c_pillar == 64 && c_mem == 0 && c_scale == 0)
        launch_scatter<64, 0, 0>(pillar_features, nullptr, nullptr, n_cells, cell_map, spatial, nullptr, s);
    else if (c_pillar == 64 && c_mem == 64 && c_scale == 0)
        launch_scatter<64, 64, 0>(pillar_features, memory_features, nullptr, n_cells, cell_map, spatial, nullptr, s);
    else
        return HVPR_ERR_UNSUPPORTED;
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
}  // namespace
