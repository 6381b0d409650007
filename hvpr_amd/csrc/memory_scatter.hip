// a3 + a4 — MemAE memory read-out (eval) and the scatter of pillars to the dense BEV canvases.
//   a3 replaces MemoryUnit_Agg.forward eval branch, map_to_bev/memory_module.py:60-77
//   a4 replaces PointPillarScatter_Agg_Memory_1_scale.forward eval branch, map_to_bev/pointpillar_scatter.py:169-222
#include "common.h"
#include "internal.h"

namespace {

// ------------------------------------------------------------------------------------------------
// a3  memory read-out.  One workgroup = kPillars pillars; the kPillars x n_items logit rows (fp32 MFMA) live in LDS
// (never in HBM: the reference materialises M x 2000 floats twice).  Selection per pillar (one wave):
//   lane-local max of the lane's 32 logits -> 64-lane bitonic sort -> tau = k-th largest lane max, a
//   lower bound of the k-th largest logit -> the expected ~23 logits >= tau are compacted and sorted
//   (value desc, index asc) -> top k.  More than 64 candidates (mass ties) takes an exact slow path.
// softmax(f . W[idx]) reuses the selected logits (memory_module.py:70-72 recomputes the same dot products).
// ------------------------------------------------------------------------------------------------
constexpr int kC = 64;          // feature channels
constexpr int kPillars = 16;    // pillars per workgroup
constexpr int kItemsPad = 2048; // logits row length in LDS
constexpr int kThreads = 1024;  // 16 waves: one pillar per wave in the selection phase, 4 waves per SIMD hide the bank stream latency

__device__ __forceinline__ unsigned ord_bits(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// k-th largest (k >= 1) of one key per lane: the largest v with count(key >= v) >= k, found bit by bit with ballots —
// compares and scalar popcounts only, no cross-lane data movement (a 64-lane bitonic sort costs 42 dependent
// ds_bpermute round trips).  Lanes that do not take part pass key 0.
__device__ __forceinline__ unsigned wave_kth_largest_u32(unsigned key, int k) {
    unsigned prefix = 0u;
#pragma unroll
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        if (__popcll(__ballot(key >= cand)) >= k) prefix = cand;
    }
    return prefix;
}
// the same on the 16 most significant bits only: a LOWER bound of the k-th largest key, for half the steps
__device__ __forceinline__ unsigned wave_kth_largest_hi16(unsigned key, int k) {
    unsigned prefix = 0u;
#pragma unroll
    for (int bit = 31; bit >= 16; --bit) {
        const unsigned cand = prefix | (1u << bit);
        if (__popcll(__ballot(key >= cand)) >= k) prefix = cand;
    }
    return prefix;
}
__device__ __forceinline__ unsigned long long wave_kth_largest_u64(unsigned long long key, int k) {
    unsigned long long prefix = 0ull;
#pragma unroll
    for (int bit = 63; bit >= 0; --bit) {
        const unsigned long long cand = prefix | (1ull << bit);
        if (__popcll(__ballot(key >= cand)) >= k) prefix = cand;
    }
    return prefix;
}

// sort 64 keys descending across the wave
__device__ __forceinline__ unsigned long long bitonic64_desc_u64(unsigned long long v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffull), j, 64);
            const unsigned hi = __shfl_xor((unsigned)(v >> 32), j, 64);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            const bool up = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            v = (lower == up) ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    return v;
}

__global__ void __launch_bounds__(kThreads) k_memory_readout(const float *__restrict__ f, int M,
                                                             const int *__restrict__ m_device,
                                                             const float *__restrict__ bank,
                                                             const float *__restrict__ bank_packed, int n_items, int k,
                                                             float *__restrict__ out, int *__restrict__ topk_idx,
                                                             const int4 *__restrict__ coords, int batch, int nx, int ny,
                                                             int *__restrict__ cell_map, float *__restrict__ canvas,
                                                             int canvas_channels, int canvas_offset) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_logit = (float *)smem;                              // [kPillars][kItemsPad]
    float *s_f = s_logit + kPillars * kItemsPad;                 // [kPillars][kC]
    unsigned long long *s_cand = (unsigned long long *)(s_f + kPillars * kC);   // [waves][64]
    if (m_device) M = min(M, *m_device);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p0 = blockIdx.x * kPillars;
    if (p0 >= M) return;
    const int np = min(kPillars, M - p0);

#ifdef HVPR_EXP_TIMING
    const long long tt0 = __builtin_readcyclecounter();
#endif
    for (int i = tid; i < kPillars * kC; i += kThreads) {
        const int p = i / kC;
        s_f[i] = p < np ? f[(size_t)(p0 + p) * kC + (i % kC)] : 0.f;
    }
    __syncthreads();

    // ---- phase 1: logits[p][j] = f[p] . bank[j] on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) ----
    // D (16 items x 16 pillars) = A (items x K) . B (K x pillars), K = 64 channels.  Lane quarter q feeds channels
    // 16g+4q .. 16g+4q+3 of group g to four consecutive MFMAs from ONE 16-byte load per operand (the MFMA k index is a
    // free permutation as long as A and B agree).  B (the 16 pillars) stays in registers for the whole kernel; A (the
    // bank, 512 KB, L2 resident) is streamed 16 items at a time, next tile prefetched while the current one multiplies.
    {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int l15 = lane & 15, q = lane >> 4;
        float4 bf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bf[g] = *(const float4 *)(s_f + l15 * kC + 16 * g + 4 * q);
        // items past n_items never win: their logits are -inf (written once, not tested per tile)
        for (int i = tid; i < kPillars * (kItemsPad - n_items); i += kThreads)
            s_logit[(i / (kItemsPad - n_items)) * kItemsPad + n_items + i % (kItemsPad - n_items)] = -INFINITY;
        // The fp32 MFMA runs on the vector ALU lanes: every VALU instruction of a wave costs its SIMD one MFMA slot, and four
        // waves share a SIMD.  The tile loop therefore carries no vector arithmetic: the tile index is a scalar (SGPR base
        // + fixed per-lane offset addressing), registers ping-pong instead of being copied, bounds are handled outside.
        const unsigned wave_u = __builtin_amdgcn_readfirstlane((unsigned)wid);
        constexpr int kWaves = kThreads / 64;
        const int n_full = n_items / 16;                           // whole 16-item tiles
        // This lane's float offset inside a 16-item tile (1024 floats in both layouts) and the step between its four 16-byte
        // pieces.  Row-major bank: 16 rows x 64 contiguous bytes per load instruction (half cache lines).  Packed bank
        // (hvpr_memory_bank_pack_f32: [tile][piece][lane] float4): every load instruction reads 1 KB contiguous — the same
        // bytes, but the logits phase drops from 36.5 k to 24 k cycles (the bank streams out of L2 in full-line requests).
        const int lane_elem = bank_packed ? 4 * l15 + 64 * q : l15 * kC + 4 * q;
        const int piece = bank_packed ? 256 : 16;
        float *const lrow = s_logit + l15 * kItemsPad + 4 * q;     // this lane's logits slot inside tile 0
        const int rot = (int)((blockIdx.x * 7u) % (unsigned)(n_full > 0 ? n_full : 1));   // de-phase the workgroups' bank streams
        auto tile_of = [&](int j) { int t = j + rot; return t >= n_full ? t - n_full : t; };
        auto load_a = [&](int t, float4 (&a)[4]) {
            const float *tb = (bank_packed ? bank_packed : bank) + (size_t)t * (16 * kC);   // wave-uniform
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = *(const float4 *)(tb + lane_elem + piece * g);
        };
        auto mul_store = [&](int t, const float4 (&a)[4]) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].x, bf[g].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].y, bf[g].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].z, bf[g].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g].w, bf[g].w, acc, 0, 0, 0);
            }
            // C/D map of 16x16x4: column (pillar) = lane & 15, row (item) = 4 * (lane >> 4) + reg
            *(float4 *)(lrow + t * 16) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        };
        // wave w owns tiles w, w + 16, w + 32, ... (rotated); two register sets alternate, the next tile's loads are in flight
        // while the current one multiplies
        float4 a0[4], a1[4];
        int j = (int)wave_u;
        if (j < n_full) load_a(tile_of(j), a0);
        for (; j < n_full; j += 2 * kWaves) {
            const int j1 = j + kWaves, j2 = j + 2 * kWaves;
            if (j1 < n_full) load_a(tile_of(j1), a1);
            mul_store(tile_of(j), a0);
            if (j1 < n_full) {
                if (j2 < n_full) load_a(tile_of(j2), a0);
                mul_store(tile_of(j1), a1);
            }
        }
        if ((n_items & 15) && wave_u == 0) {                       // the partial last tile: rows clamped, tail stays -inf
            const int item = min(n_full * 16 + l15, n_items - 1);
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a = bank_packed ? *(const float4 *)(bank_packed + (size_t)n_full * (16 * kC) + lane_elem + piece * g)   // zero padded
                                             : *(const float4 *)(bank + (size_t)item * kC + 4 * q + 16 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bf[g].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bf[g].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bf[g].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bf[g].w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n_full * 16 + 4 * q + r < n_items) lrow[n_full * 16 + r] = acc[r];
        }
    }
    __syncthreads();
#ifdef HVPR_EXP_TIMING
    const long long tt1 = __builtin_readcyclecounter();
#endif

    // ---- phase 2/3: one wave per pillar ----
    unsigned long long *cand = s_cand + wid * 64;
    for (int p = wid; p < np; p += kThreads / 64) {
        const float *row = s_logit + p * kItemsPad;
        float v[kItemsPad / 64];
        float lmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < kItemsPad / 64; ++t) { v[t] = row[lane + 64 * t]; lmax = fmaxf(lmax, v[t]); }
        // tau <= k-th largest of the 64 lane maxima (its 16 leading bits: half the ballot steps, a handful of extra
        // candidates): a lower bound of the k-th largest logit
        const unsigned tau_bits = wave_kth_largest_hi16(ord_bits(lmax), k);
        const unsigned tb = (tau_bits & 0x80000000u) ? (tau_bits & 0x7fffffffu) : ~tau_bits;
        const float tau = __uint_as_float(tb);                    // > -inf whenever at least k lanes hold a finite logit
        // candidates (>= tau) per lane as a bit mask (float compares only), then compacted rank by rank: rank r of every
        // lane with more than r candidates goes to LDS at a ballot-prefix position.  ~23 candidates, at most a few per lane.
        unsigned hits = 0u;
#pragma unroll
        for (int t = 0; t < kItemsPad / 64; ++t) hits |= (v[t] >= tau && v[t] > -INFINITY) ? (1u << t) : 0u;
        int cnt = 0;
        for (unsigned left = hits; __ballot(left != 0u) != 0ull;) {
            const bool has = left != 0u;
            const unsigned long long m = __ballot(has);
            const int t = has ? __ffs((int)left) - 1 : 0;
            left &= left - 1u;
            const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
            if (has && pos < 64)
                cand[pos] = ((unsigned long long)ord_bits(row[lane + 64 * t]) << 32) | (unsigned)(0xffffffffu - (unsigned)(lane + 64 * t));
            cnt += __popcll(m);
        }
        // the k largest keys (value desc, index asc; keys are unique) end up in lanes [0,k), in no particular order
        unsigned long long key = 0ull;
        bool sel;
        if (cnt <= 64) {
            key = lane < cnt ? cand[lane] : 0ull;
            const int kk = min(k, cnt);
            // select on the 32 value bits; the 32 index bits only matter when equal values straddle the k-th place
            const unsigned hi = (unsigned)(key >> 32);
            const unsigned kth_hi = wave_kth_largest_u32(hi, kk);
            const int n_gt = __popcll(__ballot(hi > kth_hi)), n_eq = __popcll(__ballot(key != 0ull && hi == kth_hi));
            if (n_gt + n_eq == kk) {
                sel = key != 0ull && hi >= kth_hi;
            } else {
                const unsigned long long kth = wave_kth_largest_u64(key, kk);
                sel = key != 0ull && key >= kth;
            }
        } else {
            // exact slow path (mass ties): k rounds of wave arg-max with (value desc, index asc) order
            unsigned long long prev = ~0ull;
            for (int r = 0; r < k; ++r) {
                unsigned long long best = 0ull;
#pragma unroll
                for (int t = 0; t < kItemsPad / 64; ++t) {
                    const unsigned long long c = ((unsigned long long)ord_bits(v[t]) << 32) |
                                                 (unsigned)(0xffffffffu - (unsigned)(lane + 64 * t));
                    if (c < prev && c > best && v[t] > -INFINITY) best = c;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const unsigned lo = __shfl_xor((unsigned)(best & 0xffffffffull), o, 64);
                    const unsigned hi = __shfl_xor((unsigned)(best >> 32), o, 64);
                    const unsigned long long ob = ((unsigned long long)hi << 32) | lo;
                    best = ob > best ? ob : best;
                }
                if (lane == r) key = best;
                prev = best;
            }
            sel = lane < k && key != 0ull;
        }
        // move the selected keys to lanes [0, #selected)
        {
            const unsigned long long m = __ballot(sel);
            const int dst = __popcll(m & ((1ull << lane) - 1ull));
            if (sel) cand[dst] = key;
            const int nsel = __popcll(m);
            key = lane < nsel ? cand[lane] : 0ull;
            sel = lane < nsel;
        }
        const unsigned ub = (unsigned)(key >> 32);
        const float logit = sel ? __uint_as_float((ub & 0x80000000u) ? (ub & 0x7fffffffu) : ~ub) : -INFINITY;
        const int idx = sel ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffull)) : 0;
        // softmax over the selected logits: two DPP wave reductions (lanes past the selection carry -inf / 0)
        const float mx = hvpr_reduce_max<64>(logit);
        const float e = sel ? __expf(logit - mx) : 0.f;
        const float a = e / hvpr_reduce_sum<64>(e);
        if (topk_idx && lane < k) topk_idx[(size_t)(p0 + p) * k + lane] = idx;
        long long cell = -1;           // fused a3+a4: this pillar's BEV cell (pointpillar_scatter.py:192, nz == 1)
        if (cell_map || canvas) {
            const int4 c = coords[p0 + p];
            if ((unsigned)c.x < (unsigned)batch && (unsigned)c.z < (unsigned)ny && (unsigned)c.w < (unsigned)nx)
                cell = ((long long)c.x * ny + c.z) * nx + c.w;
            if (cell_map && lane == 0 && cell >= 0) cell_map[cell] = p0 + p;   // gather-form scatter: k_cell_map's job
        }
        // lane = channel.  All k selected rows are requested before the first is used, so the gather costs one L2 round
        // trip instead of k dependent ones (row-major copy of the bank: 256 B per row).
        float rows[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) rows[r] = r < k ? bank[(size_t)__builtin_amdgcn_readlane(idx, r) * kC + lane] : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r)
            if (r < k) acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), r)), rows[r], acc);
        out[(size_t)(p0 + p) * kC + lane] = acc;
        // fused encode path: the memory channels of this pillar's cell, straight into the pre-cleared NHWC canvas
        if (canvas && cell >= 0) canvas[(size_t)cell * canvas_channels + canvas_offset + lane] = acc;
    }
#ifdef HVPR_EXP_TIMING
    if (blockIdx.x == 3 && lane == 0) printf("readout wave %d: mfma phase %lld cycles, select+gather %lld cycles\n", wid, tt1 - tt0, (long long)__builtin_readcyclecounter() - tt1);
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
    if (!f || !bank || !out) return HVPR_ERR_INVALID_ARG;
    if ((cell_map || canvas) && !coords) return HVPR_ERR_INVALID_ARG;
    if (canvas && canvas_offset + kC > canvas_channels) return HVPR_ERR_INVALID_ARG;
    const size_t lds = (size_t)kPillars * kItemsPad * 4 + kPillars * kC * 4 + (kThreads / 64) * 64 * 8;
    static unsigned long long lds_set = 0ull;   // per device
    if (hvpr_ensure_dyn_lds((const void *)k_memory_readout, (int)lds, &lds_set) != 0) return HVPR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_memory_readout, dim3(hvpr_cdiv(M, kPillars)), dim3(kThreads), lds, stream, f, M, m_device, bank,
                       bank_packed, n_items, k, out, topk_idx, (const int4 *)coords, batch, nx, ny, cell_map, canvas, canvas_channels,
                       canvas_offset);
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

// bank (n_items, 64) row-major -> packed [ceil(n_items / 16)][4 pieces][64 lanes] float4, rows past n_items zero
__global__ void __launch_bounds__(256) k_bank_pack(const float *__restrict__ bank, int n_items, float4 *__restrict__ packed, int n_out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_out) return;
    const int t = o >> 8, g = (o >> 6) & 3, ln = o & 63, row = 16 * t + (ln & 15);
    packed[o] = row < n_items ? *(const float4 *)(bank + (size_t)row * kC + 16 * g + 4 * (ln >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
}

extern "C" size_t hvpr_memory_bank_packed_floats(int n_items) { return n_items < 1 ? 0 : (size_t)hvpr_cdiv(n_items, 16) * 16 * kC; }

extern "C" int hvpr_memory_bank_pack_f32(const float *bank, int n_items, float *packed, hvpr_stream_t stream) {
    if (!bank || !packed || n_items < 1) return HVPR_ERR_INVALID_ARG;
    const int n_out = hvpr_cdiv(n_items, 16) * 256;
    hipLaunchKernelGGL(k_bank_pack, dim3(hvpr_cdiv(n_out, 256)), dim3(256), 0, (hipStream_t)stream, bank, n_items, (float4 *)packed, n_out);
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
