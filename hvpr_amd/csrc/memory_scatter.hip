// a3 + a4 — MemAE memory read-out (eval) and the scatter of pillars to the dense BEV canvases.
//   a3 replaces MemoryUnit_Agg.forward eval branch, map_to_bev/memory_module.py:60-77
//   a4 replaces PointPillarScatter_Agg_Memory_1_scale.forward eval branch, map_to_bev/pointpillar_scatter.py:169-222
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// a3  memory read-out.  One workgroup = kPillars pillars; the kPillars x n_items logit rows live in LDS
// (never in HBM: the reference materialises M x 2000 floats twice).  Selection per pillar (one wave):
//   lane-local max of the lane's 32 logits -> 64-lane bitonic sort -> tau = k-th largest lane max, a
//   lower bound of the k-th largest logit -> the expected ~23 logits >= tau are compacted and sorted
//   (value desc, index asc) -> top k.  More than 64 candidates (mass ties) takes an exact slow path.
// softmax(f . W[idx]) reuses the selected logits (memory_module.py:70-72 recomputes the same dot products).
// ------------------------------------------------------------------------------------------------
constexpr int kC = 64;          // feature channels
constexpr int kPillars = 16;    // pillars per workgroup
constexpr int kItemsPad = 2048; // logits row length in LDS
constexpr int kThreads = 256;

__device__ __forceinline__ unsigned ord_bits(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
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
                                                             const float *__restrict__ bank, int n_items, int k,
                                                             float *__restrict__ out, int *__restrict__ topk_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_logit = (float *)smem;                              // [kPillars][kItemsPad]
    float *s_f = s_logit + kPillars * kItemsPad;                 // [kPillars][kC]
    unsigned long long *s_cand = (unsigned long long *)(s_f + kPillars * kC);   // [waves][64]
    if (m_device) M = min(M, *m_device);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p0 = blockIdx.x * kPillars;
    if (p0 >= M) return;
    const int np = min(kPillars, M - p0);

    for (int i = tid; i < kPillars * kC; i += kThreads) {
        const int p = i / kC;
        s_f[i] = p < np ? f[(size_t)(p0 + p) * kC + (i % kC)] : 0.f;
    }
    __syncthreads();

    // ---- phase 1: logits[p][j] = f[p] . bank[j]  (thread = item, 16 pillars per pass over the row) ----
    for (int j = tid; j < kItemsPad; j += kThreads) {
        if (j < n_items) {
            float4 w[kC / 4];
            const float4 *row = (const float4 *)(bank + (size_t)j * kC);
#pragma unroll
            for (int c = 0; c < kC / 4; ++c) w[c] = row[c];
#pragma unroll 4
            for (int p = 0; p < kPillars; ++p) {
                const float4 *fp = (const float4 *)(s_f + p * kC);
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < kC / 4; ++c) {
                    const float4 x = fp[c];
                    a = fmaf(w[c].x, x.x, a); a = fmaf(w[c].y, x.y, a);
                    a = fmaf(w[c].z, x.z, a); a = fmaf(w[c].w, x.w, a);
                }
                s_logit[p * kItemsPad + j] = a;
            }
        } else {
#pragma unroll 4
            for (int p = 0; p < kPillars; ++p) s_logit[p * kItemsPad + j] = -INFINITY;
        }
    }
    __syncthreads();

    // ---- phase 2/3: one wave per pillar ----
    unsigned long long *cand = s_cand + wid * 64;
    for (int p = wid; p < np; p += kThreads / 64) {
        const float *row = s_logit + p * kItemsPad;
        float v[kItemsPad / 64];
        float lmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < kItemsPad / 64; ++t) { v[t] = row[lane + 64 * t]; lmax = fmaxf(lmax, v[t]); }
        // tau = k-th largest of the 64 lane maxima
        const unsigned long long sorted = bitonic64_desc_u64(((unsigned long long)ord_bits(lmax) << 32) | (unsigned)lane, lane);
        const unsigned tau_bits = __shfl((unsigned)(sorted >> 32), k - 1, 64);
        // compact the candidates (>= tau) into LDS, wave-uniform counter
        int cnt = 0;
#pragma unroll
        for (int t = 0; t < kItemsPad / 64; ++t) {
            const bool hit = ord_bits(v[t]) >= tau_bits && v[t] > -INFINITY;
            const unsigned long long m = __ballot(hit);
            if (m) {
                const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                if (hit && pos < 64)
                    cand[pos] = ((unsigned long long)ord_bits(v[t]) << 32) | (unsigned)(0xffffffffu - (unsigned)(lane + 64 * t));
                cnt += __popcll(m);
            }
        }
        unsigned long long key;
        if (cnt <= 64) {
            key = lane < cnt ? cand[lane] : 0ull;
            key = bitonic64_desc_u64(key, lane);
        } else {
            // exact slow path (mass ties): k rounds of wave arg-max with (value desc, index asc) order
            key = 0ull;
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
        }
        // lanes [0,k): selected (logit, index), descending
        const bool sel = lane < k && key != 0ull;
        const unsigned ub = (unsigned)(key >> 32);
        const float logit = sel ? __uint_as_float((ub & 0x80000000u) ? (ub & 0x7fffffffu) : ~ub) : -INFINITY;
        const int idx = sel ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffull)) : 0;
        const float mx = hvpr_reduce_max<64>(logit);
        const float e = sel ? __expf(logit - mx) : 0.f;
        const float a = e / hvpr_reduce_sum<64>(e);
        if (topk_idx && lane < k) topk_idx[(size_t)(p0 + p) * k + lane] = idx;
        float acc = 0.f;   // lane = channel
        for (int r = 0; r < k; ++r) {
            const float ar = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), r));
            const int ir = __builtin_amdgcn_readlane(idx, r);
            acc = fmaf(ar, bank[(size_t)ir * kC + lane], acc);
        }
        out[(size_t)(p0 + p) * kC + lane] = acc;
    }
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

extern "C" int hvpr_memory_readout_fwd_f32(const float *f, int M, const int32_t *m_device, const float *bank,
                                           int n_items, int k, float *out, int32_t *topk_idx, hvpr_stream_t stream) {
    if (M < 0 || n_items < 1 || k < 1) return HVPR_ERR_INVALID_ARG;
    if (k > 32 || k > n_items || n_items > kItemsPad) return HVPR_ERR_UNSUPPORTED;
    if (M == 0) return HVPR_OK;
    if (!f || !bank || !out) return HVPR_ERR_INVALID_ARG;
    const size_t lds = (size_t)kPillars * kItemsPad * 4 + kPillars * kC * 4 + (kThreads / 64) * 64 * 8;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_memory_readout, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return HVPR_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_memory_readout, dim3(hvpr_cdiv(M, kPillars)), dim3(kThreads), lds, (hipStream_t)stream, f, M,
                       m_device, bank, n_items, k, out, topk_idx);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
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
