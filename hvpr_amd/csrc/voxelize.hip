// a1 — GPU voxelizer with the sequential first-touch semantics of spconv's VoxelGenerator
// (call site in the reference: pcdet/datasets/processor/data_processor.py:43-75; in-tree twin of the
// loop: tools/vis.py:9-60).  The sequential hash is re-expressed as order statistics so that it is
// bit-identical on a parallel machine:
//   voxel id of a cell   = rank of the cell among cells ordered by their smallest point index
//   slot of a point      = rank of its index among the points of the same cell (first max_points kept)
//   max_voxels (V2)      = cells of rank >= max_voxels are dropped, everything else unchanged
//   max_voxels (V1)      = additionally every point with index >= first index of the rank==max_voxels cell
//
// Four launches, no memsets: the two persistent cell maps are returned to their idle state by K3.
//   K1 cell keys (fp32 sub/div/floor, IEEE), atomicMin(first index) + atomicAdd(count) per cell
//   K2 single-pass decoupled-look-back scan over points of (is_first, count) -> voxel rank, arena offset
//   K3 each point appends its index to its voxel's arena segment (unordered)
//   K4 one wave per voxel: bitonic selection of the max_points smallest indices (ascending), gather.
#include <stdlib.h>

#include <mutex>

#include "common.h"
#include "internal.h"

namespace {

__device__ __forceinline__ int frame_of(const int *__restrict__ off, int batch, int i) {
    // largest b in [0,batch) with off[b] <= i
    int lo = 0, hi = batch;   // invariant: off[lo] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// Inclusive add-scan over the 64 lanes on DPP moves (row_shr inside the 16-lane rows, row_bcast across them): ~10 instructions
// per value instead of six ds_bpermute round trips.
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);     // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);     // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);     // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);     // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2, 3
    return v;
}

// A LiDAR sweep delivers the points of a cell in bursts, so consecutive lanes of a wave often hold the same cell — and atomics on
// one word are performed one after the other (~12 ns each, the hottest cell's queue is what phase 1 waits for).  Runs of equal
// cells inside the wave are folded first: the run's head (its lowest lane = its smallest point index) issues ONE atomicMin and
// ONE atomicAdd of the run length.  Returns the head's lane and the run length; `g` < 0 never forms a run.
__device__ __forceinline__ void cell_runs(int g, int lane, int &head, int &len) {
    const int prev = __builtin_amdgcn_update_dpp(-2, g, 0x138, 0xf, 0xf, false);        // wave_shr:1 (lane 0 keeps -2)
    const unsigned long long heads = __ballot(g != prev || g < 0) | 1ull;                // bit l: lane l starts a run
    const unsigned long long upto = heads & (~0ull >> (63 - lane));                      // heads at or below this lane
    head = 63 - __clzll((long long)upto);
    const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);            // next head above this lane
    len = (above ? lane + 1 + (__ffsll((long long)above) - 1) : 64) - head;
}

__global__ void k_reset(int *cell_first, int *cell_count, long long n, int *sync) {
    if (blockIdx.x == 0 && threadIdx.x < 8 + 64) sync[threadIdx.x] = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        cell_first[i] = kIdle;
        cell_count[i] = 0;
    }
}

__device__ __forceinline__ int k1_cell(const float *__restrict__ pts, int i, int stride, int xyz_col,
                                       const int *__restrict__ foff, int batch, float lox, float loy, float loz,
                                       float vsx, float vsy, float vsz, int nx, int ny, int nz, const VoxWs &w) {
    const float *p = pts + (size_t)i * stride + xyz_col;
    // exact IEEE fp32: one subtract, one divide, floor (no fast-math; hipcc's default division is correctly rounded)
    const float cx = floorf(__fdiv_rn(__fsub_rn(p[0], lox), vsx));
    const float cy = floorf(__fdiv_rn(__fsub_rn(p[1], loy), vsy));
    const float cz = floorf(__fdiv_rn(__fsub_rn(p[2], loz), vsz));
    int g = -1;
    if (cx >= 0.f && cx < (float)nx && cy >= 0.f && cy < (float)ny && cz >= 0.f && cz < (float)nz) {
        const int b = frame_of(foff, batch, i);
        g = ((b * nz + (int)cz) * ny + (int)cy) * nx + (int)cx;
    }
    w.pt_cell[i] = g;
    return g;
}

// SLOTS (fused encode path beyond the one-launch sizes): the count's atomicAdd returns the point's arrival number in its cell —
// its arena slot, left in w.arena[point] — so that K3 needs no atomic of its own (the count map is returned to idle by the canvas
// clear of the pillar launch, with the first-index map): three
// memory-side atomics per point become two (batch 16: K1 28.9 -> 31 us, K3 20.3 -> 10.9 us).
template <bool SLOTS>
__global__ void __launch_bounds__(256) k1_keys(const float *__restrict__ pts, int n, int stride, int xyz_col,
                                               const int *__restrict__ foff, int batch, float lox, float loy, float loz,
                                               float vsx, float vsy, float vsz, int nx, int ny, int nz, VoxWs w, int tiles) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tiles) w.tile_state[i] = 0ull;
    if (i == 0) *w.ticket = 0;
    const int g = i < n ? k1_cell(pts, i, stride, xyz_col, foff, batch, lox, loy, loz, vsx, vsy, vsz, nx, ny, nz, w) : -1;
    const int lane = threadIdx.x & 63;
    int head, len;
    cell_runs(g, lane, head, len);
    int slot = 0;
    if (g >= 0 && head == lane) {      // one pair of atomics per run of equal cells; the run's points share its block of slots
        atomicMin(&w.cell_first[g], i);
        if (SLOTS) slot = atomicAdd(&w.cell_count[g], len);
        else atomicAdd(&w.cell_count[g], len);
    }
    if (SLOTS) {
        slot = __shfl(slot, head, 64) + (lane - head);
        if (g >= 0) w.arena[i] = slot;
    }
}

// status (2 bits) | first-touch sum (31 bits) | count sum (31 bits)
__device__ __forceinline__ unsigned long long pack(unsigned st, unsigned a, unsigned b) {
    return ((unsigned long long)st << 62) | ((unsigned long long)a << 31) | (unsigned long long)b;
}

// ITEMS points per thread: 2 keeps a single frame spread over many workgroups; 8 shortens the look-back chain of a
// large batch (config 5: 46 -> 33 us).
template <int ITEMS>
__global__ void __launch_bounds__(kScanThreads) k2_scan(int n, const int *__restrict__ foff, int batch, VoxWs w) {
    __shared__ int s_tile;
    __shared__ unsigned s_wave_a[kScanThreads / 64], s_wave_b[kScanThreads / 64];
    __shared__ unsigned s_excl_a, s_excl_b;
    if (threadIdx.x == 0) s_tile = atomicAdd(w.ticket, 1);
    __syncthreads();
    const int tile = s_tile;
    const int base = tile * (kScanThreads * ITEMS) + threadIdx.x * ITEMS;

    int gcell[ITEMS];
    unsigned fl[ITEMS], ct[ITEMS];
    unsigned ta = 0, tb = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) gcell[k] = (base + k < n) ? w.pt_cell[base + k] : -1;
    int first[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) first[k] = gcell[k] >= 0 ? w.cell_first[gcell[k]] : -1;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        fl[k] = 0; ct[k] = 0;
        if (gcell[k] >= 0 && first[k] == base + k) { fl[k] = 1; ct[k] = (unsigned)w.cell_count[gcell[k]]; }
        else gcell[k] = -1;
        ta += fl[k]; tb += ct[k];
    }
    // block exclusive scan of (ta, tb)
    unsigned ia = ta, ib = tb;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned ua = __shfl_up(ia, o, 64), ub = __shfl_up(ib, o, 64);
        if (lane >= o) { ia += ua; ib += ub; }
    }
    if (lane == 63) { s_wave_a[wid] = ia; s_wave_b[wid] = ib; }
    __syncthreads();
    unsigned wa = 0, wb = 0, tot_a = 0, tot_b = 0;
#pragma unroll
    for (int k = 0; k < kScanThreads / 64; ++k) {
        if (k < wid) { wa += s_wave_a[k]; wb += s_wave_b[k]; }
        tot_a += s_wave_a[k]; tot_b += s_wave_b[k];
    }
    // Decoupled look-back by wave 0, 64 predecessors per step (the tile word is both data and flag — one 8-byte
    // agent-scope store).  Every tile publishes its own aggregate as soon as it has it, so a tile normally needs ONE
    // round of loads: it adds the aggregates down to the nearest tile that already knows its inclusive prefix.
    if (wid == 0) {
        unsigned ea = 0, eb = 0;
        if (lane == 0)
            __hip_atomic_store(&w.tile_state[tile], pack(tile == 0 ? 2u : 1u, tot_a, tot_b), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        for (int hi = tile - 1; hi >= 0; hi -= 64) {
            const int t = hi - lane;
            unsigned long long st = 0ull;
            for (;;) {
                if (t >= 0 && (st >> 62) == 0)
                    st = __hip_atomic_load(&w.tile_state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__ballot(t >= 0 && (st >> 62) == 0) == 0ull) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const unsigned long long incl = __ballot(t >= 0 && (st >> 62) == 2);
            const int stop = incl ? __ffsll((long long)incl) - 1 : 63;   // nearest predecessor with an inclusive prefix
            unsigned pa = (t >= 0 && lane <= stop) ? (unsigned)((st >> 31) & 0x7fffffffu) : 0u;
            unsigned pb = (t >= 0 && lane <= stop) ? (unsigned)(st & 0x7fffffffu) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { pa += __shfl_xor(pa, o, 64); pb += __shfl_xor(pb, o, 64); }
            ea += pa; eb += pb;
            if (incl) break;
        }
        if (lane == 0) {
            if (tile != 0)
                __hip_atomic_store(&w.tile_state[tile], pack(2u, ea + tot_a, eb + tot_b), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_excl_a = ea; s_excl_b = eb;
        }
    }
    __syncthreads();
    unsigned ra = s_excl_a + wa + (ia - ta);   // exclusive prefix of this thread's first item
    unsigned rb = s_excl_b + wb + (ib - tb);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int i = base + k;
        if (i < n) {
            // frame bases: the thread owning the first point of a frame knows that frame's first voxel rank
            const int b = frame_of(foff, batch, i);
            if (foff[b] == i)
                for (int bb = b; bb >= 0 && foff[bb] == i; --bb) w.frame_base[bb] = (int)ra;
            if (fl[k]) {
                w.cell_vid[gcell[k]] = (int)ra;
                w.vox_rec[ra] = make_int4(gcell[k], (int)ct[k], (int)rb, i);
            }
            if (i == n - 1) {
                const int total = (int)(ra + fl[k]);
                *w.arena_total = (int)(rb + ct[k]);
                for (int bb = batch; bb >= 0 && foff[bb] == n; --bb) w.frame_base[bb] = total;
            }
        }
        ra += fl[k]; rb += ct[k];
    }
}

constexpr int kWarmParts = 32;   // L2 warmers per XCD (below)

// ---- K1 + K2 + K3 as the three phases of ONE launch (fused encode path, up to kFusedMaxOwners x T points) --------------------------
// A frame's index work is three short dependent chains (~13 dependent memory round trips in all); as three launches each also pays
// a kernel's start and drain.  Here every workgroup owns the same T points in all three phases (cell, arena slot and the point stay
// in registers) and the phases are separated by two grid barriers that cost no cache maintenance (no release / acquire fence: those
// write back / invalidate whole caches and made round 4's one-launch form slower than the three launches).
//   * Placement-independent part.  Phase 1's cell-map atomics and the first barrier are device-scope (performed at the memory
//     side, ~1 us per dependent hop).  The barrier word is a census: every workgroup adds 1 to the byte of the XCD it runs on
//     (HW_REG_XCC_ID), so that after the barrier all of them know, consistently, whether they share one XCD.
//   * If they do (the dispatcher deals block b to XCD b % 8 and the owners are the blocks with b % 8 == 0 — observed, not
//     promised), everything phases 2 and 3 exchange stays in that XCD's L2: plain (sc0) stores keep their line there, sc1 loads
//     bypass only the CU's L1 and are served by the L2, workgroup-scope atomics are performed in it: ~0.3-0.4 us per hop.
//   * If they do not, the same code runs with device-scope stores / atomics (sc1: write-through, dropped from the L2; "8-B agent
//     atomics both sides"), correct for ANY placement, at the memory side's latency.
// What the NEXT kernels read (arena, per-voxel records, cell_vid) is written with plain stores and published by the kernel end.
// Differences from the three kernels, none visible outside: the arena slot of a point is handed out in phase 1 (returning
// atomicAdd: arrival order) instead of phase 3, phase 2 returns the count map to idle, pt_cell is not written.
constexpr int kFusedMaxOwners = 32;   // one workgroup per CU of ONE XCD: all owners resident whatever else runs (they spin on each other)
// Every wait of the one-launch kernel is BOUNDED: an owner that has waited kSpinLimitTicks of the constant 100 MHz clock
// (s_memrealtime) — two seconds; a wait that is going to end ends within the run time of whatever else occupies the compute units
// — or that sees the workspace's error word raised stops waiting, raises the (sticky) error word sync[4] and leaves through the
// normal exit protocol, which returns the barrier words to idle.  The launch then reports ZERO pillars (voxel_offsets all 0), and so
// does every later one-launch call on this workspace until hvpr_voxelize_workspace_reset: a launch that could not complete never
// hangs the device and never hands out partial results (hvpr_voxelize_workspace_status reads the word).
constexpr long long kSpinLimitTicks = 200000000ll;
constexpr int kSyncExit = 3, kSyncError = 4;

struct SpinWatch {
    long long t0;
    unsigned polls;
    int *err;
    __device__ __forceinline__ SpinWatch(int *error_word) : t0(0), polls(0), err(error_word) {}
    // true when this wait has to be given up; looks at the clock / the error word every 256 polls only
    __device__ __forceinline__ bool give_up() {
        if ((++polls & 255u) != 0u) return false;
        const long long now = (long long)__builtin_amdgcn_s_memrealtime();
        if (t0 == 0) t0 = now;
        return now - t0 > kSpinLimitTicks || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    }
};

template <typename V>
__device__ __forceinline__ void xstore(V *p, V v, bool local) {
    if (local) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename V>
__device__ __forceinline__ V xload(V *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // sc1: past the L1

__device__ __forceinline__ unsigned byte_sum(unsigned long long v) {
    v = (v & 0x00ff00ff00ff00ffull) + ((v >> 8) & 0x00ff00ff00ff00ffull);
    v = (v & 0x0000ffff0000ffffull) + ((v >> 16) & 0x0000ffff0000ffffull);
    return (unsigned)(v + (v >> 32));
}

template <int T>
__device__ __forceinline__ void warm_l2(int part, int parts, const float4 *__restrict__ warm0, long long warm0_v4,
                                        const float4 *__restrict__ warm1, long long warm1_v4, const WarmSmall &small, int *sink) {
    float acc = 0.f;
    // one warmer per XCD also reads the small arrays (the pillar VFE's weights) — requested first, added last, so that their
    // round trip runs beside the big arrays' (up to 256 * {1, 1, 8, 1, 1, 1, 2, 1} floats: what the pillar VFE has)
    constexpr int kSmallLoads[8] = {1, 1, 8, 1, 1, 1, 2, 1};
    float sm[16];
    const bool smalls = part == 0 && threadIdx.x < 256;
    if (smalls) {
        int q = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int u = 0; u < kSmallLoads[j]; ++u, ++q) {
                const int i = threadIdx.x + 256 * u;
                sm[q] = i < small.n[j] ? small.p[j][i] : 0.f;
            }
    }
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        const float4 *src = which ? warm1 : warm0;
        const long long n = which ? warm1_v4 : warm0_v4, per = (n + parts - 1) / parts;
        const long long lo = part * per, hi = lo + per < n ? lo + per : n;
        for (long long i = lo + threadIdx.x; i < hi; i += T * 8) {
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = i + T * u < hi ? src[i + T * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += q[u].x + q[u].w;
        }
    }
    if (smalls) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += sm[q];
    }
    if (acc == 1.2345678e-30f) *sink = 1;      // never true in practice: keeps the loads alive
}

template <int T>
// (1024-thread form: second launch bound = eight waves per SIMD, i.e. at most 64 VGPRs, so that TWO 1024-thread owners / four 512-thread owners fit a CU — the 32 owners of a 32 768-point
// call then need 16 CUs of their XCD, not all 32: with 65 registers the batch-2 frame pipeline lost 12 % waiting for them, round 6)
__global__ void __launch_bounds__(T, T == 1024 ? 8 : 1) k_index(const float *__restrict__ pts, int n, int stride, int xyz_col,
                                             const int *__restrict__ foff, int batch, float lox, float loy, float loz, float vsx,
                                             float vsy, float vsz, int nx, int ny, int nz, int max_voxels, VoxWs w,
                                             int *__restrict__ voxel_offsets, const float *__restrict__ vfe_w1,
                                             const float *__restrict__ vfe_b0, int point_blocks, int warm_parts,
                                             const float4 *__restrict__ warm0, long long warm0_v4, const float4 *__restrict__ warm1,
                                             long long warm1_v4, WarmSmall small, int *__restrict__ sink, int force_agent) {
    const int owners_end = 8 * point_blocks;   // the owner of tile t is block 8 t: the blocks a round-robin deal puts on ONE XCD
    if ((int)blockIdx.x > owners_end) {        // L2 warmers (internal.h), dealt to the XCDs in turn; they take no part in the barriers
        warm_l2<T>(((int)blockIdx.x - owners_end - 1) >> 3, warm_parts, warm0, warm0_v4, warm1, warm1_v4, small, sink);
        return;
    }
    if ((int)blockIdx.x == owners_end) {       // the pillar VFE's padded-slot column (internal.h)
        if (vfe_w1 && threadIdx.x < 64) {
            const int lane = threadIdx.x, h = lane >> 5, slot = lane & 31;
            float aw[2][8], b0h[8];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int t = 0; t < 8; ++t) aw[mb][t] = vfe_w1[(32 * mb + slot) * 32 + 8 * (t >> 2) + 4 * h + (t & 3)];
#pragma unroll
            for (int t = 0; t < 8; ++t) b0h[t] = vfe_b0[8 * (t >> 2) + 4 * h + (t & 3)];
            hvpr_vfe_padded_slot(aw, b0h, w.vfe_aux);
        }
        return;
    }
    if (blockIdx.x & 7) return;
    __shared__ unsigned s_wave_a[T / 64], s_wave_b[T / 64];
    __shared__ unsigned s_excl_a, s_excl_b;
    __shared__ int s_local, s_abort;
    const int tile = blockIdx.x >> 3, i = tile * T + (int)threadIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#ifdef HVPR_EXP_TIMING
    long long ts[12];
#define IDX_STAMP(k) ts[k] = __builtin_amdgcn_s_memrealtime()
#else
#define IDX_STAMP(k)
#endif
    IDX_STAMP(0);
    unsigned long long *const census = reinterpret_cast<unsigned long long *>(w.sync);   // sync[0..1]; sync[3]: exits; sync[4]: error; sync[8..71]: barrier 2's flags
    int *const err_word = w.sync + kSyncError;
    // a workspace whose error word is up (an earlier launch gave up a wait) is not touched: zero pillars until it is reset.  The word
    // was written before this launch started (stream order), so every owner reads the same value and none waits for another.
    // (requested here, looked at after the point has been loaded: the two round trips run side by side)
    const int poisoned = __hip_atomic_load(err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) s_abort = 0;

    // ---- phase 1: cell key, first index and count of the cell; the point's arena slot = its arrival number in the cell
    if (threadIdx.x == 0) __hip_atomic_store(&w.tile_state[tile], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
    int g = -1, slot = 0, b = 0;
    if (i < n) {
        const float *p = pts + (size_t)i * stride + xyz_col;
        pt = make_float4(p[0], p[1], p[2], p[3]);
        const float cx = floorf(__fdiv_rn(__fsub_rn(pt.x, lox), vsx));
        const float cy = floorf(__fdiv_rn(__fsub_rn(pt.y, loy), vsy));
        const float cz = floorf(__fdiv_rn(__fsub_rn(pt.z, loz), vsz));
        b = frame_of(foff, batch, i);
        if (cx >= 0.f && cx < (float)nx && cy >= 0.f && cy < (float)ny && cz >= 0.f && cz < (float)nz) {
            g = ((b * nz + (int)cz) * ny + (int)cy) * nx + (int)cx;
        }
    }
    if (poisoned != 0) {
        if (tile == 0 && (int)threadIdx.x <= batch) voxel_offsets[threadIdx.x] = 0;
        return;
    }
    {
        int head, len;
        cell_runs(g, lane, head, len);
        if (g >= 0 && head == lane) {      // one pair of atomics per run of equal cells; the run's points share its block of slots
            __hip_atomic_fetch_min(&w.cell_first[g], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            slot = __hip_atomic_fetch_add(&w.cell_count[g], len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        slot = __shfl(slot, head, 64) + (lane - head);
    }
    IDX_STAMP(1);
    // barrier 1 (device scope) + census of the XCDs the owners run on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's atomics have been performed
    IDX_STAMP(2);
    __syncthreads();
    IDX_STAMP(3);
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        __hip_atomic_fetch_add(census, 1ull << (8 * xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // three polls in flight, a quarter of a round trip apart: a poll that left just before the last arrival costs a whole
        // memory-side round trip (~1 us) otherwise
        unsigned long long c0 = __hip_atomic_load(census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_sleep(4);
        unsigned long long c1 = __hip_atomic_load(census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_sleep(4);
        unsigned long long c2 = __hip_atomic_load(census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long c = c0;
        SpinWatch watch(err_word);
        while (byte_sum(c) < (unsigned)point_blocks) {
            if (watch.give_up()) { s_abort = 1; break; }
            c = c1; c1 = c2;
            __builtin_amdgcn_s_sleep(4);
            c2 = __hip_atomic_load(census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_local = (!force_agent && ((c >> (8 * xcc)) & 0xffull) == (unsigned long long)point_blocks) ? 1 : 0;
    }
    __syncthreads();
    const bool local = s_local != 0;       // the same answer in every owner: they all read the same final census
    IDX_STAMP(4);
    bool done_ok = false;
    do {       // (left with `break` when a wait was given up: straight to the exit protocol)
    if (s_abort) break;

    // ---- phase 2: (is-first, count) scanned over the points -> voxel rank, arena offset
    unsigned fl = 0, ct = 0;
    if (g >= 0) {      // both requested at once: one round trip (the count is only used by the cell's first point)
        const int first = xload(&w.cell_first[g]), count = xload(&w.cell_count[g]);
        if (first == i) { fl = 1; ct = (unsigned)count; }
    }
    IDX_STAMP(5);
    const unsigned ia = wave_scan_incl(fl), ib = wave_scan_incl(ct);
    if (lane == 63) { s_wave_a[wid] = ia; s_wave_b[wid] = ib; }
    __syncthreads();
    unsigned wa = 0, wb = 0, tot_a = 0, tot_b = 0;
#pragma unroll
    for (int k = 0; k < T / 64; ++k) {
        if (k < wid) { wa += s_wave_a[k]; wb += s_wave_b[k]; }
        tot_a += s_wave_a[k]; tot_b += s_wave_b[k];
    }
    if (wid == 0) {     // decoupled look-back as in k2_scan; all tiles are resident, tile order = owner order
        unsigned ea = 0, eb = 0;
        bool lb_ok = true;
        if (lane == 0) xstore(&w.tile_state[tile], pack(tile == 0 ? 2u : 1u, tot_a, tot_b), local);
        for (int hi = tile - 1; hi >= 0; hi -= 64) {
            const int t = hi - lane;
            unsigned long long st = 0ull;
            SpinWatch watch(err_word);
            bool gave_up = false;
            for (;;) {
                if (t >= 0 && (st >> 62) == 0) st = xload(&w.tile_state[t]);
                if (__ballot(t >= 0 && (st >> 62) == 0) == 0ull) break;
                if (__ballot(watch.give_up()) != 0ull) { gave_up = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (gave_up) { if (lane == 0) s_abort = 1; lb_ok = false; break; }
            const unsigned long long incl = __ballot(t >= 0 && (st >> 62) == 2);
            const int stop = incl ? __ffsll((long long)incl) - 1 : 63;
            unsigned pa = (t >= 0 && lane <= stop) ? (unsigned)((st >> 31) & 0x7fffffffu) : 0u;
            unsigned pb = (t >= 0 && lane <= stop) ? (unsigned)(st & 0x7fffffffu) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { pa += __shfl_xor(pa, o, 64); pb += __shfl_xor(pb, o, 64); }
            ea += pa; eb += pb;
            if (incl) break;
        }
        if (lane == 0 && lb_ok) {
            if (tile != 0) xstore(&w.tile_state[tile], pack(2u, ea + tot_a, eb + tot_b), local);
            s_excl_a = ea; s_excl_b = eb;
        }
    }
    __syncthreads();
    if (s_abort) break;
    const unsigned ra = s_excl_a + wa + (ia - fl), rb = s_excl_b + wb + (ib - ct);   // exclusive prefixes of this point
    IDX_STAMP(6);
    if (i < n) {
        if (foff[b] == i)
            for (int bb = b; bb >= 0 && foff[bb] == i; --bb) xstore(&w.frame_base[bb], (int)ra, local);
        if (fl) {
            w.cell_count[g] = 0;      // the count map is idle again (nobody reads it any more in this launch)
            xstore(&w.cell_pack[g], (unsigned long long)ra | ((unsigned long long)ct << 20) | ((unsigned long long)rb << 40), local);
            w.cell_vid[g] = (int)ra;
            w.vox_rec[ra] = make_int4(g, (int)ct, (int)rb, i);
        }
        if (i == n - 1) {
            const int total = (int)(ra + fl);
            *w.arena_total = (int)(rb + ct);
            for (int bb = batch; bb >= 0 && foff[bb] == n; --bb) xstore(&w.frame_base[bb], total, local);
        }
    }
    // barrier 2, without a read-modify-write (those are performed at the memory side whatever their scope): every owner raises
    // its own flag word and wave 0 watches all of them — in the XCD's L2 when the owners share it (sc0 store, sc1 loads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDX_STAMP(7);
    __syncthreads();
    if (wid == 0) {
        int *const flags = w.sync + 8;
        if (lane == 0) xstore(&flags[tile], 1, local);
        SpinWatch watch(err_word);
        while (__ballot(lane < point_blocks && xload(&flags[lane]) == 0) != 0ull) {
            if (__ballot(watch.give_up()) != 0ull) { if (lane == 0) s_abort = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (s_abort) break;

    IDX_STAMP(8);
    // ---- phase 3: every point files its record and itself at its arena position
    if (i == 0) {
        int acc = 0;
        for (int bb = 0; bb < batch; ++bb) {
            voxel_offsets[bb] = acc;
            const int c = xload(&w.frame_base[bb + 1]) - xload(&w.frame_base[bb]);
            acc += c < max_voxels ? c : max_voxels;
        }
        voxel_offsets[batch] = acc;
    }
    if (g >= 0) {
        const unsigned long long pk = xload(&w.cell_pack[g]);
        const int r = (int)(pk & 0xfffffull), cnt = (int)((pk >> 20) & 0xfffffull), off = (int)(pk >> 40);
        const int fb = batch > 1 ? xload(&w.frame_base[b]) : 0;
        const bool kept = r - fb < max_voxels;
        const int pos = off + (cnt - 1 - slot);      // from the top, like K3's atomicSub: a cell filled in index order reads descending
        w.arena_rec[pos] = make_int4(i, kept ? r : -1, cnt, g);
        if (kept) w.arena_pt[pos] = pt;
    }
    done_ok = true;
    } while (false);
    if (!done_ok) {     // a wait was given up (here or in another owner): sticky error word, zero pillars
        if (threadIdx.x == 0) __hip_atomic_store(err_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tile == 0 && (int)threadIdx.x <= batch) voxel_offsets[threadIdx.x] = 0;
    }
    IDX_STAMP(9);
#ifdef HVPR_EXP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IDX_STAMP(10);
    if (threadIdx.x == 0 && (tile == 0 || tile == point_blocks - 1 || tile == point_blocks / 2))
        printf("k_index tile %d local %d: start %lld | p1 issued +%lld | drained +%lld | sync +%lld | barrier1 +%lld | p2 loads +%lld | scan+lookback +%lld | "
               "stores drained +%lld | barrier2 +%lld | p3 issued +%lld | drained +%lld (x10 ns)\n", tile, (int)local, ts[0], ts[1] - ts[0], ts[2] - ts[1],
               ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[7] - ts[6], ts[8] - ts[7], ts[9] - ts[8], ts[10] - ts[9]);
#endif
    // the last owner out returns the counters to idle (nobody is behind it: all have passed both barriers); device scope, so that
    // the next launch finds them whatever its placement
    // (an owner that gave up a wait comes through here as well: the words are idle again after ANY launch whose owners all ran)
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(&w.sync[kSyncExit], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == point_blocks - 1) {
        __hip_atomic_store(census, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&w.sync[kSyncExit], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int t = 0; t < point_blocks; ++t) __hip_atomic_store(&w.sync[8 + t], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void __launch_bounds__(256) k3_fill(int n, const int *__restrict__ foff, int batch, int max_voxels, VoxWs w,
                                               int *__restrict__ voxel_offsets, int for_encode, int have_slots,
                                               const float *__restrict__ pts, int stride, int xyz_col,
                                               const float *__restrict__ vfe_w1, const float *__restrict__ vfe_b0, int point_blocks,
                                               const float4 *__restrict__ warm0, long long warm0_v4, const float4 *__restrict__ warm1,
                                               long long warm1_v4, WarmSmall small, int *__restrict__ sink) {
    if ((int)blockIdx.x > point_blocks) {          // L2 warmers (internal.h): workgroups are dealt to the XCDs in turn;
        // warmer w = 8 * part + x reads slice `part` (of kWarmParts) of both arrays on the XCD its index lands on, all its loads
        // (six per thread for the 768 KB of the memory bank) in flight at once
        warm_l2<256>(((int)blockIdx.x - point_blocks - 1) >> 3, kWarmParts, warm0, warm0_v4, warm1, warm1_v4, small, sink);
        return;
    }
    if (vfe_w1 && (int)blockIdx.x == point_blocks) {   // the extra workgroup: the pillar VFE's padded-slot column (internal.h)
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x, h = lane >> 5, slot = lane & 31;
            float aw[2][8], b0h[8];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int t = 0; t < 8; ++t) aw[mb][t] = vfe_w1[(32 * mb + slot) * 32 + 8 * (t >> 2) + 4 * h + (t & 3)];
#pragma unroll
            for (int t = 0; t < 8; ++t) b0h[t] = vfe_b0[8 * (t >> 2) + 4 * h + (t & 3)];
            hvpr_vfe_padded_slot(aw, b0h, w.vfe_aux);
        }
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        int acc = 0;
        for (int b = 0; b < batch; ++b) {
            voxel_offsets[b] = acc;
            const int c = w.frame_base[b + 1] - w.frame_base[b];
            acc += c < max_voxels ? c : max_voxels;
        }
        voxel_offsets[batch] = acc;
    }
    if (i >= n) return;
    const int g = w.pt_cell[i];
    if (g < 0) return;
    float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (for_encode) {   // requested before the dependent chain below, lands for free
        const float *src = pts + (size_t)i * stride + xyz_col;
        pt = make_float4(src[0], src[1], src[2], src[3]);
    }
    const int r = w.cell_vid[g];
    const int b = frame_of(foff, batch, i);
    const int local = r - w.frame_base[b];
    // the point's slot in its voxel: handed out by K1 (k1_keys<true>: arrival numbers, counted from the top here like the atomicSub) or
    // taken now — which also returns the count map to its idle 0
    const int slot_k1 = have_slots ? w.arena[i] : 0;
    const int4 rec = w.vox_rec[r];
    const int slot = have_slots ? rec.y - 1 - slot_k1 : atomicSub(&w.cell_count[g], 1) - 1;
    if (!for_encode) w.cell_first[g] = kIdle;              // idle again (benign same-value race)
    const int pos = rec.z + slot;
    if (for_encode) {
        // every arena position gets its record, the ones of voxels beyond the cap too (rank -1): a pillar wave reads
        // fixed windows of the arena and must be able to tell what it is looking at
        w.arena_rec[pos] = make_int4(i, local < max_voxels ? r : -1, rec.y, g);
        if (local < max_voxels) w.arena_pt[pos] = pt;
    } else if (local < max_voxels) {
        w.arena[pos] = i;
    }
}

__device__ __forceinline__ int bitonic64_asc(int v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = __shfl_xor(v, j, 64);
            const bool up = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            v = (lower == up) ? min(v, o) : max(v, o);
        }
    }
    return v;
}

__global__ void __launch_bounds__(256) k4_gather(const float *__restrict__ pts, int stride, int xyz_col, int n_feat,
                                                 int batch, int nx, int ny, int nz, int max_points, int max_voxels,
                                                 int cap_mode, VoxWs w, const int *__restrict__ voxel_offsets,
                                                 float *__restrict__ voxels, int *__restrict__ coords,
                                                 int *__restrict__ num_points, int capacity) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    const int total = w.frame_base[batch];
    for (int r = wave; r < total; r += n_waves) {
        // frame of this rank: largest b with frame_base[b] <= r and a non-empty frame
        int b = 0;
        for (int bb = 1; bb < batch; ++bb) if (w.frame_base[bb] <= r) b = bb;
        const int local = r - w.frame_base[b];
        if (local >= max_voxels) continue;
        const int o = voxel_offsets[b] + local;
        if (o >= capacity) continue;
        int cutoff = kIdle;
        if (cap_mode == 1) {
            const int rc = w.frame_base[b] + max_voxels;
            if (rc < w.frame_base[b + 1]) cutoff = w.vox_rec[rc].w;
        }
        const int4 rec = w.vox_rec[r];
        const int cnt = rec.y;
        const int a0 = rec.z;
        // the max_points smallest indices, ascending, in lanes [0, max_points)
        int v = lane < cnt ? w.arena[a0 + lane] : kIdle;
        v = bitonic64_asc(v, lane);
        const int chunk = 64 - max_points;
        for (int done = 64; done < cnt; done += chunk) {
            if (lane >= max_points) {
                const int j = done + (lane - max_points);
                v = j < cnt ? w.arena[a0 + j] : kIdle;
            }
            v = bitonic64_asc(v, lane);
        }
        const bool live = lane < max_points && v < cutoff;   // v == kIdle is never < cutoff
        const int num = __popcll(__ballot(live));
        if (lane < max_points) {
            float *dst = voxels + ((size_t)o * max_points + lane) * n_feat;
            if (live) {
                const float *src = pts + (size_t)v * stride + xyz_col;
                for (int f = 0; f < n_feat; ++f) dst[f] = src[f];
            } else {
                for (int f = 0; f < n_feat; ++f) dst[f] = 0.f;
            }
        }
        if (lane == 0) {
            const int g = rec.x;
            const int cx = g % nx, cy = (g / nx) % ny, cz = (g / (nx * ny)) % nz;
            reinterpret_cast<int4 *>(coords)[o] = make_int4(b, cz, cy, cx);
            num_points[o] = num;
        }
    }
}

}  // namespace

// The one-launch index kernel's owners wait for each other on the compute units they occupy, so two such launches in flight at once
// could each hold units the other's owners still need.  The library therefore takes the one-launch form only when it can tell that
// no other one-launch kernel of this process is in flight on the device: the previous one went to the SAME stream (stream order
// serialises them), or its completion event has fired.  Otherwise the three-launch form runs (same results, no waiting between
// workgroups other than the scan's look-back).  One record per device, behind a mutex; it only ever selects between two
// equivalent launch forms.  Inside a stream capture there is nothing to ask (the event would become part of the graph and the
// decision is taken once for every replay): a capturing caller that passes index_mode 1 takes on the rule itself — replay such
// graphs one at a time per device (the detector's frame pipeline has one encode lane).
namespace {
struct FusedInFlight { hipEvent_t done = nullptr; hipStream_t stream = nullptr; bool valid = false; };
std::mutex g_fused_mutex;
FusedInFlight g_fused[64];

// 1: take the one-launch form (and, outside capture, `note_fused_launch` must follow the launch); 0: take the three launches
int fused_allowed(hipStream_t s, bool *capturing) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    *capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    if (*capturing) return 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    FusedInFlight &f = g_fused[dev];
    if (!f.valid || f.stream == s) return 1;
    const hipError_t q = hipEventQuery(f.done);
    if (q != hipSuccess) (void)hipGetLastError();      // "not ready" is an answer, not a failure: it must not reach HVPR_CHECK_LAUNCH
    return q == hipSuccess ? 1 : 0;
}

void note_fused_launch(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    FusedInFlight &f = g_fused[dev];
    if (!f.done && hipEventCreateWithFlags(&f.done, hipEventDisableTiming) != hipSuccess) { f.done = nullptr; f.valid = false; return; }
    f.valid = hipEventRecord(f.done, s) == hipSuccess;
    f.stream = s;
}
}  // namespace

int hvpr_i_voxel_index(const VoxelizeArgs &a, const VoxWs &w, int32_t *voxel_offsets, bool for_encode, hipStream_t s,
                       const float *vfe_w1, const float *vfe_b0, const void *warm0, size_t warm0_bytes, const void *warm1,
                       size_t warm1_bytes, const WarmSmall *warm_small, int index_mode) {
    if (for_encode && a.n_feat != 4) return HVPR_ERR_UNSUPPORTED;
    // kernel experiments only: HVPR_INDEX_FUSED=0 / 1 overrides the caller's index_mode, HVPR_INDEX_AGENT=1 makes the one-launch form
    // take its placement-independent path (device-scope hand-offs) although its owners share an XCD
    static const int env_fused = [] { const char *e = getenv("HVPR_INDEX_FUSED"); return e ? atoi(e) : -1; }();
    static const int force_agent = [] { const char *e = getenv("HVPR_INDEX_AGENT"); return e ? atoi(e) : 0; }();
    bool fused = for_encode && (env_fused >= 0 ? env_fused != 0 : index_mode == 1) && a.n_points <= kFusedMaxOwners * 1024;
    bool capturing = false;
    std::unique_lock<std::mutex> guard(g_fused_mutex, std::defer_lock);
    if (fused) {
        guard.lock();                      // held until the launch is noted: two host threads cannot both see "nothing in flight"
        fused = fused_allowed(s, &capturing) != 0;
        if (!fused) guard.unlock();
    }
    if (fused) {
        // one launch: the owners of the point tiles (every 8th of the first 8 x tiles blocks), the padded-slot workgroup, the
        // warmers.  512 points per owner up to 16 384 points (32 owners: in-frame 16.3 us against 17.8 with 1024), 1024 beyond
        const int T = a.n_points <= kFusedMaxOwners * 512 ? 512 : 1024;
        const int pb = hvpr_cdiv(a.n_points, T);
        const int parts = kWarmParts * 256 / T;
        const int warmers = (vfe_w1 && (warm0 || warm1)) ? 8 * parts : 0;
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(8 * pb + 1 + warmers), dim3(T), 0, s, a.points, a.n_points, a.point_stride, a.xyz_col, a.frame_offsets,
                               a.batch, a.lo_x, a.lo_y, a.lo_z, a.vs_x, a.vs_y, a.vs_z, a.nx, a.ny, a.nz, a.max_voxels, w, voxel_offsets,
                               vfe_w1, vfe_b0, pb, parts, (const float4 *)warm0, (long long)(warm0 ? warm0_bytes / 16 : 0),
                               (const float4 *)warm1, (long long)(warm1 ? warm1_bytes / 16 : 0),
                               (warm_small && warmers) ? *warm_small : WarmSmall{}, (int *)(w.vfe_aux + 64), force_agent);
        };
        if (T == 1024) launch(k_index<1024>);
        else launch(k_index<512>);
        if (!capturing) note_fused_launch(s);
        HVPR_CHECK_LAUNCH();
        return HVPR_OK;
    }
    const int tiles = hvpr_cdiv(a.n_points, kScanTile);
    const int pblocks = hvpr_cdiv(a.n_points, 256);
    // K1: on the fused path it also hands out the arena slots
    const int have_slots = for_encode ? 1 : 0;
    if (have_slots)
        hipLaunchKernelGGL(k1_keys<true>, dim3(pblocks), dim3(256), 0, s, a.points, a.n_points, a.point_stride, a.xyz_col, a.frame_offsets,
                           a.batch, a.lo_x, a.lo_y, a.lo_z, a.vs_x, a.vs_y, a.vs_z, a.nx, a.ny, a.nz, w, tiles);
    else
        hipLaunchKernelGGL(k1_keys<false>, dim3(pblocks), dim3(256), 0, s, a.points, a.n_points, a.point_stride, a.xyz_col, a.frame_offsets,
                           a.batch, a.lo_x, a.lo_y, a.lo_z, a.vs_x, a.vs_y, a.vs_z, a.nx, a.ny, a.nz, w, tiles);
    if (a.n_points > 300000)
        hipLaunchKernelGGL(k2_scan<8>, dim3(hvpr_cdiv(a.n_points, kScanThreads * 8)), dim3(kScanThreads), 0, s, a.n_points,
                           a.frame_offsets, a.batch, w);
    else
        hipLaunchKernelGGL(k2_scan<kScanItems>, dim3(tiles), dim3(kScanThreads), 0, s, a.n_points, a.frame_offsets, a.batch, w);
    if (!for_encode) { vfe_w1 = nullptr; warm0 = warm1 = nullptr; }
    const int warmers = (vfe_w1 && (warm0 || warm1)) ? 8 * kWarmParts : 0;       // (the warmers sit behind the padded-slot workgroup)
    hipLaunchKernelGGL(k3_fill, dim3(pblocks + (vfe_w1 ? 1 : 0) + warmers), dim3(256), 0, s, a.n_points, a.frame_offsets, a.batch, a.max_voxels, w,
                       voxel_offsets, for_encode ? 1 : 0, have_slots, a.points, a.point_stride, a.xyz_col, vfe_w1, vfe_b0, pblocks, (const float4 *)warm0,
                       (long long)(warm0 ? warm0_bytes / 16 : 0), (const float4 *)warm1, (long long)(warm1 ? warm1_bytes / 16 : 0),
                       (warm_small && warmers) ? *warm_small : WarmSmall{}, (int *)(w.vfe_aux + 64));
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" size_t hvpr_voxelize_workspace_bytes(int batch, int n_points, int nx, int ny, int nz) {
    if (batch < 1 || n_points < 0 || nx < 1 || ny < 1 || nz < 1) return 0;
    return hvpr_vox_ws_bytes(batch, n_points > 0 ? n_points : 1, (long long)nx * ny * nz);
}

extern "C" int hvpr_voxelize_workspace_reset(void *workspace, size_t workspace_bytes, int batch, int n_points, int nx,
                                             int ny, int nz, hvpr_stream_t stream) {
    if (!workspace || batch < 1 || n_points < 0 || nx < 1 || ny < 1 || nz < 1) return HVPR_ERR_INVALID_ARG;
    const long long ncell = (long long)nx * ny * nz;
    if (workspace_bytes < hvpr_vox_ws_bytes(batch, n_points > 0 ? n_points : 1, ncell)) return HVPR_ERR_WORKSPACE;
    VoxWs w = hvpr_vox_carve(workspace, batch, n_points > 0 ? n_points : 1, ncell);
    hipLaunchKernelGGL(k_reset, dim3(1024), dim3(256), 0, (hipStream_t)stream, w.cell_first, w.cell_count, batch * ncell, w.sync);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_voxelize_workspace_status(const void *workspace, size_t workspace_bytes, int batch, int n_points, int nx, int ny,
                                              int nz, hvpr_stream_t stream) {
    if (!workspace || batch < 1 || n_points < 0 || nx < 1 || ny < 1 || nz < 1) return HVPR_ERR_INVALID_ARG;
    const long long ncell = (long long)nx * ny * nz;
    if (workspace_bytes < hvpr_vox_ws_bytes(batch, n_points > 0 ? n_points : 1, ncell)) return HVPR_ERR_WORKSPACE;
    const VoxWs w = hvpr_vox_carve(const_cast<void *>(workspace), batch, n_points > 0 ? n_points : 1, ncell);
    int word = 0;
    if (hipMemcpyAsync(&word, w.sync + kSyncError, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return HVPR_ERR_LAUNCH;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return HVPR_ERR_LAUNCH;
    return word == 0 ? HVPR_OK : HVPR_ERR_TIMEOUT;
}

extern "C" int hvpr_voxelize_f32(const float *points, int n_points, int point_stride, int xyz_col, int n_feat,
                                 const int32_t *frame_offsets, int batch, float lo_x, float lo_y, float lo_z, float vs_x,
                                 float vs_y, float vs_z, int nx, int ny, int nz, int max_points, int max_voxels,
                                 int cap_mode, float *voxels, int32_t *coords, int32_t *num_points,
                                 int32_t *voxel_offsets, int capacity, void *workspace, size_t workspace_bytes,
                                 int ws_max_batch, int ws_max_points, hvpr_stream_t stream) {
    if (!points || !frame_offsets || !voxels || !coords || !num_points || !voxel_offsets || !workspace)
        return HVPR_ERR_INVALID_ARG;
    if (batch < 1 || n_points < 0 || n_feat < 3 || xyz_col < 0 || point_stride < xyz_col + n_feat || nx < 1 || ny < 1 ||
        nz < 1 || max_points < 1 || max_voxels < 1 || capacity < 0 || (cap_mode != 0 && cap_mode != 1))
        return HVPR_ERR_INVALID_ARG;
    if (max_points > 63 || (long long)batch * nx * ny * nz > 0x7ffffff0ll) return HVPR_ERR_UNSUPPORTED;
    const long long ncell = (long long)nx * ny * nz;
    // the workspace is carved with the dimensions it was sized and reset for, not with this call's
    if (ws_max_batch < batch || ws_max_points < n_points || ws_max_points < 1) return HVPR_ERR_WORKSPACE;
    if (workspace_bytes < hvpr_vox_ws_bytes(ws_max_batch, ws_max_points, ncell)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (n_points == 0) {
        if (hipMemsetAsync(voxel_offsets, 0, sizeof(int) * (batch + 1), s) != hipSuccess) return HVPR_ERR_LAUNCH;
        return HVPR_OK;
    }
    VoxWs w = hvpr_vox_carve(workspace, ws_max_batch, ws_max_points, ncell);
    const VoxelizeArgs a{points, n_points, point_stride, xyz_col, n_feat, frame_offsets, batch, lo_x, lo_y, lo_z, vs_x, vs_y, vs_z,
                         nx, ny, nz, max_points, max_voxels, cap_mode};
    const int st = hvpr_i_voxel_index(a, w, voxel_offsets, false, s);
    if (st != HVPR_OK) return st;
    int gblocks = hvpr_cdiv(n_points, 4);
    if (gblocks > 2048) gblocks = 2048;
    hipLaunchKernelGGL(k4_gather, dim3(gblocks), dim3(256), 0, s, points, point_stride, xyz_col, n_feat, batch, nx, ny,
                       nz, max_points, max_voxels, cap_mode, w, voxel_offsets, voxels, coords, num_points, capacity);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
