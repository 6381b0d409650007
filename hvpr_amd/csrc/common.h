// Shared device/host helpers for the hvpr_amd HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hvpr_amd.h"

#define HVPR_WAVE 64

#define HVPR_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return HVPR_ERR_LAUNCH; \
    } while (0)

static inline int hvpr_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// SyncBatchNorm (abi.hip): sums buf[0..n) over the ranks through the caller's hook, ordered on `s`; 0 without a hook / on success
int hvpr_i_bn_allreduce(double *buf, int n, hipStream_t s);

// 256-byte aligned carve helper for workspaces
struct hvpr_carver {
    char *base;
    size_t off;
    explicit hvpr_carver(void *p) : base((char *)p), off(0) {}
    template <typename T>
    T *take(size_t n) {
        T *r = (T *)(base + off);
        off += ((n * sizeof(T) + 255) / 256) * 256;
        return r;
    }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: remember it per device (bit d of a
// per-call-site mask), not once per process.  The call is idempotent, so a lost update only repeats it.
static inline int hvpr_ensure_dyn_lds(const void *fn, int bytes, unsigned long long *done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64 && ((__atomic_load_n(done_mask, __ATOMIC_RELAXED) >> dev) & 1ull)) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64) __atomic_fetch_or(done_mask, 1ull << dev, __ATOMIC_RELAXED);
    return 0;
}

__device__ __forceinline__ int hvpr_lane() { return threadIdx.x & 63; }

// All-lanes reductions over the 64-lane wave or within its 32-lane halves, without the LDS crossbar: two quad permutes, the two
// row mirrors (after the quad steps every lane of a quad holds the same value, so mirroring combines exactly the groups an
// xor butterfly would), then v_permlane16_swap / v_permlane32_swap (gfx950) across the 16-lane rows.  Same reduction tree as
// the __shfl_xor butterfly — bit-identical results, commutativity aside nothing changes — at a few VALU cycles per step
// instead of a ds_bpermute round trip (~60+ cycles each: 95 of them were 3 us of the pillar VFE).
template <int CTRL>
__device__ __forceinline__ float hvpr_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
struct hvpr_op_sum { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct hvpr_op_max { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
template <int WIDTH, typename Op>
__device__ __forceinline__ float hvpr_reduce(float v, Op op) {
    static_assert(WIDTH == 32 || WIDTH == 64, "wave halves or the whole wave");
    v = op(v, hvpr_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
    v = op(v, hvpr_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
    v = op(v, hvpr_dpp<0x141>(v));   // row_half_mirror
    v = op(v, hvpr_dpp<0x140>(v));   // row_mirror
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    if (WIDTH == 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float hvpr_reduce_sum(float v) { return hvpr_reduce<WIDTH>(v, hvpr_op_sum()); }
template <int WIDTH>
__device__ __forceinline__ float hvpr_reduce_max(float v) { return hvpr_reduce<WIDTH>(v, hvpr_op_max()); }
