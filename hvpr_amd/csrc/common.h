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

__device__ __forceinline__ int hvpr_lane() { return threadIdx.x & 63; }

// butterfly reductions over the full 64-lane wave or within 32-lane halves
template <int WIDTH>
__device__ __forceinline__ float hvpr_reduce_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float hvpr_reduce_max(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
