// Wave-level selection helpers shared by the memory read-out (memory_scatter.hip) and the point <-> pillar top-k (point_topk.hip):
// order-preserving float keys, ballot radix selects, the fp16 operand conversion and the 32-vector butterfly sum.
#pragma once
#include "common.h"

namespace hvpr_sel {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned ord_bits(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_to_float(unsigned ub) {
    return __uint_as_float((ub & 0x80000000u) ? (ub & 0x7fffffffu) : ~ub);
}
// fp32 -> fp16 bits, round to nearest even (v_cvt_f16_f32 under the default rounding mode)
__device__ __forceinline__ unsigned f16_rne(float x) {
    const _Float16 h = (_Float16)x;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}

// k-th largest (k >= 1) of one key per lane: the largest v with count(key >= v) >= k, found bit by bit with ballots —
// compares and scalar popcounts only, no cross-lane data movement.  Lanes that do not take part pass key 0.
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

// lane l <-> lane l ^ 4 inside every row of 16: two DPP moves (row_shl:4 into banks 0 and 2, row_shr:4 into banks 1 and 3)
__device__ __forceinline__ float xchg_xor4(float v) {
    const int iv = __float_as_int(v);
    int r = __builtin_amdgcn_update_dpp(0, iv, 0x104, 0xf, 0x5, false);
    r = __builtin_amdgcn_update_dpp(r, iv, 0x114, 0xf, 0xa, false);
    return __int_as_float(r);
}

// Sums over the 64 lanes of 32 vectors at once: on return lane l holds sum_lanes x[l & 31].  A vector-halving butterfly —
// at lane bit b every lane keeps the half of the vectors whose index has bit b equal to its own lane bit and adds its
// partner's copy of them — 31 + 1 exchange-adds instead of 32 x 6.  Each level adds the same two numbers the plain
// butterfly of hvpr_reduce_sum<64> adds (addition is commutative), so the result is bit-identical to 32 separate reductions.
__device__ __forceinline__ float wave_sum32(float (&x)[32], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8, b4 = lane & 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float keep = b0 ? x[2 * i + 1] : x[2 * i], send = b0 ? x[2 * i] : x[2 * i + 1];
        x[i] = keep + hvpr_dpp<0xB1>(send);                       // quad_perm [1,0,3,2]: lane ^ 1
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b1 ? x[2 * i + 1] : x[2 * i], send = b1 ? x[2 * i] : x[2 * i + 1];
        x[i] = keep + hvpr_dpp<0x4E>(send);                       // quad_perm [2,3,0,1]: lane ^ 2
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b2 ? x[2 * i + 1] : x[2 * i], send = b2 ? x[2 * i] : x[2 * i + 1];
        x[i] = keep + xchg_xor4(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b3 ? x[2 * i + 1] : x[2 * i], send = b3 ? x[2 * i] : x[2 * i + 1];
        x[i] = keep + hvpr_dpp<0x128>(send);                      // row_ror:8: lane ^ 8
    }
    {
        const float keep = b4 ? x[1] : x[0], send = b4 ? x[0] : x[1];
        // v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second: with the same value
        // in both, result 0 holds the even row's value and result 1 the odd row's in BOTH rows of a pair — the partner's
        // (lane ^ 16) value is result 1 for an even-row lane and result 0 for an odd-row lane
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(send), __float_as_uint(send), false, false);
        x[0] = keep + __uint_as_float(b4 ? r[0] : r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0]), __float_as_uint(x[0]), false, false);
        x[0] = __uint_as_float(r[0]) + __uint_as_float(r[1]);     // lane ^ 32
    }
    return x[0];
}


}  // namespace hvpr_sel
