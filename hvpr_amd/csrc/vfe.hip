// a2 — fused per-pillar PointNet VFE (eval mode, BatchNorm folded into weights/bias by the caller).
// Replaces the ~10 PyTorch ops of PillarVFE_Scale.forward (pcdet/models/backbones_3d/vfe/pillar_vfe.py:184-221)
// and the two PFNLayer.forward calls (:29-49) with one launch.
//
// Round 4 form: POINTS are the matrix-core columns, not pillar slots.  A KITTI pillar holds 4.4 points on average (3.3 in the
// dense config), so a wave that gives each pillar 32 slot columns spends > 85 % of its matrix-core columns, sort-network lanes
// and reduction steps on padding.  Here one pass of a wave covers 32 CONSECUTIVE POSITIONS OF THE VOXELIZER'S ARENA (points
// grouped by voxel) = the points of ~6 (KITTI) / ~17 (dense config) whole pillars, and every per-pillar reduction is a
// wavefront SEGMENTED reduction over the pillar's run of columns.
//
//   gather      (fused encode form) a wave owns the pillars that START inside its window of kWin arena positions.  It loads
//               64 positions {index, rank, count, cell} + point in ONE round trip (K3 left them there, voxelize.hip), finds the
//               segment boundaries from the rank changes, and walks its pillars in passes of 32 columns that always begin at a
//               pillar start and hold whole pillars only.  Inside a pass a 32-lane bitonic network on DPP exchanges orders
//               (segment, point index, source column) keys — the arena is unordered inside a voxel — and the points follow
//               their keys by ds_bpermute.  Voxels with more than max_points points (~1 %) take a pass of their own after a
//               ballot radix select of the max_points smallest indices.
//   decoration  pillar mean by a Kogge-Stone scan that never crosses a segment start: the summation tree of a pillar depends
//               only on its own point order, not on where the pillar sits in the wave (bit-identical results whatever the
//               packing — single frame, batch, separate calls); 10-d decoration; masked INPUT (pillar_vfe.py:205-208).
//   layer 0     D0^T (16 ch x 32 points) = W0 . F^T on v_mfma_f32_32x32x2_f32 (exact fp32), bias, ReLU.
//   x_max       rows through LDS, lane = channel, a running max that restarts at every segment end -> one row per pillar.
//   layer 1     D1^T (64 ch x 32 points) = W1a . y0 per POINT, and C^T (64 ch x pillars) = W1b . x_max per PILLAR (the x_max
//               half of the concat is the same for every point of a pillar); max over a pillar's points of (D1 + C), bias,
//               ReLU — bias, the per-pillar constant and ReLU are monotone and commute with the max.
//   padded slot the reference masks the INPUT only, so a padded slot yields ReLU(folded bias) in layer 0 and takes part in both
//               maxes.  Its columns would be the same for every pillar: z0 = ReLU(b0) and
//               Z1 = W1a . z0 are computed once per wave (on the matrix cores, so they round like a real column) and enter
//               as a floor of the maxes of every pillar with fewer than max_points points.
//   scale       [n, |mean|, mean] -> 16 -> 32 per pillar, 3 + 8 MFMAs with pillars as columns.
//
// k_vfe_gather is the fused encode form (hvpr_encode_fwd_f32): it also writes voxels (optional) / coords / num_points and the
// pillar / scale cells of the NHWC canvases, whose other cells are cleared by extra workgroups of the same launch.
// k_vfe_rows is the separate-call form (hvpr_pillar_vfe_fwd_f32: padded voxels in, one pillar per pass); both run the same
// pass body, so the two forms agree bit for bit (tests/test_gpu_stage1.py).
#include "common.h"
#include "internal.h"

namespace {

constexpr int C0 = 16;    // layer-0 outputs (NUM_FILTERS[0] / 2)
constexpr int C1 = 64;    // layer-1 outputs
constexpr int CIN = 10;   // x y z r + cluster(3) + center(3)
constexpr int CS0 = 16, CS1 = 32;
constexpr int kWin = 28;    // arena positions whose pillars one wave owns (a pass is 32 wide: 28 leaves room for the last pillar's tail)
constexpr int kPA = 68;     // LDS row pitch (floats) of the 64-channel rows: the 16 lanes of a b128 service group land in distinct banks
constexpr int kMaxB = 64;   // frames whose offsets are staged in LDS (larger batches read them from memory)

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WaveLds {
    float a[32 * kPA];   // the per-pillar constant rows [q][kPA]; then the per-point layer-1 rows [col][kPA]
    float s[32 * 16 + 32 * 8];   // x_max rows [q][16]; scale-stream inputs + output row / cell [q][8]
};

// n / d for n < 2^31 with a precomputed multiplier: m = ceil(2^p / d), p = 31 + ceil(log2 d)
struct FastDiv {
    unsigned m;
    int p;
};
static inline FastDiv make_fastdiv(unsigned d) {
    int l = 0;
    while ((1ull << l) < d) ++l;
    const int p = 31 + l;
    return FastDiv{(unsigned)(((1ull << p) + d - 1) / d), p};
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) { return (int)(((unsigned long long)(unsigned)n * f.m) >> f.p); }

struct GatherSrc {
    const float *pts;
    int stride, xyz_col, batch, nx, ny, nz, max_voxels, cap_mode, capacity;
    VoxWs w;
    const int *voxel_offsets;
    float *voxels_out;      // optional
    int *coords_out, *num_out;
    float *spatial;         // NHWC canvas, 128 channels per cell: pillar features in [0, 64), memory read-out in [64, 128)
    int spatial_channels;
    float *spatial_scale;   // NHWC canvas, 32 channels per cell
    int work_blocks;        // the LAST work_blocks workgroups of the grid encode pillars, the ones before clear canvas cells
    ClearJob clear;         // ... (internal.h)
    int idx_bits;           // bits of the largest point index
    int n_points;
    FastDiv div_nx, div_cells;   // by nx, by nx * ny
    int split;              // one round of windows: a second set of workgroups takes every wave's passes after the first (below)
};

struct VfeParams {
    float vsx, vsy, vsz, offx, offy, offz;
    const float *w0, *b0, *w1, *b1, *ws0, *bs0, *ws1, *bs1;
    float *pillar_features, *scale_features, *pillar_mask;
};

// weights in matrix-core operand layout, resident for the whole kernel
struct Wreg {
    float a0w[CIN / 2], b0h[8];   // layer 0: A operand, bias of this lane's 8 D rows
    float aw[2][8], awb[2][8];    // layer 1: the half that multiplies y0 (per point) / x_max (per pillar)
    float as0[3], bs0h[8], as1[8];   // scale stream
    float b1c, z0c;               // lane = channel constants of the two LDS loops
};

// lane ^ j exchanges without the LDS crossbar (common.h, select.h use the same ones)
__device__ __forceinline__ unsigned xchg(unsigned v, int j) {
    const int iv = (int)v;
    if (j == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xf, 0xf, true);
    if (j == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xf, 0xf, true);
    if (j == 4) {
        int r = __builtin_amdgcn_update_dpp(0, iv, 0x104, 0xf, 0x5, false);
        return (unsigned)__builtin_amdgcn_update_dpp(r, iv, 0x114, 0xf, 0xa, false);
    }
    if (j == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, iv, 0x128, 0xf, 0xf, true);
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // j == 16
    return (threadIdx.x & 16) ? r[0] : r[1];
}
// ascending sort of one key per column inside each 32-lane half
__device__ __forceinline__ unsigned bitonic32_asc(unsigned v, int col) {
#pragma unroll
    for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const unsigned o = xchg(v, j);
            const bool up = (col & k) == 0;
            const bool lower = (col & j) == 0;
            v = (lower == up) ? min(v, o) : max(v, o);
        }
    }
    return v;
}
__device__ __forceinline__ int bitonic_asc(int v, int lane, int width) {
    for (int k = 2; k <= width; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = __shfl_xor(v, j, 64);
            const bool up = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            v = (lower == up) ? min(v, o) : max(v, o);
        }
    }
    return v;
}

__device__ __forceinline__ f32x16 zero16() { return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// Stage the weights through LDS once per workgroup (every wave needs them in a gathered per-lane layout) and set up `zt`, this
// wave's 160-float table: the padded slot's layer-1 column Z1 by channel (taken from z1_pre when the index kernels left it in
// the workspace, else computed here), the scale stream's output bias, a row of -inf.  `stage`: the workgroup's staging area
// (3104 floats).  Contains one workgroup barrier.
__device__ __forceinline__ void load_weights(const VfeParams &v, float *stage, float *zt, Wreg &W, const float *z1_pre) {
    const int lane = threadIdx.x & 63, h = lane >> 5, slot = lane & 31;
    float *s_w1 = stage, *s_ws1 = stage + C1 * 36, *s_w0 = s_ws1 + CS1 * 20;
    for (int i = threadIdx.x; i < C1 * 32 / 4; i += 256) *(float4 *)&s_w1[(i >> 3) * 36 + (i & 7) * 4] = ((const float4 *)v.w1)[i];
    if (threadIdx.x < CS1 * CS0 / 4) *(float4 *)&s_ws1[(threadIdx.x >> 2) * 20 + (threadIdx.x & 3) * 4] = ((const float4 *)v.ws1)[threadIdx.x];
    if (threadIdx.x < C0 * CIN / 4) ((float4 *)s_w0)[threadIdx.x] = ((const float4 *)v.w0)[threadIdx.x];
    __syncthreads();
    // v_mfma_f32_32x32x2_f32: A lane (i = lane % 32, k = lane / 32), B lane (k = lane / 32, j = lane % 32), D register r of
    // lane (j, hh) = row 8 * (r / 4) + 4 * hh + r % 4, column j.  Everything is computed TRANSPOSED — rows = output channels
    // (weights are the A operand), columns = points or pillars — so that a layer's output already sits in the B-operand layout
    // of the next (the k index of an MFMA is a free permutation when A and B agree): lane (col, hh) ends layer 0 with channels
    // chm(r) = 8 * (r / 4) + 4 * hh + r % 4, r < 8, and feeds exactly those to layer 1.
#pragma unroll
    for (int t = 0; t < CIN / 2; ++t) W.a0w[t] = slot < C0 ? s_w0[slot * CIN + 2 * t + h] : 0.f;
    {
        const float4 lo = *(const float4 *)(v.b0 + 4 * h), hi = *(const float4 *)(v.b0 + 8 + 4 * h);
        W.b0h[0] = lo.x; W.b0h[1] = lo.y; W.b0h[2] = lo.z; W.b0h[3] = lo.w; W.b0h[4] = hi.x; W.b0h[5] = hi.y; W.b0h[6] = hi.z; W.b0h[7] = hi.w;
        const float4 sl = *(const float4 *)(v.bs0 + 4 * h), sh = *(const float4 *)(v.bs0 + 8 + 4 * h);
        W.bs0h[0] = sl.x; W.bs0h[1] = sl.y; W.bs0h[2] = sl.z; W.bs0h[3] = sl.w; W.bs0h[4] = sh.x; W.bs0h[5] = sh.y; W.bs0h[6] = sh.z; W.bs0h[7] = sh.w;
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 w4 = *(const float4 *)&s_w1[(32 * mb + slot) * 36 + 8 * q + 4 * h];
            W.aw[mb][4 * q] = w4.x; W.aw[mb][4 * q + 1] = w4.y; W.aw[mb][4 * q + 2] = w4.z; W.aw[mb][4 * q + 3] = w4.w;
            // the x_max half: lane (q, h) of the B operand supplies x_max channel 8 h + t to MFMA t
            const float4 x4 = *(const float4 *)&s_w1[(32 * mb + slot) * 36 + C0 + 8 * h + 4 * q];
            W.awb[mb][4 * q] = x4.x; W.awb[mb][4 * q + 1] = x4.y; W.awb[mb][4 * q + 2] = x4.z; W.awb[mb][4 * q + 3] = x4.w;
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)   // scale layer 0: input k = 3 h + t of [n, |mean|, mean_x, mean_y, mean_z, 0]
        W.as0[t] = (slot < CS0 && 3 * h + t < 5) ? v.ws0[slot * 5 + 3 * h + t] : 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float4 w4 = *(const float4 *)&s_ws1[slot * 20 + 8 * q + 4 * h];
        W.as1[4 * q] = w4.x; W.as1[4 * q + 1] = w4.y; W.as1[4 * q + 2] = w4.z; W.as1[4 * q + 3] = w4.w;
    }
    W.b1c = v.b1[lane];
    W.z0c = fmaxf(0.f + v.b0[lane & 15], 0.f);
    if (z1_pre) zt[lane] = z1_pre[lane];
    else hvpr_vfe_padded_slot(W.aw, W.b0h, zt);
    if (lane < CS1) zt[C1 + lane] = v.bs1[lane];
    zt[96 + lane] = -INFINITY;
}

#ifdef HVPR_EXP_TIMING
#define VFE_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tstamp[i] = __builtin_readcyclecounter(); } while (0)
#define VFE_TARG , long long *tstamp
#define VFE_TPASS , tstamp
#else
#define VFE_STAMP(i)
#define VFE_TARG
#define VFE_TPASS
#endif

// v_max_f32 as is: fmaxf() first quiets both operands (two more instructions per max under the IEEE mode bit)
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// lane i <- lane i - D of the same 16-lane row (row_shr), ANDed with a per-lane mask; lanes without a source get 0
template <int D>
__device__ __forceinline__ float row_up_and(float x, int mask) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x110 + D, 0xf, 0xf, true) & mask);
}
// rows 1 and 3 <- lane 15 of the row before (row_bcast:15), rows 0 and 2 get 0
__device__ __forceinline__ float row_carry_and(float x, int mask) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xa, 0xf, false) & mask);
}
__device__ __forceinline__ float fand(float x, int mask) { return __int_as_float(__float_as_int(x) & mask); }
// the same two moves on a double (both halves), for the pillar sums
template <int D>
__device__ __forceinline__ double row_up_and_d(double x, int mask) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x110 + D, 0xf, 0xf, true) & mask;
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x110 + D, 0xf, 0xf, true) & mask;
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double row_carry_and_d(double x, int mask) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x142, 0xa, 0xf, false) & mask;
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x142, 0xa, 0xf, false) & mask;
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// Pillar sums that do not depend on the ORDER of the points (the arena is unordered inside a voxel): every coordinate is taken
// to a fixed-point grid as an integer-valued double, 2^k units per metre with k from the pillar's own position — all its points
// lie within the cell, |v| < bound, and 32 * bound * 2^k <= 2^52 — so the (up to 32) terms add EXACTLY in any order; values keep
// every bit above 2^-k m (k = 41 at 47 m, more near the origin).  k of a bound: bound < 2^e, k = 47 - e.
__device__ __forceinline__ int sum_scale(float bound) { return 47 - ((int)((__float_as_uint(bound) >> 23) & 0xffu) - 126); }
__device__ __forceinline__ double to_grid(float v, int k) { return trunc(ldexp((double)v, k)); }

// One pass: 32 point columns (both 32-lane halves hold the same columns, lane = (col, h)) that belong to whole pillars.
//   pt, live     the column's point; live = it is a real point of a pillar of this pass (others carry zeros)
//   inmax        the column takes part in the maxes (a live point, or the stand-in column of a pillar without points)
//   proc         the column belongs to a pillar of this pass; seg0 = first column of its pillar, n = live points of the pillar
//   cd, row, cell   the pillar's [b, z, y, x], output row and canvas cell
//   hz           the column's pillar has a padded slot (n < max_points)
//   endmask      bit c: column c is the last column of a pillar of this pass;  startsproc  bit c: column c is the first one
template <bool GATHER>
__device__ __forceinline__ void vfe_pass(const Wreg &W, WaveLds &L, const float *zt, const VfeParams &v, float4 pt, bool live, bool inmax,
                                         bool proc, bool hz, int n, int seg0, int4 cd, int row, int cell, unsigned endmask,
                                         unsigned startsproc, float *spatial, int spatial_channels, float *spatial_scale VFE_TARG) {
    const int lane = threadIdx.x & 63, h = lane >> 5, col = lane & 31;
    VFE_STAMP(3);
    endmask = uni(endmask); startsproc = uni(startsproc);
    const int q = proc ? __popc(startsproc & (0xffffffffu >> (31 - col))) - 1 : 0;   // pillar of this column inside the pass
    const bool isend = (endmask >> col) & 1u;
    const float ninf = -INFINITY;
    // all ones where column col - d belongs to the same pillar
    const int g1 = col - 1 >= seg0 ? -1 : 0, g2 = col - 2 >= seg0 ? -1 : 0, g4 = col - 4 >= seg0 ? -1 : 0, g8 = col - 8 >= seg0 ? -1 : 0;

    // ---- decoration (pillar_vfe.py:187-208) -----------------------------------------------------------------------------
    // pillar sums: exact fixed-point terms (above), segmented inclusive scan along the columns (DPP inside the 16-lane rows + one
    // carry across the row border); the pillar's LAST column ends up with the sum over the pillar
    const float cx = (float)cd.w * v.vsx + v.offx, cy = (float)cd.z * v.vsy + v.offy, cz = (float)cd.y * v.vsz + v.offz;   // cell centre
    const int kx = sum_scale(fabsf(cx) + v.vsx), ky = sum_scale(fabsf(cy) + v.vsy), kz = sum_scale(fabsf(cz) + v.vsz);
    const int gx = (col >= 16 && seg0 < 16) ? -1 : 0;
    double sx = to_grid(pt.x, kx), sy = to_grid(pt.y, ky), sz = to_grid(pt.z, kz);   // columns that are not live carry zeros (pillar_vfe.py:187)
    sx += row_up_and_d<1>(sx, g1); sy += row_up_and_d<1>(sy, g1); sz += row_up_and_d<1>(sz, g1);
    sx += row_up_and_d<2>(sx, g2); sy += row_up_and_d<2>(sy, g2); sz += row_up_and_d<2>(sz, g2);
    sx += row_up_and_d<4>(sx, g4); sy += row_up_and_d<4>(sy, g4); sz += row_up_and_d<4>(sz, g4);
    sx += row_up_and_d<8>(sx, g8); sy += row_up_and_d<8>(sy, g8); sz += row_up_and_d<8>(sz, g8);
    sx += row_carry_and_d(sx, gx); sy += row_carry_and_d(sy, gx); sz += row_carry_and_d(sz, gx);
    const int endcol = col + __ffs((int)((endmask >> col) | 0x80000000u)) - 1;     // last column of this column's pillar
    const float fn = (float)n;
    const float mx = __shfl((float)ldexp(sx, -kx), endcol, 32) / fn, my = __shfl((float)ldexp(sy, -ky), endcol, 32) / fn,
                mz = __shfl((float)ldexp(sz, -kz), endcol, 32) / fn;
    // B operand of layer 0: lane (col, h) supplies input 2 t + h to MFMA t.  The mask is applied to the INPUT (:205-208): a
    // padded slot still yields ReLU(folded bias) and takes part in both maxes
    float fb0 = h ? pt.y : pt.x, fb1 = h ? pt.w : pt.z, fb2 = h ? pt.y - my : pt.x - mx, fb3 = h ? pt.x - cx : pt.z - mz,
          fb4 = h ? pt.z - cz : pt.y - cy;
    if (!live) { fb0 = 0.f; fb1 = 0.f; fb2 = 0.f; fb3 = 0.f; fb4 = 0.f; }
    if (proc && col == seg0 && h == 0) {   // the pillar's scale-stream inputs (pillar_vfe.py:213-214) + where its outputs go
        const float nrm = sqrtf(mx * mx + my * my + mz * mz);
        *(float4 *)&L.s[512 + q * 8] = make_float4(fn, nrm, mx, my);
        *(float4 *)&L.s[512 + q * 8 + 4] = make_float4(mz, 0.f, __int_as_float(row), __int_as_float(cell));
    }
    VFE_STAMP(4);

    // ---- layer 0: (16 x 10) . (10 x 32 points) ---------------------------------------------------------------------------
    f32x16 acc0 = zero16();
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a0w[0], fb0, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a0w[1], fb1, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a0w[2], fb2, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a0w[3], fb3, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a0w[4], fb4, acc0, 0, 0, 0);
    float y0[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) y0[r] = fmaxf(acc0[r] + W.b0h[r], 0.f);

    // ---- layer 1, the per-point half: (64 x 16) . (y0 x 32 points) -----------------------------------------------------------
    f32x16 acca[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        acca[mb] = zero16();
#pragma unroll
        for (int t = 0; t < 8; ++t) acca[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.aw[mb][t], y0[t], acca[mb], 0, 0, 0);
    }
    VFE_STAMP(5);

    // ---- x_max per pillar: segmented inclusive max-scan along the columns (DPP inside the 16-lane rows + one carry across the
    // row border); the pillar's last column ends up with the max over the pillar and leaves it as row q of the pillar table
    {
        // layer-0 outputs are >= 0 (ReLU), so 0 is the neutral element of this scan: a masked-off or missing neighbour reads 0
        const int im = inmax ? -1 : 0, hm = hz ? -1 : 0;
        float xr[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float x = fand(y0[r], im);
            x = vmax(x, row_up_and<1>(x, g1));
            x = vmax(x, row_up_and<2>(x, g2));
            x = vmax(x, row_up_and<4>(x, g4));
            x = vmax(x, row_up_and<8>(x, g8));
            x = vmax(x, row_carry_and(x, gx));
            xr[r] = vmax(x, fand(fmaxf(0.f + W.b0h[r], 0.f), hm));   // the padded slot's layer-0 output
        }
        if (isend && proc) {   // lane (col, h) holds channels 4 h .. 4 h + 3 and 8 + 4 h .. 8 + 4 h + 3
            *(float4 *)&L.s[q * 16 + 4 * h] = make_float4(xr[0], xr[1], xr[2], xr[3]);
            *(float4 *)&L.s[q * 16 + 8 + 4 * h] = make_float4(xr[4], xr[5], xr[6], xr[7]);
        }
    }
    VFE_STAMP(6);

    // ---- per-pillar matrix products, pillars as columns: C = W1b . x_max, and the scale stream ---------------------------
    {
        const float4 xa = *(const float4 *)&L.s[col * 16 + 8 * h], xb = *(const float4 *)&L.s[col * 16 + 8 * h + 4];
        const float4 sa = *(const float4 *)&L.s[512 + col * 8], sb = *(const float4 *)&L.s[512 + col * 8 + 4];
        const float sci0 = h ? sa.w : sa.x, sci1 = h ? sb.x : sa.y, sci2 = h ? sb.y : sa.z;   // input 3 h + t
        const int orow = __float_as_int(sb.z), ocell = __float_as_int(sb.w);
        f32x16 accs = zero16();
        accs = __builtin_amdgcn_mfma_f32_32x32x2f32(W.as0[0], sci0, accs, 0, 0, 0);
        accs = __builtin_amdgcn_mfma_f32_32x32x2f32(W.as0[1], sci1, accs, 0, 0, 0);
        accs = __builtin_amdgcn_mfma_f32_32x32x2f32(W.as0[2], sci2, accs, 0, 0, 0);
        f32x16 accc[2];
        accc[0] = zero16(); accc[1] = zero16();
        accc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[0][0], xa.x, accc[0], 0, 0, 0);
        accc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[0][1], xa.y, accc[0], 0, 0, 0);
        accc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[0][2], xa.z, accc[0], 0, 0, 0);
        accc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[0][3], xa.w, accc[0], 0, 0, 0);
        accc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[0][4], xb.x, accc[0], 0, 0, 0);
        accc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[0][5], xb.y, accc[0], 0, 0, 0);
        accc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[0][6], xb.z, accc[0], 0, 0, 0);
        accc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[0][7], xb.w, accc[0], 0, 0, 0);
        float hid[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) hid[r] = fmaxf(accs[r] + W.bs0h[r], 0.f);
        accc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[1][0], xa.x, accc[1], 0, 0, 0);
        accc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[1][1], xa.y, accc[1], 0, 0, 0);
        accc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[1][2], xa.z, accc[1], 0, 0, 0);
        accc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[1][3], xa.w, accc[1], 0, 0, 0);
        accc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[1][4], xb.x, accc[1], 0, 0, 0);
        accc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[1][5], xb.y, accc[1], 0, 0, 0);
        accc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[1][6], xb.z, accc[1], 0, 0, 0);
        accc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.awb[1][7], xb.w, accc[1], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4 *)&L.a[col * kPA + 32 * mb + 8 * g + 4 * h] = make_float4(accc[mb][4 * g], accc[mb][4 * g + 1], accc[mb][4 * g + 2], accc[mb][4 * g + 3]);
        f32x16 accs1 = zero16();
#pragma unroll
        for (int t = 0; t < 8; ++t) accs1 = __builtin_amdgcn_mfma_f32_32x32x2f32(W.as1[t], hid[t], accs1, 0, 0, 0);
        const int n_pillars = __popc(endmask);
        if (col < n_pillars) {   // lane (pillar, hh) holds scale channels 8 g + 4 hh .. + 3
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bs = *(const float4 *)&zt[C1 + 8 * g + 4 * h];
                const float4 o = make_float4(fmaxf(accs1[4 * g] + bs.x, 0.f), fmaxf(accs1[4 * g + 1] + bs.y, 0.f),
                                             fmaxf(accs1[4 * g + 2] + bs.z, 0.f), fmaxf(accs1[4 * g + 3] + bs.w, 0.f));
                *(float4 *)&v.scale_features[(size_t)orow * CS1 + 8 * g + 4 * h] = o;
                if (GATHER) *(float4 *)&spatial_scale[(size_t)ocell * CS1 + 8 * g + 4 * h] = o;
            }
        }
    }
    VFE_STAMP(7);

    // ---- per point: floor by the padded slot's column (at ONE column of the pillar: the max over the pillar sees it), add
    // the pillar's constant; rows back to LDS.  Columns outside the maxes read a row of -inf as their constant
    {
        const float *crow = inmax ? &L.a[q * kPA] : &zt[96];
        // (at the pillar's first column that takes part in the maxes: under the V1 cap a pillar's dropped points sit anywhere)
        const unsigned inmask = (unsigned)(__ballot(inmax) & 0xffffffffull);
        const bool floor_here = hz && proc && inmax && ((inmask >> seg0) << seg0 & ((1u << col) - 1u)) == 0u;
        const float *zrow = floor_here ? zt : &zt[96];
        float4 cv[2][4], zv[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                cv[mb][g] = *(const float4 *)&crow[32 * mb + 8 * g + 4 * h];
                zv[mb][g] = *(const float4 *)&zrow[32 * mb + 8 * g + 4 * h];
            }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 o;
                o.x = vmax(acca[mb][4 * g], zv[mb][g].x) + cv[mb][g].x;
                o.y = vmax(acca[mb][4 * g + 1], zv[mb][g].y) + cv[mb][g].y;
                o.z = vmax(acca[mb][4 * g + 2], zv[mb][g].z) + cv[mb][g].z;
                o.w = vmax(acca[mb][4 * g + 3], zv[mb][g].w) + cv[mb][g].w;
                *(float4 *)&L.a[col * kPA + 32 * mb + 8 * g + 4 * h] = o;
            }
    }
    VFE_STAMP(8);

    // ---- max over each pillar's points, bias, ReLU: lane = channel, one coalesced row per pillar --------------------------
    {
        float av[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) av[c] = L.a[c * kPA + lane];
        float m = ninf;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            m = vmax(m, av[c]);
            if ((endmask >> c) & 1u) {
                const float o = fmaxf(m + W.b1c, 0.f);
                const int r = __builtin_amdgcn_readlane(row, c);
                v.pillar_features[(size_t)r * C1 + lane] = o;
                if (GATHER) spatial[(size_t)__builtin_amdgcn_readlane(cell, c) * spatial_channels + lane] = o;   // pointpillar_scatter.py:192,207
                m = ninf;
            }
        }
    }
    VFE_STAMP(9);
}

__global__ void __launch_bounds__(256, 2) k_vfe_gather(int P, VfeParams v, GatherSrc g) {
    const int fill_blocks = (int)gridDim.x - g.work_blocks;   // dispatched first: they are the bandwidth work
    if ((int)blockIdx.x < fill_blocks) {
#ifdef HVPR_EXP_TIMING
        const long long f0 = __builtin_amdgcn_s_memrealtime();
#endif
        hvpr_canvas_clear(g.clear, blockIdx.x, fill_blocks);
#ifdef HVPR_EXP_TIMING
        if ((blockIdx.x == 0 || blockIdx.x == fill_blocks - 1 || blockIdx.x == fill_blocks / 2) && threadIdx.x == 0)
            printf("vfe-abs fill blk %d: %lld .. %lld (x10 ns)\n", (int)blockIdx.x, f0, (long long)__builtin_amdgcn_s_memrealtime());
#endif
        return;
    }
#ifdef HVPR_EXP_TIMING
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    __shared__ __attribute__((aligned(16))) WaveLds s_wave[4];
    __shared__ __attribute__((aligned(16))) float s_stage[C1 * 36 + CS1 * 20 + C0 * CIN];
    __shared__ __attribute__((aligned(16))) float s_zt[4][160];
    __shared__ int s_sel[4][32];
    __shared__ __attribute__((aligned(16))) float4 s_selp[4][32];
    __shared__ int s_voff[kMaxB + 1], s_fbase[kMaxB + 1], s_cut[kMaxB];
#ifdef HVPR_EXP_TIMING
    long long tstamp[10];
    tstamp[0] = __builtin_readcyclecounter();
    int t_passes = 0;
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_front0 = 0, t_win = 0, t_winacc = 0;
#endif
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31;
    // One frame is a single round of windows with most SIMDs idle, and the launch lasts as long as its slowest wave: a wave
    // whose window needs a second pass (a voxel with more points than slots, a pillar that did not fit) takes twice as long as
    // the others.  With g.split the upper half of the work groups walks the same windows and takes over every pass AFTER the
    // first (role 1), the lower half stops after its first pass (role 0): same results, the chain of every wave is one pass long.
    const int role_blocks = g.split ? g.work_blocks / 2 : g.work_blocks;
    const int wblk = (int)blockIdx.x - fill_blocks;
    const int role = wblk >= role_blocks ? 1 : 0;
    const long long wave0 = ((long long)(wblk - role * role_blocks) * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)role_blocks * blockDim.x) >> 6;

    // the first window is requested before anything else: the weight staging below hides its round trip
    int4 rec = make_int4(0, -2, 0, 0);
    float4 apt = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const long long pos = wave0 * kWin - 1 + lane;
        if (pos >= 0 && pos < g.n_points) { rec = g.w.arena_rec[pos]; apt = g.w.arena_pt[pos]; }
    }
    const int tot = *g.w.arena_total;
    // frame tables: output row of a voxel = voxel_offsets[b] + rank - frame_base[b]; V1 cap: first point index of the first
    // dropped voxel of the frame
    for (int b = threadIdx.x; g.batch > 1 && b <= g.batch && b <= kMaxB; b += 256) {
        s_voff[b] = g.voxel_offsets[b];
        s_fbase[b] = g.w.frame_base[b];
        if (b < g.batch && b < kMaxB) {
            int cutoff = kIdle;
            if (g.cap_mode == 1) {
                const int rc = g.w.frame_base[b] + g.max_voxels;
                if (rc < g.w.frame_base[b + 1]) cutoff = g.w.vox_rec[rc].w;
            }
            s_cut[b] = cutoff;
        }
    }
    int cut1 = kIdle;   // one frame: the cutoff is a wave-uniform scalar
    if (g.batch == 1 && g.cap_mode == 1) {
        const int rc = g.w.frame_base[0] + g.max_voxels;
        if (rc < g.w.frame_base[1]) cut1 = g.w.vox_rec[rc].w;
    }
    Wreg W;
    load_weights(v, s_stage, s_zt[wid], W, g.w.vfe_aux);   // a barrier inside: the frame tables are visible afterwards
    WaveLds &L = s_wave[wid];
    VFE_STAMP(1);
    const int cells_per_frame = g.nx * g.ny;
    const unsigned idxmask = (1u << g.idx_bits) - 1u;

    for (long long win = wave0; win * kWin < tot; win += n_waves) {
        const long long base = win * kWin - 1;     // lane l looks at arena position base + l; the wave owns positions base + 1 .. base + kWin
        // the next window of this wave is requested now and lands while this one is worked on
        int4 rec_next = make_int4(0, -2, 0, 0);
        float4 apt_next = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            const long long pos = (win + n_waves) * kWin - 1 + lane;
            if (pos < g.n_points && (win + n_waves) * kWin < tot) { rec_next = g.w.arena_rec[pos]; apt_next = g.w.arena_pt[pos]; }
        }
        if (base + lane < 0 || base + lane >= tot) rec.y = -2;
        // a pillar starts where the rank changes (dropped voxels all carry -1: they merge into blobs nobody processes)
        const int rprev = __shfl_up(rec.y, 1, 64);
        const bool flag = lane >= 1 && rec.y != rprev;
        const unsigned long long startmask = __ballot(flag);
        unsigned long long todo = __ballot(flag && lane <= kWin && rec.y >= 0);

        bool first = true;
#ifdef HVPR_EXP_TIMING
        t_win = __builtin_readcyclecounter();
#endif
        while (todo != 0ull) {
#ifdef HVPR_EXP_TIMING
            t_front0 = __builtin_readcyclecounter();
            t_winacc += t_front0 - t_win;
#endif
            const bool mine = !g.split || (first ? role == 0 : role == 1);
            if (g.split && role == 0 && !first) break;
            first = false;
            const int s = __ffsll((long long)todo) - 1;   // wave-uniform: the pass begins at this pillar start
            const int cnt_s = __builtin_amdgcn_readlane(rec.z, s);
            float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
            bool live = false, proc = false;
            int n = 0, seg0 = 0, row = 0, cell = 0, cutoff = kIdle, b = 0;
            unsigned endmask = 0u, startsproc = 0u;
            if (cnt_s > P) {
                // ---- a voxel with more points than slots (~1 %): a pass of its own.  Select the P smallest point indices of
                // its arena segment, ascending, then fetch the points
                todo &= todo - 1ull;
                if (!mine) continue;
                const long long a0 = base + s;
                const int cnt = cnt_s, slot = col;
                cell = __builtin_amdgcn_readlane(rec.w, s);
                const int rk = __builtin_amdgcn_readlane(rec.y, s);
                b = g.batch > 1 ? fdiv(cell, g.div_cells) : 0;
                row = g.batch > 1 ? (b < kMaxB ? s_voff[b] + rk - s_fbase[b] : g.voxel_offsets[b] + rk - g.w.frame_base[b]) : rk;
                if (row >= g.capacity) continue;
                if (g.cap_mode == 1) {
                    if (g.batch == 1) cutoff = cut1;
                    else if (b < kMaxB) cutoff = s_cut[b];
                    else {
                        const int rc = g.w.frame_base[b] + g.max_voxels;
                        if (rc < g.w.frame_base[b + 1]) cutoff = g.w.vox_rec[rc].w;
                    }
                }
                int vsel = kIdle;
                float4 psel = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cnt <= 512) {
                    // radix select: T = the P-th smallest index (indices are distinct), bit by bit with ballots.  The points come
                    // along with their records (one round trip); the selected (index, point) pairs are compacted through LDS
                    int vals[8];
                    float4 pv[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const bool in = r * 64 + lane < cnt;
                        vals[r] = in ? g.w.arena_rec[a0 + r * 64 + lane].x : kIdle;
                        pv[r] = in ? g.w.arena_pt[a0 + r * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    unsigned T = 0u;
                    for (int bit = g.idx_bits - 1; bit >= 0; --bit) {
                        const unsigned test = T | ((1u << bit) - 1u);
                        int c = 0;
#pragma unroll
                        for (int r = 0; r < 8; ++r) c += __popcll(__ballot((unsigned)vals[r] <= test));
                        if (c < P) T |= 1u << bit;
                    }
                    int fill = 0;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const bool sel = (unsigned)vals[r] <= T;
                        const unsigned long long m = __ballot(sel);
                        if (sel) {
                            const int at = fill + __popcll(m & ((1ull << lane) - 1ull));
                            s_sel[wid][at] = vals[r];
                            s_selp[wid][at] = pv[r];
                        }
                        fill += __popcll(m);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    vsel = slot < P ? ((volatile int *)s_sel[wid])[slot] : kIdle;
                    int at = slot;
                    if (g.voxels_out) {       // the optional `voxels` output lists the points by ascending index
                        const int key = bitonic_asc(vsel == kIdle ? kIdle : (vsel << 5) | slot, slot, 32);
                        vsel = key == kIdle ? kIdle : key >> 5;
                        at = key & 31;
                    }
                    if (slot < P) {
                        const volatile float *pp = (const volatile float *)&s_selp[wid][at];
                        psel = make_float4(pp[0], pp[1], pp[2], pp[3]);
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {
                    // very dense cell: 64-lane bitonic selection with chunked merging, then the points from the point array
                    int u = lane < cnt ? g.w.arena_rec[a0 + lane].x : kIdle;
                    u = bitonic_asc(u, lane, 64);
                    const int chunk = 64 - P;
                    for (int done = 64; done < cnt; done += chunk) {
                        if (lane >= P) {
                            const int j = done + (lane - P);
                            u = j < cnt ? g.w.arena_rec[a0 + j].x : kIdle;
                        }
                        u = bitonic_asc(u, lane, 64);
                    }
                    vsel = __shfl(u, slot, 64);       // ascending
                    if (slot < P && vsel != kIdle) {
                        const float *src = g.pts + (size_t)vsel * g.stride + g.xyz_col;
                        psel = make_float4(src[0], src[1], src[2], src[3]);
                    }
                }
                live = slot < P && vsel < cutoff;     // vsel == kIdle is never < cutoff
                n = __popcll(__ballot(live) & 0xffffffffull);
                if (live) pt = psel;
                proc = true;
                seg0 = 0;
                startsproc = 1u;
                endmask = 1u << (P - 1);              // all P columns are the pillar's; the ones past the cap's cutoff stay out of the maxes
            } else {
                // ---- packed pass: columns = arena positions base + s .. base + s + 31, whole pillars only ----------------------
                const int src = s + col;
                const int idx0 = __shfl(rec.x, src, 64), rk = __shfl(rec.y, src, 64), cn = __shfl(rec.z, src, 64);
                cell = __shfl(rec.w, src, 64);
                const unsigned sm = (unsigned)(startmask >> s);   // bit c: column c starts a segment; bit 0 is set
                seg0 = 31 - __clz((int)(sm & (0xffffffffu >> (31 - col))));
                b = g.batch > 1 ? fdiv(cell, g.div_cells) : 0;
                row = rk;
                if (g.batch > 1 && rk >= 0) row = b < kMaxB ? s_voff[b] + rk - s_fbase[b] : g.voxel_offsets[b] + rk - g.w.frame_base[b];
                // taken by this pass: whole (the pillar ends inside the pass), owned (it starts inside this wave's window), small.
                // The pass's first pillar always is, so `todo` shrinks with every pass.  Emitted: its output row exists — a
                // caller's capacity below the voxel count truncates, the pillars past it are taken and dropped
                const bool own = rk >= 0 && cn <= P && seg0 + cn <= 32 && s + seg0 <= kWin;
                proc = own && row < g.capacity;
                if (g.cap_mode == 1 && proc) {
                    if (g.batch == 1) cutoff = cut1;
                    else if (b < kMaxB) cutoff = s_cut[b];
                    else {
                        const int rc = g.w.frame_base[b] + g.max_voxels;
                        if (rc < g.w.frame_base[b + 1]) cutoff = g.w.vox_rec[rc].w;
                    }
                }
                if (!mine) {   // the other role's pass: only which pillars it takes
                    todo &= ~((__ballot(own && col == seg0) & 0xffffffffull) << s);
                    continue;
                }
                // Inside a pillar the arena holds the points in arrival order.  Nothing of the VFE depends on it (the sums above
                // are exact, the maxes commute); only the optional `voxels` output lists a pillar's points by ascending index.
                // K3 hands out a voxel's slots from the top, so a segment that arrived in index order is DESCENDING and only has
                // to be read backwards; otherwise a sorting network on (segment, index, source column) keys orders all segments
                // at once (segments keep their columns)
                int from = col;
                if (g.voxels_out) {
                    const int before = __shfl_up(idx0, 1, 32);
                    if (__ballot(proc && col > seg0 && before <= idx0) == 0ull) {
                        if (proc) from = 2 * seg0 + cn - 1 - col;
                    } else {
                        unsigned key = ((unsigned)seg0 << (g.idx_bits + 5)) | (((unsigned)idx0 & idxmask) << 5) | (unsigned)col;
                        key = bitonic32_asc(key, col);
                        from = (int)(key & 31u);
                    }
                }
                const int idx = __shfl(idx0, from, 32);
                from += s;
                pt.x = __shfl(apt.x, from, 64); pt.y = __shfl(apt.y, from, 64);
                pt.z = __shfl(apt.z, from, 64); pt.w = __shfl(apt.w, from, 64);
                live = proc && idx < cutoff;
                if (!live) pt = make_float4(0.f, 0.f, 0.f, 0.f);
                const unsigned livemask = (unsigned)(__ballot(live) & 0xffffffffull);
                const unsigned segmask = (cn >= 32 ? 0xffffffffu : ((1u << cn) - 1u)) << seg0;
                n = __popc(livemask & segmask);
                startsproc = (unsigned)(__ballot(proc && col == seg0) & 0xffffffffull);
                endmask = (unsigned)(__ballot(proc && col == seg0 + cn - 1) & 0xffffffffull);
                todo &= ~((__ballot(own && col == seg0) & 0xffffffffull) << s);
                if (uni(startsproc) == 0u) continue;   // every pillar of the pass lies past the capacity
            }
            const int cib = proc ? cell - b * cells_per_frame : 0;   // nz == 1: cell = (b * ny + y) * nx + x
            const int cy = fdiv(cib, g.div_nx);
            const int4 cd = make_int4(b, 0, cy, cib - cy * g.nx);
            if (proc && lane < 32 && col == seg0) {
                reinterpret_cast<int4 *>(g.coords_out)[row] = cd;
                g.num_out[row] = n;
            }
            if (g.voxels_out || v.pillar_mask) {   // optional padded outputs, one pillar at a time
                for (unsigned left = uni(startsproc); left != 0u; left &= left - 1u) {
                    const int c0 = __ffs((int)left) - 1;
                    const int pn = __builtin_amdgcn_readlane(n, c0), prow = __builtin_amdgcn_readlane(row, c0);
                    const int from = c0 + col < 32 ? c0 + col : 31;
                    float4 o;
                    o.x = __shfl(pt.x, from, 32); o.y = __shfl(pt.y, from, 32); o.z = __shfl(pt.z, from, 32); o.w = __shfl(pt.w, from, 32);
                    if (col >= pn) o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (lane < P) {
                        if (g.voxels_out) reinterpret_cast<float4 *>(g.voxels_out)[(size_t)prow * P + lane] = o;
                        if (v.pillar_mask) v.pillar_mask[(size_t)prow * P + lane] = lane < pn ? 1.f : 0.f;
                    }
                }
            }
#ifdef HVPR_EXP_TIMING
            t_passes++;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            tstamp[2] = __builtin_readcyclecounter();
            tacc[0] += tstamp[2] - t_front0;
#endif
            vfe_pass<true>(W, L, s_zt[wid], v, pt, live, live, proc, proc && n < P, n, seg0, cd, row, cell, endmask, startsproc, g.spatial,
                           g.spatial_channels, g.spatial_scale VFE_TPASS);
#ifdef HVPR_EXP_TIMING
            for (int q = 0; q < 7; ++q) tacc[1 + q] += tstamp[3 + q] - tstamp[2 + q];
            t_win = __builtin_readcyclecounter();
#endif
#ifdef HVPR_EXP_TIMING
            if (false)
                printf("vfe wg %d pass %d (pillars %d): prologue %lld | to first pass %lld | front %lld | decor %lld | l0+a %lld | xmax %lld | "
                       "pillar mfma %lld | combine %lld | final %lld cycles\n", (int)blockIdx.x - fill_blocks, t_passes, __popc(endmask),
                       tstamp[1] - tstamp[0], tstamp[2] - tstamp[1], tstamp[3] - tstamp[2], tstamp[4] - tstamp[3], tstamp[5] - tstamp[4],
                       tstamp[6] - tstamp[5], tstamp[7] - tstamp[6], tstamp[8] - tstamp[7], tstamp[9] - tstamp[8]);
#endif
        }
        rec = rec_next;
        apt = apt_next;
    }
#ifdef HVPR_EXP_TIMING
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if ((wblk % 37) == 0 && threadIdx.x == 0)
        printf("vfe-abs work blk %d role %d passes %d: %lld x10 ns, prologue %lld | sums over passes: between passes %lld front %lld | s23 %lld s34 %lld s45 %lld s56 %lld s67 %lld s78 %lld s89 %lld cycles\n", wblk, role, t_passes,
               (long long)__builtin_amdgcn_s_memrealtime() - rt0, tstamp[1] - tstamp[0], t_winacc, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5], tacc[6], tacc[7]);
#endif
}

// separate-call form: padded voxels in, one pillar per pass
__global__ void __launch_bounds__(256, 2) k_vfe_rows(const float4 *__restrict__ voxels, const int *__restrict__ num_points,
                                                     const int4 *__restrict__ coords, int M, int P, const int *__restrict__ m_device,
                                                     VfeParams v) {
    __shared__ __attribute__((aligned(16))) WaveLds s_wave[4];
    __shared__ __attribute__((aligned(16))) float s_stage[C1 * 36 + CS1 * 20 + C0 * CIN];
    __shared__ __attribute__((aligned(16))) float s_zt[4][160];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = lane & 31;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    if (m_device) M = min(M, *m_device);
    Wreg W;
    load_weights(v, s_stage, s_zt[wid], W, nullptr);
    WaveLds &L = s_wave[wid];
    for (int p = wave; p < M; p += n_waves) {
        int n = num_points[p];
        n = n < 0 ? 0 : (n > P ? P : n);
        const int4 cd = coords[p];
        float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool live = col < n;
        if (live) pt = voxels[(size_t)p * P + col];
        if (v.pillar_mask && lane < P) v.pillar_mask[(size_t)p * P + lane] = live ? 1.f : 0.f;
        const unsigned endmask = 1u << (n > 0 ? n - 1 : 0);
        // a pillar without points still has its padded slots: column 0 stands in for them
#ifdef HVPR_EXP_TIMING
        long long tstamp[10];
#endif
        vfe_pass<false>(W, L, s_zt[wid], v, pt, live, live || (n == 0 && col == 0), true, n < P, n, 0, cd, p, 0, endmask, 1u,
                        nullptr, 0, nullptr VFE_TPASS);
    }
}

}  // namespace

int hvpr_i_vfe_gather(const VoxelizeArgs &a, const VoxWs &w, const int32_t *voxel_offsets, int capacity, const VfeWeights &v,
                      float *voxels, int32_t *coords, int32_t *num_points, float *pillar_features, float *scale_features,
                      float *pillar_mask, float *spatial, int spatial_channels, float *spatial_scale, unsigned char *canvas_state,
                      hipStream_t s) {
    // the sort keys of a pass hold (segment, point index, source column) in 32 bits: 5 + idx_bits + 5
    if (a.n_feat != 4 || a.max_points > 32 || a.nz != 1 || a.n_points >= (1 << 22)) return HVPR_ERR_UNSUPPORTED;
    if (!spatial || !spatial_scale || spatial_channels != 2 * C1) return HVPR_ERR_INVALID_ARG;
    // one wave per window of kWin arena positions (one frame: 586 waves, each a single pass); big batches stride
    const int windows = hvpr_cdiv(a.n_points, kWin);
    int blocks = hvpr_cdiv(windows, 4);
    if (blocks > 512) blocks = 512;   // two workgroups per CU: beyond one round the waves walk their windows (one weight staging each)
    if (blocks < 1) blocks = 1;
    const int split = blocks <= 256 ? 1 : 0;   // one round with room to spare: first passes and later passes on different waves
    if (split) blocks *= 2;
    int idx_bits = 1;
    while (idx_bits < 22 && (1ll << idx_bits) < (long long)a.n_points) ++idx_bits;
    const long long n_cells = (long long)a.batch * a.nx * a.ny;
    const ClearJob cj{w.cell_first, w.cell_count, w.cell_vid, w.frame_base, voxel_offsets, a.batch, a.nx, a.ny, a.max_voxels, capacity, spatial,
                      spatial_scale, 0, n_cells, canvas_state};
    GatherSrc g{a.points, a.point_stride, a.xyz_col, a.batch, a.nx, a.ny, a.nz, a.max_voxels, a.cap_mode, capacity, w,
                voxel_offsets, voxels, coords, num_points, spatial, spatial_channels, spatial_scale, blocks, cj, idx_bits, a.n_points,
                make_fastdiv((unsigned)a.nx), make_fastdiv((unsigned)(a.nx * a.ny)), split};
    const VfeParams vp{v.vs_x, v.vs_y, v.vs_z, v.off_x, v.off_y, v.off_z, v.w0, v.b0, v.w1, v.b1, v.ws0, v.bs0, v.ws1, v.bs1,
                       pillar_features, scale_features, pillar_mask};
    // three 64-cell steps per wave: few, long-lived workgroups — they hold slots the pillar workgroups want
    long long fill = (n_cells + 767) / 768;
    if (fill > 1024) fill = 1024;
    hipLaunchKernelGGL(k_vfe_gather, dim3(blocks + (int)fill), dim3(256), 0, s, a.max_points, vp, g);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_pillar_vfe_fwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coords, int M,
                                       int P, const int32_t *m_device, float vs_x, float vs_y, float vs_z, float off_x,
                                       float off_y, float off_z, const float *w0, const float *b0, const float *w1,
                                       const float *b1, const float *ws0, const float *bs0, const float *ws1,
                                       const float *bs1, float *pillar_features, float *pillar_scale_features,
                                       float *pillar_mask, hvpr_stream_t stream) {
    if (M < 0 || P < 1) return HVPR_ERR_INVALID_ARG;
    if (P > 32) return HVPR_ERR_UNSUPPORTED;
    if (M == 0) return HVPR_OK;
    if (!voxels || !num_points || !coords || !w0 || !b0 || !w1 || !b1 || !ws0 || !bs0 || !ws1 || !bs1 ||
        !pillar_features || !pillar_scale_features)
        return HVPR_ERR_INVALID_ARG;
    int blocks = hvpr_cdiv(M, 4);
    if (blocks > 2048) blocks = 2048;
    const VfeParams vp{vs_x, vs_y, vs_z, off_x, off_y, off_z, w0, b0, w1, b1, ws0, bs0, ws1, bs1, pillar_features, pillar_scale_features,
                       pillar_mask};
    hipLaunchKernelGGL(k_vfe_rows, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)voxels, num_points,
                       (const int4 *)coords, M, P, m_device, vp);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
