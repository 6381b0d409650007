// a5 / a11 — the stride-1 3x3 convolutions of the BEV backbone by Winograd F(2x2, 3x3) on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32).  Same call sites as conv_igemm.hip (BaseBEVBackbone_Scale.forward,
// pcdet/models/backbones_2d/base_bev_backbone.py:228-315; behind them cuDNN, which picks Winograd for fp32 3x3 itself): every
// 2x2 block of outputs costs 16 multiplies per (cin, cout) pair instead of 36, all of them in fp32 —
//     Y = At [ (G g Gt) . (Bt d B) ] A      Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//                                           At = [1 1 1 0; 0 1 -1 -1]
// (d = 4x4 input patch of the block, g = 3x3 filter; Lavin & Gray's minimal filtering form).  The direct kernel is bounded by the
// matrix pipes at the clock the power budget leaves; this one issues 2.25 x fewer MFMAs for the same layer.
//
// GEMM view: 16 independent products  M_xi[block, co] = sum_ci V_xi[block, ci] * U_xi[ci, co],  xi = (a, b) in 4 x 4.
// Workgroup = NG groups of 4 waves; a group owns 32 blocks (4 x 8 blocks = 8 x 16 output pixels) x 64 output channels and
// wave `a` of the group owns the four products xi = (a, 0..3): 4 x 2 accumulator blocks of 32 x 32 = 128 registers.  K walks
// in chunks of 8 input channels; per chunk the (halo) input patch and the 16 transformed filter slabs U stream into one of two
// LDS stages by LDS-DMA (the groups share U: NG = 2 halves the dominant L2 -> LDS stream).  The INPUT TRANSFORM never touches
// LDS: row a of Bt has two non-zeros, so the wave reads two patch rows x four columns (eight ds_read_b128, lane = block,
// half-wave = channels 4h..4h+3, the operand trick of conv_igemm.hip), forms t = d[r0] +- d[r1] and V[a][b] from t — 32 vector
// ALU operations per 32 MFMAs — directly in the registers the MFMAs read.  The patch is laid out [row][column parity][column/2]
// with a 20-slot row pitch: the 16 lanes of a ds_read_b128 service group (8 blocks of two block rows) hit 16 distinct slots.
// OUTPUT TRANSFORM: R_a[q] = sum_b M[a][b] At[q][b] in registers, the four waves exchange R through LDS (the U stages, idle by
// then) and wave (p, q) finishes output pixel (2 by + p, 2 bx + q) of its lane's block: bias / ReLU / SFM step / float4 store as
// in the direct kernel.  Persistent, XCD-aware tile walk as in conv_igemm.hip.
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KC = 8;             // input channels per K chunk
constexpr int BN = 64;            // output channels per workgroup
constexpr int TW = 16, PW = TW + 2;
constexpr int ROWP = 20;          // LDS slots (float4) per patch row: [parity 2][10]
constexpr int W_V4 = 16 * 2 * BN; // float4 per U stage: [xi 16][half 2][co 64]

// see conv_igemm.hip: LDS-DMA issued from inline asm so that hipcc does not serialise it against the ds_reads
__device__ __forceinline__ void lds_dma16(const void *sbase_uniform, unsigned voff_bytes, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(sbase_uniform), "s"(lds_dst_uniform)
                 : "memory");
}

__device__ __forceinline__ unsigned lds_addr_of(const void *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}

struct WinoArgs {
    const float *in;      // [N, H, W, Cin]
    const float *wpk;     // [cout_pad/64][Cin/8][xi 16][half 2][co 64][4]   (hvpr_conv2d_wino_pack_f32)
    const float *bias;    // [cout_pad]
    float *out;           // [N, H, W, out_cstride]
    const float *gate;    // [N, H, W] or null
    const float *resid;   // [N, H, W, resid_cstride] or null
    int N, H, W, Cin;
    int cout, cout_pad;
    int out_cstride, out_coff, resid_cstride;
    int relu;
    int tiles_x, tiles_y, n_ct;
    float *stats;         // optional [pixel tiles][2][cout]: per-tile sum / sum of squares of the raw output (train-mode BatchNorm)
};

__device__ __forceinline__ float4 f4_fma(float s, float4 a, float4 b) {      // s * a + b, s = +-1: an exact add / subtract
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

template <int NG, bool STATS, int NB>
__global__ void __launch_bounds__(256 * NG) __attribute__((amdgpu_waves_per_eu(2, NB == 2 ? 2 : 3))) k_wino(WinoArgs a) {
    constexpr int NT = 256 * NG;
    constexpr int TH = 8 * NG, PH = TH + 2;
    constexpr int PPAD = (PH * ROWP + 63) / 64 * 64;          // float4 per channel-half plane
    constexpr int PATCH_V4 = 2 * PPAD;
    // NB = 32-channel blocks per workgroup: 2 (64 output channels) or 1 (32: twice the tiles with half the work each, for levels
    // whose tile count does not fill the chip — the filter stage is then the half of the packed 64-channel image it needs)
    constexpr int BNT = 32 * NB, W_V4T = 16 * 2 * BNT;
    constexpr int NLD_P = (PATCH_V4 + NT - 1) / NT, NLD_W = W_V4T / NT;
    constexpr int PATCH_PAD = NLD_P * NT;
    static_assert(W_V4T % NT == 0 && (NB == 2 || NG == 1), "whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4 *const s_w = smem;                                  // [2][W_V4T]; the output transform's exchange area afterwards
    float4 *const s_patch = smem + 2 * W_V4T;                  // [2][PATCH_PAD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned wave_s = __builtin_amdgcn_readfirstlane((unsigned)(threadIdx.x >> 6));
    const int grp = wid >> 2, wa = wid & 3;                    // pixel group, row of the 4 x 4 product grid
    const int half = lane >> 5, l31 = lane & 31;
    const int by = l31 >> 3, bx = l31 & 7;                     // this lane's 2 x 2 output block inside the group's 4 x 8

    // ---- tile-independent staging descriptors ----
    int p_py[NLD_P], p_px[NLD_P], p_part[NLD_P];
    bool p_live[NLD_P];
#pragma unroll
    for (int i = 0; i < NLD_P; ++i) {
        const int v = tid + i * NT;
        const int plane = v / PPAD, slot = v % PPAD;
        const int y = slot / ROWP, rem = slot % ROWP;
        const int x = 2 * (rem % 10) + rem / 10;
        p_py[i] = y; p_px[i] = x; p_part[i] = plane * 4;
        p_live[i] = plane < 2 && y < PH && x < PW;
    }
    // ---- per-lane LDS read offsets (float4 units) ----
    // rows of the block's 4 x 4 patch that row `wa` of Bt combines:  a=0: d0 - d2   a=1: d1 + d2   a=2: d2 - d1   a=3: d1 - d3
    const int r0 = wa == 0 ? 0 : (wa == 2 ? 2 : 1), r1 = wa == 2 ? 1 : (wa == 3 ? 3 : 2);
    const float sgn = wa == 1 ? 1.f : -1.f;
    const int pbase = half * PPAD + (2 * (4 * grp + by)) * ROWP + bx;
    const int p_off0 = pbase + r0 * ROWP, p_off1 = pbase + r1 * ROWP;
    const int u_off = (wa * 4 * 2 + half) * BNT + l31;
    const unsigned lds_w0 = lds_addr_of(s_w) + wave_s * 1024u;
    const unsigned lds_patch0 = lds_addr_of(s_patch) + wave_s * 1024u;

    const int n_pt = a.tiles_x * a.tiles_y * a.N;
    const int total_walk = ((n_pt + 7) / 8) * 8 * a.n_ct;
    const int n_chunks = a.Cin / KC;
    for (int it = blockIdx.x; it < total_walk; it += gridDim.x) {
    const int xcd = it & 7, j = it >> 3;
    const int ct = j % a.n_ct;
    int pt = (j / a.n_ct) * 8 + xcd;
    if (pt >= n_pt) continue;
    const int tx = pt % a.tiles_x; pt /= a.tiles_x;
    const int ty = pt % a.tiles_y;
    const int n = pt / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int co0 = ct * BNT;

    unsigned poff[NLD_P];
    bool pok[NLD_P];
#pragma unroll
    for (int i = 0; i < NLD_P; ++i) {
        const int iy = iy0 + p_py[i], ix = ix0 + p_px[i];
        pok[i] = p_live[i] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        poff[i] = pok[i] ? (unsigned)(((iy * a.W + ix) * a.Cin + p_part[i]) * (int)sizeof(float)) : 0u;
    }
    const char *in_n = (const char *)(a.in + (size_t)n * a.H * a.W * a.Cin);                 // uniform
    // packed image: [64-channel tile][chunk][xi 16][half 2][co 64] float4; an NB = 1 workgroup reads the 32-column half it owns
    const char *w_ct = (const char *)(a.wpk + (size_t)(NB == 2 ? ct : ct >> 1) * n_chunks * (W_V4 * 4));           // uniform
    const unsigned w_half = NB == 2 ? 0u : (unsigned)(ct & 1) * 32u * 16u;

    // one staging piece of (chunk, buf): pieces 0 .. NLD_P-1 = input patch, NLD_P .. NLD_P+NLD_W-1 = filter slab
    auto stage_piece = [&](int chunk, int buf, int i) {
        if (i < NLD_P) {
            const char *pbase_g = in_n + (size_t)chunk * (KC * sizeof(float));
            if (pok[i]) lds_dma16(pbase_g, poff[i], lds_patch0 + (unsigned)(buf * PATCH_PAD + i * NT) * 16u);
        } else {
            const int j = i - NLD_P;
            const char *wbase = w_ct + (size_t)chunk * (W_V4 * 16);
            lds_dma16(wbase, NB == 2 ? (unsigned)((tid + j * NT) * 16) : (unsigned)((((tid + j * NT) >> 5) * 64 + ((tid + j * NT) & 31)) * 16) + w_half,
                      lds_w0 + (unsigned)(buf * W_V4T + j * NT) * 16u);
        }
    };
    auto stage = [&](int chunk, int buf) {
#pragma unroll
        for (int i = 0; i < NLD_P + NLD_W; ++i) stage_piece(chunk, buf, i);
    };
    const bool border = iy0 < 0 || ix0 < 0 || iy0 + PH > a.H || ix0 + PW > a.W;
    if (border) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < NLD_P; ++i) s_patch[b * PATCH_PAD + tid + i * NT] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    // This tile's bias row (BNT floats) is parked in the patch area's unused tail (slots PH * ROWP .. of plane 0, stage 0: never a DMA
    // destination, never a K-loop operand) by sixteen lanes now; the epilogue then takes its eight float4 from LDS instead of issuing
    // eight global loads per lane between its stores — a vector-memory instruction costs the issuing wave ≈145 cycles on this part.
    static_assert(PH * ROWP + BNT / 4 <= PPAD, "room for the bias row behind the patch");
    if (tid < BNT / 4) s_patch[PH * ROWP + tid] = *(const float4 *)(a.bias + co0 + tid * 4);
    f32x16 acc[4][NB];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][nb][r] = 0.f;

    auto chunk_step = [&](int c, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float4 *sp = s_patch + BUF * PATCH_PAD;
        const float4 *sw = s_w + BUF * W_V4T;
        float4 d0[4], d1[4], u[4][NB];
        // this chunk's first operands
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            d0[jx] = sp[p_off0 + (jx & 1) * 10 + (jx >> 1)];
            d1[jx] = sp[p_off1 + (jx & 1) * 10 + (jx >> 1)];
        }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) u[b][nb] = sw[u_off + b * 2 * BNT + nb * 32];
        __builtin_amdgcn_sched_barrier(0);
        // input transform: t_j = d[r0][j] +- d[r1][j];  V[a][.] = (t0 - t2, t1 + t2, t2 - t1, t1 - t3)
        float4 t[4], v[4];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) t[jx] = f4_fma(sgn, d1[jx], d0[jx]);
        v[0] = f4_sub(t[0], t[2]); v[1] = f4_add(t[1], t[2]); v[2] = f4_sub(t[2], t[1]); v[3] = f4_sub(t[1], t[3]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b + 2 < 4) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) u[b + 2][nb] = sw[u_off + (b + 2) * 2 * BNT + nb * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[b][nb].x, v[b].x, acc[b][nb], 0, 0, 0);
                acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[b][nb].y, v[b].y, acc[b][nb], 0, 0, 0);
                acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[b][nb].z, v[b].z, acc[b][nb], 0, 0, 0);
                acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[b][nb].w, v[b].w, acc[b][nb], 0, 0, 0);
                // The next stage's DMA pieces go out a few at a time after the first DEAL of the chunk's 4 NB groups of four MFMAs, in
                // the shadow of the MFMAs just queued.  Issued in one go ahead of the transform (the round-4 form) the ~10 pieces are
                // several hundred cycles in which this wave feeds the matrix cores nothing; after the last groups they would land too
                // late for the next chunk.  Frame pipeline, frames/s: in front 424 | after the b groups: first two 435, first three
                // 439, all four 428 | after the groups of four: first four 440.5, first five 444.7, first six 444.0 (NOTES_r05 section 11).
                constexpr int DEAL = 4 * NB > 5 ? 5 : 4 * NB - 1, NPF = (NLD_P + NLD_W + DEAL - 1) / DEAL;
                static_assert(DEAL * NPF >= NLD_P + NLD_W, "every piece goes out");
                if (c + 1 < n_chunks && b * NB + nb < DEAL) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = (b * NB + nb) * NPF; i < (b * NB + nb + 1) * NPF && i < NLD_P + NLD_W; ++i) stage_piece(c + 1, BUF ^ 1, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    stage(0, 0);
    for (int c = 0; c + 1 < n_chunks; c += 2) {
        chunk_step(c, std::integral_constant<int, 0>{});
        chunk_step(c + 1, std::integral_constant<int, 1>{});
    }
    if (n_chunks & 1) chunk_step(n_chunks - 1, std::integral_constant<int, 0>{});

    // ---- output transform + epilogue.  A lane holds block l31 and 16 channels per accumulator block (rows (r&3) + 8*(r>>2) +
    // 4*half: four runs of four).  Exchange area (float4): [grp][a][q][channel half of the pass][run][lane]. ----
    const int p = wa >> 1, q = wa & 1;
    const int oy = oy0 + 2 * (4 * grp + by) + p, ox = ox0 + 2 * bx + q;
    const bool live_px = oy < a.H && ox < a.W;
    const size_t pix = ((size_t)n * a.H + (live_px ? oy : 0)) * a.W + (live_px ? ox : 0);
    const float gate = a.gate && live_px ? a.gate[pix] : 0.f;
    const float *rrow = a.gate ? a.resid + pix * a.resid_cstride : nullptr;
    const float sg = p ? -1.f : 1.f;
    // one pass over both 32-channel halves when the exchange fits the two filter stages (NG = 1: 64 KB), else one half per pass
    constexpr int NPASS = NG == 1 ? 1 : 2, NBP = NB / NPASS;
    float4 ykeep[2][4];                 // this lane's outputs (statistics pass below); zero where the pixel / channel is not live
#pragma unroll
    for (int i = 0; i < 8; ++i) ykeep[i >> 2][i & 3] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        __syncthreads();               // the U stages (first pass) / the previous pass's exchange reads are done
#pragma unroll
        for (int nl = 0; nl < NBP; ++nl) {
            const int nb = ps * NBP + nl;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 m[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) m[b] = make_float4(acc[b][nb][4 * g], acc[b][nb][4 * g + 1], acc[b][nb][4 * g + 2], acc[b][nb][4 * g + 3]);
                s_w[((((grp * 4 + wa) * 2 + 0) * NBP + nl) * 4 + g) * 64 + lane] = f4_add(f4_add(m[0], m[1]), m[2]);
                s_w[((((grp * 4 + wa) * 2 + 1) * NBP + nl) * 4 + g) * 64 + lane] = f4_sub(f4_sub(m[1], m[2]), m[3]);
            }
        }
        __syncthreads();
        if (live_px) {
#pragma unroll
            for (int nl = 0; nl < NBP; ++nl) {
                const int nb = ps * NBP + nl;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = co0 + nb * 32 + 8 * g + 4 * half;
                    if (col >= a.cout) continue;
                    const float4 x0 = s_w[((((grp * 4 + p) * 2 + q) * NBP + nl) * 4 + g) * 64 + lane];
                    const float4 x1 = s_w[((((grp * 4 + p + 1) * 2 + q) * NBP + nl) * 4 + g) * 64 + lane];
                    const float4 x2 = s_w[((((grp * 4 + p + 2) * 2 + q) * NBP + nl) * 4 + g) * 64 + lane];
                    const float4 bias = s_patch[PH * ROWP + (col - co0) / 4];
                    float4 y = f4_add(f4_fma(sg, x2, f4_fma(sg, x1, x0)), bias);       // p = 0: x0 + x1 + x2;  p = 1: x1 - x2 - x3
                    if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
                    if (a.gate) {
                        const float4 r = *(const float4 *)(rrow + col);
                        y.x = fmaf(gate, y.x, r.x); y.y = fmaf(gate, y.y, r.y); y.z = fmaf(gate, y.z, r.z); y.w = fmaf(gate, y.w, r.w);
                    }
                    *(float4 *)(a.out + pix * a.out_cstride + a.out_coff + col) = y;
                    if (STATS) ykeep[nl][g] = y;
                }
            }
        }
    }
    static_assert(!STATS || (NG == 1 && NB == 2), "the statistics pass is written for the 8 x 16 px x 64 channel tile");
    if (STATS) {
        // Batch statistics of the layer's BatchNorm, fused: per-channel sum and sum of squares over this tile's live pixels go
        // to row (pixel tile) of a.stats (finished in double by hvpr_bn_finalize_partials_f32) — the separate pass over the
        // written tensor (601 MB per level-0 layer at batch 16) disappears.  Through LDS: lanes park their 8 float4, thread
        // (channel, wave) adds the 32 blocks of that wave's pixel position, then the four positions.
        __syncthreads();               // the exchange reads are done
#pragma unroll
        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
            for (int g = 0; g < 4; ++g) s_w[((wid * 2 + nl) * 4 + g) * 64 + lane] = ykeep[nl][g];
        __syncthreads();
        {
            const int c = tid & 63, part = tid >> 6;
            const float *base = (const float *)(s_w + ((part * 2 + (c >> 5)) * 4 + ((c >> 3) & 3)) * 64 + ((c >> 2) & 1) * 32) + (c & 3);
            float sm = 0.f, sq = 0.f;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) {
                const float v = base[((i + c) & 31) * 4];       // rotated start: spreads the lanes over the banks
                sm += v; sq = fmaf(v, v, sq);
            }
            float *s_part = (float *)(s_w + 2048);               // [4 waves][2][64]
            s_part[(part * 2 + 0) * 64 + c] = sm;
            s_part[(part * 2 + 1) * 64 + c] = sq;
        }
        __syncthreads();
        if (tid < 128) {
            const float *s_part = (const float *)(s_w + 2048);
            const int c = tid & 63, k = tid >> 6;
            const float v = (s_part[(0 * 2 + k) * 64 + c] + s_part[(1 * 2 + k) * 64 + c]) + (s_part[(2 * 2 + k) * 64 + c] + s_part[(3 * 2 + k) * 64 + c]);
            const int pt_lin = (n * a.tiles_y + ty) * a.tiles_x + tx;
            if (co0 + c < a.cout) a.stats[((size_t)pt_lin * 2 + k) * a.cout + co0 + c] = v;
        }
    }
    __syncthreads();   // the next tile refills the LDS stages
    }
}

// weights (Cout, Cin, 3, 3) [adjoint: the data-gradient filter w'[o][i][u][v] = w[i][o][2-u][2-v] of a (Cin', Cout') = (Cout, Cin)
// layer] -> U = G g Gt per (o, i), scaled per output channel, in the stage image the kernel streams.  One thread per (o, i).
__global__ void k_wino_pack(const float *__restrict__ w, const float *__restrict__ scale, int cout, int cin, int cout_pad, int adjoint,
                            float *__restrict__ out) {
    const long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (id >= (long long)cout_pad * cin) return;
    const int o = (int)(id / cin), i = (int)(id % cin);
    double g[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            double x = 0.0;
            if (o < cout) x = adjoint ? (double)w[(((size_t)i * cout + o) * 3 + (2 - u)) * 3 + (2 - v)] : (double)w[(((size_t)o * cin + i) * 3 + u) * 3 + v];
            g[u][v] = (scale && o < cout) ? x * (double)scale[o] : x;
        }
    double gg[4][3];       // G g
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        gg[0][v] = g[0][v];
        gg[1][v] = 0.5 * (g[0][v] + g[1][v] + g[2][v]);
        gg[2][v] = 0.5 * (g[0][v] - g[1][v] + g[2][v]);
        gg[3][v] = g[2][v];
    }
    const int ct = o / BN, co_l = o % BN, chunk = i / KC, hf = (i % KC) / 4, e = i % 4;
    float *dst = out + ((size_t)ct * (cin / KC) + chunk) * (W_V4 * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double U[4] = {gg[r][0], 0.5 * (gg[r][0] + gg[r][1] + gg[r][2]), 0.5 * (gg[r][0] - gg[r][1] + gg[r][2]), gg[r][2]};
#pragma unroll
        for (int b = 0; b < 4; ++b) dst[(((size_t)(r * 4 + b) * 2 + hf) * BN + co_l) * 4 + e] = (float)U[b];
    }
}

template <int NG, bool STATS, int NB>
int launch(WinoArgs a, hipStream_t s) {
    constexpr int NT = 256 * NG, TH = 8 * NG, PH = TH + 2;
    constexpr int PPAD = (PH * ROWP + 63) / 64 * 64;
    constexpr int NLD_P = (2 * PPAD + NT - 1) / NT;
    constexpr int lds = (2 * (16 * 2 * 32 * NB) + 2 * NLD_P * NT) * 16;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    a.n_ct = a.cout_pad / (32 * NB);
    static unsigned long long lds_set = 0ull;
    if (hvpr_ensure_dyn_lds((const void *)k_wino<NG, STATS, NB>, lds, &lds_set) != 0) return -1;
    const long long tiles = (long long)a.N * a.tiles_x * a.tiles_y * a.n_ct;
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wino<NG, STATS, NB>, NT, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        resident = per_cu * cus;
    }
    long long blocks = tiles < resident ? (tiles + 7) / 8 * 8 : resident;
    if (blocks > resident && resident >= 8) blocks = resident / 8 * 8;
    hipLaunchKernelGGL((k_wino<NG, STATS, NB>), dim3((unsigned)blocks), dim3(NT), lds, s, a);
    return 0;
}

}  // namespace

extern "C" int hvpr_conv2d_wino_stats_rows(int N, int H, int W) {     // rows of bn_partials: the 8 x 16 pixel tiles
    if (N < 1 || H < 1 || W < 1) return 0;
    return N * ((H + 7) / 8) * ((W + TW - 1) / TW);
}

extern "C" size_t hvpr_conv2d_wino_packed_floats(int cin, int cout) {
    if (cin < 1 || cout < 1) return 0;
    return (size_t)((cout + BN - 1) / BN * BN) * (size_t)cin * 16;
}

extern "C" int hvpr_conv2d_wino_pack_f32(const float *weight, const float *scale, int cout, int cin, int adjoint, float *packed,
                                         hvpr_stream_t stream) {
    if (!weight || !packed || cout < 1 || cin < 1) return HVPR_ERR_INVALID_ARG;
    if (cin % KC != 0) return HVPR_ERR_UNSUPPORTED;
    const int cout_pad = (cout + BN - 1) / BN * BN;
    const long long n = (long long)cout_pad * cin;
    hipLaunchKernelGGL(k_wino_pack, dim3(hvpr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, weight, scale, cout, cin, cout_pad,
                       adjoint ? 1 : 0, packed);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_conv2d_wino_nhwc_f32(const float *in, int N, int H, int W, int Cin, const float *w_packed, const float *bias,
                                         int cout, int relu, const float *gate, const float *resid, int resid_cstride, float *out,
                                         int out_cstride, int out_coff, int px_groups, float *bn_partials, hvpr_stream_t stream) {
    if (!in || !w_packed || !bias || !out || N < 1 || H < 1 || W < 1 || Cin < 8 || cout < 1) return HVPR_ERR_INVALID_ARG;
    if ((gate == nullptr) != (resid == nullptr)) return HVPR_ERR_INVALID_ARG;
    if (Cin % KC != 0) return HVPR_ERR_UNSUPPORTED;
    if (cout % 4 != 0 || out_cstride % 4 != 0 || out_coff % 4 != 0 || (resid && resid_cstride % 4 != 0)) return HVPR_ERR_UNSUPPORTED;
    if ((long long)H * W * Cin * 4 >= (1ll << 31)) return HVPR_ERR_UNSUPPORTED;      // byte offsets inside one image are formed in signed 32-bit arithmetic
    WinoArgs a;
    a.in = in; a.wpk = w_packed; a.bias = bias; a.out = out; a.gate = gate; a.resid = resid;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin;
    a.cout = cout; a.cout_pad = (cout + BN - 1) / BN * BN;
    a.out_cstride = out_cstride; a.out_coff = out_coff; a.resid_cstride = resid_cstride;
    a.relu = relu;
    a.stats = bn_partials;
    if (bn_partials && (px_groups != 1 || relu || gate)) return HVPR_ERR_UNSUPPORTED;      // statistics of the RAW output, 8 x 16 tiles
    int rc;
    if (px_groups == 1) rc = bn_partials ? launch<1, true, 2>(a, (hipStream_t)stream) : launch<1, false, 2>(a, (hipStream_t)stream);
    else if (px_groups == 2) rc = launch<2, false, 2>(a, (hipStream_t)stream);
    else if (px_groups == 4) rc = launch<1, false, 1>(a, (hipStream_t)stream);      // 8 x 16 px x 32 channels: twice the tiles
    else return HVPR_ERR_INVALID_ARG;
    if (rc != 0) return HVPR_ERR_LAUNCH;
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
