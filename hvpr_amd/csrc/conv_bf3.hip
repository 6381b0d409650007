// a5 (optional precision mode) — the BEV backbone convolutions on the bf16 matrix cores with 3-term split operands:
//   x = x_hi + x_lo (two bf16, round-to-nearest),  w = w_hi + w_lo   ->   x*w ~= x_hi*w_hi + x_hi*w_lo + x_lo*w_hi
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (dropped term ~2^-16 relative: "bf16x3").  SURVEY.md §8d allows a
// reduced-precision path "evidenced within tolerance" (north_star: feature tensors within 1e-3 relative of fp32); the exact
// fp32 kernel (conv_igemm.hip) stays the default and the parity reference.  Same implicit-GEMM structure as k_conv:
// NHWC, LDS-DMA double buffer, persistent XCD-aware tile walk, weights as the MFMA "A" operand.
//
// NPL = number of bf16 planes per value: 2 = "bf16x3" above; 3 = "bf16x6": x = x_hi + x_mid + x_lo EXACTLY (3 x 8 mantissa
// bits), six products hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi — the dropped terms are <= 2^-23 relative, the accuracy of
// the fp32 matrix-core kernel itself (fp32 emulation on the bf16 matrix cores).
//
// Split-bf16 NHWC: every group of 8 channels of a pixel is stored as [8 x bf16 hi | 8 x bf16 lo] = 32 bytes — the same
// footprint and addressing as 8 fp32 channels, so staging moves the same 16-byte pieces.  One MFMA consumes K = 16 channels:
// lanes 0-31 supply the 8 channels of the even chunk, lanes 32-63 those of the odd chunk.
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int KC = 8;

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
__device__ __forceinline__ bf16x8 as_bf(float4 v) {
    union { float4 f; bf16x8 b; } u;
    u.f = v;
    return u.b;
}
// x -> NPL bf16 planes by successive round-to-nearest residuals (p0 = rne(x), p1 = rne(x - p0), p2 = rne(x - p0 - p1));
// returned as the 16 bits of each
template <int NPL>
__device__ __forceinline__ void split1(float x, unsigned (&pl)[NPL]) {
    float r = x;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const __bf16 h = (__bf16)r;
        pl[k] = (unsigned)__builtin_bit_cast(unsigned short, h);
        r -= (float)h;
    }
}

struct Conv3Args {
    const float4 *in;     // split-bf16 NHWC: [N, H, W, Cin/8][hi 16 B | lo 16 B]
    const float4 *wpk;    // [TAPS, Cin/8, 2 (hi|lo), CoutPad] x 16 B (8 x bf16 over the chunk's input channels)
    const float *bias;    // [CoutPad]
    void *out;            // fp32 NHWC [N, OH, OW, out_cstride] or split-bf16 NHWC with out_cstride channels
    const float *gate;    // [N, OH, OW] or null
    const float4 *resid;  // split-bf16 NHWC, resid_cstride channels, or null
    int N, H, W, Cin, OH, OW, cout_gemm, cout_pad, out_cstride, out_coff, resid_cstride, relu, out_split;
    int up, cout_real;    // 1x1 kernel as ConvTranspose2d(kernel = stride = up): gemm column = (ky*up + kx) * cout_real + co
    int tiles_x, tiles_y, n_ct;
};

// WGM x WGN = the waves of a workgroup along pixels x channels: 4 waves as 2 x 2 (or 4 x 1 for the 256-pixel tile), or 8 waves
// as 4 x 2.  Eight waves = two per SIMD: an LDS-DMA piece blocks the issuing wave for ~100-180 cycles, and with one wave per SIMD
// (98+ KB of LDS stages = one workgroup per CU) nothing else can issue MFMAs meanwhile — the second wave does.
// ROWS: the weight slab is staged one kernel row (3 taps) at a time instead of all 9 taps per K=16 step: stages of 20-30 KB
// instead of 45-73 KB, so that 2-3 workgroups fit a CU and hide each other's LDS-DMA issue stalls and barrier waits (the regime
// the fp32 kernel lives in); the patch of a K step is staged with its first row and kept for the three rows.
// ONE: 1x1 kernel (the ConvTranspose k = s layers): no halo, and a stage holds FOUR K=16 steps ("virtual taps" at the same pixel).
template <int TH, int TW, int BN, int S, int WGM, int NPL, int NWAVES = 4, bool ROWS = false, bool ONE = false>
__global__ void __launch_bounds__(64 * NWAVES) k_conv3(Conv3Args a) {
    static_assert(!ONE || (!ROWS && S == 1), "1x1: stride 1, whole stages");
    constexpr int HALO = ONE ? 0 : 2, NSUB = 2;
    constexpr int STAPS = ONE ? 4 : (ROWS ? 3 : 9);    // taps (1x1: K=16 steps) per weight stage
    constexpr int KSTEPS = ONE ? 4 : 1;                // K=16 steps per stage
    constexpr int KSTAGE = KC * NSUB * KSTEPS;         // input channels per stage
    constexpr int NT = 64 * NWAVES;
    constexpr int WGN = NWAVES / WGM;
    constexpr int BM = TH * TW;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MB = WM / 32, NB = WN / 32;
    static_assert(TH % WGM == 0 && (TH / WGM) * TW == WM, "a wave owns whole tile rows");
    static_assert(MB >= 1 && NB >= 1, "wave tile must hold at least one 32x32 block");
    constexpr int PH = (TH - 1) * S + 1 + HALO, PW = (TW - 1) * S + 1 + HALO;
    // Row pitch of the LDS patch in 16-byte slots.  ds_read_b128 is serviced in four 16-lane groups ({0-3,12-15,20-27},
    // {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md §LDS) and a group is conflict-free only if its 16 lanes hit 16 distinct
    // slots of the 256-byte bank row.  A 32-lane half covers 2 rows of 16 pixels (pitch 32) or 4 rows of 8 pixels (pitch 24):
    // with these pitches every group sees all 16 slots, for every tap offset; the natural pitch PW (18 / 10) is 2-3 way
    // conflicted (SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE before).
    constexpr int PITCH = (S != 1 || NPL == 3 || ROWS || ONE) ? PW : (TW == 16 ? 32 : (TW == 8 ? 24 : PW));   // pad only where LDS allows
    static_assert(PITCH >= PW, "pitch must hold a patch row");
    constexpr int PPAD = (PH * PITCH + 63) / 64 * 64;
    constexpr int PATCH_V4 = KSTEPS * NSUB * NPL * PPAD;   // patch   [K step][sub][plane][PPAD]
    constexpr int W_V4 = STAPS * NSUB * NPL * BN;      // weights [tap of the stage][sub][plane][BN]
    constexpr int NLD_P = (PATCH_V4 + NT - 1) / NT, NLD_W = (W_V4 + NT - 1) / NT;
    constexpr int PATCH_PAD = NLD_P * NT, W_PAD = NLD_W * NT;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4 *s_patch = smem;                            // [2][PATCH_PAD]
    float4 *s_w = smem + 2 * PATCH_PAD;                // [2][W_PAD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned wave_s = __builtin_amdgcn_readfirstlane((unsigned)(threadIdx.x >> 6));
    const int wm = wid / WGN, wn = wid % WGN;
    const int half = lane >> 5, l31 = lane & 31;       // half = which 8-channel chunk of the K=16 step this lane feeds

    int p_py[NLD_P], p_px[NLD_P], p_part[NLD_P];
    bool p_live[NLD_P];
#pragma unroll
    for (int i = 0; i < NLD_P; ++i) {
        const int v = tid + i * NT;
        const int part = v / PPAD, pix = v % PPAD;     // part = sub * NPL + plane: 16-byte piece `part` of the stage's bytes
        p_py[i] = pix / PITCH; p_px[i] = pix % PITCH; p_part[i] = part;
        p_live[i] = v < PATCH_V4 && pix < PH * PITCH && pix % PITCH < PW;
    }
    unsigned woff0[NLD_W];
    bool wok[NLD_W];
    const size_t w_chunk_stride = (size_t)a.cout_pad * NPL;            // float4 units between cin chunks
    const size_t w_tap_stride = (size_t)(a.Cin / KC) * w_chunk_stride;
#pragma unroll
    for (int i = 0; i < NLD_W; ++i) {
        const int v = tid + i * NT;                                    // = ((tap*NSUB + sub)*NPL + plane)*BN + co_local
        wok[i] = v < W_V4;
        const int co_l = v % BN, r = v / BN;
        const int hl = r % NPL, sub = (r / NPL) % NSUB, tap = (r / NPL) / NSUB;
        const size_t tap_stride = ONE ? NSUB * w_chunk_stride : w_tap_stride;      // 1x1: the next K=16 step
        woff0[i] = wok[i] ? (unsigned)(((size_t)tap * tap_stride + (size_t)sub * w_chunk_stride + (size_t)hl * a.cout_pad + co_l) * 16) : 0u;
    }
    int a_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int q = mb * 32 + l31;
        const int py = wm * (TH / WGM) + q / TW, px = q % TW;
        a_off[mb] = (half * NPL) * PPAD + (py * S) * PITCH + px * S;   // plane 0 of this lane's chunk; plane k = + k * PPAD
    }
    int b_off[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b_off[nb] = (half * NPL) * BN + wn * WN + nb * 32 + l31;   // plane k = + k*BN; next tap = + NSUB*NPL*BN
    const unsigned lds_patch0 = lds_addr_of(s_patch) + wave_s * 1024u;
    const unsigned lds_w0 = lds_addr_of(s_w) + wave_s * 1024u;

    const int n_pt = a.tiles_x * a.tiles_y * a.N;
    const int total_walk = ((n_pt + 7) / 8) * 8 * a.n_ct;
    const int n_chunks = a.Cin / KSTAGE;
    for (int it = blockIdx.x; it < total_walk; it += gridDim.x) {
    const int xcd = it & 7, j = it >> 3;
    const int ct = j % a.n_ct;
    int pt = (j / a.n_ct) * 8 + xcd;
    if (pt >= n_pt) continue;
    const int tx = pt % a.tiles_x; pt /= a.tiles_x;
    const int ty = pt % a.tiles_y;
    const int n = pt / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - HALO / 2, ix0 = ox0 * S - HALO / 2;
    const int co0 = ct * BN;

    unsigned poff[NLD_P];
    bool pok[NLD_P];
#pragma unroll
    for (int i = 0; i < NLD_P; ++i) {
        const int iy = iy0 + p_py[i], ix = ix0 + p_px[i];
        pok[i] = p_live[i] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        poff[i] = pok[i] ? (unsigned)(((iy * a.W + ix) * (a.Cin / 8 * NPL) + p_part[i]) * 16) : 0u;   // Cin/8 groups x NPL pieces
    }
    const unsigned w_co0 = (unsigned)(co0 * 16);
    const char *in_n = (const char *)(a.in + (size_t)n * a.H * a.W * (a.Cin / 8 * NPL));

    // one stage: the weights of kernel row `ky` (all rows when !ROWS) of K step `chunk` into weight slot wbuf and, if asked,
    // the patch of that K step into patch slot pbuf
    auto stage = [&](int chunk, int ky, int wbuf, int pbuf, bool with_patch) {
        const char *pbase = in_n + (size_t)chunk * (KSTEPS * NSUB * NPL * 16);
        const char *wbase = (const char *)a.wpk + ((size_t)chunk * KSTEPS * NSUB * w_chunk_stride + (size_t)ky * 3 * w_tap_stride) * 16;
        if (with_patch) {
#pragma unroll
            for (int i = 0; i < NLD_P; ++i)
                if (pok[i]) lds_dma16(pbase, poff[i], lds_patch0 + (unsigned)(pbuf * PATCH_PAD + i * NT) * 16u);
        }
#pragma unroll
        for (int i = 0; i < NLD_W; ++i)
            if (wok[i]) lds_dma16(wbase, woff0[i] + w_co0, lds_w0 + (unsigned)(wbuf * W_PAD + i * NT) * 16u);
    };
    const bool border = iy0 < 0 || ix0 < 0 || iy0 + PH > a.H || ix0 + PW > a.W;
    if (border) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < NLD_P; ++i) s_patch[b * PATCH_PAD + tid + i * NT] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    // one stage of multiplies: wait for it, let the next stage stream in, multiply.  Slots are compile-time constants (the loops
    // below are unrolled over them) so that every LDS address is an immediate offset.
    auto step = [&](int c, int ky, auto wbuf_tag, auto pbuf_tag) {
        constexpr int WBUF = decltype(wbuf_tag)::value, PBUF = decltype(pbuf_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ROWS) {
            if (ky < 2) stage(c, ky + 1, WBUF ^ 1, PBUF, false);
            else if (c + 1 < n_chunks) stage(c + 1, 0, WBUF ^ 1, PBUF ^ 1, true);
        } else if (c + 1 < n_chunks) {
            stage(c + 1, 0, WBUF ^ 1, PBUF ^ 1, true);
        }
        const float4 *sp = s_patch + PBUF * PATCH_PAD;
        const float4 *sw = s_w + WBUF * W_PAD;
        constexpr int PF = 2;
        float4 av[PF + 1][MB][NPL], bv[PF + 1][NB][NPL];
        auto lds_load = [&](int t, int slot) {            // t = tap inside the stage
            const int row = ROWS ? ky : t / 3, kx = ROWS ? t : t % 3;
            const int shift = ONE ? t * NSUB * NPL * PPAD : row * PITCH + kx;     // 1x1: the planes of K step t, same pixel
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int k = 0; k < NPL; ++k) av[slot][mb][k] = sp[a_off[mb] + shift + k * PPAD];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int k = 0; k < NPL; ++k) bv[slot][nb][k] = sw[b_off[nb] + t * NSUB * NPL * BN + k * BN];
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) lds_load(t, t);
#pragma unroll
        for (int tap = 0; tap < STAPS; ++tap) {
            if (tap + PF < STAPS) lds_load(tap + PF, (tap + PF) % (PF + 1));
            __builtin_amdgcn_sched_barrier(0);
            const int cur = tap % (PF + 1);
            // plane pairs (weight plane, activation plane), smallest terms first; every pass walks all blocks so that
            // consecutive MFMAs write different accumulators
            constexpr int NPAIR = NPL == 2 ? 3 : 6;
            constexpr int PW_[6] = {NPL == 2 ? 1 : 2, NPL == 2 ? 0 : 0, NPL == 2 ? 0 : 1, 1, 0, 0};
            constexpr int PA_[6] = {NPL == 2 ? 0 : 0, NPL == 2 ? 1 : 2, NPL == 2 ? 0 : 1, 0, 1, 0};
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(bv[cur][nb][PW_[pr]]), as_bf(av[cur][mb][PA_[pr]]),
                                                                              acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    stage(0, 0, 0, 0, true);
    if (ROWS) {     // six stages per pair of K steps: weight slots 0,1,0,1,0,1, patch slots 0,0,0,1,1,1
        for (int c = 0; c + 1 < n_chunks; c += 2) {
            step(c, 0, I0{}, I0{}); step(c, 1, I1{}, I0{}); step(c, 2, I0{}, I0{});
            step(c + 1, 0, I1{}, I1{}); step(c + 1, 1, I0{}, I1{}); step(c + 1, 2, I1{}, I1{});
        }
        if (n_chunks & 1) { step(n_chunks - 1, 0, I0{}, I0{}); step(n_chunks - 1, 1, I1{}, I0{}); step(n_chunks - 1, 2, I0{}, I0{}); }
    } else {
        for (int c = 0; c + 1 < n_chunks; c += 2) {
            step(c, 0, I0{}, I0{});
            step(c + 1, 0, I1{}, I1{});
        }
        if (n_chunks & 1) step(n_chunks - 1, 0, I0{}, I0{});
    }

    // epilogue: a lane holds one pixel x 16 channels in four runs of four consecutive channels (8g + 4*half + 0..3)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int q = mb * 32 + l31;
        const int oy = oy0 + wm * (TH / WGM) + q / TW, ox = ox0 + q % TW;
        if (oy >= a.OH || ox >= a.OW) continue;
        const size_t pix = ((size_t)n * a.OH + oy) * a.OW + ox;
        const float gate = a.gate ? a.gate[pix] : 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = co0 + wn * WN + nb * 32 + 8 * g + 4 * half;
                if (col >= a.cout_gemm) continue;
                const float4 bias = *(const float4 *)(a.bias + col);
                float y[4] = {acc[mb][nb][4 * g] + bias.x, acc[mb][nb][4 * g + 1] + bias.y, acc[mb][nb][4 * g + 2] + bias.z,
                              acc[mb][nb][4 * g + 3] + bias.w};
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
                }
                if (a.gate) {       // residual in split form: group (col/8), 4 channels at position 4*half of every plane
                    const uint2 *rp = (const uint2 *)(a.resid + (pix * (a.resid_cstride / 8) + col / 8) * NPL);
                    float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = NPL - 1; k >= 0; --k) {            // smallest plane first
                        const uint2 r = rp[2 * k + half];
                        rs[0] += __uint_as_float(r.x << 16); rs[1] += __uint_as_float(r.x & 0xffff0000u);
                        rs[2] += __uint_as_float(r.y << 16); rs[3] += __uint_as_float(r.y & 0xffff0000u);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = fmaf(gate, y[e], rs[e]);
                }
                if (a.out_split) {
                    unsigned pl[4][NPL];
#pragma unroll
                    for (int e = 0; e < 4; ++e) split1<NPL>(y[e], pl[e]);
                    const int oc = a.out_coff + col;
                    uint2 *op = (uint2 *)((float4 *)a.out + (pix * (a.out_cstride / 8) + oc / 8) * NPL);
#pragma unroll
                    for (int k = 0; k < NPL; ++k) op[2 * k + half] = make_uint2(pl[0][k] | (pl[1][k] << 16), pl[2][k] | (pl[3][k] << 16));
                } else if (a.up > 1) {      // ConvTranspose2d(kernel = stride = up): column -> (sub-pixel, channel)
                    const int sub = col / a.cout_real, co = col - sub * a.cout_real;
                    const int sy = sub / a.up, sx = sub % a.up;
                    const size_t opix = ((size_t)n * a.OH * a.up + (oy * a.up + sy)) * ((size_t)a.OW * a.up) + (ox * a.up + sx);
                    *(float4 *)((float *)a.out + opix * a.out_cstride + a.out_coff + co) = make_float4(y[0], y[1], y[2], y[3]);
                } else {
                    *(float4 *)((float *)a.out + pix * a.out_cstride + a.out_coff + col) = make_float4(y[0], y[1], y[2], y[3]);
                }
            }
        }
    }
    __syncthreads();
    }
}

// fp32 NHWC -> split-bf16 NHWC (one thread per 8-channel group)
template <int NPL>
__global__ void __launch_bounds__(256) k_split(const float4 *__restrict__ src, long long groups, uint4 *__restrict__ dst) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const float4 a = src[g * 2], b = src[g * 2 + 1];
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned pl[8][NPL];
#pragma unroll
    for (int e = 0; e < 8; ++e) split1<NPL>(x[e], pl[e]);
#pragma unroll
    for (int k = 0; k < NPL; ++k)
        dst[g * NPL + k] = make_uint4(pl[0][k] | (pl[1][k] << 16), pl[2][k] | (pl[3][k] << 16), pl[4][k] | (pl[5][k] << 16),
                                      pl[6][k] | (pl[7][k] << 16));
}

template <int TH, int TW, int BN, int S, int WGM, int NPL, int NWAVES = 4, bool ROWS = false, bool ONE = false>
int launch3(Conv3Args a, hipStream_t s) {
    a.tiles_x = (a.OW + TW - 1) / TW;
    a.tiles_y = (a.OH + TH - 1) / TH;
    a.n_ct = a.cout_pad / BN;
    constexpr int PH = (TH - 1) * S + (ONE ? 1 : 3), PW = (TW - 1) * S + (ONE ? 1 : 3);
    constexpr int PITCH = (S != 1 || NPL == 3 || ROWS || ONE) ? PW : (TW == 16 ? 32 : (TW == 8 ? 24 : PW));
    constexpr int PPAD = (PH * PITCH + 63) / 64 * 64;
    constexpr int NT = 64 * NWAVES;
    constexpr int NLD_P = ((ONE ? 4 : 1) * 2 * NPL * PPAD + NT - 1) / NT, NLD_W = ((ONE ? 4 : (ROWS ? 3 : 9)) * 2 * NPL * BN + NT - 1) / NT;
    const size_t lds = (size_t)2 * (NLD_P + NLD_W) * NT * 16;
    const long long tiles = (long long)a.N * a.tiles_x * a.tiles_y * a.n_ct;
    static int resident = 0;
    if (resident == 0) {
        if (hipFuncSetAttribute((const void *)k_conv3<TH, TW, BN, S, WGM, NPL, NWAVES, ROWS, ONE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HVPR_ERR_LAUNCH;
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_conv3<TH, TW, BN, S, WGM, NPL, NWAVES, ROWS, ONE>, NT, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        resident = per_cu * cus;
    }
    long long blocks = tiles < resident ? (tiles + 7) / 8 * 8 : resident;
    if (blocks > resident && resident >= 8) blocks = resident / 8 * 8;
    hipLaunchKernelGGL((k_conv3<TH, TW, BN, S, WGM, NPL, NWAVES, ROWS, ONE>), dim3((unsigned)blocks), dim3(NT), lds, s, a);
    return HVPR_OK;
}

// split-bf16 NHWC -> fp32 NHWC: sum of the planes, smallest first
template <int NPL>
__global__ void __launch_bounds__(256) k_unsplit(const uint4 *__restrict__ src, long long groups, float4 *__restrict__ dst) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = NPL - 1; k >= 0; --k) {
        const uint4 v = src[g * NPL + k];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[2 * e] += __uint_as_float(w[e] << 16);
            x[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
        }
    }
    dst[g * 2] = make_float4(x[0], x[1], x[2], x[3]);
    dst[g * 2 + 1] = make_float4(x[4], x[5], x[6], x[7]);
}

template <int NPL>
int dispatch_conv3(Conv3Args a, int stride, int tile_cfg, hipStream_t s) {
    if (a.cout_pad % 64) return HVPR_ERR_INVALID_ARG;
    if (NPL == 3 && stride != 1 && tile_cfg != 4) return HVPR_ERR_UNSUPPORTED;   // three-plane stride-2 stages fit only row-staged
    if (tile_cfg == 0)              // 128 px x 64 ch
        return stride == 1 ? launch3<8, 16, 64, 1, 2, NPL>(a, s) : launch3<8, 16, 64, 2, 2, NPL == 3 ? 2 : NPL>(a, s);
    if (tile_cfg == 1)              // 64 px x 64 ch
        return stride == 1 ? launch3<8, 8, 64, 1, 2, NPL>(a, s) : launch3<8, 8, 64, 2, 2, NPL == 3 ? 2 : NPL>(a, s);
    if (tile_cfg == 2) {            // 256 px x 64 ch, waves 4 x 1 (each 64 px x 64 ch): stride 1, two planes only
        if (stride != 1 || NPL == 3) return HVPR_ERR_UNSUPPORTED;
        return launch3<16, 16, 64, 1, 4, 2>(a, s);
    }
    if (tile_cfg == 3) {            // 128 px x 64 ch, EIGHT waves 4 x 2 (each 32 px x 32 ch): stride 1
        if (stride != 1) return HVPR_ERR_UNSUPPORTED;
        return launch3<8, 16, 64, 1, 4, NPL, 8>(a, s);
    }
    if (tile_cfg == 4)              // 64 px x 64 ch, weights staged per kernel row (2-3 workgroups per CU)
        return stride == 1 ? launch3<8, 8, 64, 1, 2, NPL, 4, true>(a, s) : launch3<8, 8, 64, 2, 2, NPL, 4, true>(a, s);
    if (tile_cfg == 5) {            // 128 px x 64 ch, per kernel row
        if (stride != 1) return HVPR_ERR_UNSUPPORTED;
        return launch3<8, 16, 64, 1, 2, NPL, 4, true>(a, s);
    }
    return HVPR_ERR_INVALID_ARG;
}

}  // namespace

extern "C" int hvpr_split_bf16_f32(const float *src, long long n_floats, int n_planes, void *dst, hvpr_stream_t stream) {
    if (!src || !dst || n_floats < 0 || n_floats % 8 != 0 || (n_planes != 2 && n_planes != 3)) return HVPR_ERR_INVALID_ARG;
    if (n_floats == 0) return HVPR_OK;
    const long long groups = n_floats / 8;
    const dim3 grid((unsigned)((groups + 255) / 256));
    if (n_planes == 2) hipLaunchKernelGGL(k_split<2>, grid, dim3(256), 0, (hipStream_t)stream, (const float4 *)src, groups, (uint4 *)dst);
    else hipLaunchKernelGGL(k_split<3>, grid, dim3(256), 0, (hipStream_t)stream, (const float4 *)src, groups, (uint4 *)dst);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_unsplit_bf16_f32(const void *src, long long n_floats, int n_planes, float *dst, hvpr_stream_t stream) {
    if (!src || !dst || n_floats < 0 || n_floats % 8 != 0 || (n_planes != 2 && n_planes != 3)) return HVPR_ERR_INVALID_ARG;
    if (n_floats == 0) return HVPR_OK;
    const long long groups = n_floats / 8;
    const dim3 grid((unsigned)((groups + 255) / 256));
    if (n_planes == 2) hipLaunchKernelGGL(k_unsplit<2>, grid, dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, groups, (float4 *)dst);
    else hipLaunchKernelGGL(k_unsplit<3>, grid, dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, groups, (float4 *)dst);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_conv2d_nhwc_bf16x3(const void *in_split, int N, int H, int W, int Cin, const void *w_split,
                                       const float *bias, int stride, int cout, int cout_pad, int relu, const float *gate,
                                       const void *resid_split, int resid_cstride, void *out, int out_split,
                                       int out_cstride, int out_coff, int tile_cfg, int n_planes, hvpr_stream_t stream) {
    if (!in_split || !w_split || !bias || !out || N < 1 || H < 1 || W < 1 || cout < 1) return HVPR_ERR_INVALID_ARG;
    if ((gate == nullptr) != (resid_split == nullptr) || (n_planes != 2 && n_planes != 3)) return HVPR_ERR_INVALID_ARG;
    if (Cin % 16 != 0 || (stride != 1 && stride != 2)) return HVPR_ERR_UNSUPPORTED;
    if (cout % 4 != 0 || out_cstride % 8 != 0 || out_coff % 8 != 0 || (resid_split && resid_cstride % 8 != 0)) return HVPR_ERR_UNSUPPORTED;
    Conv3Args a;
    a.in = (const float4 *)in_split; a.wpk = (const float4 *)w_split; a.bias = bias; a.out = out; a.gate = gate;
    a.resid = (const float4 *)resid_split;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin;
    a.OH = (H + 2 - 3) / stride + 1; a.OW = (W + 2 - 3) / stride + 1;
    a.cout_gemm = cout; a.cout_pad = cout_pad; a.out_cstride = out_cstride; a.out_coff = out_coff; a.resid_cstride = resid_cstride;
    a.relu = relu; a.out_split = out_split; a.up = 1; a.cout_real = cout;
    const int st = n_planes == 2 ? dispatch_conv3<2>(a, stride, tile_cfg, (hipStream_t)stream)
                                 : dispatch_conv3<3>(a, stride, tile_cfg, (hipStream_t)stream);
    if (st != HVPR_OK) return st;
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_deconv_nhwc_bf16x3(const void *in_split, int N, int H, int W, int Cin, const void *w_split, const float *bias,
                                       int cout, int cout_pad, int up, int relu, float *out, int out_cstride, int out_coff,
                                       int n_planes, hvpr_stream_t stream) {
    if (!in_split || !w_split || !bias || !out || N < 1 || H < 1 || W < 1 || cout < 1 || up < 1) return HVPR_ERR_INVALID_ARG;
    if (n_planes != 2 && n_planes != 3) return HVPR_ERR_INVALID_ARG;
    if (Cin % 64 != 0 || cout % 4 != 0 || out_cstride % 4 != 0 || out_coff % 4 != 0 || cout_pad % 64 != 0 || cout_pad < cout * up * up)
        return HVPR_ERR_UNSUPPORTED;
    Conv3Args a;
    a.in = (const float4 *)in_split; a.wpk = (const float4 *)w_split; a.bias = bias; a.out = out; a.gate = nullptr; a.resid = nullptr;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = H; a.OW = W;
    a.cout_gemm = cout * up * up; a.cout_pad = cout_pad; a.out_cstride = out_cstride; a.out_coff = out_coff; a.resid_cstride = 0;
    a.relu = relu; a.out_split = 0; a.up = up; a.cout_real = cout;
    const int st = n_planes == 2 ? launch3<8, 8, 64, 1, 2, 2, 4, false, true>(a, (hipStream_t)stream)
                                 : launch3<8, 8, 64, 1, 2, 3, 4, false, true>(a, (hipStream_t)stream);
    if (st != HVPR_OK) return st;
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
