// a5 — BEV backbone convolutions as implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact
// fp32, 157 TFLOP/s peak on gfx950 — there is no xf32/TF32 path on this chip).
// Replaces the cuDNN calls behind BaseBEVBackbone_Scale.forward (eval),
// pcdet/models/backbones_2d/base_bev_backbone.py:280-315:
//   ZeroPad2d(1)+Conv3x3(stride 1|2)+BN+ReLU, Conv3x3(pad 1)+BN+ReLU, the weight-shared SFM step
//   x_att = gate * ReLU(BN(conv(x_att))) + x_att  (:291-295, gate from spatial_attention.py:57-63), and
//   ConvTranspose2d(k = s)+BN+ReLU (:177-188) written straight into its slice of the 384-channel concat (:303-304).
//
// Layout: activations are NHWC (torch channels_last), so the K dimension (tap, cin) is contiguous per pixel.
// GEMM view: M = output pixels, N = output channels, K = taps * Cin.  BatchNorm is folded by the caller into
// the packed weights (scale) and a per-channel bias.
//
// Workgroup = 4 waves (2 x 2), tile = (TH x TW) pixels x BN channels; each wave owns MB x NB blocks of 32 x 32.
// K is walked in chunks of 8 input channels: the (halo) input patch and all taps of the weight slab for the
// chunk are streamed into one of two LDS stages by LDS-DMA while the other stage is multiplied; one barrier per chunk.  Operand trick: lane half h reads channels 4h..4h+3 of the chunk
// as ONE ds_read_b128 for A and for B and feeds them to four consecutive MFMAs — the MFMA k index is a free
// permutation as long as A and B agree.
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KC = 8;   // input channels per K chunk

// One LDS-DMA piece: 16 B per lane, global (per-lane address) -> LDS (M0 = wave-uniform base, + lane * 16).
// Issued through inline asm on purpose: hipcc would otherwise put s_waitcnt vmcnt(0) in front of the next ds_read (it
// must assume the DMA write aliases it), which serialises the stream behind the multiply.  The wait is placed by hand
// in front of the per-chunk barrier instead (cdna_hip_programming.md §5.7 recipe).
// Address form: SGPR base + 32-bit VGPR byte offset.  Everything that changes from chunk to chunk lives in the SGPR base
// and in M0, so the steady-state loop issues NO vector-ALU instruction: measured on this chip, v_mfma_f32_32x32x2_f32
// runs on the f32 vector lanes (it is rated at the vector FMA rate), and a co-resident wave that needs VALU slots for
// address arithmetic is starved for as long as its neighbour multiplies.
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

struct ConvArgs {
    const float *in;      // [N, H, W, Cin]
    const float *wpk;     // [TAPS, Cin/8, 2, CoutPad, 4]   (BN scale folded; channel halves split)
    const float *bias;    // [CoutPad]  (gemm column index)
    float *out;           // [N, OH*up, OW*up, out_cstride]
    const float *gate;    // [N, OH, OW] or null
    const float *resid;   // [N, OH, OW, resid_cstride] or null
    int N, H, W, Cin;
    int OH, OW;           // conv output size (before pixel shuffle)
    int cout_gemm;        // live gemm columns
    int cout_pad;         // packed columns (multiple of BN)
    int out_cstride, out_coff, resid_cstride;
    int relu;
    int up;               // 1, or the ConvTranspose stride s (kernel == stride): column = (ky*s+kx)*cout_real + co
    int cout_real;        // channels per sub-pixel when up > 1 (== cout_gemm when up == 1)
    int tiles_x, tiles_y, n_ct;
};

#ifdef HVPR_EXP_TIMING
// kernel experiments: cycle stamps of the phases of every chunk of the SECOND tile of a few workgroups (all four waves), read
// back by hvpr_exp_conv_dbg: [block slot 0..7][wave][chunk 0..63][4] = wait start, barrier exit, DMA issued, multiply done
__device__ long long g_conv_dbg[8 * 4 * 64 * 4];
#endif

// UP2 (TAPS == 9, S == 1): the DATA GRADIENT of a stride-2 3x3 (pad 1) convolution as a gather over its output gradient, without the
// zero-upsampled tensor.  in = dz [N, H, W, Cin] at half resolution, out = dx [N, OH, OW, cout] at full resolution (OH, OW given by the
// caller: H = (OH + 2 - 3) / 2 + 1), weights = the layer's filter with (cout, cin) swapped, tap t = ky * 3 + kx NOT flipped:
//     dx[2a + py, 2b + px] = sum over the taps with ky = 1 (py == 0) | ky in {0, 2} (py == 1), kx likewise, of
//                            W[ky][kx]^T dz[a + (ky == 0), b + (kx == 0)]
// — 1 / 2 / 2 / 4 taps for the four parity classes instead of the 9 a stride-1 convolution over a zero-upsampled dz multiplies
// (conv_train.py ran that one on the Winograd kernel: 4.7 ms per layer at batch 16 for 173 GFLOP of useful work).  A 16 x 16 output
// tile = 8 x 8 positions (a, b) x 4 classes; MFMA block mb of a wave IS class (mb >> 1, mb & 1) for the wave's 4 x 8 positions, so which
// taps a block multiplies is known at compile time and every wave does the same 9 (tap, class) products per chunk.
template <int TH, int TW, int BN, int S, int TAPS, bool UP2 = false>
__global__ void __launch_bounds__(256) k_conv(ConvArgs a) {
    static_assert(!UP2 || (TAPS == 9 && S == 1 && TH == 16 && TW == 16), "UP2: 16 x 16 output tiles of a 3x3 stride-2 layer's data gradient");
    constexpr int BM = TH * TW;
    constexpr int WM = BM / 2, WN = BN / 2;        // per-wave tile
    constexpr int MB = WM / 32, NB = WN / 32;
    static_assert(MB >= 1 && NB >= 1, "wave tile must hold at least one 32x32 block");
    // TAPS: 9 = 3x3 conv; 1 = 1x1 conv, one 8-channel chunk per stage; 4 = 1x1 conv with FOUR 8-channel chunks per stage
    // ("virtual taps" at the same pixel: 4x the multiply work per barrier — a 1x1 stage is otherwise 4 MFMAs per wave)
    constexpr int HALO = TAPS == 9 ? 2 : 0;
    constexpr int NSUB = TAPS == 4 ? 4 : 1;        // 8-channel sub-chunks per stage
    constexpr int KSTAGE = KC * NSUB;              // input channels per stage
    constexpr int PH = UP2 ? TH / 2 + 1 : (TH - 1) * S + 1 + HALO, PW = UP2 ? TW / 2 + 1 : (TW - 1) * S + 1 + HALO;
    // LDS images (float4 units), both split into two channel-half planes so that a half-wave (32 lanes, same half)
    // reads 32 consecutive float4 = conflict-free ds_read_b128:
    //   patch   [half][PPAD]       pixel-major inside a plane
    //   weights [tap][half][BN]
    constexpr int PPAD = (PH * PW + 63) / 64 * 64;
    constexpr int PATCH_V4 = NSUB * 2 * PPAD;     //   patch   [sub][half][PPAD]
    constexpr int W_V4 = TAPS * BN * 2;
    constexpr int NLD_P = (PATCH_V4 + 255) / 256, NLD_W = (W_V4 + 255) / 256;

    // two LDS stages per operand, filled by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass);
    // padded to whole 1-KiB pieces because a piece always lands wave-uniform base + lane * 16
    constexpr int PATCH_PAD = NLD_P * 256, W_PAD = NLD_W * 256;
    __shared__ __attribute__((aligned(16))) float4 s_patch[2][PATCH_PAD];
    __shared__ __attribute__((aligned(16))) float4 s_w[2][W_PAD];
    __shared__ __attribute__((aligned(16))) float4 s_bias[BN / 4];     // this tile's bias row: the epilogue reads it from LDS (conv_wino.hip)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned wave_s = __builtin_amdgcn_readfirstlane((unsigned)(threadIdx.x >> 6));
    const int wm = wid >> 1, wn = wid & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // ---- tile-independent staging descriptors.  Everything a tile needs per DMA piece is two adds, four compares and
    // two multiply-adds: vector-ALU instructions of a workgroup that is between tiles issue only in the gaps its
    // neighbours' MFMAs leave (one per 64 cycles), so the per-tile scalar work is what costs. ----
    int p_py[NLD_P], p_px[NLD_P], p_part[NLD_P];
    bool p_live[NLD_P];
#pragma unroll
    for (int i = 0; i < NLD_P; ++i) {
        const int v = tid + i * 256;
        const int part = v / PPAD, pix = v % PPAD;          // part = sub * 2 + half: channels part*4 .. part*4+3 of the stage
        p_py[i] = pix / PW; p_px[i] = pix % PW; p_part[i] = part * 4;
        p_live[i] = v < PATCH_V4 && pix < PH * PW;
    }
    unsigned woff0[NLD_W];   // byte offset inside the packed weights without the cout-tile and chunk terms
    bool wok[NLD_W];
    const size_t w_chunk_stride = (size_t)a.cout_pad * KC;                 // floats between cin chunks
    const size_t w_tap_stride = (size_t)(a.Cin / KC) * w_chunk_stride;     // floats between taps
#pragma unroll
    for (int i = 0; i < NLD_W; ++i) {
        const int v = tid + i * 256;                                       // = (tap*2 + half)*BN + co_local
        wok[i] = v < W_V4;
        const int th = v / BN, co_l = v % BN;
        const int tap = th >> 1, hf = th & 1;
        const size_t tap_stride = TAPS == 4 ? w_chunk_stride : w_tap_stride;   // a virtual tap is the next cin chunk
        woff0[i] = wok[i] ? (unsigned)(((size_t)tap * tap_stride + ((size_t)hf * a.cout_pad + co_l) * 4) * sizeof(float)) : 0u;
    }
    // ---- per-lane LDS read offsets (float4 units) ----
    int a_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int q = mb * 32 + l31;
        const int py = wm * (TH / 2) + q / TW, px = q % TW;
        a_off[mb] = half * PPAD + (py * S) * PW + px * S;
        if (UP2) a_off[mb] = half * PPAD + (wm * 4 + (l31 >> 3)) * PW + (l31 & 7);      // position (a, b) of the tile; the class is mb
    }
    int b_off[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b_off[nb] = half * BN + wn * WN + nb * 32 + l31;
    const unsigned lds_patch0 = lds_addr_of(&s_patch[0][0]) + wave_s * 1024u;    // uniform (wave_s is an SGPR value)
    const unsigned lds_w0 = lds_addr_of(&s_w[0][0]) + wave_s * 1024u;

    // Persistent workgroups: the grid is (resident workgroups per CU) x 256 and workgroup w walks tiles w, w+G, w+2G ...
    // so that every CU ends up with the same number of tiles (+-1) whatever the dispatcher does after the first wave.
    // XCD-aware order: consecutive workgroup ids land on consecutive XCDs (8 of them, each with its own L2), so the walk
    // index is split as (xcd = it % 8, j = it / 8) and the n_ct cout tiles of ONE pixel tile are given to consecutive j of
    // the SAME xcd: they run at the same time on the same L2 and the input patch is fetched from HBM once, not n_ct times.
    const int n_pt = a.tiles_x * a.tiles_y * a.N;
    const int total_walk = ((n_pt + 7) / 8) * 8 * a.n_ct;
#ifdef HVPR_EXP_TIMING
    int dbg_tile = 0;
    const int dbg_slot = (blockIdx.x % 97 == 5 && blockIdx.x / 97 < 8) ? (int)(blockIdx.x / 97) : -1;
#endif
    for (int it = blockIdx.x; it < total_walk; it += gridDim.x) {
    const int xcd = it & 7, j = it >> 3;
    const int ct = j % a.n_ct;
    int pt = (j / a.n_ct) * 8 + xcd;
    if (pt >= n_pt) continue;
    const int tx = pt % a.tiles_x; pt /= a.tiles_x;
    const int ty = pt % a.tiles_y;
    const int n = pt / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = UP2 ? oy0 / 2 : oy0 * S - (HALO / 2), ix0 = UP2 ? ox0 / 2 : ox0 * S - (HALO / 2);
    const int co0 = ct * BN;

    unsigned poff[NLD_P];    // byte offset of this lane's 16 B inside image n (without the chunk term)
    bool pok[NLD_P];
#pragma unroll
    for (int i = 0; i < NLD_P; ++i) {
        const int iy = iy0 + p_py[i], ix = ix0 + p_px[i];
        pok[i] = p_live[i] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        poff[i] = pok[i] ? (unsigned)(((iy * a.W + ix) * a.Cin + p_part[i]) * (int)sizeof(float)) : 0u;
    }
    const unsigned w_co0 = (unsigned)(co0 * 4 * sizeof(float));
    const char *in_n = (const char *)(a.in + (size_t)n * a.H * a.W * a.Cin);   // uniform

    // one staging piece of (chunk, buf): pieces 0 .. NLD_P-1 = input patch, NLD_P .. NLD_P+NLD_W-1 = weight slab
    auto stage_piece = [&](int chunk, int buf, int i) {
        if (i < NLD_P) {
            const char *pbase = in_n + (size_t)chunk * (KSTAGE * sizeof(float));
            if (pok[i]) lds_dma16(pbase, poff[i], lds_patch0 + (unsigned)(buf * PATCH_PAD + i * 256) * 16u);
        } else {
            const int j = i - NLD_P;
            const char *wbase = (const char *)a.wpk + (size_t)chunk * NSUB * w_chunk_stride * sizeof(float);
            if (wok[j]) lds_dma16(wbase, woff0[j] + w_co0, lds_w0 + (unsigned)(buf * W_PAD + j * 256) * 16u);
        }
    };
    auto stage = [&](int chunk, int buf) {
#pragma unroll
        for (int i = 0; i < NLD_P + NLD_W; ++i) stage_piece(chunk, buf, i);
    };
    // taps of a chunk after whose MFMAs the next stage's DMA pieces go out (conv_wino.hip: issued in one go in front of the chunk
    // they are several hundred cycles in which this wave feeds the matrix cores nothing); 0 = in front (one-tap chunks are too short)
    constexpr int DEAL = TAPS == 9 ? 6 : (TAPS == 4 ? 3 : 0);
    // a DMA piece outside the image is skipped and must read as zero: clear both stages first when this tile's patch
    // sticks out of the image (uniform decision; interior tiles overwrite every slot they read)
    const bool border = iy0 < 0 || ix0 < 0 || iy0 + PH > a.H || ix0 + PW > a.W;
    if (border) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < NLD_P; ++i) s_patch[b][tid + i * 256] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    if (tid < BN / 4) s_bias[tid] = *(const float4 *)(a.bias + co0 + tid * 4);   // (visible by the chunk barriers; the previous tile's
    f32x16 acc[MB][NB];                                                           //  readers are behind its closing barrier)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    const int n_chunks = a.Cin / KSTAGE;
    // one chunk: wait for it, let the next one stream into the other stage, multiply.  `buf` is a compile-time constant in
    // both call sites (the loop is unrolled by two) so that every LDS address is an immediate offset.
    auto chunk_step = [&](int c, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        // chunk c has landed: this wave's DMA is waited for by hand, the barrier covers the other waves' and also
        // says that every wave is done reading the other stage
#ifdef HVPR_EXP_TIMING
        const long long q0 = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef HVPR_EXP_TIMING
        const long long q1 = __builtin_readcyclecounter();
#endif
        if (DEAL == 0 && c + 1 < n_chunks) stage(c + 1, BUF ^ 1);
#ifdef HVPR_EXP_TIMING
        const long long q2 = __builtin_readcyclecounter();
#endif
        const float4 *sp = s_patch[BUF];
        const float4 *sw = s_w[BUF];
        // operand ring: LDS reads run PF taps ahead of the MFMAs that consume them (a dependent-accumulator
        // MFMA chain cannot hide a ds_read issued right before its wait)
        constexpr int PF = TAPS >= 3 ? 2 : 0;
        float4 av[PF + 1][MB], bv[PF + 1][NB];
        // UP2: class (mb >> 1, mb & 1) multiplies tap (ky, kx) iff ky is 1 for an even row / 0 or 2 for an odd one (kx likewise), with
        // the gradient pixel one row down / one column right for ky == 0 / kx == 0
        auto up2_uses = [](int mb, int tap) { return (((mb >> 1) == 0) == (tap / 3 == 1)) && (((mb & 1) == 0) == (tap % 3 == 1)); };
        auto lds_load = [&](int tap, int slot) {
            const int ky = TAPS == 9 ? tap / 3 : 0, kx = TAPS == 9 ? tap % 3 : 0;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                if (UP2) {
                    if (up2_uses(mb, tap)) av[slot][mb] = sp[a_off[mb] + (ky == 0 ? PW : 0) + (kx == 0 ? 1 : 0)];
                    continue;
                }
                av[slot][mb] = sp[a_off[mb] + (TAPS == 4 ? tap * 2 * PPAD : ky * PW + kx)];
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[slot][nb] = sw[b_off[nb] + tap * BN * 2];
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) lds_load(t, t);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            if (tap + PF < TAPS || PF == 0) lds_load(tap + PF < TAPS ? tap + PF : tap, (tap + PF) % (PF + 1));
            __builtin_amdgcn_sched_barrier(0);
            const int cur = tap % (PF + 1);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    if (UP2 && !up2_uses(mb, tap)) continue;
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[cur][nb].x, av[cur][mb].x, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[cur][nb].y, av[cur][mb].y, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[cur][nb].z, av[cur][mb].z, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[cur][nb].w, av[cur][mb].w, acc[mb][nb], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (DEAL > 0 && tap < DEAL && c + 1 < n_chunks) {
                constexpr int NPC = (NLD_P + NLD_W + DEAL - 1) / (DEAL > 0 ? DEAL : 1);
#pragma unroll
                for (int i = tap * NPC; i < (tap + 1) * NPC && i < NLD_P + NLD_W; ++i) stage_piece(c + 1, BUF ^ 1, i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef HVPR_EXP_TIMING
        if (dbg_slot >= 0 && dbg_tile == 1 && lane == 0 && c < 64) {
            asm volatile("" ::"v"(acc[0][0][0]));
            long long *d = g_conv_dbg + (((size_t)dbg_slot * 4 + wid) * 64 + c) * 4;
            d[0] = q0; d[1] = q1; d[2] = q2; d[3] = __builtin_readcyclecounter();
        }
#endif
    };
    stage(0, 0);
    for (int c = 0; c + 1 < n_chunks; c += 2) {        // straight-line body: no accumulator shuffling at a join
        chunk_step(c, std::integral_constant<int, 0>{});
        chunk_step(c + 1, std::integral_constant<int, 1>{});
    }
    if (n_chunks & 1) chunk_step(n_chunks - 1, std::integral_constant<int, 0>{});

    // ---- epilogue: bias (+ReLU) (+gate * y + residual), NHWC store.  The weights are the MFMA "A" operand, so a lane
    // holds ONE pixel (column lane&31 of the block) and 16 channels: rows (r&3) + 8*(r>>2) + 4*(lane>>5), i.e. four runs of
    // four consecutive channels -> float4 bias / residual loads and float4 stores, one address per pixel. ----
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int q = mb * 32 + l31;
        const int oy = UP2 ? oy0 + 2 * (wm * 4 + (l31 >> 3)) + (mb >> 1) : oy0 + wm * (TH / 2) + q / TW;
        const int ox = UP2 ? ox0 + 2 * (l31 & 7) + (mb & 1) : ox0 + q % TW;
        if (oy >= a.OH || ox >= a.OW) continue;
        const size_t pix = ((size_t)n * a.OH + oy) * a.OW + ox;
        const float gate = a.gate ? a.gate[pix] : 0.f;
        const float *rrow = a.gate ? a.resid + pix * a.resid_cstride : nullptr;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = co0 + wn * WN + nb * 32 + 8 * g + 4 * half;
                if (col >= a.cout_gemm) continue;
                const float4 bias = s_bias[(col - co0) >> 2];
                float4 y = make_float4(acc[mb][nb][4 * g] + bias.x, acc[mb][nb][4 * g + 1] + bias.y,
                                       acc[mb][nb][4 * g + 2] + bias.z, acc[mb][nb][4 * g + 3] + bias.w);
                if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
                int co = col, sy = 0, sx = 0;
                if (a.up > 1) { const int sub = col / a.cout_real; co = col - sub * a.cout_real; sy = sub / a.up; sx = sub % a.up; }
                if (a.gate) {
                    const float4 r = *(const float4 *)(rrow + co);
                    y.x = fmaf(gate, y.x, r.x); y.y = fmaf(gate, y.y, r.y); y.z = fmaf(gate, y.z, r.z); y.w = fmaf(gate, y.w, r.w);
                }
                const size_t opix = a.up > 1 ? ((size_t)n * a.OH * a.up + (oy * a.up + sy)) * ((size_t)a.OW * a.up) + (ox * a.up + sx) : pix;
                *(float4 *)(a.out + opix * a.out_cstride + a.out_coff + co) = y;
            }
        }
    }
    __syncthreads();   // the next tile re-zeroes and refills the LDS stages
#ifdef HVPR_EXP_TIMING
    ++dbg_tile;
#endif
    }
}

template <int TH, int TW, int BN, int S, int TAPS, bool UP2 = false>
int launch(ConvArgs a, hipStream_t s) {
    a.tiles_x = (a.OW + TW - 1) / TW;
    a.tiles_y = (a.OH + TH - 1) / TH;
    a.n_ct = a.cout_pad / BN;
    const long long tiles = (long long)a.N * a.tiles_x * a.tiles_y * a.n_ct;
    static int resident = 0;   // workgroups of this instantiation that fit the chip at once
    if (resident == 0) {
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_conv<TH, TW, BN, S, TAPS, UP2>, 256, 0) != hipSuccess || per_cu < 1)
            per_cu = 1;
        resident = per_cu * cus;
    }
    long long blocks = tiles < resident ? (tiles + 7) / 8 * 8 : resident;   // a multiple of 8: workgroup id % 8 = its XCD
    if (blocks > resident && resident >= 8) blocks = resident / 8 * 8;
    hipLaunchKernelGGL((k_conv<TH, TW, BN, S, TAPS, UP2>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    return 0;
}

}  // namespace

#ifdef HVPR_EXP_TIMING
extern "C" int hvpr_exp_conv_dbg(long long *host_out, int n_words) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_conv_dbg), sizeof(long long) * n_words) == hipSuccess ? 0 : -1;
}
#endif

// Data gradient of a 3x3 stride-2 (pad 1) convolution: dz [N, H, W, Cin] (the layer's OUTPUT gradient; Cin = the layer's output channels) ->
// dx [N, OH, OW, cout] (its input's gradient, H == (OH + 2 - 3) / 2 + 1 likewise W), w_packed = hvpr_conv_pack_weights_f32 of the layer's
// filter with the two channel axes swapped ((cout = layer Cin, Cin = layer Cout, 3, 3), taps not flipped), bias [cout_pad] (zeros).
extern "C" int hvpr_conv2d_s2_dgrad_nhwc_f32(const float *dz, int N, int H, int W, int Cin, const float *w_packed, const float *bias, int cout,
                                             int cout_pad, int OH, int OW, float *dx, int out_cstride, int out_coff, hvpr_stream_t stream) {
    if (!dz || !w_packed || !bias || !dx || N < 1 || H < 1 || W < 1 || Cin < 8 || cout < 1 || OH < 1 || OW < 1) return HVPR_ERR_INVALID_ARG;
    if (H != (OH + 2 - 3) / 2 + 1 || W != (OW + 2 - 3) / 2 + 1) return HVPR_ERR_INVALID_ARG;
    if (Cin % KC != 0 || cout % 4 != 0 || out_cstride % 4 != 0 || out_coff % 4 != 0 || cout_pad % 64 != 0 || cout_pad < cout) return HVPR_ERR_UNSUPPORTED;
    if ((long long)H * W * Cin * 4 >= (1ll << 31)) return HVPR_ERR_UNSUPPORTED;
    ConvArgs a;
    a.in = dz; a.wpk = w_packed; a.bias = bias; a.out = dx; a.gate = nullptr; a.resid = nullptr;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin;
    a.OH = OH; a.OW = OW;
    a.cout_gemm = cout; a.cout_pad = cout_pad; a.cout_real = cout;
    a.out_cstride = out_cstride; a.out_coff = out_coff; a.resid_cstride = 0;
    a.relu = 0; a.up = 1;
    launch<16, 16, 64, 1, 9, true>(a, (hipStream_t)stream);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_conv2d_nhwc_f32(const float *in, int N, int H, int W, int Cin, const float *w_packed,
                                    const float *bias, int taps, int stride, int cout, int cout_pad, int up,
                                    int relu, const float *gate, const float *resid, int resid_cstride, float *out,
                                    int out_cstride, int out_coff, int tile_cfg, hvpr_stream_t stream) {
    if (!in || !w_packed || !bias || !out || N < 1 || H < 1 || W < 1 || Cin < 8 || cout < 1) return HVPR_ERR_INVALID_ARG;
    if ((gate == nullptr) != (resid == nullptr)) return HVPR_ERR_INVALID_ARG;
    if (Cin % KC != 0 || (taps != 9 && taps != 1) || (stride != 1 && stride != 2) || up < 1) return HVPR_ERR_UNSUPPORTED;
    const bool wide1x1 = taps == 1 && Cin % (4 * KC) == 0;   // four cin chunks per LDS stage
    if (taps == 1 && stride != 1) return HVPR_ERR_UNSUPPORTED;
    if (up > 1 && (taps != 1 || gate)) return HVPR_ERR_UNSUPPORTED;
    // the epilogue moves four consecutive channels per access
    if (cout % 4 != 0 || out_cstride % 4 != 0 || out_coff % 4 != 0 || (resid && resid_cstride % 4 != 0)) return HVPR_ERR_UNSUPPORTED;
    if ((long long)H * W * Cin * 4 >= (1ll << 31)) return HVPR_ERR_UNSUPPORTED;      // byte offsets inside one image are formed in signed 32-bit arithmetic
    ConvArgs a;
    a.in = in; a.wpk = w_packed; a.bias = bias; a.out = out; a.gate = gate; a.resid = resid;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin;
    a.OH = taps == 9 ? (H + 2 - 3) / stride + 1 : H;
    a.OW = taps == 9 ? (W + 2 - 3) / stride + 1 : W;
    a.cout_gemm = up > 1 ? cout * up * up : cout;
    a.cout_pad = cout_pad; a.cout_real = cout;
    a.out_cstride = out_cstride; a.out_coff = out_coff; a.resid_cstride = resid_cstride;
    a.relu = relu; a.up = up;
    if (cout_pad < a.cout_gemm) return HVPR_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    // tile_cfg: 0 = 128 px x 128 ch, 1 = 64 px x 64 ch, 2 = 128 px x 64 ch
    const int bn = tile_cfg == 0 ? 128 : 64;
    if (cout_pad % bn != 0) return HVPR_ERR_INVALID_ARG;
#define HV_CASE(TH, TW, BN)                                                        \
    if (taps == 9 && stride == 1) launch<TH, TW, BN, 1, 9>(a, s);                   \
    else if (taps == 9 && stride == 2) launch<TH, TW, BN, 2, 9>(a, s);              \
    else if (wide1x1) launch<TH, TW, BN, 1, 4>(a, s);                                \
    else launch<TH, TW, BN, 1, 1>(a, s);
    if (tile_cfg == 0) { HV_CASE(8, 16, 128) }
    else if (tile_cfg == 1) { HV_CASE(8, 8, 64) }
    else if (tile_cfg == 2) { HV_CASE(8, 16, 64) }
    else return HVPR_ERR_INVALID_ARG;
#undef HV_CASE
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
