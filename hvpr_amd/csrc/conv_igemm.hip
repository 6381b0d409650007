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
// chunk are staged in LDS, the next chunk is prefetched into registers while the current one is multiplied
// (issue-early / write-late), one LDS buffer.  Operand trick: lane half h reads channels 4h..4h+3 of the chunk
// as ONE ds_read_b128 for A and for B and feeds them to four consecutive MFMAs — the MFMA k index is a free
// permutation as long as A and B agree.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KC = 8;   // input channels per K chunk

struct ConvArgs {
    const float *in;      // [N, H, W, Cin]
    const float *wpk;     // [TAPS, Cin/8, CoutPad, 8]   (BN scale folded)
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

template <int TH, int TW, int BN, int S, int TAPS>
__global__ void __launch_bounds__(256) k_conv(ConvArgs a) {
    constexpr int BM = TH * TW;
    constexpr int WM = BM / 2, WN = BN / 2;        // per-wave tile
    constexpr int MB = WM / 32, NB = WN / 32;
    static_assert(MB >= 1 && NB >= 1, "wave tile must hold at least one 32x32 block");
    constexpr int HALO = TAPS == 9 ? 2 : 0;
    constexpr int PH = (TH - 1) * S + 1 + HALO, PW = (TW - 1) * S + 1 + HALO;
    constexpr int PATCH_V4 = PH * PW * 2;          // float4 per chunk
    constexpr int W_V4 = TAPS * BN * 2;
    constexpr int NLD_P = (PATCH_V4 + 255) / 256, NLD_W = (W_V4 + 255) / 256;

    __shared__ __attribute__((aligned(16))) float4 s_patch[PATCH_V4];
    __shared__ __attribute__((aligned(16))) float4 s_w[W_V4];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // workgroup -> (cout tile, pixel tile): cout tile fastest so that an XCD (block id % 8) keeps one weight slab in its L2
    const int bid = blockIdx.x;
    const int ct = bid % a.n_ct;
    int pt = bid / a.n_ct;
    const int tx = pt % a.tiles_x; pt /= a.tiles_x;
    const int ty = pt % a.tiles_y;
    const int n = pt / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - (HALO / 2), ix0 = ox0 * S - (HALO / 2);
    const int co0 = ct * BN;

    // ---- global -> register staging descriptors (constant over the K loop) ----
    const float *pin[NLD_P];
    bool pok[NLD_P];
#pragma unroll
    for (int i = 0; i < NLD_P; ++i) {
        const int v = tid + i * 256;
        const int pix = v >> 1, part = v & 1;
        const int py = pix / PW, px = pix % PW;
        const int iy = iy0 + py, ix = ix0 + px;
        pok[i] = v < PATCH_V4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        pin[i] = a.in + (((size_t)n * a.H + (pok[i] ? iy : 0)) * a.W + (pok[i] ? ix : 0)) * a.Cin + part * 4;
    }
    const float *pw[NLD_W];
    bool wok[NLD_W];
    const size_t w_chunk_stride = (size_t)a.cout_pad * KC;                 // floats between cin chunks
    const size_t w_tap_stride = (size_t)(a.Cin / KC) * w_chunk_stride;     // floats between taps
#pragma unroll
    for (int i = 0; i < NLD_W; ++i) {
        const int v = tid + i * 256;
        wok[i] = v < W_V4;
        const int tap = v / (BN * 2), rem = v % (BN * 2);                  // rem = co_local*2 + part
        pw[i] = a.wpk + (size_t)(wok[i] ? tap : 0) * w_tap_stride + (size_t)co0 * KC + rem * 4;
    }

    float4 rp[NLD_P], rw[NLD_W];
    auto issue = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NLD_P; ++i)
            rp[i] = pok[i] ? *(const float4 *)(pin[i] + chunk * KC) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NLD_W; ++i)
            rw[i] = wok[i] ? *(const float4 *)(pw[i] + (size_t)chunk * w_chunk_stride) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NLD_P; ++i) if (tid + i * 256 < PATCH_V4) s_patch[tid + i * 256] = rp[i];
#pragma unroll
        for (int i = 0; i < NLD_W; ++i) if (tid + i * 256 < W_V4) s_w[tid + i * 256] = rw[i];
    };

    // ---- per-lane LDS read offsets (float4 units) ----
    int a_off[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int q = mb * 32 + l31;
        const int py = wm * (TH / 2) + q / TW, px = q % TW;
        a_off[mb] = ((py * S) * PW + px * S) * 2 + half;
    }
    int b_off[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b_off[nb] = (wn * WN + nb * 32 + l31) * 2 + half;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    const int n_chunks = a.Cin / KC;
    issue(0);
    commit();
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        if (c + 1 < n_chunks) issue(c + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = TAPS == 9 ? tap / 3 : 0, kx = TAPS == 9 ? tap % 3 : 0;
            float4 av[MB], bv[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = s_patch[a_off[mb] + (ky * PW + kx) * 2];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = s_w[b_off[nb] + tap * BN * 2];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb].x, bv[nb].x, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb].y, bv[nb].y, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb].z, bv[nb].z, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb].w, bv[nb].w, acc[mb][nb], 0, 0, 0);
                }
        }
        __syncthreads();
        if (c + 1 < n_chunks) {
            commit();
            __syncthreads();
        }
    }

    // ---- epilogue: bias (+ReLU) (+gate * y + residual), NHWC store; C/D map: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5) ----
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = co0 + wn * WN + nb * 32 + l31;
        if (col >= a.cout_gemm) continue;
        const float bias = a.bias[col];
        int co = col, sub = 0;
        if (a.up > 1) { sub = col / a.cout_real; co = col - sub * a.cout_real; }
        const int sy = a.up > 1 ? sub / a.up : 0, sx = a.up > 1 ? sub % a.up : 0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int q = mb * 32 + row;
                const int oy = oy0 + wm * (TH / 2) + q / TW, ox = ox0 + q % TW;
                if (oy >= a.OH || ox >= a.OW) continue;
                float y = acc[mb][nb][r] + bias;
                if (a.relu) y = fmaxf(y, 0.f);
                const size_t pix = ((size_t)n * a.OH + oy) * a.OW + ox;
                if (a.gate) y = fmaf(a.gate[pix], y, a.resid[pix * a.resid_cstride + co]);
                const size_t opix = a.up > 1 ? ((size_t)n * a.OH * a.up + (oy * a.up + sy)) * ((size_t)a.OW * a.up) + (ox * a.up + sx) : pix;
                a.out[opix * a.out_cstride + a.out_coff + co] = y;
            }
        }
    }
}

template <int TH, int TW, int BN, int S, int TAPS>
int launch(ConvArgs a, hipStream_t s) {
    a.tiles_x = (a.OW + TW - 1) / TW;
    a.tiles_y = (a.OH + TH - 1) / TH;
    a.n_ct = a.cout_pad / BN;
    const long long blocks = (long long)a.N * a.tiles_x * a.tiles_y * a.n_ct;
    hipLaunchKernelGGL((k_conv<TH, TW, BN, S, TAPS>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    return 0;
}

}  // namespace

extern "C" int hvpr_conv2d_nhwc_f32(const float *in, int N, int H, int W, int Cin, const float *w_packed,
                                    const float *bias, int taps, int stride, int cout, int cout_pad, int up,
                                    int relu, const float *gate, const float *resid, int resid_cstride, float *out,
                                    int out_cstride, int out_coff, int tile_cfg, hvpr_stream_t stream) {
    if (!in || !w_packed || !bias || !out || N < 1 || H < 1 || W < 1 || Cin < 8 || cout < 1) return HVPR_ERR_INVALID_ARG;
    if ((gate == nullptr) != (resid == nullptr)) return HVPR_ERR_INVALID_ARG;
    if (Cin % KC != 0 || (taps != 9 && taps != 1) || (stride != 1 && stride != 2) || up < 1) return HVPR_ERR_UNSUPPORTED;
    if (taps == 1 && stride != 1) return HVPR_ERR_UNSUPPORTED;
    if (up > 1 && (taps != 1 || gate)) return HVPR_ERR_UNSUPPORTED;
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
    else launch<TH, TW, BN, 1, 1>(a, s);
    if (tile_cfg == 0) { HV_CASE(8, 16, 128) }
    else if (tile_cfg == 1) { HV_CASE(8, 8, 64) }
    else if (tile_cfg == 2) { HV_CASE(8, 16, 64) }
    else return HVPR_ERR_INVALID_ARG;
#undef HV_CASE
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
