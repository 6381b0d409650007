// a11 (training) — weight gradient of the stride-1 3x3 convolutions in the Winograd F(2x2, 3x3) domain (fp32 matrix cores).
// Call site: the backward of every Conv3x3 of BaseBEVBackbone_Scale.forward in training mode,
// pcdet/models/backbones_2d/base_bev_backbone.py:228-279 (autograd of the nn.Conv2d layers built at :154-175).
//
// With  Y = At [ U . V ] A  per 2x2 output block (U = G g Gt, V = Bt d B — conv_wino.hip):
//     dL/dU[a][b] = sum_blocks (A dY At)[a][b] * V[a][b]         A = [1 0; 1 1; 1 -1; 0 -1]
//     dL/dg       = Gt (dL/dU) G
// i.e. 16 GEMMs  dU_xi[co][ci] = sum_block dM_xi[block][co] * V_xi[block][ci]  with K = blocks: 16 instead of 36 multiplies per
// block and (co, ci) pair.  M = co, N = ci, K = blocks; v_mfma_f32_32x32x2_f32 takes one float per lane and operand, lane =
// channel, the two K slots of an instruction = two neighbouring blocks (as k_wgrad in conv_train.hip).
//
// Workgroup = 8 waves, 64 co x 64 ci; wave (a = wid & 3, ch = wid >> 2) owns the four products xi = (a, 0..3) for co half `ch`:
// 4 x 2 accumulator blocks of 32 x 32 (128 registers).  A pixel tile is 8 x 16 outputs (4 x 8 blocks = 16 K steps): the dz tile
// [128 px][64 co] and the input patch [10 x 18 px][64 ci] are staged in LDS (next tile prefetched in registers).  Per K step a
// wave reads the 2x2 of dY for its co (4 ds_read_b32) and the two patch rows x four columns row `a` of Bt needs for two ci
// blocks (16 ds_read_b32), forms row a of (A dY At) and of (Bt d B) in registers (~20 VALU) and issues 8 MFMAs.  The two signs
// of A's last row / column are left out of the operands and applied by the reducing kernel.  Split-K over pixel tiles into
// partials, summed in a fixed order by k_wgrad_wino_reduce, which also applies Gt . G: deterministic.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WwArgs {
    const float *x;     // [N, H, W, Cin]
    const float *dz;    // [N, H, W, Cout]
    float *part;        // [n_chunks][16][Cout][Cin]
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y, n_pt, n_chunks, n_ci_tiles;
};

constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2;     // 8 x 16 outputs = 4 x 8 blocks = 16 K steps per tile
constexpr int BM = 64, BNN = 64;                     // co x ci of the workgroup
constexpr int NT = 512;
constexpr int DZ_V4 = TH * TW * BM / 4, X_V4 = PH * PW * BNN / 4;       // 1024, 1600
constexpr int NLD_D = DZ_V4 / NT, NLD_X = (X_V4 + NT - 1) / NT;         // 4, 6
constexpr int kLds = (TH * TW * BM + PH * PW * BNN) * 4;                // 78 KB: one workgroup per CU

__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) k_wgrad_wino(WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    float *const s_dz = smem_w;                        // [pixel][co]
    float *const s_x = smem_w + TH * TW * BM;          // [patch pixel][ci]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wa = wid & 3, ch = wid >> 2, half = lane >> 5, l31 = lane & 31;
    const int ot = blockIdx.x, chunk = blockIdx.y;
    const int co0 = (ot / a.n_ci_tiles) * BM, ci0 = (ot % a.n_ci_tiles) * BNN;

    f32x16 acc[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][nb][r] = 0.f;

    float4 rd[NLD_D], rx[NLD_X];
    auto fetch = [&](int pt) {          // global -> registers (zeros outside the image / past the channel counts)
        const int tx = pt % a.tiles_x, ty = (pt / a.tiles_x) % a.tiles_y, n = pt / (a.tiles_x * a.tiles_y);
        const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 - 1, ix0 = ox0 - 1;
#pragma unroll
        for (int i = 0; i < NLD_D; ++i) {
            const int v = tid + i * NT, px = v / (BM / 4), c4 = (v % (BM / 4)) * 4;
            const int oy = oy0 + px / TW, ox = ox0 + px % TW;
            const bool ok = oy < a.H && ox < a.W && co0 + c4 < a.Cout;
            rd[i] = ok ? *(const float4 *)(a.dz + (((size_t)n * a.H + oy) * a.W + ox) * a.Cout + co0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NLD_X; ++i) {
            const int v = tid + i * NT, px = v / (BNN / 4), c4 = (v % (BNN / 4)) * 4;
            const int iy = iy0 + px / PW, ix = ix0 + px % PW;
            const bool ok = v < X_V4 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && ci0 + c4 < a.Cin;
            rx[i] = ok ? *(const float4 *)(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.Cin + ci0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {               // registers -> LDS
#pragma unroll
        for (int i = 0; i < NLD_D; ++i) *(float4 *)(s_dz + (tid + i * NT) * 4) = rd[i];
#pragma unroll
        for (int i = 0; i < NLD_X; ++i)
            if (tid + i * NT < X_V4) *(float4 *)(s_x + (tid + i * NT) * 4) = rx[i];
    };

    // Row `wa` of A (for dY) and of Bt (for d) — wave-uniform selections folded into lane base pointers and two sign constants:
    //   A dY:   a=0: dY0     a=1: dY0 + dY1    a=2: dY0 - dY1    a=3: dY1 (true: -dY1)
    //   Bt d:   a=0: d0 - d2 a=1: d1 + d2      a=2: d2 - d1      a=3: d1 - d3
    const float sA = wa == 2 ? -1.f : 1.f;                        // t = first + sA * second
    const bool single = wa == 0 || wa == 3;                        // a = 0 / 3: a single row of dY
    const int ya0 = wa == 3 ? 1 : 0, ya1 = 1;                      // dY rows combined
    const int r0 = wa == 0 ? 0 : (wa == 2 ? 2 : 1), r1 = wa == 2 ? 1 : (wa == 3 ? 3 : 2);
    const float sB = wa == 1 ? 1.f : -1.f;
    // lane half h takes block 2m + h of the tile: by = m / (TW / 4), bx = 2 * (m % (TW / 4)) + h  -> two pixels to the right for h = 1
    const float *pa0 = s_dz + (ya0 * TW + 2 * half) * BM + ch * 32 + l31;       // + ((2 by) * TW + 2 bx0 + q) * BM
    const float *pa1 = s_dz + (ya1 * TW + 2 * half) * BM + ch * 32 + l31;
    const float *pb0 = s_x + (r0 * PW + 2 * half) * BNN + l31;                  // + ((2 by) * PW + 2 bx0 + j) * BNN + nb * 32
    const float *pb1 = s_x + (r1 * PW + 2 * half) * BNN + l31;

    int pt = chunk;
    if (pt < a.n_pt) fetch(pt);
    for (; pt < a.n_pt; pt += a.n_chunks) {
        __syncthreads();                 // everybody is done reading the previous tile
        commit();
        __syncthreads();
        if (pt + a.n_chunks < a.n_pt) fetch(pt + a.n_chunks);   // the next tile travels while this one multiplies
#pragma unroll
        for (int m = 0; m < TH * TW / 8; ++m) {
            const int by = m / (TW / 4), bx0 = 2 * (m % (TW / 4));
            const int oa = ((2 * by) * TW + 2 * bx0) * BM, ob = ((2 * by) * PW + 2 * bx0) * BNN;
            // A operand: row wa of (A dY At) for this lane's co; columns: (t0, t0 + t1, t0 - t1, t1 [true: -t1])
            const float y00 = pa0[oa], y01 = pa0[oa + BM], y10 = pa1[oa], y11 = pa1[oa + BM];
            const float t0 = single ? y00 : fmaf(sA, y10, y00), t1 = single ? y01 : fmaf(sA, y11, y01);
            const float am[4] = {t0, t0 + t1, t0 - t1, t1};
            // B operand: row wa of (Bt d B) for this lane's ci, two ci blocks
            float bm[2][4];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = fmaf(sB, pb1[ob + j * BNN + nb * 32], pb0[ob + j * BNN + nb * 32]);
                bm[nb][0] = t[0] - t[2]; bm[nb][1] = t[1] + t[2]; bm[nb][2] = t[2] - t[1]; bm[nb][3] = t[1] - t[3];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[b], bm[nb][b], acc[b][nb], 0, 0, 0);
        }
    }
    // partials [chunk][xi][co][ci]; C/D map of 32x32: column (ci) = lane & 31, row (co) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float *out = a.part + (size_t)chunk * 16 * a.Cout * a.Cin;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int ci = ci0 + nb * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + ch * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < a.Cout && ci < a.Cin) out[((size_t)(wa * 4 + b) * a.Cout + co) * a.Cin + ci] = acc[b][nb][r];
            }
        }
}

// dU[xi] = sum over chunks (chunk order), signs of A's last row / column, dg = Gt dU G  ->  dw [co][ci][3][3].
// Workgroup = 16 (co, ci) pairs x 16 products: thread (xi, pair) sums its product over the chunks (four independent partial sums:
// the chunk loop is a chain of loads), the 16 sums of a pair meet in LDS and one thread per pair applies Gt . G.
__global__ void __launch_bounds__(256) k_wgrad_wino_reduce(const float *__restrict__ part, int n_chunks, int Cout, int Cin,
                                                           float *__restrict__ dw) {
    __shared__ float s_u[16][17];
    const int pr = threadIdx.x & 15, xi = threadIdx.x >> 4;
    const long long per = (long long)Cout * Cin;
    const long long i = (long long)blockIdx.x * 16 + pr;      // over [co][ci]
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < per) {
        const float *src = part + (size_t)xi * per + i;
        const size_t step = (size_t)16 * per;
        int c = 0;
        for (; c + 3 < n_chunks; c += 4) {
            s0 += src[(size_t)c * step]; s1 += src[(size_t)(c + 1) * step]; s2 += src[(size_t)(c + 2) * step]; s3 += src[(size_t)(c + 3) * step];
        }
        for (; c < n_chunks; ++c) s0 += src[(size_t)c * step];
    }
    const float s = (s0 + s1) + (s2 + s3);
    const bool neg = ((xi >> 2) == 3) != ((xi & 3) == 3);
    s_u[pr][xi] = neg ? -s : s;
    __syncthreads();
    if (threadIdx.x >= 16 || (long long)blockIdx.x * 16 + threadIdx.x >= per) return;
    const float *u = s_u[threadIdx.x];                        // u[a * 4 + b]
    // Gt = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
    float h[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        h[0][b] = u[b] + 0.5f * (u[4 + b] + u[8 + b]);
        h[1][b] = 0.5f * (u[4 + b] - u[8 + b]);
        h[2][b] = 0.5f * (u[4 + b] + u[8 + b]) + u[12 + b];
    }
    float *o = dw + ((size_t)blockIdx.x * 16 + threadIdx.x) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[r * 3 + 0] = h[r][0] + 0.5f * (h[r][1] + h[r][2]);
        o[r * 3 + 1] = 0.5f * (h[r][1] - h[r][2]);
        o[r * 3 + 2] = 0.5f * (h[r][1] + h[r][2]) + h[r][3];
    }
}

int ww_chunks(int N, int H, int W, int Cin, int Cout) {
    const long long n_pt = (long long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    const int n_ot = ((Cin + BNN - 1) / BNN) * ((Cout + BM - 1) / BM);
    long long chunks = (256 + n_ot - 1) / n_ot;                 // one 8-wave workgroup per CU
    if (chunks > n_pt) chunks = n_pt;
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}

}  // namespace

extern "C" size_t hvpr_conv2d_wino_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    if (N < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
    return (size_t)ww_chunks(N, H, W, Cin, Cout) * 16 * Cout * Cin * sizeof(float);
}

extern "C" int hvpr_conv2d_wino_wgrad_nhwc_f32(const float *x, int N, int H, int W, int Cin, const float *dz, int Cout, float *dw,
                                               void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!x || !dz || !dw || !workspace || N < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return HVPR_ERR_INVALID_ARG;
    if (Cin % 4 != 0 || Cout % 4 != 0) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < hvpr_conv2d_wino_wgrad_workspace_bytes(N, H, W, Cin, Cout)) return HVPR_ERR_WORKSPACE;
    WwArgs a;
    a.x = x; a.dz = dz; a.part = (float *)workspace;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.tiles_x = (W + TW - 1) / TW;
    a.tiles_y = (H + TH - 1) / TH;
    a.n_pt = N * a.tiles_x * a.tiles_y;
    a.n_ci_tiles = (Cin + BNN - 1) / BNN;
    a.n_chunks = ww_chunks(N, H, W, Cin, Cout);
    const int n_ot = a.n_ci_tiles * ((Cout + BM - 1) / BM);
    hipStream_t s = (hipStream_t)stream;
    static unsigned long long lds_set = 0ull;
    if (hvpr_ensure_dyn_lds((const void *)k_wgrad_wino, kLds, &lds_set) != 0) return HVPR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_wgrad_wino, dim3(n_ot, a.n_chunks), dim3(NT), kLds, s, a);
    const long long per = (long long)Cout * Cin;
    hipLaunchKernelGGL(k_wgrad_wino_reduce, dim3(hvpr_cdiv(per, 16)), dim3(256), 0, s, a.part, a.n_chunks, Cout, Cin, dw);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
