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
// LDS holds both tiles CHANNEL-major, [channel][pixel]: a lane (= channel) finds the neighbouring pixels of a block side by side
// and takes them with 8-byte reads.  Channel pitches 2 * odd: 32 consecutive channels land on 32 distinct even banks of 64.
constexpr int PA = TH * TW + 2, PB = PH * PW + 2;                       // 130, 182 floats
constexpr int DZ_IT = (TH * TW / 2) * (BM / 4), X_IT = (PH * PW / 2) * (BNN / 4);      // (pixel pair, 4 channels) items: 1024, 1440
constexpr int NLD_D = DZ_IT / NT, NLD_X = (X_IT + NT - 1) / NT, NIT = NLD_D + NLD_X;    // 2 + 3 items per thread
constexpr int STAGE_F = BM * PA + BNN * PB;                             // floats per stage: 19 968
constexpr int kLds = 2 * STAGE_F * 4;                                   // two stages, 156 KB: one workgroup per CU
constexpr int K_STEPS = TH * TW / 8;                                    // 16
static_assert(DZ_IT % NT == 0 && TW % 2 == 0 && PW % 2 == 0, "pixel pairs must not straddle tile rows");
static_assert(2 * NIT <= K_STEPS - 2, "the staging units of a tile are dealt to its K steps");
static_assert(X_IT - (NLD_X - 1) * NT > 0 && X_IT >= (NLD_X - 1) * NT, "the last patch item is partly live");

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) float *lds_cfp;       // an LDS address stays an LDS address through an empty asm
typedef __attribute__((address_space(3))) float *lds_fp;
#define LD2(p) (*(const __attribute__((address_space(3))) f32x2 *)(p))
#define ST2(p) (*(__attribute__((address_space(3))) f32x2 *)(p))

// One workgroup walks its pixel tiles with everything that is not an MFMA dealt to the K steps of the tile before: while tile k
// multiplies out of LDS stage k & 1, the registers holding tile k + 1 are written to the other stage (one 2-channel store pair per
// K step, steps 0-9) and re-filled with tile k + 2 (one 16-byte buffer load per step, steps 2-11; out-of-image / past-the-end
// pieces take an out-of-range offset and come back as zeros, so the loop has no branch).  One barrier per tile.  (Measured before:
// issuing the 10 loads of a tile in one go cost 3.3 k cycles and the 10 LDS store pairs 2.4 k of the 25 k cycles per tile, with the
// matrix cores idle — tools/wgrad_phases.py.)
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) k_wgrad_wino(WwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    const lds_fp s0 = (lds_fp)smem_w;                  // stage: [co][pixel] pitch PA, then [ci][patch pixel] pitch PB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wa = wid & 3, ch = wid >> 2, half = lane >> 5, l31 = lane & 31;
    // Workgroup -> (output tile, chunk).  Consecutive workgroup ids go round the 8 XCDs; the workgroups of one chunk read the same
    // pixel tiles at the same time (each dz tile n_ci_tiles times, each patch n_co_tiles times), so they are put on ONE XCD — behind
    // one L2 — by giving every XCD a contiguous range of (chunk, output tile) pairs.
    int vid = blockIdx.x;
    {
        const int total = gridDim.x;
        if (total % 8 == 0) vid = (vid & 7) * (total >> 3) + (vid >> 3);
    }
    const int n_ot = a.n_ci_tiles * ((a.Cout + BM - 1) / BM);
    const int ot = vid % n_ot, chunk = vid / n_ot;
    const int co0 = (ot / a.n_ci_tiles) * BM, ci0 = (ot % a.n_ci_tiles) * BNN;

    f32x16 acc[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][nb][r] = 0.f;

    // ---- staging items of this thread (tile-independent part).  An item = two neighbouring pixels x four channels: two 16-byte
    // loads, four 8-byte LDS stores (one per channel).  Items 0..1: dz tile, 2..4: input patch; the lanes past the end of the patch
    // list repeat their own previous item (same addresses, same values). ----
    // it_off: byte offset of the item's first pixel from the tile's first output pixel, inside the (channel-tile-based) image
    int it_dy[NIT], it_dx[NIT], it_lds[NIT];
    unsigned it_off[NIT];
    constexpr int kNever = 1 << 30;                     // a row number no image has: marks dead channel groups / tiles past the end
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        if (i < NLD_D) {
            const int v = tid + i * NT, pp = v / (BM / 4), c4 = (v % (BM / 4)) * 4;
            it_dy[i] = pp / (TW / 2); it_dx[i] = 2 * (pp % (TW / 2));
            it_lds[i] = c4 * PA + 2 * pp;
            it_off[i] = (unsigned)(((it_dy[i] * a.W + it_dx[i]) * a.Cout + c4) * 4);
            if (co0 + c4 >= a.Cout) it_dy[i] = kNever;
        } else {
            int v = tid + (i - NLD_D) * NT;
            if (v >= X_IT) v -= NT;
            const int pp = v / (BNN / 4), c4 = (v % (BNN / 4)) * 4;
            it_dy[i] = pp / (PW / 2) - 1; it_dx[i] = 2 * (pp % (PW / 2)) - 1;
            it_lds[i] = BM * PA + c4 * PB + 2 * pp;
            it_off[i] = (unsigned)(((it_dy[i] * a.W + it_dx[i]) * a.Cin + c4) * 4);
            if (ci0 + c4 >= a.Cin) it_dy[i] = kNever;
        }
    }
    const unsigned img_dz = (unsigned)a.H * a.W * a.Cout * 4u, img_x = (unsigned)a.H * a.W * a.Cin * 4u;    // bytes per image (< 4 GB: launcher)
    f32x4 rg[NIT][2];
    struct Tile { int oy0, ox0; unsigned off_dz, off_x; const float *img_dz, *img_x; };     // all wave-uniform
    auto tile_of = [&](int k) {
        const int pt = chunk + k * a.n_chunks;
        const bool valid = pt < a.n_pt;
        const int q = valid ? pt : 0;
        const int n = q / (a.tiles_x * a.tiles_y);
        Tile t;
        t.ox0 = (q % a.tiles_x) * TW; t.oy0 = ((q / a.tiles_x) % a.tiles_y) * TH;
        t.off_dz = (unsigned)((t.oy0 * a.W + t.ox0) * a.Cout * 4); t.off_x = (unsigned)((t.oy0 * a.W + t.ox0) * a.Cin * 4);
        t.img_dz = a.dz + (size_t)n * a.H * a.W * a.Cout + co0; t.img_x = a.x + (size_t)n * a.H * a.W * a.Cin + ci0;
        if (!valid) t.oy0 = kNever;
        return t;
    };
    auto load_unit = [&](int u, const Tile &t) {       // unit u: pixel u & 1 of item u >> 1
        const int i = u >> 1, w = u & 1;
        const bool is_dz = i < NLD_D;
        const int y = t.oy0 + it_dy[i], x = t.ox0 + it_dx[i] + w, C = is_dz ? a.Cout : a.Cin;
        const bool ok = (int)((unsigned)y < (unsigned)a.H) & (int)((unsigned)x < (unsigned)a.W);
        // a piece outside the image (or of a dead channel group, or of a tile past the end) gets an offset past num_records (< 2^31):
        // the load returns zeros
        const unsigned off = (is_dz ? t.off_dz : t.off_x) + it_off[i] + (unsigned)(w * C * 4);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(is_dz ? t.img_dz : t.img_x), 0, (int)(is_dz ? img_dz : img_x), 0x00020000);
        rg[i][w] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ok ? off : 0x80000000u, 0, 0));
    };
    auto store_unit = [&](int u, lds_fp stage) {       // unit u: channels 2 (u & 1), 2 (u & 1) + 1 of item u >> 1, both pixels
        const int i = u >> 1, e = 2 * (u & 1);
        const int P = i < NLD_D ? PA : PB;
        const lds_fp d = stage + it_lds[i] + e * P;
        ST2(d) = f32x2{rg[i][0][e], rg[i][1][e]};
        ST2(d + P) = f32x2{rg[i][0][e + 1], rg[i][1][e + 1]};
    };

    // Row `wa` of A (for dY) and of Bt (for d) — wave-uniform selections folded into lane base pointers and two coefficients:
    //   A dY:   a=0: dY0     a=1: dY0 + dY1    a=2: dY0 - dY1    a=3: dY1 (true: -dY1)      = first + cA * dY1
    //   Bt d:   a=0: d0 - d2 a=1: d1 + d2      a=2: d2 - d1      a=3: d1 - d3              = first + sB * second
    const float cA = wa == 1 ? 1.f : (wa == 2 ? -1.f : 0.f);
    const int ya0 = wa == 3 ? 1 : 0;
    const int r0 = wa == 0 ? 0 : (wa == 2 ? 2 : 1), r1 = wa == 2 ? 1 : (wa == 3 ? 3 : 2);
    const float sB = wa == 1 ? 1.f : -1.f;
    const f32x2 cA2 = {cA, cA}, sB2 = {sB, sB};
    // lane half h takes block 2m + h of the tile: by = m / (TW / 4), bx = 2 * (m % (TW / 4)) + h  -> two pixels to the right for h = 1
    const int oa0 = (ch * 32 + l31) * PA + ya0 * TW + 2 * half;            // + (2 by) * TW + 2 bx0: the 2x2 of dY, one row per read
    const int oa1 = (ch * 32 + l31) * PA + TW + 2 * half;
    const int ob0 = BM * PA + l31 * PB + r0 * PW + 2 * half;               // + nb * 32 * PB + (2 by) * PW + 2 bx0 (+ 2): four columns of a patch row
    const int ob1 = BM * PA + l31 * PB + r1 * PW + 2 * half;

    int n_tiles = 0;
    if (chunk < a.n_pt) n_tiles = (a.n_pt - chunk + a.n_chunks - 1) / a.n_chunks;
    {
        const Tile t0 = tile_of(0);
#pragma unroll
        for (int u = 0; u < 2 * NIT; ++u) load_unit(u, t0);
#pragma unroll
        for (int u = 0; u < 2 * NIT; ++u) store_unit(u, s0);
        const Tile t1 = tile_of(1);
#pragma unroll
        for (int u = 0; u < 2 * NIT; ++u) load_unit(u, t1);
    }
    __syncthreads();
#ifdef HVPR_EXP_TIMING
    unsigned long long tph[K_STEPS + 2], tprev, tnow;
    for (int i = 0; i < K_STEPS + 2; ++i) tph[i] = 0;
#define WW_NOW(t) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define WW_STAMP(k) do { WW_NOW(tnow); tph[k] += tnow - tprev; tprev = tnow; } while (0)
    WW_NOW(tprev);
#else
#define WW_STAMP(k)
#endif
    for (int k = 0; k < n_tiles; ++k) {
        const int cur = (k & 1) * STAGE_F;
        const lds_fp wr = s0 + (STAGE_F - cur);        // tile k + 1's stage
        const Tile t2 = tile_of(k + 2);
        // (LDS-typed pointers kept in registers by the empty asm: the K steps then differ only in the offset fields of the reads —
        //  no address arithmetic in the loop)
        lds_cfp pa0 = s0 + cur + oa0, pa1 = s0 + cur + oa1, pb0[2], pb1[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            pb0[nb] = s0 + cur + ob0 + nb * 32 * PB;
            pb1[nb] = s0 + cur + ob1 + nb * 32 * PB;
            asm volatile("" : "+v"(pb0[nb]), "+v"(pb1[nb]));
        }
        asm volatile("" : "+v"(pa0), "+v"(pa1));
        // the raw operands of step m + 1 are requested before step m's MFMAs are issued (a step's chain is then its ~20 vector
        // instructions + its MFMAs, without the LDS round trip)
        f32x2 qa0, qa1, qb0[2][2], qb1[2][2];
        auto request = [&](int m) {
            const int by = m / (TW / 4), bx0 = 2 * (m % (TW / 4));
            const int oa = (2 * by) * TW + 2 * bx0, ob = (2 * by) * PW + 2 * bx0;
            qa0 = LD2(pa0 + oa); qa1 = LD2(pa1 + oa);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                qb0[nb][0] = LD2(pb0[nb] + ob); qb0[nb][1] = LD2(pb0[nb] + ob + 2);
                qb1[nb][0] = LD2(pb1[nb] + ob); qb1[nb][1] = LD2(pb1[nb] + ob + 2);
            }
        };
        request(0);
#pragma unroll
        for (int m = 0; m < K_STEPS; ++m) {
            // A operand: row wa of (A dY At) for this lane's co; columns: (t0, t0 + t1, t0 - t1, t1 [true: -t1])
            const f32x2 t = __builtin_elementwise_fma(cA2, qa1, qa0);
            const float am[4] = {t.x, t.x + t.y, t.x - t.y, t.y};
            // B operand: row wa of (Bt d B) for this lane's ci, two ci blocks
            float bm[2][4];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f32x2 t01 = __builtin_elementwise_fma(sB2, qb1[nb][0], qb0[nb][0]);
                const f32x2 t23 = __builtin_elementwise_fma(sB2, qb1[nb][1], qb0[nb][1]);
                const f32x2 d = t01 - t23;                      // (t0 - t2, t1 - t3)
                bm[nb][0] = d.x; bm[nb][1] = t01.y + t23.x; bm[nb][2] = t23.x - t01.y; bm[nb][3] = d.y;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (m + 1 < K_STEPS) request(m + 1);
            __builtin_amdgcn_sched_barrier(0);
            // the staging units sit BETWEEN the MFMAs (in the shadow of the one just issued), not in front of the step's reads: an LDS
            // store ahead of the reads is waited for with them (LDS returns in order) and lengthens the step's dependent chain
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[b], bm[nb][b], acc[b][nb], 0, 0, 0);
                if (b == 0 && m < 2 * NIT) {
                    __builtin_amdgcn_sched_barrier(0);
                    store_unit(m, wr);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (b == 1 && m >= 2 && m < 2 + 2 * NIT) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_unit(m - 2, t2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            WW_STAMP(m + 1);
        }
        __syncthreads();                 // everybody is done reading stage `cur` and writing the other one
        WW_STAMP(K_STEPS + 1);
    }
#ifdef HVPR_EXP_TIMING
    if (lane == 0 && (wid == 0 || wid == 5) && ((ot == 0 && chunk == 0) || (ot == 1 && chunk == 3)))
        printf("k_wgrad_wino wg (%d, %d) wave %d: %d tiles; cycles per tile: K steps %llu %llu %llu %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | %llu %llu %llu %llu | "
               "barrier %llu\n", ot, chunk, wid, n_tiles, tph[1] / n_tiles, tph[2] / n_tiles, tph[3] / n_tiles, tph[4] / n_tiles, tph[5] / n_tiles,
               tph[6] / n_tiles, tph[7] / n_tiles, tph[8] / n_tiles, tph[9] / n_tiles, tph[10] / n_tiles, tph[11] / n_tiles, tph[12] / n_tiles,
               tph[13] / n_tiles, tph[14] / n_tiles, tph[15] / n_tiles, tph[16] / n_tiles, tph[17] / n_tiles);
#endif
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
    if ((unsigned long long)H * W * (Cin > Cout ? Cin : Cout) * sizeof(float) >= (1ull << 31)) return HVPR_ERR_UNSUPPORTED;   // 32-bit offsets inside an image
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
    hipLaunchKernelGGL(k_wgrad_wino, dim3(n_ot * a.n_chunks), dim3(NT), kLds, s, a);
    const long long per = (long long)Cout * Cin;
    hipLaunchKernelGGL(k_wgrad_wino_reduce, dim3(hvpr_cdiv(per, 16)), dim3(256), 0, s, a.part, a.n_chunks, Cout, Cin, dw);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
