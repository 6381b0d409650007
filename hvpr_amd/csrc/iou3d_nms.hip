// a8 — rotated-BEV overlap / IoU / 3D IoU / NMS on device.  Replaces the reference's absent native module
// pcdet/ops/iou3d_nms (sources named by setup.py:53-62; call sites: model_utils/model_nms_utils.py:17-19,
// detectors/detector3d_template.py:298,303).  Boxes are [x, y, z, dx, dy, dz, heading].
//
// The geometry follows, operation for operation, the same fp32 sequence as the CPU oracle
// (oracle/iou3d_nms_ref.c): both are compiled with -ffp-contract=off so that survivor ids are bit-exact.
// MI355X-specific parts: 64 x 64 tiles = one wave per tile and one 64-bit mask word per lane; an exact
// circum-circle reject (no polygon work for far-apart pairs: overlap is exactly 0 there); the sequential sweep
// runs ON DEVICE in one wave (mask words live in lanes, diagonal block resolved with readlane, rows of kept
// boxes OR-ed in with independent coalesced loads) — no mask copy to the host, no host loop, no sync.
#include "common.h"

namespace {

constexpr float kEps = 1e-8f;
constexpr float kMargin = 1e-2f;

struct P2 { float x, y; };

__device__ __forceinline__ float cross3(P2 a, P2 b, P2 o) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }

__device__ __forceinline__ bool rect_overlap(P2 p1, P2 p2, P2 q1, P2 q2) {
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ bool seg_intersect(P2 p1, P2 p0, P2 q1, P2 q0, P2 &out) {
    if (!rect_overlap(p0, p1, q0, q1)) return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0.0f && s3 * s4 > 0.0f)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > kEps) {
        out.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        out.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        out.x = (b0 * c1 - b1 * c0) / D;
        out.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

struct Box {
    float x, y, dx, dy, r;   // BEV part
    float cs, sn;            // cos/sin(heading)
    float ncs, nsn;          // cos/sin(-heading)
    P2 c[5];                 // rotated corners, closed
};

__device__ __forceinline__ void make_box(const float *__restrict__ b, Box &B) {
    B.x = b[0]; B.y = b[1]; B.dx = b[3]; B.dy = b[4]; B.r = b[6];
    B.cs = cosf(B.r); B.sn = sinf(B.r);
    B.ncs = cosf(-B.r); B.nsn = sinf(-B.r);
    const float hx = B.dx / 2, hy = B.dy / 2;
    const float x1 = B.x - hx, y1 = B.y - hy, x2 = B.x + hx, y2 = B.y + hy;
    const float rx[4] = {x1, x2, x2, x1}, ry[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        B.c[k].x = (rx[k] - B.x) * B.cs + (ry[k] - B.y) * (-B.sn) + B.x;
        B.c[k].y = (rx[k] - B.x) * B.sn + (ry[k] - B.y) * B.cs + B.y;
    }
    B.c[4] = B.c[0];
}

__device__ __forceinline__ bool in_box(const Box &B, P2 p) {
    const float rx = (p.x - B.x) * B.ncs + (p.y - B.y) * (-B.nsn);
    const float ry = (p.x - B.x) * B.nsn + (p.y - B.y) * B.ncs;
    return fabsf(rx) < B.dx / 2 + kMargin && fabsf(ry) < B.dy / 2 + kMargin;
}

// Exact reject: if the centres are farther apart than the two circum-radii plus the in-box margin, no edge pair can
// intersect and no corner can pass the margin test, so the polygon routine would return exactly 0.
__device__ __forceinline__ bool far_apart(const Box &A, const Box &B) {
    const float ra = 0.5f * sqrtf(A.dx * A.dx + A.dy * A.dy), rb = 0.5f * sqrtf(B.dx * B.dx + B.dy * B.dy);
    const float ddx = A.x - B.x, ddy = A.y - B.y;
    const float lim = ra + rb + 0.05f;
    return ddx * ddx + ddy * ddy > lim * lim * 1.0001f;
}

// Exact reject no. 2, separating axes: if the two rectangles are more than 5 cm apart along one of their four edge
// normals, no edges cross and no corner passes the 1 cm in-box margin (that margin widens a box by at most 1.42 cm in any
// direction), so the polygon routine would return exactly 0.  About half of the pairs the circum-circle test lets through
// (cars: circum-radius 2.1 m around a 3.9 x 1.6 m box) end here, for ~35 flops instead of the ~18 k-cycle clip.
struct BoxLite { float x, y, hx, hy, cs, sn, rad, pad; };   // 32 bytes: two broadcast 16-byte LDS reads

__device__ __forceinline__ BoxLite lite_of(const Box &B) {
    return BoxLite{B.x, B.y, 0.5f * B.dx, 0.5f * B.dy, B.cs, B.sn, 0.5f * sqrtf(B.dx * B.dx + B.dy * B.dy), 0.f};
}

// branch-free (all lanes run it on densely packed pairs)
__device__ __forceinline__ bool axes_separate(const BoxLite &A, const BoxLite &B) {
    const float ddx = B.x - A.x, ddy = B.y - A.y;
    const float c = fabsf(A.cs * B.cs + A.sn * B.sn), sn = fabsf(B.sn * A.cs - B.cs * A.sn);   // |cos|, |sin| of the heading difference
    const float m = 0.05f;
    const bool s0 = fabsf(ddx * A.cs + ddy * A.sn) > A.hx + B.hx * c + B.hy * sn + m;
    const bool s1 = fabsf(ddy * A.cs - ddx * A.sn) > A.hy + B.hx * sn + B.hy * c + m;
    const bool s2 = fabsf(ddx * B.cs + ddy * B.sn) > B.hx + A.hx * c + A.hy * sn + m;
    const bool s3 = fabsf(ddy * B.cs - ddx * B.sn) > B.hy + A.hx * sn + A.hy * c + m;
    return s0 | s1 | s2 | s3;
}

// Per-lane polygon scratch in LDS: up to 24 candidate vertices (16 edge crossings + 8 corners), lane-minor so that
// dynamic indexing is a conflict-free ds access.  (As a private array this spills to scratch memory: 300 B/lane and
// an order of magnitude slower.)
struct PolyStore {
    float x[24][64], y[24][64], a[24][64];
};

__device__ float box_overlap(const Box &A, const Box &B, PolyStore &ps, int ln) {
    if (far_apart(A, B)) return 0.0f;
    int cnt = 0;
    P2 ctr = {0.0f, 0.0f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 o;
            if (seg_intersect(A.c[i + 1], A.c[i], B.c[j + 1], B.c[j], o)) {
                ps.x[cnt][ln] = o.x; ps.y[cnt][ln] = o.y; ctr.x += o.x; ctr.y += o.y; ++cnt;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (in_box(A, B.c[k])) { ctr.x += B.c[k].x; ctr.y += B.c[k].y; ps.x[cnt][ln] = B.c[k].x; ps.y[cnt][ln] = B.c[k].y; ++cnt; }
        if (in_box(B, A.c[k])) { ctr.x += A.c[k].x; ctr.y += A.c[k].y; ps.x[cnt][ln] = A.c[k].x; ps.y[cnt][ln] = A.c[k].y; ++cnt; }
    }
    if (cnt == 0) return 0.0f;
    ctr.x /= (float)cnt; ctr.y /= (float)cnt;
    // bubble sort by polar angle: the angle of a vertex does not change while it is moved around, so it is computed
    // once per vertex instead of twice per comparison (identical comparisons, identical order)
    for (int i = 0; i < cnt; ++i) ps.a[i][ln] = atan2f(ps.y[i][ln] - ctr.y, ps.x[i][ln] - ctr.x);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            const float a0 = ps.a[i][ln], a1 = ps.a[i + 1][ln];
            if (a0 > a1) {
                const float tx = ps.x[i][ln], ty = ps.y[i][ln];
                ps.x[i][ln] = ps.x[i + 1][ln]; ps.y[i][ln] = ps.y[i + 1][ln]; ps.a[i][ln] = a1;
                ps.x[i + 1][ln] = tx; ps.y[i + 1][ln] = ty; ps.a[i + 1][ln] = a0;
            }
        }
    float area = 0.0f;
    const float x0 = ps.x[0][ln], y0 = ps.y[0][ln];
    for (int k = 0; k < cnt - 1; ++k) {
        const float ux = ps.x[k][ln] - x0, uy = ps.y[k][ln] - y0;
        const float wx = ps.x[k + 1][ln] - x0, wy = ps.y[k + 1][ln] - y0;
        area += ux * wy - uy * wx;
    }
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const Box &A, const Box &B, PolyStore &ps, int ln) {
    const float sa = A.dx * A.dy, sb = B.dx * B.dy;
    const float so = box_overlap(A, B, ps, ln);
    return so / fmaxf(sa + sb - so, kEps);
}

// ---- pairwise (N,M) kernels: mode 0 overlap, 1 bev iou, 2 3d iou -----------------------------------------
__global__ void __launch_bounds__(64) k_pairwise(const float *__restrict__ a, int n, const float *__restrict__ b, int m,
                                                 int mode, float *__restrict__ out) {
    __shared__ PolyStore ps;
    const int ln = threadIdx.x;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * m) return;
    const int i = (int)(t / m), j = (int)(t % m);
    const float *pa = a + (size_t)i * 7, *pb = b + (size_t)j * 7;
    Box A, B;
    make_box(pa, A);
    make_box(pb, B);
    float r;
    if (mode == 0) r = box_overlap(A, B, ps, ln);
    else if (mode == 1) r = iou_bev(A, B, ps, ln);
    else {
        const float a_top = pa[2] + pa[5] / 2, a_bot = pa[2] - pa[5] / 2;
        const float b_top = pb[2] + pb[5] / 2, b_bot = pb[2] - pb[5] / 2;
        const float va = pa[3] * pa[4] * pa[5], vb = pb[3] * pb[4] * pb[5];
        const float ob = box_overlap(A, B, ps, ln);
        const float oh = fmaxf(fminf(a_top, b_top) - fmaxf(a_bot, b_bot), 0.0f);
        const float o3 = ob * oh;
        r = o3 / fmaxf(va + vb - o3, 1e-6f);
    }
    out[t] = r;
}

// ---- NMS: bit-mask tiles -------------------------------------------------------------------------------
// k_nms_prep: candidate i = boxes[order ? order[i] : i] -> Box (corners + sin/cos computed once per box, not once per tile)
__global__ void __launch_bounds__(256) k_nms_prep(const float *__restrict__ boxes, int box_stride, const int *__restrict__ order,
                                                  const int *__restrict__ n_device, int n_max, Box *__restrict__ prepared) {
    const int n = n_device ? min(*n_device, n_max) : n_max;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Box B;
    make_box(boxes + (size_t)(order ? order[i] : i) * box_stride, B);
    prepared[i] = B;
}

// One wave per 64 x 64 tile.  Phase 1: every lane tests its row box against the 64 column boxes with the two exact
// rejects only (circum-circle, then separating axes: a few flops on a 32-byte box record).  Phase 2: the surviving (row, col) pairs of the whole tile are compacted into
// LDS and dealt out to the lanes round-robin, so the expensive polygon clip runs on densely packed lanes instead of
// 64 divergent column iterations.  Results are identical to testing every pair (the reject is exact).
__global__ void __launch_bounds__(64) k_nms_mask(const Box *__restrict__ prepared, const int *__restrict__ n_device, int n_max,
                                                 float thresh, unsigned long long *__restrict__ mask, int nb) {
    const int n = n_device ? min(*n_device, n_max) : n_max;
    const int rb = blockIdx.y, cb = blockIdx.x;
    const int t = threadIdx.x;
    const int row = rb * 64 + t;
    if (rb * 64 >= n || cb * 64 >= n) return;
    if (cb < rb) {
        if (row < n) mask[(size_t)row * nb + cb] = 0ull;
        return;
    }
    __shared__ Box s_row[64], s_col[64];
    __shared__ __attribute__((aligned(16))) BoxLite s_lrow[64], s_lcol[64];
    __shared__ __attribute__((aligned(16))) float4 s_circ[64];   // column boxes: centre and circum-radius
    __shared__ unsigned long long s_bits[64];
    __shared__ unsigned short s_pairs[64 * 32];   // one half of the columns at a time: 37 KB of LDS = four tiles per CU
    __shared__ PolyStore ps;
    const int col = cb * 64 + t;
    if (row < n) s_row[t] = prepared[row];
    if (row < n) s_lrow[t] = lite_of(s_row[t]);
    if (col < n) {
        s_col[t] = prepared[col];
        s_lcol[t] = lite_of(s_col[t]);
        s_circ[t] = make_float4(s_lcol[t].x, s_lcol[t].y, s_lcol[t].rad, 0.f);
    }
    s_bits[t] = 0ull;
    __syncthreads();
    const int ncol = min(64, n - cb * 64);
#ifdef HVPR_EXP_TIMING
    const long long tq0 = __builtin_readcyclecounter();
#endif
    int total2_all = 0;   // (only the HVPR_EXP_TIMING build reports it)
    (void)total2_all;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int c_lo = 32 * half;
        unsigned near = 0u;
        if (row < n) {   // reject no. 1 (the circum-circle test of far_apart()), one broadcast 16-byte LDS read per column box
            const float mx = s_lrow[t].x, my = s_lrow[t].y, mr = s_lrow[t].rad + 0.05f;
            const int i0 = (rb == cb) ? t + 1 : 0;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) {
                const float4 c = s_circ[c_lo + i];
                const float ddx = mx - c.x, ddy = my - c.y, lim = mr + c.z;
                const bool close = !(ddx * ddx + ddy * ddy > lim * lim * 1.0001f);
                if (close && c_lo + i >= i0 && c_lo + i < ncol) near |= 1u << i;
            }
        }
        // exclusive prefix of the per-lane pair counts
        const int cnt = __popc(near);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (t >= o) incl += u;
        }
        const int total = __shfl(incl, 63, 64);
        int pos = incl - cnt;
        while (near) {
            const int i = __ffs((int)near) - 1;
            near &= near - 1;
            s_pairs[pos++] = (unsigned short)((t << 6) | (c_lo + i));
        }
        __syncthreads();
        // reject no. 2 (separating axes) on the packed pair list, compacted in place (one wave: writes trail reads)
        int total2 = 0;
        for (int k0 = 0; k0 < total; k0 += 64) {
            const int k = k0 + t;
            const int pr = k < total ? s_pairs[k] : 0;
            const bool keep = k < total && !axes_separate(s_lrow[pr >> 6], s_lcol[pr & 63]);
            const unsigned long long m = __ballot(keep);
            if (keep) s_pairs[total2 + __popcll(m & ((1ull << t) - 1ull))] = (unsigned short)pr;
            total2 += __popcll(m);
        }
        __syncthreads();
        for (int k = t; k < total2; k += 64) {
            const int pr = s_pairs[k];
            const int r = pr >> 6, c = pr & 63;
            if (iou_bev(s_row[r], s_col[c], ps, t) > thresh) atomicOr(&s_bits[r], 1ull << c);
        }
        __syncthreads();
        total2_all += total2;
    }
    __syncthreads();
#ifdef HVPR_EXP_TIMING
    if (t == 0 && rb % 8 == 0 && cb % 8 == 0)
        printf("nms tile %d %d: %d pairs clipped, %lld cycles\n", rb, cb, total2_all, (long long)__builtin_readcyclecounter() - tq0);
#endif
    if (row < n) mask[(size_t)row * nb + cb] = s_bits[t];
}

// ---- NMS mask in two launches (n <= 4096): the rejects per tile, then the polygon clips over ONE balanced list --------------
// k_nms_mask above clips a tile's surviving pairs inside the tile's wave: tiles in dense regions of the scene hold thousands of
// pairs, tiles elsewhere none, and the launch lasts as long as its heaviest tiles (97 us for 4096 candidates of which the rejects
// leave a few 10^4 pairs).  Here the tile waves only run the two exact rejects and write their survivors to a per-tile segment of
// a global list (no atomics); k_nms_clip then walks the concatenation of all segments with every lane of the chip holding the
// same number of pairs, and sets the mask bits with atomicOr (order-independent: the mask, and with it the survivors, are
// identical to the one-launch form).
constexpr int kTileCap = 64 * 64;                     // pairs a tile can hold
__host__ __device__ inline int tri_index(int rb, int cb, int nb) { return rb * nb - rb * (rb - 1) / 2 + (cb - rb); }

__global__ void __launch_bounds__(64) k_nms_pairs(const Box *__restrict__ prepared, const int *__restrict__ n_device, int n_max,
                                                  unsigned long long *__restrict__ mask, int nb, unsigned short *__restrict__ plist,
                                                  int *__restrict__ pcount) {
    const int n = n_device ? min(*n_device, n_max) : n_max;
    const int rb = blockIdx.y, cb = blockIdx.x;
    const int t = threadIdx.x;
    const int row = rb * 64 + t;
    if (rb * 64 >= n || cb * 64 >= n) return;
    if (row < n) mask[(size_t)row * nb + cb] = 0ull;      // k_nms_clip ORs the suppression bits in
    if (cb < rb) return;
    __shared__ __attribute__((aligned(16))) BoxLite s_lrow[64], s_lcol[64];
    __shared__ __attribute__((aligned(16))) float4 s_circ[64];   // column boxes: centre and circum-radius
    __shared__ unsigned short s_pairs[64 * 32];
    const int col = cb * 64 + t;
    if (row < n) s_lrow[t] = lite_of(prepared[row]);
    if (col < n) {
        s_lcol[t] = lite_of(prepared[col]);
        s_circ[t] = make_float4(s_lcol[t].x, s_lcol[t].y, s_lcol[t].rad, 0.f);
    }
    __syncthreads();
    const int ncol = min(64, n - cb * 64);
    const int tile = tri_index(rb, cb, nb);
    unsigned short *out = plist + (size_t)tile * kTileCap;
    int n_out = 0;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int c_lo = 32 * half;
        unsigned near = 0u;
        if (row < n) {   // reject no. 1 (the circum-circle test of far_apart())
            const float mx = s_lrow[t].x, my = s_lrow[t].y, mr = s_lrow[t].rad + 0.05f;
            const int i0 = (rb == cb) ? t + 1 : 0;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) {
                const float4 c = s_circ[c_lo + i];
                const float ddx = mx - c.x, ddy = my - c.y, lim = mr + c.z;
                const bool close = !(ddx * ddx + ddy * ddy > lim * lim * 1.0001f);
                if (close && c_lo + i >= i0 && c_lo + i < ncol) near |= 1u << i;
            }
        }
        const int cnt = __popc(near);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (t >= o) incl += u;
        }
        const int total = __shfl(incl, 63, 64);
        int pos = incl - cnt;
        while (near) {
            const int i = __ffs((int)near) - 1;
            near &= near - 1;
            s_pairs[pos++] = (unsigned short)((t << 6) | (c_lo + i));
        }
        __syncthreads();
        // reject no. 2 (separating axes) on the packed pair list; survivors go to the tile's global segment
        for (int k0 = 0; k0 < total; k0 += 64) {
            const int k = k0 + t;
            const int pr = k < total ? s_pairs[k] : 0;
            const bool keep = k < total && !axes_separate(s_lrow[pr >> 6], s_lcol[pr & 63]);
            const unsigned long long m = __ballot(keep);
            if (keep) out[n_out + __popcll(m & ((1ull << t) - 1ull))] = (unsigned short)pr;
            n_out += __popcll(m);
        }
        __syncthreads();
    }
    if (t == 0) pcount[tile] = n_out;
}

// exclusive prefix of the live tiles' pair counts, in (rb, cb >= rb) order, + the tile of every position: one workgroup
constexpr int kMaxTiles = 2080;                       // nb <= 64
struct ClipIndex { int pre[kMaxTiles + 1]; unsigned char rb[kMaxTiles], cb[kMaxTiles]; };

__global__ void __launch_bounds__(1024) k_nms_scan(const int *__restrict__ n_device, int n_max, int nb, const int *__restrict__ pcount,
                                                   ClipIndex *__restrict__ ci) {
    __shared__ int s_wave[16];
    const int n = n_device ? min(*n_device, n_max) : n_max;
    const int nba = (n + 63) / 64, n_tiles = nba * (nba + 1) / 2;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    // thread t owns live tiles 3t .. 3t + 2  (3 * 1024 >= 2080)
    int rb = 0, rem = 3 * t;
    while (rb < nba && rem >= nba - rb) { rem -= nba - rb; ++rb; }
    int cb = rb + rem;
    int c[3], trb[3], tcb[3], run = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const bool live = 3 * t + i < n_tiles;
        trb[i] = rb; tcb[i] = cb;
        c[i] = live ? pcount[tri_index(rb, cb, nb)] : 0;
        run += c[i];
        if (live && ++cb == nba) { ++rb; cb = rb; }
    }
    int incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wid] = incl;
    __syncthreads();
    int base = incl - run;
    for (int w = 0; w < wid; ++w) base += s_wave[w];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (3 * t + i < n_tiles) {
            ci->pre[3 * t + i] = base;
            ci->rb[3 * t + i] = (unsigned char)trb[i]; ci->cb[3 * t + i] = (unsigned char)tcb[i];
        }
        base += c[i];
    }
    if (t == 1023) ci->pre[n_tiles] = base;
}

constexpr int kClipThreads = 128, kClipBlocks = 768;  // three workgroups per CU (a PolyStore per wave, 49 KB of LDS each)
__global__ void __launch_bounds__(kClipThreads) k_nms_clip(const Box *__restrict__ prepared, const int *__restrict__ n_device, int n_max,
                                                           float thresh, unsigned long long *__restrict__ mask, int nb,
                                                           const unsigned short *__restrict__ plist, const ClipIndex *__restrict__ ci) {
    __shared__ int s_pre[kMaxTiles + 1];
    __shared__ unsigned char s_rb[kMaxTiles], s_cb[kMaxTiles];
    __shared__ PolyStore ps[kClipThreads / 64];
    const int n = n_device ? min(*n_device, n_max) : n_max;
    const int nba = (n + 63) / 64, n_tiles = nba * (nba + 1) / 2;
    const int t = threadIdx.x;
    for (int i = t; i <= n_tiles; i += kClipThreads) s_pre[i] = ci->pre[i];
    for (int i = t; i < n_tiles; i += kClipThreads) { s_rb[i] = ci->rb[i]; s_cb[i] = ci->cb[i]; }
    __syncthreads();
    const int total = s_pre[n_tiles];
    PolyStore &my = ps[t >> 6];
    for (long long gidx = (long long)blockIdx.x * kClipThreads + t; gidx < total; gidx += (long long)gridDim.x * kClipThreads) {
        // the tile holding pair gidx: largest i with s_pre[i] <= gidx
        int lo = 0, hi = n_tiles;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pre[mid] <= (int)gidx) lo = mid; else hi = mid;
        }
        const int rb = s_rb[lo], cb = s_cb[lo];
        const int pr = plist[(size_t)tri_index(rb, cb, nb) * kTileCap + ((int)gidx - s_pre[lo])];
        const int row = rb * 64 + (pr >> 6), c = pr & 63;
        const Box A = prepared[row], B = prepared[cb * 64 + c];
        if (iou_bev(A, B, my, t & 63) > thresh) atomicOr(&mask[(size_t)row * nb + cb], 1ull << c);
    }
}

// ---- NMS: sequential sweep, one wave ----------------------------------------------------------------------
// keep[] receives positions in the sorted order (or order[pos] when `order` is given and map_through_order != 0).
__global__ void __launch_bounds__(64) k_nms_sweep(const unsigned long long *__restrict__ mask, int nb_stride,
                                                  const int *__restrict__ n_device, int n_max,
                                                  const int *__restrict__ order, int map_through_order, int max_keep,
                                                  int *__restrict__ keep, int *__restrict__ keep_count) {
    const int n = n_device ? min(*n_device, n_max) : n_max;
    const int lane = threadIdx.x;
    const int nb = (n + 63) / 64;
    int kept = 0;
    // remv words: lane w owns word w of chunk (w / 64); n_max <= 64*64*? -> up to NW words per lane
    constexpr int NW = 4;   // supports n <= 64 * 64 * NW = 16384
    unsigned long long remv[NW] = {0ull, 0ull, 0ull, 0ull};
    // diagonal words of the first 64 blocks are fetched up front (64 independent loads per lane), so the per-block
    // critical path is the resolve loop plus ONE batch of row loads
    constexpr int NPRE = 64;
    unsigned long long dpre[NPRE];
#pragma unroll
    for (int b = 0; b < NPRE; ++b) {
        const int row = b * 64 + lane;
        dpre[b] = (b < nb && row < n) ? mask[(size_t)row * nb_stride + b] : 0ull;
    }
    for (int b = 0; b < nb && kept < max_keep; ++b) {
        const int row = b * 64 + lane;
        unsigned long long diag = 0ull;
        if (b < NPRE) {
#pragma unroll
            for (int q = 0; q < NPRE; ++q) if (q == b) diag = dpre[q];
        } else {
            diag = row < n ? mask[(size_t)row * nb_stride + b] : 0ull;
        }
        // remv word of this block (uniform)
        unsigned long long rw = 0ull;
#pragma unroll
        for (int q = 0; q < NW; ++q)
            if ((b >> 6) == q) {
                const unsigned lo = __builtin_amdgcn_readlane((unsigned)(remv[q] & 0xffffffffull), b & 63);
                const unsigned hi = __builtin_amdgcn_readlane((unsigned)(remv[q] >> 32), b & 63);
                rw = ((unsigned long long)hi << 32) | lo;
            }
        unsigned long long kbits = 0ull;
        const int lim = min(64, n - b * 64);
        for (int i = 0; i < lim && kept < max_keep; ++i) {
            if (!((rw >> i) & 1ull)) {
                kbits |= 1ull << i;
                const unsigned lo = __builtin_amdgcn_readlane((unsigned)(diag & 0xffffffffull), i);
                const unsigned hi = __builtin_amdgcn_readlane((unsigned)(diag >> 32), i);
                rw |= ((unsigned long long)hi << 32) | lo;
                ++kept;
            }
        }
        // kept positions of this block -> keep[] (lane i writes its own entry)
        {
            const int base = kept - __popcll(kbits);
            if ((kbits >> lane) & 1ull) {
                const int pos = b * 64 + lane;
                keep[base + __popcll(kbits & ((1ull << lane) - 1ull))] = (order && map_through_order) ? order[pos] : pos;
            }
        }
        // OR the rows of the kept boxes into the later words: eight independent coalesced row loads in flight
        unsigned long long kb = kbits;
        while (kb) {
            int idx8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                idx8[u] = kb ? __ffsll((long long)kb) - 1 : -1;
                kb &= kb - 1;
            }
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const int w = q * 64 + lane;
                if (q * 64 < nb && w > b && w < nb) {
                    unsigned long long t8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        t8[u] = idx8[u] >= 0 ? mask[(size_t)(b * 64 + idx8[u]) * nb_stride + w] : 0ull;
                    remv[q] |= ((t8[0] | t8[1]) | (t8[2] | t8[3])) | ((t8[4] | t8[5]) | (t8[6] | t8[7]));
                }
            }
        }
    }
    if (lane == 0) *keep_count = kept;
}

// The greedy sweep for n <= 4096 candidates (nb <= 64 mask words per row).  The sweep is serial over the 64-box blocks and
// needs, per block, the mask rows of the boxes it keeps: read on demand that is one dependent L2 round trip per block
// (64 x ~3.5 us).  Here one workgroup of 8 waves streams ALL rows through an LDS ring, one block (64 rows x nb words, 32 KB)
// per phase, and wave 0 resolves phase p out of LDS.  Four of the eight loader waves OWN the phases p = g (mod 2), 16 rows
// each: they request the rows of their next phase two phases before they are needed and keep them in registers until the
// ring slot is free — two L2 round trips in flight, the barriers wait for LDS only (rounds 1-3: all loaders fetched phase
// p + 1 during phase p behind __syncthreads(), whose vmcnt(0) made every phase one round trip long: 71 us; now 59, bound
// by the ~300 dependent instructions wave 0 spends per block; more loader waves slow that wave down: 13 waves 63 us).  The
// diagonal words and the candidate ids travel through LDS too (a column read of the ring is a 64-way bank conflict, order[]
// a dependent global load).  Same greedy order, same result.
constexpr int kRingThreads = 576;                    // the resolving wave + 8 loaders
constexpr int kRingOwners = 4;                       // loader waves that share a phase (16 rows each)
constexpr int kRingDepth = (kRingThreads / 64 - 1) / kRingOwners;     // 2 phases in flight
constexpr int kRingWords = 64;                       // words per row in LDS (nb <= 64)
constexpr int kRingPieces = (64 / kRingOwners) * (kRingWords / 2) / 64;   // 16-byte pieces per lane and phase: 8

__global__ void __launch_bounds__(kRingThreads) k_nms_sweep_ring(const unsigned long long *__restrict__ mask, int nb_stride,
                                                                 const int *__restrict__ n_device, int n_max,
                                                                 const int *__restrict__ order, int map_through_order,
                                                                 int max_keep, int *__restrict__ keep, int *__restrict__ keep_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_rows[];   // [2][64][kRingWords]
    __shared__ int s_done;
    // per ring slot: the diagonal word of every row (row r of block b, word b: a column of s_rows — 64 lanes on one bank) and the
    // candidates' ids, so that the resolve reads both without a bank conflict / without a dependent global load
    __shared__ unsigned long long s_diag[2][64];
    __shared__ int s_ord[2][64];
    const int n = n_device ? min(*n_device, n_max) : n_max;
    const int nb = (n + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // a loader lane holds pieces (row 32 * half + 2u + lane / 32, words 2c, 2c + 1 with c = lane % 32), u = 0..15, of its phase
    ulonglong2 v[kRingPieces];
    int ord = 0;
    const bool ord_lane = order && map_through_order && wid != 0 && (wid - 1) % kRingOwners == 0;
    const int pr = (lane >> 5) + (64 / kRingOwners) * ((wid - 1) % kRingOwners), pc = lane & 31, grp = (wid - 1) / kRingOwners;
    // global -> registers: 32 UNCONDITIONAL loads back to back, no select on an address or a result (the compiler turns those into
    // a branch and a wait per load).  What is not wanted is never looked at by the resolve, so it may hold anything: rows past n
    // (clamped to the last row of the buffer), words left of the diagonal block (the lower triangle; those lanes fetch the
    // diagonal piece again — same cache line, no extra traffic) and words past nb.
    const bool even = !(nb_stride & 1);             // 16-byte aligned pieces whose two words both exist
    auto request = [&](int ph) {
        const int pcc = max(pc, (ph - 1) >> 1);
        if (ord_lane) ord = order[min(ph * 64 + lane, n_max - 1)];
        if (even) {
            const int col = 2 * min(pcc, (nb_stride >> 1) - 1);
#pragma unroll
            for (int u = 0; u < kRingPieces; ++u)
                v[u] = *(const ulonglong2 *)(mask + (size_t)min(ph * 64 + 2 * u + pr, n_max - 1) * nb_stride + col);
        } else {
            const int c0 = min(2 * pcc, nb_stride - 1), c1 = min(2 * pcc + 1, nb_stride - 1);
#pragma unroll
            for (int u = 0; u < kRingPieces; ++u) {
                const size_t base = (size_t)min(ph * 64 + 2 * u + pr, n_max - 1) * nb_stride;
                v[u].x = mask[base + c0];
                v[u].y = mask[base + c1];
            }
        }
    };
    auto commit = [&](int ph) {         // registers -> ring slot ph & 1
        unsigned long long *dst = s_rows + (size_t)(ph & 1) * 64 * kRingWords;
#pragma unroll
        for (int u = 0; u < kRingPieces; ++u) *(ulonglong2 *)(dst + (size_t)(2 * u + pr) * kRingWords + 2 * pc) = v[u];
        if (pc == (ph >> 1)) {
            // (bit masks, not `odd ? v.y : v.x`: the compiler turns that select into a dynamic index and the whole array into scratch)
            const unsigned long long odd = 0ull - (unsigned long long)(ph & 1);
#pragma unroll
            for (int u = 0; u < kRingPieces; ++u) s_diag[ph & 1][2 * u + pr] = (v[u].x & ~odd) | (v[u].y & odd);
        }
        if (ord_lane) s_ord[ph & 1][lane] = ord;
    };
    if (tid == 0) s_done = 0;
    if (wid != 0) request(grp);         // phases 0..3
    if (wid != 0 && grp == 0) { commit(0); request(kRingDepth); }
    // The barriers of this kernel order LDS traffic only: __syncthreads() would also wait for every outstanding GLOBAL load of the
    // wave (vmcnt(0)) — exactly the round trip per phase the ring is there to hide.  The loaded registers are guarded by the
    // compiler's own vmcnt waits at their first use (commit).
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    lds_barrier();
    unsigned long long remv = 0ull;                  // wave 0: lane w owns suppression word w
    int kept = 0;
    for (int ph = 0; ph < nb; ++ph) {
        if (wid != 0) {
            // the owner of phase ph + 1 hands it over (its slot was read last in phase ph - 1) and requests its next one
            if ((ph + 1) % kRingDepth == grp && ph + 1 < nb && !s_done) {
                commit(ph + 1);
                request(ph + 1 + kRingDepth);
            }
        } else if (kept < max_keep) {
            const unsigned long long *rows = s_rows + (size_t)(ph & 1) * 64 * kRingWords;
            const int b = ph;
            const unsigned long long diag = s_diag[ph & 1][lane];                  // row `lane` of block b, word b
            const int my_id = (order && map_through_order) ? s_ord[ph & 1][lane] : b * 64 + lane;
            unsigned long long rw;
            {
                const unsigned lo = __builtin_amdgcn_readlane((unsigned)(remv & 0xffffffffull), b);
                const unsigned hi = __builtin_amdgcn_readlane((unsigned)(remv >> 32), b);
                rw = ((unsigned long long)hi << 32) | lo;
            }
            const int lim = min(64, n - b * 64);
            const unsigned long long valid = lim == 64 ? ~0ull : ((1ull << lim) - 1ull);
            unsigned long long kbits = 0ull;
            unsigned long long avail = ~rw & valid;
            while (avail && kept < max_keep) {                 // only unsuppressed candidates are visited
                const int i = __ffsll((long long)avail) - 1;
                kbits |= 1ull << i;
                const unsigned lo = __builtin_amdgcn_readlane((unsigned)(diag & 0xffffffffull), i);
                const unsigned hi = __builtin_amdgcn_readlane((unsigned)(diag >> 32), i);
                rw |= ((unsigned long long)hi << 32) | lo;
                ++kept;
                avail = ~rw & valid & ~((2ull << i) - 1ull);
            }
            const int base = kept - __popcll(kbits);
            if ((kbits >> lane) & 1ull) keep[base + __popcll(kbits & ((1ull << lane) - 1ull))] = my_id;
            // rows of the kept boxes, straight from LDS, four at a time: the reads of a round are issued together (one LDS
            // latency per round instead of one per kept box)
            for (unsigned long long kb = kbits; kb;) {
                unsigned long long t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = kb ? __ffsll((long long)kb) - 1 : -1;
                    kb &= kb - 1ull;
                    t4[u] = i >= 0 ? rows[(size_t)i * kRingWords + lane] : 0ull;
                }
                remv |= (t4[0] | t4[1]) | (t4[2] | t4[3]);
            }
            if (kept >= max_keep && lane == 0) s_done = 1;
        }
        lds_barrier();
    }
    if (tid == 0) *keep_count = kept;
}

}  // namespace

extern "C" int hvpr_boxes_pairwise_f32(const float *boxes_a, int n, const float *boxes_b, int m, int mode, float *out,
                                       hvpr_stream_t stream) {
    if (n < 0 || m < 0 || mode < 0 || mode > 2) return HVPR_ERR_INVALID_ARG;
    if (n == 0 || m == 0) return HVPR_OK;
    if (!boxes_a || !boxes_b || !out) return HVPR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_pairwise, dim3(hvpr_cdiv((long long)n * m, 64)), dim3(64), 0, (hipStream_t)stream, boxes_a, n,
                       boxes_b, m, mode, out);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" size_t hvpr_nms_workspace_bytes(int n_max) {
    if (n_max < 1) return 0;
    const size_t nb = (n_max + 63) / 64;
    size_t bytes = (size_t)n_max * nb * sizeof(unsigned long long) + 256 + (size_t)n_max * sizeof(Box) + 256;
    if (nb <= 64) {     // the two-launch mask: per-tile pair segments + counts
        const size_t tiles = nb * (nb + 1) / 2;
        bytes += tiles * kTileCap * sizeof(unsigned short) + 256 + tiles * sizeof(int) + 256 + sizeof(ClipIndex) + 256;
    }
    return bytes;
}

extern "C" int hvpr_nms_bev_f32(const float *boxes, int box_stride, const int32_t *order, const int32_t *n_device,
                                int n_max, float thresh, int max_keep, int map_through_order, int32_t *keep,
                                int32_t *keep_count, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (n_max < 0 || box_stride < 7 || max_keep < 0 || !keep_count) return HVPR_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n_max == 0) {
        if (hipMemsetAsync(keep_count, 0, sizeof(int), s) != hipSuccess) return HVPR_ERR_LAUNCH;
        return HVPR_OK;
    }
    if (!boxes || !keep || !workspace) return HVPR_ERR_INVALID_ARG;
    if (n_max > 16384) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < hvpr_nms_workspace_bytes(n_max)) return HVPR_ERR_WORKSPACE;
    const int nb = (n_max + 63) / 64;
    unsigned long long *mask = (unsigned long long *)workspace;
    Box *prepared = (Box *)((char *)workspace + (((size_t)n_max * nb * sizeof(unsigned long long) + 255) / 256) * 256);
    hipLaunchKernelGGL(k_nms_prep, dim3(hvpr_cdiv(n_max, 256)), dim3(256), 0, s, boxes, box_stride, order, n_device, n_max, prepared);
    if (nb <= 64) {
        const size_t tiles = (size_t)nb * (nb + 1) / 2;
        char *p = (char *)prepared + (((size_t)n_max * sizeof(Box) + 255) / 256) * 256;
        unsigned short *plist = (unsigned short *)p;
        int *pcount = (int *)(p + ((tiles * kTileCap * sizeof(unsigned short) + 255) / 256) * 256);
        ClipIndex *ci = (ClipIndex *)((char *)pcount + ((tiles * sizeof(int) + 255) / 256) * 256);
        hipLaunchKernelGGL(k_nms_pairs, dim3(nb, nb), dim3(64), 0, s, prepared, n_device, n_max, mask, nb, plist, pcount);
        hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(1024), 0, s, n_device, n_max, nb, pcount, ci);
        hipLaunchKernelGGL(k_nms_clip, dim3(kClipBlocks), dim3(kClipThreads), 0, s, prepared, n_device, n_max, thresh, mask, nb, plist, ci);
    } else {
        hipLaunchKernelGGL(k_nms_mask, dim3(nb, nb), dim3(64), 0, s, prepared, n_device, n_max, thresh, mask, nb);
    }
    if (nb <= kRingWords) {
        const size_t lds = (size_t)2 * 64 * kRingWords * sizeof(unsigned long long);
        static unsigned long long lds_set = 0ull;   // per device
    if (hvpr_ensure_dyn_lds((const void *)k_nms_sweep_ring, (int)lds, &lds_set) != 0) return HVPR_ERR_LAUNCH;
        hipLaunchKernelGGL(k_nms_sweep_ring, dim3(1), dim3(kRingThreads), lds, s, mask, nb, n_device, n_max, order, map_through_order,
                           max_keep, keep, keep_count);
    } else {
        hipLaunchKernelGGL(k_nms_sweep, dim3(1), dim3(64), 0, s, mask, nb, n_device, n_max, order, map_through_order, max_keep,
                           keep, keep_count);
    }
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
