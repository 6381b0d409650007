/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (hvpr_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / reported CPU baseline.
 *
 * CPU restatement of the rotated-BEV overlap / IoU / NMS natives the reference
 * calls but does not ship:
 *   call sites  pcdet/models/model_utils/model_nms_utils.py:17-19  (nms_gpu)
 *               pcdet/models/detectors/detector3d_template.py:298,303 (boxes_iou3d_gpu)
 *               pcdet/datasets/augmentor/database_sampler.py:184-185 (boxes_bev_iou_cpu)
 *   sources     pcdet/ops/iou3d_nms/src/ (cpp + cu) are named by setup.py:53-62 but
 *               ABSENT from /root/reference (OpenPCDet v0.3.0, pcdet/version.py:1).
 * PARITY UNPINNED: no reference source, test or golden vector exists for these
 * functions; this file restates the published algorithm:
 * boxes [x, y, z, dx, dy, dz, heading], polygon-clip overlap with in-box margin
 * 1e-2 and EPS 1e-8, iou = overlap / max(sa + sb - overlap, 1e-8), strict '>' test.
 *
 * Arithmetic is fp32 with no FMA contraction (compile with -ffp-contract=off) so
 * that the HIP kernel, built the same way, can follow the identical op sequence.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HV_EPS 1e-8f
#define HV_MARGIN 1e-2f

typedef struct { float x, y; } pt2;

static inline float cross3(pt2 a, pt2 b, pt2 o)
{
    return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

static inline int rect_overlap(pt2 p1, pt2 p2, pt2 q1, pt2 q2)
{
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

/* proper intersection of segment p0-p1 with q0-q1 */
static int seg_intersect(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2 *out)
{
    if (!rect_overlap(p0, p1, q0, q1)) return 0;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0.0f && s3 * s4 > 0.0f)) return 0;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > HV_EPS) {
        out->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        out->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        out->x = (b0 * c1 - b1 * c0) / D;
        out->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static inline pt2 rotate_about(pt2 c, float cs, float sn, pt2 p)
{
    pt2 r;
    r.x = (p.x - c.x) * cs + (p.y - c.y) * (-sn) + c.x;
    r.y = (p.x - c.x) * sn + (p.y - c.y) * cs + c.y;
    return r;
}

static inline int in_box2d(const float *box, pt2 p)
{
    const float cs = cosf(-box[6]), sn = sinf(-box[6]);
    const float rx = (p.x - box[0]) * cs + (p.y - box[1]) * (-sn);
    const float ry = (p.x - box[0]) * sn + (p.y - box[1]) * cs;
    return fabsf(rx) < box[3] / 2 + HV_MARGIN && fabsf(ry) < box[4] / 2 + HV_MARGIN;
}

static void box_corners(const float *b, pt2 *c /*[5]*/)
{
    const float hx = b[3] / 2, hy = b[4] / 2;
    const float x1 = b[0] - hx, y1 = b[1] - hy, x2 = b[0] + hx, y2 = b[1] + hy;
    const pt2 ctr = { b[0], b[1] };
    const float cs = cosf(b[6]), sn = sinf(b[6]);
    const pt2 raw[4] = { { x1, y1 }, { x2, y1 }, { x2, y2 }, { x1, y2 } };
    for (int k = 0; k < 4; ++k) c[k] = rotate_about(ctr, cs, sn, raw[k]);
    c[4] = c[0];
}

float hvpr_oracle_box_overlap(const float *a, const float *b)
{
    pt2 ca[5], cb[5], v[24];
    box_corners(a, ca);
    box_corners(b, cb);
    int cnt = 0;
    pt2 ctr = { 0.0f, 0.0f };
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_intersect(ca[i + 1], ca[i], cb[j + 1], cb[j], &v[cnt])) {
                ctr.x += v[cnt].x; ctr.y += v[cnt].y; ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(a, cb[k])) { ctr.x += cb[k].x; ctr.y += cb[k].y; v[cnt++] = cb[k]; }
        if (in_box2d(b, ca[k])) { ctr.x += ca[k].x; ctr.y += ca[k].y; v[cnt++] = ca[k]; }
    }
    if (cnt == 0) return 0.0f;
    ctr.x /= (float)cnt; ctr.y /= (float)cnt;
    /* bubble sort by polar angle about the centroid, ascending */
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i) {
            const float ai = atan2f(v[i].y - ctr.y, v[i].x - ctr.x);
            const float an = atan2f(v[i + 1].y - ctr.y, v[i + 1].x - ctr.x);
            if (ai > an) { pt2 t = v[i]; v[i] = v[i + 1]; v[i + 1] = t; }
        }
    float area = 0.0f;
    for (int k = 0; k < cnt - 1; ++k) {
        const pt2 u = { v[k].x - v[0].x, v[k].y - v[0].y };
        const pt2 w = { v[k + 1].x - v[0].x, v[k + 1].y - v[0].y };
        area += u.x * w.y - u.y * w.x;
    }
    return fabsf(area) / 2.0f;
}

float hvpr_oracle_iou_bev(const float *a, const float *b)
{
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    const float so = hvpr_oracle_box_overlap(a, b);
    return so / fmaxf(sa + sb - so, HV_EPS);
}

/* (N,7) x (M,7) -> (N,M) */
void hvpr_oracle_boxes_overlap_bev(const float *a, int n, const float *b, int m, float *out)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) out[(size_t)i * m + j] = hvpr_oracle_box_overlap(a + 7 * i, b + 7 * j);
}

void hvpr_oracle_boxes_iou_bev(const float *a, int n, const float *b, int m, float *out)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) out[(size_t)i * m + j] = hvpr_oracle_iou_bev(a + 7 * i, b + 7 * j);
}

void hvpr_oracle_boxes_iou3d(const float *a, int n, const float *b, int m, float *out)
{
    for (int i = 0; i < n; ++i) {
        const float *A = a + 7 * i;
        const float a_top = A[2] + A[5] / 2, a_bot = A[2] - A[5] / 2;
        const float va = A[3] * A[4] * A[5];
        for (int j = 0; j < m; ++j) {
            const float *B = b + 7 * j;
            const float b_top = B[2] + B[5] / 2, b_bot = B[2] - B[5] / 2;
            const float vb = B[3] * B[4] * B[5];
            const float ob = hvpr_oracle_box_overlap(A, B);
            const float oh = fmaxf(fminf(a_top, b_top) - fmaxf(a_bot, b_bot), 0.0f);
            const float o3 = ob * oh;
            out[(size_t)i * m + j] = o3 / fmaxf(va + vb - o3, 1e-6f);
        }
    }
}

/* boxes already sorted by descending score.  Bit-mask + sequential sweep, as the
 * device kernel + host loop of the absent nms_gpu do.  keep[] receives indices
 * into the sorted order; returns the number kept.                              */
int hvpr_oracle_nms_sorted(const float *boxes, int n, float thresh, int64_t *keep)
{
    const int nb = (n + 63) / 64;
    uint64_t *mask = (uint64_t *)calloc((size_t)n * (nb ? nb : 1), sizeof(uint64_t));
    uint64_t *remv = (uint64_t *)calloc(nb ? nb : 1, sizeof(uint64_t));
    /* Pairs whose circumscribed circles are more than 10 cm apart cannot intersect (the in-box margin widens a box by
     * 1.42 cm at most): their overlap is exactly 0, so `iou > thresh` is false for every thresh >= 0 and the polygon routine
     * need not run.  Same result as the all-pairs form, ~10x fewer clips on a KITTI-like scene (the CPU baseline of bench.py
     * was otherwise dominated by this loop). */
    double *rad = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (int i = 0; i < n; ++i) rad[i] = 0.5 * sqrt((double)boxes[7 * i + 3] * boxes[7 * i + 3] + (double)boxes[7 * i + 4] * boxes[7 * i + 4]);
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const double dx = (double)boxes[7 * i] - boxes[7 * j], dy = (double)boxes[7 * i + 1] - boxes[7 * j + 1];
            const double reach = rad[i] + rad[j] + 0.1;
            if (thresh >= 0.f && dx * dx + dy * dy > reach * reach) continue;
            if (hvpr_oracle_iou_bev(boxes + 7 * i, boxes + 7 * j) > thresh)
                mask[(size_t)i * nb + j / 64] |= 1ULL << (j % 64);
        }
    free(rad);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (!(remv[i / 64] & (1ULL << (i % 64)))) {
            keep[nk++] = i;
            for (int w = i / 64; w < nb; ++w) remv[w] |= mask[(size_t)i * nb + w];
        }
    }
    free(mask); free(remv);
    return nk;
}
