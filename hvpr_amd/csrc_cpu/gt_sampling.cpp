// f4 ("next" row of SURVEY.md §8f) — the two CPU natives the reference's GT-sampling augmentation calls (absent from the
// reference: pcdet/ops/roiaware_pool3d and pcdet/ops/iou3d_nms are not in the snapshot, setup.py:53-70):
//   points_in_boxes_cpu   call sites pcdet/utils/box_utils.py:85 (remove_points_in_boxes3d), kitti_dataset.py:217 (database)
//   boxes_bev_iou_cpu     call sites pcdet/datasets/augmentor/database_sampler.py:184-185 (collision test of sampled boxes)
// Host code for the data-loader workers: plain C++ (g++), C-ABI, no GPU.  Boxes are (x, y, z, dx, dy, dz, heading), centre.
// PARITY UNPINNED (sources absent); semantics from SURVEY.md Appendix B.3 / upstream OpenPCDet: a point is inside when
// |z - cz| <= dz/2 and, in the box frame, |x| < dx/2 and |y| < dy/2.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

// Rotated-rectangle intersection by the published iou3d algorithm, fp32, the same steps as the HIP
// kernel (csrc/iou3d_nms.hip) so that the CPU collision test and the GPU NMS agree: vertices of the intersection polygon =
// proper edge crossings + corners of one rectangle inside the other (with the 1e-2 in-box margin), ordered by polar angle
// about their centroid, shoelace area.  Build with -ffp-contract=off.
constexpr float kEps = 1e-8f, kMargin = 1e-2f;

struct V2 { float x, y; };

inline float turn(V2 a, V2 b, V2 o) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }

struct Rect {
    std::array<V2, 4> c;      // corners, counter-clockwise from (-dx/2, -dy/2)
    float cx, cy, hx, hy, cs_inv, sn_inv;
    explicit Rect(const float *b) : cx(b[0]), cy(b[1]), hx(b[3] / 2), hy(b[4] / 2), cs_inv(std::cos(-b[6])), sn_inv(std::sin(-b[6])) {
        const float cs = std::cos(b[6]), sn = std::sin(b[6]);
        const float px[4] = {cx - hx, cx + hx, cx + hx, cx - hx}, py[4] = {cy - hy, cy - hy, cy + hy, cy + hy};
        for (int k = 0; k < 4; ++k)
            c[k] = {(px[k] - cx) * cs + (py[k] - cy) * (-sn) + cx, (px[k] - cx) * sn + (py[k] - cy) * cs + cy};
    }
    bool contains(V2 p) const {
        const float rx = (p.x - cx) * cs_inv + (p.y - cy) * (-sn_inv), ry = (p.x - cx) * sn_inv + (p.y - cy) * cs_inv;
        return std::fabs(rx) < hx + kMargin && std::fabs(ry) < hy + kMargin;
    }
};

// proper crossing of segments p0-p1 and q0-q1
bool crossing(V2 p1, V2 p0, V2 q1, V2 q0, V2 &out) {
    const bool boxes_touch = std::fmin(p0.x, p1.x) <= std::fmax(q0.x, q1.x) && std::fmin(q0.x, q1.x) <= std::fmax(p0.x, p1.x) &&
                             std::fmin(p0.y, p1.y) <= std::fmax(q0.y, q1.y) && std::fmin(q0.y, q1.y) <= std::fmax(p0.y, p1.y);
    if (!boxes_touch) return false;
    const float s1 = turn(q0, p1, p0), s2 = turn(p1, q1, p0), s3 = turn(p0, q1, q0), s4 = turn(q1, p1, q0);
    if (!(s1 * s2 > 0.0f && s3 * s4 > 0.0f)) return false;
    const float s5 = turn(q1, p1, p0);
    if (std::fabs(s5 - s1) > kEps) {
        out = {(s5 * q0.x - s1 * q1.x) / (s5 - s1), (s5 * q0.y - s1 * q1.y) / (s5 - s1)};
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        out = {(b0 * c1 - b1 * c0) / D, (a1 * c0 - a0 * c1) / D};
    }
    return true;
}

float intersection_area(const float *a, const float *b) {
    const Rect A(a), B(b);
    std::vector<V2> v;
    v.reserve(24);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            V2 x;
            if (crossing(A.c[(i + 1) & 3], A.c[i], B.c[(j + 1) & 3], B.c[j], x)) v.push_back(x);
        }
    for (int k = 0; k < 4; ++k) {
        if (A.contains(B.c[k])) v.push_back(B.c[k]);
        if (B.contains(A.c[k])) v.push_back(A.c[k]);
    }
    if (v.empty()) return 0.0f;
    V2 ctr{0.0f, 0.0f};
    for (const V2 &p : v) { ctr.x += p.x; ctr.y += p.y; }
    ctr.x /= (float)v.size(); ctr.y /= (float)v.size();
    std::stable_sort(v.begin(), v.end(), [&](const V2 &p, const V2 &q) {
        return std::atan2(p.y - ctr.y, p.x - ctr.x) < std::atan2(q.y - ctr.y, q.x - ctr.x);
    });
    float area = 0.0f;
    for (size_t k = 0; k + 1 < v.size(); ++k)
        area += (v[k].x - v[0].x) * (v[k + 1].y - v[0].y) - (v[k].y - v[0].y) * (v[k + 1].x - v[0].x);
    return std::fabs(area) / 2.0f;
}

}  // namespace

extern "C" {

// out[m * n_points + i] = 1 when point i lies in box m.  points (n_points, point_stride) with x,y,z first; boxes (n_boxes, 7).
int hvpr_points_in_boxes_cpu(const float *points, int n_points, int point_stride, const float *boxes, int n_boxes, int32_t *out) {
    if (n_points < 0 || n_boxes < 0 || point_stride < 3 || (n_points > 0 && !points) || (n_boxes > 0 && !boxes) ||
        ((long long)n_points * n_boxes > 0 && !out))
        return -1;
    for (int m = 0; m < n_boxes; ++m) {
        const float *b = boxes + (size_t)m * 7;
        const float cs = std::cos(-b[6]), sn = std::sin(-b[6]);
        for (int i = 0; i < n_points; ++i) {
            const float *p = points + (size_t)i * point_stride;
            int in = 0;
            if (std::fabs(p[2] - b[2]) <= b[5] / 2.0f) {
                const float sx = p[0] - b[0], sy = p[1] - b[1];
                const float lx = sx * cs - sy * sn, ly = sx * sn + sy * cs;
                in = (std::fabs(lx) < b[3] / 2.0f) && (std::fabs(ly) < b[4] / 2.0f);
            }
            out[(size_t)m * n_points + i] = in;
        }
    }
    return 0;
}

// out[i * m + j] = rotated BEV IoU of box_a i and box_b j
int hvpr_boxes_bev_iou_cpu(const float *boxes_a, int n, const float *boxes_b, int m, float *out) {
    if (n < 0 || m < 0 || (n > 0 && !boxes_a) || (m > 0 && !boxes_b) || ((long long)n * m > 0 && !out)) return -1;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            const float *a = boxes_a + (size_t)i * 7, *b = boxes_b + (size_t)j * 7;
            const float inter = intersection_area(a, b);
            out[(size_t)i * m + j] = inter / std::fmax(a[3] * a[4] + b[3] * b[4] - inter, kEps);
        }
    return 0;
}

}  // extern "C"
