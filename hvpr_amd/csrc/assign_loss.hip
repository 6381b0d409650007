// a12 + a13 (training) — anchor target assignment and the anchor head's losses as the library's own kernels.
//   a12 replaces AxisAlignedTargetAssigner.assign_targets / assign_targets_single
//       (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-213) with boxes3d_nearest_bev_iou
//       (pcdet/utils/box_utils.py:252-323) and ResidualCoder.encode_torch (pcdet/utils/box_coder_utils.py:13-43)
//   a13 replaces get_cls_layer_loss / get_box_reg_layer_loss / get_mem_loss (pcdet/models/dense_heads/anchor_head_template.py:
//       101-275) over SigmoidFocalClassificationLoss, WeightedSmoothL1Loss, WeightedCrossEntropyLoss (pcdet/utils/loss_utils.py:
//       9-72, 75-136, 181-206)
// The reference takes two arg-maxes through `.cpu().numpy()` per frame and runs ~40 elementwise torch kernels per loss and stream;
// here a frame's assignment is two launches (the per-ground-truth maximum needs every anchor first) and a stream's three losses WITH
// their gradients are one launch + a fixed-order sum.  The IoU arithmetic keeps torch's operation order with contraction off (the
// labels are compared exactly with the reference's: tests/golden G8, G16; hvpr_amd/build.py compiles this file with -ffp-contract=off).
#include "common.h"

namespace {

constexpr int kMaxGt = 256;          // ground-truth rows per frame (hvpr.yaml pads to ~50)
constexpr float kPi = 3.14159265358979323846f;

__device__ __forceinline__ unsigned ord_of(float v) {          // order-preserving map to unsigned; 0 is below every float
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float of_ord(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// box_utils.py:297-308: heading snapped to the nearest axis -> (x1, y1, x2, y2)
__device__ __forceinline__ float4 nearest_bev(const float *b) {
    const float r = b[6];
    const float rot = fabsf(r - floorf(r / kPi + 0.5f) * kPi);             // limit_period(r, 0.5, pi).abs()
    const bool keep = rot < (float)(3.14159265358979323846 / 4.0);
    const float dx = keep ? b[3] : b[4], dy = keep ? b[4] : b[3];
    return make_float4(b[0] - dx / 2.f, b[1] - dy / 2.f, b[0] + dx / 2.f, b[1] + dy / 2.f);
}

// box_utils.py:252-272 (boxes_iou_normal), the area of b handed in
__device__ __forceinline__ float iou_aa(const float4 a, const float area_a, const float4 g, const float area_g) {
    const float xl = fmaxf(a.x, g.x), xr = fminf(a.z, g.z), yl = fmaxf(a.y, g.y), yr = fminf(a.w, g.w);
    const float inter = fmaxf(xr - xl, 0.f) * fmaxf(yr - yl, 0.f);
    return inter / fmaxf(area_a + area_g - inter, 1e-6f);
}

struct GtLds {
    float4 box[kMaxGt];
    float area[kMaxGt];
    int use[kMaxGt];
    int cls[kMaxGt];
};

// the frame's ground truths in LDS: nearest-axis boxes, areas, and `use` = inside the valid range (trailing all-zero rows are
// padding, axis_aligned_target_assigner.py:53-57) and of this anchor set's class (:66-70; python's class_names[c - 1]: class 0
// wraps to the last class)
__device__ __forceinline__ void load_gt(GtLds &s, const float *__restrict__ gt, int b, int G, int class_index, int n_classes, int *s_last) {
    if (threadIdx.x == 0) *s_last = 0;
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const float *p = gt + ((size_t)b * G + g) * 8;
        float sum = 0.f;
        for (int j = 0; j < 8; ++j) sum += fabsf(p[j]);
        if (sum != 0.f) atomicMax(s_last, g);
    }
    __syncthreads();
    const int last = *s_last;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const float *p = gt + ((size_t)b * G + g) * 8;
        const float4 bx = nearest_bev(p);
        const int c = (int)p[7];
        int m = (c - 1) % n_classes;
        if (m < 0) m += n_classes;
        s.box[g] = bx;
        s.area[g] = (bx.z - bx.x) * (bx.w - bx.y);
        s.use[g] = (g <= last && m == class_index) ? 1 : 0;
        s.cls[g] = c;
    }
    __syncthreads();
}

// pass 1: per ground truth the best IoU over all anchors of the set (:153-156)
__global__ void __launch_bounds__(256) k_assign_gt_max(const float *__restrict__ anchors, int A, const float *__restrict__ gt, int G,
                                                       int class_index, int n_classes, unsigned *__restrict__ g2a_ord) {
    __shared__ GtLds s;
    __shared__ int s_last;
    __shared__ float s_wmax[4];
    const int b = blockIdx.y;
    load_gt(s, gt, b, G, class_index, n_classes, &s_last);
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
    float area_a = 0.f;
    if (a < A) {
        ab = nearest_bev(anchors + (size_t)a * 7);
        area_a = (ab.z - ab.x) * (ab.w - ab.y);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int g = 0; g < G; ++g) {
        if (!s.use[g]) continue;                                  // block-uniform
        float v = a < A ? iou_aa(ab, area_a, s.box[g], s.area[g]) : -2.f;
        v = hvpr_reduce_max<64>(v);
        if (lane == 0) s_wmax[wid] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3]));
            atomicMax(&g2a_ord[(size_t)b * G + g], ord_of(m));
        }
        __syncthreads();
    }
}

// pass 2: labels, regression targets, weights (:145-213), written straight into the head's anchor order
__global__ void __launch_bounds__(256) k_assign_labels(const float *__restrict__ anchors, int A, const float *__restrict__ gt, int G,
                                                       int class_index, int n_classes, float matched_thr, float unmatched_thr,
                                                       const unsigned *__restrict__ g2a_ord, int rots, int loc_stride, int loc_offset,
                                                       long long a_total, int *__restrict__ labels, float *__restrict__ targets,
                                                       float *__restrict__ weights, int *__restrict__ pos_count) {
    __shared__ GtLds s;
    __shared__ int s_last;
    __shared__ float s_g2a[kMaxGt];
    __shared__ int s_pos;
    const int b = blockIdx.y;
    load_gt(s, gt, b, G, class_index, n_classes, &s_last);
    if (threadIdx.x == 0) s_pos = 0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        float m = -2.f;
        if (s.use[g]) m = of_ord(g2a_ord[(size_t)b * G + g]);
        s_g2a[g] = m <= 0.f ? -1.f : m;                           // no overlap at all: no forced match (:155-156)
    }
    __syncthreads();
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    int fg = 0;
    if (a < A) {
        const float *an = anchors + (size_t)a * 7;
        const float4 ab = nearest_bev(an);
        const float area_a = (ab.z - ab.x) * (ab.w - ab.y);
        float best = -INFINITY;
        int arg = 0;
        bool force = false;
        for (int g = 0; g < G; ++g) {
            const float v = s.use[g] ? iou_aa(ab, area_a, s.box[g], s.area[g]) : -2.f;    // masked ground truths never match
            if (v > best) { best = v; arg = g; }                                          // first maximum
            force = force || (v == s_g2a[g]);
        }
        const int cls_of = s.cls[arg];
        int lab = -1;
        if (force) lab = cls_of;
        if (best >= matched_thr) lab = cls_of;
        if (best < unmatched_thr) lab = 0;
        if (force) lab = cls_of;                                  // forced matches win over background (:186-190)
        fg = lab > 0 ? 1 : 0;
        const long long idx = (long long)(a / rots) * loc_stride + loc_offset + (a % rots);
        const size_t o = (size_t)b * a_total + idx;
        labels[o] = lab;
        weights[o] = fg ? 1.f : 0.f;
        float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (fg) {                                                 // ResidualCoder.encode_torch, box_coder_utils.py:13-43
            const float *gp = gt + ((size_t)b * G + arg) * 8;
            const float dxa = fmaxf(an[3], 1e-5f), dya = fmaxf(an[4], 1e-5f), dza = fmaxf(an[5], 1e-5f);
            const float dxg = fmaxf(gp[3], 1e-5f), dyg = fmaxf(gp[4], 1e-5f), dzg = fmaxf(gp[5], 1e-5f);
            const float diag = sqrtf(dxa * dxa + dya * dya);
            t[0] = (gp[0] - an[0]) / diag;
            t[1] = (gp[1] - an[1]) / diag;
            t[2] = (gp[2] - an[2]) / dza;
            t[3] = logf(dxg / dxa);
            t[4] = logf(dyg / dya);
            t[5] = logf(dzg / dza);
            t[6] = gp[6] - an[6];
        }
        for (int j = 0; j < 7; ++j) targets[o * 7 + j] = t[j];
    }
    const unsigned long long m = __ballot(fg != 0);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_pos, __popcll(m));
    __syncthreads();
    if (threadIdx.x == 0 && s_pos) atomicAdd(&pos_count[b], s_pos);
}

// ---- a13: one prediction stream's three losses and their gradients -----------------------------------------------------------------
struct LossCfg {
    int num_class, anchors_per_loc, num_dir_bins;
    float alpha, beta, cls_weight, loc_weight, dir_weight, dir_offset;
    float code_w[7];
};

template <int NC>
__global__ void __launch_bounds__(256) k_rpn_losses(const float *__restrict__ cls, const float *__restrict__ box, const float *__restrict__ dir,
                                                    const int *__restrict__ labels, const float *__restrict__ targets,
                                                    const float *__restrict__ anchor_rot, const int *__restrict__ pos_count, int B,
                                                    long long A, LossCfg c, float *__restrict__ g_cls, float *__restrict__ g_box,
                                                    float *__restrict__ g_dir, float *__restrict__ partial) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // (frame, anchor)
    float l_cls = 0.f, l_loc = 0.f, l_dir = 0.f;
    if (i < (long long)B * A) {
        const int b = (int)(i / A);
        const long long a = i - (long long)b * A;
        const int lab = labels[i];
        const bool pos = lab > 0, neg = lab == 0;
        const float pos_norm = fmaxf((float)pos_count[b], 1.f);                // anchor_head_template.py:111-120
        const float inv_b = 1.f / (float)B;
        // focal loss on the sigmoid logits, loss_utils.py:51-72 (gamma = 2)
        const float cls_w = ((neg ? 1.f : 0.f) + (pos ? 1.f : 0.f)) / pos_norm;
        const int tgt = lab >= 0 ? (NC == 1 ? (pos ? 1 : lab) : lab) : 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const float x = cls[i * NC + k];
            const float t = tgt == k + 1 ? 1.f : 0.f;
            const float p = 1.f / (1.f + expf(-x));
            const float aw = t * c.alpha + (1.f - t) * (1.f - c.alpha);
            const float pt = t * (1.f - p) + (1.f - t) * p;
            const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
            l_cls += aw * pt * pt * bce * cls_w;
            const float dpt = (1.f - 2.f * t) * p * (1.f - p);
            g_cls[i * NC + k] = aw * (2.f * pt * dpt * bce + pt * pt * (p - t)) * cls_w * inv_b * c.cls_weight;
        }
        // smooth L1 on the sin-difference-encoded residuals, anchor_head_template.py:153-160,216-220, loss_utils.py:117-136
        const float reg_w = (pos ? 1.f : 0.f) / pos_norm;
        float tg[7], pr[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) { tg[j] = targets[i * 7 + j]; pr[j] = box[i * 7 + j]; }
        const float sa = sinf(pr[6]), ca = cosf(pr[6]), sb = sinf(tg[6]), cb = cosf(tg[6]);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            float pv = pr[j], tv = tg[j], dpv = 1.f;
            if (j == 6) { pv = sa * cb; tv = ca * sb; dpv = ca * cb + sa * sb; }      // d/da [sin a cos b - cos a sin b]
            if (tv != tv) { tv = pv; dpv = 0.f; }                                     // NaN targets are ignored (:117-119)
            const float diff = (pv - tv) * c.code_w[j];
            const float d = fabsf(diff);
            float loss, dl;
            if (c.beta >= 1e-5f) {
                loss = d < c.beta ? 0.5f * d * d / c.beta : d - 0.5f * c.beta;
                dl = d < c.beta ? d / c.beta : 1.f;
            } else {
                loss = d;
                dl = 1.f;
            }
            l_loc += loss * reg_w;
            const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            g_box[i * 7 + j] = dl * sgn * c.code_w[j] * dpv * reg_w * inv_b * c.loc_weight;
        }
        // direction classifier, anchor_head_template.py:162-176,235-249: cross entropy on the bin of the ground-truth heading
        if (dir) {
            const int nb = c.num_dir_bins;
            const float rot_gt = tg[6] + anchor_rot[a];
            const float v = rot_gt - c.dir_offset;
            const float period = 2.f * kPi;
            const float off = v - floorf(v / period + 0.f) * period;
            int bin = (int)floorf(off / (period / (float)nb));
            bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
            float mx = -INFINITY;
            for (int k = 0; k < nb; ++k) mx = fmaxf(mx, dir[i * nb + k]);
            float se = 0.f;
            for (int k = 0; k < nb; ++k) se += expf(dir[i * nb + k] - mx);
            const float lse = mx + logf(se);
            l_dir = (lse - dir[i * nb + bin]) * reg_w;
            for (int k = 0; k < nb; ++k)
                g_dir[i * nb + k] = (expf(dir[i * nb + k] - lse) - (k == bin ? 1.f : 0.f)) * reg_w * inv_b * c.dir_weight;
        }
    }
    // block sums in a fixed order: wave reduction, then the four waves
    __shared__ float s_part[3][4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    l_cls = hvpr_reduce_sum<64>(l_cls);
    l_loc = hvpr_reduce_sum<64>(l_loc);
    l_dir = hvpr_reduce_sum<64>(l_dir);
    if (lane == 0) { s_part[0][wid] = l_cls; s_part[1][wid] = l_loc; s_part[2][wid] = l_dir; }
    __syncthreads();
    if (threadIdx.x < 3)
        partial[(size_t)blockIdx.x * 3 + threadIdx.x] = (s_part[threadIdx.x][0] + s_part[threadIdx.x][1]) + (s_part[threadIdx.x][2] + s_part[threadIdx.x][3]);
}

// the partial sums of all blocks, in block order, in double: the same bits every run
__global__ void __launch_bounds__(256) k_loss_sum(const float *__restrict__ partial, int n_blocks, int B, float w0, float w1, float w2,
                                                  float *__restrict__ out) {
    __shared__ double s[3][256];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n_blocks; i += 256)
        for (int j = 0; j < 3; ++j) acc[j] += (double)partial[(size_t)i * 3 + j];
    for (int j = 0; j < 3; ++j) s[j][threadIdx.x] = acc[j];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int j = 0; j < 3; ++j) s[j][threadIdx.x] += s[j][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(s[0][0] / B) * w0;
        out[1] = (float)(s[1][0] / B) * w1;
        out[2] = (float)(s[2][0] / B) * w2;
    }
}

// MSE(memory, points) / rows * weight and its gradient w.r.t. memory (anchor_head_template.py:262-275: the divisor is the number of
// pillars of the batch, as the reference wrote it; the point features are a constant there: target.detach())
__global__ void __launch_bounds__(256) k_mse_partial(const float *__restrict__ x, const float *__restrict__ y, long long n, float gscale,
                                                     float *__restrict__ gx, float *__restrict__ partial) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = x[i] - y[i];
        acc += d * d;
        gx[i] = 2.f * d * gscale;
    }
    __shared__ float s_w[4];
    acc = hvpr_reduce_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ void __launch_bounds__(256) k_mse_sum(const float *__restrict__ partial, int n_blocks, double scale, float *__restrict__ out) {
    __shared__ double s[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += 256) acc += (double)partial[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(s[0] * scale);
}

}  // namespace

extern "C" size_t hvpr_assign_targets_workspace_bytes(int batch, int n_gt) {
    if (batch < 1 || n_gt < 0) return 0;
    return (((size_t)batch * (n_gt > 0 ? n_gt : 1) * sizeof(unsigned) + 255) / 256) * 256;
}

extern "C" int hvpr_assign_targets_f32(const float *anchors, int n_anchors, const float *gt_boxes, int batch, int n_gt, int class_index,
                                       int n_classes, float matched_thr, float unmatched_thr, int rots, int loc_stride, int loc_offset,
                                       long long anchors_total, int32_t *labels, float *reg_targets, float *reg_weights,
                                       int32_t *pos_count, void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!anchors || !gt_boxes || !labels || !reg_targets || !reg_weights || !pos_count || !workspace) return HVPR_ERR_INVALID_ARG;
    if (n_anchors < 1 || batch < 1 || n_gt < 0 || n_classes < 1 || class_index < 0 || class_index >= n_classes || rots < 1 ||
        n_anchors % rots != 0 || loc_stride < rots || loc_offset < 0 || loc_offset + rots > loc_stride ||
        anchors_total < (long long)(n_anchors / rots) * loc_stride)
        return HVPR_ERR_INVALID_ARG;
    if (n_gt > kMaxGt || batch > 65535) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < hvpr_assign_targets_workspace_bytes(batch, n_gt)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned *g2a = (unsigned *)workspace;
    if (hipMemsetAsync(g2a, 0, (size_t)batch * (n_gt > 0 ? n_gt : 1) * sizeof(unsigned), s) != hipSuccess) return HVPR_ERR_LAUNCH;
    const dim3 grid(hvpr_cdiv(n_anchors, 256), batch);
    if (n_gt > 0)
        hipLaunchKernelGGL(k_assign_gt_max, grid, dim3(256), 0, s, anchors, n_anchors, gt_boxes, n_gt, class_index, n_classes, g2a);
    hipLaunchKernelGGL(k_assign_labels, grid, dim3(256), 0, s, anchors, n_anchors, gt_boxes, n_gt, class_index, n_classes, matched_thr,
                       unmatched_thr, g2a, rots, loc_stride, loc_offset, anchors_total, labels, reg_targets, reg_weights, pos_count);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" size_t hvpr_rpn_losses_workspace_bytes(int batch, long long n_anchors) {
    if (batch < 1 || n_anchors < 1) return 0;
    return (size_t)hvpr_cdiv((long long)batch * n_anchors, 256) * 3 * sizeof(float);
}

extern "C" int hvpr_rpn_losses_f32(const float *cls_preds, const float *box_preds, const float *dir_preds, const int32_t *labels,
                                   const float *reg_targets, const float *anchor_rot, const int32_t *pos_count, int batch,
                                   long long n_anchors, int num_class, int num_dir_bins, float alpha, float gamma, float beta,
                                   const float *code_weights_host, float cls_weight, float loc_weight, float dir_weight, float dir_offset,
                                   float *losses, float *grad_cls, float *grad_box, float *grad_dir, void *workspace,
                                   size_t workspace_bytes, hvpr_stream_t stream) {
    if (!cls_preds || !box_preds || !labels || !reg_targets || !pos_count || !code_weights_host || !losses || !grad_cls || !grad_box ||
        !workspace || batch < 1 || n_anchors < 1)
        return HVPR_ERR_INVALID_ARG;
    if (dir_preds && (!grad_dir || !anchor_rot || num_dir_bins < 1)) return HVPR_ERR_INVALID_ARG;
    if (gamma != 2.0f || num_class < 1 || num_class > 3 || num_dir_bins > 8) return HVPR_ERR_UNSUPPORTED;   // hvpr.yaml: gamma 2, 1 or 3 classes, 2 bins
    if (workspace_bytes < hvpr_rpn_losses_workspace_bytes(batch, n_anchors)) return HVPR_ERR_WORKSPACE;
    LossCfg c;
    c.num_class = num_class; c.anchors_per_loc = 0; c.num_dir_bins = num_dir_bins; c.alpha = alpha; c.beta = beta;
    c.cls_weight = cls_weight; c.loc_weight = loc_weight; c.dir_weight = dir_weight; c.dir_offset = dir_offset;
    for (int j = 0; j < 7; ++j) c.code_w[j] = code_weights_host[j];
    hipStream_t s = (hipStream_t)stream;
    const int blocks = hvpr_cdiv((long long)batch * n_anchors, 256);
    float *partial = (float *)workspace;
#define HVPR_LAUNCH_LOSS(NC)                                                                                                        \
    hipLaunchKernelGGL(k_rpn_losses<NC>, dim3(blocks), dim3(256), 0, s, cls_preds, box_preds, dir_preds, labels, reg_targets, anchor_rot, \
                       pos_count, batch, n_anchors, c, grad_cls, grad_box, grad_dir, partial)
    if (num_class == 1) HVPR_LAUNCH_LOSS(1);
    else if (num_class == 2) HVPR_LAUNCH_LOSS(2);
    else HVPR_LAUNCH_LOSS(3);
#undef HVPR_LAUNCH_LOSS
    hipLaunchKernelGGL(k_loss_sum, dim3(1), dim3(256), 0, s, partial, blocks, batch, cls_weight, loc_weight, dir_preds ? dir_weight : 0.f, losses);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" size_t hvpr_mse_loss_workspace_bytes(void) { return 1024 * sizeof(float); }

extern "C" int hvpr_mse_loss_f32(const float *x, const float *target, long long rows, int cols, float weight, float *loss, float *grad_x,
                                 void *workspace, size_t workspace_bytes, hvpr_stream_t stream) {
    if (!x || !target || !loss || !grad_x || !workspace || rows < 1 || cols < 1) return HVPR_ERR_INVALID_ARG;
    if (workspace_bytes < hvpr_mse_loss_workspace_bytes()) return HVPR_ERR_WORKSPACE;
    const long long n = rows * cols;
    // loss = mean((x - t)^2) / rows * weight
    const double scale = (double)weight / ((double)n * (double)rows);
    int blocks = hvpr_cdiv(n, 256 * 4);
    if (blocks > 1024) blocks = 1024;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mse_partial, dim3(blocks), dim3(256), 0, s, x, target, n, (float)scale, grad_x, (float *)workspace);
    hipLaunchKernelGGL(k_mse_sum, dim3(1), dim3(256), 0, s, (const float *)workspace, blocks, scale, loss);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
