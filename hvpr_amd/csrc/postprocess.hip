// a5 (gate) / a7 (anchors + box decode) / a8 (score filter + top-k) — the small kernels around the conv stack.
//   k_spatial_gate   SpatialAttention gate, pcdet/models/backbones_2d/spatial_attention.py:47-62
//   k_head_decode    AnchorHeadTemplate.generate_predicted_boxes, dense_heads/anchor_head_template.py:293-340,
//                    ResidualCoder.decode_torch utils/box_coder_utils.py:45-77, limit_period utils/common_utils.py:20-23,
//                    + the sigmoid / class max of Detector3DTemplate.post_processing (detector3d_template.py:206-207,241-246)
//   k_score_compact / k_topk_select   score >= thresh mask and torch.topk of model_nms_utils.py:8-16, with the
//                    build's deterministic order: descending score, ascending anchor id on ties.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ gate
// gate[n,y,x] = sigmoid(s * (conv3x3_{2->1}(cat[max_c, mean_c])(y,x) + b) + t); zero padding of the pooled maps.
constexpr int GT = 16;   // tile edge
__global__ void __launch_bounds__(256) k_spatial_gate(const float *__restrict__ y, int N, int H, int W, int C,
                                                      const float *__restrict__ w18, float conv_bias, float bn_scale,
                                                      float bn_shift, float *__restrict__ gate) {
    __shared__ float s_max[(GT + 2) * (GT + 2)], s_mean[(GT + 2) * (GT + 2)];
    const int tx = blockIdx.x, ty = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x;
    const int grp = tid >> 3, sub = tid & 7;   // 8 lanes per pixel, float4 each
    for (int p = grp; p < (GT + 2) * (GT + 2); p += 32) {
        const int py = p / (GT + 2), px = p % (GT + 2);
        const int iy = ty * GT + py - 1, ix = tx * GT + px - 1;
        float mx = -INFINITY, sm = 0.f;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
        if (in) {
            const float4 *src = (const float4 *)(y + (((size_t)n * H + iy) * W + ix) * C);
            for (int c = sub; c < C / 4; c += 8) {
                const float4 v = src[c];
                mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                sm += (v.x + v.y) + (v.z + v.w);
            }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); sm += __shfl_xor(sm, o, 64); }
        if (sub == 0) { s_max[p] = in ? mx : 0.f; s_mean[p] = in ? sm / (float)C : 0.f; }
    }
    __syncthreads();
    const int ly = tid / GT, lx = tid % GT;
    const int oy = ty * GT + ly, ox = tx * GT + lx;
    if (oy < H && ox < W) {
        float a = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int p = (ly + ky) * (GT + 2) + lx + kx;
                a = fmaf(w18[ky * 3 + kx], s_max[p], a);
                a = fmaf(w18[9 + ky * 3 + kx], s_mean[p], a);
            }
        a = (a + conv_bias) * bn_scale + bn_shift;
        gate[((size_t)n * H + oy) * W + ox] = 1.f / (1.f + expf(-a));
    }
}

// ------------------------------------------------------------------------------------------------ head decode
__global__ void __launch_bounds__(256) k_head_decode(const float *__restrict__ head, int N, int H, int W, int CH, int na,
                                                     int nc, int nbins, const float *__restrict__ xs,
                                                     const float *__restrict__ ys, const float *__restrict__ anc,
                                                     float dir_offset, float dir_limit_offset, float period,
                                                     float *__restrict__ cls_out, float *__restrict__ box_out,
                                                     float *__restrict__ score_out, int *__restrict__ label_out) {
    const long long A = (long long)H * W * na;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * A) return;
    const int n = (int)(t / A);
    const long long ai = t % A;
    const int a = (int)(ai % na);
    const long long pix = ai / na;
    const int ix = (int)(pix % W), iy = (int)(pix / W);
    const float *h = head + ((size_t)n * H * W + pix) * CH;
    // class logits + sigmoid max  (post_processing :206-207, :241-246)
    float best = -INFINITY;
    int lab = 0;
    for (int c = 0; c < nc; ++c) {
        const float lg = h[a * nc + c];
        if (cls_out) cls_out[(size_t)t * nc + c] = lg;
        const float sg = 1.f / (1.f + expf(-lg));
        if (sg > best) { best = sg; lab = c; }
    }
    if (score_out) score_out[t] = best;
    if (label_out) label_out[t] = lab + 1;
    // ResidualCoder.decode_torch
    const float *bt = h + na * nc + a * 7;
    const float *an = anc + a * 5;   // z centre, dx, dy, dz, rot
    const float xa = xs[ix], ya = ys[iy], za = an[0], dxa = an[1], dya = an[2], dza = an[3], ra = an[4];
    const float diag = sqrtf(dxa * dxa + dya * dya);
    float *o = box_out + (size_t)t * 7;
    o[0] = bt[0] * diag + xa;
    o[1] = bt[1] * diag + ya;
    o[2] = bt[2] * dza + za;
    o[3] = expf(bt[3]) * dxa;
    o[4] = expf(bt[4]) * dya;
    o[5] = expf(bt[5]) * dza;
    float rg = bt[6] + ra;
    if (nbins > 0) {
        const float *dp = h + na * nc + na * 7 + a * nbins;
        int dl = 0;
        float dm = dp[0];
        for (int b = 1; b < nbins; ++b) if (dp[b] > dm) { dm = dp[b]; dl = b; }
        const float v = rg - dir_offset;
        // limit_period (common_utils.py:20-23) with the reference's roundings: product and difference rounded separately (a fused
        // multiply-subtract differs in the last bit for ~18 % of the headings: fixture G7, tests/test_gpu_eval_fixtures.py)
        {
#pragma clang fp contract(off)
            const float turns = floorf(v / period + dir_limit_offset);
            const float back = turns * period;
            const float rot = v - back;
            rg = (rot + dir_offset) + period * (float)dl;
        }
    }
    o[6] = rg;
}

// ------------------------------------------------------------------------------------------------ score filter + top-k
__device__ __forceinline__ unsigned ord_bits(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unord_bits(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// key = (orderable score bits << 32) | ~id : descending key order = descending score, ascending id
__global__ void __launch_bounds__(256) k_score_compact(const float *__restrict__ scores, int A, float thresh, int use_thresh,
                                                       unsigned long long *__restrict__ keys, int *__restrict__ counts) {
    const int n = blockIdx.y;
    const float *s = scores + (size_t)n * A;
    unsigned long long *k = keys + (size_t)n * A;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A; i += gridDim.x * blockDim.x) {
        const float v = s[i];
        const bool pass = use_thresh ? (v >= thresh) : !(v != v);
        if (pass) {
            const int pos = atomicAdd(&counts[n], 1);   // the compiler folds this into one atomic per wave
            k[pos] = ((unsigned long long)ord_bits(v) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        }
    }
}

// Pre-selection for large inputs: 16-bit histogram of the orderable score bits (sign + exponent + 7 mantissa bits), then
// the bin that holds the pre_max-th largest score; only scores in that bin or above are compacted and sorted.
constexpr int HBINS = 65536;
__global__ void __launch_bounds__(256) k_score_hist(const float *__restrict__ scores, int A, float thresh, int use_thresh,
                                                    unsigned *__restrict__ hist) {
    const int n = blockIdx.y;
    const float *s = scores + (size_t)n * A;
    unsigned *h = hist + (size_t)n * HBINS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A; i += gridDim.x * blockDim.x) {
        const float v = s[i];
        const bool pass = use_thresh ? (v >= thresh) : !(v != v);
        if (pass) atomicAdd(&h[ord_bits(v) >> 16], 1u);
    }
}

__global__ void __launch_bounds__(1024) k_hist_find(unsigned *__restrict__ hist, int pre_max, unsigned *__restrict__ tbin) {
    // The bin that holds the pre_max-th largest score.  The 64 K bins are read as 16 segments of 4096 with COALESCED 16-byte
    // loads (thread t: bins 4t .. 4t+3 of every segment; the former layout — 64 consecutive bins per thread — made every load
    // instruction touch 64 cache lines: 19 us for 256 KB).  Segment totals pick the segment, a suffix sum over the threads of that
    // segment picks the thread, which walks its four bins.
    static_assert(HBINS == 65536, "16 segments of 4096 bins");
    __shared__ float s_seg[16][16];
    __shared__ unsigned s_tot[16];
    __shared__ unsigned s_wave[16];
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    uint4 *h4 = reinterpret_cast<uint4 *>(hist + (size_t)n * HBINS);
    uint4 hv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) hv[q] = h4[q * 1024 + t];
#pragma unroll
    for (int q = 0; q < 16; ++q) {          // counts < 2^24: exact in fp32
        const float s = hvpr_reduce_sum<64>((float)((hv[q].x + hv[q].y) + (hv[q].z + hv[q].w)));
        if (lane == q) s_seg[wid][q] = s;
    }
    __syncthreads();
    if (t < 16) {
        float a = 0.f;
        for (int w = 0; w < 16; ++w) a += s_seg[w][t];
        s_tot[t] = (unsigned)a;
    }
    __syncthreads();
    int Q = -1;
    unsigned above = 0;                      // scores in the segments above Q
    for (int q = 15; q >= 0; --q) {
        if (above + s_tot[q] >= (unsigned)pre_max) { Q = q; break; }
        above += s_tot[q];
    }
    if (Q < 0) {
        if (t == 0) tbin[n] = 0u;            // fewer than pre_max pass: everything is a candidate
    } else {
        uint4 cur = hv[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) if (q == Q) cur = hv[q];
        const unsigned loc = (cur.x + cur.y) + (cur.z + cur.w);
        unsigned suf = loc;                  // inclusive suffix sum inside the wave (towards higher lanes)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_down(suf, o, 64);
            if (lane + o < 64) suf += u;
        }
        if (lane == 0) s_wave[wid] = suf;
        __syncthreads();
        unsigned above_waves = above;
        for (int w = wid + 1; w < 16; ++w) above_waves += s_wave[w];
        const unsigned incl = suf + above_waves;          // scores in bins >= this thread's first bin
        const unsigned excl = incl - loc;
        if (excl < (unsigned)pre_max && incl >= (unsigned)pre_max) {
            const unsigned b4[4] = {cur.x, cur.y, cur.z, cur.w};
            unsigned acc = excl;
            int b = 3;
            for (; b > 0; --b) {
                if (acc + b4[b] >= (unsigned)pre_max) break;
                acc += b4[b];
            }
            tbin[n] = (unsigned)(Q * 4096 + t * 4 + b);
        }
    }
    // idle state: the histogram is all-zero between calls (no memset node per frame)
#pragma unroll
    for (int q = 0; q < 16; ++q) h4[q * 1024 + t] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ void __launch_bounds__(256) k_score_compact_bin(const float *__restrict__ scores, int A, float thresh, int use_thresh,
                                                           const unsigned *__restrict__ tbin, unsigned long long *__restrict__ keys,
                                                           int *__restrict__ counts) {
    const int n = blockIdx.y;
    const float *s = scores + (size_t)n * A;
    unsigned long long *k = keys + (size_t)n * A;
    const unsigned tb = tbin[n];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A; i += gridDim.x * blockDim.x) {
        const float v = s[i];
        const bool pass = (use_thresh ? (v >= thresh) : !(v != v)) && (ord_bits(v) >> 16) >= tb;
        if (pass) {
            const int pos = atomicAdd(&counts[n], 1);
            k[pos] = ((unsigned long long)ord_bits(v) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
        }
    }
}

constexpr int SORTCAP = 8192;
constexpr int TK_THREADS = 1024;

__device__ void lds_bitonic_desc(unsigned long long *s, int n_pow2) {
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_pow2; i += TK_THREADS) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long a = s[i], b = s[p];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { s[i] = b; s[p] = a; }
                }
            }
            __syncthreads();
        }
}

__global__ void __launch_bounds__(TK_THREADS) k_topk_select(const unsigned long long *__restrict__ keys, int A,
                                                            int *__restrict__ counts, int pre_max,
                                                            int *__restrict__ order, float *__restrict__ sorted_scores,
                                                            int *__restrict__ out_counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *s_keys = (unsigned long long *)smem;            // [SORTCAP]
    unsigned *s_hist = (unsigned *)(s_keys + SORTCAP);                   // [2048]
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned s_remaining, s_fill;
    const int n = blockIdx.x;
    const unsigned long long *k = keys + (size_t)n * A;
    const int cnt = min(counts[n], A);
    const int take = min(cnt, pre_max);
    if (cnt <= SORTCAP) {      // ranked by k_rank_count / k_rank_place; only the bookkeeping is left
        if (threadIdx.x == 0) {
            out_counts[n] = take;
            counts[n] = 0;
        }
        return;
    }
    int m = cnt;   // number of keys to sort in LDS
    if (cnt > SORTCAP) {
        // radix select of the take-th largest key (keys are unique), 11 bits per pass from the top
        if (threadIdx.x == 0) { s_prefix = 0ull; s_remaining = (unsigned)take; }
        __syncthreads();
        for (int shift = 53; shift >= -2; shift -= 11) {
            const int sh = shift < 0 ? 0 : shift;
            const int bits = shift < 0 ? 11 + shift : 11;
            const unsigned long long himask = (sh + bits >= 64) ? 0ull : (~0ull << (sh + bits));
            for (int i = threadIdx.x; i < 2048; i += TK_THREADS) s_hist[i] = 0u;
            __syncthreads();
            const unsigned long long pre = s_prefix;
            for (int i = threadIdx.x; i < cnt; i += TK_THREADS) {
                const unsigned long long v = k[i];
                if ((v & himask) == (pre & himask)) atomicAdd(&s_hist[(unsigned)(v >> sh) & ((1u << bits) - 1u)], 1u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned rem = s_remaining;
                int b = (1 << bits) - 1;
                for (; b > 0; --b) {
                    if (s_hist[b] >= rem) break;
                    rem -= s_hist[b];
                }
                s_remaining = rem;
                s_prefix = pre | ((unsigned long long)b << sh);
            }
            __syncthreads();
        }
        const unsigned long long kth = s_prefix;
        if (threadIdx.x == 0) s_fill = 0u;
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += TK_THREADS) {
            const unsigned long long v = k[i];
            if (v >= kth) { const unsigned p = atomicAdd(&s_fill, 1u); if (p < (unsigned)SORTCAP) s_keys[p] = v; }
        }
        __syncthreads();
        m = take;
    } else {
        for (int i = threadIdx.x; i < cnt; i += TK_THREADS) s_keys[i] = k[i];
    }
    int p2 = 1;
    while (p2 < m) p2 <<= 1;
    for (int i = m + threadIdx.x; i < p2; i += TK_THREADS) s_keys[i] = 0ull;
    __syncthreads();
    lds_bitonic_desc(s_keys, p2);
    for (int i = threadIdx.x; i < take; i += TK_THREADS) {
        const unsigned long long v = s_keys[i];
        order[(size_t)n * pre_max + i] = (int)(0xffffffffu - (unsigned)(v & 0xffffffffull));
        if (sorted_scores) sorted_scores[(size_t)n * pre_max + i] = unord_bits((unsigned)(v >> 32));
    }
    if (threadIdx.x == 0) {
        out_counts[n] = take;
        counts[n] = 0;   // idle state: the candidate counter is zero between calls
    }
}

// Ordering <= SORTCAP candidates by counting: rank(key) = number of larger keys (keys are unique: score bits | ~index), spread
// over (key block) x (comparison chunk) workgroups — 512 comparisons per thread instead of a 91-step single-workgroup bitonic
// network (120 us).  rank[] is zero between calls: k_rank_place clears what k_rank_count added.
constexpr int RK_CHUNK = 512;

__global__ void __launch_bounds__(256) k_rank_count(const unsigned long long *__restrict__ keys, int A, const int *__restrict__ counts,
                                                    unsigned *__restrict__ rank) {
    __shared__ unsigned long long s_k[RK_CHUNK];
    const int n = blockIdx.z;
    const int cnt = min(counts[n], A);
    const int i0 = blockIdx.x * 256, j0 = blockIdx.y * RK_CHUNK;
    if (cnt > SORTCAP || i0 >= cnt || j0 >= cnt) return;
    const unsigned long long *k = keys + (size_t)n * A;
    for (int t = threadIdx.x; t < RK_CHUNK; t += 256) s_k[t] = j0 + t < cnt ? k[j0 + t] : 0ull;   // 0 is below every key
    __syncthreads();
    const int i = i0 + threadIdx.x;
    if (i >= cnt) return;
    const unsigned long long me = k[i];
    unsigned c = 0;
#pragma unroll 16
    for (int j = 0; j < RK_CHUNK; ++j) c += s_k[j] > me ? 1u : 0u;
    if (c) atomicAdd(&rank[(size_t)n * SORTCAP + i], c);
}

__global__ void __launch_bounds__(256) k_rank_place(const unsigned long long *__restrict__ keys, int A, const int *__restrict__ counts,
                                                    unsigned *__restrict__ rank, int pre_max, int *__restrict__ order,
                                                    float *__restrict__ sorted_scores) {
    const int n = blockIdx.y;
    const int cnt = min(counts[n], A);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (cnt > SORTCAP || i >= cnt) return;
    const unsigned r = rank[(size_t)n * SORTCAP + i];
    rank[(size_t)n * SORTCAP + i] = 0u;
    if ((int)r < min(cnt, pre_max)) {
        const unsigned long long v = keys[(size_t)n * A + i];
        order[(size_t)n * pre_max + r] = (int)(0xffffffffu - (unsigned)(v & 0xffffffffull));
        if (sorted_scores) sorted_scores[(size_t)n * pre_max + r] = unord_bits((unsigned)(v >> 32));
    }
}

// The tail of post_processing (detector3d_template.py:255-259): selected boxes / scores / labels of one frame in one launch
// (five small torch gathers otherwise).  Rows past *keep_count read keep[] as it is (the caller zero-fills it: anchor 0).
__global__ void __launch_bounds__(256) k_gather_predictions(const float *__restrict__ boxes, int box_stride, const float *__restrict__ scores,
                                                            const int *__restrict__ labels, const int *__restrict__ keep, int max_keep,
                                                            float *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                            long long *__restrict__ out_labels, long long *__restrict__ out_selected) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= max_keep * 8) return;
    const int r = t >> 3, c = t & 7;
    const int src = keep[r];
    if (c < 7) out_boxes[(size_t)r * 7 + c] = boxes[(size_t)src * box_stride + c];
    else {
        out_scores[r] = scores[src];
        out_labels[r] = (long long)labels[src];
        out_selected[r] = (long long)src;
    }
}

}  // namespace

extern "C" int hvpr_gather_predictions_f32(const float *boxes, int box_stride, const float *scores, const int32_t *labels,
                                           const int32_t *keep, int max_keep, float *out_boxes, float *out_scores,
                                           int64_t *out_labels, int64_t *out_selected, hvpr_stream_t stream) {
    if (max_keep < 0 || box_stride < 7) return HVPR_ERR_INVALID_ARG;
    if (max_keep == 0) return HVPR_OK;
    if (!boxes || !scores || !labels || !keep || !out_boxes || !out_scores || !out_labels || !out_selected) return HVPR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_gather_predictions, dim3(hvpr_cdiv(max_keep * 8, 256)), dim3(256), 0, (hipStream_t)stream, boxes, box_stride,
                       scores, labels, keep, max_keep, out_boxes, out_scores, (long long *)out_labels, (long long *)out_selected);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_spatial_gate_f32(const float *y, int N, int H, int W, int C, const float *w18, float conv_bias,
                                     float bn_scale, float bn_shift, float *gate, hvpr_stream_t stream) {
    if (!y || !w18 || !gate || N < 1 || H < 1 || W < 1 || C < 4) return HVPR_ERR_INVALID_ARG;
    if (C % 4 != 0) return HVPR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_spatial_gate, dim3(hvpr_cdiv(W, GT), hvpr_cdiv(H, GT), N), dim3(256), 0, (hipStream_t)stream, y, N,
                       H, W, C, w18, conv_bias, bn_scale, bn_shift, gate);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" int hvpr_head_decode_f32(const float *head, int N, int H, int W, int head_channels, int n_anchor, int n_class,
                                    int n_dir_bins, const float *x_shifts, const float *y_shifts, const float *anchor_table,
                                    float dir_offset, float dir_limit_offset, float period, float *batch_cls_preds,
                                    float *batch_box_preds, float *scores, int32_t *labels, hvpr_stream_t stream) {
    if (!head || !x_shifts || !y_shifts || !anchor_table || !batch_box_preds || N < 1 || H < 1 || W < 1 || n_anchor < 1 ||
        n_class < 1 || n_dir_bins < 0)
        return HVPR_ERR_INVALID_ARG;
    if (head_channels != n_anchor * (n_class + 7 + n_dir_bins)) return HVPR_ERR_INVALID_ARG;
    const long long total = (long long)N * H * W * n_anchor;
    hipLaunchKernelGGL(k_head_decode, dim3(hvpr_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, head, N, H, W,
                       head_channels, n_anchor, n_class, n_dir_bins, x_shifts, y_shifts, anchor_table, dir_offset,
                       dir_limit_offset, period, batch_cls_preds, batch_box_preds, scores, labels);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}

extern "C" size_t hvpr_score_topk_workspace_bytes(int batch, int n_scores) {
    if (batch < 1 || n_scores < 1) return 0;
    return (size_t)batch * n_scores * sizeof(unsigned long long) + 256 + (size_t)batch * (2 * sizeof(int) + 256) +
           (size_t)batch * HBINS * sizeof(unsigned) + 256 + (size_t)batch * SORTCAP * sizeof(unsigned);
}

extern "C" int hvpr_score_topk_f32(const float *scores, int batch, int n_scores, float score_thresh, int use_thresh,
                                   int pre_max, int32_t *order, float *sorted_scores, int32_t *counts, void *workspace,
                                   size_t workspace_bytes, hvpr_stream_t stream) {
    if (!scores || !order || !counts || !workspace || batch < 1 || n_scores < 1 || pre_max < 1) return HVPR_ERR_INVALID_ARG;
    if (pre_max > SORTCAP) return HVPR_ERR_UNSUPPORTED;
    if (workspace_bytes < hvpr_score_topk_workspace_bytes(batch, n_scores)) return HVPR_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *keys = (unsigned long long *)workspace;
    char *tail = (char *)workspace + (((size_t)batch * n_scores * sizeof(unsigned long long) + 255) / 256) * 256;
    int *cnt = (int *)tail;                                              // [batch]
    unsigned *tbin = (unsigned *)(tail + ((batch * sizeof(int) + 255) / 256) * 256);   // [batch]
    unsigned *hist = (unsigned *)((char *)tbin + ((batch * sizeof(unsigned) + 255) / 256) * 256);   // [batch][HBINS], adjacent to cnt/tbin
    unsigned *rank = (unsigned *)((char *)hist + (((size_t)batch * HBINS * sizeof(unsigned) + 255) / 256) * 256);   // [batch][SORTCAP]
    int bx = hvpr_cdiv(n_scores, 256 * 4);
    if (bx > 1024) bx = 1024;
    // cnt and hist are zero on entry (workspace contract) and are returned to zero by k_hist_find / k_topk_select
    if (n_scores > SORTCAP) {
        hipLaunchKernelGGL(k_score_hist, dim3(bx, batch), dim3(256), 0, s, scores, n_scores, score_thresh, use_thresh, hist);
        hipLaunchKernelGGL(k_hist_find, dim3(batch), dim3(1024), 0, s, hist, pre_max, tbin);
        hipLaunchKernelGGL(k_score_compact_bin, dim3(bx, batch), dim3(256), 0, s, scores, n_scores, score_thresh, use_thresh, tbin, keys, cnt);
    } else {
        hipLaunchKernelGGL(k_score_compact, dim3(bx, batch), dim3(256), 0, s, scores, n_scores, score_thresh, use_thresh, keys, cnt);
    }
    const size_t lds = (size_t)SORTCAP * 8 + 2048 * 4;
    static unsigned long long lds_set = 0ull;   // per device
    if (hvpr_ensure_dyn_lds((const void *)k_topk_select, (int)lds, &lds_set) != 0) return HVPR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_rank_count, dim3(SORTCAP / 256, SORTCAP / RK_CHUNK, batch), dim3(256), 0, s, keys, n_scores, cnt, rank);
    hipLaunchKernelGGL(k_rank_place, dim3(SORTCAP / 256, batch), dim3(256), 0, s, keys, n_scores, cnt, rank, pre_max, order,
                       sorted_scores);
    hipLaunchKernelGGL(k_topk_select, dim3(batch), dim3(TK_THREADS), lds, s, keys, n_scores, cnt, pre_max, order,
                       sorted_scores, counts);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
