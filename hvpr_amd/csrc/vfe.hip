// a2 — fused per-pillar PointNet VFE (eval mode, BatchNorm folded into weights/bias by the caller).
// Replaces the ~10 PyTorch ops of PillarVFE_Scale.forward (pcdet/models/backbones_3d/vfe/pillar_vfe.py:184-221)
// and the two PFNLayer.forward calls (:29-49) with one launch.
//
// Mapping (wave64): one wave works on two pillars at a time.
//   phase A  lane = slot (32 lanes per pillar): load the point, pillar mean by a 32-lane butterfly,
//            10-d decoration, masked; layer 0 (10->16) in registers; 32-lane max.
//   phase B  lane = output channel (64 lanes = 64 channels), one pillar after the other: layer 1 over the
//            DISTINCT slots only — the n valid points plus, when n < 32, ONE virtual zero-input slot
//            (pillar_vfe.py masks the input, not the output, so every padded slot yields the same
//            ReLU(folded bias) and takes part in both maxes).  The x_max half of
//            the concat is a per-pillar constant and is folded into the bias once.
//   scale    5 -> 16 -> 32 on lanes 0..15 / 0..31.
//
// k_vfe<true> is the fused encode form (hvpr_encode_fwd_f32): the wave first does the voxelizer's K4 for its two pillars
// — the max_points smallest point indices of the voxel's arena segment, ascending, by a 32-lane bitonic network (a
// 64-lane one with chunked merging for the ~1 % of voxels with more than 32 points) — and reads the points straight
// from the point array, so the padded `voxels` tensor is an optional output instead of an intermediate; it also writes
// the pillar / scale cells of the pre-cleared NHWC canvases (no scatter pass).
#include "common.h"
#include "internal.h"

namespace {

constexpr int C0 = 16;    // layer-0 outputs (NUM_FILTERS[0] / 2)
constexpr int C1 = 64;    // layer-1 outputs
constexpr int CIN = 10;   // x y z r + cluster(3) + center(3)
constexpr int CS0 = 16, CS1 = 32;

__device__ __forceinline__ float rlf(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

struct GatherSrc {
    const float *pts;
    int stride, xyz_col, batch, nx, ny, nz, max_voxels, cap_mode, capacity;
    VoxWs w;
    const int *voxel_offsets;
    float *voxels_out;      // optional
    int *coords_out, *num_out;
    float *spatial;         // NHWC canvas, 128 channels per cell: pillar features in [0, 64), memory read-out in [64, 128)
    int spatial_channels;
    float *spatial_scale;   // NHWC canvas, 32 channels per cell
    int work_blocks;        // workgroups >= work_blocks clear the canvases (see canvas_clear)
};

// The dense canvases are cleared by extra workgroups of this latency-bound launch (47 MB at hvpr_car, hidden under the
// VFE waves' dependent loads).  No race with the cells the VFE waves and the read-out write: a cell whose voxel is emitted
// is skipped — the voxelizer's cell maps still say which cells are occupied (K3 leaves cell_first alone on this path) and
// this pass returns them to idle while it is there.
__device__ __forceinline__ void canvas_clear(const GatherSrc &g, int blk, int nblk) {
    constexpr int CELLS = 8, V = 32, VS = 8;   // cells per wave step, float4 per cell of the main / scale canvas
    const int lane = threadIdx.x & 63;
    const long long n_cells = (long long)g.batch * g.nx * g.ny;
    const long long wave = ((long long)blk * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)nblk * blockDim.x) >> 6;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long c0 = wave * CELLS; c0 < n_cells; c0 += n_waves * CELLS) {
        int emitted = 0;
        if (lane < CELLS && c0 + lane < n_cells) {
            const long long c = c0 + lane;
            if (g.w.cell_first[c] != kIdle) {
                g.w.cell_first[c] = kIdle;
                const int b = (int)(c / ((long long)g.nx * g.ny));
                const int local = g.w.cell_vid[c] - g.w.frame_base[b];
                emitted = local < g.max_voxels && g.voxel_offsets[b] + local < g.capacity;
            }
        }
        const unsigned skip = (unsigned)__ballot(emitted != 0);   // bit i: cell c0 + i belongs to a pillar
#pragma unroll
        for (int i = lane; i < CELLS * V; i += 64) {
            const int cell = i / V;
            if (c0 + cell < n_cells && !((skip >> cell) & 1u))
                reinterpret_cast<float4 *>(g.spatial)[(c0 + cell) * V + i % V] = zero;
        }
        {
            const int cell = lane / VS;
            if (c0 + cell < n_cells && !((skip >> cell) & 1u))
                reinterpret_cast<float4 *>(g.spatial_scale)[(c0 + cell) * VS + lane % VS] = zero;
        }
    }
}

__device__ __forceinline__ int bitonic_asc(int v, int lane, int width) {
    for (int k = 2; k <= width; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = __shfl_xor(v, j, 64);
            const bool up = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            v = (lower == up) ? min(v, o) : max(v, o);
        }
    }
    return v;
}

template <bool GATHER>
__global__ void __launch_bounds__(256) k_vfe(const float4 *__restrict__ voxels, const int *__restrict__ num_points,
                                             const int4 *__restrict__ coords, int M, int P,
                                             const int *__restrict__ m_device, float vsx, float vsy, float vsz,
                                             float offx, float offy, float offz, const float *__restrict__ w0,
                                             const float *__restrict__ b0, const float *__restrict__ w1,
                                             const float *__restrict__ b1, const float *__restrict__ ws0,
                                             const float *__restrict__ bs0, const float *__restrict__ ws1,
                                             const float *__restrict__ bs1, float *__restrict__ pillar_features,
                                             float *__restrict__ scale_features, float *__restrict__ pillar_mask,
                                             GatherSrc g) {
    if (GATHER && (int)blockIdx.x >= g.work_blocks) {
        canvas_clear(g, blockIdx.x - g.work_blocks, gridDim.x - g.work_blocks);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, slot = lane & 31;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = ((GATHER ? g.work_blocks : (int)gridDim.x) * blockDim.x) >> 6;
    if (m_device) M = min(M, *m_device);
    if (wave * 2 >= M) return;   // the grid is sized for the capacity, the live count is a device word
#ifdef HVPR_EXP_TIMING
    const long long tt0 = __builtin_readcyclecounter();
    long long tt1 = 0, tt2 = 0, tt3 = 0;
#endif

    // lane = channel weights, resident for the whole grid-stride loop
    float w1a[C0], w1b[C0];
#pragma unroll
    for (int j = 0; j < C0; ++j) { w1a[j] = w1[lane * 32 + j]; w1b[j] = w1[lane * 32 + C0 + j]; }
    const float bias1 = b1[lane];
    float wsa[5], wsb[CS0];
#pragma unroll
    for (int j = 0; j < 5; ++j) wsa[j] = ws0[(lane & 15) * 5 + j];
#pragma unroll
    for (int j = 0; j < CS0; ++j) wsb[j] = ws1[(lane & 31) * CS0 + j];
    const float bsa = bs0[lane & 15], bsb = bs1[lane & 31];

    for (int pair = wave; pair * 2 < M; pair += n_waves) {
        const int p = pair * 2 + half;
        const bool pv = p < M;
        int n = 0;
        int4 cd = make_int4(0, 0, 0, 0);
        float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (GATHER) {
            // K4 of the voxelizer for this wave's two pillars (output rows p): rank r in the uncapped order of frame b
            int b = 0, cnt = 0, a0 = 0, cell = 0, cutoff = kIdle;
            if (pv) {
                for (int bb = 1; bb < g.batch; ++bb) if (g.voxel_offsets[bb] <= p) b = bb;
                const int fb = g.w.frame_base[b];
                const int4 rec = g.w.vox_rec[fb + (p - g.voxel_offsets[b])];   // {cell, count, arena offset, first index}
                cell = rec.x; cnt = rec.y; a0 = rec.z;
                if (g.cap_mode == 1) {
                    const int rc = fb + g.max_voxels;
                    if (rc < g.w.frame_base[b + 1]) cutoff = g.w.vox_rec[rc].w;
                }
            }
            int v = slot < cnt ? g.w.arena[a0 + slot] : kIdle;
            v = bitonic_asc(v, slot, 32);                       // both halves at once; exact when cnt <= 32
            const unsigned long long big = __ballot(cnt > 32);
            if (big) {                                          // rare: the 64-lane selection of K4, one pillar at a time
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    if (!((big >> (32 * h)) & 1ull)) continue;
                    const int c = __builtin_amdgcn_readlane(cnt, 32 * h), base = __builtin_amdgcn_readlane(a0, 32 * h);
                    int u = lane < c ? g.w.arena[base + lane] : kIdle;
                    u = bitonic_asc(u, lane, 64);
                    const int chunk = 64 - P;
                    for (int done = 64; done < c; done += chunk) {
                        if (lane >= P) {
                            const int j = done + (lane - P);
                            u = j < c ? g.w.arena[base + j] : kIdle;
                        }
                        u = bitonic_asc(u, lane, 64);
                    }
                    const int moved = __shfl(u, slot, 64);      // lanes [0, 32) -> the slots of half h
                    if (half == h) v = moved;
                }
            }
            const bool live = pv && slot < P && v < cutoff;     // v == kIdle is never < cutoff
            n = __popcll(__ballot(live) & (0xffffffffull << (32 * half)));
            if (live) {
                const float *src = g.pts + (size_t)v * g.stride + g.xyz_col;
                pt = make_float4(src[0], src[1], src[2], src[3]);
            }
            cd = make_int4(b, (cell / (g.nx * g.ny)) % g.nz, (cell / g.nx) % g.ny, cell % g.nx);
            if (pv && p < g.capacity) {
                if (g.voxels_out && slot < P) reinterpret_cast<float4 *>(g.voxels_out)[(size_t)p * P + slot] = pt;
                if (slot == 0) {
                    reinterpret_cast<int4 *>(g.coords_out)[p] = cd;
                    g.num_out[p] = n;
                }
            }
        } else {
            n = pv ? num_points[p] : 0;
            if (pv) cd = coords[p];
            if (pv && slot < P) pt = voxels[(size_t)p * P + slot];
        }
        const bool valid = pv && slot < n && slot < P;
#ifdef HVPR_EXP_TIMING
        asm volatile("" ::"v"(pt.x), "v"(pt.w), "v"(n));
        tt1 = __builtin_readcyclecounter();
#endif
        // ---- phase A: decoration + layer 0, lane = slot -------------------------------------------------
        const float fn = (float)n;
        const float mx = hvpr_reduce_sum<32>(pt.x) / fn;     // padded slots are zero (pillar_vfe.py:187)
        const float my = hvpr_reduce_sum<32>(pt.y) / fn;
        const float mz = hvpr_reduce_sum<32>(pt.z) / fn;
        float f[CIN];
        f[0] = pt.x; f[1] = pt.y; f[2] = pt.z; f[3] = pt.w;
        f[4] = pt.x - mx; f[5] = pt.y - my; f[6] = pt.z - mz;
        f[7] = pt.x - ((float)cd.w * vsx + offx);
        f[8] = pt.y - ((float)cd.z * vsy + offy);
        f[9] = pt.z - ((float)cd.y * vsz + offz);
        if (!valid) {
#pragma unroll
            for (int j = 0; j < CIN; ++j) f[j] = 0.f;
        }
        float y0[C0], xmax[C0];
#pragma unroll
        for (int c = 0; c < C0; ++c) {
            float a = b0[c];
#pragma unroll
            for (int j = 0; j < CIN; ++j) a = fmaf(w0[c * CIN + j], f[j], a);
            y0[c] = fmaxf(a, 0.f);
            xmax[c] = hvpr_reduce_max<32>(slot < P ? y0[c] : -INFINITY);   // all P slots take part, padded ones included
        }
        if (pillar_mask && pv && slot < P) pillar_mask[(size_t)p * P + slot] = valid ? 1.f : 0.f;

#ifdef HVPR_EXP_TIMING
        asm volatile("" ::"v"(xmax[0]), "v"(xmax[15]));
        tt2 = __builtin_readcyclecounter();
#endif
        // ---- phase B: layer 1, lane = channel, one pillar of the pair after the other ------------------
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ph = pair * 2 + h;
            if (ph >= M) break;
            const int src0 = h * 32;
            const int nh = __builtin_amdgcn_readlane(n, src0);
            float cst = bias1;   // bias + W1[:,16:32] . xmax  (per-pillar constant)
#pragma unroll
            for (int j = 0; j < C0; ++j) cst = fmaf(w1b[j], rlf(xmax[j], src0), cst);
            float best = -INFINITY;
            const int n_real = min(nh, P);
            const int n_eval = n_real < P ? n_real + 1 : n_real;   // +1 virtual zero-input slot
            for (int s = 0; s < n_eval; ++s) {
                // lane (src0 + s) holds y0 of slot s; slot n_real (if evaluated) is a padded slot -> ReLU(b0)
                float a = cst;
#pragma unroll
                for (int j = 0; j < C0; ++j)
                    a = fmaf(w1a[j], rlf(y0[j], src0 + s), a);
                best = fmaxf(best, fmaxf(a, 0.f));
            }
            pillar_features[(size_t)ph * C1 + lane] = best;
            size_t cell = 0;
            if (GATHER && g.spatial) {   // the pillar cell of the pre-cleared NHWC canvas (pointpillar_scatter.py:192,207)
                cell = ((size_t)__builtin_amdgcn_readlane(cd.x, src0) * g.ny + __builtin_amdgcn_readlane(cd.z, src0)) * g.nx +
                       __builtin_amdgcn_readlane(cd.w, src0);
                g.spatial[cell * g.spatial_channels + lane] = best;
            }

            // scale stream: [n, |mean|, mean_x, mean_y, mean_z] -> 16 -> 32   (pillar_vfe.py:213-216)
            const float smx = rlf(mx, src0);
            const float smy = rlf(my, src0);
            const float smz = rlf(mz, src0);
            const float nrm = sqrtf(smx * smx + smy * smy + smz * smz);
            float s1 = bsa;
            s1 = fmaf(wsa[0], (float)nh, s1);
            s1 = fmaf(wsa[1], nrm, s1);
            s1 = fmaf(wsa[2], smx, s1);
            s1 = fmaf(wsa[3], smy, s1);
            s1 = fmaf(wsa[4], smz, s1);
            s1 = fmaxf(s1, 0.f);   // lanes 0..15 hold channel (lane & 15)
            float s2 = bsb;
#pragma unroll
            for (int j = 0; j < CS0; ++j)
                s2 = fmaf(wsb[j], rlf(s1, j), s2);
            if (lane < CS1) scale_features[(size_t)ph * CS1 + lane] = fmaxf(s2, 0.f);
            if (GATHER && g.spatial_scale && lane < CS1) g.spatial_scale[cell * CS1 + lane] = fmaxf(s2, 0.f);
        }
#ifdef HVPR_EXP_TIMING
        tt3 = __builtin_readcyclecounter();
        if ((blockIdx.x == 0 || blockIdx.x == 200 || blockIdx.x == 400) && (threadIdx.x == 0 || threadIdx.x == 32))
            printf("vfe blk %d lane %d n %d: load/select %lld, phase A %lld, phase B %lld cycles (entry->end %lld)\n", (int)blockIdx.x,
                   (int)threadIdx.x, n, tt1 - tt0, tt2 - tt1, tt3 - tt2, tt3 - tt0);
#endif
    }
}

}  // namespace

int hvpr_i_vfe_gather(const VoxelizeArgs &a, const VoxWs &w, const int32_t *voxel_offsets, int capacity, const VfeWeights &v,
                      float *voxels, int32_t *coords, int32_t *num_points, float *pillar_features, float *scale_features,
                      float *pillar_mask, float *spatial, int spatial_channels, float *spatial_scale, hipStream_t s) {
    if (a.n_feat != 4 || a.max_points > 32 || a.nz != 1) return HVPR_ERR_UNSUPPORTED;
    if (!spatial || !spatial_scale || spatial_channels != 2 * C1) return HVPR_ERR_INVALID_ARG;
    int blocks = hvpr_cdiv(hvpr_cdiv(capacity, 2), 4);
    if (blocks > 2048) blocks = 2048;
    GatherSrc g{a.points, a.point_stride, a.xyz_col, a.batch, a.nx, a.ny, a.nz, a.max_voxels, a.cap_mode, capacity, w,
                voxel_offsets, voxels, coords, num_points, spatial, spatial_channels, spatial_scale, blocks};
    hipLaunchKernelGGL(k_vfe<true>, dim3(blocks + 768), dim3(256), 0, s, nullptr, nullptr, nullptr, capacity, a.max_points,
                       voxel_offsets + a.batch, v.vs_x, v.vs_y, v.vs_z, v.off_x, v.off_y, v.off_z, v.w0, v.b0, v.w1, v.b1, v.ws0,
                       v.bs0, v.ws1, v.bs1, pillar_features, scale_features, pillar_mask, g);
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}


extern "C" int hvpr_pillar_vfe_fwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coords, int M,
                                       int P, const int32_t *m_device, float vs_x, float vs_y, float vs_z, float off_x,
                                       float off_y, float off_z, const float *w0, const float *b0, const float *w1,
                                       const float *b1, const float *ws0, const float *bs0, const float *ws1,
                                       const float *bs1, float *pillar_features, float *pillar_scale_features,
                                       float *pillar_mask, hvpr_stream_t stream) {
    if (M < 0 || P < 1) return HVPR_ERR_INVALID_ARG;
    if (P > 32) return HVPR_ERR_UNSUPPORTED;
    if (M == 0) return HVPR_OK;
    if (!voxels || !num_points || !coords || !w0 || !b0 || !w1 || !b1 || !ws0 || !bs0 || !ws1 || !bs1 ||
        !pillar_features || !pillar_scale_features)
        return HVPR_ERR_INVALID_ARG;
    int blocks = hvpr_cdiv(hvpr_cdiv(M, 2), 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_vfe<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)voxels, num_points,
                       (const int4 *)coords, M, P, m_device, vs_x, vs_y, vs_z, off_x, off_y, off_z, w0, b0, w1, b1, ws0,
                       bs0, ws1, bs1, pillar_features, pillar_scale_features, pillar_mask, GatherSrc{});
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
