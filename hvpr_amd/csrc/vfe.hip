// a2 — fused per-pillar PointNet VFE (eval mode, BatchNorm folded into weights/bias by the caller).
// Replaces the ~10 PyTorch ops of PillarVFE_Scale.forward (pcdet/models/backbones_3d/vfe/pillar_vfe.py:184-221)
// and the two PFNLayer.forward calls (:29-49) with one launch.
//
// Mapping (wave64): one wave per pillar, both PFN layers on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//   decoration  lane = slot (both 32-lane halves hold the pillar's 32 slots): pillar mean by a DPP reduction, 10-d
//               decoration, masked INPUT (pillar_vfe.py:205-208 masks the input, not the output, so a padded slot yields
//               ReLU(folded bias) and takes part in both maxes; SURVEY.md §8a a2 quirk — it falls out of the matrix form).
//   layer 0     D0^T (16 ch x 32 slots) = W0 (A operand) . F^T (B operand), 5 MFMAs; bias, ReLU; max over the slots.
//   layer 1     D1^T (64 ch x 32 slots) = [W1a | W1b] . [y0 ; max y0], 2 x 16 MFMAs — the layer-0 output is already in the
//               B-operand layout; max over the slots by a transposing DPP reduction, then bias + ReLU (monotone, so they
//               commute with the max).  The cost no longer depends on the point count of the pillar.
//   scale       5 -> 16 -> 32 on lanes 0..15 / 0..31.
//
// k_vfe<true> is the fused encode form (hvpr_encode_fwd_f32): the wave first does the voxelizer's K4 for its pillar — the
// max_points smallest point indices of the voxel's arena segment, ascending, by a 32-lane bitonic network (for the ~1 % of
// voxels with more than 32 points: a ballot radix select of the 32nd smallest index first) — and reads the points straight
// from the point array, so the padded `voxels` tensor is an optional output instead of an intermediate; it also writes
// the pillar / scale cells of the NHWC canvases, whose other cells are cleared by extra workgroups of the same launch.
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
    int work_blocks;        // the LAST work_blocks workgroups of the grid encode pillars, the ones before clear canvas cells
    ClearJob clear;         // ... (internal.h)
    int idx_bits;           // bits of the largest point index
};

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

typedef float f32x16 __attribute__((ext_vector_type(16)));

// one step of a transposing max-reduction: lanes whose bit `LBIT` is clear keep register `a` and receive the partner
// lane's `a`, the others keep `b` and receive the partner's `b`; CTRL is the DPP pattern that reaches lane ^ (1 << LBIT)
template <int CTRL>
__device__ __forceinline__ float halve_dpp(float a, float b, bool bit) {
    const float keep = bit ? b : a, send = bit ? a : b;
    return fmaxf(keep, hvpr_dpp<CTRL>(send));
}
__device__ __forceinline__ float halve_row(float a, float b) {   // partner = lane ^ 16 (v_permlane16_swap)
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <bool GATHER>
__global__ void __launch_bounds__(256, 4) k_vfe(const float4 *__restrict__ voxels, const int *__restrict__ num_points,
                                             const int4 *__restrict__ coords, int M, int P,
                                             const int *__restrict__ m_device, float vsx, float vsy, float vsz,
                                             float offx, float offy, float offz, const float *__restrict__ w0,
                                             const float *__restrict__ b0, const float *__restrict__ w1,
                                             const float *__restrict__ b1, const float *__restrict__ ws0,
                                             const float *__restrict__ bs0, const float *__restrict__ ws1,
                                             const float *__restrict__ bs1, float *__restrict__ pillar_features,
                                             float *__restrict__ scale_features, float *__restrict__ pillar_mask,
                                             GatherSrc g) {
    const int fill_blocks = GATHER ? (int)gridDim.x - g.work_blocks : 0;   // dispatched first: they are the bandwidth work
    if (GATHER && (int)blockIdx.x < fill_blocks) {
#ifdef HVPR_EXP_TIMING
        const long long f0 = __builtin_amdgcn_s_memrealtime();
#endif
        hvpr_canvas_clear(g.clear, blockIdx.x, fill_blocks);
#ifdef HVPR_EXP_TIMING
        if ((blockIdx.x == 0 || blockIdx.x == fill_blocks - 1 || blockIdx.x == fill_blocks / 2) && threadIdx.x == 0)
            printf("vfe-abs fill blk %d: %lld .. %lld (x10 ns)\n", (int)blockIdx.x, f0, (long long)__builtin_amdgcn_s_memrealtime());
#endif
        return;
    }
    __shared__ int s_sel[4][32];
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5, slot = lane & 31;
    const int wave = ((blockIdx.x - fill_blocks) * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = ((GATHER ? g.work_blocks : (int)gridDim.x) * blockDim.x) >> 6;
    // the live pillar count is a device word; the first pillar's voxel record is requested before it is looked at (one
    // frame: rank == output row), so the two loads share a round trip
    int4 rec0 = make_int4(0, 0, 0, 0);
    if (GATHER && g.batch == 1 && wave < g.capacity) rec0 = g.w.vox_rec[wave];
    if (m_device) M = min(M, *m_device);
    // The three weight matrices go through LDS once per workgroup, coalesced (every wave needs them in a gathered
    // per-lane layout: read straight from memory, 4096 waves x ~170 cache-line requests on the same 11 KB were the
    // longest wait of the kernel).  Rows padded to 36 / 20 floats: conflict-free 16-byte reads.
    __shared__ __attribute__((aligned(16))) float s_w1[C1 * 36];
    __shared__ __attribute__((aligned(16))) float s_ws1[CS1 * 20];
    __shared__ __attribute__((aligned(16))) float s_w0[C0 * CIN];
    for (int i = threadIdx.x; i < C1 * 32 / 4; i += 256)
        *(float4 *)&s_w1[(i >> 3) * 36 + (i & 7) * 4] = ((const float4 *)w1)[i];
    if (threadIdx.x < CS1 * CS0 / 4) *(float4 *)&s_ws1[(threadIdx.x >> 2) * 20 + (threadIdx.x & 3) * 4] = ((const float4 *)ws1)[threadIdx.x];
    if (threadIdx.x < C0 * CIN / 4) ((float4 *)s_w0)[threadIdx.x] = ((const float4 *)w0)[threadIdx.x];
    __syncthreads();
    if (wave >= M) return;
#ifdef HVPR_EXP_TIMING
    const long long tt0 = __builtin_readcyclecounter();
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
    long long tt1 = 0, tt2 = 0, tt3 = 0;
#endif

    // ---- weights in matrix-core operand layout, resident for the whole grid-stride loop --------------------------------
    // v_mfma_f32_32x32x2_f32: A lane (i = lane % 32, k = lane / 32), B lane (k = lane / 32, j = lane % 32), D register r of
    // lane (j, hh) = row 8 * (r / 4) + 4 * hh + r % 4, column j.  Both layers are computed TRANSPOSED — rows = output
    // channels (weights are the A operand), columns = the 32 point slots — so that the output of layer 0 already sits in
    // the B-operand layout of layer 1 (the k index of an MFMA is a free permutation when A and B agree): lane (slot, hh)
    // ends layer 0 with channels chm(r) = 8 * (r / 4) + 4 * hh + r % 4, r < 8, and feeds exactly those to layer 1.
    float a0w[CIN / 2], b0h[8];
#pragma unroll
    for (int t = 0; t < CIN / 2; ++t) a0w[t] = slot < C0 ? s_w0[slot * CIN + 2 * t + h] : 0.f;
    {
        const float4 lo = *(const float4 *)(b0 + 4 * h), hi = *(const float4 *)(b0 + 8 + 4 * h);
        b0h[0] = lo.x; b0h[1] = lo.y; b0h[2] = lo.z; b0h[3] = lo.w; b0h[4] = hi.x; b0h[5] = hi.y; b0h[6] = hi.z; b0h[7] = hi.w;
    }
    float aw[2][8];   // layer 1, the half that multiplies the layer-0 output: two 16-byte loads per row
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 w4 = *(const float4 *)&s_w1[(32 * mb + slot) * 36 + 8 * q + 4 * h];
            aw[mb][4 * q] = w4.x; aw[mb][4 * q + 1] = w4.y; aw[mb][4 * q + 2] = w4.z; aw[mb][4 * q + 3] = w4.w;
        }
    // after the transposing reduction lane l holds output channels 32 mb + 16 l4 + 8 l3 + 4 hh + 2 l1 + l0 (both mb, and
    // twice: lanes l and l ^ 4); lane l finishes channel oc = 32 l2 + ...: bias, ReLU and the other half of layer 1 — the
    // x_max part of the concat is the same for every slot, so it is a 16-term dot product per channel, not a matrix product
    const int oc = 32 * ((lane >> 2) & 1) + 16 * ((lane >> 4) & 1) + 8 * ((lane >> 3) & 1) + 4 * h + (lane & 3);
    float wb[C0];
#pragma unroll
    for (int q = 0; q < C0 / 4; ++q) {
        const float4 w4 = *(const float4 *)&s_w1[oc * 36 + C0 + 4 * q];
        wb[4 * q] = w4.x; wb[4 * q + 1] = w4.y; wb[4 * q + 2] = w4.z; wb[4 * q + 3] = w4.w;
    }
    const float b1l = b1[oc];
    float wsa[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) wsa[j] = ws0[(lane & 15) * 5 + j];
    const float bsa = bs0[lane & 15], bsb = bs1[lane & 31];

    for (int p = wave; p < M; p += n_waves) {
        // both 32-lane halves hold the same pillar: lane (slot, hh) has point `slot`
        int n = 0;
        int4 cd = make_int4(0, 0, 0, 0);
        float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (GATHER) {
            // K4 of the voxelizer for output row p: rank in the uncapped order of frame b -> {cell, count, arena, first}
            // (one frame: rank == row, so this load does not wait for anything)
            int b = 0, fb = 0, r = p;
            if (g.batch > 1) {
                for (int bb = 1; bb < g.batch; ++bb) if (g.voxel_offsets[bb] <= p) b = bb;
                fb = g.w.frame_base[b];
                r = fb + (p - g.voxel_offsets[b]);
            }
            const int4 rec = (g.batch == 1 && p == wave) ? rec0 : g.w.vox_rec[r];
            const int cnt = rec.y, a0 = rec.z;
            int cutoff = kIdle;
            if (g.cap_mode == 1) {
                const int rc = fb + g.max_voxels;
                if (rc < g.w.frame_base[b + 1]) cutoff = g.w.vox_rec[rc].w;
            }
            // the P smallest point indices of the arena segment, ascending, in slots [0, P); K3 left each point next to its
            // index, so index and point arrive in the same round trip and the point travels with the sort key
            int v = kIdle;
            float4 apt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (slot < cnt) { v = g.w.arena[a0 + slot]; apt = g.w.arena_pt[a0 + slot]; }
            bool from_arena = true;
            if (cnt > 32) {   // wave-uniform, ~1 % of the pillars: select by index, fetch the points afterwards
                from_arena = false;
                if (cnt <= 512) {
                    // radix select: T = the P-th smallest index (indices are distinct), bit by bit with ballots
                    int vals[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) vals[r] = (r * 64 + lane < cnt) ? g.w.arena[a0 + r * 64 + lane] : kIdle;
                    unsigned T = 0u;
                    for (int bit = g.idx_bits - 1; bit >= 0; --bit) {
                        const unsigned test = T | ((1u << bit) - 1u);
                        int c = 0;
#pragma unroll
                        for (int r = 0; r < 8; ++r) c += __popcll(__ballot((unsigned)vals[r] <= test));
                        if (c < P) T |= 1u << bit;
                    }
                    int base = 0;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const bool sel = (unsigned)vals[r] <= T;
                        const unsigned long long m = __ballot(sel);
                        if (sel) s_sel[threadIdx.x >> 6][base + __popcll(m & ((1ull << lane) - 1ull))] = vals[r];
                        base += __popcll(m);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    v = slot < P ? ((volatile int *)s_sel[threadIdx.x >> 6])[slot] : kIdle;
                    __builtin_amdgcn_wave_barrier();
                } else {
                    // very dense cell: 64-lane bitonic selection with chunked merging (the form K4 uses)
                    int u = lane < cnt ? g.w.arena[a0 + lane] : kIdle;
                    u = bitonic_asc(u, lane, 64);
                    const int chunk = 64 - P;
                    for (int done = 64; done < cnt; done += chunk) {
                        if (lane >= P) {
                            const int j = done + (lane - P);
                            u = j < cnt ? g.w.arena[a0 + j] : kIdle;
                        }
                        u = bitonic_asc(u, lane, 64);
                    }
                    v = __shfl(u, slot, 64);
                }
            }
            if (from_arena) {   // sort (index, source slot) keys, then pull the point from its source lane
                const int c = min(cnt, 32);
                int key = v == kIdle ? kIdle : (v << 5) | slot;
                if (c > 1) key = bitonic_asc(key, slot, c <= 2 ? 2 : c <= 4 ? 4 : c <= 8 ? 8 : c <= 16 ? 16 : 32);
                v = key == kIdle ? kIdle : key >> 5;
                const int src = (lane & 32) | (key & 31);
                apt.x = __shfl(apt.x, src, 64); apt.y = __shfl(apt.y, src, 64);
                apt.z = __shfl(apt.z, src, 64); apt.w = __shfl(apt.w, src, 64);
            } else {
                v = bitonic_asc(v, slot, 32);
            }
            const bool live = slot < P && v < cutoff;     // v == kIdle is never < cutoff
            n = __popcll(__ballot(live) & 0xffffffffull);
            if (live) {
                if (from_arena) {
                    pt = apt;
                } else {
                    const float *src = g.pts + (size_t)v * g.stride + g.xyz_col;
                    pt = make_float4(src[0], src[1], src[2], src[3]);
                }
            }
            const int cell = rec.x;
            cd = make_int4(b, (cell / (g.nx * g.ny)) % g.nz, (cell / g.nx) % g.ny, cell % g.nx);
            if (p < g.capacity && h == 0) {
                if (g.voxels_out && slot < P) reinterpret_cast<float4 *>(g.voxels_out)[(size_t)p * P + slot] = pt;
                if (slot == 0) {
                    reinterpret_cast<int4 *>(g.coords_out)[p] = cd;
                    g.num_out[p] = n;
                }
            }
        } else {
            n = num_points[p];
            cd = coords[p];
            if (slot < P) pt = voxels[(size_t)p * P + slot];
        }
        const bool valid = slot < n && slot < P;
#ifdef HVPR_EXP_TIMING
        asm volatile("" ::"v"(pt.x), "v"(pt.w), "v"(n));
        tt1 = __builtin_readcyclecounter();
#endif
        // ---- decoration (pillar_vfe.py:187-208) ---------------------------------------------------------------------
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
        if (!valid) {                                        // the mask is applied to the INPUT (:205-208): a padded slot
#pragma unroll                                               // still yields ReLU(folded bias) and takes part in both maxes
            for (int j = 0; j < CIN; ++j) f[j] = 0.f;
        }
        if (pillar_mask && slot < P && h == 0) pillar_mask[(size_t)p * P + slot] = valid ? 1.f : 0.f;

        // ---- layer 0: (16 x 10) . (10 x 32 slots) ---------------------------------------------------------------------
        f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < CIN / 2; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0w[t], h ? f[2 * t + 1] : f[2 * t], acc0, 0, 0, 0);
        float y0[8], xm[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            y0[r] = fmaxf(acc0[r] + b0h[r], 0.f);
            xm[r] = hvpr_reduce_max<32>(slot < P ? y0[r] : -INFINITY);   // all P slots take part, padded ones included
        }
#ifdef HVPR_EXP_TIMING
        asm volatile("" ::"v"(xm[0]), "v"(xm[7]));
        tt2 = __builtin_readcyclecounter();
#endif
        // ---- layer 1: (64 x 16) . (y0 x 32 slots) on the matrix cores + the x_max half as a per-channel constant, then the
        // max over the slots, bias, ReLU.  max_s relu(x_s + c) = relu(max_s x_s + c): constant, bias and ReLU move behind
        // the reduction
        float cst = b1l;
#pragma unroll
        for (int c = 0; c < C0; ++c)   // channel c of x_max sits in register 4 (c / 8) + c % 4 of the lanes of half (c / 4) % 2
            cst = fmaf(wb[c], rlf(xm[4 * (c >> 3) + (c & 3)], 32 * ((c >> 2) & 1)), cst);
        float qm[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[mb][t], y0[t], acc, 0, 0, 0);
            float q[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r] = slot < P ? acc[r] : -INFINITY;
            // transposing reduction over the 32 slots: 16 -> 8 -> 4 -> 2 -> 1 registers along lane bits 0, 1, 3, 4 ...
            float q8[8], q4[4], q2[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) q8[i] = halve_dpp<0xB1>(q[2 * i], q[2 * i + 1], lane & 1);            // r bit 0
#pragma unroll
            for (int i = 0; i < 4; ++i) q4[i] = halve_dpp<0x4E>(q8[2 * i], q8[2 * i + 1], lane & 2);          // r bit 1
#pragma unroll
            for (int i = 0; i < 2; ++i) q2[i] = halve_dpp<0x128>(q4[2 * i], q4[2 * i + 1], lane & 8);         // r bit 2, row_ror:8
            qm[mb] = halve_row(q2[0], q2[1]);                                                                 // r bit 3
        }
        float q1 = (lane & 4) ? qm[1] : qm[0];
        {   // ... and a plain step along lane bit 2: the partner holds this lane's channel block in its other register
            const float other = __shfl_xor((lane & 4) ? qm[0] : qm[1], 4, 64);
            q1 = fmaxf(q1, other);
        }
        const float o = fmaxf(q1 + cst, 0.f);
        pillar_features[(size_t)p * C1 + oc] = o;
        size_t cell = 0;
        if (GATHER) {   // the pillar cell of the NHWC canvas (pointpillar_scatter.py:192,207)
            cell = ((size_t)cd.x * g.ny + cd.z) * g.nx + cd.w;
            g.spatial[cell * g.spatial_channels + oc] = o;
        }

        // ---- scale stream: [n, |mean|, mean_x, mean_y, mean_z] -> 16 -> 32   (pillar_vfe.py:213-216) ---------------------
        const float nrm = sqrtf(mx * mx + my * my + mz * mz);
        float s1 = bsa;
        s1 = fmaf(wsa[0], fn, s1);
        s1 = fmaf(wsa[1], nrm, s1);
        s1 = fmaf(wsa[2], mx, s1);
        s1 = fmaf(wsa[3], my, s1);
        s1 = fmaf(wsa[4], mz, s1);
        s1 = fmaxf(s1, 0.f);   // lanes 0..15 hold channel (lane & 15)
        float s2 = bsb;
        {
            // second scale layer: the 16 weights of this lane's channel are read from LDS per pillar instead of living in
            // registers — with them the kernel does not fit 128 VGPRs = four workgroups per CU
            const float4 *wp = (const float4 *)&s_ws1[(lane & 31) * 20];
#pragma unroll
            for (int q = 0; q < CS0 / 4; ++q) {
                const float4 w4 = wp[q];
                s2 = fmaf(w4.x, rlf(s1, 4 * q), s2);
                s2 = fmaf(w4.y, rlf(s1, 4 * q + 1), s2);
                s2 = fmaf(w4.z, rlf(s1, 4 * q + 2), s2);
                s2 = fmaf(w4.w, rlf(s1, 4 * q + 3), s2);
            }
        }
        s2 = fmaxf(s2, 0.f);
        if (lane < CS1) {
            scale_features[(size_t)p * CS1 + lane] = s2;
            if (GATHER) g.spatial_scale[cell * CS1 + lane] = s2;
        }
#ifdef HVPR_EXP_TIMING
        tt3 = __builtin_readcyclecounter();
        if (((blockIdx.x - fill_blocks) % 100 == 0) && threadIdx.x == 0)
            printf("vfe-abs work blk %d: %lld .. %lld (x10 ns)\n", (int)blockIdx.x - fill_blocks, rt0, (long long)__builtin_amdgcn_s_memrealtime());
        if ((tt3 - tt0 > 30000 || blockIdx.x == fill_blocks) && lane == 0)
            printf("vfe blk %d wave %d n %d: load/select %lld, layer 0 %lld, layer 1 %lld cycles (entry->end %lld = %lld ns)\n", (int)blockIdx.x,
                   (int)(threadIdx.x >> 6), n, tt1 - tt0, tt2 - tt1, tt3 - tt2, tt3 - tt0, 10 * ((long long)__builtin_amdgcn_s_memrealtime() - rt0));
#endif
    }
}

}  // namespace

int hvpr_i_vfe_gather(const VoxelizeArgs &a, const VoxWs &w, const int32_t *voxel_offsets, int capacity, const VfeWeights &v,
                      float *voxels, int32_t *coords, int32_t *num_points, float *pillar_features, float *scale_features,
                      float *pillar_mask, float *spatial, int spatial_channels, float *spatial_scale, unsigned char *canvas_state,
                      hipStream_t s) {
    if (a.n_feat != 4 || a.max_points > 32 || a.nz != 1 || a.n_points >= (1 << 26)) return HVPR_ERR_UNSUPPORTED;
    if (!spatial || !spatial_scale || spatial_channels != 2 * C1) return HVPR_ERR_INVALID_ARG;
    // one frame (capacity <= 16 K pillars): 1024 workgroups = four per CU, all resident at once next to the clearing ones,
    // one pillar per wave; batches: up to 8192 workgroups, each wave walks a few pillars
    int blocks = hvpr_cdiv(capacity, 16);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1024) blocks = hvpr_cdiv(capacity, 4) < 1024 ? hvpr_cdiv(capacity, 4) : 1024;
    if (blocks < 1) blocks = 1;
    int idx_bits = 1;
    while (idx_bits < 30 && (1ll << idx_bits) < (long long)a.n_points) ++idx_bits;
    const long long n_cells = (long long)a.batch * a.nx * a.ny;
    const ClearJob cj{w.cell_first, w.cell_vid, w.frame_base, voxel_offsets, a.batch, a.nx, a.ny, a.max_voxels, capacity, spatial,
                      spatial_scale, 0, n_cells, canvas_state};
    GatherSrc g{a.points, a.point_stride, a.xyz_col, a.batch, a.nx, a.ny, a.nz, a.max_voxels, a.cap_mode, capacity, w,
                voxel_offsets, voxels, coords, num_points, spatial, spatial_channels, spatial_scale, blocks, cj, idx_bits};
    // three 64-cell steps per wave: few, long-lived workgroups — they hold slots the pillar workgroups want.  (One or two steps
    // per wave with correspondingly more workgroups, tried for the sparse clear of persistent canvases: 48.8 / 49.0 vs 48.3 us
    // for the group — no difference.)
    long long fill = (n_cells + 767) / 768;
    if (fill > 1024) fill = 1024;
    hipLaunchKernelGGL(k_vfe<true>, dim3(blocks + (int)fill), dim3(256), 0, s, nullptr, nullptr, nullptr, capacity, a.max_points,
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
    int blocks = hvpr_cdiv(M, 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_vfe<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)voxels, num_points,
                       (const int4 *)coords, M, P, m_device, vs_x, vs_y, vs_z, off_x, off_y, off_z, w0, b0, w1, b1, ws0,
                       bs0, ws1, bs1, pillar_features, pillar_scale_features, pillar_mask, GatherSrc{});
    HVPR_CHECK_LAUNCH();
    return HVPR_OK;
}
