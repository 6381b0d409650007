// Probe 2: conv-like inner loop (ring of LDS operand reads PF taps ahead, single accumulator chain), optional barrier per
// 9 taps, at different occupancies.  (tools/, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PF, int BAR, int NACC>
__global__ void __launch_bounds__(256) k(float *out, int chunks) {
    __shared__ float4 lds[1024];   // 16 KB
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int base = (threadIdx.x & 63) * 2;
    for (int c = 0; c < chunks; ++c) {
        float4 av[PF + 1], bv[PF + 1];
#pragma unroll
        for (int t = 0; t < PF; ++t) { av[t] = lds[(base + t * 8 + c) & 1023]; bv[t] = lds[(base + 1 + t * 8 + c) & 1023]; }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + PF < 9 || PF == 0) {
                const int t = tap + PF < 9 ? tap + PF : tap;
                av[(tap + PF) % (PF + 1)] = lds[(base + t * 8 + c) & 1023];
                bv[(tap + PF) % (PF + 1)] = lds[(base + 1 + t * 8 + c) & 1023];
            }
            __builtin_amdgcn_sched_barrier(0);
            const int cur = tap % (PF + 1);
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur].x, bv[cur].x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur].y, bv[cur].y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur].z, bv[cur].z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur].w, bv[cur].w, acc[a], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int PF, int BAR, int NACC>
void run(int wg_per_cu, int chunks, const char *name) {
    const int blocks = 256 * wg_per_cu;
    float *out;
    (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<PF, BAR, NACC><<<blocks, 256>>>(out, chunks);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<PF, BAR, NACC><<<blocks, 256>>>(out, chunks);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * chunks * 9 * NACC * 4 * 4096.0;
    printf("%-34s wg/cu %d: %8.1f us  %6.1f TF/s\n", name, wg_per_cu, ms * 1e3, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 3, 6}) {
        run<0, 0, 1>(w, 2000 / w, "PF0 nobar 1acc");
        run<2, 0, 1>(w, 2000 / w, "PF2 nobar 1acc");
        run<2, 1, 1>(w, 2000 / w, "PF2 bar   1acc");
        run<2, 1, 2>(w, 1000 / w, "PF2 bar   2acc");
        run<2, 1, 4>(w, 500 / w, "PF2 bar   4acc");
    }
    return 0;
}
