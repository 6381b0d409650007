// Probe: what fp32 MFMA rate and shader clock does this MI355X sustain?  (tools/, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int LDSR>
__global__ void __launch_bounds__(256) k(float *out, long long *cyc, int iters) {
    __shared__ float4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float4 av = make_float4(threadIdx.x, 1.f, 2.f, 3.f), bv = make_float4(1.f, threadIdx.x, 2.f, 3.f);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (LDSR) { av = lds[(threadIdx.x * 2 + it * 64) & 4095]; bv = lds[(threadIdx.x * 2 + 1 + it * 64) & 4095]; }
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[a], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int LDSR>
void run(int blocks, int iters, const char *name) {
    float *out; long long *cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDSR><<<blocks, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDSR><<<blocks, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * NACC * 4 * 4096.0;
    printf("%-28s blocks %5d: %8.1f us  %6.1f TF/s  wave cycles %lld -> %.2f GHz(memtime) cyc/mfma/simd %.1f\n", name, blocks, ms * 1e3,
           flops / ms / 1e9, h[0], h[0] / (ms * 1e6), (double)h[0] / (iters * NACC * 4.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<4, 0>(256, 4000, "4acc noLDS 1wg/cu");
    run<4, 0>(512, 4000, "4acc noLDS 2wg/cu");
    run<1, 0>(256, 16000, "1acc noLDS 1wg/cu");
    run<1, 0>(256 * 3, 16000, "1acc noLDS 3wg/cu");
    run<1, 0>(256 * 6, 8000, "1acc noLDS 6wg/cu");
    run<1, 1>(256, 16000, "1acc LDS 1wg/cu");
    run<1, 1>(256 * 3, 16000, "1acc LDS 3wg/cu");
    run<1, 1>(256 * 6, 8000, "1acc LDS 6wg/cu");
    run<2, 1>(256 * 3, 8000, "2acc LDS 3wg/cu");
    run<4, 1>(256 * 2, 4000, "4acc LDS 2wg/cu");
    return 0;
}
