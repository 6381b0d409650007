// Probe: issue rate of v_mfma_f32_32x32x16_bf16 from one wave per SIMD (256 threads / workgroup, one workgroup per CU),
// with 1, 2 or 4 independent accumulators, operands in registers.  Prints shader cycles per MFMA and the chip-level rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(256) k_probe(float *out, long long *cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(const char *name) {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_probe<NACC><<<256, 256>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_probe<NACC><<<256, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NACC;
    printf("%s: %.1f cycles/MFMA (wave), %.2f ms, chip %.0f TFLOP/s\n", name, c / n, ms, 256.0 * 4 * n * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e12);
}

int main() {
    run<1>("1 accumulator (dependent chain)");
    run<2>("2 accumulators");
    run<4>("4 accumulators");
    return 0;
}
