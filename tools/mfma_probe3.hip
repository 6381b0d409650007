// Probe 3: does a wave running v_mfma_f32_32x32x2_f32 starve the VALU / SALU / LDS of a co-resident wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// blocks with (blockIdx.x & 1) == role 0 run MFMA chains, role 1 run the probe instruction stream
template <int KIND>
__global__ void __launch_bounds__(512) k(float *out, long long *cyc, int iters, int mfma_on) {
    __shared__ float4 lds[256];
    lds[threadIdx.x & 255] = make_float4(1, 2, 3, 4);
    __syncthreads();
    const bool mf = (threadIdx.x >> 8) == 0;   // waves 0-3 multiply, waves 4-7 (same SIMDs) run the probe stream
    float r = threadIdx.x;
    if (mf) {
        if (!mfma_on) return;
        long long m0t = __builtin_readcyclecounter();
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int it = 0; it < iters * 4; ++it) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r, 1.f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r, 2.f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r, 3.f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r, 4.f, acc, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) r += acc[i];
        if (threadIdx.x == 0) cyc[256 + blockIdx.x] = __builtin_readcyclecounter() - m0t;
    } else {
        if (mfma_on == 2) return;   // MFMA waves alone
        long long t0 = __builtin_readcyclecounter();
        if (KIND == 0) {          // dependent VALU chain
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) r = r * 1.0001f + 0.5f;
            }
        } else if (KIND == 1) {   // SALU chain
            int s = __builtin_amdgcn_readfirstlane(iters);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) asm volatile("s_add_i32 %0, %0, 3\n\ts_nop 0" : "+s"(s));
            }
            r += s;
        } else {                  // LDS reads
            float4 a = make_float4(0, 0, 0, 0);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) { float4 v = lds[(threadIdx.x + u * 7 + it) & 255]; a.x += v.x; }
            }
            r += a.x;
        }
        long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 256) cyc[blockIdx.x] = t1 - t0;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KIND>
void run(const char *name) {
    const int blocks = 256, iters = 2000;
    float *out; long long *cyc;
    (void)hipMalloc(&out, blocks * 512 * 4); (void)hipMalloc(&cyc, 2 * blocks * 8);
    for (int on = 0; on < 3; ++on) {
        (void)hipMemset(cyc, 0, 2 * blocks * 8);
        k<KIND><<<blocks, 512>>>(out, cyc, iters, on);
        (void)hipDeviceSynchronize();
        long long h[4], hm[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost); (void)hipMemcpy(hm, cyc + 256, 32, hipMemcpyDeviceToHost);
        printf("%-10s mode %d: probe %8.1f cycles per 16-instr group; mfma wave %6.1f cycles per mfma\n", name, on, (double)h[1] / iters,
               (double)hm[1] / (iters * 16.0));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0>("VALU fma");
    run<1>("SALU add");
    run<2>("LDS b128");
    return 0;
}
