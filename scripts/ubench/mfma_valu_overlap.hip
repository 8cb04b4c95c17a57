// Microbenchmark: what hides under v_mfma_f32_32x32x2_f32 with ONE wave per SIMD (the field kernel's regime)?
//   chains = 1: one dependent accumulator chain;  2: two accumulators alternating
//   K VALU (independent v_fma_f32) are placed after every MFMA (clump = 1) or K*4 after every 4th MFMA (clump = 4).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int CHAINS, int CLUMP>
__global__ __launch_bounds__(256) void bench(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (CHAINS == 2 && (m & 1))
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            else
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            if ((m + 1) % CLUMP == 0) {
#pragma unroll
                for (int k = 0; k < K * CLUMP; ++k) v[k % 16] = __builtin_fmaf(v[k % 16], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K, int CHAINS, int CLUMP>
void run() {
    float* out;
    unsigned long long* cyc;
    const int blocks = 256, iters = 2000;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 4 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((bench<K, CHAINS, CLUMP>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += h[i];
    printf("chains=%d clump=%d K=%2d VALU/MFMA : %.1f cycles per MFMA\n", CHAINS, CLUMP, K, s / 1024 / iters / 16);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 1, 1>(); run<2, 1, 1>(); run<4, 1, 1>(); run<8, 1, 1>(); run<12, 1, 1>();
    run<0, 2, 1>(); run<2, 2, 1>(); run<4, 2, 1>(); run<8, 2, 1>(); run<12, 2, 1>(); run<16, 2, 1>();
    run<4, 1, 4>(); run<8, 1, 4>(); run<4, 2, 4>(); run<8, 2, 4>();
    return 0;
}
