// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 on one wave per SIMD: one dependent chain vs two chains, with K
// independent VALU (v_fma_f32) behind every MFMA.  Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int CHAINS>
__global__ __launch_bounds__(256) void bench(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i * 0.1f); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (CHAINS == 2 && (m & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) v[k % 16] = __builtin_fmaf(v[k % 16], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int K, int CHAINS>
void run() {
    float* out; unsigned long long* cyc;
    const int blocks = 256, iters = 2000;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((bench<K, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    printf("bf16 32x32x16: chains=%d K=%2d VALU/MFMA : %.1f cycles per MFMA\n", CHAINS, K, s / 1024 / iters / 16);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 1>(); run<2, 1>(); run<4, 1>(); run<6, 1>(); run<8, 1>(); run<12, 1>();
    run<0, 2>(); run<4, 2>(); run<6, 2>(); run<8, 2>(); run<12, 2>();
    return 0;
}
