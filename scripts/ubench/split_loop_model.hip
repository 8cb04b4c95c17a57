// Microbenchmark: the inner loop of field_bf6_kernel in isolation, one wave per SIMD on every CU.
// Per k-chunk: 3 x 16-B weight fragment loads (ring AHEAD chunks ahead) from a buffer of `bytes` (L2 resident at 1.2 MB,
// L1 resident at 3 KiB), 6 dependent v_mfma_f32_32x32x16_bf16, K VALU ops (scalar v_fma_f32 or packed v_pk_fma_f32).
// Prints cycles per chunk (ideal: 6 x 32 = 192).   Build: hipcc --offload-arch=gfx950 -O3 split_loop_model.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int PK, int LOAD, int AHEAD, int CHAINS, int NM = 6>
__global__ __launch_bounds__(256) void bench(const bf16x8* __restrict__ w, int frags, float* out, unsigned long long* cyc, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16x8* lw = reinterpret_cast<const bf16x8*>(smem);
    const bf16x8* gw = w;
    const int gfrags = frags;
    if (LOAD >= 3) { w = lw; frags = 32 * 3 * 64; }      // 96 KiB of LDS (uninitialised: timing only)
    const int lane = threadIdx.x & 63;
    f32x16 acc, acc1;
    bf16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)(1.0f + i * 0.1f);
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x + i, 1.0f + i};
    float sum = 0;
    const int chunks_per_pass = frags / (3 * 64);
    unsigned long long t0 = __builtin_readcyclecounter();
    int c0 = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 7) % chunks_per_pass;      // waves start at different places
    int gtile = blockIdx.x % 20;
    for (int tile = 0; tile < tiles; ++tile) {
        if (LOAD == 4) {
            __syncthreads();
            const int wave = threadIdx.x >> 6;
            bf16x8* dst = const_cast<bf16x8*>(lw) + ((tile + 1) & 1) * (16 * 3 * 64);
            const bf16x8* src = gw + (size_t)gtile * (16 * 3 * 64);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int piece = wave + 4 * i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(dst + piece * 64), 16, 0, 0);
            }
            gtile = (gtile + 1) % (gfrags / (16 * 3 * 64));
            c0 = (tile & 1) * 16;
        }
        for (int i = 0; i < 16; ++i) acc[i] = acc1[i] = 0.f;
        bf16x8 ring[AHEAD][3];
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) ring[i][k] = w[(size_t)(((c0 + i) % chunks_per_pass) * 3 + k) * 64 + lane];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            bf16x8 a[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) a[k] = ring[c % AHEAD][k];
            if (LOAD) {
                const int cn = LOAD == 2 ? 0 : (LOAD == 4 ? (c0 + (c + AHEAD) % 16) : (c0 + c + AHEAD) % chunks_per_pass);   // LOAD == 2: always the same 3 KiB (L1 hits)
#pragma unroll
                for (int k = 0; k < 3; ++k) ring[c % AHEAD][k] = w[(size_t)(cn * 3 + k) * 64 + lane];
            }
            if (NM == 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc, 0, 0, 0);
            } else if (CHAINS == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (PK) v[k % 8] = __builtin_elementwise_fma(v[k % 8], f32x2{1.0001f, 0.9999f}, f32x2{0.5f, 0.25f});
                else v[k % 8][0] = __builtin_fmaf(v[k % 8][0], 1.0001f, 0.5f);
            }
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (K + NM - 1) / NM, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LOAD != 4) c0 = (c0 + 16) % chunks_per_pass;
        for (int i = 0; i < 16; ++i) sum += acc[i] + acc1[i];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) sum += v[i][0] + v[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K, int PK, int LOAD, int AHEAD, int CHAINS = 1, int NM = 6>
void run(const bf16x8* w, int frags, const char* what) {
    float* out; unsigned long long* cyc;
    const int blocks = 256, tiles = 400;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    hipFuncSetAttribute((const void*)bench<K, PK, LOAD, AHEAD, CHAINS, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<K, PK, LOAD, AHEAD, CHAINS, NM>), dim3(blocks), dim3(256), LOAD >= 3 ? 98304 : 0, 0, w, frags, out, cyc, tiles);
    hipEventRecord(e0);
    hipLaunchKernelGGL((bench<K, PK, LOAD, AHEAD, CHAINS, NM>), dim3(blocks), dim3(256), LOAD >= 3 ? 98304 : 0, 0, w, frags, out, cyc, tiles);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    const double per_chunk = s / 1024 / tiles / 16;
    printf("%-34s mfma/chunk=%d chains=%d K=%2d %s AHEAD=%d : %.1f counter ticks/chunk, %.3f ms -> %.1f ns/chunk\n", what, NM, CHAINS, K, PK ? "pk " : "f32", AHEAD, per_chunk, ms,
           ms * 1e6 / tiles / 16);
    hipFree(out); hipFree(cyc);
}
int main() {
    const int frags = 400 * 3 * 64;                    // 400 chunks x 3 KiB = 1.2 MB
    bf16x8* w; hipMalloc(&w, (size_t)frags * 16); hipMemset(w, 0, (size_t)frags * 16);
    run<0, 0, 0, 3>(w, frags, "mfma only");
    run<0, 0, 0, 3, 1, 3>(w, frags, "mfma only");
    run<20, 0, 3, 2, 1, 3>(w, frags, "mfma + valu + LDS reads");
    run<16, 0, 3, 2, 1, 3>(w, frags, "mfma + valu + LDS reads");
    run<12, 0, 3, 2, 1, 3>(w, frags, "mfma + valu + LDS reads");
    run<20, 0, 4, 2, 1, 3>(w, frags, "mfma + valu + staged");
    run<20, 0, 1, 3, 1, 3>(w, frags, "mfma + valu + L2 stream");
    run<20, 0, 0, 2, 1, 3>(w, frags, "mfma + valu");
    run<40, 0, 0, 2, 1, 6>(w, frags, "mfma + valu");
    return 0;
}
