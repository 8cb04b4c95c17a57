// Throughput of LDS accumulation forms on gfx950, in the access shape of scatter_patch_kernel: a half-wave adds 32 consecutive
// channels of one voxel of an 8 x 8 x 8 x 32 fp32 box, 8 corners per point.  Build: hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE, int SAME>
__global__ __launch_bounds__(256) void bench(float* out, int iters) {
    extern __shared__ float box[];
    for (int i = threadIdx.x; i < 16384; i += 256) box[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ch = lane & 31, h = lane >> 5;
    uint32_t s = 1234567u + wave * 977u + blockIdx.x * 31u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const uint32_t r = SAME ? (s >> 8) : ((s >> 8) + h * 0x9e37u);
        const int rx = r % 7, ry = (r >> 4) % 7, rz = (r >> 9) % 7;
        const int base = ((rz * 8 + ry) * 8 + rx) * 32 + ch;
        const float g = (float)(it & 7) + 1.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int off = ((k & 1) + ((k & 2) ? 8 : 0) + ((k & 4) ? 64 : 0)) * 32;
            if (MODE == 0) __hip_atomic_fetch_add(box + base + off, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(box) + base + off, (uint32_t)(it + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) { volatile float* p = box + base + off; *p = *p + g; }
        }
    }
    __syncthreads();
    float acc = 0.0f;
    for (int i = threadIdx.x; i < 16384; i += 256) acc += box[i];
    if (acc == 123.456f) out[blockIdx.x] = acc;
}
template <int MODE, int SAME>
void run(const char* name, float* out) {
    const int iters = 2048, blocks = 512;
    hipFuncSetAttribute((const void*)bench<MODE, SAME>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    bench<MODE, SAME><<<blocks, 256, 65536>>>(out, iters);
    hipEventRecord(a);
    bench<MODE, SAME><<<blocks, 256, 65536>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)blocks / 256 * 4 * iters * 8;
    printf("%-34s halves on %s voxel: %8.3f ms  %7.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, SAME ? "the same" : "different", ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}
int main() {
    float* out; hipMalloc(&out, 4096);
    run<0, 0>("ds_add_f32", out); run<0, 1>("ds_add_f32", out);
    run<1, 0>("ds_add_u32", out); run<1, 1>("ds_add_u32", out);
    run<2, 0>("ds_read + add + ds_write (racy)", out); run<2, 1>("ds_read + add + ds_write (racy)", out);
    return 0;
}
