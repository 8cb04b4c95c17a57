// Probe of ds_read_b64_tr_b16 on gfx950: which LDS element lands in which lane / element.
// LDS image: 16-bit element index = its own value; lane L supplies address of 4 contiguous elements.
// Layout under test: rows of 32 elements (64 B); lane L (grp = (L>>4)&1, h = L>>5, q = (L&15)>>2, p = L&3) supplies
// row 8h + q, columns 16 grp + 4p .. +3.   Expect: lane i of a group gets column 16 grp + (L&15) of rows 8h+0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 hv4 __attribute__((__vector_size__(8)));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) __fp16 lds[32 * 32];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (__fp16)(float)i;
    __syncthreads();
    const int L = threadIdx.x, grp = (L >> 4) & 1, h = L >> 5, q = (L & 15) >> 2, p = L & 3;
    const int row = 8 * h + q, col = 16 * grp + 4 * p;
    hv4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hv4*)(lds + row * 32 + col));
    for (int e = 0; e < 4; ++e) out[L * 4 + e] = (float)v[e];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int L = 0; L < 64; ++L) {
        const int grp = (L >> 4) & 1, hh = L >> 5, i = L & 15;
        for (int e = 0; e < 4; ++e) {
            const int expect = (8 * hh + e) * 32 + 16 * grp + i;
            if ((int)h[L * 4 + e] != expect) { if (bad < 8) printf("lane %d e %d got %d expect %d\n", L, e, (int)h[L * 4 + e], expect); ++bad; }
        }
    }
    printf("tr16 probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "as expected", bad);
    return 0;
}
