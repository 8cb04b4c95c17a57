// v_sin_f32 takes revolutions and is specified for |u| <= 256.  Does it need the caller's reduction u - rint(u) (exact in fp32)
// inside that range?  Compares sin(u) with sin(u - rint(u)) bit by bit and both, and sin(fract(u)), with sin(2 pi u) in double, u uniform in [-R, R].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <vector>
__global__ void k(const float* x, float* raw, float* red, float* fra, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = x[i];
    raw[i] = __builtin_amdgcn_sinf(u);
    red[i] = __builtin_amdgcn_sinf(u - __builtin_rintf(u));
    fra[i] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(u));
}
int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n), ha(n), hb(n), hc(n);
    float *dx, *da, *db, *dc;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4); (void)hipMalloc(&dc, n * 4);
    for (float R : {0.5f, 4.0f, 48.0f, 250.0f}) {
        for (int i = 0; i < n; ++i) hx[i] = -R + 2.0f * R * (float)rand() / (float)RAND_MAX;
        (void)hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, dc, n);
        (void)hipMemcpy(ha.data(), da, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hb.data(), db, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
        double ea = 0, eb = 0, ec = 0; long differ = 0;
        for (int i = 0; i < n; ++i) {
            const double t = sin(6.283185307179586476925 * (double)hx[i]);
            ea = fmax(ea, fabs((double)ha[i] - t)); eb = fmax(eb, fabs((double)hb[i] - t)); ec = fmax(ec, fabs((double)hc[i] - t));
            differ += memcmp(&ha[i], &hb[i], 4) != 0;
        }
        printf("|u| <= %6.1f rev: v_sin(u) max abs err %.3e, v_sin(u - rint(u)) %.3e (bit patterns differ from v_sin(u) in %ld of %d), v_sin(fract(u)) %.3e\n", R, ea, eb, differ, n, ec);
    }
    return 0;
}
