// Accuracy of the hardware v_sin_f32 (input in revolutions) after an exact Cody-Waite reduction by 2 pi, against sin in double.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__global__ void k(const float* x, float* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i];
    const float t = __builtin_fmaf(a, 0.15915494309189535f, 12582912.0f);
    const float nn = t - 12582912.0f;
    float r = __builtin_fmaf(-nn, 6.2831854820251465f, a);
    r = __builtin_fmaf(-nn, -1.7484555314695172e-07f, r);
    y[i] = __builtin_amdgcn_sinf(r * 0.15915494309189535f);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n), hy(n);
    for (int i = 0; i < n; ++i) hx[i] = -300.0f + 600.0f * (float)rand() / (float)RAND_MAX;
    float *dx, *dy; hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
    hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
    double maxe = 0, sum2 = 0;
    for (int i = 0; i < n; ++i) { double e = fabs((double)hy[i] - sin((double)hx[i])); if (e > maxe) maxe = e; sum2 += e * e; }
    printf("v_sin_f32 after exact reduction: max abs err %.3e, rms %.3e\n", maxe, sqrt(sum2 / n));
    return 0;
}
