// Weight-gradient reduction of the backward pass:  dW[b] += G[b]^T X[b]  and  colsum[b] += sum_n G[b][n]  per image b,
// with G (points, H) the gradient w.r.t. a layer's sine argument and X (points, K) that layer's input, both row-major as
// the field kernels wrote them (field_kernel.hip, STORE / backward).  This is the "tall-skinny" GEMM of SURVEY 2.1 K-bwd:
// M = H, N = K <= 256, reduction over ~1e6 points per image.
//
// Mapping: v_mfma_f32_32x32x2_f32 computes D[i][j] += A[i][k] B[k][j] with k = 2 points per instruction.  A = G^T, B = X,
// so lane (c = lane & 31, p = lane >> 5) needs G[point p][channel 32 it + c] and X[point p][channel 32 jt + c]: for both
// operands the 32 lanes of a half read 32 consecutive floats of one row -- the row-major chunk buffers are consumed as
// they are, no transposition.  A block is 4 waves; wave w of row group y owns output rows [128 y + 32 w, +32) x all K
// columns (NJ column tiles: 128 accumulator registers at K = 256).  Operands are staged per 32 points through LDS by
// LDS-DMA (X tile shared by the four waves), double-buffered, one barrier per stage; the LDS reads of k-step k+1 are
// pinned ahead of the MFMAs of k-step k.  Blocks split the points of an image -- never more blocks than CUs: with 96 KiB
// of LDS each a 257th block would run alone after the others and double the time -- and every block adds its partial
// tile to dW[b] with float atomics whose wave instructions each cover two whole 128-byte rows.
// Measured (3 images x 1,048,576 points, H = K = 256): 4.2 ms = 98 TFLOP/s (62 % of the fp32 MFMA peak), against 6.0 ms for
// torch.bmm + column sum (rocBLAS); first layer (K = 32): 0.98 ms against 2.3 ms.
#include "cnerf_dev.hpp"
#include "cnerf_kernels.hpp"

namespace cnerf {

struct WeightGradArgs {
    const float* G;      // (cnt, npi, H)
    const float* X;      // (cnt, npi, K)
    float* dW;           // (cnt, H, K)   accumulated into
    float* colsum;       // (cnt, H)      accumulated into
    long long npi;
    int cnt, H, K;
    int blocks_per_image;
};

constexpr int WG_P = 32;         // points per stage (16 MFMA k-steps)

// One stage of operands in LDS: X rows [WG_P][K] and the block's 128 columns of the G rows [WG_P][128], both parked by
// LDS-DMA (16 bytes per lane, lane-linear destination, per-lane source so that rows past the end of the image can be
// clamped instead of read).  Two stages are double-buffered: (32 + 16) KB x 2 at K = 256.
template <int NJ>
__global__ __launch_bounds__(256) void weight_grad_kernel(WeightGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 32 * NJ;
    constexpr int X_F4 = WG_P * K / 4;                 // float4 per X stage
    constexpr int G_F4 = WG_P * 128 / 4;               // float4 per G stage
    constexpr int STAGE_F4 = X_F4 + G_F4;
    f32x4* lds4 = reinterpret_cast<f32x4*>(smem);
    const float* ldsf = reinterpret_cast<const float*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int c = lane & 31, p = lane >> 5;
    const int b = blockIdx.x / a.blocks_per_image, part = blockIdx.x - b * a.blocks_per_image;
    const int col0 = 128 * blockIdx.y;                 // first G column (= dW row) of this block
    const bool active = col0 + 32 * wave < a.H;        // narrow networks: surplus waves only help with the copies
    const long long stages = (a.npi + WG_P - 1) / WG_P;
    const long long s_begin = stages * part / a.blocks_per_image, s_end = stages * (part + 1) / a.blocks_per_image;
    const float* Gb = a.G + (size_t)b * a.npi * a.H;
    const float* Xb = a.X + (size_t)b * a.npi * K;

    auto dma_stage = [&](long long s, int buf) {
        f32x4* dst = lds4 + buf * STAGE_F4;
        const long long pt0 = s * WG_P;
        // X: X_F4 / 64 pieces of 1 KiB, piece q = float4 [64 q, 64 q + 64) of the row-major [WG_P][K] tile
#pragma unroll
        for (int i = 0; i < X_F4 / 256; ++i) {
            const int q = wave_u * (X_F4 / 256) + i;
            const int e = q * 64 + lane;
            const int row = e / (K / 4), col4 = e - row * (K / 4);
            long long pt = pt0 + row;
            pt = pt < a.npi ? pt : a.npi - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xb + (size_t)pt * K + 4 * col4),
                                             (__attribute__((address_space(3))) void*)(dst + q * 64), 16, 0, 0);
        }
        // G: 16 pieces, piece q = the 128 columns of points 2q, 2q + 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave_u * 4 + i;
            long long pt = pt0 + 2 * q + p;
            pt = pt < a.npi ? pt : a.npi - 1;
            int ch = col0 + 4 * c;
            ch = ch + 4 <= a.H ? ch : a.H - 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Gb + (size_t)pt * a.H + ch),
                                             (__attribute__((address_space(3))) void*)(dst + X_F4 + q * 64), 16, 0, 0);
        }
    };

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    float cs = 0.0f;

    if (s_begin < s_end) dma_stage(s_begin, 0);
    int cur = 0;
    for (long long s = s_begin; s < s_end; ++s) {
        __syncthreads();                                           // stage s has landed; the other buffer is free
        if (s + 1 < s_end) dma_stage(s + 1, cur ^ 1);
        if (active) {
            const float* xs = ldsf + (size_t)cur * STAGE_F4 * 4 + p * K + c;
            const float* gs = ldsf + ((size_t)cur * STAGE_F4 + X_F4) * 4 + p * 128 + 32 * wave + c;
            const long long pt0 = s * WG_P;
            float ga = gs[0], xb[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) xb[j] = xs[32 * j];
#pragma unroll
            for (int k = 0; k < WG_P / 2; ++k) {
                float g_cur = ga, x_cur[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) x_cur[j] = xb[j];
                if (pt0 + 2 * k + p >= a.npi) g_cur = 0.0f;        // G rows beyond the image contribute nothing
                cs += g_cur;
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < WG_P / 2) {                            // operands of the next k-step fly under this one's MFMAs
                    ga = gs[(2 * (k + 1)) * 128];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) xb[j] = xs[(2 * (k + 1)) * K + 32 * j];
                }
                __builtin_amdgcn_sched_barrier(0);                 // (pinned: the scheduler otherwise waits on every pair of reads)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(g_cur, x_cur[j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        cur ^= 1;
    }
    if (!active) return;
    // D[i][j]: register 4g + e of lane (c, p) is row 8g + 4p + e, column c of the tile
    float* dWb = a.dW + (size_t)b * a.H * K;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = col0 + 32 * wave + 8 * (r >> 2) + 4 * p + (r & 3);
            atomicAdd(dWb + (size_t)row * K + 32 * j + c, acc[j][r]);
        }
    const float tot = cs + __shfl_xor(cs, 32, WAVE);
    if (p == 0) atomicAdd(a.colsum + (size_t)b * a.H + col0 + 32 * wave + c, tot);
}

template <int NJ>
static hipError_t launch_wg(const WeightGradArgs& a, hipStream_t stream) {
    const unsigned row_groups = (unsigned)((a.H + 127) / 128);     // 4 waves x one 32-row tile per block
    const int lds_bytes = 2 * (WG_P * 32 * NJ + WG_P * 128) * 4;
    if (hipError_t e = hipFuncSetAttribute((const void*)weight_grad_kernel<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) return e;
    hipLaunchKernelGGL((weight_grad_kernel<NJ>), dim3((unsigned)(a.cnt * a.blocks_per_image), row_groups), dim3(256), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_weight_grad(int cnt, long long npi, int H, int K, const float* G, const float* X, float* dW, float* colsum,
                              hipStream_t stream) {
    if (cnt < 1 || npi < 1 || (H != 64 && H != 128 && H != 256) || K < 32 || K > 256 || K % 32) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    WeightGradArgs a{G, X, dW, colsum, npi, cnt, H, K, 1};
    const long long stages = (npi + WG_P - 1) / WG_P;
    const int groups = (H + 127) / 128;
    long long bpi = cus / (cnt * groups);          // one block per CU (96 KiB of LDS each): never more blocks than CUs, or the tail doubles the time
    if (bpi > stages) bpi = stages;
    if (bpi < 1) bpi = 1;
    a.blocks_per_image = (int)bpi;
    switch (K / 32) {
        case 1: return launch_wg<1>(a, stream);
        case 2: return launch_wg<2>(a, stream);
        case 3: return launch_wg<3>(a, stream);
        case 4: return launch_wg<4>(a, stream);
        case 5: return launch_wg<5>(a, stream);
        case 6: return launch_wg<6>(a, stream);
        case 7: return launch_wg<7>(a, stream);
        case 8: return launch_wg<8>(a, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace cnerf
