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
        wait_vmcnt<0>();                                           // this wave's LDS-DMA pieces of stage s have landed (written out: the fence of
        __syncthreads();                                           // __syncthreads() is not a reliable wait for LDS-DMA) ... and everybody's; the other buffer is free
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

// ---------------------------------------------------------------------------------------------------------------
// Parameter gradients from the per-image reductions (cnerf_render_backward): what autograd does behind FiLMLayer / SirenLayer
// (siren.py:146-199) once dWarg[b] = G[b]^T X[b] and cs[b] = column sums of G[b] exist (weight_grad_kernel / weight_grad16_kernel):
//     dW[i][k] += sum_b f_b[i] dWarg[b][i][k]        db[i] += sum_b f_b[i] cs[b][i]                 (f = 1 without FiLM)
//     dphase[b][i] += cs[b][i]                       dfreq[b][i] += sum_k W[i][k] dWarg[b][i][k] + bias[i] cs[b][i]
// One wave per output row i, lanes over k; every output element has one owner: plain read-modify-write, no atomics.
// ---------------------------------------------------------------------------------------------------------------
struct ParamReduceArgs {
    const float* dWarg;   // (cnt, rows, ld)
    const float* cs;      // (cnt, rows)
    const float* freq;    // (cnt, film_stride) already offset to this matrix' H-vector of image 0 of the chunk, or null
    const float* W;       // (rows, k_real) raw weight (dfreq) or null
    const float* bias;    // (rows) or null
    float* dW;            // (rows, k_real) or null
    float* db;            // (rows) or null
    float* g_freq;        // (cnt, film_stride) offset like freq, or null
    float* g_phase;
    int cnt, rows, ld, k_real, film_stride;
};
__global__ __launch_bounds__(64) void param_reduce_kernel(ParamReduceArgs a) {
    const int i = blockIdx.x, lane = threadIdx.x;
    constexpr int SLOTS = 4;                          // k_real <= 256
    float dw[SLOTS] = {0.f, 0.f, 0.f, 0.f};
    float dbi = 0.0f;
    for (int b = 0; b < a.cnt; ++b) {
        const float f = a.freq ? a.freq[(size_t)b * a.film_stride + i] : 1.0f;
        const float c = a.cs[(size_t)b * a.rows + i];
        const float* row = a.dWarg + ((size_t)b * a.rows + i) * a.ld;
        float dot = 0.0f;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int k = lane + 64 * q;
            if (k < a.k_real) {
                const float v = row[k];
                dw[q] = __builtin_fmaf(f, v, dw[q]);
                if (a.g_freq) dot = __builtin_fmaf(a.W[(size_t)i * a.k_real + k], v, dot);
            }
        }
        dbi = __builtin_fmaf(f, c, dbi);
        if (a.g_freq) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d, WAVE);
            if (lane == 0) {
                a.g_freq[(size_t)b * a.film_stride + i] += dot + a.bias[i] * c;
                a.g_phase[(size_t)b * a.film_stride + i] += c;
            }
        }
    }
    if (a.dW) {
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int k = lane + 64 * q;
            if (k < a.k_real) a.dW[(size_t)i * a.k_real + k] += dw[q];
        }
    }
    if (a.db && lane == 0) a.db[i] += dbi;
}
hipError_t launch_param_reduce(int cnt, int rows, int ld, int k_real, const float* dWarg, const float* cs, const float* freq, int film_stride,
                               const float* W, const float* bias, float* dW, float* db, float* g_freq, float* g_phase, hipStream_t stream) {
    if (k_real > 256 || rows < 1 || cnt < 1) return hipErrorInvalidValue;
    ParamReduceArgs a{dWarg, cs, freq, W, bias, dW, db, freq ? g_freq : nullptr, freq ? g_phase : nullptr, cnt, rows, ld, k_real, film_stride};
    hipLaunchKernelGGL(param_reduce_kernel, dim3((unsigned)rows), dim3(64), 0, stream, a);
    return hipGetLastError();
}

// Head of the exact fp32 backward: dW_head (4, H) += go^T x, db_head (4) += column sums of go, over n points; go (n, 4) = d / d
// head pre-activation, x (n, H) = the last hidden activation, both row-major fp32 (cnerf_field_backward's act_go / act_h).
// Thread c of a block owns channel c: x[n][c] is a coalesced row read, go[n][0..3] a broadcast; blocks split the points.
__global__ __launch_bounds__(256) void head_grad32_kernel(const float* __restrict__ go, const float* __restrict__ x, long long n, int H,
                                                          float* __restrict__ dW, float* __restrict__ db) {
    const int c = threadIdx.x;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long n0 = (long long)blockIdx.x * per, n1 = n0 + per < n ? n0 + per : n;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, cs[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long p = n0; p < n1; ++p) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(go + p * 4);
        const float xv = c < H ? x[p * H + c] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] = __builtin_fmaf(g[r], xv, acc[r]);
            cs[r] += g[r];
        }
    }
    if (c < H)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(dW + (size_t)r * H + c, acc[r]);
    if (c == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(db + r, cs[r]);
}
hipError_t launch_head_grad32(const float* go, const float* x, long long n, int H, float* dW, float* db, hipStream_t stream) {
    if (H > 256 || n < 1) return hipErrorInvalidValue;
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(head_grad32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, go, x, n, H, dW, db);
    return hipGetLastError();
}

// max |v| of n floats as the bit pattern of a non-negative float (atomicMax on the bits; *slot zeroed by the caller)
__global__ void absmax_bits_kernel(const float* __restrict__ v, long long n, uint32_t* slot) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(v[i]));
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, WAVE));
    if ((threadIdx.x & 63) == 0 && m == m) atomicMax(slot, __float_as_uint(m));
}
hipError_t launch_absmax_bits(const float* v, long long n, uint32_t* slot, hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, v, n, slot);
    return hipGetLastError();
}

// {S, 1 / S} per sampled maximum: S = 2^(11 - ceil(log2 max)) -- the maximum lands in (2^10, 2^11], a factor 32 below fp16's
// largest number for values the sample missed (the chain clamps beyond) -- and 1 where the maximum is 0.
__global__ void pow2_scales_kernel(const uint32_t* __restrict__ gmax_bits, int n, float* __restrict__ scales) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = __uint_as_float(gmax_bits[i]);
    float S = 1.0f;
    if (g > 0.0f && g < 3e38f) {
        int e;
        const float m = frexpf(g, &e);                 // g = m 2^e, m in [0.5, 1): ceil(log2 g) = e, or e - 1 when g is a power of two
        const int ce = m == 0.5f ? e - 1 : e;
        int se = 11 - ce;
        se = se > 100 ? 100 : (se < -100 ? -100 : se);
        S = ldexpf(1.0f, se);
    }
    scales[2 * i] = S;
    scales[2 * i + 1] = 1.0f / S;
}
hipError_t launch_pow2_scales(const uint32_t* gmax_bits, int n, float* scales, hipStream_t stream) {
    hipLaunchKernelGGL(pow2_scales_kernel, dim3(1), dim3(64), 0, stream, gmax_bits, n, scales);
    return n <= 64 ? hipGetLastError() : hipErrorInvalidValue;
}

}  // namespace cnerf
