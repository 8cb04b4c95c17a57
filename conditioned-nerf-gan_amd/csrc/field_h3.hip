// fp16x3 variant of the fused point pass: fp32-accurate products on the fp16 matrix pipe.
//
// Every fp32 operand is split into two fp16 parts a = a0 + a1 (a0 = a truncated to 11 significant bits, a1 = the
// remainder rounded to fp16: 22 significant bits in total), weights are pre-scaled per matrix by a power of two so that
// both parts stay in fp16's normal range, and a.w is evaluated as the three partial products a0 w1 + a1 w0 + a0 w0 on
// v_mfma_f32_32x32x16_f16 (products of fp16 values are exact in fp32, accumulation is fp32; the dropped a1 w1 term is
// O(2^-22)).  That is 3 x 32 matrix cycles per 16 k instead of 8 x 64 on v_mfma_f32_32x32x2_f32 (x 0.19), and half the
// operand registers of a three-way bf16 split, which is what lets the next tile's lookups stay in flight in registers.
// Measured against the reference (tests/test_gpu_parity.py::test_split_precision): the same 1e-4 gate as the fp32
// kernel, errors at the fp32 noise floor (simulated on the CPU first: rgb 1.1e-5 / sigma 3.2e-5 on short_fg_small
// against 6.7e-6 / 2.1e-5 for exact fp32 products; a single fp16 product would be ~1e-3).
//
// The register chaining of field_kernel.hip carries over: registers 8s..8s+7 of an accumulator tile, converted pairwise,
// are the B fragment of k-chunk 2t+s of the next layer (k order inside the chunk: 8(j>>2) + 4h + (j&3)), and the packed
// A fragments follow the same order (pack_h3_kernel).
//
// Weight units through LDS.  Per point tile the kernel consumes a flat sequence of equally sized weight units: one per
// layer-0 input tile (its NT output tiles x 2 k-chunks), then for every hidden layer its NT output tiles (2*NT k-chunks
// each), then the head -- each 4*NT pieces of 1 KiB (32 KiB at H = 256).  The four waves of a block walk that sequence in
// lockstep on four point tiles of the same image: each wave copies a quarter of the NEXT unit into the idle half of a
// double buffer with LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPRs), all four read the CURRENT
// one with ds_read_b128 (conflict-free: lane-linear 16 B), and one s_barrier per unit both publishes the DMA'd half and
// retires the reads of the other.  Why (scripts/ubench/split_loop_model.hip, one wave per SIMD on all CUs, 20 VALU per
// k-chunk): with every wave streaming its own copy of the weights from L2 (4x the L2 -> CU traffic, ~20 TB/s aggregate)
// a chunk of three MFMAs takes 123 ns, staged through LDS 97 ns, 65 ns for the MFMAs alone.  Biases, the per-layer
// 2^-S factors and the image's freq / phase vectors live in LDS as well: vmcnt retires in order, so a wait for any
// global load behind a DMA burst is a wait for the whole burst.
#include "cnerf_dev.hpp"
#include "cnerf_kernels.hpp"
#include "field_common.hpp"
#include "bwd16.hpp"
#include "h3_dev.hpp"
#include <stdlib.h>

// This translation unit is compiled TWICE (build.py): CNERF_H3_PARTS = 2 (default) is the fp16x3 kernel described above,
// CNERF_H3_PARTS = 1 (-> field_h1.o, precision "fp16") keeps only the leading fp16 part of every operand: one MFMA per 16 k-values
// instead of three, no remainder arithmetic -- plain fp16 products with fp32 accumulation, the numerics class of the reference's
// own GPU path (torch.cuda.amp.autocast, utils.py:327,643) and of BASELINE config 5 ("bf16 SIREN on MFMA"; fp16 rather than bf16:
// same MFMA rate, 8x smaller operand rounding, and the operands here are bounded -- sines -- or scaled by a power of two).  Same
// kernel structure, same packed-stream order with one plane per fragment pair, its own symbol names (inner namespace; h3_dev.hpp).
namespace cnerf {
namespace H3_NS {

// ---------------------------------------------------------------------------------------------------------------
// packing: fragment index ((t*KC + c)*2 + part)*64 + lane (layer 0: (((c>>1)*OT + t)*2 + (c&1))*2 + part), 8 fp16 each:
//   element j of lane (i = lane&31, h = lane>>5) = part_k( S * W[32t + i][16c + 8(j>>2) + 4h + (j&3)] )
// S = 2^floor(log2(16384 / max|W|)) per matrix (on the device, no host round trip); 1/S goes to the kernel.
// ---------------------------------------------------------------------------------------------------------------
__global__ void absmax_kernel(const float* __restrict__ w, long long n, uint32_t* slot) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, WAVE));
    if ((threadIdx.x & 63) == 0 && m == m) atomicMax(slot, __float_as_uint(m));     // non-negative floats order like their bits
}

__device__ __forceinline__ float pow2_scale(float wmax) {
    if (!(wmax > 1e-30f) || !(wmax < 3e38f)) return 1.0f;
    int e;
    (void)frexpf(16384.0f / wmax, &e);          // 16384 / wmax = m 2^e, m in [0.5, 1)
    e = e - 1 > 100 ? 100 : e - 1;
    return ldexpf(1.0f, e);
}

// t_stride: fragment pairs between successive output tiles in dst (KC: dense; larger: the tiles of several matrices interleaved,
// field_pw16.hip)
__global__ void pack_h3_kernel(const float* __restrict__ w, int n_out, int K_real, int KC, int OT, int k_outer, long long t_stride,
                               const uint32_t* wmax_slot, float* inv_scale_slot, _Float16* __restrict__ dst) {
    const float S = pow2_scale(__uint_as_float(*wmax_slot));
    if (blockIdx.x == 0 && threadIdx.x == 0) *inv_scale_slot = 1.0f / S;
    const long long total = (long long)OT * KC * 64 * 8;          // one thread per (t, c, lane, j): writes both parts
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long tc = idx >> 9;
        const int c = (int)(tc % KC), t = (int)(tc / KC);
        const int row = 32 * t + (lane & 31);
        const int col = 16 * c + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
        const float v = (row < n_out && col < K_real) ? w[(size_t)row * K_real + col] * S : 0.0f;
        const _Float16 a0 = (_Float16)v;
        const _Float16 a1 = (_Float16)(v - (float)a0);
        // fragment pair index: (t, c) row-major, or -- layer 0 -- input tile outermost so that the two chunks of one input tile
        // for all output tiles form one contiguous weight unit
        const size_t pair = k_outer ? ((size_t)(c >> 1) * OT + t) * 2 + (c & 1) : (size_t)t * t_stride + c;
        const size_t base = (pair * PARTS) * 64 * 8 + (size_t)lane * 8 + j;
        dst[base] = a0;
        if (PARTS == 2) dst[base + 64 * 8] = a1;
    }
}

static hipError_t pack_impl(const float* w, int n_out, int K_real, int OT, bool k_outer, void* dst, float* inv_scale_slot, float* wmax_slot,
                            hipStream_t stream, long long t_stride) {
    if (hipError_t e = hipMemsetAsync(wmax_slot, 0, sizeof(float), stream)) return e;
    const long long n = (long long)n_out * K_real;
    long long rb = (n + 255) / 256;
    if (rb > 256) rb = 256;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)rb), dim3(256), 0, stream, w, n, (uint32_t*)wmax_slot);
    const int KC = (K_real + 31) / 32 * 2;
    const long long total = (long long)OT * KC * 64 * 8;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_h3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, n_out, K_real, KC, OT, k_outer ? 1 : 0,
                       t_stride > 0 ? t_stride : (long long)KC, (const uint32_t*)wmax_slot, inv_scale_slot, (_Float16*)dst);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Weight folding per image (cnerf_render_forward; the WF instantiations of the kernel below).  As in field_kernel.hip: a scale of
// output rows is a scale of weight rows, so per call and image the packed matrices are re-split with row i scaled by
//     r_b[i] = freq_b[i] / 2 pi   (FiLM)        1 / 2 pi   (plain sine, both matrices of a residual block)
// and the accumulators start from K_b[i] = (freq_b[i] bias[i] + phase_b[i]) / 2 pi: the accumulator is the argument in revolutions.
// The per-matrix power-of-two scale S moves by the power of two above max_i r_b[i] so that both fp16 parts stay in range:
// S' = S 2^-e.  fold16_prepare_kernel: per (image, matrix) the row multipliers r 2^-e, 1 / S', and K S' (in double);
// scale16_kernel: every fragment element = (hi + lo) * multiplier, split again (22 significant bits in, 22 out).  Two launches
// per call for both field passes; 0.8 MB per image (one image per XCD's L2 under the tile banding).
// ---------------------------------------------------------------------------------------------------------------
struct Fold16Args {
    const _Float16* packed;     // the shared packed weight stream (pack_h3_kernel), all matrices then the head
    const float* bias;          // behind it: biases (n_mats x H), head bias (4), 1 / S per matrix and the head's
    const float* freq;          // (B, film_stride) or null
    const float* phase;
    _Float16* img;              // (B, img_elems): the per-image copies
    float* fold;                // (B, n_mats * (H + 1)): K S' per matrix and channel, then 1 / S' per matrix
    float* rowf;                // (B, n_mats, H) scratch: row multipliers r 2^-e
    long long img_elems;
    int B, H, NT, n_mats, L, film_stride;
    int layer_kind[CNERF_MAX_LAYERS];
    int pair_begin[2 * CNERF_MAX_LAYERS + 2];   // first fragment pair of every matrix, then of the head, then the end
};

__global__ __launch_bounds__(256) void fold16_prepare_kernel(Fold16Args a) {
    __shared__ float s_max[4];
    const int b = blockIdx.x / a.n_mats, m = blockIdx.x - b * a.n_mats, i = threadIdx.x;
    int film = -1, mats = 0, films = 0;                 // FiLM index of matrix m (-1: plain sine / residual half)
    for (int l = 0; l < a.L; ++l) {
        const int kind = a.layer_kind[l];
        if (kind == CNERF_LAYER_FILM && mats == m) film = films;
        films += kind == CNERF_LAYER_FILM;
        mats += kind == CNERF_LAYER_RES ? 2 : 1;
    }
    const bool on = i < a.H;
    const double fr = (on && film >= 0) ? (double)a.freq[(size_t)b * a.film_stride + (size_t)film * a.H + i] : 1.0;
    const double ph = (on && film >= 0) ? (double)a.phase[(size_t)b * a.film_stride + (size_t)film * a.H + i] : 0.0;
    const double r = fr * 0.15915494309189533577;
    float rm = on ? fabsf((float)r) : 0.0f;
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) rm = fmaxf(rm, __shfl_xor(rm, d, WAVE));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = rm;
    __syncthreads();
    rm = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    int e = 0;
    if (rm > 1e-30f && rm < 3e38f) {
        const float mant = frexpf(rm, &e);              // rm = mant 2^e, mant in [0.5, 1): rm <= 2^e
        if (mant == 0.5f) e -= 1;                       // an exact power of two
    }
    e = e > 60 ? 60 : (e < -60 ? -60 : e);
    const double shift = ldexp(1.0, -e);                // r 2^-e <= 1 for every row
    const double inv_s = (double)a.bias[(size_t)a.n_mats * a.H + 4 + m];
    if (on) {
        a.rowf[((size_t)b * a.n_mats + m) * a.H + i] = (float)(r * shift);
        a.fold[(size_t)b * a.n_mats * (a.H + 1) + (size_t)m * a.H + i] =
            (float)((fr * (double)a.bias[(size_t)m * a.H + i] + ph) * 0.15915494309189533577 * shift / inv_s);
    }
    if (i == 0) a.fold[(size_t)b * a.n_mats * (a.H + 1) + (size_t)a.n_mats * a.H + m] = (float)(inv_s / shift);
}

__global__ void scale16_kernel(Fold16Args a) {
    const long long per_image = (long long)a.pair_begin[a.n_mats + 1] * 512;          // (pair, lane, j) triples per image
    const long long total = per_image * a.B;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / per_image);
        const long long q = idx - (long long)b * per_image;
        const int j = (int)(q & 7), lane = (int)((q >> 3) & 63), pair = (int)(q >> 9);
        int m = 0;
        while (m < a.n_mats && pair >= a.pair_begin[m + 1]) ++m;
        const size_t base = ((size_t)pair * PARTS) * 512 + (size_t)lane * 8 + j;
        float v = (float)a.packed[base];
        if (PARTS == 2) v += (float)a.packed[base + 512];
        if (m < a.n_mats) {                              // (the head's unit is copied as it is)
            const int pl = pair - a.pair_begin[m];
            const int t = m == 0 ? (pl >> 1) % a.NT : pl / (2 * a.NT);
            v *= a.rowf[((size_t)b * a.n_mats + m) * a.H + 32 * t + (lane & 31)];
        }
        _Float16* dst = a.img + (size_t)b * a.img_elems + base;
        const _Float16 a0 = (_Float16)v;
        dst[0] = a0;
        if (PARTS == 2) dst[512] = (_Float16)(v - (float)a0);
    }
}

static hipError_t fold_impl(const FieldArgs& f, int B, int H, void* img, float* fold, float* rowf, long long img_elems, hipStream_t stream) {
    Fold16Args a;
    a.packed = reinterpret_cast<const _Float16*>(f.packed);
    a.bias = f.bias;
    a.freq = f.freq;
    a.phase = f.phase;
    a.img = reinterpret_cast<_Float16*>(img);
    a.fold = fold;
    a.rowf = rowf;
    a.img_elems = img_elems;
    a.B = B;
    a.H = H;
    a.NT = H / 32;
    a.n_mats = f.n_mats;
    a.L = f.L;
    a.film_stride = f.film_stride;
    for (int l = 0; l < CNERF_MAX_LAYERS; ++l) a.layer_kind[l] = l < f.L ? f.layer_kind[l] : 0;
    int pairs = 0;
    for (int m = 0; m < f.n_mats; ++m) {
        a.pair_begin[m] = pairs;
        pairs += a.NT * 2 * (m == 0 ? f.n_in : a.NT);
    }
    a.pair_begin[f.n_mats] = pairs;                      // head: one 32-row tile
    pairs += 2 * a.NT;
    a.pair_begin[f.n_mats + 1] = pairs;
    if ((long long)pairs * PARTS * 512 != img_elems) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fold16_prepare_kernel, dim3((unsigned)(B * f.n_mats)), dim3(256), 0, stream, a);
    const long long total = (long long)pairs * 512 * B;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// epilogue: bias, FiLM, sine, split
// ---------------------------------------------------------------------------------------------------------------
// bias / freq / phase of channels (r, r+1) of output tile t, lane half h, from LDS.  Issued one step ahead of their use:
// an LDS read that is consumed at once stalls the wave for the full ds latency (lgkmcnt retires in order, behind the
// A-fragment prefetches).
// Folded FiLM.  arg = freq (acc / S + b) + phase is affine in the accumulator, and v_sin_f32 takes
// revolutions, so per image, matrix and channel three constants are prepared in LDS (in double, once per image and block):
//     M = freq / (2 pi S) = Mh + Ml (two floats),   K = (freq b + phase) / (2 pi)
// and the epilogue is  n = rint(acc Mh);  u = fma(acc, Ml, fma(acc, Mh, -n)) + K;  sin(2 pi u)  -- 5 vector ops + v_sin where the
// unfolded form (bias fma, FiLM mul, add, two-term Cody-Waite reduction by 2 pi, scale to revolutions) takes 8.  It is also the
// more exact of the two: acc Mh - n is a single-rounding fma of magnitude <= 1/2, the argument never exists rounded at its
// full magnitude (|arg| ~ 200 rad, ulp 1.5e-5 rad), so the result is within ~2e-7 rev of the exact affine map, where the
// reference's own fp32 sequence (x = acc + b, freq x, + phase: three roundings at full magnitude) is ~1e-5 rad from it.
// A plain sine layer is freq = 1, phase = 0.  The second matrix of a residual block (x + W2 y + b2) keeps the unfolded form.
// Ml is dropped (M rounded to fp32: one vector op and one LDS read per activation pair less; the error is the size of one of the
// reference's own roundings of the argument, measured in round 2: DESIGN.md 3.1)

struct FilmPair {
    f32x2 fr, ph, bs;       // FOLD (non-residual matrices): fr = Mh, ph = Ml, bs = K
};
__device__ __forceinline__ FilmPair film_pair_load(const float* lbias, const float* lfr, const float* lph, int t, int h, int r) {
    FilmPair f;
    const int ch = 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
    f.bs = *reinterpret_cast<const f32x2*>(lbias + ch);
    f.fr = *reinterpret_cast<const f32x2*>(lfr + ch);
    f.ph = *reinterpret_cast<const f32x2*>(lph + ch);
    return f;
}

// Activation rows of the lane's point for the layer being produced (activation-storing forward of the backward pass):
// sin(arg) goes to act_h, cos(arg) to act_c, four consecutive channels (two pairs) per 16-byte store.
struct ActStore {
    float* row_h;
    float* row_c;
    float ks[2], kc[2];      // the first pair of a quad, held until the second one arrives
    // STORE == 2: fp16 in the TB16 layout (bwd16.hpp): the lane's row (point j) inside the 2-KiB block of channel tile 0 of
    // the slab being produced; channel tile t is 1024 elements further on.  `live` is false for an idle wave (a tile past the
    // end of its image: it has no block of its own).
    _Float16* blk_h;
    _Float16* blk_c;
    bool live;
    float pv, pc;            // balanced epilogue: sine (and cosine) of a pair's first element, computed one k-chunk before the second
};
constexpr int STORE_NONE = 0, STORE_F32 = 1, STORE_TB16 = 2;

// acc holds S * (W x) of accumulator elements r, r+1 (r even): pre = acc / S + bias in one rounding (S is a power of
// two), FiLM with product and sum rounded separately like the reference (a plain sine layer runs with freq = 1,
// phase = 0, which is exact -- one branch-free code path), sine, split into the fragments of the chunk pair `out2`
// (element r of the tile is element r & 7 of chunk r >> 3).  Pairs of a tile must arrive in order r = 0, 2, 4, ...

// RESID (second matrix of a residual block, siren.py:218-230): the slot the result goes to still holds the block's input
// x as its two fp16 parts; x = hi + lo is added to W2 y + b2 before the sine, and the slot is overwritten.
// PHASE 0: both elements at once (layer 0, tails).  PHASE 1: only the pair's first element -> st.pv / st.pc; PHASE 2: the second
// element and everything that needs both (stores, split) -- the two halves of the balanced epilogue, one k-chunk apart, so
// that every chunk of the MFMA loop carries about the same number of vector instructions (an un-split pair every second
// chunk puts ~9 vector slots into each of three MFMA gaps and none into the next three).
// WF (weight-folded, cnerf_render_forward; scale16_kernel below): the unit came from the IMAGE's copy of the weights, rows scaled by
// freq / 2 pi (FiLM) or 1 / 2 pi (plain sine, both matrices of a residual block), and the accumulator started from (freq bias + phase) / 2 pi
// in the same units -- acc * inv_s IS the argument in revolutions: one multiply, the range reduction (WF = 1: v_fract, FiLM networks;
// WF = 2: u - rint(u), exact, networks of plain sine layers whose few-radian arguments feel fract's half ulp: as in field_kernel.hip),
// v_sin.  No per-channel constants in the epilogue at all; a residual block's second matrix adds its input, x / 2 pi, first.
template <int STORE, bool RESID, int PHASE = 0, int WF = 0>
__device__ __forceinline__ void film_split_pair(const f32x16& acc, float inv_s, const FilmPair& f, int t, int h, int r, Split2* out2,
                                                ActStore& st) {
    float a0 = 0.0f, a1 = 0.0f;
    constexpr bool FOLD = !RESID || WF != 0;
    if (WF) {
        float u0 = acc[r] * inv_s, u1 = acc[r + 1] * inv_s;
        if (RESID) {
            const uint32_t xh = out2[r >> 3].p[0][(r & 7) >> 1], xl = PARTS == 2 ? out2[r >> 3].p[PARTS - 1][(r & 7) >> 1] : 0u;
            u0 = __builtin_fmaf(half_lo(xh) + half_lo(xl), 0.15915494309189535f, u0);
            u1 = __builtin_fmaf(half_hi(xh) + half_hi(xl), 0.15915494309189535f, u1);
        }
        a0 = WF == 1 ? __builtin_amdgcn_fractf(u0) : u0 - __builtin_rintf(u0);
        a1 = WF == 1 ? __builtin_amdgcn_fractf(u1) : u1 - __builtin_rintf(u1);
    } else if (FOLD) {             // a0, a1 = the argument in revolutions, reduced to about [-1/2, 1/2] + K
        if (PHASE != 2) {
            const float n0 = __builtin_rintf(acc[r] * f.fr[0]);
            a0 = __builtin_fmaf(acc[r], f.fr[0], -n0) + f.bs[0];
        }
        if (PHASE != 1) {
            const float n1 = __builtin_rintf(acc[r + 1] * f.fr[1]);
            a1 = __builtin_fmaf(acc[r + 1], f.fr[1], -n1) + f.bs[1];
        }
    } else {
    if (PHASE != 2) a0 = __builtin_fmaf(acc[r], inv_s, f.bs[0]);
    if (PHASE != 1) a1 = __builtin_fmaf(acc[r + 1], inv_s, f.bs[1]);
    }
    if (FOLD) {
    } else if (RESID) {
        const uint32_t xh = out2[r >> 3].p[0][(r & 7) >> 1], xl = PARTS == 2 ? out2[r >> 3].p[PARTS - 1][(r & 7) >> 1] : 0u;
        if (PHASE != 2) a0 = (half_lo(xh) + half_lo(xl)) + a0;
        if (PHASE != 1) a1 = (half_hi(xh) + half_hi(xl)) + a1;
    } else {
        if (PHASE != 2) a0 = f.fr[0] * a0 + f.ph[0];
        if (PHASE != 1) a1 = f.fr[1] * a1 + f.ph[1];
    }
    float v0 = 0.0f, v1 = 0.0f;
    auto sin_of = [](float x) { return FOLD ? __builtin_amdgcn_sinf(x) : sin_2pi_reduced_hw(x); };
    auto sincos_of = [](float x, float& sn, float& cs) {
        if (FOLD) {
            sn = __builtin_amdgcn_sinf(x);
            cs = __builtin_amdgcn_cosf(x);
        } else {
            sincos_2pi_reduced_hw(x, sn, cs);
        }
    };
    if (PHASE == 1) {
        if (STORE) sincos_of(a0, st.pv, st.pc);
        else st.pv = sin_of(a0);
        return;
    }
    if (STORE) {
        float c0, c1;
        if (PHASE == 2) {
            v0 = st.pv;
            c0 = st.pc;
        } else {
            sincos_of(a0, v0, c0);
        }
        sincos_of(a1, v1, c1);
        if ((r & 2) == 0) {
            st.ks[0] = v0;
            st.ks[1] = v1;
            st.kc[0] = c0;
            st.kc[1] = c1;
        } else if (STORE == STORE_F32) {
            const int ch = 32 * t + 8 * (r >> 2) + 4 * h;
            *reinterpret_cast<f32x4*>(st.row_h + ch) = f32x4{st.ks[0], st.ks[1], v0, v1};
            *reinterpret_cast<f32x4*>(st.row_c + ch) = f32x4{st.kc[0], st.kc[1], c0, c1};
        } else if (st.live) {        // four consecutive channels of the lane's point: one 8-byte store per matrix
            const int off = t * 1024 + 8 * (r >> 2) + 4 * h;
            *reinterpret_cast<u32x2_*>(st.blk_h + off) = u32x2_{pk_f16(st.ks[0], st.ks[1]), pk_f16(v0, v1)};
            // cos is read back only by the chain kernel, lane for lane: fragment-major (bwd16.hpp "COS16"), 512 contiguous bytes
            // per wave instruction
            *reinterpret_cast<u32x2_*>(st.blk_c + (t * 4 + (r >> 2)) * 256) = u32x2_{pk_f16(st.kc[0], st.kc[1]), pk_f16(c0, c1)};
        }
    } else {
        v0 = PHASE == 2 ? st.pv : sin_of(a0);
        v1 = sin_of(a1);
    }
    Split2& d = out2[r >> 3];
    switch (r & 7) {
        case 0: split_pair<0>(v0, v1, d); break;
        case 2: split_pair<1>(v0, v1, d); break;
        case 4: split_pair<2>(v0, v1, d); break;
        default: split_pair<3>(v0, v1, d); break;
    }
}

// whole tile at once (layer 0 and the last output tile of a layer), FiLM pairs fetched one step ahead
template <int STORE, bool RESID, int WF = 0>
__device__ __forceinline__ void film_split(const f32x16& acc, float inv_s, const float* lbias, const float* lfr, const float* lph, int t,
                                           int h, Split2* out2, ActStore& st) {
    if constexpr (WF != 0) {
        const FilmPair none{};
#pragma unroll
        for (int r = 0; r < 16; r += 2) film_split_pair<STORE, RESID, 0, WF>(acc, inv_s, none, t, h, r, out2, st);
    } else {
        FilmPair f = film_pair_load(lbias, lfr, lph, t, h, 0);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const FilmPair fcur = f;
            if (r + 2 < 16) f = film_pair_load(lbias, lfr, lph, t, h, r + 2);
            film_split_pair<STORE, RESID>(acc, inv_s, fcur, t, h, r, out2, st);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight units
// ---------------------------------------------------------------------------------------------------------------
// Two weight units in LDS, one barrier per unit (four slots and a barrier per two units measured no different in round 2: the four
// waves arrive together).  Epilogue placement inside the MFMA loop of a hidden layer: one element pair every second k-chunk, up to
// H3_VPM vector instructions behind each MFMA (splitting every pair over two chunks measured 1-3 % slower: the distribution of the
// vector work over the gaps is not what limits this kernel, total issue of the one wave per SIMD is -- profiles/r02_kernel_counters.md).
constexpr int H3_VPM = 7;

template <int NT>
struct H3Lds {
    static constexpr int KC = 2 * NT;
    static constexpr int PIECES = KC * PARTS;          // 1-KiB pieces (64 fragments) per weight unit
    static constexpr int FRAGS = PIECES * 64;          // f16x8 fragments per weight unit
    static constexpr int PER_WAVE = PIECES / 4;        // pieces each wave copies
};


// every weight unit is contiguous in the packed stream (pack_h3_kernel); wave w moves the
// pieces [w*PER_WAVE, (w+1)*PER_WAVE), four per base address (instruction offsets 0, 1, 2, 3 KiB)
template <int NT>
__device__ __forceinline__ void dma_unit_flat(const f16x8* __restrict__ src, f16x8* lds_dst, int wave_u, int lane) {
    constexpr int PW = H3Lds<NT>::PER_WAVE;
    const f16x8* s0 = src + (size_t)wave_u * PW * 64 + lane;
    f16x8* d0 = lds_dst + wave_u * PW * 64;
#pragma unroll
    for (int q = 0; q < (PW + 3) / 4; ++q) {
        const f16x8* sq = s0 + q * 256;
        f16x8* dq = d0 + q * 256;
        if (4 * q + 0 < PW) dma_piece<0>(sq, dq);
        if (4 * q + 1 < PW) dma_piece<1024>(sq, dq);
        if (4 * q + 2 < PW) dma_piece<2048>(sq, dq);
        if (4 * q + 3 < PW) dma_piece<3072>(sq, dq);
    }
}

// acc (one 32-row output tile) += W_unit * x, A fragments from the LDS copy of the unit; the caller's functor runs once
// per k-chunk (the pipelined epilogue of the previous output tile).  VALU_PER_MFMA sizes the interleave groups.
template <int NT, int VALU_PER_MFMA, typename PerChunk>
__device__ __forceinline__ f32x16 h3_tile_from_lds(const f16x8* lds_tile, const Split2* x, f32x16 acc, int lane, PerChunk per_chunk) {
    constexpr int KC = 2 * NT;
    constexpr int AHEAD = 2;
    f16x8 ring[AHEAD][PARTS];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
#pragma unroll
        for (int k = 0; k < PARTS; ++k) ring[i][k] = lds_tile[(i * PARTS + k) * 64 + lane];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        f16x8 a[PARTS];
#pragma unroll
        for (int k = 0; k < PARTS; ++k) a[k] = ring[c % AHEAD][k];
        if (c + AHEAD < KC) {
#pragma unroll
            for (int k = 0; k < PARTS; ++k) ring[c % AHEAD][k] = lds_tile[((c + AHEAD) * PARTS + k) * 64 + lane];
        }
        H3_MFMA3(acc, a, x[c]);          // small terms first, the leading product last
        per_chunk(c);
        if (VALU_PER_MFMA > 0) {
#pragma unroll
            for (int m = 0; m < (PARTS == 1 ? 1 : 3); ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, PARTS == 1 ? 3 * VALU_PER_MFMA : VALU_PER_MFMA, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// layer-0 unit: all NT output tiles, two k-chunks each (the 32 channels of one input tile), one ring across the tiles
template <int NT>
__device__ __forceinline__ void h3_layer0_from_lds(const f16x8* lds_unit, const Split2* f2, f32x16* acc0, int lane) {
    constexpr int Q = 2 * NT;
    constexpr int AHEAD = 2;
    f16x8 ring[AHEAD][PARTS];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
#pragma unroll
        for (int k = 0; k < PARTS; ++k) ring[i][k] = lds_unit[(i * PARTS + k) * 64 + lane];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        f16x8 a[PARTS];
#pragma unroll
        for (int k = 0; k < PARTS; ++k) a[k] = ring[q % AHEAD][k];
        if (q + AHEAD < Q) {
#pragma unroll
            for (int k = 0; k < PARTS; ++k) ring[q % AHEAD][k] = lds_unit[((q + AHEAD) * PARTS + k) * 64 + lane];
        }
        f32x16 acc = acc0[q >> 1];
        H3_MFMA3(acc, a, f2[q & 1]);
        acc0[q >> 1] = acc;
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifdef CNERF_STAMPS
// Diagnostic build only: per-phase s_memtime totals summed over all tiles of all waves into a.stamps[0..7].
#define BSTAMP(i)                                                                                    \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        const unsigned long long now_ = __builtin_readcyclecounter();                                \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        st_[i] += now_ - last_;                                                                      \
        last_ = now_;                                                                                \
    } while (0)
#else
#define BSTAMP(i)
#endif

template <int N>
struct Younger { static constexpr int value = N; };            // compile-time tag of unit_begin() in the kernel
struct ResidNo { static constexpr bool value = false; };      // compile-time tags for the `matrix` lambda of the kernel
struct ResidYes { static constexpr bool value = true; };

// the point a wave works on: group g of 4 consecutive tiles of one image, tile `wave` of the group
struct TilePoint {
    int b;
    long long nn;      // point inside the image (clamped to the last one for idle waves / padded lanes)
    bool valid;
};
__device__ __forceinline__ TilePoint tile_of_group(const FieldArgs& a, long long g, long long G, int wave, int j) {
    TilePoint p;
    p.b = (int)(g / G);
    const long long n = ((g - (long long)p.b * G) * 4 + wave) * 32 + j;
    p.valid = n < a.n_per_image;
    p.nn = p.valid ? n : (a.n_per_image - 1);
    return p;
}

// One block of four waves per CU (512 registers per wave), two weight units in LDS, one barrier per unit.  (Measured in round 2
// and not kept: four slots with a barrier per two units -- no change; two waves per SIMD by a register cap -- spills, slower; a
// two-waves-per-tile variant of this kernel -- 11.7 vs 10.9 ms)
template <int NT, int STORE, bool HAS_RES, int WF>
__global__ __launch_bounds__(256) void field_h3_kernel(FieldArgs a) {
#ifdef CNERF_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_readcyclecounter();
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int UNIT_FR = H3Lds<NT>::FRAGS;
    constexpr int H = NT * 32;
    constexpr int KCH = 2 * NT;
    constexpr int SLOTS = 2;                                        // weight units in LDS: one being read, one being copied
    f16x8* lds = reinterpret_cast<f16x8*>(smem);                                    // SLOTS weight units
    float* lds_bias = reinterpret_cast<float*>(smem + SLOTS * (size_t)UNIT_FR * 16);   // biases, head bias, 1/S per matrix, ones, zeros
    const float* lds_ones = lds_bias + a.bias_floats;
    const float* lds_zeros = lds_ones + H;
    float* lds_freq = lds_bias + a.bias_floats + 2 * H;                             // FiLM vectors of the block's image
    float* lds_phase = lds_freq + a.film_stride;
    float* lds_mh = lds_phase + a.film_stride;                       // folded FiLM constants of the block's image: n_mats x H each
    float* lds_ml = lds_mh + (size_t)a.n_mats * H;
    float* lds_k = lds_ml + (size_t)a.n_mats * H;
    const float* lds_inv_s = lds_bias + (size_t)a.n_mats * H + 4;    // 1/S per matrix (a residual block has two), then the head's
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int j = lane & 31, h = lane >> 5;

    // Work units are groups of 4 consecutive tiles of ONE image (so the block shares freq / phase); the four waves run in
    // lockstep on the tiles of a group.  XCD-aware ownership as in tile_range(): block class c owns a contiguous eighth.
    const long long G = (a.tiles_per_image + 3) / 4;
    const long long total_groups = (a.total_tiles / a.tiles_per_image) * G;
    const int nblk = gridDim.x;
    const int cls = blockIdx.x & 7, idx_in_cls = blockIdx.x >> 3;
    const int blk_per_cls = (nblk + 7 - cls) / 8;
    const long long g_begin = total_groups * cls / 8 + idx_in_cls, g_end = total_groups * (cls + 1) / 8;

    // the per-tile unit sequence, back to back.  WF: every image has its own copy (rows scaled by its FiLM frequencies); the copy runs
    // one unit ahead of the MFMAs, across tile boundaries: `dma_base` is the image of the tile whose units are being requested,
    // `next_base` the image of the group after this one -- switched when the request sequence wraps to unit 0
    const f16x8* w_units = reinterpret_cast<const f16x8*>(a.packed);
    const f16x8* img_units = reinterpret_cast<const f16x8*>(a.packed_img);
    const size_t img_stride = (size_t)a.packed_img_stride;           // f16x8 fragments per image
    int staged_b = -1;                                               // image whose FiLM vectors are in LDS
    if (g_begin >= g_end) return;                                    // block-uniform

    // The block consumes one flat sequence of weight units: n_units per point tile (layer-0 unit per input tile, NT per
    // hidden layer, head), tile after tile.  Unit i lives in LDS slot i % 2; the barrier at the start of a unit publishes the copy of
    // that unit (issued one unit earlier) and frees the other slot for the next copy.
    const int n_units = a.n_in + (a.n_mats - 1) * NT + 1;           // >= 2
    int dma_k = 0, dma_slot = 0;                                     // next unit to copy (index in the tile sequence), its slot
    int use_slot = 0;                                                // slot of the next unit to consume
    const f16x8* dma_base = w_units;
    const f16x8* next_base = w_units;
    auto dma_next = [&]() {       // (past the block's last tile this re-copies units nobody reads: harmless, drained at the end)
        dma_unit_flat<NT>(dma_base + (size_t)dma_k * UNIT_FR, lds + dma_slot * UNIT_FR, wave_u, lane);
        dma_k = dma_k + 1 == n_units ? 0 : dma_k + 1;
        if (WF && dma_k == 0) dma_base = next_base;
        dma_slot = (dma_slot + 1) & (SLOTS - 1);
    };
    // every wave, at the start of every unit: this wave's share of the unit's copy must have LANDED before the barrier publishes it.
    // LDS-DMA copies are counted by vmcnt and the wait is written out: the release fence of __syncthreads() is NOT a reliable one --
    // the compiler drains vmcnt there for the stores and loads it keeps score of, and it left this kernel's first hidden-layer
    // barrier (and the layer-0 barrier of multi-tile inputs) without any wait: a block whose copy was slow -- the first tile group of a
    // launch, weights cold in L2 -- then multiplied by a half-copied weight unit (round 3: test_full_size_properties differing between
    // two runs in ~1 of 300 forwards, always whole tiles, almost always a block's first group).  Plain forward: wait for everything
    // (the next tile's lookups are consumed right behind the layer-0 barrier anyway).  Activation-storing forward: the barrier must NOT drain the activation stores in flight (an HBM write round trip at every unit:
    // this alone made the storing forward 16.4 ms against 10.7 plain) -- lds_only_barrier() behind a counted wait.  YOUNGER = a lower
    // bound, known at every call site, on the vector-memory operations this wave has issued since it requested the copy of the unit
    // it is about to read (the request of the previous unit_begin): the activation stores of the epilogues in between, 8 per output
    // tile (4 quads x {sin, cos}); a wave without a tile of its own (`store_live` false) stores nothing and waits for everything.
    bool store_live = true;
    auto unit_begin = [&](auto younger_tag) -> const f16x8* {
        constexpr int YOUNGER = decltype(younger_tag)::value;
        if constexpr (STORE == STORE_NONE) {
            wait_vmcnt<0>();
            __syncthreads();
        } else {
            if (YOUNGER >= 16 && store_live) wait_vmcnt<16>();
            else if (YOUNGER >= 8 && store_live) wait_vmcnt<8>();
            else wait_vmcnt<0>();
            lds_only_barrier();
        }
        dma_next();
        const f16x8* unit = lds + use_slot * UNIT_FR;
        use_slot = (use_slot + 1) & (SLOTS - 1);
        return unit;
    };

    // prologue: biases and scales into LDS, the first weight unit (the first barrier publishes both); position and lookups of
    // the first tile
    for (int i = threadIdx.x; i < a.bias_floats + 2 * H; i += 256) lds_bias[i] = a.bias[i];
    TilePoint tp = tile_of_group(a, g_begin, G, wave, j);
    if (WF) dma_base = next_base = img_units + (size_t)(tp.b + a.image0) * img_stride;
    dma_next();
    float px, py, pz;
    tile_point(a, tp.b, tp.nn, tp.valid, h, true, px, py, pz);
    InputTile it;
    input_tile_issue_volume(a, tp.b, 0, px, py, pz, h, it);      // input tile 0 is a volume tile (checked by the launcher)

    for (long long g = g_begin; g < g_end; g += blk_per_cls) {
        const int b = tp.b;
        const long long nn = tp.nn;
        const bool valid = tp.valid;
        BSTAMP(0);
        if (WF && b != staged_b) {                            // block-uniform: this image's accumulator starts K' S' and its 1 / S' per matrix
            __syncthreads();                                        // nobody still reads the previous image's constants
            const float* fsrc = a.fold + (size_t)(b + a.image0) * (a.n_mats * (H + 1));
            for (int i = threadIdx.x; i < a.n_mats * H; i += 256) lds_k[i] = fsrc[i];
            for (int i = threadIdx.x; i < a.n_mats; i += 256) lds_mh[i] = fsrc[a.n_mats * H + i];
            staged_b = b;
            __syncthreads();
        }
        if (!WF && b != staged_b && (a.freq || staged_b < 0)) {     // block-uniform
            __syncthreads();                                        // nobody still reads the previous image's vectors
            if (a.freq)
                for (int i = threadIdx.x; i < a.film_stride; i += 256) {
                    lds_freq[i] = a.freq[(size_t)b * a.film_stride + i];
                    lds_phase[i] = a.phase[(size_t)b * a.film_stride + i];
                }
            for (int i = threadIdx.x; i < a.n_mats * H; i += 256) {
                const int mm = i / H, ch = i - mm * H;
                int film = -1, mats = 0, films = 0;                 // FiLM index of matrix mm (-1: plain sine / residual half)
                for (int l = 0; l < a.L; ++l) {
                    const int kind = a.layer_kind[l];
                    if (kind == CNERF_LAYER_FILM && mats == mm) film = films;
                    films += kind == CNERF_LAYER_FILM;
                    mats += kind == CNERF_LAYER_RES ? 2 : 1;
                }
                const double fr = film >= 0 ? (double)a.freq[(size_t)b * a.film_stride + (size_t)film * H + ch] : 1.0;
                const double ph = film >= 0 ? (double)a.phase[(size_t)b * a.film_stride + (size_t)film * H + ch] : 0.0;
                const double M = fr * (double)lds_inv_s[mm] * 0.15915494309189533577;
                const float mh = (float)M;
                lds_mh[i] = mh;
                lds_ml[i] = (float)(M - (double)mh);
                lds_k[i] = (float)((fr * (double)lds_bias[i] + ph) * 0.15915494309189533577);
            }
            staged_b = b;
            __syncthreads();                                        // (once per image: not every unit starts with a barrier)
        }
        // raw sample coordinate of the NEXT tile: one load now, consumed behind the head's barrier
        const bool has_next = g + blk_per_cls < g_end;
        const TilePoint tn = tile_of_group(a, has_next ? g + blk_per_cls : g, G, wave, j);
        const TileRaw raw_next = tile_point_fetch(a, tn.b, tn.nn);
        if (WF) next_base = img_units + (size_t)(tn.b + a.image0) * img_stride;
        BSTAMP(1);

        const float* bias = lds_bias;
        const float* lfr = lds_freq;
        const float* lph = lds_phase;

        Split2 x[KCH], y[KCH];
        ActStore st;
        const size_t gpt = (size_t)b * a.n_per_image + nn;            // row of the lane's point in the chunk's activation buffers
        const size_t act_layer = (size_t)a.act_points * H;
        st.row_h = STORE == STORE_F32 ? a.act_h + gpt * H : nullptr;
        st.row_c = STORE == STORE_F32 ? a.act_c + gpt * H : nullptr;
        // TB16: tile T = image * tiles_per_image + tile in image; an idle wave (tile past the image's last) stores nothing
        const long long tile_in_image = (g - (long long)b * G) * 4 + wave;
        const long long tile_T = (long long)b * a.tiles_per_image + tile_in_image;
        const size_t slab16 = (size_t)a.total_tiles * NT * 1024;      // fp16 elements per slab (a.total_tiles = tiles of the chunk)
        st.live = tile_in_image < a.tiles_per_image;
        store_live = STORE == STORE_F32 || st.live;                   // (fp32 rows: idle waves re-store the image's last point)
        st.blk_h = STORE == STORE_TB16 ? reinterpret_cast<_Float16*>(a.act_h) + ((size_t)tile_T * NT * 32 + j) * 32 : nullptr;
        st.blk_c = STORE == STORE_TB16 ? reinterpret_cast<_Float16*>(a.act_c) + ((size_t)tile_T * NT * 256 + lane) * 4 : nullptr;
        // ---- layer 0: one weight unit per input tile ------------------------------------------------------------------
        {
            f32x16 acc0[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (WF) acc0[t] = load_chan16(lds_k, t, h);       // (freq bias + phase) / 2 pi in accumulator units
                else
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc0[t][r] = 0.0f;
            }
            for (int tk = 0; tk < a.n_in; ++tk) {
                if (tk > 0) input_tile_issue(a, b, tk, px, py, pz, h, it);   // tile 0 was issued during the previous head
                const f16x8* unit = unit_begin(Younger<0>{});      // (behind the head: the next tile's lookups are in flight and needed now)
                const f32x16 feat = input_tile_reduce(it, px, py, pz, h);
                float fv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) fv[r] = feat[r];
                if (STORE == STORE_F32) {
                    float* fo = a.act_feat + gpt * (32 * a.n_in) + 32 * tk + 4 * h;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(fo + 8 * g) = f32x4{fv[4 * g], fv[4 * g + 1], fv[4 * g + 2], fv[4 * g + 3]};
                }
                if (STORE == STORE_TB16 && st.live) {        // layer-0 input tile tk, clamped to fp16's range like the MFMA operand
                    _Float16* fo = reinterpret_cast<_Float16*>(a.act_feat) + (((size_t)tile_T * a.n_in + tk) * 32 + j) * 32 + 4 * h;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float c4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) c4[e] = __builtin_amdgcn_fmed3f(fv[4 * g + e], -65504.0f, 65504.0f);
                        *reinterpret_cast<u32x2_*>(fo + 8 * g) = u32x2_{pk_f16(c4[0], c4[1]), pk_f16(c4[2], c4[3])};
                    }
                }
                Split2 f2[2];
                f2[0] = split8_clamped(fv);
                f2[1] = split8_clamped(fv + 8);
                h3_layer0_from_lds<NT>(unit, f2, acc0, lane);
            }
            BSTAMP(2);
            const bool film = a.layer_kind[0] == CNERF_LAYER_FILM;
            const float inv_s = WF ? lds_mh[0] : lds_inv_s[0];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                film_split<STORE, false, WF>(acc0[t], inv_s, lds_k, lds_mh, lds_ml, t, h, &x[2 * t], st);
            }
            if (STORE == STORE_F32) {
                st.row_h += act_layer;
                st.row_c += act_layer;
            }
            if (STORE == STORE_TB16) {
                st.blk_h += slab16;
                st.blk_c += slab16;
            }
            bias += H;
            if (film) {
                lfr += H;
                lph += H;
            }
        }
        BSTAMP(3);
        // ---- hidden layers: NT weight units per matrix -------------------------------------------------------------------
        // One matrix: out[t] = epilogue(W[t] in) for the NT output tiles, the epilogue of tile t-1 pipelined under the MFMAs
        // of tile t.  RESID: `out` is the residual block's input x and is updated in place.
        int m = 1;                                                   // matrix counter (scales, activation slabs)
        auto matrix = [&](const Split2* in, Split2* out, auto resid_tag, const float* fr_arg, const float* ph_arg) {
            constexpr bool RESID = decltype(resid_tag)::value;
            const float inv_s = WF ? lds_mh[m] : lds_inv_s[m];
            const float* ksc = lds_k + (size_t)m * H;                // WF: the accumulators' starting values of this matrix
            // folded constants of matrix m stand in for (bias, freq, phase) where the epilogue is the affine-then-sine one
            constexpr bool FOLDED = !RESID;
            const float* bias_l = FOLDED ? lds_k + (size_t)m * H : bias;
            const float* fr_l = FOLDED ? lds_mh + (size_t)m * H : fr_arg;
            const float* ph_l = FOLDED ? lds_ml + (size_t)m * H : ph_arg;
            f32x16 acc_prev;
            FilmPair fp{};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // since the previous unit's request: t = 0: the epilogues of the last two tiles of the matrix before (first hidden matrix:
                // layer 0's, NT >= 2 tiles) = 16 stores; t = 1: nothing (tile 0 has no epilogue to run under it); t >= 2: tile t-2's 8
                const f16x8* unit = t == 0 ? unit_begin(Younger<16>{}) : t == 1 ? unit_begin(Younger<0>{}) : unit_begin(Younger<8>{});
                f32x16 acc;
                if (WF) acc = load_chan16(ksc, t, h);
                else
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                acc = h3_tile_from_lds<NT, H3_VPM>(unit, in, acc, lane, [&](int c) {
                    if (t > 0 && c < 16) {                     // epilogue of tile t-1, one pair of elements per two chunks
                        if (WF) {
                            if (c & 1) film_split_pair<STORE, RESID, 0, WF>(acc_prev, inv_s, fp, t - 1, h, c - 1, &out[2 * (t - 1)], st);
                        } else {
                            if (!(c & 1)) fp = film_pair_load(bias_l, fr_l, ph_l, t - 1, h, c);
                            else film_split_pair<STORE, RESID>(acc_prev, inv_s, fp, t - 1, h, c - 1, &out[2 * (t - 1)], st);
                        }
                    }
                });
                if (t > 0 && KCH < 16) {                       // narrow networks: the rest of tile t-1's elements
#pragma unroll
                    for (int r = KCH; r < 16; r += 2) {
                        if (WF) film_split_pair<STORE, RESID, 0, WF>(acc_prev, inv_s, fp, t - 1, h, r, &out[2 * (t - 1)], st);
                        else film_split_pair<STORE, RESID>(acc_prev, inv_s, film_pair_load(bias_l, fr_l, ph_l, t - 1, h, r), t - 1, h, r,
                                                           &out[2 * (t - 1)], st);
                    }
                }
                acc_prev = acc;
            }
            film_split<STORE, RESID, WF>(acc_prev, inv_s, bias_l, fr_l, ph_l, NT - 1, h, &out[2 * (NT - 1)], st);
            if (STORE == STORE_F32) {
                st.row_h += act_layer;
                st.row_c += act_layer;
            }
            if (STORE == STORE_TB16) {
                st.blk_h += slab16;
                st.blk_c += slab16;
            }
            bias += H;
            ++m;
        };
        for (int l = 1; l < a.L; ++l) {
            const int kind = a.layer_kind[l];
            if (HAS_RES && kind == CNERF_LAYER_RES) {          // y = sin(W1 x + b1);  x = sin(x + W2 y + b2), in place
                matrix(x, y, ResidNo{}, lds_ones, lds_zeros);
                matrix(y, x, ResidYes{}, lds_ones, lds_zeros);
            } else {
                const bool film = kind == CNERF_LAYER_FILM;
                matrix(x, y, ResidNo{}, film ? lfr : lds_ones, film ? lph : lds_zeros);
#pragma unroll
                for (int c = 0; c < KCH; ++c) x[c] = y[c];
                if (film) {
                    lfr += H;
                    lph += H;
                }
            }
        }
        BSTAMP(4);
        // ---- head (last unit of the sequence).  Behind its barrier: layer-0 unit 0 streams in for the next group, the
        // next tile's position is finished and its 32 lookups are issued -- they fly under the head's MFMAs.
        {
            const f16x8* unit = unit_begin(Younger<16>{});      // the last matrix' tiles NT-2 and NT-1
            float nx, ny, nz;                                   // without a next group: this tile again (harmless, keeps `it` dead above)
            tile_point_finish(a, tn.b, tn.nn, raw_next, tn.valid, h, has_next, nx, ny, nz);
            input_tile_issue_volume(a, tn.b, 0, nx, ny, nz, h, it);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            acc = h3_tile_from_lds<NT, 0>(unit, x, acc, lane, [](int) {});
            if (valid && h == 0) {
                const f32x4 hb = *reinterpret_cast<const f32x4*>(bias);
                const float inv_s = lds_inv_s[a.n_mats];
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = __builtin_fmaf(acc[i], inv_s, hb[i]);
                if (a.flags & CNERF_F_SIGMOID_RGB) {
                    o[0] = sigmoidf_(o[0]);
                    o[1] = sigmoidf_(o[1]);
                    o[2] = sigmoidf_(o[2]);
                }
                *reinterpret_cast<f32x4*>(a.rgb_sigma + ((size_t)b * a.n_per_image + nn) * 4) = o;
            }
            px = nx;
            py = ny;
            pz = nz;
            tp = tn;
        }
        BSTAMP(5);
    }
    wait_vmcnt<0>();                                                 // drain the copies issued for tiles this block does not have: an LDS-DMA
    __syncthreads();                                                 // write must not land after the block has given its LDS back
#ifdef CNERF_STAMPS
    if (a.stamps && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(a.stamps + i, st_[i]);
#endif
}

constexpr size_t LDS_LIMIT = 160 * 1024;

template <int NT>
static size_t h3_lds_bytes(const FieldArgs& a, int slots) {
    // weight units (32 KiB each at H = 256), biases + scales, ones / zeros, freq / phase of one image
    // ... and the folded FiLM constants (3 per matrix and channel)
    return slots * (size_t)H3Lds<NT>::FRAGS * 16 +
           ((size_t)a.bias_floats + 2 * NT * 32 + 2 * (size_t)a.film_stride + 3 * (size_t)a.n_mats * NT * 32) * 4;
}

template <int NT, int STORE, bool HAS_RES, int WF>
static hipError_t launch_h3_inst(const FieldArgs& a, hipStream_t stream) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const size_t lds_bytes = h3_lds_bytes<NT>(a, 2);
    if (lds_bytes > LDS_LIMIT) return hipErrorInvalidValue;
    // (per launch, not once per process: the attribute is per device, and a cached flag would be unsynchronised global state)
    if (hipError_t e = hipFuncSetAttribute((const void*)field_h3_kernel<NT, STORE, HAS_RES, WF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT)) return e;
    const long long want = (a.total_tiles / a.tiles_per_image) * ((a.tiles_per_image + 3) / 4);
    int blocks = (int)(want < cus ? want : cus);        // one block of four waves per CU (512 registers per wave)
    if (blocks < 8) blocks = 8;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((field_h3_kernel<NT, STORE, HAS_RES, WF>), dim3(blocks), dim3(256), lds_bytes, stream, a);
    return hipGetLastError();
}

template <int NT, int STORE>
static hipError_t launch_h3_nt(const FieldArgs& a, hipStream_t stream) {
    bool res = false, film = false;
    for (int l = 0; l < a.L; ++l) {
        res |= a.layer_kind[l] == CNERF_LAYER_RES;
        film |= a.layer_kind[l] == CNERF_LAYER_FILM;
    }
    if constexpr (STORE != STORE_F32) {       // the image's own weight copy (cnerf_render_forward prepared it): the one-op epilogue
        if (a.packed_img && a.fold) {
            if (res) return launch_h3_inst<NT, STORE, true, 2>(a, stream);        // (residual networks have no FiLM layers: exact reduction)
            return film ? launch_h3_inst<NT, STORE, false, 1>(a, stream) : launch_h3_inst<NT, STORE, false, 2>(a, stream);
        }
    }
    return res ? launch_h3_inst<NT, STORE, true, 0>(a, stream) : launch_h3_inst<NT, STORE, false, 0>(a, stream);
}

static hipError_t field_impl(const FieldArgs& a, int H, hipStream_t stream) {
    if (a.n_in < 1 || a.in_level[0] < 0) return hipErrorInvalidValue;      // the cross-tile lookup prefetch assumes a volume tile first
    // a.act_h set: activation-storing forward of the backward pass, fp32 rows or (a.act_tb16) fp16 tile blocks
    const int store = a.act_h == nullptr ? STORE_NONE : (a.act_tb16 ? STORE_TB16 : STORE_F32);
    switch (H / 32) {
        case 2: return store == STORE_TB16 ? launch_h3_nt<2, STORE_TB16>(a, stream) : store ? launch_h3_nt<2, STORE_F32>(a, stream) : launch_h3_nt<2, STORE_NONE>(a, stream);
        case 4: return store == STORE_TB16 ? launch_h3_nt<4, STORE_TB16>(a, stream) : store ? launch_h3_nt<4, STORE_F32>(a, stream) : launch_h3_nt<4, STORE_NONE>(a, stream);
        case 8: return store == STORE_TB16 ? launch_h3_nt<8, STORE_TB16>(a, stream) : store ? launch_h3_nt<8, STORE_F32>(a, stream) : launch_h3_nt<8, STORE_NONE>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace H3_NS

hipError_t H3_LAUNCH_PACK(const float* w, int n_out, int K_real, int OT, bool k_outer, void* dst, float* inv_scale_slot, float* wmax_slot,
                          hipStream_t stream, long long t_stride) {
    return H3_NS::pack_impl(w, n_out, K_real, OT, k_outer, dst, inv_scale_slot, wmax_slot, stream, t_stride);
}

hipError_t H3_LAUNCH_FIELD(const FieldArgs& a, int H, hipStream_t stream) { return H3_NS::field_impl(a, H, stream); }

hipError_t H3_LAUNCH_FOLD(const FieldArgs& a, int B, int H, void* img, float* fold, float* rowf, long long img_elems, hipStream_t stream) {
    return H3_NS::fold_impl(a, B, H, img, fold, rowf, img_elems, stream);
}

}  // namespace cnerf
