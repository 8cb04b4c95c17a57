// Half-precision backward of the field network for gfx950 (the counterpart of the reference's fp16 autocast training,
// utils.py:643-711: activations and their gradients in fp16, sums in fp32).  Two kernels:
//
//   weight_grad16_kernel   dW[b] += G[b]^T X[b], colsum[b] += sum G[b]   per image b, G = d loss / d (sine argument) of one
//                          matrix, X = that matrix' input, both fp16 in the TB16 layout (bwd16.hpp), v_mfma_f32_32x32x16_f16
//                          with the point dimension as k, fp32 accumulators.
//   (chain16, further down) the gradient chain g_{l-1} = W_l^T (g_l cos(arg_l) freq_l) on the same instruction.
//
// Why fp16 here: on gfx950 the fp32 MFMA runs at the vector rate (1/16 of the fp16 rate), which made the two gradient GEMM
// families 60 % of a training step of the render path; their operands are bounded (sines, cosines) or scaled per layer by a
// power of two, every sum is fp32, and the quantity they feed (a weight gradient over ~1e6 points) carries the operand
// rounding (2^-11 relative, zero mean) at the 4e-4 level -- inside the gradient gate of the tests (2e-3 of the reference's
// autograd).  The exact fp32 path (field_kernel.hip / grad_kernels.hip) stays available: ops.py `backward_precision`.
#include "bwd16.hpp"
#include "cnerf_dev.hpp"
#include "cnerf_kernels.hpp"
#include "field_common.hpp"

namespace cnerf {

typedef __fp16 hv4 __attribute__((__vector_size__(8)));

// ds_read_b64_tr_b16: per group of 16 lanes a block of 4 rows x 16 columns of 16-bit elements, delivered column-major: lane i
// of the group receives column i of the 4 rows.  Lane 4q + p of the group supplies the address of row q, columns 4p .. 4p+3.
// (Layout and lane map verified on hardware by scripts/ubench/tr16_probe.hip.)
__device__ __forceinline__ u32x2_ lds_tr16(const char* p) {
    const hv4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hv4*)p);
    return __builtin_bit_cast(u32x2_, v);
}
__device__ __forceinline__ f16x8 frag8(u32x2_ lo, u32x2_ hi) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(f16x8, v);
}

// acc += lo(x) + hi(x) for a packed fp16 pair, fp32 accumulate
__device__ __forceinline__ void dot2_ones(float& acc, uint32_t x) {
    asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(0x3c003c00u));
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradients
// ---------------------------------------------------------------------------------------------------------------
struct WeightGrad16Args {
    const _Float16* G;        // TB16 (tiles, g_ct, 32, 32): the first NOC channel tiles are this call's output rows
    const _Float16* X;        // TB16 (tiles, x_ct, 32, 32): channel tiles [x_t0, x_t0 + NJ) are this call's columns
    float* dW;                // (cnt, n_rows, ld) accumulated into, true units
    float* colsum;            // (cnt, n_rows) accumulated into (null: skipped)
    const float* inv_scale;   // device scalar: G holds scale * g; null: 1
    long long tiles_per_image;
    int cnt, g_ct, x_ct, x_t0, ld, n_rows, blocks_per_image;
};

// One block = NOC waves; wave w owns output rows [32 w, 32 w + 32) x NJ column tiles (NJ * 16 accumulator registers).  A stage
// = one 32-point tile: its G block (NOC x 2 KiB) and X block (NJ x 2 KiB) are copied into LDS by linear LDS-DMA (1 KiB per
// wave instruction), double-buffered, one barrier per stage.  Per 16-point k-chunk a wave reads its A fragment (8 points of
// its output channel per lane) and the NJ B fragments with two transposed reads each and issues NJ MFMAs.
template <int NOC, int NJ>
__global__ __launch_bounds__(64 * NOC) void weight_grad16_kernel(WeightGrad16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    constexpr int STAGE = (NOC + NJ) * 2048;                 // bytes
    constexpr int PIECES = (NOC + NJ) * 2;                   // 1-KiB pieces per stage
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int b = blockIdx.x / a.blocks_per_image, part = blockIdx.x - b * a.blocks_per_image;
    const long long t_begin = a.tiles_per_image * part / a.blocks_per_image, t_end = a.tiles_per_image * (part + 1) / a.blocks_per_image;
    const long long tile0 = (long long)b * a.tiles_per_image;
    const char* Gb = reinterpret_cast<const char*>(a.G);
    const char* Xb = reinterpret_cast<const char*>(a.X);

    auto dma_stage = [&](long long t, int buf) {
        char* dst = smem16 + buf * STAGE;
        const char* gsrc = Gb + (size_t)(tile0 + t) * a.g_ct * 2048;
        const char* xsrc = Xb + ((size_t)(tile0 + t) * a.x_ct + a.x_t0) * 2048;
#pragma unroll
        for (int i = 0; i < (PIECES + NOC - 1) / NOC; ++i) {
            const int p = wave_u + i * NOC;                  // wave-uniform
            if (p < PIECES) {
                const char* src = (p < 2 * NOC ? gsrc + p * 1024 : xsrc + (p - 2 * NOC) * 1024) + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
            }
        }
    };

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    float cs = 0.0f;
    // the lane's address inside a 2-KiB block for k-chunk 0, read 0: row (point) 8h + q, columns 16 grp + 4p
    const int grp = (lane >> 4) & 1, h = lane >> 5, q = (lane & 15) >> 2, p4 = lane & 3;
    const int lane_off = (8 * h + q) * 64 + (16 * grp + 4 * p4) * 2;

    if (t_begin < t_end) dma_stage(t_begin, 0);
    int cur = 0;
    for (long long t = t_begin; t < t_end; ++t) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's pieces of stage t have landed
        __syncthreads();                                     // ... and everybody else's; the other buffer is free
        if (t + 1 < t_end) dma_stage(t + 1, cur ^ 1);
        const char* gs = smem16 + cur * STAGE + wave * 2048 + lane_off;
        const char* xs = smem16 + cur * STAGE + NOC * 2048 + lane_off;
#pragma unroll
        for (int c = 0; c < 2; ++c) {                        // two k-chunks of 16 points
            const u32x2_ a0 = lds_tr16(gs + c * 1024), a1 = lds_tr16(gs + c * 1024 + 256);
            const f16x8 A = frag8(a0, a1);
            u32x2_ b0[NJ], b1[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                b0[j] = lds_tr16(xs + j * 2048 + c * 1024);
                b1[j] = lds_tr16(xs + j * 2048 + c * 1024 + 256);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, frag8(b0[j], b1[j]), acc[j], 0, 0, 0);
            // column sums of G: sum of the lane's 8 points of its channel, fp32 (v_dot2_f32_f16 against (1, 1); written as asm
            // because hipcc 7.2 folds the builtin's second dword of a transposed-read pair onto the first: v_dot2c ... v8, v8)
            dot2_ones(cs, a0[0]);
            dot2_ones(cs, a0[1]);
            dot2_ones(cs, a1[0]);
            dot2_ones(cs, a1[1]);
        }
        cur ^= 1;
    }
    // D[i][j]: lane (c = lane & 31, h), register r holds row (r & 3) + 8 (r >> 2) + 4 h, column c of the 32 x 32 tile
    const float inv = a.inv_scale ? *a.inv_scale : 1.0f;
    const int col = lane & 31;
    float* dWb = a.dW + (size_t)b * a.n_rows * a.ld + 32 * a.x_t0;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < a.n_rows) atomicAdd(dWb + (size_t)row * a.ld + 32 * j + col, acc[j][r] * inv);
        }
    if (a.colsum) {
        const float tot = cs + __shfl_xor(cs, 32, WAVE);     // the two lane halves hold the two halves of the points
        const int row = 32 * wave + col;
        if (h == 0 && row < a.n_rows) atomicAdd(a.colsum + (size_t)b * a.n_rows + row, tot * inv);
    }
}

template <int NOC, int NJ>
static hipError_t launch_wg16_inst(const WeightGrad16Args& a, hipStream_t stream) {
    const int lds_bytes = 2 * (NOC + NJ) * 2048;
    if (hipError_t e = hipFuncSetAttribute((const void*)weight_grad16_kernel<NOC, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) return e;
    hipLaunchKernelGGL((weight_grad16_kernel<NOC, NJ>), dim3((unsigned)(a.cnt * a.blocks_per_image)), dim3(64 * NOC), lds_bytes, stream, a);
    return hipGetLastError();
}

template <int NOC>
static hipError_t launch_wg16_noc(const WeightGrad16Args& a, int nj, hipStream_t stream) {
    switch (nj) {
        case 1: return launch_wg16_inst<NOC, 1>(a, stream);
        case 2: return launch_wg16_inst<NOC, 2>(a, stream);
        case 4: return launch_wg16_inst<NOC, 4>(a, stream);
        case 8: return launch_wg16_inst<NOC, 8>(a, stream);
    }
    return hipErrorInvalidValue;
}

// dW (cnt, n_rows, K) += G^T X over the tiles of each image; K = 32 * x_ct columns in runs of 8 / 4 / 2 / 1 channel tiles.
hipError_t launch_weight_grad16(int cnt, long long tiles_per_image, int n_rows, int g_ct, int x_ct, const void* G, const void* X, float* dW,
                                float* colsum, const float* inv_scale, hipStream_t stream) {
    if (cnt < 1 || tiles_per_image < 1 || n_rows < 1 || g_ct < 1 || x_ct < 1 || x_ct > 8 || n_rows > 32 * g_ct) return hipErrorInvalidValue;
    const int noc = (n_rows + 31) / 32;
    if (noc != 1 && noc != 2 && noc != 4 && noc != 8) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    // blocks per image: one block per CU where a block is 8 waves (H = 256); narrow outputs (the head: one wave per block, 36 KiB of
    // LDS) get several blocks per CU -- with 256 single-wave blocks the head's reduction read its 4.3 GB at 2 TB/s
    const int per_cu = noc >= 4 ? 1 : (noc == 2 ? 2 : 4);
    long long bpi = (long long)cus * per_cu / cnt;
    if (bpi > tiles_per_image) bpi = tiles_per_image;
    if (bpi < 1) bpi = 1;
    WeightGrad16Args a{(const _Float16*)G, (const _Float16*)X, dW, colsum, inv_scale, tiles_per_image, cnt, g_ct, x_ct, 0, 32 * x_ct, n_rows, (int)bpi};
    bool first = true;
    for (int t0 = 0; t0 < x_ct;) {                           // column runs of 8, 4, 2, 1 channel tiles
        int nj = 8;
        while (nj > x_ct - t0) nj >>= 1;
        a.x_t0 = t0;
        WeightGrad16Args run = a;
        if (!first) run.colsum = nullptr;                    // the column sums of G are accumulated by the first run only
        hipError_t e = hipSuccess;
        switch (noc) {
            case 1: e = launch_wg16_noc<1>(run, nj, stream); break;
            case 2: e = launch_wg16_noc<2>(run, nj, stream); break;
            case 4: e = launch_wg16_noc<4>(run, nj, stream); break;
            case 8: e = launch_wg16_noc<8>(run, nj, stream); break;
        }
        if (e != hipSuccess) return e;
        first = false;
        t0 += nj;
    }
    return hipSuccess;
}


// =====================================================================================================================
// chain16: the gradient chain of the field network on v_mfma_f32_32x32x16_f16
// =====================================================================================================================
//   go' = d loss / d head pre-activation (sigmoid' applied)                            -> go16 (TB16, 1 channel tile, x S_go)
//   g_h = W_head^T go'
//   for slab m = last .. 0:   ga_m = g_h * cos(arg_m)                                  -> g16[m] (TB16, x S_m)
//                             gp_m = ga_m * freq_m            (plain sine layers: freq = 1)
//                             g_h  = W_m^T gp_m               (m = 0: the 32 channels of every input tile -> trilinear
//                                                              scatter-add into the gradient volume, fp32 atomics)
// Like the forward kernels, a wave owns a 32-point tile and the accumulator registers of one product are the B operand of
// the next (channel = k).  Gradient VALUES are handled in fp32 registers in true units; fp16 appears only at the two conversion
// points, each with a power-of-two scale:
//   * MFMA B operand: per POINT -- column j of a product depends on column j of B only, so every lane scales its own point.
//     The scale of slab m is fixed BEFORE its values exist, from a bound: |gp_m| <= max|freq_m| * ||W_{m+1}||_1 * max_k|gp_{m+1}[k]|
//     (the point's own largest operand of the product that produces g_h_m; ||W||_1 from the packing kernel, max|freq| per image
//     when the FiLM vectors are staged): T = 2^14 / bound rounded down to a power of two, the accumulators are un-scaled by
//     1 / (s_{m+1} T_{m+1}).  A floating-point format loses nothing to a loose bound except range at the bottom (the bound is
//     ~20x the typical value: 4 of fp16's 30 binades), nothing overflows, tiny-gradient points keep all 11 bits -- and because
//     the scale is known up front the epilogue of output tile t (cos, store, freq, convert) runs element by element in the
//     shadow of the MFMAs of tile t + 1, as in field_h3.hip;
//   * stored g16[m]: one scale S_m per slab for the whole call (the weight-gradient kernel sums over points on the k axis and
//     cannot un-scale per point), chosen by the caller from the slab's largest |ga| as sampled by a DRY instantiation of this
//     kernel over every k-th tile group (no stores, atomicMax per slab); fmed3 clamps what a sample might have missed.
// Residual blocks (HAS_RES; x' = sin(x + W2 sin(W1 x + b1) + b2), siren.py:218-230) are two slabs of the same chain -- fc1 then
// fc2, both with freq = 1 -- plus the identity path: the output of the slab below the block also feeds fc2's argument directly,
// so  g_h of that slab = W1^T gp_fc1 + ga_fc2.  ga_fc2 = gp_fc2 is exactly the B operand fc2's epilogue produced (fp16, times
// that slab's per-point scale T): a copy of those 16 fragments stays in registers across the two products and is added, re-scaled,
// to the accumulators in the epilogue two slabs further down; the operand bound of that slab grows by the point's max |ga_fc2|.
// Weight units (two 32-row output tiles of a transposed matrix: 2 * KCH KiB) stream through a three-slot ring in LDS by
// LDS-DMA, shared by the block's four waves, which work on four tiles of one image in lockstep (one barrier per unit) -- the
// scheme of field_h3.hip.  cos(arg) is fetched a slab's worth of output tiles ahead of the epilogue that consumes it.
constexpr int C16_MAX_SLABS = 2 * CNERF_MAX_LAYERS;
enum { C16_FILM = 0, C16_SINE = 1, C16_RES_FC1 = 2, C16_RES_FC2 = 3 };
struct Chain16Args {
    FieldArgs f;              // geometry, tiles, freq, flags, layer kinds, gradient volumes, grad_out / saved_out
    const f16x8* units;       // transposed weight units in consumption order (pack_chain16)
    const f16x8* head_t;      // head^T fragments (NT x 64 lanes)
    const float* winv;        // device: 1 / s_m per slab (index m), then the head's; then ||W_m||_1 per slab and the head's
    const float* scales;      // device: per slab m {S_m, 1 / S_m}, then {S_go, 1 / S_go}
    const _Float16* cos16;    // COS16 (nslab, tiles, NT, 4, 64, 4): fragment-major, bwd16.hpp
    _Float16* g16;            // TB16 (nslab, tiles, NT, 32, 32)
    _Float16* go16;           // TB16 (tiles, 1, 32, 32)
    unsigned int* gmax;       // dry run: per slab the bits of max |ga| (non-negative floats order like their bits), then max |go'|
    unsigned int* sat;        // chain run, optional: += 1 per (tile, slab) whose stored gradients left fp16's range and were clamped
    int nslab;
    int group_step;           // process every group_step-th tile group of a block's range (dry-run sampling)
    unsigned char slab_kind[C16_MAX_SLABS];   // per slab: C16_FILM / C16_SINE / C16_RES_FC1 / C16_RES_FC2
};

__device__ __forceinline__ float pow2_scale_to_2p14(float bound) {
    // T = 2^(14 - e) with bound = m 2^e, m in [0.5, 1): bound * T in [2^13, 2^14).  bound == 0 (or denormal / huge) -> 1.
    const int e = (int)((__float_as_uint(bound) >> 23) & 255u) - 126;
    const int te = 127 + 14 - e;
    return (bound >= 1e-30f && te > 0 && te < 255) ? __uint_as_float((uint32_t)te << 23) : 1.0f;
}

// per-tile state of the pipelined epilogue
struct Epi16 {
    float US, UT;             // accumulator -> stored scale (U * S_m) and -> operand scale (U * T)
    const float* fr;          // LDS: freq of the slab (FiLM) or null (plain sine)
    _Float16* gdst;           // g16 row of the lane's point, channel tile 0 of the slab (+ 4 h)
    float vmax;               // running max |acc * cos| of the point (this lane's channels; accumulator units): ONE tracked maximum serves the
                              // dry run's sampled maximum and the chain run's clamp report (x US = the stored value before its clamp, exactly) and,
                              // times the slab's max |freq|, the bound on the next slab's operands
    float s4[4];              // the stored quad being assembled
    float gp_even;            // the operand pair being assembled
    bool live;
    float kskip;              // HAS_RES: identity-path fragments (x T_skip) -> accumulator units; 0: this slab has no identity input
};

__device__ __forceinline__ float h16_lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float h16_hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }

// one element r of output tile t: acc -> (store, operand fragment)
template <bool DRY, bool HAS_RES>
__device__ __forceinline__ void epi16_element(const f32x16& acc, const f16x4* cosq, int t, int h, int r, Epi16& st, u32x4* frag_out,
                                              const u32x4* frag_skip) {
    const int gq = r >> 2, e = r & 3;
    float av = acc[r];
    if (HAS_RES && st.kskip != 0.0f) {       // (wave-uniform) + the identity path of the residual block above
        const uint32_t w = frag_skip[2 * t + (r >> 3)][(r & 7) >> 1];
        av = __builtin_fmaf((r & 1) ? h16_hi(w) : h16_lo(w), st.kskip, av);
    }
    const float ac = av * (float)cosq[gq][e];
    const float gs = ac * st.US;                                             // ga * S_m
    const float f = st.fr ? st.fr[32 * t + 8 * gq + 4 * h + e] : 1.0f;
    const float gt = (ac * f) * st.UT;                                       // gp * T
    st.vmax = fmaxf(st.vmax, fabsf(ac));
    if (!DRY) {
        st.s4[e] = __builtin_amdgcn_fmed3f(gs, -65504.0f, 65504.0f);
        if (e == 3 && st.live)
            *reinterpret_cast<u32x2_*>(st.gdst + t * 1024 + 8 * gq) = u32x2_{pk_f16(st.s4[0], st.s4[1]), pk_f16(st.s4[2], st.s4[3])};
    }
    if ((r & 1) == 0) st.gp_even = gt;
    else frag_out[2 * t + (r >> 3)][(r & 7) >> 1] = pk_f16(st.gp_even, gt);
}

template <int NT, bool DRY, bool HAS_RES>
__global__ __launch_bounds__(256) void chain16_kernel(Chain16Args A) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const FieldArgs& a = A.f;
    constexpr int KCH = 2 * NT;                         // k-chunks of 16 per matrix row tile
    constexpr int UNIT_FR = 2 * KCH * 64;               // f16x8 fragments per weight unit (two output tiles)
    constexpr int PW = (2 * KCH) / 4;                   // 1-KiB pieces each wave copies per unit
    // cos prefetch distance, in output tiles.  A quarter of a slab (2 tiles at H = 256, ~8 us of MFMAs: several HBM round trips) instead
    // of round 2's whole slab: 48 fewer registers for the ring -- with one tracked maximum (Epi16::vmax) the three H = 256 instantiations
    // spill 20 / 8 / 20 bytes (chain / dry run / residual) where a whole slab ahead spilled 20 / 308 / 120 and round 2's code 20 / 316 / 132;
    // measured equal in time (78.5-79.6 ms per step whichever distance), kept for the margin to the 512-register limit
    constexpr int CD = NT >= 8 ? NT / 4 : (NT >= 2 ? 2 : 1);
    constexpr int EPC = 16 / KCH > 0 ? 16 / KCH : 1;    // epilogue elements per k-chunk (KCH = 16: one)
    f16x8* lds_units = reinterpret_cast<f16x8*>(smem_c);                                   // 3 slots
    f16x8* lds_head = lds_units + 3 * UNIT_FR;                                              // NT * 64 fragments
    float* lds_freq = reinterpret_cast<float*>(lds_head + NT * 64);                        // film_stride floats (image of the block)
    float* lds_fmax = lds_freq + (a.film_stride > 0 ? a.film_stride : 4);                  // per slab: max |freq| of the image (1: sine)
    float* s_g = lds_fmax + C16_MAX_SLABS;                                              // [4][32][33] scatter transpose
    int* s_base = reinterpret_cast<int*>(s_g + 4 * 32 * 33);                               // [4][32][8]
    float* s_w = reinterpret_cast<float*>(s_base + 4 * 32 * 8);                            // [4][32][8]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int j = lane & 31, h = lane >> 5;

    const long long G = (a.tiles_per_image + 3) / 4;
    const long long total_groups = (a.total_tiles / a.tiles_per_image) * G;
    const int nblk = gridDim.x;
    const int cls = blockIdx.x & 7, idx_in_cls = blockIdx.x >> 3;
    const int blk_per_cls = (nblk + 7 - cls) / 8;
    const long long g_begin = total_groups * cls / 8 + (long long)idx_in_cls * A.group_step, g_end = total_groups * (cls + 1) / 8;
    const long long g_stride = (long long)blk_per_cls * A.group_step;
    if (g_begin >= g_end) return;                                                          // block-uniform

    const int n_l0_units = (a.n_in + 1) / 2;
    const int n_units = (A.nslab - 1) * (NT / 2) + n_l0_units;
    // Weight units travel through a ring of THREE LDS slots: unit k + 2 is requested when unit k starts, so a copy has two
    // units' worth of MFMAs to land.
    int dma_k = 0, dma_slot = 0, use_slot = 0;
    auto dma_next = [&]() {
        const f16x8* src = A.units + (size_t)dma_k * UNIT_FR + (size_t)wave_u * PW * 64 + lane;
        f16x8* dst = lds_units + dma_slot * UNIT_FR + wave_u * PW * 64;
#pragma unroll
        for (int q = 0; q < PW; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 64),
                                             (__attribute__((address_space(3))) void*)(dst + q * 64), 16, 0, 0);
        dma_k = dma_k + 1 == n_units ? 0 : dma_k + 1;
        dma_slot = dma_slot == 2 ? 0 : dma_slot + 1;
    };
    // every wave, at the start of every unit: this wave's share of the unit's copy (requested two units ago) must have landed.
    // vmcnt retires in order.  `steady` (second and later units of a matrix): since that request the wave has issued, in
    // order, the cos loads of the tiles in between (4 per prefetch; unit 0 of a matrix prefetches once, every other unit twice, the
    // tail of a matrix once more) and the NEXT unit's PW copy pieces, plus -- unless it is idle or this is the dry run -- 4 gradient
    // stores per tile: at least 12 + PW operations in every case (u = 2 at H = 256: 4 + 8 loads + PW; u = 1, 3: 16 + PW), so waiting
    // until at most 12 + PW are outstanding guarantees the copy and leaves the most recent stores / loads in flight under the MFMAs.
    // Elsewhere the operation count is irregular: full drain.  The barrier itself orders LDS only (lds_only_barrier, cnerf_dev.hpp):
    // __syncthreads() would drain every gradient store in flight at every unit (round 2 did, unknowingly: its counted wait sat in
    // front of a release fence that the compiler turned into vmcnt(0) -- and its count, 16 + PW, was 4 too many for u = 2).
    constexpr int STEADY_N = 12 + PW;
    auto unit_begin = [&](bool steady) -> const f16x8* {
        if (steady) wait_vmcnt<STEADY_N>();
        else wait_vmcnt<0>();
        lds_only_barrier();                                     // everybody's share has landed; nobody still reads the slot refilled next
        dma_next();
        const f16x8* u = lds_units + use_slot * UNIT_FR;
        use_slot = use_slot == 2 ? 0 : use_slot + 1;
        return u;
    };

    for (int i = threadIdx.x; i < NT * 64; i += 256) lds_head[i] = A.head_t[i];
    dma_next();
    dma_next();
    int staged_b = -1;
    const float S_go = A.scales[2 * A.nslab], winv_head = A.winv[A.nslab];
    const float* anorm = A.winv + A.nslab + 1;                  // ||W_m||_1 per slab, then the head's
    if (!a.freq) {
        for (int i = threadIdx.x; i < C16_MAX_SLABS; i += 256) lds_fmax[i] = 1.0f;
        __syncthreads();
    }

    for (long long g = g_begin; g < g_end; g += g_stride) {
        const int b = (int)(g / G);
        const long long tile_in_image = (g - (long long)b * G) * 4 + wave;
        const bool live = tile_in_image < a.tiles_per_image;                               // wave-uniform
        const long long n = tile_in_image * 32 + j;
        const bool valid = live && n < a.n_per_image;
        const long long nn = n < a.n_per_image ? n : a.n_per_image - 1;
        const size_t gpt = (size_t)b * a.n_per_image + nn;
        const long long tile_T = (long long)b * a.tiles_per_image + (live ? tile_in_image : a.tiles_per_image - 1);
        const size_t slab16 = (size_t)a.total_tiles * NT * 1024;
        const size_t row16 = ((size_t)tile_T * NT * 32 + j) * 32;                           // element (T, t = 0, j, 0) inside a slab
        if (b != staged_b && a.freq) {                                                     // block-uniform: FiLM vectors of the image
            __syncthreads();
            for (int i = threadIdx.x; i < a.film_stride; i += 256) lds_freq[i] = a.freq[(size_t)b * a.film_stride + i];
            for (int i = threadIdx.x; i < C16_MAX_SLABS; i += 256) lds_fmax[i] = 1.0f;
            __syncthreads();
            int fi = 0;
            for (int m = 0; m < A.nslab; ++m) {
                if (A.slab_kind[m] != C16_FILM) continue;
                float v = 0.0f;
                for (int i = threadIdx.x; i < NT * 32; i += 256) v = fmaxf(v, fabsf(lds_freq[fi * NT * 32 + i]));
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, WAVE));
                if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(lds_fmax + m), __float_as_uint(v));   // (>= 1 already there: FiLM freq ~ 30)
                ++fi;
            }
            staged_b = b;
            __syncthreads();
        }
        // cos(arg) ring: CD output tiles ahead of their epilogue, over the linear sequence (slab last .. 0) x (tile 0 .. NT-1)
        f16x4 cosr[CD][4];
        auto fetch_cos = [&](int m, int t, int slot) {
            const _Float16* src = A.cos16 + (size_t)m * slab16 + (((size_t)tile_T * NT + t) * 256 + lane) * 4;      // COS16 layout (bwd16.hpp)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) cosr[slot][gq] = *reinterpret_cast<const f16x4*>(src + gq * 256);
        };
#pragma unroll
        for (int t = 0; t < CD; ++t) fetch_cos(A.nslab - 1, t, t);

        // ---- head backward -------------------------------------------------------------------------------------------
        f32x4 go = *reinterpret_cast<const f32x4*>(a.grad_out + gpt * 4);
        if (!valid) go = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.flags & CNERF_F_SIGMOID_RGB) {
            const f32x4 so = *reinterpret_cast<const f32x4*>(a.saved_out + gpt * 4);
            go[0] = go[0] * (so[0] * (1.0f - so[0]));
            go[1] = go[1] * (so[1] * (1.0f - so[1]));
            go[2] = go[2] * (so[2] * (1.0f - so[2]));
        }
        float gs[4], gl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gs[i] = __builtin_amdgcn_fmed3f(go[i] * S_go, -60000.0f, 60000.0f);
            gl[i] = gs[i] - (float)(_Float16)gs[i];
        }
        const float gomax = fmaxf(fmaxf(fabsf(go[0]), fabsf(go[1])), fmaxf(fabsf(go[2]), fabsf(go[3])));
        if (DRY) {
            float m4 = gomax;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m4 = fmaxf(m4, __shfl_xor(m4, d, WAVE));
            if (lane == 0) atomicMax(A.gmax + A.nslab, __float_as_uint(m4));
        } else if (live && h == 0) {                    // row j of the tile's single channel block: channels 0..3 (the rest stays zero)
            *reinterpret_cast<u32x2_*>(A.go16 + ((size_t)tile_T * 32 + j) * 32) = u32x2_{pk_f16(gs[0], gs[1]), pk_f16(gs[2], gs[3])};
        }

        int film_idx = 0;
        for (int l = 0; l < A.nslab; ++l) film_idx += (A.slab_kind[l] == C16_FILM);
        // B operands: the one being consumed and the one being produced; copied over after every slab (64 moves: swapping
        // pointers instead sends both arrays to scratch memory -- the register indices must be static)
        u32x4 frag_in[KCH], frag_out[KCH];
        // HAS_RES: gp (= ga) of the last fc2 slab, its per-point scale and the point's max, for the identity path two slabs below
        u32x4 frag_skip[HAS_RES ? KCH : 1];
        float T_skip = 1.0f, skip_max = 0.0f;

        // epilogue state of the slab whose g_h is being produced
        Epi16 st;
        st.live = live;
        auto begin_slab = [&](int m, float U, float bound_gp) -> float {          // returns T
            const bool film = A.slab_kind[m] == C16_FILM;
            if (film) --film_idx;
            st.fr = film ? lds_freq + (size_t)film_idx * (NT * 32) : nullptr;
            const float T = pow2_scale_to_2p14(bound_gp);
            st.US = U * A.scales[2 * m];
            st.UT = U * T;
            st.gdst = A.g16 + (size_t)m * slab16 + row16 + 4 * h;
            st.vmax = 0.0f;
            st.kskip = 0.0f;
            return T;
        };
        auto keep_skip = [&](int m, float T_m, float gpmax_m) {                    // after a slab's epilogue: is it the fc2 of a block?
            if (HAS_RES && A.slab_kind[m] == C16_RES_FC2) {
#pragma unroll
                for (int c = 0; c < KCH; ++c) frag_skip[c] = frag_out[c];
                T_skip = T_m;
                skip_max = gpmax_m;
            }
        };
        auto end_slab = [&](int m, float T) -> float {                             // returns a bound on the point's max |gp| in true units
            (void)T;
            const float vm = fmaxf(st.vmax, __shfl_xor(st.vmax, 32, WAVE)) * st.US;   // the two lane halves of a point: max |ga| * S_m
            if (DRY) {
                float gm = vm / A.scales[2 * m];
#pragma unroll
                for (int d = 16; d >= 1; d >>= 1) gm = fmaxf(gm, __shfl_xor(gm, d, WAVE));
                if (lane == 0) atomicMax(A.gmax + m, __float_as_uint(gm));
            } else if (A.sat && st.live && __any(vm > 65504.0f)) {
                // The stored scale came from a SAMPLED maximum: say so when it was too small -- exactly, and at no cost per element: the
                // maximum tracked for the operand bound IS the stored quantity before its clamp.  (A second per-lane maximum spilled 70
                // dwords in this kernel; a per-element lane mask stalled on the vector-compare -> scalar-or hazard, 123 -> 142 ms per
                // step on the residual network.)
                if (lane == 0) atomicAdd(A.sat, 1u);
            }
            return vm / A.scales[2 * m] * lds_fmax[m];                          // |gp| = |ga| |freq| <= max |ga| * max |freq| of the slab
        };
        // linear (slab, tile) position -> the cos prefetch CD tiles ahead
        auto prefetch_after = [&](int m, int t) {
            const int tn = t + CD;
            if (tn < NT) fetch_cos(m, tn, t % CD);
            else if (m > 0) fetch_cos(m - 1, tn - NT, t % CD);
        };

        // ---- head product: g_h of the last slab, epilogue right behind each tile (2 MFMAs per tile: nothing to hide under) --
        int m = A.nslab - 1;
        float T = begin_slab(m, winv_head / S_go, lds_fmax[m] * anorm[A.nslab] * gomax);
        {
            const u32x4 bh = h == 0 ? u32x4{pk_f16(gs[0], gs[1]), pk_f16(gs[2], gs[3]), 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
            const u32x4 bl = h == 0 ? u32x4{pk_f16(gl[0], gl[1]), pk_f16(gl[2], gl[3]), 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                const f16x8 aw = lds_head[t * 64 + lane];
                z = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, bl), z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, bh), z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) epi16_element<DRY, HAS_RES>(z, cosr[t % CD], t, h, r, st, frag_out, frag_skip);
                prefetch_after(m, t);
            }
        }
        float gpmax = end_slab(m, T);
        keep_skip(m, T, gpmax);
#pragma unroll
        for (int c = 0; c < KCH; ++c) frag_in[c] = frag_out[c];
        // ---- slabs last-1 .. 0: g_h = W_{m+1}^T gp_{m+1}, epilogue of tile t under the MFMAs of tile t + 1 --------------------
        for (m = A.nslab - 2; m >= 0; --m) {
            const bool skip = HAS_RES && m + 2 < A.nslab && A.slab_kind[m + 2] == C16_RES_FC2;      // block-uniform
            const float Tn = begin_slab(m, A.winv[m + 1] / T, lds_fmax[m] * (anorm[m + 1] * gpmax + (skip ? skip_max : 0.0f)));
            if (skip) st.kskip = T / (T_skip * A.winv[m + 1]);         // fragment units (x T_skip) -> accumulator units (x T / winv)
            f32x16 acc_prev;
#pragma unroll
            for (int u = 0; u < NT / 2; ++u) {
                const f16x8* unit = unit_begin(u >= 1);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int t = 2 * u + sub;
                    const f16x8* tile = unit + sub * KCH * 64 + lane;
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                    f16x8 ring[2] = {tile[0], tile[64]};
#pragma unroll
                    for (int c = 0; c < KCH; ++c) {
                        const f16x8 aw = ring[c & 1];
                        if (c + 2 < KCH) ring[c & 1] = tile[(c + 2) * 64];
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, frag_in[c]), z, 0, 0, 0);
                        if (t > 0) {
#pragma unroll
                            for (int q = 0; q < EPC; ++q)
                                epi16_element<DRY, HAS_RES>(acc_prev, cosr[(t - 1) % CD], t - 1, h, c * EPC + q, st, frag_out, frag_skip);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 9 * EPC, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (t > 0) prefetch_after(m, t - 1);
                    acc_prev = z;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) epi16_element<DRY, HAS_RES>(acc_prev, cosr[(NT - 1) % CD], NT - 1, h, r, st, frag_out, frag_skip);
            prefetch_after(m, NT - 1);
            gpmax = end_slab(m, Tn);
            T = Tn;
            keep_skip(m, T, gpmax);
#pragma unroll
            for (int c = 0; c < KCH; ++c) frag_in[c] = frag_out[c];
        }
        const float U0 = A.winv[0] / T;                  // accumulators of the layer-0 products -> true units
        // ---- layer 0: one 32-channel gradient tile per input tile; feature tiles are scattered, the xyz tile is dropped ----------
        float px, py, pz;
        tile_point(a, b, nn, valid, h, false, px, py, pz);
        const int ch = lane & 31;
        float* sg = s_g + wave * 32 * 33;
        int* sb = s_base + wave * 32 * 8;
        float* sw = s_w + wave * 32 * 8;
        int fi = 0;                                                          // index among the feature tiles (gin slab)
        for (int u = 0; u < n_l0_units; ++u) {
            const f16x8* unit = unit_begin(false);
            for (int sub = 0; sub < 2; ++sub) {
                const int tk = 2 * u + sub;
                if (tk >= a.n_in) break;
                const int lvl = a.in_level[tk];
                if (lvl < 0 || DRY) continue;                                // no gradient flows to the sample positions
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
#pragma unroll
                for (int c = 0; c < KCH; ++c)
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(unit[(sub * KCH + c) * 64 + lane], __builtin_bit_cast(f16x8, frag_in[c]), z, 0, 0, 0);
                if (a.gin) {
                    // ray passes: the tile's 32 x 32 gradients go to HBM in true units (128 B per point) and scatter_patch_kernel adds them
                    // into the volume patch by patch, pre-reduced in LDS (scatter_patch.hip) -- a tile of 32 samples of one ray shares
                    // almost no corners, 8 x 8 neighbouring pixels share most
                    if (valid) {
                        float* dst = a.gin + ((size_t)fi * (size_t)(a.total_tiles / a.tiles_per_image) * a.n_per_image + gpt) * 32 + 4 * h;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq)
                            *reinterpret_cast<f32x4*>(dst + 8 * gq) = f32x4{z[4 * gq] * U0, z[4 * gq + 1] * U0, z[4 * gq + 2] * U0, z[4 * gq + 3] * U0};
                    }
                    ++fi;
                    continue;
                }
                const int V = a.lvl_V[lvl], C = a.lvl_C[lvl];
                Corner8 cr;
                trilinear_corners(px, py, pz, a.half_voxel, V, cr);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sg[j * 33 + 8 * gq + 4 * h + e] = z[4 * gq + e] * U0;
                if (h == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        sb[j * 8 + k] = cr.base[k];
                        sw[j * 8 + k] = valid ? cr.w[k] : 0.0f;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                float* gv = a.lvl_grad[lvl] + (size_t)b * V * V * V * C + a.in_chan[tk];
                // Two points per wave instruction, 32 channels each.  The eight corner weights / voxel indices of a point are two
                // 16-byte LDS reads each (the half-wave reads one address: a broadcast), fetched for the whole iteration up front
                // and unconditionally added -- a zero weight (clamped border corner, padded lane) adds 0.0 to a valid voxel;
                // one LDS round trip per pair of points instead of sixteen dependent ones behind eight branches.
#pragma unroll 2
                for (int pp = 0; pp < 16; ++pp) {
                    const int p = 2 * pp + h;
                    const float gval = sg[p * 33 + ch];
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(sw + p * 8), w1 = *reinterpret_cast<const f32x4*>(sw + p * 8 + 4);
                    const u32x4 b0 = *reinterpret_cast<const u32x4*>(sb + p * 8), b1 = *reinterpret_cast<const u32x4*>(sb + p * 8 + 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        atomicAdd(gv + (size_t)b0[k] * C + ch, gval * w0[k]);
                        atomicAdd(gv + (size_t)b1[k] * C + ch, gval * w1[k]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                                    // drain the copies issued for units this block does not consume
}

// ---------------------------------------------------------------------------------------------------------------
// packing: transposed matrices as fp16 A fragments, scaled per matrix by a power of two (max |W| s = 2^14 .. 2^15)
// ---------------------------------------------------------------------------------------------------------------
// M = W^T (rows = inputs of the layer = outputs of the transposed product, K = H columns).  Fragment index inside the stream
// of one matrix: ((t * KCH + c) * 64 + lane), t = output tile (padded to an even count with zero tiles), element jj of lane
// (i = lane & 31, hh = lane >> 5) = s * M[32 t + i][16 c + 8 (jj >> 2) + 4 hh + (jj & 3)]: the k order in which the
// accumulator registers of the previous product become the B operand (registers 8 s' .. 8 s' + 7 of tile t' are k-chunk
// 2 t' + s': element jj <-> channel 32 t' + 16 s' + 8 (jj >> 2) + 4 hh + (jj & 3)).
__global__ void absmax16_kernel(const float* __restrict__ w, long long n, uint32_t* slot) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, WAVE));
    if ((threadIdx.x & 63) == 0 && m == m) atomicMax(slot, __float_as_uint(m));
}

__global__ void pack_t16_kernel(const float* __restrict__ w, int n_rows_w, int n_cols_w, int n_cols_real, int OT, int KCH, const uint32_t* wmax_slot,
                                float* winv_slot, _Float16* __restrict__ dst) {
    // w: (n_rows_w, n_cols_w) row-major = W; M[r][k] = W[k][r], r < n_cols_real (else 0), k < n_rows_w (= 16 * KCH)
    const float wmax = __uint_as_float(*wmax_slot);
    float s = 1.0f;
    if (wmax > 1e-30f && wmax < 3e38f) {
        int e;
        (void)frexpf(16384.0f / wmax, &e);
        s = ldexpf(1.0f, e - 1 > 100 ? 100 : e - 1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *winv_slot = 1.0f / s;
    const long long total = (long long)OT * KCH * 64 * 8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int jj = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long tc = idx >> 9;
        const int c = (int)(tc % KCH), t = (int)(tc / KCH);
        const int r = 32 * t + (lane & 31);
        const int k = 16 * c + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3);
        const float v = (r < n_cols_real && k < n_rows_w) ? w[(size_t)k * n_cols_w + r] * s : 0.0f;
        dst[idx] = (_Float16)v;
    }
}

// head^T: fragment (t * 64 + lane), element jj = s * W_head[k = 8 hh + jj][32 t + i] for k < 4, else 0
__global__ void pack_head_t16_kernel(const float* __restrict__ w, int H, const uint32_t* wmax_slot, float* winv_slot, _Float16* __restrict__ dst) {
    const float wmax = __uint_as_float(*wmax_slot);
    float s = 1.0f;
    if (wmax > 1e-30f && wmax < 3e38f) {
        int e;
        (void)frexpf(16384.0f / wmax, &e);
        s = ldexpf(1.0f, e - 1 > 100 ? 100 : e - 1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *winv_slot = 1.0f / s;
    const int total = (H / 32) * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int jj = idx & 7, lane = (idx >> 3) & 63, t = idx >> 9;
        const int k = 8 * (lane >> 5) + jj;
        dst[idx] = (_Float16)(k < 4 ? w[(size_t)k * H + 32 * t + (lane & 31)] * s : 0.0f);
    }
}

// ||W||_1 = max over columns r of sum_k |W[k][r]| (W row-major (n_rows, n_cols)): the bound on |W^T x|_inf / |x|_inf the chain
// kernel scales its operands by.  One thread per column, atomicMax of the bits into *slot (zeroed by the caller).
__global__ void col_abs_sum_max_kernel(const float* __restrict__ w, int n_rows, int n_cols, uint32_t* slot) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.0f;
    if (r < n_cols)
        for (int k = 0; k < n_rows; ++k) s += fabsf(w[(size_t)k * n_cols + r]);
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) s = fmaxf(s, __shfl_xor(s, d, WAVE));
    if ((threadIdx.x & 63) == 0 && s == s) atomicMax(slot, __float_as_uint(s));
}

hipError_t launch_col_abs_sum_max(const float* w, int n_rows, int n_cols, float* slot, hipStream_t stream) {
    if (hipError_t e = hipMemsetAsync(slot, 0, sizeof(float), stream)) return e;
    hipLaunchKernelGGL(col_abs_sum_max_kernel, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, stream, w, n_rows, n_cols, (uint32_t*)slot);
    return hipGetLastError();
}

// One transposed matrix: OT output tiles (padded to even), KCH = rows of W / 16 chunks.  scratch: one uint32 per call.
hipError_t launch_pack_t16(const float* w, int n_rows_w, int n_cols_w, int n_cols_real, int OT_padded, void* dst, float* winv_slot, uint32_t* wmax_slot,
                           hipStream_t stream) {
    if (hipError_t e = hipMemsetAsync(wmax_slot, 0, sizeof(uint32_t), stream)) return e;
    const long long n = (long long)n_rows_w * n_cols_w;
    long long rb = (n + 255) / 256;
    if (rb > 256) rb = 256;
    hipLaunchKernelGGL(absmax16_kernel, dim3((unsigned)rb), dim3(256), 0, stream, w, n, wmax_slot);
    const int KCH = n_rows_w / 16;
    const long long total = (long long)OT_padded * KCH * 64 * 8;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_t16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, n_rows_w, n_cols_w, n_cols_real, OT_padded, KCH, (const uint32_t*)wmax_slot,
                       winv_slot, (_Float16*)dst);
    return hipGetLastError();
}

hipError_t launch_pack_head_t16(const float* w, int H, void* dst, float* winv_slot, uint32_t* wmax_slot, hipStream_t stream) {
    if (hipError_t e = hipMemsetAsync(wmax_slot, 0, sizeof(uint32_t), stream)) return e;
    hipLaunchKernelGGL(absmax16_kernel, dim3(4), dim3(256), 0, stream, w, (long long)4 * H, wmax_slot);
    hipLaunchKernelGGL(pack_head_t16_kernel, dim3((unsigned)((H / 32) * 2)), dim3(256), 0, stream, w, H, (const uint32_t*)wmax_slot, winv_slot, (_Float16*)dst);
    return hipGetLastError();
}

struct Chain16Launch {
    const void* units;
    const void* head_t;
    const float* winv;
    const float* scales;
    const void* cos16;
    void* g16;
    void* go16;
    unsigned int* gmax;
    unsigned int* sat;
    int nslab, dry, group_step;
};

// slab kinds from the layer kinds: a residual block contributes fc1 and fc2
static bool slab_kinds_of(const FieldArgs& f, unsigned char* kinds, int nslab) {
    int m = 0;
    bool res = false;
    for (int l = 0; l < f.L; ++l) {
        if (f.layer_kind[l] == CNERF_LAYER_RES) {
            if (m + 2 > C16_MAX_SLABS) return false;
            kinds[m++] = C16_RES_FC1;
            kinds[m++] = C16_RES_FC2;
            res = true;
        } else {
            if (m + 1 > C16_MAX_SLABS) return false;
            kinds[m++] = f.layer_kind[l] == CNERF_LAYER_FILM ? C16_FILM : C16_SINE;
        }
    }
    (void)res;
    return m == nslab;
}

template <int NT, bool DRY, bool HAS_RES>
static hipError_t launch_chain16_nt(const FieldArgs& f, const Chain16Launch& c, hipStream_t stream) {
    Chain16Args A;
    A.f = f;
    A.units = (const f16x8*)c.units;
    A.head_t = (const f16x8*)c.head_t;
    A.winv = c.winv;
    A.scales = c.scales;
    A.cos16 = (const _Float16*)c.cos16;
    A.g16 = (_Float16*)c.g16;
    A.go16 = (_Float16*)c.go16;
    A.gmax = c.gmax;
    A.sat = c.sat;
    A.nslab = c.nslab;
    A.group_step = c.group_step < 1 ? 1 : c.group_step;
    if (!slab_kinds_of(f, A.slab_kind, c.nslab)) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const size_t lds_bytes = (size_t)3 * (2 * 2 * NT * 64) * 16 + (size_t)NT * 64 * 16 + (size_t)(f.film_stride > 0 ? f.film_stride : 4) * 4 +
                             (size_t)C16_MAX_SLABS * 4 + (size_t)4 * 32 * 33 * 4 + (size_t)2 * 4 * 32 * 8 * 4;
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = hipFuncSetAttribute((const void*)chain16_kernel<NT, DRY, HAS_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) return e;
    const long long want = (f.total_tiles / f.tiles_per_image) * ((f.tiles_per_image + 3) / 4);
    int blocks = (int)(want < cus ? want : cus);
    if (blocks < 8) blocks = 8;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((chain16_kernel<NT, DRY, HAS_RES>), dim3(blocks), dim3(256), lds_bytes, stream, A);
    return hipGetLastError();
}

hipError_t launch_chain16(const FieldArgs& f, int H, const void* units, const void* head_t, const float* winv, const float* scales, const void* cos16,
                          void* g16, void* go16, unsigned int* gmax, unsigned int* sat, int nslab, int dry, int group_step, hipStream_t stream) {
    const Chain16Launch c{units, head_t, winv, scales, cos16, g16, go16, gmax, sat, nslab, dry, group_step};
    bool res = false;
    for (int l = 0; l < f.L; ++l) res |= f.layer_kind[l] == CNERF_LAYER_RES;
#define C16_CASE(NT_)                                                                                                                   \
    case NT_:                                                                                                                           \
        if (res) return dry ? launch_chain16_nt<NT_, true, true>(f, c, stream) : launch_chain16_nt<NT_, false, true>(f, c, stream);     \
        return dry ? launch_chain16_nt<NT_, true, false>(f, c, stream) : launch_chain16_nt<NT_, false, false>(f, c, stream);
    switch (H / 32) {
        C16_CASE(2)
        C16_CASE(4)
        C16_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef C16_CASE
}

}  // namespace cnerf
