// bf16x6 variant of the fused point pass: fp32-accurate products on the bf16 matrix pipe.
//
// Every fp32 operand is split into three bf16 parts a = a0 + a1 + a2 (each the round-to-nearest bf16 of the running
// remainder, 3 x 8 = 24 mantissa bits), and a.b is evaluated as the six partial products a_i b_j with i + j <= 2 on
// v_mfma_f32_32x32x16_bf16 (products of bf16 values are exact in fp32; the dropped terms are O(2^-24)).  That costs
// 6 x 32 cycles per 16 k instead of 8 x 64 cycles on v_mfma_f32_32x32x2_f32 -- 0.375 of the matrix time -- and, unlike
// the fp32 MFMA (which shares the fp32 vector ALUs, scripts/ubench/mfma_valu_overlap.hip), leaves the VALU free for the
// FiLM / sine epilogue and the splitting.  Measured against the reference: rgb 8.6e-6, sigma 3.2e-5 in the scaled metric
// of the parity tests (gate 1e-4), i.e. the fp32 noise floor; three products (bf16x3) would miss the gate (1.4e-4 / 5.4e-4).
//
// The register chaining of field_kernel.hip carries over: registers 8s..8s+7 of an accumulator tile, converted pairwise,
// are the B fragment of k-chunk 2t+s of the next layer (k order inside the chunk: 8(j>>2) + 4h + (j&3)), and the packed
// A fragments follow the same order (pack_bf6_kernel).
#include "cnerf_dev.hpp"
#include "cnerf_kernels.hpp"
#include "field_common.hpp"

namespace cnerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split8 {          // eight fp32 values as three bf16 fragments
    bf16x8 p[3];
};

__device__ __forceinline__ Split8 split8(const float* v) {
    Split8 s;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a0 = (__bf16)v[j];
        const float r1 = v[j] - (float)a0;
        const __bf16 a1 = (__bf16)r1;
        const float r2 = r1 - (float)a1;
        s.p[0][j] = a0;
        s.p[1][j] = a1;
        s.p[2][j] = (__bf16)r2;
    }
    return s;
}

// packed A stream: fragment index ((t*KC + c)*3 + split)*64 + lane, 8 bf16 each:
//   element j of lane (i = lane&31, h = lane>>5) = split_k( W[32t + i][16c + 8(j>>2) + 4h + (j&3)] )
__global__ void pack_bf6_kernel(const float* __restrict__ w, int n_out, int K_real, int KC, int OT, __bf16* __restrict__ dst) {
    const long long total = (long long)OT * KC * 64 * 8;          // one thread per (t, c, lane, j): writes the 3 splits
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const long long tc = idx >> 9;
        const int c = (int)(tc % KC), t = (int)(tc / KC);
        const int row = 32 * t + (lane & 31);
        const int col = 16 * c + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
        const float v = (row < n_out && col < K_real) ? w[(size_t)row * K_real + col] : 0.0f;
        const __bf16 a0 = (__bf16)v;
        const float r1 = v - (float)a0;
        const __bf16 a1 = (__bf16)r1;
        const __bf16 a2 = (__bf16)(r1 - (float)a1);
        const size_t base = ((size_t)tc * 3) * 64 * 8 + (size_t)lane * 8 + j;
        dst[base] = a0;
        dst[base + 64 * 8] = a1;
        dst[base + 2 * 64 * 8] = a2;
    }
}

hipError_t launch_pack_bf6(const float* w, int n_out, int K_real, int OT, void* dst, hipStream_t stream) {
    const int KC = (K_real + 31) / 32 * 2;
    const long long total = (long long)OT * KC * 64 * 8;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_bf6_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, n_out, K_real, KC, OT, (__bf16*)dst);
    return hipGetLastError();
}

constexpr int BRING = 3;     // k-chunks of A fragments in flight (3 x 3 x 16 B per lane)

// acc += sum over KC chunks of the six partial products; A fragments stream through a ring BRING chunks ahead.
template <int KC>
__device__ __forceinline__ f32x16 bf6_accumulate(const bf16x8* __restrict__ wp /* at (t, c = 0) */, const Split8* x, f32x16 acc,
                                                 int lane) {
#ifdef CNERF_BF6_NOLOAD   // timing experiment: every chunk re-reads the first fragments (L1 hits) -> wrong results
#define BF6_IDX(i, k) ((k) * 64 + lane)
#else
#define BF6_IDX(i, k) (((i) * 3 + (k)) * 64 + lane)
#endif
    bf16x8 ring[BRING][3];
#pragma unroll
    for (int i = 0; i < BRING; ++i)
        if (i < KC) {
#pragma unroll
            for (int k = 0; k < 3; ++k) ring[i][k] = wp[BF6_IDX(i, k)];
        }
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        bf16x8 a[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] = ring[c % BRING][k];
        if (c + BRING < KC) {
#pragma unroll
            for (int k = 0; k < 3; ++k) ring[c % BRING][k] = wp[BF6_IDX(c + BRING, k)];
        }
        // small terms first, the leading product last
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], x[c].p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], x[c].p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], x[c].p[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], x[c].p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], x[c].p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], x[c].p[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// FiLM / sine epilogue of one accumulator tile, result split into the two B chunks it provides to the next layer
__device__ __forceinline__ void film_split(const f32x16& acc, const f32x16& fr, const f32x16& ph, Split8* out2) {
    float y[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = sin_pi_reduced(fr[r] * acc[r] + ph[r]);
    out2[0] = split8(y);
    out2[1] = split8(y + 8);
}

// Three bf16 parts of two consecutive values written into elements j, j+1 of the three fragments (one v_cvt_pk each).
template <int J>
__device__ __forceinline__ void split_pair(float v0, float v1, Split8& s) {
    const __bf16 a0 = (__bf16)v0, b0 = (__bf16)v1;
    const float ra = v0 - (float)a0, rb = v1 - (float)b0;
    const __bf16 a1 = (__bf16)ra, b1 = (__bf16)rb;
    s.p[0][J] = a0;
    s.p[0][J + 1] = b0;
    s.p[1][J] = a1;
    s.p[1][J + 1] = b1;
    s.p[2][J] = (__bf16)(ra - (float)a1);
    s.p[2][J + 1] = (__bf16)(rb - (float)b1);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight tiles through LDS.  After layer 0 the kernel consumes a flat sequence of equally sized weight tiles -- for every
// hidden layer its NT output tiles, then the head -- each 2*NT k-chunks x 3 splits x 1 KiB (48 KiB at H = 256) and stored
// back to back in the packed stream.  The four waves of a block walk that sequence in lockstep on four different point
// tiles: each wave copies a quarter of the NEXT weight tile into the idle half of a double buffer with LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPRs), all four read the CURRENT one with ds_read_b128
// (conflict-free: lane-linear 16 B), and one s_barrier per weight tile both publishes the DMA'd half and retires the
// reads of the other.  L2 -> CU traffic drops 4x (the per-CU L1 fill path was the limit: 3 KiB per 6 MFMAs per wave =
// 64 B/clk/CU), the sequence wraps from the head of one point tile to hidden layer 1 of the next without a bubble.
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
struct Bf6Lds {
    static constexpr int KC = 2 * NT;
    static constexpr int FRAGS = KC * 3 * 64;          // bf16x8 fragments per weight tile
    static constexpr int PER_WAVE = KC * 3 / 4;        // 1-KiB DMA pieces per wave (KC*3 is a multiple of 4 for NT >= 2)
};

// Measured A/B on one MI355X (bench.py --precision bf16x6, batch 8): fragments straight from L2 19.5 ms per launch, the
// LDS-DMA staged version 20.15 ms.  PMC of the staged version: s_waitcnt + barrier stalls 22 % of wave cycles (the four
// waves drift apart through the lookups of layer 0 and meet at every weight tile), VALU issue 34 %, matrix pipe busy 46 %,
// clock 2.0 GHz.  The L1 fill path is therefore not what limits this kernel and the direct stream is the default; the
// staged variant stays selectable (-DCNERF_BF6_DIRECT=0) for shapes where L2 -> L1 does become the limit.
#ifndef CNERF_BF6_DIRECT
#define CNERF_BF6_DIRECT 1
#endif

template <int NT>
__device__ __forceinline__ void dma_weight_tile(const bf16x8* __restrict__ src, bf16x8* lds_dst, int wave, int lane) {
    if (CNERF_BF6_DIRECT) return;
    // piece p (1 KiB = 64 fragments) of the tile goes to lds_dst + 64 p; wave w moves pieces w, w+4, ...
#pragma unroll
    for (int i = 0; i < Bf6Lds<NT>::PER_WAVE; ++i) {
        const int piece = wave + 4 * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(lds_dst + piece * 64), 16, 0, 0);
    }
}

// acc (one 32-row output tile) += W_tile * x, A fragments from the LDS copy of the weight tile; optional pipelined
// epilogue of the previous output tile (see bf6 hidden pipeline below) is injected per chunk by the caller's functor.
template <int NT, typename PerChunk>
__device__ __forceinline__ f32x16 bf6_tile_from_lds(const bf16x8* lds_tile, const Split8* x, f32x16 acc, int lane,
                                                    PerChunk per_chunk) {
    constexpr int KC = 2 * NT;
    constexpr int AHEAD = CNERF_BF6_DIRECT ? 3 : 2;
    bf16x8 ring[AHEAD][3];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) ring[i][k] = lds_tile[(i * 3 + k) * 64 + lane];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        bf16x8 a[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] = ring[c % AHEAD][k];
        if (c + AHEAD < KC) {
#pragma unroll
            for (int k = 0; k < 3; ++k) ring[c % AHEAD][k] = lds_tile[((c + AHEAD) * 3 + k) * 64 + lane];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], x[c].p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], x[c].p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], x[c].p[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], x[c].p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], x[c].p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], x[c].p[0], acc, 0, 0, 0);
        per_chunk(c);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

template <int NT>
__global__ __launch_bounds__(256) void field_bf6_kernel(FieldArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16x8* lds = reinterpret_cast<bf16x8*>(smem);               // two weight tiles of Bf6Lds<NT>::FRAGS fragments
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    constexpr int H = NT * 32;
    constexpr int KCH = 2 * NT;
    constexpr size_t FR = 64;
    constexpr int TILE_FR = Bf6Lds<NT>::FRAGS;

    // block-uniform trip count (the four waves run in lockstep on tiles base, base+1, base+2, base+3)
    const int nblk = gridDim.x;
    const int cls = blockIdx.x & 7, idx_in_cls = blockIdx.x >> 3;
    const int blk_per_cls = (nblk + 7 - cls) / 8;
    const long long t_begin = a.total_tiles * cls / 8, t_end = a.total_tiles * (cls + 1) / 8;
    const long long stride = (long long)blk_per_cls * 4;

    const int KC0 = 2 * a.n_in;
    const bf16x8* w_layer0 = reinterpret_cast<const bf16x8*>(a.packed);
    const bf16x8* w_seq = w_layer0 + (size_t)NT * KC0 * 3 * FR;      // weight tile 0 of the flat sequence
    const int n_seq = (a.L - 1) * NT + 1;                            // hidden tiles + head
    int cur = 0;                                                     // LDS half holding the current weight tile
    bool first = true;

    for (long long base = t_begin + (long long)idx_in_cls * 4; base < t_end; base += stride) {
        const long long tile_raw = base + wave;
        const bool live = tile_raw < t_end;
        const long long tile = live ? tile_raw : (t_end - 1);       // idle waves shadow the last tile and store nothing
        const int b = (int)(tile / a.tiles_per_image);
        const long long n = (tile - (long long)b * a.tiles_per_image) * 32 + j;
        const bool valid = live && n < a.n_per_image;
        const long long nn = (n < a.n_per_image) ? n : (a.n_per_image - 1);
        float px, py, pz;
        tile_point(a, b, nn, valid, h, true, px, py, pz);

        if (first) {      // prologue: weight tile 0 into half 0
            dma_weight_tile<NT>(w_seq, lds, wave, lane);
            first = false;
        }
        const float* bias = a.bias;
        const float* ones = a.bias + a.bias_floats;
        const float* zeros = ones + H;
        const float* freq = a.freq ? a.freq + (size_t)b * a.film_stride : nullptr;
        const float* phase = a.phase ? a.phase + (size_t)b * a.film_stride : nullptr;

        Split8 x[KCH], y[KCH];
        // ---- layer 0 straight from L2 (6 KiB per output tile), k-outer over the input tiles --------------------------
        {
            f32x16 acc0[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc0[t] = load_chan16(bias, t, h);
            for (int tk = 0; tk < a.n_in; ++tk) {
                const f32x16 feat = input_tile(a, b, tk, px, py, pz, h);
                float fv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) fv[r] = feat[r];
                Split8 f2[2];
                f2[0] = split8(fv);
                f2[1] = split8(fv + 8);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc0[t] = bf6_accumulate<2>(w_layer0 + ((size_t)(t * KC0 + 2 * tk) * 3) * FR, f2, acc0[t], lane);
            }
            const bool film = a.layer_kind[0] == CNERF_LAYER_FILM;
            const float* fp = film ? freq : ones;
            const float* pp = film ? phase : zeros;
#pragma unroll
            for (int t = 0; t < NT; ++t) film_split(acc0[t], load_chan16(fp, t, h), load_chan16(pp, t, h), &x[2 * t]);
            bias += H;
            if (film) {
                freq += H;
                phase += H;
            }
        }
        // ---- hidden layers + head: the flat weight-tile sequence through LDS ------------------------------------------
        int seq = 0;
        for (int l = 1; l < a.L; ++l) {
            const bool film = a.layer_kind[l] == CNERF_LAYER_FILM;
            const float* fp = film ? freq : ones;
            const float* pp = film ? phase : zeros;
            f32x16 acc_prev, fr_prev, ph_prev;
            float v_even = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (!CNERF_BF6_DIRECT) __syncthreads();        // weight tile `seq` has landed; the other half is free
                dma_weight_tile<NT>(w_seq + (size_t)(seq + 1) * TILE_FR, lds + (cur ^ 1) * TILE_FR, wave, lane);
                const f32x16 fr = load_chan16(fp, t, h);
                const f32x16 ph = load_chan16(pp, t, h);
                f32x16 acc = load_chan16(bias, t, h);
                acc = bf6_tile_from_lds<NT>(CNERF_BF6_DIRECT ? w_seq + (size_t)seq * TILE_FR : lds + cur * TILE_FR, x, acc, lane, [&](int c) {
                    if (t > 0 && c < 16) {
                        const float v = sin_pi_reduced(fr_prev[c] * acc_prev[c] + ph_prev[c]);
                        if (c & 1) {
                            Split8& d = y[2 * (t - 1) + (c >> 3)];
                            switch (c & 7) {
                                case 1: split_pair<0>(v_even, v, d); break;
                                case 3: split_pair<2>(v_even, v, d); break;
                                case 5: split_pair<4>(v_even, v, d); break;
                                default: split_pair<6>(v_even, v, d); break;
                            }
                        } else {
                            v_even = v;
                        }
                    }
                });
                if (t > 0 && KCH < 16) {                       // narrow networks: the rest of tile t-1's elements
#pragma unroll
                    for (int r = KCH; r < 16; r += 2) {
                        const float v0 = sin_pi_reduced(fr_prev[r] * acc_prev[r] + ph_prev[r]);
                        const float v1 = sin_pi_reduced(fr_prev[r + 1] * acc_prev[r + 1] + ph_prev[r + 1]);
                        Split8& d = y[2 * (t - 1) + (r >> 3)];
                        switch (r & 7) {
                            case 0: split_pair<0>(v0, v1, d); break;
                            case 2: split_pair<2>(v0, v1, d); break;
                            case 4: split_pair<4>(v0, v1, d); break;
                            default: split_pair<6>(v0, v1, d); break;
                        }
                    }
                }
                acc_prev = acc;
                fr_prev = fr;
                ph_prev = ph;
                cur ^= 1;
                ++seq;
            }
            film_split(acc_prev, fr_prev, ph_prev, &y[2 * (NT - 1)]);
#pragma unroll
            for (int c = 0; c < KCH; ++c) x[c] = y[c];
            bias += H;
            if (film) {
                freq += H;
                phase += H;
            }
        }
        // ---- head (last weight tile of the sequence); meanwhile weight tile 0 streams in for the next point tile -------
        {
            if (!CNERF_BF6_DIRECT) __syncthreads();
            dma_weight_tile<NT>(w_seq, lds + (cur ^ 1) * TILE_FR, wave, lane);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            if (h == 0) {
                acc[0] = bias[0];
                acc[1] = bias[1];
                acc[2] = bias[2];
                acc[3] = bias[3];
            }
            acc = bf6_tile_from_lds<NT>(CNERF_BF6_DIRECT ? w_seq + (size_t)seq * TILE_FR : lds + cur * TILE_FR, x, acc, lane, [](int) {});
            cur ^= 1;
            (void)n_seq;
            if (valid && h == 0) {
                f32x4 o;
                const bool sg = a.flags & CNERF_F_SIGMOID_RGB;
                o[0] = sg ? sigmoidf_(acc[0]) : acc[0];
                o[1] = sg ? sigmoidf_(acc[1]) : acc[1];
                o[2] = sg ? sigmoidf_(acc[2]) : acc[2];
                o[3] = acc[3];
                *reinterpret_cast<f32x4*>(a.rgb_sigma + ((size_t)b * a.n_per_image + nn) * 4) = o;
            }
        }
    }
    // drain the DMA that was issued for a point tile this block will not process
    __syncthreads();
}

template <int NT>
static hipError_t launch_bf6_nt(const FieldArgs& a, hipStream_t stream) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const size_t lds_bytes = CNERF_BF6_DIRECT ? 0 : 2 * (size_t)Bf6Lds<NT>::FRAGS * 16;   // staged: 96 KiB at H = 256
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)field_bf6_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    const long long want = (a.total_tiles + 3) / 4, cap = (long long)cus;
    int blocks = (int)(want < cap ? want : cap);
    if (blocks < 8) blocks = 8;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL(field_bf6_kernel<NT>, dim3(blocks), dim3(256), lds_bytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_field_bf6(const FieldArgs& a, int H, hipStream_t stream) {
    switch (H / 32) {
        case 2: return launch_bf6_nt<2>(a, stream);
        case 4: return launch_bf6_nt<4>(a, stream);
        case 8: return launch_bf6_nt<8>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace cnerf
