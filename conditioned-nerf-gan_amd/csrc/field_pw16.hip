// Per-point FiLM family (TALLSIREN, siren.py:232-331) on the fp16 matrix pipe: the point pass of field_pw_kernel (field_kernel.hip) with
// every fp32 product evaluated as in field_h3.hip -- two fp16 parts per operand, three MFMAs per 16 k-values (precision "fp16x3"), or
// the leading part only (precision "fp16"; this file is compiled twice like field_h3.hip, see h3_dev.hpp).
//
//   m = LeakyReLU_0.2(Wm1 feat + bm1)            feat = the 32 looked-up channels, m 256 wide
//   per layer l:  fr = Wm2[l H ..] m + bm2,  ph = Wm2[(L + l) H ..] m + bm2,  pre = W_l x + b_l   (x_0 = world position)
//                 x_{l+1} = sin((15 fr + 30) pre + ph)                                             siren.py:81-101, 158, 318-327
//
// 1.5 M MACs per point at H = 256, two thirds of them in the mapping network's second Linear, whose 2 L H outputs are never
// materialised: per 32-channel output tile the kernel runs three accumulations (fr and ph against m, pre against x) and combines them
// in the epilogue.  Same construction as field_h3.hip: one wave per 32-point tile, accumulator registers converted pairwise are the
// next B operands, four waves of a block walk one flat sequence of weight units through a three-slot LDS ring filled by LDS-DMA (one
// barrier per unit, counted vmcnt wait in front of it).  The sequence per tile, in the order the packed stream holds it:
//     Wm1 (8 output tiles x 2 k-chunks) | W_0 (NT x 2) | layer 0: per output tile t  fr_t, ph_t  (16 k-chunks each: K = 256)
//     | layers 1..L-1: per t  fr_t, pre_t (2 NT chunks: K = H), ph_t | head (1 tile x 2 NT chunks)
// i.e. two unit sizes: 16 PARTS pieces of 1 KiB for everything multiplied by m (and Wm1), 2 NT PARTS for what is multiplied by x.
//
// Epilogue.  Every accumulator starts from its bias in accumulator units (S b), and 15 fr + 30 = 15 (fr + 2), so
//     u = arg / 2 pi = (accf + 2 S_f) accp c1 + accph c2,    c1 = 15 / (2 pi S_f S_pre),  c2 = 1 / (2 pi S_ph)      (per layer, prepared on the device)
// is an add, two multiplies and one fma per element (the 2 S_f is added to the FINISHED accumulator: started from S_f (bm2 + 2) the 48
// partial sums of a product each round at the magnitude of 2 S_f -- rgb of `tallsiren_small` 9.6e-5 from the reference instead of 5.5e-5),
// then the exact reduction u - rint(u) and v_sin_f32 (which takes revolutions).  Rounding: the product accf accp and the fma round at the
// magnitude of the argument like the reference's own freq * x + phase does.
// The epilogue of output tile t-1 runs under the 48 MFMAs of fr_t (always 16 k-chunks: one element pair per two chunks), accf accp
// under those of ph_t: 9 MFMAs per accumulator register against 3 in field_h3.hip -- the vector work is covered.
#include "cnerf_dev.hpp"
#include "cnerf_kernels.hpp"
#include "field_common.hpp"
#include "bwd16.hpp"
#include "h3_dev.hpp"

#if CNERF_H3_PARTS == 1
#define PW_LAUNCH_FIELD launch_field_pw1
#else
#define PW_LAUNCH_FIELD launch_field_pw3
#endif

namespace cnerf {
namespace H3_NS {
namespace pw {

constexpr int KCM = 16;            // k-chunks of everything multiplied by m (the mapping network's hidden width is 256)

// PIECES 1-KiB pieces of a weight unit, contiguous in the packed stream: wave w moves the pieces [w PIECES / 4, (w + 1) PIECES / 4)
template <int PIECES>
__device__ __forceinline__ void dma_unit(const f16x8* __restrict__ src, f16x8* lds_dst, int wave_u, int lane) {
    static_assert(PIECES % 4 == 0, "four waves share a unit");
    constexpr int PW = PIECES / 4;
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

// acc (one 32-row output tile) += W_unit x over KC k-chunks, A fragments from the LDS copy of the unit; the caller's functor runs
// once per k-chunk (vector work that rides under the MFMAs: VPM vector instructions are scheduled behind each of them)
template <int KC, int VPM, typename PerChunk>
__device__ __forceinline__ f32x16 tile_kc(const f16x8* lds_tile, const Split2* x, f32x16 acc, int lane, PerChunk per_chunk) {
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
        H3_MFMA3(acc, a, x[c]);
        per_chunk(c);
        if (VPM > 0) {
#pragma unroll
            for (int m = 0; m < (PARTS == 1 ? 1 : 3); ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, PARTS == 1 ? 3 * VPM : VPM, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// a unit of OT output tiles x 2 k-chunks (one 32-wide input tile; pack_h3_kernel's k_outer order): acc0[t] += W[t] f
template <int OT>
__device__ __forceinline__ void input_unit(const f16x8* lds_unit, const Split2* f2, f32x16* acc0, int lane) {
    constexpr int Q = 2 * OT;
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

// elements r, r + 1 (r even) of an output tile -> dword (r & 7) / 2 of the fragments of chunk r >> 3 (out2 = the tile's chunk pair)
__device__ __forceinline__ void split_into(Split2* out2, int r, float v0, float v1) {
    Split2& d = out2[r >> 3];
    switch (r & 7) {
        case 0: split_pair<0>(v0, v1, d); break;
        case 2: split_pair<1>(v0, v1, d); break;
        case 4: split_pair<2>(v0, v1, d); break;
        default: split_pair<3>(v0, v1, d); break;
    }
}

__device__ __forceinline__ float sin_rev(float u) { return __builtin_amdgcn_sinf(u - __builtin_rintf(u)); }

// q = accf accp, ph = accph of elements r, r + 1:  sin(2 pi (q c1 + ph c2))
__device__ __forceinline__ void film_pair(const f32x16& q, const f32x16& ph, float c1, float c2, int r, Split2* out2) {
    const float v0 = sin_rev(__builtin_fmaf(q[r], c1, ph[r] * c2));
    const float v1 = sin_rev(__builtin_fmaf(q[r + 1], c1, ph[r + 1] * c2));
    split_into(out2, r, v0, v1);
}

// Activation-storing forward of the half-precision backward (chain_pw16.hip): per layer and point, fp16, y = sin(arg) as TB16 (bwd16.hpp: the X
// operand of the next layer's weight gradient) and cos(arg) as a COS16 slab (fragment-major).  The other two derivative rows the chain
// multiplies by, cos f and cos 15 pre, come from pw_deriv_kernel below: formed here they need accf and accp of the previous tile next to accph
// and the accumulator being filled, which this kernel's register file does not have (337 spilled registers, 38 ms per launch against 16 plain).
struct PwStore {
    _Float16* blk_h;       // y_l: the lane's row in channel tile 0 of the slab (+ 4 h)
    _Float16* blk_c;       // cos: fragment (tile, t = 0, quad 0, lane) of slab 3 l
    float s[2], c[2];      // the first pair of a quad, held until the second arrives
    bool live;
};

// the plain epilogue plus the cosine; y and cos(arg) leave quad by quad
__device__ __forceinline__ void film_pair_store(const f32x16& q, const f32x16& ph, float c1, float c2, int t, int r, Split2* out2, PwStore& st) {
    float sn[2], cs[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float u = __builtin_fmaf(q[r + e], c1, ph[r + e] * c2);
        const float a = u - __builtin_rintf(u);
        sn[e] = __builtin_amdgcn_sinf(a);
        cs[e] = __builtin_amdgcn_cosf(a);
    }
    if ((r & 2) == 0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            st.s[e] = sn[e];
            st.c[e] = cs[e];
        }
    } else if (st.live) {
        *reinterpret_cast<u32x2_*>(st.blk_h + t * 1024 + 8 * (r >> 2)) = u32x2_{pk_f16(st.s[0], st.s[1]), pk_f16(sn[0], sn[1])};
        *reinterpret_cast<u32x2_*>(st.blk_c + (size_t)(t * 4 + (r >> 2)) * 256) = u32x2_{pk_f16(st.c[0], st.c[1]), pk_f16(cs[0], cs[1])};
    }
    split_into(out2, r, sn[0], sn[1]);
}

struct TilePoint {
    int b;
    long long nn;      // point inside the image (clamped to the last one for idle waves / padded lanes)
    bool valid;
};
// group g of 4 consecutive tiles of one image, tile `wave` of the group (as field_h3.hip)
__device__ __forceinline__ TilePoint tile_of_group(const FieldArgs& a, long long g, long long G, int wave, int j) {
    TilePoint p;
    p.b = (int)(g / G);
    const long long n = ((g - (long long)p.b * G) * 4 + wave) * 32 + j;
    p.valid = n < a.n_per_image;
    p.nn = p.valid ? n : (a.n_per_image - 1);
    return p;
}

struct FirstLayer { static constexpr bool value = true; };
struct LaterLayer { static constexpr bool value = false; };

// Constants in LDS (a.bias, written by pw16_consts_kernel):  bm1 S (256) | per layer: b_l S_pre, bm2 S_f, bm2 S_ph (3 H) | head bias (4)
// | 1 / S of Wm1, 1 / S of the head, per layer c1, c2, 15 / S_f, 15 / S_pre | the raw 1 / S and max|W| slots of the packing (unused here)
// STORE: act_feat = TB16 (tiles, 2, 32, 32): the looked-up feature, the position; act_h = L slabs y_l (tiles, NT, 32, 32) then m (tiles, 8,
// 32, 32); act_c = 3 L COS16 slabs of which this kernel writes cos(arg) (slab 3 l) -- pw_deriv_kernel fills the other two and act_amax.
template <int NT, bool STORE>
__global__ __launch_bounds__(256) void field_pw16_kernel(FieldArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = NT * 32;
    constexpr int KCH = 2 * NT;
    constexpr int BIG = KCM * PARTS, SMALL = KCH * PARTS;           // 1-KiB pieces per weight unit
    constexpr int SLOT_FR = BIG * 64;                               // f16x8 fragments per LDS slot
    f16x8* lds = reinterpret_cast<f16x8*>(smem);
    constexpr int SLOTS = 3;                                        // weight units in LDS: one being read, two being copied
    constexpr int PWMIN = SMALL / 4;                                // copy instructions per wave and unit: at least this many
    float* lds_c = reinterpret_cast<float*>(smem + SLOTS * (size_t)SLOT_FR * 16);
    const int L = a.L;
    const float* c_m1 = lds_c;
    const float* c_lay = lds_c + 256;
    const float* c_head = c_lay + (size_t)3 * L * H;
    const float* c_scal = c_head + 4;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int j = lane & 31, h = lane >> 5;

    // groups of 4 consecutive tiles of one image, XCD-aware ownership as in field_h3.hip / tile_range()
    const long long G = (a.tiles_per_image + 3) / 4;
    const long long total_groups = (a.total_tiles / a.tiles_per_image) * G;
    const int nblk = gridDim.x;
    const int cls = blockIdx.x & 7, idx_in_cls = blockIdx.x >> 3;
    const int blk_per_cls = (nblk + 7 - cls) / 8;
    const long long g_begin = total_groups * cls / 8 + idx_in_cls, g_end = total_groups * (cls + 1) / 8;
    if (g_begin >= g_end) return;                                    // block-uniform

    // the flat unit sequence of a tile; the copy runs one unit ahead of the MFMAs, across tile boundaries
    const int n_units = 2 + 2 * NT + (L - 1) * 3 * NT + 1;
    const f16x8* w_units = reinterpret_cast<const f16x8*>(a.packed);
    const f16x8* dma_src = w_units;
    int dma_k = 0, dma_slot = 0, use_slot = 0;
    auto dma_next = [&]() {      // (past the block's last tile this copies a unit nobody reads: drained at the end)
        bool big = true;
        if (BIG != SMALL) {
            const int k = dma_k;
            if (k == 1 || k == n_units - 1) big = false;                                    // W_0, head
            else if (k >= 2 + 2 * NT) big = (k - 2 - 2 * NT) % 3 != 1;                      // fr, PRE, ph
        }
        if (big) {
            dma_unit<BIG>(dma_src, lds + dma_slot * SLOT_FR, wave_u, lane);
            dma_src += BIG * 64;
        } else {
            dma_unit<SMALL>(dma_src, lds + dma_slot * SLOT_FR, wave_u, lane);
            dma_src += SMALL * 64;
        }
        if (++dma_k == n_units) {
            dma_k = 0;
            dma_src = w_units;
        }
        dma_slot = dma_slot == SLOTS - 1 ? 0 : dma_slot + 1;
    };
    // Every wave, at the start of every unit: its share of the unit's copy has landed.  The copy of unit k is requested two units
    // earlier (three slots); LDS-DMA is counted by vmcnt, which retires in order, and the wait is written out: at most N
    // operations may stay in flight, N = a lower bound on what the wave has issued since that request -- the copy instructions of unit
    // k + 1 (>= PWMIN) and, in the activation-storing forward, the `stores` of the epilogue in between (8 per output tile: 4 quads x
    // {y, cos}; an idle wave stores nothing).  The barrier orders LDS only: the stores stay in flight across it.
    bool store_live = false;
    auto unit_begin = [&](int stores) -> const f16x8* {
        if (STORE && store_live && stores >= 8) wait_vmcnt<8 + PWMIN>();
        else wait_vmcnt<PWMIN>();
        lds_only_barrier();
        dma_next();
        const f16x8* unit = lds + use_slot * SLOT_FR;
        use_slot = use_slot == SLOTS - 1 ? 0 : use_slot + 1;
        return unit;
    };

    for (int i = threadIdx.x; i < a.bias_floats; i += 256) lds_c[i] = a.bias[i];
    dma_next();
    dma_next();
    __syncthreads();             // the constants are plain LDS stores: published here, once (the unit barriers order LDS-DMA data only)

    for (long long g = g_begin; g < g_end; g += blk_per_cls) {
        const TilePoint tp = tile_of_group(a, g, G, wave, j);
        float px, py, pz;
        tile_point(a, tp.b, tp.nn, tp.valid, h, true, px, py, pz);
        InputTile it;
        input_tile_issue_volume(a, tp.b, 0, px, py, pz, h, it);
        // activation store: TB16 tile T = image * tiles_per_image + tile in image; an idle wave (a tile past the image's last) stores nothing
        const long long tile_in_image = (g - (long long)tp.b * G) * 4 + wave;
        const long long tile_T = (long long)tp.b * a.tiles_per_image + tile_in_image;
        const size_t slab16 = (size_t)a.total_tiles * NT * 1024;
        PwStore st;
        st.live = STORE && tile_in_image < a.tiles_per_image;
        store_live = st.live;
        st.blk_h = STORE ? reinterpret_cast<_Float16*>(a.act_h) + ((size_t)tile_T * NT * 32 + j) * 32 + 4 * h : nullptr;
        st.blk_c = STORE ? reinterpret_cast<_Float16*>(a.act_c) + ((size_t)tile_T * NT * 256 + lane) * 4 : nullptr;

        // ---- mapping hidden layer: m = LeakyReLU_0.2(Wm1 feat + bm1), 8 output tiles ----------------------------------------
        Split2 m[KCM];
        {
            const f16x8* unit = unit_begin(0);
            const f32x16 feat = input_tile_reduce(it, px, py, pz, h);
            float fv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) fv[r] = feat[r];
            if (STORE && st.live) {          // X operand of dWm1: the looked-up feature, clamped to fp16's range like the MFMA operand
                _Float16* fo = reinterpret_cast<_Float16*>(a.act_feat) + (((size_t)tile_T * 2 + 0) * 32 + j) * 32 + 4 * h;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float c4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) c4[e] = __builtin_amdgcn_fmed3f(fv[4 * gq + e], -65504.0f, 65504.0f);
                    *reinterpret_cast<u32x2_*>(fo + 8 * gq) = u32x2_{pk_f16(c4[0], c4[1]), pk_f16(c4[2], c4[3])};
                }
            }
            Split2 f2[2];
            f2[0] = split8_clamped(fv);
            f2[1] = split8_clamped(fv + 8);
            f32x16 accm[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) accm[t] = load_chan16(c_m1, t, h);
            input_unit<8>(unit, f2, accm, lane);
            const float inv_s = c_scal[0];
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float v0 = accm[t][r] * inv_s, v1 = accm[t][r + 1] * inv_s;
                    v0 = v0 > 0.0f ? v0 : v0 * 0.2f;
                    v1 = v1 > 0.0f ? v1 : v1 * 0.2f;
                    v0 = __builtin_amdgcn_fmed3f(v0, -65504.0f, 65504.0f);      // (unbounded, unlike sine outputs)
                    v1 = __builtin_amdgcn_fmed3f(v1, -65504.0f, 65504.0f);
                    split_into(&m[2 * t], r, v0, v1);
                    if (STORE) {                                                 // X operand of dWm2, and the sign for LeakyReLU'
                        if ((r & 2) == 0) {
                            st.s[0] = v0;
                            st.s[1] = v1;
                        } else if (st.live) {
                            _Float16* mo = reinterpret_cast<_Float16*>(a.act_h) + (size_t)a.L * slab16 + (((size_t)tile_T * 8 + t) * 32 + j) * 32 + 4 * h;
                            *reinterpret_cast<u32x2_*>(mo + 8 * (r >> 2)) = u32x2_{pk_f16(st.s[0], st.s[1]), pk_f16(v0, v1)};
                        }
                    }
                }
        }
        // ---- layer 0 reads the world position: pre_0 of all NT output tiles from one unit -----------------------------------------
        f32x16 acc0[NT];
        {
            const f16x8* unit = unit_begin(0);
            float fv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) fv[r] = 0.0f;
            if (h == 0) {                      // k = 0, 1, 2 live in lane half 0, elements 0..2 (the xyz tile of input_tile())
                fv[0] = px;
                fv[1] = py;
                fv[2] = pz;
            }
            if (STORE && st.live) {          // X operand of dW_0: the position in channels 0..2 of a 32-channel tile, zeros behind
                _Float16* fo = reinterpret_cast<_Float16*>(a.act_feat) + (((size_t)tile_T * 2 + 1) * 32 + j) * 32 + 4 * h;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<u32x2_*>(fo + 8 * gq) = gq == 0 ? u32x2_{pk_f16(fv[0], fv[1]), pk_f16(fv[2], 0.0f)} : u32x2_{0u, 0u};
            }
            Split2 f2[2];
            f2[0] = split8_clamped(fv);
#pragma unroll
            for (int k = 0; k < PARTS; ++k) f2[1].p[k] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < NT; ++t) acc0[t] = load_chan16(c_lay, t, h);
            input_unit<NT>(unit, f2, acc0, lane);
        }
        // ---- the FiLM layers ------------------------------------------------------------------------------------------------
        Split2 x[KCH], y[KCH];
        auto layer = [&](const Split2* in, Split2* out, auto first_tag, int l) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const float* ci = c_lay + (size_t)l * 3 * H;              // starting values: pre, fr, ph
            const float c1 = c_scal[2 + 4 * l], c2 = c_scal[3 + 4 * l];
            const float k2 = 30.0f / c_scal[4 + 4 * l];          // 2 S_f (a power of two): 15 fr + 30 = 15 (fr + 2), added to the finished accumulator
            f32x16 q_prev, ph_prev;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 fr = load_chan16(ci + H, t, h);
                // (stores since the request of this unit's copy, two units ago: none -- the epilogue of tile t-1, 8 stores, rides under THIS unit)
                fr = tile_kc<KCM, 6>(unit_begin(0), m, fr, lane, [&](int c) {
                    if (t > 0 && (c & 1)) {
                        if constexpr (STORE) film_pair_store(q_prev, ph_prev, c1, c2, t - 1, c - 1, &out[2 * (t - 1)], st);
                        else film_pair(q_prev, ph_prev, c1, c2, c - 1, &out[2 * (t - 1)]);
                    }
                });
                f32x16 pre;
                if constexpr (FIRST) {
                    pre = acc0[t];
                } else {
                    pre = load_chan16(ci, t, h);
                    pre = tile_kc<KCH, 0>(unit_begin(t > 0 ? 8 : 0), in, pre, lane, [](int) {});        // (tile t-1's epilogue rode under fr_t: 8 stores)
                }
                f32x16 ph = load_chan16(ci + 2 * H, t, h);
                f32x16 q;
                ph = tile_kc<KCM, 1>(unit_begin(t > 0 ? 8 : 0), m, ph, lane, [&](int c) { q[c] = (fr[c] + k2) * pre[c]; });
                q_prev = q;
                ph_prev = ph;
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if constexpr (STORE) film_pair_store(q_prev, ph_prev, c1, c2, NT - 1, r, &out[2 * (NT - 1)], st);
                else film_pair(q_prev, ph_prev, c1, c2, r, &out[2 * (NT - 1)]);
            }
            if (STORE) {
                st.blk_h += slab16;
                st.blk_c += 3 * slab16;
            }
        };
        layer(x, x, FirstLayer{}, 0);
        for (int l = 1; l < L; ++l) {
            layer(x, y, LaterLayer{}, l);
#pragma unroll
            for (int c = 0; c < KCH; ++c) x[c] = y[c];
        }
        // ---- head -------------------------------------------------------------------------------------------------------------
        {
            const f16x8* unit = unit_begin(8);                   // (the last layer's last epilogue sits in the interval before this one)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            acc = tile_kc<KCH, 0>(unit, x, acc, lane, [](int) {});
            if (tp.valid && h == 0) {
                const f32x4 hb = *reinterpret_cast<const f32x4*>(c_head);
                const float inv_s = c_scal[1];
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = __builtin_fmaf(acc[i], inv_s, hb[i]);
                if (a.flags & CNERF_F_SIGMOID_RGB) {
                    o[0] = sigmoidf_(o[0]);
                    o[1] = sigmoidf_(o[1]);
                    o[2] = sigmoidf_(o[2]);
                }
                *reinterpret_cast<f32x4*>(a.rgb_sigma + ((size_t)tp.b * a.n_per_image + tp.nn) * 4) = o;
            }
        }
    }
    wait_vmcnt<0>();                                                 // the copy issued for a tile this block does not have must not land
    __syncthreads();                                                 // after the block has given its LDS back
}

#if CNERF_H3_PARTS == 2
// ---------------------------------------------------------------------------------------------------------------
// pw_deriv_kernel: the two derivative rows the storing forward leaves out -- cos(arg) f and cos(arg) 15 pre per layer and point (COS16 slabs
// 3 l + 1, 3 l + 2) and the per-point maxima the chain scales its operands from -- out of what that forward stored: m and y_{l-1} (TB16
// rows, loaded straight into MFMA B fragments: a lane's eight channels of a k-chunk are two 8-byte pieces of its point's row) and cos(arg).
// One MFMA per 16 k-values on the LEADING fp16 part of the forward's own two-part weight stream (the hi pieces of a unit are every other
// KiB; same scales, same starting values): f and 15 pre are derivative factors about to be rounded to fp16, and the operands they come
// from are fp16 here (relative 6e-4 on 15 pre, 1e-4 on f).  Units used per tile: W_0 | layer 0: fr_t | layers >= 1: fr_t, pre_t; the
// epilogue of tile t-1 rides under fr_t.  HBM-bound: 8.5 KiB read + 8 KiB written per point and 8 layers.
// ---------------------------------------------------------------------------------------------------------------
template <int KC, typename PerChunk>
__device__ __forceinline__ f32x16 tile_hi(const f16x8* lds_tile, const u32x4* x, f32x16 acc, int lane, PerChunk per_chunk) {
    f16x8 ring[2] = {lds_tile[lane], lds_tile[64 + lane]};
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const f16x8 aw = ring[c & 1];
        if (c + 2 < KC) ring[c & 1] = lds_tile[(c + 2) * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, x[c]), acc, 0, 0, 0);
        per_chunk(c);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// B fragments of a point tile out of a TB16 matrix with CT channel tiles: chunk c = channels 16 c + 8 (jj >> 2) + 4 h + (jj & 3) of the lane's point
template <int NCH>
__device__ __forceinline__ void load_frags(const _Float16* row, int h, u32x4* x) {       // row = element (tile, channel tile 0, point j, 0)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const _Float16* p = row + (c >> 1) * 1024 + 16 * (c & 1) + 4 * h;
        const u32x2_ lo = *reinterpret_cast<const u32x2_*>(p), hi = *reinterpret_cast<const u32x2_*>(p + 8);
        x[c] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
}

template <int NT>
__global__ __launch_bounds__(256) void pw_deriv_kernel(FieldArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = NT * 32;
    constexpr int KCH = 2 * NT;
    constexpr int SLOT_FR = KCM * 64;                                // hi pieces only: at most 16 KiB per unit
    constexpr int SLOTS = 3;
    constexpr int PWMIN = KCH / 4;                                   // copy instructions per wave and unit: at least this many (KCH <= KCM)
    constexpr int CD = NT < 4 ? NT : 4;                              // cos(arg) rows: this many output tiles ahead (NT % CD == 0: the slot of a
                                                                     // tile is a compile-time number)
    f16x8* lds = reinterpret_cast<f16x8*>(smem);
    float* lds_c = reinterpret_cast<float*>(smem + SLOTS * (size_t)SLOT_FR * 16);
    const int L = a.L;
    const float* c_lay = lds_c + 256;
    const float* c_scal = c_lay + (size_t)3 * L * H + 4;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int j = lane & 31, h = lane >> 5;
    const long long G = (a.tiles_per_image + 3) / 4;
    const long long total_groups = (a.total_tiles / a.tiles_per_image) * G;
    const int nblk = gridDim.x;
    const int cls = blockIdx.x & 7, idx_in_cls = blockIdx.x >> 3;
    const int blk_per_cls = (nblk + 7 - cls) / 8;
    const long long g_begin = total_groups * cls / 8 + idx_in_cls, g_end = total_groups * (cls + 1) / 8;
    if (g_begin >= g_end) return;

    // the used units of a tile and where their hi pieces start in the two-part stream (1-KiB pieces; a unit of n k-chunk fragments is 2 n pieces)
    const int n_used = 1 + NT + (L - 1) * 2 * NT;
    const f16x8* w_units = reinterpret_cast<const f16x8*>(a.packed);
    int dma_k = 0, dma_slot = 0, use_slot = 0;
    auto dma_next = [&]() {
        const int k = dma_k;
        long long piece;                                             // first piece of the unit
        bool big = true;
        if (k == 0) {
            piece = 32;                                              // W_0, behind Wm1 (8 tiles x 2 chunks x 2 parts)
            big = false;
        } else if (k <= NT) {
            piece = 32 + 4 * NT + (long long)(k - 1) * 64;           // layer 0: per tile [fr | ph]
        } else {
            const int kk = k - 1 - NT, l1 = kk / (2 * NT), r = kk - l1 * 2 * NT, t = r >> 1;
            piece = 32 + 4 * NT + (long long)NT * 64 + ((long long)l1 * NT + t) * (64 + 4 * NT) + ((r & 1) ? 32 : 0);   // per tile [fr | W_l | ph]
            big = (r & 1) == 0;
        }
        const f16x8* src = w_units + piece * 64 + lane;             // hi piece i of the unit: 2 i KiB further on
        f16x8* dst = lds + dma_slot * SLOT_FR;
        if (big) {
#pragma unroll
            for (int q = 0; q < KCM / 4; ++q) {
                const int i = wave_u * (KCM / 4) + q;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * 128),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 64), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < KCH / 4; ++q) {
                const int i = wave_u * (KCH / 4) + q;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * 128),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 64), 16, 0, 0);
            }
        }
        dma_k = dma_k + 1 == n_used ? 0 : dma_k + 1;
        dma_slot = dma_slot == SLOTS - 1 ? 0 : dma_slot + 1;
    };
    // counted wait as in the forward: `younger` = vector-memory operations issued in the two intervals before (12 per epilogue: 4 loads, 8 stores)
    auto unit_begin = [&](int younger) -> const f16x8* {
        if (younger >= 12) wait_vmcnt<12 + PWMIN>();
        else wait_vmcnt<PWMIN>();
        lds_only_barrier();
        dma_next();
        const f16x8* unit = lds + use_slot * SLOT_FR;
        use_slot = use_slot == SLOTS - 1 ? 0 : use_slot + 1;
        return unit;
    };
    for (int i = threadIdx.x; i < a.bias_floats; i += 256) lds_c[i] = a.bias[i];
    dma_next();
    dma_next();
    __syncthreads();             // the constants are plain LDS stores: published here, once (the unit barriers order LDS-DMA data only)

    for (long long g = g_begin; g < g_end; g += blk_per_cls) {
        const int b = (int)(g / G);
        const long long tile_in_image = (g - (long long)b * G) * 4 + wave;
        const bool live = tile_in_image < a.tiles_per_image;
        const long long n = tile_in_image * 32 + j;
        const bool valid = live && n < a.n_per_image;
        const long long nn = n < a.n_per_image ? n : a.n_per_image - 1;
        const long long tile_T = (long long)b * a.tiles_per_image + (live ? tile_in_image : a.tiles_per_image - 1);   // (idle waves re-read a real tile, store nothing)
        const size_t slab16 = (size_t)a.total_tiles * NT * 1024;
        const _Float16* act_h = reinterpret_cast<const _Float16*>(a.act_h);
        _Float16* act_c = reinterpret_cast<_Float16*>(a.act_c);
        u32x4 m[KCM];
        load_frags<KCM>(act_h + (size_t)L * slab16 + ((size_t)tile_T * 8 * 32 + j) * 32, h, m);
        float px, py, pz;
        tile_point(a, b, nn, valid, h, false, px, py, pz);
        // cos(arg) rows, CD output tiles ahead over the linear sequence (layer, tile)
        f16x4 cosr[CD][4];
        auto fetch_cos = [&](int l, int t, int slot) {
            const _Float16* src = act_c + (size_t)(3 * l) * slab16 + (((size_t)tile_T * NT + t) * 256 + lane) * 4;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) cosr[slot][gq] = *reinterpret_cast<const f16x4*>(src + gq * 256);
        };
        auto prefetch_after = [&](int l, int t) {
            int tn = t + CD, ln = l;
            while (tn >= NT) {
                tn -= NT;
                ln += 1;
            }
            if (ln >= L) {
                ln = L - 1;
                tn = NT - 1;
            }
            fetch_cos(ln, tn, t % CD);
        };
#pragma unroll
        for (int t = 0; t < CD; ++t) fetch_cos(t / NT < L ? t / NT : L - 1, t % NT, t);

        // layer 0 reads the position
        f32x16 acc0[NT];
        {
            const f16x8* unit = unit_begin(0);
            float fv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) fv[r] = 0.0f;
            if (h == 0) {
                fv[0] = px;
                fv[1] = py;
                fv[2] = pz;
            }
            u32x4 f2[2];
            f2[0] = u32x4{pk_f16(fv[0], fv[1]), pk_f16(fv[2], fv[3]), 0u, 0u};
            f2[1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int t = 0; t < NT; ++t) acc0[t] = load_chan16(c_lay, t, h);
#pragma unroll
            for (int q = 0; q < 2 * NT; ++q)        // k_outer unit: fragment pair q = 2 t + c
                acc0[q >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(unit[q * 64 + lane], __builtin_bit_cast(f16x8, f2[q & 1]), acc0[q >> 1], 0, 0, 0);
        }
        u32x4 x[KCH];
        auto layer = [&](auto first_tag, int l) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const float* ci = c_lay + (size_t)l * 3 * H;
            const float kf = c_scal[4 + 4 * l], kp = c_scal[5 + 4 * l];
            const float k2 = 30.0f / kf;
            float amax = 1.0f;                                       // |cos| <= 1
            _Float16* dst = act_c + (size_t)(3 * l + 1) * slab16 + ((size_t)tile_T * NT * 256 + lane) * 4;
            f32x16 fr_prev, pre_prev;
            auto epi_quad = [&](const f32x16& fr, const f32x16& pre, int t, int gq) {       // quad gq of tile t: 4 consecutive channels
                float cf[4], cp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = (float)cosr[t % CD][gq][e];
                    cf[e] = __builtin_amdgcn_fmed3f(c * ((fr[4 * gq + e] + k2) * kf), -65504.0f, 65504.0f);
                    cp[e] = __builtin_amdgcn_fmed3f(c * (pre[4 * gq + e] * kp), -65504.0f, 65504.0f);
                    amax = fmaxf(amax, fmaxf(fabsf(cf[e]), fabsf(cp[e])));
                }
                if (live) {
                    _Float16* d = dst + (size_t)(t * 4 + gq) * 256;
                    *reinterpret_cast<u32x2_*>(d) = u32x2_{pk_f16(cf[0], cf[1]), pk_f16(cf[2], cf[3])};
                    *reinterpret_cast<u32x2_*>(d + slab16) = u32x2_{pk_f16(cp[0], cp[1]), pk_f16(cp[2], cp[3])};
                }
            };
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 fr = load_chan16(ci + H, t, h);
                // (the two intervals before: pre_{t-1} none, fr_{t-1} the epilogue of tile t-2 -- 12 operations)
                fr = tile_hi<KCM>(unit_begin(t >= 2 ? 12 : 0), m, fr, lane, [&](int c) {
                    if (t > 0 && (c & 3) == 3) epi_quad(fr_prev, pre_prev, t - 1, c >> 2);
                });
                if (t > 0) prefetch_after(l, t - 1);
                f32x16 pre;
                if constexpr (FIRST) {
                    pre = acc0[t];
                } else {
                    pre = load_chan16(ci, t, h);
                    pre = tile_hi<KCH>(unit_begin(t >= 1 ? 12 : 0), x, pre, lane, [](int) {});      // (fr_t held the epilogue of tile t-1)
                }
                fr_prev = fr;
                pre_prev = pre;
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) epi_quad(fr_prev, pre_prev, NT - 1, gq);
            prefetch_after(l, NT - 1);
            const float am = fmaxf(amax, __shfl_xor(amax, 32, WAVE));
            if (live && h == 0) a.act_amax[((size_t)l * a.total_tiles + tile_T) * 32 + j] = am;
        };
        layer(FirstLayer{}, 0);
        for (int l = 1; l < L; ++l) {
            load_frags<KCH>(act_h + (size_t)(l - 1) * slab16 + ((size_t)tile_T * NT * 32 + j) * 32, h, x);
            layer(LaterLayer{}, l);
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
}
#endif

constexpr size_t LDS_LIMIT = 160 * 1024;

template <int NT, bool STORE>
static hipError_t launch_inst(const FieldArgs& a, hipStream_t stream) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const size_t lds_bytes = 3 * (size_t)KCM * PARTS * 1024 + (size_t)a.bias_floats * 4;
    if (lds_bytes > LDS_LIMIT) return hipErrorInvalidValue;
    if (hipError_t e = hipFuncSetAttribute((const void*)field_pw16_kernel<NT, STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT)) return e;
    const long long want = (a.total_tiles / a.tiles_per_image) * ((a.tiles_per_image + 3) / 4);
    int blocks = (int)(want < cus ? want : cus);        // one block of four waves per CU (512 registers per wave)
    if (blocks < 8) blocks = 8;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((field_pw16_kernel<NT, STORE>), dim3(blocks), dim3(256), lds_bytes, stream, a);
    return hipGetLastError();
}

static hipError_t field_impl(const FieldArgs& a, int H, hipStream_t stream) {
    if (a.n_in != 1 || a.in_level[0] < 0 || a.L < 1) return hipErrorInvalidValue;      // one 32-channel volume tile
    const bool store = a.act_h != nullptr;             // activation-storing forward of the half-precision backward (fp16 tile blocks)
    if (store && (PARTS != 2 || !a.act_tb16 || !a.act_feat || !a.act_c || !a.act_amax)) return hipErrorInvalidValue;
    switch (H / 32) {
        case 2: return store ? launch_inst<2, true>(a, stream) : launch_inst<2, false>(a, stream);
        case 4: return store ? launch_inst<4, true>(a, stream) : launch_inst<4, false>(a, stream);
        case 8: return store ? launch_inst<8, true>(a, stream) : launch_inst<8, false>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

#if CNERF_H3_PARTS == 2
template <int NT>
static hipError_t launch_deriv_inst(const FieldArgs& a, hipStream_t stream) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const size_t lds_bytes = 3 * (size_t)KCM * 1024 + (size_t)a.bias_floats * 4;
    if (lds_bytes > LDS_LIMIT) return hipErrorInvalidValue;
    if (hipError_t e = hipFuncSetAttribute((const void*)pw_deriv_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT)) return e;
    const long long want = (a.total_tiles / a.tiles_per_image) * ((a.tiles_per_image + 3) / 4);
    int blocks = (int)(want < cus ? want : cus);
    if (blocks < 8) blocks = 8;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((pw_deriv_kernel<NT>), dim3(blocks), dim3(256), lds_bytes, stream, a);
    return hipGetLastError();
}
#endif

}  // namespace pw
}  // namespace H3_NS

hipError_t PW_LAUNCH_FIELD(const FieldArgs& a, int H, hipStream_t stream) {
    if (hipError_t e = H3_NS::pw::field_impl(a, H, stream)) return e;
#if CNERF_H3_PARTS == 2
    if (a.act_h) {      // activation-storing forward: the derivative rows the field kernel leaves out (same stream: ordered behind it)
        switch (H / 32) {
            case 2: return H3_NS::pw::launch_deriv_inst<2>(a, stream);
            case 4: return H3_NS::pw::launch_deriv_inst<4>(a, stream);
            case 8: return H3_NS::pw::launch_deriv_inst<8>(a, stream);
        }
    }
#endif
    return hipSuccess;
}

#if CNERF_H3_PARTS == 2
// ---------------------------------------------------------------------------------------------------------------
// The constants region behind the packed weight stream (one launch per packing, shared by both precisions): starting values of the
// accumulators and the epilogue's scalars from the raw biases and the 1 / S slots the packing kernels wrote.  inv_s: [Wm1 | per layer:
// W_l, Wm2 freq rows, Wm2 phase rows | head].
// ---------------------------------------------------------------------------------------------------------------
struct PwConstArgs {
    const float* bm1;
    const float* b[CNERF_MAX_LAYERS];
    const float* bm2;
    const float* b_head;
    const float* inv_s;
    float* out;
    int L, H;
};

__global__ void pw16_consts_kernel(PwConstArgs a) {
    const int LH3 = 3 * a.L * a.H;
    const int n_scal = (2 + 4 * a.L + 3) / 4 * 4;
    const int total = 256 + LH3 + 4 + n_scal;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (i < 256) {
            v = (float)((double)a.bm1[i] / (double)a.inv_s[0]);
        } else if (i < 256 + LH3) {
            const int q = i - 256, l = q / (3 * a.H), r = q - l * 3 * a.H, kind = r / a.H, ch = r - kind * a.H;
            const double is = (double)a.inv_s[1 + 3 * l + kind];
            if (kind == 0) v = (float)((double)a.b[l][ch] / is);
            else if (kind == 1) v = (float)((double)a.bm2[(size_t)l * a.H + ch] / is);
            else v = (float)((double)a.bm2[(size_t)(a.L + l) * a.H + ch] / is);
        } else if (i < 256 + LH3 + 4) {
            v = a.b_head[i - 256 - LH3];
        } else {
            const int s = i - 256 - LH3 - 4;
            const double inv_2pi = 0.15915494309189533577;
            if (s == 0) v = a.inv_s[0];
            else if (s == 1) v = a.inv_s[3 * a.L + 1];
            else if (s < 2 + 4 * a.L) {
                const int l = (s - 2) >> 2, w = (s - 2) & 3;
                if (w == 0) v = (float)(15.0 * (double)a.inv_s[2 + 3 * l] * (double)a.inv_s[1 + 3 * l] * inv_2pi);      // c1
                else if (w == 1) v = (float)((double)a.inv_s[3 + 3 * l] * inv_2pi);                                      // c2
                else if (w == 2) v = 15.0f * a.inv_s[2 + 3 * l];                                                         // accf -> f
                else v = 15.0f * a.inv_s[1 + 3 * l];                                                                     // accp -> 15 pre
            }
        }
        a.out[i] = v;
    }
}

hipError_t launch_pw16_consts(const cnerf_field_params* p, int L, int H, const float* inv_s, float* consts, hipStream_t stream) {
    PwConstArgs a;
    a.bm1 = p->map_b1;
    for (int l = 0; l < CNERF_MAX_LAYERS; ++l) a.b[l] = l < L ? p->b[l] : nullptr;
    a.bm2 = p->map_b2;
    a.b_head = p->b_final;
    a.inv_s = inv_s;
    a.out = consts;
    a.L = L;
    a.H = H;
    hipLaunchKernelGGL(pw16_consts_kernel, dim3(32), dim3(256), 0, stream, a);
    return hipGetLastError();
}
#endif

}  // namespace cnerf
