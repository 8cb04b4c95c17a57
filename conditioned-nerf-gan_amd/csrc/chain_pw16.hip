// Half-precision gradient chain of the per-point FiLM family (TALLSIREN, siren.py:232-331) on v_mfma_f32_32x32x16_f16: the counterpart of
// chain16_kernel (bwd16.hip) for networks whose frequencies and phases come per POINT from a mapping MLP.  With
//     m = LeakyReLU_0.2(Wm1 feat + bm1),   fr_l = Wm2[l H ..] m + ..,   ph_l = Wm2[(L + l) H ..] m + ..,   pre_l = W_l y_{l-1} + b_l,
//     y_l = sin(f_l pre_l + ph_l),  f_l = 15 fr_l + 30
// and g_y_l = d loss / d y_l, per layer (the storing forward, field_pw16.hip, left cos, cos f, cos 15 pre as fp16 rows):
//     g_ph = g_y cos      g_pre = g_y (cos f)      g_fr = g_y (cos 15 pre)          -> g16, three TB16 slabs per layer
//     g_y_{l-1} = W_l^T g_pre_l                                                      (l >= 1; the position carries no gradient)
//     g_m += Wm2[fr rows of l]^T g_fr_l + Wm2[ph rows of l]^T g_ph_l
// then g_mpre = g_m LeakyReLU'(m) -> g16 (TB16, 8 channel tiles), g_feat = Wm1^T g_mpre -> trilinear scatter into the gradient volume.
// The weight gradients are reductions over the stored slabs (weight_grad16_kernel): dW_l = g_pre_l^T y_{l-1}, dWm2 = [g_fr | g_ph]^T m,
// dWm1 = g_mpre^T feat, dW_head = go^T y_{L-1}.
//
// Two kernels, each within the register file.  (The first version did all of it in one: 128 resident accumulators of g_m next to two
// operand arrays, three derivative rows two tiles ahead and the epilogue's three stored quads -- 195 spilled dwords at H = 256, 64 GB of
// scratch traffic per launch, 31 ms; commit 9b8cfcc has it.)  One wave per 32-point tile, accumulator registers converted pairwise are the
// next B operands (channel = k) as in the forward kernels, weight units through a three-slot LDS ring by LDS-DMA, counted vmcnt waits.
#include "bwd16.hpp"
#include "cnerf_dev.hpp"
#include "cnerf_kernels.hpp"
#include "field_common.hpp"

namespace cnerf {
namespace pwchain {

__device__ __forceinline__ float pow2_to_2p14(float bound) {     // T = 2^(14 - e), bound = m 2^e, m in [0.5, 1); 0 / denormal / huge -> 1
    const int e = (int)((__float_as_uint(bound) >> 23) & 255u) - 126;
    const int te = 127 + 14 - e;
    return (bound >= 1e-30f && te > 0 && te < 255) ? __uint_as_float((uint32_t)te << 23) : 1.0f;
}

template <int PIECES>
__device__ __forceinline__ void dma_pieces(const f16x8* __restrict__ src, f16x8* lds_dst, int wave_u, int lane) {
    constexpr int PW = PIECES / 4;
    const f16x8* s0 = src + (size_t)wave_u * PW * 64 + lane;
    f16x8* d0 = lds_dst + wave_u * PW * 64;
#pragma unroll
    for (int q = 0; q < PW; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s0 + q * 64), (__attribute__((address_space(3))) void*)(d0 + q * 64), 16,
                                         0, 0);
}

struct RescaleYes { static constexpr bool value = true; };
struct RescaleNo { static constexpr bool value = false; };

// chain_pre_kernel   propagates g_y through W_l^T only -- operand g_y (cos f) -- and stores g_y itself, one TB16 slab per layer with its own
//                    scale (DRY instantiation: sampled maxima); one derivative row to load, nothing resident but the two operand arrays.
// pw_gm_kernel       per layer and channel tile: g_y (fp16) x (cos, cos f, cos 15 pre) -> the three stored gradient slabs (scale = g_y's
//                    x 2^-ceil(log2 largest derivative of the layer): no overflow by construction) and the operands of the mapping products
//                    (M(l, t): the 4 k-chunks of tile t -- 2 of g_fr, 2 of g_ph -- against all 8 output tiles, one 32-KiB unit, so the operands of
//                    a tile are consumed as soon as its epilogue has produced them), g_m resident in 8 accumulator tiles and moved to the next
//                    layer's units tile by tile, then the tail (g_mpre, Wm1^T, volume scatter).  Operands carry per-LAYER scales here (the stored
//                    g_y has no per-point one): a point whose gradient is 2^-20 of the largest contributes nothing to g_m -- or to any sum over
//                    points.  The stored slabs see two fp16 roundings (g_y, then the product) instead of one.
struct PreArgs {
    FieldArgs f;
    const f16x8* units;       // Y units: stages L-2 .. 0, NT units of KCH pieces each
    const f16x8* head_t;
    const float* winv;        // 1 / s: [W_l^T: L (entry 0 unused) | Wm2 pair of layer l: L | Wm1^T | head^T]
    const float* anorm;       // ||W_l||_1: L (entry 0 unused), then the head's
    const float* scales;      // {S, 1 / S}: index 3 L + 1: go; 3 L + 2 + l: g_y of layer l
    const _Float16* cos16;    // 3 L COS16 slabs; this kernel reads cos f (slab 3 l + 1)
    const float* amax;
    _Float16* gy16;           // TB16: L slabs (tiles, NT, 32, 32)
    _Float16* go16;
    unsigned int* gmax;       // dry run: 3 L + 1: go, 3 L + 2 + l: max |g_y_l|
    unsigned int* sat;
    int group_step;
};

template <int NT, bool DRY>
__global__ __launch_bounds__(256) void chain_pre_kernel(PreArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_q[];
    const FieldArgs& a = A.f;
    constexpr int KCH = 2 * NT;
    constexpr int SLOT_FR = KCH * 64;
    constexpr int SLOTS = 3;
    constexpr int PW = KCH / 4;
    constexpr int CD = 2;
    f16x8* lds_units = reinterpret_cast<f16x8*>(smem_q);
    f16x8* lds_head = lds_units + SLOTS * SLOT_FR;
    const int L = a.L;
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
    if (g_begin >= g_end) return;
    const int n_units = (L - 1) * NT;                   // 0 for a one-layer network: no unit is ever requested
    int dma_k = 0, dma_slot = 0, use_slot = 0;
    auto dma_next = [&]() {
        if (n_units == 0) return;
        dma_pieces<KCH>(A.units + (size_t)dma_k * SLOT_FR, lds_units + dma_slot * SLOT_FR, wave_u, lane);
        dma_k = dma_k + 1 == n_units ? 0 : dma_k + 1;
        dma_slot = dma_slot == SLOTS - 1 ? 0 : dma_slot + 1;
    };
    // counted wait: since the request of this unit's copy, two units ago, every wave has issued the copy of the unit in
    // between (PW) and `younger` more (the prefetch of a derivative row: 4 loads per output tile)
    auto unit_begin = [&](int younger) -> const f16x8* {
        if (younger >= 4) wait_vmcnt<4 + PW>();
        else wait_vmcnt<PW>();
        lds_only_barrier();
        dma_next();
        const f16x8* u = lds_units + use_slot * SLOT_FR;
        use_slot = use_slot == SLOTS - 1 ? 0 : use_slot + 1;
        return u;
    };
    for (int i = threadIdx.x; i < NT * 64; i += 256) lds_head[i] = A.head_t[i];
    dma_next();
    dma_next();
    __syncthreads();                                    // the head fragments are plain LDS stores: published here, once
    const float winv_head = A.winv[2 * L + 1];
    const float S_go = A.scales[2 * (3 * L + 1)];

    for (long long g = g_begin; g < g_end; g += g_stride) {
        const int b = (int)(g / G);
        const long long tile_in_image = (g - (long long)b * G) * 4 + wave;
        const bool live = tile_in_image < a.tiles_per_image;
        const long long n = tile_in_image * 32 + j;
        const bool valid = live && n < a.n_per_image;
        const long long nn = n < a.n_per_image ? n : a.n_per_image - 1;
        const size_t gpt = (size_t)b * a.n_per_image + nn;
        const long long tile_T = (long long)b * a.tiles_per_image + (live ? tile_in_image : a.tiles_per_image - 1);
        const size_t slab16 = (size_t)a.total_tiles * NT * 1024;
        const size_t row16 = ((size_t)tile_T * NT * 32 + j) * 32;
        f16x4 cfr[CD][4];                                // cos f rows, CD tiles ahead over (stage L-1 .. 0) x (tile 0 .. NT-1)
        auto fetch_row = [&](int lam, int t, int slot) {
            const _Float16* src = A.cos16 + (size_t)(3 * lam + 1) * slab16 + (((size_t)tile_T * NT + t) * 256 + lane) * 4;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) cfr[slot][gq] = *reinterpret_cast<const f16x4*>(src + gq * 256);
        };
        auto prefetch_after = [&](int lam, int t) {
            int tn = t + CD, ln = lam;
            if (tn >= NT) {
                tn -= NT;
                ln -= 1;
            }
            if (ln < 0) {
                ln = 0;
                tn = NT - 1;
            }
            fetch_row(ln, tn, t % CD);
        };
#pragma unroll
        for (int t = 0; t < CD; ++t) fetch_row(L - 1, t, t);

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
            if (lane == 0) atomicMax(A.gmax + 3 * L + 1, __float_as_uint(m4));
        } else if (live && h == 0) {
            *reinterpret_cast<u32x2_*>(A.go16 + ((size_t)tile_T * 32 + j) * 32) = u32x2_{pk_f16(gs[0], gs[1]), pk_f16(gs[2], gs[3])};
        }

        u32x4 frag_in[KCH], frag_out[KCH];
        float T_prev = 1.0f, gpremax_prev = 0.0f;
        for (int lam = L - 1; lam >= 0; --lam) {
            const float am = A.amax[((size_t)lam * a.total_tiles + tile_T) * 32 + j];
            const bool first = lam == L - 1;
            const float U = first ? winv_head / S_go : A.winv[lam + 1] / T_prev;            // accumulator -> true g_y
            const float T = pow2_to_2p14(am * (first ? A.anorm[L] * gomax : A.anorm[lam + 1] * gpremax_prev));
            const float UT = U * T, USy = U * A.scales[2 * (3 * L + 2 + lam)];
            _Float16* gdst = A.gy16 + (size_t)lam * slab16 + row16 + 4 * h;
            float mpre = 0.0f, my = 0.0f, s4[4], epre = 0.0f;
            auto epi = [&](const f32x16& z, int t, int r) {
                const int gq = r >> 2, e = r & 3;
                const float av = z[r];
                const float vpre = av * (float)cfr[t % CD][gq][e];
                mpre = fmaxf(mpre, fabsf(vpre));
                my = fmaxf(my, fabsf(av));
                if (!DRY) {
                    s4[e] = __builtin_amdgcn_fmed3f(av * USy, -65504.0f, 65504.0f);
                    if (e == 3 && live) *reinterpret_cast<u32x2_*>(gdst + t * 1024 + 8 * gq) = u32x2_{pk_f16(s4[0], s4[1]), pk_f16(s4[2], s4[3])};
                }
                const float op = vpre * UT;
                if ((r & 1) == 0) epre = op;
                else frag_out[2 * t + (r >> 3)][(r & 7) >> 1] = pk_f16(epre, op);
            };
            f32x16 z_prev;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                if (first) {
                    const u32x4 bh = h == 0 ? u32x4{pk_f16(gs[0], gs[1]), pk_f16(gs[2], gs[3]), 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
                    const u32x4 bl = h == 0 ? u32x4{pk_f16(gl[0], gl[1]), pk_f16(gl[2], gl[3]), 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
                    const f16x8 aw = lds_head[t * 64 + lane];
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, bl), z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, bh), z, 0, 0, 0);
                    if (t > 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) epi(z_prev, t - 1, r);
                        prefetch_after(lam, t - 1);
                    }
                } else {
                    // (the two intervals before hold the prefetch behind epilogue(t-2) / the last one of the stage above: >= 4 loads)
                    const f16x8* unit = unit_begin(4) + lane;
                    f16x8 ring[2] = {unit[0], unit[64]};
#pragma unroll
                    for (int c = 0; c < KCH; ++c) {
                        const f16x8 aw = ring[c & 1];
                        if (c + 2 < KCH) ring[c & 1] = unit[(c + 2) * 64];
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, frag_in[c]), z, 0, 0, 0);
                        if (t > 0) {
#pragma unroll
                            for (int q = 0; q < 16 / KCH; ++q) epi(z_prev, t - 1, c * (16 / KCH) + q);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 10 * (16 / KCH), 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (t > 0) prefetch_after(lam, t - 1);
                }
                z_prev = z;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) epi(z_prev, NT - 1, r);
            prefetch_after(lam, NT - 1);
            const float mp = fmaxf(mpre, __shfl_xor(mpre, 32, WAVE));
            if (DRY) {
                float m4 = my * U;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) m4 = fmaxf(m4, __shfl_xor(m4, d, WAVE));
                if (lane == 0) atomicMax(A.gmax + 3 * L + 2 + lam, __float_as_uint(m4));
            } else if (A.sat && live && __any(my * USy > 65504.0f) && lane == 0) {
                atomicAdd(A.sat, 1u);
            }
            gpremax_prev = mp * U;
            T_prev = T;
#pragma unroll
            for (int c = 0; c < KCH; ++c) frag_in[c] = frag_out[c];
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
}

struct GmArgs {
    FieldArgs f;
    const f16x8* units;       // M(l, t): (l NT + t) x 32 pieces, then Wm1^T (16 pieces)
    const float* winv;
    const float* scales;      // {S, 1 / S}: 3 l + s: the three stored slabs of layer l; 3 L: g_mpre; 3 L + 2 + l: g_y
    const float* lay;         // per layer {r_l, To_l}: stored slab = g_y16 x derivative x r_l, operand = ... x To_l
    const _Float16* gy16;
    const _Float16* cos16;
    const _Float16* m16;
    _Float16* g16;
    unsigned int* gmax;       // dry run: 3 L: max |g_mpre|
    unsigned int* sat;
    int group_step;
};

template <int NT, bool DRY>
__global__ __launch_bounds__(256) void pw_gm_kernel(GmArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    const FieldArgs& a = A.f;
    constexpr int SLOT_FR = 32 * 64;
    constexpr int SLOTS = 3;
    constexpr int CD = 2;
    f16x8* lds_units = reinterpret_cast<f16x8*>(smem_g);
    char* lds_wave = smem_g + SLOTS * SLOT_FR * 16 + (threadIdx.x >> 6) * 6400;
    const int L = a.L;
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
    if (g_begin >= g_end) return;
    const int n_units = L * NT + 1;
    int dma_k = 0, dma_slot = 0, use_slot = 0;
    auto dma_next = [&]() {
        f16x8* dst = lds_units + dma_slot * SLOT_FR;
        if (dma_k == n_units - 1) dma_pieces<16>(A.units + (size_t)dma_k * SLOT_FR, dst, wave_u, lane);
        else dma_pieces<32>(A.units + (size_t)dma_k * SLOT_FR, dst, wave_u, lane);
        dma_k = dma_k + 1 == n_units ? 0 : dma_k + 1;
        dma_slot = dma_slot == SLOTS - 1 ? 0 : dma_slot + 1;
    };
    // counted wait: the two intervals before hold >= `younger` operations of this wave besides the copy in between (4 pieces at least)
    auto unit_begin = [&](int younger) -> const f16x8* {
        if (younger >= 16) wait_vmcnt<16 + 4>();
        else wait_vmcnt<4>();
        lds_only_barrier();
        dma_next();
        const f16x8* u = lds_units + use_slot * SLOT_FR;
        use_slot = use_slot == SLOTS - 1 ? 0 : use_slot + 1;
        return u;
    };
    dma_next();
    dma_next();
    const float* winvM = A.winv + L;
    const float winvW1 = A.winv[2 * L];

    for (long long g = g_begin; g < g_end; g += g_stride) {
        const int b = (int)(g / G);
        const long long tile_in_image = (g - (long long)b * G) * 4 + wave;
        const bool live = tile_in_image < a.tiles_per_image;
        const long long n = tile_in_image * 32 + j;
        const bool valid = live && n < a.n_per_image;
        const long long nn = n < a.n_per_image ? n : a.n_per_image - 1;
        const long long tile_T = (long long)b * a.tiles_per_image + (live ? tile_in_image : a.tiles_per_image - 1);
        const size_t slab16 = (size_t)a.total_tiles * NT * 1024;
        const size_t row16 = ((size_t)tile_T * NT * 32 + j) * 32 + 4 * h;
        // rows of a (layer, tile): g_y quads (TB16) and the three derivative quads (COS16), CD tiles ahead over (layer 0 .. L-1) x (tile 0 .. NT-1)
        f16x4 rows[CD][4][4];
        auto fetch_rows = [&](int l, int t, int slot) {
            const _Float16* gsrc = A.gy16 + (size_t)l * slab16 + row16 + t * 1024;
            const _Float16* csrc = A.cos16 + (size_t)(3 * l) * slab16 + (((size_t)tile_T * NT + t) * 256 + lane) * 4;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                rows[slot][0][gq] = *reinterpret_cast<const f16x4*>(gsrc + 8 * gq);
#pragma unroll
                for (int s = 0; s < 3; ++s) rows[slot][1 + s][gq] = *reinterpret_cast<const f16x4*>(csrc + s * slab16 + gq * 256);
            }
        };
        auto prefetch_after = [&](int l, int t) {
            int tn = t + CD, ln = l;
            if (tn >= NT) {
                tn -= NT;
                ln += 1;
            }
            if (ln >= L) {
                ln = L - 1;
                tn = NT - 1;
            }
            fetch_rows(ln, tn, t % CD);
        };
#pragma unroll
        for (int t = 0; t < CD; ++t) fetch_rows(0, t, t);
        f32x16 gm[8];
#pragma unroll
        for (int ot = 0; ot < 8; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) gm[ot][r] = 0.0f;
        float unit_prev = 1.0f;                          // gm = true g_m x unit_prev
        auto m_product = [&](const f16x8* unit, const u32x4* ffr, const u32x4* fph, auto rescale_tag, float rho, auto ride) {
            constexpr bool RESCALE = decltype(rescale_tag)::value;
            f16x8 ring[2] = {unit[lane], unit[64 + lane]};
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int cc = i >> 3, ot = i & 7;
                const f16x8 aw = ring[i & 1];
                if (i + 2 < 32) ring[i & 1] = unit[(i + 2) * 64 + lane];
                if (RESCALE && i < 8) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) gm[ot][r] *= rho;
                }
                const u32x4 bq = cc < 2 ? ffr[cc] : fph[cc - 2];
                gm[ot] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, bq), gm[ot], 0, 0, 0);
                ride(i);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int l = 0; l < L; ++l) {
            const float r_l = A.lay[2 * l], To = A.lay[2 * l + 1];
            const float unit_l = A.scales[2 * (3 * L + 2 + l)] * To / winvM[l];            // accumulator units of this layer's products
            const float rho = l == 0 ? 1.0f : unit_l / unit_prev;
            unit_prev = unit_l;
            _Float16* gdst = A.g16 + (size_t)(3 * l) * slab16 + row16;
            u32x4 ffr[2][2], fph[2][2];
            float s3[3][4], ofr_e = 0.0f, oph_e = 0.0f;
            auto epi = [&](int t, int r, u32x4* nfr, u32x4* nph) {
                const int gq = r >> 2, e = r & 3;
                const float gy = (float)rows[t % CD][0][gq][e];
                const float vph = gy * (float)rows[t % CD][1][gq][e], vpre = gy * (float)rows[t % CD][2][gq][e], vfr = gy * (float)rows[t % CD][3][gq][e];
                if (!DRY) {
                    s3[0][e] = vpre * r_l;
                    s3[1][e] = vfr * r_l;
                    s3[2][e] = vph * r_l;
                    if (e == 3 && live) {
                        _Float16* d = gdst + t * 1024 + 8 * gq;
#pragma unroll
                        for (int s = 0; s < 3; ++s)
                            *reinterpret_cast<u32x2_*>(d + s * slab16) = u32x2_{pk_f16(s3[s][0], s3[s][1]), pk_f16(s3[s][2], s3[s][3])};
                    }
                }
                const float ofr = vfr * To, oph = vph * To;
                if ((r & 1) == 0) {
                    ofr_e = ofr;
                    oph_e = oph;
                } else {
                    nfr[r >> 3][(r & 7) >> 1] = pk_f16(ofr_e, ofr);
                    nph[r >> 3][(r & 7) >> 1] = pk_f16(oph_e, oph);
                }
            };
#pragma unroll
            for (int r = 0; r < 16; ++r) epi(0, r, ffr[0], fph[0]);
            prefetch_after(l, 0);
#pragma unroll
            for (int t = 1; t <= NT; ++t) {
                // (the two intervals before: the epilogue of tile t-1 -- 12 stores -- and its prefetch of 16 rows; t = 1: the plain epilogue above)
                const f16x8* unit = unit_begin(16);
                auto ride = [&](int i) {
                    if (t < NT && (i & 1)) epi(t < NT ? t : 0, i >> 1, ffr[t & 1], fph[t & 1]);
                };
                if (t == 1) m_product(unit, ffr[0], fph[0], RescaleYes{}, rho, ride);
                else m_product(unit, ffr[(t - 1) & 1], fph[(t - 1) & 1], RescaleNo{}, 1.0f, ride);
                if (t < NT) prefetch_after(l, t);
            }
        }
        // ---- mapping network: g_mpre = g_m LeakyReLU'(m) (its operand scale from the point's exact maximum), g_feat = Wm1^T g_mpre ---------------
        const float Um = 1.0f / unit_prev;
        const _Float16* mrow = A.m16 + ((size_t)tile_T * 8 * 32 + j) * 32 + 4 * h;
        auto mpre_of = [&](int ot, int gq, f32x4& v) {
            const f16x4 mq = *reinterpret_cast<const f16x4*>(mrow + ot * 1024 + 8 * gq);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gm[ot][4 * gq + e] * Um * ((float)mq[e] > 0.0f ? 1.0f : 0.2f);
        };
        float mx = 0.0f;
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 v;
                mpre_of(ot, gq, v);
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
        const float S_mp = A.scales[2 * (3 * L)];
        if (DRY) {
            float m4 = mx;
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) m4 = fmaxf(m4, __shfl_xor(m4, d, WAVE));
            if (lane == 0) atomicMax(A.gmax + 3 * L, __float_as_uint(m4));
        } else if (A.sat && live && __any(mx * S_mp > 65504.0f) && lane == 0) {
            atomicAdd(A.sat, 1u);
        }
        const float Tm = pow2_to_2p14(mx);
        u32x4 fm[16];
        _Float16* gmp_dst = A.g16 + (size_t)(3 * L) * slab16 + ((size_t)tile_T * 8 * 32 + j) * 32 + 4 * h;
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 v;
                mpre_of(ot, gq, v);
                if (!DRY && live) {
                    float q4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) q4[e] = __builtin_amdgcn_fmed3f(v[e] * S_mp, -65504.0f, 65504.0f);
                    *reinterpret_cast<u32x2_*>(gmp_dst + ot * 1024 + 8 * gq) = u32x2_{pk_f16(q4[0], q4[1]), pk_f16(q4[2], q4[3])};
                }
                fm[2 * ot + (gq >> 1)][2 * (gq & 1)] = pk_f16(v[0] * Tm, v[1] * Tm);
                fm[2 * ot + (gq >> 1)][2 * (gq & 1) + 1] = pk_f16(v[2] * Tm, v[3] * Tm);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const f16x8* unit = unit_begin(0) + lane;
            if (!DRY) {
                f32x16 zf;
#pragma unroll
                for (int r = 0; r < 16; ++r) zf[r] = 0.0f;
#pragma unroll
                for (int c = 0; c < 16; ++c) zf = __builtin_amdgcn_mfma_f32_32x32x16_f16(unit[c * 64], __builtin_bit_cast(f16x8, fm[c]), zf, 0, 0, 0);
                const float U0 = winvW1 / Tm;
                float px, py, pz;
                tile_point(a, b, nn, valid, h, false, px, py, pz);
                const int ch = lane & 31;
                float* sg = reinterpret_cast<float*>(lds_wave);
                int* sb = reinterpret_cast<int*>(sg + 32 * 33);
                float* sw = reinterpret_cast<float*>(sb + 32 * 8);
                const int V = a.lvl_V[0], C = a.lvl_C[0];
                Corner8 cr;
                trilinear_corners(px, py, pz, a.half_voxel, V, cr);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sg[j * 33 + 8 * gq + 4 * h + e] = zf[4 * gq + e] * U0;
                if (h == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        sb[j * 8 + k] = cr.base[k];
                        sw[j * 8 + k] = valid ? cr.w[k] : 0.0f;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                float* gv = a.lvl_grad[0] + (size_t)b * V * V * V * C;
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
    wait_vmcnt<0>();
    __syncthreads();
}

// per layer: r_l = 2^-ceil(log2 max |derivative|), To_l = 8 r_l; the three stored slabs' scales = g_y's x r_l
__global__ void pw_split_scales_kernel(const uint32_t* amaxg_bits, int L, float* scales, float* lay) {
    const int l = threadIdx.x;
    if (l >= L) return;
    float am = __uint_as_float(amaxg_bits[l]);
    if (!(am >= 1.0f) || !(am < 3e38f)) am = 1.0f;
    int e;
    const float mant = frexpf(am, &e);                  // am = mant 2^e, mant in [0.5, 1)
    const int ce = mant == 0.5f ? e - 1 : e;             // ceil(log2 am)
    const float r = ldexpf(1.0f, -ce);
    lay[2 * l] = r;
    lay[2 * l + 1] = 8.0f * r;
    const float Sy = scales[2 * (3 * L + 2 + l)];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        scales[2 * (3 * l + s)] = Sy * r;
        scales[2 * (3 * l + s) + 1] = 1.0f / (Sy * r);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// packing.  Element jj of lane (i = lane & 31, hh = lane >> 5) of a fragment = s * (transposed weight)[row 32 t' + i][k], k in the order
// in which accumulator registers become B operands: 16 c + 8 (jj >> 2) + 4 hh + (jj & 3) (bwd16.hip).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow2_weight_scale(uint32_t wmax_bits) {
    const float wmax = __uint_as_float(wmax_bits);
    if (!(wmax > 1e-30f) || !(wmax < 3e38f)) return 1.0f;
    int e;
    (void)frexpf(16384.0f / wmax, &e);
    return ldexpf(1.0f, e - 1 > 100 ? 100 : e - 1);
}

// Y(l, t) for all t: rows [32 t, 32 t + 32) of W_l^T (W_l row-major (H, H)), KCH k-chunks
__global__ void pack_y_kernel(const float* __restrict__ w, int H, int NT, const uint32_t* wmax_slot, float* winv_slot, _Float16* __restrict__ stage_dst) {
    const float s = pow2_weight_scale(*wmax_slot);
    if (blockIdx.x == 0 && threadIdx.x == 0) *winv_slot = 1.0f / s;
    const int KCH = 2 * NT;
    const long long total = (long long)NT * KCH * 512;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int jj = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int c = (int)((idx >> 9) % KCH), t = (int)((idx >> 9) / KCH);
        const int r = 32 * t + (lane & 31), k = 16 * c + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3);
        stage_dst[(size_t)t * KCH * 512 + ((size_t)c * 64 + lane) * 8 + jj] = (_Float16)(w[(size_t)k * H + r] * s);
    }
}

// M(l, t) for all t: fragment (cc, ot), cc = 0, 1: freq rows, 2, 3: phase rows of the layer; k = channel 32 t + 16 (cc & 1) + ..; output row
// 32 ot + i of the 256 inputs of the mapping network's second Linear (Wm2 row-major (2 L H, 256))
__global__ void pack_m_kernel(const float* __restrict__ wm2, int H, int NT, int row_f, int row_p, const uint32_t* wmax_slot, float* winv_slot,
                              _Float16* __restrict__ stage_dst) {
    const float s = pow2_weight_scale(*wmax_slot);
    if (blockIdx.x == 0 && threadIdx.x == 0) *winv_slot = 1.0f / s;
    const long long total = (long long)NT * 32 * 512;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int jj = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int f = (int)((idx >> 9) & 31), t = (int)(idx >> 14);
        const int cc = f >> 3, ot = f & 7;
        const int chan = 32 * t + 16 * (cc & 1) + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3);
        const int row = (cc < 2 ? row_f : row_p) + chan;
        stage_dst[(size_t)t * 32 * 512 + ((size_t)f * 64 + lane) * 8 + jj] = (_Float16)(wm2[(size_t)row * 256 + 32 * ot + (lane & 31)] * s);
    }
}

template <int NT, bool DRY>
static hipError_t launch_pre_nt(const PreArgs& A, hipStream_t stream) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const size_t lds_bytes = (size_t)3 * 2 * NT * 1024 + (size_t)NT * 1024;
    if (hipError_t e = hipFuncSetAttribute((const void*)chain_pre_kernel<NT, DRY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) return e;
    const FieldArgs& f = A.f;
    const long long want = (f.total_tiles / f.tiles_per_image) * ((f.tiles_per_image + 3) / 4);
    int blocks = (int)(want < cus ? want : cus);
    if (blocks < 8) blocks = 8;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((chain_pre_kernel<NT, DRY>), dim3(blocks), dim3(256), lds_bytes, stream, A);
    return hipGetLastError();
}

template <int NT, bool DRY>
static hipError_t launch_gm_nt(const GmArgs& A, hipStream_t stream) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const size_t lds_bytes = (size_t)3 * 32 * 1024 + (size_t)4 * 6400;
    if (hipError_t e = hipFuncSetAttribute((const void*)pw_gm_kernel<NT, DRY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) return e;
    const FieldArgs& f = A.f;
    const long long want = (f.total_tiles / f.tiles_per_image) * ((f.tiles_per_image + 3) / 4);
    int blocks = (int)(want < cus ? want : cus);
    if (blocks < 8) blocks = 8;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((pw_gm_kernel<NT, DRY>), dim3(blocks), dim3(256), lds_bytes, stream, A);
    return hipGetLastError();
}

}  // namespace pwchain

hipError_t launch_chain_pre(const FieldArgs& f, int H, const PwChainBuffers& c, int dry, int group_step, hipStream_t stream) {
    if (f.n_in != 1 || f.in_level[0] < 0 || f.L < 1 || 4 * f.L + 2 > 64) return hipErrorInvalidValue;
    pwchain::PreArgs A;
    A.f = f;
    A.units = (const f16x8*)c.units_y;
    A.head_t = (const f16x8*)c.head_t;
    A.winv = c.winv;
    A.anorm = c.anorm;
    A.scales = c.scales;
    A.cos16 = (const _Float16*)c.cos16;
    A.amax = c.amax;
    A.gy16 = (_Float16*)c.gy16;
    A.go16 = (_Float16*)c.go16;
    A.gmax = c.gmax;
    A.sat = c.sat;
    A.group_step = group_step < 1 ? 1 : group_step;
    switch (H / 32) {
        case 2: return dry ? pwchain::launch_pre_nt<2, true>(A, stream) : pwchain::launch_pre_nt<2, false>(A, stream);
        case 4: return dry ? pwchain::launch_pre_nt<4, true>(A, stream) : pwchain::launch_pre_nt<4, false>(A, stream);
        case 8: return dry ? pwchain::launch_pre_nt<8, true>(A, stream) : pwchain::launch_pre_nt<8, false>(A, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pw_gm(const FieldArgs& f, int H, const PwChainBuffers& c, int dry, int group_step, hipStream_t stream) {
    pwchain::GmArgs A;
    A.f = f;
    A.units = (const f16x8*)c.units_m;
    A.winv = c.winv;
    A.scales = c.scales;
    A.lay = c.lay;
    A.gy16 = (const _Float16*)c.gy16;
    A.cos16 = (const _Float16*)c.cos16;
    A.m16 = (const _Float16*)c.m16;
    A.g16 = (_Float16*)c.g16;
    A.gmax = c.gmax;
    A.sat = c.sat;
    A.group_step = group_step < 1 ? 1 : group_step;
    switch (H / 32) {
        case 2: return dry ? pwchain::launch_gm_nt<2, true>(A, stream) : pwchain::launch_gm_nt<2, false>(A, stream);
        case 4: return dry ? pwchain::launch_gm_nt<4, true>(A, stream) : pwchain::launch_gm_nt<4, false>(A, stream);
        case 8: return dry ? pwchain::launch_gm_nt<8, true>(A, stream) : pwchain::launch_gm_nt<8, false>(A, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pw_split_scales(const uint32_t* amaxg_bits, int L, float* scales, float* lay, hipStream_t stream) {
    hipLaunchKernelGGL(pwchain::pw_split_scales_kernel, dim3(1), dim3(64), 0, stream, amaxg_bits, L, scales, lay);
    return L <= 64 ? hipGetLastError() : hipErrorInvalidValue;
}

// Y units (stages L-2 .. 0, NT x KCH pieces each) | M units (layers 0 .. L-1, NT x 32 pieces each) + Wm1^T (16 pieces)
size_t pw_chain_y_bytes(int L, int H) { return (size_t)(L > 1 ? L - 1 : 0) * (H / 32) * (2 * (H / 32)) * 1024; }
size_t pw_chain_m_bytes(int L, int H) { return ((size_t)L * (H / 32) * 32 + 16) * 1024; }

// both streams; winv: [W_l^T: L (0 unused) | Wm2 pair of layer l: L | Wm1^T | head^T], anorm: [||W_l||_1: L (0 unused) | head], wmax: 2 L + 2 scratch words
hipError_t launch_pack_pw_chain(const cnerf_field_params* p, int L, int H, void* units_y, void* units_m, void* head_t, float* winv, float* anorm,
                                uint32_t* wmax, hipStream_t stream) {
    const int NT = H / 32, KCH = 2 * NT;
    if (hipError_t e = hipMemsetAsync(wmax, 0, (size_t)(2 * L + 2) * sizeof(uint32_t), stream)) return e;
    if (hipError_t e = hipMemsetAsync(winv, 0, (size_t)(2 * L + 2) * sizeof(float), stream)) return e;
    if (hipError_t e = hipMemsetAsync(anorm, 0, (size_t)(L + 1) * sizeof(float), stream)) return e;
    const size_t LH = (size_t)L * H;
    for (int l = 1; l < L; ++l)
        if (hipError_t e = launch_absmax_bits(p->w[l], (long long)H * H, wmax + l, stream)) return e;
    for (int l = 0; l < L; ++l) {       // the freq rows and the phase rows of a layer share one scale: they accumulate into the same registers
        if (hipError_t e = launch_absmax_bits(p->map_w2 + (size_t)l * H * 256, (long long)H * 256, wmax + L + l, stream)) return e;
        if (hipError_t e = launch_absmax_bits(p->map_w2 + (LH + (size_t)l * H) * 256, (long long)H * 256, wmax + L + l, stream)) return e;
    }
    _Float16* dy = (_Float16*)units_y;
    for (int lam = L - 2; lam >= 0; --lam) {             // consumption order of chain_pre_kernel: the stage that produces g_y of layer lam multiplies by W_{lam+1}^T
        hipLaunchKernelGGL(pwchain::pack_y_kernel, dim3(64), dim3(256), 0, stream, p->w[lam + 1], H, NT, (const uint32_t*)(wmax + lam + 1), winv + lam + 1, dy);
        dy += (size_t)NT * KCH * 512;
    }
    _Float16* dm = (_Float16*)units_m;
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(pwchain::pack_m_kernel, dim3(128), dim3(256), 0, stream, p->map_w2, H, NT, l * H, (int)(LH + (size_t)l * H),
                           (const uint32_t*)(wmax + L + l), winv + L + l, dm);
        dm += (size_t)NT * 32 * 512;
    }
    // Wm1^T: one output tile (the 32 feature channels), K = 256; Wm1 row-major (256, 32)
    if (hipError_t e = launch_pack_t16(p->map_w1, 256, 32, 32, 1, dm, winv + 2 * L, wmax + 2 * L, stream)) return e;
    if (hipError_t e = launch_pack_head_t16(p->w_final, H, head_t, winv + 2 * L + 1, wmax + 2 * L + 1, stream)) return e;
    for (int l = 1; l < L; ++l)
        if (hipError_t e = launch_col_abs_sum_max(p->w[l], H, H, anorm + l, stream)) return e;
    if (hipError_t e = launch_col_abs_sum_max(p->w_final, 4, H, anorm + L, stream)) return e;
    return hipGetLastError();
}

}  // namespace cnerf
