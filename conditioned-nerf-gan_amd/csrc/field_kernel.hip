// Fused point pass of the render path for gfx950: sample position -> trilinear lookup -> FiLM/sine-SIREN MLP ->
// rgb_sigma, one 32-point tile per wavefront at a time, activations never leave registers.
//
// Replaces, per point: get_initial_rays_trig / perturb_points / transform_sampled_points
// (volumetric_rendering.py:73-199), F.grid_sample + permute (siren.py:555-571), the FiLM / Siren / ResSiren layers
// (siren.py:146-230), the head and _sigmoid_rgb (siren.py:667, 1227-1234).
//
// MFMA formulation (fp32 in / fp32 accumulate, v_mfma_f32_32x32x2_f32, exact fmaf chain):
//   each layer computes  Y^T = W * X^T  for a tile of 32 points: A = W (i = output channel), B = X^T (j = point),
//   so the accumulator of output tile t holds, in lane (j, h = lane>>5), register r = 4g+e, channel
//   32t + 8g + 4h + e of point j.  The B operand of the NEXT layer's k-step s = 16t'+4g+e needs, in lane (j, h),
//   "some k for half 0 and some k for half 1" -- we choose exactly k(s,h) = 32t' + 8g + 4h + e, i.e. the activation
//   registers ARE the next layer's B operands, with no LDS round trip and no cross-lane traffic; the permutation is
//   absorbed into the order the weights are packed in (pack_field_kernel below): lane (i, h) of A k-step s holds
//   W[32t + i][k(s,h)].
#include "cnerf_dev.hpp"
#include "cnerf_kernels.hpp"

namespace cnerf {

// ---------------------------------------------------------------------------------------------------------------
// packed-weight addressing (units: float4 = one lane's A operands for 4 consecutive k-steps)
// ---------------------------------------------------------------------------------------------------------------
// matrix with K inputs (K = 32*KT) and 32*OT outputs: float4 index ((t*KT + tk)*4 + g)*64 + lane
//   value[e] = W[32t + (lane&31)][32tk + 8g + 4(lane>>5) + e]     (rows >= n_out are zero: the head has 4 rows)

__global__ void pack_matrix_kernel(const float* __restrict__ w, int n_out, int K, int OT, float* __restrict__ dst) {
    const int KT = K / 32;
    const int total = OT * KT * 4 * 64 * 4;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const int lane = (idx >> 2) & 63;
        const int g = (idx >> 8) & 3;
        const int rest = idx >> 10;
        const int tk = rest % KT, t = rest / KT;
        const int row = 32 * t + (lane & 31);
        const int col = 32 * tk + 8 * g + 4 * (lane >> 5) + e;
        dst[idx] = (row < n_out) ? w[(size_t)row * K + col] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the point-tile kernel
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
struct Act {
    f32x16 v[NT];
};

// acc registers of output tile t, lane half h  <-  per-channel vector p[32t + 8g + 4h + e]
__device__ __forceinline__ f32x16 load_chan16(const float* __restrict__ p, int t, int h) {
    f32x16 r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(p + 32 * t + 8 * g + 4 * h);
        r[4 * g + 0] = q[0];
        r[4 * g + 1] = q[1];
        r[4 * g + 2] = q[2];
        r[4 * g + 3] = q[3];
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// One matrix product of the MLP for a 32-point tile:  out[t] = epilogue( bias[t] + W[t, :] * in )  for the OT output
// tiles of 32 channels, KT input tiles.  The A operands (packed weights, 16 B per lane per 4 k-steps) stream from L2
// through a ring of RING float4 registers that runs RING groups (= 4*RING MFMAs = 256*RING cycles) ahead of the MFMA
// that consumes them, across output-tile boundaries and epilogues, so the L2 latency is hidden behind the matrix pipe.
// Everything is unrolled: register indices of the ring, the B operands and the accumulators are compile-time.
// ---------------------------------------------------------------------------------------------------------------
#ifndef CNERF_RING
#define CNERF_RING 6
#endif
constexpr int RING = CNERF_RING;

// epilogue kinds
enum { EPI_FILM = 0, EPI_FILM_RES = 1 };

template <int EPI>
__device__ __forceinline__ float epilogue_one(float acc, float res, float fr, float ph) {
    float pre = acc;
    if (EPI == EPI_FILM_RES) pre = res + pre;
    return sin_pi_reduced(fr * pre + ph);
}

// Software pipeline, pinned with sched_barrier(0) at every group boundary (left alone, the scheduler turns the loop
// nest inside out -- k outermost, all OT accumulators live, loads in bursts, every epilogue at the end of the layer):
//   group = { 1 ring load (RING groups ahead), 4 MFMAs of output tile t, the epilogue of 16/GPT... elements of tile t-1 }
// so only two accumulator tiles are live and the weight stream stays RING groups ahead.  Note (measured,
// scripts/ubench/mfma_valu_overlap.hip): VALU instructions do NOT hide under v_mfma_f32_32x32x2_f32 -- each adds its
// full 2 cycles whether spread between the MFMAs or clumped, with one or two accumulator chains -- so for the fp32
// path  time = MFMA + VALU + stalls  and the epilogue placement only matters for register pressure and load distance.
template <int OT, int KT, int EPI>
__device__ __forceinline__ void mlp_matrix(const f32x4* __restrict__ wp, const float* __restrict__ bias,
                                           const float* __restrict__ freq, const float* __restrict__ phase,
                                           const f32x16* in, const f32x16* res, f32x16* out, int lane, int h) {
    constexpr int GPT = KT * 4;                       // groups (of 4 MFMAs) per output tile
    constexpr int NG = OT * GPT;
    constexpr int EPG = GPT >= 16 ? 1 : 16 / GPT;     // epilogue elements handled per group
    constexpr int ESTEP = GPT >= 16 ? GPT / 16 : 1;   // ... every ESTEP-th group
    f32x4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i)
        if (i < NG) ring[i] = wp[i * 64 + lane];
    f32x16 acc_prev, fr_prev, ph_prev;
    f32x16 bias_next = load_chan16(bias, 0, h);       // per-channel vectors are fetched one output tile ahead
#pragma unroll
    for (int t = 0; t < OT; ++t) {
        f32x16 acc = bias_next;
        if (t + 1 < OT) bias_next = load_chan16(bias, t + 1, h);
        const f32x16 fr = load_chan16(freq, t, h);
        const f32x16 ph = load_chan16(phase, t, h);
#pragma unroll
        for (int gi = 0; gi < GPT; ++gi) {
            const int tk = gi >> 2, g = gi & 3;
            const int idx = t * GPT + gi;
            const f32x4 a = ring[idx % RING];
            if (idx + RING < NG) ring[idx % RING] = wp[(idx + RING) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], in[tk][4 * g + e], acc, 0, 0, 0);
            if (t > 0 && gi % ESTEP == 0 && gi / ESTEP < 16 / EPG) {
#pragma unroll
                for (int q = 0; q < EPG; ++q) {
                    const int r = (gi / ESTEP) * EPG + q;
                    out[t - 1][r] = epilogue_one<EPI>(acc_prev[r], EPI == EPI_FILM_RES ? res[t - 1][r] : 0.0f, fr_prev[r], ph_prev[r]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        acc_prev = acc;
        fr_prev = fr;
        ph_prev = ph;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        out[OT - 1][r] = epilogue_one<EPI>(acc_prev[r], EPI == EPI_FILM_RES ? res[OT - 1][r] : 0.0f, fr_prev[r], ph_prev[r]);
}

#ifdef CNERF_STAMPS
// Diagnostic build only: per-phase cycle totals (s_memtime) summed over all tiles of all waves into a.stamps[0..7].
#define STAMP(i)                                                                                     \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        const unsigned long long now_ = __builtin_readcyclecounter();                                \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        st_[i] += now_ - last_;                                                                      \
        last_ = now_;                                                                                \
    } while (0)
#else
#define STAMP(i)
#endif

template <int NT, bool HAS_RES>
__global__ __launch_bounds__(256) void field_tile_kernel(FieldArgs a) {
#ifdef CNERF_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_readcyclecounter();
#endif
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    constexpr int H = NT * 32;
    constexpr size_t TILE4 = 4 * 64;   // float4 per (t, tk) pair

    // XCD-aware tile ownership: blocks b and b+8 share an XCD (round-robin dispatch), so give each of the 8 block
    // classes one contiguous eighth of the tiles (a band of neighbouring rays -> a compact slab of the feature grid in
    // that XCD's L2).  Placement only changes speed, never results.
    const int nblk = gridDim.x;
    const int cls = blockIdx.x & 7, idx_in_cls = blockIdx.x >> 3;
    const int blk_per_cls = (nblk + 7 - cls) / 8;   // blocks whose id % 8 == cls
    const long long T = a.total_tiles;
    const long long t_begin = T * cls / 8, t_end = T * (cls + 1) / 8;
    const long long stride = (long long)blk_per_cls * 4;

    for (long long tile = t_begin + idx_in_cls * 4 + wave_in_block; tile < t_end; tile += stride) {
        const int b = (int)(tile / a.tiles_per_image);
        const long long n = (tile - (long long)b * a.tiles_per_image) * 32 + j;  // point index inside image b
        const bool valid = n < a.n_per_image;
        const long long nn = valid ? n : (a.n_per_image - 1);                     // padded lanes recompute the last point

        STAMP(0);   // loop overhead / previous store
        // ---- sample position -------------------------------------------------------------------------------------
        float px, py, pz;
        if (a.mode == FIELD_MODE_POINTS) {
            const float* p = a.points + ((size_t)b * a.n_per_image + nn) * 3;
            px = p[0];
            py = p[1];
            pz = p[2];
        } else {
            const int S = a.geom.S, R = a.geom.R;
            const int ray = (int)(nn / S), s = (int)(nn - (long long)ray * S);
            const int row = ray / R, col = ray - row * R;
            float dx, dy, dz;
            camera_dir(a.geom, row, col, dx, dy, dz);
            const float* m = a.cam2world + (size_t)b * 16;
            if (a.mode == FIELD_MODE_COARSE) {
                const float u = a.u_strat ? a.u_strat[(size_t)b * a.n_per_image + nn] : 0.5f;
                float zj;
                coarse_sample(a.geom, m, dx, dy, dz, s, u, zj, px, py, pz);
                if (valid && h == 0) a.z_out[(size_t)b * a.n_per_image + nn] = zj;
            } else {
                const float t = a.fine_z[(size_t)b * a.n_per_image + nn];
                fine_sample(m, dx, dy, dz, t, px, py, pz);
            }
        }
        if (a.points_out && valid && h == 0) {
            float* po = a.points_out + ((size_t)b * a.n_per_image + nn) * 3;
            po[0] = px;
            po[1] = py;
            po[2] = pz;
        }

        // ---- trilinear lookup: this lane's 16 of the 32 channels (8g + 4h + e) ---------------------------------------
        Corner8 cr;
        trilinear_corners(px, py, pz, a.half_voxel, a.V, cr);
        const float* vol = a.fvol + (size_t)b * a.V * a.V * a.V * 32 + 4 * h;
        f32x16 feat;
        {
            f32x4 q[8][4];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* cp = vol + (size_t)cr.base[k] * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) q[k][g] = *reinterpret_cast<const f32x4*>(cp + 8 * g);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) feat[r] = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k)     // ATen order: corners sequentially, product and sum rounded separately
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) feat[4 * g + e] = feat[4 * g + e] + q[k][g][e] * cr.w[k];
        }

        asm volatile("" :: "v"(feat[0]), "v"(feat[15]));
        STAMP(1);   // position + gather
        // ---- MLP ---------------------------------------------------------------------------------------------------
        // A plain sine layer is a FiLM layer with freq = 1, phase = 0 (1*x and +0 are exact): one code path.
        Act<NT> x, y;
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.packed);
        const float* bias = a.bias;              // concatenated biases, layer after layer (H each, RES: 2H)
        const float* ones = a.bias + a.bias_floats;
        const float* zeros = ones + H;
        const float* freq = a.freq ? a.freq + (size_t)b * a.film_stride : nullptr;
        const float* phase = a.phase ? a.phase + (size_t)b * a.film_stride : nullptr;
        {
            const bool film = a.layer_kind[0] == CNERF_LAYER_FILM;
            mlp_matrix<NT, 1, EPI_FILM>(wp, bias, film ? freq : ones, film ? phase : zeros, &feat, nullptr, x.v, lane, h);
            wp += (size_t)NT * TILE4;
            bias += H;
            if (film) {
                freq += H;
                phase += H;
            }
        }
        asm volatile("" :: "v"(x.v[0][0]), "v"(x.v[NT - 1][15]));
        STAMP(2);   // layer 0
        for (int l = 1; l < a.L; ++l) {
            const int kind = a.layer_kind[l];
            if (!HAS_RES || kind != CNERF_LAYER_RES) {
                const bool film = kind == CNERF_LAYER_FILM;
                mlp_matrix<NT, NT, EPI_FILM>(wp, bias, film ? freq : ones, film ? phase : zeros, x.v, nullptr, y.v, lane, h);
                wp += (size_t)NT * NT * TILE4;
                bias += H;
                if (film) {
                    freq += H;
                    phase += H;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) x.v[t] = y.v[t];
                asm volatile("" :: "v"(x.v[0][0]), "v"(x.v[NT - 1][15]));
                STAMP(3);   // hidden layers
            } else {
                // y = sin(W1 x + b1);  x = sin(x + W2 y + b2)   (tile t of x is dead once its own residual is added)
                mlp_matrix<NT, NT, EPI_FILM>(wp, bias, ones, zeros, x.v, nullptr, y.v, lane, h);
                wp += (size_t)NT * NT * TILE4;
                bias += H;
                mlp_matrix<NT, NT, EPI_FILM_RES>(wp, bias, ones, zeros, y.v, x.v, x.v, lane, h);
                wp += (size_t)NT * NT * TILE4;
                bias += H;
            }
        }

        // ---- head: 4 output rows padded to one 32-row tile; rows 0..3 land in registers 0..3 of half 0 --------------
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            if (h == 0) {
                acc[0] = bias[0];
                acc[1] = bias[1];
                acc[2] = bias[2];
                acc[3] = bias[3];
            }
            constexpr int NG = NT * 4;
            f32x4 ring[RING];
#pragma unroll
            for (int i = 0; i < RING; ++i)
                if (i < NG) ring[i] = wp[i * 64 + lane];
#pragma unroll
            for (int tk = 0; tk < NT; ++tk) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int idx = tk * 4 + g;
                    const f32x4 aw = ring[idx % RING];
                    if (idx + RING < NG) ring[idx % RING] = wp[(idx + RING) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[e], x.v[tk][4 * g + e], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (valid && h == 0) {
                f32x4 o;
                if (a.flags & CNERF_F_SIGMOID_RGB) {
                    o[0] = sigmoidf_(acc[0]);
                    o[1] = sigmoidf_(acc[1]);
                    o[2] = sigmoidf_(acc[2]);
                } else {
                    o[0] = acc[0];
                    o[1] = acc[1];
                    o[2] = acc[2];
                }
                o[3] = acc[3];
                *reinterpret_cast<f32x4*>(a.rgb_sigma + ((size_t)b * a.n_per_image + nn) * 4) = o;
            }
        }
        STAMP(4);   // head
    }
#ifdef CNERF_STAMPS
    if (a.stamps && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(a.stamps + i, st_[i]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// host-side launchers (called from cnerf_abi.hip)
// ---------------------------------------------------------------------------------------------------------------
__global__ void fill_kernel(float* dst, float value, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

hipError_t launch_fill(float* dst, float value, int n, hipStream_t stream) {
    hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dst, value, n);
    return hipGetLastError();
}

hipError_t launch_pack_matrix(const float* w, int n_out, int K, int OT, float* dst, hipStream_t stream) {
    const int total = OT * (K / 32) * 4 * 64 * 4;
    const int blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_matrix_kernel, dim3(blocks), dim3(256), 0, stream, w, n_out, K, OT, dst);
    return hipGetLastError();
}

template <int NT, bool HAS_RES>
static hipError_t launch_field_nt(const FieldArgs& a, hipStream_t stream) {
    int dev = 0, cus = 256, per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, field_tile_kernel<NT, HAS_RES>, 256, 0) != hipSuccess || per_cu < 1)
        per_cu = 1;
    long long want = (a.total_tiles + 3) / 4;
    long long cap = (long long)cus * per_cu;
    int blocks = (int)(want < cap ? want : cap);
    if (blocks < 8) blocks = 8;               // every XCD class owns an eighth of the tiles
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL((field_tile_kernel<NT, HAS_RES>), dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_field(const FieldArgs& a, int H, hipStream_t stream) {
    bool res = false;
    for (int l = 0; l < a.L; ++l) res |= a.layer_kind[l] == CNERF_LAYER_RES;
    switch (H / 32) {
        case 2: return res ? launch_field_nt<2, true>(a, stream) : launch_field_nt<2, false>(a, stream);
        case 4: return res ? launch_field_nt<4, true>(a, stream) : launch_field_nt<4, false>(a, stream);
        case 8: return res ? launch_field_nt<8, true>(a, stream) : launch_field_nt<8, false>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace cnerf
