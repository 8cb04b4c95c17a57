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
#include "field_common.hpp"

namespace cnerf {

// ---------------------------------------------------------------------------------------------------------------
// packed-weight addressing (units: float4 = one lane's A operands for 4 consecutive k-steps)
// ---------------------------------------------------------------------------------------------------------------
// matrix with K inputs (K = 32*KT) and 32*OT outputs: float4 index ((t*KT + tk)*4 + g)*64 + lane
//   value[e] = W[32t + (lane&31)][32tk + 8g + 4(lane>>5) + e]     (rows >= n_out are zero: the head has 4 rows)

__global__ void pack_matrix_kernel(const float* __restrict__ w, int n_out, int K_real, int K, int OT, float* __restrict__ dst) {
    const int KT = K / 32;          // K = K_real padded to a multiple of 32 (columns >= K_real are zero)
    const int total = OT * KT * 4 * 64 * 4;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const int lane = (idx >> 2) & 63;
        const int g = (idx >> 8) & 3;
        const int rest = idx >> 10;
        const int tk = rest % KT, t = rest / KT;
        const int row = 32 * t + (lane & 31);
        const int col = 32 * tk + 8 * g + 4 * (lane >> 5) + e;
        dst[idx] = (row < n_out && col < K_real) ? w[(size_t)row * K_real + col] : 0.0f;
    }
}

// Head (4 outputs) for v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products per instruction.  Block b = lane/4
// covers points 4b..4b+3 of the tile for lanes 0..31 and the same points again for lanes 32..63 (which hold the other
// half of the channels), so the B operand is again an activation register as it stands; lane (b, i = lane&3) of the A
// operand carries W_head[i][channel(s, lane>>5)].  float4 index (s/4)*64 + lane, element s%4; s = 16 t + r runs over the
// activation registers (channel 32t + 8(r>>2) + 4h + (r&3)).
__global__ void pack_head_kernel(const float* __restrict__ w, int H, float* __restrict__ dst) {
    const int total = (H / 2) * 64;                 // H/2 registers per lane x 64 lanes
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) {
        const int e = idx & 3, lane = (idx >> 2) & 63, q = idx >> 8;
        const int sreg = 4 * q + e, t = sreg >> 4, r = sreg & 15, h = lane >> 5;
        const int ch = 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
        dst[idx] = w[(size_t)(lane & 3) * H + ch];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the point-tile kernel
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
struct Act {
    f32x16 v[NT];
};

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

// Forward epilogue sine.  v_sin_f32 behind an exact reduction by 2 pi (max abs error 3.8e-7, scripts/ubench/vsin_accuracy.hip)
// instead of the 12-op polynomial (1.2e-7) is +1.7 % throughput, but on this exact path it pushes one random-input parity
// case (test_render_matches_oracle_random_inputs, sigma head x30) from 0.9e-4 to 1.03e-4 of the 1e-4 gate: off.

// epilogue kinds
enum { EPI_FILM = 0, EPI_FILM_RES = 1 };

// FOLD (plain forward of cnerf_render_forward): arg = freq * pre + phase is affine in the accumulator (which already holds the bias;
// a plain sine layer or a residual matrix is freq = 1, phase = 0, with the block input added to pre first)
// and v_sin_f32 takes revolutions, so with the per-image constants M = freq / 2 pi = Mh + Ml and K = phase / 2 pi
// (fold_film_kernel, once per call) the activation is  n = rint(pre Mh);  u = fma(pre, Ml, fma(pre, Mh, -n)) + K;  sin(2 pi u):
// 5 vector ops + v_sin instead of 2 + the 12-op reduce-and-polynomial sine.  pre * Mh - n is a single-rounding fma of
// magnitude <= 1/2: the argument never exists rounded at its full magnitude (~200 rad, ulp 1.5e-5), which more than pays
// for v_sin's 3.8e-7 against the polynomial's 1.2e-7 (the reason the hardware sine alone was rejected above).
// CNERF_F32_FOLD_ML = 0 (default) drops the low part of M, one vector op per activation less: M is then freq / 2 pi rounded to
// fp32, a relative 6e-8, i.e. up to ~2e-6 revolutions on an argument of ~30 -- the size of ONE of the three roundings the
// reference's own fp32 sequence makes at that magnitude.  Measured: 28.9 -> 28.4 ms per launch, parity unchanged (timed image of
// bench.py: rgb / sigma 6.9e-5 / 7.8e-5 coarse / fine with Ml, 7.1e-5 / 7.2e-5 without; every parity test passes either way).
__device__ __forceinline__ float folded_sine(float pre, float mh, float ml, float k) {
    const float n = __builtin_rintf(pre * mh);
    return __builtin_amdgcn_sinf(__builtin_fmaf(pre, mh, -n) + k);
}

// WFOLD: the scale is folded into the WEIGHTS, per image (a residual matrix adds its block input, scaled by 1 / 2 pi, in the epilogue) -- W'_b = diag(freq_b / 2 pi) W, packed per call into the
// workspace (scale_packed_kernel), the accumulator starts from K_b = (freq_b bias + phase_b) / 2 pi -- so the accumulator IS the
// argument in revolutions and the activation is sin(2 pi fract(acc)): one vector op + v_sin, no per-channel constants in the
// epilogue.  Numerically the sum is rounded at the same relative precision as before; the weights carry one more rounding (6e-8).
// Reduction of the accumulator (revolutions) in front of v_sin_f32, which is specified for |u| <= 256 only -- the WFOLD template
// argument of the kernels (0 = no weight folding):
//   1: v_fract_f32 -- one vector op, any magnitude; [0, 1) instead of [-0.5, 0.5] costs half an ulp of 1 on a negative argument
//      (<= 2.7e-7 abs instead of 1.2e-7).  Networks with FiLM layers (arguments of tens of radians: ulp 4e-6 and more).
//   2: u - rint(u) -- two vector ops, exact, 1.2e-7.  Networks of plain sine layers / residual blocks, whose hidden arguments are
//      a few radians: there the half ulp of `fract` is visible (sigma of tests/golden/short_f_small 1.6e-5 -> 6.3e-5 of its scale).
// (v_sin_f32 alone also reduces by itself, 1.2e-7 for |u| <= 250, but returns 0 beyond 256 revolutions: measured in round 2 and
// not shipped.)  scripts/ubench/vsin_raw_range.hip, profiles/r02_vsin_reduction.txt
template <int WFOLD>
__device__ __forceinline__ float wfolded_sine(float u) {
    return __builtin_amdgcn_sinf(WFOLD == 1 ? __builtin_amdgcn_fractf(u) : u - __builtin_rintf(u));
}

template <int EPI, bool STORE>
__device__ __forceinline__ float epilogue_one(float acc, float res, float fr, float ph, float& cs) {
    float pre = acc;
    if (EPI == EPI_FILM_RES) pre = res + pre;
    if (STORE) {
        float sn;
        sincos_pi_reduced(fr * pre + ph, sn, cs);
        return sn;
    }
    return sin_pi_reduced(fr * pre + ph);
}

// row-major activation store of one output tile: lane (j,h) owns channels 32t + 8g + 4h + e of its point
__device__ __forceinline__ void store_tile_rows(float* __restrict__ row /* &buf[point][0] */, int t, int h, const f32x16& v) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = v[4 * g + e];
        *reinterpret_cast<f32x4*>(row + 32 * t + 8 * g + 4 * h) = q;
    }
}

// Software pipeline, pinned with sched_barrier(0) at every group boundary (left alone, the scheduler turns the loop
// nest inside out -- k outermost, all OT accumulators live, loads in bursts, every epilogue at the end of the layer):
//   group = { 1 ring load (RING groups ahead), 4 MFMAs of output tile t, the epilogue of 16/GPT... elements of tile t-1 }
// so only two accumulator tiles are live and the weight stream stays RING groups ahead.  Note (measured,
// scripts/ubench/mfma_valu_overlap.hip): VALU instructions do NOT hide under v_mfma_f32_32x32x2_f32 -- each adds its
// full 2 cycles whether spread between the MFMAs or clumped, with one or two accumulator chains -- so for the fp32
// path  time = MFMA + VALU + stalls  and the epilogue placement only matters for register pressure and load distance.
// DROP: the finished tile (and its cosine row) is multiplied by the dropout factors of layer `drop_d` at point `drop_gp`.
template <int OT, int KT, int EPI, bool STORE, bool DROP = false, bool FOLD = false, int WFOLD = 0>
__device__ __forceinline__ void mlp_matrix(const f32x4* __restrict__ wp, const float* __restrict__ bias,
                                           const float* __restrict__ freq, const float* __restrict__ phase,
                                           const f32x16* in, const f32x16* res, f32x16* out, int lane, int h,
                                           float* __restrict__ row_h, float* __restrict__ row_c,
                                           const FieldArgs* da = nullptr, unsigned long long drop_gp = 0, int drop_d = 0,
                                           const float* __restrict__ fold_ml = nullptr) {     // FOLD: freq = Mh, phase = K
    constexpr int GPT = KT * 4;                       // groups (of 4 MFMAs) per output tile
    constexpr int NG = OT * GPT;
    constexpr int EPG = GPT >= 16 ? 1 : 16 / GPT;     // epilogue elements handled per group
    constexpr int ESTEP = GPT >= 16 ? GPT / 16 : 1;   // ... every ESTEP-th group
    f32x4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i)
        if (i < NG) ring[i] = wp[i * 64 + lane];
    f32x16 acc_prev, fr_prev, ph_prev, ml_prev, cos_t;
    f32x16 bias_next = load_chan16(bias, 0, h);       // per-channel vectors are fetched one output tile ahead
#pragma unroll
    for (int t = 0; t < OT; ++t) {
        f32x16 acc = bias_next;
        if (t + 1 < OT) bias_next = load_chan16(bias, t + 1, h);
        f32x16 fr, ph, ml;
        if (!WFOLD) {
            fr = load_chan16(freq, t, h);
            ph = load_chan16(phase, t, h);
        }
        if (FOLD && !WFOLD) ml = load_chan16(fold_ml, t, h);
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
                    float cs_ = 0.0f;
                    if (WFOLD) out[t - 1][r] = wfolded_sine<WFOLD>(EPI == EPI_FILM_RES ? __builtin_fmaf(res[t - 1][r], 0.15915494309189535f, acc_prev[r]) : acc_prev[r]);
                    else if (FOLD) out[t - 1][r] = folded_sine(EPI == EPI_FILM_RES ? res[t - 1][r] + acc_prev[r] : acc_prev[r], fr_prev[r], ml_prev[r], ph_prev[r]);
                    else out[t - 1][r] = epilogue_one<EPI, STORE>(acc_prev[r], EPI == EPI_FILM_RES ? res[t - 1][r] : 0.0f, fr_prev[r],
                                                                  ph_prev[r], cs_);
                    if (STORE) cos_t[r] = cs_;
                }
            }
            if (DROP && t > 0 && gi == GPT - 1) {
                if (STORE) drop_tile2(*da, drop_gp, drop_d, OT * 32, t - 1, h, out[t - 1], cos_t);
                else drop_tile(*da, drop_gp, drop_d, OT * 32, t - 1, h, out[t - 1]);
            }
            if (STORE && t > 0 && gi == GPT - 1) {      // tile t-1 is complete: spill it for the backward pass
                store_tile_rows(row_h, t - 1, h, out[t - 1]);
                store_tile_rows(row_c, t - 1, h, cos_t);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        acc_prev = acc;
        if (!WFOLD) {
            fr_prev = fr;
            ph_prev = ph;
        }
        if (FOLD && !WFOLD) ml_prev = ml;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float cs_ = 0.0f;
        if (WFOLD) out[OT - 1][r] = wfolded_sine<WFOLD>(EPI == EPI_FILM_RES ? __builtin_fmaf(res[OT - 1][r], 0.15915494309189535f, acc_prev[r]) : acc_prev[r]);
        else if (FOLD) out[OT - 1][r] = folded_sine(EPI == EPI_FILM_RES ? res[OT - 1][r] + acc_prev[r] : acc_prev[r], fr_prev[r], ml_prev[r], ph_prev[r]);
        else out[OT - 1][r] = epilogue_one<EPI, STORE>(acc_prev[r], EPI == EPI_FILM_RES ? res[OT - 1][r] : 0.0f, fr_prev[r], ph_prev[r],
                                                       cs_);
        if (STORE) cos_t[r] = cs_;
    }
    if (DROP) {
        if (STORE) drop_tile2(*da, drop_gp, drop_d, OT * 32, OT - 1, h, out[OT - 1], cos_t);
        else drop_tile(*da, drop_gp, drop_d, OT * 32, OT - 1, h, out[OT - 1]);
    }
    if (STORE) {
        store_tile_rows(row_h, OT - 1, h, out[OT - 1]);
        store_tile_rows(row_c, OT - 1, h, cos_t);
    }
}


// y[t] += W0[t, tile tk] * feat   for all NT output tiles (k-outer form of layer 0: any number of input tiles)
template <int NT>
__device__ __forceinline__ void layer0_accumulate(const f32x4* __restrict__ wp, int n_in, int tk, const f32x16& feat,
                                                  f32x16* y, int lane) {
    constexpr int NG = NT * 4;
    auto addr = [&](int i) { return wp + ((size_t)((i >> 2) * n_in + tk) * 4 + (i & 3)) * 64 + lane; };
    f32x4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i)
        if (i < NG) ring[i] = *addr(i);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int t = i >> 2, g = i & 3;
        const f32x4 aw = ring[i % RING];
        if (i + RING < NG) ring[i % RING] = *addr(i + RING);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[e], feat[4 * g + e], y[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// x[t] = sin(freq * y[t] + phase) for all tiles (+ activation store)
template <int NT, bool STORE, bool DROP = false, bool FOLD = false, int WFOLD = 0>
__device__ __forceinline__ void film_all(const f32x16* y, f32x16* x, const float* __restrict__ freq,
                                         const float* __restrict__ phase, int h, float* row_h, float* row_c,
                                         const FieldArgs* da = nullptr, unsigned long long drop_gp = 0,
                                         const float* __restrict__ fold_ml = nullptr) {
    f32x16 fr_n = load_chan16(freq, 0, h), ph_n = load_chan16(phase, 0, h);      // per-channel vectors one tile ahead of their use
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x16 fr = fr_n, ph = ph_n;
        if (t + 1 < NT) {
            fr_n = load_chan16(freq, t + 1, h);
            ph_n = load_chan16(phase, t + 1, h);
        }
        f32x16 o, cs;
        if (WFOLD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = wfolded_sine<WFOLD>(y[t][r]);
        } else if (FOLD) {
            const f32x16 ml = load_chan16(fold_ml, t, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = folded_sine(y[t][r], fr[r], ml[r], ph[r]);
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float c_ = 0.0f;
            o[r] = epilogue_one<EPI_FILM, STORE>(y[t][r], 0.0f, fr[r], ph[r], c_);
            if (STORE) cs[r] = c_;
        }
        }
        if (DROP) {
            if (STORE) drop_tile2(*da, drop_gp, 0, NT * 32, t, h, o, cs);
            else drop_tile(*da, drop_gp, 0, NT * 32, t, h, o);
        }
        x[t] = o;
        if (STORE) {
            store_tile_rows(row_h, t, h, o);
            store_tile_rows(row_c, t, h, cs);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
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

// Measured on one MI355X (bench.py, batch 8): parking the next tile's lookups in LDS by DMA removes 7.4 k cycles of wait
// from layer 0 but the 32 scattered global_load_lds instructions cost 8.6 k cycles to issue in the head (~200 cycles each:
// M0 rewrite + 32 distinct lines per instruction), a net loss of 0.5 %; only the one-tile-ahead fetch of the raw sample
// coordinate is kept by default.

template <int NT, bool HAS_RES, bool STORE, bool DROP, bool FOLD = false, int WFOLD = 0>
__global__ __launch_bounds__(256) void field_tile_kernel(FieldArgs a) {
#ifdef CNERF_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_readcyclecounter();
#endif
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    constexpr int H = NT * 32;
    constexpr size_t TILE4 = 4 * 64;   // float4 per (t, tk) pair

    // The head's weights (4x4x1 MFMA layout: NT*4 float4 per lane) are read in full by every wave for every tile -- as many
    // bytes as the lookups.  They are parked in LDS once per block instead of streamed from L2 once per tile.
    __shared__ f32x4 s_head[NT * 4 * 64];
    {
        const f32x4* head_w = reinterpret_cast<const f32x4*>(a.packed) + ((size_t)NT * a.n_in + (size_t)(a.n_mats - 1) * NT * NT) * TILE4;
        for (int i = threadIdx.x; i < NT * 4 * 64; i += 256) s_head[i] = head_w[i];
        __syncthreads();
    }

    const TileRange tr = tile_range(a.total_tiles);
    auto point_of = [&](long long tile, int& b_, long long& nn_, bool& valid_) {
        b_ = (int)(tile / a.tiles_per_image);
        const long long n = (tile - (long long)b_ * a.tiles_per_image) * 32 + j;  // point index inside image b
        valid_ = n < a.n_per_image;
        nn_ = valid_ ? n : (a.n_per_image - 1);                                     // padded lanes recompute the last point
    };
    int b = 0;
    long long nn = 0;
    bool valid = false;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (tr.begin < tr.end) {
        point_of(tr.begin, b, nn, valid);
        tile_point(a, b, nn, valid, h, true, px, py, pz);
    }
    for (long long tile = tr.begin; tile < tr.end; tile += tr.stride) {
        STAMP(0);   // loop overhead / previous store
        // raw sample coordinate of the next tile of this wave (this tile again at the end of the range)
        const bool has_next = tile + tr.stride < tr.end;
        int nb;
        long long nnn;
        bool nvalid;
        point_of(has_next ? tile + tr.stride : tile, nb, nnn, nvalid);
        const TileRaw raw_next = tile_point_fetch(a, nb, nnn);

        // ---- layer 0: lookups feed the matrix pipe tile by tile ---------------------------------------------------------
        // A plain sine layer is a FiLM layer with freq = 1, phase = 0 (1*x and +0 are exact): one code path.
        Act<NT> x, y;
        const size_t gpt = (size_t)b * a.n_per_image + nn;                 // global point row of the activation buffers
        const unsigned long long drop_gp = (unsigned long long)(b + a.image0) * a.n_per_image + nn;   // ... of the whole call
        int drop_d = 0;                                                    // dropout layers seen so far (layer 0 is one)
        const size_t act_layer = (size_t)a.act_points * H;                 // floats per layer in act_h / act_c
        float* row_h = STORE ? a.act_h + gpt * H : nullptr;
        float* row_c = STORE ? a.act_c + gpt * H : nullptr;
        // WFOLD: this image's row-scaled weights; `bias` walks K_b = (freq bias + phase) / 2 pi, the accumulators' starting values
        const f32x4* wp = reinterpret_cast<const f32x4*>(WFOLD ? a.packed_img + (size_t)(b + a.image0) * a.packed_img_stride : a.packed);
        const float* bias = WFOLD ? a.fold + 3 * (size_t)a.fold_images * a.n_mats * H + (size_t)(b + a.image0) * a.n_mats * H : a.bias;
        const float* ones = a.bias + a.bias_floats;
        const float* zeros = ones + H;
        // FOLD: `freq` / `phase` walk the folded constants Mh / K of the image instead, `fml` the low parts Ml (a.fold: three
        // arrays of (B, n_mats * H) floats: Mh, Ml, K -- one H-vector per MATRIX, whatever its layer kind)
        const size_t fold_stride = (size_t)a.n_mats * H;
        const size_t fold_n = FOLD ? (size_t)a.fold_images * fold_stride : 0;
        const float* freq = FOLD ? a.fold + (size_t)(b + a.image0) * fold_stride : (a.freq ? a.freq + (size_t)b * a.film_stride : nullptr);
        const float* phase = FOLD ? freq + 2 * fold_n : (a.phase ? a.phase + (size_t)b * a.film_stride : nullptr);
        const float* fml = FOLD ? freq + fold_n : nullptr;
#pragma unroll
        for (int t = 0; t < NT; ++t) y.v[t] = load_chan16(bias, t, h);
        for (int tk = 0; tk < a.n_in; ++tk) {
            const f32x16 feat = input_tile(a, b, tk, px, py, pz, h);
            if (STORE) {
                float* fo = a.act_feat + gpt * (32 * a.n_in) + 32 * tk + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = feat[4 * g + e];
                    *reinterpret_cast<f32x4*>(fo + 8 * g) = q;
                }
            }
            layer0_accumulate<NT>(wp, a.n_in, tk, feat, y.v, lane);
        }
        STAMP(1);   // position + lookups + layer-0 products
        {
            const bool film = a.layer_kind[0] == CNERF_LAYER_FILM;
            film_all<NT, STORE, DROP, FOLD, WFOLD>(y.v, x.v, (FOLD || film) ? freq : ones, (FOLD || film) ? phase : zeros, h, row_h, row_c, &a, drop_gp, fml);
            if (STORE) {
                row_h += act_layer;
                row_c += act_layer;
            }
            wp += (size_t)NT * a.n_in * TILE4;
            bias += H;
            if (FOLD || film) {
                freq += H;
                phase += H;
                if (FOLD) fml += H;
            }
        }
        asm volatile("" :: "v"(x.v[0][0]), "v"(x.v[NT - 1][15]));
        STAMP(2);   // layer 0 epilogue
        for (int l = 1; l < a.L; ++l) {
            const int kind = a.layer_kind[l];
            if (!HAS_RES || kind != CNERF_LAYER_RES) {
                const bool film = kind == CNERF_LAYER_FILM;
                ++drop_d;
                mlp_matrix<NT, NT, EPI_FILM, STORE, DROP, FOLD, WFOLD>(wp, bias, (FOLD || film) ? freq : ones, (FOLD || film) ? phase : zeros, x.v,
                                                                       nullptr, y.v, lane, h, row_h, row_c, &a, drop_gp, drop_d, fml);
                if (STORE) {
                    row_h += act_layer;
                    row_c += act_layer;
                }
                wp += (size_t)NT * NT * TILE4;
                bias += H;
                if (FOLD || film) {
                    freq += H;
                    phase += H;
                    if (FOLD) fml += H;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) x.v[t] = y.v[t];
                asm volatile("" :: "v"(x.v[0][0]), "v"(x.v[NT - 1][15]));
                STAMP(3);   // hidden layers
            } else {
                // y = sin(W1 x + b1);  x = sin(x + W2 y + b2)   (tile t of x is dead once its own residual is added)
                mlp_matrix<NT, NT, EPI_FILM, STORE, false, FOLD, WFOLD>(wp, bias, FOLD ? freq : ones, FOLD ? phase : zeros, x.v, nullptr, y.v, lane, h,
                                                                        row_h, row_c, nullptr, 0, 0, fml);
                if (STORE) {
                    row_h += act_layer;
                    row_c += act_layer;
                }
                wp += (size_t)NT * NT * TILE4;
                bias += H;
                if (FOLD) {
                    freq += H;
                    phase += H;
                    fml += H;
                }
                mlp_matrix<NT, NT, EPI_FILM_RES, STORE, false, FOLD, WFOLD>(wp, bias, FOLD ? freq : ones, FOLD ? phase : zeros, y.v, x.v, x.v, lane, h,
                                                                            row_h, row_c, nullptr, 0, 0, fml);
                if (FOLD) {
                    freq += H;
                    phase += H;
                    fml += H;
                }
                if (STORE) {
                    row_h += act_layer;
                    row_c += act_layer;
                }
                wp += (size_t)NT * NT * TILE4;
                bias += H;
            }
        }

        // ---- next tile: finish its position, send its lookups off; they land under the head and the loop overhead ----------
        float nx, ny, nz;
        tile_point_finish(a, nb, nnn, raw_next, nvalid, h, has_next, nx, ny, nz);

        // ---- head: 4 outputs on the 4x4x1 MFMA (16 blocks of 4 points), see pack_head_kernel -------------------------------
        {
            // (WFOLD: `bias` / `wp` walked the per-image constants / weights: the head's own bias and weights are the shared ones)
            const float* head_bias = WFOLD ? a.bias + (size_t)a.n_mats * H : bias;
            const f32x4 acc = head_forward<NT>(s_head, head_bias, x.v, lane);
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
        b = nb;
        nn = nnn;
        valid = nvalid;
        px = nx;
        py = ny;
        pz = nz;
        STAMP(4);   // head
    }
#ifdef CNERF_STAMPS
    if (a.stamps && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(a.stamps + i, st_[i]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Per-point FiLM family (TALLSIREN, siren.py:232-331): layer input = world xyz; frequencies and phases of every layer
// come per POINT from a mapping MLP of the looked-up feature:  m = LeakyReLU_0.2(Wm1 feat + bm1) (256 wide),
// [freq | phase] = Wm2 m + bm2 (2*L*H wide), freq = freq*15+30.  The 2*L*H mapping outputs are never materialised: for
// output tile t of layer l the kernel runs three accumulations -- W_l x, Wm2[freq rows] m, Wm2[phase rows] m -- and
// combines them in the epilogue.
// STORE (the activation-storing re-run of the backward pass, see field_pw_backward_kernel): the looked-up feature, m, and per
// layer four rows per point -- y = sin(arg), cos(arg), cos(arg) * freq, cos(arg) * 15 * pre -- i.e. d arg / d (phase, pre, raw
// frequency output) already multiplied into the cosine, so that the gradient chain is three multiplies per element.
// Packed stream: Wm1 (8 x 1 tiles) | per layer: W_l (NT x KT), Wm2 freq rows (NT x 8), Wm2 phase rows (NT x 8) | head.
// Bias stream:   bm1 (256) | per layer: b_l (H), bm2 freq slice (H), bm2 phase slice (H) | head bias (4).
// ---------------------------------------------------------------------------------------------------------------
template <int KT>
__device__ __forceinline__ f32x16 mfma_accumulate(const f32x4* __restrict__ wp, const f32x16* in, f32x16 acc, int lane) {
    constexpr int NG = KT * 4;
    f32x4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i)
        if (i < NG) ring[i] = wp[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int tk = i >> 2, g = i & 3;
        const f32x4 aw = ring[i % RING];
        if (i + RING < NG) ring[i % RING] = wp[(i + RING) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[e], in[tk][4 * g + e], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

template <int NT, int KT, bool STORE, bool DROP>
__device__ __forceinline__ void pfilm_layer(const f32x4* __restrict__ w_main, const f32x4* __restrict__ w_freq,
                                            const f32x4* __restrict__ w_phase, const float* __restrict__ b_main,
                                            const float* __restrict__ b_freq, const float* __restrict__ b_phase,
                                            const f32x16* x, const f32x16* m, f32x16* y, int lane, int h,
                                            float* __restrict__ row_y, float* __restrict__ row_c, size_t slab,
                                            const FieldArgs& a, unsigned long long drop_gp, int drop_d) {
    constexpr size_t TILE4 = 4 * 64;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x16 fr = mfma_accumulate<8>(w_freq + (size_t)t * 8 * TILE4, m, load_chan16(b_freq, t, h), lane);
        f32x16 ph = mfma_accumulate<8>(w_phase + (size_t)t * 8 * TILE4, m, load_chan16(b_phase, t, h), lane);
        f32x16 pre = mfma_accumulate<KT>(w_main + (size_t)t * KT * TILE4, x, load_chan16(b_main, t, h), lane);
        f32x16 o;
        if (STORE) {
            f32x16 cs, cf, cp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = fr[r] * 15.0f + 30.0f;
                float sn, c_;
                sincos_pi_reduced(f * pre[r] + ph[r], sn, c_);
                o[r] = sn;
                cs[r] = c_;
                cf[r] = c_ * f;
                cp[r] = c_ * (15.0f * pre[r]);
            }
            if (DROP) {
                drop_tile2(a, drop_gp, drop_d, NT * 32, t, h, o, cs);
                drop_tile2(a, drop_gp, drop_d, NT * 32, t, h, cf, cp);
            }
            if (row_y) {                       // (padded lanes of an image's last tile shadow its last point: they do not store)
                store_tile_rows(row_y, t, h, o);
                store_tile_rows(row_c, t, h, cs);
                store_tile_rows(row_c + slab, t, h, cf);
                store_tile_rows(row_c + 2 * slab, t, h, cp);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = sin_pi_reduced((fr[r] * 15.0f + 30.0f) * pre[r] + ph[r]);
            if (DROP) drop_tile(a, drop_gp, drop_d, NT * 32, t, h, o);
        }
        y[t] = o;
    }
}

template <int NT, bool STORE, bool DROP>
__global__ __launch_bounds__(256) void field_pw_kernel(FieldArgs a) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    constexpr int H = NT * 32;
    constexpr size_t TILE4 = 4 * 64;
    const TileRange tr = tile_range(a.total_tiles);
    for (long long tile = tr.begin; tile < tr.end; tile += tr.stride) {
        const int b = (int)(tile / a.tiles_per_image);
        const long long n = (tile - (long long)b * a.tiles_per_image) * 32 + j;
        const bool valid = n < a.n_per_image;
        const long long nn = valid ? n : (a.n_per_image - 1);
        float px, py, pz;
        tile_point(a, b, nn, valid, h, true, px, py, pz);
        // activation store (STORE): act_h = L slabs (n,H) of y, then m (n,256); act_c = 3L slabs (n,H): cos, cos*freq, cos*15*pre
        const size_t gpt = (size_t)b * a.n_per_image + nn;
        const size_t slab = (size_t)a.act_points * H;
        const unsigned long long drop_gp = (unsigned long long)(b + a.image0) * a.n_per_image + nn;
        const bool st = STORE && valid;
        float* row_y = st ? a.act_h + gpt * H : nullptr;
        float* row_c = st ? a.act_c + gpt * H : nullptr;

        const f32x4* wp = reinterpret_cast<const f32x4*>(a.packed);
        const float* bias = a.bias;
        // mapping hidden layer: 256 wide, LeakyReLU(0.2)
        Act<8> m;
        {
            const f32x16 feat = input_tile(a, b, 0, px, py, pz, h);
            if (st) store_tile_rows(a.act_feat + gpt * 32, 0, h, feat);
#pragma unroll
            for (int t = 0; t < 8; ++t) m.v[t] = load_chan16(bias, t, h);
            layer0_accumulate<8>(wp, 1, 0, feat, m.v, lane);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) m.v[t][r] = m.v[t][r] > 0.0f ? m.v[t][r] : m.v[t][r] * 0.2f;
            if (st) {
                float* row_m = a.act_h + (size_t)a.L * slab + gpt * 256;
#pragma unroll
                for (int t = 0; t < 8; ++t) store_tile_rows(row_m, t, h, m.v[t]);
            }
            wp += 8 * TILE4;
            bias += 256;
        }
        Act<NT> x, y;
        f32x16 xyz;
#pragma unroll
        for (int r = 0; r < 16; ++r) xyz[r] = 0.0f;
        if (h == 0) {
            xyz[0] = px;
            xyz[1] = py;
            xyz[2] = pz;
        }
        for (int l = 0; l < a.L; ++l) {
            if (l == 0) {
                const f32x4* w_main = wp;
                const f32x4* w_freq = w_main + (size_t)NT * 1 * TILE4;
                const f32x4* w_phase = w_freq + (size_t)NT * 8 * TILE4;
                pfilm_layer<NT, 1, STORE, DROP>(w_main, w_freq, w_phase, bias, bias + H, bias + 2 * H, &xyz, m.v, x.v, lane, h, row_y, row_c, slab, a, drop_gp, l);
                wp = w_phase + (size_t)NT * 8 * TILE4;
            } else {
                const f32x4* w_main = wp;
                const f32x4* w_freq = w_main + (size_t)NT * NT * TILE4;
                const f32x4* w_phase = w_freq + (size_t)NT * 8 * TILE4;
                pfilm_layer<NT, NT, STORE, DROP>(w_main, w_freq, w_phase, bias, bias + H, bias + 2 * H, x.v, m.v, y.v, lane, h, row_y, row_c, slab, a, drop_gp, l);
                wp = w_phase + (size_t)NT * 8 * TILE4;
#pragma unroll
                for (int t = 0; t < NT; ++t) x.v[t] = y.v[t];
            }
            bias += 3 * H;
            if (st) {
                row_y += slab;
                row_c += 3 * slab;
            }
        }
        // head
        const f32x4 acc = head_forward<NT>(wp, bias, x.v, lane);
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

// ---------------------------------------------------------------------------------------------------------------
// Backward of the point pass for one 32-point tile per wave (FiLM / sine layers; residual blocks are not supported yet).
//
//   go' = d loss / d head pre-activation (sigmoid' applied)                          -> act_go (n,4)
//   g_h_L = W_head^T go'            (MFMA, K = 4 outputs in two k-steps)
//   for l = L..1:  g_arg_l = g_h_l * cos(arg_l)   -> act_g[l] (n,H)   (the weight / FiLM gradients are GEMMs and column
//                  g_pre_l = g_arg_l * freq_l                           sums over these buffers: done by the caller with
//                  g_h_{l-1} = W_l^T g_pre_l   (MFMA, transposed pack)  rocBLAS, they are plain library GEMMs)
//   g_feat (32 ch) -> trilinear scatter-add into the channel-last gradient volume with fp32 atomics; the tile is
//   transposed through LDS so that one wave instruction adds two whole 128-byte corner lines (the shape the memory-side
//   atomic units run at full rate for: MI355X_MICROARCH.md "Global float atomics").
// The same chaining as the forward applies: the accumulator registers of g_h are the B operands of the next product.
// ---------------------------------------------------------------------------------------------------------------
template <int OT, int KT>
__device__ __forceinline__ void bwd_matrix(const f32x4* __restrict__ wp, const f32x16* in, f32x16* out, int lane) {
    constexpr int GPT = KT * 4;
    constexpr int NG = OT * GPT;
    f32x4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i)
        if (i < NG) ring[i] = wp[i * 64 + lane];
#pragma unroll
    for (int t = 0; t < OT; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int gi = 0; gi < GPT; ++gi) {
            const int tk = gi >> 2, g = gi & 3;
            const int idx = t * GPT + gi;
            const f32x4 a = ring[idx % RING];
            if (idx + RING < NG) ring[idx % RING] = wp[(idx + RING) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], in[tk][4 * g + e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        out[t] = acc;
    }
}

// g_arg = g_h * cos(arg) (stored), g_pre = g_arg * freq       for all NT tiles of one layer
// (`valid`: padded lanes of an image's last tile shadow its last point -- same row addresses -- with zero gradients; they must
// not store, or their zeros race with the real row of that point)
template <int NT>
__device__ __forceinline__ void bwd_activation(f32x16* g, const float* __restrict__ row_c, float* __restrict__ row_g,
                                               const float* __restrict__ freq, int h, bool valid) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x16 ga;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(row_c + 32 * t + 8 * gq + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) ga[4 * gq + e] = g[t][4 * gq + e] * c[e];
        }
        if (valid) store_tile_rows(row_g, t, h, ga);
        const f32x16 fr = load_chan16(freq, t, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) g[t][r] = ga[r] * fr[r];
    }
}

// go' = d loss / d head pre-activation (stored), g = W_head^T go'; returns the transposed-weight stream behind the head
template <int NT>
__device__ __forceinline__ const float* head_backward(const FieldArgs& a, size_t gpt, bool valid, int h, int lane, f32x16* g) {
    f32x4 go = *reinterpret_cast<const f32x4*>(a.grad_out + gpt * 4);
    if (!valid) go = f32x4{0.f, 0.f, 0.f, 0.f};        // padded lanes shadow the last point: they must add nothing
    if (a.flags & CNERF_F_SIGMOID_RGB) {
        const f32x4 so = *reinterpret_cast<const f32x4*>(a.saved_out + gpt * 4);
        go[0] = go[0] * (so[0] * (1.0f - so[0]));
        go[1] = go[1] * (so[1] * (1.0f - so[1]));
        go[2] = go[2] * (so[2] * (1.0f - so[2]));
    }
    if (valid && h == 0) *reinterpret_cast<f32x4*>(a.act_go + gpt * 4) = go;
    // packed_t layout: [head^T: NT tiles x 2 k-steps x 64 lanes floats, padded to float4 groups][layers L..1 transposed]
    const float* wt = a.packed_t;
    const float b0 = h ? go[1] : go[0], b1 = h ? go[3] : go[2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[(t * 2 + 0) * 64 + lane], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[(t * 2 + 1) * 64 + lane], b1, acc, 0, 0, 0);
        g[t] = acc;
    }
    return wt + (size_t)NT * 2 * 64;
}

// ---------------------------------------------------------------------------------------------------------------
// Gradient chain of the per-point FiLM family (TALLSIREN) over the rows field_pw_kernel<NT, true> stored.  Per layer, from
// g_y = d loss / d y_l:   g_phase = g_y cos,  g_pre = g_y (cos freq),  g_rawfreq = g_y (cos 15 pre)   (three stored rows),
// g_y_{l-1} = W_l^T g_pre (MFMA on the transposed pack).  act_g: L slabs (n,H) of g_pre -- dW_l = g_pre^T y_{l-1}, db_l its
// column sums: cnerf_weight_grad -- followed by G (n, 2 L H) = [g_rawfreq of layers 0..L-1 | g_phase of layers 0..L-1], i.e.
// d loss / d (output of the mapping network's second Linear) row by row in that Linear's own output order: its weight
// gradient G^T m and the gradient G Wm2 that continues into the mapping MLP, the looked-up feature and the volume are
// plain GEMMs over this buffer, left to the caller (library GEMMs + cnerf_scatter_features).  The sample positions carry
// no gradient (generators.py:57,111), so the chain stops at layer 0.
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void field_pw_backward_kernel(FieldArgs a) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    constexpr int H = NT * 32;
    constexpr size_t TILE4 = 4 * 64;
    const TileRange tr = tile_range(a.total_tiles);
    for (long long tile = tr.begin; tile < tr.end; tile += tr.stride) {
        const int b = (int)(tile / a.tiles_per_image);
        const long long n = (tile - (long long)b * a.tiles_per_image) * 32 + j;
        const bool valid = n < a.n_per_image;
        const long long nn = valid ? n : (a.n_per_image - 1);
        const size_t gpt = (size_t)b * a.n_per_image + nn;
        const size_t slab = (size_t)a.act_points * H;
        Act<NT> g, g2;
        const f32x4* wp = reinterpret_cast<const f32x4*>(head_backward<NT>(a, gpt, valid, h, lane, g.v));
        float* row_G = a.act_g + (size_t)a.L * slab + gpt * (size_t)(2 * a.L * H);
        for (int l = a.L - 1; l >= 0; --l) {
            const float* rc = a.act_c + (size_t)(3 * l) * slab + gpt * H;
            float* row_gp = a.act_g + (size_t)l * slab + gpt * H;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 gph, gpre, gfr;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int o = 32 * t + 8 * gq + 4 * h;
                    const f32x4 c = *reinterpret_cast<const f32x4*>(rc + o);
                    const f32x4 cf = *reinterpret_cast<const f32x4*>(rc + slab + o);
                    const f32x4 cp = *reinterpret_cast<const f32x4*>(rc + 2 * slab + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gy = g.v[t][4 * gq + e];
                        gph[4 * gq + e] = gy * c[e];
                        gpre[4 * gq + e] = gy * cf[e];
                        gfr[4 * gq + e] = gy * cp[e];
                    }
                }
                if (valid) {
                    store_tile_rows(row_gp, t, h, gpre);
                    store_tile_rows(row_G + (size_t)l * H, t, h, gfr);
                    store_tile_rows(row_G + (size_t)(a.L + l) * H, t, h, gph);
                }
                g.v[t] = gpre;
            }
            if (l > 0) {
                bwd_matrix<NT, NT>(wp, g.v, g2.v, lane);
                wp += (size_t)NT * NT * TILE4;
#pragma unroll
                for (int t = 0; t < NT; ++t) g.v[t] = g2.v[t];
            }
        }
    }
}

template <int NT, bool HAS_RES>
__global__ __launch_bounds__(256) void field_backward_kernel(FieldArgs a) {
    __shared__ float s_g[4][32][33];     // per wave: g_feat [point][channel] (padded)
    __shared__ int s_base[4][32][8];
    __shared__ float s_w[4][32][8];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    constexpr int H = NT * 32;
    constexpr size_t TILE4 = 4 * 64;
    const TileRange tr = tile_range(a.total_tiles);
    for (long long tile = tr.begin; tile < tr.end; tile += tr.stride) {
        const int b = (int)(tile / a.tiles_per_image);
        const long long n = (tile - (long long)b * a.tiles_per_image) * 32 + j;
        const bool valid = n < a.n_per_image;
        const long long nn = valid ? n : (a.n_per_image - 1);
        const size_t gpt = (size_t)b * a.n_per_image + nn;
        const size_t act_layer = (size_t)a.act_points * H;

        // ---- head backward -------------------------------------------------------------------------------------------
        Act<NT> g, g2;
        const f32x4* wp = reinterpret_cast<const f32x4*>(head_backward<NT>(a, gpt, valid, h, lane, g.v));
        const float* ones = a.bias + a.bias_floats;
        int film_idx = 0, sub = -1;       // sub: index of the activation slab (a residual block owns two)
        for (int l = 0; l < a.L; ++l) {
            film_idx += (a.layer_kind[l] == CNERF_LAYER_FILM);
            sub += (a.layer_kind[l] == CNERF_LAYER_RES) ? 2 : 1;
        }
        for (int l = a.L - 1; l >= 0; --l) {
            const int kind = a.layer_kind[l];
            if (HAS_RES && kind == CNERF_LAYER_RES) {
                // y = sin(W1 x + b1) [slab sub-1];  x' = sin(x + W2 y + b2) [slab sub]
                float* row_gb = a.act_g + (size_t)sub * act_layer + gpt * H;
                bwd_activation<NT>(g.v, a.act_c + (size_t)sub * act_layer + gpt * H, row_gb, ones, h, valid);  // g = g_arg_b
                bwd_matrix<NT, NT>(wp, g.v, g2.v, lane);                                                       // g2 = W2^T g_arg_b
                wp += (size_t)NT * NT * TILE4;
                bwd_activation<NT>(g2.v, a.act_c + (size_t)(sub - 1) * act_layer + gpt * H,
                                   a.act_g + (size_t)(sub - 1) * act_layer + gpt * H, ones, h, valid);         // g2 = g_arg_a
                bwd_matrix<NT, NT>(wp, g2.v, g.v, lane);                                                       // g = W1^T g_arg_a
                wp += (size_t)NT * NT * TILE4;
#pragma unroll
                for (int t = 0; t < NT; ++t) {       // + the identity path: g_arg_b, re-read from the rows this lane just wrote
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 q = *reinterpret_cast<const volatile f32x4*>(row_gb + 32 * t + 8 * gq + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) g.v[t][4 * gq + e] = g.v[t][4 * gq + e] + q[e];
                    }
                }
                sub -= 2;
                continue;
            }
            const bool film = kind == CNERF_LAYER_FILM;
            if (film) --film_idx;
            const float* fr = film ? a.freq + (size_t)b * a.film_stride + (size_t)film_idx * H : ones;
            bwd_activation<NT>(g.v, a.act_c + (size_t)sub * act_layer + gpt * H, a.act_g + (size_t)sub * act_layer + gpt * H, fr, h, valid);
            --sub;
            if (l > 0) {
                bwd_matrix<NT, NT>(wp, g.v, g2.v, lane);
                wp += (size_t)NT * NT * TILE4;
#pragma unroll
                for (int t = 0; t < NT; ++t) g.v[t] = g2.v[t];
            }
        }
        // ---- layer 0: one 32-channel gradient tile per input tile; feature tiles are scattered, the xyz tile is dropped ----
        float px, py, pz;
        tile_point(a, b, nn, valid, h, false, px, py, pz);
        const int ch = lane & 31;
        for (int tk = 0; tk < a.n_in; ++tk) {
            const int lvl = a.in_level[tk];
            if (lvl < 0) continue;                              // no gradient flows to the sample positions
            f32x16 gfeat;
            bwd_matrix<1, NT>(wp + (size_t)tk * NT * TILE4, g.v, &gfeat, lane);
            const int V = a.lvl_V[lvl], C = a.lvl_C[lvl];
            Corner8 cr;
            trilinear_corners(px, py, pz, a.half_voxel, V, cr);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int e = 0; e < 4; ++e) s_g[wv][j][8 * gq + 4 * h + e] = gfeat[4 * gq + e];
            if (h == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    s_base[wv][j][k] = cr.base[k];
                    s_w[wv][j][k] = valid ? cr.w[k] : 0.0f;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float* gv = a.lvl_grad[lvl] + (size_t)b * V * V * V * C + a.in_chan[tk];
            for (int pp = 0; pp < 16; ++pp) {
                const int p = 2 * pp + h;                      // two points per wave instruction, 32 channels each
                const float gval = s_g[wv][p][ch];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float wk = s_w[wv][p][k];
                    if (wk != 0.0f) atomicAdd(gv + (size_t)s_base[wv][p][k] * C + ch, gval * wk);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host-side launchers (called from cnerf_abi.hip)
// ---------------------------------------------------------------------------------------------------------------
// Transposed pack: the matrix M = W^T (rows = inputs of the layer, cols = outputs) in the same A-operand order:
//   value[e] = M[32t + (lane&31)][32tk + 8g + 4(lane>>5) + e] = W[32tk + 8g + 4(lane>>5) + e][32t + (lane&31)]
// w is (n_rows_w, n_cols_w) row-major; OT = n_cols_w / 32 output tiles, KT = n_rows_w / 32.
__global__ void pack_matrix_t_kernel(const float* __restrict__ w, int n_rows_w, int n_cols_w, int OT, float* __restrict__ dst) {
    const int KT = n_rows_w / 32;
    const int total = OT * KT * 4 * 64 * 4;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const int lane = (idx >> 2) & 63;
        const int g = (idx >> 8) & 3;
        const int rest = idx >> 10;
        const int tk = rest % KT, t = rest / KT;
        const int mrow = 32 * t + (lane & 31);                       // column of W (>= n_cols_w: padding)
        const int mcol = 32 * tk + 8 * g + 4 * (lane >> 5) + e;      // row of W
        dst[idx] = (mrow < n_cols_w) ? w[(size_t)mcol * n_cols_w + mrow] : 0.0f;
    }
}

// head^T for the backward: float index (t*2 + s)*64 + lane = W_head[2s + (lane>>5)][32t + (lane&31)]
__global__ void pack_head_t_kernel(const float* __restrict__ w, int H, float* __restrict__ dst) {
    const int total = (H / 32) * 2 * 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) {
        const int lane = idx & 63, s = (idx >> 6) & 1, t = idx >> 7;
        dst[idx] = w[(size_t)(2 * s + (lane >> 5)) * H + 32 * t + (lane & 31)];
    }
}

hipError_t launch_pack_matrix_t(const float* w, int n_rows_w, int n_cols_w, int OT, float* dst, hipStream_t stream) {
    const int total = OT * (n_rows_w / 32) * 4 * 64 * 4;
    hipLaunchKernelGGL(pack_matrix_t_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, n_rows_w, n_cols_w, OT, dst);
    return hipGetLastError();
}

hipError_t launch_pack_head(const float* w, int H, float* dst, hipStream_t stream) {
    const int total = (H / 2) * 64;
    hipLaunchKernelGGL(pack_head_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, H, dst);
    return hipGetLastError();
}

hipError_t launch_pack_head_t(const float* w, int H, float* dst, hipStream_t stream) {
    const int total = (H / 32) * 2 * 64;
    hipLaunchKernelGGL(pack_head_t_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, H, dst);
    return hipGetLastError();
}

__global__ void fill_kernel(float* dst, float value, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

hipError_t launch_fill(float* dst, float value, int n, hipStream_t stream) {
    hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dst, value, n);
    return hipGetLastError();
}

// folded FiLM constants of a call (see folded_sine): out = [Mh | Ml | K | Kb], each (B, n_mats, H): one H-vector per image and MATRIX;
// film_of[m] = index of matrix m's FiLM vectors inside an image's (film_stride) freq / phase rows, or -1 (plain sine layer,
// residual matrix: freq = 1, phase = 0)
struct FoldMap {
    int film_of[2 * CNERF_MAX_LAYERS];
};
__global__ void fold_film_kernel(const float* __restrict__ freq, const float* __restrict__ phase, const float* __restrict__ bias, FoldMap map,
                                 int B, int n_mats, int H, int film_stride, float* __restrict__ out) {
    const long long n = (long long)B * n_mats * H;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int ch = (int)(i % H), m = (int)((i / H) % n_mats), b = (int)(i / ((long long)H * n_mats));
        const int f = map.film_of[m];
        const double fr = f >= 0 ? (double)freq[(size_t)b * film_stride + (size_t)f * H + ch] : 1.0;
        const double ph = f >= 0 ? (double)phase[(size_t)b * film_stride + (size_t)f * H + ch] : 0.0;
        const double M = fr * 0.15915494309189533577;
        const float mh = (float)M;
        out[i] = mh;
        out[n + i] = (float)(M - (double)mh);
        out[2 * n + i] = (float)(ph * 0.15915494309189533577);
        // WFOLD: the accumulator's starting value, bias included (one rounding)
        out[3 * n + i] = (float)(M * (double)bias[(size_t)m * H + ch] + ph * 0.15915494309189533577);
    }
}

// WFOLD: out[b] = the layer part of the packed weight stream with every row scaled by Mh[b][matrix][row]
// (packed layout: float index ((t * KT + tk) * 4 + g) * 256 + lane * 4 + e  <->  row 32 t + (lane & 31); layer 0 has n_in k-tiles)
__global__ void scale_packed_kernel(const float* __restrict__ packed, const float* __restrict__ mh, const float* __restrict__ ml, int n_in, int NT,
                                    int n_mats, long long layer_floats, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int H = NT * 32;
    const long long l0 = (long long)NT * n_in * 1024, lh = (long long)NT * NT * 1024;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < layer_floats; idx += (long long)gridDim.x * blockDim.x) {
        int m;
        long long li;
        int KT;
        if (idx < l0) {
            m = 0;
            li = idx;
            KT = n_in;
        } else {
            m = 1 + (int)((idx - l0) / lh);
            li = (idx - l0) - (long long)(m - 1) * lh;
            KT = NT;
        }
        const int lane = (int)((li >> 2) & 63);
        const long long rest = li >> 10;
        const int t = (int)(rest / KT);
        const int row = 32 * t + (lane & 31);
        const size_t mi = ((size_t)b * n_mats + m) * H + row;
        out[(size_t)b * layer_floats + idx] = (float)((double)packed[idx] * ((double)mh[mi] + (double)ml[mi]));      // M = Mh + Ml: one rounding
    }
}

hipError_t launch_scale_packed(const FieldArgs& a, int B, int H, const float* mh, long long layer_floats, float* out, hipStream_t stream) {
    long long bx = (layer_floats + 255) / 256;
    if (bx > 1024) bx = 1024;
    const float* ml = mh + (size_t)B * a.n_mats * H;                 // fold = [Mh | Ml | K | Kb]
    hipLaunchKernelGGL(scale_packed_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, stream, a.packed, mh, ml, a.n_in, H / 32, a.n_mats, layer_floats,
                       out);
    return hipGetLastError();
}

hipError_t launch_fold_film(const FieldArgs& a, int B, int H, float* out, hipStream_t stream) {
    FoldMap map;
    int m = 0, films = 0;
    for (int l = 0; l < a.L; ++l) {
        if (a.layer_kind[l] == CNERF_LAYER_RES) {
            map.film_of[m++] = -1;
            map.film_of[m++] = -1;
        } else {
            map.film_of[m++] = a.layer_kind[l] == CNERF_LAYER_FILM ? films++ : -1;
        }
    }
    const long long n = (long long)B * a.n_mats * H;
    hipLaunchKernelGGL(fold_film_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a.freq, a.phase, a.bias, map, B, a.n_mats, H,
                       a.film_stride, out);
    return hipGetLastError();
}

hipError_t launch_pack_matrix(const float* w, int n_out, int K_real, int OT, float* dst, hipStream_t stream) {
    const int K = (K_real + 31) / 32 * 32;
    const int total = OT * (K / 32) * 4 * 64 * 4;
    const int blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_matrix_kernel, dim3(blocks), dim3(256), 0, stream, w, n_out, K_real, K, OT, dst);
    return hipGetLastError();
}

static int field_grid(const void* kernel, long long total_tiles) {
    int dev = 0, cus = 256, per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    const long long want = (total_tiles + 3) / 4, cap = (long long)cus * per_cu;
    int blocks = (int)(want < cap ? want : cap);
    if (blocks < 8) blocks = 8;               // every XCD class owns an eighth of the tiles
    return (blocks + 7) / 8 * 8;
}

static int field_grid_one_per_cu(long long total_tiles) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const long long want = (total_tiles + 3) / 4;
    int blocks = (int)(want < cus ? want : cus);
    if (blocks < 8) blocks = 8;
    return (blocks + 7) / 8 * 8;
}

template <int NT, bool HAS_RES>
static hipError_t launch_field_backward_nt(const FieldArgs& a, hipStream_t stream) {
    const int blocks = field_grid((const void*)field_backward_kernel<NT, HAS_RES>, a.total_tiles);
    hipLaunchKernelGGL((field_backward_kernel<NT, HAS_RES>), dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

static bool has_res(const FieldArgs& a) {
    bool res = false;
    for (int l = 0; l < a.L; ++l) res |= a.layer_kind[l] == CNERF_LAYER_RES;
    return res;
}

template <int NT>
static hipError_t launch_field_pw_backward_nt(const FieldArgs& a, hipStream_t stream) {
    const int blocks = field_grid((const void*)field_pw_backward_kernel<NT>, a.total_tiles);
    hipLaunchKernelGGL((field_pw_backward_kernel<NT>), dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_field_backward(const FieldArgs& a, int H, hipStream_t stream) {
    if (a.layer_kind[0] == CNERF_LAYER_PFILM) {
        switch (H / 32) {
            case 2: return launch_field_pw_backward_nt<2>(a, stream);
            case 4: return launch_field_pw_backward_nt<4>(a, stream);
            case 8: return launch_field_pw_backward_nt<8>(a, stream);
            default: return hipErrorInvalidValue;
        }
    }
    const bool res = has_res(a);
    switch (H / 32) {
        case 2: return res ? launch_field_backward_nt<2, true>(a, stream) : launch_field_backward_nt<2, false>(a, stream);
        case 4: return res ? launch_field_backward_nt<4, true>(a, stream) : launch_field_backward_nt<4, false>(a, stream);
        case 8: return res ? launch_field_backward_nt<8, true>(a, stream) : launch_field_backward_nt<8, false>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

template <int NT, bool HAS_RES, bool STORE, bool DROP, bool FOLD = false, int WFOLD = 0>
static hipError_t launch_field_tile(const FieldArgs& a, hipStream_t stream) {
    if (a.n_in < 1 || a.in_level[0] < 0) return hipErrorInvalidValue;      // the lookup prefetch assumes a volume tile first
    const void* fn = (const void*)field_tile_kernel<NT, HAS_RES, STORE, DROP, FOLD, WFOLD>;
    const int lds_bytes = 0;
    // (per launch, not once per process: the attribute is per device, and a cached flag would be unsynchronised global state)
    if (lds_bytes)
        if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) return e;
    const int blocks = lds_bytes ? field_grid_one_per_cu(a.total_tiles) : field_grid(fn, a.total_tiles);
    hipLaunchKernelGGL((field_tile_kernel<NT, HAS_RES, STORE, DROP, FOLD, WFOLD>), dim3(blocks), dim3(256), lds_bytes, stream, a);
    return hipGetLastError();
}

template <int NT, bool HAS_RES>
static hipError_t launch_field_nt(const FieldArgs& a, hipStream_t stream) {
    // a.act_h set: activation-storing forward of the backward pass; a.drop_scale != 0: dropout (training mode)
    if (a.drop_scale != 0.0f)
        return a.act_h ? launch_field_tile<NT, HAS_RES, true, true>(a, stream) : launch_field_tile<NT, HAS_RES, false, true>(a, stream);
    if (a.fold && a.packed_img && !a.act_h) {      // (which reduction in front of v_sin: wfolded_sine above)
        if constexpr (HAS_RES) return launch_field_tile<NT, true, false, false, true, 2>(a, stream);
        else return a.freq ? launch_field_tile<NT, false, false, false, true, 1>(a, stream) : launch_field_tile<NT, false, false, false, true, 2>(a, stream);
    }
    if (a.fold && !a.act_h) return launch_field_tile<NT, HAS_RES, false, false, true>(a, stream);
    return a.act_h ? launch_field_tile<NT, HAS_RES, true, false>(a, stream) : launch_field_tile<NT, HAS_RES, false, false>(a, stream);
}

template <int NT, bool STORE, bool DROP>
static hipError_t launch_field_pw(const FieldArgs& a, hipStream_t stream) {
    const int blocks = field_grid((const void*)field_pw_kernel<NT, STORE, DROP>, a.total_tiles);
    hipLaunchKernelGGL((field_pw_kernel<NT, STORE, DROP>), dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

template <int NT>
static hipError_t launch_field_pw_nt(const FieldArgs& a, hipStream_t stream) {
    if (a.drop_scale != 0.0f) return a.act_h ? launch_field_pw<NT, true, true>(a, stream) : launch_field_pw<NT, false, true>(a, stream);
    return a.act_h ? launch_field_pw<NT, true, false>(a, stream) : launch_field_pw<NT, false, false>(a, stream);
}

hipError_t launch_field(const FieldArgs& a, int H, hipStream_t stream) {
    if (a.layer_kind[0] == CNERF_LAYER_PFILM) {
        switch (H / 32) {
            case 2: return launch_field_pw_nt<2>(a, stream);
            case 4: return launch_field_pw_nt<4>(a, stream);
            case 8: return launch_field_pw_nt<8>(a, stream);
            default: return hipErrorInvalidValue;
        }
    }
    const bool res = has_res(a);
    switch (H / 32) {
        case 2: return res ? launch_field_nt<2, true>(a, stream) : launch_field_nt<2, false>(a, stream);
        case 4: return res ? launch_field_nt<4, true>(a, stream) : launch_field_nt<4, false>(a, stream);
        case 8: return res ? launch_field_nt<8, true>(a, stream) : launch_field_nt<8, false>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace cnerf
