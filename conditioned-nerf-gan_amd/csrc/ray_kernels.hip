// Per-ray kernels of the render path for gfx950: alpha compositing with a wavefront prefix-product for the
// transmittance, inverse-CDF resampling, depth merge, plus the layout transposes and the unfused trilinear gather.
// One wavefront owns one ray; samples sit on the lanes (64 per chunk), scans are wave-wide DPP/shuffle scans in double.
//
// Replaces fancy_integration (volumetric_rendering.py:18-70), sample_pdf (:297-342) with its call site
// (generators.py:123-137), the cat/sort/gather merge (generators.py:162-167), the epilogue (generators.py:182-186,
// distance2depth volumetric_rendering.py:345-356) and F.grid_sample + permute (siren.py:555-571).
#include "cnerf_dev.hpp"
#include "cnerf_kernels.hpp"

namespace cnerf {

constexpr int MAX_N = 256;               // samples per ray after the merge (2 * S, S <= 128)
constexpr int CHUNKS = MAX_N / WAVE;     // 4
constexpr int RAYS_PER_BLOCK = 4;        // 4 waves

// Each wave only ever touches its own LDS rows, and the LDS executes one wave's instructions in issue order, so a
// write by one lane is visible to a later read by another lane of the SAME wave as long as the compiler keeps the
// program order: a wavefront-scope fence, no s_barrier (waves of a block may exit independently).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct RaySums {
    float r, g, b, dist, wsum;
};

// Composites n samples of one ray.  rs/z/eps are indexed by sample; w_keep[c] receives the weight of sample
// c*64+lane.  Every lane returns the same sums.
// density noise of sample i: injected tensor, or Philox (stream `ph_stream`, element ph_base + i), or none
struct NoiseSrc {
    const float* eps;
    PhiloxKey ph;
    uint32_t ph_stream;
    unsigned long long ph_base;
    __device__ __forceinline__ float at(int i, float noise_std) const {
        if (eps) return eps[i] * noise_std;
        if (ph.on && noise_std != 0.0f) return philox_normal(ph, ph_stream, ph_base + (unsigned long long)i) * noise_std;
        return 0.0f;
    }
};

__device__ __forceinline__ RaySums composite_ray(const f32x4* rs, const float* z, const NoiseSrc& eps, int n,
                                                 float noise_std, uint32_t flags, float (&w_keep)[CHUNKS], int lane) {
    double carry = 1.0;   // product of (1 - alpha + 1e-10) over all earlier chunks
    double wsum_d = 0.0;
    f32x4 c_keep[CHUNKS];
    float z_keep[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int i = c * WAVE + lane;
        const bool act = i < n;
        float alpha = 0.0f, shifted = 1.0f;
        c_keep[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        z_keep[c] = 0.0f;
        if (c * WAVE < n) {
            if (act) {
                const f32x4 v = rs[i];
                const float zi = z[i];
                const float delta = (i == n - 1) ? 1e10f : (z[i + 1] - zi);
                const float noisy = v[3] + eps.at(i, noise_std);
                const float dens = (flags & CNERF_F_SOFTPLUS) ? softplus20(noisy) : fmaxf(noisy, 0.0f);
                alpha = 1.0f - expf(-delta * dens);
                shifted = 1.0f - alpha + 1e-10f;
                c_keep[c] = v;
                z_keep[c] = zi;
            }
            const double incl = wave_incl_prod((double)shifted, lane);
            double excl = __shfl_up(incl, 1, WAVE);
            if (lane == 0) excl = 1.0;
            const float trans = (float)(carry * excl);   // exclusive cumprod, rounded to fp32 like ATen's output
            w_keep[c] = act ? alpha * trans : 0.0f;
            carry *= __shfl(incl, WAVE - 1, WAVE);
            wsum_d += wave_sum((double)w_keep[c]);
        } else {
            w_keep[c] = 0.0f;
        }
    }
    RaySums out;
    out.wsum = (float)wsum_d;
    if (flags & CNERF_F_LAST_BACK) {
        const int last = n - 1;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c)
            if (c * WAVE + lane == last) w_keep[c] = w_keep[c] + (1.0f - out.wsum);
    }
    double r = 0.0, g = 0.0, b = 0.0, d = 0.0;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        if (c * WAVE < n) {
            r += (double)(w_keep[c] * c_keep[c][0]);
            g += (double)(w_keep[c] * c_keep[c][1]);
            b += (double)(w_keep[c] * c_keep[c][2]);
            d += (double)(w_keep[c] * z_keep[c]);
        }
    }
    out.r = (float)wave_sum(r);
    out.g = (float)wave_sum(g);
    out.b = (float)wave_sum(b);
    out.dist = (float)wave_sum(d);
    if (flags & CNERF_F_WHITE_BACK) {
        out.r = out.r + 1.0f - out.wsum;
        out.g = out.g + 1.0f - out.wsum;
        out.b = out.b + 1.0f - out.wsum;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_kernel(CompositeArgs a) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= a.rays) return;
    const f32x4* rs = reinterpret_cast<const f32x4*>(a.rgb_sigma) + ray * a.n;
    const float* z = a.z + ray * a.n;
    const NoiseSrc eps{a.eps ? a.eps + ray * a.n : nullptr, PhiloxKey{0u, 0u, 0u, 0u}, 0u, 0ull};
    float w[CHUNKS];
    const RaySums s = composite_ray(rs, z, eps, a.n, a.noise_std, a.flags, w, lane);
    if (a.weights) {
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c)
            if (c * WAVE + lane < a.n) a.weights[ray * a.n + c * WAVE + lane] = w[c];
    }
    if (lane == 0) {
        if (a.rgb) {
            a.rgb[ray * 3 + 0] = s.r;
            a.rgb[ray * 3 + 1] = s.g;
            a.rgb[ray * 3 + 2] = s.b;
        }
        if (a.dist) a.dist[ray] = s.dist;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Inverse-CDF resampling.  LDS per wave: cdf[S] (S-1 used) and the bin mid-points.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs a) {
    __shared__ float s_cdf[RAYS_PER_BLOCK][MAX_N / 2 + 4];
    __shared__ float s_bin[RAYS_PER_BLOCK][MAX_N / 2 + 4];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * RAYS_PER_BLOCK + wv;
    if (ray >= a.rays) return;   // whole wave exits together; no block-level barrier below
    const int S = a.S;
    const float* z = a.z + ray * S;

    float w[CHUNKS];
    if (a.rgb_sigma) {
        const f32x4* rs = reinterpret_cast<const f32x4*>(a.rgb_sigma) + ray * S;
        const NoiseSrc eps{a.eps ? a.eps + ray * S : nullptr, a.philox, PHILOX_EPS_COARSE, (unsigned long long)ray * S};
        (void)composite_ray(rs, z, eps, S, a.noise_std, a.flags & CNERF_F_SOFTPLUS, w, lane);
    } else {
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const int i = c * WAVE + lane;
            w[c] = (i < S) ? a.weights[ray * S + i] : 0.0f;
        }
    }
    if (a.weights_out) {
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c)
            if (c * WAVE + lane < S) a.weights_out[ray * S + c * WAVE + lane] = w[c];
    }

    // interior weights, eps added by the caller and again inside sample_pdf
    float wi[CHUNKS];
    double tot = 0.0;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int i = c * WAVE + lane;
        const bool interior = (i >= 1) && (i <= S - 2);
        wi[c] = interior ? (w[c] + 1e-5f) + 1e-5f : 0.0f;
        if (c * WAVE < S) tot += wave_sum((double)wi[c]);
    }
    const float total = (float)tot;
    // cdf[0] = 0, cdf[i] = sum_{m<i} pdf[m] with pdf[m] = wi(sample m+1) / total  -> cdf index == sample index i
    double carry = 0.0;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        if (c * WAVE < S) {
            const int i = c * WAVE + lane;
            const bool interior = (i >= 1) && (i <= S - 2);
            const float pdf = interior ? wi[c] / total : 0.0f;
            const double incl = wave_incl_sum((double)pdf, lane);
            if (i <= S - 2) s_cdf[wv][i] = (float)(carry + incl);   // i = 0 gets 0 (pdf forced to 0 there)
            carry += __shfl(incl, WAVE - 1, WAVE);
            if (i <= S - 2) s_bin[wv][i] = 0.5f * (z[i] + z[i + 1]);
        }
    }
    wave_lds_sync();

    const int ncdf = S - 1;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int s = c * WAVE + lane;
        if (s < S) {
            const float u = a.u ? a.u[ray * S + s] : philox_uniform(a.philox, PHILOX_U_FINE, (unsigned long long)ray * S + s);
            int lo = 0, hi = ncdf;   // first index with cdf[idx] >= u
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_cdf[wv][mid] < u) lo = mid + 1; else hi = mid;
            }
            const int below = max(lo - 1, 0), above = min(lo, S - 2);
            const float c0 = s_cdf[wv][below], c1 = s_cdf[wv][above];
            const float b0 = s_bin[wv][below], b1 = s_bin[wv][above];
            float den = c1 - c0;
            if (den < 1e-5f) den = 1.0f;
            a.fine_z[ray * S + s] = b0 + (u - c0) / den * (b1 - b0);
            if (a.inds) a.inds[ray * S + s] = lo;
        }
    }
    if (a.cdf) {
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const int i = c * WAVE + lane;
            if (i < ncdf) a.cdf[ray * ncdf + i] = s_cdf[wv][i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// merge [fine, coarse] by depth (rank sort in LDS) + final composite + output formatting
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_composite_kernel(MergeArgs a) {
    __shared__ float s_z[RAYS_PER_BLOCK][MAX_N];
    __shared__ f32x4 s_rs[RAYS_PER_BLOCK][MAX_N];
    __shared__ float s_zs[RAYS_PER_BLOCK][MAX_N];
    __shared__ f32x4 s_rss[RAYS_PER_BLOCK][MAX_N];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long ray_raw = (long long)blockIdx.x * RAYS_PER_BLOCK + wv;
    const bool live = ray_raw < a.rays;
    const long long ray = live ? ray_raw : a.rays - 1;   // dead waves shadow the last ray and skip the stores
    const int S = a.S;
    const bool hier = a.fine_z != nullptr;
    const int n = hier ? 2 * S : S;

    const f32x4* crs = reinterpret_cast<const f32x4*>(a.coarse_rgb_sigma) + ray * S;
    const float* cz = a.coarse_z + ray * S;
    const f32x4* rs_sorted;
    const float* z_sorted;
    if (hier) {
        const f32x4* frs = reinterpret_cast<const f32x4*>(a.fine_rgb_sigma) + ray * S;
        const float* fz = a.fine_z + ray * S;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const int i = c * WAVE + lane;
            if (i < n) {
                s_z[wv][i] = (i < S) ? fz[i] : cz[i - S];
                s_rs[wv][i] = (i < S) ? frs[i] : crs[i - S];
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const int i = c * WAVE + lane;
            if (i < n) {
                const float zi = s_z[wv][i];
                int rank = 0;
                for (int j = 0; j < n; ++j) {
                    const float zj = s_z[wv][j];
                    rank += (zj < zi) || (zj == zi && j < i);
                }
                s_zs[wv][rank] = zi;
                s_rss[wv][rank] = s_rs[wv][i];
                if (a.sort_idx && live) a.sort_idx[ray * n + rank] = i;
            }
        }
        wave_lds_sync();
        rs_sorted = s_rss[wv];
        z_sorted = s_zs[wv];
    } else {
        rs_sorted = crs;
        z_sorted = cz;
    }
    const NoiseSrc eps{a.eps ? a.eps + ray * n : nullptr, a.philox, PHILOX_EPS_FINAL, (unsigned long long)ray * n};
    float w[CHUNKS];
    const RaySums s = composite_ray(rs_sorted, z_sorted, eps, n, a.noise_std, a.flags, w, lane);
    if (!live) return;
    if (a.final_weights) {
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c)
            if (c * WAVE + lane < n) a.final_weights[ray * n + c * WAVE + lane] = w[c];
    }
    if (lane == 0) {
        const int R = a.geom.R;
        const long long P = (long long)R * R;
        const long long b = ray / P, p = ray - b * P;
        const int row = (int)(p / R), col = (int)(p - (long long)row * R);
        float dx, dy, dz;
        camera_dir(a.geom, row, col, dx, dy, dz);
        float* px = a.pixels + b * 3 * P + p;
        px[0] = s.r * 2.0f - 1.0f;
        px[P] = s.g * 2.0f - 1.0f;
        px[2 * P] = s.b * 2.0f - 1.0f;
        a.depth[ray] = dz * s.dist;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of merge + final composite + epilogue: d(loss)/d(pixels, depth) -> d(loss)/d(rgb_sigma) of the coarse and
// fine samples.  One wave per ray; the forward quantities (alpha, transmittance, weights) are recomputed from the saved
// rgb_sigma / z, the dependence of T_i on earlier alphas is a reverse (suffix) scan.
//   w_i = alpha_i T_i, T_i = prod_{j<i} s_j, s_j = 1 - alpha_j + 1e-10
//   dL/dalpha_i = Gw_i T_i - (sum_{k>i} Gw_k w_k) / s_i ,   dL/dsigma_i = dL/dalpha_i * delta_i exp(-delta_i dens_i) act'(.)
// Autograd twin of fancy_integration (volumetric_rendering.py:18-70) and generators.py:162-186.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_composite_backward_kernel(MergeBwdArgs a) {
    __shared__ float s_z[RAYS_PER_BLOCK][MAX_N];
    __shared__ f32x4 s_rs[RAYS_PER_BLOCK][MAX_N];
    __shared__ float s_zs[RAYS_PER_BLOCK][MAX_N];
    __shared__ f32x4 s_rss[RAYS_PER_BLOCK][MAX_N];
    __shared__ int s_src[RAYS_PER_BLOCK][MAX_N];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long ray_raw = (long long)blockIdx.x * RAYS_PER_BLOCK + wv;
    const bool live = ray_raw < a.rays;
    const long long ray = live ? ray_raw : a.rays - 1;
    const int S = a.S;
    const bool hier = a.fine_z != nullptr;
    const int n = hier ? 2 * S : S;

    const f32x4* crs = reinterpret_cast<const f32x4*>(a.coarse_rgb_sigma) + ray * S;
    const float* cz = a.coarse_z + ray * S;
    if (hier) {
        const f32x4* frs = reinterpret_cast<const f32x4*>(a.fine_rgb_sigma) + ray * S;
        const float* fz = a.fine_z + ray * S;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const int i = c * WAVE + lane;
            if (i < n) {
                s_z[wv][i] = (i < S) ? fz[i] : cz[i - S];
                s_rs[wv][i] = (i < S) ? frs[i] : crs[i - S];
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const int i = c * WAVE + lane;
            if (i < n) {
                const float zi = s_z[wv][i];
                int rank = 0;
                for (int j = 0; j < n; ++j) {
                    const float zj = s_z[wv][j];
                    rank += (zj < zi) || (zj == zi && j < i);
                }
                s_zs[wv][rank] = zi;
                s_rss[wv][rank] = s_rs[wv][i];
                s_src[wv][rank] = i;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            const int i = c * WAVE + lane;
            if (i < n) {
                s_zs[wv][i] = cz[i];
                s_rss[wv][i] = crs[i];
                s_src[wv][i] = i + S;    // "coarse" slot of the cat[fine, coarse] index space
            }
        }
    }
    wave_lds_sync();

    // upstream gradients of this ray
    const int R = a.geom.R;
    const long long P = (long long)R * R;
    const long long b = ray / P, p = ray - b * P;
    const int row = (int)(p / R), col = (int)(p - (long long)row * R);
    float dx, dy, dz;
    camera_dir(a.geom, row, col, dx, dy, dz);
    const float* gp = a.grad_pixels + b * 3 * P + p;
    const float gr = 2.0f * gp[0], gg = 2.0f * gp[P], gb = 2.0f * gp[2 * P];   // pixels = 2 rgb - 1
    const float gd = a.grad_depth ? dz * a.grad_depth[ray] : 0.0f;              // depth = dir_z * dist
    const NoiseSrc eps{a.eps ? a.eps + ray * n : nullptr, a.philox, PHILOX_EPS_FINAL, (unsigned long long)ray * n};

    // forward recompute (same arithmetic as composite_ray) keeping alpha, trans, exp(-delta dens), act'
    float alpha[CHUNKS], trans[CHUNKS], w[CHUNKS], dsig[CHUNKS], sft[CHUNKS];
    f32x4 cv[CHUNKS];
    float zv[CHUNKS];
    double carry = 1.0, wsum_d = 0.0;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int i = c * WAVE + lane;
        const bool act = i < n;
        alpha[c] = 0.f; trans[c] = 0.f; w[c] = 0.f; dsig[c] = 0.f; sft[c] = 1.0f;
        cv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        zv[c] = 0.f;
        if (c * WAVE < n) {
            if (act) {
                const f32x4 v = s_rss[wv][i];
                const float zi = s_zs[wv][i];
                const float delta = (i == n - 1) ? 1e10f : (s_zs[wv][i + 1] - zi);
                const float noisy = v[3] + eps.at(i, a.noise_std);
                float dens, dact;
                if (a.flags & CNERF_F_SOFTPLUS) {
                    dens = softplus20(noisy);
                    dact = noisy > 20.0f ? 1.0f : sigmoidf_(noisy);
                } else {
                    dens = fmaxf(noisy, 0.0f);
                    dact = noisy > 0.0f ? 1.0f : 0.0f;
                }
                const float ex = expf(-delta * dens);
                alpha[c] = 1.0f - ex;
                sft[c] = 1.0f - alpha[c] + 1e-10f;
                dsig[c] = delta * ex * dact;        // d alpha / d sigma
                cv[c] = v;
                zv[c] = zi;
            }
            const double incl = wave_incl_prod((double)sft[c], lane);
            double excl = __shfl_up(incl, 1, WAVE);
            if (lane == 0) excl = 1.0;
            trans[c] = (float)(carry * excl);
            w[c] = act ? alpha[c] * trans[c] : 0.0f;
            carry *= __shfl(incl, WAVE - 1, WAVE);
            wsum_d += wave_sum((double)w[c]);
        }
    }
    const float wsum = (float)wsum_d;
    // dL/dw' (w' = weights after last_back), then dL/dw
    float G[CHUNKS];
    float g_last = 0.0f;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        G[c] = gr * cv[c][0] + gg * cv[c][1] + gb * cv[c][2] + gd * zv[c];
        if (c * WAVE <= n - 1 && n - 1 < (c + 1) * WAVE) g_last = __shfl(G[c], (n - 1) - c * WAVE, WAVE);
    }
    const float white = (a.flags & CNERF_F_WHITE_BACK) ? (gr + gg + gb) : 0.0f;
    float wl[CHUNKS];   // weights as used in rgb = sum w' c (needed for dL/dc)
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int i = c * WAVE + lane;
        wl[c] = w[c];
        if (a.flags & CNERF_F_LAST_BACK) {
            if (i == n - 1) {
                wl[c] = w[c] + (1.0f - wsum);
                G[c] = 0.0f;                 // w'_{n-1} = 1 - sum_{j<n-1} w_j does not depend on w_{n-1}
            } else {
                G[c] = G[c] - g_last;
            }
        }
        G[c] = (i < n) ? G[c] - white : 0.0f;
    }
    // suffix sums of G_k w_k  (k > i), in double, chunk by chunk from the far end
    double tail = 0.0;
    float gsig[CHUNKS];
#pragma unroll
    for (int c = CHUNKS - 1; c >= 0; --c) {
        gsig[c] = 0.0f;
        if (c * WAVE < n) {
            const double v = (double)G[c] * (double)w[c];
            // inclusive suffix within the chunk = total - inclusive prefix + own
            const double pre = wave_incl_sum(v, lane);
            const double tot = __shfl(pre, WAVE - 1, WAVE);
            const double suf_excl = (tot - pre) + tail;          // sum over k > i inside chunk + later chunks
            const float dalpha = G[c] * trans[c] - (float)(suf_excl / (double)sft[c]);
            gsig[c] = dalpha * dsig[c];
            tail += tot;
        }
    }
    if (!live) return;
    // scatter back to the unsorted coarse / fine arrays
    float* gc = a.grad_coarse + ray * S * 4;
    float* gf = hier ? a.grad_fine + ray * S * 4 : nullptr;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int i = c * WAVE + lane;
        if (i < n) {
            const int src = s_src[wv][i];
            f32x4 o = {gr * wl[c], gg * wl[c], gb * wl[c], gsig[c]};
            float* dst = (src < S) ? gf + src * 4 : gc + (src - S) * 4;
            *reinterpret_cast<f32x4*>(dst) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// (B,C,V^3) <-> (B,V^3,C), C == 32: tiles of 32 channels x 64 voxels through LDS
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_cl_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           long long V3, int C, int to_channel_last) {
    __shared__ float tile[32][65];
    const int b = blockIdx.y;
    const int cb = blockIdx.z * 32;                 // this block's group of 32 channels
    const long long v0 = (long long)blockIdx.x * 64;
    const int t = threadIdx.x;
    if (to_channel_last) {
        const float* s = src + ((size_t)b * C + cb) * V3;
        float* d = dst + (size_t)b * C * V3 + cb;
        const int vv = t & 63, c0 = t >> 6;
#pragma unroll
        for (int c = 0; c < 32; c += 4)
            if (v0 + vv < V3) tile[c + c0][vv] = s[(size_t)(c + c0) * V3 + v0 + vv];
        __syncthreads();
        const int cc = t & 31, w0 = t >> 5;
#pragma unroll
        for (int k = 0; k < 64; k += 8)
            if (v0 + k + w0 < V3) d[(size_t)(v0 + k + w0) * C + cc] = tile[cc][k + w0];
    } else {
        const float* s = src + (size_t)b * C * V3 + cb;
        float* d = dst + ((size_t)b * C + cb) * V3;
        const int cc = t & 31, w0 = t >> 5;
#pragma unroll
        for (int k = 0; k < 64; k += 8)
            if (v0 + k + w0 < V3) tile[cc][k + w0] = s[(size_t)(v0 + k + w0) * C + cc];
        __syncthreads();
        const int vv = t & 63, c0 = t >> 6;
#pragma unroll
        for (int c = 0; c < 32; c += 4)
            if (v0 + vv < V3) d[(size_t)(c + c0) * V3 + v0 + vv] = tile[c + c0][vv];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// unfused trilinear gather: 8 lanes per point, 4 channels (one 16-byte load per corner) per lane
// ---------------------------------------------------------------------------------------------------------------
// XCD-aware point ownership (as tile_range() of the fused kernels): blocks b and b + 8 share an XCD, so block class
// b % 8 sweeps one contiguous eighth of the points -- a band of neighbouring rays, whose corner lines then stay in that
// XCD's 4 MiB L2 instead of being fetched by all eight.  Placement only changes speed, never results.
struct PointRange {
    long long begin, end, stride;
};
__device__ __forceinline__ PointRange point_range(long long total) {
    const int per_block = blockDim.x >> 3;             // 8 lanes per point
    const int cls = blockIdx.x & 7, idx_in_cls = blockIdx.x >> 3;
    const int blk_per_cls = (gridDim.x + 7 - cls) / 8;
    PointRange r;
    r.begin = total * cls / 8 + (long long)idx_in_cls * per_block + (threadIdx.x >> 3);
    r.end = total * (cls + 1) / 8;
    r.stride = (long long)blk_per_cls * per_block;
    return r;
}

__device__ __forceinline__ void gather_point(const GatherArgs& a, long long pt, int sub) {
    const long long b = pt / a.n_per_image;
    const float* p = a.points + pt * 3;
    Corner8 cr;
    trilinear_corners(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), a.half_voxel, a.V, cr);
    const float* vol = a.fvol + (size_t)b * a.V * a.V * a.V * 32 + 4 * sub;
    f32x4 q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = *reinterpret_cast<const f32x4*>(vol + (size_t)cr.base[k] * 32);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] + q[k][e] * cr.w[k];
    // streamed once: the 128 B per point written here (and the positions read above) must not evict the volume's corner
    // lines from the XCD's L2, which every neighbouring ray re-reads
    __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(a.feat + pt * 32 + 4 * sub));
}

__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
    const int sub = threadIdx.x & 7;
    const PointRange pr = point_range((long long)a.B * a.n_per_image);
    for (long long pt = pr.begin; pt < pr.end; pt += pr.stride) gather_point(a, pt, sub);
}

// Round 3 measured two re-orderings of this kernel and kept neither (profiles/r03_gather_counters.md): visiting a render's samples
// patch by patch (4 x 4 pixels x 2 depths per block iteration, so that the repeated corner lines of neighbouring pixels would hit
// in L1) 1.23 -> 1.29-1.37 ms per 16.8 M lookups -- the duplicates are in flight together and each still takes its own trip to L2;
// and fetching every distinct corner line of such a patch once into LDS (box-local voxel keys, wave prefix sum for compact slots)
// 1.19 -> 1.54 ms -- a patch has 30-70 distinct lines, not the 20-40 first estimated, and the dependent phases of a tile (positions,
// box, lines, corners) run at 12 waves per CU instead of streaming at full occupancy.  Counters of this kernel: HBM-side traffic
// within 1.1x of compulsory, 1 KiB per point through the texture addresser at 42 % of its rate, half the corner lines miss L1
// at 255 cycles average L2 round trip, the L1 stalled on pending misses for 42 % of its active cycles.

// Adjoint of gather_kernel: grad_fvol[corner k of point] += w_k * grad_feat[point] (8 lanes per point, 4 channels each, so
// the 8 lanes of a point add one whole 128-B corner line per atomic instruction).  Used by the backward of the per-point
// FiLM family, whose MLP gradients are evaluated by library GEMMs on the host side (ops._pfilm_backward).
__global__ __launch_bounds__(256) void scatter_kernel(GatherArgs a, const float* __restrict__ grad_feat, float* __restrict__ grad_fvol) {
    const int sub = threadIdx.x & 7;
    const PointRange pr = point_range((long long)a.B * a.n_per_image);
    for (long long pt = pr.begin; pt < pr.end; pt += pr.stride) {
        const long long b = pt / a.n_per_image;
        const float* p = a.points + pt * 3;
        Corner8 cr;
        trilinear_corners(p[0], p[1], p[2], a.half_voxel, a.V, cr);
        const f32x4 g = *reinterpret_cast<const f32x4*>(grad_feat + pt * 32 + 4 * sub);
        float* vol = grad_fvol + (size_t)b * a.V * a.V * a.V * 32 + 4 * sub;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float* dst = vol + (size_t)cr.base[k] * 32;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dst + e, g[e] * cr.w[k]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the draws as tensors (tests, and hosts that want to look at them): out[i] = uniform / normal of element i of a stream
__global__ void philox_fill_kernel(PhiloxKey k, uint32_t stream_id, long long n, int normal, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = normal ? philox_normal(k, stream_id, (unsigned long long)i) : philox_uniform(k, stream_id, (unsigned long long)i);
}
hipError_t launch_philox_fill(const PhiloxKey& k, uint32_t stream_id, long long n, int normal, float* out, hipStream_t stream) {
    long long blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, k, stream_id, n, normal, out);
    return hipGetLastError();
}

hipError_t launch_composite(const CompositeArgs& a, hipStream_t stream) {
    const unsigned blocks = (unsigned)((a.rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    hipLaunchKernelGGL(composite_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_resample(const ResampleArgs& a, hipStream_t stream) {
    const unsigned blocks = (unsigned)((a.rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    hipLaunchKernelGGL(resample_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_merge_composite(const MergeArgs& a, hipStream_t stream) {
    const unsigned blocks = (unsigned)((a.rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    hipLaunchKernelGGL(merge_composite_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_merge_composite_backward(const MergeBwdArgs& a, hipStream_t stream) {
    const unsigned blocks = (unsigned)((a.rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
    hipLaunchKernelGGL(merge_composite_backward_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_transpose_cl(int B, int C, int V, const float* src, float* dst, bool to_channel_last,
                               hipStream_t stream) {
    if (C < 32 || C % 32 != 0) return hipErrorInvalidValue;
    const long long V3 = (long long)V * V * V;
    dim3 grid((unsigned)((V3 + 63) / 64), (unsigned)B, (unsigned)(C / 32));
    hipLaunchKernelGGL(transpose_cl_kernel, grid, dim3(256), 0, stream, src, dst, V3, C, to_channel_last ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_gather(const GatherArgs& a, hipStream_t stream) {
    const long long total = (long long)a.B * a.n_per_image;
    long long blocks = (total + 31) / 32;
    if (blocks > 256 * 16) blocks = 256 * 16;
    blocks = (blocks + 7) / 8 * 8;                    // every XCD class owns an eighth of the points
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_scatter(const GatherArgs& a, const float* grad_feat, float* grad_fvol, hipStream_t stream) {
    const long long total = (long long)a.B * a.n_per_image;
    long long blocks = (total + 31) / 32;
    if (blocks > 65536) blocks = 65536;
    blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, grad_feat, grad_fvol);
    return hipGetLastError();
}

}  // namespace cnerf
