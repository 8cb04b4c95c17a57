// scatter_sorted_kernel: the feature-volume gradient of a rendering pass (the backward of the trilinear lookup, siren.py:555-567 ->
// grid_sampler_3d_backward) from the input-tile gradients the fp16 gradient chain stores (`gin`: its layer-0 products in true units, fp32,
// 128 B per point -- exactly the values it otherwise adds to the volume itself), pre-reduced per pixel patch.
//
// Why.  chain16_kernel finishes every 32-point tile (32 consecutive samples of ONE ray, ~1.4 voxels apart: almost no shared corners) with
// 32 channels x 8 corners of fp32 atomics per point: 1 KiB per point at the chip's float-atomic rate (1.3 TB/s), 5.4 of its 17.4 ms per
// launch at batch 8.  Neighbouring PIXELS are 0.4-0.7 voxel apart: an 8 x 8-pixel patch x four strata lands in ~110 voxels (0.43 voxel
// rows added per point instead of 8, measured).
//
// How.  A block owns a patch x depth bin = 256 points.  It sorts their 2048 (voxel, point, weight) corner records by voxel with a counting
// sort on LDS integer atomics (ds_add_rtn_u32: 5-6 cycles per wave instruction; ds_add_f32 takes 194 -- scripts/ubench/lds_atomic.hip --
// which is what stopped a box of fp32 sums in LDS, csrc/experiments/scatter_patch.hip); the points' gradient rows are parked in LDS;
// then each half-wave walks an equal share of the sorted records, 32 channels across its lanes, sums a run of equal voxels in a register
// and adds the run to the volume once (a 128-byte row per half-wave: the shape float atomics run at full rate in).  53.6 KiB of LDS per
// block, three blocks per CU; sums stay fp32.  The window of a block is 8 x 8 x 8 voxels from the smallest voxel index of the 8 corners of
// its frustum piece; a point with a corner outside it (0.3 % at 128 x 128 x 64 in a 64-voxel volume; most points when pixels are
// several voxels apart) is added directly -- the result never depends on the window, only the number of atomics does.
//
// Depth bins.  Coarse pass: bin q = strata 4q .. 4q+3 of every ray (the jitter keeps a sample inside its stratum): one round, one batch.
// Fine pass: the resampled depths of a ray are unordered (inverse-CDF draws), so a block tests all S depths of its 64 rays against its bin
// (16 depth loads in flight per lane, one result bit per round), queues the matches and processes a batch whenever 256 wait; a tail of at
// most 48 points is added directly.  bench.py train_step at batch 8, 128 x 128 x (64 + 64), random cameras: 84.2 ms with the chain's own
// atomics, 77.6 with the coarse pass here, 74.9 with both (the default; CNERF_SCATTER=chain / coarse select the others: A/B and tests).
#include <hip/hip_runtime.h>

#include "cnerf_kernels.hpp"
#include "field_common.hpp"

namespace cnerf {

namespace {
constexpr int SS_BOX = 8;                                   // voxel window per axis
constexpr int SS_VOX = SS_BOX * SS_BOX * SS_BOX;
constexpr int SS_PTS = 256;                                 // points per batch (coarse pass: 64 rays x 4 strata)
constexpr int SS_DIRECT = 48;                               // a last batch of at most this many points skips the sort

struct ScatterSortedArgs {
    FieldArgs f;          // geometry, mode (COARSE / FINE), u_strat / fine_z / philox, levels, gradient volumes (of the launch's first image)
    const float* gin;     // (feature input tiles, points of the launch, 32) fp32
    long long n_points;   // points of the launch = images * n_per_image
    int n_images;
};
}  // namespace

__device__ __forceinline__ int depth_bin(const RayGeom& g, float t, int NQ) {
    const float half = g.S > 1 ? 0.5f * (g.ray_end - g.ray_start) / (float)(g.S - 1) : 0.5f;
    const float x = (t - (g.ray_start - half)) / ((g.ray_end - g.ray_start) + 2.0f * half) * (float)NQ;
    return (int)fminf(fmaxf(floorf(x), 0.0f), (float)(NQ - 1));      // NaN -> 0
}

__global__ __launch_bounds__(256) void scatter_sorted_kernel(ScatterSortedArgs A) {
    extern __shared__ __attribute__((aligned(16))) float ss_smem[];
    const FieldArgs& a = A.f;
    float* rows = ss_smem;                                                       // [256][32] gradient rows of the batch's points
    uint2* rec = reinterpret_cast<uint2*>(rows + SS_PTS * 32);                   // [2048] {voxel << 8 | entry, weight bits}, sorted by voxel
    int* cnt = reinterpret_cast<int*>(rec + SS_PTS * 8);                         // [512] records per voxel, then (in place) their first index
    int* queue = cnt + SS_VOX;                                                   // [512] ring of point indices waiting for a batch
    int* misc = queue + 2 * SS_PTS;                                              // [0] total records, [1] entries outside the window, [2] queue tail, [3] spare
    int* s_corner = misc + 4;                                                    // [8][3]
    unsigned char* outside = reinterpret_cast<unsigned char*>(s_corner + 24);    // [256] the entries outside the window
    // (53.6 KiB in all: three blocks per CU with room for the allocation granule; a fourth KiB would cost the third block)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = a.geom.R, S = a.geom.S;
    const int rows_img = (int)(a.n_per_image / ((long long)R * S));
    const int PC = (R + 7) / 8, PR = (rows_img + 7) / 8, NQ = (S + 3) / 4;
    long long idx = blockIdx.x;
    const int q = (int)(idx % NQ);
    idx /= NQ;
    const int pc = (int)(idx % PC);
    idx /= PC;
    const int pr = (int)(idx % PR);
    const int b = (int)(idx / PR);
    if (b >= A.n_images) return;                                                 // block-uniform

    // thread -> candidate: ray of the patch, sample of a quad
    const int r = wave * 16 + (lane >> 2), s4 = lane & 3;
    const int row = pr * 8 + (r >> 3), col = pc * 8 + (r & 7);
    const bool ray_ok = row < rows_img && col < R;
    const long long ray = ray_ok ? (long long)row * R + col : 0;
    const float* m = a.cam2world + (size_t)b * 16;
    const bool fine = a.mode == FIELD_MODE_FINE;
    const int ch = lane & 31, h = lane >> 5;

    int fi = -1;
    for (int tk = 0; tk < a.n_in; ++tk) {
        const int lvl = a.in_level[tk];
        if (lvl < 0) continue;
        ++fi;
        const int V = a.lvl_V[lvl], C = a.lvl_C[lvl];
        float* gv = a.lvl_grad[lvl] + (size_t)b * V * V * V * C + a.in_chan[tk];
        const float* gin = A.gin + ((size_t)fi * A.n_points + (size_t)b * a.n_per_image) * 32;

        // ---- window origin: smallest voxel index over the 8 corners of the patch-bin's frustum piece -----------------------------------
        if (tid == 0) misc[2] = 0;
        if (tid < 8) {
            const int rc = (tid & 1) ? min(pr * 8 + 7, rows_img - 1) : pr * 8, cc = (tid & 2) ? min(pc * 8 + 7, R - 1) : pc * 8;
            float dx, dy, dz, px, py, pz;
            camera_dir(a.geom, rc, cc, dx, dy, dz);
            if (fine) {
                const float half = S > 1 ? 0.5f * (a.geom.ray_end - a.geom.ray_start) / (float)(S - 1) : 0.5f;
                const float wbin = ((a.geom.ray_end - a.geom.ray_start) + 2.0f * half) / (float)NQ;
                fine_sample(m, dx, dy, dz, (a.geom.ray_start - half) + wbin * (float)(q + ((tid & 4) ? 1 : 0)), px, py, pz);
            } else {
                float zj;
                coarse_sample(a.geom, m, dx, dy, dz, (tid & 4) ? min(4 * q + 3, S - 1) : 4 * q, (tid & 4) ? 1.0f : 0.0f, zj, px, py, pz);
            }
            int i0;
            float lo, hi;
            unnormalize(px, a.half_voxel, V, i0, lo, hi);
            s_corner[tid * 3 + 0] = i0;
            unnormalize(py, a.half_voxel, V, i0, lo, hi);
            s_corner[tid * 3 + 1] = i0;
            unnormalize(pz, a.half_voxel, V, i0, lo, hi);
            s_corner[tid * 3 + 2] = i0;
        }
        __syncthreads();
        int ox = s_corner[0], oy = s_corner[1], oz = s_corner[2];
#pragma unroll
        for (int c = 1; c < 8; ++c) {
            ox = min(ox, s_corner[c * 3 + 0]);
            oy = min(oy, s_corner[c * 3 + 1]);
            oz = min(oz, s_corner[c * 3 + 2]);
        }

        // ---- one batch: the n <= 256 points queue[(base + e) & 511], e = thread ---------------------------------------------------------
        auto process_batch = [&](int base, int n) {
            const bool ok = tid < n;
            const long long nn = ok ? queue[(base + tid) & (2 * SS_PTS - 1)] : 0;
            const TileRaw raw = tile_point_fetch(a, b, nn);
            // the batch's gradient rows: requested now, parked in LDS after the corner arithmetic.  lane -> (row of an 8-row group,
            // 16-byte piece): a wave instruction covers 8 rows of 128 B
            f32x4 piece[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int e = wave * 64 + it * 8 + (lane >> 3);
                const long long nne = e < n ? queue[(base + e) & (2 * SS_PTS - 1)] : 0;
                piece[it] = *reinterpret_cast<const f32x4*>(gin + (size_t)nne * 32 + 4 * (lane & 7));
            }
            cnt[tid] = 0;
            cnt[tid + 256] = 0;
            if (tid == 0) misc[1] = 0;
            __syncthreads();
            // the point's eight corners: counted per voxel
            int vox[8], pos[8];
            float w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) vox[k] = -1;
            if (ok) {
                float px, py, pz;
                tile_point_finish(a, b, nn, raw, true, 0, false, px, py, pz);
                int ix, iy, iz;
                float lx, hx, ly, hy, lz, hz;
                unnormalize(px, a.half_voxel, V, ix, lx, hx);
                unnormalize(py, a.half_voxel, V, iy, ly, hy);
                unnormalize(pz, a.half_voxel, V, iz, lz, hz);
                const int x1 = ix + 1 < V ? 1 : 0, y1 = iy + 1 < V ? 1 : 0, z1 = iz + 1 < V ? 1 : 0;   // trilinear_corners' clamped +1 corners
                const int rx = ix - ox, ry = iy - oy, rz = iz - oz;
                if (rx >= 0 && ry >= 0 && rz >= 0 && rx + x1 < SS_BOX && ry + y1 < SS_BOX && rz + z1 < SS_BOX) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        // a clamped +1 corner coincides with corner 0 and weighs exactly 0: no record
                        const bool act = (!(k & 1) || x1) && (!(k & 2) || y1) && (!(k & 4) || z1);
                        w[k] = ((k & 1) ? lx : hx) * ((k & 2) ? ly : hy) * ((k & 4) ? lz : hz);
                        if (act) {
                            vox[k] = ((rz + ((k >> 2) & 1)) * SS_BOX + ry + ((k >> 1) & 1)) * SS_BOX + rx + (k & 1);
                            pos[k] = atomicAdd(&cnt[vox[k]], 1);
                        }
                    }
                } else {
                    outside[atomicAdd(&misc[1], 1)] = (unsigned char)tid;
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it)
                *reinterpret_cast<f32x4*>(rows + (wave * 64 + it * 8 + (lane >> 3)) * 32 + 4 * (lane & 7)) = piece[it];
            __syncthreads();
            // exclusive scan of the 512 counts, in place (wave 0: 8 consecutive voxels per lane)
            if (wave == 0) {
                int c[8], sum = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    c[k] = cnt[lane * 8 + k];
                    sum += c[k];
                }
                int incl = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int up = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += up;
                }
                int run = incl - sum;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    cnt[lane * 8 + k] = run;
                    run += c[k];
                }
                if (lane == 63) misc[0] = incl;
            }
            __syncthreads();
            // records into their sorted places
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (vox[k] >= 0) rec[cnt[vox[k]] + pos[k]] = uint2{(uint32_t)(vox[k] << 8) | (uint32_t)tid, __float_as_uint(w[k])};
            __syncthreads();
            // every half-wave an equal share of the sorted records: runs of one voxel summed in a register, one volume add per run
            {
                const int total = misc[0];
                const int hw = 2 * wave + h;
                const int lo = (int)((long long)total * hw / 8), hi = (int)((long long)total * (hw + 1) / 8);
                const int span = (total + 7) / 8 + 1;                             // >= hi - lo of both halves (wave-uniform trip count)
                int cur = -1;
                float acc = 0.0f;
                auto flush = [&]() {
                    const int gx = ox + (cur & 7), gy = oy + ((cur >> 3) & 7), gz = oz + (cur >> 6);
                    atomicAdd(gv + ((size_t)(gz * V + gy) * V + gx) * C + ch, acc);
                };
                for (int i0 = 0; i0 < span; i0 += 8) {
                    uint2 rr[8];
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) rr[u] = lo + i0 + u < hi ? rec[lo + i0 + u] : uint2{0xffffff00u, 0u};
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = rows[(rr[u].x & 255u) * 32 + ch];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (lo + i0 + u < hi) {
                            const int v = (int)(rr[u].x >> 8);
                            if (v != cur) {
                                if (cur >= 0) flush();
                                cur = v;
                                acc = 0.0f;
                            }
                            acc += x[u] * __uint_as_float(rr[u].y);
                        }
                    }
                }
                if (cur >= 0) flush();
            }
            // points outside the window: straight to the volume, two per wave instruction (every lane recomputes the point's corners)
            {
                const int n_out = misc[1];
                for (int i = 2 * wave + h; i < n_out; i += 8) {
                    const int e = outside[i];
                    float px, py, pz;
                    tile_point(a, b, (long long)queue[(base + e) & (2 * SS_PTS - 1)], true, 0, false, px, py, pz);
                    Corner8 cr;
                    trilinear_corners(px, py, pz, a.half_voxel, V, cr);
                    const float gval = rows[e * 32 + ch];
#pragma unroll
                    for (int k = 0; k < 8; ++k) atomicAdd(gv + (size_t)cr.base[k] * C + ch, gval * cr.w[k]);
                }
            }
            __syncthreads();                                                     // before rows / rec / cnt / the queue slots are reused
        };

        // ---- a handful of queued points (the tail of a fine-pass bin, a sliver of a patch at the image edge): straight to the volume, two
        // per wave instruction -- a batch costs its barriers and latencies whatever it holds
        auto direct = [&](int base, int n) {
            for (int i = 2 * wave + h; i < n; i += 8) {
                const long long nnd = queue[(base + i) & (2 * SS_PTS - 1)];
                float px, py, pz;
                tile_point(a, b, nnd, true, 0, false, px, py, pz);
                Corner8 cr;
                trilinear_corners(px, py, pz, a.half_voxel, V, cr);
                const float gval = gin[(size_t)nnd * 32 + ch];
#pragma unroll
                for (int k = 0; k < 8; ++k) atomicAdd(gv + (size_t)cr.base[k] * C + ch, gval * cr.w[k]);
            }
        };

        // ---- rounds: a quad of samples per ray; the points of this block's depth bin queue up and leave in batches of 256 -----------------
        // Fine pass: which of the next 16 quads' samples fall in the bin is decided up front from 16 depth loads in flight together
        // (one bit per round in a register), not one dependent load per round.
        const int q_begin = fine ? 0 : q, q_end = fine ? NQ : q + 1;
        int done = 0;
        for (int q0 = q_begin; q0 < q_end; q0 += 16) {
            uint32_t minebits = 0;
            if (fine) {
                float zt[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int sj = 4 * (q0 + j) + s4;
                    zt[j] = a.fine_z[(size_t)b * a.n_per_image + (size_t)(ray * S + (ray_ok && sj < S ? sj : 0))];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (ray_ok && 4 * (q0 + j) + s4 < S && depth_bin(a.geom, zt[j], NQ) == q) minebits |= 1u << j;
            } else {
                minebits = ray_ok && 4 * q + s4 < S ? 1u : 0u;
            }
            const int nr = q_end - q0 < 16 ? q_end - q0 : 16;
            for (int j = 0; j < nr; ++j) {
                const bool mine = (minebits >> j) & 1u;
                const unsigned long long mask = __ballot(mine);
                if (mask) {                                                      // wave-uniform
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&misc[2], __builtin_popcountll(mask));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (mine) queue[(base + __builtin_popcountll(mask & ((1ull << lane) - 1ull))) & (2 * SS_PTS - 1)] = (int)(ray * S + 4 * (q0 + j) + s4);
                }
                __syncthreads();
                int avail = misc[2] - done;                                      // block-uniform
                __syncthreads();                                                 // everybody has read the tail before the next round moves it
                const bool last = q0 + j + 1 == q_end;
                while (avail >= SS_PTS || (last && avail > 0)) {
                    const int n = avail < SS_PTS ? avail : SS_PTS;
                    if (n <= SS_DIRECT) direct(done, n);
                    else process_batch(done, n);
                    done += n;
                    avail -= n;
                }
            }
        }
        __syncthreads();                                                         // the next input tile resets the queue
    }
}

hipError_t launch_scatter_patch(const FieldArgs& f, const float* gin, hipStream_t stream) {
    if (f.mode != FIELD_MODE_COARSE && f.mode != FIELD_MODE_FINE) return hipErrorInvalidValue;
    ScatterSortedArgs A;
    A.f = f;
    A.gin = gin;
    A.n_images = (int)(f.total_tiles / f.tiles_per_image);
    A.n_points = (long long)A.n_images * f.n_per_image;
    const int R = f.geom.R, S = f.geom.S;
    if (R < 1 || S < 1 || f.n_per_image % ((long long)R * S) != 0) return hipErrorInvalidValue;
    const long long rows = f.n_per_image / ((long long)R * S);
    const long long blocks = (long long)A.n_images * ((rows + 7) / 8) * ((R + 7) / 8) * ((S + 3) / 4);
    if (blocks < 1 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds_bytes = (size_t)SS_PTS * 32 * 4 + (size_t)SS_PTS * 8 * 8 + (size_t)SS_VOX * 4 + (size_t)2 * SS_PTS * 4 + 4 * 4 + 8 * 3 * 4 + SS_PTS;
    if (hipError_t e = hipFuncSetAttribute((const void*)scatter_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) return e;
    hipLaunchKernelGGL(scatter_sorted_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, A);
    return hipGetLastError();
}

}  // namespace cnerf
