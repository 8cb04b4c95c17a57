// EXPERIMENT, NOT BUILT (build.py does not list this file; scatter_patch_integration.diff beside it is the wiring into bwd16.hip /
// cnerf_abi.hip it was measured with).  Result, batch 8, 128 x 128 x (64 + 64), fp16x3 forward + fp16 backward, scripts/time_backward.py:
//   chain16 with its own atomics (the product path)            80.1 ms per step   (chain16 17.4 ms per launch)
//   chain16 storing gin + this kernel                          88.0 ms            (chain16 12.1, this kernel 5.4 coarse / 8.5 fine)
//   ... with the LDS read-add-writes skipped (wrong results)   79.2 ms            -> 4.4 ms per launch are the owner's serial LDS round trips
//   ... with no point entering a list (empty kernel)           73.4 ms            -> the floor: what a free pre-reduction would give (-8 %)
//   first version (ds_add_f32 into the box, no owners)         98.9 ms            (11.0 / 14.7 ms per launch; 99.7 % of the points in the box,
//                                                                                  0.43 flushed voxel rows per point: the reduction itself works)
// ds_add_f32 retires one wave instruction per 194 cycles per CU on gfx950, ds_add_u32 per 5-6 (scripts/ubench/lds_atomic.hip): an exact
// fp32 box needs owners, and eight owners (half-waves) per block doing dependent LDS round trips at two blocks per CU (the 64 KiB box)
// are slower than the atomics they replace.  What won instead: a counting sort of the (voxel, point) records by LDS integer atomics and
// per-run sums in registers (no box) -- csrc/scatter_patch.hip, the shipped kernel.  A fixed-point box (ds_add_u32) would change the
// sums' rounding: rejected.
//
// scatter_patch_kernel: the feature-volume gradient of a rendering pass (the backward of the trilinear lookup, siren.py:555-567 ->
// grid_sampler_3d_backward) from the stored input-tile gradients of the fp16 gradient chain, pre-reduced in LDS.
//
// Why a kernel of its own.  chain16_kernel used to finish every 32-point tile (32 consecutive samples of ONE ray) with 32 channels x
// 8 corners of fp32 atomics per point: 1 KiB of atomic traffic per point at the chip's float-atomic rate (1.3 TB/s), issued by a
// kernel that runs one wave per SIMD -- 5.4 of its 17.4 ms per launch at batch 8.  Consecutive samples of a ray are ~1.4 voxels apart
// and share few corners, but neighbouring PIXELS are 0.4-0.7 voxel apart: an 8 x 8 pixel patch x one depth bin (four strata) lands in
// ~110 voxels.  Here a block owns such a patch-bin, sums its points' corner contributions in an 8 x 8 x 8-voxel box of fp32 in LDS and
// flushes the touched voxels once: 0.43 voxel rows of global atomics per point instead of 8 (measured), exactly the same addends
// (the chain writes its layer-0 products in true units, fp32, 128 B per point: `gin`).
//
// How the box is summed.  LDS float atomics are no option on gfx950: ds_add_f32 retires one wave instruction per 194 cycles per CU
// (three cycles a lane; ds_add_u32 takes 5-6, a plain read-add-write 20: scripts/ubench/lds_atomic.hip) -- the first version of this
// kernel spent 11-15 ms per launch in them.  Instead every box voxel has ONE owner: the box is cut into 8 slices along the axis in
// which the patch-bin extends furthest, half-wave s owns slice s, and a point is appended (LDS integer atomics: two per point) to the
// list of each slice its corners fall in (one or two).  A half-wave walks its list and adds its four corners by plain LDS
// read-add-write, 32 channels across the lanes: no two lanes share an address, the owner is the only writer, and one wave's LDS
// operations execute in order.  Sums are fp32 throughout.
//
// Depth bins.  Coarse pass: bin q = strata 4q .. 4q+3 of every ray (the jitter keeps a sample inside its stratum): one round.  Fine
// pass: the resampled depths of a ray are unordered (inverse-CDF draws), so a block tests all S depths of its 64 rays against its
// bin, a quad of samples per round -- 4 B x S/4 re-reads per point, nothing against the 128 B of the gradient row.  The box origin is
// the smallest voxel index of the 8 corners of the patch-bin's frustum piece; a point with a corner outside the box (0.3 % at
// 128 x 128 x 64 in a 64-voxel volume) goes to global memory directly -- the result does not depend on the box, only the atomic count.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "cnerf_kernels.hpp"
#include "field_common.hpp"

namespace cnerf {

namespace {
constexpr int SP_BOX = 8;                                   // voxels per box edge = slices = half-waves of a block
constexpr int SP_BOX_FLOATS = SP_BOX * SP_BOX * SP_BOX * 32;
constexpr int SP_ENTRY = 9;                                 // words per point entry
constexpr int SP_ROUND = 256;                               // points per round (64 rays x 4 samples)

struct ScatterPatchArgs {
    FieldArgs f;          // geometry, mode, u_strat / fine_z / philox, levels, gradient volumes (of the launch's first image)
    const float* gin;     // (feature input tiles, points of the launch, 32) fp32
    long long n_points;   // points of the launch = images * n_per_image
    int n_images;
    int dbg;
};

__device__ __forceinline__ int depth_bin(const RayGeom& g, float t, int NQ) {
    const float half = g.S > 1 ? 0.5f * (g.ray_end - g.ray_start) / (float)(g.S - 1) : 0.5f;
    const float x = (t - (g.ray_start - half)) / ((g.ray_end - g.ray_start) + 2.0f * half) * (float)NQ;
    return (int)fminf(fmaxf(floorf(x), 0.0f), (float)(NQ - 1));      // NaN -> 0
}

// entry: [0] point index inside the image, [1] voxel of corner 0 (box-relative (rz*8+ry)*8+rx, or volume index when not in the box),
// [2] flags: bit 0 in the box, bits 1..3 the +1 corner exists along x / y / z (else it coincides with corner 0 and weighs exactly 0),
// [3..8] lx, hx, ly, hy, lz, hz: weights of the +1 / +0 corner along each axis (cnerf_dev.hpp unnormalize)
template <int AX>
__device__ __forceinline__ void slice_add(float* box, const uint32_t* E, float gval, int sl, int ch) {
    constexpr int B = AX == 0 ? 1 : 0, C = AX == 2 ? 1 : 2;                 // the two other axes
    constexpr int st[3] = {1, SP_BOX, SP_BOX * SP_BOX};
    const uint32_t base = E[1], fl = E[2];
    const float hiw[3] = {__uint_as_float(E[3]), __uint_as_float(E[5]), __uint_as_float(E[7])};     // +1 corner
    const float low[3] = {__uint_as_float(E[4]), __uint_as_float(E[6]), __uint_as_float(E[8])};     // +0 corner
    const int one[3] = {(int)(fl >> 1) & 1, (int)(fl >> 2) & 1, (int)(fl >> 3) & 1};
    const int side = sl - (int)((base >> (3 * AX)) & 7u);                   // 0: the slice of corner 0; 1: the next one (listed only if one[AX])
    float* p0 = box + (size_t)(base + side * st[AX]) * 32 + ch;
    float W[3];
    W[AX] = side ? hiw[AX] : low[AX];
    float* p[4];
    float w[4], v[4];
    bool act[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int jb = j & 1, jc = j >> 1;
        act[j] = (!jb || one[B]) && (!jc || one[C]);                        // a clamped +1 corner weighs exactly 0: skipped (and it aliases corner 0)
        p[j] = p0 + (jb * one[B] * st[B] + jc * one[C] * st[C]) * 32;
        W[B] = jb ? hiw[B] : low[B];
        W[C] = jc ? hiw[C] : low[C];
        w[j] = W[0] * W[1] * W[2];                                          // wx * wy * wz as trilinear_corners forms it
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *p[j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (act[j]) *p[j] = v[j] + gval * w[j];
}
}  // namespace

__global__ __launch_bounds__(256) void scatter_patch_kernel(ScatterPatchArgs A) {
    extern __shared__ __attribute__((aligned(16))) float sp_smem[];
    const FieldArgs& a = A.f;
    float* box = sp_smem;                                                        // [8][8][8][32]
    uint32_t* ent = reinterpret_cast<uint32_t*>(box + SP_BOX_FLOATS);            // [256][SP_ENTRY]
    unsigned short* slist = reinterpret_cast<unsigned short*>(ent + SP_ROUND * SP_ENTRY);   // [9][256]: slices 0..7, then the points outside the box
    int* cnt2 = reinterpret_cast<int*>(slist + 9 * SP_ROUND);                    // [2][16]: list lengths, double-buffered by round parity
    int* s_corner = cnt2 + 32;                                                   // [8][3]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = a.geom.R, S = a.geom.S;
    const int rows = (int)(a.n_per_image / ((long long)R * S));
    const int PC = (R + 7) / 8, PR = (rows + 7) / 8, NQ = (S + 3) / 4;
    long long idx = blockIdx.x;
    const int q = (int)(idx % NQ);
    idx /= NQ;
    const int pc = (int)(idx % PC);
    idx /= PC;
    const int pr = (int)(idx % PR);
    const int b = (int)(idx / PR);
    if (b >= A.n_images) return;                                                 // block-uniform

    // lane -> (ray of the patch, sample of a quad)
    const int r = wave * 16 + (lane >> 2), s4 = lane & 3;
    const int row = pr * 8 + (r >> 3), col = pc * 8 + (r & 7);
    const bool ray_ok = row < rows && col < R;
    const long long ray = ray_ok ? (long long)row * R + col : 0;
    const float* m = a.cam2world + (size_t)b * 16;
    const bool fine = a.mode == FIELD_MODE_FINE;
    const int ch = lane & 31, h = lane >> 5;
    const int sl = 2 * wave + h;                                                 // the slice this half-wave owns

    for (int i = tid; i < SP_BOX_FLOATS / 4; i += 256) reinterpret_cast<f32x4*>(box)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    int fi = -1;
    for (int tk = 0; tk < a.n_in; ++tk) {
        const int lvl = a.in_level[tk];
        if (lvl < 0) continue;
        ++fi;
        const int V = a.lvl_V[lvl], C = a.lvl_C[lvl];
        float* gv = a.lvl_grad[lvl] + (size_t)b * V * V * V * C + a.in_chan[tk];
        const float* gin = A.gin + ((size_t)fi * A.n_points + (size_t)b * a.n_per_image) * 32;

        // ---- box origin and slicing axis from the 8 corners of the patch-bin ------------------------------------------------------
        if (tid >= 64 && tid < 96) cnt2[tid - 64] = 0;
        if (tid < 8) {
            const int rc = (tid & 1) ? min(pr * 8 + 7, rows - 1) : pr * 8, cc = (tid & 2) ? min(pc * 8 + 7, R - 1) : pc * 8;
            float dx, dy, dz, px, py, pz;
            camera_dir(a.geom, rc, cc, dx, dy, dz);
            if (fine) {
                const float half = S > 1 ? 0.5f * (a.geom.ray_end - a.geom.ray_start) / (float)(S - 1) : 0.5f;
                const float w = ((a.geom.ray_end - a.geom.ray_start) + 2.0f * half) / (float)NQ;
                const float t = (a.geom.ray_start - half) + w * (float)(q + ((tid & 4) ? 1 : 0));
                fine_sample(m, dx, dy, dz, t, px, py, pz);
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
        __syncthreads();                                                         // also: the box is zero (first tile: the fill above; later: the flush)
        int o[3], e3[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            o[d] = s_corner[d];
            e3[d] = s_corner[d];
#pragma unroll
            for (int c = 1; c < 8; ++c) {
                o[d] = min(o[d], s_corner[c * 3 + d]);
                e3[d] = max(e3[d], s_corner[c * 3 + d]);
            }
            e3[d] -= o[d];
        }
        const int ox = o[0], oy = o[1], oz = o[2];
        const int axis = (e3[2] >= e3[1] && e3[2] >= e3[0]) ? 2 : (e3[1] >= e3[0] ? 1 : 0);       // block-uniform

        auto point_of = [&](int qq, bool& ok) -> long long {
            const int s = 4 * qq + s4;
            ok = ray_ok && s < S;
            return ok ? ray * S + s : 0;
        };
        const int q_begin = fine ? 0 : q, q_end = fine ? NQ : q + 1;
        bool ok_next;
        long long nn_next = point_of(q_begin, ok_next);
        TileRaw raw_next = tile_point_fetch(a, b, nn_next);                      // the one dependent load of a round, requested a round ahead
        for (int qq = q_begin; qq < q_end; ++qq) {
            const bool ok = ok_next;
            const long long nn = nn_next;
            const TileRaw raw = raw_next;
            if (qq + 1 < q_end) {
                nn_next = point_of(qq + 1, ok_next);
                raw_next = tile_point_fetch(a, b, nn_next);
            }
            int* cnt = cnt2 + ((qq - q_begin) & 1) * 16;
            __syncthreads();                                                     // nobody still reads the previous round's entries / lists
            // ---- phase A: every lane its candidate point -> entry, appended to the list of each slice it touches ----------------------
            const bool mine = ok && (!fine || depth_bin(a.geom, raw.v[0], NQ) == q);
            if (mine && !(A.dbg & 8)) {
                float px, py, pz;
                tile_point_finish(a, b, nn, raw, true, 0, false, px, py, pz);
                int ix, iy, iz;
                float lx, hx, ly, hy, lz, hz;
                unnormalize(px, a.half_voxel, V, ix, lx, hx);
                unnormalize(py, a.half_voxel, V, iy, ly, hy);
                unnormalize(pz, a.half_voxel, V, iz, lz, hz);
                const int x1 = ix + 1 < V ? 1 : 0, y1 = iy + 1 < V ? 1 : 0, z1 = iz + 1 < V ? 1 : 0;   // trilinear_corners' clamped +1 corners
                const int rx = ix - ox, ry = iy - oy, rz = iz - oz;
                const bool inbox = rx >= 0 && ry >= 0 && rz >= 0 && rx + x1 < SP_BOX && ry + y1 < SP_BOX && rz + z1 < SP_BOX;
                const int id = tid;
                uint32_t* e = ent + id * SP_ENTRY;
                e[0] = (uint32_t)nn;
                e[1] = inbox ? (uint32_t)((rz * SP_BOX + ry) * SP_BOX + rx) : (uint32_t)((iz * V + iy) * V + ix);
                e[2] = (inbox ? 1u : 0u) | (x1 << 1) | (y1 << 2) | (z1 << 3);
                e[3] = __float_as_uint(lx);
                e[4] = __float_as_uint(hx);
                e[5] = __float_as_uint(ly);
                e[6] = __float_as_uint(hy);
                e[7] = __float_as_uint(lz);
                e[8] = __float_as_uint(hz);
                if (inbox) {
                    const int s0 = axis == 2 ? rz : (axis == 1 ? ry : rx), a1 = axis == 2 ? z1 : (axis == 1 ? y1 : x1);
                    slist[s0 * SP_ROUND + atomicAdd(&cnt[s0], 1)] = (unsigned short)id;
                    if (a1) slist[(s0 + 1) * SP_ROUND + atomicAdd(&cnt[s0 + 1], 1)] = (unsigned short)id;
                } else {
                    slist[8 * SP_ROUND + atomicAdd(&cnt[8], 1)] = (unsigned short)id;
                }
            }
            __syncthreads();
            // ---- phase B: each half-wave adds the corners that fall in its slice, 32 channels across its lanes ----------------------
            if (tid < 16) cnt2[(((qq - q_begin) & 1) ^ 1) * 16 + tid] = 0;       // the next round's counters: their last readers have passed this round's first barrier
            const int n = cnt[sl], n_wave = max(cnt[2 * wave], cnt[2 * wave + 1]);
            const unsigned short* my = slist + sl * SP_ROUND;
            for (int i0 = 0; i0 < n_wave; i0 += 8) {
                int id[8];
                float g[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) id[u] = i0 + u < n ? (int)my[i0 + u] : -1;
#pragma unroll
                for (int u = 0; u < 8; ++u) g[u] = (id[u] >= 0 && !(A.dbg & 4)) ? gin[(size_t)ent[id[u] * SP_ENTRY] * 32 + ch] : 0.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (id[u] >= 0 && !(A.dbg & 1)) {
                        const uint32_t* E = ent + id[u] * SP_ENTRY;
                        if (axis == 2) slice_add<2>(box, E, g[u], sl, ch);
                        else if (axis == 1) slice_add<1>(box, E, g[u], sl, ch);
                        else slice_add<0>(box, E, g[u], sl, ch);
                    }
            }
            // points outside the box: straight to the volume, two per wave instruction
            const int n_out = cnt[8];
            for (int i = 2 * wave + h; i < n_out; i += 8) {
                const uint32_t* E = ent + (int)slist[8 * SP_ROUND + i] * SP_ENTRY;
                const float gval = gin[(size_t)E[0] * 32 + ch];
                const uint32_t fl = E[2];
                const float lx = __uint_as_float(E[3]), hx = __uint_as_float(E[4]), ly = __uint_as_float(E[5]), hy = __uint_as_float(E[6]),
                            lz = __uint_as_float(E[7]), hz = __uint_as_float(E[8]);
                const int x1 = (fl >> 1) & 1, y1 = (fl >> 2) & 1, z1 = (fl >> 3) & 1;
                float* gp = gv + (size_t)E[1] * C + ch;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float w = ((k & 1) ? lx : hx) * ((k & 2) ? ly : hy) * ((k & 4) ? lz : hz);
                    const size_t off = (size_t)((k & 1) ? x1 : 0) + (size_t)((k & 2) ? y1 * V : 0) + (size_t)((k & 4) ? z1 * V * V : 0);
                    atomicAdd(gp + off * C, gval * w);
                }
            }
        }
        __syncthreads();

        // ---- flush the touched voxels and leave the box zero ---------------------------------------------------------------------
        for (int v = tid >> 5; v < SP_BOX * SP_BOX * SP_BOX; v += 8) {
            const float val = box[v * 32 + ch];
            if (val != 0.0f) {
                box[v * 32 + ch] = 0.0f;
                if (A.dbg & 2) continue;
                const int gx = ox + (v & 7), gy = oy + ((v >> 3) & 7), gz = oz + (v >> 6);
                atomicAdd(gv + ((size_t)(gz * V + gy) * V + gx) * C + ch, val);
            }
        }
        __syncthreads();
    }
}

hipError_t launch_scatter_patch(const FieldArgs& f, const float* gin, hipStream_t stream) {
    if (f.mode != FIELD_MODE_COARSE && f.mode != FIELD_MODE_FINE) return hipErrorInvalidValue;
    ScatterPatchArgs A;
    A.f = f;
    A.gin = gin;
    A.n_images = (int)(f.total_tiles / f.tiles_per_image);
    A.n_points = (long long)A.n_images * f.n_per_image;
    const int R = f.geom.R, S = f.geom.S;
    if (R < 1 || S < 1 || f.n_per_image % ((long long)R * S) != 0) return hipErrorInvalidValue;
    const long long rows = f.n_per_image / ((long long)R * S);
    const long long blocks = (long long)A.n_images * ((rows + 7) / 8) * ((R + 7) / 8) * ((S + 3) / 4);
    if (blocks < 1 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds_bytes = (size_t)SP_BOX_FLOATS * 4 + (size_t)SP_ROUND * SP_ENTRY * 4 + (size_t)9 * SP_ROUND * 2 + 32 * 4 + 8 * 3 * 4;
    if (hipError_t e = hipFuncSetAttribute((const void*)scatter_patch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) return e;
    const char* dbg = getenv("CNERF_SP_DEBUG");
    A.dbg = dbg ? atoi(dbg) : 0;
    hipLaunchKernelGGL(scatter_patch_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, A);
    return hipGetLastError();
}

}  // namespace cnerf
