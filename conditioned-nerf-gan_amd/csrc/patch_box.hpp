// Distinct voxels of a 32-point tile, without sorting (shared by the patch-ordered lookup and scatter kernels:
// gather_box_kernel in ray_kernels.hip, scatter_patch_kernel in bwd16.hip).
//
// A tile of 4 x 4 neighbouring pixels x 2 consecutive depths asks for 32 x 8 = 256 corner lines of the feature volume, but
// neighbouring pixels are ~0.4 voxel apart at the same depth: typically only 20-40 of the 256 are distinct.  The corners of
// such a patch live in a small box of the voxel grid, so a voxel's KEY is its offset inside the box (PB_BOX^3 <= 256 keys):
// every (point, corner) marks its key in a 256-entry LDS table, a wave-wide exclusive prefix sum over the table hands out
// compact SLOT numbers 0 .. U-1, and slot -> voxel index is recorded.  The lookup kernel then fetches each distinct line ONCE
// into LDS and serves the 256 corner reads from there; the scatter kernel accumulates into one LDS line per distinct voxel and
// flushes each line once.  A tile whose box is larger than PB_BOX per axis or that has more than PB_SLOTS distinct voxels
// (grazing rays, a very fine grid, arbitrary point lists) is reported as not reducible and takes the callers' direct route.
//
// All LDS arrays are private to the calling wave (wavefront-scope fences, no s_barrier).
#pragma once
#include "cnerf_dev.hpp"

namespace cnerf {

constexpr int PB_BOX = 6, PB_SLOTS = 64;

struct PatchBox {
    int x0, y0, z0;     // origin of the box (voxel coordinates)
    int U;              // distinct voxels (valid when reducible)
    bool boxed;         // the tile's corners fit the box: keys are box-local offsets (else `key` holds voxel indices)
    bool reducible;     // boxed && U <= PB_SLOTS: slot[] / svox[] are valid
};

__device__ __forceinline__ void pb_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int pb_voxel_of_key(const PatchBox& pb, int kk, int V) {
    return ((pb.z0 + kk / (PB_BOX * PB_BOX)) * V + (pb.y0 + (kk / PB_BOX) % PB_BOX)) * V + pb.x0 + kk % PB_BOX;
}

// Lane (j = lane & 31, h = lane >> 5) holds the corners of point j of the tile (both halves compute them, half 0 writes).
//   key  [32 * 8] int    out: per (point, corner) the box-local key, -1 where the corner is skipped (SKIP_ZERO_WEIGHT and w == 0,
//                             or an invalid lane); when !boxed: the voxel index itself
//   sw   [32 * 8] float  out: the corner weights (0 for invalid lanes)
//   slot [256] int       out: key -> compact slot number (reducible tiles)
//   svox [PB_SLOTS] int  out: slot -> voxel index
template <bool SKIP_ZERO_WEIGHT>
__device__ __forceinline__ PatchBox patch_box_build(const Corner8& cr, const int* lo, const int* hi, bool valid, int V, int lane, int* key,
                                                    float* sw, int* slot, int* svox) {
    const int j = lane & 31, h = lane >> 5;
    PatchBox pb;
    int x0 = valid ? lo[0] : V, y0 = valid ? lo[1] : V, z0 = valid ? lo[2] : V;
    int x1 = valid ? lo[0] : 0, y1 = valid ? lo[1] : 0, z1 = valid ? lo[2] : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        x0 = min(x0, __shfl_xor(x0, d, WAVE));
        y0 = min(y0, __shfl_xor(y0, d, WAVE));
        z0 = min(z0, __shfl_xor(z0, d, WAVE));
        x1 = max(x1, __shfl_xor(x1, d, WAVE));
        y1 = max(y1, __shfl_xor(y1, d, WAVE));
        z1 = max(z1, __shfl_xor(z1, d, WAVE));
    }
    pb.x0 = x0;
    pb.y0 = y0;
    pb.z0 = z0;
    // the far corners sit at most one voxel beyond the largest floor corner
    pb.boxed = x1 - x0 + 2 <= PB_BOX && y1 - y0 + 2 <= PB_BOX && z1 - z0 + 2 <= PB_BOX;          // wave-uniform
    pb.U = 0;
    pb.reducible = false;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<u4*>(slot + 4 * lane) = u4{0u, 0u, 0u, 0u};
    pb_wave_sync();
    if (h == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int vx = (k & 1) ? hi[0] : lo[0], vy = (k & 2) ? hi[1] : lo[1], vz = (k & 4) ? hi[2] : lo[2];
            const float w = valid ? cr.w[k] : 0.0f;
            const int kk = ((vz - z0) * PB_BOX + (vy - y0)) * PB_BOX + (vx - x0);
            const bool on = valid && !(SKIP_ZERO_WEIGHT && w == 0.0f);
            key[j * 8 + k] = pb.boxed ? (on ? kk : -1) : (on ? cr.base[k] : -1);
            sw[j * 8 + k] = w;
            if (pb.boxed && on) slot[kk] = 1;
        }
    }
    pb_wave_sync();
    if (pb.boxed) {         // compact slot numbers: exclusive prefix sum of the marks, 4 keys per lane
        const u4 mk = *reinterpret_cast<const u4*>(slot + 4 * lane);
        const int cnt = (int)(mk[0] + mk[1] + mk[2] + mk[3]);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const int up = __shfl_up(incl, d, WAVE);
            if (lane >= d) incl += up;
        }
        pb.U = __shfl(incl, WAVE - 1, WAVE);
        pb.reducible = pb.U <= PB_SLOTS;
        int sl = incl - cnt;
        u4 out;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            out[q] = mk[q] ? (uint32_t)sl : 0xffffffffu;
            if (mk[q] && sl < PB_SLOTS) svox[sl] = pb_voxel_of_key(pb, 4 * lane + q, V);
            sl += mk[q] ? 1 : 0;
        }
        *reinterpret_cast<u4*>(slot + 4 * lane) = out;
    }
    pb_wave_sync();
    return pb;
}

// tile `ti` of an image, row p (0..31) of the tile -> point index inside the image.  patch: the image's points are R x R rays x S
// depths ray-major with R % 4 == 0 and S % 2 == 0, tiles are (patch row, patch column, depth slot) with the depth slot
// innermost, rows are (dx, dy, dd) with dx fastest; else 32 consecutive points.
__device__ __forceinline__ long long patch_point(bool patch, long long ti, int p, int R, int S) {
    if (!patch) return ti * 32 + p;
    const int n_ds = S / 2, n_pc = R / 4;
    const int ds = (int)(ti % n_ds), pc = (int)((ti / n_ds) % n_pc), prow = (int)(ti / ((long long)n_ds * n_pc));
    return (long long)((prow * 4 + ((p >> 2) & 3)) * R + pc * 4 + (p & 3)) * S + ds * 2 + (p >> 4);
}

}  // namespace cnerf
