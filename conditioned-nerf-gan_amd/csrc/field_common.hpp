// Device helpers shared by the point-pass kernels (field_kernel.hip: fp32 MFMA, field_h3.hip: fp16x3 split).
#pragma once
#include "cnerf_dev.hpp"
#include "cnerf_kernels.hpp"

namespace cnerf {

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
// Layer-0 input tiles.  The input of layer 0 is a runtime list of 32-wide tiles: 32 channels of one pyramid level's
// feature volume (siren.py:555-571, 1444-1473) or the world position padded to 32 (feature || xyz, siren.py:1158).
// Lane (j, h) holds channels 8g + 4h + e of its point in register 4g + e, like every activation tile.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 input_tile(const FieldArgs& a, int b, int tk, float px, float py, float pz, int h) {
    f32x16 feat;
#pragma unroll
    for (int r = 0; r < 16; ++r) feat[r] = 0.0f;
    const int lvl = a.in_level[tk];
    if (lvl < 0) {                       // xyz tile: channels 0,1,2 = x,y,z live in half 0, registers 0..2
        if (h == 0) {
            feat[0] = px;
            feat[1] = py;
            feat[2] = pz;
        }
        return feat;
    }
    const int V = a.lvl_V[lvl], C = a.lvl_C[lvl];
    Corner8 cr;
    trilinear_corners(px, py, pz, a.half_voxel, V, cr);
    const float* vol = a.lvl_vol[lvl] + (size_t)b * V * V * V * C + a.in_chan[tk] + 4 * h;
    f32x4 q[8][4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float* cp = vol + (size_t)cr.base[k] * C;
#pragma unroll
        for (int g = 0; g < 4; ++g) q[k][g] = *reinterpret_cast<const f32x4*>(cp + 8 * g);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)     // ATen order: corners sequentially, product and sum rounded separately
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) feat[4 * g + e] = feat[4 * g + e] + q[k][g][e] * cr.w[k];
    return feat;
}


// The same in two steps, so that a kernel can put a barrier between the issue of the 32 loads and their first use.
struct InputTile {
    f32x4 q[8][4];
    float w[8];
    bool is_volume;
};
// tk must be a volume tile (in_level[tk] >= 0): every field of `it` is overwritten
__device__ __forceinline__ void input_tile_issue_volume(const FieldArgs& a, int b, int tk, float px, float py, float pz, int h, InputTile& it) {
    const int lvl = a.in_level[tk];
    it.is_volume = true;
    const int V = a.lvl_V[lvl], C = a.lvl_C[lvl];
    Corner8 cr;
    trilinear_corners(px, py, pz, a.half_voxel, V, cr);
    const float* vol = a.lvl_vol[lvl] + (size_t)b * V * V * V * C + a.in_chan[tk] + 4 * h;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        it.w[k] = cr.w[k];
        const float* cp = vol + (size_t)cr.base[k] * C;
#pragma unroll
        for (int g = 0; g < 4; ++g) it.q[k][g] = *reinterpret_cast<const f32x4*>(cp + 8 * g);
    }
}
__device__ __forceinline__ void input_tile_issue(const FieldArgs& a, int b, int tk, float px, float py, float pz, int h, InputTile& it) {
    if (a.in_level[tk] >= 0) input_tile_issue_volume(a, b, tk, px, py, pz, h, it);
    else it.is_volume = false;
}
__device__ __forceinline__ f32x16 input_tile_reduce(const InputTile& it, float px, float py, float pz, int h) {
    f32x16 feat;
#pragma unroll
    for (int r = 0; r < 16; ++r) feat[r] = 0.0f;
    if (!it.is_volume) {
        if (h == 0) {
            feat[0] = px;
            feat[1] = py;
            feat[2] = pz;
        }
        return feat;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) feat[4 * g + e] = feat[4 * g + e] + it.q[k][g][e] * it.w[k];
    return feat;
}

// LDS-DMA form of the lookups of a volume tile (in_level[tk] >= 0): 32 wave instructions park the 8 corner lines of every
// point of the tile in the wave's own 32 KiB of LDS -- piece (k, g) at lds_wave + (4k + g) * 64 float4, lane-linear -- with
// no VGPRs and no wait, so they can be issued a whole tile ahead.  input_tile_from_lds() finishes the job: it waits for
// the wave's outstanding loads, recomputes the corner weights from the position and accumulates in ATen's order.
__device__ __forceinline__ void input_tile_dma(const FieldArgs& a, int b, int tk, float px, float py, float pz, int h, f32x4* lds_wave) {
    const int lvl = a.in_level[tk];
    const int V = a.lvl_V[lvl], C = a.lvl_C[lvl];
    Corner8 cr;
    trilinear_corners(px, py, pz, a.half_voxel, V, cr);
    const float* vol = a.lvl_vol[lvl] + (size_t)b * V * V * V * C + a.in_chan[tk] + 4 * h;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float* cp = vol + (size_t)cr.base[k] * C;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cp + 8 * g),
                                             (__attribute__((address_space(3))) void*)(lds_wave + (4 * k + g) * 64), 16, 0, 0);
    }
}
__device__ __forceinline__ f32x16 input_tile_from_lds(const FieldArgs& a, int tk, float px, float py, float pz, int h, const f32x4* lds_wave,
                                                      int lane) {
    const int lvl = a.in_level[tk];
    Corner8 cr;
    trilinear_corners(px, py, pz, a.half_voxel, a.lvl_V[lvl], cr);
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): the DMA of this tile has landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x16 feat;
#pragma unroll
    for (int r = 0; r < 16; ++r) feat[r] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k)     // ATen order: corners sequentially, product and sum rounded separately
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 q = lds_wave[(4 * k + g) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) feat[4 * g + e] = feat[4 * g + e] + q[e] * cr.w[k];
        }
    return feat;
}

// sample position of point nn of image b, in two steps so that a kernel can issue the one dependent load (the jitter
// draw, the resampled depth or the explicit point) long before it needs the position
struct TileRaw {
    float v[3];     // POINTS: x, y, z;  COARSE: v[0] = u;  FINE: v[0] = t
};
__device__ __forceinline__ TileRaw tile_point_fetch(const FieldArgs& a, int b, long long nn) {
    TileRaw r;
    r.v[0] = r.v[1] = r.v[2] = 0.0f;
    const size_t gp = (size_t)b * a.n_per_image + nn;
    if (a.mode == FIELD_MODE_POINTS) {
        const float* p = a.points + gp * 3;
        r.v[0] = p[0];
        r.v[1] = p[1];
        r.v[2] = p[2];
    } else if (a.mode == FIELD_MODE_COARSE) {
        // the jitter draw: injected tensor, or Philox on the element's index inside the whole call, or none (0.5 = no jitter)
        r.v[0] = a.u_strat ? a.u_strat[gp] : (a.philox.on ? philox_uniform(a.philox, PHILOX_U_STRAT, (unsigned long long)(b + a.image0) * a.n_per_image + nn) : 0.5f);
    } else {
        r.v[0] = a.fine_z[gp];
    }
    return r;
}
__device__ __forceinline__ void tile_point_finish(const FieldArgs& a, int b, long long nn, const TileRaw& raw, bool valid, int h, bool write,
                                                  float& px, float& py, float& pz) {
    if (a.mode == FIELD_MODE_POINTS) {
        px = raw.v[0];
        py = raw.v[1];
        pz = raw.v[2];
    } else {
        const int S = a.geom.S, R = a.geom.R;
        const int ray = (int)(nn / S), s = (int)(nn - (long long)ray * S);
        const int row = ray / R, col = ray - row * R;
        float dx, dy, dz;
        camera_dir(a.geom, row, col, dx, dy, dz);
        const float* m = a.cam2world + (size_t)b * 16;
        if (a.mode == FIELD_MODE_COARSE) {
            float zj;
            coarse_sample(a.geom, m, dx, dy, dz, s, raw.v[0], zj, px, py, pz);
            if (write && a.z_out && valid && h == 0) a.z_out[(size_t)b * a.n_per_image + nn] = zj;
        } else {
            fine_sample(m, dx, dy, dz, raw.v[0], px, py, pz);
        }
    }
    if (write && a.points_out && valid && h == 0) {
        float* po = a.points_out + ((size_t)b * a.n_per_image + nn) * 3;
        po[0] = px;
        po[1] = py;
        po[2] = pz;
    }
}
__device__ __forceinline__ void tile_point(const FieldArgs& a, int b, long long nn, bool valid, int h, bool write,
                                           float& px, float& py, float& pz) {
    const TileRaw raw = tile_point_fetch(a, b, nn);
    tile_point_finish(a, b, nn, raw, valid, h, write, px, py, pz);
}

// ---------------------------------------------------------------------------------------------------------------
// Dropout (nn.Dropout behind the sine, siren.py:158-159,175-176,197-198; training mode only): the factors 0 or 1/(1-p) of
// the 4 consecutive channels c0..c0+3 of dropout layer d at point `gp` (index in the whole call).  Injected keep bytes, or
// one Philox block per 4 channels: counter index ((gp * n_drop + d) * H + c0) / 4, word e decides channel c0 + e.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 drop_factors(const FieldArgs& a, unsigned long long gp, int d, int H, int c0) {
    f32x4 f;
    if (a.drop_mask) {
        const uint32_t m = *reinterpret_cast<const uint32_t*>(a.drop_mask + ((size_t)d * a.drop_points + gp) * H + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = ((m >> (8 * e)) & 0xffu) ? a.drop_scale : 0.0f;
    } else {
        const unsigned long long idx = ((gp * (unsigned long long)a.n_drop + d) * H + c0) >> 2;
        // the ten rounds as a rolled loop: this sits inside fully unrolled matrix loops, 64 times per layer
        uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = a.drop_stream, c3 = a.philox.offset;
        uint32_t k0 = a.philox.seed_lo, k1 = a.philox.seed_hi;
#pragma nounroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
            c0 = n0;
            c1 = lo1;
            c2 = n2;
            c3 = lo0;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        f[0] = c0 >= a.drop_thresh ? a.drop_scale : 0.0f;
        f[1] = c1 >= a.drop_thresh ? a.drop_scale : 0.0f;
        f[2] = c2 >= a.drop_thresh ? a.drop_scale : 0.0f;
        f[3] = c3 >= a.drop_thresh ? a.drop_scale : 0.0f;
    }
    return f;
}
// ... applied to one activation tile (and, in the storing forward, to the rows of derivatives that travel with it)
__device__ __forceinline__ void drop_tile(const FieldArgs& a, unsigned long long gp, int d, int H, int t, int h, f32x16& v) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 f = drop_factors(a, gp, d, H, 32 * t + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * g + e] *= f[e];
    }
}
__device__ __forceinline__ void drop_tile2(const FieldArgs& a, unsigned long long gp, int d, int H, int t, int h, f32x16& v, f32x16& w) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 f = drop_factors(a, gp, d, H, 32 * t + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[4 * g + e] *= f[e];
            w[4 * g + e] *= f[e];
        }
    }
}

// XCD-aware tile ownership: blocks b and b+8 share an XCD (round-robin dispatch), so give each of the 8 block classes
// one contiguous eighth of the tiles (a band of neighbouring rays -> a compact slab of the feature grid in that XCD's
// L2).  Placement only changes speed, never results.
struct TileRange {
    long long begin, end, stride;
};
__device__ __forceinline__ TileRange tile_range(long long total_tiles) {
    const int nblk = gridDim.x;
    const int cls = blockIdx.x & 7, idx_in_cls = blockIdx.x >> 3;
    const int blk_per_cls = (nblk + 7 - cls) / 8;   // blocks whose id % 8 == cls
    TileRange r;
    r.begin = total_tiles * cls / 8 + idx_in_cls * 4 + (threadIdx.x >> 6);
    r.end = total_tiles * (cls + 1) / 8;
    r.stride = (long long)blk_per_cls * 4;
    return r;
}

// rgb_sigma pre-activations of the tile: returns, in lanes 0..31, the 4 head outputs of the lane's point.
template <int NT>
__device__ __forceinline__ f32x4 head_forward(const f32x4* __restrict__ wp, const float* __restrict__ bias, const f32x16* x,
                                              int lane) {
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NT * 4; ++q) {                 // 4 activation registers per 16-byte weight load
        const f32x4 aw = wp[q * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int sreg = 4 * q + e;
            acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(aw[e], x[sreg >> 4][sreg & 15], acc[e], 0, 0, 0);   // 4 independent chains
        }
    }
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
        o[i] = v + __shfl_xor(v, 32, WAVE) + bias[i];  // the two lane halves hold the two halves of the channels
    }
    return o;
}


}  // namespace cnerf
