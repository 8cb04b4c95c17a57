// Device-side helpers shared by the gfx950 kernels of libcnerf_hip.so.
//
// Arithmetic contract (see DESIGN.md "Numerics"): every translation unit is compiled with -ffp-contract=off, so
// `a * b + c` is two roundings exactly as the ATen CPU elementwise ops the reference runs, and a fused multiply-add
// happens only where fmaf() is written -- which is where the reference's CPU path itself fuses (torch.linspace, the
// 4x4 bmm, the 3-vector norm; verified bit-for-bit against the golden vectors).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cnerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------------------------------------------
// sin(x) for |x| up to a few 1e4: two-term Cody-Waite reduction by pi (fused), odd degree-9 polynomial on
// [-pi/2, pi/2], sign from the parity of the quotient.  Max error 1.2e-7 abs / 1.9 ulp on [-300, 300] (the
// FiLM arguments are |freq*x+phase| < ~200).  No transcendental unit, no branches.
// The quotient n = rint(x/pi) comes from the magic-number trick: t = fma(x, 1/pi, 1.5*2^23) has n in its low mantissa
// bits (so parity = bit 0 of t) and n = t - 1.5*2^23 exactly; 12 VALU ops scalar, 17 for a PAIR of arguments on the
// packed fp32 instructions of gfx950 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth per issue slot).
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr float SIN_MAGIC = 12582912.0f;              // 1.5 * 2^23
constexpr float SIN_INV_PI = 0.31830987334251404f;
constexpr float SIN_PI_HI = 3.1415927410125732f;
constexpr float SIN_PI_LO = -8.742277657347586e-08f;
constexpr float SIN_C9 = 2.629978780532838e-06f, SIN_C7 = -0.00019821235036943108f, SIN_C5 = 0.008333230391144753f,
                SIN_C3 = -0.1666666567325592f;

__device__ __forceinline__ float sin_pi_reduced(float x) {
    const float t = __builtin_fmaf(x, SIN_INV_PI, SIN_MAGIC);
    const float n = t - SIN_MAGIC;
    float r = __builtin_fmaf(-n, SIN_PI_HI, x);
    r = __builtin_fmaf(-n, SIN_PI_LO, r);
    const float s = r * r;
    float p = SIN_C9;
    p = __builtin_fmaf(p, s, SIN_C7);
    p = __builtin_fmaf(p, s, SIN_C5);
    p = __builtin_fmaf(p, s, SIN_C3);
    const float y = __builtin_fmaf(r * s, p, r);
    return __uint_as_float(__float_as_uint(y) ^ (__float_as_uint(t) << 31));
}

__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

// the same arithmetic on two arguments at once (bit-identical per element to sin_pi_reduced)
__device__ __forceinline__ f32x2 sin_pi_reduced2(f32x2 x) {
    const f32x2 t = __builtin_elementwise_fma(x, splat2(SIN_INV_PI), splat2(SIN_MAGIC));
    const f32x2 n = t - splat2(SIN_MAGIC);
    f32x2 r = __builtin_elementwise_fma(-n, splat2(SIN_PI_HI), x);
    r = __builtin_elementwise_fma(-n, splat2(SIN_PI_LO), r);
    const f32x2 s = r * r;
    f32x2 p = splat2(SIN_C9);
    p = __builtin_elementwise_fma(p, s, splat2(SIN_C7));
    p = __builtin_elementwise_fma(p, s, splat2(SIN_C5));
    p = __builtin_elementwise_fma(p, s, splat2(SIN_C3));
    const f32x2 y = __builtin_elementwise_fma(r * s, p, r);
    const u32x2 yb = __builtin_bit_cast(u32x2, y) ^ (__builtin_bit_cast(u32x2, t) << 31);
    return __builtin_bit_cast(f32x2, yb);
}

// sin(x) on the transcendental unit: exact two-term Cody-Waite reduction by 2 pi (so the argument of v_sin_f32, which
// takes revolutions, is in [-0.5, 0.5] with full relative accuracy), then v_sin_f32.  Measured on gfx950 over [-300, 300]
// (scripts/ubench/vsin_accuracy.hip): max abs error 3.8e-7, rms 7.5e-8 -- three times the polynomial's maximum, five VALU
// ops and one transcendental instead of twelve VALU ops.  Used by the split-precision kernel.
__device__ __forceinline__ float sin_2pi_reduced_hw(float x) {
    const float t = __builtin_fmaf(x, 0.15915494309189535f, SIN_MAGIC);
    const float n = t - SIN_MAGIC;
    float r = __builtin_fmaf(-n, 6.2831854820251465f, x);
    r = __builtin_fmaf(-n, -1.7484555314695172e-07f, r);
    return __builtin_amdgcn_sinf(r * 0.15915494309189535f);
}

// sine and cosine of the same argument on the transcendental unit (shared reduction): the activation-storing forward
__device__ __forceinline__ void sincos_2pi_reduced_hw(float x, float& sn, float& cs) {
    const float t = __builtin_fmaf(x, 0.15915494309189535f, SIN_MAGIC);
    const float n = t - SIN_MAGIC;
    float r = __builtin_fmaf(-n, 6.2831854820251465f, x);
    r = __builtin_fmaf(-n, -1.7484555314695172e-07f, r);
    const float u = r * 0.15915494309189535f;
    sn = __builtin_amdgcn_sinf(u);
    cs = __builtin_amdgcn_cosf(u);
}

__device__ __forceinline__ f32x2 sin_2pi_reduced_hw2(f32x2 x) {     // two arguments, reduction on the packed fp32 ops
    const f32x2 t = __builtin_elementwise_fma(x, splat2(0.15915494309189535f), splat2(SIN_MAGIC));
    const f32x2 n = t - splat2(SIN_MAGIC);
    f32x2 r = __builtin_elementwise_fma(-n, splat2(6.2831854820251465f), x);
    r = __builtin_elementwise_fma(-n, splat2(-1.7484555314695172e-07f), r);
    r = r * splat2(0.15915494309189535f);
    return f32x2{__builtin_amdgcn_sinf(r[0]), __builtin_amdgcn_sinf(r[1])};
}

// sin and cos of the same argument (shared reduction); cos: even degree-10 polynomial, max abs error 1.5e-7 on [-300, 300].
// Used by the activation-storing forward of the backward pass.
__device__ __forceinline__ void sincos_pi_reduced(float x, float& sn, float& cs) {
    const float t = __builtin_fmaf(x, SIN_INV_PI, SIN_MAGIC);
    const float n = t - SIN_MAGIC;
    float r = __builtin_fmaf(-n, SIN_PI_HI, x);
    r = __builtin_fmaf(-n, SIN_PI_LO, r);
    const float s = r * r;
    float p = SIN_C9;
    p = __builtin_fmaf(p, s, SIN_C7);
    p = __builtin_fmaf(p, s, SIN_C5);
    p = __builtin_fmaf(p, s, SIN_C3);
    const float y = __builtin_fmaf(r * s, p, r);
    float q = -2.6247781192978437e-07f;
    q = __builtin_fmaf(q, s, 2.4772387405391783e-05f);
    q = __builtin_fmaf(q, s, -0.0013888622634112835f);
    q = __builtin_fmaf(q, s, 0.041666656732559204f);
    q = __builtin_fmaf(q, s, -0.5f);
    const float c = __builtin_fmaf(s, q, 1.0f);
    const uint32_t flip = __float_as_uint(t) << 31;
    sn = __uint_as_float(__float_as_uint(y) ^ flip);
    cs = __uint_as_float(__float_as_uint(c) ^ flip);
}

// torch.linspace(start, end, steps)[i] in float32, as ATen computes it (symmetric halves, fused).
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? __builtin_fmaf(step, (float)i, start) : __builtin_fmaf(-step, (float)(steps - 1 - i), end);
}

// F.softplus(x) with beta=1, threshold=20
__device__ __forceinline__ float softplus20(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------------------------
// wave-wide scans in double (ATen's CPU cumsum/cumprod accumulate float32 inputs in double and round each prefix
// to float32; a double tree scan reproduces that rounding regardless of association up to ~1e-16)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_incl_prod(double v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const double o = __shfl_up(v, d, WAVE);
        if (lane >= d) v *= o;
    }
    return v;
}
__device__ __forceinline__ double wave_incl_sum(double v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const double o = __shfl_up(v, d, WAVE);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// In-kernel random draws: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), counter-based:
// the draw of element `idx` of stream `stream` (0 u_strat, 1 eps_coarse, 2 u_fine, 3 eps_final -- the reference's four draws,
// volumetric_rendering.py:39,106,319) of call `offset` under `seed` is a pure function of those four numbers, so forward,
// re-computed forward and backward see the same values without a tensor in between.  counter = (idx lo, idx hi, stream,
// offset), key = (seed lo, seed hi).  uniform: top 24 bits of word 0 -> [0, 1) in steps of 2^-24 (torch.rand's grid);
// normal: Box-Muller on words 0, 1.  oracle/philox.py is the NumPy twin (known-answer tested).
// ---------------------------------------------------------------------------------------------------------------
struct PhiloxKey {
    uint32_t on, offset, seed_lo, seed_hi;
};
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
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
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}
__device__ __forceinline__ float philox_uniform(const PhiloxKey& k, uint32_t stream, unsigned long long idx) {
    uint32_t o[4];
    philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), stream, k.offset, k.seed_lo, k.seed_hi, o);
    return (float)(o[0] >> 8) * 5.9604644775390625e-08f;           // * 2^-24
}
__device__ __forceinline__ float philox_normal(const PhiloxKey& k, uint32_t stream, unsigned long long idx) {
    uint32_t o[4];
    philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), stream, k.offset, k.seed_lo, k.seed_hi, o);
    const float u1 = (float)((o[0] >> 8) + 1u) * 5.9604644775390625e-08f;     // (0, 1]
    const float u2 = (float)(o[1] >> 8) * 5.9604644775390625e-08f;            // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853071795864f * u2);
}
enum { PHILOX_U_STRAT = 0, PHILOX_EPS_COARSE = 1, PHILOX_U_FINE = 2, PHILOX_EPS_FINAL = 3, PHILOX_DROP_COARSE = 4, PHILOX_DROP_FINE = 5, PHILOX_DROP_POINTS = 6 };

// ---------------------------------------------------------------------------------------------------------------
// rays and sample points (volumetric_rendering.py:73-199, generators.py:138-142)
// ---------------------------------------------------------------------------------------------------------------
struct RayGeom {
    int R, S;
    float focal;  // 1 / tan(fov/2), computed on the host like the reference does
    float ray_start, ray_end;
};

// unit camera-space direction of pixel (row, col)
__device__ __forceinline__ void camera_dir(const RayGeom& g, int row, int col, float& dx, float& dy, float& dz) {
    const float x = linspace_at(-1.0f, 1.0f, g.R, col);
    const float y = linspace_at(-1.0f, 1.0f, g.R, row);
    const float z = g.focal;
    float n2 = x * x;
    n2 = __builtin_fmaf(y, y, n2);
    n2 = __builtin_fmaf(z, z, n2);
    const float n = __builtin_sqrtf(n2);
    dx = x / n;
    dy = y / n;
    dz = z / n;
}

// coarse sample s of a ray: jittered distance zj and world position (px,py,pz).  m = cam2world row-major (16 floats).
__device__ __forceinline__ void coarse_sample(const RayGeom& g, const float* __restrict__ m, float dx, float dy,
                                              float dz, int s, float u, float& zj, float& px, float& py, float& pz) {
    const float zl = linspace_at(g.ray_start, g.ray_end, g.S, s);
    const float dz01 = linspace_at(g.ray_start, g.ray_end, g.S, 1) - linspace_at(g.ray_start, g.ray_end, g.S, 0);
    const float off = (u - 0.5f) * dz01;
    zj = zl + off;
    const float cx = dx * zl + off * dx;
    const float cy = dy * zl + off * dy;
    const float cz = dz * zl + off * dz;
    px = __builtin_fmaf(m[2], cz, __builtin_fmaf(m[1], cy, m[0] * cx)) + m[3];
    py = __builtin_fmaf(m[6], cz, __builtin_fmaf(m[5], cy, m[4] * cx)) + m[7];
    pz = __builtin_fmaf(m[10], cz, __builtin_fmaf(m[9], cy, m[8] * cx)) + m[11];
}

// fine sample at distance t along the world-space ray: origin + rot(dir) * t
__device__ __forceinline__ void fine_sample(const float* __restrict__ m, float dx, float dy, float dz, float t,
                                            float& px, float& py, float& pz) {
    const float wx = __builtin_fmaf(m[2], dz, __builtin_fmaf(m[1], dy, m[0] * dx));
    const float wy = __builtin_fmaf(m[6], dz, __builtin_fmaf(m[5], dy, m[4] * dx));
    const float wz = __builtin_fmaf(m[10], dz, __builtin_fmaf(m[9], dy, m[8] * dx));
    px = m[3] + wx * t;
    py = m[7] + wy * t;
    pz = m[11] + wz * t;
}

// ---------------------------------------------------------------------------------------------------------------
// trilinear lookup geometry (siren.py:555-567 -> ATen grid_sampler_3d, bilinear / border / align_corners=False)
// ---------------------------------------------------------------------------------------------------------------
struct Corner8 {
    int base[8];   // voxel index (iz*V + iy)*V + ix of each corner, clamped in range
    float w[8];    // its weight; corners in ATen accumulation order: x fastest, then y, then z
};

__device__ __forceinline__ void unnormalize(float p, float half_voxel, int V, int& i0, float& lo, float& hi) {
    const float gco = p / half_voxel;
    float ic = ((gco + 1.0f) * (float)V - 1.0f) / 2.0f;
    ic = fminf(fmaxf(ic, 0.0f), (float)(V - 1));
    const float fl = floorf(ic);
    i0 = (int)fl;
    lo = ic - fl;
    hi = (fl + 1.0f) - ic;
}

__device__ __forceinline__ void trilinear_corners(float px, float py, float pz, float half_voxel, int V, Corner8& c) {
    int ix, iy, iz;
    float lx, hx, ly, hy, lz, hz;
    unnormalize(px, half_voxel, V, ix, lx, hx);
    unnormalize(py, half_voxel, V, iy, ly, hy);
    unnormalize(pz, half_voxel, V, iz, lz, hz);
    const int ix1 = min(ix + 1, V - 1), iy1 = min(iy + 1, V - 1), iz1 = min(iz + 1, V - 1);
    // a +1 corner that would fall at index V only occurs with weight exactly 0 (coordinate clamped to V-1), so
    // re-reading the clamped voxel adds +0 where ATen skips the term.
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int xx = (k & 1) ? ix1 : ix, yy = (k & 2) ? iy1 : iy, zz = (k & 4) ? iz1 : iz;
        const float wx = (k & 1) ? lx : hx, wy = (k & 2) ? ly : hy, wz = (k & 4) ? lz : hz;
        c.base[k] = (zz * V + yy) * V + xx;
        c.w[k] = wx * wy * wz;
    }
}

// Workgroup barrier that drains nothing.  __syncthreads() carries a release fence, which the compiler implements as
// s_waitcnt vmcnt(0): every global store, float atomic and load the wave has in flight is drained at the barrier (an LDS-only
// fence -- __builtin_amdgcn_fence(..., "workgroup", "local") -- does the same as soon as an LDS-DMA copy is pending, because those
// are counted by vmcnt too; measured on the ISA).  In kernels that stream weight units through LDS behind one barrier per unit
// AND write activations or gradients to HBM (the activation-storing forward, the gradient chain) that exposed an HBM round trip
// at every unit.  Here: the bare s_barrier between two compiler-level memory barriers (no memory access is moved across it), and
// the caller states what it needs to have completed: wait_vmcnt<N>() with N = the number of vector-memory operations it has issued
// SINCE the LDS-DMA copy it is about to read (vmcnt retires in issue order; those N may stay in flight), and nothing for its LDS
// reads of the slot that is refilled next -- their data has already been consumed by issued MFMAs.
__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {      // at most N vector-memory operations outstanding (gfx9 encoding: 6 bits, split)
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

}  // namespace cnerf
