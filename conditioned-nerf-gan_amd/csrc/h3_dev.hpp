// Device helpers shared by the fp16 point-pass kernels (field_h3.hip: FiLM / sine / residual networks; field_pw16.hip: the per-point
// FiLM family): two-part fp16 operands, LDS-DMA pieces, the three-product MFMA group.  Like those translation units this header is
// compiled once per value of CNERF_H3_PARTS, each into its own inner namespace.
#pragma once
#include "cnerf_dev.hpp"
#include "bwd16.hpp"

// CNERF_H3_PARTS = 2 (default): every fp32 operand as two fp16 parts, three MFMAs per 16 k-values (precision "fp16x3");
// CNERF_H3_PARTS = 1 (precision "fp16"): only the leading part, rounded to nearest -- one MFMA per 16 k-values.
#ifndef CNERF_H3_PARTS
#define CNERF_H3_PARTS 2
#endif
#if CNERF_H3_PARTS == 1
#define H3_NS h1
#define H3_LAUNCH_FIELD launch_field_h1
#define H3_LAUNCH_PACK launch_pack_h1
#define H3_LAUNCH_FOLD launch_fold_h1
#else
#define H3_NS h3
#define H3_LAUNCH_FIELD launch_field_h3
#define H3_LAUNCH_PACK launch_pack_h3
#define H3_LAUNCH_FOLD launch_fold_h3
#endif

namespace cnerf {
namespace H3_NS {

constexpr int PARTS = CNERF_H3_PARTS;

struct Split2 {          // eight fp32 values as PARTS fp16 fragments; dword d of a fragment = elements 2d (low half), 2d+1
    u32x4 p[PARTS];
    __device__ __forceinline__ f16x8 frag(int k) const { return __builtin_bit_cast(f16x8, p[k]); }
};

// two fp32 -> packed fp16 pair, rounded toward zero (one v_cvt_pkrtz_f16_f32; saturates instead of overflowing)
__device__ __forceinline__ uint32_t pk_rtz(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b)); }

// Two parts of two consecutive values into dword D of the two fragments.  hi = value truncated to 11 significant bits
// (the fp32 with its 13 low mantissa bits cleared IS that fp16 value over fp16's normal range; below 2^-14 the two differ
// by < 6e-8 absolute, nothing at the scale of activations and scaled weights), lo = remainder, again truncated.
template <int D>
__device__ __forceinline__ void split_pair(float v0, float v1, Split2& s) {
    if constexpr (PARTS == 1) {
        s.p[0][D] = pk_f16(v0, v1);          // the only part: round to nearest (truncation would bias every product low)
    } else {
        s.p[0][D] = pk_rtz(v0, v1);
        // (a v_fma_mix_f32 against the packed half itself is one op instead of and + sub but measured no faster)
        const float r0 = v0 - __uint_as_float(__float_as_uint(v0) & 0xffffe000u);
        const float r1 = v1 - __uint_as_float(__float_as_uint(v1) & 0xffffe000u);
        s.p[PARTS - 1][D] = pk_rtz(r0, r1);
    }
}

// layer-0 inputs (looked-up features, positions) are not bounded like sine outputs: clamp to fp16's range first
__device__ __forceinline__ Split2 split8_clamped(const float* v) {
    float c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_fmed3f(v[i], -65504.0f, 65504.0f);
    Split2 s;
    split_pair<0>(c[0], c[1], s);
    split_pair<1>(c[2], c[3], s);
    split_pair<2>(c[4], c[5], s);
    split_pair<3>(c[6], c[7], s);
    return s;
}

// the two fp16 halves of a packed pair, widened (exact)
__device__ __forceinline__ float half_lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float half_hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }

// One wave instruction moves 1 KiB: lane i's 16 bytes from src_lane land at lds_dst + OFF + 16 i (the instruction offset
// applies to the global and to the LDS address alike).
template <int OFF>
__device__ __forceinline__ void dma_piece(const f16x8* src_lane, f16x8* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_lane,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, OFF, 0);
}

#if CNERF_H3_PARTS == 1
#define H3_MFMA3(acc, a, xs) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], (xs).frag(0), acc, 0, 0, 0)
#else
#define H3_MFMA3(acc, a, xs)                                                                  \
    do {                                                                                       \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], (xs).frag(0), acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], (xs).frag(1), acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], (xs).frag(0), acc, 0, 0, 0);        \
    } while (0)
#endif

}  // namespace H3_NS
}  // namespace cnerf
