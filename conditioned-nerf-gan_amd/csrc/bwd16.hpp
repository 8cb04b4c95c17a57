// Shared definitions of the half-precision backward path (bwd16.hip, field_h3.hip STORE16).
//
// "TB16" = tile-blocked fp16 matrix layout of the backward's activation / gradient buffers.  A matrix with one row per sample
// point and CT * 32 channels is stored per 32-point tile and per 32-channel tile as a dense 32 x 32 block of fp16 (2 KiB):
//     element (tile T, channel tile t, point j, channel c)  ->  fp16 index ((T * CT + t) * 32 + j) * 32 + c
// Tiles are the field kernels' own work units (image b, tile k of tiles_per_image = ceil(n_per_image / 32): T = b *
// tiles_per_image + k), so rows past the end of an image exist as rows of their own (G rows there are zero).  Why this
// shape: the producers hold, per lane (point j, half h), 4 consecutive channels of a tile at a time -> one 8-byte store into
// a 2-KiB block that four such instructions complete; the weight-gradient kernel copies a whole tile (all channel tiles of
// 32 points, contiguous) into LDS with linear LDS-DMA and reads MFMA operands out of the 64-byte rows with
// ds_read_b64_tr_b16, conflict-free (bank = 16 j + c / 2 over the 4 rows x 32 channels a half-wave reads).
//
// "COS16": cos(arg) of every slab travels from the storing forward to the chain kernel only, lane for lane (lane (j, h) of the
// wave that owns tile T holds channels 32 t + 8 g + 4 h + e of point j), so it is stored fragment-major -- same size, but every
// wave instruction moves 512 contiguous bytes instead of 16 bytes in each of 32 rows:
//     element (slab m, tile T, channel tile t, quad g, lane, e) -> fp16 index ((((m * tiles + T) * NT + t) * 4 + g) * 64 + lane) * 4 + e
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cnerf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));

__device__ __forceinline__ size_t tb16_index(long long tile, int ct, int t, int j, int c) {
    return ((((size_t)tile * ct + t) * 32 + j) * 32) + c;
}

// two fp32 -> packed fp16 pair, round to nearest even (v_cvt_pk_f16_f32 ... the compiler's cast), no saturation needed:
// the callers scale into range
__device__ __forceinline__ uint32_t pk_f16(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}

}  // namespace cnerf
