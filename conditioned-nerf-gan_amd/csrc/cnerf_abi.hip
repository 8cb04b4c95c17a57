// C-ABI entry points of libcnerf_hip.so (declared in include/cnerf.h).  Host-side only: argument validation, buffer
// carving, launch sequencing on the caller's stream.  No allocation, no synchronisation, no global state.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cnerf_kernels.hpp"

using namespace cnerf;

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(CNERF_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int check_cfg(const cnerf_cfg* c, bool need_render) {
    if (!c) return fail(CNERF_EINVAL, "cfg is NULL");
    if (c->B < 1) return fail(CNERF_EINVAL, "B=%d must be >= 1", c->B);
    if (c->C < 32 || c->C % 32 != 0 || c->C > 224) return fail(CNERF_EINVAL, "C=%d must be a multiple of 32 in [32,224]", c->C);
    if (c->n_levels < 0 || c->n_levels > CNERF_MAX_LEVELS) return fail(CNERF_EINVAL, "n_levels=%d out of range", c->n_levels);
    if (c->n_levels > 1 || (c->n_levels == 1 && (c->level_V[0] != c->V || c->level_C[0] != c->C))) {
        int sum = 0;
        for (int i = 0; i < c->n_levels; ++i) {
            if (c->level_V[i] < 2 || c->level_V[i] > 1024) return fail(CNERF_EINVAL, "level_V[%d]=%d out of range [2,1024]", i, c->level_V[i]);
            if (c->level_C[i] < 32 || c->level_C[i] % 32 != 0) return fail(CNERF_EINVAL, "level_C[%d]=%d must be a multiple of 32", i, c->level_C[i]);
            sum += c->level_C[i];
        }
        if (sum != c->C) return fail(CNERF_EINVAL, "sum of level_C (%d) != C (%d)", sum, c->C);
    }
    if (c->H != 64 && c->H != 128 && c->H != 256) return fail(CNERF_EINVAL, "H=%d must be 64, 128 or 256", c->H);
    if (c->V < 2 || c->V > 1024) return fail(CNERF_EINVAL, "V=%d out of range [2,1024]", c->V);
    if (c->L < 1 || c->L > CNERF_MAX_LAYERS) return fail(CNERF_EINVAL, "L=%d out of range [1,%d]", c->L, CNERF_MAX_LAYERS);
    for (int l = 0; l < c->L; ++l) {
        const int k = c->layer_kind[l];
        if (k != CNERF_LAYER_FILM && k != CNERF_LAYER_SINE && k != CNERF_LAYER_RES && k != CNERF_LAYER_PFILM)
            return fail(CNERF_EINVAL, "layer_kind[%d]=%d unknown", l, k);
        if ((k == CNERF_LAYER_PFILM) != (c->layer_kind[0] == CNERF_LAYER_PFILM))
            return fail(CNERF_EINVAL, "per-point FiLM layers cannot be mixed with other layer kinds");
        if (l == 0 && k == CNERF_LAYER_RES) return fail(CNERF_EINVAL, "layer 0 cannot be a residual block");
    }
    if (!(c->voxel_length > 0.f)) return fail(CNERF_EINVAL, "voxel_length must be > 0");
    if (c->precision != CNERF_PREC_FP32 && c->precision != CNERF_PREC_FP16X3 && c->precision != CNERF_PREC_FP16)
        return fail(CNERF_EINVAL, "precision=%d unknown", c->precision);
    if (!(c->drop_p >= 0.0f && c->drop_p < 1.0f)) return fail(CNERF_EINVAL, "drop_p=%g out of [0,1)", (double)c->drop_p);
    if (c->drop_p > 0.0f && c->precision != CNERF_PREC_FP32) return fail(CNERF_EINVAL, "dropout (drop_p > 0) is implemented for precision fp32 only");
    if (need_render) {
        if (c->R < 1 || c->R > 4096) return fail(CNERF_EINVAL, "R=%d out of range [1,4096]", c->R);
        if (c->S < 2 || c->S > 128) return fail(CNERF_EINVAL, "S=%d out of range [2,128]", c->S);
        if (!(c->fov_deg > 0.0 && c->fov_deg < 180.0)) return fail(CNERF_EINVAL, "fov_deg out of (0,180)");
    }
    return CNERF_OK;
}

// packed layout: [float4 weight stream][biases of every layer in order (RES: b1 then b2)][head bias (4)][ones H][zeros H]
// (ones/zeros: a plain sine layer runs as a FiLM layer with freq = 1, phase = 0)
struct PackedLayout {
    size_t weight_floats;
    size_t bias_floats;
    int n_film;
    int n_in;      // 32-wide input tiles of layer 0
    int k0;        // real input width of layer 0
};

int n_levels_of(const cnerf_cfg* c) { return c->n_levels > 0 ? c->n_levels : 1; }
int level_V_of(const cnerf_cfg* c, int i) { return c->n_levels > 0 ? c->level_V[i] : c->V; }
int level_C_of(const cnerf_cfg* c, int i) { return c->n_levels > 0 ? c->level_C[i] : c->C; }

PackedLayout packed_layout(const cnerf_cfg* c) {
    const size_t NT = c->H / 32;
    const size_t tile = 4 * 64 * 4;  // floats per (t, tk) pair
    PackedLayout p{0, 0, 0, 0, 0};
    if (c->layer_kind[0] == CNERF_LAYER_PFILM && c->precision != CNERF_PREC_FP32) {
        // field_pw16.hip: weight units in consumption order -- Wm1 | W_0 | layer 0: per tile (freq rows, phase rows) | layers >= 1: per
        // tile (freq rows, W_l, phase rows) | head; units multiplied by m have 16 k-chunks, the others 2 NT.  Behind them the
        // constants of pw16_consts_kernel, then the 1 / S and max|W| slots of the 3 L + 2 packed matrices
        const size_t parts = c->precision == CNERF_PREC_FP16 ? 1 : 2;
        const size_t frag = 64 * 8 / 2;                     // floats per (tile, k-chunk, part)
        const size_t big = 16 * parts * frag, small = 2 * NT * parts * frag;
        p.n_in = 1;
        p.k0 = 3;
        p.weight_floats = big + small + NT * 2 * big + (size_t)(c->L - 1) * NT * (2 * big + small) + small;
        p.bias_floats = 256 + 3 * (size_t)c->L * c->H + 4 + (2 + 4 * (size_t)c->L + 3) / 4 * 4 + (2 * (3 * (size_t)c->L + 2) + 3) / 4 * 4;
        return p;
    }
    if (c->layer_kind[0] == CNERF_LAYER_PFILM) {   // mapping hidden | per layer (main, freq rows, phase rows) | head
        p.n_in = 1;
        p.k0 = 3;
        p.weight_floats = 8 * tile;
        p.bias_floats = 256;
        for (int l = 0; l < c->L; ++l) {
            p.weight_floats += NT * (l == 0 ? 1 : NT) * tile + 2 * NT * 8 * tile;
            p.bias_floats += 3 * c->H;
        }
        p.weight_floats += NT * tile;
        p.bias_floats += 4;
        return p;
    }
    p.n_in = c->C / 32 + ((c->flags & CNERF_F_INPUT_XYZ) ? 1 : 0);
    p.k0 = c->C + ((c->flags & CNERF_F_INPUT_XYZ) ? 3 : 0);
    if (c->precision == CNERF_PREC_FP16X3 || c->precision == CNERF_PREC_FP16) {
        const size_t parts = c->precision == CNERF_PREC_FP16 ? 1 : 2;
        // fp16 fragments: (t, k-chunk of 16, part) x 64 lanes x 8 fp16 = 256 floats' worth of bytes per (t, c, part);
        // behind the biases: 1/S of every matrix (L + 1), then the max|W| scratch slots (L + 1), padded to 4 floats
        const size_t frag = 64 * 8 / 2;   // in floats
        size_t mats = 0;
        for (int l = 0; l < c->L; ++l) {
            const size_t kc = (l == 0) ? 2 * (size_t)p.n_in : 2 * NT;
            const size_t n = c->layer_kind[l] == CNERF_LAYER_RES ? 2 : 1;      // residual block: fc1 and fc2
            p.weight_floats += n * NT * kc * parts * frag;
            p.bias_floats += n * c->H;
            mats += n;
            if (c->layer_kind[l] == CNERF_LAYER_FILM) p.n_film++;
        }
        p.weight_floats += 1 * 2 * NT * parts * frag;   // head, one 32-row tile
        p.bias_floats += 4 + (2 * (mats + 1) + 3) / 4 * 4;
        return p;
    }
    for (int l = 0; l < c->L; ++l) {
        const size_t kt = (l == 0) ? (size_t)p.n_in : NT;
        if (c->layer_kind[l] == CNERF_LAYER_RES) {
            p.weight_floats += 2 * NT * NT * tile;
            p.bias_floats += 2 * c->H;
        } else {
            p.weight_floats += NT * kt * tile;
            p.bias_floats += c->H;
            if (c->layer_kind[l] == CNERF_LAYER_FILM) p.n_film++;
        }
    }
    p.weight_floats += 1 * NT * tile;  // head, one 32-row tile
    p.bias_floats += 4;
    return p;
}

PhiloxKey philox_of(const cnerf_cfg* c) {
    return PhiloxKey{c->philox ? 1u : 0u, c->philox_offset, (uint32_t)(c->philox_seed & 0xffffffffull), (uint32_t)(c->philox_seed >> 32)};
}

RayGeom make_geom(const cnerf_cfg* c) {
    RayGeom g;
    g.R = c->R;
    g.S = c->S;
    // z = ones(float32) / np.tan((2*pi*fov/360)/2): the double tangent is rounded to fp32, then 1/x in fp32
    g.focal = 1.0f / (float)tan((2.0 * M_PI * c->fov_deg / 360.0) / 2.0);
    g.ray_start = c->ray_start;
    g.ray_end = c->ray_end;
    return g;
}

// image0: first image of the sub-range this launch works on (per-image pointers are offset here)
int fill_field_args(FieldArgs& a, const cnerf_cfg* c, const cnerf_volumes* vols, const cnerf_grad_volumes* gvols,
                    const float* packed, const float* freq, const float* phase, int image0 = 0) {
    memset(&a, 0, sizeof(a));
    const PackedLayout pl = packed_layout(c);
    if (!vols) return fail(CNERF_EINVAL, "volumes are NULL");
    int tk = 0;
    for (int i = 0; i < n_levels_of(c); ++i) {
        const int V = level_V_of(c, i), C = level_C_of(c, i);
        if (!vols->level[i]) return fail(CNERF_EINVAL, "volume level %d is NULL", i);
        const size_t per_image = (size_t)V * V * V * C;
        a.lvl_vol[i] = vols->level[i] + (size_t)image0 * per_image;
        a.lvl_grad[i] = (gvols && gvols->level[i]) ? gvols->level[i] + (size_t)image0 * per_image : nullptr;
        a.lvl_V[i] = V;
        a.lvl_C[i] = C;
        for (int cc = 0; cc < C; cc += 32, ++tk) {
            if (tk >= 8) return fail(CNERF_EINVAL, "more than 8 input tiles");
            a.in_level[tk] = i;
            a.in_chan[tk] = cc;
        }
    }
    if (c->flags & CNERF_F_INPUT_XYZ) {
        if (tk >= 8) return fail(CNERF_EINVAL, "more than 8 input tiles");
        a.in_level[tk] = -1;
        a.in_chan[tk] = 0;
        ++tk;
    }
    a.n_in = tk;
    freq = (pl.n_film && freq) ? freq + (size_t)image0 * pl.n_film * c->H : freq;
    phase = (pl.n_film && phase) ? phase + (size_t)image0 * pl.n_film * c->H : phase;
    a.packed = packed;
    a.bias = packed + pl.weight_floats;
    a.freq = pl.n_film ? freq : nullptr;
    a.phase = pl.n_film ? phase : nullptr;
    a.film_stride = pl.n_film * c->H;
    a.bias_floats = (int)pl.bias_floats;
    a.geom = make_geom(c);
    a.half_voxel = c->voxel_length / 2.0f;
    a.L = c->L;
    a.n_mats = 0;
    for (int l = 0; l < c->L; ++l) a.n_mats += c->layer_kind[l] == CNERF_LAYER_RES ? 2 : 1;
    a.flags = c->flags;
    for (int l = 0; l < c->L; ++l) a.layer_kind[l] = c->layer_kind[l];
    a.philox = philox_of(c);
    a.image0 = image0;
    if (c->drop_p > 0.0f) {
        // ATen: noise.bernoulli_(1 - p).div_(1 - p): the factor is 1 / float(1 - p) in fp32
        a.drop_scale = 1.0f / (float)(1.0 - (double)c->drop_p);
        const double th = (double)c->drop_p * 4294967296.0 + 0.5;
        a.drop_thresh = th >= 4294967295.0 ? 0xffffffffu : (uint32_t)th;
        a.n_drop = 0;
        for (int l = 0; l < c->L; ++l) a.n_drop += c->layer_kind[l] != CNERF_LAYER_RES;
    }
    return CNERF_OK;
}

// which keep decisions a pass uses: the injected bytes, or Philox stream `stream_id` (fill_field_args set the rest)
void set_dropout(FieldArgs& a, const cnerf_cfg* c, const uint8_t* mask, uint32_t stream_id, long long n_per_image) {
    a.drop_mask = c->drop_p > 0.0f ? mask : nullptr;
    a.drop_stream = stream_id;
    a.drop_points = (long long)c->B * n_per_image;
}

hipError_t launch_forward(const FieldArgs& a, const cnerf_cfg* c, hipStream_t stream) {
    if (c->layer_kind[0] == CNERF_LAYER_PFILM && c->precision != CNERF_PREC_FP32)
        return c->precision == CNERF_PREC_FP16 ? launch_field_pw1(a, c->H, stream) : launch_field_pw3(a, c->H, stream);
    if (c->precision == CNERF_PREC_FP16X3) return launch_field_h3(a, c->H, stream);
    if (c->precision == CNERF_PREC_FP16) return launch_field_h1(a, c->H, stream);
    return launch_field(a, c->H, stream);
}

void set_points(FieldArgs& a, int B, long long n_per_image) {
    a.n_per_image = n_per_image;
    a.tiles_per_image = (n_per_image + 31) / 32;
    a.total_tiles = a.tiles_per_image * B;
}

}  // namespace

extern "C" {

int cnerf_abi_version(void) { return CNERF_ABI_VERSION; }

const char* cnerf_last_error(void) { return g_err; }

int cnerf_philox_fill(uint64_t seed, uint32_t offset, uint32_t stream_id, int64_t n, int32_t normal, float* out, void* stream) {
    g_err[0] = 0;
    if (!out || n < 1 || stream_id > 3) return fail(CNERF_EINVAL, "philox_fill: bad argument (stream_id 0..3)");
    const PhiloxKey k{1u, offset, (uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32)};
    if (hipError_t e = launch_philox_fill(k, stream_id, (long long)n, normal, out, (hipStream_t)stream)) return hip_fail(e, "philox_fill");
    return CNERF_OK;
}

int cnerf_workspace_bytes(const cnerf_cfg* cfg, size_t* packed, size_t* fvol_cl, size_t* fwd_ws) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, fwd_ws != nullptr)) return rc;
    const PackedLayout pl = packed_layout(cfg);
    if (packed) *packed = align256((pl.weight_floats + pl.bias_floats + 2 * (size_t)cfg->H) * sizeof(float));
    if (fvol_cl) {   // all levels together
        size_t fl = 0;
        for (int i = 0; i < n_levels_of(cfg); ++i)
            fl += (size_t)cfg->B * level_V_of(cfg, i) * level_V_of(cfg, i) * level_V_of(cfg, i) * level_C_of(cfg, i);
        *fvol_cl = align256(fl * sizeof(float));
    }
    if (fwd_ws) {
        const size_t N = (size_t)cfg->B * cfg->R * cfg->R * cfg->S;
        // coarse rgb_sigma + z, fine z + rgb_sigma; folded FiLM constants of the call (3 per image, matrix and channel)
        size_t mats = 0;
        for (int l = 0; l < cfg->L; ++l) mats += cfg->layer_kind[l] == CNERF_LAYER_RES ? 2 : 1;
        // ... 4 per image, matrix and channel; and, in fp32, the row-scaled layer weights per image
        const size_t NT = cfg->H / 32;
        const size_t layer_floats = (NT * pl.n_in + (mats - 1) * NT * NT) * 1024;
        *fwd_ws = 2 * align256(N * 4 * sizeof(float)) + 2 * align256(N * sizeof(float)) + align256((size_t)4 * cfg->B * mats * cfg->H * sizeof(float)) +
                  align256((size_t)cfg->B * layer_floats * sizeof(float));
        // fp16 precisions: the per-image copies of the whole packed weight stream (rows scaled by the image's FiLM frequencies), their
        // constants K S' / 1 / S' and the row multipliers (field_h3.hip, "weight folding per image") instead of the fp32 ones
        if (cfg->precision != CNERF_PREC_FP32 && cfg->layer_kind[0] != CNERF_LAYER_PFILM)
            *fwd_ws = 2 * align256(N * 4 * sizeof(float)) + 2 * align256(N * sizeof(float)) + align256((size_t)cfg->B * pl.weight_floats * sizeof(float)) +
                      align256((size_t)cfg->B * mats * (cfg->H + 1) * sizeof(float)) + align256((size_t)cfg->B * mats * cfg->H * sizeof(float));
    }
    return CNERF_OK;
}

int cnerf_fvol_channel_last(int32_t B, int32_t C, int32_t V, const float* fvol_cf, float* fvol_cl, void* stream) {
    g_err[0] = 0;
    if (B < 1 || V < 1 || C < 32 || C % 32 || !fvol_cf || !fvol_cl) return fail(CNERF_EINVAL, "fvol_channel_last: bad argument (C must be a multiple of 32)");
    if (hipError_t e = launch_transpose_cl(B, C, V, fvol_cf, fvol_cl, true, (hipStream_t)stream)) return hip_fail(e, "transpose");
    return CNERF_OK;
}

int cnerf_fvol_channel_first(int32_t B, int32_t C, int32_t V, const float* fvol_cl, float* fvol_cf, void* stream) {
    g_err[0] = 0;
    if (B < 1 || V < 1 || C < 32 || C % 32 || !fvol_cf || !fvol_cl) return fail(CNERF_EINVAL, "fvol_channel_first: bad argument (C must be a multiple of 32)");
    if (hipError_t e = launch_transpose_cl(B, C, V, fvol_cl, fvol_cf, false, (hipStream_t)stream)) return hip_fail(e, "transpose");
    return CNERF_OK;
}

int cnerf_pack_field(const cnerf_cfg* cfg, const cnerf_field_params* p, float* packed, void* stream_) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, false)) return rc;
    if (!p || !packed) return fail(CNERF_EINVAL, "pack_field: NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    const PackedLayout pl = packed_layout(cfg);
    const int H = cfg->H, NT = H / 32;
    float* wdst = packed;
    float* bdst = packed + pl.weight_floats;
    const size_t tile = 4 * 64 * 4;
    if (cfg->layer_kind[0] == CNERF_LAYER_PFILM) {
        if (cfg->C != 32 || cfg->n_levels > 1) return fail(CNERF_EINVAL, "per-point FiLM: a single 32-channel feature volume is supported");
        if (!p->map_w1 || !p->map_b1 || !p->map_w2 || !p->map_b2 || !p->w_final || !p->b_final)
            return fail(CNERF_EINVAL, "pack_field: mapping network / head is NULL");
        for (int l = 0; l < cfg->L; ++l)
            if (!p->w[l] || !p->b[l]) return fail(CNERF_EINVAL, "pack_field: layer %d weight/bias is NULL", l);
    }
    if (cfg->layer_kind[0] == CNERF_LAYER_PFILM && cfg->precision != CNERF_PREC_FP32) {      // field_pw16.hip, see packed_layout()
        const size_t parts = cfg->precision == CNERF_PREC_FP16 ? 1 : 2;
        const size_t frag = 64 * 8 / 2;
        const size_t big = 16 * parts * frag, small = 2 * (size_t)NT * parts * frag;
        auto pack16 = cfg->precision == CNERF_PREC_FP16 ? launch_pack_h1 : launch_pack_h3;
        const int L = cfg->L, n_slots = 3 * L + 2;
        float* consts = packed + pl.weight_floats;
        float* inv_s = consts + 256 + 3 * (size_t)L * H + 4 + (2 + 4 * (size_t)L + 3) / 4 * 4;     // [Wm1 | per layer: W_l, freq rows, phase rows | head]
        float* wmax = inv_s + n_slots;
        const size_t LH = (size_t)L * H;
        if (hipError_t e = pack16(p->map_w1, 256, cfg->C, 8, true, wdst, inv_s, wmax, stream, 0)) return hip_fail(e, "pack_h3");
        wdst += big;
        if (hipError_t e = pack16(p->w[0], H, 3, NT, true, wdst, inv_s + 1, wmax + 1, stream, 0)) return hip_fail(e, "pack_h3");
        wdst += small;
        for (int l = 0; l < L; ++l) {
            // the tiles of the layer's three matrices interleaved per output tile: [freq rows t | W_l t (l >= 1) | phase rows t]
            const long long t_stride = 16 + 16 + (l ? 2 * NT : 0);          // fragment pairs (one per k-chunk) per output tile
            const float* wf = p->map_w2 + (size_t)l * H * 256;
            const float* wp = p->map_w2 + (LH + (size_t)l * H) * 256;
            float* d = wdst;
            if (hipError_t e = pack16(wf, H, 256, NT, false, d, inv_s + 2 + 3 * l, wmax + 2 + 3 * l, stream, t_stride)) return hip_fail(e, "pack_h3");
            d += big;
            if (l) {
                if (hipError_t e = pack16(p->w[l], H, H, NT, false, d, inv_s + 1 + 3 * l, wmax + 1 + 3 * l, stream, t_stride)) return hip_fail(e, "pack_h3");
                d += small;
            }
            if (hipError_t e = pack16(wp, H, 256, NT, false, d, inv_s + 3 + 3 * l, wmax + 3 + 3 * l, stream, t_stride)) return hip_fail(e, "pack_h3");
            wdst += (size_t)NT * (2 * big + (l ? small : 0));
        }
        if (hipError_t e = pack16(p->w_final, 4, H, 1, false, wdst, inv_s + 3 * L + 1, wmax + 3 * L + 1, stream, 0)) return hip_fail(e, "pack_h3");
        if (hipError_t e = launch_pw16_consts(p, L, H, inv_s, consts, stream)) return hip_fail(e, "pw16_consts");
        return CNERF_OK;
    }
    if (cfg->layer_kind[0] == CNERF_LAYER_PFILM) {
        auto cp = [&](const float* src, size_t n) { return hipMemcpyAsync(bdst, src, n * sizeof(float), hipMemcpyDeviceToDevice, stream); };
        if (hipError_t e = launch_pack_matrix(p->map_w1, 256, cfg->C, 8, wdst, stream)) return hip_fail(e, "pack_matrix");
        wdst += 8 * tile;
        if (hipError_t e = cp(p->map_b1, 256)) return hip_fail(e, "bias copy");
        bdst += 256;
        const size_t LH = (size_t)cfg->L * H;
        for (int l = 0; l < cfg->L; ++l) {
            const int K = (l == 0) ? 3 : H;
            if (!p->w[l] || !p->b[l]) return fail(CNERF_EINVAL, "pack_field: layer %d weight/bias is NULL", l);
            if (hipError_t e = launch_pack_matrix(p->w[l], H, K, NT, wdst, stream)) return hip_fail(e, "pack_matrix");
            wdst += (size_t)NT * ((K + 31) / 32) * tile;
            if (hipError_t e = launch_pack_matrix(p->map_w2 + (size_t)l * H * 256, H, 256, NT, wdst, stream)) return hip_fail(e, "pack_matrix");
            wdst += (size_t)NT * 8 * tile;
            if (hipError_t e = launch_pack_matrix(p->map_w2 + (LH + (size_t)l * H) * 256, H, 256, NT, wdst, stream)) return hip_fail(e, "pack_matrix");
            wdst += (size_t)NT * 8 * tile;
            if (hipError_t e = cp(p->b[l], H)) return hip_fail(e, "bias copy");
            bdst += H;
            if (hipError_t e = cp(p->map_b2 + (size_t)l * H, H)) return hip_fail(e, "bias copy");
            bdst += H;
            if (hipError_t e = cp(p->map_b2 + LH + (size_t)l * H, H)) return hip_fail(e, "bias copy");
            bdst += H;
        }
        if (hipError_t e = launch_pack_head(p->w_final, H, wdst, stream)) return hip_fail(e, "pack_matrix");
        if (hipError_t e = cp(p->b_final, 4)) return hip_fail(e, "bias copy");
        bdst += 4;
        if (hipError_t e = launch_fill(bdst, 1.0f, H, stream)) return hip_fail(e, "fill");
        if (hipError_t e = launch_fill(bdst + H, 0.0f, H, stream)) return hip_fail(e, "fill");
        return CNERF_OK;
    }
    if (cfg->precision == CNERF_PREC_FP16X3 || cfg->precision == CNERF_PREC_FP16) {
        const size_t frag = 64 * 8 / 2;
        const size_t parts = cfg->precision == CNERF_PREC_FP16 ? 1 : 2;
        auto pack16 = cfg->precision == CNERF_PREC_FP16 ? launch_pack_h1 : launch_pack_h3;
        int mats = 0;
        for (int l = 0; l < cfg->L; ++l) mats += cfg->layer_kind[l] == CNERF_LAYER_RES ? 2 : 1;
        float* inv_scale = packed + pl.weight_floats + (size_t)mats * H + 4;     // 1/S per matrix and the head's, then the max|W| scratch
        float* wmax = inv_scale + mats + 1;
        int m = 0;
        for (int l = 0; l < cfg->L; ++l) {
            const int K = (l == 0) ? pl.k0 : H;
            const bool res = cfg->layer_kind[l] == CNERF_LAYER_RES;
            if (!p->w[l] || !p->b[l]) return fail(CNERF_EINVAL, "pack_field: layer %d weight/bias is NULL", l);
            if (res && (!p->w2[l] || !p->b2[l])) return fail(CNERF_EINVAL, "pack_field: residual layer %d fc2 is NULL", l);
            for (int half = 0; half < (res ? 2 : 1); ++half) {
                const float* w = half ? p->w2[l] : p->w[l];
                const float* b = half ? p->b2[l] : p->b[l];
                if (hipError_t e = pack16(w, H, K, NT, l == 0, wdst, inv_scale + m, wmax + m, stream, 0)) return hip_fail(e, "pack_h3");
                wdst += (size_t)NT * ((K + 31) / 32 * 2) * parts * frag;
                if (hipError_t e = hipMemcpyAsync(bdst, b, H * sizeof(float), hipMemcpyDeviceToDevice, stream)) return hip_fail(e, "bias copy");
                bdst += H;
                ++m;
            }
        }
        if (!p->w_final || !p->b_final) return fail(CNERF_EINVAL, "pack_field: head is NULL");
        if (hipError_t e = pack16(p->w_final, 4, H, 1, false, wdst, inv_scale + m, wmax + m, stream, 0)) return hip_fail(e, "pack_h3");
        if (hipError_t e = hipMemcpyAsync(bdst, p->b_final, 4 * sizeof(float), hipMemcpyDeviceToDevice, stream)) return hip_fail(e, "bias copy");
        bdst = packed + pl.weight_floats + pl.bias_floats;
        if (hipError_t e = launch_fill(bdst, 1.0f, H, stream)) return hip_fail(e, "fill");
        if (hipError_t e = launch_fill(bdst + H, 0.0f, H, stream)) return hip_fail(e, "fill");
        return CNERF_OK;
    }
    for (int l = 0; l < cfg->L; ++l) {
        const int K = (l == 0) ? pl.k0 : H;                    // layer 0: C (+3), zero padded to 32 * n_in columns
        if (!p->w[l] || !p->b[l]) return fail(CNERF_EINVAL, "pack_field: layer %d weight/bias is NULL", l);
        if (hipError_t e = launch_pack_matrix(p->w[l], H, K, NT, wdst, stream)) return hip_fail(e, "pack_matrix");
        wdst += (size_t)NT * ((K + 31) / 32) * tile;
        if (hipError_t e = hipMemcpyAsync(bdst, p->b[l], H * sizeof(float), hipMemcpyDeviceToDevice, stream)) return hip_fail(e, "bias copy");
        bdst += H;
        if (cfg->layer_kind[l] == CNERF_LAYER_RES) {
            if (!p->w2[l] || !p->b2[l]) return fail(CNERF_EINVAL, "pack_field: residual layer %d fc2 is NULL", l);
            if (hipError_t e = launch_pack_matrix(p->w2[l], H, H, NT, wdst, stream)) return hip_fail(e, "pack_matrix");
            wdst += (size_t)NT * NT * tile;
            if (hipError_t e = hipMemcpyAsync(bdst, p->b2[l], H * sizeof(float), hipMemcpyDeviceToDevice, stream)) return hip_fail(e, "bias copy");
            bdst += H;
        }
    }
    if (!p->w_final || !p->b_final) return fail(CNERF_EINVAL, "pack_field: head is NULL");
    if (hipError_t e = launch_pack_head(p->w_final, H, wdst, stream)) return hip_fail(e, "pack_matrix");
    if (hipError_t e = hipMemcpyAsync(bdst, p->b_final, 4 * sizeof(float), hipMemcpyDeviceToDevice, stream)) return hip_fail(e, "bias copy");
    bdst += 4;
    if (hipError_t e = launch_fill(bdst, 1.0f, H, stream)) return hip_fail(e, "fill");
    if (hipError_t e = launch_fill(bdst + H, 0.0f, H, stream)) return hip_fail(e, "fill");
    return CNERF_OK;
}

int cnerf_gather_features(const cnerf_cfg* cfg, const float* fvol_cl, const float* points, int64_t n_per_image,
                          float* feat, void* stream) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, false)) return rc;
    if (!fvol_cl || !points || !feat || n_per_image < 1) return fail(CNERF_EINVAL, "gather_features: bad argument");
    if (cfg->C != 32 || cfg->n_levels > 1) return fail(CNERF_EINVAL, "gather_features: single 32-channel volume only");
    GatherArgs a{fvol_cl, points, feat, (long long)n_per_image, cfg->B, cfg->V, cfg->C, cfg->voxel_length / 2.0f};
    if (hipError_t e = launch_gather(a, (hipStream_t)stream)) return hip_fail(e, "gather");
    return CNERF_OK;
}

int cnerf_weight_grad(int32_t n_images, int64_t n_per_image, int32_t H, int32_t K, const float* g_arg, const float* x, float* dW,
                      float* colsum, void* stream) {
    g_err[0] = 0;
    if (!g_arg || !x || !dW || !colsum) return fail(CNERF_EINVAL, "weight_grad: NULL argument");
    if (n_images < 1 || n_per_image < 1) return fail(CNERF_EINVAL, "weight_grad: empty chunk");
    if ((H != 64 && H != 128 && H != 256) || K < 32 || K > 256 || K % 32)
        return fail(CNERF_EINVAL, "weight_grad: H=%d must be 64/128/256 and K=%d a multiple of 32 up to 256", H, K);
    if (hipError_t e = launch_weight_grad(n_images, n_per_image, H, K, g_arg, x, dW, colsum, (hipStream_t)stream))
        return hip_fail(e, "weight_grad");
    return CNERF_OK;
}

int cnerf_scatter_features(const cnerf_cfg* cfg, const float* points, int64_t n_per_image, const float* grad_feat,
                           float* grad_fvol_cl, void* stream) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, false)) return rc;
    if (!grad_fvol_cl || !points || !grad_feat || n_per_image < 1) return fail(CNERF_EINVAL, "scatter_features: bad argument");
    if (cfg->C != 32 || cfg->n_levels > 1) return fail(CNERF_EINVAL, "scatter_features: single 32-channel volume only");
    GatherArgs a{nullptr, points, nullptr, (long long)n_per_image, cfg->B, cfg->V, cfg->C, cfg->voxel_length / 2.0f};
    if (hipError_t e = launch_scatter(a, grad_feat, grad_fvol_cl, (hipStream_t)stream)) return hip_fail(e, "scatter");
    return CNERF_OK;
}

int cnerf_field_forward(const cnerf_cfg* cfg, const cnerf_volumes* vols, const float* packed, const float* freq,
                        const float* phase, const float* points, int64_t n_per_image, float* rgb_sigma, void* stream) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, false)) return rc;
    if (!vols || !packed || !points || !rgb_sigma || n_per_image < 1) return fail(CNERF_EINVAL, "field_forward: bad argument");
    const PackedLayout pl = packed_layout(cfg);
    if (pl.n_film && (!freq || !phase)) return fail(CNERF_EINVAL, "field_forward: FiLM layers need freq and phase");
    FieldArgs a;
    if (int rc = fill_field_args(a, cfg, vols, nullptr, packed, freq, phase)) return rc;
    a.mode = FIELD_MODE_POINTS;
    a.points = points;
    a.rgb_sigma = rgb_sigma;
    set_points(a, cfg->B, n_per_image);
    set_dropout(a, cfg, nullptr, PHILOX_DROP_POINTS, n_per_image);
    if (hipError_t e = launch_forward(a, cfg, (hipStream_t)stream)) return hip_fail(e, "field kernel");
    return CNERF_OK;
}

int cnerf_composite(const cnerf_cfg* cfg, int64_t rays, int32_t n, const float* rgb_sigma, const float* z,
                    const float* eps, float* rgb, float* dist, float* weights, void* stream) {
    g_err[0] = 0;
    if (!cfg || rays < 1 || n < 1 || n > 256 || !rgb_sigma || !z) return fail(CNERF_EINVAL, "composite: bad argument (1 <= n <= 256)");
    CompositeArgs a{rgb_sigma, z, eps, rgb, dist, weights, (long long)rays, n, cfg->noise_std, cfg->flags};
    if (hipError_t e = launch_composite(a, (hipStream_t)stream)) return hip_fail(e, "composite");
    return CNERF_OK;
}

int cnerf_resample(int64_t rays, int32_t S, const float* z, const float* weights, const float* u, float* fine_z,
                   int32_t* inds, float* cdf, void* stream) {
    g_err[0] = 0;
    if (rays < 1 || S < 2 || S > 128 || !z || !weights || !u || !fine_z) return fail(CNERF_EINVAL, "resample: bad argument (2 <= S <= 128)");
    ResampleArgs a{z, weights, nullptr, nullptr, u, fine_z, inds, cdf, nullptr, (long long)rays, S, 0.0f, 0u, PhiloxKey{0u, 0u, 0u, 0u}};
    if (hipError_t e = launch_resample(a, (hipStream_t)stream)) return hip_fail(e, "resample");
    return CNERF_OK;
}

int cnerf_render_forward(const cnerf_cfg* cfg, const cnerf_volumes* vols, const float* packed, const float* freq,
                         const float* phase, const float* cam2world, const cnerf_rng* rng, float* pixels,
                         float* depth, const cnerf_aux* aux, void* workspace, void* stream_) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, true)) return rc;
    if (!vols || !packed || !cam2world || !pixels || !depth || !workspace) return fail(CNERF_EINVAL, "render_forward: NULL argument");
    const PackedLayout pl = packed_layout(cfg);
    if (pl.n_film && (!freq || !phase)) return fail(CNERF_EINVAL, "render_forward: FiLM layers need freq and phase");
    const bool hier = cfg->flags & CNERF_F_HIERARCHICAL;
    static const cnerf_rng no_rng = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!rng) rng = &no_rng;
    if (hier && !rng->u_fine && !cfg->philox) return fail(CNERF_EINVAL, "render_forward: hierarchical sampling needs rng.u_fine (or cfg.philox)");
    hipStream_t stream = (hipStream_t)stream_;

    const long long P = (long long)cfg->R * cfg->R, S = cfg->S;
    const long long npi = P * S;
    const size_t N = (size_t)cfg->B * npi;
    char* ws = (char*)workspace;
    float* c_rs = (float*)ws; ws += align256(N * 4 * sizeof(float));
    float* f_rs = (float*)ws; ws += align256(N * 4 * sizeof(float));
    float* c_z = (float*)ws;  ws += align256(N * sizeof(float));
    float* f_z = (float*)ws;  ws += align256(N * sizeof(float));
    float* fold = (float*)ws;
    {
        size_t mats = 0;
        for (int l = 0; l < cfg->L; ++l) mats += cfg->layer_kind[l] == CNERF_LAYER_RES ? 2 : 1;
        ws += align256((size_t)4 * cfg->B * mats * cfg->H * sizeof(float));
    }
    float* packed_img = (float*)ws;
    if (aux) {   // write straight into the caller's buffers where given
        if (aux->coarse_rgb_sigma) c_rs = aux->coarse_rgb_sigma;
        if (aux->fine_rgb_sigma) f_rs = aux->fine_rgb_sigma;
        if (aux->coarse_z) c_z = aux->coarse_z;
        if (aux->fine_z) f_z = aux->fine_z;
    }

    FieldArgs fa;
    if (int rc = fill_field_args(fa, cfg, vols, nullptr, packed, freq, phase)) return rc;
    set_points(fa, cfg->B, npi);
    fa.cam2world = cam2world;
    // the exact fp32 precision: FiLM (plain sine: freq 1, phase 0) folded into one affine map per matrix and channel, prepared once
    // for both passes (per-point FiLM computes its frequencies per point: not foldable)
    if (cfg->precision == CNERF_PREC_FP32 && cfg->layer_kind[0] != CNERF_LAYER_PFILM && cfg->drop_p == 0.0f) {
        if (hipError_t e = launch_fold_film(fa, cfg->B, cfg->H, fold, stream)) return hip_fail(e, "fold_film");
        fa.fold = fold;
        fa.fold_images = cfg->B;
        {                                 // the scale goes into per-image copies of the layer weights (WFOLD, field_kernel.hip)
            const size_t NT = cfg->H / 32;
            const long long layer_floats = (long long)((NT * pl.n_in + (size_t)(fa.n_mats - 1) * NT * NT) * 1024);
            if (hipError_t e = launch_scale_packed(fa, cfg->B, cfg->H, fold, layer_floats, packed_img, stream)) return hip_fail(e, "scale_packed");
            fa.packed_img = packed_img;
            fa.packed_img_stride = layer_floats;
        }
    }
    // the fp16 precisions: the same folding on two-part fp16 weights -- per-image copies of the packed stream, the accumulators start
    // from (freq bias + phase) / 2 pi, the epilogue is one multiply, the range reduction and v_sin (field_h3.hip)
    if (cfg->precision != CNERF_PREC_FP32 && cfg->layer_kind[0] != CNERF_LAYER_PFILM) {
        char* w16 = (char*)fold;                  // (the fp32 regions are not used in these precisions: same place in the workspace)
        void* img = w16;
        w16 += align256((size_t)cfg->B * pl.weight_floats * sizeof(float));
        float* fold16 = (float*)w16;
        w16 += align256((size_t)cfg->B * fa.n_mats * (cfg->H + 1) * sizeof(float));
        float* rowf = (float*)w16;
        const long long img_elems = (long long)pl.weight_floats * 2;            // fp16 elements of the packed weight stream
        if (hipError_t e = (cfg->precision == CNERF_PREC_FP16 ? launch_fold_h1 : launch_fold_h3)(fa, cfg->B, cfg->H, img, fold16, rowf, img_elems, stream))
            return hip_fail(e, "fold16");
        fa.packed_img = (const float*)img;
        fa.packed_img_stride = img_elems / 8;     // in f16x8 fragments
        fa.fold = fold16;
        fa.fold_images = cfg->B;
    }
    // 1. coarse pass
    fa.mode = FIELD_MODE_COARSE;
    fa.u_strat = rng->u_strat;
    fa.rgb_sigma = c_rs;
    fa.z_out = c_z;
    fa.points_out = aux ? aux->coarse_points : nullptr;
#ifdef CNERF_STAMPS
    fa.stamps = aux ? (unsigned long long*)aux->cdf : nullptr;   // diagnostic build, non-hierarchical call: cdf slot is never written
#endif
    auto mark = [&](int i) {
        if (aux && aux->field_events[i]) (void)hipEventRecord((hipEvent_t)aux->field_events[i], stream);
    };
    const bool keep = aux && aux->act16[0].h;
    if (keep) {
        if (cfg->precision != CNERF_PREC_FP16X3 && cfg->precision != CNERF_PREC_FP16)
            return fail(CNERF_EINVAL, "render_forward: act16 needs precision CNERF_PREC_FP16X3 or CNERF_PREC_FP16");
        if (!aux->act16[0].feat || !aux->act16[0].c || (hier && (!aux->act16[1].feat || !aux->act16[1].h || !aux->act16[1].c)))
            return fail(CNERF_EINVAL, "render_forward: act16 is incomplete");
        if (cfg->layer_kind[0] == CNERF_LAYER_PFILM) {
            if (cfg->precision != CNERF_PREC_FP16X3) return fail(CNERF_EINVAL, "render_forward: per-point FiLM keeps its activations in precision CNERF_PREC_FP16X3 only");
            if (!aux->act16[0].amax || (hier && !aux->act16[1].amax)) return fail(CNERF_EINVAL, "render_forward: act16.amax is NULL (per-point FiLM)");
        }
    }
    auto keep_pass = [&](int i) {
        fa.act_points = (long long)cfg->B * npi;
        fa.act_feat = keep ? (float*)aux->act16[i].feat : nullptr;
        fa.act_h = keep ? (float*)aux->act16[i].h : nullptr;
        fa.act_c = keep ? (float*)aux->act16[i].c : nullptr;
        fa.act_amax = keep ? (float*)aux->act16[i].amax : nullptr;
        fa.act_tb16 = keep ? 1 : 0;
    };
    keep_pass(0);
    set_dropout(fa, cfg, rng->drop_coarse, PHILOX_DROP_COARSE, npi);
    mark(0);
    if (hipError_t e = launch_forward(fa, cfg, stream)) return hip_fail(e, "field kernel (coarse)");
    mark(1);

    if (hier) {
        // 2. coarse weights -> inverse-CDF depths
        ResampleArgs ra{c_z, nullptr, c_rs, rng->eps_coarse, rng->u_fine, f_z, aux ? aux->inds : nullptr,
                        aux ? aux->cdf : nullptr, aux ? aux->coarse_weights : nullptr, (long long)cfg->B * P, (int)S,
                        cfg->noise_std, cfg->flags, philox_of(cfg)};
        if (hipError_t e = launch_resample(ra, stream)) return hip_fail(e, "resample");
        // 3. fine pass
        fa.mode = FIELD_MODE_FINE;
        fa.u_strat = nullptr;
        if (rng->fine_z) f_z = const_cast<float*>(rng->fine_z);   // teacher-forced depths (read only from here on)
        fa.fine_z = f_z;
        fa.rgb_sigma = f_rs;
        fa.z_out = nullptr;
        fa.points_out = aux ? aux->fine_points : nullptr;
        keep_pass(1);
        set_dropout(fa, cfg, rng->drop_fine, PHILOX_DROP_FINE, npi);
        mark(2);
        if (hipError_t e = launch_forward(fa, cfg, stream)) return hip_fail(e, "field kernel (fine)");
        mark(3);
    }
    // 4. merge + composite + epilogue
    MergeArgs ma{c_rs, c_z, hier ? f_rs : nullptr, hier ? f_z : nullptr, rng->eps_final, pixels, depth,
                 aux ? aux->sort_idx : nullptr, aux ? aux->final_weights : nullptr, (long long)cfg->B * P, (int)S,
                 make_geom(cfg), cfg->noise_std, cfg->flags, philox_of(cfg)};
    if (hipError_t e = launch_merge_composite(ma, stream)) return hip_fail(e, "merge_composite");
    return CNERF_OK;
}

int cnerf_backward_bytes(const cnerf_cfg* cfg, size_t* packed_t) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, false)) return rc;
    const size_t NT = cfg->H / 32, tile = 4 * 64 * 4;
    size_t fl = NT * 2 * 64;                                  // head^T
    for (int l = cfg->L - 1; l >= 1; --l) fl += (cfg->layer_kind[l] == CNERF_LAYER_RES ? 2 : 1) * NT * NT * tile;
    if (cfg->layer_kind[0] != CNERF_LAYER_PFILM)             // (per-point FiLM: layer 0 reads the sample position, no gradient)
        fl += (size_t)packed_layout(cfg).n_in * NT * tile;   // layer 0 transposed: one 32-row output tile per input tile
    if (packed_t) *packed_t = align256(fl * sizeof(float));
    return CNERF_OK;
}

int cnerf_pack_field_transposed(const cnerf_cfg* cfg, const cnerf_field_params* p, float* packed_t, void* stream_) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, false)) return rc;
    if (!p || !packed_t) return fail(CNERF_EINVAL, "pack_field_transposed: NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    const int H = cfg->H, NT = H / 32;
    const size_t tile = 4 * 64 * 4;
    float* dst = packed_t;
    if (!p->w_final) return fail(CNERF_EINVAL, "pack_field_transposed: head is NULL");
    if (hipError_t e = launch_pack_head_t(p->w_final, H, dst, stream)) return hip_fail(e, "pack_head_t");
    dst += (size_t)NT * 2 * 64;
    for (int l = cfg->L - 1; l >= 1; --l) {      // in the order the backward consumes them (a residual block: fc2 then fc1)
        if (!p->w[l]) return fail(CNERF_EINVAL, "pack_field_transposed: layer %d weight is NULL", l);
        if (cfg->layer_kind[l] == CNERF_LAYER_RES) {
            if (!p->w2[l]) return fail(CNERF_EINVAL, "pack_field_transposed: residual layer %d fc2 is NULL", l);
            if (hipError_t e = launch_pack_matrix_t(p->w2[l], H, H, NT, dst, stream)) return hip_fail(e, "pack_matrix_t");
            dst += (size_t)NT * NT * tile;
        }
        if (hipError_t e = launch_pack_matrix_t(p->w[l], H, H, NT, dst, stream)) return hip_fail(e, "pack_matrix_t");
        dst += (size_t)NT * NT * tile;
    }
    if (cfg->layer_kind[0] == CNERF_LAYER_PFILM) return CNERF_OK;      // head^T and W_l^T of layers L-1..1 only
    if (!p->w[0]) return fail(CNERF_EINVAL, "pack_field_transposed: layer 0 weight is NULL");
    const PackedLayout pl = packed_layout(cfg);
    if (hipError_t e = launch_pack_matrix_t(p->w[0], H, pl.k0, pl.n_in, dst, stream)) return hip_fail(e, "pack_matrix_t");
    return CNERF_OK;
}

int cnerf_merge_composite_backward(const cnerf_cfg* cfg, const float* coarse_rgb_sigma, const float* coarse_z,
                                   const float* fine_rgb_sigma, const float* fine_z, const float* eps_final,
                                   const float* grad_pixels, const float* grad_depth, float* grad_coarse,
                                   float* grad_fine, void* stream) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, true)) return rc;
    const bool hier = cfg->flags & CNERF_F_HIERARCHICAL;
    if (!coarse_rgb_sigma || !coarse_z || !grad_pixels || !grad_coarse) return fail(CNERF_EINVAL, "merge_composite_backward: NULL argument");
    if (hier && (!fine_rgb_sigma || !fine_z || !grad_fine)) return fail(CNERF_EINVAL, "merge_composite_backward: fine tensors missing");
    MergeBwdArgs a{coarse_rgb_sigma, coarse_z, hier ? fine_rgb_sigma : nullptr, hier ? fine_z : nullptr, eps_final,
                   grad_pixels, grad_depth, grad_coarse, hier ? grad_fine : nullptr,
                   (long long)cfg->B * cfg->R * cfg->R, cfg->S, make_geom(cfg), cfg->noise_std, cfg->flags, philox_of(cfg)};
    if (hipError_t e = launch_merge_composite_backward(a, (hipStream_t)stream)) return hip_fail(e, "merge_composite_backward");
    return CNERF_OK;
}

int cnerf_field_backward(const cnerf_cfg* cfg, int32_t pass, int32_t image0, int32_t n_images, const cnerf_volumes* vols,
                         const float* packed, const float* packed_t, const float* freq, const float* phase,
                         const float* cam2world, const float* u_strat, const float* fine_z,
                         const float* grad_rgb_sigma, const float* saved_rgb_sigma, float* act_feat, float* act_h,
                         float* act_c, float* act_g, float* act_go, const cnerf_grad_volumes* grad_vols, const uint8_t* drop_mask,
                         void* stream_) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, true)) return rc;
    if (image0 < 0 || n_images < 1 || image0 + n_images > cfg->B) return fail(CNERF_EINVAL, "field_backward: image range out of [0,B)");
    if (pass < 0 || pass > 2) return fail(CNERF_EINVAL, "field_backward: pass must be 0 (coarse), 1 (fine) or 2 (explicit points)");
    if (cfg->layer_kind[0] == CNERF_LAYER_PFILM && cfg->precision != CNERF_PREC_FP32)
        return fail(CNERF_EINVAL, "field_backward: the per-point FiLM family's fp32 chain needs a cfg (and packed weights) of precision fp32");
    if (!vols || !packed || !packed_t || !cam2world || !grad_rgb_sigma || !saved_rgb_sigma || !act_feat || !act_h || !act_c ||
        !act_g || !act_go || !grad_vols)
        return fail(CNERF_EINVAL, "field_backward: NULL argument");
    for (int i = 0; i < n_levels_of(cfg); ++i)
        if (!grad_vols->level[i]) return fail(CNERF_EINVAL, "field_backward: gradient volume %d is NULL", i);
    if (pass == 1 && !fine_z) return fail(CNERF_EINVAL, "field_backward: the fine pass needs fine_z");
    const PackedLayout pl = packed_layout(cfg);
    if (pl.n_film && (!freq || !phase)) return fail(CNERF_EINVAL, "field_backward: FiLM layers need freq and phase");
    hipStream_t stream = (hipStream_t)stream_;
    const long long npi = (long long)cfg->R * cfg->R * cfg->S;

    FieldArgs fa;
    if (int rc = fill_field_args(fa, cfg, vols, grad_vols, packed, freq, phase, image0)) return rc;
    set_points(fa, n_images, npi);
    fa.cam2world = cam2world + (size_t)image0 * 16;
    if (pass == 0) {
        fa.mode = FIELD_MODE_COARSE;
        fa.u_strat = u_strat ? u_strat + (size_t)image0 * npi : nullptr;
    } else if (pass == 1) {
        fa.mode = FIELD_MODE_FINE;
        fa.fine_z = fine_z + (size_t)image0 * npi;
    } else {
        if (!u_strat) return fail(CNERF_EINVAL, "field_backward: pass 2 takes the points (B,R*R*S,3) in the u_strat argument");
        fa.mode = FIELD_MODE_POINTS;
        fa.points = u_strat + (size_t)image0 * npi * 3;
    }
    fa.rgb_sigma = act_go;          // the re-run forward needs somewhere to put its head output: overwritten below by go'
    fa.z_out = nullptr;
    fa.act_points = (long long)n_images * npi;
    fa.act_feat = act_feat;
    fa.act_h = act_h;
    fa.act_c = act_c;
    set_dropout(fa, cfg, drop_mask, pass == 0 ? PHILOX_DROP_COARSE : pass == 1 ? PHILOX_DROP_FINE : PHILOX_DROP_POINTS, npi);
    // the activation-storing forward runs in cfg->precision (`packed` is in that precision's layout); the gradient chain
    // below is fp32 on the transposed fp32 weights either way
    if (hipError_t e = launch_forward(fa, cfg, stream)) return hip_fail(e, "field kernel (activation store)");
    fa.packed_t = packed_t;
    fa.grad_out = grad_rgb_sigma + (size_t)image0 * npi * 4;
    fa.saved_out = saved_rgb_sigma + (size_t)image0 * npi * 4;
    fa.act_g = act_g;
    fa.act_go = act_go;
    if (hipError_t e = launch_field_backward(fa, cfg->H, stream)) return hip_fail(e, "field backward kernel");
    return CNERF_OK;
}

int cnerf_weight_grad16(int32_t n_images, int64_t tiles_per_image, int32_t n_rows, int32_t g_ct, int32_t x_ct, const void* G,
                        const void* X, float* dW, float* colsum, const float* inv_scale, void* stream) {
    g_err[0] = 0;
    if (!G || !X || !dW) return fail(CNERF_EINVAL, "weight_grad16: NULL argument");
    if (n_images < 1 || tiles_per_image < 1) return fail(CNERF_EINVAL, "weight_grad16: empty chunk");
    const int noc = (n_rows + 31) / 32;
    if (n_rows < 1 || (noc != 1 && noc != 2 && noc != 4 && noc != 8) || noc > g_ct || x_ct < 1 || x_ct > 8)
        return fail(CNERF_EINVAL, "weight_grad16: n_rows=%d (1..256, 1/2/4/8 channel tiles), g_ct=%d, x_ct=%d (1..8) unsupported", n_rows, g_ct, x_ct);
    if (hipError_t e = launch_weight_grad16(n_images, tiles_per_image, n_rows, g_ct, x_ct, G, X, dW, colsum, inv_scale, (hipStream_t)stream))
        return hip_fail(e, "weight_grad16");
    return CNERF_OK;
}

namespace {
// packed16 layout: [transposed units of matrices n_mats-1 .. 1, then layer 0 (output tiles padded to even)][head^T: NT fragments
// x 64 lanes][winv: n_mats + 1 floats, then ||W||_1: n_mats + 1 floats][max|W| scratch: n_mats + 1 uint32], each section
// 256-byte aligned
struct Chain16Layout {
    size_t units_bytes, head_off, winv_off, wmax_off, total;
    int n_mats, n_in, k0, ot0;
};
int chain16_layout(const cnerf_cfg* c, Chain16Layout& l) {
    l.n_mats = 0;
    for (int i = 0; i < c->L; ++i) {
        if (c->layer_kind[i] == CNERF_LAYER_PFILM)
            return fail(CNERF_ENOSYS, "half-precision backward: FiLM / plain-sine / residual layers only (layer %d is per-point FiLM)", i);
        l.n_mats += c->layer_kind[i] == CNERF_LAYER_RES ? 2 : 1;      // a residual block: fc1 and fc2
    }
    const size_t NT = c->H / 32, KCH = 2 * NT, frag = 64 * 16;
    const PackedLayout pl = packed_layout(c);
    l.n_in = pl.n_in;
    l.k0 = pl.k0;
    l.ot0 = (pl.n_in + 1) / 2 * 2;
    l.units_bytes = ((size_t)(l.n_mats - 1) * NT + l.ot0) * KCH * frag;
    l.head_off = align256(l.units_bytes);
    l.winv_off = l.head_off + align256(NT * frag);
    l.wmax_off = l.winv_off + align256((size_t)2 * (l.n_mats + 1) * sizeof(float));
    l.total = l.wmax_off + align256((size_t)(l.n_mats + 1) * sizeof(uint32_t));
    return CNERF_OK;
}
}  // namespace

namespace {
// per-point FiLM family (chain_pw16.hip): [Y units][M units + Wm1^T][head^T: NT fragments x 64 lanes][winv: 2 L + 2][anorm: L + 1][max|W| scratch: 2 L + 2]
struct PwChainLayout {
    size_t m_off, head_off, winv_off, anorm_off, wmax_off, total;
};
PwChainLayout pw_chain_layout(const cnerf_cfg* c) {
    PwChainLayout l;
    const size_t NT = c->H / 32, L = c->L;
    l.m_off = align256(pw_chain_y_bytes(c->L, c->H));
    l.head_off = l.m_off + align256(pw_chain_m_bytes(c->L, c->H));
    l.winv_off = l.head_off + align256(NT * 1024);
    l.anorm_off = l.winv_off + align256((2 * L + 2) * sizeof(float));
    l.wmax_off = l.anorm_off + align256((L + 1) * sizeof(float));
    l.total = l.wmax_off + align256((2 * L + 2) * sizeof(uint32_t));
    return l;
}
}  // namespace

int cnerf_backward16_bytes(const cnerf_cfg* cfg, size_t* packed16) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, false)) return rc;
    if (cfg->layer_kind[0] == CNERF_LAYER_PFILM) {
        if (packed16) *packed16 = pw_chain_layout(cfg).total;
        return CNERF_OK;
    }
    Chain16Layout l;
    if (int rc = chain16_layout(cfg, l)) return rc;
    if (packed16) *packed16 = l.total;
    return CNERF_OK;
}

int cnerf_pack_field_chain16(const cnerf_cfg* cfg, const cnerf_field_params* p, void* packed16, void* stream_) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, false)) return rc;
    if (cfg->layer_kind[0] == CNERF_LAYER_PFILM) {
        if (!p || !packed16 || !p->w_final || !p->map_w1 || !p->map_w2) return fail(CNERF_EINVAL, "pack_field_chain16: NULL argument");
        for (int l = 1; l < cfg->L; ++l)
            if (!p->w[l]) return fail(CNERF_EINVAL, "pack_field_chain16: layer %d weight is NULL", l);
        if (cfg->C != 32 || cfg->n_levels > 1) return fail(CNERF_EINVAL, "per-point FiLM: a single 32-channel feature volume is supported");
        const PwChainLayout l = pw_chain_layout(cfg);
        char* base = (char*)packed16;
        if (hipError_t e = launch_pack_pw_chain(p, cfg->L, cfg->H, base, base + l.m_off, base + l.head_off, (float*)(base + l.winv_off),
                                                (float*)(base + l.anorm_off), (uint32_t*)(base + l.wmax_off), (hipStream_t)stream_))
            return hip_fail(e, "pack_pw_chain");
        return CNERF_OK;
    }
    Chain16Layout l;
    if (int rc = chain16_layout(cfg, l)) return rc;
    if (!p || !packed16 || !p->w_final) return fail(CNERF_EINVAL, "pack_field_chain16: NULL argument");
    hipStream_t stream = (hipStream_t)stream_;
    const int H = cfg->H, NT = H / 32;
    const size_t KCH = 2 * NT, frag = 64 * 16;
    char* base = (char*)packed16;
    float* winv = (float*)(base + l.winv_off);
    uint32_t* wmax = (uint32_t*)(base + l.wmax_off);
    char* dst = base;
    const float* mats[2 * CNERF_MAX_LAYERS];         // the matrices in slab order (a residual block: fc1, fc2)
    int nm = 0;
    for (int i = 0; i < cfg->L; ++i) {
        if (!p->w[i]) return fail(CNERF_EINVAL, "pack_field_chain16: layer %d weight is NULL", i);
        mats[nm++] = p->w[i];
        if (cfg->layer_kind[i] == CNERF_LAYER_RES) {
            if (!p->w2[i]) return fail(CNERF_EINVAL, "pack_field_chain16: residual layer %d fc2 is NULL", i);
            mats[nm++] = p->w2[i];
        }
    }
    const int M = l.n_mats;
    for (int m = M - 1; m >= 1; --m) {               // consumption order of the chain
        if (hipError_t e = launch_pack_t16(mats[m], H, H, H, NT, dst, winv + m, wmax + m, stream)) return hip_fail(e, "pack_t16");
        dst += (size_t)NT * KCH * frag;
    }
    if (hipError_t e = launch_pack_t16(mats[0], H, l.k0, l.k0, l.ot0, dst, winv + 0, wmax + 0, stream)) return hip_fail(e, "pack_t16");
    if (hipError_t e = launch_pack_head_t16(p->w_final, H, base + l.head_off, winv + M, wmax + M, stream)) return hip_fail(e, "pack_head_t16");
    float* anorm = winv + M + 1;                       // ||W_m||_1 per matrix, then the head's (4 x H: max over channels of the 4-term sum)
    for (int m = 0; m < M; ++m)
        if (hipError_t e = launch_col_abs_sum_max(mats[m], H, m == 0 ? l.k0 : H, anorm + m, stream)) return hip_fail(e, "col_abs_sum_max");
    if (hipError_t e = launch_col_abs_sum_max(p->w_final, 4, H, anorm + M, stream)) return hip_fail(e, "col_abs_sum_max");
    return CNERF_OK;
}

namespace {
int field_backward16_impl(const cnerf_cfg* cfg, uint32_t mode, int32_t group_step, int32_t pass, int32_t image0, int32_t n_images,
                          const cnerf_volumes* vols, const float* packed, const void* packed16, const float* freq,
                          const float* phase, const float* cam2world, const float* u_strat, const float* fine_z,
                          const float* grad_rgb_sigma, const float* saved_rgb_sigma, void* act_feat16, void* act_h16, void* act_c16,
                          void* act_g16, void* act_go16, const float* scales, uint32_t* gmax, const cnerf_grad_volumes* grad_vols,
                          uint32_t* sat, float* gin, void* stream_) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, true)) return rc;
    Chain16Layout l;
    if (int rc = chain16_layout(cfg, l)) return rc;
    if (cfg->precision != CNERF_PREC_FP16X3) return fail(CNERF_EINVAL, "field_backward16: cfg->precision must be CNERF_PREC_FP16X3 (the re-run forward is that kernel)");
    if (image0 < 0 || n_images < 1 || image0 + n_images > cfg->B) return fail(CNERF_EINVAL, "field_backward16: image range out of [0,B)");
    if (pass < 0 || pass > 2) return fail(CNERF_EINVAL, "field_backward16: pass must be 0 (coarse), 1 (fine) or 2 (explicit points)");
    if (!(mode & (CNERF_B16_STORE | CNERF_B16_DRY | CNERF_B16_CHAIN))) return fail(CNERF_EINVAL, "field_backward16: empty mode");
    if (!vols || !packed || !packed16 || !cam2world || !act_feat16 || !act_h16 || !act_c16 || !scales)
        return fail(CNERF_EINVAL, "field_backward16: NULL argument");
    if ((mode & (CNERF_B16_DRY | CNERF_B16_CHAIN)) && (!grad_rgb_sigma || !saved_rgb_sigma)) return fail(CNERF_EINVAL, "field_backward16: the chain needs grad / saved rgb_sigma");
    if ((mode & CNERF_B16_DRY) && !gmax) return fail(CNERF_EINVAL, "field_backward16: the dry run needs gmax");
    if ((mode & CNERF_B16_CHAIN) && (!act_g16 || !act_go16 || !grad_vols)) return fail(CNERF_EINVAL, "field_backward16: the chain needs act_g16, act_go16, grad_vols");
    if (mode & CNERF_B16_CHAIN)
        for (int i = 0; i < n_levels_of(cfg); ++i)
            if (!grad_vols->level[i]) return fail(CNERF_EINVAL, "field_backward16: gradient volume %d is NULL", i);
    if (pass == 1 && !fine_z) return fail(CNERF_EINVAL, "field_backward16: the fine pass needs fine_z");
    const PackedLayout pl = packed_layout(cfg);
    if (pl.n_film && (!freq || !phase)) return fail(CNERF_EINVAL, "field_backward16: FiLM layers need freq and phase");
    hipStream_t stream = (hipStream_t)stream_;
    const long long npi = (long long)cfg->R * cfg->R * cfg->S;

    FieldArgs fa;
    if (int rc = fill_field_args(fa, cfg, vols, (mode & CNERF_B16_CHAIN) ? grad_vols : nullptr, packed, freq, phase, image0)) return rc;
    set_points(fa, n_images, npi);
    fa.cam2world = cam2world + (size_t)image0 * 16;
    if (pass == 0) {
        fa.mode = FIELD_MODE_COARSE;
        fa.u_strat = u_strat ? u_strat + (size_t)image0 * npi : nullptr;
    } else if (pass == 1) {
        fa.mode = FIELD_MODE_FINE;
        fa.fine_z = fine_z + (size_t)image0 * npi;
    } else {
        if (!u_strat) return fail(CNERF_EINVAL, "field_backward16: pass 2 takes the points (B,R*R*S,3) in the u_strat argument");
        fa.mode = FIELD_MODE_POINTS;
        fa.points = u_strat + (size_t)image0 * npi * 3;
    }
    const char* base16 = (const char*)packed16;
    if (mode & CNERF_B16_STORE) {
        if (!act_g16) return fail(CNERF_EINVAL, "field_backward16: the storing re-run uses act_g16 as scratch for its head output");
        FieldArgs fs = fa;
        fs.rgb_sigma = (float*)act_g16;      // (n,4) floats of head output nobody reads; the chain overwrites act_g16 entirely
        fs.z_out = nullptr;
        fs.act_points = (long long)n_images * npi;
        fs.act_feat = (float*)act_feat16;
        fs.act_h = (float*)act_h16;
        fs.act_c = (float*)act_c16;
        fs.act_tb16 = 1;
        if (hipError_t e = launch_field_h3(fs, cfg->H, stream)) return hip_fail(e, "field kernel (fp16 activation store)");
    }
    fa.grad_out = grad_rgb_sigma ? grad_rgb_sigma + (size_t)image0 * npi * 4 : nullptr;
    fa.saved_out = saved_rgb_sigma ? saved_rgb_sigma + (size_t)image0 * npi * 4 : nullptr;
    const float* winv = (const float*)(base16 + l.winv_off);
    if (mode & CNERF_B16_DRY) {
        if (hipError_t e = launch_chain16(fa, cfg->H, base16, base16 + l.head_off, winv, scales, act_c16, nullptr, nullptr, gmax, nullptr, l.n_mats, 1,
                                          group_step < 1 ? 1 : group_step, stream))
            return hip_fail(e, "chain16 (dry run)");
    }
    if (mode & CNERF_B16_CHAIN) {
        // Ray passes: the chain stores its input-tile gradients (fp32, 128 B per point) and scatter_sorted_kernel adds them to the volume
        // pre-reduced per pixel patch (scatter_patch.hip); explicit points are added by the chain itself.  CNERF_SCATTER=chain makes the
        // chain add the ray passes' too, CNERF_SCATTER=coarse sends only the coarse pass through the patch kernel (A/B runs and
        // tests/test_gpu_parity.py::test_sorted_patch_scatter_matches_the_chain_scatter; DESIGN.md 3.7 has the three timings).
        const char* sc_env = getenv("CNERF_SCATTER");
        const bool force_chain = sc_env && !strcmp(sc_env, "chain"), coarse_only = sc_env && !strcmp(sc_env, "coarse");
        const bool patch = gin && pass != 2 && !force_chain && (pass == 0 || !coarse_only);
        fa.gin = patch ? gin : nullptr;
        if (hipError_t e = launch_chain16(fa, cfg->H, base16, base16 + l.head_off, winv, scales, act_c16, act_g16, act_go16, nullptr, sat, l.n_mats, 0, 1,
                                          stream))
            return hip_fail(e, "chain16");
        if (patch)
            if (hipError_t e = launch_scatter_patch(fa, gin, stream)) return hip_fail(e, "scatter_patch");
    }
    return CNERF_OK;
}
}  // namespace

int cnerf_field_backward16(const cnerf_cfg* cfg, uint32_t mode, int32_t group_step, int32_t pass, int32_t image0, int32_t n_images,
                           const cnerf_volumes* vols, const float* packed, const void* packed16, const float* freq,
                           const float* phase, const float* cam2world, const float* u_strat, const float* fine_z,
                           const float* grad_rgb_sigma, const float* saved_rgb_sigma, void* act_feat16, void* act_h16, void* act_c16,
                           void* act_g16, void* act_go16, const float* scales, uint32_t* gmax, const cnerf_grad_volumes* grad_vols,
                           void* stream_) {
    return field_backward16_impl(cfg, mode, group_step, pass, image0, n_images, vols, packed, packed16, freq, phase, cam2world, u_strat, fine_z,
                                 grad_rgb_sigma, saved_rgb_sigma, act_feat16, act_h16, act_c16, act_g16, act_go16, scales, gmax, grad_vols, nullptr,
                                 nullptr, stream_);
}

// ---------------------------------------------------------------------------------------------------------------------
// the whole backward in one call
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// Workspace of cnerf_render_backward, carved in this order (every piece 256-byte aligned).  n = images_per_chunk * points per
// image (fp32 backward: row-major fp32 chunk matrices), T = images_per_chunk * tiles per image (fp16 backward: TB16 blocks).
struct BackwardLayout {
    size_t gc, gf;                                   // d loss / d rgb_sigma of the coarse / fine samples, whole call
    size_t a_feat, a_h, a_c, a_g, a_go;              // chunk buffers (a_feat / a_h / a_c absent when the forward kept its activations)
    size_t a_amax;                                   // per-point FiLM family: (L, T * 32) floats
    size_t a_gy;                                     // per-point FiLM family: L TB16 slabs of g_y
    size_t a_gin;                                    // fp16 chain: (feature input tiles, chunk points, 32) fp32 input-tile gradients for scatter_patch_kernel
    size_t gmax, scales;                             // fp16: sampled maxima (n_mats + 1 uint32), {S, 1/S} pairs (n_mats + 1)
    size_t dwarg, cs, dwh, csh;                      // per-image reductions of one matrix: (cnt, H, 32 * max tiles), (cnt, H), (cnt, 4, H), (cnt, 4)
    size_t total;
    int n_mats, n_in, k0;
};
int backward_layout(const cnerf_cfg* c, int bprec, int cnt, bool have_act16, BackwardLayout& L) {
    if (bprec != CNERF_PREC_FP32 && bprec != CNERF_PREC_FP16) return fail(CNERF_EINVAL, "render_backward: backward_precision must be CNERF_PREC_FP32 or CNERF_PREC_FP16");
    if (c->layer_kind[0] == CNERF_LAYER_PFILM && bprec != CNERF_PREC_FP16)
        return fail(CNERF_ENOSYS, "render_backward: the per-point FiLM family's exact fp32 backward finishes its mapping-MLP gradients with library "
                                  "GEMMs on the host (cnerf_field_backward + cnerf_weight_grad + cnerf_scatter_features); backward_precision fp16 runs here");
    if (cnt < 1 || cnt > c->B) return fail(CNERF_EINVAL, "render_backward: images_per_chunk=%d out of [1,B]", cnt);
    if (c->layer_kind[0] == CNERF_LAYER_PFILM) {      // chain_pw16.hip: three stored derivatives and three gradient slabs per layer, m and g_mpre
        if (have_act16 && cnt != c->B) return fail(CNERF_EINVAL, "render_backward: kept activations need images_per_chunk = B");
        if (c->precision != CNERF_PREC_FP16X3) return fail(CNERF_EINVAL, "render_backward: the fp16 backward re-runs the fp16x3 forward (cfg->precision)");
        const size_t H = c->H, NT = H / 32, Lc = c->L, npi = (size_t)c->R * c->R * c->S, tpi = (npi + 31) / 32;
        const size_t N = (size_t)c->B * npi, T = (size_t)cnt * tpi;
        const bool hier = c->flags & CNERF_F_HIERARCHICAL;
        L.n_in = 2;
        L.k0 = 3;
        L.n_mats = c->L;
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
        L.gc = take(N * 4 * sizeof(float));
        L.gf = take(hier ? N * 4 * sizeof(float) : 0);
        L.a_feat = take(have_act16 ? 0 : T * 2 * 2048);
        L.a_h = take(have_act16 ? 0 : (Lc * NT + 8) * T * 2048);
        L.a_c = take(have_act16 ? 0 : 3 * Lc * NT * T * 2048);
        L.a_amax = take(have_act16 ? 0 : Lc * T * 32 * sizeof(float));
        L.a_g = take((3 * Lc * NT + 8) * T * 2048);
        L.a_go = take(T * 2048);
        L.a_gin = 0;
        L.a_gy = take(Lc * NT * T * 2048);                          // two-kernel chain: g_y slabs
        L.gmax = take((5 * Lc + 2) * sizeof(uint32_t));             // 3 L + 2 sampled maxima, L of g_y, L of the stored derivatives
        L.scales = take((2 * (4 * Lc + 2) + 2 * Lc) * sizeof(float));   // {S, 1 / S} x (4 L + 2), then per layer {r, To}
        L.dwarg = take((size_t)cnt * 256 * 256 * sizeof(float));
        L.cs = take((size_t)cnt * 256 * sizeof(float));
        L.dwh = take((size_t)cnt * 4 * H * sizeof(float));
        L.csh = take((size_t)cnt * 4 * sizeof(float));
        L.total = off;
        return CNERF_OK;
    }
    if (have_act16 && (bprec != CNERF_PREC_FP16 || cnt != c->B)) return fail(CNERF_EINVAL, "render_backward: kept activations need the fp16 backward and images_per_chunk = B");
    if (bprec == CNERF_PREC_FP16 && c->precision != CNERF_PREC_FP16X3) return fail(CNERF_EINVAL, "render_backward: the fp16 backward re-runs the fp16x3 forward (cfg->precision)");
    const PackedLayout pl = packed_layout(c);
    L.n_in = pl.n_in;
    L.k0 = pl.k0;
    L.n_mats = 0;
    for (int l = 0; l < c->L; ++l) L.n_mats += c->layer_kind[l] == CNERF_LAYER_RES ? 2 : 1;
    const size_t H = c->H, NT = H / 32, npi = (size_t)c->R * c->R * c->S, tpi = (npi + 31) / 32;
    const size_t N = (size_t)c->B * npi, n = (size_t)cnt * npi, T = (size_t)cnt * tpi;
    const bool hier = c->flags & CNERF_F_HIERARCHICAL;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    L.gc = take(N * 4 * sizeof(float));
    L.gf = take(hier ? N * 4 * sizeof(float) : 0);
    if (bprec == CNERF_PREC_FP16) {
        L.a_feat = take(have_act16 ? 0 : T * L.n_in * 2048);
        L.a_h = take(have_act16 ? 0 : (size_t)L.n_mats * T * NT * 2048);
        L.a_c = take(have_act16 ? 0 : (size_t)L.n_mats * T * NT * 2048);
        L.a_g = take((size_t)L.n_mats * T * NT * 2048);
        L.a_go = take(T * 2048);
        L.a_gin = take((size_t)L.n_in * n * 32 * sizeof(float));
    } else {
        L.a_gin = 0;
        L.a_feat = take(n * 32 * L.n_in * sizeof(float));
        L.a_h = take((size_t)L.n_mats * n * H * sizeof(float));
        L.a_c = take((size_t)L.n_mats * n * H * sizeof(float));
        L.a_g = take((size_t)L.n_mats * n * H * sizeof(float));
        L.a_go = take(n * 4 * sizeof(float));
    }
    L.a_amax = 0;
    L.a_gy = 0;
    L.gmax = take((size_t)(L.n_mats + 2) * sizeof(uint32_t));
    L.scales = take((size_t)2 * (L.n_mats + 1) * sizeof(float));
    const size_t kmax = 32 * (size_t)(L.n_in > (int)NT ? L.n_in : (int)NT);
    L.dwarg = take((size_t)cnt * H * kmax * sizeof(float));
    L.cs = take((size_t)cnt * H * sizeof(float));
    L.dwh = take((size_t)cnt * 4 * H * sizeof(float));
    L.csh = take((size_t)cnt * 4 * sizeof(float));
    L.total = off;
    return CNERF_OK;
}
}  // namespace

int cnerf_backward_workspace_bytes(const cnerf_cfg* cfg, int32_t backward_precision, int32_t images_per_chunk, int32_t have_act16, size_t* bytes) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, true)) return rc;
    BackwardLayout L;
    if (int rc = backward_layout(cfg, backward_precision, images_per_chunk, have_act16 != 0, L)) return rc;
    if (bytes) *bytes = L.total;
    return CNERF_OK;
}

int cnerf_render_backward(const cnerf_cfg* cfg, int32_t bprec, int32_t cnt_max, const cnerf_volumes* vols, const cnerf_field_params* P,
                          const float* packed, const void* packed_bwd, const float* freq, const float* phase, const float* cam2world,
                          const cnerf_rng* rng, const cnerf_saved* saved, const cnerf_aux* kept, const float* grad_pixels,
                          const float* grad_depth, const cnerf_field_param_grads* G, float* grad_freq, float* grad_phase,
                          const cnerf_grad_volumes* grad_vols, uint32_t* saturated, void* workspace, void* stream_) {
    g_err[0] = 0;
    if (int rc = check_cfg(cfg, true)) return rc;
    const bool have_act16 = kept && kept->act16[0].h;
    BackwardLayout L;
    if (int rc = backward_layout(cfg, bprec, cnt_max, have_act16, L)) return rc;
    if (!vols || !P || !packed || !packed_bwd || !cam2world || !saved || !grad_pixels || !G || !grad_vols || !workspace)
        return fail(CNERF_EINVAL, "render_backward: NULL argument");
    const bool hier = cfg->flags & CNERF_F_HIERARCHICAL;
    if (!saved->coarse_rgb_sigma || !saved->coarse_z || (hier && (!saved->fine_rgb_sigma || !saved->fine_z)))
        return fail(CNERF_EINVAL, "render_backward: saved rgb_sigma / z of the forward are incomplete");
    const PackedLayout pl = packed_layout(cfg);
    if (pl.n_film && (!freq || !phase || !grad_freq || !grad_phase)) return fail(CNERF_EINVAL, "render_backward: FiLM layers need freq, phase and their gradient buffers");
    static const cnerf_rng no_rng = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!rng) rng = &no_rng;
    hipStream_t stream = (hipStream_t)stream_;
    const int H = cfg->H, NT = H / 32, B = cfg->B;
    const long long npi = (long long)cfg->R * cfg->R * cfg->S, tpi = (npi + 31) / 32;
    char* ws = (char*)workspace;
    float* gc = (float*)(ws + L.gc);
    float* gf = hier ? (float*)(ws + L.gf) : nullptr;

    // 1. d(pixels, depth) -> d(rgb_sigma) of every coarse / fine sample
    if (int rc = cnerf_merge_composite_backward(cfg, saved->coarse_rgb_sigma, saved->coarse_z, saved->fine_rgb_sigma, saved->fine_z,
                                                cfg->noise_std != 0.0f ? rng->eps_final : nullptr, grad_pixels, grad_depth, gc, gf, stream_))
        return rc;

    if (cfg->layer_kind[0] == CNERF_LAYER_PFILM) {
        // ---- per-point FiLM family, half-precision backward: storing forward (field_pw16.hip), the two chain kernels with their dry runs
        // (chain_pw16.hip), one weight_grad16 reduction per matrix: dW_l = g_pre_l^T y_{l-1}, dWm2 rows = (g_fr_l | g_ph_l)^T m, dWm1 = g_mpre^T feat, head
        const int Lc = cfg->L, n_slots = 3 * Lc + 2;
        if (!P->map_w1 || !P->map_w2 || !P->w_final) return fail(CNERF_EINVAL, "render_backward: mapping network / head parameters are NULL");
        if (!G->map_w1 || !G->map_b1 || !G->map_w2 || !G->map_b2 || !G->w_final || !G->b_final)
            return fail(CNERF_EINVAL, "render_backward: per-point FiLM needs the mapping network's and the head's gradient buffers");
        for (int l = 0; l < Lc; ++l)
            if (!G->w[l] || !G->b[l]) return fail(CNERF_EINVAL, "render_backward: gradient buffers of layer %d are NULL", l);
        const PwChainLayout cl = pw_chain_layout(cfg);
        const char* base16 = (const char*)packed_bwd;
        float* dwarg = (float*)(ws + L.dwarg);
        float* cs = (float*)(ws + L.cs);
        float* dwh = (float*)(ws + L.dwh);
        float* csh = (float*)(ws + L.csh);
        uint32_t* gmax = (uint32_t*)(ws + L.gmax);
        float* scales = (float*)(ws + L.scales);
        char* a_g = ws + L.a_g;
        char* a_go = ws + L.a_go;
        if (hipError_t e = hipMemsetAsync(a_go, 0, (size_t)cnt_max * tpi * 2048, stream)) return hip_fail(e, "memset");
        for (int pass = 0; pass < (hier ? 2 : 1); ++pass) {
            const float* g_out = pass ? gf : gc;
            const float* s_out = pass ? saved->fine_rgb_sigma : saved->coarse_rgb_sigma;
            for (int b0 = 0; b0 < B; b0 += cnt_max) {
                const int cnt = b0 + cnt_max <= B ? cnt_max : B - b0;
                const long long T = (long long)cnt * tpi;
                // the pass's activations: kept by the forward (all images), or re-computed into the workspace
                char* a_feat = have_act16 ? (char*)kept->act16[pass].feat : ws + L.a_feat;
                char* a_h = have_act16 ? (char*)kept->act16[pass].h : ws + L.a_h;
                char* a_c = have_act16 ? (char*)kept->act16[pass].c : ws + L.a_c;
                float* a_amax = have_act16 ? (float*)kept->act16[pass].amax : (float*)(ws + L.a_amax);
                if (have_act16 && (!a_feat || !a_h || !a_c || !a_amax)) return fail(CNERF_EINVAL, "render_backward: act16 of pass %d is incomplete", pass);
                FieldArgs fa;
                if (int rc = fill_field_args(fa, cfg, vols, grad_vols, packed, nullptr, nullptr, b0)) return rc;
                if (!fa.lvl_grad[0]) return fail(CNERF_EINVAL, "render_backward: gradient volume is NULL");
                set_points(fa, cnt, npi);
                fa.cam2world = cam2world + (size_t)b0 * 16;
                if (pass == 0) {
                    fa.mode = FIELD_MODE_COARSE;
                    fa.u_strat = rng->u_strat ? rng->u_strat + (size_t)b0 * npi : nullptr;
                } else {
                    fa.mode = FIELD_MODE_FINE;
                    fa.fine_z = saved->fine_z + (size_t)b0 * npi;
                }
                if (!have_act16) {
                    FieldArgs fs = fa;
                    fs.rgb_sigma = (float*)a_g;      // (n,4) floats of head output nobody reads; the chain overwrites a_g entirely
                    fs.z_out = nullptr;
                    fs.act_points = (long long)cnt * npi;
                    fs.act_feat = (float*)a_feat;
                    fs.act_h = (float*)a_h;
                    fs.act_c = (float*)a_c;
                    fs.act_amax = a_amax;
                    fs.act_tb16 = 1;
                    if (hipError_t e = launch_field_pw3(fs, H, stream)) return hip_fail(e, "field kernel (fp16 activation store)");
                }
                fa.grad_out = g_out + (size_t)b0 * npi * 4;
                fa.saved_out = s_out + (size_t)b0 * npi * 4;
                if (hipError_t e = hipMemsetAsync(gmax, 0, (size_t)(5 * Lc + 2) * sizeof(uint32_t), stream)) return hip_fail(e, "memset");
                if (hipError_t e = launch_absmax_bits(fa.grad_out, (long long)cnt * npi * 4, gmax + n_slots - 1, stream)) return hip_fail(e, "absmax");
                if (hipError_t e = launch_pow2_scales(gmax + n_slots - 1, 1, scales + 2 * (n_slots - 1), stream)) return hip_fail(e, "pow2_scales");
                for (int m = 0; m < n_slots - 1; ++m)
                    if (hipError_t e = launch_fill(scales + 2 * m, 1.0f, 2, stream)) return hip_fail(e, "fill");
                const size_t slabH = (size_t)T * NT * 2048;            // bytes per (tiles, NT, 32, 32) slab
                float* lay = scales + 2 * (4 * Lc + 2);
                PwChainBuffers cb{base16, base16 + cl.m_off, base16 + cl.head_off, (const float*)(base16 + cl.winv_off), (const float*)(base16 + cl.anorm_off),
                                  scales, lay, a_c, a_amax, a_h + (size_t)Lc * slabH, ws + L.a_gy, a_g, a_go, gmax, nullptr};
                const long long groups = (long long)cnt * ((tpi + 3) / 4);
                long long step = groups / 1024;                       // dry-run sampling: at least 1024 tile groups (131 k points), every 32nd at most
                step = step < 1 ? 1 : (step > 32 ? 32 : step);
                // g_y through the layer matrices (dry run -> its scales), then the stored slabs and the mapping products (dry run -> g_mpre's scale)
                for (int m = 0; m < Lc; ++m)
                    if (hipError_t e = launch_fill(scales + 2 * (3 * Lc + 2 + m), 1.0f, 2, stream)) return hip_fail(e, "fill");
                if (hipError_t e = launch_chain_pre(fa, H, cb, 1, (int)step, stream)) return hip_fail(e, "chain_pre (dry run)");
                if (hipError_t e = launch_pow2_scales(gmax + 3 * Lc + 2, Lc, scales + 2 * (3 * Lc + 2), stream)) return hip_fail(e, "pow2_scales");
                for (int m = 0; m < Lc; ++m)      // the largest stored derivative of each layer over the chunk's points
                    if (hipError_t e = launch_absmax_bits(a_amax + (size_t)m * T * 32, T * 32, gmax + 4 * Lc + 2 + m, stream)) return hip_fail(e, "absmax");
                if (hipError_t e = launch_pw_split_scales(gmax + 4 * Lc + 2, Lc, scales, lay, stream)) return hip_fail(e, "split_scales");
                cb.sat = saturated;
                if (hipError_t e = launch_chain_pre(fa, H, cb, 0, 1, stream)) return hip_fail(e, "chain_pre");
                cb.sat = nullptr;
                if (hipError_t e = launch_pw_gm(fa, H, cb, 1, (int)step, stream)) return hip_fail(e, "pw_gm (dry run)");
                if (hipError_t e = launch_pow2_scales(gmax + 3 * Lc, 1, scales + 2 * (3 * Lc), stream)) return hip_fail(e, "pow2_scales");
                cb.sat = saturated;
                if (hipError_t e = launch_pw_gm(fa, H, cb, 0, 1, stream)) return hip_fail(e, "pw_gm");
                // one reduction: G (n_rows of slab `slot`) against X (x_ct channel tiles), take k_real columns from column k0 on
                auto reduce = [&](const void* Gs, int g_ct, int n_rows, int slot, const void* X, int x_ct, int k0, int k_real, float* dW, float* db) -> int {
                    if (hipError_t e = hipMemsetAsync(dwarg, 0, (size_t)cnt * n_rows * 32 * x_ct * sizeof(float), stream)) return hip_fail(e, "memset");
                    if (hipError_t e = hipMemsetAsync(cs, 0, (size_t)cnt * n_rows * sizeof(float), stream)) return hip_fail(e, "memset");
                    if (int rc = cnerf_weight_grad16(cnt, tpi, n_rows, g_ct, x_ct, Gs, X, dwarg, cs, scales + 2 * slot + 1, stream_)) return rc;
                    if (hipError_t e = launch_param_reduce(cnt, n_rows, 32 * x_ct, k_real, dwarg + k0, cs, nullptr, 0, nullptr, nullptr, dW, db, nullptr, nullptr, stream))
                        return hip_fail(e, "param_reduce");
                    return CNERF_OK;
                };
                const char* m16 = a_h + (size_t)Lc * slabH;
                const size_t LH = (size_t)Lc * H;
                for (int l = 0; l < Lc; ++l) {
                    const char* g_pre = a_g + (size_t)(3 * l) * slabH;
                    if (l == 0) {      // X = [feature | position]: the position's three columns
                        if (int rc = reduce(g_pre, NT, H, 3 * l, a_feat, 2, 32, 3, G->w[0], G->b[0])) return rc;
                    } else {
                        if (int rc = reduce(g_pre, NT, H, 3 * l, a_h + (size_t)(l - 1) * slabH, NT, 0, H, G->w[l], G->b[l])) return rc;
                    }
                    if (int rc = reduce(g_pre + slabH, NT, H, 3 * l + 1, m16, 8, 0, 256, G->map_w2 + (size_t)l * H * 256, G->map_b2 + (size_t)l * H)) return rc;
                    if (int rc = reduce(g_pre + 2 * slabH, NT, H, 3 * l + 2, m16, 8, 0, 256, G->map_w2 + (LH + (size_t)l * H) * 256, G->map_b2 + LH + (size_t)l * H))
                        return rc;
                }
                if (int rc = reduce(a_g + (size_t)(3 * Lc) * slabH, 8, 256, 3 * Lc, a_feat, 2, 0, 32, G->map_w1, G->map_b1)) return rc;
                if (hipError_t e = hipMemsetAsync(dwh, 0, (size_t)cnt * 4 * H * sizeof(float), stream)) return hip_fail(e, "memset");
                if (hipError_t e = hipMemsetAsync(csh, 0, (size_t)cnt * 4 * sizeof(float), stream)) return hip_fail(e, "memset");
                if (int rc = cnerf_weight_grad16(cnt, tpi, 4, 1, NT, a_go, a_h + (size_t)(Lc - 1) * slabH, dwh, csh, scales + 2 * (n_slots - 1) + 1, stream_)) return rc;
                if (hipError_t e = launch_param_reduce(cnt, 4, H, H, dwh, csh, nullptr, 0, nullptr, nullptr, G->w_final, G->b_final, nullptr, nullptr, stream))
                    return hip_fail(e, "param_reduce (head)");
            }
        }
        return CNERF_OK;
    }

    // the matrices in slab order (a residual block: fc1, fc2), their kinds and gradient buffers
    const float *Wm[2 * CNERF_MAX_LAYERS], *bm[2 * CNERF_MAX_LAYERS];
    float *dWm[2 * CNERF_MAX_LAYERS], *dbm[2 * CNERF_MAX_LAYERS];
    int film_of[2 * CNERF_MAX_LAYERS];
    int nm = 0, nfilm = 0;
    for (int l = 0; l < cfg->L; ++l) {
        const bool res = cfg->layer_kind[l] == CNERF_LAYER_RES;
        if (!P->w[l] || !P->b[l] || (res && (!P->w2[l] || !P->b2[l]))) return fail(CNERF_EINVAL, "render_backward: parameters of layer %d are NULL", l);
        Wm[nm] = P->w[l]; bm[nm] = P->b[l]; dWm[nm] = G->w[l]; dbm[nm] = G->b[l];
        film_of[nm] = cfg->layer_kind[l] == CNERF_LAYER_FILM ? nfilm++ : -1;
        ++nm;
        if (res) {
            Wm[nm] = P->w2[l]; bm[nm] = P->b2[l]; dWm[nm] = G->w2[l]; dbm[nm] = G->b2[l];
            film_of[nm] = -1;
            ++nm;
        }
    }
    const int film_stride = pl.n_film * H;
    float* dwarg = (float*)(ws + L.dwarg);
    float* cs = (float*)(ws + L.cs);
    float* dwh = (float*)(ws + L.dwh);
    float* csh = (float*)(ws + L.csh);
    uint32_t* gmax = (uint32_t*)(ws + L.gmax);
    float* scales = (float*)(ws + L.scales);
    void* a_g = ws + L.a_g;
    void* a_go = ws + L.a_go;
    if (bprec == CNERF_PREC_FP16)        // only channels 0..3 of a row are ever written: the rest must read as zero
        if (hipError_t e = hipMemsetAsync(a_go, 0, (size_t)cnt_max * tpi * 2048, stream)) return hip_fail(e, "memset");

    // reduce one matrix' per-image sums into the parameter gradients (and dfreq / dphase of the chunk's images)
    auto reduce_matrix = [&](int m, int cnt, int b0, int ld) -> int {
        const int k_real = m == 0 ? L.k0 : H;
        const bool film = film_of[m] >= 0;
        const size_t foff = film ? (size_t)b0 * film_stride + (size_t)film_of[m] * H : 0;
        if (hipError_t e = launch_param_reduce(cnt, H, ld, k_real, dwarg, cs, film ? freq + foff : nullptr, film_stride, Wm[m], bm[m], dWm[m], dbm[m],
                                               film ? grad_freq + foff : nullptr, film ? grad_phase + foff : nullptr, stream))
            return hip_fail(e, "param_reduce");
        return CNERF_OK;
    };

    for (int pass = 0; pass < (hier ? 2 : 1); ++pass) {
        const float* g_out = pass ? gf : gc;
        const float* s_out = pass ? saved->fine_rgb_sigma : saved->coarse_rgb_sigma;
        const uint8_t* drop = pass ? rng->drop_fine : rng->drop_coarse;
        for (int b0 = 0; b0 < B; b0 += cnt_max) {
            const int cnt = b0 + cnt_max <= B ? cnt_max : B - b0;
            if (bprec == CNERF_PREC_FP16) {
                const long long T = (long long)cnt * tpi;
                void* a_feat = have_act16 ? kept->act16[pass].feat : (void*)(ws + L.a_feat);
                void* a_h = have_act16 ? kept->act16[pass].h : (void*)(ws + L.a_h);
                void* a_c = have_act16 ? kept->act16[pass].c : (void*)(ws + L.a_c);
                if (have_act16 && (!a_feat || !a_h || !a_c)) return fail(CNERF_EINVAL, "render_backward: act16 of pass %d is incomplete", pass);
                // scales: ones for the matrices during the dry run, the head gradient's from max |d loss / d rgb_sigma| of the chunk
                if (hipError_t e = hipMemsetAsync(gmax, 0, (size_t)(L.n_mats + 2) * sizeof(uint32_t), stream)) return hip_fail(e, "memset");
                if (hipError_t e = launch_absmax_bits(g_out + (size_t)b0 * npi * 4, (long long)cnt * npi * 4, gmax + L.n_mats + 1, stream)) return hip_fail(e, "absmax");
                if (hipError_t e = launch_pow2_scales(gmax + L.n_mats + 1, 1, scales + 2 * L.n_mats, stream)) return hip_fail(e, "pow2_scales");
                for (int m = 0; m < L.n_mats; ++m)
                    if (hipError_t e = launch_fill(scales + 2 * m, 1.0f, 2, stream)) return hip_fail(e, "fill");
                const long long groups = (long long)cnt * ((tpi + 3) / 4);
                long long step = groups / 2048;                       // dry-run sampling: every 16th tile group once there are plenty
                step = step < 1 ? 1 : (step > 16 ? 16 : step);
                if (int rc = field_backward16_impl(cfg, (have_act16 ? 0u : CNERF_B16_STORE) | CNERF_B16_DRY, (int)step, pass, b0, cnt, vols, packed, packed_bwd,
                                                   freq, phase, cam2world, rng->u_strat, saved->fine_z, g_out, s_out, a_feat, a_h, a_c, a_g, a_go, scales, gmax,
                                                   grad_vols, nullptr, nullptr, stream_))
                    return rc;
                if (hipError_t e = launch_pow2_scales(gmax, L.n_mats, scales, stream)) return hip_fail(e, "pow2_scales");
                if (int rc = field_backward16_impl(cfg, CNERF_B16_CHAIN, 1, pass, b0, cnt, vols, packed, packed_bwd, freq, phase, cam2world, rng->u_strat,
                                                   saved->fine_z, g_out, s_out, a_feat, a_h, a_c, a_g, a_go, scales, gmax, grad_vols, saturated, (float*)(ws + L.a_gin), stream_))
                    return rc;
                const size_t slab = (size_t)T * NT * 2048;             // bytes per matrix in a_h / a_g
                for (int m = 0; m < L.n_mats; ++m) {
                    const int x_ct = m == 0 ? L.n_in : NT;
                    const void* X = m == 0 ? a_feat : (const void*)((const char*)a_h + (size_t)(m - 1) * slab);
                    if (hipError_t e = hipMemsetAsync(dwarg, 0, (size_t)cnt * H * 32 * x_ct * sizeof(float), stream)) return hip_fail(e, "memset");
                    if (hipError_t e = hipMemsetAsync(cs, 0, (size_t)cnt * H * sizeof(float), stream)) return hip_fail(e, "memset");
                    if (int rc = cnerf_weight_grad16(cnt, tpi, H, NT, x_ct, (const char*)a_g + (size_t)m * slab, X, dwarg, cs, scales + 2 * m + 1, stream_)) return rc;
                    if (int rc = reduce_matrix(m, cnt, b0, 32 * x_ct)) return rc;
                }
                if (hipError_t e = hipMemsetAsync(dwh, 0, (size_t)cnt * 4 * H * sizeof(float), stream)) return hip_fail(e, "memset");
                if (hipError_t e = hipMemsetAsync(csh, 0, (size_t)cnt * 4 * sizeof(float), stream)) return hip_fail(e, "memset");
                if (int rc = cnerf_weight_grad16(cnt, tpi, 4, 1, NT, a_go, (const char*)a_h + (size_t)(L.n_mats - 1) * slab, dwh, csh, scales + 2 * L.n_mats + 1, stream_))
                    return rc;
                if (hipError_t e = launch_param_reduce(cnt, 4, H, H, dwh, csh, nullptr, 0, nullptr, nullptr, G->w_final, G->b_final, nullptr, nullptr, stream))
                    return hip_fail(e, "param_reduce (head)");
            } else {
                const size_t n = (size_t)cnt * npi;
                float* a_feat = (float*)(ws + L.a_feat);
                float* a_h = (float*)(ws + L.a_h);
                float* a_c = (float*)(ws + L.a_c);
                if (int rc = cnerf_field_backward(cfg, pass, b0, cnt, vols, packed, (const float*)packed_bwd, freq, phase, cam2world, rng->u_strat, saved->fine_z,
                                                  g_out, s_out, a_feat, a_h, a_c, (float*)a_g, (float*)a_go, grad_vols, drop, stream_))
                    return rc;
                for (int m = 0; m < L.n_mats; ++m) {
                    const int K = m == 0 ? 32 * L.n_in : H;
                    const float* X = m == 0 ? a_feat : a_h + (size_t)(m - 1) * n * H;
                    if (hipError_t e = hipMemsetAsync(dwarg, 0, (size_t)cnt * H * K * sizeof(float), stream)) return hip_fail(e, "memset");
                    if (hipError_t e = hipMemsetAsync(cs, 0, (size_t)cnt * H * sizeof(float), stream)) return hip_fail(e, "memset");
                    if (int rc = cnerf_weight_grad(cnt, npi, H, K, (const float*)a_g + (size_t)m * n * H, X, dwarg, cs, stream_)) return rc;
                    if (int rc = reduce_matrix(m, cnt, b0, K)) return rc;
                }
                if (G->w_final && G->b_final)
                    if (hipError_t e = launch_head_grad32((const float*)a_go, a_h + (size_t)(L.n_mats - 1) * n * H, (long long)n, H, G->w_final, G->b_final, stream))
                        return hip_fail(e, "head_grad32");
            }
        }
    }
    return CNERF_OK;
}

}  // extern "C"
