// Internal interface between the C-ABI translation unit and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cnerf.h"
#include "cnerf_dev.hpp"

namespace cnerf {

enum { FIELD_MODE_POINTS = 0, FIELD_MODE_COARSE = 1, FIELD_MODE_FINE = 2 };

// Arguments of field_tile_kernel (passed by value).
struct FieldArgs {
    // layer-0 inputs: n_in tiles of 32 channels; in_level[tk] = pyramid level (or -1: the xyz tile), in_chan[tk] = first
    // channel inside that level.  lvl_vol: (B,V,V,V,C) channel-last; lvl_grad: same shape, backward only.
    int n_in;
    int in_level[8];
    int in_chan[8];
    const float* lvl_vol[CNERF_MAX_LEVELS];
    float* lvl_grad[CNERF_MAX_LEVELS];
    int lvl_V[CNERF_MAX_LEVELS];
    int lvl_C[CNERF_MAX_LEVELS];
    const float* packed;     // packed weights (float4 stream, see field_kernel.hip)
    const float* bias;       // biases of all layers then the head, inside the packed buffer
    const float* freq;       // (B, film_stride) or null
    const float* phase;
    const float* points;     // mode POINTS: (B,n,3)
    const float* cam2world;  // (B,16)
    const float* u_strat;    // mode COARSE: (B,n) or null
    const float* fine_z;     // mode FINE: (B,n)
    float* rgb_sigma;        // (B,n,4)
    float* z_out;            // mode COARSE: (B,n)
    float* points_out;       // optional (B,n,3)
    unsigned long long* stamps;  // diagnostic builds (-DCNERF_STAMPS) only: 8 cycle totals; else unused
    // activation store of the backward pass (all null in a plain forward): row-major [point][channel]
    long long act_points;    // rows of every activation buffer (= B * n_per_image of the chunk)
    float* act_feat;         // (n,32*n_in) layer-0 input tiles (looked-up features, xyz)
    float* act_h;            // (L,n,H) layer outputs sin(arg)
    float* act_c;            // (L,n,H) cos(arg)
    int act_tb16;            // 1: act_feat / act_h / act_c are fp16 buffers in the TB16 layout (bwd16.hpp), written by field_h3_kernel
    float* act_amax;         // per-point FiLM family, act_tb16: (L, tiles * 32) largest |stored derivative| per layer and point (field_pw16.hip)
    // field_backward_kernel only
    const float* packed_t;   // transposed packed weights (cnerf_pack_field_transposed)
    const float* grad_out;   // (n,4) d loss / d rgb_sigma
    const float* saved_out;  // (n,4) rgb_sigma of the forward (sigmoid')
    float* act_g;            // (L,n,H) d loss / d arg
    float* act_go;           // (n,4)   d loss / d head pre-activation
    float* gin;              // chain16_kernel, ray passes: (feature input tiles, points of the launch, 32) fp32 gradients of the looked-up features,
                             // scattered by scatter_patch_kernel; null: the chain scatters its tiles itself (explicit points)
    long long n_per_image;
    long long tiles_per_image;
    long long total_tiles;
    RayGeom geom;
    float half_voxel;
    int L;
    int n_mats;              // weight matrices before the head (a residual block counts two)
    int film_stride;
    int bias_floats;         // biases of all layers + head (padded to 4): ones[H], zeros[H] follow
    uint32_t flags;
    int mode;
    int layer_kind[CNERF_MAX_LAYERS];
    PhiloxKey philox;        // in-kernel draw of u_strat where the tensor is null (cnerf_cfg.philox_*)
    int image0;              // first image of this launch inside the call (global element indices of the draws)
    // dropout behind the sine of the FiLM / sine / per-point FiLM layers (training mode, cnerf_cfg.drop_p); drop_scale 0 = off
    float drop_scale;        // 1 / (1 - p)
    uint32_t drop_thresh;    // Philox word >= thresh keeps (thresh = round(p * 2^32))
    uint32_t drop_stream;    // Philox stream of this pass (PHILOX_DROP_*)
    int n_drop;              // dropout layers of the network (= layers that are not residual blocks)
    long long drop_points;   // points of the whole call (B * n_per_image): row count of one layer of drop_mask
    const uint8_t* drop_mask;  // injected keep decisions (n_drop, drop_points, H), 1 = keep; null: Philox
    // folded FiLM constants of the call (field_tile_kernel FOLD: all-FiLM networks, plain fp32 forward), or null
    const float* fold;       // [Mh | Ml | K | Kb], each (fold_images, n_mats, H)
    int fold_images;
    const float* packed_img; // WFOLD: per image, the layer part of the packed weights with rows scaled by Mh (scale_packed_kernel), or null
    long long packed_img_stride;
};

hipError_t launch_fill(float* dst, float value, int n, hipStream_t stream);
hipError_t launch_pack_matrix(const float* w, int n_out, int K_real, int OT, float* dst, hipStream_t stream);
hipError_t launch_fold_film(const FieldArgs& a, int B, int H, float* out, hipStream_t stream);
hipError_t launch_scale_packed(const FieldArgs& a, int B, int H, const float* mh, long long layer_floats, float* out, hipStream_t stream);
hipError_t launch_field(const FieldArgs& a, int H, hipStream_t stream);
hipError_t launch_field_h3(const FieldArgs& a, int H, hipStream_t stream);
hipError_t launch_field_h1(const FieldArgs& a, int H, hipStream_t stream);      // field_h3.hip compiled with CNERF_H3_PARTS=1
// per-image weight folding of the fp16 kernels (field_h3.hip): img (B, img_elems fp16), fold (B, n_mats * (H + 1)), rowf (B, n_mats, H) scratch
hipError_t launch_fold_h3(const FieldArgs& a, int B, int H, void* img, float* fold, float* rowf, long long img_elems, hipStream_t stream);
hipError_t launch_fold_h1(const FieldArgs& a, int B, int H, void* img, float* fold, float* rowf, long long img_elems, hipStream_t stream);
// t_stride: fragment pairs (one per (output tile, k-chunk)) between successive output tiles in dst; 0 = dense
hipError_t launch_pack_h1(const float* w, int n_out, int K_real, int OT, bool k_outer, void* dst, float* inv_scale_slot, float* wmax_slot,
                          hipStream_t stream, long long t_stride = 0);
hipError_t launch_pack_h3(const float* w, int n_out, int K_real, int OT, bool k_outer, void* dst, float* inv_scale_slot, float* wmax_slot,
                          hipStream_t stream, long long t_stride = 0);
// field_pw16.hip (compiled twice like field_h3.hip): the per-point FiLM family (TALLSIREN) on the fp16 matrix pipe
hipError_t launch_field_pw3(const FieldArgs& a, int H, hipStream_t stream);
hipError_t launch_field_pw1(const FieldArgs& a, int H, hipStream_t stream);
// starting values of the accumulators (biases in accumulator units) and the epilogue's scalars from the raw biases and the 1 / S slots
hipError_t launch_pw16_consts(const cnerf_field_params* p, int L, int H, const float* inv_s, float* consts, hipStream_t stream);
hipError_t launch_field_backward(const FieldArgs& a, int H, hipStream_t stream);
hipError_t launch_pack_head_t(const float* w, int H, float* dst, hipStream_t stream);
hipError_t launch_pack_head(const float* w, int H, float* dst, hipStream_t stream);
hipError_t launch_pack_matrix_t(const float* w, int n_rows_w, int n_cols_w, int OT, float* dst, hipStream_t stream);

// ray_kernels.hip
struct CompositeArgs {
    const float* rgb_sigma;  // (rays,n,4)
    const float* z;          // (rays,n)
    const float* eps;        // (rays,n) or null
    float* rgb;              // (rays,3) or null
    float* dist;             // (rays) or null
    float* weights;          // (rays,n) or null
    long long rays;
    int n;
    float noise_std;
    uint32_t flags;
};
hipError_t launch_composite(const CompositeArgs& a, hipStream_t stream);

struct ResampleArgs {
    const float* z;        // (rays,S)
    const float* weights;  // (rays,S) or null when rgb_sigma is given
    const float* rgb_sigma;  // (rays,S,4): if non-null the coarse weights are composited here first
    const float* eps;      // (rays,S) or null
    const float* u;        // (rays,S)
    float* fine_z;         // (rays,S)
    int32_t* inds;         // optional
    float* cdf;            // optional (rays,S-1)
    float* weights_out;    // optional (rays,S)
    long long rays;
    int S;
    float noise_std;
    uint32_t flags;
    PhiloxKey philox;      // draws eps (coarse) / u (fine) in the kernel where the tensors are null
};
hipError_t launch_resample(const ResampleArgs& a, hipStream_t stream);

struct MergeArgs {
    const float* coarse_rgb_sigma;  // (rays,S,4)
    const float* coarse_z;          // (rays,S)
    const float* fine_rgb_sigma;    // (rays,S,4) or null (non-hierarchical)
    const float* fine_z;            // (rays,S)   or null
    const float* eps;               // (rays,n) or null, n = 2S or S
    float* pixels;                  // (B,3,R,R)
    float* depth;                   // (B,R,R)
    int32_t* sort_idx;              // optional (rays,2S)
    float* final_weights;           // optional (rays,n)
    long long rays;
    int S;
    RayGeom geom;
    float noise_std;
    uint32_t flags;
    PhiloxKey philox;               // draws eps (final) in the kernel where the tensor is null
};
hipError_t launch_merge_composite(const MergeArgs& a, hipStream_t stream);

struct MergeBwdArgs {
    const float* coarse_rgb_sigma;  // (rays,S,4) saved by the forward
    const float* coarse_z;          // (rays,S)
    const float* fine_rgb_sigma;    // (rays,S,4) or null
    const float* fine_z;            // (rays,S)   or null
    const float* eps;               // (rays,n) or null
    const float* grad_pixels;       // (B,3,R,R)
    const float* grad_depth;        // (B,R,R) or null
    float* grad_coarse;             // (rays,S,4)
    float* grad_fine;               // (rays,S,4) or null
    long long rays;
    int S;
    RayGeom geom;
    float noise_std;
    uint32_t flags;
    PhiloxKey philox;
};
hipError_t launch_merge_composite_backward(const MergeBwdArgs& a, hipStream_t stream);

hipError_t launch_philox_fill(const PhiloxKey& k, uint32_t stream_id, long long n, int normal, float* out, hipStream_t stream);
hipError_t launch_transpose_cl(int B, int C, int V, const float* src, float* dst, bool to_channel_last, hipStream_t stream);

struct GatherArgs {
    const float* fvol;    // (B,V,V,V,C)
    const float* points;  // (B,n,3)
    float* feat;          // (B,n,C)
    long long n_per_image;
    int B, V, C;
    float half_voxel;
};
hipError_t launch_gather(const GatherArgs& a, hipStream_t stream);
// scatter_patch.hip: feature-volume gradient of a ray pass from the chain's stored input-tile gradients, pre-reduced per pixel patch x depth bin in LDS
hipError_t launch_scatter_patch(const FieldArgs& f, const float* gin, hipStream_t stream);
hipError_t launch_scatter(const GatherArgs& a, const float* grad_feat, float* grad_fvol, hipStream_t stream);

hipError_t launch_weight_grad(int cnt, long long npi, int H, int K, const float* G, const float* X, float* dW, float* colsum,
                              hipStream_t stream);
hipError_t launch_param_reduce(int cnt, int rows, int ld, int k_real, const float* dWarg, const float* cs, const float* freq, int film_stride,
                               const float* W, const float* bias, float* dW, float* db, float* g_freq, float* g_phase, hipStream_t stream);
hipError_t launch_head_grad32(const float* go, const float* x, long long n, int H, float* dW, float* db, hipStream_t stream);
hipError_t launch_absmax_bits(const float* v, long long n, uint32_t* slot, hipStream_t stream);
hipError_t launch_pow2_scales(const uint32_t* gmax_bits, int n, float* scales, hipStream_t stream);

// bwd16.hip
hipError_t launch_pack_t16(const float* w, int n_rows_w, int n_cols_w, int n_cols_real, int OT_padded, void* dst, float* winv_slot, uint32_t* wmax_slot,
                           hipStream_t stream);
hipError_t launch_col_abs_sum_max(const float* w, int n_rows, int n_cols, float* slot, hipStream_t stream);
hipError_t launch_pack_head_t16(const float* w, int H, void* dst, float* winv_slot, uint32_t* wmax_slot, hipStream_t stream);
hipError_t launch_chain16(const FieldArgs& f, int H, const void* units, const void* head_t, const float* winv, const float* scales, const void* cos16,
                          void* g16, void* go16, unsigned int* gmax, unsigned int* sat, int nslab, int dry, int group_step, hipStream_t stream);
hipError_t launch_weight_grad16(int cnt, long long tiles_per_image, int n_rows, int g_ct, int x_ct, const void* G, const void* X, float* dW,
                                float* colsum, const float* inv_scale, hipStream_t stream);

// chain_pw16.hip: half-precision gradient chain of the per-point FiLM family (chain_pre_kernel + pw_gm_kernel)
struct PwChainBuffers {
    const void* units_y;      // Y units: W_l^T, stages L-2 .. 0 (launch_pack_pw_chain)
    const void* units_m;      // M units: the Wm2 pair of every layer by channel tile, then Wm1^T
    const void* head_t;
    const float* winv;        // [W_l^T: L (0 unused) | Wm2 pair of layer l: L | Wm1^T | head^T]
    const float* anorm;       // [||W_l||_1: L (0 unused) | head]
    const float* scales;      // {S, 1 / S} x (4 L + 2): per layer (g_pre, g_fr, g_ph), g_mpre, go, then g_y per layer
    const float* lay;         // per layer {r_l, To_l} (launch_pw_split_scales)
    const void* cos16;        // 3 L COS16 slabs of the storing forward
    const float* amax;        // (L, tiles * 32)
    const void* m16;          // TB16 (tiles, 8, 32, 32)
    void* gy16;               // TB16: L slabs (tiles, NT, 32, 32) of g_y
    void* g16;                // TB16: 3 L slabs (tiles, NT, 32, 32) then g_mpre (tiles, 8, 32, 32)
    void* go16;
    unsigned int* gmax;       // dry runs: 3 L: g_mpre, 3 L + 1: go, 3 L + 2 + l: g_y of layer l
    unsigned int* sat;
};
hipError_t launch_chain_pre(const FieldArgs& f, int H, const PwChainBuffers& c, int dry, int group_step, hipStream_t stream);
hipError_t launch_pw_gm(const FieldArgs& f, int H, const PwChainBuffers& c, int dry, int group_step, hipStream_t stream);
hipError_t launch_pw_split_scales(const uint32_t* amaxg_bits, int L, float* scales, float* lay, hipStream_t stream);
size_t pw_chain_y_bytes(int L, int H);
size_t pw_chain_m_bytes(int L, int H);
hipError_t launch_pack_pw_chain(const cnerf_field_params* p, int L, int H, void* units_y, void* units_m, void* head_t, float* winv, float* anorm,
                                uint32_t* wmax, hipStream_t stream);

}  // namespace cnerf
