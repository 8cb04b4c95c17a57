/*
 * cnerf.h -- C ABI of libcnerf_hip.so: the MI355X (gfx950) render path of the conditioned-NeRF GAN.
 *
 * The reference has no FFI layer; its boundary for this path is the Python generator API
 *   generators/generators.py:33-187   ImplicitGenerator3d.forward(z, cam2worlds, img_size, fov, ...)
 *   generators/siren.py:637-668       <SIREN variant>.forward(points, z, img_size, num_steps)
 * Each entry point below names the reference code it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer on the current HIP device unless the comment says "host";
 *   - nothing is allocated, freed or synchronised here: the caller passes every buffer, including the
 *     workspace sized by cnerf_workspace_bytes(), and a stream (hipStream_t passed as void*);
 *   - all launches go to that stream; calls are re-entrant, there is no global state;
 *   - return value: 0 on success, a negative CNERF_E* code otherwise; never throws.
 *   - rays: P = R*R per image, pixel p = row*R + col; points per pass N = P*S per image.
 */
#ifndef CNERF_H
#define CNERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNERF_ABI_VERSION 7

#define CNERF_OK 0
#define CNERF_EINVAL (-22)  /* bad argument / unsupported shape (message via cnerf_last_error) */
#define CNERF_ENOSYS (-38)  /* feature not built into this library */
#define CNERF_ELAUNCH (-5)  /* the HIP runtime refused a launch */

#define CNERF_MAX_LAYERS 16
#define CNERF_MAX_LEVELS 4  /* feature-pyramid levels (siren.py:1444-1473) */

/* cnerf_cfg.flags */
#define CNERF_F_HIERARCHICAL (1u << 0) /* generators.py:110  coarse pass + importance resampling + fine pass */
#define CNERF_F_WHITE_BACK (1u << 1)   /* volumetric_rendering.py:59  rgb += 1 - sum(w) */
#define CNERF_F_LAST_BACK (1u << 2)    /* volumetric_rendering.py:53  w[last] += 1 - sum(w) */
#define CNERF_F_SOFTPLUS (1u << 3)     /* clamp_mode == "softplus" (else "relu"), volumetric_rendering.py:41-44 */
#define CNERF_F_SIGMOID_RGB (1u << 4)  /* siren.py:1227-1234  sigmoid on channels 0..2 of the head */
#define CNERF_F_INPUT_XYZ (1u << 5)    /* siren.py:1158  layer 0 sees features || world xyz (TALLSIREN_dgx) */
#define CNERF_F_RESERVED6 (1u << 6)    /* reserved (ABI v4-v5: selected an experimental kernel variant that was measured slower and
                                          removed in v6); ignored */

/* cnerf_cfg.precision */
#define CNERF_PREC_FP32 0   /* v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains */
#define CNERF_PREC_FP16 1   /* plain fp16 products, fp32 accumulation (one v_mfma_f32_32x32x16_f16 per 16 k-values; weights pre-scaled per
                             * matrix by a power of two, activations rounded to nearest): the numerics class of the reference's GPU path
                             * (torch.cuda.amp.autocast, utils.py:327,643) and of BASELINE config 5; NOT inside the 1e-4 gate (measured
                             * ~1e-3 on rgb / sigma: tests/test_gpu_parity.py::test_single_pass_fp16) -- use fp16x3 for that */
#define CNERF_PREC_FP16X3 2 /* every fp32 operand split into two fp16 parts (22 significant bits), three fp16 MFMAs per product
                             * with fp32 accumulation, weights pre-scaled per matrix by a power of two: fp32-level accuracy
                             * (same parity gate) at 3/16 of the fp32 matrix time; forward and the activation-storing re-run of the
                             * backward; FiLM / sine / residual layers (not the per-point FiLM family) */

/* layer kinds of the field network (siren.py:146-230) */
#define CNERF_LAYER_FILM 0 /* y = sin(freq * (W x + b) + phase), freq/phase per image */
#define CNERF_LAYER_SINE 1 /* y = sin(W x + b) */
#define CNERF_LAYER_RES 2  /* y = sin(x + W2 sin(W1 x + b1) + b2) */
#define CNERF_LAYER_PFILM 3 /* y = sin(freq(p) * (W x + b) + phase(p)): per-point FiLM from the mapping MLP of the looked-up
                               feature (siren.py:163-177, 81-101); all layers of the network must be of this kind, layer 0
                               then reads the world position (K = 3) and the features feed the mapping MLP only.  All three
                               precisions (ABI v7: CNERF_PREC_FP16X3 / CNERF_PREC_FP16 in csrc/field_pw16.hip) */

typedef struct cnerf_cfg {
    int32_t B;            /* images in this call */
    int32_t R;            /* img_size */
    int32_t S;            /* num_steps (coarse samples per ray) */
    int32_t V;            /* feature volume side (level 0) */
    int32_t C;            /* feature channels over all levels; input width of layer 0 = C (+3 with CNERF_F_INPUT_XYZ) */
    int32_t H;            /* hidden width (multiple of 32, <= 256) */
    int32_t L;            /* number of entries in layer_kind */
    int32_t layer_kind[CNERF_MAX_LAYERS];
    float ray_start;      /* forward(ray_start) */
    float ray_end;        /* forward(ray_end) */
    float voxel_length;   /* 1.2, siren.py:555 */
    float noise_std;      /* kwargs["nerf_noise"] */
    uint32_t flags;       /* CNERF_F_* */
    double fov_deg;       /* forward(fov), degrees; kept in double because the reference takes tan() of the
                             Python float before rounding the focal length to fp32 (volumetric_rendering.py:85-87) */
    int32_t n_levels;     /* feature volumes looked up and concatenated (0 or 1: the single volume V, C) */
    int32_t level_V[CNERF_MAX_LEVELS]; /* side of level i */
    int32_t level_C[CNERF_MAX_LEVELS]; /* channels of level i (multiple of 32); sum = C */
    int32_t precision;    /* CNERF_PREC_*: arithmetic of the MLP products in the FORWARD kernels */
    /* In-kernel random draws (ABI v4).  philox != 0: every draw whose tensor in cnerf_rng is NULL is generated inside the
     * kernels by Philox4x32-10 with key = philox_seed and counter = (element index in the whole call, stream, philox_offset) --
     * stream 0 u_strat (B,P,S), 1 eps_coarse (B,P,S), 2 u_fine (B,P,S), 3 eps_final (B,P,S') indexed in sorted order; the noise
     * streams only when noise_std != 0.  A pure function of (seed, offset, index): the backward entry points given the same cfg
     * see the same draws, no tensor travels.  Use a fresh philox_offset per call.  philox == 0: NULL tensors mean "no jitter /
     * no noise" as before (and u_fine is required).  cnerf_philox_fill materialises a stream; oracle/philox.py is its NumPy twin. */
    uint32_t philox;
    uint32_t philox_offset;
    uint64_t philox_seed;
    /* Dropout in training mode (ABI v5): nn.Dropout(p) behind the sine of every FiLMLayer / SirenLayer / PointwiseFiLMLayer
     * (siren.py:158-159,175-176,197-198; residual blocks have none).  drop_p in [0,1), 0 = off (eval mode).  Kept values are
     * multiplied by 1 / (1 - p) like ATen does.  Keep decisions per (dropout layer d, point, channel): the bytes of
     * cnerf_rng.drop_coarse / drop_fine where given, else Philox4x32-10 under (philox_seed, philox_offset) -- whether or not
     * `philox` is set -- stream 4 (coarse pass), 5 (fine pass), 6 (cnerf_field_forward), one block per 4 channels: counter index
     * ((point index in the whole call * n_drop + d) * H + c) / 4, word c % 4 keeps iff >= round(p * 2^32); n_drop = number of
     * layers that are not residual blocks.  The backward entry points given the same cfg (and masks) see the same decisions.
     * Precision fp32 only (CNERF_EINVAL otherwise). */
    float drop_p;
    uint32_t reserved0;
} cnerf_cfg;

/* Feature volumes, channel-last: level[i] is (B, V_i, V_i, V_i, C_i).  HOST struct of device pointers.  The gradient
 * twin of the backward has the same shapes and is accumulated into. */
typedef struct cnerf_volumes {
    const float* level[CNERF_MAX_LEVELS];
} cnerf_volumes;
typedef struct cnerf_grad_volumes {
    float* level[CNERF_MAX_LEVELS];
} cnerf_grad_volumes;

/* Raw parameters of the field network, exactly as the nn.Module holds them (row-major [out][in]).
 * For a RES layer i: w[i]/b[i] = fc1, w2[i]/b2[i] = fc2.  Replaces the state the reference keeps in
 * siren.network[i].layer.{weight,bias} / .fc1 / .fc2 and siren.final_layer (siren.py:611-625). */
typedef struct cnerf_field_params {
    const float* w[CNERF_MAX_LAYERS];
    const float* b[CNERF_MAX_LAYERS];
    const float* w2[CNERF_MAX_LAYERS];
    const float* b2[CNERF_MAX_LAYERS];
    const float* w_final; /* [4][H] */
    const float* b_final; /* [4] */
    /* per-point FiLM family only (siren.mapping_network.network.{0,2}, siren.py:81-101) */
    const float* map_w1;  /* [256][C]      */
    const float* map_b1;  /* [256]         */
    const float* map_w2;  /* [2*L*H][256]  freq rows first, then phase rows */
    const float* map_b2;  /* [2*L*H]       */
} cnerf_field_params;

/* The four random tensors the reference draws, in its draw order; any may be NULL (then: cnerf_cfg.philox):
 *   u_strat    (B,P,S)   uniform, stratified jitter          volumetric_rendering.py:106  (NULL -> 0.5, no jitter)
 *   eps_coarse (B,P,S)   normal, density noise, coarse pass  volumetric_rendering.py:39   (NULL -> 0)
 *   u_fine     (B,P,S)   uniform, inverse-CDF draws          volumetric_rendering.py:319  (required if hierarchical)
 *   eps_final  (B,P,S')  normal, final pass, S' = 2S or S    volumetric_rendering.py:39   (NULL -> 0)
 * eps_final is indexed in SORTED sample order, as the reference adds its noise after the merge.
 *   fine_z     (B,P,S)   test hook, normally NULL: depths that REPLACE the resampled ones for the fine pass and the merge
 *                        (the resampling still runs and fills aux).  The field is chaotic in position, so parity of the
 *                        fine pass / merge / final composite is pinned by forcing the reference's own fine depths.
 *   drop_coarse, drop_fine (n_drop, B, P*S, H) uint8, 1 = keep: dropout decisions of the two field passes (cnerf_cfg.drop_p;
 *                        normally NULL: Philox).  What F.dropout drew in the reference, layer after layer, for tests. */
typedef struct cnerf_rng {
    const float* u_strat;
    const float* eps_coarse;
    const float* u_fine;
    const float* eps_final;
    const float* fine_z;
    const uint8_t* drop_coarse;
    const uint8_t* drop_fine;
} cnerf_rng;

/* Optional intermediate outputs (each may be NULL).  Shapes per image-major layout:
 *   coarse_points (B,P,S,3) coarse_z (B,P,S) coarse_rgb_sigma (B,P,S,4) coarse_weights (B,P,S)
 *   cdf (B,P,S-1) inds (B,P,S) int32   fine_z (B,P,S) fine_rgb_sigma (B,P,S,4)
 *   sort_idx (B,P,2S) int32 (index into cat[fine, coarse])   final_weights (B,P,S')   fine_points (B,P,S,3) */
typedef struct cnerf_aux {
    float* coarse_points;
    float* coarse_z;
    float* coarse_rgb_sigma;
    float* coarse_weights;
    float* cdf;
    int32_t* inds;
    float* fine_z;
    float* fine_rgb_sigma;
    int32_t* sort_idx;
    float* final_weights;
    float* fine_points; /* (B,P,S,3) */
    /* Optional profiling hooks (HOST handles): hipEvent_t created by the caller, recorded on the call's stream
     * immediately before / after the field kernel of the coarse pass ([0],[1]) and of the fine pass ([2],[3]).
     * NULL entries are skipped.  bench.py uses them to time the dominant kernel inside the timed region. */
    void* field_events[4];
    /* Optional (ABI v4), precision CNERF_PREC_FP16X3 only: keep the activations of the two field passes for the half-precision
     * backward instead of re-computing them there -- act16[0] coarse pass, act16[1] fine pass; each {feat (T, n_in, 32, 32),
     * h (n_mats, T, H/32, 32, 32), c (same)} fp16 in the TB16 layout over ALL images of the call (T = B * ceil(R*R*S / 32)).
     * 4 KiB per sample point at H = 256, 4 layers: sized for the 288 GB of HBM of an MI355X (batch 8 at 128x128x(64+64): 69 GB).
     * All-NULL entries: nothing is kept. */
    struct {
        void* feat;
        void* h;
        void* c;
        void* amax; /* ABI v7, CNERF_LAYER_PFILM networks only: (L, T * 32) floats; their feat is (T, 2, 32, 32), h L slabs (T, H/32, 32, 32)
                       then m (T, 8, 32, 32), c 3 L slabs -- 16.5 KiB per sample point and pass at H = 256, L = 8 */
    } act16[2];
} cnerf_aux;

int cnerf_abi_version(void);
/* out[i] = draw i of stream `stream_id` under (seed, offset): uniform in [0,1) (normal == 0) or standard normal (normal != 0),
 * exactly what the kernels generate when cnerf_cfg.philox is set.  Replaces torch.rand / torch.randn at
 * volumetric_rendering.py:39,106,319 for hosts that want the tensors. */
int cnerf_philox_fill(uint64_t seed, uint32_t offset, uint32_t stream_id, int64_t n, int32_t normal, float* out, void* stream);
/* Host string describing the last error raised on the calling thread ("" if none). */
const char* cnerf_last_error(void);

/* Validates cfg and returns the byte sizes the caller must provide:
 *   packed   : packed field weights written by cnerf_pack_field
 *   fvol_cl  : channel-last copy of the feature volume written by cnerf_fvol_channel_last
 *   fwd_ws   : scratch of cnerf_render_forward (coarse / fine rgb_sigma and depths, and the folded FiLM constants of the call:
 *              per image, matrix and channel  freq / 2 pi  and  (freq * bias + phase) / 2 pi, and the layer weights with their rows scaled
 *              by freq / 2 pi per image, prepared once for both field passes in fp32 precision) */
int cnerf_workspace_bytes(const cnerf_cfg* cfg, size_t* packed, size_t* fvol_cl, size_t* fwd_ws);

/* (B,C,V,V,V) channel-first, as unet3d emits it (generators/unet3d.py) -> (B,V,V,V,C) channel-last, so that one
 * trilinear corner is one contiguous C*4-byte line.  Replaces the layout F.grid_sample reads (siren.py:561-567). */
int cnerf_fvol_channel_last(int32_t B, int32_t C, int32_t V, const float* fvol_cf, float* fvol_cl, void* stream); /* C % 32 == 0 */
/* Transpose of the above (used for the gradient of the feature volume). */
int cnerf_fvol_channel_first(int32_t B, int32_t C, int32_t V, const float* fvol_cl, float* fvol_cf, void* stream);

/* Re-orders nn.Linear weights into MFMA A-operand order (see DESIGN.md "packed weights").  params is a HOST struct
 * of device pointers. */
int cnerf_pack_field(const cnerf_cfg* cfg, const cnerf_field_params* params, float* packed, void* stream);

/* Trilinear lookup only: points (B,n,3) world -> feat (B,n,C).  Replaces F.grid_sample + permute
 * (siren.py:555-571, K4/K5 of SURVEY.md 2.1).  This is the unfused "sample pass" kernel the HBM roofline is quoted on. */
int cnerf_gather_features(const cnerf_cfg* cfg, const float* fvol_cl, const float* points, int64_t n_per_image,
                          float* feat, void* stream);

/* Field network at explicit points: points (B,n,3) -> rgb_sigma (B,n,4).
 * Replaces <SIREN>.forward(points, z, img_size, num_steps) (siren.py:637-668 and siblings; extract_shapes.py:63-69).
 * freq/phase: (B, n_film*H) with freq already *15+30 (siren.py:650), NULL when the network has no FiLM layer. */
int cnerf_field_forward(const cnerf_cfg* cfg, const cnerf_volumes* vols, const float* packed, const float* freq,
                        const float* phase, const float* points, int64_t n_per_image, float* rgb_sigma, void* stream);

/* Alpha compositing of n samples per ray: rgb_sigma (rays,n,4), z (rays,n), eps (rays,n) or NULL ->
 * rgb (rays,3), dist (rays), weights (rays,n) (each output may be NULL).
 * Replaces fancy_integration (volumetric_rendering.py:18-70).  Uses cfg->noise_std and flags. */
int cnerf_composite(const cnerf_cfg* cfg, int64_t rays, int32_t n, const float* rgb_sigma, const float* z,
                    const float* eps, float* rgb, float* dist, float* weights, void* stream);

/* Inverse-CDF resampling: z, weights, u (rays,S) -> fine_z (rays,S); optional inds (rays,S) int32, cdf (rays,S-1).
 * Replaces generators.py:123-137 + sample_pdf (volumetric_rendering.py:297-342). */
int cnerf_resample(int64_t rays, int32_t S, const float* z, const float* weights, const float* u, float* fine_z,
                   int32_t* inds, float* cdf, void* stream);

/* The whole path: ImplicitGenerator3d.forward (generators.py:33-187).
 *   cam2world (B,4,4) row-major;  pixels (B,3,R,R) = 2*rgb-1;  depth (B,R,R).  aux may be NULL. */
int cnerf_render_forward(const cnerf_cfg* cfg, const cnerf_volumes* vols, const float* packed, const float* freq,
                         const float* phase, const float* cam2world, const cnerf_rng* rng, float* pixels,
                         float* depth, const cnerf_aux* aux, void* workspace, void* stream);

/* ---- backward (autograd twin of cnerf_render_forward; gradients flow to the field parameters, freq/phase and the
 * feature volume, never to cam2world or the sample positions: generators.py:57,111 run them under no_grad) ----------
 *
 * Step 1  cnerf_merge_composite_backward: d(pixels, depth) -> d(rgb_sigma) of the coarse and the fine samples.
 * Step 2  cnerf_field_backward per pass (coarse, fine) and per chunk of images: re-runs the field forward storing the
 *         activations, back-propagates through the MLP on the MFMA units, scatter-adds d(feature volume) with fp32
 *         atomics, and leaves row-major activation / gradient matrices in the caller's chunk buffers:
 *             act_feat (n,32k)  x0 = layer-0 input tiles            act_go (n,4)   d/d head pre-activation
 *             act_h (L,n,H)     x_l = sin(arg_l)                    act_g  (L,n,H) d/d arg_l
 *             act_c (L,n,H)     cos(arg_l)                          (n = images_in_chunk * points_per_image)
 *         The parameter gradients are then plain GEMMs / column sums over these matrices (rocBLAS territory):
 *             dWarg_l = act_g[l]^T x_{l-1};  dW_l = diag(freq_l) dWarg_l;  db_l = freq_l * colsum(act_g[l])
 *             dphase_l = colsum(act_g[l]);   dfreq_l = rowsum(W_l * dWarg_l) + b_l * dphase_l   (per image)
 *             dW_head = act_go^T x_L;        db_head = colsum(act_go)
 * A residual block owns two consecutive slabs of act_h / act_c / act_g (fc1 then fc2); L in the shapes above then counts
 * slabs:  dW_fc1 = act_g[s]^T x_in, dW_fc2 = act_g[s+1]^T act_h[s].
 *
 * Per-point FiLM family (CNERF_LAYER_PFILM, TALLSIREN siren.py:232-331:  m = LeakyReLU_0.2(Wm1 feat + bm1), [f | p] = Wm2 m + bm2,
 * y_l = sin((15 f_l + 30) (W_l y_{l-1} + b_l) + p_l), y_-1 = xyz): the same call with larger chunk buffers --
 *             act_feat (n,32)                  looked-up feature            act_go (n,4)  d/d head pre-activation
 *             act_h  L (n,H) slabs of y_l, then m (n,256)                    (L*n*H + n*256 floats)
 *             act_c  3L (n,H) slabs: per layer cos(arg), cos*freq, cos*15*pre   (private to the call)
 *             act_g  L (n,H) slabs of g_pre_l = d/d (W_l y_{l-1} + b_l), then G (n, 2*L*H) = d/d (Wm2 m + bm2), row by row in
 *                    that Linear's output order [f of layers 0..L-1 | p of layers 0..L-1]           (3*L*n*H floats)
 *         and no volume scatter (grad_vols is not touched).  The caller finishes with plain GEMMs:
 *             dW_l = act_g[l]^T y_{l-1} (cnerf_weight_grad; layer 0: y_-1 = the sample positions), db_l = colsum(act_g[l]),
 *             dWm2 = G^T m, dbm2 = colsum(G), g_m = (G Wm2) * (m > 0 ? 1 : 0.2), dWm1 = g_m^T feat, dbm1 = colsum(g_m),
 *             d feat = g_m Wm1 -> cnerf_scatter_features. */

/* packed_t: transposed packed weights for the backward; bytes via cnerf_backward_bytes. */
int cnerf_backward_bytes(const cnerf_cfg* cfg, size_t* packed_t);
int cnerf_pack_field_transposed(const cnerf_cfg* cfg, const cnerf_field_params* params, float* packed_t, void* stream);

/* saved tensors are the forward's coarse/fine rgb_sigma and z (cnerf_aux); fine_* may be NULL when not hierarchical.
 * grad_depth may be NULL.  Outputs grad_coarse / grad_fine: (B,P,S,4). */
int cnerf_merge_composite_backward(const cnerf_cfg* cfg, const float* coarse_rgb_sigma, const float* coarse_z,
                                   const float* fine_rgb_sigma, const float* fine_z, const float* eps_final,
                                   const float* grad_pixels, const float* grad_depth, float* grad_coarse,
                                   float* grad_fine, void* stream);

/* pass: 0 = coarse samples (needs u_strat as in the forward), 1 = fine samples (needs fine_z), 2 = explicit points:
 * the u_strat argument then carries points (B, R*R*S, 3) (backward of cnerf_field_forward).  Images [image0,
 * image0 + n_images) of the call described by cfg.  All per-image inputs are the FULL tensors of the forward (the
 * function offsets them); grad_rgb_sigma / saved_rgb_sigma are the (B,P,S,4) tensors of that pass; grad_vols are the
 * full gradient volumes, accumulated into (zero them first).  act_* are chunk buffers for n_images images; act_feat is
 * (n, 32 * input tiles): the concatenated looked-up features (and xyz, zero padded) that layer 0 saw.
 * `packed` is in the layout of cfg->precision (the activation-storing forward runs in that precision); packed_t is always
 * the fp32 transposed layout (the gradient chain is fp32).  drop_mask: cnerf_rng.drop_coarse / drop_fine of that pass (the FULL
 * tensor) or NULL; with cfg->drop_p > 0 the stored sine and cosine rows carry the dropout factor, so the chain, the weight
 * reductions and the scatter are the ones of the eval-mode network. */
int cnerf_field_backward(const cnerf_cfg* cfg, int32_t pass, int32_t image0, int32_t n_images, const cnerf_volumes* vols,
                         const float* packed, const float* packed_t, const float* freq, const float* phase,
                         const float* cam2world, const float* u_strat, const float* fine_z,
                         const float* grad_rgb_sigma, const float* saved_rgb_sigma, float* act_feat, float* act_h,
                         float* act_c, float* act_g, float* act_go, const cnerf_grad_volumes* grad_vols, const uint8_t* drop_mask,
                         void* stream);

/* Weight-gradient reduction over a chunk written by cnerf_field_backward:  dW[b] (H,K) += g_arg[b]^T x[b],
 * colsum[b] (H) += sum over points of g_arg[b], per image b < n_images; g_arg (n_images, n_per_image, H) = act_g of one
 * matrix, x (n_images, n_per_image, K) = that matrix' input (act_feat for layer 0, act_h of the previous matrix otherwise),
 * K a multiple of 32.  From these the host forms dW, db, dfreq, dphase (autograd of FiLMLayer / SirenLayer,
 * siren.py:146-199).  Both outputs are accumulated into (zero them first). */
int cnerf_weight_grad(int32_t n_images, int64_t n_per_image, int32_t H, int32_t K, const float* g_arg, const float* x, float* dW,
                      float* colsum, void* stream);

/* Adjoint of cnerf_gather_features: grad_fvol_cl (B,V,V,V,32) += scatter of grad_feat (B,n,32) with the trilinear weights of
 * points (B,n,3).  Autograd twin of F.grid_sample (siren.py:555-571) for hosts that evaluate the MLP gradients themselves
 * (the per-point FiLM family, siren.py:232-331). */
int cnerf_scatter_features(const cnerf_cfg* cfg, const float* points, int64_t n_per_image, const float* grad_feat,
                           float* grad_fvol_cl, void* stream);

/* ---- half-precision backward (ABI v4): the counterpart of the reference's fp16 autocast training (utils.py:643-711) ------
 * Activations and their gradients travel in fp16, every sum is fp32.  Buffers use the TB16 layout: a matrix with one row per
 * sample point and 32 * CT channels is stored per 32-point tile (the field kernels' work unit: image b, tile k of
 * tiles_per_image = ceil(R*R*S / 32), T = b * tiles_per_image + k) and per 32-channel tile as a dense 32 x 32 fp16 block:
 *     element (T, t, j, c) -> fp16 index ((T * CT + t) * 32 + j) * 32 + c.
 * Rows past the end of an image are rows of their own (gradient rows there are written as zeros). */

/* dW[b] (n_rows, 32 * x_ct) += G[b]^T X[b] and colsum[b] (n_rows) += column sums of G[b], per image b < n_images, over the
 * tiles of that image; G: TB16 with g_ct channel tiles of which the first ceil(n_rows / 32) are used (n_rows = 4 for the head:
 * dW_head = go'^T x_L), X: TB16 with x_ct <= 8 channel tiles.  G holds scale * g; inv_scale (DEVICE scalar, may be NULL = 1)
 * undoes it.  Outputs are accumulated into (zero them first); colsum may be NULL.  Autograd of nn.Linear inside
 * FiLMLayer / SirenLayer (siren.py:146-199) like cnerf_weight_grad, on v_mfma_f32_32x32x16_f16. */
int cnerf_weight_grad16(int32_t n_images, int64_t tiles_per_image, int32_t n_rows, int32_t g_ct, int32_t x_ct, const void* G,
                        const void* X, float* dW, float* colsum, const float* inv_scale, void* stream);

/* Packed operands of the half-precision gradient chain: every W_l^T (and the head's) as fp16 MFMA fragments, each matrix
 * pre-scaled by a power of two, plus their inverse scales.  Bytes via cnerf_backward16_bytes.  FiLM / plain-sine layers
 * (FiLM, plain-sine and residual-block networks; a residual block is two matrices -- fc1, fc2 -- of every per-matrix array below.
 * CNERF_ENOSYS for the per-point FiLM family: use the fp32 backward there). */
int cnerf_backward16_bytes(const cnerf_cfg* cfg, size_t* packed16);
int cnerf_pack_field_chain16(const cnerf_cfg* cfg, const cnerf_field_params* params, void* packed16, void* stream);

/* cnerf_field_backward16.mode */
#define CNERF_B16_STORE 1u /* re-run the field forward (fp16x3 kernel) storing x0, sin(arg_m), cos(arg_m) as fp16 TB16 */
#define CNERF_B16_DRY 2u   /* run the chain over every `group_step`-th tile group WITHOUT stores: only gmax[m] = max |d/d arg_m| */
#define CNERF_B16_CHAIN 4u /* the chain: g16, go16, feature-volume gradients */

/* Twin of cnerf_field_backward (same pass / image-range / per-image-input conventions) for the half-precision path.
 *   cfg->precision must be CNERF_PREC_FP16X3 and `packed` its forward layout (the re-run is that kernel);
 *   act_feat16 (T, n_in, 32, 32), act_h16 / act_g16 (n_mats, T, H/32, 32, 32), act_go16 (T, 1, 32, 32; ZERO it first: only
 *   channels 0..3 are written): fp16 TB16 chunk buffers, T = n_images * ceil(R*R*S / 32); act_c16: the same number of bytes as
 *   act_h16, private to the storing forward and the chain (fragment-major, csrc/bwd16.hpp "COS16");
 *   scales (DEVICE, 2 * (n_mats + 1) floats): per matrix m {S_m, 1 / S_m} = power-of-two scale of act_g16[m], then
 *   {S_go, 1 / S_go} of act_go16 -- the caller derives them from gmax;  gmax (DEVICE, n_mats + 1 uint32, zero it first): bit
 *   patterns of the sampled maxima written by a CNERF_B16_DRY call (max |d/d arg_m| per matrix, then max |go'|).
 * Typical sequence per chunk: STORE | DRY (group_step 16)  ->  scales from gmax  ->  CHAIN  ->  cnerf_weight_grad16 per matrix
 * (G = act_g16[m], X = act_feat16 or act_h16[m-1]; head: G = act_go16, n_rows 4, X = act_h16[last]). */
int cnerf_field_backward16(const cnerf_cfg* cfg, uint32_t mode, int32_t group_step, int32_t pass, int32_t image0, int32_t n_images,
                           const cnerf_volumes* vols, const float* packed, const void* packed16, const float* freq,
                           const float* phase, const float* cam2world, const float* u_strat, const float* fine_z,
                           const float* grad_rgb_sigma, const float* saved_rgb_sigma, void* act_feat16, void* act_h16, void* act_c16,
                           void* act_g16, void* act_go16, const float* scales, uint32_t* gmax, const cnerf_grad_volumes* grad_vols,
                           void* stream);

/* ---- the whole backward in ONE call (ABI v6) --------------------------------------------------------------------------
 * Autograd twin of cnerf_render_forward for hosts without an autograd engine of their own (and the path the PyTorch mirror
 * takes): given d loss / d pixels and d loss / d depth it runs the steps above -- cnerf_merge_composite_backward, then per
 * pass (coarse, fine) and per chunk of images the activation-storing re-run of the forward (unless the forward kept its
 * activations: aux->act16), the gradient chain, the volume scatter, one weight reduction per matrix -- and finishes what the
 * round-2 ABI left to the host: dW_l = diag(freq_l) dWarg_l, db_l, dfreq_l, dphase_l per image, the head's dW / db.
 * Replaces loss.backward() through ImplicitGenerator3d.forward (utils.py:638-711; generators.py:33-187 under autograd).
 *
 *   backward_precision  CNERF_PREC_FP32: exact fp32 chain and weight reductions (re-run in cfg->precision);
 *                       CNERF_PREC_FP16: fp16 operands, fp32 sums (cnerf_field_backward16 / cnerf_weight_grad16; cfg->precision must
 *                       be CNERF_PREC_FP16X3).  Per-point FiLM networks (ABI v7): CNERF_PREC_FP16 runs here (csrc/chain_pw16.hip; grads->map_*
 *                       receive the mapping network's gradients); CNERF_PREC_FP32 answers CNERF_ENOSYS -- the exact path finishes its
 *                       mapping-MLP reductions with library GEMMs on the host (cnerf_field_backward + cnerf_weight_grad + cnerf_scatter_features).
 *   params              the raw parameters (dfreq needs W_l and b_l);  packed: cnerf_pack_field in cfg->precision;
 *   packed_bwd          cnerf_pack_field_transposed (fp32 backward) or cnerf_pack_field_chain16 (fp16 backward).
 *   saved               the forward's coarse / fine rgb_sigma and z (cnerf_aux of that call; fine_* NULL when not hierarchical).
 *   rng                 the forward's cnerf_rng -- u_strat, eps_final, dropout masks; Philox draws follow cfg as in the forward.
 *   act16               the activations the forward kept (cnerf_aux.act16, fp16 backward only) or NULL: re-computed per chunk.
 *   grads               per parameter a buffer of the parameter's shape, ACCUMULATED INTO (zero them first); NULL entries are
 *                       skipped.  grad_freq / grad_phase (B, n_film * H), grad_vols (shapes of vols): accumulated into as well.
 *   images_per_chunk    images whose activation / gradient buffers live in the workspace at a time (1..B; with act16 given
 *                       all B).  workspace: cnerf_backward_workspace_bytes(cfg, backward_precision, images_per_chunk, act16 != NULL).
 *   saturated           optional DEVICE uint32 (zero it first), fp16 backward: incremented once per (tile, matrix) in which a stored
 *                       gradient exceeded fp16's range and was clamped -- the per-matrix scale comes from a SAMPLED maximum
 *                       (every 16th tile group once there are >= 32768 of them); non-zero means: repeat with exhaustive sampling
 *                       (cnerf_field_backward16 with group_step 1) or in fp32. */
typedef struct cnerf_field_param_grads {
    float* w[CNERF_MAX_LAYERS];
    float* b[CNERF_MAX_LAYERS];
    float* w2[CNERF_MAX_LAYERS];
    float* b2[CNERF_MAX_LAYERS];
    float* w_final;
    float* b_final;
    /* per-point FiLM family only (ABI v7): gradients of siren.mapping_network.network.{0,2} */
    float* map_w1; /* [256][C]      */
    float* map_b1; /* [256]         */
    float* map_w2; /* [2*L*H][256]  */
    float* map_b2; /* [2*L*H]       */
} cnerf_field_param_grads;

typedef struct cnerf_saved {
    const float* coarse_rgb_sigma; /* (B,P,S,4) */
    const float* coarse_z;         /* (B,P,S)   */
    const float* fine_rgb_sigma;   /* (B,P,S,4) or NULL */
    const float* fine_z;           /* (B,P,S) or NULL: the depths the fine pass used */
} cnerf_saved;

/* Half-precision backward, feature-volume gradient: the ray passes' input-tile gradients go through the workspace (128 B per point) and
 * are added to grad_vols pre-reduced per 8 x 8-pixel patch (csrc/scatter_patch.hip); explicit points are added by the gradient chain
 * itself.  Same addends either way.  Environment CNERF_SCATTER=chain | coarse makes the chain add both ray passes' / the fine pass's
 * too (A/B runs and tests only). */
int cnerf_backward_workspace_bytes(const cnerf_cfg* cfg, int32_t backward_precision, int32_t images_per_chunk, int32_t have_act16,
                                   size_t* bytes);
int cnerf_render_backward(const cnerf_cfg* cfg, int32_t backward_precision, int32_t images_per_chunk, const cnerf_volumes* vols,
                          const cnerf_field_params* params, const float* packed, const void* packed_bwd, const float* freq,
                          const float* phase, const float* cam2world, const cnerf_rng* rng, const cnerf_saved* saved,
                          const cnerf_aux* act16, const float* grad_pixels, const float* grad_depth,
                          const cnerf_field_param_grads* grads, float* grad_freq, float* grad_phase,
                          const cnerf_grad_volumes* grad_vols, uint32_t* saturated, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CNERF_H */
