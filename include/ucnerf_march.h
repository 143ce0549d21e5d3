/*
 * ucnerf_march.h -- C ABI of libucnerf_march.so (MI355X / gfx950 only).
 *
 * The drop-in boundary for UC-NeRF's per-ray sampling + integration hot path.
 * All entry points are `extern "C"`, take plain device/host pointers and sizes (no
 * torch / ATen types), enqueue work on the HIP stream they are given and return
 * immediately (asynchronous, like the reference's kernels).  Return value: 0 on
 * success, non-zero on a precondition or launch failure; ucn_last_error() then holds
 * the message (the Python host raises RuntimeError with it -- the counterpart of the
 * reference's TORCH_CHECK / std::runtime_error, gridencoder.cu:15-18,381,398).
 *
 * "ref:" comments cite the interface each function replaces, relative to
 * /root/reference/nerf/.
 *
 * Pointer residency: every `const float*` / `float*` data pointer is DEVICE memory
 * unless the comment says HOST.  Small metadata (level offsets, grid sizes, the field
 * descriptor structs) is HOST memory: the reference keeps `offsets` on the device and
 * re-derives per-level constants inside every thread (gridencoder.cu:137-139); here
 * they are derived once on the host and travel as kernel arguments, so host and
 * device (and the CPU oracle) agree on them bit-for-bit.
 */
#ifndef UCNERF_MARCH_H
#define UCNERF_MARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UCN_MAX_LEVELS 24
#define UCN_DTYPE_F32 0
#define UCN_DTYPE_F16 1

typedef void *ucn_stream_t; /* hipStream_t; NULL = the legacy default stream */

/* ------------------------------------------------------------------ diagnostics */
const char *ucn_last_error(void);
/* ABI version: bump when a signature changes (checked by the Python loader). */
uint32_t ucn_abi_version(void);
/* HBM bandwidth probe: device-to-device copy kernel, returns 0 and leaves timing to the caller. */
int ucn_probe_copy(const float *src, float *dst, uint64_t n_floats, ucn_stream_t stream);

/* ------------------------------------------------- (b2) the `_gridencoder` operator
 * ref: gridencoder/src/gridencoder.h:12-15, bindings.cpp:5-9, gridencoder.cu:448-503,639-645.
 * Same argument order and meaning as the pybind functions; tensors become pointers,
 * `offsets` is a HOST int32[L+1] array, `emb_dtype` says whether embeddings/outputs/grad
 * are float32 or float16 (the reference dispatches on the tensor dtype, :467,498).
 * Caller allocates every output (grid.py:47-52,77-82); grad_embeddings must be pre-zeroed.
 * D in {2,3,4,5}, C in {1,2,4,8} else error (gridencoder.cu:376-399). */
int ucn_grid_encode_forward(const float *inputs /*[B,D] in [0,1]*/, const void *embeddings /*[rows,C]*/,
                            const int32_t *offsets_host /*[L+1]*/, void *outputs /*[L,B,C]*/,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                            void *dy_dx /*[B,L*D*C] or NULL*/, uint32_t gridtype, int align_corners,
                            uint32_t interp, int emb_dtype, ucn_stream_t stream);

int ucn_grid_encode_backward(const void *grad /*[L,B,C]*/, const float *inputs, const void *embeddings,
                             const int32_t *offsets_host, void *grad_embeddings /*[rows,C] +=*/,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             const void *dy_dx /*or NULL*/, void *grad_inputs /*[B,D] or NULL*/,
                             uint32_t gridtype, int align_corners, uint32_t interp, int emb_dtype,
                             ucn_stream_t stream);

int ucn_grad_total_variation(const float *inputs, const float *embeddings, float *grad /*[rows,C] +=*/,
                             const int32_t *offsets_host, float weight, uint32_t B, uint32_t D, uint32_t C,
                             uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                             ucn_stream_t stream);

/* ------------------------------------------------- (b1) fused ray-march entry points
 * These are what Model.forward (internal/models.py:97-365) calls instead of the
 * unfused chain stepfun -> render.cast_rays -> coord.track_linearize -> GridEncoder ->
 * MLP -> render.compute_alpha_weights -> render.volumetric_rendering. */

/* One hash-grid + MLP "field" (ref: models.py:367-483 MLP.__init__; NerfMLP / PropMLP).
 * HOST struct of DEVICE pointers in PyTorch nn.Linear layout ([out,in] row-major). */
typedef struct ucn_field {
    /* grid (ref: gridencoder/grid.py:96-149) */
    const float *embeddings;         /* [rows, level_dim] float32 */
    const int32_t *offsets_host;     /* HOST [num_levels+1] */
    const int32_t *grid_sizes_host;  /* HOST [num_levels]  (grid.py:142; squared in int32, models.py:495) */
    uint32_t num_levels, level_dim, base_resolution;
    float log2_per_level_scale;
    /* density MLP (models.py:438-441): Linear(F,64) ReLU Linear(64, n_bottleneck) */
    const float *w_d0, *b_d0, *w_d1, *b_d1;
    uint32_t n_bottleneck;           /* 1 for PropMLP (disable_rgb), else bottleneck_width */
    /* colour MLP (models.py:456-483), NULL when n_bottleneck == 1 */
    const float *w_c0, *b_c0;        /* [n_width, n_bottleneck + n_dir] */
    const float *w_c1, *b_c1;        /* [n_width, n_width + n_bottleneck + n_dir] */
    const float *w_rgb, *b_rgb;      /* [3, n_width] */
    uint32_t n_width, n_dir;         /* n_dir = 3 + 6*deg_view = 27 */
    float density_bias, rgb_premultiplier, rgb_bias, rgb_padding;
    /* MFMA-ordered weight copies, filled by ucn_field_pack; size from ucn_field_packed_floats */
    float *packed;
    /* 0: fp32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 products, 157 TF ceiling);
     * 1: split-f16 MFMA (hi/lo f16 operands, 3 x v_mfma_f32_32x32x16_f16, fp32 accumulate; ~3e-7
     *    relative per product).  Must be the same at ucn_field_pack and ucn_field_mlp time. */
    uint32_t mlp_mode;
} ucn_field_t;

uint64_t ucn_field_packed_floats(const ucn_field_t *f);
/* Re-pack after every weight update (render: once; train: once per step). */
int ucn_field_pack(const ucn_field_t *f, ucn_stream_t stream);

/* ref: stepfun.py:75-105 max_dilate_weights + models.py:168-191 (trim, anneal, logits) +
 * stepfun.py:251-294 sample_intervals.  n_prev == 0 selects the first level (sdist=[0,1], w=[1]).
 * u_table: DEVICE [S] = the linspace of stepfun.py:206 (eval) or :215 (train).
 * jitter : NULL (eval) or DEVICE [N, jitter_cols] U[0,1) draws (stepfun.py:216).
 * dilation <= 0 selects the reference's use_dilation == False branch (models.py:167-168, both dilation knobs 0): the
 * previous fenceposts / weights are resampled as they are (no envelope, no trim, no renormalisation). */
int ucn_resample(const float *sdist_prev /*[N,n_prev+1]*/, const float *weights_prev /*[N,n_prev]*/,
                 uint32_t n_prev, float dilation, float anneal, float resample_padding,
                 const float *u_table, const float *jitter, uint32_t jitter_cols, float max_jitter,
                 uint32_t N, uint32_t S, float *sdist_out /*[N,S+1]*/, ucn_stream_t stream);

/* ref: render.py:139-146 -- the two cone cross-section axes from cam_dirs x rand_vec. */
int ucn_cone_basis(const float *cam_dirs /*[N,3]*/, const float *rand_vec /*[N,3]*/, uint32_t N,
                   float *basis_out /*[N,6] = e1,e2*/, ucn_stream_t stream);

/* ref: tsdf.py:115-219 TSDF.integrate_tsdf -- fuse B depth (and colour) images into the volume, one thread per voxel,
 * the views of the call applied in order (running weighted mean, new weight 1).  voxel_world = the reference's
 * `voxel_world_coords` [4][N] (homogeneous, inverse-contracted voxel centres); w2c = rows 0..2 of inverse(c2w) per view;
 * sampling = grid_sample(nearest, zeros, align_corners=False).  colors / color NULL together. */
int ucn_tsdf_integrate(const float *voxel_world, uint32_t N, const float *w2c /*[B][3][4]*/, const float *K /*[3][3]*/,
                       const float *depth /*[B][H][W]*/, const float *color /*[B][3][H][W]|NULL*/, uint32_t B,
                       uint32_t H, uint32_t W, float truncation, float *values /*[N]*/, float *weights /*[N]*/,
                       float *colors /*[N][3]|NULL*/, ucn_stream_t stream);

/* Launch-shape flag, OR-ed into ucn_march_features' `sample_major` and ucn_field_mlp's `rays_fastest` argument:
 * the two kernels are meant to run SIMULTANEOUSLY on two HIP streams and share every CU -- the featurisation as
 * 512-thread workgroups (two waves per SIMD) that reserve 88 KiB of LDS, the MLP with its 64 KiB weight ring
 * (72 KiB in all), so that exactly one workgroup of each fits a CU (registers: 2 x 104 + 296 of 512 per SIMD lane).
 * Results do not depend on it. */
#define UCN_LAUNCH_CORESIDENT 0x100
/* OR-ed into ucn_march_features' layout argument: field->embeddings points to an IEEE half copy of the table ([rows, C]
 * _Float16) -- what the reference gathers under autocast (gridencoder/grid.py:41-44: `embeddings.to(torch.half)` when
 * autocast is on and C is even).  The interpolation arithmetic stays fp32. */
#define UCN_TABLE_F16 0x200
/* ... and (with UCN_TABLE_F16, level_dim 2, a level-major layout) features_out receives [num_levels][N*S] PAIRS OF BF16
 * (4 bytes per level and sample, round to nearest even) instead of float pairs: the operand format of the bf16 inference
 * MLP (ucn_train_fwd with feat_level_dim = 2 | UCN_FEAT_BF16), which would round the floats the same way. */
#define UCN_FEATURES_BF16 0x400
/* OR-ed into ucn_march_features_backward's layout: accumulate the row blocks in int32 FIXED POINT (one fire-and-forget 64-bit LDS add
 * per channel pair instead of a compare-and-swap round trip; scale = a power of two from a guaranteed per-task bound on every row sum,
 * so it cannot overflow).  Order-independent, hence bit-reproducible per task; resolution ~2^-30 of the task's summed |gradient| --
 * the autocast training step's mode (the reference accumulates that step's table gradient in fp16, gridencoder.cu:319-334); the fp32
 * step and every parity test of it keep exact fp32 adds.  Ignored for level_dim 1 and for calls of more than 2^22 samples. */
#define UCN_BWD_FIXED_POINT 0x800
/* OR-ed into ucn_march_features' layout: the rays of the call are NOT neighbouring pixels (a training batch of random rays) -- the lanes of
 * a wave share no cache lines on any hashed level, so every hashed level takes the lane-paired corner fetch (rendering: only the levels
 * finer than 2048, where neighbouring pixels stop sharing lines).  A scheduling hint: the features are bit-identical either way. */
#define UCN_RAYS_INCOHERENT 0x1000
/* OR-ed into ucn_train_fwd's feat_level_dim: `feat` holds the bf16 pairs UCN_FEATURES_BF16 produced. */
#define UCN_FEAT_BF16 0x100

/* ref: render.py:94-152 cast_rays + coord.py:60-116 contraction + grid.py:158-174 /
 * gridencoder.cu:87-199 + models.py:494-496 (erf damping, mean over the 6 multisamples).
 * features_out layout [num_levels][N*S][level_dim]  (level-major like gridencoder.cu:108).
 * flip/spin NULL = deterministic hexagon pattern (rand=False). coord_out/tmean_out optional.
 * levels_per_block: levels handled by one thread (the sample geometry is derived once per group);
 * 0 = auto (levels of resolution <= 2048 in groups of eight, finer levels alone); results do not depend on it. */
int ucn_march_features(const ucn_field_t *f, const float *sdist /*[N,S+1]*/, const float *near_ /*[N]*/,
                       const float *far_ /*[N]*/, const float *origins, const float *directions,
                       const float *basis /*[N,6]*/, const float *radii /*[N]*/,
                       const float *flip /*[N,S]|NULL*/, const float *spin /*[N,S]|NULL*/,
                       float std_scale, uint32_t N, uint32_t S, uint32_t levels_per_block,
                       int sample_major /*1: features_out is [N*S][num_levels*level_dim] instead*/,
                       float *features_out, float *coord_out /*[N,S,3]|NULL*/,
                       float *tmean_out /*[N,S]|NULL = mean t of the 6 multisamples (GradientScaler input)*/,
                       ucn_stream_t stream);

/* Backward of ucn_march_features w.r.t. the table (ref: grid.py:68-89 -> gridencoder.cu:248-340 composed
 * with the erf damping and mean of models.py:495-496; positions carry no gradient, coord.py:75).
 * grad_embeddings [rows, level_dim] is accumulated into (pre-zero it, like grid.py:77).
 * levels_per_block = 0 picks the algorithm: row-block ownership in LDS without global atomics while the
 * table has <= 64 blocks of 128 KiB per level, else the atomic scatter; >= 1 forces the atomic scatter.
 * sample_major: 0 = grad_features [num_levels][N*S][level_dim], 1 = [N*S][num_levels*level_dim] (what autograd
 * hands to grid.py:68), 3 = [num_levels*level_dim][N*S] (the output of a transposed dgrad GEMM; row-block
 * algorithms only), 4 (ABI 26) = layout 0 with every value ALREADY divided by 6 (ucn_train_bwd with UCN_GFEAT_LEVEL_MAJOR wrote it):
 * the row-block kernel reads it in place -- no level-major copy, no division in the mask pass; needs a workspace, refused where the
 * call would fall back to the atomic kernels. */
int ucn_march_features_backward(const ucn_field_t *f, const float *sdist, const float *near_, const float *far_,
                                const float *origins, const float *directions, const float *basis,
                                const float *radii, const float *flip, const float *spin, float std_scale,
                                uint32_t N, uint32_t S, uint32_t levels_per_block, int sample_major,
                                const float *grad_features, float *grad_embeddings,
                                float *workspace /*DEVICE, ucn_march_features_backward_ws_floats(f,N,S) floats (the
                                                   samples' geometry cache + per-level row-block masks + a
                                                   level-major copy of a layout-1/3 gradient), or NULL: the row-block
                                                   algorithm then re-derives the geometry in every workgroup and
                                                   cannot compact its work (several times slower)*/,
                                ucn_stream_t stream);
uint64_t ucn_march_features_backward_ws_floats(const ucn_field_t *f, uint32_t N, uint32_t S);
/* ABI 26: 1 if that call (levels_per_block = 0, with its workspace) runs the compacted row-block kernel -- the only route that accepts
 * sample_major = 4; 0 where it would fall back (more than 512 row blocks per level, >= 2^29 samples). */
int ucn_march_features_backward_row_blocks(const ucn_field_t *f, uint32_t N, uint32_t S);

/* Introspection of the fused featurisation's geometry stage (the parity tests of SURVEY 8 rows a5 / a6; not on the
 * rendering path): the six multisample Gaussians of every sample exactly as ucn_march_features derives them
 * (ref: render.py:94-152 cast_rays, then coord.py:60-116 track_linearize('contract') and the /2 of models.py:491-493).
 * out [N, S, 6, UCN_CAST_PROBE_FLOATS] = {mean x, y, z, std, t  (cast_rays' returns),
 *                                          contracted mean / 2 x, y, z, contracted std / 2, 1 / sqrt(8 std_c^2)}. */
#define UCN_CAST_PROBE_FLOATS 10
int ucn_cast_probe(const float *sdist, const float *near_, const float *far_, const float *origins,
                   const float *directions, const float *basis, const float *radii, const float *flip /*|NULL*/,
                   const float *spin /*|NULL*/, float std_scale, uint32_t N, uint32_t S, float *out, ucn_stream_t stream);
/* coord.py:60-72 contract_mean_std through the kernels' own device function: means [B,3], stds [B] ->
 * contracted mean / 2 [B,3], contracted std / 2 [B]. */
int ucn_contract_probe(const float *means, const float *stds, uint32_t B, float *out_mean, float *out_std,
                       ucn_stream_t stream);

/* Same featurisation for caller-supplied Gaussians (ref: models.py:485-512 predict_density as
 * called by extract.py:56-57,96): means [B,G,3], stds [B,G]; warp=0 skips the contraction. */
int ucn_points_features(const ucn_field_t *f, const float *means, const float *stds, uint32_t B,
                        uint32_t G, int warp, uint32_t levels_per_block, float *features_out /*[L][B][C]*/,
                        float *coord_out /*[B,3]|NULL*/, ucn_stream_t stream);

/* ref: coord.py:214-225 pos_enc(viewdirs): the per-ray direction inputs of the colour MLP.
 * mlp_mode 0: folded through the direction columns of lin_second_stage_{0,1} into additive terms
 *             [N,2,n_width];  mlp_mode 1: the encoding itself as one 32-wide input tile [N,32]
 *             (k < n_dir: pos_enc, k = n_dir: 1 -- the bias slot, then 0).
 * ucn_field_dir_floats = number of floats of dir_bias_out for N rays in the field's mode. */
uint64_t ucn_field_dir_floats(const ucn_field_t *f, uint32_t N);
int ucn_field_dir_bias(const ucn_field_t *f, const float *viewdirs /*[N,3]*/, uint32_t N,
                       float *dir_bias_out, ucn_stream_t stream);

/* ref: models.py:507-508,581,599-674 -- density MLP, softplus, colour MLP, sigmoid + padding.
 * MFMA (fp32-input or split-f16 per mlp_mode), activations chained through registers.
 * features [L][B][C] as written by ucn_march_features: rays_fastest = 0 -> b = ray*samples_per_ray + s
 * (layout 0), rays_fastest = 1 -> b = s*n_rays + ray (layout 2).  Outputs are always [ray][s]-ordered.
 * bottleneck_out optional [B, n_bottleneck]. rgb_out NULL for PropMLP. */
int ucn_field_mlp(const ucn_field_t *f, const float *features, uint32_t B, uint32_t samples_per_ray,
                  int rays_fastest, const float *dir_bias /*from ucn_field_dir_bias|NULL*/,
                  float *density_out /*[B]*/, float *rgb_out /*[B,3]|NULL*/, float *bottleneck_out,
                  ucn_stream_t stream);

/* Early-termination sample compaction (BASELINE.json north_star).  The reference runs the colour MLP on every sample
 * (models.py:221-243 -> :581-674) although a sample whose compositing weight w = alpha * T (render.py:155-174) is ~0
 * cannot change the pixel.  The path here, per pass of the last level: density head first (ucn_field_mlp with
 * rgb_out = NULL), weights by ucn_composite (rgbs = NULL), then
 *   ucn_compact_alive: idx_out[0 .. *count) = the FEATURE indices b (b = s * n_rays + ray when rays_fastest, else
 *     ray * S + s) of the samples with weight >= min_weight, in unspecified order (per wave: ballot + popcount, one
 *     atomic per wave for the base); *count is reset by the call (stream-ordered);
 *   ucn_field_rgb_compacted: the colour layers of those samples only, rgb_out[ray][s] written for them (the caller
 *     zero-fills the rest).  Workgroups beyond *count exit at once: no host read-back.
 * Pixel error <= S * min_weight * max|rgb| by construction. */
int ucn_compact_alive(const float *weights /*[N,S]*/, uint32_t N, uint32_t S, int rays_fastest, float min_weight,
                      uint32_t *idx_out /*[N*S]*/, uint32_t *count /*[1]*/, ucn_stream_t stream);
int ucn_field_rgb_compacted(const ucn_field_t *f, const float *features, uint32_t B, uint32_t samples_per_ray,
                            int rays_fastest, const float *dir_bias, const uint32_t *idx, const uint32_t *count,
                            float *rgb_out /*[B,3]*/, ucn_stream_t stream);

/* ref: render.py:155-174 compute_alpha_weights + :177-244 volumetric_rendering +
 * stepfun.py:329-339 weighted_percentile.  rgbs NULL = PropMLP zeros (models.py:584-585).
 * out_main [N,5] = r,g,b,depth,acc ; out_extras [N,4] = distance_mean, p5, median, p95 (or NULL). */
int ucn_composite(const float *density /*[N,S]*/, const float *rgbs /*[N,S,3]|NULL*/,
                  const float *sdist /*[N,S+1]*/, const float *near_, const float *far_,
                  const float *directions, float bg_intensity, int opaque_background, uint32_t N,
                  uint32_t S, float *weights_out /*[N,S]*/, float *out_main, float *out_extras,
                  ucn_stream_t stream);

/* Backward of ucn_composite's differentiable outputs w.r.t. density and rgbs: what autograd derives from
 * render.py:155-174 + :203-216 in the reference's training step (train.py:165-221).  g_weights [N,S]|NULL is the
 * gradient arriving at `weights` directly (interlevel / distortion losses), g_main [N,5] the gradient of
 * out_main (r,g,b,depth,acc).  g_rgbs NULL iff rgbs NULL. */
int ucn_composite_backward(const float *density, const float *rgbs, const float *sdist, const float *near_,
                           const float *far_, const float *directions, float bg_intensity,
                           int opaque_background, uint32_t N, uint32_t S, const float *g_weights,
                           const float *g_main, float *g_density /*[N,S]*/, float *g_rgbs /*[N,S,3]|NULL*/,
                           ucn_stream_t stream);

/* ------------------------------------------------- training-side reductions + optimiser
 * ucn_adam_step: torch.optim.Adam (amsgrad = False, weight_decay = 0; what train_utils.py:347-366 builds) on one
 * fp32 tensor of n elements, in place, one pass; step counts from 1; sanitize_grad != 0 folds in the
 * `param.grad.nan_to_num_()` of train_utils.py:343 (and stores the sanitised gradient like the reference).
 * ucn_distortion_loss: stepfun.py:297-307 lossfun_distortion per ray in O(S): g_loss NULL -> out [N] = loss per
 * ray; g_loss [N] -> out [N,S] = d(sum_n g_loss[n] loss[n]) / d w (t carries no gradient). */
int ucn_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n, float lr, float beta1,
                  float beta2, float eps, uint32_t step, int sanitize_grad, ucn_stream_t stream);
/* ref: models.py:297-306 (hash decay: mean over levels and channels of the per-level mean of embeddings^2).
 * embeddings DEVICE [rows, C]; offsets_host HOST int32 [L+1] (first row of each level, offsets[L] = rows).
 * g_dev NULL: forward -- out DEVICE [1] = the loss, workspace DEVICE [1024] floats.
 * g_dev DEVICE [1] (gradient of the loss): backward -- out DEVICE [rows, C] = g * d loss / d embeddings. */
int ucn_hash_decay(const float *embeddings, const int32_t *offsets_host, uint32_t L, uint32_t C, const float *g_dev,
                   float *out, float *workspace, ucn_stream_t stream);
/* ref: train_utils.py:342-344 `param.grad.nan_to_num_()` for every parameter: `count` fp32 DEVICE tensors (HOST arrays of
 * their pointers and element counts) sanitised in place by one launch per 48 tensors. */
int ucn_nan_to_num_many(float *const *tensors_host, const uint64_t *numel_host, uint32_t count, ucn_stream_t stream);
/* ucn_adam_step on `count` small tensors that share the hyper-parameters and the step count (one parameter group) in
 * one launch per 24 tensors: HOST arrays of DEVICE pointers / element counts. */
int ucn_adam_step_many(float *const *params_host, float *const *grads_host, float *const *exp_avg_host,
                       float *const *exp_avg_sq_host, const uint64_t *numel_host, uint32_t count, float lr, float beta1,
                       float beta2, float eps, uint32_t step, int sanitize_grad, ucn_stream_t stream);
int ucn_distortion_loss(const float *t /*[N,S+1]*/, const float *w /*[N,S]*/, uint32_t N, uint32_t S,
                        const float *g_loss, float *out, ucn_stream_t stream);
/* ref: train_utils.py:247-270 anti_interlevel_loss for ONE proposal level (stepfun.py:395-403 blur_stepfun +
 * math.py:110-133 sorted_interp_quad): c [N,S_nerf+1], w [N,S_nerf] = the (detached) NeRF level; cp [N,S_prop+1],
 * wp [N,S_prop] = the proposal level.  loss_ray [N] = sum_j max(w_s - wp, 0)^2 / (wp + 1e-5); dterm [N,S_prop] =
 * d loss_ray / d wp (the caller applies the mean and the loss multiplier). */
int ucn_interlevel_loss(const float *c, const float *w, uint32_t S_nerf, const float *cp, const float *wp,
                        uint32_t S_prop, float pulse_width, uint32_t N, float *loss_ray, float *dterm,
                        ucn_stream_t stream);

/* ------------------------------------------------- virtual-pose depth warping
 * ref: train_utils.py:19-55 img_warping / :58-98 img_warping_for_depth, called per training step on a full depth
 * map by datasets.py:511-529.  rel_pose_host = HOST float[12], rows of the upper 3x4 of inv(src_pose) @ ref_pose
 * (OpenCV convention, float32 like the reference); intrinsic_host = HOST float[9].  pts_out [H,W,2] = projected
 * (x, y) in the source frame, mask_out [H,W] uint8 = depth > 0 and inside the frame, z_src_out [H,W]|NULL = depth in
 * the source camera.  ucn_warp_scatter_depth: depth_tgt[int(y), int(x)] = z for the masked pixels, later pixels
 * (row-major) overwriting earlier ones; owner_ws = DEVICE uint32[H*W] scratch. */
int ucn_img_warping(const float *depth, const float *rel_pose_host, const float *intrinsic_host, uint32_t H, uint32_t W,
                    float *pts_out, uint8_t *mask_out, float *z_src_out, ucn_stream_t stream);
int ucn_warp_scatter_depth(const float *pts, const uint8_t *mask, const float *z_src, uint32_t H, uint32_t W,
                           uint32_t *owner_ws, float *depth_tgt, ucn_stream_t stream);

/* Elementwise halves of the colour MLP's hidden layers in the training graph (models.py:615-640 under autograd):
 * ucn_bias_relu: pre_inout [N*S, W] <- relu(pre + per_ray[ray]) with per_ray [N, W] (the direction block of the
 * layer times the ray's encoding, plus the bias); ucn_relu_backward_reduce: d_pre = gy * [h > 0] and
 * d_per_ray [N, W] = sum over the ray's S samples of d_pre.  dtype 0 = float32, 2 = bfloat16 (autocast); W % 8 == 0. */
int ucn_bias_relu(void *pre_inout, const void *per_ray, uint32_t N, uint32_t S, uint32_t W, int dtype, ucn_stream_t stream);
int ucn_relu_backward_reduce(const void *gy, const void *h, void *d_pre, void *d_per_ray, uint32_t N, uint32_t S,
                             uint32_t W, int dtype, ucn_stream_t stream);

/* Training-time forward of the NeRF field's dense layers in one kernel (models.py:507-674 under
 * accelerator.autocast(): bf16 operands, fp32 accumulation), writing every activation the backward needs once:
 * h0 [M,64], x [M,256] (bottleneck), h1, h2 [M,256] as bf16, raw [M] = x[:,0], y [M,3] = pre-sigmoid colour, M = N*S,
 * and the ReLU masks m0 / m1 / m2 (16 bits per 32-feature tile and wave half) for ucn_train_bwd.
 * packed = ucn_train_fwd_fragments() fragments of 64 lanes x 8 bf16 (1 KiB), of which the forward reads the first 256:
 * W_d0, W_d1, then the colour layers COMPOSED with the activation-free bottleneck -- (W0x W_d1) [256 x 64] and
 * [W1h | W1x W_d1] [256 x 320] -- and W_rgb woven behind each output-tile pair of the last hidden layer, all in MFMA
 * A-operand order [out tile pair][in tile][k-step][tile of the pair], k permuted to the accumulator layout of the
 * producing layer (ucnerf_amd/internal/train_graph.py::_head_gather_index); pr0 / pr1 carry W0x b_d1 / W1x b_d1; biases / per-ray terms in accumulator order
 * [tile][wave half][16] (pr0, pr1: [N, 8, 2, 16] = direction block of the colour layer times the ray's encoding
 * plus its bias).  Widths are the reference's (64, 256, 256, 256, 3); feat [M,F] fp32 with F <= 64 (one feature tile up to 32, two above:
 * the first and last matrices then have 4 more fragments). */
uint64_t ucn_train_fwd_fragments(void);
/* act_ld: 0 = h0 / x / h1 / h2 are dense [M,64] / [M,256] matrices; otherwise all four are column blocks of ONE row-major
 * [M, act_ld] bf16 buffer (the caller passes the four column-offset pointers), so that the weight-gradient GEMMs can
 * take adjacent blocks as one operand ([h1 | x | per-ray columns] is the input of the reference's concatenated layer,
 * models.py:620-640).  ray_cols [N,32] bf16 | NULL: per-RAY columns (direction encoding, a constant 1 for the bias)
 * copied into every sample's row at ray_dst (a column-offset pointer into the same buffer).  Rows that start on 128-byte
 * lines matter: act_ld = 864 (1728-byte rows) costs 15 % against 1024.  feat_bf16 | NULL (row stride act_ld when act_ld != 0,
 * i.e. one more column block of the same buffer, else F).  h0 / x / h1 / h2 / m0 / m1 / m2 all NULL = INFERENCE: nothing but
 * raw / y (density / rgb with head) is written -- the mixed-precision render path (render_image under autocast).  There
 * pr0 = pr1 = NULL selects the stream form with the direction tile inside: ray_cols [N,32] bf16 = [dir_enc (27), 1, 0...]
 * is one more input tile of the two colour layers (its column 27 carries the bias), packed by
 * train_graph.py::_head_gather_index(dir_in_stream=True): bf16 copy of the features (operand of the first layer's weight gradient;
 * F % 8 == 0).  head: HOST float[4] {density_bias, rgb_premultiplier, rgb_bias, rgb_padding} | NULL.  With head the
 * output activations (models.py:515 softplus, :667-672 sigmoid + padding) are applied in fp32 before the store:
 * raw := density, y := rgb.  r04: x alone may be NULL in the training form -- the bottleneck is linear in h0 (models.py:508), so
 * its weight-gradient operands can be formed from d^T h0 (train_graph.py::_FusedHeads) -- and ucn_train_bwd's gx likewise; dy's column 3
 * then carries the density head's gradient at the bottleneck's feature 0. */
int ucn_train_fwd(const float *feat, uint32_t F, const void *packed, const float *bias_d0, const float *bias_d1,
                  const float *bias_rgb, const float *pr0, const float *pr1, uint32_t N, uint32_t S, void *h0, void *x,
                  void *h1, void *h2, uint32_t act_ld, const void *ray_cols, void *ray_dst, void *feat_bf16,
                  const float *head, float *raw, float *y, uint32_t *m0 /*[M,2]*/, void *m1 /*[M,2] x 16 B*/, void *m2,
                  uint32_t feat_level_dim /*0: feat [M,F]; C (the inference form only: pr0 = pr1 = NULL, ray_cols = the
                  rays' direction tiles, no stores): feat = ucn_march_features' layout 2, [L][B][C] with b = s N + ray -- a
                  wave's lanes are then neighbouring rays, outputs stay [ray][sample]*/,
                  ucn_stream_t stream);
/* The same chain backwards (dgrad): gy [M,3] bf16 (gradient of y), graw [M] bf16|NULL (gradient of raw), packed_t =
 * the TRANSPOSED weights in the same fragment format (Wr^T, W1h^T, [W1x^T | W0x^T], Wd1^T, Wd0^T), m0/m1/m2 = the ReLU
 * masks ucn_train_fwd wrote.  With head (the same HOST float[4]) gy / graw are the fp32 gradients of rgb [M,3] /
 * density [M] and the activation derivatives are taken from the forward's outputs `density`, `rgb`.
 * Outputs: the pre-activation gradients the weight-gradient GEMMs need, d1, d0, gx
 * [M,256] and gh0 [M,64] as bf16, and the feature gradient gfeat [M,F] fp32.  F | UCN_GFEAT_LEVEL_MAJOR (ABI 26, F % 4 == 0, features in
 * pairs of level_dim 2): gfeat is written as [F / 2][M][2] with every value divided by 6 -- ucn_march_features_backward's layout 4, which
 * then neither copies nor divides it. */
#define UCN_GFEAT_LEVEL_MAJOR 0x10000
/* ... the same for level_dim 4 (the reference's own waymo.gin grid): [F / 4][M][4], / 6. */
#define UCN_GFEAT_LEVEL_MAJOR4 0x20000
int ucn_train_bwd(const void *gy, const void *graw, const float *head, const float *density, const float *rgb,
                  const void *packed_t, const uint32_t *m0, const void *m1, const void *m2, uint32_t N, uint32_t S,
                  uint32_t F, void *d1, void *d0, void *gx, void *gh0, void *dy /*[M, dy_ld] bf16 | NULL: the colour-logit
                  gradient consumed (columns 0-2) and the density head's gradient at the bottleneck (column 3), for the rgb layer's and
                  the bottleneck's weight gradients*/, uint32_t dy_ld /*0 = 4; 32: a zero-filled tile ucn_wgrad_bf16 takes as A*/,
                  float *gfeat, ucn_stream_t stream);

/* The PROPOSAL field's dense part in training (models.py:507-516 with disable_rgb: Linear(F,64) + ReLU, Linear(64,1),
 * softplus(raw + density_bias)), forward and backward, as VALU kernels (prop_train.hip) instead of ~45 library launches.
 * feat [M,F] fp32 (F <= 24), W0 [64,F], b0 [64], w1 [64], b1 [1] = the module's fp32 parameters.  round_bf16 != 0: operands
 * and layer outputs rounded to bf16 with fp32 accumulation (what accelerator.autocast() makes of these layers); 0: fp32.
 * bwd: density = the forward's output, g_density its gradient; gfeat [M,F] | NULL; gW0 / gb0 / gw1 / gb1 fp32, summed in a
 * fixed order (deterministic); workspace of ucn_prop_train_bwd_ws_floats(F, M) floats. */
int ucn_prop_train_fwd(const float *feat, uint32_t F, uint32_t hidden, const float *W0, const float *b0, const float *w1,
                       const float *b1, float density_bias, int round_bf16, uint64_t M, float *density,
                       uint32_t n_rays, uint32_t feat_level_dim /*0: feat [M,F], density [M]; C (inference): feat =
                       ucn_march_features' layout 2 ([L][B][C], b = s n_rays + ray), density [ray][sample]*/,
                       ucn_stream_t stream);
uint64_t ucn_prop_train_bwd_ws_floats(uint32_t F, uint64_t M);
int ucn_prop_train_bwd(const float *feat, uint32_t F, uint32_t hidden, const float *W0, const float *b0, const float *w1,
                       const float *b1, float density_bias, int round_bf16, uint64_t M, const float *density,
                       const float *g_density, float *gfeat, float *gW0, float *gb0, float *gw1, float *gb1, float *workspace,
                       ucn_stream_t stream);

/* ------------------------------------------------- ray generation
 * ref: camera_utils.py:448-557 pixels_to_rays (perspective pinhole, no distortion, no NDC) + :560-608
 * cast_ray_batch + datasets.py:421-447,476 (_make_ray_batch: cam_dirs, near/far/lossmult/cam_idx columns, the
 * final float32 cast).  Computed in float64 with the reference's operation order, rounded once at the store.
 * pix_x / pix_y DEVICE int32 [n_rays] or both NULL = every pixel of a width x height frame in 'xy' meshgrid order
 * (camera_utils.py:368-370, n_rays = width*height).  cam_idx DEVICE int32 [n_rays] or NULL = cam_idx_scalar.
 * pixtocams DEVICE double [n_cams,3,3] (inverse intrinsics), camtoworlds DEVICE double [n_cams,3,4].
 * Outputs float32: origins/directions/viewdirs/cam_dirs [n,3], radii [n,1], imageplane [n,2]|NULL, and the
 * broadcast columns near/far/lossmult/cam_idx [n,1] (each may be NULL). */
int ucn_generate_rays(const int32_t *pix_x, const int32_t *pix_y, const int32_t *cam_idx, int32_t cam_idx_scalar,
                      const double *pixtocams, const double *camtoworlds, uint32_t n_cams, uint32_t width,
                      uint32_t height, uint32_t n_rays, float near_, float far_, float *origins, float *directions,
                      float *viewdirs, float *radii, float *imageplane, float *cam_dirs, float *near_out,
                      float *far_out, float *lossmult_out, float *cam_idx_out, ucn_stream_t stream);

/* ------------------------------------------------- sky layer + colour correction
 * ref: models.py:326-337,743-904 (sky NeRF, 120 samples, 8x256 MLP) and
 * extrinsic_optimizer.py:4-48 + models.py:339-363 (per-camera 3x4 affine). */
typedef struct ucn_sky {
    const float *w_pts[8], *b_pts[8]; /* pts_linears.{0..7}; layer 5 is [256,259] */
    const float *w_alpha, *b_alpha, *w_feat, *b_feat, *w_view, *b_view, *w_rgb, *b_rgb;
    float *packed;
} ucn_sky_t;
uint64_t ucn_sky_packed_floats(void);
int ucn_sky_pack(const ucn_sky_t *s, ucn_stream_t stream);
uint64_t ucn_sky_workspace_floats(uint32_t N);
/* t_vals: DEVICE [120] = linspace(0,1,120) (models.py:870); far0_times_1p5 = 1.5*far[0] (:329);
 * workspace: DEVICE ucn_sky_workspace_floats(N) floats (per-ray view bias + per-sample raw outputs).
 * mixed = 0: fp32-class products (split-f16 MFMA, the parity path).  mixed = 1: what the reference's NeRF.forward is
 * under a bf16 autocast (models.py:957, :786-815): bf16 operands, fp32 accumulation, for the 256-wide layers; layer 0,
 * the two heads and the compositing stay fp32. */
int ucn_sky_render(const ucn_sky_t *s, const float *origins, const float *directions,
                   const float *cam_dirs, const float *far_ /*[N]*/, float far0_times_1p5,
                   const float *t_vals, uint32_t N, float *workspace, float *sky_rgb_out /*[N,3]*/,
                   int mixed, ucn_stream_t stream);

/* ---- training step of the sky layer (ref: models.py:326-337, :743-904 under train.py:165-171's bf16 autocast, and
 * autograd's way back).  The host passes the two 9-tile layers COMPOSED, as differentiable fp32 matrices it forms itself:
 *   m5 [256, 288] = [W5[:, 3:259] | W5[:, 0:3] | b5 | 0 (28)]
 *   mv [128, 288] = [W_view[:, :256] W_feat | 0, 0, 0 | b_view + W_view[:, :256] b_feat | W_view[:, 256:283] | 0]
 * (columns = the kernels' input tiles [h (256) | aux = (p (3), 1, embed(cam_dir) (27), 0)]).
 * ucn_sky_train_fwd writes, per sample b = ray * 120 + s: raw [M, 4] (colour logits, sigma), the bf16 activation buffer
 * act [M, ucn_sky_train_act_ld()] = h_0 .. h_7 (256 each) | aux (32) | hv (128), and the ReLU masks; sky_rgb_out [N, 3].
 * ucn_sky_train_bwd turns d loss / d sky_rgb [N, 3] into the bf16 pre-activation gradients grad [M, ucn_sky_train_grad_ld()]
 * = d0 .. d7 (256 each) | dv (128) | g (32: d logits, d sigma, 0 ...).  Weight gradients are then ucn_wgrad_bf16 passes:
 *   d_l^T [h_{l-1} | aux] = [dW_l | . | db_l at column 259 | .],  d0^T aux = [dW0 (3) | db0],  [dv | g]^T [h7 | aux] -> d mv,
 *   d w_alpha (row 131), g^T hv -> dW_rgb.   sky_far = 1.5 * far[0] is read on the device (no host sync). */
typedef struct ucn_sky_train {
    const float *w_pts[8], *b_pts[8]; /* pts_linears.{0..7}; entry 5 unused (m5) */
    const float *m5, *mv;
    const float *w_alpha, *b_alpha, *w_rgb, *b_rgb;
    void *packed;                     /* DEVICE, ucn_sky_train_packed_bytes() bytes */
} ucn_sky_train_t;
uint64_t ucn_sky_train_packed_bytes(void);
uint32_t ucn_sky_train_act_ld(void);
uint32_t ucn_sky_train_grad_ld(void);
int ucn_sky_train_pack(const ucn_sky_train_t *s, ucn_stream_t stream);
int ucn_sky_train_fwd(const void *packed, const float *origins, const float *directions, const float *cam_dirs,
                      const float *far_ /*[N]*/, const float *t_vals /*DEVICE [120]*/, uint32_t N,
                      float *aux_ws /*[N,32]*/, float *raw /*[N*120,4]*/, void *act, void *mask /*[8][N*120][2] uint4*/,
                      void *mask_v /*[N*120][2] uint2*/, float *sky_rgb_out /*[N,3]*/, ucn_stream_t stream);
int ucn_sky_train_bwd(const void *packed, const float *g_sky_rgb /*[N,3]*/, const float *raw, const float *directions,
                      const float *far_, const float *t_vals, uint32_t N, const void *mask, const void *mask_v,
                      float *g_raw_ws /*[N*120,4]*/, void *grad, ucn_stream_t stream);

/* Weight gradient of a dense layer over a training batch (ref: autograd through nn.Linear, models.py:438-483, :743-820):
 *   out[KA][kb1 + kb2] (fp32) = A^T [B1 | B2],  A = pre-activation gradients [M, lda] bf16 (KA columns from A),
 *   B1 / B2 = column blocks of the layer's inputs [M, ldb*] bf16 (B2 optional: e.g. the tile holding the constant 1 whose
 *   column is the bias gradient).  KA, kb1, kb2 multiples of 32, KA <= 256, kb1 + kb2 <= 288; row strides multiples of 8.
 * Split-K over the samples with a fixed-order reduction (deterministic); workspace: ucn_wgrad_ws_floats(KA, KB, M) floats. */
uint64_t ucn_wgrad_ws_floats(uint32_t KA, uint32_t KB, uint64_t M);
int ucn_wgrad_bf16(const void *A, uint32_t lda, uint32_t KA, const void *B1, uint32_t ldb1, uint32_t kb1, const void *B2,
                   uint32_t ldb2, uint32_t kb2, uint64_t M, float *workspace, float *out, ucn_stream_t stream);

/* ---- fp32 dense layers of the NON-autocast training step (r04; csrc/gemm_f32.hip): exact fp32 products on v_mfma_f32_32x32x2_f32.
 * The reference's shipped launch trains in fp32 (scripts/train_waymo.sh:3, train.py:165): every nn.Linear of the NeRF field
 * (internal/models.py:438-483, 581-674), the sky NeRF (models.py:743-820) and the colour-correction head
 * (internal/extrinsic_optimizer.py:4-48) is `F.linear` forward and two GEMMs backward.  These replace the library GEMMs of that route.
 * Row-major fp32 operands with leading dimensions (column slices of wider buffers are passed as views).
 * ucn_gemm_f32:  Y[M, N] = (Y if UCN_GEMM_ACCUMULATE) + X[M, K] W[N, K]^T + bias[N] (may be NULL), then ReLU if UCN_GEMM_RELU.
 *   = torch.nn.functional.linear(X, W, bias); the input gradient d X = d Y W is the same call on the transposed weight.
 *   K, ldx, ldw multiples of 4, X and W 16-byte aligned (16-byte operand loads); any M, N. */
#define UCN_GEMM_ACCUMULATE 1
#define UCN_GEMM_RELU 2
#define UCN_GEMM_MASK 4
int ucn_gemm_f32(const float *X, uint32_t ldx, const float *W, uint32_t ldw, const float *bias, uint32_t M, uint32_t N, uint32_t K,
                 int flags, float *Y, uint32_t ldy, ucn_stream_t stream);
/* ucn_gemm_f32_ex (r05): the same product with two more epilogue terms, so that a chain of Linear + ReLU layers needs no elementwise pass:
 *   rowbias != NULL: + rowbias[row / rgroup][N] (ldr floats per row) before the ReLU -- a per-RAY term under a per-sample GEMM (the view
 *     branch of the sky NeRF: the direction encoding is the same for a ray's 120 samples, models.py:796-806);
 *   UCN_GEMM_MASK: Y = mask[M, N] > 0 ? Y : 0 as the last step -- the ReLU derivative of the layer BELOW (mask = that layer's stored
 *     output) fused into the d X GEMM that produces its output gradient (autograd's threshold_backward pass over [M, N] disappears:
 *     3 x M x N x 4 bytes per layer). */
int ucn_gemm_f32_ex(const float *X, uint32_t ldx, const float *W, uint32_t ldw, const float *bias, uint32_t M, uint32_t N, uint32_t K,
                    int flags, float *Y, uint32_t ldy, const float *mask, uint32_t ldm, const float *rowbias, uint32_t ldr,
                    uint32_t rgroup, ucn_stream_t stream);
/* ucn_wgrad_f32: GW[N, K] = GY[M, N]^T X[M, K] (the weight gradient of the layer above) and, if gb != NULL, gb[N] = column sums of GY
 *   (its bias gradient), reduction over the M samples in fixed-order partial sums (deterministic).  `ws`: ucn_wgrad_f32_ws_floats floats.
 *   N, K, ldg, ldx multiples of 4, GY and X 16-byte aligned. */
uint64_t ucn_wgrad_f32_ws_floats(uint32_t N, uint32_t K, uint64_t M);
int ucn_wgrad_f32(const float *GY, uint32_t ldg, const float *X, uint32_t ldx, uint32_t M, uint32_t N, uint32_t K, float *ws, float *GW,
                  float *gb, ucn_stream_t stream);

/* ---- the same dense layers on the split-f16 MFMA engine (r06; csrc/gemm_h3.hip): "fp32-class" products, x w ~ x_hi w_hi + x_hi w_lo +
 * x_lo w_hi with hi = f16(x), lo = f16(x - hi) (three v_mfma_f32_32x32x16_f16 per 16 k against eight fp32 MFMAs; fp32 accumulation).
 * Replaces the same reference code as ucn_gemm_f32 / ucn_wgrad_f32 (F.linear forward and its two backward GEMMs: models.py:438-483,
 * 581-674, 743-820, extrinsic_optimizer.py:4-48 under scripts/train_waymo.sh:3's fp32 launch); the exact-fp32 kernels stay behind
 * the host-side switch.  Every operand carries a power-of-two scale derived from its absolute maximum, a DEVICE float:
 *   ucn_amax_f32:  *slot = max(*slot, max |X[M, K]|) (atomic on the bit pattern; zero the slot first) for operands no kernel here made;
 *   ucn_pack_h3:   W[N, K] (ldw floats per row; transposed != 0: the operand is given as [K, N] and used as its transpose) -> the packed
 *                  A-operand stream (ucn_pack_h3_bytes(N, K) bytes, 16-byte aligned), *wmax_out = max |W|;  N <= 256;
 *   ucn_gemm_h3:   ucn_gemm_f32_ex's product and epilogue (flags, mask, rowbias as there) from X, the packed stream and the two maxima;
 *                  ymax != NULL: *ymax = max(*ymax, max |Y|) (the xmax of the GEMM that consumes Y);  K, ldx multiples of 4, N <= 256;
 *   ucn_wgrad_h3:  ucn_wgrad_f32's result from GY, X and their maxima (gb: exact fp32 column sums); ws: ucn_wgrad_h3_ws_floats floats. */
int ucn_amax_f32(const float *X, uint32_t ldx, uint64_t M, uint32_t K, float *slot, ucn_stream_t stream);
uint64_t ucn_pack_h3_bytes(uint32_t N, uint32_t K);
int ucn_pack_h3(const float *W, uint32_t ldw, uint32_t N, uint32_t K, int transposed, void *packed, float *wmax_out, ucn_stream_t stream);
int ucn_gemm_h3(const float *X, uint32_t ldx, const void *packed, const float *xmax, const float *wmax, const float *bias, uint32_t M,
                uint32_t N, uint32_t K, int flags, float *Y, uint32_t ldy, const float *mask, uint32_t ldm, const float *rowbias,
                uint32_t ldr, uint32_t rgroup, float *ymax, ucn_stream_t stream);
/* ucn_gemm_h3_x2: the same with a second, 4-wide operand pair added in the epilogue (exact fp32 FMAs) before the ReLU / mask:
 *   Y = ... + X2[M, 4] W2[N, 4]^T -- the [hidden | 3-d point] input of the sky NeRF's skip layer (models.py:790-795) and the
 *   [view branch | density row] gradient into its last trunk layer (models.py:800-806) as ONE pass over the [M, 256] output. */
int ucn_gemm_h3_x2(const float *X, uint32_t ldx, const void *packed, const float *xmax, const float *wmax, const float *bias, uint32_t M,
                   uint32_t N, uint32_t K, int flags, float *Y, uint32_t ldy, const float *mask, uint32_t ldm, const float *rowbias,
                   uint32_t ldr, uint32_t rgroup, const float *X2, uint32_t ldx2, const float *W2, uint32_t ldw2,
                   uint64_t *relu_bits_out, const uint64_t *mask_bits, float *ymax, ucn_stream_t stream);
/* ReLU derivatives as bit masks (128- / 256-wide outputs): relu_bits_out != NULL: the call leaves "Y > 0" of every output as one bit
 * (ucn_relu_bits_words(M, N) 64-bit words; 32-bit word ((row / 32) * (N / 64) + column / 64) * 64 + lane, bit 4 u + j = row 32 (row / 32) + 4 u + lane / 16, column 64 (column / 64) + 4 (lane % 16) + j: the epilogue's own store order); mask_bits != NULL: Y = bit ? Y : 0 as the last step, from
 * the bits a forward call of the same [M, N] shape left -- the d X GEMM of a Linear + ReLU layer without reading the layer's stored
 * fp32 output (UCN_GEMM_MASK's 4 bytes per element become 1 bit). */
uint64_t ucn_relu_bits_words(uint64_t M, uint32_t N);
uint64_t ucn_wgrad_h3_ws_floats(uint32_t N, uint32_t K, uint64_t M);
int ucn_wgrad_h3(const float *GY, uint32_t ldg, const float *X, uint32_t ldx, const float *gmax, const float *xmax, uint32_t M, uint32_t N,
                 uint32_t K, float *ws, float *GW, float *gb, ucn_stream_t stream);

/* Iso-surface extraction from a dense lattice of values on the device (ref: skimage.measure.marching_cubes as called by
 * extract.py:379-383, :420-460 and tsdf.py:98-102): volume [X][Y][Z] float32 (z fastest), inside = value < level.
 * Two calls around one host read of the two counts (the outputs have to be allocated):
 *   ucn_marching_cubes_count -> counts_out[0] = vertices, [1] = triangles (DEVICE); workspace ucn_marching_cubes_ws_bytes(X,Y,Z)
 *   ucn_marching_cubes_emit  -> verts [nv,3] = lattice coordinates x spacing, normals [nv,3] (unit gradient; NULL to skip),
 *                               faces [nt,3] int32 into verts (NULL to skip); same volume / level / workspace as the count call.
 * Shared vertices, deterministic order (vertices by owning lattice point then axis, triangles by cell then table order). */
uint64_t ucn_marching_cubes_ws_bytes(uint32_t X, uint32_t Y, uint32_t Z);
int ucn_marching_cubes_count(const float *volume, uint32_t X, uint32_t Y, uint32_t Z, float level, void *workspace,
                             uint32_t *counts_out, ucn_stream_t stream);
int ucn_marching_cubes_emit(const float *volume, uint32_t X, uint32_t Y, uint32_t Z, float level, float sx, float sy, float sz,
                            void *workspace, float *verts, float *normals, int32_t *faces, ucn_stream_t stream);

/* PSNR and SSIM of a rendered frame against the ground truth with the reference's conventions (ref: internal/image.py:114-133
 * MetricHarness: uint8 quantisation, skimage PSNR with data_range 255, skimage SSIM defaults on OpenCV's 8-bit grey images):
 * pred, gt [H, W, 3] float32 in [0, 1] (DEVICE) -> out[0] = psnr (dB), out[1] = ssim, out[2] = mse on the uint8 scale (DEVICE doubles). */
uint64_t ucn_image_metrics_ws_bytes(uint32_t H, uint32_t W);
int ucn_image_metrics(const float *pred, const float *gt, uint32_t H, uint32_t W, void *workspace, double *out, ucn_stream_t stream);

/* generic small dense layer y = act(x W^T + b), used for the brightness MLP (4->256->256->256->12) */
int ucn_dense(const float *x /*[M,K]*/, const float *w /*[Nout,K]*/, const float *b, uint32_t M,
              uint32_t K, uint32_t Nout, int relu, float *y /*[M,Nout]*/, ucn_stream_t stream);
/* rgb' = A[idx] rgb + b[idx] (+ (1-acc_last) (A_sky sky + b_sky)); ref: models.py:350-354 */
int ucn_apply_affine(const float *rgb_in /*[N,3]*/, const float *affine /*[M,12]*/,
                     const int64_t *ray_to_row /*[N]|NULL = row 0*/, const float *weights_last /*[N,S]|NULL*/,
                     uint32_t S, const float *sky_rgb /*[N,3]|NULL*/, const float *affine_sky /*[M,12]|NULL*/,
                     uint32_t N, float *rgb_out /*[N,3]*/, ucn_stream_t stream);

/* ---- the tail of the TRAINING step with the colour-correction head and the sky layer on (r03; csrc/heads_train.hip).
 * Each call is one launch; g == NULL selects the forward, g != NULL the backward of the same entry point.
 *
 * ucn_affine_blend: rgb' = A rgb + t (+ (1 - acc_last) (A_sky sky + t_sky)) with PER-RAY affine maps [N,12] (row-major [3,4] = [A | t];
 * models.py:339-363, the training form where every ray carries its camera's map).  Forward (g_out NULL): out_or_g_rgb = rgb'.
 * Backward: out_or_g_rgb = d/d rgb; g_affine / g_acc / g_sky / g_affine_sky receive (accumulate = 0) or add up (accumulate = 1:
 * the same maps, sky colours and acc serve every level's call) the other gradients. */
int ucn_affine_blend(const float *g_out /*[N,3]|NULL*/, const float *rgb /*[N,3]*/, const float *affine /*[N,12]*/,
                     const float *acc_last /*[N]|NULL*/, const float *sky_rgb /*[N,3]|NULL*/, const float *affine_sky /*[N,12]|NULL*/,
                     uint32_t N, int accumulate, float *out_or_g_rgb /*[N,3]*/, float *g_affine /*[N,12]*/, float *g_acc /*[N]*/,
                     float *g_sky /*[N,3]*/, float *g_affine_sky /*[N,12]*/, ucn_stream_t stream);
/* ucn_data_loss (train_utils.py:171-230): per level l mse_l = sum(m r^2) / sum(m), charb_l = sum(m sqrt(r^2 + pad^2)) / sum(m), r = rgb_l -
 * target, m = lossmult per ray (NULL = 1); fwd_out = [L][2] {mse, charb}, sum(m) at [2 L], loss = sum_l w_mse[l] mse_l + w_charb[l] charb_l
 * at [2 L + 1].  Backward (g = d / d loss, [1] on the device): g_rgb_levels[l] = d loss / d rgb_l.  L <= 4; fixed-order sums. */
int ucn_data_loss(const float *const *rgb_levels_host /*[L] device pointers, each [N,3]*/, uint32_t L, const float *w_mse_host /*[L]*/,
                  const float *w_charb_host /*[L]*/, const float *target /*[N,3]*/, const float *lossmult /*[N]|NULL*/, uint32_t N,
                  float charb_padding, float *fwd_out /*[2 L + 2]*/, const float *g /*[1]|NULL*/, float *const *g_rgb_levels_host,
                  ucn_stream_t stream);
/* ucn_sky_loss (train_utils.py:149-157): sum_l mean BCE(clip(acc_l, 1e-3, 0.999), 1 - sky_segs) */
int ucn_sky_loss(const float *const *acc_levels_host /*[L] device pointers, each [N]*/, uint32_t L, const float *sky_segs /*[N]*/, uint32_t N,
                 float *loss_out /*[1]*/, const float *g /*[1]|NULL*/, float *const *g_acc_levels_host, ucn_stream_t stream);
/* ucn_identity_loss (train_utils.py:159-169): mean over [N,3,4] of |eye - A| (+ |eye - A_sky|), accumulated in float64 like the reference */
int ucn_identity_loss(const float *affine /*[N,12]*/, const float *affine_sky /*[N,12]|NULL*/, uint32_t N, double *loss_out /*[1]*/,
                      const double *g /*[1]|NULL*/, float *g_affine /*[N,12]*/, float *g_affine_sky /*[N,12]|NULL*/, ucn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UCNERF_MARCH_H */
