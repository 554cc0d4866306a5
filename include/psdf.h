/* psdf.h -- C ABI of libpsdf_hip.so, the MI355X (gfx950) implementation of the PermutoSDF rendering/training hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Each entry point replaces one
 * operator of the reference's Python-facing API -- the pybind11 module `permuto_sdf` (src/PyBridge.cxx:36-166) and
 * the external `permutohedral_encoding` package -- and cites the reference interface it stands in for.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 / int32 / 1-byte-bool data unless marked "host";
 *   - the callee never allocates and never synchronises: outputs are caller-allocated (zero-filled where a comment
 *     says "accumulated into" or where the reference allocates with torch::zeros), launches are asynchronous on `stream`
 *     (a hipStream_t; pass the framework's current stream, NULL = default stream);
 *   - return value: 0 = ok, -1 = argument error, -2 = unsupported configuration, > 0 = hipError_t of the launch;
 *   - ray-index arguments (nr_rays, ray_start_end_idx [R,2] int32, rays_have_equal_nr_of_samples,
 *     fixed_nr_of_samples_per_ray, max_nr_samples) mirror RaySamplesPacked (include/permuto_sdf/RaySamplesPacked.cuh:6-46);
 *   - PCG32 generators are passed by value as (state, inc) exactly as the reference passes `pcg32 rng` to its kernels;
 *     the owner advances its host copy by 2^32 after every jittered call (src/OccupancyGrid.cu:252-254).
 */
#ifndef PSDF_H
#define PSDF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- encode.hip ---- */
/* The frozen conventions of the encoding (upstream source absent, PARITY UNPINNED) are the #defines of
   permuto_sdf_amd/csrc/encode_conventions.h; `concat_points` below is one of PSDF_ENC_CONCAT_NONE (0),
   PSDF_ENC_CONCAT_PSEUDO_LEVELS (1: F*(L + ceil(P/F)) channels, zero padded) or PSDF_ENC_CONCAT_APPEND (2: F*L + P channels,
   `cat([sliced, scaling * points])`, what permuto_sdf_py/models/models.py:149,154 consumes through output_dims()).
   psdf_encode_convention(i) returns the value IN FORCE of convention i (0 hash multiplier, 1 rank tie rule -- the two that
   live in device code, runtime values -- 2 sqrt term of scale_factor, 3 inverse-std-dev term, 4 default concatenation layout --
   host-side defaults: `scale_factor` and `concat_points` are arguments of every entry point); host only.
   psdf_encode_set_conventions() replaces the two device-side ones for the process (kernel arguments from then on), so that
   matching an upstream build that disagrees with the recollection is a flag flip, not a rebuild (INTEGRATION.md,
   tools/dump_upstream_encoding_vectors.py, tests/test_upstream_vectors.py).  No reference counterpart. */
int64_t psdf_encode_convention(int which);
int psdf_encode_set_conventions(uint32_t hash_multiplier, int rank_tie_raises_later);

/* replaces: permutohedral_encoding CUDA op `forward_gpu` (un-vendored; call sites permuto_sdf_py/models/models.py:186,370,500,542)
   -2 (unsupported) when one level of the table exceeds 4 GiB (capacity * nr_feat * 4 bytes: the kernel gathers with 32-bit
   offsets from the level's base; the reference's tables are 2 MiB per level) */
int psdf_encode_forward(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
    const float* lattice, const float* scale_factor, const float* shifts, const float* window, int concat_points,
    float points_scaling, float* sliced, void* stream);

/* same operator; additionally sets touched_blocks[level][row >> block_rows_log2] = 1 for every table row the batch reads
   (bytes, [nr_levels, ceil(capacity / 2^block_rows_log2)], never cleared here): what psdf_adamw_step_blocks consumes */
int psdf_encode_forward_mark(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
    const float* lattice, const float* scale_factor, const float* shifts, const float* window, int concat_points,
    float points_scaling, float* sliced, unsigned char* touched_blocks, int block_rows_log2, void* stream);

/* replaces: permutohedral_encoding `backward_gpu` / `backward_gpu_only_pos` (autograd of models.py:186; positions grad needed by models.py:240-251) */
int psdf_encode_backward(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
    const float* lattice, const float* scale_factor, const float* shifts, const float* window, int concat_points,
    float points_scaling, const float* grad_sliced, float* grad_lattice, float* grad_positions, void* stream);

/* same operator with caller-provided device scratch: enables the binned queue + LDS-reduction lattice-gradient path
   for large batches (psdf_encode_backward_workspace_bytes() == 0 means "not applicable", pass NULL) */
int64_t psdf_encode_backward_workspace_bytes(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity);
int psdf_encode_backward_ws(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
    const float* lattice, const float* scale_factor, const float* shifts, const float* window, int concat_points,
    float points_scaling, const float* grad_sliced, float* grad_lattice, float* grad_positions, void* workspace,
    int64_t workspace_bytes, void* stream);
/* debug query: how the last balanced binning launch of psdf_encode_backward_ws dealt its resident round of workgroups over the
   levels (counts[0 .. nr_levels)); returns nr_levels, 0 when no balanced launch has run.  Shares follow the measured duration
   of each level's workgroups in the previous call (PSDF_ENC_BWD_BALANCE=0: equal shares); closed levels fall to the minimum.
   REPRODUCIBILITY: the deal -- and with it the order of the float additions of the lattice gradient -- depends on the timing of
   the previous call; PSDF_ENC_BWD_BALANCE=0 is the switch for run-to-run comparisons (state is kept per device, lattice and
   shape class).  NON-FINITE INPUTS: contributions of neighbouring samples to one table row are merged in registers with
   multiply-adds (encode.hip, combine_runs16), so ONE non-finite upstream gradient makes the gradient of up to 15 other table
   rows handled by the same 16 lanes NaN as well -- a NaN batch is a NaN batch, but more rows show it than carry it. */
int psdf_encode_backward_level_shares(int* counts, int max_levels);


/* replaces: permutohedral_encoding `double_backward_from_positions_gpu` (create_graph=True at models.py:245-251) */
int psdf_encode_double_backward(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float*
    positions, const float* lattice, const float* scale_factor, const float* shifts, const float* window, int
    concat_points, float points_scaling, const float* dd_positions, const float* grad_sliced, float* grad_lattice,
    float* grad_grad_sliced, void* stream);
/* Same, with the backward's device workspace (psdf_encode_backward_workspace_bytes; NULL = none): large batches send the
 * lattice scatter through the binning + reduce kernels instead of float atomics.  grad_grad_sliced may be NULL when only the
 * lattice gradient is wanted.  grad_sliced_direct (optional, [C, N]): the lattice scatter of the PLAIN backward for this upstream
 * gradient is added in the same pass (grad_lattice += d<sliced, grad_sliced_direct>/d lattice): a training step sends both onto
 * the same rows. */
int psdf_encode_double_backward_ws(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
                                   const float* lattice, const float* scale_factor, const float* shifts, const float* window,
                                   int concat_points, float points_scaling, const float* dd_positions, const float* grad_sliced,
                                   float* grad_lattice, float* grad_grad_sliced, const float* grad_sliced_direct,
                                   void* workspace, int64_t workspace_bytes, void* stream);

/* ---- mlp.hip, opt-in arithmetic ---- */
/* psdf_mlp_pack / psdf_mlp_forward with TWO fp16 pieces per fp32 operand (three products on v_mfma_f32_32x32x16_f16) instead of
   three bf16 pieces (six products): the BASELINE net only (dims = {<= 64, 64, 64, 64, <= 4}; -2 otherwise); max error ~3e-6 of
   the largest output instead of ~1e-6; inputs, activations and weights must stay below 65504 in magnitude.  A buffer made by
   psdf_mlp_pack_f16 is consumed by psdf_mlp_forward_f16 only. */
int psdf_mlp_pack_f16(int n_layers, const int* dims, const float* const* weights, const float* const* biases, float* packed,
    void* stream);
int psdf_mlp_forward_f16(int n_layers, const int* dims, int64_t N, const float* X, const float* packed, float* Y, void* stream);

/* ---- composite_fused.hip ---- */
/* replaces, fused: VolumeRenderingNeus.compute_weights + integrate (permuto_sdf_py/volume_rendering/volume_rendering_modules.py:
   129-190), i.e. the chain psdf_neus_alpha_forward -> psdf_cumprod_alpha2transmittance -> (alpha * T) ->
   psdf_integrate_with_weights in ONE launch: pred [R,3] (every ray written: 0 for invalid / empty rays), bg [R] (optional; 1 for
   such rays), weights [N] (optional).  Same arithmetic and summation order as the separate entry points. */
int psdf_neus_composite_forward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float* sdf,
    const float* dirs, const float* gradients, const float* dt, const float* rgb, const float* inv_s, float cos_anneal_ratio,
    float* pred, float* bg, float* weights, void* stream);
/* its backward in one launch (integrate_with_weights_backward, the transmittance backward with its inverse cumulative sum,
   the opacity backward: volume_rendering_funcs.py:55-190): grad_pred [R,3], grad_bg [R] or NULL -> grad_sdf [N]; optional
   grad_gradients [N,3], grad_rgb [N,3], grad_inv_s [1] (accumulated into).  max_per_ray = an upper bound of the samples of any
   ray, at most 256 (-2 beyond: use the per-operator entry points); reference_compat as in psdf_integrate_with_weights_backward. */
int psdf_neus_composite_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, int max_per_ray,
    const float* grad_pred, const float* grad_bg, const float* sdf, const float* dirs, const float* gradients, const float* dt,
    const float* rgb, const float* inv_s, float cos_anneal_ratio, int reference_compat, float* grad_sdf, float* grad_gradients,
    float* grad_rgb, float* grad_inv_s, void* stream);

/* Background NeRF (NerfHash + VolumeRenderingNerf.compute_weights + integrate: models.py:520, volume_rendering_modules.py:72-86,
   176-190) and the composition with the foreground (train_permuto_sdf.py:160-165), one launch per direction:
   density = softplus(raw_density); alpha = 1 - exp(-density dt); T = exclusive cumprod(1 - alpha + 1e-7); w = alpha T;
   pred_bg [R,3] = sum w rgb; with fg_pred [R,3] and fg_bg [R] (both or neither) also pred [R,3] = fg_pred + fg_bg * pred_bg. */
int psdf_nerf_composite_forward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float* raw_density,
                                const float* dt, const float* rgb, const float* fg_pred, const float* fg_bg, float* pred_bg,
                                float* pred, void* stream);
/* its backward for grad_pred [R,3] (of the composed radiance when fg_bg is given, else of pred_bg): grad_raw_density [M], grad_rgb
   [M,3], optional grad_fg_bg [R] = <grad_pred, pred_bg>; rays of at most max_per_ray <= 256 samples (-2 beyond) */
int psdf_nerf_composite_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, int max_per_ray,
                                 const float* grad_pred, const float* fg_bg, const float* raw_density, const float* dt,
                                 const float* rgb, int reference_compat, float* grad_raw_density, float* grad_rgb,
                                 float* grad_fg_bg, void* stream);

/* ---- debug query (mlp_bwd.hip) ---- */
/* Which kernel variant the LAST call of an operator family dispatched to (host only, no device work): lets a parity test
   assert that the configuration it compares with the oracle ran the kernels the benchmark times.  No reference counterpart.
     family 0 encode backward: 1 LDS scatter cache + atomics, 2 queue mode (binning + encode_bwd_reduce_kernel), 3 positions only
     family 1 MLP backward   : 1 fp32-MFMA kernel, 2 split-bf16 kernel (mlp_bwd_split.hip), 3 wide workgroup kernel,
                               4 split-fp16 kernel (mlp_bwd_split_f16.hip)
     family 2 MLP forward    : 1 fp32-MFMA kernel, 2 split-bf16 kernel, 3 split-fp16 kernel (psdf_mlp_forward_f16)
   0 = no call yet, -1 = unknown family. */
int psdf_last_path(int family);

/* ---- mlp_bwd_split.hip ---- */
/* same contract as psdf_mlp_backward for dims = {K0 <= 52, 64, 64, 64, 1} with dW/db requested, computed on the bf16 matrix
   pipe with split fp32 operands (three bf16 pieces, six products kept: fp32-level accuracy); -2 for every other net and when
   stream-ordered scratch is unavailable (stream capture).  psdf_mlp_backward routes large batches here by itself. */
int psdf_mlp_backward_split(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
    const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db, void* stream);

/* ---- mlp_bwd_split_f16.hip ---- */
/* same contract as psdf_mlp_backward for dims = {K0 <= 64, 64, 64, 64, 1} with dW/db requested, computed on the fp16 matrix pipe
   with TWO fp16 pieces per fp32 operand (three products in the chains, four in the dW products); the gradient chain of every
   sample runs on the mantissa of its dY and the factor 2^e is restored exactly, so the accuracy does not depend on the size or
   spread of dY (errors of a few 1e-6 of the largest entry against float64; activations / weights must stay below 65504 in
   magnitude).  -2 for every other net and when stream-ordered scratch is unavailable.  psdf_mlp_backward routes large batches
   here by default (PSDF_MLP_BWD_SPLIT=bf16 selects psdf_mlp_backward_split instead). */
int psdf_mlp_backward_split_f16(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
    const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db, void* stream);
/* 1 once psdf_mlp_backward_split_f16 has launched mlp_bwd_split_f16_kernel (one wave per SIMD: the only form built since round 6;
   the two two-waves-per-SIMD forms of round 5 were slower and live in attic/rejected/), 0 = none yet.  Debug query (host only). */
int psdf_mlp_backward_split_f16_form(void);
/* range guard of the split-fp16 backward: inputs / hidden activations of magnitude >= 255 or weights >= 65504 leave the range
   of its two-piece arithmetic; such a batch is redone on the device by the three-piece bf16 kernel (K0 <= 52; no host
   round trip) and counted here -- the number of launches that raised the guard so far (exact after a synchronisation).
   The first such event also prints one line to stderr at the next call. */
unsigned psdf_mlp_f16_range_events(void);

/* ---- mlp_wide.hip ---- */
/* (also, round 6: the background density / feature net 52 -> 64 x 3 -> 65, models.py:451-459: dims[0] <= 64, 32 < dims[1..3] <= 64,
   16 < dims[4] <= 80)
   Forward of the reference's colour network shape (LipshitzMLP 111 -> 128 -> 128 -> 64 -> 3, models.py:54-129,349-350: dims[0] <= 112,
   dims[1], dims[2] <= 128, dims[3] <= 64, dims[4] <= 16) on the fp16 matrix pipe with two pieces per fp32 operand: X [dims[0], N],
   Y [dims[4], N] feature-major; weights[l] [dims[l+1], dims[l]] (for a LipshitzMLP the NORMALISED weights), biases[l]; GELU between
   the layers, the last one linear.  -2: another shape, stream capture, PSDF_MLP_WIDE_SPLIT=f32, or a value beyond the fp16 range met
   earlier (psdf_mlp_forward evaluates every shape with fp32 MFMAs). */
int psdf_mlp_forward_wide_f16(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
    const float* const* biases, float* Y, void* stream);
/* which kernel the last psdf_mlp_backward_wide launched: 1 = fp32 MFMAs (mlp_wide_bwd_kernel), 2 = two fp16 pieces per operand on
   the fp16 matrix pipe (mlp_wide_bwd_f16_kernel, the default since round 6; PSDF_MLP_WIDE_SPLIT=f32 selects the other, and so does a
   value beyond the fp16 range met by an earlier launch); 0 = none yet.  Debug query (host only). */
int psdf_mlp_backward_wide_form(void);
/* same contract as psdf_mlp_backward for 4-layer nets wider than one wave's register file holds (dims[0] <= 112, dims[1],
   dims[2] <= 128, dims[3] <= 64, dims[4] <= 16): the colour network LipshitzMLP 111 -> 128 -> 128 -> 64 -> 3 of
   permuto_sdf_py/models/models.py:54-129,349-350 (weights = the already normalised ones).  Workgroup-cooperative: 8 waves
   share a 32-sample tile through LDS, each owns one output tile per layer.  -2 for other widths / no stream-ordered scratch;
   psdf_mlp_backward falls through to it.  Also (split-fp16 kernel only): the 64-wide nets with many outputs (background density
   net 52 -> 64 x 3 -> 65, models.py:451-459) and, with n_layers = 3, the background colour head 80 -> 64 -> 64 -> 3
   (models.py:463-469: 64 < dims[0] <= 80, 32 < dims[1], dims[2] <= 64, dims[3] <= 16). */
int psdf_mlp_backward_wide(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
    const float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db, void* stream);

/* replaces: LipshitzMLP.normalization (models.py:98-104): Wn[r][:] = W[r][:] * min(1, softplus(c[0]) / sum_j |W[r][j]|), and its
   autograd (grad_W written, grad_c[0] accumulated into) */
int psdf_lipshitz_normalize_forward(int out, int in, const float* W, const float* c, float* Wn, void* stream);
int psdf_lipshitz_normalize_backward(int out, int in, const float* W, const float* c, const float* grad_Wn, float* grad_W,
    float* grad_c, void* stream);
/* all layers of a LipshitzMLP (n_layers <= 8) in one launch per direction: W[l] [out[l], in[l]], c[l] [1] on the device */
int psdf_lipshitz_normalize_forward_multi(int n_layers, const int* out, const int* in, const float* const* W, const float* const* c,
                                          float* const* Wn, void* stream);
int psdf_lipshitz_normalize_backward_multi(int n_layers, const int* out, const int* in, const float* const* W, const float* const* c,
                                           const float* const* grad_Wn, float* const* grad_W, float* const* grad_c, void* stream);

/* ---- neus.hip ---- */
/* replaces: the torch elementwise chain of VolumeRenderingNeus.compute_weights, permuto_sdf_py/volume_rendering/
   volume_rendering_modules.py:129-172 (cos anneal, section-point SDFs, two sigmoids, (p+1e-5)/(c+1e-5) clipped to [0,1]);
   inv_s is a DEVICE pointer to one float (the clipped exp(10*variance) of SingleVarianceNetwork, :96-115), samples are the
   packed [N,1]/[N,3] tensors of RaySamplesPacked; one_minus_alpha (optional) = 1 - alpha + 1e-7, the transmittance input */
int psdf_neus_alpha_forward(int64_t N, const float* sdf, const float* dirs, const float* gradients, const float* dt,
    const float* inv_s, float cos_anneal_ratio, float* alpha, float* one_minus_alpha, void* stream);
/* replaces: torch autograd of the same chain.  grad_gradients [N,3] and grad_inv_s [1] are optional; grad_inv_s is
   ACCUMULATED into (zero it first) */
int psdf_neus_alpha_backward(int64_t N, const float* grad_alpha, const float* sdf, const float* dirs, const float*
    gradients, const float* dt, const float* inv_s, float cos_anneal_ratio, float* grad_sdf, float* grad_gradients,
    float* grad_inv_s, void* stream);
/* replaces: rgb_loss, permuto_sdf_py/utils/permuto_sdf_utils.py:43-47: loss[0] += scale * sum |gt - pred| * mask[ray]
   (mask: optional [R] bytes), grad_pred (optional) = scale * sign(pred - gt) * mask */
int psdf_l1_loss(int64_t R, int C, const float* pred, const float* gt, const unsigned char* mask, float scale,
    float* loss, float* grad_pred, void* stream);
/* replaces: eikonal_loss, permuto_sdf_utils.py:49-51: loss[0] += scale * sum (|g| - 1)^2, grad (optional) [N,3] */
int psdf_eikonal_loss(int64_t N, const float* gradients, float scale, float* loss, float* grad_gradients, void* stream);
/* replaces: F.normalize(x, dim=-1) and its autograd backward (torch), used on the SDF gradients at
   permuto_sdf_py/models/models.py:272,280,367: grad_y == NULL -> out = x / max(|x|, 1e-12); else out = d/dx applied to grad_y */
int psdf_normalize3(int64_t N, const float* x, const float* grad_y, float* out, void* stream);
/* sigmoid at the end of the colour heads (models.py:386,525), fused with the layout change the compositing operators need:
   y [N, C] = sigmoid(x_fm [C, N]);  grad_x_fm [C, N] = grad_y [N, C] * y * (1 - y);  C <= 16 */
int psdf_sigmoid_rows(int64_t N, int C, const float* x_fm, float* y, void* stream);
int psdf_sigmoid_rows_backward(int64_t N, int C, const float* grad_y, const float* y, float* grad_x_fm, void* stream);
/* replaces: the shifted points of the curvature loss, models.py:266-277: out = points + epsilon * cross(normalize(gradients),
   normalize(rand_directions)) (grad_shifted == NULL), or the gradient of that w.r.t. `gradients` applied to grad_shifted */
int psdf_curvature_shift(int64_t N, const float* points, const float* gradients, const float* rand_directions, float
    epsilon, const float* grad_shifted, float* out, void* stream);
/* replaces: models.py:280-289 + the mean at train_permuto_sdf.py:363: loss[0] += scale * sum acos(clamp(n(g) . n(g_shifted),
   -1+1e-6, 1-1e-6)) / pi; the two gradient outputs are optional (both or none) */
int psdf_curvature_loss(int64_t N, const float* gradients, const float* gradients_shifted, float scale, float* loss,
    float* grad_gradients, float* grad_gradients_shifted, void* stream);
/* replaces: the off-surface loss, train_permuto_sdf.py:372-375: loss[0] += scale * sum exp(-sharpness |sdf|) */
int psdf_offsurface_loss(int64_t N, const float* sdf, float sharpness, float scale, float* loss, float* grad_sdf, void*
    stream);
/* replaces: NerfHash density activation + VolumeRenderingNerf alpha, models.py:520 (softplus) and
   volume_rendering_modules.py:72-86: alpha = 1 - exp(-softplus(raw) dt), one_minus_alpha = 1 - alpha + 1e-7 */
int psdf_nerf_alpha_forward(int64_t N, const float* raw_density, const float* dt, float* alpha, float* one_minus_alpha,
    void* stream);
int psdf_nerf_alpha_backward(int64_t N, const float* raw_density, const float* dt, const float* grad_alpha, const float*
    grad_one_minus_alpha, float* grad_raw_density, void* stream);

/* ---- mlp.hip ---- */
/* replaces: torch.nn.Sequential(Linear,GELU,...) evaluators, permuto_sdf_py/models/models.py:153-161,451-470 */
int64_t psdf_mlp_packed_size(int n_layers, const int* dims);

/* replaces: same evaluators (parameter re-ordering for the MFMA kernels).  `packed` holds psdf_mlp_packed_size floats:
   the fp32 operand image and, for nets whose image fits 80 KB of LDS, a second image of the same parameters as three
   bf16 pieces per weight (16-byte aligned, after the first) that psdf_mlp_forward multiplies on the bf16 matrix pipe
   with six products kept per fp32 multiply (fp32-level accuracy; no range restriction on inputs or weights). */
int psdf_mlp_pack(int n_layers, const int* dims, const float* const* weights, const float* const* biases, float*
    packed, void* stream);

/* replaces: models.py:187 `self.mlp_sdf(point_features)` and :508-517 (cuBLAS + elementwise GELU in the reference) */
int psdf_mlp_forward(int n_layers, const int* dims, int64_t N, const float* X, const float* packed, float* Y, void*
    stream);

/* ---- encode.hip / mlp.hip: masked forward ---- */
/* per-sample masks for fixed-shape callers (one slot per ray: the sphere tracer's converged rays): masked points /
   fully masked 32-sample tiles are not evaluated, their outputs are left untouched */
int psdf_encode_forward_masked(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float* positions,
    const float* lattice, const float* scale_factor, const float* shifts, const float* window, int concat_points, float
    points_scaling, const uint8_t* skip, float* sliced, void* stream);
int psdf_mlp_forward_masked(int n_layers, const int* dims, int64_t N, const float* X, const float* packed, const uint8_t*
    skip, float* Y, void* stream);
/* the analytic normal of the same callers (reference: get_sdf_and_gradient on the traced end points,
   permuto_sdf_py/utils/sdf_utils.py:203-208, which in the reference holds only the rays that met occupancy): data gradient
   of the MLP and position gradient of the encoding under the same mask; masked samples keep the contents of their outputs */
int psdf_mlp_backward_data_masked(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
    const float* const* biases, const float* dY, const uint8_t* skip, float* dX, void* stream);
int psdf_encode_backward_positions_masked(int pos_dim, int nr_feat, int64_t N, int nr_levels, int capacity, const float*
    positions, const float* lattice, const float* scale_factor, const float* shifts, const float* window, int
    concat_points, float points_scaling, const float* grad_sliced, const uint8_t* skip, float* grad_positions, void*
    stream);

/* ---- mlp_bwd.hip ---- */
/* replaces: autograd backward of the same evaluators (dX, dW_l, db_l in one launch; forward recomputed from X).
   weights[l]/biases[l]: torch-layout parameters; dW[l]/db[l] are accumulated into (caller zero-fills); dW = db = NULL:
   data gradient only (lighter kernel: analytic normals at inference).  dY = NULL (data gradient only; also accepted by
   psdf_mlp_backward_data_masked and psdf_mlp_double_backward): the unit gradient of output 0, i.e. d y_0 / d X -- the
   `torch.autograd.grad(sdf, points, torch.ones_like(sdf))` of models.py:236-251 without a [rows, N] tensor that is 1 in one row */
int psdf_mlp_backward(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights, const
    float* const* biases, const float* dY, float* dX, float* const* dW, float* const* db, void* stream);

/* replaces: the second differentiation torch autograd performs for the reference's eikonal / curvature losses
   (get_sdf_and_gradient with create_graph=True, models.py:236-251): VJP of (X, params) -> dX = J^T dY with upstream
   gradient V [dims[0], N]; dX2 receives d/dX, dW[l] / db[l] are accumulated into; 3 hidden layers */
int psdf_mlp_double_backward(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
    const float* const* biases, const float* dY, const float* V, float* dX2, float* const* dW, float* const* db,
    void* stream);
/* the same pass with the plain backward of an upstream gradient dY2 [dims[n_layers], N] of the net's OUTPUTS folded in (round 6: the
   training step runs both on the same samples -- g_y of (sdf, geometry features) beside g_n of the normals; one forward
   recomputation, one backward sweep, one launch instead of psdf_mlp_backward + psdf_mlp_double_backward): dX2 = the double
   backward's data gradient + the dX psdf_mlp_backward would return for dY2; dW[l] / db[l] accumulate both.  The reference's SDF net
   shapes only (models.py:153-161: <= 64 inputs, 32 x 3 hidden, a matrix output layer of 5 .. 48 rows); -2 otherwise */
int psdf_mlp_double_backward_plus(int n_layers, const int* dims, int64_t N, const float* X, const float* const* weights,
    const float* const* biases, const float* dY, const float* V, const float* dY2, float* dX2, float* const* dW,
    float* const* db, void* stream);

/* ---- fused.hip ---- */
/* replaces: models.py:186-192 `point_features=self.encoding(points, window); sdf_and_feat=self.mlp_sdf(point_features)`
   as ONE launch that never materialises the feature tensor (pos_dim 3, 2 features/level).  dims[0] must be
   2*(nr_levels + (concat_points?2:0)); skip [N] bytes and feat [dims[0],N] are optional (NULL) */
int psdf_encode_mlp_forward(int64_t N, int nr_levels, int capacity, const float* positions, const float* lattice,
    const float* scale_factor, const float* shifts, const float* window, int concat_points, float points_scaling,
    int n_layers, const int* dims, const float* packed, const uint8_t* skip, float* feat, float* Y, void* stream);

/* ---- optim.hip ---- */
/* replaces: torch.optim.AdamW at permuto_sdf_py/train_permuto_sdf.py:293-304,418 */
int psdf_adamw_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float
    beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* the same update for up to 64 tensors sharing the step count, one launch (host arrays of sizes / device pointers) */
int psdf_adamw_step_multi(int n_tensors, const int64_t* sizes, float* const* params, const float* const* grads, float*
    const* exp_avgs, float* const* exp_avg_sqs, float lr, float beta1, float beta2, float eps, float weight_decay, int
    step, float grad_scale, void* stream);

/* replaces: the same dense update (train_permuto_sdf.py:293-304: weight_decay 0 on the lattices) restricted to the blocks
   where it is not the identity.  The tensor is n_blocks blocks of block_elems floats; a block whose gradient and both
   moments are exactly zero (no batch ever touched its rows) is skipped, every other block gets the dense update -- moments
   of rows a batch does NOT touch keep decaying, as in torch.  touched [n_blocks] bytes: from psdf_encode_forward_mark,
   consumed (reset to 0); active [n_blocks] bytes: 1 once a block has been updated; zero_grad != 0: the gradient of the
   processed blocks is cleared in the same pass (persistent gradient buffer, no fill launch).  Bit-identical to
   psdf_adamw_step over the whole tensor. */
int psdf_adamw_step_blocks(int64_t n_blocks, int block_elems, float* param, float* grad, float* exp_avg, float*
    exp_avg_sq, unsigned char* touched, unsigned char* active, float lr, float beta1, float beta2, float eps, int step,
    float grad_scale, int zero_grad, void* stream);
/* psdf_adamw_step_blocks for up to 8 tensors in ONE launch: host arrays [n_tensors] of block counts, block sizes, device pointers
   and per-tensor lr / beta1 / beta2 / eps / step (the lattices of a training step sit in different parameter groups). */
int psdf_adamw_step_blocks_multi(int n_tensors, const int64_t* n_blocks, const int* block_elems, float* const* params,
    float* const* grads, float* const* exp_avgs, float* const* exp_avg_sqs, unsigned char* const* touched,
    unsigned char* const* active, const float* lr, const float* beta1, const float* beta2, const float* eps, const int* step,
    float grad_scale, int zero_grad, void* stream);


/* ---- sampling.hip ---- */
/* replaces: OccupancyGrid::compute_grid_points / compute_random_sample_of_grid_points, src/OccupancyGrid.cu:88-117,179-208 */
int psdf_grid_points(int count, int nr_voxels_per_dim, float extent, const float* grid_translation, const int*
    voxel_indices, uint64_t rng_state, uint64_t rng_inc, int randomize, float* out_points, void* stream);

/* replaces: OccupancyGrid::update_with_density[_random_sample], src/OccupancyGrid.cu:368-420 */
int psdf_grid_update_with_density(int count, const int* voxel_indices, const float* density, float decay, float
    thresh, float* grid_values, uint8_t* grid_occupancy, void* stream);

/* replaces: OccupancyGrid::update_with_sdf[_random_sample], src/OccupancyGrid.cu:422-475 */
int psdf_grid_update_with_sdf(int count, const int* voxel_indices, const float* sdf, int nr_voxels_per_dim, float
    extent, const float* grid_translation, float inv_s, const float* inv_s_tensor, int full_update, float thresh,
    float* grid_values, uint8_t* grid_occupancy, void* stream);

/* replaces: OccupancyGrid::check_occupancy, src/OccupancyGrid.cu:339-365 */
int psdf_grid_check_occupancy(int count, int nr_voxels_per_dim, float extent, const float* grid_translation, const
    uint8_t* grid_occupancy, const float* points, uint8_t* out, void* stream);

/* which form the last psdf_march_samples launched: 1 = march_kernel (a thread per ray), 2 = march_quad_kernel (four lanes per
   ray, unit grids with nr_rays <= 6144; PSDF_MARCH_FORM=thread|quad overrides); 0 = none yet.  Debug query (host only). */
int psdf_march_form(void);
/* replaces: OccupancyGrid::compute_samples_in_occupied_regions (src/OccupancyGrid.cu:212-257) and RaySampler::compute_samples_fg (src/RaySampler.cu:104-152);
   scratch: nr_rays * (3 + max_nr_samples_per_ray) 4-byte words */
int psdf_march_samples(int use_grid, int nr_rays, int nr_voxels_per_dim, float extent, const float* grid_translation,
    const uint8_t* grid_occupancy, const float* ray_origins, const float* ray_dirs, const float* ray_t_entry, const
    float* ray_t_exit, float min_dist_between_samples, int max_nr_samples_per_ray, int max_nr_samples, uint64_t
    rng_state, uint64_t rng_inc, int jitter, float* samples_pos, float* samples_dirs, float* samples_z, float*
    samples_dt, float* ray_fixed_dt, int* ray_start_end_idx, int* cur_nr_samples, int* scratch, const uint32_t*
    coarse_mask, void* stream);

/* replaces: OccupancyGrid::compute_first_sample_start_of_occupied_regions, src/OccupancyGrid.cu:259-300 */
int psdf_first_hit_samples(int nr_rays, int nr_voxels_per_dim, float extent, const float* grid_translation, const
    uint8_t* grid_occupancy, const float* ray_origins, const float* ray_dirs, const float* ray_t_entry, const float*
    ray_t_exit, int max_nr_samples, float* samples_pos, float* samples_dirs, float* samples_z, float* samples_dt,
    float* ray_fixed_dt, int* ray_start_end_idx, int* cur_nr_samples, int* scratch, const uint32_t* coarse_mask,
    void* stream);

/* replaces: OccupancyGrid::advance_sample_to_next_occupied_voxel, src/OccupancyGrid.cu:302-337 */
int psdf_advance_to_next_occupied_voxel(int count, int nr_voxels_per_dim, float extent, const float* grid_translation,
    const uint8_t* grid_occupancy, const float* samples_dirs, float* samples_pos, uint8_t* is_within_bounds, const
    uint32_t* coarse_mask, void* stream);

/* helper of the four marches above/below (no reference counterpart): "some voxel occupied" bit per 8x8x8 block of the
   Morton-ordered occupancy (kernels/permuto_sdf/OccupancyGridGPU.cuh:50-55 gives the order), psdf_occupancy_coarse_words(n)
   32-bit words; the marches copy it into LDS and answer the probes of empty blocks there (same results bit for bit).
   Optional everywhere: coarse_mask = NULL probes the bytes only.  Rebuild after every change of the occupancy. */
int psdf_occupancy_coarse_words(int nr_voxels_per_dim);
int psdf_occupancy_coarse_mask(int nr_voxels_per_dim, const uint8_t* grid_occupancy, uint32_t* coarse_mask, void*
    stream);

/* replaces: RaySampler::compute_samples_bg, src/RaySampler.cu:37-101 */
int psdf_samples_bg(int nr_rays, int nr_samples_per_ray, const float* ray_origins, const float* ray_dirs, const float*
    ray_t_exit, float sphere_radius, const float* sphere_center, uint64_t rng_state, uint64_t rng_inc, int randomize,
    int contract_3d_samples, float* samples_3d, float* samples_4d, float* samples_dirs, float* samples_z, float*
    samples_dt, float* ray_fixed_dt, int* ray_start_end_idx, void* stream);

/* replaces: Sphere::ray_intersection, src/Sphere.cu:42-79 */
int psdf_sphere_ray_intersection(int nr_rays, float radius, const float* center, const float* ray_origins, const
    float* ray_dirs, float* points_entry, float* t_entry, float* points_exit, float* t_exit, uint8_t* does_intersect,
    void* stream);

/* replaces: Sphere::rand_points_inside, src/Sphere.cu:82-108 */
int psdf_sphere_rand_points_inside(int count, float radius, const float* phi, const float* costheta, const float* u,
    float* points, void* stream);

/* replaces: RaySamplesPacked::compute_exact_nr_samples + compact_to_valid_samples, src/RaySamplesPacked.cu:44-95 */
int psdf_compact_offsets(int nr_rays, const int* ray_start_end_idx, int* scratch, int* total, void* stream);

/* replaces: RaySamplesPacked::compact_to_valid_samples, src/RaySamplesPacked.cu:57-95 */
int psdf_compact_copy(int nr_rays, const int* ray_start_end_idx, const int* offsets, const float* pos, const float*
    pos4, const float* dirs, const float* z, const float* dt, const float* sdf, const float* fixed_dt, float* o_pos,
    float* o_pos4, float* o_dirs, float* o_z, float* o_dt, float* o_sdf, float* o_fixed_dt, int* o_start_end, void*
    stream);

/* replaces: RaySamplesPacked::compute_per_sample_ray_idx, src/RaySamplesPacked.cu:124-146 */
int psdf_per_sample_ray_idx(int nr_rays, int nr_samples, const int* ray_start_end_idx, int* out, void* stream);

/* replaces: PermutoSDF::spherical_harmonics, src/PermutoSDF.cu:167-204 */
int psdf_spherical_harmonics(int count, int degree, const float* dirs, float* out, void* stream);

/* replaces: PermutoSDF::random_rays_from_reel, src/PermutoSDF.cu:67-112 */
int psdf_random_rays_from_reel(int nr_rays, int nr_images, int height, int width, const float* rgb_reel, const float*
    mask_reel, const float* K_reel, const float* tf_world_cam_reel, const int* pixel_indices, const int* img_indices,
    int has_mask, float* ray_origins, float* ray_dirs, float* gt_rgb, float* gt_mask, void* stream);

/* replaces: the Python loop of sphere_trace(), permuto_sdf_py/utils/sdf_utils.py:120-218 (first hit :127-133, per
   iteration step :167-185), with one slot per ray so that the loop is a fixed launch sequence (hipGraph) */
int psdf_first_hit_dense(int nr_rays, int nr_voxels_per_dim, float extent, const float* grid_translation, const
    uint8_t* grid_occupancy, const float* ray_origins, const float* ray_dirs, const float* ray_t_entry, const float*
    ray_t_exit, float* pos, uint8_t* converged, const uint32_t* coarse_mask, void* stream);
int psdf_sphere_trace_step(int count, int nr_voxels_per_dim, float extent, const float* grid_translation, const
    uint8_t* grid_occupancy, const float* dirs, const float* sdf, float sdf_multiplier, float sdf_converged_thresh,
    float* pts, uint8_t* converged, const uint32_t* coarse_mask, void* stream);
/* the same iteration with the long marches compacted into a second, densely packed launch (same results): work_flags
   [count] bytes, work_list [count] ints, work_count [1] int that the caller has zeroed; coarse_mask optional (the marches) */
int psdf_sphere_trace_step_compacted(int count, int nr_voxels_per_dim, float extent, const float* grid_translation, const
    uint8_t* grid_occupancy, const float* dirs, const float* sdf, float sdf_multiplier, float sdf_converged_thresh,
    float* pts, uint8_t* converged, uint8_t* work_flags, int* work_list, int* work_count, const uint32_t* coarse_mask,
    void* stream);

/* ---- volume_rendering.hip ---- */
/* replaces: (helper) replaces the atomicAdd slot counters, e.g. kernels/permuto_sdf/OccupancyGridGPU.cuh:599 */
int psdf_exclusive_scan_i32(int n, const int* in, int* out, int* total, void* stream);

/* replaces: VolumeRendering::cumprod_alpha2transmittance, src/VolumeRendering.cu:169-201 */
int psdf_cumprod_alpha2transmittance(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
    const float* alpha, float* transmittance, float* bg_transmittance, void* stream);

/* replaces: VolumeRendering::cumprod_alpha2transmittance_backward, src/VolumeRendering.cu:530-564 */
int psdf_cumprod_alpha2transmittance_backward(int nr_rays, const int* start_end, int equal, int fixed, int
    max_nr_samples, const float* grad_bg, const float* alpha, const float* bg, const float* cumsumLV, float*
    grad_alpha, void* stream);

/* replaces: VolumeRendering::integrate_with_weights, src/VolumeRendering.cu:204-232 */
int psdf_integrate_with_weights(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const
    float* rgb, const float* weights, float* pred, void* stream);

/* replaces: VolumeRendering::integrate_with_weights_backward, src/VolumeRendering.cu:567-601 */
int psdf_integrate_with_weights_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
    const float* grad_pred, const float* rgb, const float* weights, float* grad_rgb, float* grad_weights, int
    reference_compat, void* stream);

/* replaces: VolumeRendering::sum_over_each_ray, src/VolumeRendering.cu:272-344 */
int psdf_sum_over_each_ray(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, int channels,
    const float* values, float* sum_per_ray, float* sum_per_sample, void* stream);

/* replaces: VolumeRendering::sum_over_each_ray_backward, src/VolumeRendering.cu:604-668 */
int psdf_sum_over_each_ray_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, int
    channels, const float* grad_per_ray, const float* grad_per_sample, float* grad_values, void* stream);

/* replaces: VolumeRendering::cumsum_over_each_ray (:346-376) and compute_cdf (:379-408) */
int psdf_cumsum_over_each_ray(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const
    float* values, int inverse, int exclusive, float* out, void* stream);

/* replaces: VolumeRendering::sdf2alpha, src/VolumeRendering.cu:234-269 */
int psdf_sdf2alpha(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float*
    ray_fixed_dt, const float* samples_dt, const float* sdf, float inv_s, int dynamic_inv_s, float inv_s_multiplier,
    float* alpha, void* stream);
/* sdf2alpha -> clip(0,1) -> 1 - alpha + 1e-7 -> cumprod_alpha2transmittance -> alpha * T -> sum_over_each_ray -> / clamp(sum, 1e-6)
   -> compute_cdf in ONE launch: the operator chain of importance_sampling_sdf_model (permuto_sdf_py/utils/sdf_utils.py:383-423),
   bit-identical to calling the operators one by one; cdf [M,1] (zero-filled by the caller where slots belong to no ray) */
int psdf_sdf_importance_cdf(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float* ray_fixed_dt,
                            const float* samples_dt, const float* sdf, float inv_s, int dynamic_inv_s, float inv_s_multiplier,
                            float* cdf, void* stream);

/* replaces: VolumeRendering::compute_dt, src/VolumeRendering.cu:135-167 */
int psdf_compute_dt(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float*
    samples_z, const float* ray_t_exit, int use_ray_t_exit, float* dt, void* stream);

/* replaces: VolumeRendering::volume_render_nerf, src/VolumeRendering.cu:40-83 */
int psdf_volume_render_nerf(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float*
    rgb, const float* density, const float* samples_z, const float* samples_dt, float* pred_rgb, float* pred_depth,
    float* bg_transmittance, float* weight_per_sample, void* stream);

/* replaces: VolumeRendering::volume_render_nerf_backward, src/VolumeRendering.cu:86-132 */
int psdf_volume_render_nerf_backward(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples,
    const float* grad_pred_rgb, const float* grad_bg_transmittance, const float* pred_rgb, const float*
    bg_transmittance, const float* rgb, const float* density, const float* samples_dt, float* grad_rgb, float*
    grad_density, void* stream);

/* replaces: VolumeRendering::importance_sample, src/VolumeRendering.cu:410-461 */
int psdf_importance_sample(int nr_rays, const int* start_end, int equal, int fixed, int max_nr_samples, const float*
    ray_origins, const float* ray_dirs, const float* ray_fixed_dt, const float* samples_z, const float* cdf, int
    nr_importance_samples, uint64_t rng_state, uint64_t rng_inc, int jitter, float* out_pos, float* out_dirs, float*
    out_z, void* stream);

/* replaces: VolumeRendering::combine_uniform_samples_with_imp, src/VolumeRendering.cu:464-525 */
int psdf_combine_uniform_samples_with_imp(int nr_rays, const int* uni_start_end, int uni_equal, int uni_fixed, int
    uni_max_nr_samples, const float* ray_origins, const float* ray_dirs, const float* ray_t_exit, const float*
    uni_fixed_dt, const float* uni_z, const float* uni_sdf, int has_sdf, int nr_imp, const float* imp_z, const float*
    imp_sdf, int out_max_nr_samples, float* out_pos, float* out_dirs, float* out_z, float* out_dt, float* out_sdf,
    float* out_fixed_dt, int* out_start_end, int* out_cur_nr_samples, int* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PSDF_H */
