/* libishap_hip.so -- C ABI of the MI355X (gfx950) denoise-and-drag core.
 *
 * The reference (jinli99/iShapEditing) has no native boundary: its hot path is reached through
 * Python calls into stock torch ops.  Each entry point below names the reference function whose
 * arithmetic it replaces (paths relative to the reference root; gd = neural_field_diffusion/
 * guided_diffusion).  Conventions:
 *   - every function returns 0 on success, non-zero on failure; ishap_last_error() gives the
 *     thread-local message; nothing throws across this boundary;
 *   - all tensors are caller-allocated DEVICE buffers (torch owns memory); dims are explicit;
 *   - `stream` is a hipStream_t (0 = default stream); calls only enqueue work, they never sync;
 *   - torch-visible tensors use the reference's layouts (NCHW, fp32 unless noted).
 */
#ifndef ISHAP_H
#define ISHAP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* ishap_last_error(void);
/* Asynchronous device-side failures (the reference's analogue: a CUDA error raised by a later torch call).  A kernel of
 * this library that cannot go on correctly -- today: a bounded cross-workgroup wait that gave up -- poisons its outputs
 * with NaN and raises a process-wide status word; every ishap_unet_* call that enqueues work checks the word first and
 * fails (-3, message in ishap_last_error) if an EARLIER launch raised it.  This call checks on demand, e.g. after a
 * stream synchronise at the end of a loop: 0 = no failure since the last report; the word is cleared once reported. */
int ishap_device_status(void);
/* Tenancy.  The group-local GroupNorm kernels of the small maps run several workgroups per (image, group) that meet INSIDE
 * one launch (gd/nn.py:16-18 needs group-wide sums); such a grid only completes when all of its workgroups are resident
 * together.  Within one process the library arbitrates: per device, ONE (model context, stream) pair at a time may launch
 * such grids; a call on another context / thread / stream that arrives while the holder's work is still in flight runs the
 * same kernels with one workgroup per group, and the 8x8-map AttentionBlock kernel (which also hands data between workgroups
 * inside a launch) as two launches of the same code (BITWISE the same values either way, a few microseconds slower per
 * launch) -- no action needed.
 * The library is otherwise SINGLE-TENANT per GPU: another PROCESS using the same device, or a caller stream created with a
 * compute-unit mask, can keep part of such a grid from becoming resident; the wait is bounded, and the failure is reported
 * as described above (status word, NaN outputs, -3 from the next call), never a hang or a silently wrong result.
 * ishap_rendezvous_would_grant: diagnostic, no side effects -- 1 if a launch sequence of `owner` (a model context, or NULL
 * for the stand-alone operator calls) on `stream` would be allowed in-launch rendezvous right now. */
int ishap_rendezvous_would_grant(const void* owner, void* stream);
int ishap_version(void);   /* 2 since ishap_mesh_smooth takes (and checks) the size of its scratch buffer; 3 since ishap_step_coefs
                            * ends with rng / rng_seed / rng_offset / noise_out */

/* ---------------------------------------------------------------- UNet (gd/unet.py:396-671) */
typedef struct ishap_unet ishap_unet;

typedef struct {
  int image_size;          /* 128 */
  int in_channels;         /* 96 */
  int model_channels;      /* 256 */
  int out_channels;        /* 192 (learn_sigma) */
  int num_res_blocks;      /* 2 */
  int n_mult;              /* length of channel_mult */
  int channel_mult[8];     /* (1,1,2,3,4) for image_size 128, gd/script_util.py:151-161 */
  int n_att;
  int attention_ds[8];     /* downsample rates with attention: image_size // res, script_util.py:163-165 */
  int num_head_channels;   /* 64 */
  int max_batch;           /* largest N a forward may be called with */
} ishap_unet_config;

/* UNetModel.__init__ with use_scale_shift_norm, resblock_updown, use_fp16 (drag_utils.py:44-57):
 * builds the block graph, sizes the activation arena and workspaces on `device`. */
int ishap_unet_create(const ishap_unet_config* cfg, int device, ishap_unet** out);
void ishap_unet_destroy(ishap_unet* u);

/* state_dict key table (same names/shapes torch gives the reference model; drag_utils.py:229-230
 * loads with strict=True).  shape has up to 4 entries. */
int ishap_unet_num_params(const ishap_unet* u);
int ishap_unet_param_info(const ishap_unet* u, int index, char* name, int name_cap, int* ndim, long long* shape);
/* model.load_state_dict + convert_to_fp16 (gd/unet.py:618-624, gd/fp16_util.py:14-21) for one tensor:
 * `data` = fp32 device buffer of `numel` values in the reference's layout.  Conv weights are re-packed
 * to fp16 MFMA operands (forward and input-gradient forms); torso conv biases are rounded through fp16. */
int ishap_unet_load_param(ishap_unet* u, const char* name, const float* data, long long numel, void* stream);
int ishap_unet_params_loaded(const ishap_unet* u);   /* number of distinct tensors loaded so far */

/* UNetModel.forward(x, timesteps, feat_layer) (gd/unet.py:634-671) with timesteps already mapped to the
 * original 0..999 index (gd/respace.py:122-127).
 *   x            [N][in_channels][S][S] fp32
 *   timesteps    HOST array of N floats
 *   feat_layer   output-block index whose activation is tapped (h.clone() at unet.py:665-666), or -1
 *   out          [N][out_channels][S][S] fp32
 *   inter_feat   optional [N][C_tap][S_tap][S_tap] fp16 copy of the tap in the reference layout (may be NULL;
 *                the tap always stays resident inside the context for ishap_drag_* / backward)
 *   keep_for_backward  non-zero: keep every intermediate needed by ishap_unet_backward_input */
int ishap_unet_forward(ishap_unet* u, const float* x, const float* timesteps, int N, int feat_layer,
                       float* out, void* inter_feat, int keep_for_backward, void* stream);
int ishap_unet_tap_shape(const ishap_unet* u, int feat_layer, int* channels, int* size);
/* device pointer of the resident tap of the last forward: NHWC fp16 [N][S_tap*S_tap][C_tap] */
const void* ishap_unet_tap_ptr(const ishap_unet* u);
/* keep_for_backward is a bit set: bit 0 = keep what a following backward re-reads; bit 1 (with feat_layer >= 0) = the part of the
 * network AFTER the tapped output block (the remaining output blocks and the fp32 head) is only PLANNED by the forward and later
 * enqueued on a stream owned by the context.  The drag step needs only the tap for its loss and backward pass
 * (drag_utils.py:355-384), the model output only for the DDPM update after them (:385-393): the two then run side by side.
 * `out` is complete on a stream only after ishap_unet_join_tail(u, that stream); the next ishap_unet_forward, a
 * full-depth backward and ishap_unet_block_output join by themselves.
 * LIFETIME: with bit 1 the library keeps the raw `out` pointer and writes through it when the tail runs -- `out` must stay allocated until ishap_unet_join_tail has been called for this forward (or the next
 * forward / full-depth backward has joined it); freeing it earlier lets the tail write into memory that may have a new owner. */
int ishap_unet_join_tail(ishap_unet* u, void* stream);
/* Enqueues the planned tail on the context's own stream, behind the point the backward pass marked after its first output
 * blocks (or behind the tap when no backward ran).  No-op when nothing is planned; ishap_unet_join_tail runs a plan that was
 * never enqueued.  On failure the launches already enqueued stay ordered before the next join (the context remains usable). */
int ishap_unet_run_tail(ishap_unet* u);
/* Diagnostics (contexts created with ISHAP_BWD_MARKS=1 in the environment; otherwise returns 0): elapsed milliseconds from the start of
 * the last backward pass to the timing event recorded after each of its blocks -- tags: 0 start, 100 + i after output block i, 200
 * after the middle block, 300 + i after input block i, 999 end -- and to the begin / end of the forward tail that ran beside it on the
 * context's side stream (-1 when none did).  Returns the number of marks written (<= cap); synchronises with the device. */
int ishap_unet_marks(ishap_unet* u, int* tags, float* ms, int cap, float* tail_begin_ms, float* tail_end_ms);
/* copy it into a caller buffer of N*S_tap^2*C_tap halfs (the guidance cache of drag_utils.py:275-276, kept
 * on the device in the tap's own layout instead of resized fp32 copies on the host) */
int ishap_unet_copy_tap(const ishap_unet* u, void* dst, void* stream);
/* The output of any TimestepEmbedSequential of the last forward(keep_for_backward=1): group 0 = input_blocks[index]
 * (the `hs` list, gd/unet.py:658-660), 1 = middle_block (:661), 2 = output_blocks[index] (:662-666).  Writes the
 * block's channel count and side to *channels / *size; when dst is non-NULL also copies the activation as fp16
 * [N][channels][size][size] (the reference's layout).  Parity tests use it to localise a mismatch to one block. */
int ishap_unet_block_output(const ishap_unet* u, int group, int index, int* channels, int* size, void* dst_nchw_f16,
                            void* stream);
/* Optional, ahead of a sampling loop: compute the timestep-embedding products of n timesteps once (timestep_embedding,
 * time_embed and every ResBlock's emb_layers, gd/unet.py:651,245-250 -- none of them depends on x).  A later
 * ishap_unet_forward whose timesteps all equal one prepared value reuses its row and skips those four launches; any other
 * forward computes them as before.  Results are bit-identical either way.  n = 0 drops the prepared rows; loading a
 * parameter drops them too.  Call it between steps, not between a kept forward and its backward (that forward's
 * intermediates are invalidated if it used a prepared row). */
int ishap_unet_prepare_timesteps(ishap_unet* u, const float* timesteps, int n, void* stream);
/* Device bytes the context holds besides the packed weights: activation arena + split-K partials + GroupNorm scratch
 * (SURVEY 8b's ishap_workspace_bytes; the context allocates them itself at create time, sized by dry runs of
 * forward + backward at every batch size 1..max_batch). */
long long ishap_unet_workspace_bytes(const ishap_unet* u);

/* d(sum(tap * cot)) / dx through output block feat_layer ... input block 0: what loss.backward()
 * computes for img.grad at drag_utils.py:383, without the weight gradients the reference discards.
 *   cot_nhwc   fp16 cotangent of the tap in the tap's resident layout [N][S_tap^2][C_tap], already
 *              multiplied by the loss scale *scale2[0] (see ishap_grad_to_scaled_f16)
 *   scale2     device float[2] = {scale, 1/scale} or NULL for scale 1
 *   dx         [N][in_channels][S][S] fp32 = gradient w.r.t. x (loss scale removed) */
int ishap_unet_backward_input(ishap_unet* u, const void* cot_nhwc, const float* scale2, float* dx, void* stream);
/* same, from a cotangent of the model output [N][out_channels][S][S] fp32 (full-depth backward,
 * drag_utils.py:458 in train_triplane) */
int ishap_unet_backward_from_output(ishap_unet* u, const void* cot_out, int cot_is_f16, const float* scale2, float* dx,
                                    void* stream);
/*   cot_out: fp32, or fp16 already multiplied by scale2[0] (ishap_grad_to_scaled_f16); scale2 as above */

/* ---------------------------------------------------------------- GroupNorm32 alone (gd/nn.py:16-18,92-99)
 * normalization(C) = GroupNorm32(32, C): statistics and normalisation in fp32 on x.float(), eps 1e-5, result cast back
 * to the torso's fp16; followed by nn.SiLU in ResBlock.in_layers / out_layers and the head (gd/unet.py:179-183,
 * 205-209, 612-614).  The executor fuses this into the neighbouring kernels and picks one of several statistics
 * routes by map size; these calls run a chosen route on a caller-given tensor, so the reference's own primitive
 * fixtures reach each of them.
 *   x, y, g, dx   NHWC fp16 [N][H*W][C]        stats  [N][32][2] fp32 (mean, rstd): written by the forward, read by the backward
 *   silu          0: y = GN(x);  1: y = SiLU(GN(x))
 *   route         0 = the executor's choice for this map; 1 = two-pass statistics; 2 = group-local, one workgroup per
 *                 (image, group); 3 = group-local, several workgroups meeting in the in-launch rendezvous; 4 (forward
 *                 only) = fixed-point per-channel sums gathered by an implicit-GEMM epilogue (C % 64 == 0)
 *   scratch       device buffer of ishap_group_norm32_scratch_bytes(N, H*W, C) bytes (contents irrelevant) */
long long ishap_group_norm32_scratch_bytes(int N, int HW, int C);
int ishap_group_norm32(const void* x_nhwc_f16, const float* gamma, const float* beta, int N, int H, int W, int C, int silu,
                       int route, void* y_nhwc_f16, float* stats, void* scratch, void* stream);
/* input gradient of sum(y * g) w.r.t. x (no parameter gradients), as autograd derives it at drag_utils.py:383 */
int ishap_group_norm32_backward(const void* g_nhwc_f16, const void* x_nhwc_f16, const float* stats, const float* gamma,
                                const float* beta, int N, int H, int W, int C, int silu, int route, void* dx_nhwc_f16,
                                void* scratch, void* stream);
/* workgroups per (image, group) route 3 uses for this shape on the current device (> 1: the rendezvous is exercised) */
int ishap_group_norm32_parts(int N, int HW, int C);

/* ------------------------------------------- diffusion step (gd/gaussian_diffusion.py:232-331, 400-510) */
typedef struct {
  float min_log;        /* posterior_log_variance_clipped[t]  (float64 table cast to fp32, :1035-1048) */
  float max_log;        /* log(betas[t]) */
  float sqrt_recip;     /* sqrt_recip_alphas_cumprod[t] */
  float sqrt_recipm1;   /* sqrt_recipm1_alphas_cumprod[t] */
  float coef1, coef2;   /* posterior_mean_coef1/2[t] */
  float nonzero;        /* 0 when t == 0 else 1 */
  int clip_denoised;
  int mode;             /* 0: mean + sqrt(var)*noise (p_sample_guidance :503/:508)
                           1: mean + exp(0.5*logvar)*noise (p_sample :443)
                           2: mean + variance_noise (:498-499; `noise` holds variance_noise)
                           3: DDIM (ddim_sample :654-705): x0*ddim_a + ddim_b*eps(x0) + nonzero*ddim_sigma*noise */
  float ddim_a;         /* sqrt(alphas_cumprod_prev[t]) */
  float ddim_b;         /* sqrt(1 - alphas_cumprod_prev[t] - sigma^2) */
  float ddim_sigma;     /* eta * sqrt((1-abar_prev)/(1-abar)) * sqrt(1 - abar/abar_prev) */
  /* round 5: the step draws its own noise (the reference's th.randn_like(x), gaussian_diffusion.py:443 / :493) when rng != 0 and
   * `noise` is NULL -- no randn launch and no noise tensor in front of the step.  Philox4x32-10 (Salmon et al., Random123) with
   * key = rng_seed ^ 0x9E3779B97F4A7C15 and counter = {index of the 4-float vector in x (64 bit), rng_offset (64 bit)}; the four
   * words become four standard normals by two Box-Muller pairs (u = ((w >> 8) + 0.5) * 2^-24): element 4 i + j of x takes normal j
   * of vector i.  noise_out: optional [N][C][HW] that receives the noise used (the dict's "noise"). */
  int rng;
  unsigned long long rng_seed, rng_offset;
  float* noise_out;
} ishap_step_coefs;
/* any of sample / pred_xstart / variance / mean may be NULL; noise NULL = zeros (or drawn, ishap_step_coefs::rng); variance_in optional */
int ishap_ddpm_step(const float* x, const float* model_out, const float* noise, const float* variance_in,
                    const ishap_step_coefs* k, int N, int C, int HW,
                    float* sample, float* pred_xstart, float* variance, float* mean, void* stream);
/* The step of p_sample_guidance (mode 0) and the guided update of the drag loop in ONE pass (round 5; gd/gaussian_diffusion.py:
 * 446-510 followed by drag_utils.py:384-392): guided = sample + variance * (scale * grad [* *grad_mul_dev]) with the sample and
 * variance this step computes; `sample` / `variance` are optional outputs (null: not written). */
int ishap_ddpm_step_guided(const float* x, const float* model_out, const float* noise, const float* variance_in,
                           const ishap_step_coefs* k, int N, int C, int HW, const float* grad, float scale,
                           const float* grad_mul_dev, float* guided, float* sample, float* variance, void* stream);
/* img = sample + variance * (scale * grad)   (drag_utils.py:384-392); grad_mul_dev: optional device scalar */
int ishap_guided_update(const float* sample, const float* variance, const float* grad, float scale,
                        const float* grad_mul_dev, long long numel, float* out, void* stream);
/* out = a*x + b*y  (forward-noising chain of ddpm_inversion, gd/gaussian_diffusion.py:520-522) */
int ishap_axpby(const float* x, const float* y, float a, float b, long long numel, float* out, void* stream);

/* ---------------------------------------------------------- drag loss (drag_utils.py:141-159, 309-382) */
typedef struct {
  int W;                /* side of the tap feature map (64) */
  int ld;               /* channels of the tap (512) */
  int Cc;               /* channels per plane after resize_feat_align (170) */
  const int* chmap;     /* device int[3*Cc]: (plane, c) -> tap channel (resize_feat_align's mapping) */
  const float* sources; /* device [B][3] */
  const float* targets; /* device [B][3] */
  int B;
  int r;                /* lattice radius r1 (12) */
  float voxel;          /* 2 / shape_resolution */
  float cof;
  int l1;               /* loss_type == 'l1' */
  unsigned char* touched; /* device scratch [3*W*W], 3*W*W % 4 == 0: bit 0 = the reference's rounded-texel sets, bit 1 = target footprints */
  int* nmask;             /* device scratch [1] */
  void* acc;              /* device scratch, 16 bytes (two 64-bit fixed-point loss sums) */
  void* grad_fx;          /* device scratch, W*W*ld*8 bytes: the gradient scatter accumulates in 64-bit fixed point so
                           * that repeated edits are bitwise identical (integer atomics commute) */
  unsigned char* chan_weight; /* device scratch [3*ld] bytes: inverse of chmap, filled by ishap_drag_setup */
} ishap_drag_args;
/* once per edit: rounded-texel bitmap and complement count (drag_utils.py:322-334); also zeroes acc and grad_fx, which
 * every loss call below expects zero on entry and leaves zero on return */
int ishap_drag_setup(const ishap_drag_args* a, void* stream);
/* per step: loss (device float[1]) and d loss / d tap as fp32 NHWC [W*W][ld] (drag_utils.py:355-383) */
int ishap_drag_loss_grad(const ishap_drag_args* a, const void* edit_nhwc_f16, const void* orig_nhwc_f16,
                         float* grad_nhwc, float* loss, void* stream);
/* the same followed by ishap_grad_to_scaled_f16 (below) on that gradient, as three launches instead of ten: the form the
 * guided step uses (drag_utils.py:355-383 up to loss.backward()) */
int ishap_drag_loss_cotangent(const ishap_drag_args* a, const void* edit_nhwc_f16, const void* orig_nhwc_f16,
                              float* grad_nhwc, float* loss, void* cot_f16, unsigned* bits, float* scale2, void* stream);
/* fp32 gradient -> fp16 cotangent times a power-of-two loss scale picked from max|g| on the device;
 * bits: device scratch uint32[1]; scale2: device float[2] = {scale, 1/scale} */
int ishap_grad_to_scaled_f16(const float* grad, void* out_f16, unsigned* bits, float* scale2, long long numel,
                             void* stream);

/* ------------------------------------- decoder (triplane_decoder/axisnetworks.py:517-562, visualize.py:76-97) */
typedef struct {
  const float* B;       /* net.0._B      [32][64] */
  const float* W1;      /* net.1.weight  [128][128] */
  const float* b1;
  const float* W2;      /* net.3.weight  [128][128] */
  const float* b2;
  const float* w3;      /* net.5.weight  [1][128] */
  const float* b3;      /* net.5.bias    [1] */
} ishap_decoder_weights;
/* (latent * range + middle).reshape(3,32,S,S) (drag_utils.py:295) into channels-last planes [3][S][S][32];
 * range/middle: device float[96] or NULL for 1/0 */
int ishap_planes_prepare(const float* latent, const float* range, const float* middle, int S, float* planes,
                         void* stream);
/* MultiTriplane.forward: logits for explicit coords [npts][3] */
int ishap_triplane_decode_points(const float* planes, int S, const ishap_decoder_weights* w, const float* coords,
                                 long long npts, float* logits, void* stream);
/* create_obj_o3d's dense grid (visualize.py:79-97): axis = device float[res] (torch.linspace(-1,1,res)),
 * volume[res][res][res] with x slowest, no host round trips */
int ishap_triplane_decode_grid(const float* planes, int S, const ishap_decoder_weights* w, const float* axis, int res,
                               float* volume, void* stream);

/* real-shape guidance (drag_utils.py:447-463): prediction = decoder(0, coord); loss = -BCEWithLogitsLoss()(prediction, gt);
 * loss.backward() -- forward and backward of the decoder on sampled points.  W1T / W2T: transposed copies of
 * net.1.weight / net.3.weight.  Outputs: loss[1], dplanes [3][S][S][32] = d loss / d planes, optional logits[npts]. */
int ishap_triplane_points_loss_grad(const float* planes, int S, const ishap_decoder_weights* w, const float* W1T,
                                    const float* W2T, const float* coords, const float* gt, long long npts,
                                    float* dplanes, float* loss, float* logits, void* stream);
/* chain rule from planes = clamp(sqrt_recip*x - sqrt_recipm1*eps, -1, 1)*range + middle back to the step's inputs
 * (drag_utils.py:448-450, gd/gaussian_diffusion.py:333-338,299-301): g_direct [96][S][S] = explicit d/dx term,
 * cot_out [192][S][S] = cotangent of the model output (eps half; variance half zero) for the UNet backward */
int ishap_x0_grad_to_cotangent(const float* dplanes, const float* range, const float* x, const float* model_out,
                               float sqrt_recip, float sqrt_recipm1, int clip_denoised, int S, float* g_direct,
                               float* cot_out, void* stream);

/* ------------------------------------------------------------------ surface of the decoded volume
 * Replaces the third-party CPU calls after the decode: mcubes.marching_cubes(volume, 0) (visualize.py:100),
 * mesh.filter_smooth_simple(10) (drag_utils.py:300) and the nearest-neighbour part of meshProcess.py:18-35.
 * `method` 1 = MARCHING CUBES (what the reference calls): one vertex per sign-changing grid edge at the linear zero crossing,
 * shared by the cells around the edge (so the vertex count is the marching-cubes vertex count), triangles from a 256-case
 * table derived by tools/make_mc_table.py (PyMCubes' own table is not available here: triangle-level parity unpinned).
 * `method` 0 = marching tetrahedra (6 per cell; extra vertices on face / body diagonals).  Grid coordinates,
 * deterministic voxel order.  volume: device float[res^3], x slowest.
 * Two calls because the caller allocates the outputs: count -> read counts -> emit. */
long long ishap_surface_scratch_bytes(int res);
/* counts: device unsigned[2] = {vertices, triangles} */
int ishap_surface_count(const float* volume, int res, float level, int method, void* scratch, unsigned* counts, void* stream);
/* verts: device float[3*vertices]; tris: device int[3*triangles]; same volume / level / method / scratch as the count call */
int ishap_surface_emit(const float* volume, int res, float level, int method, void* scratch, float* verts, int* tris,
                       void* stream);
/* in place: v <- (v + sum of neighbours) / (1 + number of neighbours), `iterations` Jacobi sweeps (each neighbour once:
 * Open3D's filter_smooth_simple).  box_max > 0: the vertices are in grid coordinates of a [0, box_max]^3 volume and the
 * mesh may be open where the surface leaves the box (edges lying in a box face belong to one triangle); box_max <= 0: the
 * mesh is closed.  scratch: ishap_mesh_smooth_scratch_bytes(nverts, ntris) device bytes (the vertex adjacency, built once per
 * call, and a second vertex buffer; ~28 bytes per vertex + 24 per triangle).  Vertex indices and 6*ntris must fit 32 bits. */
long long ishap_mesh_smooth_scratch_bytes(long long nverts, long long ntris);
/* ABI version 2: `scratch_bytes` = the size of the caller's buffer; a buffer smaller than
 * ishap_mesh_smooth_scratch_bytes(nverts, ntris) fails the call (version 1 took 32 * nverts bytes on trust). */
int ishap_mesh_smooth(float* verts, long long nverts, const int* tris, long long ntris, int iterations, float box_max,
                      void* scratch, long long scratch_bytes, void* stream);
/* out2[0] = mean over a of min_b |a-b|^2, out2[1] = mean over b of min_a |a-b|^2 (device floats; their sum is the
 * reference's chamfer distance); nearest: device scratch float[max(na, nb)] */
int ishap_chamfer(const float* a, long long na, const float* b, long long nb, float* nearest, float* out2, void* stream);

/* ------------------------------------------------------------------ occupancy samples of an input mesh (8(f) rank 2)
 * Replaces the Open3D calls of train_triplane's data preparation (drag_utils.py:411-440): mesh.sample_points_uniformly
 * (area-weighted triangle choice -- the caller draws the triangle indices from `areas` and the uniforms) and
 * RaycastingScene.compute_occupancy (here: parity of the crossings of the +x ray with the closed triangle mesh).
 * verts: device float[3*nverts]; tris: device int[3*ntris]. */
int ishap_mesh_tri_areas(const float* verts, const int* tris, long long ntris, float* areas, void* stream);
/* pts[i] = uniform point of triangle tri_idx[i] from the uniforms uw[2i], uw[2i+1] */
int ishap_mesh_points_on_tris(const float* verts, const int* tris, const int* tri_idx, const float* uw, long long n,
                              float* pts, void* stream);
/* occ[i] = 1 inside / 0 outside */
int ishap_mesh_occupancy(const float* verts, const int* tris, long long ntris, const float* pts, long long npts,
                         float* occ, void* stream);

/* ------------------------------------------------------------------ measurement aid (bench.py roofline leg)
 * Brackets every implicit-GEMM launch with HIP events on its own stream between begin and end.
 * out[v*3+{0,1,2}] = {launches, total ms, algorithmic FLOPs}; one v per kernel symbol: 0 conv3x3 128^2 tile,
 * 1 conv3x3 64^2 tile, 2 GEMM 128^2 tile, 3 GEMM 64^2 tile, 4 conv3x3 64^2 tile two-team, 5 small-map GEMM kernel,
 * 6 register-staged stem kernel, 7 small-map 3x3 weight-streaming kernel. */
int ishap_profile_begin(void);
int ishap_profile_end(double* out, int nvar);
/* Per-shape CSV ("M,N,K,conv3,tile,ksplit,launches,main_ms,reduce_ms,gflop" lines) of the same records; call
 * before the next ishap_profile_begin.  Returns the number of lines, -2 when `cap` is too small. */
int ishap_profile_shapes(char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
