/*
 * flocoder_amd.h -- C ABI of the MI355X (gfx950) implementation of flocoder's latent-flow hot path.
 *
 * The reference (drscotthawley/flocoder @ 2025-08-08) is pure Python and has no FFI; what it has are
 * Python protocols.  Each entry point below names the reference interface it replaces (file:line in
 * the reference tree) -- INTEGRATION.md shows the ctypes stub a maintainer would add on their side.
 *
 * Conventions
 *   - every pointer marked "dev" is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr());
 *     "host" pointers are plain host memory.  fp32 throughout.  Boundary tensors are NCHW contiguous,
 *     exactly the reference's layout; the library's internal activations are NHWC.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Launch functions enqueue work and
 *     return; they never allocate, free or synchronise.  Allocation happens in *_create / *_reserve only.
 *   - return value: 0 = FC_OK, negative = error; fc_last_error() gives the message (thread-local).
 *   - objects are thread-compatible: use one object per host thread.
 */
#ifndef FLOCODER_AMD_H
#define FLOCODER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FC_OK 0
#define FC_E_ARG (-1)    /* bad argument / unknown name                     -> Python ValueError   */
#define FC_E_SHAPE (-2)  /* shape the kernels do not support / not reserved -> Python ValueError   */
#define FC_E_ARCH (-3)   /* device is not gfx950                            -> Python RuntimeError */
#define FC_E_HIP (-4)    /* a HIP runtime call failed                       -> Python RuntimeError */
#define FC_E_STATE (-5)  /* weights not loaded, plan not built ...          -> Python RuntimeError */

#define FC_ABI_VERSION 1

int fc_abi_version(void);
const char* fc_last_error(void);
/* 0 if `device` is a gfx950 part, FC_E_ARCH otherwise. */
int fc_check_device(int device);

/* ------------------------------------------------------------------------------------------------
 * Velocity U-Net  (replaces flocoder/unet.py:164-377  Unet.__init__ / Unet.forward)
 * ---------------------------------------------------------------------------------------------- */
typedef struct fc_unet fc_unet;

typedef struct fc_unet_config {
    int dim;             /* Unet(dim=...)            unet.py:167 */
    int channels;        /* Unet(channels=...)       unet.py:169 */
    int n_levels;        /* len(dim_mults)           unet.py:168 */
    int dim_mults[8];
    int groups;          /* resnet_block_groups      unet.py:170 */
    int n_classes;       /* 0 = no class_cond_mlp    unet.py:172,181 */
    int mask_cond;       /* inpainting branches      unet.py:173,214-235 */
} fc_unet_config;

/* device >= 0: a gfx950 HIP device.  device < 0: description only (parameter table, no GPU touched). */
int fc_unet_create(const fc_unet_config* cfg, int device, fc_unet** out);
void fc_unet_destroy(fc_unet* u);

/* Parameter table, in the order fc_unet_load_params expects.  Names and shapes are the reference's
 * state_dict keys; `shape` receives up to 4 dims, unused = 0. */
int fc_unet_param_count(const fc_unet* u);
int fc_unet_param_info(const fc_unet* u, int i, const char** name, int64_t shape[4], int64_t* offset);
int64_t fc_unet_param_numel(const fc_unet* u);
/* Copy the flat fp32 parameter vector (all params concatenated in table order) and re-pack it into the
 * kernels' layouts.  `on_device` says whether `flat` is a device or a host pointer.
 * Replaces Unet.load_state_dict (generate_samples.py:101-106). */
int fc_unet_load_params(fc_unet* u, const float* flat, int64_t numel, int on_device, void* stream);

/* Build the launch plan and allocate the activation arena for batches up to `max_batch` of HxW latents.
 * `max_batch` counts U-Net rows: a CFG sampler of B samples needs 2B. */
int fc_unet_reserve(fc_unet* u, int max_batch, int height, int width);
/* The current reservation (0, 0, 0 before the first one).  The launch plan depends on it -- tiles are picked for max_batch rows -- so two
 * handles give bit-identical results only under equal reservations: Unet.replica() copies it (sampling.sample_many, bench.py two-in-flight). */
int fc_unet_reserved(const fc_unet* u, int* max_batch, int* height, int* width);

/* v = Unet(x, time, cond)   (unet.py:374-377).
 *   x_dev [B,C,H,W]; time_dev [B] (already multiplied by t_scale, sampling.py:63);
 *   class_ids_dev [B] int64 or NULL (cond['class_cond'] is None), an id < 0 disables the class
 *   embedding for that row; mask_dev [B,C,H,W] or NULL (cond['mask_cond']); mask_is_ones = the
 *   reference's allclose(mask,1) bypass (unet.py:301) decided by the caller; out_dev [B,C,H,W]. */
int fc_unet_forward(fc_unet* u, const float* x_dev, const float* time_dev, const int64_t* class_ids_dev,
                    const float* mask_dev, int mask_is_ones, float* out_dev, int batch, int height, int width,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * ODE integrators  (replace flocoder/sampling.py:36-122 rk4_step / v_func_cfg / generate_latents_rk4,
 *                   and the legacy Euler loop legacy/train_sd_flowers.py:50-67)
 * ---------------------------------------------------------------------------------------------- */
#define FC_METHOD_EULER 0
#define FC_METHOD_RK4 1

/* Integrate x_dev [B,C,H,W] in place along the caller-supplied fp32 time grid.
 *   RK4  : ts_host has n_points entries (= warp_time(linspace(...)), sampling.py:102-111); the loop runs
 *          n_points-1 intervals with stages at t, t+dt/2, t+dt/2, t+dt  (sampling.py:43-48,116-117).
 *   Euler: ts_host has n_points = N entries t_i = i/N*(1-eps)+eps; x += v*dt with dt_euler = 1/N
 *          (train_sd_flowers.py:58-64).
 *   t_scale multiplies every time value before it reaches the U-Net (999, sampling.py:57).
 *   class_ids_dev/mask_dev as in fc_unet_forward.  cfg_strength != 0 with class ids present runs the
 *   classifier-free-guidance pair as one 2B-row pass (sampling.py:69-74).
 * A run of consecutive integration steps (as many as fit ~6000 graph nodes: all 64 of the Euler sampler) is captured in a hipGraph
 * the first time a (method, B, H, W, cfg, mask, run length) variant is seen and replayed from then on; step counter, time grid and
 * conditioning table live in device memory, so replays take any grid.  On return the trajectory is queued on `stream`, not finished:
 * fc_unet_check(u, stream, 1) waits for it and reports a timed-out cross-workgroup wait (see fc_unet_set_shared). */
int fc_unet_integrate(fc_unet* u, int method, float* x_dev, int batch, int height, int width,
                      const float* ts_host, int n_points, float dt_euler, float t_scale,
                      const int64_t* class_ids_dev, float cfg_strength, const float* mask_dev, int mask_is_ones,
                      void* stream);

/* Number of kernel launches in one U-Net forward of the current plan, and its algorithmic FLOPs per
 * sample (2 x MACs over every conv / linear / attention contraction). */
/* The reserved batch runs as this many independent row ranges ("chains") on concurrent streams; *rows_per_chain = the rows
 * each chain's plan holds (= the batch every kernel launch of that chain sees). */
int fc_unet_chains(const fc_unet* u, int* rows_per_chain);
int fc_unet_plan_launches(const fc_unet* u);
/* Fused Block tails meet the workgroups of a sample at a device-side counter (bounded wait).  *count = launches whose wait ever
 * timed out since the plan was built -- must be 0; anything else means the residency assumption broke and results are invalid.
 * Synchronises. */
int fc_unet_fused_tail_errors(const fc_unet* u, int* count);
/* Device sharing.  The default ("exclusive") plan closes most Blocks inside their second convolution by letting the workgroups of a
 * sample exchange GroupNorm partials inside the launch; such a launch is only correct while its whole grid is resident, i.e. while
 * nothing else competes for the CUs.  Within one process the library orders these plans against each other across streams by itself.
 * A caller that runs the handle beside other GPU work it does not order against (a second replica meant to overlap, collectives of a
 * training job, another process on the same GPU) declares that with shared = 1: plans are then built without cross-workgroup waits
 * (same results to fp32 rounding -- the GroupNorm partials are combined in another order --, one more launch per Block).  Changing the mode drops the current plan; the next reserve rebuilds it.
 * No counterpart in the reference (PyTorch kernels never wait for each other). */
int fc_unet_set_shared(fc_unet* u, int shared);
/* Launches of the current plan whose workgroups wait for each other (0 for a shared-mode or training plan). */
int fc_unet_meeting_launches(const fc_unet* u);
/* Has a wait of such a launch ever timed out?  A timed-out wait turns its samples into NaN (never finite garbage) and makes every later
 * call on the handle -- this one included -- return FC_E_STATE until the plan is rebuilt.  With `synchronize` != 0 the call first waits
 * for `stream` (the stream the last forward / integration was given), so that its answer covers that work. */
int fc_unet_check(fc_unet* u, void* stream, int synchronize);
/* Test hook: makes the next run of the plan's first meeting launch time out. */
int fc_debug_unet_break_meeting(fc_unet* u);
int fc_debug_unet_break_meeting_kind(fc_unet* u, int kind);   /* kind: 0 a convolution's Block tail, 1 the linear attention's fused close */
/* Experiment switch: plans built after the call use (1) / do not use (0) the cross-workgroup Block tails; < 0 restores the default
 * (on, unless FLOCODER_AMD_FUSED_TAIL says otherwise). */
int fc_debug_set_fused_tail(int on);
/* FLOPs per sample and evaluation as the REFERENCE computes the network (2 x MACs of every module, unet.py:289-372; SURVEY 8(d): 1.0008e9 at
 * dim 32) -- the unit of every end-to-end TFLOP/s figure.  Inference plans execute less where nn.Upsample + conv3x3 is folded into four 2x2
 * kernels (9 -> 4 taps per output pixel): fc_unet_op_info reports what each launch executes. */
double fc_unet_flops_per_sample(const fc_unet* u);
/* Launch i of the plan: kernel family, the reference module it implements, the FLOPs per sample it executes. */
int fc_unet_op_info(const fc_unet* u, int i, const char** kernel, const char** module, double* flops_per_sample);
/* Algorithmic HBM bytes of launch i (convolution launches; 0 for the others): per sample = every input / output / residual element
 * once, per launch = the weights once -- the figure bench.py's roofline.traffic is read against. */
int fc_unet_op_bytes(const fc_unet* u, int i, double* bytes_per_sample, double* bytes_per_launch);
/* Measurement hook for bench.py: average device milliseconds of every launch of the plan at `batch` rows, each
 * timed alone with HIP events on `stream` over `repeats` back-to-back launches (`batch` is clamped to the rows of one
 * chain, see fc_unet_chains).  Synchronises. */
int fc_unet_profile_ops(fc_unet* u, int batch, int repeats, float* ms_out, int n_out, void* stream);

/* The sinusoidal frequency table exp(-k ln(1e4)/(dim/2-1)) (unet.py:26-27).  The library builds it in double
 * precision; a host that wants the exact fp32 values its own framework produces may override it. */
int fc_unet_set_time_freqs(fc_unet* u, const float* freqs_host, int n);

/* ---- debug / test hooks: not part of the drop-in surface --------------------------------------- */
/* Device pointer + NHWC extent of an internal activation of the current plan, by reference module name
 * ("downs.0.0", "downs.0.2", "mid_attn", "ups.3.3", ...). */
int fc_unet_debug_tensor(const fc_unet* u, const char* name, const float** ptr, int* channels, int* height, int* width);
int fc_debug_copy(void* dst_dev, const void* src_dev, int64_t bytes, void* stream);
/* Memory diagnostics (csrc/devmem.hip): with the switch on (or FLOCODER_AMD_POISON=1 in the environment) every long-lived buffer the
 * library allocates FROM THEN ON -- arenas, statistics, integrator state, parameter stores, job tables -- is filled with 0xFFFFFFFF words
 * (NaN as fp32) and fenced by 64 KiB of the same pattern on both sides: a value read before it was written, or read past a buffer's end,
 * poisons the result instead of depending on what the memory held before; fc_debug_poison_check counts the buffers whose fences were
 * WRITTEN (fc_last_error names them) and synchronises the device. */
int fc_debug_set_poison(int on);
int fc_debug_poison_check(int* corrupted, int* live);
/* Diagnostics: subsequent pipelined-conv launches write shader-clock phase stamps to buf_dev
 * ([block][wave][16] uint64); NULL switches them off again. */
int fc_debug_set_conv_stamps(void* buf_dev);
/* Diagnostics: fc_unet_profile_ops runs plan entry `op_index` once more with the stamps of fc_debug_set_conv_stamps going to buf_dev
 * (the phase timeline of ONE launch of a real plan, fused tail included: tools/fin_stamps.py); NULL switches it off. */
int fc_debug_set_stamp_op(int op_index, void* buf_dev);
/* One implicit-GEMM launch on caller tensors (NHWC activations, OIHW weights as torch stores them).
 * Synchronises; allocates a scratch weight buffer. stats_out [B][G][T][2] gets (mean, M2) partials, T and the
 * per-slot count come back through stats_T / stats_nt. tile_cfg = -1 picks automatically.  repeats > 0 additionally
 * times that many back-to-back launches with HIP events (average milliseconds in *ms_out). */
int fc_debug_set_conv_precision(int mode);    /* test hook: arithmetic of fc_debug_conv launches (0 fp32, 1 split-bf16) */
int fc_debug_conv(const float* src0_nhwc, int c0, const float* src1_nhwc, int c1, const float* w_oihw_dev, const float* bias_dev,
                  const float* add_nhwc, float* out_nhwc, float* stats_out, int groups_out, int* stats_T, float* stats_nt, int batch,
                  int hs, int ws, int cout, int ksize, int pad, int stride, int upsample, int out_act, int tile_cfg, int repeats,
                  float* ms_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flow training step  (replaces train_flow.py:346-397: interpolation, loss.backward() through flocoder/unet.py,
 * clip_grad_norm_, torch.optim.Adam.step, EMA.update :46-54).  All vectors are caller-owned device memory.
 * ---------------------------------------------------------------------------------------------- */
/* Forward plan (fc_unet_reserve) plus the backward plan over the same arena. */
int fc_unet_train_reserve(fc_unet* u, int max_batch, int height, int width);
/* Gradients of all parameters for the LAST fc_unet_forward(x, time, ids) on this handle (same arguments again), given d(out):
 * grads_flat_dev[numel] in the parameter table's layout (fc_unet_param_info offsets; padding stays zero).  With ids == NULL
 * the class_cond_mlp.* range receives no gradient (zeros); an optimiser must skip it as torch skips p.grad is None. */
int fc_unet_backward(fc_unet* u, const float* x_dev, const float* time_dev, const int64_t* class_ids_dev, const float* d_out_dev,
                     float* grads_flat_dev, int64_t numel, int batch, int height, int width, void* stream);
/* The same with mask conditioning (cond['mask_cond'], unet.py:298-305,336-340,360-364) and the input gradients: mask_dev /
 * mask_is_ones as in fc_unet_forward; dx_out_dev / dmask_out_dev [B,C,H,W] receive d(x) / d(mask) when not NULL (the inpainting
 * step needs both: source and mask reach x through mask_blending, train_flow.py:146-147). */
int fc_unet_backward_ex(fc_unet* u, const float* x_dev, const float* time_dev, const int64_t* class_ids_dev, const float* mask_dev,
                        int mask_is_ones, const float* d_out_dev, float* grads_flat_dev, int64_t numel, float* dx_out_dev,
                        float* dmask_out_dev, int batch, int height, int width, void* stream);
/* The same in two parts, for data-parallel training that overlaps the gradient all-reduce with the backward (train_flow.py:371 under
 * DDP): part 0 runs the data-gradient chain from the output through mid_block1 and every weight / norm / FiLM gradient of final_*,
 * ups.* and mid_*, after which [split_offset, numel) of grads_flat_dev is final (fc_unet_grad_buckets); part 1 runs the rest and completes
 * [0, split_offset).  first_part = 0, last_part = 1 is fc_unet_backward_ex.  Part 1 must follow part 0 of the same forward. */
int fc_unet_backward_parts(fc_unet* u, const float* x_dev, const float* time_dev, const int64_t* class_ids_dev, const float* mask_dev,
                           int mask_is_ones, const float* d_out_dev, float* grads_flat_dev, int64_t numel, float* dx_out_dev,
                           float* dmask_out_dev, int batch, int height, int width, int first_part, int last_part, void* stream);
/* Ask for (1) / do without (0, default) the two-bucket form of the backward plan; takes effect at the next fc_unet_train_reserve.  One
 * process gains nothing from it (the deferred table launches run twice); a data-parallel trainer sets it once. */
int fc_unet_set_grad_buckets(fc_unet* u, int on);
/* Number of gradient buckets of the current backward plan (0 none, 1, 2) and the flat offset where the early bucket starts. */
int fc_unet_grad_buckets(const fc_unet* u, int64_t* split_offset);
/* The backward reads the activations the last training forward left in the handle's single arena.  Every call that writes the arena
 * (fc_unet_forward, fc_unet_integrate, fc_unet_profile_ops, a re-plan by fc_unet_reserve) moves this counter; a caller that keeps
 * several forwards in flight (autograd with two micro-batches, gradient accumulation) compares the value it saw after ITS forward
 * with the current one and re-runs the forward when they differ.  fc_unet_backward[_ex] itself fails with FC_E_STATE when the
 * arena does not hold a training forward of `batch` rows.  (loss.backward() over a graph built earlier, train_flow.py:358-371.) */
uint64_t fc_unet_arena_serial(const fc_unet* u);
/* [lo, hi) of class_cond_mlp.* inside the flat table (0,0 without classes). */
int fc_unet_class_param_range(const fc_unet* u, int64_t* lo, int64_t* hi);
/* x = (1-t) source + t target ; v* = target - source   (train_flow.py:350-353), t per sample. */
int fc_flow_interp(const float* source_dev, const float* target_dev, const float* t_dev, float* x_out_dev, float* v_out_dev, int batch,
                   int64_t per_sample, void* stream);
/* The per-step prologue of train_flow.py:346-357 in one launch: t = warp_time(u (1 - t_eps) + t_eps, s = warp_s) (sampling.py:23-33,
 * the same rounded operations torch runs), time = t * t_scale (the U-Net's time input), x / v* as fc_flow_interp with
 * target row pairing[b] when pairing_dev != NULL (the OT pairing's gather, train_flow.py:350), and the range check of the class ids:
 * bit 0 of *id_flag_dev is set (never cleared) when an id lies outside [0, n_classes) -- nn.Embedding's IndexError, reported when the
 * host next reads the flag; bit 1 when a pairing entry lies outside [0, batch) (torch's target[ot_indices] raises IndexError; the row
 * then reads its own target instead of memory out of bounds).  pairing_dev, class_ids_dev and id_flag_dev may be NULL. */
int fc_flow_prepare(const float* source_dev, const float* target_dev, const int64_t* pairing_dev, const float* u_dev, float t_eps, float warp_s,
                    float t_scale, const int64_t* class_ids_dev, int n_classes, float* t_out_dev, float* time_out_dev, float* x_out_dev,
                    float* v_out_dev, int* id_flag_dev, int batch, int64_t per_sample, void* stream);
/* loss = mean((v - v*)^2) (train_flow.py:359) and, when dv_out_dev != NULL, its gradient 2 (v - v*) / numel.  ws: 256 floats. */
int fc_mse_loss_grad(const float* v_dev, const float* target_dev, float* dv_out_dev, float* loss_out_dev, float* ws256_dev, int64_t numel,
                     void* stream);
/* clip_grad_norm_ (train_flow.py:392): out[0] = 2-norm over both ranges, out[1] = min(1, max_norm / (norm + 1e-6)). */
int fc_grad_clip_coef(const float* grads_dev, int64_t numel, const float* grads2_dev, int64_t numel2, float max_norm, float* norm_coef_out_dev,
                      float* ws256_dev, void* stream);
/* One torch.optim.Adam step (no weight decay / amsgrad) on grads * (*clip_coef_dev), bias corrections for step count `step`
 * (1-based), followed by ema = decay ema + (1 - decay) p.  apply_adam == 0: EMA only (parameters without a gradient). */
int fc_adam_ema_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* ema_dev, int64_t numel,
                     const float* clip_coef_dev, float lr, float beta1, float beta2, float eps, int step, float ema_decay, int apply_adam,
                     void* stream);
/* The same, skipped entirely (no Adam, no EMA) when *skip_flag_dev != 0 (NULL: never): FlowTrainer passes fc_flow_prepare's flag, so a
 * step whose class ids / pairing were out of range changes nothing before the host has seen the flag and raised. */
int fc_adam_ema_step_guarded(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* ema_dev, int64_t numel,
                             const float* clip_coef_dev, float lr, float beta1, float beta2, float eps, int step, float ema_decay, int apply_adam,
                             const int* skip_flag_dev, void* stream);
/* test hook: weight / bias gradient of one convolution (NHWC operands, dw in [O][I][KH][KW]) */
int fc_debug_conv_wgrad(const float* src0, int c0, const float* src1, int c1, const float* dy, int cout, int batch, int hs, int ws, int ksize,
                        int pad, int stride, int upsample, float* dw_out, float* db_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SD-VAE codec  (replaces flocoder/codecs.py:631-663 SD_VAE_Wrapper -> diffusers AutoencoderKL, sd-vae-ft-mse config)
 * ---------------------------------------------------------------------------------------------- */
typedef struct fc_vae fc_vae;
/* device >= 0: a gfx950 device; device < 0: description only (parameter table). */
int fc_vae_create(int device, fc_vae** out);
void fc_vae_destroy(fc_vae* v);
/* Parameter table: names/shapes are the upstream AutoencoderKL state_dict keys ("encoder.conv_in.weight", ...). */
int fc_vae_param_count(const fc_vae* v);
int64_t fc_vae_param_numel(const fc_vae* v);
int fc_vae_param_info(const fc_vae* v, int i, const char** name, int64_t shape[4], int64_t* offset);
int fc_vae_load_params(fc_vae* v, const float* flat, int64_t numel, int on_device, void* stream);
/* Plans + arenas.  Images HxW (powers of two >= 64), latents (H/8)x(W/8). */
int fc_vae_reserve_encode(fc_vae* v, int max_batch, int height, int width);
int fc_vae_reserve_decode(fc_vae* v, int max_batch, int lat_height, int lat_width);
/* mean = vae.encode(x).latent_dist.mean  (codecs.py:642): x_dev [B,3,H,W] -> mean_out_dev [B,4,H/8,W/8]; no scaling factor. */
int fc_vae_encode(fc_vae* v, const float* x_dev, float* mean_out_dev, int batch, int height, int width, void* stream);
/* x = vae.decode(z).sample  (codecs.py:651): z_dev [B,4,h,w] -> x_out_dev [B,3,8h,8w]. */
int fc_vae_decode(fc_vae* v, const float* z_dev, float* x_out_dev, int batch, int lat_height, int lat_width, void* stream);
/* Arithmetic of the plans built from now on: 0 = exact fp32 (default; what every parity test and the headline benchmark use), 1 = split-bf16
 * -- each fp32 operand as bf16 hi + lo, a product as hi*hi + hi*lo + lo*hi on the bf16 matrix pipe with fp32 accumulation (~1e-5 relative per
 * layer): an OPT-IN for callers that decode many images and accept that (sampling.py:150-183 decode_latents).  No counterpart upstream.
 * Changing the mode drops the current plans. */
int fc_vae_set_precision(fc_vae* v, int mode);
double fc_vae_flops_per_sample(const fc_vae* v, int decode);
int fc_vae_plan_launches(const fc_vae* v, int decode);
/* Measurement only: launch i of the encode / decode plan, and every launch timed alone (see fc_unet_profile_ops). */
int fc_vae_op_info(const fc_vae* v, int decode, int i, const char** kernel, const char** module, double* flops_per_sample);
int fc_vae_profile_ops(fc_vae* v, int decode, const float* in_dev, float* out_dev, int batch, int repeats, float* ms_out, int n_out,
                       void* stream);
int fc_vae_op_bytes(const fc_vae* v, int decode, int i, double* bytes_per_sample, double* bytes_per_launch);   /* as fc_unet_op_bytes */

/* ------------------------------------------------------------------------------------------------
 * VQVAE codec, encode / decode  (replaces flocoder/codecs.py:386-525 VQVAE.encode / VQVAE.decode with
 * EncDecResidualBlock :150-214, AttnBlock :53-89, Decoder :217-316, SpatialNonLocalAttention :337-383; no NATTEN, eval)
 * quantize() (third-party ResidualVQ) is not part of this library.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fc_vqvae fc_vqvae;
/* Constructor arguments of VQVAE (codecs.py:399-405); widths must be multiples of 4.  device < 0: description only. */
int fc_vqvae_create(int in_channels, int hidden_channels, int num_downsamples, int internal_dim, int vq_embedding_dim, int decoder_nonlocal,
                    int device, fc_vqvae** out);
/* The same with NATTENBlocks (codecs.py:93-145; 7x7 neighbourhood attention, 8 heads) in the EncDecResidualBlocks the reference
 * gives them to when the natten package is importable: the last two encoder levels, the bottleneck block, the first decoder block
 * (decoder_nonlocal == 0) and the first upsampling level (codecs.py:266,275-278,414-429).  natten_layout: 0 = none (fc_vqvae_create),
 * 1 = windows over image rows x columns per head (the intended reading), 2 = what natten >= 0.20 computes from the reference's
 * [B, heads, H, W, d] tensors (windows over head index x image rows, per image column).  PARITY UNPINNED: the package is absent. */
int fc_vqvae_create_ex(int in_channels, int hidden_channels, int num_downsamples, int internal_dim, int vq_embedding_dim,
                       int decoder_nonlocal, int natten_layout, int device, fc_vqvae** out);
/* 2-D neighbourhood attention by itself: qkv_nhwc_dev [B][H][W][3C] (channel = which*C + head*(C/heads) + d), out [B][H][W][C],
 * kernel_size odd <= 7, windows clamped (shifted) at the borders, scale (C/heads)^-0.5, optional scalar gate gamma_dev[0].
 * Replaces natten.functional.na2d / na2d_qk + softmax + na2d_av (codecs.py:130-135). */
int fc_na2d(const float* qkv_nhwc_dev, float* out_nhwc_dev, const float* gamma_dev, int batch, int height, int width, int channels,
            int heads, int kernel_size, int layout_mode, void* stream);
void fc_vqvae_destroy(fc_vqvae* v);
/* Parameter table: the reference's state_dict keys ("encoder.0.conv1.weight", "decoder.layers.0.q_proj.weight", ...). */
int fc_vqvae_param_count(const fc_vqvae* v);
int64_t fc_vqvae_param_numel(const fc_vqvae* v);
int fc_vqvae_param_info(const fc_vqvae* v, int i, const char** name, int64_t shape[4], int64_t* offset);
int fc_vqvae_load_params(fc_vqvae* v, const float* flat, int64_t numel, int on_device, void* stream);
/* Plans + arenas.  Images HxW (powers of two), latents (H >> num_downsamples) x (W >> num_downsamples), >= 4 per side. */
int fc_vqvae_reserve_encode(fc_vqvae* v, int max_batch, int height, int width);
int fc_vqvae_reserve_decode(fc_vqvae* v, int max_batch, int lat_height, int lat_width);
/* z = vqvae.encode(x)  (codecs.py:492-502): x_dev [B,in_channels,H,W] -> z_out_dev [B,vq_embedding_dim,h,w] (pre-quantisation). */
int fc_vqvae_encode(fc_vqvae* v, const float* x_dev, float* z_out_dev, int batch, int height, int width, void* stream);
/* x = vqvae.decode(z_q)  (codecs.py:523-525, noise_strength 0): z_dev [B,vq_embedding_dim,h,w] -> x_out_dev [B,in_channels,H,W]. */
int fc_vqvae_decode(fc_vqvae* v, const float* z_dev, float* x_out_dev, int batch, int lat_height, int lat_width, void* stream);
int fc_vqvae_set_precision(fc_vqvae* v, int mode);     /* as fc_vae_set_precision */
double fc_vqvae_flops_per_sample(const fc_vqvae* v, int decode);
/* Measurement only (bench.py's config-5 leg): launches of the encode / decode plan, launch i's kernel family, module, algorithmic FLOPs
 * per sample and HBM bytes (per sample / per launch; 0 where not stated), and every launch timed alone as in fc_unet_profile_ops. */
int fc_vqvae_plan_launches(const fc_vqvae* v, int decode);
int fc_vqvae_op_info(const fc_vqvae* v, int decode, int i, const char** kernel, const char** module, double* flops_per_sample,
                     double* bytes_per_sample, double* bytes_per_launch);
int fc_vqvae_profile_ops(fc_vqvae* v, int decode, const float* in_dev, float* out_dev, int batch, int repeats, float* ms_out, int n_out,
                         void* stream);
/* z_q, indices = ResidualVQ(z) in inference form (replaces VQVAE.quantize, codecs.py:504-521 -> vector_quantize_pytorch.ResidualVQ,
 * third party, parity unpinned): per level the nearest codeword of the running residual, z_q = their sum.
 * z_dev / zq_out_dev [B,dim,hw] (NCHW with hw = h*w); codebooks_dev [levels][codebook_size][dim]; indices_out_dev [B*hw][levels]
 * int64 or NULL. */
int fc_rvq_quantize(const float* z_dev, const float* codebooks_dev, float* zq_out_dev, int64_t* indices_out_dev, int batch, int dim, int hw,
                    int codebook_size, int levels, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Inpainting conditioning  (replaces flocoder/inpainting.py:161-253 MaskEncoder / mask_blending)
 * ---------------------------------------------------------------------------------------------- */
typedef struct fc_mask_encoder fc_mask_encoder;
int fc_mask_encoder_create(int device, fc_mask_encoder** out);        /* device < 0: description only */
void fc_mask_encoder_destroy(fc_mask_encoder* m);
int fc_mask_encoder_param_count(const fc_mask_encoder* m);            /* names: "layers.0.conv1.weight", ... */
int64_t fc_mask_encoder_param_numel(const fc_mask_encoder* m);
int fc_mask_encoder_param_info(const fc_mask_encoder* m, int i, const char** name, int64_t shape[4], int64_t* offset);
int fc_mask_encoder_load_params(fc_mask_encoder* m, const float* flat, int64_t numel, int on_device, void* stream);
int fc_mask_encoder_reserve(fc_mask_encoder* m, int max_batch, int height, int width);
/* MaskEncoder.forward (inpainting.py:235-245): mask_pixels_dev [B,1,H,W] fp32 -> mask_latents_dev [B,4,H/16,W/16]. */
int fc_mask_encoder_forward(fc_mask_encoder* m, const float* mask_pixels_dev, float* mask_latents_dev, int batch, int height, int width,
                            void* stream);
/* Parameter gradients of the LAST fc_mask_encoder_forward on this object (same mask_pixels again, its output in mask_latents_dev)
 * for d(mask_latents): grads_flat_dev[numel] in the parameter-table layout, overwritten or (accumulate != 0) added to.  Replaces
 * the MaskEncoder leg of loss.backward() in the inpainting step (train_flow.py:312-318,361-371). */
int fc_mask_encoder_backward(fc_mask_encoder* m, const float* mask_pixels_dev, const float* mask_latents_dev, const float* d_latents_dev,
                             float* grads_flat_dev, int64_t numel, int accumulate, int batch, int height, int width, void* stream);
/* mask_blending (inpainting.py:250-253): out = source + mask * (noise - source), elementwise over numel floats. */
int fc_mask_blend(const float* source_dev, const float* mask_dev, const float* noise_dev, float* out_dev, int64_t numel, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Greedy OT pairing  (replaces flocoder/ot.py:63-84 compute_ot_pairing)
 * ---------------------------------------------------------------------------------------------- */
/* source_dev/target_dev [B,D] fp32; dist_ws_dev workspace of B*B floats; perm_out_dev [B] int64. */
int fc_ot_pairing(const float* source_dev, const float* target_dev, int batch, int64_t dim, float* dist_ws_dev,
                  int64_t* perm_out_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Parity metrics  (replace flocoder/metrics.py:40-54 sinkhorn_loss: geomloss SamplesLoss("sinkhorn", p=2, blur=0.05))
 * ---------------------------------------------------------------------------------------------- */
/* Debiased Sinkhorn divergence S_eps(x, y) between the point clouds x_dev [n][dim] and y_dev [m][dim] (fp32, uniform weights),
 * cost |x-y|^2/2, eps-scaling from diameter^2 down to blur^2 by factors scaling^2 (geomloss's tensorized algorithm, restated:
 * PARITY UNPINNED, the package is absent).  diameter <= 0: computed from the data as geomloss does.  Results on the host
 * (the call synchronises `stream`): *value_out_host, the diameter used, the number of eps steps.  n, m <= 8192. */
int fc_sinkhorn_divergence(const float* x_dev, const float* y_dev, int n, int m, int64_t dim, double blur, double scaling,
                           double diameter, double* value_out_host, double* diameter_out_host, int* iterations_out_host,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOCODER_AMD_H */
