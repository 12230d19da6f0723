/* swapnet_hip.h -- C ABI of libswapnet_hip.so, the MI355X (gfx950) back end of the SwapNet
 * two-stage GAN training hot path.
 *
 * The reference (andrewjong/SwapNet) has no native boundary of its own: it is pure Python
 * on torch.nn.  Each entry point below therefore cites the reference
 * *Python* interface it stands in for; swapnet_amd/_C.py is the ctypes binding and
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; swn_last_error() returns
 *     the message of the last failure on the calling thread (the Python shim re-raises it
 *     as the exception type the reference would raise: ValueError / NotImplementedError /
 *     RuntimeError).
 *   - all tensor arguments are raw DEVICE pointers to fp32 (NCHW, contiguous -- the layout
 *     of the reference's torch tensors) unless the name says host; the library borrows them
 *     for the duration of the call and never frees caller memory.
 *   - one context per process/GPU, driven by one host thread; all work is enqueued on the
 *     context's HIP stream.
 *   - arithmetic: storage and accumulation are fp32 throughout.  The large implicit-GEMM kernels form each fp32 product
 *     on the 16-bit matrix cores from a split of both operands into two fp16 planes, x * 2^k = h + l with k chosen per TENSOR
 *     from its amax: h + l carries 22 of fp32's 24 mantissa bits for every element within 2^-14 of the tensor's largest and
 *     progressively fewer below that (absolute floor amax * 2^-37), and three of the four partial products are formed (the
 *     dropped l*l term is < 2^-22 relative).  This is NOT an exact representation of fp32: it is held to the f32-MFMA
 *     kernels' error on well-conditioned and on heavy-tailed operands by tests/test_ops.py (rel-L2 and element-wise).
 *     Environment: SWN_PC_PLANES=3 / SWN_WGRAD_PLANES=3 (three bf16 planes by truncation -- that split IS exact, 24 bits
 *     in, 24 out -- six MFMAs, dropped terms < 2^-24), SWN_SPLIT=0 (v_mfma_f32_32x32x2_f32); DESIGN.md section 4.
 */
#ifndef SWAPNET_HIP_H
#define SWAPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct swn_ctx swn_ctx;
typedef struct swn_model swn_model;

int swn_abi_version(void);   /* 2: swn_hyper gained d_b1, d_b2; 3: gp_mode, lambda_gp; 4: swn_route_*, swn_model_step_captured, swn_model_create_shared;
                                5: swn_ctx_attach_comm, swn_model_step_dp; 6: swn_probe_mfma */
const char* swn_last_error(void);
/* 1 when this library executes on a HIP device (libswapnet_hip.so), 0 for the CI simulator */
int swn_is_device_build(void);

/* ---- context ------------------------------------------------------------------------
 * replaces BaseModel.__init__'s device selection (models/base_model.py:36-40).
 * create_stream == 0: all work is enqueued on `hip_stream` exactly as given -- pass
 * torch.cuda.current_stream().cuda_stream so torch-side copies / allocator reuse stay ordered
 * with the library's kernels (NULL is the legacy default stream, which IS torch's default).
 * create_stream != 0: the library creates and owns a private non-blocking stream. */
int swn_ctx_create(int device, void* hip_stream, int create_stream, size_t workspace_bytes, swn_ctx** out);
int swn_ctx_destroy(swn_ctx* ctx);
/* The context runs weight-gradient work and the derived-weight refresh on an internal second stream
 * (DESIGN.md section 4).  on = 0 keeps everything in order on the caller's stream (used by bench.py's
 * per-kernel roofline pass so that kernel durations are not inflated by co-running kernels); results
 * are identical either way.  Synchronises. */
int swn_ctx_set_overlap(swn_ctx* ctx, int on);
/* discriminators.define_D(input_nc, 64, opt.discriminator, opt.n_layers_D, opt.norm) (modules/discriminators.py:45-88,
 * models/base_gan.py:147-149): the number of stride-2 levels of the NLayerDiscriminator (:91-136) of every model created on
 * this context AFTERWARDS; 3 = "basic", the 70x70 PatchGAN (default).  1..5; the input must keep >= 3 pixels per side after
 * the stride-2 levels.  0 = --discriminator pixel: the 1x1 PixelDiscriminator (:139-170; three 1x1 convs, a prediction per
 * pixel, state_dict keys net.{0,2,5}.{weight,bias}).  The gradient-penalty modes exist for every PatchGAN depth, not for the
 * PixelDiscriminator (swn_model_set_hyper fails). */
int swn_ctx_set_patchgan_layers(swn_ctx* ctx, int n_layers);
int swn_ctx_sync(swn_ctx* ctx);
int swn_ctx_bytes_allocated(swn_ctx* ctx, size_t* out);

/* per-launch timing of the implicit-GEMM kernels with HIP events on the context's stream
 * (bench.py's roofline leg).  report: one line per kernel variant "<name> <launches> <total_ms>
 * <total_flops>"; returns the number of bytes the full report needs. */
int swn_prof_enable(int on);
int swn_prof_reset(void);
int swn_prof_report(char* buf, int len);
/* What the matrix pipe of THIS chip sustains for the arithmetic of the ring kernels ("Builder must confirm peaks on
 * the box"; bench.py's roofline object quotes it beside the nominal 2.5 PFLOP/s): `iters` rounds of the product loop's twelve
 * v_mfma_f32_32x32x16_f16 per wave (4 accumulators x 3 terms) on operand fragments that stay in registers -- no LDS, no HBM --
 * 1024 workgroups x 4 waves, zeros = 0: pseudo-random fp16 operands (what a training step multiplies), 1: all-zero operands (the
 * clock the same code gets when the multipliers do not toggle).  out[0] = fp16 TFLOP/s, out[1] = the shader clock in GHz measured
 * INSIDE the kernel (s_memtime over s_memrealtime), out[2] = ms per launch, out[3] = matrix-pipe occupancy at that clock (0..1).
 * Replaces nothing in the reference (measurement only); synchronises. */
int swn_probe_mfma(swn_ctx* ctx, int zeros, int iters, float* out4);
/* Routing trace: which kernel family / algorithmic form each layer takes under the CURRENT environment (the SWN_* switches of
 * DESIGN.md section 4 select among kernels; models/base_gan.py:194-203 is the step whose launches are listed).  While on, every
 * implicit-GEMM launch (name, M, N, K, batch, split schedule) and every Winograd transform is recorded against the layer
 * (state_dict prefix) and phase (f forward, b backward, r operand refresh) that issued it; report = the distinct lines in
 * first-use order, returns the bytes the full report needs.  tests/ compare the list of a scrubbed-environment run (what
 * bench.py times) with the list of the run they check against the oracle. */
int swn_route_trace(int on);
int swn_route_report(char* buf, int len);

/* ---- models ---------------------------------------------------------------------------
 * swn_warp_model_create    <-> models.create_model(opt) with --model warp
 *                              (models/__init__.py:33-44, models/warp_model.py:42-76,
 *                               models/base_gan.py:130-176): WarpModule generator and, when
 *                               is_train, the 22-channel conditional PatchGAN + both AdamW states.
 * swn_texture_model_create <-> --model texture (models/texture_model.py:54-111): TextureModule
 *                              generator (RoIAlign -> UNetDown -> pix2pix U-Net), PatchGAN,
 *                              VGG16 perceptual network. */
int swn_warp_model_create(swn_ctx* ctx, int batch, int height, int width, int is_train, float dropout,
                          swn_model** out);
int swn_texture_model_create(swn_ctx* ctx, int batch, int height, int width, int is_train, int num_roi,
                             swn_model** out);
/* The same with the reference's representation options (options/base_options.py:75-105, models/warp_model.py:49-55,
 * models/texture_model.py:94-109): body_channels = 3 for --body_representation rgb, --body_channels (12) for labels;
 * cloth_channels = --cloth_channels (19) for --cloth_representation labels, 3 for rgb.  The plain constructors above
 * are the defaults (3, 19).  swn_model_destroy releases every device buffer the model allocated. */
int swn_warp_model_create_ex(swn_ctx* ctx, int batch, int height, int width, int is_train, float dropout,
                             int body_channels, int cloth_channels, swn_model** out);
int swn_texture_model_create_ex(swn_ctx* ctx, int batch, int height, int width, int is_train, int num_roi,
                                int cloth_channels, swn_model** out);
/* A second model on the SAME training state: it uses `sharer`'s parameter arenas (weights, gradients, both Adam moments, step
 * counters -- per network one flat buffer each) and owns only its activations and derived operands.  For the reference's loops
 * that is the model of another batch size: the last, smaller batch of an epoch (train.py:62-64 with drop_last off) or the batch-1
 * pass of inference.py:67 beside a training model -- without a second copy of the state and without copying it back and forth.
 * Same kind, channel options and PatchGAN depth as the sharer; hyper-parameters are copied at creation (set them on both
 * afterwards).  The sharer's buffers live until the last sharing model is destroyed, whatever the order of swn_model_destroy. */
int swn_model_create_shared(swn_model* sharer, int batch, int height, int width, swn_model** out);
int swn_model_destroy(swn_model* m);

/* hyper-parameters = the opt.* fields read by the step (models/base_gan.py:87-120,
 * optimizers/__init__.py:25-59, models/warp_model.py:27-34, models/texture_model.py:31-50) */
typedef struct swn_hyper {
  float lr, d_lr, weight_decay, d_weight_decay, b1, b2;
  float lambda_gan, lambda_ce, lambda_l1, lambda_content, lambda_style;
  int gan_mode;      /* 0 vanilla, 1 lsgan, 2 wgan */
  int warp_mode_ce;  /* 1 = --warp_mode ce (generator only) */
  float grad_scale;  /* multiplies every loss gradient: 1/world_size under data parallelism so the
                        RCCL all-reduce(SUM) of the arenas yields the mean without an extra pass */
  float d_b1, d_b2;  /* AdamW betas of optimizer_D (negative = same as b1 / b2; 0 is a valid beta): the two torch optimizers of the
                        reference are independent objects (models/base_gan.py:87-120) */
  int gp_mode;       /* gradient penalty of --gan_mode (modules/loss.py:133-184): 0 none, 1 wgan-gp (with gan_mode 2),
                        2 dragan-gp, 3 dragan-lp (with gan_mode 0); warp model only */
  float lambda_gp;   /* --lambda_gp (models/base_gan.py:54-58) */
} swn_hyper;
int swn_model_set_hyper(swn_model* m, const swn_hyper* h);

/* parameters: net 0 = generator, 1 = discriminator, 2 = VGG16 features (texture model).
 * Names and shapes are the reference's state_dict() keys (SURVEY.md 8(b) "Checkpoint
 * format").  which: 0 weight, 1 grad, 2 exp_avg, 3 exp_avg_sq (torch.optim.AdamW state). */
int swn_model_param_count(swn_model* m, int net, int* out);
int swn_model_param_info(swn_model* m, int net, int index, char* name, int name_len, int shape[4], int* ndim);
int swn_model_param_set(swn_model* m, int net, int which, const char* name, const float* dev_src);
int swn_model_param_get(swn_model* m, int net, int which, const char* name, float* dev_dst);
int swn_model_optim_step_get(swn_model* m, int net, int* step);   /* AdamW `step` counter */
int swn_model_optim_step_set(swn_model* m, int net, int step);

/* BaseModel.set_input (models/warp_model.py:99-104, models/texture_model.py:113-119).
 * warp slots: 0 bodys (B,3,H,W), 1 input_cloths (B,19,H,W), 2 target_cloths (B,19,H,W)
 * texture slots: 0 input_textures (B,3,H,W), 1 rois (B,R,4 as N=B,C=R,H=4,W=1), 2 cloths
 * (B,19,H,W), 3 target_textures (B,3,H,W) */
int swn_model_set_input(swn_model* m, int slot, const float* dev_nchw, int n, int c, int h, int w);
/* Same slots, but for the cloth segmentations (warp 1, 2; texture 2) the caller hands over the
 * integer label map (B,H,W) int32 -- the on-disk format, datasets/data_utils.py:298-343 -- and the
 * one-hot expansion (label 0 -> all-zero vector) happens on the device: 1/19 of the H2D bytes of
 * set_input(one_hot).  Bit-identical to to_onehot_tensor + set_input. */
int swn_model_set_input_labels(swn_model* m, int slot, const int32_t* dev_labels, int n, int h, int w);
/* slot 0: self.fakes (B,19,H,W) warp / (B,3,H,W) texture */
int swn_model_get_output(swn_model* m, int slot, float* dev_nchw);
/* named intermediate activation (debug / per-level parity tests), copied out as NCHW */
int swn_model_get_tap(swn_model* m, int net, const char* name, float* dev_nchw, int shape[4]);

/* gradient w.r.t. a named activation as left by the last backward pass through `net` (diagnostics) */
int swn_model_get_tap_grad(swn_model* m, int net, const char* name, float* dev_nchw, int shape[4]);

/* nn.Dropout sites of the generator in forward order (WarpModule: body_down4, cloth_down5, cloth_down6 and the
 * four ResidualBlocks, modules/swapnet_modules.py:37,46-47,58 / modules/layers.py:22-23,136; pix2pix U-Net:
 * the three inner ngf*8 blocks, modules/pix2pix_modules.py:251-252).  swn_model_dropout_mask writes the factor
 * (0 or 1/(1-p)) that a training-mode forward AND backward with `dropout_seed` apply at `site`, as an NCHW
 * tensor of `shape` (channel count = the padded channel count of the layer).  dst may be NULL to query the
 * shape.  Diagnostic: lets a parity test replay the exact masks in the CPU oracle. */
int swn_model_dropout_sites(swn_model* m, int net, int* count);
/* Diagnostic export for parity tests: the branch every piecewise-linear op of the last pass took.  LeakyReLU / ReLU
 * (kind 1): byte 1 where the activation output is > 0, i.e. the side the backward pass differentiates on; MaxPool2d
 * (kind 2): the window position 2*kh + kw that receives the gradient.  NCHW bytes on the device, `shape` = (N,C,H,W)
 * with C the buffer's (possibly padded) channel count; dev_nchw = NULL only queries shape / kind.  Sites are numbered
 * in forward order per network: net 0 = generator, 1 = discriminator of the D step (samples [0,B) generated, [B,2B)
 * real), 2 = discriminator as re-evaluated inside the G step, 3 = VGG16 on the generated image (texture model).
 * Why: an fp32 pass in a different summation order leaves a handful of pre-activations on the other side of zero and
 * each flip moves the gradient by O(1) of that element (rel-L2 ~1e-3 on PatchGAN at 256x256, for torch's own fp32
 * backward as much as for this library); with the pattern replayed in the float64 oracle the comparison is ~1e-5. */
int swn_model_act_sites(swn_model* m, int net, int* count);
int swn_model_act_pattern(swn_model* m, int net, int site, uint8_t* dev_nchw, int shape[4], int* kind);
int swn_model_dropout_mask(swn_model* m, int net, int site, uint64_t dropout_seed, float* dev_nchw, int shape[4],
                           float* p);

/* inference.py's warp -> texture hand-off (inference.py:94-126; .npz round trip of :140-149,169-180 /
 * datasets/data_utils.py:311-343) kept in HBM: warp forward -> argmax label map -> one-hot into the texture model's
 * cloth inputs -> texture forward.  Inputs are staged with swn_model_set_input on the two models beforehand (warp
 * slots 0,1; texture slots 0,1 -- texture slot 2 is produced here); the result is the texture model's output.
 * use_graph != 0: the first call runs eagerly and captures the sequence into a hipGraph, later calls replay it
 * (*graph_replayed = 1).  Both models must be inference models (is_train = 0) of one context and one (B,H,W). */
typedef struct swn_pipeline swn_pipeline;
int swn_pipeline_create(swn_model* warp, swn_model* texture, swn_pipeline** out);
int swn_pipeline_destroy(swn_pipeline* p);
int swn_pipeline_run(swn_pipeline* p, int use_graph, int* graph_replayed);
int swn_pipeline_labels(swn_pipeline* p, int32_t** dev_labels);   /* (B,H,W) int32 of the last run, library-owned */

/* Data-parallel texture stage: the style term is MSE(Gram(output), Gram(target)) with the Gram over the WHOLE batch viewed
 * as (B*C, H*W) (modules/losses/perceptual.py:6-10,58-63), so it couples the samples of all ranks.  The host all-gathers the
 * generated and target images ((n_total, 3, H, W) each; this rank's samples start at n0) after swn_model_forward; the next
 * backward_G evaluates the Gram over that global batch and back-propagates into the local samples -- the step then equals
 * the one-process big-batch step.  One-shot. */
int swn_model_set_style_context(swn_model* m, const float* all_out_nchw, const float* all_tgt_nchw, int n_total, int n0);
/* The random draws of the next gradient-penalty pass (modules/loss.py:141-147): alpha = torch.rand(B,1,1,1) as B
 * device floats; beta = torch.rand_like(conditioned_real) as (B, 22, H, W) in the reference's channel order (dragan
 * modes only).  Either may be NULL (the library then draws it from its own counter RNG).  One-shot. */
int swn_model_set_gp_random(swn_model* m, const float* alpha_dev, const float* beta_nchw_dev);
/* NLayerDiscriminator.forward(input) (modules/discriminators.py:134-136) as a standalone call on the model's
 * discriminator weights: x = conditioned input in the reference's channel order, (B, 22, H, W); pred receives
 * (B, 1, (H >> n) - 2, (W >> n) - 2), n = the PatchGAN depth (3: H/8-2).  Uses a private activation set: self.fakes and the staged batch stay untouched. */
int swn_model_discriminate(swn_model* m, const float* x_nchw, float* pred_nchw);
/* PerceptualLoss(use_style)(output, target) -> (content, style) (modules/losses/perceptual.py:49-66), texture model:
 * output / target (B, 3, H, W); out2 = device float[2] = { sum over the 5 VGG16 slices of MSE(normalised features),
 * 5 x MSE(Gram(output), Gram(target)) }.  d_output (optional, (B,3,H,W)) receives content_w * d(content)/d(output)
 * + style_w * d(style)/d(output) -- what autograd would return for content_w*content + style_w*style. */
int swn_model_perceptual(swn_model* m, const float* output_nchw, const float* target_nchw, int use_style, float* out2,
                         float content_w, float style_w, float* d_output_nchw);

/* BaseModel.forward (models/warp_model.py:106-107, models/texture_model.py:121-125) */
int swn_model_forward(swn_model* m, int training, uint64_t dropout_seed);
/* WarpModel.backward_D / TextureModel.backward_D (warp_model.py:109-139, texture_model.py:
 * 127-155); the two labels are the smooth-label scalars GANLoss draws (modules/loss.py:79-108) */
int swn_model_backward_D(swn_model* m, float label_fake, float label_real);
/* backward_G (warp_model.py:141-167, texture_model.py:157-180) */
int swn_model_backward_G(swn_model* m, float label_real);
/* backward_G in `nparts` buckets (part = 0 .. nparts-1, in this order) so the data-parallel exchange of
 * the generator gradients overlaps the rest of the backward pass: part 0 runs the loss head and the
 * last layers (decoder + late residual blocks), each further part the next group of earlier layers;
 * on return the arena range [ready_off, ready_off+ready_count) holds final gradients and can be handed to
 * the all-reduce while the next part runs.  The ranges tile the arena from its end to its start. */
int swn_model_backward_G_parts(swn_model* m, int* nparts);
int swn_model_backward_G_part(swn_model* m, float label_real, int part, size_t* ready_off, size_t* ready_count);
/* optimizer_{G,D}.step() (models/base_gan.py:199,203): fused AdamW over the net's arena */
int swn_model_optimizer_step(swn_model* m, int net);
/* the same AdamW step restricted to the arena range [off, off+count) (the ranges swn_model_backward_G_part reports):
 * under data parallelism a bucket is stepped as soon as its all-reduce has landed, while the next bucket is still
 * being back-propagated.  first != 0 on the first range of an optimizer step (advances AdamW's step counter). */
int swn_model_optimizer_step_range(swn_model* m, int net, size_t off, size_t count, int first);
/* BaseGAN.optimize_parameters (models/base_gan.py:194-203; warp_model.py:169-183) in one call.  With the context's second
 * stream on, optimizer_G.step() (:203) is applied bucket by bucket behind each bucket's weight gradients, under the
 * back-propagation of the earlier layers -- the same element-wise update from the same gradients: bit-identical to the
 * phased calls above (SWN_STREAM_ADAMW=0: one launch after the pass). */
int swn_model_step(swn_model* m, const float labels[3], int training, uint64_t dropout_seed);
/* Data parallelism with the exchange owned by this library (SURVEY.md 8(b) "allreduce_attach(rccl_comm)", 8(e); the reference has no
 * multi-GPU code: models/base_gan.py:194-203 is the step being sharded).  swn_ctx_attach_comm hands over an all-reduce entry point
 * with ncclAllReduce's signature -- RCCL's own symbol, taken from the librccl the process already holds, so no second copy of RCCL
 * is linked -- and the communicator to call it on (one rank per GPU, ncclCommInitRank done by the caller: swapnet_amd/parallel.py
 * NativeComm).  The library gives the exchange a HIP stream of its own and orders it against its compute streams with events it
 * records itself.  swn_model_step_dp is optimize_parameters under data parallelism in one call: backward_D, all-reduce(SUM) of D's
 * gradient arena, optimizer_D, then the generator's backward pass bucket by bucket, each bucket's all-reduce enqueued on the
 * exchange stream the moment its gradients are final and its AdamW on the same stream behind it; the compute stream joins the
 * exchange stream once, at the end of the step.  swn_hyper.grad_scale = 1 / world makes SUM the mean.  after_forward != 0: the
 * caller ran swn_model_forward itself (texture stage with the style term: swn_model_set_style_context goes in between).
 * fn == NULL detaches.  dtype / op are passed as ncclFloat32 (7) / ncclSum (0). */
typedef int (*swn_allreduce_fn)(const void* sendbuf, void* recvbuf, size_t count, int dtype, int op, void* comm, void* stream);
int swn_ctx_attach_comm(swn_ctx* ctx, swn_allreduce_fn fn, void* comm, int world_size);
int swn_model_step_dp(swn_model* m, const float labels[3], int training, uint64_t dropout_seed, int after_forward);
/* The same step recorded once into a hipGraph and replayed (BASELINE.json C5's "hipGraph-captured step"): the three label
 * draws of GANLoss (modules/loss.py:77-104), the dropout seed and both AdamW bias corrections travel through a 40-byte device
 * block uploaded in stream order, so one recorded launch sequence serves every step; the loss read-back of train.py:74
 * (swn_model_get_losses) stays the only synchronisation.  First call per `training` value runs eagerly, the second records,
 * later ones replay.  Results are bit-identical to swn_model_step.  Falls back to swn_model_step for the gradient-penalty
 * modes; under data parallelism use the phased calls (the exchange runs between them). */
int swn_model_step_captured(swn_model* m, const float labels[3], int training, uint64_t dropout_seed);
/* BaseModel.get_current_losses (models/base_model.py:139-147): host array of 9 floats
 * D, D_real, D_fake, G, G_gan, G_ce, G_l1, G_content, G_style, D_gp  (one small D2H + sync; n <= 10) */
int swn_model_get_losses(swn_model* m, float* host_out, int n);

/* data-parallel hook: flat gradient arena of a net (device pointer, float count) so the host
 * can all-reduce it with RCCL (torch.distributed, backend "nccl") between backward and step */
int swn_model_grad_arena(swn_model* m, int net, float** dev_ptr, size_t* count);
int swn_model_weight_arena(swn_model* m, int net, float** dev_ptr, size_t* count);
/* any of the four arenas (which: 0 weight, 1 grad, 2 exp_avg, 3 exp_avg_sq); the layout does
 * not depend on the batch shape, so a whole training state moves with flat copies */
int swn_model_arena(swn_model* m, int net, int which, float** dev_ptr, size_t* count);

/* ---- operator-level entry points (parity tests, integer work, inference helpers) --------- */
/* torchvision.ops.RoIAlign((128,128),1,1) as used at modules/swapnet_modules.py:166-168,234.
 * tex (B,C,H,W) NCHW, rois (B,R,4) -> out (B,R*C,PH,PW) NCHW */
int swn_op_roi_align(swn_ctx* ctx, const float* tex, int b, int c, int h, int w, const float* rois, int r,
                     int ph, int pw, float* out);
/* integer part only: idx (B*R,PH,PW,4) int32 = yl,yh,xl,xh ; valid (B*R,PH,PW) uint8 */
int swn_op_roi_align_indices(swn_ctx* ctx, const float* rois, int k, int h, int w, int ph, int pw, int32_t* idx,
                             uint8_t* valid);
/* util.decode_labels.decode_cloth_labels (util/decode_labels.py:24-55): (B,C,H,W) f32 -> (B,3,H,W) u8 */
int swn_op_decode_labels(swn_ctx* ctx, const float* x, int b, int c, int h, int w, uint8_t* rgb);
/* datasets/data_utils.py:322 (argmax for compress_and_save_cloth) and :330-343 (to_onehot_tensor) */
int swn_op_argmax_labels(swn_ctx* ctx, const float* x, int b, int c, int h, int w, int32_t* labels);
int swn_op_labels_to_onehot(swn_ctx* ctx, const int32_t* labels, int b, int c, int h, int w, float* out);
/* a single convolution through the MFMA implicit-GEMM kernel (or the naive checker):
 * kind 0 k4s2p1, 1 k3s1 reflect, 2 k4s1p1, 3 k3s1 zero-pad, 4 upsample-pad-conv tail, 5 k1s1 (PixelDiscriminator);
 * transposed!=0 -> ConvTranspose2d k4s2p1 (weight (Ci,Co,4,4)).  x (N,Ci,H,W), y NCHW.
 * what: 0 forward, 1 weight gradient (y = dY in, w = dW out), 2 input gradient (y = dY in, x = dX out) */
int swn_op_conv(swn_ctx* ctx, int kind, int transposed, int what, int naive, float* x, int n, int ci, int h, int w,
                float* wgt, int co, const float* bias, int act, float* y);
/* InstanceNorm(+act) forward / backward on NCHW tensors (modules/__init__.py:66-69) */
int swn_op_instance_norm_act(swn_ctx* ctx, const float* x, int n, int c, int h, int w, int act, float* y);
int swn_op_instance_norm_act_bwd(swn_ctx* ctx, const float* x, const float* dy, int n, int c, int h, int w, int act,
                                 float* dx);
/* WarpDataset's per-channel augmentation (datasets/warp_dataset.py:131-137, datasets/data_utils.py:346-361
 * per_channel_transform: an independent random flip / affine / perspective chain for each of the 19 cloth channels
 * of each sample) as ONE device gather over the batch instead of B*19*ntransforms PIL calls.  src, dst (B,C,H,W)
 * fp32; maps = device doubles [B*C][nmaps][9] = {kind, c0..c7} applied in index order: kind 0 identity, 1 affine in
 * Pillow's 16.16 fixed point (bit-identical to Image.transform(AFFINE, NEAREST) and to the flips), 2 perspective with
 * NEAREST sampling.  swapnet_amd/datasets/gpu_augment.py draws the parameters like torchvision's transforms do. */
int swn_op_affine_gather(swn_ctx* ctx, const float* src, float* dst, int b, int c, int h, int w, const double* maps,
                         int nmaps);
/* GANLoss(gan_mode)(prediction, target_is_real) (modules/loss.py:110-130) with the target scalar already drawn
 * (`label`; the smooth-label draw stays host-side like in the reference, loss.py:65-108): mean BCE-with-logits /
 * MSE / +-mean over ALL elements of pred (N,C,H,W).  loss_out = device float; dpred (optional, same shape) receives
 * grad_scale * d(loss)/d(pred). */
int swn_op_gan_loss(swn_ctx* ctx, int gan_mode, const float* pred, int n, int c, int h, int w, float label,
                    int target_is_real, float grad_scale, float* loss_out, float* dpred);
/* [InstanceNorm] -> act -> nn.Dropout(p) in TRAINING mode as ONE op (UNetDown / ResidualBlock, modules/layers.py:
 * 18-23,133-136): y, the keep/scale mask it used (0 or 1/(1-p); may be NULL) and, when dy and dx are given, the
 * input gradient computed with the same mask.  NCHW fp32, C % 4 == 0. */
int swn_op_norm_act_dropout(swn_ctx* ctx, const float* x, const float* dy, int n, int c, int h, int w, int norm, int act,
                            float p, uint64_t seed, float* y, float* mask, float* dx);
/* second-order step through y = act(InstanceNorm(x)) (gradient-penalty pass, csrc/gp.cpp): with gx = d<y,gy>/dx the first
 * backward, returns uy = d<gx,u>/d gy and ax = d<gx,u>/d x.  NCHW fp32, C % 4 == 0. */
int swn_op_norm_act_bwd2(swn_ctx* ctx, const float* x, const float* gy, const float* u, int n, int c, int h, int w, int act,
                         float* uy, float* ax);
/* torch.optim.AdamW single step on flat arrays (optimizers/__init__.py:52-59) */
int swn_op_adamw(swn_ctx* ctx, float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                 float eps, float wd, int step);

#ifdef __cplusplus
}
#endif
#endif /* SWAPNET_HIP_H */
