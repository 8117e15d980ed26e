/* mst_engine.h -- C ABI of the MI355X denoising engine (libmst_engine.so).
 *
 * The reference (hlcdyy/diffusion-based-motion-style-transfer) is 100 % Python and has no
 * FFI/operator interface of its own, so nothing here mirrors an existing
 * binding; each entry point instead REPLACES a group of reference Python functions, cited per
 * function below as file:line under the reference root.  The library is called only from the
 * boundary package (diffusion-based-motion-style-transfer_amd/_native.py via ctypes), never from
 * user scripts.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; mst_last_error() then returns a
 *     thread-local, NUL-terminated description.
 *   - all pointers named *_dev are device (HBM) pointers owned by the caller (PyTorch tensors);
 *     *_host are host pointers.  The engine owns only its weights copy and workspace.
 *   - all work is enqueued on the hipStream_t passed as `void* stream` (NULL = default stream)
 *     and is stream-ordered; one host thread per handle.
 *   - tensors use the reference's layouts: clips are float32 [B, F, 1, T] (F = njoints*nfeats,
 *     T contiguous), timesteps are int64 [B].
 */
#ifndef MST_ENGINE_H
#define MST_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mst_engine mst_engine;     /* one MDM-shaped denoiser (weights + workspace)     */
typedef struct mst_schedule mst_schedule; /* one (possibly respaced) diffusion process' tables */

typedef struct mst_config {
    int32_t feats;       /* F = njoints * nfeats, e.g. 263 (humanml), 181 (stylexia), 190 (bandai) */
    int32_t max_frames;  /* largest T the engine must accept (<= 223; tokens S = T + 1)            */
    int32_t max_rows;    /* largest number of clips through the transformer at once
                            (= 2 * batch under classifier-free guidance)                           */
    int32_t latent_dim;  /* must be 512  (utils/parser_util.py default; model_util.py:160-167)     */
    int32_t num_heads;   /* must be 4    (hard-coded in utils/model_util.py:160-167)               */
    int32_t ff_size;     /* must be 1024 (same)                                                    */
    int32_t num_layers;  /* 1..16 (default 8)                                                      */
    int32_t clip_dim;    /* width of the text embedding, 512                                       */
    int32_t pe_len;      /* rows of the positional table (5000, mdm_forstyledataset.py:388)        */
    int32_t device;      /* HIP device ordinal                                                     */
} mst_config;

/* -------------------------------------------------------------------------------------------
 * lifetime
 * ----------------------------------------------------------------------------------------- */
const char* mst_last_error(void);
int  mst_version(void);
const char* mst_source_hash(void);   /* hash of csrc/ + this header the library was compiled from (stale-build check) */
int  mst_engine_create(const mst_config* cfg, mst_engine** out);
void mst_engine_destroy(mst_engine* e);

/* Copy one parameter (float32, device memory) into the engine, converting dense matrices to the
 * engine's padded f16 operand layout.  `name` is the key of the tensor in an MDM state dict
 * (model/mdm_forstyledataset.py:183-270), i.e. the reference checkpoint layout:
 *   seqTransEncoder.layers.{i}.self_attn.in_proj_weight [1536,512] / in_proj_bias
 *   seqTransEncoder.layers.{i}.self_attn.out_proj.weight [512,512] / .bias
 *   seqTransEncoder.layers.{i}.linear1.weight [1024,512] / .bias, linear2.weight [512,1024] / .bias
 *   seqTransEncoder.layers.{i}.norm1.weight/.bias, norm2.weight/.bias
 *   input_process.poseEmbedding.weight [512,F] / .bias      (InputProcess  :425-449)
 *   output_process.poseFinal.weight   [F,512] / .bias       (OutputProcess :452-478)
 *   embed_timestep.time_embed.0.weight/.bias, .2.weight/.bias (TimestepEmbedder :408-422)
 *   embed_text.weight [512,clip_dim] / .bias                (:258)
 *   sequence_pos_encoder.pe [pe_len,512]                    (PositionalEncoding :387-404)
 * For StyleDiffusion the caller passes its own `seqTransEncoder.*` tensors and the frozen prior's
 * (`motion_enc.mdm_model.*`) projections, which is exactly what StyleDiffusion.forward (:602-625)
 * reads.  Replaces: nn.Module.load_state_dict / .to(device) for this path.
 * Stream contract: the f16 copies are written on `stream`.  The fused kernels' pre-packed copies (fragment streams of W_in and of
 * W_out | W1 | W2) are produced lazily on the stream of the FIRST mst_forward / mst_sample_loop after an upload (for a loop: the
 * engine's loop stream, which is ordered behind the caller's stream); a caller that uploads on one stream and samples on another
 * orders the two itself, exactly as for the plain copies. */
int mst_load_weight(mst_engine* e, const char* name, const float* src_dev,
                    const int64_t* shape, int32_t ndim, void* stream);
/* The 12 tensors of EVERY encoder layer in one call (one launch per 8 layers): srcs_host_array = num_layers x 12 device pointers, per
 * layer in the order in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, linear1.weight, linear1.bias, linear2.weight,
 * linear2.bias, norm1.weight, norm1.bias, norm2.weight, norm2.bias.  Same result as the 12 x num_layers mst_load_weight calls; what
 * a fine-tune iteration does after every optimizer step (train/training_loop.py:297-303 updates all 96).  Not in precise mode. */
int mst_load_layers(mst_engine* e, const float* const* srcs_host_array, void* stream);
/* 0 when every tensor of the list above has been loaded. */
int mst_weights_complete(const mst_engine* e);

/* -------------------------------------------------------------------------------------------
 * schedule: replaces GaussianDiffusion.__init__'s tables as consumed through
 * _extract_into_tensor (diffusion/gaussian_diffusion.py:183-219, :1605-1618) and
 * _WrappedModel's timestep_map (diffusion/respace.py:129-134).
 * `tables_host` is float32 [MST_NTAB][num_steps] in the order of mst_table below: the float64
 * numpy tables cast to float32, exactly what `.float()` at gaussian_diffusion.py:1615 yields.
 * ----------------------------------------------------------------------------------------- */
enum mst_table {
    MST_TAB_SQRT_AC = 0,        /* sqrt_alphas_cumprod                */
    MST_TAB_SQRT_1M_AC = 1,     /* sqrt_one_minus_alphas_cumprod      */
    MST_TAB_COEF1 = 2,          /* posterior_mean_coef1               */
    MST_TAB_COEF2 = 3,          /* posterior_mean_coef2               */
    MST_TAB_LOGVAR = 4,         /* log-variance of the configured var type
                                   (FIXED_SMALL: posterior_log_variance_clipped) */
    MST_TAB_SQRT_RECIP_AC = 5,  /* sqrt_recip_alphas_cumprod          */
    MST_TAB_SQRT_RECIPM1_AC = 6,/* sqrt_recipm1_alphas_cumprod        */
    MST_TAB_AC = 7,             /* alphas_cumprod                     */
    MST_TAB_AC_PREV = 8,        /* alphas_cumprod_prev                */
    MST_NTAB = 9
};
int  mst_schedule_create(int32_t num_steps, const float* tables_host,
                         const int32_t* timestep_map_host, int32_t device, mst_schedule** out);
void mst_schedule_destroy(mst_schedule* s);

/* -------------------------------------------------------------------------------------------
 * conditioning: replaces embed_text(mask_cond(encode_text(...))) of StyleDiffusion.forward
 * (:611-615, :592-600) / MDM.forward (:324-327).  CLIP itself stays outside; `text_emb_dev` is its
 * float32 [batch, clip_dim] output.  keep_dev (float32 [batch], may be NULL = all ones) is the
 * cond mask: 0 drops the text embedding of that clip (y['uncond'] / the Bernoulli mask).
 * With cfg != 0 the engine prepares a doubled batch: rows [0,batch) conditional, rows
 * [batch, 2*batch) unconditional (model/cfg_sampler.py:36-43).  The projection is constant over a
 * sampling loop, so it is computed here once instead of once per step.
 * ----------------------------------------------------------------------------------------- */
int mst_set_text(mst_engine* e, const float* text_emb_dev, const float* keep_dev,
                 int32_t batch, int32_t cfg, void* stream);
/* The training-mode mask_cond (model/mdm_forstyledataset.py:288-296 / :592-600: `cond * (1. - bernoulli(ones(bs) * p))`) with the
 * mask handed over as drawn: drop_dev float32 [batch], 1 = the clip's text embedding is dropped.  Same projection, one
 * elementwise launch less on the caller's side than masking first and calling mst_set_text. */
int mst_set_text_dropped(mst_engine* e, const float* text_emb_dev, const float* drop_dev, int32_t batch, void* stream);

/* -------------------------------------------------------------------------------------------
 * one model evaluation: replaces StyleDiffusion.forward / MDM.forward (:602-625, :315-364) and,
 * with cfg != 0, ClassifierFreeSampleModel.forward (model/cfg_sampler.py:36-43).
 *   x_dev      float32 [batch, F, 1, frames]
 *   t_dev      int64 [batch] ORIGINAL-process timesteps (after timestep_map)
 *   scale_dev  float32 [batch] guidance scale (cfg only)
 *   out_dev    float32 [batch, F, 1, frames]
 * mst_set_text must have been called for this batch/cfg.
 * ----------------------------------------------------------------------------------------- */
int mst_forward(mst_engine* e, const float* x_dev, const int64_t* t_dev, const float* scale_dev,
                int32_t batch, int32_t frames, int32_t cfg, float* out_dev, void* stream);

/* -------------------------------------------------------------------------------------------
 * sampling loop / single step: replaces p_sample_loop(_progressive), ddim_sample_loop
 * (_progressive) (diffusion/gaussian_diffusion.py:644-794, :948-1082) with p_mean_variance
 * (:311-424), p_sample (:532-585 / inpainting_gaussian_diffusion.py:25-64) and ddim_sample
 * (inpainting_gaussian_diffusion.py:125-177) fused into the output-projection kernel.
 * Runs diffusion indices t_start, t_start-1, ..., t_end (t_start == t_end: one step).
 * ----------------------------------------------------------------------------------------- */
enum { MST_SAMPLER_DDPM = 0, MST_SAMPLER_DDIM = 1 };
enum { MST_NOISE_BUFFER = 0, MST_NOISE_PHILOX = 1 };

typedef struct mst_loop_args {
    int32_t batch;                  /* clips B                                                   */
    int32_t frames;                 /* T                                                         */
    int32_t cfg;                    /* classifier-free guidance: cond/uncond as one 2B batch     */
    int32_t sampler;                /* MST_SAMPLER_*                                             */
    int32_t mask_noise;             /* 1: InpaintingGaussianDiffusion (noise *= 1 - mask)        */
    int32_t clip_denoised;          /* clamp x0-hat to [-1, 1] (callers pass 0)                  */
    int32_t noise_mode;             /* MST_NOISE_*                                               */
    int32_t t_start, t_end;         /* inclusive, t_start >= t_end >= 0                          */
    float   eta;                    /* DDIM eta                                                  */
    uint64_t seed;                  /* Philox key (MST_NOISE_PHILOX)                             */
    const float* scale_dev;         /* [B] guidance scale (cfg)                                  */
    const float* inpainting_mask_dev;   /* [B,F,1,T] float32 0/1, or NULL                        */
    const float* inpainted_motion_dev;  /* [B,F,1,T], or NULL (blend needs both)                 */
    const float* noise_dev;         /* MST_NOISE_BUFFER: [nsteps][B,F,1,T], step j at j*B*F*T    */
    float* x_dev;                   /* [B,F,1,T]: x_{t_start} on entry, final sample on exit     */
    float* xstart_dump_dev;         /* optional [nsteps][B,F,1,T]: x0-hat of every step          */
} mst_loop_args;

int mst_sample_loop(mst_engine* e, const mst_schedule* s, const mst_loop_args* a, void* stream);

/* Number of independent clip slices (1..3) mst_sample_loop runs on separate streams for this
 * batch of `frames`-frame clips (frames <= 0: the engine's max_frames; the policy depends on the
 * token-row count, so pass the loop's own frame count when it is below the cap): clips never interact (no cross-sample op in
 * mdm_forstyledataset.py:602-625), so slices overlap each other's launch gaps, prologues and
 * tails.  Chosen per call from the tile count (one slice when every tile of the batch is
 * resident at once, up to three beyond that and on the small-tile path); MST_STREAMS=1..3 in
 * the environment at engine creation fixes it.  Per-launch work = batch / slices clips. */
int mst_loop_slices(const mst_engine* e, int32_t batch, int32_t cfg, int32_t frames);

/* -------------------------------------------------------------------------------------------
 * stand-alone elementwise kernels for callers that bring their own model callable
 * (any nn.Module passed to p_sample / ddim_sample / q_sample):
 *   mst_q_sample        diffusion/gaussian_diffusion.py:267-285,
 *                       diffusion/inpainting_gaussian_diffusion.py:6-23
 *   mst_step_epilogue   gaussian_diffusion.py:341-349 (blend), :387-412 (mean/variance),
 *                       :569-585 / inpainting_gaussian_diffusion.py:51-63 (p_sample),
 *                       inpainting_gaussian_diffusion.py:157-177 (ddim_sample)
 * t_dev is int64 [batch] of indices into the schedule.  sample_out_dev / xstart_out_dev may be NULL.
 * ----------------------------------------------------------------------------------------- */
int mst_q_sample(const mst_schedule* s, const float* x_start_dev, const float* noise_dev,
                 const float* mask_dev, const int64_t* t_dev, int32_t batch, int64_t per_clip,
                 float* out_dev, void* stream);

int mst_step_epilogue(const mst_schedule* s, const float* model_out_dev, const float* x_dev,
                      const float* noise_dev, const float* mask_dev, const float* motion_dev,
                      const int64_t* t_dev, int32_t batch, int64_t per_clip, int32_t sampler,
                      float eta, int32_t mask_noise, int32_t clip_denoised,
                      float* sample_out_dev, float* xstart_out_dev, void* stream);

/* The same step for a model that predicts something else than x_start (enum ModelMeanType, gaussian_diffusion.py:69-76; branch
 * :398-412): mean_type 0 = x_start, 1 = epsilon (x0-hat = sqrt_recip_alphas_cumprod x - sqrt_recipm1_alphas_cumprod out, :426-431),
 * 2 = previous x (x0-hat = out / coef1 - coef2 / coef1 x, :433-441; the posterior mean is then the model output itself, :399-403).
 * As in the reference the inpainting blend (:341-349) acts on the raw model output, in front of the conversion. */
int mst_step_epilogue_mt(const mst_schedule* s, const float* model_out_dev, const float* x_dev,
                         const float* noise_dev, const float* mask_dev, const float* motion_dev,
                         const int64_t* t_dev, int32_t batch, int64_t per_clip, int32_t sampler, int32_t mean_type,
                         float eta, int32_t mask_noise, int32_t clip_denoised,
                         float* sample_out_dev, float* xstart_out_dev, void* stream);

/* Standard-normal fill with the engine's Philox stream (the generator MST_NOISE_PHILOX uses
 * inside the fused step), so a caller can reproduce in-loop noise: element (clip, f, t) of step
 * `step`.  Replaces th.randn / th.randn_like draws (gaussian_diffusion.py:754, :569). */
/* Backward of mst_step_epilogue for the `*_with_grad` samplers (diffusion/inpainting_gaussian_diffusion.py:66-123,
 * :179-239; the fine-tune objective keeps every x0-hat in the autograd graph, gaussian_diffusion.py:1364-1378):
 *     d_model_out = (g_pred + g_sample * d sample / d pred) * (1 - mask)
 * g_sample / g_pred: upstream gradients of the two outputs, either may be NULL (= zero).  has_blend: the forward blended
 * with (mask, motion).  pred_clipped_dev: NULL, or -- when the forward ran with clip_denoised (the reference signature's default,
 * gaussian_diffusion.py:389-395) -- its x0-hat output: the clamp's gradient mask (zero where the prediction saturated at +-1). */
int mst_step_backward(const mst_schedule* s, const float* g_sample_dev, const float* g_pred_dev, const float* mask_dev,
                      int32_t has_blend, const int64_t* t_dev, int32_t batch, int64_t per_clip, int32_t sampler, float eta,
                      const float* pred_clipped_dev, float* d_model_out_dev, void* stream);

/* K13 of SURVEY section 2.1 -- the reductions of the fine-tune objective, one launch each way:
 * mst_masked_l2: `masked_l2` (gaussian_diffusion.py:223-235) of a, b [n][feats][1][frames] with a frame mask
 *   [n][1][1][frames]; a_stride / mask_stride = elements between consecutive samples (0 = broadcast, the
 *   `.expand(num_step, ...)` of :1380).  g == NULL: out[n] = loss; g != NULL ([n] upstream gradient): out = dL/db
 *   ([n][feats][1][frames]; dL/da is its negative).
 * mst_text_cosine: `(1 - cosine_similarity(f / |f|, m / |m|, eps = 1e-6)).mean()` (:1384-1388) of two [batch][dim]
 *   matrices.  g == NULL: out[0] = loss; g != NULL ([1]): out = dL/dm ([batch][dim]). */
int mst_masked_l2(const float* a_dev, int64_t a_stride, const float* b_dev, const float* mask_dev, int64_t mask_stride,
                  int32_t n, int32_t feats, int32_t frames, const float* g_dev, float* out_dev, void* stream);
int mst_text_cosine(const float* f_dev, const float* m_dev, int32_t batch, int32_t dim, const float* g_dev, float* out_dev,
                    void* stream);

int mst_philox_normal(float* out_dev, int32_t batch, int32_t feats, int32_t frames, uint64_t seed,
                      uint32_t step, void* stream);

/* -------------------------------------------------------------------------------------------
 * Training path of the TRAINABLE encoder stack: StyleDiffusion.seqTransEncoder, 8 x
 * nn.TransformerEncoderLayer(d_model=512, nhead=4, dim_feedforward=1024, dropout=0.1, gelu),
 * model/mdm_forstyledataset.py:539-546, called at :622 inside the graph that
 * few_shot_style_finetune_losses (diffusion/gaussian_diffusion.py:1317-1399) back-propagates
 * through.  Replaces the torch autograd graph of that call:
 *   mst_train_forward   forward in model.train() semantics: dropout p at the four sites of each
 *                       layer (attention probabilities, out-proj output, FFN hidden, FFN output),
 *                       masks drawn from a counter-based generator keyed by `seed`; writes the
 *                       activation tape into caller-owned memory of mst_train_tape_bytes() bytes.
 *   mst_train_backward  given dL/d(h_out): dL/d(h_in) and the 96 parameter gradients, ACCUMULATED
 *                       (+=) into the caller's float32 buffers.  `grads` is a HOST array of
 *                       num_layers*12 device pointers in nn.TransformerEncoderLayer parameter order:
 *                       self_attn.in_proj_weight, .in_proj_bias, self_attn.out_proj.weight, .bias,
 *                       linear1.weight, .bias, linear2.weight, .bias, norm1.weight, .bias,
 *                       norm2.weight, .bias.  grads == NULL: no parameter gradients (frozen stack, input
 *                       gradient only).  rows / S / p_drop / seed must repeat the forward's.
 * h_in, h_out, d_out, d_in: float32 [rows][S][512] (clip-major; the reference's [S, B, 512] permuted).
 * key_keep: NULL, or uint8 [rows][S] with 0 marking padding keys -- the inverse of the `src_key_padding_mask`
 * the frozen MotionEncoder passes to its own 8-layer stack (model/mdm_forstyledataset.py:90-124); every clip
 * must keep at least one key.
 * The engine's weights are the ones last uploaded with mst_load_weight.
 * mst_dropout_mask: the keep-multipliers (0 or 1/(1-p)) of the first n elements of site
 * (layer, site 0..3) -- lets a test rebuild the masked forward exactly in PyTorch.
 * ----------------------------------------------------------------------------------------- */
int64_t mst_train_tape_bytes(const mst_engine* e, int32_t rows, int32_t S);
int mst_train_forward(mst_engine* e, const float* h_in_dev, int32_t rows, int32_t S, float p_drop,
                      uint64_t seed, const uint8_t* key_keep_dev, void* tape_dev, float* h_out_dev,
                      void* stream);
int mst_train_backward(mst_engine* e, const void* tape_dev, const float* d_out_dev, int32_t rows,
                       int32_t S, float p_drop, uint64_t seed, const uint8_t* key_keep_dev,
                       float* d_in_dev, float* const* grads_host_array, void* stream);
/* The whole denoiser as one training node: StyleDiffusion.forward / MDM.forward in train mode
 * (model/mdm_forstyledataset.py:602-625, :315-364): conditioning token (timestep MLP + text projection, text set with
 * mst_set_text as for mst_forward), pose embedding + positional rows, PositionalEncoding's dropout p_pe (:404) on the
 * assembled sequence, the trainable stack (p_drop), output projection.  Backward returns dL/dx and accumulates the 96
 * stack gradients (the projections, the timestep MLP and the text projection are frozen in every shipped script).
 * x, out, d_out, d_x: float32 [batch][feats][1][frames]; t_idx: int64 [batch] original-process timesteps.
 * clip0 / tape_clips (tape_clips <= 0: the tape is this call's own): the call writes clips [clip0, clip0 + batch) of a tape laid out
 * for tape_clips clips, and draws the dropout masks those clips have in a pass over the WHOLE tape with the same seed.  For model calls
 * whose inputs are cut from each other's graphs -- the chained x0-hat steps of the fine-tune objective (gaussian_diffusion.py:1364-1378:
 * `x.detach()` between steps) -- so that ONE mst_train_model_backward over tape_clips clips differentiates all of them (their forward
 * passes are sequential, their backward passes independent). */
int mst_train_model_forward(mst_engine* e, const float* x_dev, const int64_t* t_idx_dev, int32_t batch,
                            int32_t frames, float p_drop, float p_pe, uint64_t seed, void* tape_dev,
                            float* out_dev, int32_t clip0, int32_t tape_clips, void* stream);
int mst_train_model_backward(mst_engine* e, const void* tape_dev, const float* d_out_dev, int32_t batch,
                             int32_t frames, float p_drop, float p_pe, uint64_t seed, float* d_x_dev,
                             float* const* grads_host_array, void* stream);
/* MotionEncoder.forward (model/mdm_forstyledataset.py:90-124), the frozen "semantic discriminator" of the fine-tune
 * objective (gaussian_diffusion.py:1340-1343), as one native call each way:
 *     frames = mdm_model.input_process(x);  seq = pos_encoder(cat(muQuery, sigmaQuery, frames));
 *     mu = seqTransEncoder(seq, src_key_padding_mask=~keep)[0]
 * x_dev [batch][feats][1][frames]; mu_query / sigma_query [512]; key_keep_dev [batch][frames + 2] bytes (1 = real key,
 * the two query tokens first); mu_out_dev [batch][512].  Dropout (p_drop in the layers, p_pe behind the positional rows) is
 * counter-based from `seed` as in mst_train_forward.  The backward call returns dL/dx only: every parameter on this path
 * is frozen (train/finetune_style_diffusion.py:256, load_motion_enc :579-588).  The engine must hold the encoder's own
 * layers and the prior's pose embedding / positional table, with max_frames >= frames + 1. */
int mst_motion_encoder_forward(mst_engine* e, const float* x_dev, const float* mu_query_dev, const float* sigma_query_dev,
                               const uint8_t* key_keep_dev, int32_t batch, int32_t frames, float p_drop, float p_pe,
                               uint64_t seed, void* tape_dev, float* mu_out_dev, void* stream);
int mst_motion_encoder_backward(mst_engine* e, const void* tape_dev, const float* d_mu_dev, const uint8_t* key_keep_dev,
                                int32_t batch, int32_t frames, float p_drop, float p_pe, uint64_t seed, float* d_x_dev,
                                void* stream);

/* Data-parallel fine-tuning (BASELINE.json configs[3]; the reference is single-device, train/training_loop.py:73).
 * Make `stream` wait until every kernel that writes layer `layer`'s 12 gradient tensors in the MOST RECENT
 * mst_train_backward / mst_train_model_backward call (grads != NULL) has finished.  The backward calls only ENQUEUE
 * work; a reducer calls this right after the backward call returns, layer num_layers-1 first, and launches that
 * layer's gradient all-reduce on `stream`: it then runs on the GPU while the layers below are still being
 * differentiated. */
int mst_train_wait_layer_grads(mst_engine* e, int32_t layer, void* stream);
int mst_dropout_mask(uint64_t seed, int32_t layer, int32_t site, float p, uint64_t n, float* out_dev,
                     void* stream);

/* -------------------------------------------------------------------------------------------
 * Optimizer step of the fine-tune loop: `self.mp_trainer.optimize(self.opt)`,
 * train/training_loop.py:196-200 -> diffusion/fp16_util.py:208-223 (`_compute_norms`: one
 * `.item()` sync per tensor for grad and param norms) + torch.optim.AdamW.step (training_loop.py:96).
 * ONE launch over all tensors: AdamW update in place (torch's arithmetic: decoupled weight decay,
 * step_size = lr/(1-b1^t), denom = sqrt(v)/sqrt(1-b2^t) + eps) and norms_dev[0] += sum g^2,
 * norms_dev[1] += sum p^2 of the parameters BEFORE the update (what the reference logs).
 * All pointer arrays are HOST arrays of device pointers; workspace_dev holds the device-side tables
 * (mst_adamw_workspace_bytes).  `step` is the 1-based step count of these tensors.
 * `upload_tables` != 0: (re)write the device-side tables from the host arrays before the launch (one
 * stream synchronisation + two small copies).  The OWNER of the workspace decides: it must pass 1 on the
 * first call with a workspace, whenever any pointer / size changed since the tables were last written
 * into THIS workspace allocation, and whenever the workspace memory may have been reused in between;
 * 0 re-uses the tables already there (the steady state: torch's caching allocator returns the same
 * blocks every iteration).  The library keeps no state of its own about workspaces.
 * ----------------------------------------------------------------------------------------- */
int64_t mst_adamw_workspace_bytes(int32_t n_tensors, const int64_t* numel_host);
int mst_adamw_step(int32_t n_tensors, float* const* params, const float* const* grads,
                   float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel_host, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int32_t step,
                   float* norms_dev, void* workspace_dev, int64_t workspace_bytes, int32_t upload_tables,
                   void* stream);

/* -------------------------------------------------------------------------------------------
 * Post-sampling tensor ops, one launch (sample/demo_style_transfer.py:265-267,
 * train/finetune_style_diffusion.py:331-332):
 *     sample = dataset.inv_transform(sample.cpu().permute(0, 2, 3, 1)).float()   dataset.py:478-479
 *     joints = recover_from_ric(sample, n_joints)                                 motion_process.py:444-461
 * sample_dev: [batch][feats][1][frames] float32 normalised hml_vec (the samplers' output layout),
 * mean_dev / std_dev: [feats]; out_dev: [batch][1][frames][joints][3].  The clip never leaves the GPU.
 * ----------------------------------------------------------------------------------------- */
int mst_recover_from_ric(const float* sample_dev, const float* mean_dev, const float* std_dev, int32_t batch,
                         int32_t feats, int32_t frames, int32_t joints, float* out_dev, void* stream);

/* Per-kernel device timing of the most recent mst_sample_loop / mst_forward when profiling is
 * enabled: HIP events recorded around every launch on the caller's stream.  names/ms are arrays
 * of `cap` entries filled with per-kernel-family totals; returns the number of families. */
int mst_profile_enable(mst_engine* e, int32_t on);
float mst_profile_event_overhead_us(const mst_engine* e);   /* what an empty HIP-event pair reports: the fixed part of every event-timed launch */
int mst_profile_read(mst_engine* e, const char** names, float* total_ms, int32_t* launches,
                     int32_t cap);

/* Precise mode (no reference counterpart: the reference computes in fp32, model/mdm_forstyledataset.py:539-546).  on != 0: every
 * layer GEMM of the sampling path (mst_forward, mst_sample_loop) multiplies its activation operand as hi + lo -- f16(x) and
 * f16(x - hi), ~22 significant bits -- instead of f16(x): for checkpoints whose statistics (LayerNorm-gain outlier channels, large
 * FFN / attention weights) put plain f16 operands above the 1e-3 relative-L2 bar.  Takes the small-tile kernels at any batch size
 * (about half the throughput of the default path at 64 clips); default off, also MST_PRECISE=1.
 * The weights' lo halves are written by mst_load_weight only while precise mode is on (a fine-tune iteration re-uploads 96 tensors and
 * never reads them).  Returns 0, or 2 when it was switched on over weights uploaded without them: upload those again (mst_forward /
 * mst_sample_loop fail until then). */
int mst_set_precise(mst_engine* e, int32_t on);

/* Resident-group trunk (no reference counterpart; csrc/mst_trunk.h).  on != 0: a sampling step's encoder stack (the 8 x [fused QKV +
 * attention, fused layer tail] of model/mdm_forstyledataset.py:539-546, 602-625) runs as ONE launch in which the four workgroups of a
 * clip stay resident and hand their phase outputs to each other through a per-clip arrival counter; shapes it does not cover (token
 * counts other than 193 .. 208, precise mode, debug stops, instrumented steps) keep the two launches per layer.  Results are
 * bit-identical either way.  Also MST_TRUNK=1 / 0.  mst_trunk_check: after the caller has synchronised, 0 when every hand-off of every
 * such launch arrived (a bounded wait that gives up sets a host-visible word instead of hanging the device). */
int mst_set_trunk_groups(mst_engine* e, int32_t on);
int mst_trunk_check(mst_engine* e);

/* Debug / test hooks (no reference counterpart): stop the encoder stack after (layer, stage) --
 * stage 0 = token stream assembled, 1 = QKV, 2 = attention, 3 = out-proj + LayerNorm1, 4 = FFN1,
 * 5 = FFN2 + LayerNorm2; layer = stage = -1 runs everything -- and copy a workspace buffer
 * ("hs" f32 stream, "hx"/"qkv"/"att"/"hid" f16, "temb"/"textproj" f32) into caller memory. */
int mst_debug_stop_after(mst_engine* e, int32_t layer, int32_t stage);
int mst_debug_copy(mst_engine* e, const char* which, void* dst_dev, uint64_t nbytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MST_ENGINE_H */
