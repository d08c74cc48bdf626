/* etainv.h -- C ABI of the MI355X-native Eta-Inversion engine (libetainv_hip.so).
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json `north_star`: the
 * forward (DDIM inversion) / backward (eta-sampling) loop of furiosa-ai/eta-inversion's `etainv`
 * with the simple / prompt-to-prompt / MasaCtrl editors on an SD1.x UNet.  Every entry point
 * cites the reference interface it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  All tensor pointers are DEVICE pointers
 *     unless the parameter is documented as host.  Tensors are contiguous, NCHW where spatial.
 *   - `stream` is a hipStream_t passed as void* (e.g. torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it, no hidden synchronisation, no per-call allocation.
 *   - ownership: the caller owns every buffer it passes; the library never frees or retains it
 *     past the call.  The engine handle owns its weights, workspace and attention-map store.
 *   - errors: every function returns 0 on success, non-zero otherwise; a thread-local message
 *     is available from etainv_last_error().  No C++ exception crosses the ABI.
 *   - threading: a handle is not thread-safe (the reference is single-threaded, one stream).
 *   - batch layout for B image pairs ("n_img"):  latents of the backward pass are 2*B rows
 *     [src_0..src_{B-1}, tgt_0..tgt_{B-1}]; UNet rows / contexts are 4*B rows
 *     [u_src x B, u_tgt x B, c_src x B, c_tgt x B].  With B = 1 this is exactly the reference's
 *     [u_s, u_t, c_s, c_t] (modules/inversion/diffusion_inversion.py:462-479).
 */
#ifndef ETAINV_H
#define ETAINV_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ETAINV_ABI_VERSION 1
#define ETAINV_MAX_WORDS 77

enum etainv_dtype { ETAINV_F32 = 0, ETAINV_F16 = 1, ETAINV_BF16 = 2 };

int etainv_abi_version(void);
const char* etainv_last_error(void);

/* ---------------------------------------------------------------- elementwise step kernels
 * (usable without an engine handle; io_dtype is the element type of every tensor argument) */

/* eps = eps_u + g * (eps_c - eps_u).  Replaces the CFG combine of EtaInversion.predict_noise
 * (modules/inversion/eta_inversion.py:328). n = element count. */
int etainv_cfg_combine(const void* eps_u, const void* eps_c, float g, void* out, int64_t n,
                       int io_dtype, void* stream);

/* x' = sqrt(a_to) * (x - sqrt(1-a_from) eps)/sqrt(a_from) + sqrt(1-a_to) eps.
 * Replaces DDIMInverseScheduler.ddim_step (modules/inverse_schedulers/scheduling_ddim_inverse.py:71-100);
 * a_from/a_to are the alphas_cumprod the host looked up (clamp/negative rule :85-92). */
int etainv_ddim_step(const void* x, const void* eps, float a_from, float a_to, void* out, int64_t n,
                     int io_dtype, void* stream);

/* x' = DDIMScheduler.step(eps, t, x, eta, variance_noise) for `rows` latents ([3P] diffusers scheduler as called by
 * DiffusionInversion.step_backward, modules/inversion/diffusion_inversion.py:301-312, and with a per-pixel eta by
 * modules/inversion/eta_inversion.py:245).  eta_px = eta * eta_mask[row % n_mask][pixel] (eta_mask may be NULL);
 * noise [c*hw] is shared by all rows (may be NULL = no noise term). */
int etainv_ddim_eta_step(const void* x, const void* eps, float eta, const void* eta_mask, int n_mask, const void* noise,
                         float a_t, float a_p, float var, int rows, int c, int hw, void* out, int io_dtype, void* stream);

/* Fused backward step of EtaInversion.predict_step_backward (modules/inversion/eta_inversion.py:207-273)
 * for n_img independent (src,tgt) pairs, everything after the UNet call:
 *   CFG combine (:328) -> best-of-n variance noise on the source row (:330-375, argmin on device, NaN counts
 *   as minimal like torch.argmin) -> per-pixel eta = 1[mask_map > thres] * eta (:159-205,:236-243) ->
 *   DDIM-eta update of both rows ([3P] DDIMScheduler.step as called at :245) -> source row replay
 *   x_src += (x_prev_src - x_src) (:247-249; exact assignment when use_mask == 0, :260-261).
 * x        [2*n_img][chw]   current latents (rows src.., tgt..)
 * eps_all  [4*n_img][chw]   UNet output rows [u_s.., u_t.., c_s.., c_t..]
 * x_prev_src [n_img][chw]   stored inversion latent latents[-(k+2)] (:291)
 * noise    [n_cand][chw]    candidates drawn by sample_variance_noise (:145-156), shared by all images
 * mask_map [n_img][hw]      forward-pass mean attention map of the source edit word (may be NULL if !use_mask);
 *                            use_mask == 2: the map IS the per-pixel eta multiplier (non-default mask modes: no threshold,
 *                            `pow`, ground-truth mask -- eta_inversion.py:164-201), mask_thres ignored
 * a_t, a_p, var             alphas_cumprod[t], alphas_cumprod[t-Delta] (or final), _get_variance(t, t-Delta)
 * out_x    [2*n_img][chw]   new latents;  out_eps [2*n_img][chw] guided noise (may be NULL)
 * best_idx [n_img] int32, losses [n_img][n_cand] float (device, may be NULL); scratch: >= n_img*16*64 floats */
int etainv_eta_backward_step(const void* x, const void* eps_all, float g, const void* x_prev_src,
                             const void* noise, int n_cand, float eta, const void* mask_map, float mask_thres,
                             int use_mask, float a_t, float a_p, float var, int n_img, int c, int hw,
                             void* out_x, void* out_eps, int32_t* best_idx, float* losses, float* scratch,
                             int io_dtype, void* stream);
/* same + the reference's `target_dirinv` option (eta_inversion.py:251-256): x_tgt += target_dirinv * dirinv_map * (x_prev_src - x_src_new),
 * dirinv_map [n_img][hw] = 1 - mask_dirinv prepared by the caller (NULL = 1 everywhere); needs use_mask != 0 */
int etainv_eta_backward_step_ex(const void* x, const void* eps_all, float g, const void* x_prev_src, const void* noise, int n_cand, float eta,
                                const void* mask_map, float mask_thres, int use_mask, float a_t, float a_p, float var, int n_img, int c, int hw,
                                void* out_x, void* out_eps, int32_t* best_idx, float* losses, float* scratch, int io_dtype,
                                float target_dirinv, const void* dirinv_map, void* stream);

/* out = a x + b y + c z over n elements (z may be NULL): the latent update of the multistep DPM-Solver++ scheduler pair
 * (reference modules/inverse_schedulers/scheduling_dpmsolver_multistep_inverse.py:83-159 and the [3P] diffusers
 * DPMSolverMultistepScheduler.step behind DiffusionInversion.step_backward, modules/inversion/diffusion_inversion.py:279-312) */
int etainv_lincomb3(const void* x, float a, const void* y, float b, const void* z, float c, void* out, int64_t n, int io_dtype, void* stream);

/* ---------------------------------------------------------------- engine */
typedef struct etainv_engine etainv_engine_t;

typedef struct etainv_engine_config {
  int compute_dtype;      /* ETAINV_F16 / ETAINV_BF16: MFMA operand type (fp32 accumulate); ETAINV_F32: fp32 operands on
                           * v_mfma_f32_32x32x2_f32, fp32 activations -- the reference's default precision (edit_image.py:147),
                           * 1/16 of the 16-bit matrix rate: the parity mode                                              */
  int max_unet_batch;     /* largest number of UNet rows per call (4 * n_img for the backward pass)      */
  int latent_size;        /* L: 64 for 512x512, 96 for 768x768; multiple of 8                           */
  int max_img;            /* largest n_img (attention-map store is sized for it)                        */
  int reserved[4];
} etainv_engine_config;

/* Memory: weights + one activation workspace are allocated here and freed by etainv_engine_destroy; calls never allocate,
 * except two process-wide one-time buffers of the GEMM launcher (a 256-byte zero page for the 3x3 halo and a 64 MiB split-K
 * workspace, created by the first launch that needs them).  One engine per device and process; not thread-safe. */
int etainv_engine_create(const etainv_engine_config* cfg, etainv_engine_t** out);
int etainv_engine_destroy(etainv_engine_t* e);

/* Weights: the engine enumerates the SD1.x UNet parameters under their diffusers state-dict names
 * (what `model.unet` holds in the reference, modules/models/__init__.py:135).  The caller uploads each as a
 * contiguous fp32 DEVICE buffer in diffusers layout; the engine converts / re-lays-out on `stream`. */
int etainv_engine_num_weights(etainv_engine_t* e);
int etainv_engine_weight_info(etainv_engine_t* e, int i, char* name, int name_cap, int64_t shape[4], int* ndim);
int etainv_engine_set_weight(etainv_engine_t* e, const char* name, const float* data, int64_t numel, void* stream);
int etainv_engine_weights_ready(etainv_engine_t* e); /* 1 when every parameter has been set */

/* Declarative attention control for one UNet call: replaces the per-layer Python callbacks installed by
 * register_attention_control (modules/utils/ptp_utils.py:196-302) and register_attention_editor_diffusers
 * (modules/utils/masactrl_utils.py:74-153).  Device pointers; per-image tables have n_img rows. */
enum etainv_attn_mode { ETAINV_ATTN_PLAIN = 0, ETAINV_ATTN_STORE = 1, ETAINV_ATTN_PTP = 2, ETAINV_ATTN_MASA = 3 };

typedef struct etainv_attn_ctrl {
  int mode;
  int n_img;
  /* STORE / PTP: accumulate the cond-half cross-attention probabilities of the layers the store keeps -- the five (L/4)^2-token layers by default,
   * the (L/2)^2 or (L/8)^2 ones after etainv_maps_configure -- (AttentionStore, modules/utils/ptp.py:143-183) into the engine's map store. */
  int store_maps;
  /* PTP cross edit (AttentionControlEdit.forward, modules/utils/ptp.py:205-218; Refine :245-258; Reweight
   * :261-274; Replace :234-242).  cross_alpha is the row of cross_replace_alpha for the current step. */
  const int32_t* mapper;      /* [n_img][77] or NULL */
  const float* alphas;        /* [n_img][77] or NULL */
  const float* replace_mat;   /* [n_img][77][77] or NULL (AttentionReplace) */
  const float* equalizer;     /* [n_img][77] or NULL */
  const float* cross_alpha;   /* [n_img][77] */
  /* PTP self edit: target probabilities := source probabilities when active and N <= self_max_tokens */
  int self_replace_active;
  int self_max_tokens;        /* 32^2 at L = 64 (modules/utils/ptp.py:226) */
  /* MASA: mutual self-attention active for transformer blocks >= masa_first_block (cur_att_layer // 2) */
  int masa_active;
  int masa_first_block;
  /* PTP only: the call carries rows [first_row, 4 n_img) of the [u_s, u_t, c_s, c_t] x n_img layout -- 0 (all rows) or n_img (no uncond source rows:
   * n_rows == 3 n_img, rows [u_t, c_s, c_t]).  A backward step whose eta(t) is 0 does not need eps(uncond source): the source row is replayed from the
   * inversion trajectory (modules/inversion/eta_inversion.py:247-249) and its guided noise only feeds the best-of-n choice of a noise that is then
   * multiplied by eta = 0 (:232, :330-375); the cond source row stays (its attention probabilities are what prompt-to-prompt injects). */
  int first_row;
  /* PTP with first_row = n_img only.  != 0: the 3 n_img rows are ordered [u_t, c_t, c_s] and the cond source rows LEAVE the network after transformer block
   * `src_exit_block` (execution order 0..15): from there on only the first 2 n_img rows are computed and the last n_img rows of `out` are not written.
   * For backward steps with eta == 0 in which nothing is injected from the source any more (cross_replace_alpha row all zero, self-replace over): the
   * cond source row then only feeds the AttentionStore of the five (L/4)^2 cross layers (LocalBlend, modules/utils/ptp.py:37-39), the last of which is
   * block 9 -- its noise prediction is unused (the source latent is replayed).  Needs mapper == replace_mat == NULL; while self_replace_active the exit
   * must lie behind the last (L/2)^2-token self-attention (block 12); with store_maps it must not lie in front of the last layer the store keeps
   * (etainv_maps_configure: res_div 2 -> block 12, 4 -> block 9, 8 -> block 6) -- an earlier exit is an error, not a silent loss of maps. */
  int src_exit_block;
  int reserved[2];
} etainv_attn_ctrl;

/* eps = UNet(latent, t, ctx).  Replaces `self.unet(latent_input, t, encoder_hidden_states=context)["sample"]`
 * (modules/inversion/eta_inversion.py:321).
 * latent  [n_lat][4][L][L]  io_dtype, 1 <= n_lat <= n_rows; UNet row r reads latent row r % n_lat (torch.cat([latent]*2), :320)
 * t_host  [n_rows] int64 HOST timesteps (one per UNet row)
 * ctx     [n_rows][77][768] io_dtype
 * out     [n_rows][4][L][L] io_dtype */
int etainv_unet_forward(etainv_engine_t* e, const void* latent, int n_lat, const int64_t* t_host, const void* ctx,
                        int n_rows, const etainv_attn_ctrl* ctrl, void* out, int io_dtype, void* stream);

/* Attention-map store (AttentionStore.attention_store restricted to what the default path reads:
 * the five (L/4)^2 cross layers; modules/utils/ptp.py:37-39, 288-303). */
int etainv_maps_reset(etainv_engine_t* e, void* stream);

/* Forward-pass word maps (ControllerAttentionStorePerStep.end_step, modules/inversion/eta_inversion.py:44-49;
 * get_attention_map, modules/editing/ptp_editor.py:43-85): for each image and each of n_tok token indices,
 * mean over the 40 head-layers of (store / steps_done), / max, bicubic -> LxL, clamp[0,1].
 * tokens [n_img][n_tok] int32 device.  out [n_img][n_tok][L][L] float: written (accumulate == 0) or
 * accumulated with weight `scale` (accumulate == 1; scale = 1/S gives the "fwd_mean" map, :392-396). */
int etainv_maps_word_maps(etainv_engine_t* e, int n_img, const int32_t* tokens, int n_tok, int steps_done,
                          float* out, int accumulate, float scale, void* stream);
/* same, for one role of the backward-pass store: row_sel 0 = source cond row, 1 = target cond row (the `bwd_source` /
 * `bwd_target` eta-mask sources, reference modules/inversion/eta_inversion.py:176-183 via ptp_editor.py:43-85) */
int etainv_maps_word_maps_role(etainv_engine_t* e, int n_img, const int32_t* tokens, int n_tok, int steps_done, int row_sel, float* out,
                               int accumulate, float scale, void* stream);

/* Which cross-attention layers the store keeps: res_div 4 (default) = the five (L/4)^2 layers [down, down, up, up, up]; 2 = the five (L/2)^2 layers
 * (same order; allocated on first use, 4x the default store); 8 = the mid block's (L/8)^2 layer.  Non-default `attn_res` of the eta mask
 * (modules/inversion/eta_inversion.py:161; aggregate_attention keeps the layers whose token count is res^2, modules/utils/ptp.py:288-303).  Clears the
 * store.  LocalBlend needs res_div 4. */
int etainv_maps_configure(etainv_engine_t* e, int res_div, void* stream);
/* etainv_maps_word_maps_role with `attn_from_where` (eta_inversion.py:162) as a mask over the stored layers: bit l = layer l of the order above
 * ("down" = 0x03, "up" = 0x1c, both = 0x1f; res_div 8: bit 0 = "mid").  A mask that selects no stored layer is an error (the reference fails in
 * torch.cat of an empty list, ptp.py:301). */
int etainv_maps_word_maps_ex(etainv_engine_t* e, int n_img, const int32_t* tokens, int n_tok, int steps_done, int row_sel, unsigned layer_mask,
                             float* out, int accumulate, float scale, void* stream);

/* LocalBlend (modules/utils/ptp.py:18-47) on the backward latents x [2*n_img][4][L][L] (in place, fp32):
 * blend_alpha [n_img][2][77] selects the blend-word tokens of (source, target) prompt. */
int etainv_local_blend(etainv_engine_t* e, float* x, int n_img, const float* blend_alpha, float thres, void* stream);

/* Workspace statistics for DESIGN.md / bench (bytes). */
/* The loops of DiffusionInversion.diffusion_forward / sample (reference modules/inversion/diffusion_inversion.py:388-418, 493-528) pass the SAME context
 * tensor to every UNet call.  enable = 1: etainv_unet_forward reuses the cross-attention K / V projections of the context when the context
 * pointer, row count and dtype equal the previous call's -- the caller promises not to change the buffer's contents in between and switches the
 * cache off (enable = 0) when the loop ends.  Off by default: every call projects the context it is given. */
int etainv_engine_cache_context(etainv_engine_t* e, int enable);
/* Generation of the context buffer: a cached projection is reused only while the generation equals the one it was computed under.  A caller
 * that rewrites the context tensor IN PLACE between two calls (same pointer, rows, dtype) bumps it; the built-in loops pass a fresh value per
 * loop.  The cache entry is recorded only after a forward has succeeded: an error half-way never leaves stale K / V behind. */
int etainv_engine_context_generation(etainv_engine_t* e, uint64_t generation);
/* hipGraph replay of small UNet calls (opt-in: ETAINV_GRAPH_MAX_ROWS=<rows>, read at engine creation; measured no faster than the eager launches
 * at batch 1 -- the dependent-kernel gap is the same inside a graph): from its
 * second occurrence on, a call signature (rows, latents, dtype, attention-control flags without per-step device tables) is captured once and replayed
 * -- latent / context / output staged through engine-owned buffers, the timesteps through a device vector.  Same results as the eager launches
 * (the replay IS those launches); captures / replays counted since engine creation. */
int etainv_engine_graph_stats(etainv_engine_t* e, int64_t* captures, int64_t* replays);
/* Default since round 4 (ETAINV_QKV_HM=0 at engine creation switches it off): the fused QKV projection of a transformer block writes
 * three head-major planes [q|k|v][row][head][token][head_dim] where both its GEMM kernel and the self-attention kernel can (head_dim 40 / 80, 16-bit modes,
 * whole 256-row tiles inside one batch row), so that a 64-key tile is one contiguous block.  Same values, same arithmetic: results are bit-identical.
 * `launches` = how many QKV projections took that path since the engine was created. */
int etainv_engine_qkv_head_major_count(etainv_engine_t* e, long long* launches);
int64_t etainv_engine_workspace_bytes(etainv_engine_t* e);
int64_t etainv_engine_weight_bytes(etainv_engine_t* e);

/* Opt-in per-kernel-class timing (HIP events recorded on the launch stream around every launch of the class).
 * Classes: 0 implicit-GEMM (conv/linear, work = FLOPs), 1 self-attention (FLOPs), 2 cross-attention (FLOPs),
 * 3 GroupNorm (bytes), 4 LayerNorm (bytes).  etainv_prof_read synchronises the device. */
int etainv_prof_enable(int on);
int etainv_prof_reset(void);
int etainv_prof_read(int cls, double* ms, double* work, int64_t* launches);
/* The individual launches of one class since the last reset, in launch order: fills ms[i], work[i] for i < min(n, cap) and
 * returns n through *launches (per-shape breakdown of a UNet call: tools/unet_call.py --shapes). */
int etainv_prof_records(int cls, double* ms, double* work, int64_t cap, int64_t* launches);
/* same + bytes[i] = the algorithmic HBM bytes of launch i (every operand read once, the result written once; 0 where the class records none) */
int etainv_prof_records_ex(int cls, double* ms, double* work, double* bytes, int64_t cap, int64_t* launches);

/* Roofline split of the launches of one class by arithmetic intensity (algorithmic FLOPs / algorithmic HBM bytes of each launch): out6 =
 * {ms, FLOPs, bytes} of the launches at or above `ridge` FLOP/byte (MFMA-bound), then of those below it (HBM-bound); launches2 = their counts.
 * Only the implicit-GEMM class records bytes. */
int etainv_prof_split(int cls, double ridge, double* out6, int64_t* launches2);

/* Per-op entry points used by the parity tests (tests/test_kernels_gpu.py) -- the same launchers the
 * executor uses, exposed so every kernel is checked against a plain fp32 reference in isolation.
 * Activations NHWC in the compute dtype; weights in the engine layouts described in DESIGN.md. */
int etainv_op_gemm(const void* a, const void* w, const void* bias, const void* residual, void* out,
                   int m, int n, int k, int geglu, int dtype, void* stream);
/* LayerNorm folded into the two GEMMs around it (reference: BasicTransformerBlock norm1/2/3 of [3P] diffusers 0.21.1 attention.py, called from
 * Transformer2DModel inside `self.unet(...)`, modules/inversion/eta_inversion.py:321).  The GEMM that writes the LayerNorm's input leaves per-row
 * (mean, M2) partials [m][P][2] in stat_out (*stat_p_out = P, each over n / P columns; 0 = this launch shape emits none: use etainv_op_row_stats);
 * etainv_op_ln_finalize turns them into stat[m] = (mean, rstd); the GEMM that consumes the LayerNorm runs on the RAW rows with the weights of
 * etainv_op_ln_fold: out = rstd (a W'^T - mean s) + c.  stat == NULL: a plain GEMM with bias c_vec. */
int etainv_op_gemm_ln(const void* a, const void* w_folded, const float* c_vec, const float* s_vec, const float* stat,
                      const void* residual, void* out, float* stat_out, int* stat_p_out, int m, int n, int k, int geglu,
                      int dtype, void* stream);
/* W' = gamma . W (packed, compute dtype; geglu: the value / gate row interleave of the GEGLU projection), s = row sums of the rounded W',
 * c = beta W^T + bias (bias in logical row order, may be NULL) */
int etainv_op_ln_fold(const float* w, const float* gamma, const float* beta, const float* bias, int n, int k, int geglu, float scale,
                      void* w_out, float* s_out, float* c_out, int dtype, void* stream);
int etainv_op_row_stats(const void* x, float* stat, int rows, int c, float eps, int dtype, void* stream);
int etainv_op_ln_finalize(const float* partials, int p, int cw, float eps, float* stat, int rows, void* stream);
/* GroupNorm statistics from the producing GEMM's epilogue (reference: the GroupNorms of [3P] diffusers ResnetBlock2D / Transformer2DModel inside the
 * UNet call, eta_inversion.py:321): etainv_op_gemm_gnstat = etainv_op_gemm that also leaves per-channel (sum, sum of squares) partials of its stored
 * output, part[m / wm][2][n] (*wm_out rows per block; 0 = this launch shape emits none); etainv_op_groupnorm_pre = GroupNorm(+SiLU) of cat[x1, x2]
 * from such partials (no statistics pass over x); final_stats: b * (2 * groups + 2 * (c1 + c2)) floats of scratch ((mean, rstd) per group, then the
 * apply pass's per-channel scale / shift planes). */
int etainv_op_gemm_gnstat(const void* a, const void* w, const float* bias, const void* residual, void* out, float* part, int* wm_out, int m,
                          int n, int k, int rows_per_image, int dtype, void* stream);
/* etainv_op_conv3x3 (stride 1, single source, nine taps) that also leaves the GroupNorm partials of its stored output, as etainv_op_gemm_gnstat does for
 * the 1x1 layers: conv1 / conv2 of [3P] diffusers ResnetBlock2D feed the next GroupNorm (inside the UNet call of eta_inversion.py:321). */
int etainv_op_conv3x3_gnstat(const void* x_nhwc, const void* w_okkc, const float* bias, const float* rowvec, const void* residual, void* out,
                             float* part, int* wm_out, int b, int h, int wd, int cin, int cout, int dtype, void* stream);
int etainv_op_groupnorm_pre(const void* x1, const void* x2, int c1, int c2, const float* part1, int wm1, const float* part2, int wm2,
                            const float* gamma, const float* beta, void* out, int b, int hw, int groups, float eps, int silu,
                            float* final_stats, int dtype, void* stream);
/* conv3x3 behind a nearest-2x upsample (diffusers Upsample2D inside the UNet call of eta_inversion.py:321) in PHASE form: the nine taps on the upsampled image
 * are four 2x2 convs on the source image, one per output phase (y & 1, x & 1) -- taps that land on the same source pixel are summed in fp32 and rounded
 * once (4 / 9 of the FLOPs, the same result up to that rounding).  w_oc33 [cout][cin][3][3] fp32 -> dst [4][cout][4][cin] in the compute dtype; run it with
 * etainv_op_conv3x3(..., upsample = 2, taps = 4): the 256 x 160 ring only (h * wd % 256 == 0, enough rows; otherwise an error -- upsample = 1 is the 9-tap form). */
int etainv_op_pack_ups4(const float* w_oc33, void* dst, int cout, int cin, int dtype, void* stream);
int etainv_op_conv3x3(const void* x_nhwc, const void* x2_nhwc, int c1, int c2, const void* w_okkc, const void* bias,
                      const float* rowvec, const void* residual, void* out, int b, int h, int wd, int cout,
                      int stride, int upsample, int taps, int dtype, void* stream);
/* extended conv for the VAE: pad0 = taps start at the output origin (asymmetric (0,1,0,1) padding of its stride-2
 * downsamplers); out_nchw = 1..4: cout must be 4 and the first out_nchw channels are written as io-dtype NCHW */
int etainv_op_conv3x3_ex(const void* x_nhwc, const void* w_okkc, const void* bias, const void* residual, void* out, int b, int h,
                         int wd, int cin, int cout, int stride, int upsample, int pad0, int out_nchw, int out_io_dtype, int dtype,
                         void* stream);
/* VAE / CLIP text-encoder helpers (third-party networks outside the DDIM loop; reference call sites
 * modules/inversion/diffusion_inversion.py:183-247) */
int etainv_op_im2col3x3(const void* x_nchw, int io_dtype, int cin, int h, int w, int rows, const float* premix, void* out,
                        int dtype, void* stream);
int etainv_op_row_softmax(void* x, int rows, int n, float scale, int dtype, void* stream);
int etainv_op_quick_gelu(const void* x, void* out, int64_t n, int dtype, void* stream);
int etainv_op_embed(const int64_t* ids, const void* tok, const void* pos, int b, int n_pos, int d, void* out, int dtype,
                    void* stream);
int etainv_op_causal_attention(const void* qkv, void* out, int b, int n, int heads, int d, int dtype, void* stream);
int etainv_op_groupnorm(const void* x_nhwc, const void* x2_nhwc, int c1, int c2, const float* gamma,
                        const float* beta, void* out, int b, int hw, int groups, float eps, int silu,
                        float* scratch, int dtype, void* stream);
int etainv_op_layernorm(const void* x, const float* gamma, const float* beta, void* out, int rows, int c,
                        float eps, int dtype, void* stream);
/* mode 0 plain, 1 ptp self-replace, 2 masactrl (modes 1/2: b == 4*n_img rows [u_s,u_t,c_s,c_t]) */
int etainv_op_self_attention(const void* qkv, void* out, int b, int n, int heads, int d, int mode, int n_img,
                             int dtype, void* stream);
int etainv_op_cross_attention(const void* q, const void* kv, void* out, int b, int n, int heads, int d,
                              int n_ctx, const etainv_attn_ctrl* ctrl, int map_layer, int n_img_cap,
                              float* maps_acc, int dtype, void* stream);
int etainv_op_word_maps(const float* maps_acc, int n_layers, int n_img_cap, int heads, int res, int L, int n_img,
                        const int32_t* tokens, int n_tok, int steps_done, float* out, int accumulate, float scale,
                        void* stream);
int etainv_op_local_blend(const float* maps_acc, int n_layers, int n_img_cap, int heads, int res, int L, float* x,
                          int n_img, const float* blend_alpha, float thres, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ETAINV_H */
