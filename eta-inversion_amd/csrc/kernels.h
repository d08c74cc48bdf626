// Internal launcher interface between the UNet executor (engine.cpp) and the kernel files.
#pragma once
#include "common.h"

namespace etainv {

// ---- igemm.hip
struct IGemmParams {
  const void* a1 = nullptr;   // activations, NHWC [B][H][W][c1]
  const void* a2 = nullptr;   // optional second source (channel concat), NHWC [B][H][W][c2]
  const void* w = nullptr;    // weights [N][taps][c1+c2]
  const float* bias = nullptr;      // [N] (physical column order)
  const float* rowvec = nullptr;    // [batch][rowvec_stride] fp32, added per batch row (time-embedding projection)
  int rowvec_stride = 0;
  int out_f32 = 0;                  // store fp32 instead of T (time-embedding projections)
  int out_nchw = 0;                 // conv_out: N == 4, store the first out_nchw channels as io-dtype NCHW [batch][out_nchw][Ho*Wo]
  int pad0 = 0;                     // 3x3 taps start at the output origin (VAE downsampler: F.pad(x,(0,1,0,1)) + stride-2 conv, no top/left pad)
  int out_io_dtype = 0;
  const void* residual = nullptr;   // [M][N]
  void* out = nullptr;              // [M][N]  (or [M][N/2] with geglu)
  const void* zeros = nullptr;      // filled in by launch_igemm
  // timing experiments only (ETAINV_IGEMM_DEBUG bit mask): 1 no DMA in the loop, 2 no epilogue, 4 no MFMA, 8 stores hit cache-resident
  // rows, 16 no young-store vmcnt allowance, 32 8-byte stores (no lane swap), 64 epilogue without its stores, 128 scalar A&S erf in GEGLU
  int debug = 0;
  // Head-major output of a fused QKV projection (round 4): out = three planes [part q|k|v][batch row][head][token][hm_dim] instead of [M][3 * heads * hm_dim],
  // so that a 64-key tile of one head is one contiguous block for the attention kernel (igemm_hm_ok says whether a launch can do it: LayerNorm
  // consumer on the 256 x 160 ring, hm_dim 40 / 80, whole tiles inside one batch row).  0 = row-major
  int hm_heads = 0, hm_dim = 0, hm_tokens = 0, hm_magic = 0;   // (hm_magic: filled in by launch_igemm)
  unsigned long* stamps = nullptr;         // diagnostic build (-DETAINV_IGEMM_STAMPS) only
  int stagger = 0;                  // experiment (ETAINV_STAGGER): start delay of the second co-resident block, 64-cycle ticks
  int M = 0, N = 0;
  int c1 = 0, c2 = 0;
  int H = 1, W = 1;           // source spatial dims (before the fused upsample)
  int Ho = 1, Wo = 1;         // output spatial dims
  int stride = 1, ups = 0, taps = 1;
  int ups_pm = 0;                   // ups == 2: virtual rows ordered phase-major [phase][image][source pixel] (filled in by launch_igemm: source images that are not
                                    // whole 256-row tiles -- 8 x 8, 24 x 24 -- where [image][phase][pixel] would put two phases' kernels into one tile)
  int geglu = 0;
  int xcd_gn = 1;             // XCD grid along N for the tile order (1, 2, 4 or 8; chosen by launch_igemm_t from a traffic model, see there)
  int ksplit = 1;             // split-K parts (filled in by launch_igemm for small M*N with deep K)
  float* ws = nullptr;        // [ksplit][M][N] fp32 partials
  int rows_per_batch = 1;     // Ho*Wo for convs; M/batch for linears
  // per-image operands (a GroupNorm folded into the 1x1 conv that follows it): rows of image b use w + b * w_batch_stride (elements) and
  // bias + b * bias_batch_stride; 0 = one weight matrix.  Needs rows_per_batch % (M tile) == 0.
  int64_t w_batch_stride = 0;
  int bias_batch_stride = 0;
  // ---- LayerNorm folded into the GEMM pair around it (the transformer blocks' norm1/2/3: no LayerNorm pass over HBM).
  // Producer side (the GEMM that writes the LayerNorm's input x): per row and per column chunk of the STORED (rounded) output, the pair
  // (mean, M2 = sum of squared deviations from that mean) -> stat_out[m][stat_P][2]; stat_P is chosen by launch_igemm (columns of one wave
  // tile per chunk) and reported through its stat_P argument; 0 = this launch could not emit them (the caller runs launch_row_stats).
  float* stat_out = nullptr;
  int stat_P = 0;
  // stat_kind 1: GroupNorm producer instead -- stat_out[row block][2][N]: per channel, the sum and the sum of squares of the stored output over the
  // rows of each wave tile (stat_P = rows per block, reported by launch_igemm); launch_groupnorm_pre consumes them
  int stat_kind = 0;
  // Consumer side (K = LayerNorm width, weights pre-multiplied by gamma: launch_ln_fold): out = rstd[m] * (acc - mean[m] * ln_s[n]) + bias[n],
  // bias = the folded c vector; ln_stat = (mean, rstd) per row, [M][2] (launch_ln_finalize of the producer's partials, or launch_row_stats).
  const float* ln_stat = nullptr;
  const float* ln_s = nullptr;      // [N] physical column order, like bias
};
// stat_P (optional): receives the number of partials per row written to p.stat_out (0: none written)
int launch_igemm(const IGemmParams& p, int dtype, hipStream_t s, int* stat_P = nullptr);
bool igemm_hm_ok(const IGemmParams& p, int dtype);     // may this launch (with hm_* set) write the head-major layout?
bool igemm_ups4_ok(const IGemmParams& p, int dtype);   // may this launch run the phase form (ups == 2, taps == 4) of a fused-upsample conv?

// ---- xsgemm.hip: K = 320 LayerNorm-consumer projections (GEGLU, fused QKV) of the L^2-token blocks on a stationary activation tile with two wave
// groups in anti-phase; launch_igemm routes to it when xs_gemm_applicable (bit-identical results)
bool xs_gemm_applicable(const IGemmParams& p, int dtype);
int launch_xs_gemm(const IGemmParams& p, int dtype, hipStream_t s);

// ---- ppgemm.hip: the 256 x 160 tile with its two wave groups in anti-phase (ping-pong) and the epilogue of a tile under the next tile's main loop (two
// accumulator sets): 1x1 / Linear with bias (+ residual, + LayerNorm row statistics), K >= 320; launch_igemm routes to it when pp_gemm_applicable
bool pp_gemm_applicable(const IGemmParams& p, int dtype);
int launch_pp_gemm(const IGemmParams& p, int dtype, hipStream_t s, int* stat_P = nullptr);
// dual-N form: a 256 x 320 output tile as two 160-column halves sharing one staged activation K tile (-31 % LDS-DMA bytes per FLOP; the 1x1 GEMMs are bound
// by the DMA fill rate): bias (+ residual) (+ LayerNorm row statistics), LayerNorm consumer (row-major / head-major QKV planes), and -- as a 256 x 256 tile
// of two 128-column halves -- the LayerNorm-consumer GEGLU projection
bool pp_dualn_applicable(const IGemmParams& p, int dtype);
bool pp_dualn_hm_ok(const IGemmParams& p, int dtype);   // ... as a LayerNorm consumer that writes the head-major QKV planes (hm_* set)?
int launch_pp_dualn(const IGemmParams& p, int dtype, hipStream_t s, int* stat_P = nullptr);

// ---- ppconv.hip: conv3x3 stride 1 in PATCH form (16 x 16 pixel tiles, halo'd activation patch per channel chunk) with the wave groups in anti-phase and a
// lean issue side; bias + time row + residual + GroupNorm-statistics epilogues; launch_igemm routes to it when pp_conv_applicable
bool pp_conv_applicable(const IGemmParams& p, int dtype);
int launch_pp_conv(const IGemmParams& p, int dtype, hipStream_t s, int* stat_P = nullptr);

// ---- f32path.hip: the fp32-operand execution (dtype == ETAINV_F32 routes here from the launchers of igemm / norm / attention)
int launch_igemm_f32(const IGemmParams& p, hipStream_t s);
int launch_groupnorm_f32(const void* x1, const void* x2, int c1, int c2, const float* gamma, const float* beta, void* out, int b, int hw, int groups,
                         float eps, int silu, float* scratch, hipStream_t s);
int launch_layernorm_f32(const void* x, const float* gamma, const float* beta, void* out, int rows, int c, float eps, hipStream_t s);
int launch_self_attention_f32(const void* qkv, void* out, int b, int n, int heads, int d, int mode, int n_img, hipStream_t s, int first_row = 0);
struct CrossParams;
int launch_cross_attention_f32(const void* q, const void* kv, void* out, int b, int d, const CrossParams& p, hipStream_t s);

// ---- norm.hip
// GroupNorm(32 groups) over NHWC with optional second (concatenated) source and fused SiLU.
// scratch: >= b * GN_CHUNKS_MAX * groups * 2 floats
constexpr int GN_MAX_CHUNKS = 64;
int launch_groupnorm(const void* x1, const void* x2, int c1, int c2, const float* gamma, const float* beta, void* out,
                     int b, int hw, int groups, float eps, int silu, float* scratch, int dtype, hipStream_t s);
// GroupNorm whose statistics arrive as per-channel partials from the epilogues of the GEMMs that wrote x1 / x2 (IGemmParams::stat_kind 1; part =
// [b * hw / wm][2][c] floats): a finalize launch (sums in double, fixed order) -> final_stats = [b][groups][2] (mean, rstd) followed by the apply
// pass's per-channel scale / shift planes [b][2][c1 + c2] (the buffer holds b * (2 * groups + 2 * C) floats), then the apply pass of launch_groupnorm
int launch_groupnorm_pre(const void* x1, const void* x2, int c1, int c2, const float* part1, int wm1, const float* part2, int wm2, const float* gamma,
                         const float* beta, void* out, int b, int hw, int groups, float eps, int silu, float* final_stats, int dtype, hipStream_t s);
// the finalize half of launch_groupnorm_pre alone: final_stats[b][groups] = (mean, rstd)
int launch_gn_finalize(int c1, int c2, const float* part1, int wm1, const float* part2, int wm2, int b, int hw, int groups, float eps, float* final_stats,
                       hipStream_t s, const float* gamma = nullptr, const float* beta = nullptr, float* scsh = nullptr);
// GroupNorm (no activation) folded into the 1x1 conv / Linear W [n][k] that follows it: per image b, W_b[n][k] = W[n][k] * rstd[b][g(k)] * gamma[k]
// (rounded to the compute dtype) and c_b[n] = sum_k (beta[k] - mean[b][g(k)] * rstd * gamma[k]) * W[n][k] + bias[n]
int launch_gn_fold(const float* w, const float* gamma, const float* beta, const float* bias, const float* final_stats, int groups, int b, int n, int k,
                   void* wb_out, float* cb_out, int dtype, hipStream_t s);
int launch_layernorm(const void* x, const float* gamma, const float* beta, void* out, int rows, int c, float eps, int dtype,
                     hipStream_t s);
// stat[row] = (mean, rstd) of x[row][0..c): IGemmParams::ln_stat computed by a pass over x (when the producing GEMM could not emit partials)
int launch_row_stats(const void* x, float* stat, int rows, int c, float eps, int dtype, hipStream_t s);
// partials[row][P] (mean, M2) pairs over cw columns each (IGemmParams::stat_out) -> stat[row] = (mean, rstd); Chan's update in index order
int launch_ln_finalize(const float* partials, int P, int cw, float eps, float* stat, int rows, hipStream_t s);

// ---- attention.hip
// self-attention modes: 0 plain; 1 ptp self-replace (cond target rows use Q,K of their source row);
// 2 masactrl (target rows use K,V of their source row).  Modes 1/2 need the 4*n_img backward row layout.
// A/B switch ETAINV_ATT_OLD: head_dim 40 on the generic 16x16x32 kernel (then the engine does not fold the scale into to_q)
bool self_attn40_v2_enabled();
// q_prescaled (d == 40 only): the queries already carry softmax scale * log2(e) (the engine folds it into the to_q weights)
// first_row: the call carries rows [first_row, 4 n_img) of the [u_s, u_t, c_s, c_t] x n_img layout (0, or n_img when the uncond source rows are left out:
// backward steps with eta == 0, etainv/pipeline.py)
int launch_self_attention_mode(const void* qkv, void* out, int b, int n, int heads, int d, int mode, int n_img, int dtype,
                               hipStream_t s, int q_prescaled = 0, int first_row = 0, int head_major = 0);
bool self_attn_head_major_ok(int d, int dtype);
struct CrossParams {
  int N = 0, heads = 8, n_ctx = 77;
  float scale_log2 = 0.f;
  int layout = 0;      // 0: no roles (plain); 1: forward store layout [u x B, c x B] or [c x B]; 2: backward 4B layout
  int n_img = 1;
  int rows = 0;        // batch rows in this call
  int first_row = 0;   // layout 2: the call carries rows [first_row, 4 n_img) of [u_s, u_t, c_s, c_t] x n_img (0 or n_img)
  int edit = 0;        // ptp cross edit on cond target rows
  int map_layer = -1;  // >= 0: accumulate cond-half probabilities into maps_acc layer `map_layer`
  int n_img_cap = 1;
  const int32_t* mapper = nullptr;
  const float* alphas = nullptr;
  const float* replace_mat = nullptr;
  const float* equalizer = nullptr;
  const float* cross_alpha = nullptr;
  float* maps_acc = nullptr;
  int xcd_gx = 0;      // > 0: 1-D grid with the heads of one (row, query range) on one XCD; the value = query-block lanes per (row, head) (attention.hip)
};
int launch_cross_attention_p(const void* q, const void* kv, void* out, int b, int d, const CrossParams& p, int dtype, hipStream_t s);

// ---- misc.hip
// conv_in: NCHW io-dtype latent [n_lat][4][L][L] (row r reads r % n_lat) -> NHWC T [rows][L*L][cout], 3x3 pad 1
int launch_conv_in(const void* latent, int io_dtype, int n_lat, int rows, int L, const void* w, const float* bias, int cout,
                   void* out, int dtype, hipStream_t s);
// im2col of the 3x3 / 4-channel input conv: NCHW io-dtype latent [n_lat][4][L][L] -> T [rows][L*L][64] (k = tap*4 + ci, 36..63 zero)
int launch_im2col_in(const void* latent, int io_dtype, int n_lat, int rows, int L, void* out, int dtype, hipStream_t s);
// conv_out: NHWC T [rows][L*L][cin] (already GroupNorm+SiLU'd) -> NCHW io-dtype [rows][4][L][L]
int launch_conv_out(const void* x, int rows, int L, int cin, const void* w, const float* bias, void* out, int io_dtype, int dtype,
                    hipStream_t s);
// sinusoidal timestep embedding (flip_sin_to_cos, freq_shift 0): t_host [rows] (HOST array, passed by value in the kernel arguments) -> [rows][dim] T
int launch_time_embedding(const int64_t* t_host, int rows, int dim, void* out, int dtype, hipStream_t s);
// the same from DEVICE timesteps t_dev [rows] floats (written by launch_set_timesteps): replayable inside a captured hipGraph
int launch_set_timesteps(const int64_t* t_host, int rows, float* t_dev, hipStream_t s);
int launch_time_embedding_dev(const float* t_dev, int rows, int dim, void* out, int dtype, hipStream_t s);
// y = silu(x) elementwise on T
int launch_silu(const void* x, void* out, int64_t n, int dtype, hipStream_t s);
// cast fp32 -> T with optional row permutation (weights)
int launch_pack_ups4(const float* src, void* dst, int cout, int cin, int dtype, hipStream_t s);   // conv3x3 weights -> four 2x2 phase kernels (IGemmParams::ups == 2)
int launch_pack_weight(const float* src, void* dst, int64_t rows, int64_t cols, int mode, int taps, int dtype, hipStream_t s, float scale = 1.0f,
                       const float* colscale = nullptr);
// LayerNorm(gamma, beta) folded into the Linear (w_src [rows][cols] fp32, pack mode 0 / 2) that consumes it: packed operand W' = gamma . W, the
// row sums s of the rounded W' and c = beta W^T + bias (bias_packed may be null) -- see IGemmParams::ln_stat
int launch_ln_fold(const float* w_src, const float* gamma, const float* beta, const float* bias_packed, int64_t rows, int64_t cols, int mode, float scale,
                   void* w_dst, float* s_dst, float* c_dst, int dtype, hipStream_t s);
int launch_cast_f32(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, hipStream_t s);

// ---- maps.hip
int launch_word_maps(const float* maps_acc, int n_layers, int n_img_cap, int rows_per_img, int row_sel, int heads, int res, int L,
                     int n_img, const int32_t* tokens, int n_tok, int steps_done, float* out, int accumulate, float scale,
                     hipStream_t s, unsigned layer_mask = ~0u);
int launch_local_blend(const float* maps_acc, int n_layers, int n_img_cap, int heads, int res, int L, float* x, int n_img,
                       const float* blend_alpha, float thres, hipStream_t s);

}  // namespace etainv
