/*
 * advgrpo.h -- C ABI of libadvgrpo_hip.so: the MI355X (gfx950) kernels behind the Adv-GRPO
 * SD3 rollout-and-update hot path.
 *
 * The reference (showlab/Adv-GRPO) is 100 % Python and owns no native boundary; each entry
 * point below names the reference Python site(s) it replaces (paths relative to the upstream
 * checkout).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer unless the name ends in _host;
 *     outputs are pre-allocated by the caller; no entry point allocates, frees or synchronises
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on that stream and
 *     thread-safe for distinct streams (the reward scorers are called from worker threads,
 *     scripts/train_sd3_fast_pickscore.py:668,816-817)
 *   - return 0 on success, <0 on error (message: advgrpo_last_error(), thread-local)
 *   - tensors are dense row-major; dtype codes below; "tokens" layouts are [rows, features]
 */
#ifndef ADVGRPO_H
#define ADVGRPO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADVGRPO_ABI_VERSION 1

enum advgrpo_dtype { ADVGRPO_F32 = 0, ADVGRPO_BF16 = 1, ADVGRPO_F64 = 2 };

int advgrpo_abi_version(void);
const char* advgrpo_last_error(void);

/* ------------------------------------------------------------------ RNG
 * Philox4x32-10 standard normals, element i uses counter (offset + i/4).  Replaces the
 * global-device-RNG draws of prepare_latents (sd3_pipeline_with_logprob_fast.py:559-568) and
 * randn_tensor (sd3_sde_with_logprob.py:125-130).  Not bit-compatible with torch's stream
 * (SURVEY.md section 7 "RNG parity"): parity tests inject epsilon instead. */
int advgrpo_randn(void* out, int out_dtype, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* ------------------------------------------------------------------ SDE step
 * Fused CFG combine + Flow-CPS SDE step + per-sample Gaussian log-prob.
 * Replaces sd3_pipeline_with_logprob_fast.py:640-655 (CFG, step, cast back) and
 * sd3_sde_with_logprob.py:77-139; in replay mode, train_sd3_fast_pickscore.py:242-265.
 *
 *   v      = v_text ? v_uncond + guidance*(v_text - v_uncond) : v_uncond     (in v_dtype; bf16 =>
 *            three bf16 roundings exactly like the three torch ops), then promoted to f32
 *   std    = sigma_prev * sin_coeff          sin_coeff = (float)sin(noise_level*pi/2), host-side
 *   mean   = (x - sigma*v)*(1-sigma_prev) + (x + v*(1-sigma))*sqrt(sigma_prev^2 - std^2)
 *   next   = mode==REPLAY ? prev_sample : mean + std*eps
 *   log_prob[b] = mean_over_n( -(next-mean)^2 )
 * All f32, one rounding per reference op (no FMA contraction): mean/next are bit-exact with the
 * reference's torch-CPU result; log_prob differs only by summation order.
 *
 * sigma / sigma_prev: device f32, element b*sigma_stride (stride 0 = one value for all samples).
 * eps: f32 [B,n] (mode EPS) ; seed/offset (mode PHILOX); prev_sample [B,n] in prev_dtype (REPLAY).
 * out_next_f32, out_next_cast (dtype out_cast_dtype), out_mean: optional (NULL to skip).
 * workspace: >= advgrpo_sde_step_workspace_bytes(B, n) bytes. */
enum advgrpo_sde_mode { ADVGRPO_SDE_EPS = 0, ADVGRPO_SDE_PHILOX = 1, ADVGRPO_SDE_REPLAY = 2 };

int64_t advgrpo_sde_step_workspace_bytes(int B, int64_t n);

int advgrpo_sde_step(const void* v_uncond, const void* v_text, int v_dtype, float guidance_scale,
                     const void* x, int x_dtype,
                     const float* sigma, const float* sigma_prev, int sigma_stride, float sin_coeff,
                     int mode, const float* eps, uint64_t seed, uint64_t offset,
                     const void* prev_sample, int prev_dtype,
                     float* out_next_f32, void* out_next_cast, int out_cast_dtype,
                     float* out_mean, float* out_log_prob, float* out_std,
                     void* workspace, int B, int64_t n, void* stream);

/* Backward of the replay-mode step w.r.t. the two CFG halves of the transformer output
 * (autograd of train_sd3_fast_pickscore.py:242-265 as reached from loss.backward(), :1165):
 *   g_v = grad_log_prob[b] * 2*(prev-mean)/n * ((1-sigma)*sqrt(sigma_prev^2-std^2) - sigma*(1-sigma_prev))
 *   grad_v_text = guidance*g_v ; grad_v_uncond = (1-guidance)*g_v      (written in v_dtype)
 * Without CFG (v_text NULL) grad_v_uncond = g_v and grad_v_text is ignored. */
int advgrpo_sde_step_bwd(const void* v_uncond, const void* v_text, int v_dtype, float guidance_scale,
                         const void* x, int x_dtype,
                         const float* sigma, const float* sigma_prev, int sigma_stride, float sin_coeff,
                         const void* prev_sample, int prev_dtype, const float* grad_log_prob,
                         void* grad_v_uncond, void* grad_v_text, int B, int64_t n, void* stream);
/* The same backward with the KL term of TP:1105-1108,1126-1130 (config.train.beta > 0):
 *   loss += kl_weight * mean_b mean_elem (mean - mean_ref)^2,
 * mean_ref [B, n] f32 = prev_sample_mean of the same step under the adapter-free transformer (advgrpo_sde_step's out_mean of
 * that forward).  The KL gradient joins the policy gradient before the bf16 rounding autograd applies once to their sum.
 * kl_out[b] = mean_elem (mean - mean_ref)^2 (partials through `workspace`, advgrpo_sde_step_workspace_bytes, fixed order). */
int advgrpo_sde_step_bwd_kl(const void* v_uncond, const void* v_text, int v_dtype, float guidance_scale,
                            const void* x, int x_dtype, const float* sigma, const float* sigma_prev,
                            int sigma_stride, float sin_coeff, const void* prev_sample, int prev_dtype,
                            const float* grad_log_prob, const float* mean_ref, float kl_weight,
                            void* grad_v_uncond, void* grad_v_text, float* kl_out, void* workspace, int B,
                            int64_t n, void* stream);

/* ------------------------------------------------------------------ group advantage
 * PerPromptStatTracker.update(type='grpo') on a fresh tracker -- adv_grpo/stat_tracking.py:18-47:
 * float64, per-group mean over rows, std = global (np.std over all rows, per column) or
 * per-group, +1e-4; adv = (r - mean_g)/std.  group_id replaces the decoded prompt strings
 * (train_sd3_fast_pickscore.py:960-970): any int32 key, equal key <=> same prompt.
 * rewards: [N,T] in rewards_dtype (F32 as gathered, or F64); out_adv: [N,T] f64.
 * Sums run in row order like numpy's axis-0 reduction, so results are bit-exact with it. */
int advgrpo_group_advantage(const void* rewards, int rewards_dtype, const int32_t* group_id,
                            int N, int T, int global_std, double* out_adv, void* stream);
/* same launch + the logging statistics of calculate_zero_std_ratio, scripts/train_sd3_fast_pickscore.py:195-229
 * (logged at :975-988): out_stats[0] = zero_std_ratio (share of groups whose reward std is exactly 0),
 * out_stats[1] = reward_std_mean (mean of the per-group np.std), both from column 0 of `rewards` in ITS dtype's
 * arithmetic (float32 as gathered upstream), groups in ascending key order, numpy summation orders: bit-exact. */
int advgrpo_group_advantage_stats(const void* rewards, int rewards_dtype, const int32_t* group_id,
                                  int N, int T, int global_std, double* out_adv, double* out_stats, void* stream);

/* ------------------------------------------------------------------ GRPO loss
 * Clipped surrogate forward + backward + diagnostics, train_sd3_fast_pickscore.py:1111-1162
 * (beta == 0).  log_prob/old_log_prob/advantages: f32 [B].
 * out_scalars[6] = {loss, approx_kl, clipfrac, clipfrac_gt_one, clipfrac_lt_one, policy_loss};
 * out_grad_log_prob[B] = d loss / d log_prob (NULL to skip). */
int advgrpo_grpo_loss(const float* log_prob, const float* old_log_prob, const float* advantages,
                      int B, float adv_clip_max, float clip_range,
                      float* out_scalars, float* out_grad_log_prob, void* stream);

/* ------------------------------------------------------------------ dense contraction
 * C[M,N] = epilogue(A[M,K] . W[N,K]^T): bf16 operands (K contiguous, torch nn.Linear weight
 * layout), f32 accumulation on v_mfma_f32_16x16x32_bf16.  Carries every nn.Linear the hot path
 * reaches inside diffusers' SD3Transformer2DModel (sd3_pipeline_with_logprob_fast.py:630-637,
 * train_sd3_fast_pickscore.py:235-255), transformers' CLIPModel (adv_grpo/pickscore_scorer.py:40-44)
 * and timm's DINOv2 (adv_grpo/rewards.py:397).
 *   y = act(alpha*acc + bias[n]);  y *= gate[(m / gate_rows)*gate_stride + n];  y += residual[orow*ldr + n]
 *   orow = seg_rows ? (m / seg_rows)*seg_stride + seg_off + m % seg_rows : m      (row-segment scatter)
 *   A row m is read from arow = a_seg_rows ? (m / a_seg_rows)*a_seg_stride + a_seg_off + m % a_seg_rows : m
 * bias/gate/residual: bf16, optional (NULL).  act: 0 none, 1 GELU(tanh), 2 GELU(erf), 3 SiLU.
 * out_dtype: BF16 or F32.  K % 64 == 0; lda/ldw % 8 == 0.  batch > 1: grid-batched with element
 * strides (residual uses strideC). */
int advgrpo_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                      int out_dtype, int M, int N, int K, const void* bias, int act, float alpha,
                      const void* gate, int64_t gate_stride, int gate_rows, const void* residual,
                      int64_t ldr, int seg_rows, int64_t seg_stride, int64_t seg_off, int a_seg_rows,
                      int64_t a_seg_stride, int64_t a_seg_off, int batch,
                      int64_t strideA, int64_t strideW, int64_t strideC, void* stream);

/* Which kernel instance the GEMM dispatcher uses for a shape: tile 0 = 128x128 (BK 64, 2 stages), 1 = 128x64,
 * 2 = 64x128, 3 = 256x256, 6 = 256x128; with the implicit-conv loader 4, 5, 7, 8; 9..13 = deep-pipelined BK 32
 * variants (13 = 128x128, 3 stages, 8 waves).  (Per-kernel accounting in bench.py.) */
int advgrpo_gemm_variant(int M, int N, int K, int batch, int conv);

/* ------------------------------------------------------------------ row kernels (HBM bound)
 * layernorm_mod: out = (LN(x) [*w + b]) * (1 + scale[m / rows_per_batch]) + shift[...]; optional second
 * output with a second (scale1, shift1) from the same statistics.  Replaces nn.LayerNorm +
 * AdaLayerNormZero / AdaLayerNormContinuous modulation inside the transformer call and the ViT LayerNorms.
 * All tensors bf16; w/b/scale/shift optional; D % 8 == 0, D <= 2048. */
int advgrpo_layernorm_mod(const void* x, int64_t ldx, void* out0, void* out1, int64_t ldo,
                          const void* w, const void* b, const void* scale0, const void* shift0,
                          const void* scale1, const void* shift1, int64_t mod_stride, int rows_per_batch,
                          int M, int D, float eps, void* stream);
/* The same with fp8 outputs for the fp8 Linears (below): q0 / q1 [M, D] e4m3 codes (pitch ldq bytes) + qs0 / qs1 [M] f32 row
 * scales, bit for bit what advgrpo_quant_fp8_rows makes of out0 / out1 -- which may then be null (not written at all). */
/* Two such problems in ONE launch: the image-stream and the text-stream norm of an MMDiT block (norm1 / norm1_context and the
 * LayerNorms in front of the two feed-forwards, diffusers JointTransformerBlock.forward behind PF:630-637).  The text stream's 154
 * rows per sample are launch-bound alone; bit-identical to two separate calls.  Fields as the arguments above; q0 / qs0 (/ q1 /
 * qs1, ldq) non-null selects the fp8 form for BOTH problems. */
typedef struct advgrpo_ln_desc {
    const void* x; int64_t ldx;
    void* out0; void* out1; int64_t ldo;
    const void* w; const void* b;
    const void* scale0; const void* shift0; const void* scale1; const void* shift1;
    int64_t mod_stride; int rows_per_batch;
    int M, D; float eps;
    void* q0; float* qs0; void* q1; float* qs1; int64_t ldq;
} advgrpo_ln_desc;
int advgrpo_layernorm_mod_pair(const advgrpo_ln_desc* a, const advgrpo_ln_desc* b, void* stream);
int advgrpo_layernorm_mod_fp8(const void* x, int64_t ldx, void* out0, void* out1, int64_t ldo, const void* w,
                              const void* b, const void* scale0, const void* shift0, const void* scale1,
                              const void* shift1, int64_t mod_stride, int rows_per_batch, int M, int D, float eps,
                              void* q0, float* qs0, void* q1, float* qs1, int64_t ldq, void* stream);
/* T5LayerNorm (transformers' T5: bf16(x * rsqrt(mean(x^2) + eps)) * w, f32 statistics, no bias): the norms of the T5-XXL
 * text encoder called by encode_prompt (adv_grpo/diffusers_patch/train_dreambooth_lora_sd3.py:19-56,125-133). */
int advgrpo_rmsnorm_rows(const void* x, int64_t ldx, void* out, int64_t ldo, const void* w, int M, int D, float eps,
                         void* stream);
/* In-place RMSNorm over 64-wide heads (SD3.5 qk_norm): rows of buf[., ld], heads at columns
 * [col0, col0 + 64*nheads); head hh is scaled by weight[(hh / heads_per_weight)*64 ...].  Row m maps
 * through (seg_rows, seg_stride, seg_off) like the GEMM output map. */
int advgrpo_rmsnorm_heads(void* buf, int64_t ld, int M, int col0, int nheads, const void* weight,
                          int heads_per_weight, float eps, int seg_rows, int64_t seg_stride,
                          int64_t seg_off, float* rs_out /* optional f32 [rows, nheads]: 1/rms, for the backward */,
                          void* stream);
/* Per-head RMSNorm (affine, eps) + rotary position embedding, in place, on the q | k heads of a packed JOINT QKV buffer
 * buf[rows = B * S, ld]: heads of head_dim (64 or 128) at columns [col0, col0 + nheads * head_dim).  Token row m is position
 * s = m % S of its sample; rows with s < n_first (image tokens) use w_first [nheads / heads_per_weight, head_dim], the others
 * (text tokens) w_rest.  rope [S, head_dim] f32 holds (cos, sin) of pair i of position s at [s, 2i], [s, 2i + 1] (NULL: norm only):
 *     y = bf16(bf16(x * rsqrt(mean x^2 + eps)) * w);   out[2i] = y[2i] cos - y[2i+1] sin,  out[2i+1] = y[2i] sin + y[2i+1] cos.
 * Replaces attn.norm_q / norm_k / norm_added_q / norm_added_k + apply_rotary_emb_qwen of diffusers' Qwen-Image attention
 * processor (BASELINE config 5's model; no code in the reference beyond README.md:75, config/grpo.py:324,330), as reached
 * through the transformer call of adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:630-637.
 * rs_out (optional, f32 [rows, nheads]): 1/rms per head, for the backward. */
int advgrpo_qk_norm_rope(void* buf, int64_t ld, int rows, int S, int n_first, int col0, int nheads, int head_dim,
                         const void* w_first, const void* w_rest, int heads_per_weight, float eps,
                         const float* rope, float* rs_out, void* stream);
/* Prompt encoder of BASELINE config 5 (the Qwen2.5-VL language model on text-only input behind QwenImagePipeline's `encode_prompt`;
 * the Qwen-Image twin of scripts/train_dreambooth_lora_sd3.py:98-144 as used at TP:628-651 -- the reference names the config at
 * config/grpo.py:324,330 and ships no code for it):
 *   rope_half: Qwen2's rotary embedding (pairs (i, i + head_dim / 2)) in place on nheads heads at columns [col0, ...) of token rows
 *     [rows, ld] bf16; cos_sin [T, head_dim / 2, 2] f32, position = row % T.
 *   softmax_rows_causal: row r of materialised scores sc f32 [rows, n] -> bf16 softmax over columns <= r % n, zeros after (n <= 512). */
int advgrpo_rope_half(void* buf, int64_t ld, int rows, int T, int col0, int nheads, int head_dim, const float* cos_sin, void* stream);
int advgrpo_softmax_rows_causal(const float* sc, void* p16, int64_t rows, int n, void* stream);
/* diffusers Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): t f32 [B] -> bf16 [B, dim]. */
int advgrpo_timestep_embedding(const float* t, void* out, int B, int dim, void* stream);
/* y = act(x [+ x2]) on bf16, n % 8 == 0 (act 0 none, 3 SiLU). */
int advgrpo_unary(const void* x, const void* x2, void* y, int64_t n, int act, void* stream);
/* [B,C,H,W] latents <-> 2x2 patch token rows (PatchEmbed conv k2 s2 as im2col; proj_out unpatchify). */
int advgrpo_patchify(const void* x, int x_dtype, void* out, int B, int C, int H, int W, void* stream);
int advgrpo_unpatchify(const void* tokens, void* out, int out_dtype, int B, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------ fused attention
 * o = softmax(q k^T * scale [+ causal mask]) v per (batch, head); bf16 in/out, f32 softmax.
 * Replaces F.scaled_dot_product_attention as reached through the transformer / ViT calls
 * (sd3_pipeline_with_logprob_fast.py:630-637; adv_grpo/rewards.py:397; pickscore_scorer.py:40-44).
 * q/k/v/o are [B, S, H*head_dim]-shaped VIEWS: element (b, s, h, d) at
 * ptr + b*bs + s*ld + h*head_dim + d, so packed QKV GEMM outputs are consumed in place.
 * head_dim: 64, 80 or 128 (128: the Qwen-Image MMDiT; non-causal, 16-byte aligned output rows).  ld{q,k,v} % 8 == 0, ldo % 4 == 0.
 * lse (optional, f32 [B,H,Sq]): base-2 log-sum-exp of the scaled scores, consumed by the backward. */
int advgrpo_attention_fwd(const void* q, const void* k, const void* v, void* o,
                          int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                          int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso,
                          int B, int H, int Sq, int Skv, int head_dim, float scale, int causal,
                          float* lse, void* stream);

/* Diagnostic (HOST call; the one entry point that synchronises the device): the softmax of the pipelined forward kernels
 * (head dim 64 without mask / bias, head dim 128) takes every probability relative to the row maximum of the FIRST key tile
 * and never rescales; a workgroup whose row sums leave (1e-30, 1e30) redoes its queries with a per-tile running maximum.
 * count_host receives the number of workgroups that took that slow path on the current device since the last reset. */
int advgrpo_attention_fallback_count(long long* count_host, int reset);

/* advgrpo_attention_fwd + an additive score bias [H,Sq,Skv] f32 shared over the batch: softmax(q k^T scale + bias) v.
 * T5 self-attention (relative position bias, scale = 1) inside encode_prompt, train_dreambooth_lora_sd3.py:98-144. */
int advgrpo_attention_fwd_bias(const void* q, const void* k, const void* v, void* o, int64_t ldq, int64_t ldk,
                               int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso,
                               int B, int H, int Sq, int Skv, int head_dim, float scale, int causal,
                               const float* bias, void* stream);
/* Backward of the fused attention (autograd of the transformer call inside compute_log_prob,
 * scripts/train_sd3_fast_pickscore.py:233-267, reached from loss.backward() at :1165).  head_dim 64.
 * d_o: gradient w.r.t. o (same view convention, pitch lddo / bsdo); lse from the forward; work: f32 scratch of
 * B * H * ceil(Sq / 32) * 64 elements (filled here: per block of 32 queries, -lse / (scale log2 e) and -rowsum(o * d_o));
 * dq/dk/dv: bf16 views with pitches lddq / bsdq (one packed buffer).  All pointers and pitches 16-byte aligned.
 * head_dim 128 (the Qwen-Image MMDiT, BASELINE config 5): the same call; work holds B * H * Sq f32 (rowsum(o * d_o)); the tile
 * loaders address with 32-bit byte offsets from a per-(b, h) base, so Sq * ldq, Sq * lddo, Skv * ldk and Skv * ldv must each
 * stay below 2^30 elements (checked; the call fails otherwise). */
int advgrpo_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                          const float* lse, float* work, void* dq, void* dk, void* dv,
                          int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq,
                          int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso, int64_t bsdo, int64_t bsdq,
                          int B, int H, int Sq, int Skv, int head_dim, float scale, void* stream);

/* ------------------------------------------------------------------ VAE decoder pieces
 * Replace diffusers AutoencoderKL.decode + VaeImageProcessor.postprocess("pt") as reached at
 * sd3_pipeline_with_logprob_fast.py:667-670.  Activations are NHWC bf16; 3x3 convolutions are implicit
 * GEMMs on the MFMA kernel (f32 accumulate).  DEVIATION: the reference runs the VAE in fp32
 * (train_sd3_fast_pickscore.py:481); tolerance is stated in tests/test_gpu_vae.py and DESIGN.md. */
/* y[b,yo,xo,:] = act(conv3x3(x)[...] + bias) + residual ; x: [B, Hout>>up, Wout>>up, Cin] (nearest x2
 * upsample fused when upsample != 0); w: [Cout, 9*Cin] with k = (ky*3+kx)*Cin + c; Cin % 64 == 0;
 * zero_page: >= 128 bytes of zeros (padding taps are read from it). */
int advgrpo_conv3x3_nhwc(const void* x, const void* w, void* y, int out_dtype, int B, int Hout, int Wout,
                         int Cin, int Cout, int upsample, const void* bias, int act, const void* residual,
                         const void* zero_page, void* stream);
/* GroupNorm(G) over NHWC [B, HW, C] + affine (+ SiLU).  `stats`: scratch of advgrpo_groupnorm_scratch_bytes(B, HW, G) bytes
 * (per-block f64 partial sums, added in a fixed order: the result is bitwise reproducible; then the f32 (mean, 1/std) pairs). */
int64_t advgrpo_groupnorm_scratch_bytes(int B, int HW, int G);
int advgrpo_groupnorm_nhwc(const void* x, void* y, double* stats, const void* weight, const void* bias,
                           int B, int HW, int C, int G, float eps, int silu, void* stream);
/* in-place softmax over rows of a bf16 [rows, n] matrix (n % 8 == 0, n <= 8192). */
int advgrpo_softmax_rows(void* s, int64_t rows, int n, void* stream);
/* z [B,C,H,W] -> NHWC bf16 [B,H,W,Cpad] of z/scaling_factor + shift_factor (zero in the pad channels). */
int advgrpo_latents_to_nhwc(const void* z, int z_dtype, void* out, int B, int C, int H, int W, int Cpad,
                            float scaling_factor, float shift_factor, void* stream);
/* decoder output NHWC (first 3 of ldc channels) -> image [B,3,H,W] f32 = clamp(y/2 + 0.5, 0, 1). */
int advgrpo_image_postprocess(const void* y, int y_dtype, int ldc, float* image, int B, int H, int W,
                              void* stream);

/* ---- split-bf16 ("bf16x3") decode mode: closes the gap to the reference's fp32 VAE (train_sd3_fast_pickscore.py:481)
 * without fp32 matrix math.  An f32 value v is carried as hi = bf16(v), lo = bf16(v - hi); x*w = xh*wh + xh*wl + xl*wh on
 * the bf16 MFMA with f32 accumulation (3x the flops of the bf16 mode, ~2^-16 relative error per product).  Left operands
 * (activations) are laid out along K as [hi | hi | lo] (order 0), right operands (weights) as [hi | lo | hi] (order 1);
 * everything between two matrix products (bias, residual, GroupNorm input, softmax input) stays f32. */
/* f32 [rows, K] (+ bias[K], optional) -> bf16 [rows, 3K];  K % 8 == 0.  order 2 = [hi | unwritten | lo]: the activations of
 * advgrpo_conv3x3_nhwc_x3 with Cout >= 128, whose kernel reads the hi and lo thirds only (4 instead of 6 bytes written
 * per element); every other consumer needs order 0. */
int advgrpo_split_bf16x3(const float* x, const float* bias, void* out, int64_t rows, int K, int order, void* stream);
/* advgrpo_conv3x3_nhwc over split operands: x3 [B,Hin,Win,Cin3=3C], w3 [Cout, 9*Cin3] (each tap [hi|lo|hi]); bias,
 * residual, y: f32 */
int advgrpo_conv3x3_nhwc_x3(const void* x3, const void* w3, float* y, int B, int Hout, int Wout, int Cin3, int Cout,
                            int upsample, const float* bias, int act, const float* residual, const void* zero_page,
                            void* stream);
/* ---- "f16x2": the same fp32-equivalent decode for decoder weights that are EXACT in fp16 -- the released SD3 / SD3.5 VAE is
 * stored in fp16 and only upcast by vae.to(torch.float32) (train_sd3_fast_pickscore.py:481).  A 3x3 convolution with such a
 * weight needs two MFMA products, x_hi w + x_lo w on v_mfma_f32_16x16x32_f16, instead of three: activations travel as the fp16
 * pair hi = f16(s v), lo = f16(s v - hi) (22 significant bits) in the thirds [hi | unwritten | lo] of a [.., 3C] row, weights as
 * ONE fp16 piece [Cout, 9 C] (k = (ky*3+kx)*C + c); s = prescale is a power of two that keeps un-normalised activations inside
 * the fp16 range, alpha = 1 / s multiplies the accumulators back (exact).  Everything between two products stays f32.
 *   groupnorm_nhwc_f16x2 / split_f16x2: the producers (GroupNorm + SiLU of a resnet, the plain split in front of an upsampler);
 *   conv3x3_nhwc_f16x2: Cout >= 128, bias / residual / y f32.
 * GroupNorm statistics without a pass over the activations: gn_partial (optional; (B Hout Wout / 16) * (Cout / 4) * 2 floats;
 * needs 16 | Hout Wout) receives {sum, sum of squares} of every block of 16 consecutive output pixels x 4 consecutive output
 * channels; the GroupNorm that reads y takes them as tile_partial with tile_rows = 16 (ADVGRPO_CONV_F16X2_STAT_ROWS) and skips its
 * statistics kernel.  Deterministic (fixed summation order, no atomics) and independent of the image's position in the batch. */
#define ADVGRPO_CONV_F16X2_STAT_ROWS 16
int advgrpo_groupnorm_nhwc_f16x2(const float* x, void* y3, double* stats, const float* weight, const float* bias, int B,
                                 int HW, int C, int G, float eps, int silu, float prescale, const float* tile_partial,
                                 int tile_rows, void* stream);
int advgrpo_split_f16x2(const float* x, const float* bias, void* out3, int64_t rows, int K, float prescale, void* stream);
int advgrpo_conv3x3_nhwc_f16x2(const void* x2, const void* w16, float* y, int B, int Hout, int Wout, int Cin3, int Cout,
                               int upsample, const float* bias, int act, const float* residual, const void* zero_page,
                               float alpha, float* gn_partial, void* stream);
/* Qwen-Image VAE decode, first step: latents [B,C,H,W] (f32 | bf16) de-normalised per channel (z / inv_std[c] + mean[c]: the
 * pipeline's `latents / latents_std + latents_mean` with latents_std = 1 / std) and sent through the 1x1x1 `post_quant_conv`
 * (P [C,C] row-major, bias [C]; f32) -> NHWC [B,H,W,Cpad] bf16 (x3 = 0) or split rows [B,H,W,3 Cpad] (x3 = 1), channels >= C zero.
 * Replaces the head of AutoencoderKLQwenImage.decode behind the rollout's decode call (see advgrpo_rmsnorm_nhwc). */
int advgrpo_latents_mix_to_nhwc(const void* z, int z_dtype, void* out, int x3, int B, int C, int H, int W, int Cpad,
                                const float* inv_std, const float* mean, const float* P, const float* bias, void* stream);
/* Per-pixel RMS norm over the channel axis of NHWC activations [pixels, C] (f32 or bf16 by x_dtype), f32 gamma[C] (+ SiLU):
 * y = x / max(||x||_2, 1e-12) * mult * gamma.  Replaces diffusers' QwenImageRMS_norm (F.normalize(x, dim=1) * dim**0.5 * gamma)
 * of the Qwen-Image VAE decoder behind the decode call of the rollout (the Qwen-Image twin of
 * adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:667-670; BASELINE config 5, README.md:75); mult = sqrt(real
 * channel count) when C is padded.  out_mode 0: bf16 [pixels, C]; 1: split rows [hi | hi | lo]; 2: [hi | unwritten | lo]. */
int advgrpo_rmsnorm_nhwc(const void* x, int x_dtype, void* y, const float* gamma, int64_t pixels, int C, float mult, int silu,
                         int out_mode, void* stream);
/* The same convolution writing, INSTEAD of y, the fp16-pair rows of pair_prescale * y ([B,Hout,Wout,3 Cout] 16-bit, thirds
 * [hi | unwritten | lo]: advgrpo_split_f16x2 of y, bit for bit) -- the operand of the next f16x2 convolution when nothing else reads y
 * (a resnet's second convolution in front of an upsampler of the decoder, PF:667-670): saves the f32 round trip and the split pass. */
int advgrpo_conv3x3_nhwc_f16x2_pair(const void* x2, const void* w16, void* pair_out, float pair_prescale, int B, int Hout, int Wout,
                                    int Cin3, int Cout, int upsample, const float* bias, int act, const float* residual,
                                    const void* zero_page, float alpha, void* stream);
/* "f16x1" (round 6): the TF32-CLASS form of the two entries above -- same operands, same arguments, ONE fp16 product per f32 product (the
 * activation's hi half only: 11 significant bits, what a TF32 operand keeps; weights exact; f32 accumulation, f32 between kernels).  The
 * reference sets allow_tf32 = True (config/base.py:22-23, train_sd3_fast_pickscore.py:537-538), so its fp32 VAE convolutions round both
 * operands to TF32 on the hardware it was written for; this is that arithmetic class on the fp16 MFMA.  An opt-in decoder mode
 * (AutoencoderKLDecoder(f16_single=True)), priced as a leg of the bench line; the default stays the fp32-equivalent two-product form. */
int advgrpo_conv3x3_nhwc_f16x1(const void* x2, const void* w16, float* y, int B, int Hout, int Wout, int Cin3, int Cout,
                               int upsample, const float* bias, int act, const float* residual, const void* zero_page,
                               float alpha, float* gn_partial, void* stream);
int advgrpo_conv3x3_nhwc_f16x1_pair(const void* x2, const void* w16, void* pair_out, float pair_prescale, int B, int Hout, int Wout,
                                    int Cin3, int Cout, int upsample, const float* bias, int act, const float* residual,
                                    const void* zero_page, float alpha, void* stream);
/* "bf16x2": the same two-product convolution for weights that are EXACT in bf16 (the released Qwen-Image VAE is a bf16
 * checkpoint; BASELINE config 5's decode, see advgrpo_rmsnorm_nhwc): x2 = split rows [hi | unwritten | lo] of bf16 pieces
 * (advgrpo_split_bf16x3 order 2, advgrpo_rmsnorm_nhwc out_mode 2), w16 = ONE bf16 piece [Cout, 9 C]; x_hi w + x_lo w on the bf16
 * MFMA.  Same arguments as advgrpo_conv3x3_nhwc_f16x2. */
int advgrpo_conv3x3_nhwc_bf16x2(const void* x2, const void* w16, float* y, int B, int Hout, int Wout, int Cin3, int Cout,
                                int upsample, const float* bias, int act, const float* residual, const void* zero_page,
                                float alpha, float* gn_partial, void* stream);
/* GroupNorm over f32 NHWC [B,HW,C], f32 affine (+ SiLU) -> split output [B,HW,3C]; stats scratch as above.
 * pair_only != 0 leaves the middle third unwritten (order 2 above: output consumed by the Cout >= 128 3x3 kernel only) */
int advgrpo_groupnorm_nhwc_x3(const float* x, void* y3, double* stats, const float* weight, const float* bias, int B,
                              int HW, int C, int G, float eps, int silu, int pair_only, void* stream);
/* softmax over f32 rows [rows, n] -> split rows [rows, 3n] (left operand of P.V) */
int advgrpo_softmax_rows_x3(const float* s, void* out3, int64_t rows, int n, void* stream);
/* y[r, c] = a[r, c] + b[r, c] (optional) + bias[c] (optional), f32; C % 4 == 0 */
int advgrpo_add_rows_f32(const float* a, const float* b, const float* bias, float* y, int64_t rows, int C, void* stream);
/* ---- bf16x3 ViT towers: the fp32 PickScore scorer of rewards.py:561-574 (PickScoreScorer(dtype=torch.float32), config 4's
 * `pickscore` reward) on the same split-bf16 products.  Row kernels between two matrix products (csrc/x3.hip):
 *   layernorm_x3:            x f32 [M, D] -> LayerNorm(w, b f32; D <= 2048) -> split rows [M, 3D], order 0
 *   split_act_bf16x3:        split(act(x + bias)); act 0 none, 1 GELU (erf), 2 quick_gelu
 *   softmax_rows_x3_masked:  softmax(alpha * s) over the first n_valid keys of every row (causal_period > 0: row r keeps
 *                            min(n_valid, r % causal_period + 1) keys), zeros past them; s [rows, n] f32 -> [rows, 3n] */
int advgrpo_layernorm_x3(const float* x, const float* w, const float* b, void* out3, int M, int D, float eps, void* stream);
int advgrpo_split_act_bf16x3(const float* x, const float* bias, void* out3, int64_t rows, int K, int order, int act,
                             void* stream);
int advgrpo_softmax_rows_x3_masked(const float* s, void* out3, int64_t rows, int n, int n_valid, int causal_period,
                                   float alpha, void* stream);
/* advgrpo_latents_to_nhwc writing the split layout [B,H,W,3*Cpad] */
int advgrpo_latents_to_nhwc_x3(const void* z, int z_dtype, void* out3, int B, int C, int H, int W, int Cpad,
                               float scaling_factor, float shift_factor, void* stream);

/* ------------------------------------------------------------------ reward preprocessing + epilogues
 * CLIP path: (x*255).round().clamp -> uint8 (adv_grpo/rewards.py:567), Pillow 8-bit antialiased bicubic
 * resize H x W -> OH x OW (bit-exact emulation of CLIPProcessor's PIL resize, pickscore_scorer.py:21-27),
 * rescale 1/255, normalise, and im2col for the 14x14 patch embedding: patches bf16 [B*(OH/14)*(OW/14), 640]
 * (column = c*196 + iy*14 + ix; 588..639 zero).  bounds_{h,v}: int [out, 2] = (first tap, taps);
 * coefs_{h,v}: int [out, ksize] 22-bit fixed point (device; built on the host, adv_grpo_amd/preprocess.py).
 * tmp: B*3*H*OW bytes.  mean3_host/std3_host: HOST float[3].  quant_trunc != 0: uint8 by truncation
 * ((x*255).astype(uint8), tensor_to_pil_list of the D-step, train_sd3_fast_dino_patch.py:135-149) instead of round. */
int advgrpo_clip_preprocess_patches(const void* image, int image_dtype, void* patches, uint8_t* tmp,
                                    int B, int H, int W, int OH, int OW,
                                    const int* bounds_h, const int* coefs_h, int ksize_h,
                                    const int* bounds_v, const int* coefs_v, int ksize_v,
                                    const float* mean3_host, const float* std3_host, int quant_trunc,
                                    void* stream);
/* The same with the normalised pixels kept in f32 and written as the split-bf16 left operand [hi | hi | lo]:
 * patches3 [B*P, 3*640] -- the input of the fp32-equivalent scorer towers (below, "bf16x3 ViT towers"). */
int advgrpo_clip_preprocess_patches_x3(const void* image, int image_dtype, void* patches3, uint8_t* tmp, int B,
                                       int H, int W, int OH, int OW, const int* bounds_h, const int* coefs_h,
                                       int ksize_h, const int* bounds_v, const int* coefs_v, int ksize_v,
                                       const float* mean3_host, const float* std3_host, int quant_trunc,
                                       void* stream);
/* DINO path (adv_grpo/rewards.py:379-391): F.interpolate(bicubic, align_corners=False) to OH x OW on
 * bf16-rounded pixels, bf16 round, (x-mean)/std in f32, bf16; same im2col output. */
int advgrpo_dino_preprocess_patches(const void* image, int image_dtype, void* patches, int B, int H, int W,
                                    int OH, int OW, const float* mean3_host, const float* std3_host,
                                    void* stream);
/* The fp32 pipeline of image_similarity_score (rewards.py:147-203): f32 bicubic, no bf16 rounding anywhere, the normalised
 * pixels written as the split-bf16 left operand: patches3 [B*P, 3*640]. */
int advgrpo_dino_preprocess_patches_x3(const void* image, int image_dtype, void* patches3, int B, int H, int W,
                                       int OH, int OW, const float* mean3_host, const float* std3_host, void* stream);
/* rows [B*(1+n), D]: row 0 = feats[b,0], rows 1.. = feats[b, 1+idx[b,j]], each x/(||x||+eps) (rewards.py:400-412). */
int advgrpo_gather_l2norm_rows(const void* feats, const int64_t* idx, void* out, int B, int T, int D, int n,
                               float eps, void* stream);
/* DINO head second Linear + cls/patch mix (rewards.py:414-421): hidden [B*(1+n), Hd] bf16 after Linear+GELU. */
int advgrpo_dino_head_combine(const void* hidden, const void* w2, const void* b2, int B, int Hd, int n,
                              float cls_weight, float* hybrid, float* cls_score, float* patch_scores,
                              void* stream);
/* PickScore (pickscore_scorer.py:40-52): scores[b] = logit_scale_exp * cos(text_b, image_b) / 26. */
int advgrpo_pickscore_pairs(const void* image_embs, const void* text_embs, int B, int P, float logit_scale_exp,
                            float* scores, void* stream);

/* Descriptor form of the Linear (same semantics as advgrpo_gemm_bf16 / _train; unused fields zero) plus
 *   - a fused QK-norm: RMSNorm(64, eps, affine) applied per head to the first rms_nheads 64-wide column groups of
 *     the output (attn.norm_q / norm_k of diffusers' JointAttnProcessor2_0 on a fused to_q|to_k|to_v projection,
 *     called from the transformer at adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:168-186):
 *     y = bf16(bf16(x) * rsqrt(mean(x^2) + eps)) * w[head / heads_per_weight]; rms_rs_out[orow, head] keeps 1/rms;
 *   - a grouped launch: descs[0] and descs[1] (count = 2) are served by ONE kernel launch, the second problem's
 *     tiles filling the tail of the first (the text-stream Linear of a joint block next to its image-stream twin). */
typedef struct advgrpo_gemm_desc {
    const void* A; const void* W; void* C;
    int64_t lda, ldw, ldc;
    int32_t out_dtype, M, N, K;
    const void* bias; int32_t act; float alpha;
    const void* gate; int64_t gate_stride; int32_t gate_rows;
    const void* residual; int64_t ldr;
    int32_t seg_rows; int64_t seg_stride, seg_off;
    int32_t a_seg_rows; int64_t a_seg_stride, a_seg_off;
    void* aux_out; const void* aux_in; int64_t ld_aux;
    const void* rms_weight; int32_t rms_nheads, rms_heads_per_weight; float rms_eps; float* rms_rs_out;
} advgrpo_gemm_desc;
int advgrpo_gemm_grouped(const advgrpo_gemm_desc* descs, int count /* 1 or 2 */, void* stream);

/* ------------------------------------------------------------------ one MMDiT block behind one entry (csrc/mmdit_block.cpp)
 * diffusers JointTransformerBlock of SD3 / SD3.5 ("MMDiT-X": dual = the block has the second, image-only attention; last = the
 * context_pre_only block) as called by the transformer forward at sd3_pipeline_with_logprob_fast.py:630-637 and
 * train_sd3_fast_pickscore.py:235-255: norms + adaLN modulation, fused q|k|v projections with QK RMSNorm, joint attention, gated
 * output projections, [second attention,] gated GELU-tanh feed-forwards -- both streams, ~10 launches on `stream`, x / c updated in
 * place.  bf16 everywhere, head dim 64; weights in nn.Linear layout [N, K] (the q | k | v weights of a projection stacked along N);
 * mods [B, >= ..] bf16 holds the block's modulation rows, chunk j of the image stream at column mod_x + j D (j: shift_msa, scale_msa,
 * gate_msa, shift_mlp, scale_mlp, gate_mlp [, shift_msa2, scale_msa2, gate_msa2]), of the text stream at mod_c + j D (last block:
 * scale, shift).  rms_*: [2, 64] q / k RMSNorm weights, or NULL for no QK norm.  Bit-identical to the same launches issued one by
 * one (adv_grpo_amd/mmdit.py does exactly that when a feature this entry does not cover is on: fp8 Linears, LoRA side columns). */
typedef struct advgrpo_mmdit_block_desc {
    int32_t B, Ni, Nt, D, H, dual, last;
    void* x; void* c;                                            /* [B Ni, D], [B Nt, D] residual streams, updated in place */
    const void* mods; int64_t mod_stride, mod_x, mod_c;          /* elements */
    const void *qkv_w, *qkv_b, *cqkv_w, *cqkv_b, *out_w, *out_b, *cout_w, *cout_b, *qkv2_w, *qkv2_b, *out2_w, *out2_b;
    const void *ff1_w, *ff1_b, *ff2_w, *ff2_b, *cff1_w, *cff1_b, *cff2_w, *cff2_b;
    const void *rms_x, *rms_c, *rms_2;
} advgrpo_mmdit_block_desc;
int64_t advgrpo_mmdit_block_workspace_bytes(int B, int Ni, int Nt, int D, int dual);
int advgrpo_mmdit_block_forward(const advgrpo_mmdit_block_desc* block, void* workspace /* 256-byte aligned */, int64_t workspace_bytes,
                                void* stream);

/* ------------------------------------------------------------------ one MMDiT block's BACKWARD behind one entry (csrc/mmdit_block_bwd.cpp)
 * The data-gradient chain of the same JointTransformerBlock inside loss.backward() (train_sd3_fast_pickscore.py:1165) of the transformer call of
 * compute_log_prob (:233-267): feed-forward data gradients with GELU' in the epilogue, LayerNorm-modulate backward of both norms (writing the
 * gated copies the next data-gradient GEMMs read), output-projection data gradients scattered into the joint rows, attention backward,
 * QK-norm backward, q | k | v data gradients -- both streams, ~14 launches on `stream`.  *_wT: the forward weights TRANSPOSED, in nn.Linear
 * layout ([in, out] -> a [N = in, K = out] matrix); saved activations as advgrpo_mmdit_block_forward's caller kept them: x_in / c_in (block
 * inputs), x_mid / c_mid (streams after the attention residual), pre / cpre (feed-forward pre-activations, [M, 4 D]), qkv [B S, 3 D] (after
 * QK-norm), rs [B S, 2 H] f32 (1 / rms), att [B, S, D] (row pitch ld_att, 0 = D), lse [B, H, S] f32 (base 2), and the second attention's
 * qkv2 / rs2 / att2 / lse2 over the image rows.  Gradients: dx / dc of the block outputs, dyg / dcyg = gate_mlp * dx / c_gate_mlp * dc (written by
 * the pass that produced dx / dc: block i + 1's call, or advgrpo_layernorm_mod_bwd_gated of the final layer); outputs dx_out / dc_out and, unless
 * `first`, their copies gated with block i - 1's feed-forward gates (modulation rows at mod_x_prev / mod_c_prev).  dyo [B Ni, D], dyc [B Nt, D]
 * (= gate_msa * d(stream after attention)) and dqkv [B S, 3 D] are OUTPUTS too: with att and the block's normalised inputs they are the
 * operands of the LoRA adapter gradients (advgrpo_gemm_tn_grouped), which stay with the caller.  last: no text-stream output projection /
 * feed-forward (context_pre_only block; dc, dcyg, dyc, c_mid, cpre unused).  bf16 unless noted, dense rows; bit-identical to the same launches
 * issued one by one (adv_grpo_amd/mmdit_train.py keeps that sequencing for the LoRA side-path mode). */
typedef struct advgrpo_mmdit_block_bwd_desc {
    int32_t B, Ni, Nt, D, H, dual, last, first;
    const void* mods; int64_t mod_stride, mod_x, mod_c, mod_x_prev, mod_c_prev, ld_att;
    const void *ff2_wT, *ff1_wT, *cff2_wT, *cff1_wT, *out_wT, *cout_wT, *qkv_wT, *cqkv_wT, *out2_wT, *qkv2_wT;
    const void *rms_x, *rms_c, *rms_2;
    const void *x_in, *c_in, *x_mid, *c_mid, *pre, *cpre, *qkv, *rs, *att, *lse, *qkv2, *rs2, *att2, *lse2;
    const void *dx, *dc, *dyg, *dcyg;
    void *dx_out, *dc_out, *dyg_prev, *dcyg_prev, *dyo, *dyc, *dqkv;
} advgrpo_mmdit_block_bwd_desc;
int64_t advgrpo_mmdit_block_backward_workspace_bytes(int B, int Ni, int Nt, int D, int H, int dual);
int advgrpo_mmdit_block_backward(const advgrpo_mmdit_block_bwd_desc* block, void* workspace /* 256-byte aligned */, int64_t workspace_bytes,
                                 void* stream);

/* ------------------------------------------------------------------ the VAE decoder behind one entry (csrc/vae_decode.cpp)
 * `pipeline.vae.decode(latents / scaling_factor + shift_factor)` + `image_processor.postprocess(image, "pt")` of the rollout
 * (adv_grpo/diffusers_patch/sd3_pipeline_with_logprob_fast.py:667-670) in the fp32-EQUIVALENT arithmetic (the reference decodes in fp32,
 * train_sd3_fast_pickscore.py:481): diffusers' AutoencoderKL decoder of SD3 / SD3.5 -- conv_in, mid block (resnet, single-head attention,
 * resnet), n_up up blocks of resnets_per_up resnets with a nearest x2 upsample + convolution between them, GroupNorm + SiLU + conv_out --
 * ~190 launches on `stream`.  A 3x3 convolution is described by advgrpo_vae_conv: form 1 = w is ONE fp16 piece [cout, 9 cin] (weights exact
 * in fp16: the two-product "f16x2" kernels, or the one-product "f16x1" ones when f16_single != 0), form 0 = w is the split-bf16 operand
 * [cout, 9 * 3 cin] (advgrpo_split_bf16x3 order 1 per tap: three products); cin is the padded channel count (a multiple of 64); bias f32.
 * A resnet's 1x1 shortcut weight is the split-bf16 [cout, 3 cin] operand and its bias is already added to conv2's.  The attention projections
 * are split-bf16 [C, 3 C] operands with f32 biases; GroupNorm affines f32.  up_resnets / upsamplers are HOST arrays.  latents [B, C, h, w]
 * f32 or bf16 as the rollout holds them (pre-scaling); image [B, 3, 8 h, 8 w] f32 in [0, 1].  Bit-identical to the launches issued one by one
 * (adv_grpo_amd/vae.py keeps that sequencing as the check). */
typedef struct advgrpo_vae_conv {
    const void* w; const float* bias;
    int32_t cin, cout, form, reserved;
} advgrpo_vae_conv;
typedef struct advgrpo_vae_resnet {
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b;
    advgrpo_vae_conv conv1, conv2;
    const void* shortcut_w;
} advgrpo_vae_resnet;
typedef struct advgrpo_vae_decoder_desc {
    int32_t B, h, w, latent_channels, groups, n_up, resnets_per_up, f16_single;
    float scaling_factor, shift_factor;
    advgrpo_vae_conv conv_in, conv_out;
    const float *norm_out_w, *norm_out_b;
    advgrpo_vae_resnet mid[2];
    const float *attn_norm_w, *attn_norm_b;
    const void *attn_q_w, *attn_k_w, *attn_v_w, *attn_o_w;
    const float *attn_q_b, *attn_k_b, *attn_v_b, *attn_o_b;
    const advgrpo_vae_resnet* up_resnets;      /* [n_up * resnets_per_up] */
    const advgrpo_vae_conv* upsamplers;        /* [n_up - 1] */
    const void* zero_page;                     /* >= 256 zero bytes of device memory (the padding taps read it) */
} advgrpo_vae_decoder_desc;
int64_t advgrpo_vae_decode_workspace_bytes(const advgrpo_vae_decoder_desc* decoder);
int advgrpo_vae_decode(const advgrpo_vae_decoder_desc* decoder, const void* latents, int latents_dtype, float* image,
                       void* workspace /* 256-byte aligned */, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ a ViT encoder stack behind one entry (csrc/vit_encoder.cpp)
 * The pre-LN transformer encoder of the reward towers: transformers' CLIPEncoderLayer x n behind CLIPModel.get_image_features /
 * get_text_features (adv_grpo/pickscore_scorer.py:40-44, adv_grpo/pick_score_training.py:95-106; ViT-H/14: 32 layers, 16 heads x 80; text:
 * 24 causal layers, 16 x 64) and timm's vit_base_patch14_dinov2 blocks with LayerScale behind forward_features (adv_grpo/rewards.py:397,
 * scripts/train_sd3_fast_dino_patch.py:183-184; 12 layers, 12 x 64).  Per layer: x += [ls1 *] attn(LN(x)); x += [ls2 *] fc2(act(fc1(LN(x)))),
 * q | k | v stacked along N in qkv_w / qkv_b, weights in nn.Linear layout [N, K], all bf16; x [B S, D] token rows, updated in place;
 * `layers` is a HOST array (device pointers inside).  act: the GEMM activation codes (2 GELU-erf, 4 quick-GELU); head dim D / H must be 64
 * or 80.  The embedding in front and the final norm / projection behind stay with the caller.  Bit-identical to the same launches issued one
 * by one (adv_grpo_amd/vit.py keeps that sequencing as the check). */
typedef struct advgrpo_vit_layer {
    const void *ln1_w, *ln1_b, *qkv_w, *qkv_b, *out_w, *out_b, *ls1;      /* ls1 / ls2: LayerScale gamma [D] or NULL */
    const void *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *ls2;
} advgrpo_vit_layer;
typedef struct advgrpo_vit_desc {
    int32_t B, S, D, H, mlp, n_layers, act, causal;
    float eps;
    void* x;
    const advgrpo_vit_layer* layers;
} advgrpo_vit_desc;
int64_t advgrpo_vit_workspace_bytes(int B, int S, int D, int mlp);
int advgrpo_vit_forward(const advgrpo_vit_desc* tower, void* workspace /* 256-byte aligned */, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ fp8 Linears (BASELINE config 5: "fp8 MFMA path")
 * The reference has no fp8 code (config 5 changes pretrained.model / resolution only); the scheme is this
 * library's: OCP e4m3 codes, one f32 scale per token row of the activation and per output channel of the weight,
 *     x[r,k] ~= scale[r] * q[r,k],   scale[r] = max(max_k |x[r,k]| / 448, 2^-126)  (1 for an all-zero row),  q = RNE(clamp(x / scale)).
 * advgrpo_quant_fp8_rows: x [M, K] bf16 (pitch ldx) -> q [M, K] one byte per element (pitch ldq) + scale [M].
 * split_period > 0: rows of a joint [B, period, K] buffer are compacted, the first split_first rows of every period (image
 * tokens) to output rows b * split_first + s, the others behind ALL of them (B * split_first + b * (period - split_first) + ..).
 * advgrpo_gemm_fp8_grouped: descriptors as for advgrpo_gemm_grouped with A / W pointing at fp8 codes (lda / ldw in bytes),
 *     C = epilogue( a_scale[m] * w_scale[n] * sum_k A[m,k] W[n,k] ),
 * f32 accumulation on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), bf16 output, K % 128 == 0, no A row map, and one
 * of the epilogues bias | bias + QK-norm | bias + GELU-tanh (aux_out: keeps the pre-activation) | bias + gate + residual. */
typedef struct advgrpo_fp8_scales { const float* a_scale; const float* w_scale; } advgrpo_fp8_scales;
int advgrpo_quant_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int M, int K,
                           int split_first, int split_period, void* stream);
int advgrpo_gemm_fp8_grouped(const advgrpo_gemm_desc* descs, const advgrpo_fp8_scales* scales, int count /* 1 or 2 */, void* stream);

/* ------------------------------------------------------------------ G-step (training) kernels
 * The update half of the path: loss.backward() / clip_grad_norm_ / AdamW / EMA at
 * scripts/train_sd3_fast_pickscore.py:1165-1171,1186-1187 and adv_grpo/ema.py:39-52; the backward of the
 * transformer call inside compute_log_prob (TP:233-267) is assembled from these + advgrpo_attention_bwd. */
/* GEMM with training extras: aux_out[orow,n] = pre-activation (bf16, pitch ld_aux); act 5/6 = multiply by
 * GELU-tanh' / GELU-erf' evaluated at aux_in[orow,n]; splitk > 1: contraction split over workgroups, partial
 * tiles atomically added into an f32 C (LoRA weight gradients, whose contraction runs over all tokens). */
int advgrpo_gemm_bf16_train(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                            int out_dtype, int M, int N, int K, const void* bias, int act, float alpha,
                            const void* gate, int64_t gate_stride, int gate_rows, const void* residual,
                            int64_t ldr, void* aux_out, const void* aux_in, int64_t ld_aux, int splitk,
                            void* stream);
/* Token-contracted GEMM for the LoRA weight gradients (PEFT LoRA layers set up at TP:490-511, differentiated by
 * loss.backward() at TP:1165): C[n1,n2] += alpha * sum_m P[row_p(m), n1] * Q[row_q(m), n2], f32 atomic accumulation;
 * P [M, N1] and Q [M, 64] bf16, token-major, rows through the (seg_rows, seg_stride, seg_off) map; transpose_out
 * writes C[n2 * ldc + n1]; slices of the token range are summed through `workspace` (no atomics).  dB = s dY^T (X A^T): P = dY, Q = X A^T; dA = s (dY B)^T X: P = X, Q = dY B, transposed. */
int advgrpo_gemm_tn_f32acc(const void* P, int64_t ldp, int p_seg_rows, int64_t p_seg_stride, int64_t p_seg_off,
                           const void* Q, int64_t ldq, int q_seg_rows, int64_t q_seg_stride, int64_t q_seg_off,
                           float* C, int64_t ldc, int transpose_out, int M, int N1, int N2, float alpha,
                           void* workspace /* advgrpo_gemm_tn_workspace_bytes(M, N1), 16-byte aligned */, void* stream);
int64_t advgrpo_gemm_tn_workspace_bytes(int M, int N1);
/* The same products for a whole adapter group in ONE launch (round 5): up to 12 problems, each
 *   C[n1, n2] += alpha * sum_m P[row_p(m), n1] * Q[row_q(m), n2]     (Q 64 wide; transpose_out: C[n2 * ldc + n1])
 * -- the dB of every adapter of a group (P = dY_j, Q = X A_j^T) and the dA of every adapter (P = X, Q = dY_j B_j, transposed) together.
 * Token slices meet in `workspace` and the last workgroup to reach a tile adds them in slice order (reproducible; no float atomics,
 * no reduce launch).  workspace: 256-byte aligned, advgrpo_gemm_tn_grouped_workspace_bytes(descs, n) bytes, its first 4096 bytes
 * (arrival counters) zero at every launch -- a workspace only ever used by this function stays that way (workspace_is_zeroed = 1);
 * 0 = memset first.  Reference site: the LoRA layers' weight gradients inside loss.backward() (TP:490-511, TP:1165). */
typedef struct advgrpo_tn_desc {
    const void* P; int64_t ldp; int32_t p_seg_rows; int64_t p_seg_stride, p_seg_off;      /* wide operand [M, N1] bf16, token-major */
    const void* Q; int64_t ldq; int32_t q_seg_rows; int64_t q_seg_stride, q_seg_off;      /* [M, 64] bf16 */
    float* C; int64_t ldc; int32_t transpose_out;
    int32_t M, N1;
    float alpha;
} advgrpo_tn_desc;
int64_t advgrpo_gemm_tn_grouped_workspace_bytes(const advgrpo_tn_desc* descs, int n);
int advgrpo_gemm_tn_grouped(const advgrpo_tn_desc* descs, int n, void* workspace, int64_t workspace_bytes, int workspace_is_zeroed,
                            void* stream);
/* ---- LoRA re-merge after an optimizer step (TP:1166-1171; the adapters are PEFT's on the attention projections, TP:490-511) ----
   One adapter of one projection: W_eff = base + alpha * B A and its transpose, plus the adapter's slices of the two operands the
   adapter-gradient GEMMs read (the group's stacked A and block-diagonal B^T).  All bf16, N and K multiples of 64, the rank padded to 64
   with zero rows / columns. */
typedef struct advgrpo_lora_merge_item {
    const void* A;        /* [64, K], row pitch K */
    const void* B;        /* [N, 64], row pitch 64 */
    const void* base;     /* [N, K] rows of the frozen weight, pitch ld_base */
    void* w;              /* out [N, K], pitch ld_w; NULL: not written (lora_mode = "side" keeps the base weight in the forward) */
    void* wT;             /* out [K, N] (this adapter's N columns of its group's transposed weight), pitch ld_wT */
    void* a_cat;          /* out [64, K], pitch K: copy of A (this adapter's rows of the group's stacked A); NULL: skipped */
    void* b_bd;           /* out [64, N], pitch ld_bd: B^T (this adapter's diagonal block of the group's block-diagonal B^T); NULL: skipped */
    int64_t ld_base, ld_w, ld_wT, ld_bd;
    int32_t N, K;
} advgrpo_lora_merge_item;
/* items_device: n_items descriptors in DEVICE memory; max_N / max_K: the largest N / K among them; rank_padded must be 64.
   acc = sum_r B[n, r] A[r, k] in f32 (fused multiply-adds, r ascending), out = bf16(acc * alpha + base): w and wT hold the same bits. */
int advgrpo_lora_merge(const advgrpo_lora_merge_item* items_device, int n_items, int max_N, int max_K, int rank_padded, float alpha,
                       void* stream);
/* out[c, r] = in[row(r), c] for r < R, zero for R <= r < Rpad (row(r) = the GEMM row-segment map when seg_rows > 0). */
int advgrpo_transpose_bf16(const void* in, void* out, int R, int C, int64_t ldi, int64_t ldo, int Rpad,
                           int seg_rows, int64_t seg_stride, int64_t seg_off, void* stream);
/* dx = dres + d/dx [ LN(x)*(1+scale0)+shift0 -> dy0  (+ LN(x)*(1+scale1)+shift1 -> dy1) ]; all bf16. */
int advgrpo_layernorm_mod_bwd(const void* x, int64_t ldx, const void* dy0, const void* dy1, int64_t lddy,
                              const void* scale0, const void* scale1, int64_t mod_stride, int rows_per_batch,
                              const void* dres, void* dx, int64_t lddx, int M, int D, float eps, void* stream);
/* The same, plus up to two gated copies of the result for the data-gradient GEMMs of the gated projections that read dx next
   (autograd of `hidden_states + gate.unsqueeze(1) * branch`, diffusers' JointTransformerBlock behind TP:235-255 / loss.backward() TP:1165):
   gout_k[m, :] = gate_k[m / rows_per_batch, :] * dx[m, :] with dx as stored (bf16), dense rows of D; gate rows gate_stride elements apart.
   gate_b / gout_b may both be NULL.  Same bits as advgrpo_layernorm_mod_bwd followed by advgrpo_gate_mul. */
int advgrpo_layernorm_mod_bwd_gated(const void* x, int64_t ldx, const void* dy0, const void* dy1, int64_t lddy,
                                    const void* scale0, const void* scale1, int64_t mod_stride, int rows_per_batch,
                                    const void* dres, void* dx, int64_t lddx, int M, int D, float eps,
                                    const void* gate_a, void* gout_a, const void* gate_b, void* gout_b,
                                    int64_t gate_stride, void* stream);
/* in place: dy (grad of the normalised+weighted q|k heads) -> grad of the un-normalised heads. */
int advgrpo_rmsnorm_heads_bwd(void* dy, int64_t lddy, const void* y, int64_t ldy, const float* rs, int M,
                              int col0, int nheads, const void* weight, int heads_per_weight, int seg_rows,
                              int64_t seg_stride, int64_t seg_off, void* stream);
/* backward of advgrpo_qk_norm_rope, in place: dy (grad of the rotated + normalised + weighted q | k heads of the joint buffer) ->
 * grad of the raw projections; y = the forward's saved output, rs = its rs_out; row / weight / rope conventions as the forward.
 * Autograd of norm_q / norm_k / norm_added_* + apply_rotary_emb_qwen in the G-step of BASELINE config 5 (TP:1165 on the
 * Qwen-Image model the reference names at README.md:75). */
int advgrpo_qk_norm_rope_bwd(void* dy, int64_t lddy, const void* y, int64_t ldy, const float* rs, int rows, int S, int n_first,
                             int col0, int nheads, int head_dim, const void* w_first, const void* w_rest, int heads_per_weight,
                             const float* rope, void* stream);
int advgrpo_gate_mul(const void* x, const void* gate, void* y, int M, int D, int rows_per_batch,
                     int64_t gate_stride, void* stream);
/* workspace: advgrpo_sumsq_workspace_bytes() bytes of device memory (per-block partial sums, added in a fixed order: the result
 * is bitwise repeatable) */
int64_t advgrpo_sumsq_workspace_bytes(void);
int advgrpo_sumsq_f32(const float* g, int64_t n, float* out /* += */, float* workspace, void* stream);
/* torch.optim.AdamW step on a flat f32 vector (+ bf16 copy), with clip_grad_norm_ folded in
 * (grad_sumsq = device scalar sum of squares of the UNSCALED grads; grad_scale multiplies every grad, e.g.
 * 1/accumulation steps); grads are zeroed. */
int advgrpo_adamw_step(float* param_f32, void* param_bf16, float* grad, float* exp_avg, float* exp_avg_sq,
                       int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                       const float* grad_sumsq, float max_grad_norm, float grad_scale, void* stream);
int advgrpo_ema_step(float* ema, const float* param, int64_t n, float one_minus_decay, void* stream);

/* ------------------------------------------------------------------ D-step, DINO variant
 * train_dino (scripts/train_sd3_fast_dino_patch.py:156-232) on frozen-backbone features: hinge loss on the CLS logit
 * + 0.3 x hinge on n sampled patch logits; head = Linear(D,Hd) -> GELU -> Linear(Hd,1) (TD:592-603).
 * Rows are [cls, n patches] per image, B_real real images first.  The two Linears run on advgrpo_gemm_bf16_train
 * (pre-activation kept), these kernels do the rest; weight gradients of the first Linear are a split-K GEMM. */
int advgrpo_gather_rows(const void* feats, const int64_t* idx, void* out, int B, int T, int D, int n, void* stream);
/* logits[r] = hidden[r,:].w2 + b2; dlogits = d loss / d logit; stats4 = {loss, #real cls correct, #fake cls correct,
 * sum(dlogits) = grad b2}. */
int advgrpo_dino_head_loss(const void* hidden, const void* w2, const void* b2, int R, int Hd, int n, int B_real,
                           int B_total, float patch_loss_weight, float* logits, float* dlogits, float* stats4,
                           void* stream);
/* dpre = dlogits (x) w2 * GELU'(pre) (bf16); grad_w2[c] += sum_r dlogits[r] hidden[r,c]; grad_b1[c] += sum_r dpre[r,c]. */
int advgrpo_dino_head_dpre(const void* pre, const void* hidden, const void* w2, const float* dlogits, void* dpre,
                           float* grad_w2, float* grad_b1, int R, int Hd, void* stream);

/* ------------------------------------------------------------------ D-step, PickScore (CLIP) variant
 * train_pickscore (scripts/train_sd3_fast_pickscore.py:151-183) + CLIPCriterion (adv_grpo/pick_score_training.py:
 * 89-224), trainable = vision_model.encoder.layers[-1] (TP:1016-1020 with tune_layer = -1).  CLIP pools the CLS
 * token, so the last layer's attention is differentiated for ONE query per image. */
/* qkv [Bt,S,3*H*hd] bf16 -> o_cls [Bt,H*hd] bf16 (attention output of token 0), probs [Bt,H,S] f32. */
int advgrpo_cls_attention_fwd(const void* qkv, void* o_cls, float* probs, int Bt, int S, int H, int head_dim,
                              float scale, void* stream);
/* d o_cls -> dqkv [Bt,S,3*H*hd] bf16 (fully written; dq non-zero on token 0 only). */
int advgrpo_cls_attention_bwd(const void* qkv, const float* probs, const void* do_cls, void* dqkv, int Bt, int S,
                              int H, int head_dim, float scale, void* stream);
/* loss = mean_i softplus(s (cos(t_i, e_{B+i}) - cos(t_i, e_i))) over pairs (real e_i, fake e_{B+i}) and its gradient
 * w.r.t. the un-normalised image embeddings e [2B,P]; == CLIPCriterion.calc_loss with label_0 = 1, label_1 = 0. */
int advgrpo_clip_pair_loss(const void* image_embs, const void* text_embs, int B, int P, float logit_scale_exp,
                           float* loss, void* d_image_embs, void* stream);
/* The same criterion with the batch's labels (CLIPCriterion.forward, adv_grpo/pick_score_training.py:205-224 -> calc_loss :118-199,
 * in_batch_negatives = False): label_0 / label_1 device f32 [B] (a 0-dim label of the reference broadcast by the caller), or both NULL
 * for (1, 0).  loss = mean_i label_0 softplus(z_i) + label_1 softplus(-z_i) + [label_0 == label_1] log(0.5), z = s (cos(t, e1) - cos(t, e0)). */
int advgrpo_clip_pair_loss_labels(const void* image_embs, const void* text_embs, int B, int P, float logit_scale_exp,
                                  const float* label_0, const float* label_1, float* loss, void* d_image_embs, void* stream);
/* tune_layer = -k, k > 1 (TP:1016-1020): softmax forward + backward over materialised attention-score rows of the trainable CLIP layers
 * (autograd of F.scaled_dot_product_attention inside CLIPModel.vision_model, reached from train_pickscore TP:151-183): sc / dp f32 [rows, n]
 * (scaled scores, dO V^T) -> p16 = softmax over the first n_valid columns, ds16 = scale * P (dP - sum P dP), bf16; padding rows (query
 * index r % n >= n_valid) and columns come out zero. */
int advgrpo_softmax_bwd_rows(const float* sc, const float* dp, void* p16, void* ds16, int64_t rows, int n, int n_valid, float scale,
                             void* stream);
int advgrpo_colsum_bf16(const void* x, int64_t ld, int R, int C, float* out /* += */, void* stream);
int advgrpo_ln_affine_grads(const void* x, int64_t ldx, const void* dy, int64_t lddy, int M, int D, float eps,
                            float* grad_w /* += */, float* grad_b /* += */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVGRPO_H */
