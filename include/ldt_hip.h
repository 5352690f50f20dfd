/*
 * ldt_hip.h — C-ABI of libldt_hip.so: the MI355X (gfx950) kernels behind LDT's sampling hot path.
 *
 * The upstream reference (Negai-98/LDT) has NO native/FFI seam on this path — everything is torch.nn
 * (SURVEY.md §8b).  The seam a drop-in keeps is the Python class surface (ldt_amd/: Score, Compressor,
 * DiffusionVPSDE, Trainer); this library sits underneath it and each entry point below names the
 * reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + sizes + a hipStream_t passed as void* (0 = default stream);
 *     no torch types, no exceptions, no ownership transfer, no allocation of caller-visible memory.
 *   - every function returns int: 0 ok; <0 argument/shape/alignment error (LDT_E*); >0 a hipError_t.
 *     ldt_last_error() returns a thread-local message for the last non-zero status.
 *   - bf16 buffers are uint16_t* (raw bfloat16 bits); matrices are row-major with explicit leading dims
 *     in ELEMENTS; "token-major": activations are [rows = batch*tokens][channels].
 *   - step-dependent operands (AdaLN tables, sampler coefficients, injected noise) are addressed as
 *     base + (*step_ptr) * step_stride with step_ptr a DEVICE int, so one captured HIP graph of a single
 *     reverse-SDE step can be replayed for every step (NULL step_ptr = step 0 / host step where given).
 */
#ifndef LDT_HIP_H
#define LDT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LDT_ABI_VERSION 22
#define LDT_OK 0
#define LDT_EARG (-1)    /* null / inconsistent argument */
#define LDT_ESHAPE (-2)  /* unsupported shape */
#define LDT_EALIGN (-3)  /* pointer / leading-dim alignment */
#define LDT_MAX_BLOCKS 64

int ldt_abi_version(void);
const char* ldt_last_error(void);

/* ---- packing ------------------------------------------------------------------------------------
 * fp32 [rows][cols] (ld_src) -> bf16 [rows][cols_pad] (ld_dst), zero padded to cols_pad (multiple of 4).
 * Used to pack Conv1d/Linear weights (out,in,1) once per weight version (after
 * EMA.swap_parameters_with_ema, tools/utils.py:80-101) and to feed fp32 latents to the MFMA GEMM. */
int ldt_cast_pad_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst,
                      int64_t rows, int32_t cols, int32_t cols_pad, void* stream);

/* ---- token-linear layers: bf16 MFMA GEMM with fused epilogue --------------------------------------
 * out[M,N] = epi( X[M,K] · W[N,K]^T + bias[N] ).  Replaces every 1x1 nn.Conv1d / nn.Linear on the path:
 * model/layers.py:159-161 (fc_q, fc_kv, fc_o), :121-124 (MLP fc/out), model/scorenet/score.py:110 (ln_in),
 * model/layers.py:239 (FinalLayer.ln) and the Compressor's twins.  K % 64 == 0 (pad), rows 16-B aligned. */
enum ldt_epilogue {
    LDT_EPI_F32 = 0,        /* out fp32 = acc + bias */
    LDT_EPI_BF16 = 1,       /* out bf16 = acc + bias */
    LDT_EPI_GELU_BF16 = 2,  /* out bf16 = gelu_erf(acc + bias)            (layers.py:127-129) */
    LDT_EPI_RELU_BF16 = 3,  /* out bf16 = relu(acc + bias [+ skip bf16])  (Compressor/layers.py:115-160) */
    LDT_EPI_RESID_F32 = 4   /* out fp32 = resid + gate[s,:] * (acc + bias) (layers.py:218-219; gate NULL = 1) */
};
int ldt_gemm_bf16(int32_t epilogue, const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw,
                  const float* bias, void* out, int64_t ldo,
                  const float* resid, int64_t ldr, const uint16_t* skip, int64_t ldskip,
                  const float* gate, int64_t gate_sample_stride, int32_t rows_per_sample,
                  const int32_t* step_ptr, int64_t gate_step_stride,
                  int32_t M, int32_t N, int32_t K, void* stream);

/* ---- LayerNorm folded into the neighbouring GEMMs (batch-shared AdaLN modulation) ------------------------------
 * The reference's  x = x + gate * fc(...) ; h = LN(x) * (1 + scale) + shift ; y = W h + b  (model/layers.py:218-219 with
 * :136-137 and tools/utils.py:127-133) without a LayerNorm pass over x.  With mu, r = mean and rstd of a row of x:
 *     y = r * (W xs) - r * mu * S + C,   xs = x (1 + scale),  S[n] = sum_k (1 + scale[k]) W[n][k],  C[n] = sum_k shift[k] W[n][k] + b[n].
 * ldt_gemm_resid_lnstats (producer):  out = out + gate * (X W^T + bias) in place (LDT_EPI_RESID_F32), plus
 *     xs[M][N] = bf16(out * (1 + ln_scale[n])) and stats_out[N/256][M][2] = per-row (sum, sum of squares) of out over
 *     each 256-column tile (deterministic: no atomics).
 * ldt_gemm_lnfold (consumer):  out bf16 = epi(r * (Xs W^T) - r * mu * fold_S + fold_C), epilogue LDT_EPI_BF16 or
 *     LDT_EPI_GELU_BF16; the row statistics are those of the K = stats_parts*256 input channels (K <= 1024).
 * ln_scale / fold_S / fold_C / gate are addressed base + (*step_ptr) * their step stride (step_ptr NULL = 0).
 * stats_parts selects the statistics granule and with it the kernel family: N/256 (producer) | K/256 (consumer) partials per row — the
 *     256-tile kernels: M, N multiples of 256, K >= 256, K <= 1024 for the consumer; N/32 | K/32 — the small-batch kernels (batches
 *     of 1-2 k rows, where every GEMM of a Score block runs 64-wide tiles): M a multiple of 128, N of 64.  A producer and the consumer
 *     of its statistics must use the same granule.  The host builds S and C in fp32 from the SAME bf16 W the GEMM reads. */
int ldt_gemm_resid_lnstats(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias,
                           float* out, int64_t ldo, const float* gate, int64_t gate_sample_stride,
                           int32_t rows_per_sample, const float* ln_scale, uint16_t* xs, int64_t ldxs,
                           float* stats_out, const int32_t* step_ptr, int64_t gate_step_stride,
                           int64_t ln_step_stride, int32_t M, int32_t N, int32_t K, int32_t stats_parts, void* stream);
int ldt_gemm_lnfold(int32_t epilogue, const uint16_t* Xs, int64_t ldx, const uint16_t* W, int64_t ldw,
                    const float* stats_in, const float* fold_S, const float* fold_C, uint16_t* out, int64_t ldo,
                    const int32_t* step_ptr, int64_t fold_step_stride, int32_t M, int32_t N, int32_t K, int32_t stats_parts,
                    void* stream);

/* ---- LayerNorm(eps 1e-6) [+affine] [+AdaLN modulate] -> bf16 -----------------------------------------
 * y = LN(x)[*w+b] * (1 + scale[s]) + shift[s].  tools/utils.py:127-133 + model/layers.py:136-137,218-219.
 * shift/scale: fp32 [samples or 1][...] addressed base + step*mod_step_stride + sample*mod_sample_stride. */
int ldt_layernorm_modulate(const float* x, int64_t ldx, uint16_t* y, int64_t ldy,
                           const float* w, const float* b, const float* shift, const float* scale,
                           int64_t mod_sample_stride, int32_t rows_per_sample,
                           const int32_t* step_ptr, int64_t mod_step_stride,
                           int64_t M, int32_t C, void* stream);

/* ---- fused multi-head attention ---------------------------------------------------------------------
 * O[b,h,n,:] = softmax(Q K^T / sqrt(Dh)) V, heads at channel offset h*Dh of each row; output is the
 * contiguous [B][H][Nq][Dh] buffer the reference reinterprets as (B,N,C) (model/layers.py:190-197, Q1).
 * head_dim 32 or 64.  K and V share kv_batch_stride. */
int ldt_attention_fwd(const uint16_t* Q, int64_t ldq, int64_t q_batch_stride,
                      const uint16_t* K, int64_t ldk, const uint16_t* V, int64_t ldv, int64_t kv_batch_stride,
                      uint16_t* O, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t head_dim, void* stream);
/* Which kernel ldt_attention_fwd runs for a problem of this shape (a query, not a launch; the return value is the route, not a status):
 * 0 = streaming (attn_fwd_kernel<head_dim>), 1 = resident (attn_fwd_resident_kernel<head_dim>), 2 = whole-head
 * (attn_fwd_head_kernel<64, ceil(Nk / 64)>).  bench.py uses it to name the rocprofv3 symbol of the kernel it timed. */
int ldt_attention_route(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t head_dim);
/* Attention + output projection + gated residual in one kernel, for narrow blocks (the Compressor: Dh = 32,
 * C = H*Dh in {64, 128}; model/layers.py:183-200 then :218 / :225):
 *     X[b] += gate[b] * (Wo . O'[b] + bo),   O' = softmax(Q K^T / sqrt(Dh)) V written as [H][Nq][Dh] and re-read as
 * (Nq, C) rows without permuting the heads back (quirk Q1) — so the rows a workgroup produces for head h and query
 * block q0 are rows h*Nq/H + q0/H ... of X.  Needs Nq % H == 0.  X fp32 [B*Nq][ldx] updated in place (X must not alias
 * Q/K/V); Wo bf16 [C][C] dense; gate fp32 per-sample vectors (nullable = 1). */
int ldt_attention_oproj_resid(const uint16_t* Q, int64_t ldq, int64_t q_batch_stride, const uint16_t* K, int64_t ldk,
                              const uint16_t* V, int64_t ldv, int64_t kv_batch_stride, int32_t B, int32_t H,
                              int32_t Nq, int32_t Nk, int32_t head_dim, const uint16_t* Wo, const float* bo, float* X,
                              int64_t ldx, const float* gate, int64_t gate_sample_stride, void* stream);

/* ---- fp32 linear for the small precision-critical layers ---------------------------------------------
 * C[M,N] = act_out( act_in(A[M,K]) · Bw[N,K]^T + bias ), fp32 FMA accumulate; out fp32 or bf16.
 * TimeEmbedding.mlp (model/layers.py:17), adaLN Linear (:172,:238), K<=64 convs, MiniPointnet, prior heads. */
enum ldt_act { LDT_ACT_NONE = 0, LDT_ACT_SILU = 1, LDT_ACT_RELU = 2, LDT_ACT_GELU = 3 };
int ldt_sgemm(const float* A, int64_t lda, const float* Bw, int64_t ldb, const float* bias,
              void* C, int64_t ldc, int32_t out_bf16, int32_t act_in, int32_t act_out,
              int32_t M, int32_t N, int32_t K, void* stream);

/* sinusoidal embedding of continuous t (model/layers.py:20-36): e[n][2*half] = [sin(t f) | cos(t f)];
 * freq[half] is built by the host with the reference's fp32 expression (quirk Q5). */
int ldt_sinusoid(const float* t, const float* freq, float* e, int32_t n, int32_t half, void* stream);

/* ---- reverse-SDE predictor update (diffusion/diffusion_continuous.py:141-191) ---------------------------
 * mode 0 = ancestral in the reference's op order (coef[step] = {beta, std, sqrt(1-beta), sqrt(beta)});
 * mode 1 = folded x_mean = A x + B params, x = x_mean + C z (coef[step] = {A,B,C,0}).
 * z = noise[step*noise_step_stride + i] if noise != NULL (parity mode: injected CPU draws), else Philox4x32-10
 * keyed by (seed, stream id = step*philox_mul + philox_add, elem_offset + i) — independent of how the batch is
 * sharded across GPUs; predictor-only loops use (1, 0), predictor+corrector loops number every draw.
 * step = *step_ptr if step_ptr else step_host.  x_out may alias x; x_mean_out may be NULL. */
int ldt_sampler_step(const float* x, const float* params, const float* noise, int64_t noise_step_stride,
                     float* x_out, float* x_mean_out, const float* coef,
                     const int32_t* step_ptr, int32_t step_host, int32_t mode,
                     int64_t n, int64_t elem_offset, uint64_t seed,
                     int32_t philox_mul, int32_t philox_add, void* stream);
int ldt_philox_normal(float* out, int64_t n, int64_t elem_offset, int32_t step, uint64_t seed, void* stream);

/* ---- LangevinCorrector (diffusion/diffusion_continuous.py:193-210) --------------------------------------
 * ldt_batch_norm_sum: *sum_out = sum_b ||x[b,:]||_2 over B samples of per_sample fp32 values each — the numerator of
 *   torch.norm(v.reshape(B,-1), dim=-1).mean() (:204-205); norms_scratch[B] receives the per-sample norms.
 *   Fixed reduction order, no atomics.  When the batch is sharded the caller all-reduces the two sums.
 * ldt_langevin_coef: sums = {sum_b ||params_b||, sum_b ||z_b||} over n_total samples -> coef_out[4] = {1, -step/std,
 *   sqrt(2 step), 0}, step = (snr * noise_norm / grad_norm)^2 * 2 with grad = -params/std (:206, alpha = 1), the row
 *   ldt_sampler_step(mode 1) applies: x_mean = x + step*grad, x = x_mean + sqrt(2 step) z (:207-208). */
int ldt_batch_norm_sum(const float* x, int32_t B, int64_t per_sample, float* norms_scratch, float* sum_out, void* stream);
int ldt_langevin_coef(const float* sums, int32_t n_total, float snr, float std_t, float* coef_out, void* stream);

/* ---- PNDM (diffusion/diffusion_continuous.py:260-316) -----------------------------------------------------
 * ldt_pndm_transfer: out = x + d * (p*x - q*et), the transfer() of :263-274 with the batch-uniform schedule scalars
 *   d = at_next - at, p = 1/(sqrt(at)(sqrt(at)+sqrt(at_next))), q = 1/(sqrt(at)(sqrt((1-at_next)at)+sqrt((1-at)at_next)))
 *   formed by the host in fp32; reference op order, no FMA contraction.  out may alias x.
 * ldt_lincomb4: out = s * (((c0 a0 + c1 a1) + c2 a2) + c3 a3): the Runge-Kutta average of :291 and the 4-step
 *   multistep combination of :300. */
int ldt_pndm_transfer(const float* x, const float* et, float d, float p, float q, float* out, int64_t n, void* stream);
int ldt_lincomb4(const float* a0, const float* a1, const float* a2, const float* a3, float c0, float c1, float c2, float c3,
                 float s, float* out, int64_t n, void* stream);

/* ---- small fp32 element-wise helpers of the host orchestration ---------------------------------------------
 * ldt_vpsde_score: out[b,:] = -params[b,:] / sqrt(var(t[b])), var(t) = 1 - (1 - sigma2_0) exp(-beta0 t - (beta1-beta0) t^2 / 2)
 *   in fp32 — the `score` half of Trainer.score_fn's return value (trainer/Latent_SDE_Trainer.py:57-61 with
 *   DiffusionVPSDE.var, diffusion/diffusion_continuous.py:649-651).
 * ldt_add_f32: out = a + b (c = t_emb + label / image-condition embedding, model/scorenet/score.py:135).
 * ldt_widen_bf16: fp32 copy of a packed bf16 panel (exact). */
int ldt_vpsde_score(const float* params, const float* t, float beta0, float beta1, float sigma2_0, float* out, int32_t B,
                    int64_t per_sample, void* stream);
/* ldt_sde_score: the same for the other SDE families make_diffusion builds (diffusion/diffusion_continuous.py:18-29):
 *   kind 0 vpsde (c0..c2 = beta0, beta1, sigma2_0), 1 sub_vpsde (:705-707; same constants), 2 vesde / geometric_sde
 *   (:746-747 / :615-616; c0..c2 = sigma2_min, sigma2_max / sigma2_min, sigma2_0). */
int ldt_sde_score(const float* params, const float* t, int32_t kind, float c0, float c1, float c2, float* out, int32_t B,
                  int64_t per_sample, void* stream);
int ldt_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
/* ldt_block_activation: x = act(x) in place on bf16 rows [M][C] (row stride ld, elements) — the block activation of the reference's
 *   no-condition ResidualBlock branches (`decoder_act`: model/layers.py:224-226, `self.act` from tools/utils.py:104-124), applied to the
 *   LayerNorm output that fc_q / the MLP read.  rrelu in its eval-mode form. */
enum ldt_block_act { LDT_BACT_NONE = 0, LDT_BACT_GELU = 1, LDT_BACT_SILU = 2, LDT_BACT_RELU = 3, LDT_BACT_LEAKY_001 = 4, LDT_BACT_LEAKY_02 = 5,
                     LDT_BACT_RRELU_EVAL = 6, LDT_BACT_HARDSWISH = 7, LDT_BACT_SELU = 8 };
int ldt_block_activation(uint16_t* x, int64_t ld, int64_t M, int32_t C, int32_t kind, void* stream);
int ldt_widen_bf16(const uint16_t* src, float* dst, int64_t n, void* stream);

/* ---- the other norms `get_norm` builds (tools/utils.py:168-181; ResidualBlock.norm1 / norm2, FinalLayer.norm: model/layers.py:163-164,235) ----
 * `norm: group_norm` = nn.GroupNorm(min(C / 4, 16), C, eps = 1e-6) applied to the channels-first (B, C, N) activations: statistics per sample and
 * group over (C / G channels) x (N tokens), ALWAYS affine (the elementwise_affine flag is ignored upstream).  `norm: ~` = Identity.
 * (`norm: batch_norm` is undefined upstream: its wrapper feeds (B, N, C) to BatchNorm1d(C) and fails unless tokens == channels.)
 * ldt_group_stats: stats[b][g] = (mean, rstd) of sample b's `T` token rows x[b*T .. (b+1)*T)[g*C/G .. (g+1)*C/G), biased variance.
 * ldt_norm_apply:  y[m][c] = bf16( ((x[m][c] - mean) * rstd * w[c] + b[c]) * (1 + scale[s][c]) + shift[s][c] ), (mean, rstd) =
 *   stats[m / rows_per_stat][c / (C/G)] (stats NULL: 0, 1), w / b NULL: 1 / 0, s = m / rows_per_sample, shift / scale NULL: 0 (layers.py:136-137). */
int ldt_group_stats(const float* x, int64_t ldx, int32_t B, int32_t T, int32_t C, int32_t G, float eps, float* stats, void* stream);
int ldt_norm_apply(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int64_t M, int32_t C, const float* stats, int32_t G,
                   int32_t rows_per_stat, const float* w, const float* b, const float* shift, const float* scale,
                   int64_t mod_sample_stride, int32_t rows_per_sample, void* stream);

/* LN-folding monitor: *out = max over the M rows of mean^2 / variance, from the row statistics stats[parts][M][2] a
 * ldt_gemm_resid_lnstats launch wrote (K = parts * 256 channels).  The folded projections' rounding error grows as
 * (1 + mean^2 / variance); the sampler falls back to the LayerNorm kernels when this exceeds its bound. */
int ldt_fold_mean_ratio(const float* stats, int32_t parts, int64_t M, int32_t K, float* out, void* stream);

/* ---- Compressor encoder front end (model/Compressor/layers.py:65-112, 288-319; Network.py:26-29,76,86-107) ----
 * Clouds are fp32 [B][n][3]; index outputs are int32.
 * ldt_fps: farthest point sampling, m centres per cloud, semantics of the vendored CUDA twin
 *   model/functional/src/sampling/sampling.cu:86-167 (start index 0, ties to the smaller (k%512, k/512)).
 *   The reference calls pointnet2_ops.furthest_point_sample (Compressor/layers.py:106; third party, not vendored).
 * ldt_knn: k nearest points of each of the S centres by square_distance (layers.py:65-84) + topk(largest=False,
 *   sorted=False) (:97): an UNORDERED index set [B][S][k]; dist_out (nullable) receives the [B][S][n] distances.
 * ldt_group_normalize: LocalGrouper normalisation (:297-315): per-sample unbiased std of (g - origin)
 *   (stats: fp64 scratch [2*B]), rows U[b,s,j,:] = [alpha*(g-origin)/(std+1e-5)+beta | centre feature], bf16, K padded to ldu.
 *   center_mode 0 = 'anchor' (origin = the centre point's [feature|xyz], :309-311; the Compressor, cluster_norm),
 *   1 = 'center' (origin = mean over the group's k rows, :307-308; ConditionNet, model/scorenet/score.py:22) with
 *   group_mean an fp32 workspace [B][S][D+3] (may be NULL in mode 0).
 * ldt_gather_rows: index_points (:46-62), out[b,s,:] = src[b, idx[b,s], :].
 * ldt_maxpool: max over the middle axis of [G][n][C] (bf16 or fp32 input) -> fp32 [G][C] (:186, Network.py:97).
 * ldt_actnorm: in place (x - shift[t,c]) * exp(-log_scale[t,c]) (model/layers.py:103-107, eval).
 * ldt_reparam: mu|logvar split, clamp(logvar, lo, hi), eps = mu + exp(logvar/2)*noise into a strided slice
 *   (Network.py:26-29,75-77); mu_out/logvar_out nullable.
 * ldt_chamfer: distChamfer (evaluation/evaluation_metrics.py:23-33): dl[b][nb] = min over a, dr[b][na] = min over b. */
int ldt_fps(const float* xyz, int32_t B, int32_t n, int32_t m, int32_t skip_near_origin, int32_t* idx_out, void* stream);
/* ldt_norm_points: Compressor.norm_pts (Network.py:170-174; cfg.compressor.norm_input): per cloud and coordinate (p - mean) / std,
 *   unbiased std over the n points; out may alias xyz.
 * ldt_mixture_seed: InitialSet with max_outputs None (Compressor/layers.py:17-24,38-41): out[r][d] = sum_m (eps[r][m][d] sig[m][d] +
 *   mu[m][d]) softmax(logits)[m] over rows r = b*N + n; the `output` MLP that follows is two ldt_sgemm calls. */
int ldt_norm_points(const float* xyz, int32_t B, int32_t n, float* out, void* stream);
int ldt_mixture_seed(const float* eps, const float* sig, const float* mu, const float* logits, int32_t n_mix, int32_t D, int64_t rows,
                     float* out, void* stream);
int ldt_knn(const float* xyz, const float* centers, int32_t B, int32_t n, int32_t S, int32_t k,
            int32_t* idx_out, float* dist_out, void* stream);
int ldt_group_normalize(const float* feat, const float* xyz, const int32_t* fps_idx, const int32_t* knn_idx,
                        const float* alpha, const float* beta, double* stats, int32_t B, int32_t n, int32_t S,
                        int32_t k, int32_t D, uint16_t* U, int32_t ldu, int32_t center_mode, float* group_mean,
                        void* stream);
int ldt_gather_rows(const float* src, const int32_t* idx, int32_t B, int32_t n, int32_t S, int32_t C, float* out, void* stream);
/* Grouping + PreExtraction + neighbour max in ONE kernel (Compressor/layers.py:288-319 LocalGrouper.forward with
 * normalize='anchor', :115-160 PreExtraction, :186 max over the k neighbours): out fp32 [B*S][128] = max_j relu(W3 . relu(W2 . h1 + b2)
 * + b3 + h1), h1 = relu(W1 . u_j + b1), u_j = the grouped row [alpha*((g_j - anchor)/(std+1e-5)) + beta | anchor features] that
 * ldt_group_normalize would write (never materialised here).  D must be 128, k one of 8, 16 or a multiple of 32.  stats: workspace double[2B].
 * wimg: the three eval-BatchNorm-folded weight panels as bf16 MFMA fragments, 132 x 1 KB:
 *   fragment f, lane l = 32 h + i, slot e (0..7)  ->  W[32 blk + i][col(step, h, e)]
 *   layer 1  f = 4 step + blk, step 0..16: col = 16 step + 8 h + e (steps 0..7: normalised features), 131 + 16 (step - 8) + 8 h + e
 *            (steps 8..15: anchor features), step 16: 128 + e for h = 0, e < 3 (xyz), zero weight elsewhere;
 *   layer 2  f = 68 + 4 step + blk,  layer 3  f = 100 + 4 step + blk, step 0..7: col = 16 step + 8 (e >> 2) + 4 h + (e & 3). */
int ldt_grouper_mlp(const float* feat, const float* xyz, const int32_t* fps_idx, const int32_t* knn_idx, const float* alpha,
                    const float* beta, double* stats, int32_t B, int32_t n, int32_t S, int32_t k, int32_t D,
                    const uint16_t* wimg, const float* b1, const float* b2, const float* b3, float* out, void* stream);
int ldt_maxpool(const void* in, int32_t in_bf16, int64_t ld, int64_t G, int32_t n, int32_t C, float* out, void* stream);
int ldt_actnorm(float* x, const float* shift, const float* log_scale, int64_t B, int64_t per_sample, void* stream);
int ldt_reparam(const float* post, const float* noise, float* out, int64_t ldo, float* mu_out, float* logvar_out,
                int64_t rows, int32_t z, float lo, float hi, void* stream);
int ldt_chamfer(const float* a, const float* b, int32_t B, int32_t na, int32_t nb, float* dl, float* dr, void* stream);

/* ---- fused MLP half of a narrow ResidualBlock (the Compressor's d = 128 blocks; model/layers.py:219,226 + :110-133) ----
 * In place on x fp32 [M][ldx]:  x += gate * (W_dn . GELU(W_up . h + b_up) + b_dn),  h = LN(x) * ln_w + ln_b  (affine,
 * no-condition blocks) or LN(x) * (1 + scale) + shift (AdaLN; shift/scale/gate are per-sample vectors, sample =
 * row / rows_per_sample, consecutive samples mod_sample_stride floats apart).  C in {64, 128}; w_up bf16 [4C][C],
 * w_dn bf16 [C][4C], both row-major and dense.  One pass over x instead of LayerNorm + two GEMMs.
 * x_bf16 (or NULL): bf16 [M][ldxb] copy of the updated x, written in the same pass — the next DecoderBlock level reads the
 * decoded set as its K/V source (Network.py:229 `att(x, o)`), which otherwise costs a separate cast pass over x. */
int ldt_ln_mlp_resid(float* x, int64_t ldx, int64_t M, int32_t C, const float* ln_w, const float* ln_b,
                     const float* shift, const float* scale, const float* gate, int64_t mod_sample_stride,
                     int32_t rows_per_sample, const uint16_t* w_up, const float* b_up, const uint16_t* w_dn,
                     const float* b_dn, uint16_t* x_bf16, int64_t ldxb, void* stream);

/* The same kernel followed, on the rows it has just produced, by the NEXT block's LayerNorm + first projection
 * (model/layers.py:218 / :225 of the ResidualBlock that follows: nx_out bf16 [M][nx_ldo] = LN(x_new)[nx affine | nx modulated] . nx_w^T
 * + nx_bias, nx_w bf16 [nx_N][C], nx_N a multiple of 64) — what ldt_ln_linear would compute from x_new in a pass of its own.  In the
 * Compressor's decoder that is the next level's query projection (Network.py:80-83 via layers.py:225). */
int ldt_ln_mlp_resid_next(float* x, int64_t ldx, int64_t M, int32_t C, const float* ln_w, const float* ln_b,
                          const float* shift, const float* scale, const float* gate, int64_t mod_sample_stride,
                          int32_t rows_per_sample, const uint16_t* w_up, const float* b_up, const uint16_t* w_dn,
                          const float* b_dn, uint16_t* x_bf16, int64_t ldxb,
                          const float* nx_ln_w, const float* nx_ln_b, const float* nx_shift, const float* nx_scale,
                          int64_t nx_mod_sample_stride, int32_t nx_rows_per_sample, const uint16_t* nx_w, const float* nx_bias,
                          int32_t nx_N, uint16_t* nx_out, int64_t nx_ldo, void* stream);

/* LayerNorm + linear for the same narrow blocks: out bf16 [M][ldo] = LN(x)[affine | modulated] . W^T + bias, W bf16 [N][C]
 * dense, N % 64 == 0, C in {64, 128} — fc_q (and fc_kv when the block attends to its own normalised input,
 * model/layers.py:184-189) fed straight from the LayerNorm without writing the normalised activations. */
int ldt_ln_linear(const float* x, int64_t ldx, int64_t M, int32_t C, const float* ln_w, const float* ln_b,
                  const float* shift, const float* scale, int64_t mod_sample_stride, int32_t rows_per_sample,
                  const uint16_t* w, const float* bias, int32_t N, uint16_t* out, int64_t ldo, void* stream);

/* ---- generation-quality metrics of the validation loop (evaluation/evaluation_metrics.py:112-277) ----
 * ldt_chamfer_pairwise: cd[s][r] = dl.mean(1) + dr.mean(1) of distChamfer(x[s], y[r]) for all S*R cloud pairs — the
 *   matrix `_pairwise_CD_` (:165-199) / `_pairwise_EMD_CD_` (:112-162) build row by row.  x fp32 [S][n][3],
 *   y fp32 [R][m][3], cd fp32 [S][R].
 * ldt_emd_approx: transport cost of the approximate matching of
 *   evaluation/pytorch_structural_losses/src/approxmatch.cu (approxmatchkernel :3-186 + matchcostkernel :188-224),
 *   i.e. what `match_cost(xyz1, xyz2)` returns (emd_approx_cuda, evaluation_metrics.py:40-46, divides it by n).
 *   pairwise = 0: out[b] for the pairs (x[b], y[b]), S == R;  pairwise = 1: out[s][r] for all S*R pairs.
 *   n + m <= 7680 points (both clouds and the marginals stay in LDS). */
int ldt_chamfer_pairwise(const float* x, const float* y, int32_t S, int32_t R, int32_t n, int32_t m, float* cd, void* stream);
int ldt_emd_approx(const float* x, const float* y, int32_t S, int32_t R, int32_t n, int32_t m, int32_t pairwise,
                   float* out, void* stream);

/* ---- Score network forward: model/scorenet/score.py:117-151 (Transformer path, unet False) ------------- */
typedef struct ldt_score_plan {
    int32_t hidden, heads, blocks, z_dim, z_pad, mlp_hidden, tokens, batch;
    /* packed weights (bf16 [N][K] row-major, K padded) and fp32 biases */
    const uint16_t* w_in;  const float* b_in;                       /* ln_in   [hidden][z_pad]       score.py:110 */
    const uint16_t* w_qkv[LDT_MAX_BLOCKS]; const float* b_qkv[LDT_MAX_BLOCKS]; /* fc_q|fc_kv rows q,k,v [3h][h] layers.py:159-160 */
    const uint16_t* w_o[LDT_MAX_BLOCKS];   const float* b_o[LDT_MAX_BLOCKS];   /* fc_o   [h][h]        layers.py:161 */
    const uint16_t* w_up[LDT_MAX_BLOCKS];  const float* b_up[LDT_MAX_BLOCKS];  /* mlp.fc.0.0 [4h][h]   layers.py:121 */
    const uint16_t* w_dn[LDT_MAX_BLOCKS];  const float* b_dn[LDT_MAX_BLOCKS];  /* mlp.out [h][4h]      layers.py:124 */
    const uint16_t* w_out; const float* b_out;                      /* ln_out.ln [z_dim][hidden]     layers.py:239 */
    /* Optional cross-attention (ViPC: even blocks attend to the point-cloud condition, score.py:148-149 with
       layers.py:183-189 y != None): when kv_cond[l] != NULL block l takes K|V from kv_cond[l]
       [batch*cond_tokens][2*hidden] bf16 — fc_kv applied to the RAW condition tokens, step-invariant, so the caller
       projects it once per sample() call — and projects only the query with w_q[l]/b_q[l] ([hidden][hidden]). */
    const uint16_t* w_q[LDT_MAX_BLOCKS];   const float* b_q[LDT_MAX_BLOCKS];
    const uint16_t* kv_cond[LDT_MAX_BLOCKS];
    int32_t cond_tokens; int32_t _pad0;
    /* AdaLN modulation: mod[step][sample][blocks*6*hidden + 2*hidden] fp32.  Block l at l*6*hidden:
       shift_msa|scale_msa|gate_msa|shift_mlp|scale_mlp|gate_mlp (layers.py:214); FinalLayer shift|scale (:244) last. */
    const float* mod; int64_t mod_step_stride; int64_t mod_sample_stride;
    /* workspace (caller-owned), M = batch*tokens rows */
    uint16_t* xin;   /* [M][z_pad]       */
    float*    X;     /* [M][hidden] fp32 residual stream */
    uint16_t* Hb;    /* [M][hidden]      */
    uint16_t* QKV;   /* [M][3*hidden]    */
    uint16_t* Ob;    /* [M][hidden]  == [B][H][T][Dh] */
    uint16_t* U;     /* [M][mlp_hidden]  */
    /* Optional LN folding (see ldt_gemm_resid_lnstats): fold[step][blocks][S_qkv 3h | C_qkv 3h | S_up mlp | C_up mlp] fp32
       built by the host from mod and the packed weights; stats = fp32 scratch [hidden/256][M][2].  Used when both are
       non-NULL, mod_sample_stride == 0, no cross-attention, M % 256 == 0, hidden % 256 == 0 <= 1024, mlp_hidden % 256 == 0;
       otherwise the LayerNorm kernels run (the host passes fold only where whole 256x256 tiles fill the chip). */
    const float* fold; int64_t fold_step_stride; float* stats;
    /* Cap on the persistent GEMM grids of this plan (0 = one workgroup per CU = 256).  Two sub-batches sampled on two
       streams give each plan 128: their kernels then share the chip CU-wise and one stream's HBM-bound epilogues /
       attention overlap the other's MFMA-bound main loops (ldt_amd/diffusion.py, `streams`). */
    int32_t gemm_wgs;
    /* ldt_sample_loop only: with fold_monitor set, the steps i with i % fold_monitor_every == 0 and the last step run the monitored
       forward, the others the plain one (0 = every step).  47 monitor launches of ~5 us per monitored forward: every 50th step costs
       0.05 % of a call and the running maximum covers the whole trajectory, not its two ends. */
    int32_t fold_monitor_every;
    /* Optional LN-folding monitor (NULL = off): one device float, raised (atomic max; zero it first) after every folded
       residual GEMM of a forward to max over rows of mean^2 / variance of that GEMM's output — the quantity that scales
       the folded projections' rounding error, (1 + mean^2 / variance) x the LayerNorm kernel's.  The sampler sets it on a
       probe forward before the loop and, since round 6, inside the loop every fold_monitor_every steps (ldt_amd/diffusion.py),
       and reads the running maximum once when the loop has ended. */
    float* fold_monitor;
} ldt_score_plan;

/* Which LN-folded route ldt_score_forward takes for a batch of M = batch*tokens rows (hidden D, mlp_hidden F, gemm_wgs as in the plan) when the plan
 * offers `fold` tables: 0 = none (LayerNorm kernels), 1 = the 256-tile kernels (statistics per 256 columns), 2 = the mid-size tile kernels of small
 * batches (statistics per 32 columns).  The host (Score.can_fold) asks before it builds the tables. */
int ldt_score_lnfold_route(int32_t M, int32_t D, int32_t F, int32_t gemm_wgs);

/* eps_out[M][z_dim] = Score(x[M][z_dim]) with the AdaLN row selected by *step_ptr (NULL = row 0). */
int ldt_score_forward(const ldt_score_plan* plan, const float* x, float* eps_out,
                      const int32_t* step_ptr, void* stream);

/* Same forward with a hipEvent pair around EVERY launch (same stream); returns summed milliseconds and launch
 * counts per kernel class.  Synchronises the stream.  bench.py's live roofline measurement. */
enum ldt_prof_class {
    LDT_PROF_OTHER = 0,       /* cast/pad */
    LDT_PROF_GEMM_IO = 1,     /* ln_in, FinalLayer.ln      (gemm<F32>)   */
    LDT_PROF_LN = 2,          /* LayerNorm + AdaLN modulate               */
    LDT_PROF_GEMM_QKV = 3,    /* fc_q|fc_kv                (gemm<BF16>)  */
    LDT_PROF_ATTN = 4,        /* fused attention                          */
    LDT_PROF_GEMM_O = 5,      /* fc_o + gate + residual    (gemm<RESID>, K = hidden)     */
    LDT_PROF_GEMM_GELU = 6,   /* mlp.fc + GELU             (gemm<GELU>)  */
    LDT_PROF_GEMM_DN = 7,     /* mlp.out + gate + residual (gemm<RESID>, K = mlp_hidden) */
    LDT_PROF_NCLASS = 8
};
int ldt_score_forward_profile(const ldt_score_plan* plan, const float* x, float* eps_out, const int32_t* step_ptr,
                              float* ms_by_class, int32_t* launches_by_class, void* stream);

/* Per-sample conditioning inside the loop (label / ViPC image embedding, score.py:125-135): the AdaLN rows then
 * differ per sample and per step, c[b] = TimeEmbedding(t_i) + extra[b], mod[b] = W_ada · SiLU(c[b]) + b_ada, and are
 * recomputed every step by two extra launches (row add, fp32 SGEMM over the stacked adaLN weights). */
typedef struct ldt_cond_args {
    const float* temb;    /* [n_steps][t_dim]  TimeEmbedding(t_i) for every step (host builds it once per call) */
    const float* extra;   /* [batch][t_dim]    label / image-condition embedding, or NULL */
    const float* w_ada;   /* [n_mod][t_dim]    every block's adaLN.1 weight stacked in plan order (+ FinalLayer's); may be NULL when w_ada_bf16 is given */
    const float* b_ada;   /* [n_mod] */
    float* c_buf;         /* [batch][t_dim]    scratch */
    float* mod_buf;       /* [batch][n_mod]    scratch; plan->mod must point here with mod_sample_stride = n_mod */
    int32_t t_dim, n_mod;
    /* optional (both or neither): the stacked adaLN weights as a bf16 panel [n_mod][t_dim] and a bf16 scratch [batch][t_dim].  When
     * given, the per-step rows are one bf16 MFMA GEMM (fp32 accumulation and bias) that streams 2 bytes per weight instead of 4 —
     * the fp32 form is HBM-bound on its 4 n_mod t_dim bytes per step (BASELINE configs[4]: 604 MB, 9 % of a step).  Rows then carry
     * bf16 operand rounding (relative MSE ~5e-6), like every token GEMM of the model.  The MFMA GEMM needs t_dim % 64 == 0: with another
     * width the loop uses the fp32 w_ada rows if they were given too, and returns LDT_ESHAPE otherwise. */
    const void* w_ada_bf16;
    void* c_buf_bf16;
} ldt_cond_args;

/* The whole reverse-SDE loop (pc_sampling, diffusion_continuous.py:231-258 with corrector None): n_steps x
 * [(AdaLN rows if cond) -> Score forward -> predictor update -> ++step].  x is updated in place, x_mean receives the
 * last x_mean (denoise=True returns it, quirk Q8).  step_counter: device int32 scratch.  cond: NULL for the
 * unconditional sampler (plan->mod = table of all steps).  x_traj: NULL, or [n_steps][batch*tokens*z_dim] fp32 that
 * receives x after every step (the per-step parity curve against the CPU reference).  use_graph != 0 captures one step
 * into a hipGraph and replays it. */
int ldt_sample_loop(const ldt_score_plan* plan, float* x, float* x_mean, float* eps_tmp,
                    const float* coef, int32_t mode, const float* noise, int64_t noise_step_stride,
                    int64_t elem_offset, uint64_t seed, int32_t* step_counter, int32_t n_steps,
                    const ldt_cond_args* cond, float* x_traj, int32_t use_graph, void* stream);

/* ---- measurement knob (tools/dbg/gm_bench.py): rows per group of the persistent GEMM's grouped tile order
 * (1 = row-major, the default; LDT_GEMM_GM sets it at start-up).  Process-wide; not used by the product path. */
int ldt_dbg_gemm_group_m(int32_t rows_per_group);
/* tools/dbg/epi_ablate.py: skip parts of the residual epilogue (bit 1 residual read, 2 fp32 store, 4 bf16 x(1+scale) store,
 * 8 row statistics) to attribute its time; -1 restores the product behaviour.  Process-wide; not used by the product path. */
int ldt_dbg_gemm_epi(int32_t bits);
/* tools/dbg + tests: 1 = every 256-tile GEMM launch without a fragment-order weight copy packs one on the fly (cached by pointer and
 * shape, never invalidated) and runs the W-from-registers form; 0 = that form off; -1 restores the product behaviour.  Process-wide. */
int ldt_dbg_gemm_wreg(int32_t on);

#ifdef __cplusplus
}
#endif
#endif
