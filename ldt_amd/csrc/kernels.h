// Internal argument blocks + launcher prototypes shared by the kernel files and api.hip.
#pragma once
#include "common.h"

enum { EPI_F32 = 0, EPI_BF16 = 1, EPI_GELU_BF16 = 2, EPI_RELU_BF16 = 3, EPI_RESID_F32 = 4 };
#define LDT_NUM_CUS 256      // MI355X
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_GELU = 3 };

struct GemmArgs {
    const bf16_t* X; long ldx;
    const bf16_t* W; long ldw;
    const float* bias;
    void* out; long ldo;
    const float* resid; long ldr;       // EPI_RESID_F32 (may alias out)
    const bf16_t* skip; long lds_;      // EPI_RELU_BF16 optional skip
    const float* gate;                  // [samples or 1][gate_sample_stride] fp32, nullable
    long gate_sample_stride;            // 0 = one gate vector shared by every row
    int rows_per_sample;
    const int* step_ptr; long gate_step_stride;   // gate += *step_ptr * gate_step_stride (device-side step counter)
    int M, N, K;
    // ---- LayerNorm folding (256-tile kernel, interior tiles only; gemm_bf16.hip "LN folding") ----
    // producer (EPI_RESID_F32): second output xs[m][n] = bf16(out[m][n] * (1 + ln_scale[n])) and per-row partial
    // (sum, sum of squares) of out over each 256-column tile: stats_out[(n/256)][M][2]
    bf16_t* xs; long ldxs; const float* ln_scale; long ln_step_stride; float* stats_out;
    // consumer (EPI_BF16 / EPI_GELU_BF16): X is the producer's xs; out = epi( rstd*acc - rstd*mean*fold_S[n] + fold_C[n] )
    // with (mean, rstd) of each row from stats_in[stats_parts][M][2] over the K input channels
    const float* stats_in; int stats_parts; const float* fold_S; const float* fold_C; long fold_step_stride;
    int group_m;                        // 256-tile kernel: row panels per group of the tile order (set by the launcher)
    int dbg;                            // tools/dbg only (LDT_DBG_EPI bits: 1 no residual read, 2 no fp32 store, 4 no xs store, 8 no statistics)
    int max_wgs;                        // 256-tile kernel: cap on the persistent grid (0 = one workgroup per CU); sub-batch streams use 128
    int col_major;                      // v1 kernel: tile order (set by the launcher; see gemm_bf16_nt_kernel)
    // ---- QKV projection + self-attention in one launch (gemm_mid.hip, 32-token samples, head dim 64): N = 3 * hidden columns [q | k | v];
    // the kernel writes O[B][H][32][64] (the reference's raw (B,N,C) buffer, model/layers.py:190-197) to attn_o and NOT the q | k | v rows
    bf16_t* attn_o; float attn_scale_log2e;
    // fused q projection + cross-attention (gemm_mid.hip, ldt_gemm_mid_q_xattn_try): the condition's cached K | V rows (elements)
    const bf16_t* attn_k; const bf16_t* attn_v; long attn_ldkv; long attn_kv_batch_stride;
    // 256-tile kernel, WREG form: W once more in MFMA-fragment order (gemm_bf16.hip "W from registers"; ldt_gemm_pack_wfrag), nullable
    const bf16_t* Wp;
};

struct LnArgs {
    const float* x; long ldx;
    bf16_t* y; long ldy;
    const float* w; const float* b;              // affine (decoder blocks), nullable
    const float* shift; const float* scale;      // nullable (no modulation)
    long mod_sample_stride; int rows_per_sample;
    const int* step_ptr; long mod_step_stride;
    long M; int C;
};

struct AttnArgs {
    const bf16_t* Q; long ldq; long q_batch_stride;     // Q[b][n][h*Dh + d]
    const bf16_t* K; long ldk; long kv_batch_stride;    // K[b][m][h*Dh + d]
    const bf16_t* V; long ldv;                          // V[b][m][h*Dh + d] (same batch stride as K)
    bf16_t* O;                                          // [B][H][Nq][Dh]
    int B, H, Nq, Nk;
    float scale_log2e;                                  // Dh^-0.5 * log2(e)
    // fused output projection (narrow blocks, Dh = 32, C = H*Dh in {64,128}):  X += gate * (Wo . O' + bo), O' = the
    // [B][H][Nq][Dh] result re-read as (B*Nq, C) rows (quirk Q1); all null/0 for the plain kernel
    const bf16_t* Wo; const float* bo; float* X; long ldx; const float* gate; long gate_sample_stride;
};

struct StepArgs {
    const float* x; const float* params; const float* noise;
    float* x_out; float* x_mean_out;
    const float* coef; const int* step_ptr; int step_host; int mode;
    long n; long elem_offset; long noise_step_stride;
    uint32_t seed_lo, seed_hi;
    int philox_mul, philox_add;          // Philox stream id of this draw = step * philox_mul + philox_add
    float* traj;                         // optional [n_steps][n]: x after this step is also stored at traj + step*n (parity curves)
};

struct SgemmArgs {
    const float* A; long lda; const float* B; long ldb; const float* bias;
    void* C; long ldc; int out_bf16; int act_in; int act_out;
    int M, N, K;
};

int ldt_gemm_launch(int epi, const GemmArgs* a, hipStream_t stream);
int ldt_gemm_mid_shape(int epi, const GemmArgs* a);                // gemm_mid.hip: (BM << 16) | BN of the mid-size tile kernel for this problem, 0 = not taken
int ldt_gemm_mid_launch(int epi, int shape, const GemmArgs* a, hipStream_t stream);
bool ldt_gemm_mid_lnfold_takes(int epi, int M, int N, int K);       // would the LN-folded form (statistics per 32 columns) of this GEMM be taken?
bool ldt_gemm_mid_lnfold_try(int epi, const GemmArgs* a, hipStream_t stream, int* status);
bool ldt_gemm_mid_qkv_attn_try(const GemmArgs* a, int tokens, int head_dim, bool folded, hipStream_t stream, int* status);
bool ldt_gemm_qkv_attn256_try(const GemmArgs* a, int tokens, int head_dim, bool folded, hipStream_t stream, int* status);   // gemm_bf16.hip: fused QKV + self-attention at 256 tokens, Dh 64; false = not taken
bool ldt_gemm_mid_q_xattn_try(const GemmArgs* a, int tokens, int cond_tokens, int head_dim, hipStream_t stream, int* status);   // fused q projection + cross-attention (32 x 32 tokens, Dh 64); false = not taken
//   // fused QKV + attention (32 tokens, Dh 64); false = not taken   // LN-folded producer / consumer (statistics per 32 columns); false = not taken
bool ldt_gemm_lnfold_v1_route(int M, int D, int F, int max_wgs);   // small batch: every GEMM of a Score block folds through the mid-size tile kernel (statistics per 32 columns)
int ldt_gemm_lnfold_launch(int epi, const GemmArgs* a, hipStream_t stream);   // producer (RESID + xs/stats) or consumer (stats_in)
int ldt_ln_launch(const LnArgs* a, hipStream_t s);
int ldt_attn_launch(const AttnArgs* a, int dh, hipStream_t s);
int ldt_attn_route(int B, int H, int Nq, int Nk, int dh);       // 0 streaming, 1 resident, 2 whole-head (attention.hip)
int ldt_attn_oproj_launch(const AttnArgs* a, int dh, hipStream_t s);
int ldt_cast_pad_launch(const float* src, long lds, bf16_t* dst, long ldd, long rows, int cols, int cols_pad, hipStream_t s);
int ldt_sampler_step_launch(const StepArgs* a, hipStream_t s);
int ldt_advance_step_launch(int* step_ptr, hipStream_t s);
int ldt_philox_normal_launch(float* out, long n, long elem_offset, int step, uint32_t k0, uint32_t k1, hipStream_t s);
int ldt_sinusoid_launch(const float* t, const float* freq, float* e, int n, int half, hipStream_t s);
int ldt_sgemm_launch(const SgemmArgs* a, hipStream_t s);
bool ldt_sgemm_mfma_try(const SgemmArgs* a, hipStream_t s, int* status);   // sgemm_mfma.hip: false = shape not taken
bool ldt_skinny_linear_try(const SgemmArgs* a, hipStream_t s, int* status);   // skinny_linear.hip

int ldt_fps_launch(const float* xyz, int B, int n, int m, int skip_near_origin, int* idx, hipStream_t s);
int ldt_fps_wave_launch(const float* xyz, int B, int n, int m, int skip_near_origin, int* idx, hipStream_t s);   // fps_wave.hip
int ldt_knn_launch(const float* xyz, const float* centers, int B, int n, int S, int k, int* out, float* dist_out, hipStream_t s);
int ldt_group_launch(const float* feat, const float* xyz, const int* fps_idx, const int* knn_idx, const float* alpha,
                     const float* beta, double* stats, int B, int n, int S, int k, int D, bf16_t* U, int ldu,
                     int center_mode, float* gmean, hipStream_t s);
int ldt_group_stats_launch(const float* feat, const float* xyz, const int* fps_idx, const int* knn_idx, double* stats,
                           int B, int n, int S, int k, int D, hipStream_t s);
// grouping + PreExtraction + neighbour max in one kernel (grouper_mlp.hip): D = 128 channels, k in {8, 16, 32 m}, 'anchor' mode
struct GroupMlpArgs {
    const float* feat; const float* xyz;               // [B][n][128], [B][n][3]
    const int* fps_idx; const int* knn_idx;            // [B][S], [B][S][k]
    const float* alpha; const float* beta;             // [131]
    const double* stats;                               // [2B] from ldt_group_stats_launch
    const bf16_t* wimg;                                // MFMA fragment image of the three weight panels (132 x 1 KB)
    const float* b1; const float* b2; const float* b3; // [128] each
    int B, n, S, k, flat;
    float* out;                                        // [B*S][128]
};
int ldt_grouper_mlp_launch(const GroupMlpArgs* a, hipStream_t s);
int ldt_norm_points_launch(const float* xyz, int B, int n, float* out, hipStream_t s);
int ldt_mixture_seed_launch(const float* eps, const float* sig, const float* mu, const float* logits, int n_mix, int D, long rows, float* out, hipStream_t s);
int ldt_gather_rows_launch(const float* src, const int* idx, int B, int n, int S, int C, float* out, hipStream_t s);
int ldt_maxpool_launch(const void* in, int in_bf16, long ld, long G, int n, int C, float* out, hipStream_t s);
int ldt_actnorm_launch(float* x, const float* shift, const float* log_scale, long B, long per_sample, hipStream_t s);
int ldt_reparam_launch(const float* post, const float* noise, float* out, long ldo, float* mu_out, float* lv_out,
                       long rows, int z, float lo, float hi, hipStream_t s);
int ldt_chamfer_launch(const float* a, const float* b, int B, int na, int nb, float* dl, float* dr, hipStream_t s);
// fused LayerNorm + MLP + gated residual for narrow blocks (fused_mlp.hip)
struct LnLinArgs {
    const float* x; long ldx; long M;
    const float* ln_w; const float* ln_b; const float* shift; const float* scale; long mod_sample_stride; int rows_per_sample;
    const bf16_t* w; const float* bias; int N;          // [N][C], [N] (nullable)
    bf16_t* out; long ldo;
};
struct MlpArgs {
    float* x; long ldx; long M;
    const float* ln_w; const float* ln_b;              // affine LayerNorm (no-condition blocks) or null
    const float* shift; const float* scale;            // AdaLN modulation (per sample) or null
    const float* gate;                                 // per-sample gate or null (= 1)
    long mod_sample_stride; int rows_per_sample;
    const bf16_t* w_up; const float* b_up;             // [4C][C], [4C]
    const bf16_t* w_dn; const float* b_dn;             // [C][4C], [C]
    bf16_t* x_bf16; long ldxb;                          // optional bf16 mirror of the updated x (the next block's K/V source) or null
    LnLinArgs next;                                     // optional (next.w != null): LN + linear of the block that follows, on the new x (next.x unused)
};
int ldt_ln_mlp_launch(const MlpArgs* a, int C, hipStream_t st);
// fused LayerNorm + linear (bf16 out) for narrow blocks (fused_mlp.hip)
int ldt_ln_linear_launch(const LnLinArgs* a, int C, hipStream_t st);
int ldt_chamfer_pairwise_launch(const float* x, const float* y, int S, int R, int n, int m, float* cd, hipStream_t st);
int ldt_emd_approx_launch(const float* x, const float* y, int S, int R, int n, int m, int pairwise, float* out, hipStream_t st);
int ldt_fold_monitor_launch(const float* stats, int parts, long M, int K, float* out, hipStream_t s);   // samplers.hip
int ldt_cond_rows_launch(const float* temb, const float* extra, float* c, void* c_bf16, const int* step_ptr, int batch, int t_dim, int silu_out, hipStream_t s);
