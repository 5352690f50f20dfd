// C-ABI of libldt_hip.so (include/ldt_hip.h): argument marshalling, error reporting, and the two
// host-side orchestrators that enqueue the whole Score forward / reverse-SDE loop on one HIP stream.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/ldt_hip.h"
#include "kernels.h"

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void ldt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int ldt_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ldt_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return LDT_OK;
}
extern "C" const char* ldt_last_error(void) { return g_err; }
extern "C" int ldt_abi_version(void) { return LDT_ABI_VERSION; }

// ------------------------------------------------------------------------------ internal launchers
#define BF(p) reinterpret_cast<const bf16_t*>(p)
#define BFM(p) reinterpret_cast<bf16_t*>(p)
#define ST(s) reinterpret_cast<hipStream_t>(s)
#define TRY(expr)                   \
    do {                            \
        const int _rc = (expr);     \
        if (_rc != LDT_OK) return _rc; \
    } while (0)

// ------------------------------------------------------------------------------ primitive ops
extern "C" int ldt_cast_pad_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows,
                                 int32_t cols, int32_t cols_pad, void* stream) {
    LDT_REQUIRE(src && dst, LDT_EARG, "cast_pad: null pointer");
    return ldt_cast_pad_launch(src, ld_src, BFM(dst), ld_dst, rows, cols, cols_pad, ST(stream));
}

extern "C" int ldt_gemm_bf16(int32_t epilogue, const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw,
                             const float* bias, void* out, int64_t ldo, const float* resid, int64_t ldr,
                             const uint16_t* skip, int64_t ldskip, const float* gate, int64_t gate_sample_stride,
                             int32_t rows_per_sample, const int32_t* step_ptr, int64_t gate_step_stride,
                             int32_t M, int32_t N, int32_t K, void* stream) {
    LDT_REQUIRE(X && W && out, LDT_EARG, "gemm: null pointer");
    GemmArgs a{BF(X), ldx, BF(W), ldw, bias, out, ldo, resid, ldr, BF(skip), ldskip, gate, gate_sample_stride,
               rows_per_sample, step_ptr, gate_step_stride, M, N, K};
    return ldt_gemm_launch(epilogue, &a, ST(stream));
}

extern "C" int ldt_gemm_resid_lnstats(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias,
                                      float* out, int64_t ldo, const float* gate, int64_t gate_sample_stride,
                                      int32_t rows_per_sample, const float* ln_scale, uint16_t* xs, int64_t ldxs,
                                      float* stats_out, const int32_t* step_ptr, int64_t gate_step_stride,
                                      int64_t ln_step_stride, int32_t M, int32_t N, int32_t K, int32_t stats_parts, void* stream) {
    LDT_REQUIRE(X && W && out && xs && ln_scale && stats_out, LDT_EARG, "gemm_resid_lnstats: null pointer");
    LDT_REQUIRE(stats_parts == N / 256 || stats_parts == N / 32, LDT_EARG, "gemm_resid_lnstats: stats_parts=%d is neither N/256 nor N/32 (N=%d)", stats_parts, N);
    GemmArgs a{BF(X), ldx, BF(W), ldw, bias, out, ldo, out, ldo, nullptr, 0, gate, gate_sample_stride, rows_per_sample, step_ptr,
               gate_step_stride, M, N, K, BFM(xs), ldxs, ln_scale, ln_step_stride, stats_out};
    a.stats_parts = stats_parts;
    return ldt_gemm_lnfold_launch(EPI_RESID_F32, &a, ST(stream));
}

extern "C" int ldt_gemm_lnfold(int32_t epilogue, const uint16_t* Xs, int64_t ldx, const uint16_t* W, int64_t ldw,
                               const float* stats_in, const float* fold_S, const float* fold_C, uint16_t* out, int64_t ldo,
                               const int32_t* step_ptr, int64_t fold_step_stride, int32_t M, int32_t N, int32_t K, int32_t stats_parts,
                               void* stream) {
    LDT_REQUIRE(Xs && W && out && stats_in && fold_S && fold_C, LDT_EARG, "gemm_lnfold: null pointer");
    LDT_REQUIRE(stats_parts == K / 256 || stats_parts == K / 32, LDT_EARG, "gemm_lnfold: stats_parts=%d is neither K/256 nor K/32 (K=%d)", stats_parts, K);
    GemmArgs a{BF(Xs), ldx, BF(W), ldw, nullptr, out, ldo, nullptr, 0, nullptr, 0, nullptr, 0, 0, step_ptr, 0, M, N, K,
               nullptr, 0, nullptr, 0, nullptr, stats_in, stats_parts, fold_S, fold_C, fold_step_stride};
    return ldt_gemm_lnfold_launch(epilogue, &a, ST(stream));
}

extern "C" int ldt_layernorm_modulate(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, const float* w,
                                      const float* b, const float* shift, const float* scale,
                                      int64_t mod_sample_stride, int32_t rows_per_sample, const int32_t* step_ptr,
                                      int64_t mod_step_stride, int64_t M, int32_t C, void* stream) {
    LDT_REQUIRE(x && y, LDT_EARG, "ln: null pointer");
    LnArgs a{x, ldx, BFM(y), ldy, w, b, shift, scale, mod_sample_stride, rows_per_sample, step_ptr, mod_step_stride, M, C};
    return ldt_ln_launch(&a, ST(stream));
}

extern "C" int ldt_attention_fwd(const uint16_t* Q, int64_t ldq, int64_t q_batch_stride, const uint16_t* K,
                                 int64_t ldk, const uint16_t* V, int64_t ldv, int64_t kv_batch_stride, uint16_t* O,
                                 int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t head_dim, void* stream) {
    LDT_REQUIRE(Q && K && V && O, LDT_EARG, "attention: null pointer");
    AttnArgs a{BF(Q), ldq, q_batch_stride, BF(K), ldk, kv_batch_stride, BF(V), ldv, BFM(O), B, H, Nq, Nk,
               1.4426950408889634f / sqrtf((float)head_dim), nullptr, nullptr, nullptr, 0, nullptr, 0};
    return ldt_attn_launch(&a, head_dim, ST(stream));
}
extern "C" int ldt_attention_route(int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t head_dim) { return ldt_attn_route(B, H, Nq, Nk, head_dim); }
extern "C" int ldt_attention_oproj_resid(const uint16_t* Q, int64_t ldq, int64_t q_batch_stride, const uint16_t* K,
                                         int64_t ldk, const uint16_t* V, int64_t ldv, int64_t kv_batch_stride,
                                         int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t head_dim,
                                         const uint16_t* Wo, const float* bo, float* X, int64_t ldx, const float* gate,
                                         int64_t gate_sample_stride, void* stream) {
    LDT_REQUIRE(Q && K && V && Wo && bo && X, LDT_EARG, "attention_oproj: null pointer");
    AttnArgs a{BF(Q), ldq, q_batch_stride, BF(K), ldk, kv_batch_stride, BF(V), ldv, nullptr, B, H, Nq, Nk,
               1.4426950408889634f / sqrtf((float)head_dim), BF(Wo), bo, X, ldx, gate, gate_sample_stride};
    return ldt_attn_oproj_launch(&a, head_dim, ST(stream));
}

extern "C" int ldt_sgemm(const float* A, int64_t lda, const float* Bw, int64_t ldb, const float* bias, void* C,
                         int64_t ldc, int32_t out_bf16, int32_t act_in, int32_t act_out, int32_t M, int32_t N,
                         int32_t K, void* stream) {
    SgemmArgs a{A, lda, Bw, ldb, bias, C, ldc, out_bf16, act_in, act_out, M, N, K};
    return ldt_sgemm_launch(&a, ST(stream));
}

extern "C" int ldt_sinusoid(const float* t, const float* freq, float* e, int32_t n, int32_t half, void* stream) {
    LDT_REQUIRE(t && freq && e, LDT_EARG, "sinusoid: null pointer");
    return ldt_sinusoid_launch(t, freq, e, n, half, ST(stream));
}

extern "C" int ldt_sampler_step(const float* x, const float* params, const float* noise, int64_t noise_step_stride,
                                float* x_out, float* x_mean_out, const float* coef, const int32_t* step_ptr,
                                int32_t step_host, int32_t mode, int64_t n, int64_t elem_offset, uint64_t seed,
                                int32_t philox_mul, int32_t philox_add, void* stream) {
    StepArgs a{x, params, noise, x_out, x_mean_out, coef, step_ptr, step_host, mode, n, elem_offset, noise_step_stride,
               (uint32_t)seed, (uint32_t)(seed >> 32), philox_mul, philox_add};
    return ldt_sampler_step_launch(&a, ST(stream));
}

extern "C" int ldt_philox_normal(float* out, int64_t n, int64_t elem_offset, int32_t step, uint64_t seed, void* stream) {
    LDT_REQUIRE(out, LDT_EARG, "philox_normal: null pointer");
    return ldt_philox_normal_launch(out, n, elem_offset, step, (uint32_t)seed, (uint32_t)(seed >> 32), ST(stream));
}

// ------------------------------------------------------------------------------ Compressor encoder front end
extern "C" int ldt_fps(const float* xyz, int32_t B, int32_t n, int32_t m, int32_t skip_near_origin, int32_t* idx_out, void* stream) {
    LDT_REQUIRE(xyz && idx_out, LDT_EARG, "fps: null pointer");
    return ldt_fps_launch(xyz, B, n, m, skip_near_origin, idx_out, ST(stream));
}
extern "C" int ldt_knn(const float* xyz, const float* centers, int32_t B, int32_t n, int32_t S, int32_t k,
                       int32_t* idx_out, float* dist_out, void* stream) {
    LDT_REQUIRE(xyz && centers && idx_out, LDT_EARG, "knn: null pointer");
    return ldt_knn_launch(xyz, centers, B, n, S, k, idx_out, dist_out, ST(stream));
}
extern "C" int ldt_group_normalize(const float* feat, const float* xyz, const int32_t* fps_idx, const int32_t* knn_idx,
                                   const float* alpha, const float* beta, double* stats, int32_t B, int32_t n, int32_t S,
                                   int32_t k, int32_t D, uint16_t* U, int32_t ldu, int32_t center_mode, float* group_mean,
                                   void* stream) {
    LDT_REQUIRE(feat && xyz && fps_idx && knn_idx && alpha && beta && stats && U, LDT_EARG, "group: null pointer");
    return ldt_group_launch(feat, xyz, fps_idx, knn_idx, alpha, beta, stats, B, n, S, k, D, BFM(U), ldu, center_mode, group_mean,
                            ST(stream));
}
extern "C" int ldt_grouper_mlp(const float* feat, const float* xyz, const int32_t* fps_idx, const int32_t* knn_idx, const float* alpha,
                               const float* beta, double* stats, int32_t B, int32_t n, int32_t S, int32_t k, int32_t D,
                               const uint16_t* wimg, const float* b1, const float* b2, const float* b3, float* out, void* stream) {
    LDT_REQUIRE(feat && xyz && fps_idx && knn_idx && alpha && beta && stats && wimg && b1 && b2 && b3 && out, LDT_EARG, "grouper_mlp: null pointer");
    LDT_REQUIRE(D == 128, LDT_ESHAPE, "grouper_mlp: the fused kernel is built for 128 channels (got D=%d)", D);
    const int rc = ldt_group_stats_launch(feat, xyz, fps_idx, knn_idx, stats, B, n, S, k, D, ST(stream));
    if (rc != LDT_OK) return rc;
    GroupMlpArgs a{feat, xyz, fps_idx, knn_idx, alpha, beta, stats, BF(wimg), b1, b2, b3, B, n, S, k, 0, out};
    return ldt_grouper_mlp_launch(&a, ST(stream));
}
extern "C" int ldt_norm_points(const float* xyz, int32_t B, int32_t n, float* out, void* stream) {
    LDT_REQUIRE(xyz && out, LDT_EARG, "norm_points: null pointer");
    return ldt_norm_points_launch(xyz, B, n, out, ST(stream));
}
extern "C" int ldt_mixture_seed(const float* eps, const float* sig, const float* mu, const float* logits, int32_t n_mix, int32_t D, int64_t rows,
                                float* out, void* stream) {
    LDT_REQUIRE(eps && sig && mu && logits && out, LDT_EARG, "mixture_seed: null pointer");
    return ldt_mixture_seed_launch(eps, sig, mu, logits, n_mix, D, rows, out, ST(stream));
}
extern "C" int ldt_gather_rows(const float* src, const int32_t* idx, int32_t B, int32_t n, int32_t S, int32_t C, float* out, void* stream) {
    LDT_REQUIRE(src && idx && out, LDT_EARG, "gather_rows: null pointer");
    return ldt_gather_rows_launch(src, idx, B, n, S, C, out, ST(stream));
}
extern "C" int ldt_maxpool(const void* in, int32_t in_bf16, int64_t ld, int64_t G, int32_t n, int32_t C, float* out, void* stream) {
    LDT_REQUIRE(in && out, LDT_EARG, "maxpool: null pointer");
    return ldt_maxpool_launch(in, in_bf16, ld, G, n, C, out, ST(stream));
}
extern "C" int ldt_actnorm(float* x, const float* shift, const float* log_scale, int64_t B, int64_t per_sample, void* stream) {
    LDT_REQUIRE(x && shift && log_scale, LDT_EARG, "actnorm: null pointer");
    return ldt_actnorm_launch(x, shift, log_scale, B, per_sample, ST(stream));
}
extern "C" int ldt_reparam(const float* post, const float* noise, float* out, int64_t ldo, float* mu_out, float* logvar_out,
                           int64_t rows, int32_t z, float lo, float hi, void* stream) {
    LDT_REQUIRE(post && noise && out, LDT_EARG, "reparam: null pointer");
    return ldt_reparam_launch(post, noise, out, ldo, mu_out, logvar_out, rows, z, lo, hi, ST(stream));
}
extern "C" int ldt_chamfer(const float* a, const float* b, int32_t B, int32_t na, int32_t nb, float* dl, float* dr, void* stream) {
    LDT_REQUIRE(a && b && dl && dr, LDT_EARG, "chamfer: null pointer");
    return ldt_chamfer_launch(a, b, B, na, nb, dl, dr, ST(stream));
}

extern "C" int ldt_ln_mlp_resid(float* x, int64_t ldx, int64_t M, int32_t C, const float* ln_w, const float* ln_b,
                                const float* shift, const float* scale, const float* gate, int64_t mod_sample_stride,
                                int32_t rows_per_sample, const uint16_t* w_up, const float* b_up, const uint16_t* w_dn,
                                const float* b_dn, uint16_t* x_bf16, int64_t ldxb, void* stream) {
    LDT_REQUIRE(x && w_up && b_up && w_dn && b_dn, LDT_EARG, "ln_mlp: null pointer");
    MlpArgs a{x, ldx, M, ln_w, ln_b, shift, scale, gate, mod_sample_stride, rows_per_sample, BF(w_up), b_up, BF(w_dn), b_dn,
              BFM(x_bf16), ldxb};
    return ldt_ln_mlp_launch(&a, C, ST(stream));
}
extern "C" int ldt_ln_mlp_resid_next(float* x, int64_t ldx, int64_t M, int32_t C, const float* ln_w, const float* ln_b,
                                     const float* shift, const float* scale, const float* gate, int64_t mod_sample_stride,
                                     int32_t rows_per_sample, const uint16_t* w_up, const float* b_up, const uint16_t* w_dn,
                                     const float* b_dn, uint16_t* x_bf16, int64_t ldxb,
                                     const float* nx_ln_w, const float* nx_ln_b, const float* nx_shift, const float* nx_scale,
                                     int64_t nx_mod_sample_stride, int32_t nx_rows_per_sample, const uint16_t* nx_w, const float* nx_bias,
                                     int32_t nx_N, uint16_t* nx_out, int64_t nx_ldo, void* stream) {
    LDT_REQUIRE(x && w_up && b_up && w_dn && b_dn && nx_w && nx_out, LDT_EARG, "ln_mlp_next: null pointer");
    MlpArgs a{x, ldx, M, ln_w, ln_b, shift, scale, gate, mod_sample_stride, rows_per_sample, BF(w_up), b_up, BF(w_dn), b_dn,
              BFM(x_bf16), ldxb,
              LnLinArgs{nullptr, 0, M, nx_ln_w, nx_ln_b, nx_shift, nx_scale, nx_mod_sample_stride, nx_rows_per_sample, BF(nx_w), nx_bias,
                        nx_N, BFM(nx_out), nx_ldo}};
    return ldt_ln_mlp_launch(&a, C, ST(stream));
}
extern "C" int ldt_ln_linear(const float* x, int64_t ldx, int64_t M, int32_t C, const float* ln_w, const float* ln_b,
                             const float* shift, const float* scale, int64_t mod_sample_stride, int32_t rows_per_sample,
                             const uint16_t* w, const float* bias, int32_t N, uint16_t* out, int64_t ldo, void* stream) {
    LDT_REQUIRE(x && w && out, LDT_EARG, "ln_linear: null pointer");
    LnLinArgs a{x, ldx, M, ln_w, ln_b, shift, scale, mod_sample_stride, rows_per_sample, BF(w), bias, N, BFM(out), ldo};
    return ldt_ln_linear_launch(&a, C, ST(stream));
}
extern "C" int ldt_chamfer_pairwise(const float* x, const float* y, int32_t S, int32_t R, int32_t n, int32_t m, float* cd, void* stream) {
    LDT_REQUIRE(x && y && cd, LDT_EARG, "chamfer_pairwise: null pointer");
    return ldt_chamfer_pairwise_launch(x, y, S, R, n, m, cd, ST(stream));
}
extern "C" int ldt_emd_approx(const float* x, const float* y, int32_t S, int32_t R, int32_t n, int32_t m, int32_t pairwise,
                              float* out, void* stream) {
    LDT_REQUIRE(x && y && out, LDT_EARG, "emd_approx: null pointer");
    return ldt_emd_approx_launch(x, y, S, R, n, m, pairwise, out, ST(stream));
}

// ------------------------------------------------------------------------------ Score forward
extern "C" int ldt_score_lnfold_route(int32_t M, int32_t D, int32_t F, int32_t gemm_wgs) {
    if (M <= 0 || D <= 0 || F <= 0 || D % 256 != 0 || D > 1024 || F % 256 != 0) return 0;
    if (ldt_gemm_lnfold_v1_route(M, D, F, gemm_wgs)) return 2;
    const int lim = (gemm_wgs > 0 && gemm_wgs < LDT_NUM_CUS) ? gemm_wgs : LDT_NUM_CUS;
    return (M % 256 == 0 && (long)(M / 256) * (D / 256) * 8 >= (long)lim * 5) ? 1 : 0;
}

static int check_plan(const ldt_score_plan* p) {
    LDT_REQUIRE(p, LDT_EARG, "score: null plan");
    LDT_REQUIRE(p->blocks > 0 && p->blocks <= LDT_MAX_BLOCKS, LDT_ESHAPE, "score: blocks=%d out of range", p->blocks);
    LDT_REQUIRE(p->hidden > 0 && p->heads > 0 && p->hidden % p->heads == 0, LDT_ESHAPE, "score: hidden %% heads");
    const int dh = p->hidden / p->heads;
    LDT_REQUIRE(dh == 32 || dh == 64, LDT_ESHAPE, "score: head dim %d not built (32, 64)", dh);
    LDT_REQUIRE(p->hidden % 64 == 0 && p->mlp_hidden % 64 == 0 && p->z_pad % 64 == 0 && p->z_pad >= p->z_dim, LDT_ESHAPE,
                "score: hidden/mlp_hidden/z_pad must be multiples of 64");
    LDT_REQUIRE(p->z_dim % 4 == 0, LDT_ESHAPE, "score: z_dim %% 4");
    LDT_REQUIRE(p->tokens > 0 && p->batch > 0, LDT_ESHAPE, "score: empty batch");
    LDT_REQUIRE(p->w_in && p->w_out && p->mod && p->xin && p->X && p->Hb && p->QKV && p->Ob && p->U, LDT_EARG, "score: null buffer in plan");
    for (int l = 0; l < p->blocks; ++l) {
        LDT_REQUIRE(p->w_qkv[l] && p->w_o[l] && p->w_up[l] && p->w_dn[l], LDT_EARG, "score: block %d weights missing", l);
        LDT_REQUIRE(!p->kv_cond[l] || (p->w_q[l] && p->cond_tokens > 0), LDT_EARG, "score: block %d cross-attention needs w_q and cond_tokens", l);
    }
    return LDT_OK;
}

// Optional per-launch HIP-event timing (bench.py's live roofline measurement): events are recorded on the
// SAME stream the kernels run on, bracketing every launch; elapsed times are summed per kernel class.
struct Prof {
    hipStream_t s;
    std::vector<hipEvent_t> ev;      // 2 per launch
    std::vector<int> cls;
    void begin(int c) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); ev.push_back(a); ev.push_back(b); cls.push_back(c); (void)hipEventRecord(a, s); }
    void end() { (void)hipEventRecord(ev.back(), s); }
    void cancel() { (void)hipEventDestroy(ev.back()); ev.pop_back(); (void)hipEventDestroy(ev.back()); ev.pop_back(); cls.pop_back(); }   // begin() without a launch
};
// a `..._try` launcher (fused QKV + attention forms): timed as class cls_ when it takes the launch
#define LAUNCH_TRY(cls_, took_, try_expr, st_)      \
    do {                                            \
        if (prof) prof->begin(cls_);                \
        took_ = (try_expr);                         \
        if (prof) { if (took_) prof->end(); else prof->cancel(); } \
        if (took_ && (st_) != LDT_OK) return (st_); \
    } while (0)
#define LAUNCH(cls_, expr)                \
    do {                                  \
        if (prof) prof->begin(cls_);      \
        const int _rc = (expr);           \
        if (prof) prof->end();            \
        if (_rc != LDT_OK) return _rc;    \
    } while (0)

static int score_forward_impl(const ldt_score_plan* p, const float* x, float* eps_out, const int32_t* step_ptr,
                              hipStream_t s, Prof* prof) {
    TRY(check_plan(p));
    LDT_REQUIRE(x && eps_out, LDT_EARG, "score: null x/out");
    const int D = p->hidden, T = p->tokens, M = p->batch * p->tokens, F = p->mlp_hidden;
    const long sstr = p->mod_sample_stride, tstr = p->mod_step_stride;
    // LN folding (gemm_bf16.hip): with batch-shared modulation (unconditional sampling) the LayerNorm + modulate between a
    // residual GEMM and the next projection is folded into the two GEMMs' epilogues; the host supplies the per-step
    // S / C tables (plan->fold) when that pays (whole 256x256 tiles that fill the chip: Score.can_fold).
    // Small batches whose GEMMs all run the v1 kernels fold through those (statistics per 32 columns), the rest through the 256-tile kernel.
    const bool fold_v1 = ldt_gemm_lnfold_v1_route(M, D, F, p->gemm_wgs);
    bool fold = p->fold && p->stats && sstr == 0 && (M % 256 == 0 || fold_v1) && D % 256 == 0 && D <= 1024 && F % 256 == 0;
    for (int l = 0; l < p->blocks && fold; ++l) fold = !p->kv_cond[l];
    const int sparts = fold_v1 ? D / 32 : D / 256;              // row-statistics partials per row in plan->stats ([sparts][M][2])
    const long fstep = p->fold_step_stride, fblk = 6L * D + 2L * F;   // per block: S_qkv[3D] | C_qkv[3D] | S_up[F] | C_up[F]
    // ln_in (score.py:136-137): latents fp32 -> bf16 (K padded) -> X fp32
    LAUNCH(LDT_PROF_OTHER, ldt_cast_pad_launch(x, p->z_dim, BFM(p->xin), p->z_pad, M, p->z_dim, p->z_pad, s));
    {
        GemmArgs g{BF(p->xin), p->z_pad, BF(p->w_in), p->z_pad, p->b_in, p->X, D, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, 0, M, D, p->z_pad};
        g.max_wgs = p->gemm_wgs; LAUNCH(LDT_PROF_GEMM_IO, ldt_gemm_launch(LDT_EPI_F32, &g, s));
    }
    for (int l = 0; l < p->blocks; ++l) {                       // score.py:148-149, layers.py:212-219
        const float* m = p->mod + (long)l * 6 * D;              // shift_msa|scale_msa|gate_msa|shift_mlp|scale_mlp|gate_mlp
        const float* fl = fold ? p->fold + (long)l * fblk : nullptr;
        if (fold && l > 0) {                                    // Hb = x (1 + scale_msa) and the row statistics came from block l-1's mlp.out
            GemmArgs gq{BF(p->Hb), D, BF(p->w_qkv[l]), D, nullptr, p->QKV, 3L * D, nullptr, 0, nullptr, 0, nullptr, 0, 0, step_ptr, 0, M, 3 * D, D,
                        nullptr, 0, nullptr, 0, nullptr, p->stats, sparts, fl, fl + 3L * D, fstep};
            gq.max_wgs = p->gemm_wgs;
            // 32-token samples: projection + attention in one launch (gemm_mid.hip, mid_epilogue_attn): q | k | v never reach HBM
            gq.attn_o = BFM(p->Ob); gq.attn_scale_log2e = 1.4426950408889634f / sqrtf((float)(D / p->heads));
            int fst = LDT_OK;
            bool took = false;
            if (fold_v1) LAUNCH_TRY(LDT_PROF_GEMM_QKV, took, ldt_gemm_mid_qkv_attn_try(&gq, T, D / p->heads, true, s, &fst), fst);
            else LAUNCH_TRY(LDT_PROF_GEMM_QKV, took, ldt_gemm_qkv_attn256_try(&gq, T, D / p->heads, true, s, &fst), fst);   // 256-token samples: the 256 x 192-tile form of the persistent kernel
            if (!took) {
            LAUNCH(LDT_PROF_GEMM_QKV, ldt_gemm_lnfold_launch(LDT_EPI_BF16, &gq, s));
            AttnArgs at{BF(p->QKV), 3L * D, (long)T * 3 * D, BF(p->QKV) + D, 3L * D, (long)T * 3 * D, BF(p->QKV) + 2 * D, 3L * D,
                        BFM(p->Ob), p->batch, p->heads, T, T, 1.4426950408889634f / sqrtf((float)(D / p->heads))};
            LAUNCH(LDT_PROF_ATTN, ldt_attn_launch(&at, D / p->heads, s));
            }
        } else {
        LnArgs n1{p->X, D, BFM(p->Hb), D, nullptr, nullptr, m, m + D, sstr, T, step_ptr, tstr, M, D};
        LAUNCH(LDT_PROF_LN, ldt_ln_launch(&n1, s));
        if (p->kv_cond[l]) {                                    // cross-attention: q from the modulated x, K|V from the condition
            const int S = p->cond_tokens;
            GemmArgs gq{BF(p->Hb), D, BF(p->w_q[l]), D, p->b_q[l], p->QKV, 3L * D, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, 0, M, D, D};
            gq.max_wgs = p->gemm_wgs;
            gq.attn_o = BFM(p->Ob); gq.attn_scale_log2e = 1.4426950408889634f / sqrtf((float)(D / p->heads));
            gq.attn_k = BF(p->kv_cond[l]); gq.attn_v = BF(p->kv_cond[l]) + D; gq.attn_ldkv = 2L * D; gq.attn_kv_batch_stride = (long)S * 2 * D;
            int fst = LDT_OK;
            bool took = false;
            LAUNCH_TRY(LDT_PROF_GEMM_QKV, took, ldt_gemm_mid_q_xattn_try(&gq, T, S, D / p->heads, s, &fst), fst);   // 32 x 32 tokens: projection + attention in one launch
            if (!took) {
            LAUNCH(LDT_PROF_GEMM_QKV, ldt_gemm_launch(LDT_EPI_BF16, &gq, s));
            AttnArgs at{BF(p->QKV), 3L * D, (long)T * 3 * D, BF(p->kv_cond[l]), 2L * D, (long)S * 2 * D, BF(p->kv_cond[l]) + D, 2L * D,
                        BFM(p->Ob), p->batch, p->heads, T, S, 1.4426950408889634f / sqrtf((float)(D / p->heads))};
            LAUNCH(LDT_PROF_ATTN, ldt_attn_launch(&at, D / p->heads, s));
            }
        } else {                                                // self-attention: fused q|k|v projection of the modulated x
            GemmArgs gq{BF(p->Hb), D, BF(p->w_qkv[l]), D, p->b_qkv[l], p->QKV, 3L * D, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, 0, M, 3 * D, D};
            gq.max_wgs = p->gemm_wgs;
            gq.attn_o = BFM(p->Ob); gq.attn_scale_log2e = 1.4426950408889634f / sqrtf((float)(D / p->heads));
            int fst = LDT_OK;
            bool took = false;
            LAUNCH_TRY(LDT_PROF_GEMM_QKV, took, ldt_gemm_mid_qkv_attn_try(&gq, T, D / p->heads, false, s, &fst), fst);   // 32-token samples: projection + attention in one launch
            if (!took) LAUNCH_TRY(LDT_PROF_GEMM_QKV, took, ldt_gemm_qkv_attn256_try(&gq, T, D / p->heads, false, s, &fst), fst);   // 256-token samples
            if (!took) {
            LAUNCH(LDT_PROF_GEMM_QKV, ldt_gemm_launch(LDT_EPI_BF16, &gq, s));
            AttnArgs at{BF(p->QKV), 3L * D, (long)T * 3 * D, BF(p->QKV) + D, 3L * D, (long)T * 3 * D, BF(p->QKV) + 2 * D, 3L * D,
                        BFM(p->Ob), p->batch, p->heads, T, T, 1.4426950408889634f / sqrtf((float)(D / p->heads))};
            LAUNCH(LDT_PROF_ATTN, ldt_attn_launch(&at, D / p->heads, s));
            }
        }
        }
        if (fold) {
            // fc_o + gate + residual, also emitting Hb = x (1 + scale_mlp) and the row statistics; mlp.fc consumes them
            GemmArgs go{BF(p->Ob), D, BF(p->w_o[l]), D, p->b_o[l], p->X, D, p->X, D, nullptr, 0, m + 2 * D, sstr, T, step_ptr, tstr, M, D, D,
                        BFM(p->Hb), D, m + 4 * D, tstr, p->stats};
            go.max_wgs = p->gemm_wgs; go.stats_parts = sparts; LAUNCH(LDT_PROF_GEMM_O, ldt_gemm_lnfold_launch(LDT_EPI_RESID_F32, &go, s));
            if (p->fold_monitor) TRY(ldt_fold_monitor_launch(p->stats, sparts, M, D, p->fold_monitor, s));
            GemmArgs gu{BF(p->Hb), D, BF(p->w_up[l]), D, nullptr, p->U, F, nullptr, 0, nullptr, 0, nullptr, 0, 0, step_ptr, 0, M, F, D,
                        nullptr, 0, nullptr, 0, nullptr, p->stats, sparts, fl + 6L * D, fl + 6L * D + F, fstep};
            gu.max_wgs = p->gemm_wgs; LAUNCH(LDT_PROF_GEMM_GELU, ldt_gemm_lnfold_launch(LDT_EPI_GELU_BF16, &gu, s));
            GemmArgs gd{BF(p->U), F, BF(p->w_dn[l]), F, p->b_dn[l], p->X, D, p->X, D, nullptr, 0, m + 5 * D, sstr, T, step_ptr, tstr, M, D, F,
                        BFM(p->Hb), D, m + 6 * D + D, tstr, p->stats};      // next block's scale_msa
            gd.max_wgs = p->gemm_wgs; gd.stats_parts = sparts;
            if (l + 1 < p->blocks) {
                LAUNCH(LDT_PROF_GEMM_DN, ldt_gemm_lnfold_launch(LDT_EPI_RESID_F32, &gd, s));
                if (p->fold_monitor) TRY(ldt_fold_monitor_launch(p->stats, sparts, M, D, p->fold_monitor, s));
            }
            else LAUNCH(LDT_PROF_GEMM_DN, ldt_gemm_launch(LDT_EPI_RESID_F32, &gd, s));    // FinalLayer's LN runs as a kernel
            continue;
        }
        LnArgs n2{p->X, D, BFM(p->Hb), D, nullptr, nullptr, m + 3 * D, m + 4 * D, sstr, T, step_ptr, tstr, M, D};
        {
            GemmArgs go{BF(p->Ob), D, BF(p->w_o[l]), D, p->b_o[l], p->X, D, p->X, D, nullptr, 0, m + 2 * D, sstr, T, step_ptr, tstr, M, D, D};
            go.max_wgs = p->gemm_wgs; LAUNCH(LDT_PROF_GEMM_O, ldt_gemm_launch(LDT_EPI_RESID_F32, &go, s));
        }
        LAUNCH(LDT_PROF_LN, ldt_ln_launch(&n2, s));
        GemmArgs gu{BF(p->Hb), D, BF(p->w_up[l]), D, p->b_up[l], p->U, F, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, 0, M, F, D};
        gu.max_wgs = p->gemm_wgs; LAUNCH(LDT_PROF_GEMM_GELU, ldt_gemm_launch(LDT_EPI_GELU_BF16, &gu, s));
        {
            GemmArgs gd{BF(p->U), F, BF(p->w_dn[l]), F, p->b_dn[l], p->X, D, p->X, D, nullptr, 0, m + 5 * D, sstr, T, step_ptr, tstr, M, D, F};
            gd.max_wgs = p->gemm_wgs; LAUNCH(LDT_PROF_GEMM_DN, ldt_gemm_launch(LDT_EPI_RESID_F32, &gd, s));
        }
    }
    {                                                           // FinalLayer (layers.py:240-248)
        const float* m = p->mod + (long)p->blocks * 6 * D;
        LnArgs nf{p->X, D, BFM(p->Hb), D, nullptr, nullptr, m, m + D, sstr, T, step_ptr, tstr, M, D};
        LAUNCH(LDT_PROF_LN, ldt_ln_launch(&nf, s));
        GemmArgs gf{BF(p->Hb), D, BF(p->w_out), D, p->b_out, eps_out, p->z_dim, nullptr, 0, nullptr, 0, nullptr, 0, 0, nullptr, 0, M, p->z_dim, D};
        gf.max_wgs = p->gemm_wgs; LAUNCH(LDT_PROF_GEMM_IO, ldt_gemm_launch(LDT_EPI_F32, &gf, s));
    }
    return LDT_OK;
}

extern "C" int ldt_score_forward(const ldt_score_plan* p, const float* x, float* eps_out, const int32_t* step_ptr, void* stream) {
    return score_forward_impl(p, x, eps_out, step_ptr, ST(stream), nullptr);
}

extern "C" int ldt_score_forward_profile(const ldt_score_plan* p, const float* x, float* eps_out, const int32_t* step_ptr,
                                         float* ms_by_class, int32_t* launches_by_class, void* stream) {
    LDT_REQUIRE(ms_by_class && launches_by_class, LDT_EARG, "score_profile: null output");
    Prof prof{ST(stream), {}, {}};
    const int rc = score_forward_impl(p, x, eps_out, step_ptr, ST(stream), &prof);
    (void)hipStreamSynchronize(ST(stream));
    for (int c = 0; c < LDT_PROF_NCLASS; ++c) { ms_by_class[c] = 0.f; launches_by_class[c] = 0; }
    for (size_t i = 0; i < prof.cls.size(); ++i) {
        float ms = 0.f;
        if (rc == LDT_OK && hipEventElapsedTime(&ms, prof.ev[2 * i], prof.ev[2 * i + 1]) == hipSuccess) {
            ms_by_class[prof.cls[i]] += ms;
            launches_by_class[prof.cls[i]] += 1;
        }
        (void)hipEventDestroy(prof.ev[2 * i]);
        (void)hipEventDestroy(prof.ev[2 * i + 1]);
    }
    return rc;
}

// ------------------------------------------------------------------------------ reverse-SDE loop
static int enqueue_step(const ldt_score_plan* p, float* x, float* x_mean, float* eps_tmp, const float* coef, int mode,
                        const float* noise, long noise_step_stride, long elem_offset, uint64_t seed, int* step_counter,
                        const ldt_cond_args* cond, float* x_traj, hipStream_t s) {
    if (cond) {                                                 // per-sample AdaLN rows of this step
        TRY(ldt_cond_rows_launch(cond->temb, cond->extra, cond->c_buf, cond->c_buf_bf16, step_counter, p->batch, cond->t_dim, 1, s));   // c_buf = silu(c)
        if (cond->w_ada_bf16 && (cond->t_dim % 64 == 0 || !cond->w_ada)) {   // bf16 weight panel: half the bytes of the HBM-bound row GEMM (fp32 accumulation + bias); a width the MFMA GEMM does not take falls back to the fp32 rows when they were given
            GemmArgs g{};
            g.X = BF(cond->c_buf_bf16); g.ldx = cond->t_dim;
            g.W = BF(cond->w_ada_bf16); g.ldw = cond->t_dim;
            g.bias = cond->b_ada; g.out = cond->mod_buf; g.ldo = cond->n_mod;
            g.M = p->batch; g.N = cond->n_mod; g.K = cond->t_dim;
            g.max_wgs = p->gemm_wgs;
            TRY(ldt_gemm_launch(EPI_F32, &g, s));
        } else {
            SgemmArgs g{cond->c_buf, cond->t_dim, cond->w_ada, cond->t_dim, cond->b_ada, cond->mod_buf, cond->n_mod, 0, LDT_ACT_NONE,
                        LDT_ACT_NONE, p->batch, cond->n_mod, cond->t_dim};
            TRY(ldt_sgemm_launch(&g, s));
        }
    }
    TRY(score_forward_impl(p, x, eps_tmp, cond ? nullptr : step_counter, s, nullptr));
    const long n = (long)p->batch * p->tokens * p->z_dim;
    StepArgs st{x, eps_tmp, noise, x, x_mean, coef, step_counter, 0, mode, n, elem_offset, noise_step_stride,
                (uint32_t)seed, (uint32_t)(seed >> 32), 1, 0, x_traj};
    TRY(ldt_sampler_step_launch(&st, s));
    return ldt_advance_step_launch(step_counter, s);
}

extern "C" int ldt_sample_loop(const ldt_score_plan* p, float* x, float* x_mean, float* eps_tmp, const float* coef,
                               int32_t mode, const float* noise, int64_t noise_step_stride, int64_t elem_offset,
                               uint64_t seed, int32_t* step_counter, int32_t n_steps, const ldt_cond_args* cond,
                               float* x_traj, int32_t use_graph, void* stream) {
    TRY(check_plan(p));
    LDT_REQUIRE(x && x_mean && eps_tmp && coef && step_counter && n_steps > 0, LDT_EARG, "sample_loop: null pointer / n_steps");
    LDT_REQUIRE(!cond || !cond->w_ada_bf16 == !cond->c_buf_bf16, LDT_EARG, "sample_loop: w_ada_bf16 and c_buf_bf16 go together");
    LDT_REQUIRE(!cond || (cond->temb && (cond->w_ada || cond->w_ada_bf16) && cond->b_ada && cond->c_buf && cond->mod_buf && cond->mod_buf == p->mod &&
                          p->mod_sample_stride == cond->n_mod && cond->t_dim > 0), LDT_EARG,
                "sample_loop: inconsistent conditioning block (plan->mod must be cond->mod_buf with sample stride n_mod)");
    hipStream_t s = ST(stream);
    hipError_t e = hipMemsetAsync(step_counter, 0, sizeof(int), s);
    if (e != hipSuccess) { ldt_set_error("sample_loop: memset: %s", hipGetErrorString(e)); return (int)e; }
    // LN-fold monitor inside the loop (plan->fold_monitor, every plan->fold_monitor_every steps and on the last one): the other steps run a
    // copy of the plan without it
    ldt_score_plan plain = *p;
    plain.fold_monitor = nullptr;
    const bool mon = p->fold_monitor != nullptr;
    const int every = p->fold_monitor_every > 0 ? p->fold_monitor_every : 1;
    auto monitored = [&](int i) { return mon && (i % every == 0 || i == n_steps - 1); };
    if (!use_graph) {
        for (int i = 0; i < n_steps; ++i)
            TRY(enqueue_step(monitored(i) ? p : &plain, x, x_mean, eps_tmp, coef, mode, noise, noise_step_stride, elem_offset, seed, step_counter, cond, x_traj, s));
        return LDT_OK;
    }
    // One step captured, replayed n_steps times; every step-dependent operand is indexed by *step_counter.
    // Capture runs on a private non-blocking stream (the caller's may be the legacy default stream, which
    // cannot be captured), fenced against the caller's stream with events on both sides.
    hipStream_t gs = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipGraph_t graph = nullptr, graph_m = nullptr;
    hipGraphExec_t exec = nullptr, exec_m = nullptr;                     // the plain step and (fold monitor on) the monitored step
    int status = LDT_OK;
#define HIPTRY(call, what)                                                                   \
    do {                                                                                     \
        const hipError_t _e = (call);                                                        \
        if (_e != hipSuccess && status == LDT_OK) {                                          \
            ldt_set_error("sample_loop: %s: %s", what, hipGetErrorString(_e));               \
            status = (int)_e;                                                                \
        }                                                                                    \
    } while (0)
    HIPTRY(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking), "stream create");
    HIPTRY(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming), "event create");
    HIPTRY(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming), "event create");
    if (status == LDT_OK) {
        HIPTRY(hipEventRecord(ev_in, s), "event record");
        HIPTRY(hipStreamWaitEvent(gs, ev_in, 0), "stream wait");
    }
    if (status == LDT_OK) {
        for (int pass = 0; pass < (mon ? 2 : 1) && status == LDT_OK; ++pass) {      // pass 0: the plain step; pass 1: the monitored one
            HIPTRY(hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal), "begin capture");
            if (status == LDT_OK) {
                const int rc = enqueue_step(pass ? p : &plain, x, x_mean, eps_tmp, coef, mode, noise, noise_step_stride, elem_offset, seed, step_counter, cond, x_traj, gs);
                const hipError_t ee = hipStreamEndCapture(gs, pass ? &graph_m : &graph);
                if (rc != LDT_OK) status = rc;
                else if (ee != hipSuccess) { ldt_set_error("sample_loop: end capture: %s", hipGetErrorString(ee)); status = (int)ee; }
            }
        }
    }
    if (status == LDT_OK) HIPTRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "graph instantiate");
    if (status == LDT_OK && mon) HIPTRY(hipGraphInstantiate(&exec_m, graph_m, nullptr, nullptr, 0), "graph instantiate");
    for (int i = 0; i < n_steps && status == LDT_OK; ++i) HIPTRY(hipGraphLaunch(monitored(i) ? exec_m : exec, gs), "graph launch");
    if (gs) {
        if (status == LDT_OK) {
            HIPTRY(hipEventRecord(ev_out, gs), "event record");
            HIPTRY(hipStreamWaitEvent(s, ev_out, 0), "stream wait");
        }
        (void)hipStreamSynchronize(gs);          // the exec must outlive its in-flight launches
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (exec_m) (void)hipGraphExecDestroy(exec_m);
    if (graph) (void)hipGraphDestroy(graph);
    if (graph_m) (void)hipGraphDestroy(graph_m);
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    if (gs) (void)hipStreamDestroy(gs);
#undef HIPTRY
    return status;
}
