// Device ops of the remaining samplers around the Score forward: LangevinCorrector (diffusion/diffusion_continuous.py:193-210)
// and PNDM (:260-316).  All are HBM-streaming passes over the latents [B][tokens*z] fp32 — a few MB, far below the
// Score forward they sit next to — written for determinism (fixed reduction order, no atomics) rather than speed.
#include "../../include/ldt_hip.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------
// per-sample L2 norms (torch.norm(v.reshape(B, -1), dim=-1), :204-205): one 256-thread workgroup per sample,
// 16-B loads, wave shuffle + LDS tree in a fixed order.
__global__ __launch_bounds__(256) void batch_norms_kernel(const float* __restrict__ x, long per, float* __restrict__ out) {
    const float* row = x + (long)blockIdx.x * per;
    float acc = 0.f;
    for (long i = threadIdx.x; i < per / 4; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * i);
        acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sqrtf((part[0] + part[1]) + (part[2] + part[3]));
}
// sum over the batch of the per-sample norms (the numerator of .mean()): one wave, fixed order
__global__ __launch_bounds__(64) void norm_sum_kernel(const float* __restrict__ norms, int B, float* __restrict__ sum_out) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += 64) acc += norms[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) *sum_out = acc;
}

extern "C" int ldt_batch_norm_sum(const float* x, int32_t B, int64_t per_sample, float* norms_scratch, float* sum_out, void* stream) {
    LDT_REQUIRE(x && norms_scratch && sum_out, LDT_EARG, "batch_norm_sum: null pointer");
    LDT_REQUIRE(B > 0 && per_sample > 0 && per_sample % 4 == 0 && ldt_aligned16(x), LDT_ESHAPE,
                "batch_norm_sum: per_sample=%ld must be a multiple of 4, x 16-byte aligned", (long)per_sample);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(batch_norms_kernel, dim3((unsigned)B), dim3(256), 0, s, x, (long)per_sample, norms_scratch);
    hipLaunchKernelGGL(norm_sum_kernel, dim3(1), dim3(64), 0, s, norms_scratch, B, sum_out);
    return ldt_check_launch("batch_norm_sum");
}

// Langevin step size -> one coefficient row {A, B, C, 0} for ldt_sampler_step(mode 1) (x_mean = A x + B params, x = x_mean + C z):
//   grad = score = -params / std  =>  grad_norm = mean_b ||params_b|| / std ;  noise_norm = mean_b ||z_b||
//   step_size = (snr * noise_norm / grad_norm)^2 * 2 * alpha,  alpha = 1 (the reference's class test at :195 is never true)
//   x_mean = x + step_size * grad = x - (step_size / std) params ;  x = x_mean + sqrt(2 step_size) z
// sums[0] = sum_b ||params_b||, sums[1] = sum_b ||z_b|| over the WHOLE batch (all-reduced by the caller when sharded).
__global__ void langevin_coef_kernel(const float* __restrict__ sums, int n_total, float snr, float std_t, float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float grad_norm = (sums[0] / (float)n_total) / std_t;
    const float noise_norm = sums[1] / (float)n_total;
    const float r = snr * noise_norm / grad_norm;
    const float step = r * r * 2.0f;
    coef[0] = 1.0f; coef[1] = -step / std_t; coef[2] = sqrtf(step * 2.0f); coef[3] = 0.f;
}
extern "C" int ldt_langevin_coef(const float* sums, int32_t n_total, float snr, float std_t, float* coef_out, void* stream) {
    LDT_REQUIRE(sums && coef_out && n_total > 0 && std_t > 0.f, LDT_EARG, "langevin_coef: bad argument");
    hipLaunchKernelGGL(langevin_coef_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sums, n_total, snr, std_t, coef_out);
    return ldt_check_launch("langevin_coef");
}

// ------------------------------------------------------------------------------------------------
// PNDM (:260-316).  transfer(): x_next = x + d * (p * x - q * et) with the three schedule scalars of :267-271 formed by
// the host in fp32 (they are batch-uniform: every sample is at the same t); the element-wise op order is the reference's.
__global__ __launch_bounds__(256) void pndm_transfer_kernel(const float* __restrict__ x, const float* __restrict__ et, float d, float p,
                                                            float q, float* __restrict__ out, long nvec) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + 4 * i);
        const f32x4 ev = *reinterpret_cast<const f32x4*>(et + 4 * i);
        f32x4 o;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = p * xv[j];
                const float b = q * ev[j];
                const float dl = d * (a - b);
                o[j] = xv[j] + dl;
            }
        }
        *reinterpret_cast<f32x4*>(out + 4 * i) = o;
    }
}
extern "C" int ldt_pndm_transfer(const float* x, const float* et, float d, float p, float q, float* out, int64_t n, void* stream) {
    LDT_REQUIRE(x && et && out, LDT_EARG, "pndm_transfer: null pointer");
    LDT_REQUIRE(n > 0 && n % 4 == 0 && ldt_aligned16(x) && ldt_aligned16(et) && ldt_aligned16(out), LDT_EALIGN,
                "pndm_transfer: n %% 4 and 16-byte aligned buffers");
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pndm_transfer_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, et, d, p, q, out, n / 4);
    return ldt_check_launch("pndm_transfer");
}

// out = s * (((c0 a0 + c1 a1) + c2 a2) + c3 a3): the Runge-Kutta average (1/6)(e1 + 2 e2 + 2 e3 + e4) (:291) and the
// 4-step linear multistep combination (1/24)(55 e[-1] - 59 e[-2] + 37 e[-3] - 9 e[-4]) (:300), left to right as torch does.
struct Lin4Args { const float* a[4]; float c[4]; float s; float* out; long nvec; };
__global__ __launch_bounds__(256) void lincomb4_kernel(const Lin4Args g) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < g.nvec; i += (long)gridDim.x * 256) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(g.a[k] + 4 * i);
        f32x4 o;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t0 = g.c[0] * v[0][j];
                const float t1 = g.c[1] * v[1][j];
                const float t2 = g.c[2] * v[2][j];
                const float t3 = g.c[3] * v[3][j];
                const float acc = ((t0 + t1) + t2) + t3;
                o[j] = g.s * acc;
            }
        }
        *reinterpret_cast<f32x4*>(g.out + 4 * i) = o;
    }
}
extern "C" int ldt_lincomb4(const float* a0, const float* a1, const float* a2, const float* a3, float c0, float c1, float c2, float c3,
                            float s, float* out, int64_t n, void* stream) {
    LDT_REQUIRE(a0 && a1 && a2 && a3 && out, LDT_EARG, "lincomb4: null pointer");
    LDT_REQUIRE(n > 0 && n % 4 == 0 && ldt_aligned16(a0) && ldt_aligned16(a1) && ldt_aligned16(a2) && ldt_aligned16(a3) && ldt_aligned16(out),
                LDT_EALIGN, "lincomb4: n %% 4 and 16-byte aligned buffers");
    Lin4Args g{{a0, a1, a2, a3}, {c0, c1, c2, c3}, s, out, n / 4};
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lincomb4_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g);
    return ldt_check_launch("lincomb4");
}

// ------------------------------------------------------------------------------------------------
// score = -params / sqrt(var(t)) of Trainer.score_fn (trainer/Latent_SDE_Trainer.py:57-61) with var(t) of the VP-SDE
// (diffusion/diffusion_continuous.py:649-651) evaluated per sample in fp32 in the reference's op order:
//   var = 1 - (1 - sigma2_0) * exp(-beta0 t - 0.5 (beta1 - beta0) t t)
__global__ __launch_bounds__(256) void vpsde_score_kernel(const float* __restrict__ params, const float* __restrict__ t, float beta0, float beta1,
                                                          float sigma2_0, float* __restrict__ out, long per4) {
    const float tb = t[blockIdx.y];
    float sd;
    {
#pragma clang fp contract(off)
        const float a = -beta0 * tb;
        const float b = (0.5f * (beta1 - beta0)) * tb * tb;
        const float var = 1.0f - (1.0f - sigma2_0) * expf(a - b);
        sd = sqrtf(var);
    }
    const float* src = params + (long)blockIdx.y * per4 * 4;
    float* dst = out + (long)blockIdx.y * per4 * 4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < per4; i += (long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = -v[j] / sd;
        *reinterpret_cast<f32x4*>(dst + 4 * i) = o;
    }
}
extern "C" int ldt_vpsde_score(const float* params, const float* t, float beta0, float beta1, float sigma2_0, float* out, int32_t B,
                               int64_t per_sample, void* stream) {
    LDT_REQUIRE(params && t && out, LDT_EARG, "vpsde_score: null pointer");
    LDT_REQUIRE(B > 0 && B <= 65535 && per_sample > 0 && per_sample % 4 == 0 && ldt_aligned16(params) && ldt_aligned16(out), LDT_ESHAPE,
                "vpsde_score: B=%d in [1, 65535], per_sample=%ld a multiple of 4, 16-byte aligned buffers", B, (long)per_sample);
    long bx = (per_sample / 4 + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(vpsde_score_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params, t,
                       beta0, beta1, sigma2_0, out, (long)(per_sample / 4));
    return ldt_check_launch("vpsde_score");
}

// The same score for the other SDE families of diffusion/diffusion_continuous.py (make_diffusion :18-29): kind 1 = sub-VP
// (:705-707: var = (1 - e)^2 + sigma2_0 e, e = exp(-beta0 t - (beta1 - beta0) t^2 / 2); c = beta0, beta1, sigma2_0),
// kind 2 = VE and geometric (:746-747, :615-616: var = smin ratio^t - smin + sigma2_0; c = smin, ratio = smax / smin, sigma2_0),
// kind 0 = the VP form above.  fp32, the reference's op order.
__global__ __launch_bounds__(256) void sde_score_kernel(const float* __restrict__ params, const float* __restrict__ t, int kind, float c0, float c1,
                                                        float c2, float* __restrict__ out, long per4) {
    const float tb = t[blockIdx.y];
    float sd;
    {
#pragma clang fp contract(off)
        float var;
        if (kind == 2) {
            var = c0 * powf(c1, tb) - c0 + c2;
        } else {
            const float a = -c0 * tb;
            const float b = (0.5f * (c1 - c0)) * tb * tb;
            const float e = expf(a - b);
            if (kind == 1) {
                const float om = 1.0f - e;
                var = om * om + c2 * e;
            } else {
                var = 1.0f - (1.0f - c2) * e;
            }
        }
        sd = sqrtf(var);
    }
    const float* src = params + (long)blockIdx.y * per4 * 4;
    float* dst = out + (long)blockIdx.y * per4 * 4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < per4; i += (long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = -v[j] / sd;
        *reinterpret_cast<f32x4*>(dst + 4 * i) = o;
    }
}
extern "C" int ldt_sde_score(const float* params, const float* t, int32_t kind, float c0, float c1, float c2, float* out, int32_t B,
                             int64_t per_sample, void* stream) {
    LDT_REQUIRE(params && t && out, LDT_EARG, "sde_score: null pointer");
    LDT_REQUIRE(kind >= 0 && kind <= 2, LDT_EARG, "sde_score: kind=%d (0 vpsde, 1 sub_vpsde, 2 vesde / geometric_sde)", kind);
    LDT_REQUIRE(B > 0 && B <= 65535 && per_sample > 0 && per_sample % 4 == 0 && ldt_aligned16(params) && ldt_aligned16(out), LDT_ESHAPE,
                "sde_score: B=%d in [1, 65535], per_sample=%ld a multiple of 4, 16-byte aligned buffers", B, (long)per_sample);
    long bx = (per_sample / 4 + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(sde_score_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params, t,
                       (int)kind, c0, c1, c2, out, (long)(per_sample / 4));
    return ldt_check_launch("sde_score");
}

// x = act(x) in place on bf16 rows [M][C] (row stride ld): the block activation of the reference's no-condition ResidualBlock branches
// (`decoder_act`, model/layers.py:224-226 via tools/utils.py:104-124), applied to the LayerNorm output the projections read.  kind = enum
// ldt_block_act; the arithmetic runs in fp32.  rrelu is its eval-mode form (slope (1/8 + 1/3) / 2).
__device__ __forceinline__ float block_act(float v, int kind) {
    switch (kind) {
        case LDT_BACT_GELU: return gelu_erf(v);
        case LDT_BACT_SILU: return silu(v);
        case LDT_BACT_RELU: return fmaxf(v, 0.f);
        case LDT_BACT_LEAKY_001: return v > 0.f ? v : 0.01f * v;
        case LDT_BACT_LEAKY_02: return v > 0.f ? v : 0.2f * v;
        case LDT_BACT_RRELU_EVAL: return v > 0.f ? v : v * ((1.0f / 8.0f + 1.0f / 3.0f) * 0.5f);
        case LDT_BACT_HARDSWISH: return v * fminf(fmaxf(v + 3.0f, 0.f), 6.0f) * (1.0f / 6.0f);
        case LDT_BACT_SELU: return 1.0507009873554804934193349852946f * (v > 0.f ? v : 1.6732632423543772848170429916717f * (expf(v) - 1.0f));
        default: return v;
    }
}
__global__ __launch_bounds__(256) void block_act_kernel(bf16_t* __restrict__ x, long ld, long M, int C, int kind) {
    const long n = M * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        bf16_t* p = x + (i / C) * ld + (i % C);
        *p = (bf16_t)block_act((float)*p, kind);
    }
}
extern "C" int ldt_block_activation(uint16_t* x, int64_t ld, int64_t M, int32_t C, int32_t kind, void* stream) {
    LDT_REQUIRE(x && M > 0 && C > 0 && ld >= C, LDT_EARG, "block_activation: bad argument");
    LDT_REQUIRE(kind > LDT_BACT_NONE && kind <= LDT_BACT_SELU, LDT_EARG, "block_activation: unknown activation %d", kind);
    long blocks = (M * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(block_act_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<bf16_t*>(x), (long)ld, (long)M, (int)C, (int)kind);
    return ldt_check_launch("block_activation");
}

// out = a + b (fp32; c = t_emb + label / image-condition embedding, model/scorenet/score.py:135).  out may alias a or b.
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a[i] + b[i];
}
extern "C" int ldt_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    LDT_REQUIRE(a && b && out && n > 0, LDT_EARG, "add_f32: bad argument");
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(add_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, out, (long)n);
    return ldt_check_launch("add_f32");
}

// bf16 -> fp32 widening of a packed weight panel (exact), for the fp32 table builds that must see the SAME rounded weights
// the MFMAs multiply by (Score.fold_table).
__global__ __launch_bounds__(256) void widen_bf16_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (float)src[i];
}
extern "C" int ldt_widen_bf16(const uint16_t* src, float* dst, int64_t n, void* stream) {
    LDT_REQUIRE(src && dst && n > 0, LDT_EARG, "widen_bf16: bad argument");
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(widen_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16_t*>(src), dst, (long)n);
    return ldt_check_launch("widen_bf16");
}

// ------------------------------------------------------------------------------------------------
// LN-folding monitor: max over rows of mean^2 / variance from the producer's row statistics stats[parts][M][2] (partial
// (sum, sum of squares) per 256-column tile).  The folded projections round x (1 + scale) to bf16 BEFORE the mean is removed,
// so their error variance is (1 + mean^2 / variance) x the LayerNorm kernel's; the sampler reads this once per
// sample() call and falls back to the LayerNorm kernels when the ratio says the 1e-4 parity bar is at risk.
// One workgroup of 256 threads, fixed order: deterministic.  *out = the maximum ratio (0 when M == 0).
__global__ __launch_bounds__(256) void fold_mean_ratio_kernel(const float* __restrict__ stats, int parts, long M, int K, float* __restrict__ out) {
    float best = 0.f;
    const float invk = 1.0f / (float)K;
    for (long r = threadIdx.x; r < M; r += 256) {
        float s1 = 0.f, s2 = 0.f;
        for (int p = 0; p < parts; ++p) { s1 += stats[((long)p * M + r) * 2]; s2 += stats[((long)p * M + r) * 2 + 1]; }
        const float mean = s1 * invk;
        const float var = fmaxf(s2 * invk - mean * mean, 0.f) + 1e-6f;
        best = fmaxf(best, mean * mean / var);
    }
    best = wave_max(best);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) *out = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}
// the same, accumulated into *out by atomic max (ratios are non-negative: their bit patterns order like ints); the Score
// forward calls it after every folded residual GEMM when plan->fold_monitor is set
__global__ __launch_bounds__(256) void fold_monitor_kernel(const float* __restrict__ stats, int parts, long M, int K, float* __restrict__ out) {
    float best = 0.f;
    const float invk = 1.0f / (float)K;
    for (long r = blockIdx.x * 256L + threadIdx.x; r < M; r += (long)gridDim.x * 256) {
        float s1 = 0.f, s2 = 0.f;
        for (int p = 0; p < parts; ++p) { s1 += stats[((long)p * M + r) * 2]; s2 += stats[((long)p * M + r) * 2 + 1]; }
        const float mean = s1 * invk;
        const float var = fmaxf(s2 * invk - mean * mean, 0.f) + 1e-6f;
        best = fmaxf(best, mean * mean / var);
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(out), __float_as_int(best));
}
int ldt_fold_monitor_launch(const float* stats, int parts, long M, int K, float* out, hipStream_t s) {
    long blocks = (M + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(fold_monitor_kernel, dim3((unsigned)blocks), dim3(256), 0, s, stats, parts, M, K, out);
    return ldt_check_launch("fold_monitor");
}

extern "C" int ldt_fold_mean_ratio(const float* stats, int32_t parts, int64_t M, int32_t K, float* out, void* stream) {
    LDT_REQUIRE(stats && out && parts >= 1 && M > 0 && K > 0, LDT_EARG, "fold_mean_ratio: bad argument");
    hipLaunchKernelGGL(fold_mean_ratio_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), stats, parts, (long)M, K, out);
    return ldt_check_launch("fold_mean_ratio");
}


// ------------------------------------------------------------------------------------------------- group norm / identity norm
// tools/utils.py:168-181 get_norm: `group_norm` -> nn.GroupNorm(min(C/4, 16), C, eps=1e-6) on the channels-first activations (statistics per
// sample and group over C/G channels x all tokens), `None` -> Identity.  Rows are token-major here: x[b*T + t][c].
__global__ void group_stats_kernel(const float* __restrict__ x, long ldx, int T, int C, int G, float eps, float* __restrict__ stats) {
    const int b = blockIdx.x / G, g = blockIdx.x % G, cg = C / G;
    const float* xb = x + (long)b * T * ldx + g * cg;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < T * cg; i += blockDim.x) {
        const float v = xb[(long)(i / cg) * ldx + i % cg];
        s1 += v; s2 += (double)v * v;
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)T * cg, mean = r1[0] / n;
        double var = r2[0] / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        stats[(long)blockIdx.x * 2] = (float)mean;
        stats[(long)blockIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
extern "C" int ldt_group_stats(const float* x, int64_t ldx, int32_t B, int32_t T, int32_t C, int32_t G, float eps, float* stats, void* stream) {
    LDT_REQUIRE(x && stats && B > 0 && T > 0 && C > 0 && G > 0 && C % G == 0 && ldx >= C, LDT_EARG, "group_stats: bad argument (B=%d T=%d C=%d G=%d)", B, T, C, G);
    hipLaunchKernelGGL(group_stats_kernel, dim3((unsigned)(B * G)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, T, C, G, eps, stats);
    return ldt_check_launch("group_stats");
}

__global__ void norm_apply_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, long M, int C, const float* __restrict__ stats,
                                  int G, int rows_per_stat, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ shift,
                                  const float* __restrict__ scale, long mod_sample_stride, int rows_per_sample) {
    const long n4 = M * (C / 4);
    const int cg = C / (G > 0 ? G : 1);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long m = i / (C / 4);
        const int c = (int)(i % (C / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ldx + c);
        const long moff = (m / rows_per_sample) * mod_sample_stride + c;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float mean = 0.f, rstd = 1.f;
            if (stats) {
                const float* st = stats + ((m / rows_per_stat) * G + (c + j) / cg) * 2;
                mean = st[0]; rstd = st[1];
            }
            float h = (v[j] - mean) * rstd;
            if (w) h = h * w[c + j] + b[c + j];
            if (scale) h = h * (1.0f + scale[moff + j]) + shift[moff + j];
            o[j] = (bf16_t)h;
        }
        *reinterpret_cast<bf16x4*>(y + m * ldy + c) = o;
    }
}
extern "C" int ldt_norm_apply(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int64_t M, int32_t C, const float* stats, int32_t G,
                              int32_t rows_per_stat, const float* w, const float* b, const float* shift, const float* scale,
                              int64_t mod_sample_stride, int32_t rows_per_sample, void* stream) {
    LDT_REQUIRE(x && y && M > 0 && C > 0 && C % 4 == 0 && ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0, LDT_EARG, "norm_apply: bad argument (M=%ld C=%d)", (long)M, C);
    LDT_REQUIRE(ldt_aligned16(x) && (reinterpret_cast<uintptr_t>(y) & 7u) == 0, LDT_EALIGN, "norm_apply: x must be 16-byte, y 8-byte aligned");
    LDT_REQUIRE(!stats || (G > 0 && C % G == 0 && rows_per_stat > 0), LDT_EARG, "norm_apply: statistics need G | C and rows_per_stat > 0");
    LDT_REQUIRE(!w == !b && !shift == !scale && rows_per_sample > 0, LDT_EARG, "norm_apply: w / b and shift / scale go in pairs; rows_per_sample > 0");
    long blocks = (M * (C / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(norm_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, reinterpret_cast<bf16_t*>(y), (long)ldy,
                       (long)M, C, stats, stats ? G : 1, stats ? rows_per_stat : 1, w, b, shift, scale, (long)mod_sample_stride, rows_per_sample);
    return ldt_check_launch("norm_apply");
}
