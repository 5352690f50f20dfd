// HBM-bound elementwise / row kernels of the LDT hot path (gfx950): cast+pad, LayerNorm+AdaLN
// modulate, the fused reverse-SDE update with in-kernel Philox noise, sinusoidal time embedding,
// and a plain fp32 tiled SGEMM for the tiny / precision-critical linears (time MLP, AdaLN tables,
// K<=64 convs).  All are coalesced 16-B-per-lane where the shape allows (guide G13).
#include <stdlib.h>

#include "kernels.h"

// ------------------------------------------------------------------------------------------------
// cast fp32 [rows][cols] (ld = lds) -> bf16 [rows][cols_pad] with zero padding (K padding for MFMA GEMM)
__global__ void cast_pad_kernel(const float* __restrict__ src, long lds, bf16_t* __restrict__ dst, long ldd,
                                long rows, int cols, int cols_pad) {
    const long total = rows * (long)(cols_pad / 4);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (cols_pad / 4);
        const int c = (int)(i % (cols_pad / 4)) * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (c + j < cols) ? src[r * lds + c + j] : 0.f;
        bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(dst + r * ldd + c) = pk;
    }
}

int ldt_cast_pad_launch(const float* src, long lds, bf16_t* dst, long ldd, long rows, int cols,
                                   int cols_pad, hipStream_t s) {
    LDT_REQUIRE(rows > 0 && cols > 0 && cols_pad >= cols && cols_pad % 4 == 0 && ldd % 4 == 0, LDT_ESHAPE,
                "cast_pad: bad shape rows=%ld cols=%d cols_pad=%d ldd=%ld", rows, cols, cols_pad, ldd);
    LDT_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 7) == 0, LDT_EALIGN, "cast_pad: dst must be 8-byte aligned");
    const long total = rows * (long)(cols_pad / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cast_pad_kernel, dim3(blocks), dim3(256), 0, s, src, lds, dst, ldd, rows, cols, cols_pad);
    return ldt_check_launch("cast_pad");
}

// ------------------------------------------------------------------------------------------------
// LayerNorm(eps=1e-6, biased var, over C) [* w + b] then modulate: y = ln * (1 + scale) + shift  -> bf16
//   reference: tools/utils.py:127-133 (LayerNorm wrapper), model/layers.py:136-137 (modulate), :218-219 (use)
// One wave per row; shift/scale are per-sample vectors (stride 0 = shared by the batch, the
// unconditional sampler's case: SURVEY hard part 3) selected by a device-side step counter.
template <int NV>   // NV float4 chunks per lane: C = NV*256
__global__ __launch_bounds__(256) void ln_mod_vec_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const float* xr = a.x + row * a.ldx;
    const long modoff = a.shift ? (a.step_ptr ? (long)(*a.step_ptr) * a.mod_step_stride : 0) + (row / a.rows_per_sample) * a.mod_sample_stride : 0;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        v[i] = *reinterpret_cast<const f32x4*>(xr + c);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)a.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)a.C + 1e-6f);
    const float* sh = a.shift; const float* sc = a.scale;
    if (sh) { sh += modoff; sc += modoff; }
    bf16_t* yr = a.y + row * a.ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd;
        if (a.w) { const f32x4 w = *reinterpret_cast<const f32x4*>(a.w + c); const f32x4 b = *reinterpret_cast<const f32x4*>(a.b + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = o[j] * w[j] + b[j]; }
        if (sh) { const f32x4 h = *reinterpret_cast<const f32x4*>(sh + c); const f32x4 g = *reinterpret_cast<const f32x4*>(sc + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = o[j] * (1.f + g[j]) + h[j]; }
        bf16x4 pk = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        *reinterpret_cast<bf16x4*>(yr + c) = pk;
    }
}

// generic C (any width): three passes over the (cache-resident) row
__global__ __launch_bounds__(256) void ln_mod_generic_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const float* xr = a.x + row * a.ldx;
    float s = 0.f;
    for (int c = lane; c < a.C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)a.C;
    float q = 0.f;
    for (int c = lane; c < a.C; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)a.C + 1e-6f);
    const float* sh = a.shift; const float* sc = a.scale;
    if (sh) {
        const long off = (a.step_ptr ? (long)(*a.step_ptr) * a.mod_step_stride : 0) +
                         (row / a.rows_per_sample) * a.mod_sample_stride;
        sh += off; sc += off;
    }
    bf16_t* yr = a.y + row * a.ldy;
    for (int c = lane; c < a.C; c += 64) {
        float o = (xr[c] - mean) * rstd;
        if (a.w) o = o * a.w[c] + a.b[c];
        if (sh) o = o * (1.f + sc[c]) + sh[c];
        yr[c] = (bf16_t)o;
    }
}

int ldt_ln_launch(const LnArgs* a, hipStream_t s) {
    LDT_REQUIRE(a->M > 0 && a->C > 0, LDT_ESHAPE, "ln: empty problem");
    LDT_REQUIRE((a->shift == nullptr) == (a->scale == nullptr), LDT_EARG, "ln: shift and scale go together");
    LDT_REQUIRE((a->w == nullptr) == (a->b == nullptr), LDT_EARG, "ln: affine weight and bias go together");
    LDT_REQUIRE(!a->shift || a->rows_per_sample > 0, LDT_EARG, "ln: rows_per_sample must be > 0 with modulation");
    dim3 grid((unsigned)((a->M + 3) / 4)), block(256);
    const bool vec = (a->C % 256 == 0) && a->C <= 1024 && a->ldx % 4 == 0 && a->ldy % 4 == 0 && ldt_aligned16(a->x) &&
                     (reinterpret_cast<uintptr_t>(a->y) & 7) == 0 && (!a->shift || (ldt_aligned16(a->shift) && ldt_aligned16(a->scale) &&
                     a->mod_sample_stride % 4 == 0 && a->mod_step_stride % 4 == 0)) && (!a->w || (ldt_aligned16(a->w) && ldt_aligned16(a->b)));
    if (vec) {
        switch (a->C / 256) {
            case 1: hipLaunchKernelGGL(ln_mod_vec_kernel<1>, grid, block, 0, s, *a); break;
            case 2: hipLaunchKernelGGL(ln_mod_vec_kernel<2>, grid, block, 0, s, *a); break;
            case 3: hipLaunchKernelGGL(ln_mod_vec_kernel<3>, grid, block, 0, s, *a); break;
            default: hipLaunchKernelGGL(ln_mod_vec_kernel<4>, grid, block, 0, s, *a); break;
        }
    } else {
        hipLaunchKernelGGL(ln_mod_generic_kernel, grid, block, 0, s, *a);
    }
    return ldt_check_launch("ln_modulate");
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011) + Box-Muller: 4 N(0,1) per counter.
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;   // (0,1]
    const float u2 = (float)b * 2.3283064365386963e-10f;            // [0,1)
    const float rr = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    z0 = rr * cs; z1 = rr * sn;
}

// Fused predictor update (one pass over the latents):
//   mode 0 (ancestral, exact op order of diffusion_continuous.py:152-162 + Latent_SDE_Trainer.py:57-61):
//       score = -params / std ; x_mean = (x + beta*score) / sqrt(1-beta) ; x = x_mean + sqrt(beta)*z
//       coef[step] = {beta, std, sqrt(1-beta), sqrt(beta)}   (host fp32 tables, SURVEY hard part 4)
//   mode 1 (folded; reversediffusion / eulermaruyama / ddim, :141-191):
//       x_mean = A*x + Bc*params ; x = x_mean + Cc*z        coef[step] = {A, Bc, Cc, 0}
// z comes from `noise` (parity mode: injected CPU draws) or from Philox keyed by
// (seed, stream id = step*philox_mul + philox_add, global element index) so that a sample's noise does not depend on how the batch is sharded.
__global__ __launch_bounds__(256) void sampler_step_kernel(const StepArgs a) {
    const int step = a.step_ptr ? *a.step_ptr : a.step_host;
    const f32x4 cf = *reinterpret_cast<const f32x4*>(a.coef + 4L * step);
    const float* nz = a.noise ? a.noise + (long)step * a.noise_step_stride : nullptr;
    const long nvec = a.n / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a.x + 4 * i);
        const f32x4 p = *reinterpret_cast<const f32x4*>(a.params + 4 * i);
        f32x4 z;
        if (nz) {
            z = *reinterpret_cast<const f32x4*>(nz + 4 * i);
        } else {
            const uint64_t e = (uint64_t)(a.elem_offset / 4 + i);
            uint32_t c[4] = {(uint32_t)e, (uint32_t)(e >> 32), (uint32_t)(step * a.philox_mul + a.philox_add), 0x4C445421u};
            philox4x32_10(c, a.seed_lo, a.seed_hi);
            float z0, z1, z2, z3;
            box_muller(c[0], c[1], z0, z1);
            box_muller(c[2], c[3], z2, z3);
            z = (f32x4){z0, z1, z2, z3};
        }
        f32x4 xm, xn;
        {
#pragma clang fp contract(off)      // the reference's separate mul/add/div roundings, no FMA (HIP's __f*_rn are plain ops)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (a.mode == 0) {
                    const float score = -p[j] / cf[1];
                    const float bs = cf[0] * score;
                    const float num = x[j] + bs;
                    xm[j] = num / cf[2];
                } else {
                    const float ax = cf[0] * x[j];
                    const float bp = cf[1] * p[j];
                    xm[j] = ax + bp;
                }
                const float cz = (a.mode == 0) ? cf[3] : cf[2];
                const float nz_ = cz * z[j];
                xn[j] = xm[j] + nz_;
            }
        }
        *reinterpret_cast<f32x4*>(a.x_out + 4 * i) = xn;
        if (a.traj) *reinterpret_cast<f32x4*>(a.traj + (long)step * a.n + 4 * i) = xn;
        if (a.x_mean_out) *reinterpret_cast<f32x4*>(a.x_mean_out + 4 * i) = xm;
    }
}

__global__ void advance_step_kernel(int* step_ptr) { if (threadIdx.x == 0 && blockIdx.x == 0) *step_ptr += 1; }

int ldt_sampler_step_launch(const StepArgs* a, hipStream_t s) {
    LDT_REQUIRE(a->n > 0 && a->n % 4 == 0 && a->elem_offset % 4 == 0, LDT_ESHAPE, "sampler_step: n and elem_offset must be multiples of 4 (n=%ld)", a->n);
    LDT_REQUIRE(a->x && a->params && a->x_out && a->coef, LDT_EARG, "sampler_step: null pointer");
    LDT_REQUIRE(ldt_aligned16(a->x) && ldt_aligned16(a->params) && ldt_aligned16(a->x_out) && ldt_aligned16(a->coef) &&
                (!a->noise || ldt_aligned16(a->noise)) && (!a->x_mean_out || ldt_aligned16(a->x_mean_out)) && a->noise_step_stride % 4 == 0,
                LDT_EALIGN, "sampler_step: buffers must be 16-byte aligned");
    LDT_REQUIRE(a->mode == 0 || a->mode == 1, LDT_EARG, "sampler_step: mode %d", a->mode);
    long blocks = (a->n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sampler_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *a);
    return ldt_check_launch("sampler_step");
}
int ldt_advance_step_launch(int* step_ptr, hipStream_t s) {
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, s, step_ptr);
    return ldt_check_launch("advance_step");
}

// standalone Philox normal fill (same stream as the fused step; used by tests and Compressor.sample(given_eps=None))
__global__ void philox_normal_kernel(float* out, long n, long elem_offset, int step, uint32_t k0, uint32_t k1) {
    const long nvec = n / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const uint64_t e = (uint64_t)(elem_offset / 4 + i);
        uint32_t c[4] = {(uint32_t)e, (uint32_t)(e >> 32), (uint32_t)step, 0x4C445421u};
        philox4x32_10(c, k0, k1);
        float z0, z1, z2, z3;
        box_muller(c[0], c[1], z0, z1);
        box_muller(c[2], c[3], z2, z3);
        *reinterpret_cast<f32x4*>(out + 4 * i) = (f32x4){z0, z1, z2, z3};
    }
}
int ldt_philox_normal_launch(float* out, long n, long elem_offset, int step, uint32_t k0, uint32_t k1, hipStream_t s) {
    LDT_REQUIRE(n > 0 && n % 4 == 0 && elem_offset % 4 == 0 && ldt_aligned16(out), LDT_ESHAPE, "philox_normal: n/offset %% 4, 16-byte aligned");
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, n, elem_offset, step, k0, k1);
    return ldt_check_launch("philox_normal");
}

// ------------------------------------------------------------------------------------------------
// sinusoidal embedding of continuous t (model/layers.py:20-36): e[i] = [sin(t_i f_k), cos(t_i f_k)];
// the frequency table f is built on the host with the reference's own fp32 expression (quirk Q5).
__global__ void sinusoid_kernel(const float* __restrict__ t, const float* __restrict__ freq, float* __restrict__ e, int n, int half) {
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        const float a = __fmul_rn(t[i], freq[k]);
        e[(long)i * 2 * half + k] = sinf(a);
        e[(long)i * 2 * half + half + k] = cosf(a);
    }
}
int ldt_sinusoid_launch(const float* t, const float* freq, float* e, int n, int half, hipStream_t s) {
    LDT_REQUIRE(n > 0 && half > 0, LDT_ESHAPE, "sinusoid: empty");
    hipLaunchKernelGGL(sinusoid_kernel, dim3(n), dim3(128), 0, s, t, freq, e, n, half);
    return ldt_check_launch("sinusoid");
}

// ------------------------------------------------------------------------------------------------
// fp32 SGEMM (NT): C[M,N] = act_out( act_in(A[M,K]) · B[N,K]^T + bias[N] ), output fp32 or bf16.
// 64x64 tile, BK=16, 4x4 outputs per thread.  For the small fp32-critical linears: time MLP
// (layers.py:17), AdaLN tables (layers.py:172,238), convs with K<=64, MiniPointnet, prior heads.
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_SILU: return v / (1.0f + expf(-v));
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_GELU: return gelu_erf(v);
        default: return v;
    }
}
__global__ __launch_bounds__(256) void sgemm_nt_kernel(const SgemmArgs a) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    // each thread stages 4 consecutive k of one A row and one B row: one 16-B load each when the rows allow it
    const bool vec = (a.lda % 4 == 0) && (a.ldb % 4 == 0) && (a.K % 4 == 0) && ldt_aligned16(a.A) && ldt_aligned16(a.B);
    for (int k0 = 0; k0 < a.K; k0 += 16) {
        const int m = m0 + lr, n = n0 + lr, kk = k0 + lk;
        f32x4 va4 = {0.f, 0.f, 0.f, 0.f}, vb4 = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            if (m < a.M && kk < a.K) va4 = *reinterpret_cast<const f32x4*>(a.A + (long)m * a.lda + kk);
            if (n < a.N && kk < a.K) vb4 = *reinterpret_cast<const f32x4*>(a.B + (long)n * a.ldb + kk);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (m < a.M && kk + j < a.K) va4[j] = a.A[(long)m * a.lda + kk + j];
                if (n < a.N && kk + j < a.K) vb4[j] = a.B[(long)n * a.ldb + kk + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[lk + j][lr] = a.act_in ? apply_act(va4[j], a.act_in) : va4[j];
            Bs[lk + j][lr] = vb4[j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&As[k][ty * 4]);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&Bs[k][tx * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= a.N) continue;
            float v = acc[i][j] + (a.bias ? a.bias[n] : 0.f);
            v = apply_act(v, a.act_out);
            if (a.out_bf16) reinterpret_cast<bf16_t*>(a.C)[(long)m * a.ldc + n] = (bf16_t)v;
            else reinterpret_cast<float*>(a.C)[(long)m * a.ldc + n] = v;
        }
    }
}
int ldt_sgemm_launch(const SgemmArgs* a, hipStream_t s) {
    LDT_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, LDT_ESHAPE, "sgemm: empty problem");
    LDT_REQUIRE(a->A && a->B && a->C, LDT_EARG, "sgemm: null pointer");
    static const bool no_mfma = getenv("LDT_SGEMM_VALU") != nullptr;      // tools/dbg A/B: force the scalar-FMA kernel
    static const bool no_skinny = getenv("LDT_SGEMM_SKINNY") && atoi(getenv("LDT_SGEMM_SKINNY")) == 0;
    int st = LDT_OK;
    if (!no_skinny && ldt_skinny_linear_try(a, s, &st)) return st;      // millions of rows x (K <= 32 | N <= 8): streaming forms (skinny_linear.hip)
    if (!no_mfma && ldt_sgemm_mfma_try(a, s, &st)) return st;            // fp32-input MFMA kernel (sgemm_mfma.hip) when rows are 16-B aligned, K % 16 == 0
    dim3 grid((a->N + 63) / 64, (a->M + 63) / 64), block(256);
    LDT_REQUIRE(grid.y < 65536, LDT_ESHAPE, "sgemm: M too large for this kernel (M=%d)", a->M);
    hipLaunchKernelGGL(sgemm_nt_kernel, grid, block, 0, s, *a);
    return ldt_check_launch("sgemm_nt");
}

// c[b][k] = silu?(temb[step][k] + extra[b][k])   (score.py:135: c = t_emb + l_emb | img condition; the SiLU that opens every AdaLN
// Sequential — model/layers.py:172 — is applied here once so that the row GEMM behind it can stream its A operand by LDS-DMA), step from the device counter
// `cb` (optional): the same rows rounded to bf16 — the A operand of the bf16-weight form of the row GEMM (ldt_cond_args.w_ada_bf16)
__global__ void cond_rows_kernel(const float* __restrict__ temb, const float* __restrict__ extra, float* __restrict__ c, bf16_t* __restrict__ cb,
                                 const int* __restrict__ step_ptr, int batch, int t_dim, int silu_out) {
    const int step = step_ptr ? *step_ptr : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < batch * t_dim; i += gridDim.x * blockDim.x) {
        const int k = i % t_dim;
        const float v = temb[(long)step * t_dim + k] + (extra ? extra[i] : 0.f);
        const float o = silu_out ? silu(v) : v;
        c[i] = o;
        if (cb) cb[i] = (bf16_t)o;
    }
}
int ldt_cond_rows_launch(const float* temb, const float* extra, float* c, void* c_bf16, const int* step_ptr, int batch, int t_dim, int silu_out, hipStream_t s) {
    LDT_REQUIRE(temb && c && batch > 0 && t_dim > 0, LDT_EARG, "cond_rows: bad arguments");
    int blocks = (batch * t_dim + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(cond_rows_kernel, dim3(blocks), dim3(256), 0, s, temb, extra, c, reinterpret_cast<bf16_t*>(c_bf16), step_ptr, batch, t_dim, silu_out);
    return ldt_check_launch("cond_rows");
}
