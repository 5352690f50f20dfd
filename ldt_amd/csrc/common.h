// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the LDT hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDT_WAVE 64

// status codes of the C-ABI (include/ldt_hip.h)
#define LDT_OK 0
#define LDT_EARG (-1)
#define LDT_ESHAPE (-2)
#define LDT_EALIGN (-3)

void ldt_set_error(const char* fmt, ...);
int ldt_check_launch(const char* what);   // hipGetLastError -> status (+ message)

#define LDT_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            ldt_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)

static inline bool ldt_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU (nn.GELU() default; model/layers.py:111-113 via tools/utils.py:107-108).
// gelu_erf: libm erff (fp32-accurate; used by the fp32 SGEMM path).
// gelu_erf_fast: GEMM epilogue form whose output is rounded to bf16 anyway — erf by Abramowitz & Stegun 7.1.26
// (|abs err| <= 1.5e-7, i.e. ~2^-13 of a bf16 ulp at |x|~1), branch-free: 1 rcp + 1 exp + 8 FMA instead of erff.
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_z = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // erfc(|x|/sqrt2)
    const float phi = 0.5f * erfc_z;                                                    // Phi(-|x|)
    return x * (x >= 0.f ? 1.0f - phi : phi);
}
// two-lane form of gelu_erf_fast for VALU-bound epilogues: the polynomial runs on v_pk_fma_f32 / v_pk_mul_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 z = ax * 0.70710678118654752440f;
    const f32x2 den = z * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2 p = t * 1.061405429f + (-1.453152027f);
    p = p * t + 1.421413741f;
    p = p * t + (-0.284496736f);
    p = p * t + 0.254829592f;
    const f32x2 e = z * z * (-1.4426950408889634f);
    const f32x2 ex = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
    // x Phi(x) = x/2 + |x| (1/2 - Phi(-|x|)),  Phi(-|x|) = erfc(z)/2 = p t ex / 2 : no compare/select
    const f32x2 r = (p * t) * ex * (-0.5f) + 0.5f;
    return x * 0.5f + ax * r;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
