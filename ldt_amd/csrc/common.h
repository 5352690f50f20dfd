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

// Raise a kernel's dynamic-LDS limit once per (call site, device): thread-safe (sub-batch streams call the launchers from
// several host threads) and per device (the attribute belongs to the device current at the call).
#include <atomic>
#define LDT_ENSURE_LDS(kernel_ptr, bytes, what)                                                                       \
    do {                                                                                                              \
        static std::atomic<unsigned long long> _lds_done{0};                                                          \
        int _dev = 0;                                                                                                 \
        (void)hipGetDevice(&_dev);                                                                                    \
        const unsigned long long _bit = 1ull << (_dev & 63);                                                          \
        if (!(_lds_done.load(std::memory_order_acquire) & _bit)) {                                                    \
            const hipError_t _e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr),                      \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));           \
            if (_e != hipSuccess) { ldt_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(_e)); return (int)_e; } \
            _lds_done.fetch_or(_bit, std::memory_order_release);                                                      \
        }                                                                                                             \
    } while (0)

__host__ __device__ static inline bool ldt_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Order-independent accumulation of fp64 partial sums across workgroups (LocalGrouper statistics): the partials are added as
// 64-bit fixed-point integers (2^-20 resolution per wave partial, |sum| < 2^43 ~ 8.8e12: integer adds commute, so the result does not depend on the
// arrival order of the atomics — two runs of the same input are bit-identical), and read back as doubles.
#define LDT_FX_SCALE 1048576.0               /* 2^20 */
__device__ __forceinline__ void fx_atomic_add(double* slot, double v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double2ll_rn(v * LDT_FX_SCALE));
}
__device__ __forceinline__ double fx_load(const double* slot) {
    return (double)(long long)(*reinterpret_cast<const unsigned long long*>(slot)) * (1.0 / LDT_FX_SCALE);
}

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
// gelu_erf_fast / gelu_erf_fast2: GEMM-epilogue forms whose output is rounded to bf16 anyway.  The lower normal tail is
// written as a power of two of a polynomial,
//     Phi(-z) = 1/2 * 2^(-z (c1 + c2 z + c3 z^2 + c4 z^3 + c5 z^4)),   z = |x|,
// (-log2(2 Phi(-z)) is smooth and nearly quadratic; coefficients: weighted minimax fit of the GELU error over z in
// [0, 13]; the polynomial stays >= c1 for every z >= 0, so large |x| saturates to 2^-inf = 0 without a clamp), then
//     x Phi(x) = x/2 + |x| (1/2 - Phi(-|x|)) :  ONE transcendental (v_exp_f32), no reciprocal, no compare/select.
// Evaluated in fp32: |gelu_fast(x) - gelu(x)| <= 8.7e-7 for every finite x (the accuracy of the A&S 7.1.26 erfc form it
// replaces, which cost a reciprocal and an exponential per element): 12 VALU per PAIR instead of 19.
#define LDT_GELU_C1 1.1510004997253418f
#define LDT_GELU_C2 0.45959582924842834f
#define LDT_GELU_C3 0.052146632224321365f
#define LDT_GELU_C4 (-0.007198718376457691f)
#define LDT_GELU_C5 0.00048810214502736926f
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x);
    float p = fmaf(LDT_GELU_C5, z, LDT_GELU_C4);
    p = fmaf(p, z, LDT_GELU_C3);
    p = fmaf(p, z, LDT_GELU_C2);
    p = fmaf(p, z, LDT_GELU_C1);
    const float e = __builtin_amdgcn_exp2f(-(p * z));            // 2 Phi(-z)
    const float r = fmaf(e, -0.5f, 0.5f);                        // 1/2 - Phi(-z)
    return fmaf(z, r, x * 0.5f);
}
// two-lane form: the polynomial runs on v_pk_fma_f32 / v_pk_mul_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
    const f32x2 z = {fabsf(x[0]), fabsf(x[1])};
    f32x2 p = z * LDT_GELU_C5 + LDT_GELU_C4;
    p = p * z + LDT_GELU_C3;
    p = p * z + LDT_GELU_C2;
    p = p * z + LDT_GELU_C1;
    const f32x2 q = p * z;
    const f32x2 e = {__builtin_amdgcn_exp2f(-q[0]), __builtin_amdgcn_exp2f(-q[1])};
    const f32x2 r = e * (-0.5f) + 0.5f;
    return x * 0.5f + z * r;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
