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

// exact-erf GELU (nn.GELU() default; model/layers.py:111-113 via tools/utils.py:107-108)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
