// fp32 linear layers with one skinny side over millions of rows — the Compressor's per-point / per-token 1x1 convs that are pure
// streaming (model/Compressor/Network.py:192 input Conv1d 3 -> C, :266 output Conv1d C -> 3, MiniPointnet's first conv):
//     out[M][N] = act_out( act_in(A[M][K]) . W[N][K]^T + bias )
// A 128-wide MFMA tile spends 98 % of its work on zero padding there (N = 3) or cannot be used at all (K = 3), and the generic kernels
// ran at 0.04 – 0.3 of the HBM rate.  Two forms, both fp32 FMA chains (the fp32 parity bars of ldt_sgemm hold), both bandwidth-bound:
//   * small K (<= 8), N a multiple of 4 dividing 1024: a thread owns 4 consecutive outputs of a row with its K x 4 weights in
//     registers for the whole (grid-strided) launch; the row's K inputs are broadcast loads; 16-B (fp32) or 8-B (bf16) stores, whole rows
//     per 32 lanes;
//   * small N (<= 8), K a multiple of 4: 32 lanes read a row 16 B each (coalesced), N partial dot products per lane, 5 xor-shuffles.
#include "kernels.h"

namespace {

__device__ __forceinline__ float sk_act(float v, int act) {
    switch (act) {
        case ACT_SILU: return v / (1.0f + expf(-v));
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_GELU: return gelu_erf(v);
        default: return v;
    }
}

template <int KP>   // K padded to KP (weights of the padding are zero, its inputs are not loaded)
__global__ __launch_bounds__(256) void linear_smallk_kernel(const SgemmArgs a) {
    constexpr int R = 4;                                            // rows per thread and trip: R x K loads in flight
    const int nq = a.N >> 2, rows_per_block = 256 / nq;
    const int q = threadIdx.x % nq, rloc = threadIdx.x / nq;
    float w[KP][4];
#pragma unroll
    for (int k = 0; k < KP; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[k][j] = k < a.K ? a.B[(long)(4 * q + j) * a.ldb + k] : 0.f;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * q);
    const long step = (long)gridDim.x * rows_per_block;
    for (long row0 = (long)blockIdx.x * rows_per_block + rloc; row0 < a.M; row0 += step * R) {
        float x[R][KP];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = row0 + r * step < a.M ? row0 + r * step : a.M - 1;
            const float* ar = a.A + row * a.lda;
#pragma unroll
            for (int k = 0; k < KP; ++k) x[r][k] = k < a.K ? ar[k] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = row0 + r * step;
            if (row >= a.M) break;
            f32x4 o = b4;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const float xv = a.act_in ? sk_act(x[r][k], a.act_in) : x[r][k];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fmaf(xv, w[k][j], o[j]);
            }
            if (a.act_out) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = sk_act(o[j], a.act_out);
            }
            if (a.out_bf16) {
                const bf16x4 p = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.C) + row * a.ldc + 4 * q) = p;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.C) + row * a.ldc + 4 * q) = o;
            }
        }
    }
}

template <int NP>   // outputs padded to NP
__global__ __launch_bounds__(256) void linear_smalln_kernel(const SgemmArgs a) {
    constexpr int R = 32 / NP;                                      // rows per 32-lane group and trip (R x NP results = one store): R loads in flight
    const int sub = threadIdx.x & 31;                               // lane inside the 32-lane row group
    const long grp = (long)blockIdx.x * 8 + (threadIdx.x >> 5), ngrp = (long)gridDim.x * 8;
    for (long row0 = grp * R; row0 < a.M; row0 += ngrp * R) {
        float acc[R][NP];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int n = 0; n < NP; ++n) acc[r][n] = 0.f;
        for (int k0 = sub * 4; k0 < a.K; k0 += 128) {
            f32x4 x[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const long row = row0 + r < a.M ? row0 + r : a.M - 1;
                x[r] = *reinterpret_cast<const f32x4*>(a.A + row * a.lda + k0);
            }
            if (a.act_in) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[r][j] = sk_act(x[r][j], a.act_in);
            }
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                if (n < a.N) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.B + (long)n * a.ldb + k0);   // (L1-resident: N x K floats)
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[r][n] = fmaf(x[r][j], w4[j], acc[r][n]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int n = 0; n < NP; ++n)
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) acc[r][n] += __shfl_xor(acc[r][n], o, 64);
        // lane (r, n) of the group writes output n of row r: R x N <= 32 results, one store instruction
        {
            const int r = sub / NP, n = sub % NP;
            float v = 0.f;
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int nn = 0; nn < NP; ++nn) v = (rr == r && nn == n) ? acc[rr][nn] : v;
            if (r < R && n < a.N && row0 + r < a.M) {
                v = sk_act(v + (a.bias ? a.bias[n] : 0.f), a.act_out);
                if (a.out_bf16) reinterpret_cast<bf16_t*>(a.C)[(row0 + r) * a.ldc + n] = (bf16_t)v;
                else reinterpret_cast<float*>(a.C)[(row0 + r) * a.ldc + n] = v;
            }
        }
    }
}

}  // namespace

// -> true when one of the streaming forms took the problem
bool ldt_skinny_linear_try(const SgemmArgs* a, hipStream_t s, int* status) {
    if (a->M < 8192) return false;                                   // short problems: the tiled kernels are fine
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (a->N <= 8 && a->K % 4 == 0 && a->lda % 4 == 0 && a->ldb % 4 == 0 && ldt_aligned16(a->A) && ldt_aligned16(a->B)) {
        long blocks = ((long)a->M + 31) / 32;
        if (blocks > (long)cus * 16) blocks = (long)cus * 16;
        if (a->N <= 4) hipLaunchKernelGGL(linear_smalln_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, *a);
        else hipLaunchKernelGGL(linear_smalln_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, *a);
        *status = ldt_check_launch("linear_smalln");
        return true;
    }
    const int nq = a->N / 4;
    if (a->K <= 8 && a->N % 4 == 0 && nq >= 8 && nq <= 256 && 256 % nq == 0 && ldt_aligned16(a->C) &&
        a->ldc % 4 == 0 && (!a->bias || ldt_aligned16(a->bias))) {
        const int rpb = 256 / nq;
        long blocks = ((long)a->M + rpb * 4 - 1) / (rpb * 4);
        if (blocks > (long)cus * 8) blocks = (long)cus * 8;
        const dim3 g((unsigned)blocks), b(256);
        if (a->K <= 4) hipLaunchKernelGGL(linear_smallk_kernel<4>, g, b, 0, s, *a);
        else hipLaunchKernelGGL(linear_smallk_kernel<8>, g, b, 0, s, *a);
        *status = ldt_check_launch("linear_smallk");
        return true;
    }
    return false;
}
