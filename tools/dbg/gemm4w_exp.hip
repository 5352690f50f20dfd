// EXPERIMENT (not part of the product): 4-wave, one-wave-per-SIMD bf16 GEMM main loop with 128x128 per-wave tiles.
//
// Question it answers: the shipped 8-wave 256x256 kernel reads every X fragment in 4 waves and every W fragment in 2
// (96 KB of LDS reads per 32-deep sub-tile = the LDS array's peak).  With 4 waves of 128x128 the
// same tile needs 64 KB of reads per sub-tile and no inter-group barriers, at the price of 256 accumulator registers
// per lane (AGPRs, 1 wave per SIMD) and of leaving the interleave of ds_read / LDS-DMA / MFMA to the compiler
// (sched_group_barrier).  Main loop only: Y = X[M,K] * W[N,K]^T, fp32 accumulators, optional plain bf16 store.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbg/gemm4w_exp.hip -o /tmp/gemm4w && /tmp/gemm4w [M N K] [store]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define STAGE_BYTES 32768
#define OPER_BYTES 16384
#define NSLOT 4
#ifndef INTERLEAVE
#define INTERLEAVE 1
#endif

__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

struct Args { const bf16_t* X; const bf16_t* W; bf16_t* Y; int M, N, K; int store; };

__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 15, lchk = lane >> 4;
    const int nks = a.K >> 5;
    const int tiles_n = a.N / 256;
    const int tiles = (a.M / 256) * tiles_n;

    // per-lane DMA source offsets: a sub-tile operand = 256 rows x 64 B = 16 pieces of 16 rows; wave w issues pieces 4w..4w+3
    long xsrc[4], wsrc[4];
    int xoff[8], woff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rx = wm * 128 + i * 16 + lrow, rw = wn * 128 + i * 16 + lrow;
        xoff[i] = rx * 64 + ((lchk ^ swz(rx)) << 4);
        woff[i] = OPER_BYTES + rw * 64 + ((lchk ^ swz(rw)) << 4);
    }

    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (wave * 4 + q) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ swz(r);
            xsrc[q] = (long)(m0 + r) * a.K + c * 8;
            wsrc[q] = (long)(n0 + r) * a.K + c * 8;
        }
        auto issue = [&](int g) {                                // DMA of sub-tile g into slot g & 3 (8 pieces per wave)
            char* slot = smem + (g & (NSLOT - 1)) * STAGE_BYTES;
            const int k = (g < nks ? g : nks - 1) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.X + xsrc[q] + k),
                                                 (__attribute__((address_space(3))) void*)(slot + (wave * 4 + q) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.W + wsrc[q] + k),
                                                 (__attribute__((address_space(3))) void*)(slot + OPER_BYTES + (wave * 4 + q) * 1024), 16, 0, 0);
            }
        };
        f32x4 acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        __syncthreads();                                         // previous tile's LDS reads are done
        issue(0); issue(1); issue(2);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // sub-tile 0 landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        bf16x8 xf[2][8], wf[2][8];
        {
            const char* st = smem;
#pragma unroll
            for (int i = 0; i < 8; ++i) { wf[0][i] = *reinterpret_cast<const bf16x8*>(st + woff[i]); xf[0][i] = *reinterpret_cast<const bf16x8*>(st + xoff[i]); }
        }
        for (int g = 0; g < nks; g += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                        // two sub-tiles per trip: register sets alternate statically
                const int gg = g + h;
                // sub-tile gg+1 landed?  (outstanding after this wait: gg+2 only), then everybody's pieces: barrier
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue(gg + 3);                                   // slot of gg-1: its fragment reads were issued a whole iteration ago
                const char* st = smem + ((gg + 1) & (NSLOT - 1)) * STAGE_BYTES;
#pragma unroll
                for (int i = 0; i < 8; ++i) { wf[h ^ 1][i] = *reinterpret_cast<const bf16x8*>(st + woff[i]); xf[h ^ 1][i] = *reinterpret_cast<const bf16x8*>(st + xoff[i]); }
#pragma unroll
                for (int nj = 0; nj < 8; ++nj)
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi)
                        acc[nj][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][nj], xf[h][mi], acc[nj][mi], 0, 0, 0);
#if INTERLEAVE
                // 64 MFMA, 16 ds_read_b128, 8 LDS-DMA: one DS read per 4 MFMAs, one DMA per 8
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA)
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#endif
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.store) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = m0 + wm * 128 + mi * 16 + lrow;
#pragma unroll
                for (int nj = 0; nj < 8; ++nj) {
                    const int n = n0 + wn * 128 + nj * 16 + lchk * 4;
                    const f32x4 v = acc[nj][mi];
                    const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    *reinterpret_cast<bf16x4*>(a.Y + (long)m * a.N + n) = pk;
                }
            }
        } else {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (s == 123.456f) a.Y[tid] = (bf16_t)s;             // keeps the accumulators live, never true in practice
        }
    }
}

static float bf2f(bf16_t v) { return (float)v; }

int main(int argc, char** argv) {
    int M = argc > 3 ? atoi(argv[1]) : 16384, N = argc > 3 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 1024;
    int store = argc > 4 ? atoi(argv[4]) : 0;
    if (M % 256 || N % 256 || K % 64 || (K / 32) % 2) { printf("M, N multiples of 256; K multiple of 64\n"); return 1; }
    std::vector<bf16_t> hx((size_t)M * K), hw((size_t)N * K);
    srand(1);
    for (auto& v : hx) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& v : hw) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f / sqrtf((float)K));
    bf16_t *dx, *dw, *dy;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dy, (size_t)M * N * 2);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dy, 0, (size_t)M * N * 2);
    Args a{dx, dw, dy, M, N, K, 1};
    const int lds = NSLOT * STAGE_BYTES;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int tiles = (M / 256) * (N / 256), grid = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL(gemm4w_kernel, dim3(grid), dim3(256), lds, 0, a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    std::vector<bf16_t> hy((size_t)M * N);
    hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 4000; ++t) {
        const int m = rand() % M, n = rand() % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
        const double e = fabs(ref - bf2f(hy[(size_t)m * N + n]));
        if (e > maxerr) maxerr = e;
    }
    printf("check: max abs err over 4000 samples %.4g (bf16 out, |y| ~ 0.5)\n", maxerr);
    a.store = store;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gemm4w_kernel, dim3(grid), dim3(256), lds, 0, a);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gemm4w_kernel, dim3(grid), dim3(256), lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("gemm4w M=%d N=%d K=%d store=%d interleave=%d: %.1f us  %.0f TFLOP/s\n", M, N, K, store, INTERLEAVE, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    return maxerr < 0.02 ? 0 : 3;
}
