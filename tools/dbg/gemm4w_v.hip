// EXPERIMENT (tools/dbg, round 5): the vendor library's tiling and schedule for the Score GEMM shapes, re-derived for this repository's
// LDS image — 256 x 256 tile, FOUR waves (one per SIMD), 128 x 128 per wave in 256 accumulator AGPRs, 64-deep K-tiles in two 64-KiB LDS
// buffers, and the whole K loop of a tile as ONE asm statement (tools/dbg/gen_gemm4w_v.py writes its text: every MFMA, ds_read_b128,
// LDS-DMA piece, wait and barrier is placed by the generator; no compiler-scheduled instruction, no VALU address arithmetic in the loop).
//
//   python3 tools/dbg/gen_gemm4w_v.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbg/gemm4w_v.hip -o /tmp/gemm4w_v && /tmp/gemm4w_v [M N K] [store]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Args { const bf16_t* X; const bf16_t* W; bf16_t* Y; int M, N, K; int store; };

#define ACC_READ(dst, idx) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(dst) : "n"(idx))

#include "gemm4w_v_clobbers.inc"

__global__ __launch_bounds__(256, 1) void gemm4w_v_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 15, lchk = lane >> 4;
    const int nkt = a.K >> 6;
    const int tiles_n = a.N / 256;
    // XCD-chunked tile order (blockIdx % 8 labels the XCD): an XCD's workgroups take a contiguous run of tiles (row-major: they share X panels)
    const int G_ = (int)gridDim.x, bid = (int)blockIdx.x;
    const int xcd = bid & 7, per = G_ >> 3;
    const int tile = (G_ & 7) ? bid : xcd * per + (bid >> 3);
    const int m0t = (tile / tiles_n) * 256, n0t = (tile % tiles_n) * 256;

    const int smem_base = (int)(size_t)smem;
    const int sw = (lrow >> 1) & 7;
    int xa0 = smem_base + (wm * 128 + lrow) * 128 + ((lchk ^ sw) << 4), xa1 = smem_base + (wm * 128 + lrow) * 128 + (((4 + lchk) ^ sw) << 4);
    int wa0 = smem_base + 32768 + (wn * 128 + lrow) * 128 + ((lchk ^ sw) << 4), wa1 = smem_base + 32768 + (wn * 128 + lrow) * 128 + (((4 + lchk) ^ sw) << 4);
    // LDS-DMA: piece (wave * 8 + q) = rows (wave * 8 + q) * 8 + (lane >> 3); LDS position lane & 7 of a row holds global chunk (lane & 7) ^ ((row >> 1) & 7),
    // (row >> 1) & 7 = (q & 1) * 4 + (lane >> 4)  ->  one per-lane offset for even pieces, one for odd ones
    const int ldxb = a.K * 2, ldwb = a.K * 2;
    const int vox0 = (lane >> 3) * ldxb + (((lane & 7) ^ (lane >> 4)) << 4), vox1 = (lane >> 3) * ldxb + (((lane & 7) ^ (4 + (lane >> 4))) << 4);
    const int vow0 = (lane >> 3) * ldwb + (((lane & 7) ^ (lane >> 4)) << 4), vow1 = (lane >> 3) * ldwb + (((lane & 7) ^ (4 + (lane >> 4))) << 4);
    const int sx0 = __builtin_amdgcn_readfirstlane(wave * 64 * ldxb), sxs = 8 * ldxb;
    const int sw0 = __builtin_amdgcn_readfirstlane(wave * 64 * ldwb), sws = 8 * ldwb;
    const unsigned long long xb = (unsigned long long)(a.X + (long)m0t * a.K), wb = (unsigned long long)(a.W + (long)n0t * a.K);
    const int srdx0 = __builtin_amdgcn_readfirstlane((int)(unsigned)xb), srdx1 = __builtin_amdgcn_readfirstlane((int)((xb >> 32) & 0xffff));
    const int srdw0 = __builtin_amdgcn_readfirstlane((int)(unsigned)wb), srdw1 = __builtin_amdgcn_readfirstlane((int)((wb >> 32) & 0xffff));
    const int srd2 = 0x7fffffff, srd3 = 0x00020000;
    const int ldsx = __builtin_amdgcn_readfirstlane(smem_base + wave * 8192);

    asm volatile(
#include "gemm4w_v_loop.inc"
        : [xa0] "+v"(xa0), [xa1] "+v"(xa1), [wa0] "+v"(wa0), [wa1] "+v"(wa1)
        : [vox0] "v"(vox0), [vox1] "v"(vox1), [vow0] "v"(vow0), [vow1] "v"(vow1), [srdx0] "s"(srdx0), [srdx1] "s"(srdx1), [srdx2] "s"(srd2), [srdx3] "s"(srd3),
          [srdw0] "s"(srdw0), [srdw1] "s"(srdw1), [srdw2] "s"(srd2), [srdw3] "s"(srd3), [sx0] "s"(sx0), [sxs] "s"(sxs), [sw0] "s"(sw0), [sws] "s"(sws),
          [nkt] "s"(nkt), [ldsx] "s"(ldsx)
        : G4V_CLOBBERS, "memory", "scc", "vcc");

    // accumulator (k, p) = a[(k*8+p)*4 + r] = D[n = n0t + wn*128 + k*16 + lchk*4 + r][m = m0t + wm*128 + p*16 + lrow]
    if (a.store == 2) {
        // staged epilogue (the product kernels' form): 16 output rows x the wave's 128 columns per pass through 4.25 KiB of the wave's (idle)
        // operand buffer, then 16 B per lane over whole 256-B row segments
        char* reg = smem + wave * 8192;
        constexpr int RS = 256 + 16;
#define STAGE_TILE(k, p)                                                                          \
    {                                                                                             \
        float f0, f1, f2, f3;                                                                     \
        ACC_READ(f0, ((k) * 8 + (p)) * 4 + 0); ACC_READ(f1, ((k) * 8 + (p)) * 4 + 1);             \
        ACC_READ(f2, ((k) * 8 + (p)) * 4 + 2); ACC_READ(f3, ((k) * 8 + (p)) * 4 + 3);             \
        const bf16x4 pk = {(bf16_t)f0, (bf16_t)f1, (bf16_t)f2, (bf16_t)f3};                       \
        *reinterpret_cast<bf16x4*>(reg + lrow * RS + ((k) * 16 + lchk * 4) * 2) = pk;             \
    }
#define STAGE_PASS(p)                                                                                                              \
    {                                                                                                                              \
        STAGE_TILE(0, p) STAGE_TILE(1, p) STAGE_TILE(2, p) STAGE_TILE(3, p) STAGE_TILE(4, p) STAGE_TILE(5, p) STAGE_TILE(6, p) STAGE_TILE(7, p) \
        bf16_t* o = a.Y + (long)(m0t + wm * 128 + (p) * 16) * a.N + n0t + wn * 128;                                               \
        _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                                         \
            const int row = it * 4 + (lane >> 4), ch = lane & 15;                                                                  \
            const uint4 d = *reinterpret_cast<const uint4*>(reg + row * RS + ch * 16);                                             \
            *reinterpret_cast<uint4*>(o + (long)row * a.N + ch * 8) = d;                                                           \
        }                                                                                                                          \
    }
        STAGE_PASS(0) STAGE_PASS(1) STAGE_PASS(2) STAGE_PASS(3) STAGE_PASS(4) STAGE_PASS(5) STAGE_PASS(6) STAGE_PASS(7)
    } else if (a.store) {
#define STORE_TILE(k, p)                                                                          \
    {                                                                                             \
        float f0, f1, f2, f3;                                                                     \
        ACC_READ(f0, ((k) * 8 + (p)) * 4 + 0); ACC_READ(f1, ((k) * 8 + (p)) * 4 + 1);             \
        ACC_READ(f2, ((k) * 8 + (p)) * 4 + 2); ACC_READ(f3, ((k) * 8 + (p)) * 4 + 3);             \
        const int m = m0t + wm * 128 + (p) * 16 + lrow, n = n0t + wn * 128 + (k) * 16 + lchk * 4; \
        const bf16x4 pk = {(bf16_t)f0, (bf16_t)f1, (bf16_t)f2, (bf16_t)f3};                       \
        *reinterpret_cast<bf16x4*>(a.Y + (long)m * a.N + n) = pk;                                 \
    }
#define STORE_ROW(k) STORE_TILE(k, 0) STORE_TILE(k, 1) STORE_TILE(k, 2) STORE_TILE(k, 3) STORE_TILE(k, 4) STORE_TILE(k, 5) STORE_TILE(k, 6) STORE_TILE(k, 7)
        STORE_ROW(0) STORE_ROW(1) STORE_ROW(2) STORE_ROW(3) STORE_ROW(4) STORE_ROW(5) STORE_ROW(6) STORE_ROW(7)
    } else {
        float f;
        ACC_READ(f, 17);
        if (f == 123.456f) a.Y[threadIdx.x] = (bf16_t)f;         // never true in practice
    }
}

static float bf2f(bf16_t v) { return (float)v; }

int main(int argc, char** argv) {
    int M = argc > 3 ? atoi(argv[1]) : 16384, N = argc > 3 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 4096;
    int store = argc > 4 ? atoi(argv[4]) : 0;
    if (M % 256 || N % 256 || K % 64 || K < 256) { printf("M, N multiples of 256; K multiple of 64, >= 256\n"); return 1; }
    const size_t pad = 1 << 20;
    std::vector<bf16_t> hx((size_t)M * K), hw((size_t)N * K);
    srand(1);
    for (auto& v : hx) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& v : hw) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f / sqrtf((float)K));
    bf16_t *dx, *dw, *dy;
    (void)hipMalloc(&dx, hx.size() * 2 + pad); (void)hipMalloc(&dw, hw.size() * 2 + pad); (void)hipMalloc(&dy, (size_t)M * N * 2);
    (void)hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(dy, 0, (size_t)M * N * 2);
    Args a{dx, dw, dy, M, N, K, store == 2 ? 2 : 1};
    const int lds = 131072;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_v_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int tiles = (M / 256) * (N / 256);
    hipLaunchKernelGGL(gemm4w_v_kernel, dim3(tiles), dim3(256), lds, 0, a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    std::vector<bf16_t> hy((size_t)M * N);
    (void)hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 40000; ++t) {
        const int m = rand() % M, n = rand() % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
        const double e = fabs(ref - bf2f(hy[(size_t)m * N + n]));
        if (e > maxerr) maxerr = e;
    }
    a.store = store;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(gemm4w_v_kernel, dim3(tiles), dim3(256), lds, 0, a);
    (void)hipEventRecord(e0);
    const int reps = 30;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gemm4w_v_kernel, dim3(tiles), dim3(256), lds, 0, a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("gemm4w_v M=%d N=%d K=%d store=%d: %.1f us  %.0f TFLOP/s   (check: max abs err %.4g)\n", M, N, K, store, ms * 1e3, 2.0 * M * N * K / ms / 1e9, maxerr);
    return maxerr < 0.02 ? 0 : 3;
}
