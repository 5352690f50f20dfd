// EXPERIMENT (tools/dbg): 256x256 tile, FOUR waves (one per SIMD, 128x128 per wave, 256 AGPR accumulators), hand-placed
// instruction stream (every MFMA / ds_read / LDS-DMA is an `asm volatile` statement), and — what gemm4w_asm.hip's ablations
// showed to be the bound of BOTH that kernel and the shipped 8-wave one — FULL-LINE operand DMA:
//
//   a DMA piece of 16 rows x 64 B (32-deep sub-tiles, 64-B LDS rows) fetches every 128-B line twice (two consecutive sub-tiles),
//   and the CU's 32 KiB L1 does not hold a line that long: the operand stream alone took 80 us for a 137-GFLOP GEMM (as long as
//   its MFMAs alone), 48-52 us with pieces of 8 rows x 128 B.
//
// So K is consumed in 64-deep K-tiles, LDS rows of 128 B, the 16-B chunk index XORed with (row >> 1) & 7 (conflict-free
// ds_read_b128 of 16x32 fragments; applied on the DMA source address and on the read address).  LDS = a ring of FIVE 32-KiB units,
// one unit = one operand of one K-tile (X[256][64] or W[256][64]): X(t) -> unit 2t mod 5, W(t) -> unit (2t+1) mod 5.  With five units
// the operand stream is spread EVENLY over the phases (8 pieces per wave and phase: the CU's address pipeline is ~80 % loaded by a
// 256^2 tile, any burst stalls the one-wave-per-SIMD MFMA stream):
//   phase A(t): 64 MFMAs on half 0 of K-tile t | 16 ds_reads: half 1 of K-tile t       | 8 DMA pieces: X(t+2) -> the unit W(t-1) left
//               lgkmcnt(0); vmcnt(8) (K-tile t+1 landed); s_barrier
//   phase B(t): 64 MFMAs on half 1 of K-tile t | 16 ds_reads: half 0 of K-tile t+1     | 8 DMA pieces: W(t+2) -> the unit X(t) left
//               lgkmcnt(0)
// One barrier per K-tile; a piece has 1 - 2 phases (1000 - 2000 MFMA cycles) to land.  Persistent over tiles, the operand stream
// runs across tile boundaries.  (The fifth unit doubles as the epilogue's staging area in a product kernel.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbg/gemm4w_fl.hip -o /tmp/gemm4w_fl && /tmp/gemm4w_fl [M N K] [store]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <utility>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define UNIT_BYTES 32768
#ifndef NO_MFMA
#define NO_MFMA 0
#endif
#ifndef NO_READ
#define NO_READ 0
#endif
#ifndef NO_DMA
#define NO_DMA 0
#endif

struct Args { const bf16_t* X; const bf16_t* W; bf16_t* Y; int M, N, K; int store; };

struct Ctx {
    f32x4 acc[8][8];
    i32x4 xf[2][8], wf[2][8];         // [k-half][fragment]
    int xrd[2], wrd[2];               // per-lane LDS read addresses of k-half 0 / 1 (unit of the K-tile being read included)
    const char* xptr[8];              // per-lane DMA sources of this wave's 8 X pieces / 8 W pieces at their streams' K-tiles
    const char* wptr[8];
};

template <int OFF>
__device__ __forceinline__ void dsread(i32x4& dst, int addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF)); }
__device__ __forceinline__ void m0_set(int v) { asm volatile("s_mov_b32 m0, %0" ::"s"(v)); }
template <int INC>
__device__ __forceinline__ void m0_add() { asm volatile("s_add_u32 m0, m0, %0" ::"i"(INC) : "scc"); }
__device__ __forceinline__ void glds(const char* p) { asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(p) : "memory"); }

// read number R (0..15) of a 16-fragment half: W0, X0..X7, W1..W7 (first needed first)
template <int H, int R>
__device__ __forceinline__ void read_frag(Ctx& c) {
    if constexpr (NO_READ) return;
    if constexpr (R == 0) dsread<0>(c.wf[H][0], c.wrd[H]);
    else if constexpr (R <= 8) dsread<(R - 1) * 2048>(c.xf[H][R - 1], c.xrd[H]);
    else dsread<(R - 8) * 2048>(c.wf[H][R - 8], c.wrd[H]);
}
// DMA piece D (0..7) of the operand of this phase (A: X, B: W); M0 walks the wave's 8 KiB of the unit
template <int PH, int D>
__device__ __forceinline__ void dma_piece(Ctx& c) {
    if constexpr (NO_DMA) return;
    if constexpr (PH == 0) glds(c.xptr[D]); else glds(c.wptr[D]);
}

// MFMA T (0..63) of phase PH (0 = A: half 0, 1 = B: half 1) and the filler of the gap behind it: per group of 8 gaps
//   R R D R R - S -   (groups 0..3)      - - D - - - S -   (groups 4..7)       R = ds_read, D = LDS-DMA, S = M0 step
template <int PH, int T>
__device__ __forceinline__ void gap(Ctx& c) {
    constexpr int k = T >> 3, p = T & 7;
    if constexpr (!NO_MFMA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c.acc[k][p]) : "v"(c.wf[PH][k]), "v"(c.xf[PH][p]));
    constexpr int RH = PH ^ 1;                                        // phase A reads half 1 (this K-tile), phase B half 0 (next K-tile)
    if constexpr (p == 2) dma_piece<PH, k>(c);
    else if constexpr (p == 6) { if constexpr (!NO_DMA && k < 7) m0_add<1024>(); }
    else if constexpr (k < 4 && (p == 0 || p == 1 || p == 3 || p == 4)) read_frag<RH, k * 4 + (p < 2 ? p : p - 1)>(c);
}
template <int PH, int... Ts>
__device__ __forceinline__ void phase(Ctx& c, std::integer_sequence<int, Ts...>) { (gap<PH, Ts>(c), ...); }

#define WAIT_LGKM(H)                                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                     \
                 : "+v"(c.xf[H][0]), "+v"(c.xf[H][1]), "+v"(c.xf[H][2]), "+v"(c.xf[H][3]), "+v"(c.xf[H][4]), "+v"(c.xf[H][5]), "+v"(c.xf[H][6]), \
                   "+v"(c.xf[H][7]), "+v"(c.wf[H][0]), "+v"(c.wf[H][1]), "+v"(c.wf[H][2]), "+v"(c.wf[H][3]), "+v"(c.wf[H][4]), "+v"(c.wf[H][5]), \
                   "+v"(c.wf[H][6]), "+v"(c.wf[H][7]))

__device__ __forceinline__ void nop0() { asm volatile("s_nop 0"); }
template <int PH, int... Is>
__device__ __forceinline__ void dma_all(Ctx& c, std::integer_sequence<int, Is...>) {   // prologue form: nothing between the statements -> pad the M0 hazards by hand
    ((dma_piece<PH, Is>(c), nop0(), m0_add<1024>(), nop0()), ...);
}
template <int H, int... Is>
__device__ __forceinline__ void read_all(Ctx& c, std::integer_sequence<int, Is...>) { (read_frag<H, Is>(c), ...); }

__global__ __launch_bounds__(256, 1) void gemm4w_fl_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx c;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 15, lchk = lane >> 4;
    const int nkt = a.K >> 6;
    const int tiles_n = a.N / 256;
    const int tiles = (a.M / 256) * tiles_n;
    // tiles cut into 8 contiguous chunks, one per XCD label (blockIdx % 8): the column tiles of an X row panel share an L2
    const int G_ = (int)gridDim.x, bid = (int)blockIdx.x;
    const int nx = G_ < 8 ? G_ : 8, xcd = bid % nx, jb = bid / nx;
    const int wpx = (G_ - xcd + nx - 1) / nx;
    const int c_lo = (int)((long)tiles * xcd / nx), c_hi = (int)((long)tiles * (xcd + 1) / nx);
    const int my_tiles = (c_hi - c_lo - jb + wpx - 1) / wpx > 0 ? (c_hi - c_lo - jb + wpx - 1) / wpx : 0;
    if (my_tiles <= 0) return;
    auto tile_id = [&](int it) { const int t = c_lo + jb + it * wpx; return t < c_hi ? t : c_hi - 1; };

    // LDS read addresses inside a unit: fragment i = rows base + i*16 + lrow (128-B rows), 16-B chunk h*4 + lchk, swizzled by (row >> 1) & 7 = lrow >> 1
    const int smem_base = (int)(size_t)smem;
    const int sw = (lrow >> 1) & 7;
    const int xl0 = smem_base + (wm * 128 + lrow) * 128 + ((lchk ^ sw) << 4), xl1 = smem_base + (wm * 128 + lrow) * 128 + (((4 + lchk) ^ sw) << 4);
    const int wl0 = smem_base + (wn * 128 + lrow) * 128 + ((lchk ^ sw) << 4), wl1 = smem_base + (wn * 128 + lrow) * 128 + (((4 + lchk) ^ sw) << 4);
    const int dma_w = __builtin_amdgcn_readfirstlane(smem_base + wave * 8192);       // this wave's 8 pieces inside a unit

    // DMA streams: piece q of this wave = rows (wave*8 + q)*8 + (lane >> 3) of the operand, LDS position lane & 7 holds chunk (lane & 7) ^ ((row >> 1) & 7)
    int xtile = 0, xkt = 0, wtile = 0, wkt = 0;
    auto seek = [&](int it, bool is_x) {
        const int tt = tile_id(it);                                   // past the end: re-read the last tile (never consumed)
        const int m0 = (tt / tiles_n) * 256, n0 = (tt % tiles_n) * 256;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = (wave * 8 + q) * 8 + (lane >> 3);
            const int cs = (lane & 7) ^ ((r >> 1) & 7);
            if (is_x) c.xptr[q] = reinterpret_cast<const char*>(a.X + (long)(m0 + r) * a.K + cs * 8);
            else c.wptr[q] = reinterpret_cast<const char*>(a.W + (long)(n0 + r) * a.K + cs * 8);
        }
    };
    auto advance_x = [&]() {
        if (++xkt == nkt) { xkt = 0; ++xtile; seek(xtile, true); }
        else {
#pragma unroll
            for (int q = 0; q < 8; ++q) c.xptr[q] += 128;
        }
    };
    auto advance_w = [&]() {
        if (++wkt == nkt) { wkt = 0; ++wtile; seek(wtile, false); }
        else {
#pragma unroll
            for (int q = 0; q < 8; ++q) c.wptr[q] += 128;
        }
    };
    using S8 = std::make_integer_sequence<int, 8>;
    using S16 = std::make_integer_sequence<int, 16>;
    using S64 = std::make_integer_sequence<int, 64>;
    seek(0, true); seek(0, false);
    // prologue: X(0), W(0), X(1), W(1) -> units 0..3; fragments of (K-tile 0, half 0)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        m0_set(dma_w + u * UNIT_BYTES); nop0();
        if (u & 1) { dma_all<1>(c, S8{}); advance_w(); } else { dma_all<0>(c, S8{}); advance_x(); }
    }
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");  // K-tile 0 landed everywhere
    int ux = 0;                                                        // unit of X(t); W(t) = ux + 1 (mod 5)
    c.xrd[0] = xl0; c.wrd[0] = wl0 + UNIT_BYTES;
    read_all<0>(c, S16{});
    WAIT_LGKM(0);

    for (int it = 0; it < my_tiles; ++it) {
        const int tile = tile_id(it);
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) c.acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_nop 4" ::: "memory");                         // compiler-written accumulators -> first MFMA
        for (int t = 0; t < nkt; ++t) {
            const int uw = ux + 1 >= 5 ? ux + 1 - 5 : ux + 1;
            const int ux1 = ux + 2 >= 5 ? ux + 2 - 5 : ux + 2, uw1 = ux + 3 >= 5 ? ux + 3 - 5 : ux + 3;
            const int dx = ux + 4 >= 5 ? ux + 4 - 5 : ux + 4;          // X(t+2) -> the unit W(t-1) left;  W(t+2) -> X(t)'s unit
            c.xrd[1] = xl1 + ux * UNIT_BYTES; c.wrd[1] = wl1 + uw * UNIT_BYTES;
            m0_set(dma_w + dx * UNIT_BYTES);
            phase<0>(c, S64{});                                       // half 0; reads half 1 of this K-tile; DMA X(t+2)
            advance_x();
            WAIT_LGKM(1);
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");   // K-tile t+1 landed everywhere; everyone is done with X(t), W(t)
            c.xrd[0] = xl0 + ux1 * UNIT_BYTES; c.wrd[0] = wl0 + uw1 * UNIT_BYTES;
            m0_set(dma_w + ux * UNIT_BYTES);
            phase<1>(c, S64{});                                       // half 1; reads half 0 of the next K-tile; DMA W(t+2)
            advance_w();
            WAIT_LGKM(0);
            ux = ux1;
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // last MFMAs' results -> compiler-generated readers
        if (a.store) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = m0 + wm * 128 + mi * 16 + lrow;
#pragma unroll
                for (int nj = 0; nj < 8; ++nj) {
                    const int n = n0 + wn * 128 + nj * 16 + lchk * 4;
                    const f32x4 v = c.acc[nj][mi];
                    const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    *reinterpret_cast<bf16x4*>(a.Y + (long)m * a.N + n) = pk;
                }
            }
        } else {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) s += c.acc[i][j][0] + c.acc[i][j][1] + c.acc[i][j][2] + c.acc[i][j][3];
            if (s == 123.456f) a.Y[threadIdx.x] = (bf16_t)s;         // keeps the accumulators live, never true in practice
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static float bf2f(bf16_t v) { return (float)v; }

int main(int argc, char** argv) {
    int M = argc > 3 ? atoi(argv[1]) : 16384, N = argc > 3 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 1024;
    int store = argc > 4 ? atoi(argv[4]) : 0;
    if (M % 256 || N % 256 || K % 64) { printf("M, N multiples of 256; K multiple of 64\n"); return 1; }
    const size_t pad = 1 << 20;
    std::vector<bf16_t> hx((size_t)M * K), hw((size_t)N * K);
    srand(1);
    for (auto& v : hx) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f);
    for (auto& v : hw) v = (bf16_t)((rand() % 2001 - 1000) / 1000.0f / sqrtf((float)K));
    bf16_t *dx, *dw, *dy;
    (void)hipMalloc(&dx, hx.size() * 2 + pad); (void)hipMalloc(&dw, hw.size() * 2 + pad); (void)hipMalloc(&dy, (size_t)M * N * 2);
    (void)hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(dy, 0, (size_t)M * N * 2);
    Args a{dx, dw, dy, M, N, K, 1};
    const int lds = 5 * UNIT_BYTES;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_fl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int tiles = (M / 256) * (N / 256), grid = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL(gemm4w_fl_kernel, dim3(grid), dim3(256), lds, 0, a);
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
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(gemm4w_fl_kernel, dim3(grid), dim3(256), lds, 0, a);
    (void)hipEventRecord(e0);
    const int reps = 30;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gemm4w_fl_kernel, dim3(grid), dim3(256), lds, 0, a);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("gemm4w_fl M=%d N=%d K=%d store=%d no_mfma=%d no_read=%d no_dma=%d: %.1f us  %.0f TFLOP/s   (check: max abs err %.4g)\n", M, N, K, store, NO_MFMA,
           NO_READ, NO_DMA, ms * 1e3, 2.0 * M * N * K / ms / 1e9, maxerr);
    return (maxerr < 0.02 || NO_MFMA || NO_READ || NO_DMA) ? 0 : 3;
}
