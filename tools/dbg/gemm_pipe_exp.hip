// Persistent, role-specialised bf16 MFMA GEMM for the LARGE token-linear layers (M >= 8k rows: the headline workload, B = 64 x 256 tokens):
// 128 x 128 tiles, 768 threads = 4 COMPUTE waves + 4 LOADER waves + 4 EPILOGUE waves, one workgroup per CU, K streamed across tiles.
//
//   Y[M,N] = epilogue( X[M,K] . W[N,K]^T )        (same operands / epilogues / LN-folding algebra as gemm_bf16.hip; model/layers.py:121-124,159-161)
//
// Why (rounds 2-3): in the 256^2 kernel one 512-thread workgroup owns a CU, so a tile's epilogue — every byte of it algorithmic:
// the fp32 residual read + write, the bf16 outputs, GELU — cannot overlap a main loop; ~107 us of each 435-us block were exposed epilogue, and
// the residual GEMMs (one tile per CU) alternated a chip-wide MFMA phase with a chip-wide HBM phase.  Here the three jobs belong to three sets
// of waves that run concurrently on every SIMD:
//   * LOADER waves issue the operand stream (global_load_lds, full 128-B lines, 8-row pieces) into a 3-stage ring of 64-deep K-tiles and run
//     across tile boundaries (gemm_mid.hip's loader with v3's persistent stream);
//   * COMPUTE waves (128 rows x 32 columns each) do nothing but fragment reads and MFMAs; at the end of a tile they park the fp32 accumulators
//     in a 64-KiB LDS tile (16 ds_write_b128 per lane) and start the next tile at once;
//   * EPILOGUE waves turn the parked tile into outputs while the next tile's main loop runs: 32-row steps spread over the K-tile periods,
//     residual rows / row statistics requested one step ahead, whole-row 16-B accesses, LN-folding producer (xs + per-row sums per 128
//     columns) and consumer (rstd / mean from the partials, S | C slices in registers) forms, exact-erf GELU.
// One s_barrier per K-tile joins the three roles: loaders arrive when K-tile g+1 has landed (counted vmcnt), compute waves when their reads of
// K-tile g have returned, epilogue waves when their slice of the previous tile is done — so the accumulator tile is free again exactly when the
// compute waves need it.  A tile with 64 flop per operand byte is bound by the L2 -> LDS DMA rate (~85 GB/s per CU) at ~1.4 PFLOP/s — what the
// 256^2 kernel's main loop reaches — with the epilogue no longer added on top; the N = hidden residual GEMMs become HBM-bound.
//
// LDS: 3 x (X[128][64] | W[128][64]) bf16 = 96 KiB ring + 64 KiB accumulator tile = 160 KiB.  <= 168 VGPRs (12 waves).
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"

#define PP_BK 64
#define PP_NS 3
#define PP_XB (128 * PP_BK * 2)
#define PP_STAGE (2 * PP_XB)
#define PP_RING (PP_NS * PP_STAGE)
#define PP_DUMP (128 * 128 * 4)
#define PP_LDS (PP_RING + PP_DUMP)
#define PP_PPW 8                      /* DMA pieces per loader wave and K-tile: 4 X + 4 W */
#define PP_STEPS 4                    /* epilogue steps per tile (32 rows each) */

enum { PP_FOLD_NONE = 0, PP_FOLD_PRODUCER = 1, PP_FOLD_CONSUMER = 2 };

#define PP_BARRIER()                          \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
#define PP_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int R, int MPR, int TOTAL>
__device__ __forceinline__ void pp_interleave() {
#pragma unroll
    for (int i = 0; i < R; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
    }
    if constexpr (TOTAL - R * MPR > 0) __builtin_amdgcn_sched_group_barrier(0x008, TOTAL - R * MPR, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// this workgroup's tile list: the tiles are cut into 8 contiguous chunks (one per XCD label bid & 7); inside a chunk workgroup j takes ids
// j, j + wpx, ...; ids sweep groups of `gm` row tiles column by column, so the 32 tiles an XCD computes at a time form a gm x 32/gm block
struct PpTiles {
    int tiles_n, tiles_m, c_lo, j, wpx, my, gm;
    __device__ __forceinline__ void init(int M, int N, int G, int bid) {
        tiles_n = N / 128; tiles_m = M / 128; gm = 4;
        const int tiles = tiles_m * tiles_n;
        const int nx = G < 8 ? G : 8;
        const int xcd = bid % nx;
        j = bid / nx;
        wpx = (G - xcd + nx - 1) / nx;
        c_lo = (int)((long)tiles * xcd / nx);
        const int c_hi = (int)((long)tiles * (xcd + 1) / nx);
        my = (c_hi - c_lo - j + wpx - 1) / wpx;
        my = my > 0 ? my : 0;
    }
    __device__ __forceinline__ void at(int it, int& m0, int& n0) const {
        const int id = c_lo + j + it * wpx;
        const int per = gm * tiles_n, g = id / per, r = id - g * per;
        const int rows = min(gm, tiles_m - g * gm);
        m0 = (g * gm + r % rows) * 128; n0 = (r / rows) * 128;
    }
};

__device__ __forceinline__ float pp_half_sum(float v) {   // sum over the 32 lanes of a wave half (all 32 end with the total), fixed order
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    v += __shfl_xor(v, 16, 64);
    return v;
}

template <int EPI, int FOLD>
__global__ __launch_bounds__(768) void gemm_bf16_nt_pipe_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_pp[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkt = a.K / PP_BK;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    PpTiles tl;
    tl.init(a.M, a.N, gridDim.x, blockIdx.x);
    if (tl.my == 0) return;                                  // whole workgroup, before any barrier
    const int P = tl.my * nkt;                               // K-tiles of this workgroup's stream; barrier g sits between K-tiles g and g+1
    char* dump = smem_pp + PP_RING;

    if (wave >= 4 && wave < 8) {
        // ------------------------------------------------------------------------------------------ loader waves
        const int lw = wave - 4;
        int xo[4], wo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (lw * 4 + q) * 8 + (lane >> 3);
            const int ch = ((lane & 7) ^ ((r >> 1) & 7)) << 4;
            xo[q] = r * (int)a.ldx * 2 + ch;
            wo[q] = r * (int)a.ldw * 2 + ch;
        }
        const char* xbase = nullptr; const char* wbase = nullptr;
        int s_it = 0, s_kt = 0;
        auto seek = [&](int it) {
            int m0, n0;
            tl.at(it, m0, n0);
            xbase = reinterpret_cast<const char*>(a.X + (long)m0 * a.ldx);
            wbase = reinterpret_cast<const char*>(a.W + (long)n0 * a.ldw);
        };
        auto issue = [&](int slot) {                         // the stream's current K-tile -> stage `slot`; then advance the stream
            char* st = smem_pp + slot * PP_STAGE;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xbase + xo[q]),
                                                 (__attribute__((address_space(3))) void*)(st + (lw * 4 + q) * 1024), 16, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + wo[q]),
                                                 (__attribute__((address_space(3))) void*)(st + PP_XB + (lw * 4 + q) * 1024), 16, 0, 0);
            xbase += PP_BK * 2; wbase += PP_BK * 2;
            if (++s_kt == nkt) { s_kt = 0; if (++s_it < tl.my) seek(s_it); }
        };
        seek(0);
        issue(0);
        if (P > 1) issue(1);
        if (P > 1) pp_wait_vmcnt<PP_PPW>(); else pp_wait_vmcnt<0>();
        PP_BARRIER();                                        // prologue barrier: K-tile 0 landed
        if (P > 2) issue(2);
        int slot = 0;
        for (int g = 0; g + 1 < P; ++g) {
            // K-tile g+1 landed: issued after it: K-tile g+2 (if any); K-tile g+3 goes out past the barrier
            if (g + 2 < P) pp_wait_vmcnt<PP_PPW>(); else pp_wait_vmcnt<0>();
            PP_BARRIER();                                    // barrier g
            if (g + 3 < P) issue(slot);
            slot = slot + 1 == PP_NS ? 0 : slot + 1;
        }
        return;
    }

    if (wave < 4) {
        // ------------------------------------------------------------------------------------------ compute waves
        const int wn = wave;
        const int lrow = lane & 15, lchk = lane >> 4;
        const int sw = (lrow >> 1) & 7;
        const int xb0 = lrow * 128 + ((lchk ^ sw) << 4), xb1 = lrow * 128 + (((4 + lchk) ^ sw) << 4);
        const int wrow = PP_XB + (wn * 32 + lrow) * 128;
        const int wb0 = wrow + ((lchk ^ sw) << 4), wb1 = wrow + (((4 + lchk) ^ sw) << 4);
        f32x4 acc[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jx = 0; jx < 8; ++jx) acc[i][jx] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 wa[2], wb[2], xa[8], xb[8];
        auto ld = [&](bf16x8 (&w)[2], bf16x8 (&x)[8], const char* st, int woff, int xoff) {
#pragma unroll
            for (int i = 0; i < 2; ++i) w[i] = *reinterpret_cast<const bf16x8*>(st + woff + i * 2048);
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const bf16x8*>(st + xoff + i * 2048);
        };
        auto mm = [&](const bf16x8 (&w)[2], const bf16x8 (&x)[8]) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ni], x[mi], acc[ni][mi], 0, 0, 0);
        };
        // park the finished accumulators: dump[row][col] fp32, 512-B rows, 16-B chunk index XORed with (row & 7); lane holds
        // D[n = wn*32 + ni*16 + lchk*4 + r][m = mi*16 + lrow] = 4 consecutive columns of one row = one chunk
        auto park = [&]() {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int row = mi * 16 + lrow, c = wn * 8 + ni * 4 + lchk;
                    *reinterpret_cast<f32x4*>(dump + row * 512 + ((c ^ (row & 7)) << 4)) = acc[ni][mi];
                    acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            __builtin_amdgcn_sched_barrier(0);
        };
        PP_BARRIER();                                        // prologue barrier
        const char* st = smem_pp;
        ld(wa, xa, st, wb0, xb0);
        PP_LGKM0();
        int slot = 0, kt = 0;
        __builtin_amdgcn_sched_barrier(0);
        for (int g = 0; g + 1 < P; ++g) {
            ld(wb, xb, st, wb1, xb1);                        // last reads of this stage
            mm(wa, xa);
            pp_interleave<10, 1, 16>();
            PP_LGKM0();
            PP_BARRIER();                                    // barrier g: K-tile g+1 landed; this stage may be refilled; (kt >= 1) the parked tile is consumed
            slot = slot + 1 == PP_NS ? 0 : slot + 1;
            st = smem_pp + slot * PP_STAGE;
            ld(wa, xa, st, wb0, xb0);
            mm(wb, xb);
            pp_interleave<10, 1, 16>();
            PP_LGKM0();
            if (++kt == nkt) { kt = 0; park(); }             // a tile's last K-tile: park it (complete before barrier g+1)
        }
        ld(wb, xb, st, wb1, xb1);
        mm(wa, xa);
        pp_interleave<10, 1, 16>();
        mm(wb, xb);
        park();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_BARRIER();                                        // drain barrier: the last tile is parked
        return;
    }

    // ---------------------------------------------------------------------------------------------- epilogue waves
    // lane -> (row pair slot r2 = lane >> 5, 16-B column chunk c = lane & 31); a wave takes 8 rows of a 32-row step as 4 row pairs:
    // row = s * 32 + ew * 8 + p * 2 + r2, p = 0..3
    const int ew = wave - 8;
    const int r2 = lane >> 5, c = lane & 31;
    const float* gate = a.gate;
    if (EPI == EPI_RESID_F32 && gate) gate += (long)step * a.gate_step_stride;
    const bool has_gate = (EPI == EPI_RESID_F32) && gate;
    const bool per_sample_gate = has_gate && a.gate_sample_stride != 0;
    const float* ln_scale = (FOLD == PP_FOLD_PRODUCER) ? a.ln_scale + (long)step * a.ln_step_stride : nullptr;
    const float* fold_S = (FOLD == PP_FOLD_CONSUMER) ? a.fold_S + (long)step * a.fold_step_stride : nullptr;
    const float* fold_C = (FOLD == PP_FOLD_CONSUMER) ? a.fold_C + (long)step * a.fold_step_stride : nullptr;
    const float invk = 1.0f / (float)a.K;
    int e_m0 = 0, e_n0 = 0;                                   // the parked tile's origin
    f32x4 add4 = {0.f, 0.f, 0.f, 0.f}, s4 = {0.f, 0.f, 0.f, 0.f}, g4 = {1.f, 1.f, 1.f, 1.f}, sc4 = {1.f, 1.f, 1.f, 1.f};
    f32x4 xq[4];                                              // residual rows of the NEXT step to run (EPI_RESID_F32)
    f32x2 pq[4];                                              // row-statistics partial (lane c < stats_parts) of the NEXT step's rows (consumer)
    auto tile_consts = [&](int n0) {                         // column-dependent vectors of a tile
        const int col = n0 + c * 4;
        if constexpr (FOLD == PP_FOLD_CONSUMER) { s4 = *reinterpret_cast<const f32x4*>(fold_S + col); add4 = *reinterpret_cast<const f32x4*>(fold_C + col); }
        else add4 = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has_gate && !per_sample_gate) g4 = *reinterpret_cast<const f32x4*>(gate + col);
        if constexpr (FOLD == PP_FOLD_PRODUCER) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(ln_scale + col);
#pragma unroll
            for (int r = 0; r < 4; ++r) sc4[r] = 1.0f + t[r];
        }
    };
    auto prefetch = [&](int m0, int n0, int s) {             // requests for step s of the tile at (m0, n0)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long row = m0 + s * 32 + ew * 8 + p * 2 + r2;
            if constexpr (EPI == EPI_RESID_F32) xq[p] = *reinterpret_cast<const f32x4*>(a.resid + row * a.ldr + n0 + c * 4);
            if constexpr (FOLD == PP_FOLD_CONSUMER) {
                const int part = c < a.stats_parts ? c : a.stats_parts - 1;
                pq[p] = *reinterpret_cast<const f32x2*>(a.stats_in + ((long)part * a.M + row) * 2);
            }
        }
    };
    auto run_step = [&](int s) {                             // step s of the parked tile (e_m0, e_n0); operands = xq / pq
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int trow = s * 32 + ew * 8 + p * 2 + r2;
            const long row = e_m0 + trow;
            f32x4 v = *reinterpret_cast<const f32x4*>(dump + trow * 512 + ((c ^ (trow & 7)) << 4));
            if constexpr (FOLD == PP_FOLD_CONSUMER) {
                const float s1 = pp_half_sum(c < a.stats_parts ? pq[p][0] : 0.f), s2 = pp_half_sum(c < a.stats_parts ? pq[p][1] : 0.f);
                const float mean = s1 * invk;
                const float rs = rsqrtf(fmaxf(s2 * invk - mean * mean, 0.f) + 1e-6f), nm = -mean * rs;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rs * v[r] + (nm * s4[r] + add4[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += add4[r];
            }
            if constexpr (EPI == EPI_RESID_F32) {
                if (per_sample_gate) g4 = *reinterpret_cast<const f32x4*>(gate + (row / a.rows_per_sample) * a.gate_sample_stride + e_n0 + c * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = xq[p][r] + g4[r] * v[r];
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + row * a.ldo + e_n0 + c * 4) = v;
                if constexpr (FOLD == PP_FOLD_PRODUCER) {
                    const bf16x4 pk = {(bf16_t)(v[0] * sc4[0]), (bf16_t)(v[1] * sc4[1]), (bf16_t)(v[2] * sc4[2]), (bf16_t)(v[3] * sc4[3])};
                    *reinterpret_cast<bf16x4*>(a.xs + row * a.ldxs + e_n0 + c * 4) = pk;
                    const float s1 = pp_half_sum((v[0] + v[1]) + (v[2] + v[3]));
                    const float s2 = pp_half_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                    if (c == 0) *reinterpret_cast<f32x2*>(a.stats_out + ((long)(e_n0 >> 7) * a.M + row) * 2) = (f32x2){s1, s2};
                }
            } else if constexpr (EPI == EPI_F32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + row * a.ldo + e_n0 + c * 4) = v;
            } else {
                if constexpr (EPI == EPI_GELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 gg = gelu_erf_fast2((f32x2){v[r], v[r + 1]});
                        v[r] = gg[0]; v[r + 1] = gg[1];
                    }
                }
                const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + row * a.ldo + e_n0 + c * 4) = pk;
            }
        }
    };
    // step s of a parked tile runs in the K-tile period kt = 1 + s * (nkt - 1) / PP_STEPS of the NEXT tile's main loop (periods 1 .. nkt - 1 are
    // free: the tile is parked during period 0); its operands were requested one step earlier
    {
        int m0, n0;
        tl.at(0, m0, n0);
        prefetch(m0, n0, 0);                                 // first step of the first tile
    }
    PP_BARRIER();                                            // prologue barrier
    int kt = 0, it = 0, s_next = 0;                          // it = tile whose K-tiles are being multiplied; the parked tile is it - 1
    for (int g = 0; g + 1 < P; ++g) {
        // interval before barrier g: K-tile (it, kt) is being multiplied
        if (it >= 1) {
            if (kt == 0) {                                   // tile it-1 is being parked right now: fetch its column vectors meanwhile
                tl.at(it - 1, e_m0, e_n0);
                tile_consts(e_n0);
                s_next = 0;
            } else if (s_next < PP_STEPS && kt == 1 + s_next * (nkt - 1) / PP_STEPS) {
                run_step(s_next);
                ++s_next;
                if (s_next < PP_STEPS) prefetch(e_m0, e_n0, s_next);
                else { int m0, n0; tl.at(it, m0, n0); prefetch(m0, n0, 0); }      // first step of the tile now in the main loop
            }
        }
        PP_BARRIER();                                        // barrier g
        if (++kt == nkt) { kt = 0; ++it; }
    }
    // the stream's last K-tile has no barrier of its own; if it opened a tile-boundary period above (kt == 0 there) nothing is pending.
    // Steps of tile my-2 not yet run (possible only when its successor's main loop was cut short — it never is: every tile has nkt periods)
    PP_BARRIER();                                            // drain barrier: the last tile is parked
    tl.at(tl.my - 1, e_m0, e_n0);
    tile_consts(e_n0);
    // xq / pq hold step 0 of the last tile (requested when the previous tile's last step ran, or in the prologue for a one-tile stream)
#pragma unroll 1
    for (int s = 0; s < PP_STEPS; ++s) {
        run_step(s);
        if (s + 1 < PP_STEPS) prefetch(e_m0, e_n0, s + 1);
    }
}

// ---------------------------------------------------------------------------------------------- launcher
// LDT_GEMM_PIPE: 0 off, 1 on (default set by the round-4 A/B: see DESIGN.md §4).  Takes: M, N multiples of 128, K a multiple of 64 with >= 9
// K-tiles, 16-byte aligned operands, enough tiles for every CU (the large-batch regime); statistics granule = 128 columns.
static int pipe_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LDT_GEMM_PIPE"); v = e ? atoi(e) : 0; }
    return v;
}

bool ldt_gemm_pipe_route(int M, int N, int K) {
    return pipe_env() && M % 128 == 0 && N % 128 == 0 && K % PP_BK == 0 && K / PP_BK >= 9 && (long)(M / 128) * (N / 128) >= 2L * LDT_NUM_CUS;
}

template <int EPI, int FOLD>
static int pipe_launch_t(const GemmArgs* a, hipStream_t stream) {
    LDT_ENSURE_LDS((&gemm_bf16_nt_pipe_kernel<EPI, FOLD>), PP_LDS, "gemm_pipe");
    const long tiles = (long)(a->M / 128) * (a->N / 128);
    const int lim = (a->max_wgs > 0 && a->max_wgs < LDT_NUM_CUS) ? a->max_wgs : LDT_NUM_CUS;
    const int grid = tiles < lim ? (int)tiles : lim;
    hipLaunchKernelGGL((gemm_bf16_nt_pipe_kernel<EPI, FOLD>), dim3(grid), dim3(768), PP_LDS, stream, *a);
    return ldt_check_launch("gemm_bf16_nt_pipe");
}

// fold: 0 plain, 1 producer (EPI_RESID_F32: stats_out[N / 128][M][2]), 2 consumer (EPI_BF16 / EPI_GELU_BF16: stats_in[K / 128][M][2])
bool ldt_gemm_pipe_try(int epi, int fold, const GemmArgs* a, hipStream_t stream, int* status) {
    if (!ldt_gemm_pipe_route(a->M, a->N, a->K)) return false;
    if (a->ldx % 8 != 0 || a->ldw % 8 != 0 || a->ldo % 8 != 0 || !ldt_aligned16(a->X) || !ldt_aligned16(a->W) || !ldt_aligned16(a->out)) return false;
    if (a->bias && !ldt_aligned16(a->bias)) return false;
    if (epi == EPI_RESID_F32) {
        if (!a->resid || a->ldr % 4 != 0 || !ldt_aligned16(a->resid)) return false;
        if (a->gate && (!ldt_aligned16(a->gate) || a->gate_sample_stride % 4 != 0 || a->gate_step_stride % 4 != 0 || a->rows_per_sample <= 0)) return false;
        if (fold == 1) {
            if (!a->xs || !a->ln_scale || !a->stats_out || a->ldxs % 4 != 0 || a->stats_parts * 128 != a->N) return false;
            *status = pipe_launch_t<EPI_RESID_F32, PP_FOLD_PRODUCER>(a, stream);
        } else if (fold == 0) *status = pipe_launch_t<EPI_RESID_F32, PP_FOLD_NONE>(a, stream);
        else return false;
        return true;
    }
    if (epi == EPI_BF16 || epi == EPI_GELU_BF16) {
        if (fold == 2) {
            if (!a->stats_in || !a->fold_S || !a->fold_C || a->stats_parts * 128 != a->K || a->stats_parts > 32) return false;
            *status = epi == EPI_BF16 ? pipe_launch_t<EPI_BF16, PP_FOLD_CONSUMER>(a, stream) : pipe_launch_t<EPI_GELU_BF16, PP_FOLD_CONSUMER>(a, stream);
        } else if (fold == 0) *status = epi == EPI_BF16 ? pipe_launch_t<EPI_BF16, PP_FOLD_NONE>(a, stream) : pipe_launch_t<EPI_GELU_BF16, PP_FOLD_NONE>(a, stream);
        else return false;
        return true;
    }
    if (epi == EPI_F32 && fold == 0) { *status = pipe_launch_t<EPI_F32, PP_FOLD_NONE>(a, stream); return true; }
    return false;
}
