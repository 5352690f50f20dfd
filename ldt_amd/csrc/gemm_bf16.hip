// bf16 MFMA GEMM for the token-linear layers of the LDT hot path (gfx950 / MI355X).
//
//   Y[M,N] = epilogue( X[M,K] (bf16, row-major, ldx) · W[N,K]^T (bf16, row-major, ldw) + bias[N] )
//
// This is every 1x1 Conv1d / Linear on the path (reference: model/layers.py:159-161 fc_q/fc_kv/fc_o,
// :121-124 MLP fc/out, model/scorenet/score.py:110,112 ln_in/ln_out.ln; Compressor twins).  A Conv1d
// weight (out,in,1) is already the [N][K] K-contiguous operand MFMA wants, so nothing is transposed.
//
// Structure (v1, "2-phase" of the CDNA4 guide): 128x128x64 tile, 256 threads = 2x2 waves of 64x64,
// mfma_f32_16x16x32_bf16 with the operands SWAPPED (D[n][m] = W·X^T) so that each lane ends up with 4
// consecutive output columns of one row -> 16-B fp32 / 8-B bf16 epilogue accesses.  Both operand tiles
// are staged HBM->LDS with global_load_lds (16 B/lane, LDS image lane-linear), double-buffered; the
// LDS bank-conflict swizzle chunk' = chunk ^ ((row>>1)&7) is applied on the SOURCE address and on the
// ds_read address (both-sides rule).  The workgroup->tile map is XCD-aware (blocks that share an XCD's
// L2 walk one row-panel of X across the N tiles of W).
//
// Fused epilogues (template EPI):
//   EPI_F32        out fp32  = acc + bias
//   EPI_BF16       out bf16  = acc + bias
//   EPI_GELU_BF16  out bf16  = gelu_erf(acc + bias)                       (MLP up, layers.py:127-129)
//   EPI_RELU_BF16  out bf16  = relu(acc + bias [+ skip bf16])             (PreExtraction, Compressor/layers.py:115-160)
//   EPI_RESID_F32  out fp32  = resid + gate[s,n] * (acc + bias)           (x + gate*(...), layers.py:218-219; gate may be null)
#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

#include "kernels.h"

#define BK 64

enum { FOLD_NONE = 0, FOLD_PRODUCER = 1, FOLD_CONSUMER = 2 };   // LN folding (described at the 256-tile kernel below)

template <int ROWS, int NW = 4>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, long ld, int row0, int nrows_total,
                                           int k0, char* lds_tile, int wave, int lane) {
    // one operand tile = ROWS rows x 128 B = ROWS/8 pieces of 1 KiB (8 rows each); ROWS/(8 NW) pieces per wave
#pragma unroll
    for (int p = 0; p < ROWS / (8 * NW); ++p) {
        const int piece = wave * (ROWS / (8 * NW)) + p;
        const int r = piece * 8 + (lane >> 3);
        const int cdst = lane & 7;
        const int csrc = cdst ^ ((r >> 1) & 7);
        int grow = row0 + r;
        grow = grow < nrows_total ? grow : nrows_total - 1;          // clamp: OOB rows are never stored
        const bf16_t* src = g + (long)grow * ld + k0 + csrc * 8;
        char* dst = lds_tile + piece * 1024;                          // wave-uniform; HW adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

// TBM x TBN output tile (each 128 or 64): 128x128 is the workhorse of mid-size problems, the smaller shapes keep all
// 256 CUs busy when M*N is small (T=32 latents: M = 2048 rows).
// NST = 2: a stage is waited for in full before the barrier (the simple 2-phase loop).  NST = 3: the DMA of stage
// kt+2 stays in flight across the barrier (counted vmcnt, raw s_barrier), which hides the load latency when only 1-3
// workgroups share a CU (small M: the T = 32 latents, half-batch shapes).
// NW = 8 (512 threads, waves 4 x 2 over the tile): the mid-size form — a 256 x 128 tile with a 3-stage ring is ONE workgroup per CU
// whose operand stream carries 85 flop per byte (128 x 64 tiles: 43), for the batches whose GEMMs are bound by the per-CU
// L2 -> LDS rate (M = 1-4 k rows; DESIGN.md §4 "small-batch regime").
template <int EPI, int TBM, int TBN, int NST = 2, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_nt_kernel(const GemmArgs a) {
    constexpr int BM = TBM, BN = TBN;
    constexpr int XB = TBM * BK * 2, WB = TBN * BK * 2;               // operand tile bytes per stage
    constexpr int WGM = NW / 2;                                       // waves along m (2 along n)
    constexpr int MT = TBM / (16 * WGM), NT = TBN / 32;               // 16x16 accumulator tiles per wave (m, n)
    constexpr int OPS = (TBM + TBN) / (8 * NW);                       // LDS-DMA instructions per wave and stage
    __shared__ __attribute__((aligned(16))) char smem[NST * (XB + WB)];  // [stage][X|W]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware bijective remap of the 1-D grid (blocks b, b+8, ... share an XCD's L2) ----
    const int tiles_n = (a.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    // tile order: row-major (an XCD's chunk of workgroups = a band of output rows x every column tile: it reads its own rows of X
    // and ALL of W) or column-major (a band of column tiles x every row tile: its own slice of W, all of X) — the launcher picks
    // column-major for wide outputs over few row tiles (M = 1024: QKV 15.9 -> 14.5 us, MLP-up 18.2 -> 16.1 us; tools/dbg/smallm_bench.py)
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tile_m = a.col_major ? wgid % tiles_m : wgid / tiles_n, tile_n = a.col_major ? wgid / tiles_m : wgid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bf16_t* Xk = a.X;
    const bf16_t* Wk = a.W;
    const int nk = a.K / BK;
    stage_tile<TBM, NW>(Xk, a.ldx, m0, a.M, 0, smem, wave, lane);
    stage_tile<TBN, NW>(Wk, a.ldw, n0, a.N, 0, smem + XB, wave, lane);
    // stages 1 .. NST-2 follow at once; wait until only they are outstanding (stage 0 landed)
    int ahead = 0;                                                    // stages issued beyond the one being waited for
#pragma unroll
    for (int st = 1; st < NST - 1; ++st)
        if (st < nk) {
            stage_tile<TBM, NW>(Xk, a.ldx, m0, a.M, st * BK, smem + st * (XB + WB), wave, lane);
            stage_tile<TBN, NW>(Wk, a.ldw, n0, a.N, st * BK, smem + st * (XB + WB) + XB, wave, lane);
            ++ahead;
        }
    if (NST >= 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
    else if (NST >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int lrow = lane & 15;
    const int lchk = lane >> 4;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        char* sx = smem + cur * (XB + WB);
        char* sw = sx + XB;
        if (kt + NST - 1 < nk) {                                      // buffer of stage kt-1 (NST = 3) / kt+1's own (NST = 2)
            int nb = cur + NST - 1; nb = nb >= NST ? nb - NST : nb;
            char* nx = smem + nb * (XB + WB);
            stage_tile<TBM, NW>(Xk, a.ldx, m0, a.M, (kt + NST - 1) * BK, nx, wave, lane);
            stage_tile<TBN, NW>(Wk, a.ldw, n0, a.N, (kt + NST - 1) * BK, nx + XB, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[NT], xf[MT];
            const int c = ks * 4 + lchk;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int rw = wn * (TBN / 2) + i * 16 + lrow;
                wf[i] = *reinterpret_cast<const bf16x8*>(sw + rw * 128 + ((c ^ ((rw >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int rx = wm * (TBM / WGM) + i * 16 + lrow;
                xf[i] = *reinterpret_cast<const bf16x8*>(sx + rx * 128 + ((c ^ ((rx >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
        // stage kt+1 landed (only the stages behind it, kt+2 .. kt+NST-1, may still be in flight), then visible to every wave
        const int left = nk - 2 - kt;                                 // stages that exist beyond kt+1
        if (NST >= 4 && left >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
        else if (NST >= 3 && left >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == NST ? 0 : cur + 1;
    }

    // ---- epilogue: lane holds D[n = nb + (lane>>4)*4 + r][m = mb + (lane&15)], r = 0..3 ----
    const float* gate = a.gate;
    if (EPI == EPI_RESID_F32 && gate && a.step_ptr) gate += (long)(*a.step_ptr) * a.gate_step_stride;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int m = m0 + wm * (TBM / WGM) + mi * 16 + lrow;
        if (m >= a.M) continue;
        const float* grow = nullptr;
        if (EPI == EPI_RESID_F32 && gate) grow = gate + (long)(m / a.rows_per_sample) * a.gate_sample_stride;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int n = n0 + wn * (TBN / 2) + ni * 16 + lchk * 4;
            if (n >= a.N) continue;
            f32x4 v = acc[ni][mi];
            const bool full = (n + 3 < a.N);
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
                if (full) { const f32x4 t = *reinterpret_cast<const f32x4*>(a.bias + n); b[0] = t[0]; b[1] = t[1]; b[2] = t[2]; b[3] = t[3]; }
                else for (int r = 0; r < 4; ++r) if (n + r < a.N) b[r] = a.bias[n + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += b[r];
            if (EPI == EPI_F32) {
                float* o = reinterpret_cast<float*>(a.out) + (long)m * a.ldo + n;
                if (full) *reinterpret_cast<f32x4*>(o) = v;
                else for (int r = 0; r < 4; ++r) if (n + r < a.N) o[r] = v[r];
            } else if (EPI == EPI_RESID_F32) {
                float* o = reinterpret_cast<float*>(a.out) + (long)m * a.ldo + n;
                const float* rs = a.resid + (long)m * a.ldr + n;
                if (full) {
                    f32x4 x = *reinterpret_cast<const f32x4*>(rs);
                    if (grow) { const f32x4 g = *reinterpret_cast<const f32x4*>(grow + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[r] = x[r] + g[r] * v[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[r] = x[r] + v[r];
                    }
                    *reinterpret_cast<f32x4*>(o) = x;
                } else {
                    for (int r = 0; r < 4; ++r) if (n + r < a.N) o[r] = rs[r] + (grow ? grow[n + r] : 1.f) * v[r];
                }
            } else {
                if (EPI == EPI_GELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {                 // two lanes of the polynomial per v_pk_* instruction
                        const f32x2 gg = gelu_erf_fast2((f32x2){v[r], v[r + 1]});
                        v[r] = gg[0]; v[r + 1] = gg[1];
                    }
                }
                if (EPI == EPI_RELU_BF16) {
                    if (a.skip) {
                        const bf16_t* sk = a.skip + (long)m * a.lds_ + n;
                        for (int r = 0; r < 4; ++r) if (n + r < a.N) v[r] += (float)sk[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (long)m * a.ldo + n;
                if (full) {
                    bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    *reinterpret_cast<bf16x4*>(o) = pk;
                } else for (int r = 0; r < 4; ++r) if (n + r < a.N) o[r] = (bf16_t)v[r];
            }
        }
    }
}


// =================================================================================================
// v2: persistent 256x256-tile kernel, 8 waves in two groups that ping-pong on each SIMD (the CDNA4 guide's
// "8-phase" structure), one workgroup per CU, K streamed continuously ACROSS output tiles.
//
//   * 512 threads = 8 waves; wave w and w+4 share a SIMD.  Group g = w>>2 owns output rows [g*128, +128) of the
//     tile, wn = w&3 owns 64 output columns: per wave 128x64 = 8x4 accumulator tiles of mfma_f32_16x16x32_bf16
//     (operands swapped as in v1: D[n][m]).
//   * K is consumed in 32-deep sub-tiles through a 4-slot LDS ring (slot = X[256][32] + W[256][32] bf16 = 32 KiB).
//     Rows are 64 B; the 16-B chunk index is XORed with f((row>>2)&3), f = {0,2,3,1}: every ds_read_b128 lane
//     group of the 16x16x32 operand read is bank-conflict-free (applied on the glds SOURCE address + read address).
//   * Each sub-tile is two phases of 16 MFMAs.  A phase = [load segment: ds_reads of this phase's operands + one
//     glds batch (2 x 1 KiB per wave) for a future sub-tile] s_barrier [16 MFMAs] s_barrier.  Group 1 runs one
//     barrier behind group 0, so on every SIMD one wave issues MFMAs while its partner reads LDS / issues DMA.
//   * The sub-tile stream does not stop at a tile boundary: the W batch issued at phase 0 of stream position g is
//     for position g+2, the X batch at phase 1 for g+3 — possibly the first sub-tiles of this workgroup's NEXT
//     tile, so the next tile's operands are already in LDS when the epilogue ends (no prologue bubble).  The only
//     VMEM wait in the loop is a counted `s_waitcnt vmcnt(6)` once per sub-tile (three batches stay in flight);
//     the first wait after an epilogue allows for the epilogue's own stores (vmcnt counts stores, in order).
//     Slot reuse distance >= 2 phases after the last read (WAR); data is read >= 1 barrier after every wave's
//     counted wait (RAW).  Past the end of the stream the batches are still issued (clamped, never consumed).
//   * Epilogue: accumulators -> per-wave 4 KiB LDS staging area (beside the ring, 160 KiB LDS in total) -> 16-B
//     per lane accesses over whole output rows (the raw fragment layout is store-issue bound, guide T21); stores
//     are not waited for, they drain under the next tile's main loop.
#define V2_STAGE_BYTES 32768
#define V2_OPER_BYTES 16384
#define V2_RING_BYTES (4 * V2_STAGE_BYTES)
#define V2_LDS_BYTES (V2_RING_BYTES + 8 * 4096)

#define V2_BARRIER()                          \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

// sum over the 16 lanes of a DPP row (all 16 end up with the total): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// LN folding (FOLD): the LayerNorm + AdaLN modulate between a residual GEMM and the next projection never runs as a
// kernel.  With h = LN(x)(1 + sc) + sh, r = rstd(x), mu = mean(x):
//     h . W^T + b  =  r * ( xs . W^T )  -  r * mu * S  +  C,     xs = x (1 + sc),  S_n = sum_k (1 + sc_k) W_nk,  C_n = sum_k sh_k W_nk + b_n
//   FOLD_PRODUCER (EPI_RESID_F32): besides x_new the epilogue stores xs = bf16(x_new (1 + sc)) and, per row, the partial
//     (sum, sum of squares) of x_new over this tile's 256 columns  -> stats_out[n0/256][M][2]  (DPP row sums, the four
//     column waves combined through LDS: one 8-B value per row and tile, summed in a fixed order — no atomics).
//   FOLD_CONSUMER (EPI_BF16 / EPI_GELU_BF16): X = xs; the tile's 256 rows x stats_parts partials are fetched by one
//     LDS-DMA piece per wave during the main loop (into the unused tail of the bf16 staging areas), r / -r mu are
//     formed per lane at the start of the epilogue and y = r acc + (-r mu S + C) replaces acc + bias.
//   S, C are batch-invariant per-step tables built by the host in fp32 from the same bf16 W the MFMAs read; they are
//   step-indexed, hence cold in every cache at every step: the tile's two 1 KiB slices ride the same mid-loop DMA slot
//   (four half-wave pieces) so that the epilogue opens on LDS reads instead of an HBM round trip.  For the same reason
//   the residual epilogue's step-indexed gate / ln_scale vectors of a workgroup's first tile are loaded before the
//   main loop and kept in 8 VGPRs.
#define V2_STATS_OFF 2304            /* bf16 staging uses 16 rows x 144 B of each wave's 4 KiB */
#define V2_SC_OFF (V2_STATS_OFF + 1024)   /* 512 B: a 128-column slice of fold_S (waves 0, 1) or fold_C (waves 2, 3) */

// FOLD_CONSUMER, once per tile and off the epilogue's critical path: thread R < 256 adds row R's partial (sum, sumsq)
// pairs (piece part*2 + (R >> 7) sits in that wave's staging tail) and overwrites the part-0 slot with (rstd, -mean*rstd).
__device__ __forceinline__ void v2_fold_finalize(char* stage_base, int R, int parts, int K) {
    char* slot = stage_base + (R >> 7) * 4096 + V2_STATS_OFF + (R & 127) * 8;
    float s1 = 0.f, s2 = 0.f;
    for (int pp = 0; pp < parts; ++pp) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(slot + pp * 2 * 4096);
        s1 += t[0]; s2 += t[1];
    }
    const float invk = 1.0f / (float)K;
    const float mean = s1 * invk;
    const float var = fmaxf(s2 * invk - mean * mean, 0.f);
    const float r = rsqrtf(var + 1e-6f);
    *reinterpret_cast<f32x2*>(slot) = (f32x2){r, -mean * r};
}

// interior tiles: per-wave LDS staging (16 output rows per pass) -> 16 B per lane over whole rows
// XRING (EPI_RESID_F32 on a workgroup's LAST tile, batch-shared gate): the fp32 residual rows are not loaded pass by pass into
// VGPRs (4 x 16 B per lane in flight per wave = 32 KB per CU: at ~2.5 us of HBM latency that caps the read at ~3.3 TB/s
// chip-wide, 18-21 us of exposed epilogue — tools/dbg/epi_ablate.py) but by LDS-DMA into the operand ring, which is idle by
// then: `xring` = this wave's 16 KiB of it = four 4 KiB pass slots.  Passes 0..3 are requested up front, pass mi + 4 when
// pass mi has consumed its slot: 16 KB per wave (128 KB per CU) in flight, lane-linear both ways (a lane reads back the 16 B it
// requested).  Counted waits: vmcnt(N), N = the ops issued after pass mi's requests (later requests + SP stores per pass).
template <int EPI, int FOLD, int XRING = 0>
__device__ __forceinline__ void v2_epilogue_staged(const GemmArgs& a, f32x4 (&acc)[4][8], int m0, int n0, int grp, int wn,
                                                   int lane, int lrow, int lchk, const float* gate, char* reg, char* stage_base,
                                                   const float* ln_scale, bool have_pre, f32x4 g4_pre, f32x4 sc4_pre,
                                                   char* xring = nullptr) {
    const int mb = m0 + grp * 128, nb = n0 + wn * 64;
    f32x4 bias4[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
        bias4[ni] = (a.bias && FOLD != FOLD_CONSUMER) ? *reinterpret_cast<const f32x4*>(a.bias + nb + ni * 16 + lchk * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16) {
        constexpr int RS = 128 + 16;                              // staged row: 64 bf16 + 16 B pad
        f32x4 s4[4];
        float rr[8], nm[8];
        if (FOLD == FOLD_CONSUMER) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {                      // (rstd, -mean*rstd) of the lane's rows, finalised mid-loop (v2_fold_finalize)
                const int R = grp * 128 + mi * 16 + lrow;
                const f32x2 t = *reinterpret_cast<const f32x2*>(stage_base + (R >> 7) * 4096 + V2_STATS_OFF + (R & 127) * 8);
                rr[mi] = t[0]; nm[mi] = t[1];
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int c = wn * 64 + ni * 16 + lchk * 4;       // S | C slices of this tile: DMA'd into waves 0..3's areas
                s4[ni] = *reinterpret_cast<const f32x4*>(stage_base + (c >> 7) * 4096 + V2_SC_OFF + (c & 127) * 4);
                bias4[ni] = *reinterpret_cast<const f32x4*>(stage_base + (2 + (c >> 7)) * 4096 + V2_SC_OFF + (c & 127) * 4);
            }
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 v = acc[ni][mi];
                if (FOLD == FOLD_CONSUMER) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * rr[mi] + (nm[mi] * s4[ni][r] + bias4[ni][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bias4[ni][r];
                }
                if (EPI == EPI_GELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {                 // two lanes of the polynomial per v_pk_* instruction
                        const f32x2 gg = gelu_erf_fast2((f32x2){v[r], v[r + 1]});
                        v[r] = gg[0]; v[r + 1] = gg[1];
                    }
                }
                if (EPI == EPI_RELU_BF16) {
                    if (a.skip) {
                        const bf16x4 sk = *reinterpret_cast<const bf16x4*>(a.skip + (long)(mb + mi * 16 + lrow) * a.lds_ + nb + ni * 16 + lchk * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)sk[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(reg + lrow * RS + (ni * 16 + lchk * 4) * 2) = pk;
            }
            bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (long)(mb + mi * 16) * a.ldo + nb;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const bf16x8 d = *reinterpret_cast<const bf16x8*>(reg + row * RS + ch * 16);
                *reinterpret_cast<bf16x8*>(o + (long)row * a.ldo + ch * 8) = d;
            }
        }
    } else {
        const int ch = lane & 15;                                 // 16-B chunk of the 256-B fp32 row (XOR-swizzled by row)
        f32x4 g4 = {1.f, 1.f, 1.f, 1.f};
        const bool has_gate = (EPI == EPI_RESID_F32) && gate;
        const bool shared_gate = has_gate && a.gate_sample_stride == 0;
        if (shared_gate) g4 = have_pre ? g4_pre : *reinterpret_cast<const f32x4*>(gate + nb + ch * 4);
        f32x4 sc4 = {1.f, 1.f, 1.f, 1.f};
        constexpr int SP = (FOLD == FOLD_PRODUCER) ? 8 : 4;       // VMEM stores a pass issues (x, and xs for the producer)
        // running source pointer: row (lane>>4) of the next 4-row group, advanced 4 rows per request (passes are requested in
        // order 0..7); kept opaque so that hipcc does not materialise all 32 addresses up front
        const float* xsrc = XRING ? a.resid + ((long)mb + (lane >> 4)) * a.ldr + nb + ch * 4 : nullptr;
        const long xstep = (long)4 * a.ldr;
        auto request_pass = [&](int p) {                          // 4 x 1 KiB: rows it*4 + (lane>>4) of pass p, 16 B per lane
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xsrc,
                                                 (__attribute__((address_space(3))) void*)(xring + (p & 3) * 4096 + it * 1024), 16, 0, 0);
                xsrc += xstep;
                asm volatile("" : "+v"(xsrc));
            }
        };
        if (XRING) { request_pass(0); request_pass(1); request_pass(2); request_pass(3); }
        float rs1[8], rs2[8];                                     // FOLD_PRODUCER: lanes with (lane & 15) < 4 keep row (lane&15)*4 + (lane>>4) of pass mi
        if (FOLD == FOLD_PRODUCER) {
            const f32x4 t = have_pre ? sc4_pre : *reinterpret_cast<const f32x4*>(ln_scale + nb + ch * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) sc4[r] = 1.0f + t[r];
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 v = acc[ni][mi];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bias4[ni][r];
                *reinterpret_cast<f32x4*>(reg + lrow * 256 + (((ni * 4 + lchk) ^ lrow) << 4)) = v;
            }
            const long mrow0 = mb + mi * 16;
            float k1 = 0.f, k2 = 0.f;
            if (XRING) {                                             // pass mi's rows have landed (ops issued after its requests: see above)
                constexpr int NW[8] = {12, 12 + SP, 12 + 2 * SP, 12 + 3 * SP, 12 + 3 * SP, 8 + 3 * SP, 4 + 3 * SP, 3 * SP};
                switch (mi) {                                        // (mi is a compile-time constant of the unrolled loop)
                    case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW[0]) : "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW[1]) : "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW[2]) : "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW[3]) : "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW[4]) : "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW[5]) : "memory"); break;
                    case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW[6]) : "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW[7]) : "memory"); break;
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 4 + (lane >> 4);
                f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * 256 + ((ch ^ row) << 4));
                float* o = reinterpret_cast<float*>(a.out) + (mrow0 + row) * a.ldo + nb + ch * 4;
                if (EPI == EPI_RESID_F32 && XRING) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(xring + (mi & 3) * 4096 + it * 1024 + lane * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = x[r] + g4[r] * v[r];
                } else if (EPI == EPI_RESID_F32) {
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (!(a.dbg & 1)) x = *reinterpret_cast<const f32x4*>(a.resid + (mrow0 + row) * a.ldr + nb + ch * 4);
                    if (has_gate && !shared_gate)
                        g4 = *reinterpret_cast<const f32x4*>(gate + ((mrow0 + row) / a.rows_per_sample) * a.gate_sample_stride + nb + ch * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = x[r] + g4[r] * v[r];
                }
                if (XRING || !(a.dbg & 2)) *reinterpret_cast<f32x4*>(o) = v;
                if (FOLD == FOLD_PRODUCER) {
                    const bf16x4 pk = {(bf16_t)(v[0] * sc4[0]), (bf16_t)(v[1] * sc4[1]), (bf16_t)(v[2] * sc4[2]), (bf16_t)(v[3] * sc4[3])};
                    if (XRING || !(a.dbg & 4)) *reinterpret_cast<bf16x4*>(a.xs + (mrow0 + row) * a.ldxs + nb + ch * 4) = pk;
                    const float s1 = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
                    const float s2 = row16_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                    const bool keep = (lane & 15) == it;
                    k1 = keep ? s1 : k1; k2 = keep ? s2 : k2;
                }
            }
            rs1[mi] = k1; rs2[mi] = k2;
            if (XRING && mi < 4) request_pass(mi + 4);               // into the slot this pass has just consumed
        }
        if (FOLD == FOLD_PRODUCER && !(a.dbg & 8)) {
            // the wave's 128 rows x (sum, sumsq) over its 64 columns -> head of its staging area; the four column waves of a
            // row group are then added in the fixed order wn = 0..3 by one thread per row
            if ((lane & 15) < 4) {
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
                    *reinterpret_cast<f32x2*>(reg + (mi * 16 + (lane & 15) * 4 + (lane >> 4)) * 8) = (f32x2){rs1[mi], rs2[mi]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            V2_BARRIER();
            const int tid = threadIdx.x;
            if (tid < 256) {
                const char* src = stage_base + (tid >> 7) * 4 * 4096 + (tid & 127) * 8;
                f32x2 t = *reinterpret_cast<const f32x2*>(src);
#pragma unroll
                for (int w = 1; w < 4; ++w) { const f32x2 u = *reinterpret_cast<const f32x2*>(src + w * 4096); t[0] += u[0]; t[1] += u[1]; }
                *reinterpret_cast<f32x2*>(a.stats_out + ((long)(n0 >> 8) * a.M + m0 + tid) * 2) = t;
            }
        }                                                         // (the staging areas are next written a whole main loop later)
    }
}

// =================================================================================================
// v3 ("full-line"): the same 8-wave ping-pong 256x256 kernel with the operand stream rebuilt around WHOLE 128-B LINES.
//
// Round 3 measurement (tools/dbg/gemm4w_asm.hip, profiles/r03_gemm4w_*): v2's DMA pieces are 16 rows x 64 B (32-deep sub-tiles),
// so every 128-B line of X / W is requested twice, one sub-tile apart, and the CU's 32 KiB L1 has seen 64 KiB of other lines in
// between: the L2 -> L1 fill traffic is 2x the operand bytes.  Alone, that operand stream takes as long as the MFMAs of the GEMM
// (80 us for 137 GFLOP); with pieces of 8 rows x 128 B it takes 48-52 us.
//
//   * K is consumed in 64-deep K-TILES: LDS = 2 buffers x (X[256][64] | W[256][64]) bf16 = 2 x 64 KiB, rows of 128 B, the 16-B chunk
//     index XORed with (row >> 1) & 7 (conflict-free ds_read_b128 of 16x32 fragments; on the DMA source address and the read address).
//   * A K-tile = two 32-deep halves = four phases p0..p3 of v2's shape (16 MFMAs per wave and phase, two staggered groups).  During
//     K-tile s the NEXT K-tile is requested into the other buffer, part by part as that buffer's rows retire (>= 3 phases after
//     their last read):   p0: W(s+1), 4 pieces per wave     p1: X rows {0..63, 128..191}(s+1), 2 pieces     p2: the other X rows, 2.
//   * Two counted waits per K-tile: p0 `vmcnt(4)` (X rows 64.. of THIS K-tile, read in p1; only W(s+1) may be in flight) and
//     p3 `vmcnt(2)` (W and the first X half of s+1, read in the next p0).  Both sit before the phase's first barrier, the reads they
//     cover come two barriers later (v2's RAW rule).  The first p0 after an epilogue allows for the epilogue's stores.
//   * Operand addresses: a wave-uniform base (SGPRs: tile origin + k) + per-lane 32-bit offsets that never change (8 VGPRs), so the
//     stream advance is scalar.  Interior tiles only (M, N multiples of 256: the launcher falls back to v2 otherwise).
// Tile order, epilogues (staged / LN-fold producer + consumer / XRING) and persistence are v2's.
#ifndef V3_PREISSUE
#define V3_PREISSUE 1                   /* request a tile's second K-tile before the previous tile's epilogue stores */
#endif
#ifndef V3_SCHED
#define V3_SCHED 0                      /* 0: requests per phase 4 (W) / 2 / 2 / 0;  1: 2 / 2 / 2 / 2 (tools/dbg) */
#endif
#ifndef V3_XRING_BREAK
#define V3_XRING_BREAK 1                /* tools/dbg A/B: 0 = the round-5 form (the tile loop's exit unknown to the compiler in the one-tile kernels) */
#endif
#define V3_BUF_BYTES 65536
#define V3_OPER_BYTES 32768

// WREG = 1 ("W from registers", round 6): the weight operand never touches LDS.  GemmArgs::Wp holds W once more in MFMA-FRAGMENT order
// (ldt_gemm_pack_wfrag, packed once per weight version): for every 64-column band n64 and 64-deep K-tile kt the eight 16 x 32 fragments
// (k-half h, n-tile i) as 1 KiB each, lane l's bf16x8 at + l * 16 — so a wave's whole W stream is contiguous (8 KiB per K-tile) and a
// fragment is ONE global_load_dwordx4 with a wave-uniform base.  Per wave and K-tile: 8 register loads + 4 LDS-DMA pieces (X) + 16
// ds_read_b128 instead of 8 pieces + 24 reads; both wave groups of a column band load the same fragments (2 x W through the L1).
// Registers: the 256 x 256 tile leaves no room for a second full set (128 accumulators + 16 X + 64 W spills inside the K loop), so there
// are TWO HALF-SETS of four fragments, each refilled as soon as its last MFMA has been issued: k-half 0 of K-tile s + 1 at p2 of K-tile s
// (read at p0, s + 1), k-half 1 of K-tile s at its own p0 (read at p2): two phases between a request and its first use.  The loads are
// asm statements with hand-counted waits like the DMA pieces (ISA lint R3); per wave and K-tile the VMEM queue is
//     p0: Wh1(s) x4          wait vmcnt(4):  XB(s) and Wh0(s) landed       (behind them: the four loads just issued)
//     p1: XA(s+1) x2
//     p2: XB(s+1) x2, Wh0(s+1) x4   wait vmcnt(8):  Wh1(s) landed           (behind it: XA, XB, Wh0 of s + 1)
//     p3:                    wait vmcnt(6):  XA(s+1) landed                 (behind it: XB(s+1), Wh0(s+1))
// Built for the one-tile-per-workgroup residual GEMMs (XRING: fc_o, mlp.out).  Same MFMA order per accumulator: bit-identical to the LDS form.
// KLONG is a NAME TAG only (same code): the one-tile residual GEMMs are launched as <.., .., 1, .., 1> when K >= 2048 (mlp.out) and as
// <.., .., 1, .., 0> otherwise (fc_o), so that rocprofv3 / PMC summaries price the headline's dominant kernel under a symbol of its own.
template <int EPI, int FOLD = FOLD_NONE, int XRING = 0, int WREG = 0, int KLONG = 0>
__global__ __launch_bounds__(512) void gemm_bf16_nt_256f_kernel(const GemmArgs a) {
    static_assert(!WREG || (V3_PREISSUE == 1 && V3_SCHED == 0 && XRING == 1), "WREG: one tile per workgroup; the counted waits assume the pre-issue form and the 4/2/2/0 request schedule");
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int lrow = lane & 15, lchk = lane >> 4;
    const int nkt = a.K >> 6;

    // ---- this workgroup's tile list (v2's: 8 contiguous chunks, one per XCD label; grouped order inside)
    const int tiles_n = a.N / 256;
    const int tiles = (a.M / 256) * tiles_n;
    const int G = gridDim.x, bid = blockIdx.x;
    const int nx = G < 8 ? G : 8;
    const int xcd = bid % nx, j = bid / nx;
    const int wpx = (G - xcd + nx - 1) / nx;
    const int c_lo = (int)((long)tiles * xcd / nx), c_hi = (int)((long)tiles * (xcd + 1) / nx);
    const int my_tiles = (c_hi - c_lo - j + wpx - 1) / wpx > 0 ? (c_hi - c_lo - j + wpx - 1) / wpx : 0;
    if (my_tiles == 0) return;
    const int tiles_m = a.M / 256, gm = a.group_m;
    auto tile_of = [&](int it, int& m0, int& n0) {
        const int id = c_lo + j + it * wpx;
        if (gm <= 1) { m0 = (id / tiles_n) * 256; n0 = (id % tiles_n) * 256; return; }
        const int per = gm * tiles_n, g = id / per, r = id - g * per;
        const int rows = min(gm, tiles_m - g * gm);
        m0 = (g * gm + r % rows) * 256; n0 = (r / rows) * 256;
    };

    // ---- operand stream: per-lane byte offsets (constant) + wave-uniform LDS destinations + uniform bases
    // piece = 8 rows x 128 B: lane -> row (lane >> 3) of the piece, LDS position lane & 7 holds chunk (lane & 7) ^ ((row >> 1) & 7)
    int wvo[4], xavo[2], xbvo[2];                                        // global byte offsets from the (tile row 0, k) element
    int wds[4], xads[2], xbds[2];                                        // LDS byte offsets inside a buffer (wave-uniform)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + (lane >> 3);
        wvo[q] = r * (int)a.ldw * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        wds[q] = V3_OPER_BYTES + (wave * 4 + q) * 1024;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pj = wave * 2 + q;                                     // 0..15: rows 0..63 then 128..191 (m-tiles 0-3 of the two groups)
        const int r0 = pj < 8 ? pj * 8 : 128 + (pj - 8) * 8;
        const int ra = r0 + (lane >> 3), rb = ra + 64;
        xavo[q] = ra * (int)a.ldx * 2 + (((lane & 7) ^ ((ra >> 1) & 7)) << 4);
        xbvo[q] = rb * (int)a.ldx * 2 + (((lane & 7) ^ ((rb >> 1) & 7)) << 4);
        xads[q] = r0 * 128;
        xbds[q] = (r0 + 64) * 128;
    }
    const char* sxb = nullptr;                                           // stream bases: X / W at the stream's (tile, K-tile)
    const char* swb = nullptr;
    int s_it = 0, s_kt = 0, s_inc = 128;                                 // tile iteration, K-tile inside it, bytes per advance (0 once parked)
    auto seek = [&](int it) {
        int m0, n0;
        tile_of(it, m0, n0);
        sxb = reinterpret_cast<const char*>(a.X + (long)m0 * a.ldx);
        swb = reinterpret_cast<const char*>(a.W + (long)n0 * a.ldw);
    };
    auto advance = [&]() {                                               // after the last part (X rows 64..) of a K-tile was requested
        sxb += s_inc; swb += s_inc;
        if (++s_kt == nkt) {
            s_kt = 0;
            if (++s_it < my_tiles) seek(s_it);
            else { sxb -= s_inc; swb -= s_inc; s_inc = 0; s_kt = -0x40000000; }   // parked: re-reads the last K-tile, never consumed
        }
    };
    int gk = 0;                                                          // global K-tile counter of the CONSUMER (buffer = gk & 1)
    auto issue_w = [&](char* buf, int q0 = 0, int q1 = 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q >= q0 && q < q1)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(swb + wvo[q]),
                                                 (__attribute__((address_space(3))) void*)(buf + wds[q]), 16, 0, 0);
    };
    auto issue_xa = [&](char* buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sxb + xavo[q]),
                                             (__attribute__((address_space(3))) void*)(buf + xads[q]), 16, 0, 0);
    };
    auto issue_xb = [&](char* buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sxb + xbvo[q]),
                                             (__attribute__((address_space(3))) void*)(buf + xbds[q]), 16, 0, 0);
    };
    seek(0);
    // WREG: this wave's fragment stream (wave-uniform position of the next K-tile to request) and the two register sets
    const char* wrp = nullptr;                                           // K-tile whose halves are requested next
    const char* wnext = nullptr;                                         // where the stream continues behind the tile's last K-tile
    int w_kt = 0;
    const int wlane = lane * 16, wlane2 = lane * 16 + 4096;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 w0[4], w1[4];                                                  // k-half 0 / 1: one fragment per n-tile
    auto wbase = [&](int it) {
        int m0, n0;
        tile_of(it, m0, n0);
        return reinterpret_cast<const char*>(a.Wp) + (long)((n0 >> 6) + wn) * nkt * 8192;
    };
    // (asm, not plain loads: with LDS-DMA in flight beside a register load it knows of, hipcc drains the whole queue — vmcnt(0) — in front
    //  of the load's first use.  The destinations stay unnamed until the hand-counted wait that covers them: ISA lint R3)
    auto wload_h0 = [&]() {
        const unsigned long wb = ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long)wrp >> 32)) << 32) |
                                 (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long)wrp);   // (uniform already; pins it to SGPRs for the asm)
#define WLD(dst, voff, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(voff), "s"(wb) : "memory")
        WLD(w0[0], wlane, 0); WLD(w0[1], wlane, 1024); WLD(w0[2], wlane, 2048); WLD(w0[3], wlane, 3072);
    };
    auto wload_h1 = [&]() {
        const unsigned long wb = ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long)wrp >> 32)) << 32) |
                                 (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long)wrp);
        WLD(w1[0], wlane2, 0); WLD(w1[1], wlane2, 1024); WLD(w1[2], wlane2, 2048); WLD(w1[3], wlane2, 3072);
#undef WLD
        wrp += 8192;
        if (++w_kt == nkt) { w_kt = 0; wrp = wnext; }                    // (no address arithmetic inside the K loop: wnext is ready)
    };
    if (WREG) wrp = wbase(0);

    // step-indexed (cache-cold) epilogue vectors of the FIRST tile, fetched ahead of everything else (as v2)
    const float* gate = a.gate;
    if (EPI == EPI_RESID_F32 && gate && a.step_ptr) gate += (long)(*a.step_ptr) * a.gate_step_stride;
    const int step = ((FOLD != FOLD_NONE) && a.step_ptr) ? *a.step_ptr : 0;
    const float* ln_scale = (FOLD == FOLD_PRODUCER) ? a.ln_scale + (long)step * a.ln_step_stride : nullptr;
    const float* fold_S = (FOLD == FOLD_CONSUMER) ? a.fold_S + (long)step * a.fold_step_stride : nullptr;
    const float* fold_C = (FOLD == FOLD_CONSUMER) ? a.fold_C + (long)step * a.fold_step_stride : nullptr;
    f32x4 g4_pre = {1.f, 1.f, 1.f, 1.f}, sc4_pre = {0.f, 0.f, 0.f, 0.f};
    const bool pre_ok = (EPI == EPI_RESID_F32) && gate && a.gate_sample_stride == 0;
    if (EPI == EPI_RESID_F32) {
        int m0, n0;
        tile_of(0, m0, n0);
        if (pre_ok) g4_pre = *reinterpret_cast<const f32x4*>(gate + n0 + wn * 64 + (lane & 15) * 4);
        if (FOLD == FOLD_PRODUCER) sc4_pre = *reinterpret_cast<const f32x4*>(ln_scale + n0 + wn * 64 + (lane & 15) * 4);
    }

    // prologue: K-tile 0 -> buffer 0 (and, V3_PREISSUE, K-tile 1 -> buffer 1: the invariant at every tile start is then "K-tiles 0 and 1
    // of this tile are requested", which lets a tile's SECOND K-tile be requested before the previous tile's epilogue stores — see below)
    if (WREG) {                                                          // Wh0(0) first, then the X pieces of K-tiles 0 and 1
        wload_h0();
        __builtin_amdgcn_sched_barrier(0);
        issue_xa(smem2); issue_xb(smem2); advance();
        issue_xa(smem2 + V3_BUF_BYTES); issue_xb(smem2 + V3_BUF_BYTES); advance();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                 // first X half of K-tile 0 landed (behind it: XB(0), XA(1), XB(1))
    } else {
    issue_w(smem2); issue_xa(smem2); issue_xb(smem2); advance();
#if V3_PREISSUE
    issue_w(smem2 + V3_BUF_BYTES); issue_xa(smem2 + V3_BUF_BYTES); issue_xb(smem2 + V3_BUF_BYTES); advance();
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                    // W + first X half of K-tile 0 landed (this wave's pieces)
#else
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                     // W + first X half landed (this wave's pieces)
#endif
    }
    V2_BARRIER();
    if (EPI == EPI_RESID_F32) asm volatile("" : "+v"(g4_pre), "+v"(sc4_pre));

    // per-lane LDS read bases inside a buffer: row * 128 + ((k-half * 4 + lchk) ^ ((row >> 1) & 7)) * 16; fragment i at + i * 2048
    const int sw = (lrow >> 1) & 7;
    const int xrb = (grp * 128 + lrow) * 128, wrb = V3_OPER_BYTES + (wn * 64 + lrow) * 128;
    const int xb0 = xrb + ((lchk ^ sw) << 4), xb1 = xrb + (((4 + lchk) ^ sw) << 4);
    const int wb0 = wrb + ((lchk ^ sw) << 4), wb1 = wrb + (((4 + lchk) ^ sw) << 4);

    constexpr int EPI_VMEM = (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16) ? 16
                             : (FOLD == FOLD_PRODUCER) ? 57 : 32;
    char* stage_reg = smem2 + V2_RING_BYTES + wave * 4096;
    bool prev_staged = false;

    for (int it = 0; it < my_tiles; ++it) {
        int m0, n0;
        tile_of(it, m0, n0);
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (WREG) wnext = wbase(it);                                     // behind its only tile the stream re-reads the first K-tile (never consumed)
        if (grp == 1) V2_BARRIER();                                      // stagger the two groups by one barrier

        // V3_PREISSUE: in-order VMEM retirement makes every request issued AFTER an epilogue's stores wait for them (a 128 KiB tile drains
        // in 2.5-5 us).  With K-tile 1 of the next tile requested BEFORE the stores, the first requests behind them (K-tile 2) are
        // needed 8 phases after the epilogue instead of 4: KT_FIRST issues nothing, KT_FIRST / KT_SECOND count the stores into their waits.
        enum { KT_PLAIN = 0, KT_FIRST = 1 /* first K-tile of a tile */, KT_SECOND = 2 /* second (V3_PREISSUE) */, KT_FOLD_DMA = 4, KT_FOLD_FINAL = 8 };
        auto ktile = [&](auto flags_c) {
            constexpr int FL = decltype(flags_c)::value;
            constexpr bool SKIP = V3_PREISSUE && (FL & KT_FIRST);        // K-tile 1 of this tile was requested ahead (prologue / before the epilogue)
            const char* st = smem2 + (gk & 1) * V3_BUF_BYTES;            // buffer being consumed
            char* nb = smem2 + ((gk + 1) & 1) * V3_BUF_BYTES;            // buffer being refilled (K-tile gk + 1)
            bf16x8 wf[4], xf[4];
            // ---------------- p0: half 0 — W (4 n-tiles) + X m-tiles 0..3; DMA: W of the next K-tile; wait: this K-tile's X rows 64.. ----------------
            if (!WREG) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + wb0 + i * 2048);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xb0 + i * 2048);
#if V3_SCHED == 1
            issue_w(nb, 0, 2);                                           // balanced form: 2 pieces per wave in every phase (tools/dbg A/B)
#else
            if (WREG) wload_h1();                                        // k-half 1 of THIS K-tile (its registers were last read at the previous p3)
            else if (!SKIP) issue_w(nb);
#endif
            if (FL & KT_FOLD_DMA) {
                if (wave < 2 * a.stats_parts) {
                    const float* src = a.stats_in + ((long)(wave >> 1) * a.M + m0 + (wave & 1) * 128) * 2 + lane * 4;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(stage_reg + V2_STATS_OFF), 16, 0, 0);
                }
                if (wave < 4 && lane < 32) {
                    const float* src = (wave < 2 ? fold_S : fold_C) + n0 + (wave & 1) * 128 + lane * 4;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(stage_reg + V2_SC_OFF), 16, 0, 0);
                }
            }
            if ((FL & KT_FOLD_FINAL) && wave < 4) v2_fold_finalize(smem2 + V2_RING_BYTES, tid, a.stats_parts, a.K);
            // requests issued after this K-tile's second X half (the data this wait is for): none / W of the next K-tile / (KT_FIRST with
            // V3_PREISSUE) the whole pre-requested K-tile 1; after an epilogue also its stores (clamped to the 6-bit counter: only stricter)
            // WREG (header): behind XB(s) and Wh0(s) sit the four loads of this p0; KT_FIRST: + the pre-requested X pieces of K-tile 1
            constexpr int P0W = WREG ? (SKIP ? 8 : 4) : SKIP ? 8 : V3_SCHED == 1 ? 2 : 4;
            constexpr bool AFTER_EPI = (FL & KT_FIRST) || (V3_PREISSUE && !WREG && (FL & KT_SECOND));
            if (AFTER_EPI && prev_staged) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P0W + EPI_VMEM > 63 ? 63 : P0W + EPI_VMEM) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P0W) : "memory");
            V2_BARRIER();
            if (WREG) {                                                  // the half-set becomes visible to the compiler only behind its covering wait
                asm volatile("" : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]));
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = __builtin_bit_cast(bf16x8, w0[i]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            V2_BARRIER();
            // ---------------- p1: half 0 — X m-tiles 4..7; DMA: first X half of the next K-tile ----------------
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xb0 + (4 + i) * 2048);
#if V3_SCHED == 1
            issue_w(nb, 2, 4);
#else
            if (!SKIP) issue_xa(nb);
#endif
            V2_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][4 + mi], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            V2_BARRIER();
            // ---------------- p2: half 1 — W + X m-tiles 0..3; DMA: second X half of the next K-tile ----------------
            if (!WREG) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + wb1 + i * 2048);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xb1 + i * 2048);
#if V3_SCHED == 1
            issue_xa(nb);
#else
            if (!SKIP) { issue_xb(nb); advance(); }
#endif
            if (WREG) {                                                  // k-half 0 of the NEXT K-tile (p1's MFMAs were the last readers); then Wh1(s) landed
                wload_h0();
                if (SKIP) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // (KT_FIRST requested no X piece: only Wh0(s+1) is younger)
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            V2_BARRIER();
            if (WREG) {
                asm volatile("" : "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]));
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = __builtin_bit_cast(bf16x8, w1[i]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            V2_BARRIER();
            // ---------------- p3: half 1 — X m-tiles 4..7; wait: W + first X half of the next K-tile ----------------
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xb1 + (4 + i) * 2048);
#if V3_SCHED == 1
            issue_xb(nb);
            advance();
#endif
            // (WREG: behind XA(s+1) sit XB(s+1) and the four Wh0(s+1) loads of p2; KT_FIRST: K-tile 1 is older than everything p2 waited for)
            constexpr int P3W = WREG ? (SKIP ? 4 : 6) : 2;
            if (!WREG && SKIP && prev_staged) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + EPI_VMEM > 63 ? 63 : 2 + EPI_VMEM) : "memory");   // W + X half of K-tile 1: older than the stores
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P3W) : "memory");
            V2_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][4 + mi], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            V2_BARRIER();
            ++gk;
        };
#define KTL(f) std::integral_constant<int, (f)>{}
        if (FOLD == FOLD_CONSUMER) {                              // K >= 256 (launcher): at least 4 K-tiles
            ktile(KTL(KT_FIRST)); ktile(KTL(KT_FOLD_DMA | KT_SECOND)); ktile(KTL(KT_PLAIN)); ktile(KTL(KT_FOLD_FINAL));
            for (int kt = 4; kt < nkt; ++kt) ktile(KTL(KT_PLAIN));
        } else {
            ktile(KTL(KT_FIRST)); ktile(KTL(KT_SECOND));                 // (launcher: at least 2 K-tiles)
            for (int kt = 2; kt < nkt; ++kt) ktile(KTL(KT_PLAIN));
        }
#undef KTL
        if (grp == 0) V2_BARRIER();                                      // un-stagger: both groups run the epilogue together

        prev_staged = true;                                              // interior, aligned tiles only (launcher)
#if V3_PREISSUE
        if (!XRING) {                                                    // K-tile 1 of the next tile -> the buffer the last K-tile has just left (all waves are past its reads)
            char* pb = smem2 + ((gk + 1) & 1) * V3_BUF_BYTES;
            if (!WREG) issue_w(pb);
            issue_xa(pb); issue_xb(pb); advance();
        }
#endif
        if (XRING) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            V2_BARRIER();
            v2_epilogue_staged<EPI, FOLD, 1>(a, acc, m0, n0, grp, wn, lane, lrow, lchk, gate, stage_reg, smem2 + V2_RING_BYTES, ln_scale,
                                                true, g4_pre, sc4_pre, smem2 + wave * 16384);
#if V3_XRING_BREAK
            break;                                                       // (launcher: one tile per workgroup — nothing of the stream state is live past here: -13 spilled VGPRs)
#else
            if (WREG) break;
#endif
        } else
            v2_epilogue_staged<EPI, FOLD>(a, acc, m0, n0, grp, wn, lane, lrow, lchk, gate, stage_reg, smem2 + V2_RING_BYTES, ln_scale,
                                          it == 0 && (pre_ok || FOLD == FOLD_PRODUCER), g4_pre, sc4_pre);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // drain the (unused) tail requests before exit
}

// =================================================================================================
// QKV projection + self-attention in ONE launch at the bench shape (256-token samples, head dim 64): the v3 kernel on 256 x 192 tiles.
// A tile is [q | k | v] of ONE head for ONE whole sample (rows = the sample's 256 tokens; the tile's W rows / bias / S | C columns are three
// 64-wide segments, `hidden` apart), so when its main loop ends the workgroup holds everything that (sample, head)'s attention needs:
//   * the finished projections (bias or LN-folded form applied) go to LDS as bf16 rows in the whole-head attention kernel's layouts —
//     q into the staging areas (32 KB), k | v into the operand buffer the last K-tile has just left (64 KB); the OTHER buffer keeps
//     receiving the next tile's first K-tile meanwhile (this form does not pre-request the second one: V3_PREISSUE needs both buffers);
//   * wave w (8 of them) then runs query rows [32 w, +32) over the four 64-key tiles with attn_tile_joint — the math, operand layouts and
//     summation order of attn_fwd_head_kernel — and stores O / l through its own (then dead) q rows: attn_o[B][H][256][64].
// The q | k | v rows never reach HBM (96 MB written + 96 MB read per block at B = 64) and the attention launch of the block is gone.
// Wave layout in the main loop: grp = wave >> 2 owns rows [128 grp, +128), wn = wave & 3 owns tile columns [48 wn, +48): 12 MFMAs per
// phase instead of 16, everything else (phases, barriers, counted waits, tile order) is v3's.
#include "attn_tile.h"
#define QA_W_ROWS 192
template <int FOLD>   // FOLD_NONE (block 0: bias) | FOLD_CONSUMER
__global__ __launch_bounds__(512) void gemm_qkv_attn256_kernel(const GemmArgs a) {
    static_assert(FOLD == FOLD_NONE || FOLD == FOLD_CONSUMER, "qkv+attention: plain or LN-folded consumer");
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int lrow = lane & 15, lchk = lane >> 4;
    const int nkt = a.K >> 6;
    const int hidden = a.N / 3, heads = hidden / 64;

    // ---- this workgroup's tile list (v3's: 8 contiguous chunks, one per XCD label; grouped order inside); tile = (sample, head)
    const int tiles_n = heads;
    const int tiles_m = a.M / 256;
    const int tiles = tiles_m * tiles_n;
    const int G = gridDim.x, bid = blockIdx.x;
    const int nx = G < 8 ? G : 8;
    const int xcd = bid % nx, j = bid / nx;
    const int wpx = (G - xcd + nx - 1) / nx;
    const int c_lo = (int)((long)tiles * xcd / nx), c_hi = (int)((long)tiles * (xcd + 1) / nx);
    const int my_tiles = (c_hi - c_lo - j + wpx - 1) / wpx > 0 ? (c_hi - c_lo - j + wpx - 1) / wpx : 0;
    if (my_tiles == 0) return;
    const int gm = a.group_m;
    auto tile_of = [&](int it, int& m0, int& hd) {
        const int id = c_lo + j + it * wpx;
        if (gm <= 1) { m0 = (id / tiles_n) * 256; hd = id % tiles_n; return; }
        const int per = gm * tiles_n, g = id / per, r = id - g * per;
        const int rows = min(gm, tiles_m - g * gm);
        m0 = (g * gm + r % rows) * 256; hd = r / rows;
    };

    // ---- operand stream: W = 24 pieces per K-tile (3 per wave; piece pj = rows [8 pj, +8) of the tile's 192 = segment pj / 8), X as v3
    int wvo[3], xavo[2], xbvo[2];
    int wds[3], xads[2], xbds[2];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pj = wave * 3 + q;
        const int r = pj * 8 + (lane >> 3);                              // row of the tile's W image
        const int gr = (pj >> 3) * hidden + (r & 63);                    // row of W relative to the head's first q row
        wvo[q] = gr * (int)a.ldw * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        wds[q] = V3_OPER_BYTES + pj * 1024;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pj = wave * 2 + q;
        const int r0 = pj < 8 ? pj * 8 : 128 + (pj - 8) * 8;
        const int ra = r0 + (lane >> 3), rb = ra + 64;
        xavo[q] = ra * (int)a.ldx * 2 + (((lane & 7) ^ ((ra >> 1) & 7)) << 4);
        xbvo[q] = rb * (int)a.ldx * 2 + (((lane & 7) ^ ((rb >> 1) & 7)) << 4);
        xads[q] = r0 * 128;
        xbds[q] = (r0 + 64) * 128;
    }
    const char* sxb = nullptr;
    const char* swb = nullptr;
    int s_it = 0, s_kt = 0, s_inc = 128;
    auto seek = [&](int it) {
        int m0, hd;
        tile_of(it, m0, hd);
        sxb = reinterpret_cast<const char*>(a.X + (long)m0 * a.ldx);
        swb = reinterpret_cast<const char*>(a.W + (long)hd * 64 * a.ldw);
    };
    auto advance = [&]() {
        sxb += s_inc; swb += s_inc;
        if (++s_kt == nkt) {
            s_kt = 0;
            if (++s_it < my_tiles) seek(s_it);
            else { sxb -= s_inc; swb -= s_inc; s_inc = 0; s_kt = -0x40000000; }   // parked: re-reads the last K-tile, never consumed
        }
    };
    int gk = 0;
    auto issue_w = [&](char* buf) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(swb + wvo[q]),
                                             (__attribute__((address_space(3))) void*)(buf + wds[q]), 16, 0, 0);
    };
    auto issue_xa = [&](char* buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sxb + xavo[q]),
                                             (__attribute__((address_space(3))) void*)(buf + xads[q]), 16, 0, 0);
    };
    auto issue_xb = [&](char* buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sxb + xbvo[q]),
                                             (__attribute__((address_space(3))) void*)(buf + xbds[q]), 16, 0, 0);
    };
    seek(0);
    const int step = ((FOLD != FOLD_NONE) && a.step_ptr) ? *a.step_ptr : 0;
    const float* fold_S = (FOLD == FOLD_CONSUMER) ? a.fold_S + (long)step * a.fold_step_stride : nullptr;
    const float* fold_C = (FOLD == FOLD_CONSUMER) ? a.fold_C + (long)step * a.fold_step_stride : nullptr;

    issue_w(smem2); issue_xa(smem2); issue_xb(smem2); advance();
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                     // W + first X half landed (this wave's pieces)
    V2_BARRIER();

    const int sw = (lrow >> 1) & 7;
    const int xrb = (grp * 128 + lrow) * 128, wrb = V3_OPER_BYTES + (wn * 48 + lrow) * 128;
    const int xb0 = xrb + ((lchk ^ sw) << 4), xb1 = xrb + (((4 + lchk) ^ sw) << 4);
    const int wb0 = wrb + ((lchk ^ sw) << 4), wb1 = wrb + (((4 + lchk) ^ sw) << 4);
    constexpr int EPI_VMEM = 4;                                          // VMEM ops of the epilogue behind the stream's last request: the four O stores
    bool prev_staged = false;
    const int r32 = lane & 31, hh = lane >> 5;
    AttnLaneOffs<64> lo;
    lo.init(lane);

    for (int it = 0; it < my_tiles; ++it) {
        int m0, hd;
        tile_of(it, m0, hd);
        f32x4 acc[3][8];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // (block 0: the bias of this lane's columns, fetched ahead of the main loop so that the epilogue issues no load)
        f32x4 add4[3];
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) {
            const int c = wn * 48 + ni * 16 + lchk * 4;
            add4[ni] = (FOLD == FOLD_NONE && a.bias) ? *reinterpret_cast<const f32x4*>(a.bias + (c >> 6) * hidden + hd * 64 + (c & 63)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (grp == 1) V2_BARRIER();                                      // stagger the two groups by one barrier

        enum { KT_PLAIN = 0, KT_FIRST = 1, KT_FOLD_DMA = 4, KT_FOLD_FINAL = 8 };
        auto ktile = [&](auto flags_c) {
            constexpr int FL = decltype(flags_c)::value;
            const char* st = smem2 + (gk & 1) * V3_BUF_BYTES;
            char* nb = smem2 + ((gk + 1) & 1) * V3_BUF_BYTES;
            bf16x8 wf[3], xf[4];
            // ---------------- p0
#pragma unroll
            for (int i = 0; i < 3; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + wb0 + i * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xb0 + i * 2048);
            issue_w(nb);
            if (FL & KT_FOLD_DMA) {
                char* stage_reg = smem2 + V2_RING_BYTES + wave * 4096;
                if (wave < 2 * a.stats_parts) {
                    const float* src = a.stats_in + ((long)(wave >> 1) * a.M + m0 + (wave & 1) * 128) * 2 + lane * 4;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(stage_reg + V2_STATS_OFF), 16, 0, 0);
                }
                if (wave < 6 && lane < 16) {                             // S segments -> waves 0-2's areas, C segments -> waves 3-5's
                    const int seg = wave < 3 ? wave : wave - 3;
                    const float* src = (wave < 3 ? fold_S : fold_C) + seg * hidden + hd * 64 + lane * 4;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(stage_reg + V2_SC_OFF), 16, 0, 0);
                }
            }
            if ((FL & KT_FOLD_FINAL) && wave < 4) v2_fold_finalize(smem2 + V2_RING_BYTES, tid, a.stats_parts, a.K);
            constexpr int P0W = 3;
            if ((FL & KT_FIRST) && prev_staged) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P0W + EPI_VMEM) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P0W) : "memory");
            V2_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 3; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            V2_BARRIER();
            // ---------------- p1
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xb0 + (4 + i) * 2048);
            issue_xa(nb);
            V2_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 3; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][4 + mi], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            V2_BARRIER();
            // ---------------- p2
#pragma unroll
            for (int i = 0; i < 3; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + wb1 + i * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xb1 + i * 2048);
            issue_xb(nb); advance();
            V2_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 3; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            V2_BARRIER();
            // ---------------- p3
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xb1 + (4 + i) * 2048);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");             // W + first X half of the next K-tile
            V2_BARRIER();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 3; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][4 + mi], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            V2_BARRIER();
            ++gk;
        };
#define KTL(f) std::integral_constant<int, (f)>{}
        if (FOLD == FOLD_CONSUMER) {                                     // K >= 256 (launcher): at least 4 K-tiles
            ktile(KTL(KT_FIRST)); ktile(KTL(KT_FOLD_DMA)); ktile(KTL(KT_PLAIN)); ktile(KTL(KT_FOLD_FINAL));
            for (int kt = 4; kt < nkt; ++kt) ktile(KTL(KT_PLAIN));
        } else {
            ktile(KTL(KT_FIRST));
            for (int kt = 1; kt < nkt; ++kt) ktile(KTL(KT_PLAIN));
        }
#undef KTL
        if (grp == 0) V2_BARRIER();                                      // un-stagger: both groups run the epilogue together
        prev_staged = true;

        // ---- epilogue 1: finish the projection; bf16 rows -> q (staging areas) | k | v (the buffer the last K-tile has just left)
        char* stage_base = smem2 + V2_RING_BYTES;
        char* kvb = smem2 + ((gk + 1) & 1) * V3_BUF_BYTES;               // K: [256 keys][128 B] at + 0, V at + 32 KiB (the other buffer holds the next tile's K-tile 0)
        f32x4 s4[3];
        float rr[8], nm[8];
        if (FOLD == FOLD_CONSUMER) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int R = grp * 128 + mi * 16 + lrow;
                const f32x2 t = *reinterpret_cast<const f32x2*>(stage_base + (R >> 7) * 4096 + V2_STATS_OFF + (R & 127) * 8);
                rr[mi] = t[0]; nm[mi] = t[1];
            }
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) {
                const int c = wn * 48 + ni * 16 + lchk * 4;
                s4[ni] = *reinterpret_cast<const f32x4*>(stage_base + (c >> 6) * 4096 + V2_SC_OFF + (c & 63) * 4);
                add4[ni] = *reinterpret_cast<const f32x4*>(stage_base + (3 + (c >> 6)) * 4096 + V2_SC_OFF + (c & 63) * 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            V2_BARRIER();                                                // every wave has its statistics / S | C: the staging areas become the q rows
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int R = grp * 128 + mi * 16 + lrow;
            const int swr = (R >> 1) & 7;
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) {
                f32x4 v = acc[ni][mi];
                if (FOLD == FOLD_CONSUMER) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * rr[mi] + (nm[mi] * s4[ni][r] + add4[ni][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += add4[ni][r];
                }
                const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                const int c = wn * 48 + ni * 16 + lchk * 4, seg = c >> 6, cc = c & 63;
                const int sz = seg == 2 ? ((R >> 1) & 1) << 2 : swr;        // V rows: the transposed-read swizzle; q, k rows: the row-read one
                char* dst = (seg == 0 ? stage_base : kvb + (seg - 1) * 32768) + R * 128 + (((cc >> 3) ^ sz) << 4) + (cc & 7) * 2;
                *reinterpret_cast<bf16x4*>(dst) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        V2_BARRIER();                                                    // the head's q | k | v are complete

        // ---- epilogue 2: wave w = query rows [32 w, +32) over the four key tiles (attn_fwd_head_kernel's loop)
        {
            const int q0 = wave * 32;
            bf16x8 qf[4];
#pragma unroll
            for (int sI = 0; sI < 4; ++sI)
                qf[sI] = *reinterpret_cast<const bf16x8*>(stage_base + (q0 + r32) * 128 + (((hh + 2 * sI) ^ ((r32 >> 1) & 7)) << 4));
            f32x16 oacc[2];
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
            float m_run = -INFINITY, l_run = 0.f;
            // software-pipelined over the four key tiles: the S^T MFMAs of tile t + 1 are issued before the softmax of tile t, so the matrix
            // pipe works under this wave's own softmax VALU (two score accumulator pairs; same math and summation order per tile)
            f32x16 sa0, sa1, sb0, sb1;
            const char* Kt = kvb;
            const char* Vt = kvb + 32768;
            const float csc = a.attn_scale_log2e;
            attn_scores<64>(Kt, qf, sa0, sa1, lo);
            __builtin_amdgcn_sched_barrier(0);
            attn_scores<64>(Kt + 8192, qf, sb0, sb1, lo);
            __builtin_amdgcn_sched_barrier(0);
            attn_softmax_pv<64>(Vt, sa0, sa1, oacc, m_run, l_run, 0, 256, hh, csc, lo);
            __builtin_amdgcn_sched_barrier(0);
            attn_scores<64>(Kt + 2 * 8192, qf, sa0, sa1, lo);
            __builtin_amdgcn_sched_barrier(0);
            attn_softmax_pv<64>(Vt + 8192, sb0, sb1, oacc, m_run, l_run, 64, 256, hh, csc, lo);
            __builtin_amdgcn_sched_barrier(0);
            attn_scores<64>(Kt + 3 * 8192, qf, sb0, sb1, lo);
            __builtin_amdgcn_sched_barrier(0);
            attn_softmax_pv<64>(Vt + 2 * 8192, sa0, sa1, oacc, m_run, l_run, 128, 256, hh, csc, lo);
            __builtin_amdgcn_sched_barrier(0);
            attn_softmax_pv<64>(Vt + 3 * 8192, sb0, sb1, oacc, m_run, l_run, 192, 256, hh, csc, lo);
            // O / l through this wave's own q rows (dead: the fragments are in registers), whole rows out
            char* ost = stage_base + q0 * 128;
            const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int chn = d * 4 + g;
                    const bf16x4 pk = {(bf16_t)(oacc[d][4 * g + 0] * inv), (bf16_t)(oacc[d][4 * g + 1] * inv),
                                       (bf16_t)(oacc[d][4 * g + 2] * inv), (bf16_t)(oacc[d][4 * g + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(ost + r32 * 128 + ((chn ^ (r32 & 7)) << 4) + hh * 8) = pk;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bf16_t* ob = a.attn_o + (((long)(m0 >> 8) * heads + hd) * 256 + q0) * 64;
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const int row = p4 * 8 + (lane >> 3), ch = lane & 7;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(ost + row * 128 + ((ch ^ (row & 7)) << 4));
                *reinterpret_cast<bf16x8*>(ob + (long)row * 64 + ch * 8) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        V2_BARRIER();                                                    // k | v (the next K-tile 1's buffer) and the staging areas are free again
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- W in MFMA-fragment order for the WREG form of the 256-tile kernel: Wp[((n64 * (K/64) + kt) * 8 + h * 4 + i) * 64 + lane][8] =
// W[n64 * 64 + i * 16 + (lane & 15)][kt * 64 + h * 32 + (lane >> 4) * 8 .. + 8]   (one thread per 16-byte chunk)
__global__ __launch_bounds__(256) void pack_wfrag_kernel(const bf16_t* __restrict__ W, long ldw, int N, int K, bf16_t* __restrict__ Wp) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;                 // chunk index in Wp
    const int nkt = K >> 6;
    if (c >= (long)N * K / 8) return;
    const int lane = (int)(c & 63), f = (int)((c >> 6) & 7);
    const long t = c >> 9;                                               // n64 * nkt + kt
    const int kt = (int)(t % nkt), n64 = (int)(t / nkt);
    const int n = n64 * 64 + (f & 3) * 16 + (lane & 15), k = kt * 64 + (f >> 2) * 32 + (lane >> 4) * 8;
    *reinterpret_cast<bf16x8*>(Wp + c * 8) = *reinterpret_cast<const bf16x8*>(W + (long)n * ldw + k);
}
int ldt_gemm_pack_wfrag_launch(const bf16_t* W, long ldw, int N, int K, bf16_t* Wp, hipStream_t stream) {
    LDT_REQUIRE(W && Wp && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0 && ldw >= K && ldw % 8 == 0 && ldt_aligned16(W) && ldt_aligned16(Wp), LDT_ESHAPE,
                "pack_wfrag: N=%d and K=%d must be multiples of 64, rows 16-byte aligned (ldw=%ld)", N, K, ldw);
    const long chunks = (long)N * K / 8;
    hipLaunchKernelGGL(pack_wfrag_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, W, ldw, N, K, Wp);
    return ldt_check_launch("pack_wfrag");
}
// tools/dbg + tests: ldt_dbg_gemm_wreg(1) makes every 256-tile launch WITHOUT a caller-packed Wp pack its W on the fly into a cache keyed by
// (pointer, shape) — never invalidated: the caller must not change those weights afterwards.  The product path passes Wp itself.
static std::atomic<int> g_dbg_wreg{-1};
extern "C" int ldt_dbg_gemm_wreg(int32_t on) { g_dbg_wreg.store(on); return LDT_OK; }
static const bf16_t* dbg_wfrag_cached(const bf16_t* W, long ldw, int N, int K, hipStream_t stream) {
    static std::mutex mu;
    static std::map<std::tuple<const void*, long, int, int>, bf16_t*> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_tuple((const void*)W, ldw, N, K);
    auto itc = cache.find(key);
    if (itc != cache.end()) return itc->second;
    bf16_t* wp = nullptr;
    if (hipMalloc((void**)&wp, (size_t)N * K * 2) != hipSuccess) return nullptr;
    if (ldt_gemm_pack_wfrag_launch(W, ldw, N, K, wp, stream) != LDT_OK) { (void)hipFree(wp); return nullptr; }
    cache[key] = wp;
    return wp;
}

// rows per group of the grouped tile order (tools/dbg sets it at run time; LDT_GEMM_GM at start-up)
static std::atomic<int> g_group_m{-1};
extern "C" int ldt_dbg_gemm_group_m(int32_t gm) { g_group_m.store(gm); return LDT_OK; }
static std::atomic<int> g_dbg_epi{-1};
extern "C" int ldt_dbg_gemm_epi(int32_t bits) { g_dbg_epi.store(bits); return LDT_OK; }   // tools/dbg/epi_ablate.py

// The 256-tile kernel takes interior, aligned tiles only (M, N multiples of 256, K a multiple of 64 with >= 2 K-tiles, 16-byte rows);
// everything else belongs to the mid-size / small-tile kernels.
static bool gemm256_takes(int epi, const GemmArgs* a) {
    return a->K % 64 == 0 && a->K >= 128 && a->M % 256 == 0 && a->N % 256 == 0 && a->ldo % 8 == 0 &&
           (epi != EPI_RESID_F32 || (a->ldr % 4 == 0 && (!a->gate || a->gate_sample_stride % 4 == 0))) &&
           (epi != EPI_RELU_BF16 || !a->skip || a->lds_ % 4 == 0);
}

template <int EPI, int FOLD = FOLD_NONE>
static int launch_256(const GemmArgs* a_in, hipStream_t stream) {
    // Tile order: wide outputs (QKV: 12 column tiles, MLP-up: 16) are swept in groups of row panels, so an XCD's 32 workgroups hold a
    // block of tiles (round 2-5: 8 x 4 = 8 X panels + 4 W panels in its 4 MiB L2; round 6: 4 x 8) instead of 2 x 16 — the W panel set is
    // then re-streamed from the fabric once per group of row panels, not once per 2 (profiles/: MLP-up fetched 2.5x its unique bytes in
    // row-major order).  LDT_GEMM_GM overrides (LDT_QKV_GM: the fused QKV + attention kernel alone).
    static const int gm_env = getenv("LDT_GEMM_GM") ? atoi(getenv("LDT_GEMM_GM")) : -1;
    const int gm_dbg = g_group_m.load();
    GemmArgs a_copy = *a_in;
    const int tn = a_in->N / 256, tm = a_in->M / 256;
    // Round 6: groups of 4 (an XCD's 32 workgroups = 4 row panels x 8 column tiles per round) instead of 8 — re-swept on the whole loop with the
    // fused QKV + attention kernel and the LN-folded MLP-up in place: 8 / 4 / 1 row panels = 10.76-10.79 / 10.59-10.64 / 10.56-10.60 ms per SDE
    // step on one box (MLP-up alone -1.1 %, the QKV kernel -0.4 %; tools/dbg/gm_loop_sweep.py, profiles/r06_tile_order_sweep.txt).
    a_copy.group_m = gm_dbg >= 0 ? gm_dbg : gm_env >= 0 ? gm_env : (tn >= 8 && tm >= 8) ? 4 : 1;
    // residual rows of a one-tile workgroup through the operand ring (v2_epilogue_staged XRING, kernel <.., .., 1>): needs the exact VMEM op
    // count of the epilogue (no per-sample gate loads, no debug skips) and 16-B aligned rows.  LDT_RESID_RING=0: A/B runs.
    static const bool xring_on = !(getenv("LDT_RESID_RING") && atoi(getenv("LDT_RESID_RING")) == 0);
    const bool xring = (EPI == EPI_RESID_F32 && xring_on && a_in->resid && ldt_aligned16(a_in->resid) && (!a_in->gate || a_in->gate_sample_stride == 0));
    static const int dbg_env = getenv("LDT_DBG_EPI") ? atoi(getenv("LDT_DBG_EPI")) : 0;
    a_copy.dbg = g_dbg_epi.load() >= 0 ? g_dbg_epi.load() : dbg_env;
    const GemmArgs* a = &a_copy;
    LDT_REQUIRE(gemm256_takes(EPI, a), LDT_ESHAPE, "gemm256: M=%d N=%d must be multiples of 256, K=%d of 64 (>= 128), rows 16-byte aligned", a->M, a->N, a->K);
    // W from registers (kernel <.., .., 1, 1>: the one-tile-per-workgroup residual GEMMs): the caller's fragment-order copy
    constexpr bool WREG_BUILT = (EPI == EPI_RESID_F32) && V3_PREISSUE == 1 && V3_SCHED == 0;   // (its counted waits assume the shipped request schedule: the tools/dbg schedule builds go without it)
    static const int wreg_env = getenv("LDT_GEMM_WREG") ? atoi(getenv("LDT_GEMM_WREG")) : -1;      // 0: off even when Wp is given; 1: pack on the fly (tools/dbg)
    const int wreg_dbg = g_dbg_wreg.load() >= 0 ? g_dbg_wreg.load() : wreg_env;
    const int tiles = tm * tn;
    static const int cap = getenv("LDT_GEMM_GRID") ? atoi(getenv("LDT_GEMM_GRID")) : LDT_NUM_CUS;   // tools/dbg: > 256 = non-persistent
    const int lim = (a->max_wgs > 0 && a->max_wgs < cap) ? a->max_wgs : cap;
    const int grid = tiles < lim ? tiles : lim;                          // one persistent workgroup per CU (or per CU of this stream's share)
    if constexpr (EPI == EPI_RESID_F32) {
        if (xring && grid == tiles && a->dbg == 0) {                     // every workgroup has exactly one tile: the ring is idle in its epilogue
            if (WREG_BUILT && wreg_dbg == 1 && !a_copy.Wp) a_copy.Wp = dbg_wfrag_cached(a->W, a->ldw, a->N, a->K, stream);
            const bool wreg = WREG_BUILT && a_copy.Wp && wreg_dbg != 0 && ldt_aligned16(a_copy.Wp);
            const bool klong = a->K >= 2048;                             // symbol tag: mlp.out vs fc_o (see the kernel's template comment)
#define LAUNCH_XR(W, KL)                                                                                                          \
    do {                                                                                                                          \
        LDT_ENSURE_LDS((&gemm_bf16_nt_256f_kernel<EPI, FOLD, 1, W, KL>), V2_LDS_BYTES, "gemm256f");                                \
        hipLaunchKernelGGL((gemm_bf16_nt_256f_kernel<EPI, FOLD, 1, W, KL>), dim3(grid), dim3(512), V2_LDS_BYTES, stream, *a);     \
    } while (0)
            if constexpr (WREG_BUILT) {
                if (wreg) {
                    if (klong) LAUNCH_XR(1, 1); else LAUNCH_XR(1, 0);
                    return ldt_check_launch("gemm_bf16_nt_256f");
                }
            }
            if (klong) LAUNCH_XR(0, 1); else LAUNCH_XR(0, 0);
#undef LAUNCH_XR
            return ldt_check_launch("gemm_bf16_nt_256f");
        }
    }
    LDT_ENSURE_LDS((&gemm_bf16_nt_256f_kernel<EPI, FOLD>), V2_LDS_BYTES, "gemm256f");
    hipLaunchKernelGGL((gemm_bf16_nt_256f_kernel<EPI, FOLD>), dim3(grid), dim3(512), V2_LDS_BYTES, stream, *a);
    return ldt_check_launch("gemm_bf16_nt_256f");
}

static int gemm_variant_env();
// QKV projection + self-attention in one launch at 256 tokens (gemm_qkv_attn256_kernel): head dim 64, N = 3 * hidden (hidden % 64 == 0),
// whole samples (M % 256 == 0), enough (sample, head) tiles to fill 5/8 of the workgroups the launch may use.  `folded`: a = the LN-folded
// consumer's arguments (statistics per 256 columns).  -> true when this kernel took the launch.  LDT_QKV_ATTN256=0: off (A/B).
bool ldt_gemm_qkv_attn256_try(const GemmArgs* a_in, int tokens, int head_dim, bool folded, hipStream_t stream, int* status) {
    static const bool on = !(getenv("LDT_QKV_ATTN256") && atoi(getenv("LDT_QKV_ATTN256")) == 0);
    const GemmArgs& g = *a_in;
    if (!on || gemm_variant_env() != 0 || tokens != 256 || head_dim != 64 || !g.attn_o) return false;
    if (g.N % 192 != 0 || (g.N / 3) % 64 != 0 || g.M % 256 != 0 || g.K % 64 != 0 || g.K < (folded ? 256 : 128)) return false;
    if (folded && (g.stats_parts <= 0 || g.stats_parts > 4 || g.stats_parts * 256 != g.K || !g.stats_in || !g.fold_S || !g.fold_C ||
                   !ldt_aligned16(g.stats_in) || !ldt_aligned16(g.fold_S) || !ldt_aligned16(g.fold_C) || g.fold_step_stride % 4 != 0))
        return false;
    if (!ldt_aligned16(g.X) || !ldt_aligned16(g.W) || !ldt_aligned16(g.attn_o) || (g.bias && !ldt_aligned16(g.bias)) || g.ldx % 8 != 0 || g.ldw % 8 != 0 ||
        g.ldx < g.K || g.ldw < g.K)
        return false;
    const int tm = g.M / 256, tn = (g.N / 3) / 64;
    const long tiles = (long)tm * tn;
    const int lim = (g.max_wgs > 0 && g.max_wgs < LDT_NUM_CUS) ? g.max_wgs : LDT_NUM_CUS;
    if (tiles * 8 < (long)lim * 5) return false;
    GemmArgs a = g;
    static const int gm_env = getenv("LDT_QKV_GM") ? atoi(getenv("LDT_QKV_GM")) : getenv("LDT_GEMM_GM") ? atoi(getenv("LDT_GEMM_GM")) : -1;   // tools/dbg
    a.group_m = gm_env >= 0 ? gm_env : (tn >= 8 && tm >= 8) ? 4 : 1;     // (4 since round 6: see launch_256)
    const int grid = tiles < lim ? (int)tiles : lim;
    auto launch = [&]() -> int {
        if (folded) {
            LDT_ENSURE_LDS((&gemm_qkv_attn256_kernel<FOLD_CONSUMER>), V2_LDS_BYTES, "gemm_qkv_attn256");
            hipLaunchKernelGGL((gemm_qkv_attn256_kernel<FOLD_CONSUMER>), dim3(grid), dim3(512), V2_LDS_BYTES, stream, a);
        } else {
            LDT_ENSURE_LDS((&gemm_qkv_attn256_kernel<FOLD_NONE>), V2_LDS_BYTES, "gemm_qkv_attn256");
            hipLaunchKernelGGL((gemm_qkv_attn256_kernel<FOLD_NONE>), dim3(grid), dim3(512), V2_LDS_BYTES, stream, a);
        }
        return ldt_check_launch("gemm_qkv_attn256");
    };
    *status = launch();
    return true;
}

// LN-folding launches.  Large batches: the 256-tile kernel (statistics per 256 columns).  Small batches — all four GEMMs of a Score block
// (N = D, 3D, F) below ldt_gemm_launch's 5/8 rule — fold through the mid-size tile kernel (gemm_mid.hip, statistics per 32 columns:
// stats[D / 32][M][2]) when it takes every one of them in a folded form; otherwise the LayerNorm kernels run.
static int gemm_variant_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LDT_GEMM_FORCE"); v = e ? atoi(e) : 0; }
    return v;
}
bool ldt_gemm_lnfold_v1_route(int M, int D, int F, int max_wgs) {
    const int lim = (max_wgs > 0 && max_wgs < LDT_NUM_CUS) ? max_wgs : LDT_NUM_CUS;
    auto small = [&](int N) { return (long)((M + 255) / 256) * ((N + 255) / 256) * 8 < (long)lim * 5; };
    if (!(gemm_variant_env() == 0 && M % 128 == 0 && D % 64 == 0 && F % 64 == 0 && D <= 1024 && small(D) && small(3 * D) && small(F))) return false;
    // the folded residual producers run 64 x 128 tiles (statistics per 32 columns need whole 128-column slabs): below 5/8 of the workgroups
    // they leave the chip half empty and the LayerNorm launches they replace are cheaper — M = 1024 (B = 32 x 32 tokens): folded 1.788 ms per
    // SDE step against 1.670 with the 64 x 64 plain forms + LayerNorm kernels (profiles/r06_c5_ln_fold_decision.txt); M = 2048: 256 tiles, folded wins
    if ((long)(M / 64) * (D / 128) * 8 < (long)lim * 5) return false;
    return ldt_gemm_mid_lnfold_takes(EPI_RESID_F32, M, D, D) && ldt_gemm_mid_lnfold_takes(EPI_RESID_F32, M, D, F) &&
           ldt_gemm_mid_lnfold_takes(EPI_BF16, M, 3 * D, D) && ldt_gemm_mid_lnfold_takes(EPI_GELU_BF16, M, F, D);
}

int ldt_gemm_lnfold_launch(int epi, const GemmArgs* a, hipStream_t stream) {
    // route: stats_parts says which statistics layout the caller's buffers use — K / 256 (N / 256 for the producer) parts: the 256-tile
    // kernel; K / 32 (N / 32): the mid-size tile kernel (ldt_gemm_lnfold_v1_route)
    const int width = epi == EPI_RESID_F32 ? a->N : a->K;
    const bool v1 = a->stats_parts > 0 && a->stats_parts * 32 == width && a->stats_parts * 256 != width;
    if (v1) LDT_REQUIRE(a->M > 0 && a->M % 128 == 0 && a->N % 64 == 0 && a->K >= 128 && a->K % BK == 0, LDT_ESHAPE,
                        "gemm_lnfold (v1 route): M=%d must be a multiple of 128, N=%d of 64, K=%d >= 128, K %% 64 == 0", a->M, a->N, a->K);
    else LDT_REQUIRE(a->M > 0 && a->N > 0 && a->K >= 256 && a->M % 256 == 0 && a->N % 256 == 0 && a->K % BK == 0, LDT_ESHAPE,
                     "gemm_lnfold: M=%d N=%d must be multiples of 256 and K=%d >= 256, K %% 64 == 0", a->M, a->N, a->K);
    LDT_REQUIRE(a->ldx % 8 == 0 && a->ldw % 8 == 0 && a->ldx >= a->K && a->ldw >= a->K && a->ldo % 8 == 0 && ldt_aligned16(a->X) &&
                ldt_aligned16(a->W) && ldt_aligned16(a->out), LDT_EALIGN, "gemm_lnfold: operands must be 16-byte aligned (ldx=%ld ldw=%ld ldo=%ld)",
                a->ldx, a->ldw, a->ldo);
    if (epi == EPI_RESID_F32) {                                          // producer
        LDT_REQUIRE(a->resid && a->ldr % 4 == 0 && ldt_aligned16(a->resid), LDT_EALIGN, "gemm_lnfold: resid missing/misaligned");
        LDT_REQUIRE(!a->gate || (a->rows_per_sample > 0 && ldt_aligned16(a->gate) && a->gate_sample_stride % 4 == 0 && a->gate_step_stride % 4 == 0),
                    LDT_EARG, "gemm_lnfold: gate needs rows_per_sample>0 and 16-byte aligned strides");
        LDT_REQUIRE(!a->bias || ldt_aligned16(a->bias), LDT_EALIGN, "gemm_lnfold: bias must be 16-byte aligned");
        LDT_REQUIRE(a->xs && a->ln_scale && a->stats_out && a->ldxs % 4 == 0 && a->ldxs >= a->N && ldt_aligned16(a->xs) &&
                    ldt_aligned16(a->ln_scale) && a->ln_step_stride % 4 == 0 && ldt_aligned16(a->stats_out), LDT_EARG,
                    "gemm_lnfold: producer needs xs / ln_scale / stats_out (16-byte aligned)");
        if (v1) {
            int st = LDT_OK;
            if (ldt_gemm_mid_lnfold_try(EPI_RESID_F32, a, stream, &st)) return st;   // mid-size tile kernel (gemm_mid.hip)
            ldt_set_error("gemm_lnfold: statistics per 32 columns are the mid-size tile kernel's; it does not take M=%d N=%d K=%d as a producer", a->M, a->N, a->K);
            return LDT_ESHAPE;
        }
        return launch_256<EPI_RESID_F32, FOLD_PRODUCER>(a, stream);
    }
    LDT_REQUIRE(epi == EPI_BF16 || epi == EPI_GELU_BF16, LDT_EARG, "gemm_lnfold: epilogue %d has no folded form", epi);
    LDT_REQUIRE(a->stats_in && a->fold_S && a->fold_C && ldt_aligned16(a->stats_in) && ldt_aligned16(a->fold_S) && ldt_aligned16(a->fold_C) &&
                a->fold_step_stride % 4 == 0, LDT_EARG, "gemm_lnfold: consumer needs stats_in, fold_S, fold_C (16-byte aligned)");
    if (v1) {
        LDT_REQUIRE(a->stats_parts <= 32, LDT_ESHAPE, "gemm_lnfold (v1 route): K=%d > 1024 input channels", a->K);
        int st = LDT_OK;
        if (ldt_gemm_mid_lnfold_try(epi, a, stream, &st)) return st;
        ldt_set_error("gemm_lnfold: statistics per 32 columns are the mid-size tile kernel's; it does not take M=%d N=%d K=%d as a consumer", a->M, a->N, a->K);
        return LDT_ESHAPE;
    }
    LDT_REQUIRE(a->stats_parts >= 1 && a->stats_parts <= 4 && a->stats_parts * 256 == a->K, LDT_EARG,
                "gemm_lnfold: consumer needs stats_in[K/256 <= 4][M][2] (or [K/32][M][2] for the small-batch kernels); K=%d parts=%d", a->K, a->stats_parts);
    return epi == EPI_BF16 ? launch_256<EPI_BF16, FOLD_CONSUMER>(a, stream) : launch_256<EPI_GELU_BF16, FOLD_CONSUMER>(a, stream);
}

// LDT_GEMM_FORCE=128|256 pins the variant (A/B runs); default: see ldt_gemm_launch.
static int gemm_variant() { return gemm_variant_env(); }

int ldt_gemm_launch(int epi, const GemmArgs* a, hipStream_t stream) {
    LDT_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, LDT_ESHAPE, "gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    LDT_REQUIRE(a->K % BK == 0, LDT_ESHAPE, "gemm: K=%d must be a multiple of %d (pad activations/weights)", a->K, BK);
    LDT_REQUIRE(a->ldx % 8 == 0 && a->ldw % 8 == 0 && ldt_aligned16(a->X) && ldt_aligned16(a->W), LDT_EALIGN,
                "gemm: X/W rows must be 16-byte aligned (ldx=%ld ldw=%ld)", a->ldx, a->ldw);
    LDT_REQUIRE(a->ldx >= a->K && a->ldw >= a->K, LDT_ESHAPE, "gemm: leading dims smaller than K");
    LDT_REQUIRE(a->ldo % 4 == 0 && ldt_aligned16(a->out), LDT_EALIGN, "gemm: out must be 16-byte aligned, ldo%%4==0 (ldo=%ld)", a->ldo);
    if (epi == EPI_RESID_F32) {
        LDT_REQUIRE(a->resid && a->ldr % 4 == 0 && ldt_aligned16(a->resid), LDT_EALIGN, "gemm: resid missing/misaligned");
        LDT_REQUIRE(!a->gate || (a->rows_per_sample > 0 && ldt_aligned16(a->gate) && a->gate_sample_stride % 4 == 0 && a->gate_step_stride % 4 == 0),
                    LDT_EARG, "gemm: gate needs rows_per_sample>0 and 16-byte aligned strides");
    }
    LDT_REQUIRE(!a->bias || ldt_aligned16(a->bias), LDT_EALIGN, "gemm: bias must be 16-byte aligned");
    if (gemm_variant() == 0) {
        // mid-size problems (gemm_mid.hip): 128 x 256 / 128 x 192 / 128 x 128 / 64 x 128 tiles with dedicated loader waves, when the 256^2
        // persistent kernel would leave CUs idle (its 5/8 rule below) — the 1-4k-row batches
        const int t256 = ((a->M + 255) / 256) * ((a->N + 255) / 256);
        const int lim = (a->max_wgs > 0 && a->max_wgs < LDT_NUM_CUS) ? a->max_wgs : LDT_NUM_CUS;
        const bool big = t256 * 8 >= lim * 5 && a->N > 128;
        if (!big) {
            const int shape = ldt_gemm_mid_shape(epi, a);
            if (shape) return ldt_gemm_mid_launch(epi, shape, a, stream);
        }
    }
    const int tiles256 = ((a->M + 255) / 256) * ((a->N + 255) / 256);
    const int force = gemm_variant();
    // 256^2 persistent kernel when its tiles fill at least 5/8 of the workgroups this launch may use (all CUs, or a
    // sub-batch stream's share): at exactly half (M = 8192, N = 1024: 128 tiles on 256 CUs) the 128^2 kernel on every CU
    // is as fast (K = 1024) or 15 % faster (K = 4096).
    const int lim256 = (a->max_wgs > 0 && a->max_wgs < LDT_NUM_CUS) ? a->max_wgs : LDT_NUM_CUS;
    // (N <= 128 — the Compressor's 128-channel convs over millions of point rows — would leave half of every 256-wide tile
    //  empty: the 128^2 kernel streams those 5.7 % faster end to end, tools/dbg/c4_chunks.py)
    if ((force == 256 || (force == 0 && tiles256 * 8 >= lim256 * 5 && a->N > 128)) && gemm256_takes(epi, a)) {
        switch (epi) {
            case EPI_F32: return launch_256<EPI_F32>(a, stream);
            case EPI_BF16: return launch_256<EPI_BF16>(a, stream);
            case EPI_GELU_BF16: return launch_256<EPI_GELU_BF16>(a, stream);
            case EPI_RELU_BF16: return launch_256<EPI_RELU_BF16>(a, stream);
            case EPI_RESID_F32: return launch_256<EPI_RESID_F32>(a, stream);
            default: ldt_set_error("gemm: unknown epilogue %d", epi); return LDT_EARG;
        }
    }
    // v1 tile shape: the largest of 128x128 / 128x64 / 64x64 that still gives every CU two tiles
    auto ntiles = [&](int bm, int bn) { return (long)((a->M + bm - 1) / bm) * ((a->N + bn - 1) / bn); };
    static const int v1_shape = getenv("LDT_GEMM_V1_SHAPE") ? atoi(getenv("LDT_GEMM_V1_SHAPE")) : -1;   // tools/dbg
    // this 2-phase kernel hides a stage's load latency only across co-resident workgroups: want >= 2 tiles per CU
    // (M = 2048: QKV 25.8 -> 22.9 us with 128x64, fc_o 14.4 -> 12.0 and mlp.out 47.7 -> 40.0 us with 64x64 tiles)
    const int shape = v1_shape >= 0 ? v1_shape : (force == 128 || ntiles(128, 128) >= 2 * LDT_NUM_CUS) ? 0 : (ntiles(128, 64) >= 2 * LDT_NUM_CUS ? 1 : 2);
    dim3 block(256);
    // 3 stages only for 64x64 tiles (48 KB of LDS, still 3 workgroups per CU; M = 2048: fc_o 13.5 -> 12.3, mlp.out 38.6 ->
    // 31.3 us; a 4th stage measured equal): at 128x64 / 128x128 the third buffer costs a co-resident workgroup and loses 20-40 %
    static const int v1_stages_env = getenv("LDT_GEMM_V1_STAGES") ? atoi(getenv("LDT_GEMM_V1_STAGES")) : 0;   // tools/dbg
    const int v1_stages = v1_stages_env ? v1_stages_env : (shape == 2 ? 3 : 2);
    static const int v1_map_env = getenv("LDT_GEMM_V1_MAP") ? atoi(getenv("LDT_GEMM_V1_MAP")) : -1;         // tools/dbg: 0 / 1 force
    GemmArgs a_v1 = *a;
    {
        const int bm = (shape == 3) ? 256 : (shape == 2 ? 64 : 128), bn = (shape == 0 || shape == 3 || shape == 4) ? 128 : 64;
        const long tm = (a->M + bm - 1) / bm, tn = (a->N + bn - 1) / bn;
        a_v1.col_major = v1_map_env >= 0 ? v1_map_env : (tn >= 3 * tm ? 1 : 0);
    }
    a = &a_v1;
#define LAUNCH_V1(E)                                                                                                     \
    do {                                                                                                                 \
        if (shape == 3) hipLaunchKernelGGL((gemm_bf16_nt_kernel<E, 256, 128, 3, 8>), dim3((unsigned)ntiles(256, 128)), dim3(512), 0, stream, *a); \
        else if (shape == 4) hipLaunchKernelGGL((gemm_bf16_nt_kernel<E, 128, 128, 4, 8>), dim3((unsigned)ntiles(128, 128)), dim3(512), 0, stream, *a); \
        else if (v1_stages >= 3) {                                                                                       \
            if (shape == 0) hipLaunchKernelGGL((gemm_bf16_nt_kernel<E, 128, 128, 3>), dim3((unsigned)ntiles(128, 128)), block, 0, stream, *a); \
            else if (shape == 1) hipLaunchKernelGGL((gemm_bf16_nt_kernel<E, 128, 64, 3>), dim3((unsigned)ntiles(128, 64)), block, 0, stream, *a); \
            else hipLaunchKernelGGL((gemm_bf16_nt_kernel<E, 64, 64, 3>), dim3((unsigned)ntiles(64, 64)), block, 0, stream, *a);  \
        } else if (shape == 0) hipLaunchKernelGGL((gemm_bf16_nt_kernel<E, 128, 128>), dim3((unsigned)ntiles(128, 128)), block, 0, stream, *a); \
        else if (shape == 1) hipLaunchKernelGGL((gemm_bf16_nt_kernel<E, 128, 64>), dim3((unsigned)ntiles(128, 64)), block, 0, stream, *a); \
        else hipLaunchKernelGGL((gemm_bf16_nt_kernel<E, 64, 64>), dim3((unsigned)ntiles(64, 64)), block, 0, stream, *a);  \
    } while (0)
    switch (epi) {
        case EPI_F32: LAUNCH_V1(EPI_F32); break;
        case EPI_BF16: LAUNCH_V1(EPI_BF16); break;
        case EPI_GELU_BF16: LAUNCH_V1(EPI_GELU_BF16); break;
        case EPI_RELU_BF16: LAUNCH_V1(EPI_RELU_BF16); break;
        case EPI_RESID_F32: LAUNCH_V1(EPI_RESID_F32); break;
        default: ldt_set_error("gemm: unknown epilogue %d", epi); return LDT_EARG;
    }
#undef LAUNCH_V1
    return ldt_check_launch("gemm_bf16_nt");
}
