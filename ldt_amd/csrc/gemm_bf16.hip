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

#include "kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * BK * 2)      // 16 KiB per operand per stage

__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, long ld, int row0, int nrows_total,
                                           int k0, char* lds_tile, int wave, int lane) {
    // one operand tile = 128 rows x 128 B = 16 pieces of 1 KiB (8 rows each); 4 pieces per wave
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int piece = wave * 4 + p;
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

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];   // [stage][X|W]
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
    const int tile_m = wgid / tiles_n, tile_n = wgid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / BK;
    stage_tile(a.X, a.ldx, m0, a.M, 0, smem, wave, lane);
    stage_tile(a.W, a.ldw, n0, a.N, 0, smem + TILE_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int lrow = lane & 15;
    const int lchk = lane >> 4;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        char* sx = smem + cur * 2 * TILE_BYTES;
        char* sw = sx + TILE_BYTES;
        if (kt + 1 < nk) {
            char* nx = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile(a.X, a.ldx, m0, a.M, (kt + 1) * BK, nx, wave, lane);
            stage_tile(a.W, a.ldw, n0, a.N, (kt + 1) * BK, nx + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[4], xf[4];
            const int c = ks * 4 + lchk;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + lrow;
                wf[i] = *reinterpret_cast<const bf16x8*>(sw + rw * 128 + ((c ^ ((rw >> 1) & 7)) << 4));
                const int rx = wm * 64 + i * 16 + lrow;
                xf[i] = *reinterpret_cast<const bf16x8*>(sx + rx * 128 + ((c ^ ((rx >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds D[n = nb + (lane>>4)*4 + r][m = mb + (lane&15)], r = 0..3 ----
    const float* gate = a.gate;
    if (EPI == EPI_RESID_F32 && gate && a.step_ptr) gate += (long)(*a.step_ptr) * a.gate_step_stride;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + lrow;
        if (m >= a.M) continue;
        const float* grow = nullptr;
        if (EPI == EPI_RESID_F32 && gate) grow = gate + (long)(m / a.rows_per_sample) * a.gate_sample_stride;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + lchk * 4;
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
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);
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
// v2: 256x256 tile, 8 waves in two groups that ping-pong on each SIMD (CDNA4 guide "8-phase" structure).
//
//   * 512 threads = 8 waves; wave w and w+4 share a SIMD.  Group g = w>>2 owns output rows [g*128, +128) of the
//     tile, wn = w&3 owns 64 output columns: per wave 128x64 = 8x4 accumulator tiles of mfma_f32_16x16x32_bf16
//     (operands swapped as in v1: D[n][m]).
//   * K is consumed in 32-deep sub-tiles through a 4-slot LDS ring (slot = X[256][32] + W[256][32] bf16 = 32 KiB;
//     128 KiB total, one workgroup per CU).  Rows are 64 B; the 16-B chunk index is XORed with f((row>>2)&3),
//     f = {0,2,3,1}, which makes every ds_read_b128 lane group of the 16x16x32 operand read conflict-free
//     (applied on the glds SOURCE address and on the read address).
//   * Each sub-tile is two phases of 16 MFMAs.  A phase = [load segment: ds_reads of this phase's operands +
//     one glds batch (2 x 1 KiB per wave) for a future sub-tile] s_barrier [16 MFMAs] s_barrier.  Group 1 runs
//     one barrier behind group 0, so on every SIMD one wave issues MFMAs while its partner reads LDS / issues DMA.
//   * glds for sub-tile v: X at phase 1 of sub-tile v-3, W at phase 0 of sub-tile v-2; the only VMEM wait in the
//     loop is a counted `s_waitcnt vmcnt(6)` once per sub-tile (three batches stay in flight across barriers).
//     Slot reuse distance >= 2 phases after the last read (WAR), data is read >= 1 barrier after every wave's
//     counted wait (RAW).  Past the end of K the batches are still issued (clamped to the last sub-tile, never
//     consumed) so the wait count stays uniform.
#ifndef V2_STAGED_EPILOGUE
#define V2_STAGED_EPILOGUE 1
#endif
#ifndef V2_ORDER
#define V2_ORDER 0
#endif
#ifndef V2_ABLATE
#define V2_ABLATE 0          // timing-only ablations for tools/dbg (1: no glds in the loop, 2: no ds_reads, 4: no barriers)
#endif
#define V2_STAGE_BYTES 32768
#define V2_OPER_BYTES 16384

__device__ __forceinline__ int v2_swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

__device__ __forceinline__ void v2_stage(const bf16_t* __restrict__ g, long ld, int row0, int nrows_total, int k0,
                                         char* lds_oper, int wave, int lane) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int piece = wave * 2 + p;                       // 16 pieces of 16 rows x 64 B
#if V2_ABLATE & 8
        const int r = piece * 8 + (lane >> 3);                // timing-only: full 128-B lines (wrong data)
        const int csrc = (lane & 7);
#else
        const int r = piece * 16 + (lane >> 2);
        const int csrc = (lane & 3) ^ v2_swz(r);
#endif
        int grow = row0 + r;
        grow = grow < nrows_total ? grow : nrows_total - 1;
        const bf16_t* src = g + (long)grow * ld + k0 + csrc * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_oper + piece * 1024), 16, 0, 0);
    }
}

#if V2_ABLATE & 4
#define V2_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define V2_BARRIER()                          \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
#endif

template <int EPI, bool INTERIOR>
__device__ __forceinline__ void v2_epilogue(const GemmArgs& a, f32x4 (&acc)[4][8], int m0, int n0, int grp, int wn,
                                            int lrow, int lchk, const float* gate) {
    // lane holds D[n = nb + lchk*4 + r][m = mb + lrow]
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = m0 + grp * 128 + mi * 16 + lrow;
        if (!INTERIOR && m >= a.M) continue;
        const float* grow = nullptr;
        if (EPI == EPI_RESID_F32 && gate) grow = gate + (long)(m / a.rows_per_sample) * a.gate_sample_stride;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + lchk * 4;
            if (!INTERIOR && n >= a.N) continue;
            f32x4 v = acc[ni][mi];
            const bool full = INTERIOR || (n + 3 < a.N);
            if (a.bias) {
                if (full) { const f32x4 t = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += t[r]; }
                else for (int r = 0; r < 4; ++r) if (n + r < a.N) v[r] += a.bias[n + r];
            }
            if (EPI == 5) {                     // timing-only: keep the accumulators live, store nothing
                if (a.M < 0) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (long)m * a.ldo + n) = v;
            } else if (EPI == EPI_F32) {
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
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);
                }
                if (EPI == EPI_RELU_BF16) {
                    if (a.skip) {
                        const bf16_t* sk = a.skip + (long)m * a.lds_ + n;
                        for (int r = 0; r < 4; ++r) if (full || n + r < a.N) v[r] += (float)sk[r];
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

// Interior-tile epilogue staged through the (now idle) LDS ring so that every global access is 16 B per lane
// over whole output rows: the MFMA fragment layout gives each lane 4 consecutive columns of 16 DIFFERENT rows,
// i.e. 32-B (bf16) / 64-B (fp32) row pieces per store instruction, which is store-ISSUE bound (guide T21).
// Each wave owns a private LDS region (no workgroup barrier): it writes its accumulator fragments row-major
// (row stride padded by 16 B against bank conflicts), reads rows back as 16-B chunks and streams them out:
//   bf16 out : 2 passes of 64 rows x 128 B  -> one store instruction = 8 rows x 128 B
//   fp32 out : 4 passes of 32 rows x 256 B  -> one access            = 4 rows x 256 B  (+ residual read, gate)
template <int EPI>
__device__ __forceinline__ void v2_epilogue_staged(const GemmArgs& a, f32x4 (&acc)[4][8], int m0, int n0, int grp, int wn,
                                                   int wave, int lane, int lrow, int lchk, const float* gate, char* smem) {
    const int mb = m0 + grp * 128, nb = n0 + wn * 64;
    f32x4 bias4[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
        bias4[ni] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nb + ni * 16 + lchk * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16) {
        constexpr int RS = 128 + 16;                              // bytes per staged row
        char* reg = smem + wave * (64 * RS);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = mi * 16 + lrow;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    f32x4 v = acc[ni][half * 4 + mi];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bias4[ni][r];
                    if (EPI == EPI_GELU_BF16) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);
                    }
                    if (EPI == EPI_RELU_BF16) {
                        if (a.skip) {
                            const bf16x4 sk = *reinterpret_cast<const bf16x4*>(a.skip + (long)(mb + half * 64 + row) * a.lds_ + nb + ni * 16 + lchk * 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (float)sk[r];
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    *reinterpret_cast<bf16x4*>(reg + row * RS + (ni * 16 + lchk * 4) * 2) = pk;
                }
            }
            bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (long)(mb + half * 64) * a.ldo + nb;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const bf16x8 d = *reinterpret_cast<const bf16x8*>(reg + row * RS + ch * 16);
                *reinterpret_cast<bf16x8*>(o + (long)row * a.ldo + ch * 8) = d;
            }
        }
    } else {
        constexpr int RS = 256 + 16;
        char* reg = smem + wave * (32 * RS);
        const int ch = lane & 15;
        f32x4 g4 = {1.f, 1.f, 1.f, 1.f};
        const bool has_gate = (EPI == EPI_RESID_F32) && gate;
        const bool shared_gate = has_gate && a.gate_sample_stride == 0;
        if (shared_gate) g4 = *reinterpret_cast<const f32x4*>(gate + nb + ch * 4);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = mi * 16 + lrow;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    f32x4 v = acc[ni][qd * 2 + mi];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bias4[ni][r];
                    *reinterpret_cast<f32x4*>(reg + row * RS + (ni * 16 + lchk * 4) * 4) = v;
                }
            }
            const long mrow0 = mb + qd * 32;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4);
                f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * RS + ch * 16);
                float* o = reinterpret_cast<float*>(a.out) + (mrow0 + row) * a.ldo + nb + ch * 4;
                if (EPI == EPI_RESID_F32) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(a.resid + (mrow0 + row) * a.ldr + nb + ch * 4);
                    if (has_gate && !shared_gate)
                        g4 = *reinterpret_cast<const f32x4*>(gate + ((mrow0 + row) / a.rows_per_sample) * a.gate_sample_stride + nb + ch * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = x[r] + g4[r] * v[r];
                }
                if (EPI != 5 || a.M < 0) *reinterpret_cast<f32x4*>(o) = v;
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_nt_256_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;

    const int tiles_n = (a.N + 255) / 256;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int m0 = (wgid / tiles_n) * 256, n0 = (wgid % tiles_n) * 256;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nks = a.K >> 5;
    const int lrow = lane & 15, lchk = lane >> 4;

#define ISSUE_X(v_) do { const int vv_ = (v_) < nks ? (v_) : nks - 1; \
        v2_stage(a.X, a.ldx, m0, a.M, vv_ << 5, smem2 + ((v_) & 3) * V2_STAGE_BYTES, wave, lane); } while (0)
#define ISSUE_W(v_) do { const int vv_ = (v_) < nks ? (v_) : nks - 1; \
        v2_stage(a.W, a.ldw, n0, a.N, vv_ << 5, smem2 + ((v_) & 3) * V2_STAGE_BYTES + V2_OPER_BYTES, wave, lane); } while (0)

    ISSUE_X(0); ISSUE_W(0); ISSUE_X(1); ISSUE_W(1); ISSUE_X(2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    V2_BARRIER();
    if (grp == 1) V2_BARRIER();                               // stagger the two groups by one barrier

    // per-lane LDS read offsets inside an operand sub-tile: row*64 + ((chunk ^ f(row)) << 4)
    int xoff[8], woff[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int r = grp * 128 + i * 16 + lrow; xoff[i] = r * 64 + ((lchk ^ v2_swz(r)) << 4); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = wn * 64 + i * 16 + lrow; woff[i] = V2_OPER_BYTES + r * 64 + ((lchk ^ v2_swz(r)) << 4); }

#if V2_ABLATE & 2
    bf16x8 wf[4], xf[4];
    for (int i = 0; i < 4; ++i) { wf[i] = *reinterpret_cast<const bf16x8*>(smem2 + woff[i]); xf[i] = *reinterpret_cast<const bf16x8*>(smem2 + xoff[i]); }
#endif
    for (int v = 0; v < nks; ++v) {
        const char* st = smem2 + (v & 3) * V2_STAGE_BYTES;
#if !(V2_ABLATE & 2)
        bf16x8 wf[4], xf[4];
#endif
        // ---------------- phase 0: W(all 4 n-tiles) + X(m-tiles 0..3) ----------------
#if V2_ORDER == 1 && !(V2_ABLATE & 1)
        ISSUE_W(v + 2);
        __builtin_amdgcn_sched_barrier(0);
#endif
#if !(V2_ABLATE & 2)
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + woff[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xoff[i]);
#else
        asm volatile("" : "+v"(wf[0]), "+v"(xf[0]) :: "memory"); (void)st;
#endif
#if V2_ORDER == 0 && !(V2_ABLATE & 1)
        ISSUE_W(v + 2);
#endif
        V2_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        V2_BARRIER();
        // ---------------- phase 1: X(m-tiles 4..7) ----------------
#if V2_ORDER == 1 && !(V2_ABLATE & 1)
        ISSUE_X(v + 3);
        __builtin_amdgcn_sched_barrier(0);
#endif
#if !(V2_ABLATE & 2)
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(st + xoff[4 + i]);
#endif
#if !(V2_ABLATE & 1)
#if V2_ORDER == 0
        ISSUE_X(v + 3);
#endif
#if !(V2_ABLATE & 16)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // sub-tile v+1 has landed (3 newer batches in flight)
#endif
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
    }
    if (grp == 0) V2_BARRIER();                               // every wave executes the same number of barriers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the (unused) tail batches ...
    V2_BARRIER();                                             // ... of EVERY wave: the ring is quiescent, LDS is reusable
#undef ISSUE_X
#undef ISSUE_W

    const float* gate = a.gate;
    if (EPI == EPI_RESID_F32 && gate && a.step_ptr) gate += (long)(*a.step_ptr) * a.gate_step_stride;
    const bool aligned = (a.ldo % 8 == 0) && (EPI != EPI_RESID_F32 || a.ldr % 4 == 0) && (EPI != EPI_RELU_BF16 || !a.skip || a.lds_ % 4 == 0);
    if ((m0 + 256 <= a.M) && (n0 + 256 <= a.N) && aligned && V2_STAGED_EPILOGUE)
        v2_epilogue_staged<EPI>(a, acc, m0, n0, grp, wn, wave, lane, lrow, lchk, gate, smem2);
    else if ((m0 + 256 <= a.M) && (n0 + 256 <= a.N)) v2_epilogue<EPI, true>(a, acc, m0, n0, grp, wn, lrow, lchk, gate);
    else v2_epilogue<EPI, false>(a, acc, m0, n0, grp, wn, lrow, lchk, gate);
}

template <int EPI>
static int launch_256(const GemmArgs* a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_nt_256_kernel<EPI>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 4 * V2_STAGE_BYTES);
        if (e != hipSuccess) { ldt_set_error("gemm256: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int tiles = ((a->M + 255) / 256) * ((a->N + 255) / 256);
    hipLaunchKernelGGL(gemm_bf16_nt_256_kernel<EPI>, dim3(tiles), dim3(512), 4 * V2_STAGE_BYTES, stream, *a);
    return ldt_check_launch("gemm_bf16_nt_256");
}

// LDT_GEMM_FORCE=128|256 pins the variant (A/B runs); default: 256^2 when it fills at least half the CUs.
static int gemm_variant() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LDT_GEMM_FORCE"); v = e ? atoi(e) : 0; }
    return v;
}

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
    const int tiles256 = ((a->M + 255) / 256) * ((a->N + 255) / 256);
    const int force = gemm_variant();
    if ((force == 256 || (force == 0 && tiles256 >= 128)) && a->M >= 16 && a->N >= 16) {
        switch (epi) {
            case EPI_F32: return launch_256<EPI_F32>(a, stream);
            case EPI_BF16: return launch_256<EPI_BF16>(a, stream);
            case EPI_GELU_BF16: return launch_256<EPI_GELU_BF16>(a, stream);
            case EPI_RELU_BF16: return launch_256<EPI_RELU_BF16>(a, stream);
            case EPI_RESID_F32: return launch_256<EPI_RESID_F32>(a, stream);
            case 5: return launch_256<5>(a, stream);       // timing-only (tools/dbg): no stores
            default: ldt_set_error("gemm: unknown epilogue %d", epi); return LDT_EARG;
        }
    }
    const int tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
    dim3 grid(tiles), block(256);
    switch (epi) {
        case EPI_F32: hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI_F32>, grid, block, 0, stream, *a); break;
        case EPI_BF16: hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI_BF16>, grid, block, 0, stream, *a); break;
        case EPI_GELU_BF16: hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI_GELU_BF16>, grid, block, 0, stream, *a); break;
        case EPI_RELU_BF16: hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI_RELU_BF16>, grid, block, 0, stream, *a); break;
        case EPI_RESID_F32: hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI_RESID_F32>, grid, block, 0, stream, *a); break;
        default: ldt_set_error("gemm: unknown epilogue %d", epi); return LDT_EARG;
    }
    return ldt_check_launch("gemm_bf16_nt");
}
