// Mid-size bf16 MFMA GEMM (gfx950): 128 x (256 | 128) output tiles, one tile per workgroup, for the batches whose GEMMs have
// too few 256^2 tiles to fill the chip (M = 1-4 k token rows: the reference's shipped 32-token regime,
// experiments/Latent_Diffusion_Trainer/airplane/config.yaml:48-49, and BASELINE configs[4]'s per-GPU share).
//
//   Y[M,N] = epilogue( X[M,K] (bf16, row-major) . W[N,K]^T (bf16, row-major) + bias[N] )     (the 1x1 Conv1d / Linear layers of
//   model/layers.py:121-124,159-161; same operand conventions and epilogues as gemm_bf16.hip)
//
// Why another kernel (round 3 measurements, profiles/r03_t32_kernel_sequence.txt, r03_smallm_tile_sweep.txt): at M = 2048 every GEMM of
// a Score block ran the 2-phase 128^2 / 64^2 kernels at 0.3-0.6 PFLOP/s.  Those kernels keep ONE K-stage in flight per workgroup, and an
// LDS-DMA request that misses to HBM (the weights: 914 MB per SDE step stream through, never cache-resident) takes ~1.1-1.3 us to land,
// so a K = 1024 tile is 16 dependent round trips: latency-bound at ~40 GB/s per CU where the L2 -> LDS path delivers ~80.  A tile with
// few bytes per flop AND enough bytes in flight needs the whole LDS of a CU and a request stream nobody has to wait for:
//
//   * 512 threads = 4 COMPUTE waves + 4 LOADER waves (one of each per SIMD).  Loader waves do nothing but issue the operand stream
//     (global_load_lds, 16 B per lane, pieces of 8 rows x 128 B = whole lines: gemm_bf16.hip v3's finding) into a ring of THREE
//     64-deep K-tile stages and keep TWO K-tiles (96 KB at BN = 256) in flight; compute waves never issue a VMEM instruction in the
//     main loop, so no MFMA stream ever stalls behind a DMA issue (60-185 cycles each) and the epilogue's loads and
//     stores share a queue with nothing.
//   * a compute wave owns 128 rows x BN/4 columns (BN = 256: 8 x 4 accumulator tiles of mfma_f32_16x16x32_bf16, operands swapped as in
//     gemm_bf16.hip so a lane holds 4 consecutive output columns of one row); its fragment reads are software-pipelined one 16-MFMA
//     block ahead in registers (two X sets, two W sets), so the single compute wave of a SIMD keeps the matrix pipe busy without a
//     partner wave.
//   * ONE s_barrier per K-tile (all 8 waves).  Loaders arrive after a counted `s_waitcnt vmcnt` says K-tile kt+1 has landed; compute
//     waves arrive after `lgkmcnt(0)` behind their last fragment read of K-tile kt.  Past the barrier the compute waves read K-tile
//     kt+1 (RAW: covering vmcnt + barrier) and the loaders refill the stage K-tile kt has just left with K-tile kt+3 (WAR: every
//     compute wave's reads of it have returned).  Loader waves exit after the last K-tile; the epilogue is per wave (no barrier).
//   * LDS: 3 x (X[128][64] | W[BN][64]) bf16 with 128-B rows, 16-B chunk index XORed with (row >> 1) & 7 on the DMA source address and
//     on the ds_read address (conflict-free ds_read_b128) + 4 KiB of staging per compute wave: 160 KiB at BN = 256, 112 KiB at BN = 128.
//   * epilogues through the per-wave staging area so that every global access is 16 B per lane over whole rows:
//     EPI_F32 (also split-K partials: blockIdx.y = K slice), EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32 (gate per step and/or per sample).
//   * tile order: XCD-aware bijective remap; column-major inside an XCD's chunk (an XCD reads its own slice of W — cold in HBM every
//     step — exactly once, and shares the small X panel set through L2 / Infinity Cache).
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"

#define MID_BM 128
#define MID_BK 64
#define MID_NS 3
#define MID_XB (MID_BM * MID_BK * 2)                     /* 16 KiB: X part of a stage */

template <int BN>
struct MidCfg {
    static constexpr int NT = BN / 64;                   // 16-column accumulator tiles per compute wave (wave = BN / 4 columns)
    static constexpr int STAGE = MID_XB + BN * MID_BK * 2;
    static constexpr int RING = MID_NS * STAGE;
    static constexpr int LDS = RING + 4 * 4096;
    static constexpr int XPW = MID_BM / 8 / 4;           // X pieces per loader wave and K-tile (4)
    static constexpr int WPW = BN / 8 / 4;               // W pieces per loader wave and K-tile (8 | 4)
    static constexpr int PPW = XPW + WPW;
};

#define MID_BARRIER()                         \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

// Instruction order of one 16-MFMA block and the R fragment reads issued for the NEXT block (one compute wave per SIMD: nothing else
// fills the matrix pipe while this wave issues ds_reads, ~16 cycles each, so they go BETWEEN the MFMAs): R x (1 read, MPR MFMAs), then
// the remaining MFMAs — the reads lead, so the block after this one does not open on an LDS round trip.
template <int R, int MPR>
__device__ __forceinline__ void mid_interleave() {
#pragma unroll
    for (int i = 0; i < R; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);     // MFMA
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16 - R * MPR, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// s_waitcnt lgkmcnt(0) as the BUILTIN (vmcnt / expcnt fields at their maxima): hipcc's wait-insertion pass sees it, an asm statement it would not
#define MID_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

template <int N>
__device__ __forceinline__ void mid_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// tools/dbg build (-DMID_STAMPS): wall-clock stamps (s_memrealtime, 10 ns) of wave 0 (compute) and wave 4 (loader) of every workgroup, kept in a
// VGPR (lane i = stamp i) and stored once at the wave's end -> g_mid_stamps[workgroup][role][64]; tools/dbg/mid_stamps.py reads them
#ifdef MID_STAMPS
__device__ long long* g_mid_stamps;
extern "C" int ldt_dbg_mid_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mid_stamps), &p, sizeof(p)); }
#define MID_STAMP_DECL() long long _stv = 0
#define MID_STAMP(idx) do { const long long _t = __builtin_amdgcn_s_memrealtime(); _stv = (lane == (idx)) ? _t : _stv; } while (0)
#define MID_STAMP_FLUSH(role) do { if (g_mid_stamps) g_mid_stamps[((long)(blockIdx.x + blockIdx.y * gridDim.x) * 2 + (role)) * 64 + lane] = _stv; } while (0)
#else
#define MID_STAMP_DECL()
#define MID_STAMP(idx)
#define MID_STAMP_FLUSH(role)
#endif

// ---------------------------------------------------------------------------------------------- loader waves
template <int BN>
__device__ __forceinline__ void mid_loader(const GemmArgs& a, char* smem, int lw, int lane, int m0, int n0, int kbase, int nkt) {
    using C = MidCfg<BN>;
    // piece = 8 rows x 128 B; lane -> row (lane >> 3), LDS position lane & 7 holds global chunk (lane & 7) ^ ((row >> 1) & 7)
    const char* xp[C::XPW];
    const char* wp[C::WPW];
#pragma unroll
    for (int q = 0; q < C::XPW; ++q) {
        const int r = (lw * C::XPW + q) * 8 + (lane >> 3);
        int grow = m0 + r;
        grow = grow < a.M ? grow : a.M - 1;              // clamp: rows past the edge are never stored
        xp[q] = reinterpret_cast<const char*>(a.X + (long)grow * a.ldx + kbase) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int q = 0; q < C::WPW; ++q) {
        const int r = (lw * C::WPW + q) * 8 + (lane >> 3);
        int grow = n0 + r;
        grow = grow < a.N ? grow : a.N - 1;
        wp[q] = reinterpret_cast<const char*>(a.W + (long)grow * a.ldw + kbase) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto issue = [&](int slot) {                         // the K-tile the pointers stand at -> stage `slot`; then advance one K-tile
        char* st = smem + slot * C::STAGE;
#pragma unroll
        for (int q = 0; q < C::XPW; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xp[q],
                                             (__attribute__((address_space(3))) void*)(st + (lw * C::XPW + q) * 1024), 16, 0, 0);
            xp[q] += MID_BK * 2;
        }
#pragma unroll
        for (int q = 0; q < C::WPW; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wp[q],
                                             (__attribute__((address_space(3))) void*)(st + MID_XB + (lw * C::WPW + q) * 1024), 16, 0, 0);
            wp[q] += MID_BK * 2;
        }
    };
    MID_STAMP_DECL();
    MID_STAMP(0);
    issue(0);
    if (nkt > 1) issue(1);
    MID_STAMP(1);
    // K-tile 0 landed (K-tile 1 may still be in flight); K-tile 2 follows the barrier: the compute waves start one K-tile's issue time earlier
    if (nkt > 1) mid_wait_vmcnt<C::PPW>();
    else mid_wait_vmcnt<0>();
    MID_STAMP(2);
    MID_BARRIER();                                       // prologue barrier
    MID_STAMP(3);
    if (nkt > 2) issue(2);
    int slot = 0;
    for (int kt = 0; kt + 1 < nkt; ++kt) {
        MID_STAMP(4 + 3 * kt);
        // K-tile kt+1 landed (only K-tile kt+2, if it exists, was issued after it; kt+3 is issued past the barrier)
        if (kt + 2 < nkt) mid_wait_vmcnt<C::PPW>();
        else mid_wait_vmcnt<0>();
        MID_STAMP(5 + 3 * kt);
        MID_BARRIER();                                   // barrier kt: compute waves are past their last read of K-tile kt
        MID_STAMP(6 + 3 * kt);
        if (kt + 3 < nkt) issue(slot);                   // K-tile kt+3 -> the stage K-tile kt has left
        slot = slot + 1 == MID_NS ? 0 : slot + 1;
    }
#ifdef MID_STAMPS
    if (lw == 0) MID_STAMP_FLUSH(1);
#endif
}

// ---------------------------------------------------------------------------------------------- compute-wave epilogue
// lane holds D[n = nb + ni*16 + lchk*4 + r][m = mb + mi*16 + lrow], r = 0..3  (operands swapped: 4 consecutive output columns of one row)
// rows / 16-B chunk of the fp32 staged-row pass: lane -> (row = it * RPI + lane / CPR, chunk = lane % CPR), CPR = NT * 4 chunks per row
template <int NT>
__device__ __forceinline__ f32x4 mid_resid_load(const GemmArgs& a, int m0, int nb, int lane, int mi, int it) {
    constexpr int CPR = NT * 4, RPI = 64 / CPR;
    long row = (long)m0 + mi * 16 + it * RPI + lane / CPR;
    row = row < a.M ? row : a.M - 1;                     // clamp, never branch (a branch per load makes hipcc wait for each one): rows past the edge are not stored
    return *reinterpret_cast<const f32x4*>(a.resid + row * a.ldr + nb + (lane % CPR) * 4);
}

// `rpre` (EPI_RESID_F32, BN = 128): the wave's whole residual tile, requested before the main loop (16 x 16 B per lane)
template <int EPI, int BN>
__device__ __forceinline__ void mid_epilogue(const GemmArgs& a, f32x4 (&acc)[MidCfg<BN>::NT][8], int m0, int nb, int lane, char* reg,
                                             const f32x4 (&rpre)[8][2]) {
    constexpr int NT = MidCfg<BN>::NT;
    const int lrow = lane & 15, lchk = lane >> 4;
    f32x4 bias4[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
        bias4[ni] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nb + ni * 16 + lchk * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU_BF16) {
        constexpr int RS = NT * 32 + 16;                 // staged row: NT*16 bf16 + 16 B pad
        constexpr int CPR = NT * 2;                      // 16-B chunks per row (8 | 4)
        constexpr int RPI = 64 / CPR;                    // rows per store instruction (8 | 16)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                f32x4 v = acc[ni][mi];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bias4[ni][r];
                if constexpr (EPI == EPI_GELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 gg = gelu_erf_fast2((f32x2){v[r], v[r + 1]});
                        v[r] = gg[0]; v[r + 1] = gg[1];
                    }
                }
                const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(reg + lrow * RS + (ni * 16 + lchk * 4) * 2) = pk;
            }
            bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (long)(m0 + mi * 16) * a.ldo + nb;
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it) {
                const int row = it * RPI + lane / CPR, ch = lane % CPR;
                const bf16x8 d = *reinterpret_cast<const bf16x8*>(reg + row * RS + ch * 16);
                if (m0 + mi * 16 + row < a.M) *reinterpret_cast<bf16x8*>(o + (long)row * a.ldo + ch * 8) = d;
            }
        }
    } else {
        constexpr int CPR = NT * 4;                      // 16-B chunks per fp32 row (16 | 8)
        constexpr int RPI = 64 / CPR;                    // rows per instruction (4 | 8)
        constexpr int RSB = CPR * 16;                    // staged row bytes (256 | 128), chunk index XORed with the row
        const int ch = lane % CPR;
        const float* gate = a.gate;
        if (EPI == EPI_RESID_F32 && gate && a.step_ptr) gate += (long)(*a.step_ptr) * a.gate_step_stride;
        const bool has_gate = (EPI == EPI_RESID_F32) && gate;
        const bool shared_gate = has_gate && a.gate_sample_stride == 0;
        f32x4 g4 = {1.f, 1.f, 1.f, 1.f};
        if (shared_gate) g4 = *reinterpret_cast<const f32x4*>(gate + nb + ch * 4);
        float* obase = reinterpret_cast<float*>(a.out) + ((EPI == EPI_F32 && a.splits > 1) ? (long)blockIdx.y * a.split_stride : 0L);
        // BN = 256: the residual rows are requested TWO passes ahead (a pass = 16 rows = 4 x 16 B per lane; the fragment registers of the
        // main loop are free by now) — loaded pass by pass, each pass would expose a whole memory round trip (8 x ~1.3 us from HBM)
        constexpr int AHEAD = 2;
        f32x4 rq[(EPI == EPI_RESID_F32 && NT == 4) ? AHEAD : 1][4];
        if constexpr (EPI == EPI_RESID_F32 && NT == 4) {
#pragma unroll
            for (int p = 0; p < AHEAD; ++p)
#pragma unroll
                for (int it = 0; it < 4; ++it) rq[p][it] = mid_resid_load<NT>(a, m0, nb, lane, p, it);
        }
        // (the per-sample-gate form is a separate copy of the pass loop: a runtime test around a load inside it would make hipcc wait for
        //  every outstanding residual request at each pass)
        auto passes = [&](auto per_sample_c) {
            constexpr bool PER_SAMPLE = decltype(per_sample_c)::value;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    f32x4 v = acc[ni][mi];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bias4[ni][r];
                    *reinterpret_cast<f32x4*>(reg + lrow * RSB + ((((ni * 4 + lchk) ^ lrow) & (CPR - 1)) << 4)) = v;
                }
                const long mrow0 = m0 + mi * 16;
#pragma unroll
                for (int it = 0; it < 16 / RPI; ++it) {
                    const int row = it * RPI + lane / CPR;
                    f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * RSB + (((ch ^ row) & (CPR - 1)) << 4));
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (EPI == EPI_RESID_F32) {
                        if constexpr (NT == 4) {
                            x = rq[mi % AHEAD][it];
                            if (mi + AHEAD < 8) rq[mi % AHEAD][it] = mid_resid_load<NT>(a, m0, nb, lane, mi + AHEAD, it);
                        } else x = rpre[mi][it];
                    }
                    if (mrow0 + row >= a.M) continue;
                    if constexpr (EPI == EPI_RESID_F32) {
                        if constexpr (PER_SAMPLE)
                            g4 = *reinterpret_cast<const f32x4*>(gate + ((mrow0 + row) / a.rows_per_sample) * a.gate_sample_stride + nb + ch * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = x[r] + g4[r] * v[r];
                    }
                    *reinterpret_cast<f32x4*>(obase + (mrow0 + row) * a.ldo + nb + ch * 4) = v;
                }
            }
        };
        if (EPI == EPI_RESID_F32 && has_gate && !shared_gate) passes(std::true_type{});
        else passes(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------- kernel
template <int EPI, int BN>
__global__ __launch_bounds__(512) void gemm_bf16_nt_mid_kernel(const GemmArgs a) {
    using C = MidCfg<BN>;
    constexpr int NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) char smem_mid[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware bijective remap of the 1-D tile index (blocks b, b+8, ... share an XCD's L2), column-major inside the chunk
    const int tiles_m = (a.M + MID_BM - 1) / MID_BM, tiles_n = a.N / BN;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tile_m = a.col_major ? wgid % tiles_m : wgid / tiles_n, tile_n = a.col_major ? wgid / tiles_m : wgid % tiles_n;
    const int m0 = tile_m * MID_BM, n0 = tile_n * BN;
    const bool split = (EPI == EPI_F32 && a.splits > 1);
    const int ksplit = split ? a.K / a.splits : a.K;
    const int kbase = split ? (int)blockIdx.y * ksplit : 0;
    const int nkt = ksplit / MID_BK;

    if (wave >= 4) {                                     // loader waves: the operand stream, nothing else
        mid_loader<BN>(a, smem_mid, wave - 4, lane, m0, n0, kbase, nkt);
        return;
    }

    // ---- compute waves: wave wn owns columns [wn * BN/4, +BN/4) of the tile, all 128 rows
    const int wn = wave;
    const int lrow = lane & 15, lchk = lane >> 4;
    const int sw = (lrow >> 1) & 7;
    // per-lane LDS read bases inside a stage: row * 128 + ((k-half * 4 + lchk) ^ ((row >> 1) & 7)) * 16; fragment i at + i * 2048
    const int xb0 = lrow * 128 + ((lchk ^ sw) << 4), xb1 = lrow * 128 + (((4 + lchk) ^ sw) << 4);
    const int wrow = MID_XB + (wn * (BN / 4) + lrow) * 128;
    const int wb0 = wrow + ((lchk ^ sw) << 4), wb1 = wrow + (((4 + lchk) ^ sw) << 4);

    f32x4 acc[NT][8];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    MID_STAMP_DECL();
    MID_STAMP(0);
    // While the first K-tile is on its way the compute waves have nothing to do:
    //  (1) BN = 128 residual epilogue: the wave's whole residual tile (128 rows x 32 columns fp32 = 16 x 16 B per lane) is requested now and
    //      arrives under the main loop;
    //  (2) W-panel touch: the tiles_m workgroups that share this W column panel (cold in HBM at every SDE step) each pull their share of its
    //      128-B lines towards the XCD's L2 — K-tile j of the panel by the workgroup with tile_m == j mod tiles_m, one line per lane, one
    //      dword-sized LDS-DMA per wave and K-tile into the (still unused) staging area — so the loaders' requests from K-tile 3 on are L2 hits.
    f32x4 rpre[8][2];
    if constexpr (EPI == EPI_RESID_F32 && NT == 2) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int it = 0; it < 2; ++it) rpre[mi][it] = mid_resid_load<NT>(a, m0, n0 + wn * (BN / 4), lane, mi, it);
    }
#ifndef MID_NO_TOUCH
    {
        const int prow = wn * 64 + lane;                 // row of the panel this lane touches (BN = 128: waves 0, 1 cover it)
        if (prow < BN) {
            const char* wl = reinterpret_cast<const char*>(a.W + (long)(n0 + prow) * a.ldw + kbase);
            char* dst = smem_mid + C::RING + wave * 4096;
            int cnt = 0;
            for (int j = tile_m + (tile_m < 3 ? tiles_m : 0); j < nkt && cnt < 16; j += tiles_m, ++cnt)   // (K-tiles 0-2 are requested at once anyway)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wl + (long)j * (MID_BK * 2)),
                                                 (__attribute__((address_space(3))) void*)(dst + cnt * 256), 4, 0, 0);
        }
    }
#endif
    MID_BARRIER();                                       // prologue barrier: K-tile 0 has landed
    MID_STAMP(1);

    if constexpr (NT == 4) {
        // K-tile = 4 blocks of 16 MFMAs: (half 0, rows 0-63), (half 0, rows 64-127), (half 1, rows 0-63), (half 1, rows 64-127);
        // each block's fragments are read during the block before it
        bf16x8 wa[4], wb[4], xa[4], xb[4];
        auto ld_w = [&](bf16x8 (&w)[4], const char* st, int off) {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const bf16x8*>(st + off + i * 2048);
        };
        auto ld_x = [&](bf16x8 (&x)[4], const char* st, int off, int part) {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const bf16x8*>(st + off + (part * 4 + i) * 2048);
        };
        auto mm = [&](const bf16x8 (&w)[4], const bf16x8 (&x)[4], int part) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][part * 4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ni], x[mi], acc[ni][part * 4 + mi], 0, 0, 0);
        };
        const char* st = smem_mid;
        ld_w(wa, st, wb0); ld_x(xa, st, xb0, 0);
        MID_LGKM0();                                     // once: both ways into the loop then carry no pending reads (see the loop's tail)
        int slot = 0;
        // (the last K-tile is peeled: with the barrier under a condition inside ONE loop body hipcc merges the two paths in front of the
        //  fourth block and waits there for the next K-tile's first fragment reads — lgkmcnt(3..0) — before MFMAs that do not need them)
        __builtin_amdgcn_sched_barrier(0);
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            ld_x(xb, st, xb0, 1);
            mm(wa, xa, 0);
            mid_interleave<4, 2>();
            ld_w(wb, st, wb1); ld_x(xa, st, xb1, 0);
            mm(wa, xb, 1);
            mid_interleave<8, 1>();
            ld_x(xb, st, xb1, 1);                        // last read of this stage
            mm(wb, xa, 0);
            mid_interleave<4, 2>();
            MID_LGKM0();
            MID_BARRIER();                               // barrier kt: K-tile kt+1 landed; this stage may be refilled
            slot = slot + 1 == MID_NS ? 0 : slot + 1;
            st = smem_mid + slot * C::STAGE;
            ld_w(wa, st, wb0); ld_x(xa, st, xb0, 0);
            mm(wb, xb, 1);
            mid_interleave<8, 1>();
            MID_LGKM0();                                 // (free behind the MFMAs; lets hipcc's wait insertion open the next trip with known counters)
        }
        ld_x(xb, st, xb0, 1);
        mm(wa, xa, 0);
        mid_interleave<4, 2>();
        ld_w(wb, st, wb1); ld_x(xa, st, xb1, 0);
        mm(wa, xb, 1);
        mid_interleave<8, 1>();
        ld_x(xb, st, xb1, 1);
        mm(wb, xa, 0);
        mid_interleave<4, 2>();
        mm(wb, xb, 1);
    } else {
        // BN = 128: a wave owns 32 columns (2 accumulator tiles x 8 row tiles); K-tile = 2 blocks of 16 MFMAs (half 0, half 1)
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
        const char* st = smem_mid;
        ld(wa, xa, st, wb0, xb0);
        MID_LGKM0();
        int slot = 0;
        __builtin_amdgcn_sched_barrier(0);
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            ld(wb, xb, st, wb1, xb1);                    // last reads of this stage
            mm(wa, xa);
            mid_interleave<10, 1>();
            MID_LGKM0();
            MID_BARRIER();
            slot = slot + 1 == MID_NS ? 0 : slot + 1;
            st = smem_mid + slot * C::STAGE;
            ld(wa, xa, st, wb0, xb0);
            mm(wb, xb);
            mid_interleave<10, 1>();
            MID_LGKM0();
        }
        ld(wb, xb, st, wb1, xb1);
        mm(wa, xa);
        mid_interleave<10, 1>();
        mm(wb, xb);
    }

    MID_STAMP(2);
    mid_epilogue<EPI, BN>(a, acc, m0, n0 + wn * (BN / 4), lane, smem_mid + C::RING + wave * 4096, rpre);
#ifdef MID_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MID_STAMP(3);
    if (wave == 0) MID_STAMP_FLUSH(0);
#endif
}

// ---------------------------------------------------------------------------------------------- launcher
// Shapes this kernel takes (everything else stays with gemm_bf16.hip): whole column tiles, 16-byte aligned rows, K a multiple of 64.
static int mid_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LDT_GEMM_MID"); v = e ? atoi(e) : 1; }
    return v;
}

// tile width for (M, N): 256 when that still gives most CUs a tile, else 128; 0 = not a mid-size problem
int ldt_gemm_mid_bn(int epi, const GemmArgs* a) {
    if (!mid_env()) return 0;
    if (!(epi == EPI_F32 || epi == EPI_BF16 || epi == EPI_GELU_BF16 || epi == EPI_RESID_F32)) return 0;
    if (a->K % MID_BK != 0 || a->N % 128 != 0 || a->M < 128 || a->ldo % 8 != 0) return 0;
    if (a->splits > 1 && (epi != EPI_F32 || a->K % (a->splits * MID_BK) != 0)) return 0;
    if (epi == EPI_RESID_F32 && (a->ldr % 4 != 0 || (a->gate && a->gate_sample_stride % 4 != 0))) return 0;
    const int sp = a->splits > 1 ? a->splits : 1;
    const long tm = (a->M + MID_BM - 1) / MID_BM;
    const long t256 = (a->N % 256 == 0) ? tm * (a->N / 256) * sp : 0, t128 = tm * (a->N / 128) * sp;
    const int forced = mid_env();                        // LDT_GEMM_MID=256 / 128 pins the width (tools/dbg)
    if (forced == 256) return t256 ? 256 : 0;
    if (forced == 128) return 128;
    if (t256 * 8 >= LDT_NUM_CUS * 5) return 256;         // >= 160 workgroups of the wide tile
    if (t128 * 8 >= LDT_NUM_CUS * 3) return 128;         // >= 96 of the narrow one
    return 0;
}

template <int EPI, int BN>
static int mid_launch_t(const GemmArgs* a_in, hipStream_t stream) {
    using C = MidCfg<BN>;
    GemmArgs a = *a_in;
    const long tm = (a.M + MID_BM - 1) / MID_BM, tn = a.N / BN;
    static const int map_env = getenv("LDT_GEMM_MID_MAP") ? atoi(getenv("LDT_GEMM_MID_MAP")) : -1;   // tools/dbg: 0 row-major, 1 column-major
    a.col_major = map_env >= 0 ? map_env : 1;
    LDT_ENSURE_LDS((&gemm_bf16_nt_mid_kernel<EPI, BN>), C::LDS, "gemm_mid");
    const unsigned sp = (EPI == EPI_F32 && a.splits > 1) ? (unsigned)a.splits : 1u;
    hipLaunchKernelGGL((gemm_bf16_nt_mid_kernel<EPI, BN>), dim3((unsigned)(tm * tn), sp), dim3(512), C::LDS, stream, a);
    return ldt_check_launch("gemm_bf16_nt_mid");
}

int ldt_gemm_mid_launch(int epi, int bn, const GemmArgs* a, hipStream_t stream) {
#define MID_CASE(E)                                                                  \
    case E: return bn == 256 ? mid_launch_t<E, 256>(a, stream) : mid_launch_t<E, 128>(a, stream)
    switch (epi) {
        MID_CASE(EPI_F32);
        MID_CASE(EPI_BF16);
        MID_CASE(EPI_GELU_BF16);
        MID_CASE(EPI_RESID_F32);
        default: ldt_set_error("gemm_mid: epilogue %d not built", epi); return LDT_EARG;
    }
#undef MID_CASE
}
