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
//     EPI_F32, EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32 (gate per step and/or per sample), and the LN-folded producer / consumer forms.
//   * tile order: XCD-aware bijective remap; column-major inside an XCD's chunk (an XCD reads its own slice of W — cold in HBM every
//     step — exactly once, and shares the small X panel set through L2 / Infinity Cache).
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"

#define MID_BK 64

// BM x BN output tile: 128 x 256 (the widest a CU's LDS and the accumulator budget take), 128 x 192 (bf16-output epilogues only: QKV with
// N = 3 x 1024 over 2k rows is exactly 256 such tiles), 128 x 128, and 64 x 128 for the N = hidden residual GEMMs of a 2k-row batch
// (256 workgroups instead of 128; the smaller stage buys a 4-deep ring), 64 x 64 for the same GEMMs of a 1k-row batch (BASELINE configs[4]'s
// per-GPU share: again 256 workgroups; 6-deep ring)
template <int BM, int BN>
struct MidCfg {
    static constexpr int MT = BM / 16;                   // 16-row accumulator tiles per compute wave (a wave spans all BM rows)
    static constexpr int NT = BN / 64;                   // 16-column accumulator tiles per compute wave (a wave = BN / 4 columns: 4 | 3 | 2 | 1 tiles)
    static constexpr int XB = BM * MID_BK * 2;           // X part of a stage (bytes)
    static constexpr int STAGE = XB + BN * MID_BK * 2;
    static constexpr int NS = BM == 64 ? (BN == 64 ? 6 : 4) : 3;   // ring stages
    static constexpr int RING = NS * STAGE;
    static constexpr int LDS = RING + 4 * 4096;
    static constexpr int XPW = BM / 8 / 4;               // X pieces per loader wave and K-tile
    static constexpr int WPW = BN / 8 / 4;               // W pieces per loader wave and K-tile
    static constexpr int PPW = XPW + WPW;
};

// LN folding (gemm_bf16.hip "LN folding"; statistics granule = 32 columns = a compute wave's share of a 128-wide tile, plan->stats[D / 32][M][2]):
//   MID_FOLD_PRODUCER (EPI_RESID_F32, BN = 128): the epilogue also stores xs = bf16(x_new (1 + ln_scale)) and the wave's per-row (sum, sum of
//     squares) over its 32 columns -> stats_out[n / 32][M][2] (8-lane DPP row sums in a fixed order: bit-reproducible, no atomics);
//   MID_FOLD_CONSUMER (EPI_BF16 / EPI_GELU_BF16): X = xs.  The LOADER waves — idle between issue bursts — fetch the tile's rows' K / 32 partials
//     and its S | C slices behind the first ring fill, form (rstd, -mean rstd) per row and leave both in the tails of the staging areas; the
//     compute waves' epilogue is y = rstd acc + (-mean rstd S + C).
enum { MID_FOLD_NONE = 0, MID_FOLD_PRODUCER = 1, MID_FOLD_CONSUMER = 2 };
enum { MID_ATTN_NONE = 0, MID_ATTN_SELF = 1 /* QKV projection + self-attention, 128 x 192 */, MID_ATTN_CROSS = 2 /* q projection + cross-attention, 64 x 64 */ };
#define MID_XKV_BYTES 16384           /* cross-attention: [sample 2][K | V][32 keys][128 B] behind the staging areas */
#define MID_TAIL_OFF 2304            /* bf16 staging uses at most 16 rows x 144 B of each compute wave's 4 KiB; the tails hold: */
#define MID_RS_OFF (0 * 4096 + MID_TAIL_OFF)     /* (rstd, -mean rstd) of the tile's <= 128 rows (1 KiB)  */
#define MID_S_OFF (1 * 4096 + MID_TAIL_OFF)      /* fold_S slice, BN floats (<= 1 KiB)                    */
#define MID_C_OFF (2 * 4096 + MID_TAIL_OFF)      /* fold_C slice                                          */

#define MID_BARRIER()                         \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

// Instruction order of one 16-MFMA block and the R fragment reads issued for the NEXT block (one compute wave per SIMD: nothing else
// fills the matrix pipe while this wave issues ds_reads, ~16 cycles each, so they go BETWEEN the MFMAs): R x (1 read, MPR MFMAs), then
// the remaining MFMAs — the reads lead, so the block after this one does not open on an LDS round trip.
template <int R, int MPR, int TOTAL = 16>
__device__ __forceinline__ void mid_interleave() {
#pragma unroll
    for (int i = 0; i < R; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);     // MFMA
    }
    if constexpr (TOTAL - R * MPR > 0) __builtin_amdgcn_sched_group_barrier(0x008, TOTAL - R * MPR, 0);
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
#define MID_STAMP_FLUSH(role) do { if (g_mid_stamps) g_mid_stamps[((long)blockIdx.x * 2 + (role)) * 64 + lane] = _stv; } while (0)
#else
#define MID_STAMP_DECL()
#define MID_STAMP(idx)
#define MID_STAMP_FLUSH(role)
#endif

// ---------------------------------------------------------------------------------------------- loader waves
// ATTN (fused QKV + attention, BN = 192): tile column c of head h = tile_n is output column / W row (c / 64) * hidden + h * 64 + c % 64
__device__ __forceinline__ int mid_attn_col(int c, int h, int hidden) { return (c >> 6) * hidden + h * 64 + (c & 63); }

template <int BM, int BN, int FOLD, int ATTN>
__device__ __forceinline__ void mid_loader(const GemmArgs& a, char* smem, int lw, int lane, int m0, int n0, int kbase, int nkt, int step) {
    using C = MidCfg<BM, BN>;
    // piece = 8 rows x 128 B; lane -> row (lane >> 3), LDS position lane & 7 holds global chunk (lane & 7) ^ ((row >> 1) & 7).
    // Addresses = a wave-uniform base (tile origin + K-tile, advanced by scalar adds) + per-lane 32-bit byte offsets that never change
    // (rows past the edge are clamped to the last row: never stored)
    int xo[C::XPW], wo[C::WPW];
#pragma unroll
    for (int q = 0; q < C::XPW; ++q) {
        const int r = (lw * C::XPW + q) * 8 + (lane >> 3);
        const int rr = m0 + r < a.M ? r : a.M - 1 - m0;
        xo[q] = rr * (int)a.ldx * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int q = 0; q < C::WPW; ++q) {
        const int r = (lw * C::WPW + q) * 8 + (lane >> 3);
        const int rr = ATTN == MID_ATTN_SELF ? mid_attn_col(r, n0 / BN, a.N / 3) : (n0 + r < a.N ? r : a.N - 1 - n0);
        wo[q] = rr * (int)a.ldw * 2 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    const char* xbase = reinterpret_cast<const char*>(a.X + (long)m0 * a.ldx + kbase);
    const char* wbase = reinterpret_cast<const char*>(a.W + (ATTN == MID_ATTN_SELF ? 0L : (long)n0 * a.ldw) + kbase);
    auto issue = [&](int slot) {                         // the K-tile the bases stand at -> stage `slot`; then advance one K-tile
        char* st = smem + slot * C::STAGE;
#pragma unroll
        for (int q = 0; q < C::XPW; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xbase + xo[q]),
                                             (__attribute__((address_space(3))) void*)(st + (lw * C::XPW + q) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < C::WPW; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + wo[q]),
                                             (__attribute__((address_space(3))) void*)(st + C::XB + (lw * C::WPW + q) * 1024), 16, 0, 0);
        xbase += MID_BK * 2; wbase += MID_BK * 2;
    };
    MID_STAMP_DECL();
    MID_STAMP(0);
    if constexpr (ATTN == MID_ATTN_CROSS) {
        // the K | V rows (32 condition tokens x 64 channels of head n0 / 64) of the tile's two samples: 16 pieces, 4 per loader wave, the OLDEST
        // requests of the wave — every counted wait below covers them, the prologue barrier publishes them
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int g = lw * 4 + q, sp = g >> 3, which = (g >> 2) & 1, row = (g & 3) * 8 + (lane >> 3);
            const bf16_t* src = (which ? a.attn_v : a.attn_k) + (long)(m0 / 32 + sp) * a.attn_kv_batch_stride + (long)row * a.attn_ldkv + n0 +
                                (((lane & 7) ^ ((row >> 1) & 7)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + C::LDS + g * 1024), 16, 0, 0);
        }
    }
    issue(0);
    if (nkt > 1) issue(1);
    MID_STAMP(1);
    // K-tile 0 landed (K-tile 1 may still be in flight); the rest of the ring follows the barrier: the compute waves start earlier.
    // (Only K-tile 0 ahead of the barrier measured 1.5 % SLOWER per SDE step at the shipped 32-token config: tools/dbg/lib_ab.py.)
    if (nkt > 1) mid_wait_vmcnt<C::PPW>();
    else mid_wait_vmcnt<0>();
    MID_STAMP(2);
    MID_BARRIER();                                       // prologue barrier
    MID_STAMP(3);
#pragma unroll
    for (int j = 2; j < C::NS; ++j)
        if (j < nkt) issue(j);
    int slot = 0;
    int kt0 = 0;
    if constexpr (FOLD == MID_FOLD_CONSUMER) {
        // (launcher: nkt >= NS + 2, so the whole ring is in flight and K-tile NS exists)  Behind the ring fill: this tile's row statistics
        // (BM rows x stats_parts <= 32 partials: thread t -> row t / TPR, a contiguous run of partials) and S | C slices (one LDS-DMA per wave).
        constexpr int TPR = 256 / BM, PPT = 32 / TPR, NSTAT = PPT + 1;
        const int t = lw * 64 + lane;
        int row = m0 + t / TPR;
        row = row < a.M ? row : a.M - 1;
        const int p0 = (t % TPR) * PPT;
        const float* sp = a.stats_in + (long)row * 2;
        static_assert(PPT == 16, "the statistics pass is written for 128-row tiles (16 partials per thread)");
        // The partials are loaded by asm statements hipcc does not count: beside LDS-DMA requests it drains the whole queue (vmcnt(0)) before
        // and after any register load it knows of (cdna_hip_programming.md §5 trap 4b), which would park this wave until the ring has landed.
        // Their completion is counted by hand below; the wait statement names every destination (§5.7 item 1, form ii).
        f32x2 part[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {                  // (clamped, never branched: partials past stats_parts are re-reads of the last one, masked below)
            const int p = p0 + i < a.stats_parts ? p0 + i : a.stats_parts - 1;
            const float* src = sp + (long)p * a.M * 2;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(part[i]) : "v"(src) : "memory");
        }
        {
            constexpr int Q = 2 * BN;                    // bytes of [S | C] per loader wave: waves 0, 1 -> S, waves 2, 3 -> C
            const long fst = (long)step * a.fold_step_stride;
            const int c0 = (lw & 1) * (Q / 4) + lane * 4;    // tile column of this lane's 16 B
            const float* src = (lw < 2 ? a.fold_S : a.fold_C) + fst + (ATTN == MID_ATTN_SELF ? mid_attn_col(c0 < BN ? c0 : 0, n0 / BN, a.N / 3) : n0 + c0);
            char* dst = smem + C::RING + (lw < 2 ? MID_S_OFF : MID_C_OFF) + (lw & 1) * Q;
            if (lane * 16 < Q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        // barrier 0: K-tile 1 landed — issued after it: K-tiles 2 .. NS-1 and the NSTAT requests above
        mid_wait_vmcnt<(C::NS - 2) * C::PPW + NSTAT>();
        MID_BARRIER();
        issue(0);                                        // K-tile NS -> stage 0
        slot = 1;
        // everything older than K-tile NS's pieces — the statistics and S | C — has landed
        asm volatile("s_waitcnt vmcnt(%16)"
                     : "+v"(part[0]), "+v"(part[1]), "+v"(part[2]), "+v"(part[3]), "+v"(part[4]), "+v"(part[5]), "+v"(part[6]), "+v"(part[7]),
                       "+v"(part[8]), "+v"(part[9]), "+v"(part[10]), "+v"(part[11]), "+v"(part[12]), "+v"(part[13]), "+v"(part[14]), "+v"(part[15])
                     : "n"(C::PPW) : "memory");
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            if (p0 + i < a.stats_parts) { s1 += part[i][0]; s2 += part[i][1]; }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) {              // partner threads are adjacent lanes: lower run + upper run, fixed order
            const float o1 = __shfl_xor(s1, o, 64), o2 = __shfl_xor(s2, o, 64);
            s1 = (t & o) ? o1 + s1 : s1 + o1;
            s2 = (t & o) ? o2 + s2 : s2 + o2;
        }
        if (t % TPR == 0) {
            const float invk = 1.0f / (float)a.K;
            const float mean = s1 * invk;
            const float var = fmaxf(s2 * invk - mean * mean, 0.f);
            const float r = rsqrtf(var + 1e-6f);
            *reinterpret_cast<f32x2*>(smem + C::RING + MID_RS_OFF + (t / TPR) * 8) = (f32x2){r, -mean * r};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // visible to the compute waves behind barrier 1 (they read it after the main loop)
        kt0 = 1;
    }
    for (int kt = kt0; kt + 1 < nkt; ++kt) {
        MID_STAMP(4 + 3 * kt);
        // K-tile kt+1 landed: the K-tiles issued after it are kt+2 .. min(kt + NS - 1, nkt - 1) (kt + NS goes out past the barrier)
        const int last = kt + C::NS - 1 < nkt - 1 ? kt + C::NS - 1 : nkt - 1;
        const int ahead = last - (kt + 1);               // 0 .. NS - 2
        if (C::NS >= 6 && ahead >= 4) mid_wait_vmcnt<(C::NS >= 6 ? 4 : 0) * C::PPW>();
        else if (C::NS >= 5 && ahead == 3) mid_wait_vmcnt<(C::NS >= 5 ? 3 : 0) * C::PPW>();
        else if (ahead >= 2) mid_wait_vmcnt<2 * C::PPW>();
        else if (ahead == 1) mid_wait_vmcnt<C::PPW>();
        else mid_wait_vmcnt<0>();
        MID_STAMP(5 + 3 * kt);
        MID_BARRIER();                                   // barrier kt: compute waves are past their last read of K-tile kt
        MID_STAMP(6 + 3 * kt);
        if (kt + C::NS < nkt) issue(slot);               // K-tile kt+NS -> the stage K-tile kt has left
        slot = slot + 1 == C::NS ? 0 : slot + 1;
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

// `rpre` (EPI_RESID_F32, BN <= 128): the wave's whole residual tile, requested before the main loop (NT x 16 B per lane and 16-row pass)
// sum over the 8 lanes of an aligned lane octet (all 8 end up with the total): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
__device__ __forceinline__ float mid_oct_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}

template <int EPI, int BM, int BN, int FOLD>
__device__ __forceinline__ void mid_epilogue(const GemmArgs& a, f32x4 (&acc)[MidCfg<BM, BN>::NT][MidCfg<BM, BN>::MT], int m0, int nb, int wn, int lane,
                                             char* reg, const char* stage_base, const f32x4 (&rpre)[MidCfg<BM, BN>::MT][MidCfg<BM, BN>::NT <= 2 ? MidCfg<BM, BN>::NT : 1], int step) {
    constexpr int NT = MidCfg<BM, BN>::NT, MT = MidCfg<BM, BN>::MT;
    const int lrow = lane & 15, lchk = lane >> 4;
    f32x4 bias4[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
        bias4[ni] = (a.bias && FOLD != MID_FOLD_CONSUMER) ? *reinterpret_cast<const f32x4*>(a.bias + nb + ni * 16 + lchk * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU_BF16) {
        constexpr int RS = NT * 32 + 16;                 // staged row: NT*16 bf16 + 16 B pad
        constexpr int CHK = NT * 2;                      // 16-B chunks per row (8 | 6 | 4)
        constexpr int CPR = NT <= 2 ? 4 : 8;             // lanes per row (a power of two; NT = 3: lanes 6, 7 of each octet idle, NT = 1: lanes 2, 3 of each quad)
        constexpr int RPI = 64 / CPR;                    // rows per store instruction (8 | 16)
        f32x4 s4[NT];
        if constexpr (FOLD == MID_FOLD_CONSUMER) {       // this wave's columns of the S | C slices the loader waves left in the staging tails
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int c = wn * (BN / 4) + ni * 16 + lchk * 4;
                s4[ni] = *reinterpret_cast<const f32x4*>(stage_base + MID_S_OFF + c * 4);
                bias4[ni] = *reinterpret_cast<const f32x4*>(stage_base + MID_C_OFF + c * 4);
            }
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            f32x2 rn = {1.f, 0.f};
            if constexpr (FOLD == MID_FOLD_CONSUMER) rn = *reinterpret_cast<const f32x2*>(stage_base + MID_RS_OFF + (mi * 16 + lrow) * 8);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                f32x4 v = acc[ni][mi];
                if constexpr (FOLD == MID_FOLD_CONSUMER) {   // y = rstd acc + (-mean rstd S + C): the bias is inside C
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rn[0] * v[r] + (rn[1] * s4[ni][r] + bias4[ni][r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bias4[ni][r];
                }
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
                if (CHK == CPR || ch < CHK) {
                    const bf16x8 d = *reinterpret_cast<const bf16x8*>(reg + row * RS + ch * 16);
                    if (m0 + mi * 16 + row < a.M) *reinterpret_cast<bf16x8*>(o + (long)row * a.ldo + ch * 8) = d;
                }
            }
        }
    } else {
        constexpr int CPR = NT * 4;                      // 16-B chunks per fp32 row (16 | 8)
        constexpr int RPI = 64 / CPR;                    // rows per instruction (4 | 8)
        constexpr int RSB = CPR * 16;                    // staged row bytes (256 | 128), chunk index XORed with the row
        const int ch = lane % CPR;
        const float* gate = a.gate;
        if (EPI == EPI_RESID_F32 && gate) gate += (long)step * a.gate_step_stride;
        const bool has_gate = (EPI == EPI_RESID_F32) && gate;
        const bool shared_gate = has_gate && a.gate_sample_stride == 0;
        f32x4 g4 = {1.f, 1.f, 1.f, 1.f};
        if (shared_gate) g4 = *reinterpret_cast<const f32x4*>(gate + nb + ch * 4);
        float* obase = reinterpret_cast<float*>(a.out);
        f32x4 sc4 = {1.f, 1.f, 1.f, 1.f};
        if constexpr (FOLD == MID_FOLD_PRODUCER) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(a.ln_scale + (long)step * a.ln_step_stride + nb + ch * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) sc4[r] = 1.0f + t[r];
        }
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
            for (int mi = 0; mi < MT; ++mi) {
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
                            if (mi + AHEAD < MT) rq[mi % AHEAD][it] = mid_resid_load<NT>(a, m0, nb, lane, mi + AHEAD, it);
                        } else x = rpre[mi][it];
                    }
                    const bool live = mrow0 + row < a.M;
                    if constexpr (EPI == EPI_RESID_F32) {
                        if constexpr (PER_SAMPLE) {
                            const long srow = live ? mrow0 + row : (long)a.M - 1;
                            g4 = *reinterpret_cast<const f32x4*>(gate + (srow / a.rows_per_sample) * a.gate_sample_stride + nb + ch * 4);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = x[r] + g4[r] * v[r];
                    }
                    if (live) *reinterpret_cast<f32x4*>(obase + (mrow0 + row) * a.ldo + nb + ch * 4) = v;
                    if constexpr (FOLD == MID_FOLD_PRODUCER) {   // (NT == 2: a row's 32 columns sit in one aligned lane octet)
                        const bf16x4 pk = {(bf16_t)(v[0] * sc4[0]), (bf16_t)(v[1] * sc4[1]), (bf16_t)(v[2] * sc4[2]), (bf16_t)(v[3] * sc4[3])};
                        if (live) *reinterpret_cast<bf16x4*>(a.xs + (mrow0 + row) * a.ldxs + nb + ch * 4) = pk;
                        const float s1 = mid_oct_sum((v[0] + v[1]) + (v[2] + v[3]));
                        const float s2 = mid_oct_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
                        if (live && ch == 0) *reinterpret_cast<f32x2*>(a.stats_out + ((long)(nb >> 5) * a.M + mrow0 + row) * 2) = (f32x2){s1, s2};
                    }
                }
            }
        };
        if (EPI == EPI_RESID_F32 && has_gate && !shared_gate) passes(std::true_type{});
        else passes(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------- fused self-attention epilogue
// QKV projection + attention in one launch for the shipped 32-token regime (model/layers.py:183-200 with N = M = 32, head dim 64).  The 128 x 192
// tile is [q | k | v] of ONE head for FOUR whole samples: the finished projections (bias or LN-folded form applied) go to LDS as bf16 rows — the
// operand ring is idle by then — and compute wave s runs sample s: S^T = K Q^T (8 MFMAs), the softmax over the 32 keys of a query (8 values
// in the lane + two cross-lane steps), O^T = V^T P^T (8 MFMAs; the key order inside the 32-deep contraction is the accumulator layout's, applied
// to both operands, so P never leaves its registers), O / l -> attn_o[B][H][32][64].  The q | k | v rows are never written to HBM and the
// attention launch of the block is gone (6.5 us + a 12.6 MB round trip per block at B = 64).
#define MID_T_STRIDE 400             /* bytes per staged row (192 bf16 + pad): conflict-free 16-B fragment reads */
template <int FOLD>
__device__ __forceinline__ void mid_epilogue_attn(const GemmArgs& a, f32x4 (&acc)[3][8], int m0, int head, int wn, int lane, char* tq,
                                                  const char* stage_base) {
    const int lrow = lane & 15, lchk = lane >> 4;
    const int hidden = a.N / 3;
    // ---- 1. finish the projection, bf16 rows into tq[128][192]
    f32x4 add4[3], s4[3];
#pragma unroll
    for (int ni = 0; ni < 3; ++ni) {
        const int c = wn * 48 + ni * 16 + lchk * 4;
        if constexpr (FOLD == MID_FOLD_CONSUMER) {
            s4[ni] = *reinterpret_cast<const f32x4*>(stage_base + MID_S_OFF + c * 4);
            add4[ni] = *reinterpret_cast<const f32x4*>(stage_base + MID_C_OFF + c * 4);
        } else {
            s4[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
            add4[ni] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + mid_attn_col(c, head, hidden)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        f32x2 rn = {1.f, 0.f};
        if constexpr (FOLD == MID_FOLD_CONSUMER) rn = *reinterpret_cast<const f32x2*>(stage_base + MID_RS_OFF + (mi * 16 + lrow) * 8);
#pragma unroll
        for (int ni = 0; ni < 3; ++ni) {
            f32x4 v = acc[ni][mi];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rn[0] * v[r] + (rn[1] * s4[ni][r] + add4[ni][r]);
            const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(tq + (mi * 16 + lrow) * MID_T_STRIDE + (wn * 48 + ni * 16 + lchk * 4) * 2) = pk;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MID_BARRIER();                                       // the four compute waves (the loader waves have ended): the tile is complete
    // ---- 2. wave wn = sample wn: rows [32 wn, +32)
    const char* ts = tq + wn * 32 * MID_T_STRIDE;
    f32x4 st[2][2];                                      // [key tile][query tile]: lane holds S[key = kt*16 + lchk*4 + r][query = qt*16 + lrow]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) st[kt][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 kf[2], qf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            kf[t] = *reinterpret_cast<const bf16x8*>(ts + (t * 16 + lrow) * MID_T_STRIDE + (64 + ks * 32 + lchk * 8) * 2);
            qf[t] = *reinterpret_cast<const bf16x8*>(ts + (t * 16 + lrow) * MID_T_STRIDE + (ks * 32 + lchk * 8) * 2);
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], st[kt][qt], 0, 0, 0);
    }
    // softmax over the 32 keys of each query (qt, lrow): 8 values in this lane, the rest in lanes lrow + 16 j
    const float cs = a.attn_scale_log2e;
    float linv[2];
    bf16x8 pf[2];                                        // P^T operand: k-slot j of lane group lchk = key (j < 4 ? lchk*4 + j : 16 + lchk*4 + j - 4)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float mx = fmaxf(fmaxf(fmaxf(st[0][qt][0], st[0][qt][1]), fmaxf(st[0][qt][2], st[0][qt][3])),
                         fmaxf(fmaxf(st[1][qt][0], st[1][qt][1]), fmaxf(st[1][qt][2], st[1][qt][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc = mx * cs;
        float p[8], l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[kt * 4 + r] = __builtin_amdgcn_exp2f(st[kt][qt][r] * cs - mc); l += p[kt * 4 + r]; }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        linv[qt] = 1.0f / l;
        pf[qt] = (bf16x8){(bf16_t)p[0], (bf16_t)p[1], (bf16_t)p[2], (bf16_t)p[3], (bf16_t)p[4], (bf16_t)p[5], (bf16_t)p[6], (bf16_t)p[7]};
    }
    // O^T[d][query] = sum_key V[key][d] P[query][key]: A operand = V^T rows d = dt*16 + lrow with the same key order in its k-slots
    bf16_t* ob = a.attn_o + (((long)(m0 / 32 + wn) * (hidden / 64) + head) * 32) * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        bf16x8 vf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = (j < 4) ? lchk * 4 + j : 16 + lchk * 4 + (j - 4);
            vf[j] = *reinterpret_cast<const bf16_t*>(ts + key * MID_T_STRIDE + (128 + dt * 16 + lrow) * 2);
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            // lane holds O[query = qt*16 + lrow][d = dt*16 + lchk*4 + r]
            const bf16x4 pk = {(bf16_t)(o[0] * linv[qt]), (bf16_t)(o[1] * linv[qt]), (bf16_t)(o[2] * linv[qt]), (bf16_t)(o[3] * linv[qt])};
            *reinterpret_cast<bf16x4*>(ob + (qt * 16 + lrow) * 64 + dt * 16 + lchk * 4) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------- fused cross-attention epilogue
// q projection + cross-attention in one launch (model/layers.py:183-200 with N = 32 queries, M = 32 condition tokens, head dim 64; the K | V rows
// of the condition are step-invariant and cached by the caller).  The 64 x 64 tile is q of ONE head for TWO whole samples; the loader waves bring
// that head's K and V rows of the two samples into LDS ahead of the operand stream (16 KB, swizzled like the operands).  Compute wave w runs
// sample w >> 1, queries [16 (w & 1), +16): S^T = K Q^T (4 MFMAs), softmax over the 32 keys, O^T = V^T P^T (4 MFMAs) — the math and the operand
// layouts of mid_epilogue_attn.  q never reaches HBM and the attention launch of every cross-attention block is gone.
#define MID_XQ_STRIDE 144            /* bytes per staged q row (64 bf16 + pad): conflict-free 16-B fragment reads */
__device__ __forceinline__ void mid_epilogue_xattn(const GemmArgs& a, f32x4 (&acc)[1][4], int m0, int head, int wn, int lane, char* tq, const char* kv) {
    const int lrow = lane & 15, lchk = lane >> 4;
    const f32x4 b4 = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + head * 64 + wn * 16 + lchk * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const f32x4 v = acc[0][mi];
        const bf16x4 pk = {(bf16_t)(v[0] + b4[0]), (bf16_t)(v[1] + b4[1]), (bf16_t)(v[2] + b4[2]), (bf16_t)(v[3] + b4[3])};
        *reinterpret_cast<bf16x4*>(tq + (mi * 16 + lrow) * MID_XQ_STRIDE + (wn * 16 + lchk * 4) * 2) = pk;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MID_BARRIER();                                       // the four compute waves (the loader waves have ended): the q tile is complete
    const int sp = wn >> 1, qh = wn & 1;
    const char* ks = kv + (sp * 2) * 4096;
    const char* vs = ks + 4096;
    const char* qs = tq + (sp * 32 + qh * 16 + lrow) * MID_XQ_STRIDE;
    f32x4 st[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};   // lane holds S[key = kt*16 + lchk*4 + r][query = lrow]
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qs + (k2 * 32 + lchk * 8) * 2);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int row = kt * 16 + lrow;
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + row * 128 + (((k2 * 4 + lchk) ^ ((row >> 1) & 7)) << 4));
            st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, st[kt], 0, 0, 0);
        }
    }
    const float cs = a.attn_scale_log2e;
    float mx = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])), fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * cs;
    float p[8], l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[kt * 4 + r] = __builtin_amdgcn_exp2f(st[kt][r] * cs - mc); l += p[kt * 4 + r]; }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float linv = 1.0f / l;
    // P^T operand: k-slot j of lane group lchk = key (j < 4 ? lchk*4 + j : 16 + lchk*4 + j - 4): the accumulator layout's key order, on both operands
    const bf16x8 pf = {(bf16_t)p[0], (bf16_t)p[1], (bf16_t)p[2], (bf16_t)p[3], (bf16_t)p[4], (bf16_t)p[5], (bf16_t)p[6], (bf16_t)p[7]};
    bf16_t* ob = a.attn_o + ((((long)(m0 / 32 + sp)) * (a.N / 64) + head) * 32 + qh * 16 + lrow) * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        bf16x8 vf;
        const int c = dt * 2 + (lrow >> 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = (j < 4) ? lchk * 4 + j : 16 + lchk * 4 + (j - 4);
            vf[j] = *reinterpret_cast<const bf16_t*>(vs + key * 128 + ((c ^ ((key >> 1) & 7)) << 4) + (lrow & 7) * 2);
        }
        const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // lane holds O[query = lrow][d = dt*16 + lchk*4 + r]
        const bf16x4 pk = {(bf16_t)(o[0] * linv), (bf16_t)(o[1] * linv), (bf16_t)(o[2] * linv), (bf16_t)(o[3] * linv)};
        *reinterpret_cast<bf16x4*>(ob + dt * 16 + lchk * 4) = pk;
    }
}

// ---------------------------------------------------------------------------------------------- kernel
template <int EPI, int BM, int BN, int FOLD = MID_FOLD_NONE, int ATTN = MID_ATTN_NONE>
__global__ __launch_bounds__(512) void gemm_bf16_nt_mid_kernel(const GemmArgs a) {
    static_assert(ATTN != MID_ATTN_SELF || (EPI == EPI_BF16 && BM == 128 && BN == 192 && FOLD != MID_FOLD_PRODUCER), "fused self-attention: the 128 x 192 bf16 form");
    static_assert(ATTN != MID_ATTN_CROSS || (EPI == EPI_BF16 && BM == 64 && BN == 64 && FOLD == MID_FOLD_NONE), "fused cross-attention: the 64 x 64 bf16 form");
    using C = MidCfg<BM, BN>;
    constexpr int NT = C::NT, MT = C::MT;
    extern __shared__ __attribute__((aligned(16))) char smem_mid[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware bijective remap of the 1-D tile index (blocks b, b+8, ... share an XCD's L2), column-major inside the chunk
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tile_m = a.col_major ? wgid % tiles_m : wgid / tiles_n, tile_n = a.col_major ? wgid / tiles_m : wgid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbase = 0;
    const int nkt = a.K / MID_BK;
    const int step = a.step_ptr ? *a.step_ptr : 0;       // device-side SDE step counter (scalar load, before any request of this kernel)

    if (wave >= 4) {                                     // loader waves: the operand stream, nothing else
        mid_loader<BM, BN, FOLD, ATTN>(a, smem_mid, wave - 4, lane, m0, n0, kbase, nkt, step);
        return;
    }

    // ---- compute waves: wave wn owns columns [wn * BN/4, +BN/4) of the tile, all 128 rows
    const int wn = wave;
    const int lrow = lane & 15, lchk = lane >> 4;
    const int sw = (lrow >> 1) & 7;
    // per-lane LDS read bases inside a stage: row * 128 + ((k-half * 4 + lchk) ^ ((row >> 1) & 7)) * 16; fragment i at + i * 2048
    const int xb0 = lrow * 128 + ((lchk ^ sw) << 4), xb1 = lrow * 128 + (((4 + lchk) ^ sw) << 4);
    const int wrow = C::XB + (wn * (BN / 4) + lrow) * 128;
    const int wb0 = wrow + ((lchk ^ sw) << 4), wb1 = wrow + (((4 + lchk) ^ sw) << 4);

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    MID_STAMP_DECL();
    MID_STAMP(0);
    // While the first K-tile is on its way the compute waves have nothing to do but one thing — the W-panel touch: the tiles_m workgroups that
    // share this W column panel (cold in HBM at every SDE step) each pull their share of its 128-B lines towards the XCD's L2 — K-tile j of the
    // panel by the workgroup with tile_m == j mod tiles_m, one line per lane, one dword-sized LDS-DMA per wave and K-tile into the (still
    // unused) staging area.
#ifndef MID_NO_TOUCH
    {
        const int prow = wn * 64 + lane;                 // row of the panel this lane touches (BN = 128: waves 0, 1 cover it)
        if (prow < BN) {
            const char* wl = reinterpret_cast<const char*>(a.W + (long)(ATTN == MID_ATTN_SELF ? mid_attn_col(prow, tile_n, a.N / 3) : n0 + prow) * a.ldw + kbase);
            char* dst = smem_mid + C::RING + wave * 4096;
            int cnt = 0;
            for (int j = tile_m + (tile_m < 3 ? tiles_m : 0); j < nkt && cnt < 6; j += tiles_m, ++cnt)   // (K-tiles 0-2 are requested at once anyway; 6 x 256 B: the staging area's head)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wl + (long)j * (MID_BK * 2)),
                                                 (__attribute__((address_space(3))) void*)(dst + cnt * 256), 4, 0, 0);
        }
    }
#endif
    // (Also tried here: pulling the NEXT GEMM's weights towards the Infinity Cache, 1 / 256 of them per workgroup — 2.3 % SLOWER per SDE step
    //  at the shipped 32-token config, 2.5 % at the ViPC share: the extra HBM requests compete with this kernel's own first K-tiles.)
    MID_BARRIER();                                       // prologue barrier: K-tile 0 has landed
    MID_STAMP(1);
    // BN = 128 residual epilogue: the wave's whole residual tile (BM rows x 32 columns fp32: 2 x 16 B per lane and 16-row pass) is requested
    // here and arrives under the main loop.  NOT before the barrier: a CU's memory pipeline is a FIFO — requested first, these
    // 32-64 KB of (HBM / Infinity-Cache) misses held the loaders' first K-tiles back by 2-4 us (profiles/r04_mid_stamps.txt).
    f32x4 rpre[MT][NT <= 2 ? NT : 1];
    if constexpr (EPI == EPI_RESID_F32 && NT <= 2) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int it = 0; it < NT; ++it) rpre[mi][it] = mid_resid_load<NT>(a, m0, n0 + wn * (BN / 4), lane, mi, it);
    }

    if constexpr (BM == 128 && NT >= 3) {
        // (NT = 4 | 3: BN = 256 | 192)  K-tile = 4 blocks of 4 NT MFMAs: (half 0, rows 0-63), (half 0, rows 64-127), (half 1, rows 0-63), (half 1, rows 64-127);
        // each block's fragments are read during the block before it
        bf16x8 wa[NT], wb[NT], xa[4], xb[4];
        auto ld_w = [&](bf16x8 (&w)[NT], const char* st, int off) {
#pragma unroll
            for (int i = 0; i < NT; ++i) w[i] = *reinterpret_cast<const bf16x8*>(st + off + i * 2048);
        };
        auto ld_x = [&](bf16x8 (&x)[4], const char* st, int off, int part) {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const bf16x8*>(st + off + (part * 4 + i) * 2048);
        };
        auto mm = [&](const bf16x8 (&w)[NT], const bf16x8 (&x)[4], int part) {
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
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
            mid_interleave<4, 2, 4 * NT>();
            ld_w(wb, st, wb1); ld_x(xa, st, xb1, 0);
            mm(wa, xb, 1);
            mid_interleave<NT + 4, 1, 4 * NT>();
            ld_x(xb, st, xb1, 1);                        // last read of this stage
            mm(wb, xa, 0);
            mid_interleave<4, 2, 4 * NT>();
            MID_LGKM0();
            MID_BARRIER();                               // barrier kt: K-tile kt+1 landed; this stage may be refilled
            slot = slot + 1 == C::NS ? 0 : slot + 1;
            st = smem_mid + slot * C::STAGE;
            ld_w(wa, st, wb0); ld_x(xa, st, xb0, 0);
            mm(wb, xb, 1);
            mid_interleave<NT + 4, 1, 4 * NT>();
            MID_LGKM0();                                 // (free behind the MFMAs; lets hipcc's wait insertion open the next trip with known counters)
        }
        ld_x(xb, st, xb0, 1);
        mm(wa, xa, 0);
        mid_interleave<4, 2, 4 * NT>();
        ld_w(wb, st, wb1); ld_x(xa, st, xb1, 0);
        mm(wa, xb, 1);
        mid_interleave<NT + 4, 1, 4 * NT>();
        ld_x(xb, st, xb1, 1);
        mm(wb, xa, 0);
        mid_interleave<4, 2, 4 * NT>();
        mm(wb, xb, 1);
    } else if constexpr (BM == 64) {
        // 64 x 128 | 64 x 64 tile: a wave owns 64 rows x 32 | 16 columns (NT x 4 accumulator tiles), a K-tile is ONE block of 16 | 8 MFMAs; the
        // whole next K-tile's fragments (2 NT W + 8 X) are read into the other register set while this one is multiplied.  The reads of K-tile kt are complete
        // before its MFMAs start, so the barrier that admits K-tile kt+1 also frees K-tile kt's stage.
        bf16x8 wa[2][NT], xa[2][4], wb[2][NT], xb[2][4];
        auto ld = [&](bf16x8 (&w)[2][NT], bf16x8 (&x)[2][4], const char* st) {
#pragma unroll
            for (int i = 0; i < NT; ++i) { w[0][i] = *reinterpret_cast<const bf16x8*>(st + wb0 + i * 2048); w[1][i] = *reinterpret_cast<const bf16x8*>(st + wb1 + i * 2048); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[0][i] = *reinterpret_cast<const bf16x8*>(st + xb0 + i * 2048); x[1][i] = *reinterpret_cast<const bf16x8*>(st + xb1 + i * 2048); }
        };
        auto mm = [&](const bf16x8 (&w)[2][NT], const bf16x8 (&x)[2][4]) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[h][ni], x[h][mi], acc[ni][mi], 0, 0, 0);
        };
        constexpr int KT_MFMA = 8 * NT, KT_RD = (2 * NT + 8) < KT_MFMA ? (2 * NT + 8) : KT_MFMA;   // MFMAs / interleaved reads per K-tile
        int slot = 0;
        auto next_stage = [&]() { slot = slot + 1 == C::NS ? 0 : slot + 1; return smem_mid + slot * C::STAGE; };
        ld(wa, xa, smem_mid);
        MID_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        int kt = 0;
        for (; kt + 2 < nkt; kt += 2) {
            MID_BARRIER();                               // barrier kt: K-tile kt+1 landed (and K-tile kt's stage is free: its reads returned before this trip)
            ld(wb, xb, next_stage());
            mm(wa, xa);
            mid_interleave<KT_RD, 1, KT_MFMA>();
            MID_LGKM0();
            MID_BARRIER();                               // barrier kt+1
            ld(wa, xa, next_stage());
            mm(wb, xb);
            mid_interleave<KT_RD, 1, KT_MFMA>();
            MID_LGKM0();
        }
        if (kt + 1 < nkt) {
            MID_BARRIER();
            ld(wb, xb, next_stage());
            mm(wa, xa);
            mid_interleave<KT_RD, 1, KT_MFMA>();
            mm(wb, xb);
        } else mm(wa, xa);
    } else {
        // 128 x 128: a wave owns 32 columns (2 accumulator tiles x 8 row tiles); K-tile = 2 blocks of 16 MFMAs (half 0, half 1)
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
            slot = slot + 1 == C::NS ? 0 : slot + 1;
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
    // The epilogue must not be scheduled in among the last MFMAs: left to itself hipcc starts the epilogue's LDS reads a few instructions behind
    // MFMAs that still read the SAME registers as their C operand, and the returned data then overwrites an operand in flight — measured: the
    // LN-folded consumer lost the mean term in some lanes of the first 16-row pass, differently from run to run (tools/dbg/mid_fold_dbg.py;
    // profiles/r04_mid_epilogue_hazard.txt).  A scheduling fence + the matrix pipe's depth in wait states; once per kernel.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ATTN == MID_ATTN_SELF) {
        // every compute wave is past its last ring read before the ring becomes the q | k | v tile
        MID_BARRIER();
        mid_epilogue_attn<FOLD>(a, acc, m0, tile_n, wn, lane, smem_mid, smem_mid + C::RING);
    } else if constexpr (ATTN == MID_ATTN_CROSS) {
        MID_BARRIER();                                   // (as above: the ring becomes the q tile)
        mid_epilogue_xattn(a, acc, m0, tile_n, wn, lane, smem_mid, smem_mid + C::LDS);
    } else
        mid_epilogue<EPI, BM, BN, FOLD>(a, acc, m0, n0 + wn * (BN / 4), wn, lane, smem_mid + C::RING + wave * 4096, smem_mid + C::RING, rpre, step);
#ifdef MID_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MID_STAMP(3);
    if (wave == 0) MID_STAMP_FLUSH(0);
#endif
}

// ---------------------------------------------------------------------------------------------- launcher
// Shapes this kernel takes (everything else stays with gemm_bf16.hip): whole column tiles, 16-byte aligned rows, K a multiple of 64,
// and a problem of at most two rounds of workgroups (longer ones belong to the persistent 256^2 kernel or the streaming v1 forms).
static int mid_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LDT_GEMM_MID"); v = e ? atoi(e) : 1; }
    return v;
}

// Tile shape for (M, N): this path is bound by what a CU can take in through the L2 -> LDS DMA (~80 GB/s, profiles/r04_mid_stamps.txt), so
// the cost of a shape is rounds x bytes per workgroup and K-tile ~ ceil(workgroups / 256) x (BM + BN).  -> (BM << 16) | BN, 0 = not taken.
// fold: 0 = any form; 1 = LN-folded producer (a wave's 32 columns are the statistics granule: BN = 128 forms only);
//       2 = LN-folded consumer (the loader waves' statistics pass: 128-row forms only)
static int mid_shape_for(int epi, const GemmArgs* a, int fold);
int ldt_gemm_mid_shape(int epi, const GemmArgs* a) { return mid_shape_for(epi, a, 0); }

static int mid_shape_for(int epi, const GemmArgs* a, int fold) {
    const int mode = mid_env();                          // 0 off, 1 automatic; tools/dbg: 256 / 192 / 128 / 64 pin 128x256 / 128x192 / 128x128 / 64x128
    if (!mode) return 0;
    if (!(epi == EPI_F32 || epi == EPI_BF16 || epi == EPI_GELU_BF16 || epi == EPI_RESID_F32)) return 0;
    if (a->K % MID_BK != 0 || a->N % 64 != 0 || a->M < 64 || a->ldo % 8 != 0) return 0;
    if (epi == EPI_RESID_F32 && (a->ldr % 4 != 0 || (a->gate && a->gate_sample_stride % 4 != 0))) return 0;
    struct { int bm, bn; } cand[5] = {{128, 256}, {128, 192}, {128, 128}, {64, 128}, {64, 64}};
    const bool bf16_out = epi == EPI_BF16 || epi == EPI_GELU_BF16;
    if (fold == 0) {
    if (mode == 256) return a->N % 256 == 0 ? (128 << 16) | 256 : 0;
    if (mode == 128) return (128 << 16) | 128;
    if (mode == 64) return (64 << 16) | 128;
    if (mode == 6464) return (64 << 16) | 64;
    }
    long best_cost = 0; int best = 0;
    if (mode == 192) return (a->N % 192 == 0 && bf16_out) ? (128 << 16) | 192 : 0;
    for (int i = 0; i < 5; ++i) {
        if (a->N % cand[i].bn != 0 || (cand[i].bn == 192 && !bf16_out)) continue;
        if ((fold == 1 && cand[i].bn != 128) || (fold == 2 && cand[i].bm != 128)) continue;
        const long wgs = (long)((a->M + cand[i].bm - 1) / cand[i].bm) * (a->N / cand[i].bn);
        if (wgs < 48 || wgs > 2 * LDT_NUM_CUS) continue;
        const long cost = ((wgs + LDT_NUM_CUS - 1) / LDT_NUM_CUS) * (cand[i].bm + cand[i].bn);
        if (!best || cost < best_cost) { best = (cand[i].bm << 16) | cand[i].bn; best_cost = cost; }
    }
    return best;
}

template <int EPI, int BM, int BN>
static int mid_launch_t(const GemmArgs* a_in, hipStream_t stream) {
    using C = MidCfg<BM, BN>;
    GemmArgs a = *a_in;
    const long tm = (a.M + BM - 1) / BM, tn = a.N / BN;
    static const int map_env = getenv("LDT_GEMM_MID_MAP") ? atoi(getenv("LDT_GEMM_MID_MAP")) : -1;   // tools/dbg: 0 row-major, 1 column-major
    a.col_major = map_env >= 0 ? map_env : 1;
    LDT_ENSURE_LDS((&gemm_bf16_nt_mid_kernel<EPI, BM, BN>), C::LDS, "gemm_mid");
    hipLaunchKernelGGL((gemm_bf16_nt_mid_kernel<EPI, BM, BN>), dim3((unsigned)(tm * tn)), dim3(512), C::LDS, stream, a);
    return ldt_check_launch("gemm_bf16_nt_mid");
}

int ldt_gemm_mid_launch(int epi, int shape, const GemmArgs* a, hipStream_t stream) {
    const int bm = shape >> 16, bn = shape & 0xffff;
    if (bn == 192) return epi == EPI_BF16 ? mid_launch_t<EPI_BF16, 128, 192>(a, stream) : mid_launch_t<EPI_GELU_BF16, 128, 192>(a, stream);
#define MID_CASE(E)                                                                                            \
    case E: return bm == 64 ? (bn == 64 ? mid_launch_t<E, 64, 64>(a, stream) : mid_launch_t<E, 64, 128>(a, stream)) : bn == 256 ? mid_launch_t<E, 128, 256>(a, stream) \
                                                                               : mid_launch_t<E, 128, 128>(a, stream)
    switch (epi) {
        MID_CASE(EPI_F32);
        MID_CASE(EPI_BF16);
        MID_CASE(EPI_GELU_BF16);
        MID_CASE(EPI_RESID_F32);
        default: ldt_set_error("gemm_mid: epilogue %d not built", epi); return LDT_EARG;
    }
#undef MID_CASE
}

// LN-folded forms (ldt_gemm_lnfold_launch's small-batch route: statistics per 32 columns): producer = the N = hidden residual GEMMs on
// 64 x 128 / 128 x 128 tiles, consumer = QKV / MLP-up on 128 x 256 / 128 x 128 tiles.  -> true when this kernel took the launch.
template <int EPI, int BM, int BN, int FOLD>
static int mid_fold_launch_t(const GemmArgs& a, hipStream_t stream) {
    constexpr int lds = MidCfg<BM, BN>::LDS;
    LDT_ENSURE_LDS((&gemm_bf16_nt_mid_kernel<EPI, BM, BN, FOLD>), lds, "gemm_mid(fold)");
    const unsigned grid = (unsigned)(((a.M + BM - 1) / BM) * (a.N / BN));
    hipLaunchKernelGGL((gemm_bf16_nt_mid_kernel<EPI, BM, BN, FOLD>), dim3(grid), dim3(512), lds, stream, a);
    return ldt_check_launch("gemm_bf16_nt_mid(fold)");
}

// would the folded form of this GEMM (M x N x K, statistics per 32 columns) be taken?  (Score.can_fold / ldt_gemm_lnfold_v1_route)
bool ldt_gemm_mid_lnfold_takes(int epi, int M, int N, int K) {
    if (!mid_env()) return false;
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.ldo = N; g.ldr = N;
    const int shape = mid_shape_for(epi, &g, epi == EPI_RESID_F32 ? 1 : 2);
    if (!shape) return false;
    if (epi == EPI_RESID_F32) return N % 32 == 0;
    return K % 32 == 0 && K / 32 <= 32 && K / MID_BK >= MidCfg<128, 256>::NS + 2;
}

bool ldt_gemm_mid_lnfold_try(int epi, const GemmArgs* a_in, hipStream_t stream, int* status) {
    if (!mid_env()) return false;
    const int shape = mid_shape_for(epi, a_in, epi == EPI_RESID_F32 ? 1 : 2);
    if (!shape) return false;
    const int bm = shape >> 16, bn = shape & 0xffff;
    GemmArgs a = *a_in;
    static const int map_env = getenv("LDT_GEMM_MID_MAP") ? atoi(getenv("LDT_GEMM_MID_MAP")) : -1;
    a.col_major = map_env >= 0 ? map_env : 1;
    if (epi == EPI_RESID_F32) {                          // producer: a wave's 32 columns are the statistics granule
        if (bn != 128 || a.stats_parts * 32 != a.N) return false;
        *status = bm == 64 ? mid_fold_launch_t<EPI_RESID_F32, 64, 128, MID_FOLD_PRODUCER>(a, stream)
                           : mid_fold_launch_t<EPI_RESID_F32, 128, 128, MID_FOLD_PRODUCER>(a, stream);
        return true;
    }
    // consumer: the loader waves' statistics pass assumes the whole ring is in flight and <= 32 partials per row
    if (bm != 128 || a.stats_parts > 32 || a.stats_parts * 32 != a.K || a.K / MID_BK < MidCfg<128, 256>::NS + 2) return false;
    if (epi == EPI_BF16)
        *status = bn == 256 ? mid_fold_launch_t<EPI_BF16, 128, 256, MID_FOLD_CONSUMER>(a, stream)
                : bn == 192 ? mid_fold_launch_t<EPI_BF16, 128, 192, MID_FOLD_CONSUMER>(a, stream)
                            : mid_fold_launch_t<EPI_BF16, 128, 128, MID_FOLD_CONSUMER>(a, stream);
    else if (epi == EPI_GELU_BF16)
        *status = bn == 256 ? mid_fold_launch_t<EPI_GELU_BF16, 128, 256, MID_FOLD_CONSUMER>(a, stream)
                : bn == 192 ? mid_fold_launch_t<EPI_GELU_BF16, 128, 192, MID_FOLD_CONSUMER>(a, stream)
                            : mid_fold_launch_t<EPI_GELU_BF16, 128, 128, MID_FOLD_CONSUMER>(a, stream);
    else return false;
    return true;
}

// QKV projection + self-attention in one launch (mid_epilogue_attn): 32-token samples, head dim 64, N = 3 * hidden with hidden % 64 == 0,
// whole 128-row tiles (four samples each).  `folded`: a = the LN-folded consumer's arguments (stats per 32 columns).  LDT_QKV_ATTN=0: off (A/B).
template <int FOLD>
static int mid_qkv_attn_launch(const GemmArgs& a, hipStream_t stream) {
    constexpr int lds = MidCfg<128, 192>::LDS;
    LDT_ENSURE_LDS((&gemm_bf16_nt_mid_kernel<EPI_BF16, 128, 192, FOLD, MID_ATTN_SELF>), lds, "gemm_mid(qkv+attention)");
    hipLaunchKernelGGL((gemm_bf16_nt_mid_kernel<EPI_BF16, 128, 192, FOLD, MID_ATTN_SELF>), dim3((unsigned)((a.M / 128) * (a.N / 192))), dim3(512), lds, stream, a);
    return ldt_check_launch("gemm_bf16_nt_mid(qkv+attention)");
}

bool ldt_gemm_mid_qkv_attn_try(const GemmArgs* a_in, int tokens, int head_dim, bool folded, hipStream_t stream, int* status) {
    static const bool on = !(getenv("LDT_QKV_ATTN") && atoi(getenv("LDT_QKV_ATTN")) == 0);
    if (!on || !mid_env() || tokens != 32 || head_dim != 64 || !a_in->attn_o) return false;
    const GemmArgs& g = *a_in;
    if (g.N % 192 != 0 || (g.N / 3) % 64 != 0 || g.M % 128 != 0 || g.K % MID_BK != 0 || g.K / MID_BK < MidCfg<128, 192>::NS + 2) return false;
    // the regime of this family (mid_shape_for): at most two rounds of workgroups — of the workgroups this launch may use when a sub-batch
    // stream caps its grids (a->max_wgs); larger batches (B >= 256 at 16 heads) stay with the persistent 256^2 GEMM + the resident
    // attention kernel.  No lower bound: a small batch saves a launch here and has nothing to lose to a bigger tile.
    const long wgs = (long)(g.M / 128) * (g.N / 192);
    const long cus = g.max_wgs > 0 && g.max_wgs < LDT_NUM_CUS ? g.max_wgs : LDT_NUM_CUS;
    if (wgs > 2 * cus) return false;
    if (folded && (g.stats_parts > 32 || g.stats_parts * 32 != g.K || !g.stats_in || !g.fold_S || !g.fold_C)) return false;
    // the argument checks of ldt_gemm_launch / ldt_gemm_lnfold_launch, which this route runs ahead of (a violation returns false:
    // the generic path then reports it)
    if (!g.X || !g.W || !ldt_aligned16(g.X) || !ldt_aligned16(g.W) || g.ldx < g.K || g.ldw < g.K || g.ldx % 8 != 0 || g.ldw % 8 != 0) return false;
    if (!ldt_aligned16(g.attn_o) || (g.bias && !ldt_aligned16(g.bias))) return false;
    if (folded && (!ldt_aligned16(g.stats_in) || !ldt_aligned16(g.fold_S) || !ldt_aligned16(g.fold_C) || g.fold_step_stride % 4 != 0)) return false;
    GemmArgs a = g;
    a.col_major = 1;
    *status = folded ? mid_qkv_attn_launch<MID_FOLD_CONSUMER>(a, stream) : mid_qkv_attn_launch<MID_FOLD_NONE>(a, stream);
    return true;
}

// q projection + cross-attention in one launch (mid_epilogue_xattn): 32-token samples, 32 condition tokens, head dim 64, N = hidden (a multiple
// of 64), whole 64-row tiles (two samples each), one or two rounds of workgroups.  a->attn_k / attn_v: the cached K | V rows of the condition
// (row stride attn_ldkv, sample stride attn_kv_batch_stride, elements).  LDT_Q_XATTN=0: off (A/B).
bool ldt_gemm_mid_q_xattn_try(const GemmArgs* a_in, int tokens, int cond_tokens, int head_dim, hipStream_t stream, int* status) {
    static const bool on = !(getenv("LDT_Q_XATTN") && atoi(getenv("LDT_Q_XATTN")) == 0);
    const GemmArgs& g = *a_in;
    if (!on || !mid_env() || tokens != 32 || cond_tokens != 32 || head_dim != 64 || !g.attn_o || !g.attn_k || !g.attn_v) return false;
    if (g.N % 64 != 0 || g.M % 64 != 0 || g.K % MID_BK != 0 || g.K / MID_BK < 2) return false;
    const long wgs = (long)(g.M / 64) * (g.N / 64);
    const long cus = g.max_wgs > 0 && g.max_wgs < LDT_NUM_CUS ? g.max_wgs : LDT_NUM_CUS;
    if (wgs < 48 || wgs > 2 * cus) return false;
    if (!g.X || !g.W || !ldt_aligned16(g.X) || !ldt_aligned16(g.W) || g.ldx < g.K || g.ldw < g.K) return false;
    if (!ldt_aligned16(g.attn_o) || !ldt_aligned16(g.attn_k) || !ldt_aligned16(g.attn_v) || g.attn_ldkv % 8 != 0 || g.attn_kv_batch_stride % 8 != 0 ||
        (g.bias && !ldt_aligned16(g.bias)) || g.ldx % 8 != 0 || g.ldw % 8 != 0)
        return false;
    GemmArgs a = g;
    a.col_major = 1;
    constexpr int lds = MidCfg<64, 64>::LDS + MID_XKV_BYTES;
    auto launch = [&]() -> int {
        LDT_ENSURE_LDS((&gemm_bf16_nt_mid_kernel<EPI_BF16, 64, 64, MID_FOLD_NONE, MID_ATTN_CROSS>), lds, "gemm_mid(q+cross-attention)");
        hipLaunchKernelGGL((gemm_bf16_nt_mid_kernel<EPI_BF16, 64, 64, MID_FOLD_NONE, MID_ATTN_CROSS>), dim3((unsigned)wgs), dim3(512), lds, stream, a);
        return ldt_check_launch("gemm_bf16_nt_mid(q+cross-attention)");
    };
    *status = launch();
    return true;
}
