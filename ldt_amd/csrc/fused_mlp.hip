// Fused MLP half of a Set-Transformer ResidualBlock for narrow channel counts (the Compressor: d = 128, 4d hidden;
// reference model/layers.py:219 / :226 with MLP :110-133 and the LayerNorm wrapper tools/utils.py:127-133):
//
//     x  <-  x + gate * ( W_dn · GELU( W_up · h + b_up ) + b_dn ),      h = LN(x)[*w + b]  or  LN(x)*(1+scale)+shift
//
// in ONE pass over x.  Unfused this is LayerNorm + two GEMMs: x is read twice and written once in fp32, h written and
// read in bf16, and the 4d-wide hidden activation written and read in bf16 — 22 B/channel·row of HBM traffic against
// the 8 B/channel·row (x in, x out) this kernel moves.  At d = 128 the GEMMs have K = 128 / 512: far below the MFMA
// ridge, so the unfused path is HBM-bound and the fused one becomes MFMA/LDS-bound.
//
// Workgroup = 4 waves x 32 rows = 128 rows; two workgroups share a CU (80 KB LDS each, <= 256 VGPRs).  Per wave:
//   1. LN: the wave's 32 rows are read in the MFMA accumulator layout (16 B per lane), statistics in-lane + two
//      cross-lane adds, h -> bf16 -> wave-private rows of an LDS image (XOR-swizzled 16-B chunks) -> read back ONCE as
//      the MFMA operand fragments of the wave's rows, which then stay in 8*C/32 VGPRs; the image's LDS becomes the
//      second weight buffer.  Without a gate the x values just read initialise the output accumulators (x + b_dn).
//   2. for each chunk of 64 hidden units (weights W_up[64 x C], W_dn[C x 64] staged L2 -> LDS by LDS-DMA, shared by
//      the 4 waves, double-buffered: chunk c+1 is in flight during the whole of chunk c, one barrier per chunk):
//        U^T[64 x 32]  = W_up_c · h^T          (mfma 16x16x32 bf16; operands swapped so a lane holds 4 consecutive
//        U = GELU(U + b_up) -> bf16 -> LDS       hidden units of one row -> one 8-B LDS store)
//        O^T[C x 32] += W_dn_c · U^T           (a lane holds 4 consecutive output channels of one row)
//   3. epilogue from registers, 16 B per lane: the accumulators ARE the new x (no gate), or x + gate * (O + b_dn).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

#define LDT_MLP_RT_DEFAULT 2          /* 16-row tiles per wave of ln_mlp_resid_kernel: 2 = 4 waves x 32 rows, 1 = 8 waves x 16 rows */

// tools/dbg build (-DMLP_STAMPS): wall-clock stamps (s_memrealtime, 10 ns) of wave 0 of every workgroup, kept in a VGPR (lane i = stamp i) and
// stored once at the wave's end -> g_mlp_stamps[workgroup][64]; lane 62 = HW_ID, lane 63 = XCC_ID (tools/dbg/mlp_stamps.py)
#ifdef MLP_STAMPS
__device__ long long* g_mlp_stamps;
extern "C" int ldt_dbg_mlp_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_stamps), &p, sizeof(p)); }
#define MLP_STAMP_DECL() long long _stv = 0
#define MLP_STAMP(idx) do { const long long _t = __builtin_amdgcn_s_memrealtime(); _stv = (lane == (idx)) ? _t : _stv; } while (0)
#define MLP_STAMP_FLUSH() do { if (wave == 0 && g_mlp_stamps) { \
        const long long _h = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)), _x = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); \
        _stv = lane == 62 ? _h : lane == 63 ? _x : _stv; g_mlp_stamps[(long)blockIdx.x * 64 + lane] = _stv; } } while (0)
#else
#define MLP_STAMP_DECL()
#define MLP_STAMP(idx)
#define MLP_STAMP_FLUSH()
#endif

namespace {

template <int C> struct MlpCfg {
    static constexpr int HID = 4 * C;
    static constexpr int HCH = 64;                         // hidden units per chunk
    static constexpr int NCH = HID / HCH;
    static constexpr int ROWB = C * 2;                     // bytes of one bf16 row of width C
    static constexpr int CB = C / 8;                       // 16-B chunks per such row
    static constexpr int H_BYTES = 128 * ROWB;             // h image: 128 rows x C
    static constexpr int U_BYTES = 128 * HCH * 2;          // GELU output chunk: 128 rows x 64
    static constexpr int WUP_BYTES = HCH * ROWB;           // W_up chunk: 64 rows x C
    static constexpr int WDN_BYTES = C * HCH * 2;          // W_dn chunk: C rows x 64
    static constexpr int LDS = H_BYTES + U_BYTES + WUP_BYTES + WDN_BYTES;   // h image (later weight set 1) | U | weight set 0
};

// bank-conflict swizzle of the 16-B chunk index inside a row of `CB` chunks (both-sides rule: the same function on
// the LDS-DMA source address, on plain LDS stores and on every ds_read)
template <int CB> __device__ __forceinline__ int swz(int row) { return CB == 16 ? (row & 15) : ((row >> 1) & 7); }

template <int C, int NW = 4>
__device__ __forceinline__ void stage_wup(const bf16_t* __restrict__ w_up, int hid0, char* lds, int wave, int lane) {
    constexpr int CB = MlpCfg<C>::CB, RPP = 64 / CB, PIECES = MlpCfg<C>::WUP_BYTES / 1024;    // rows per 1-KiB piece
#pragma unroll
    for (int p = wave; p < PIECES; p += NW) {
        const int row = p * RPP + lane / CB, phys = lane % CB;
        const bf16_t* src = w_up + (long)(hid0 + row) * C + ((phys ^ swz<CB>(row)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
    }
}
template <int C, int NW = 4>
__device__ __forceinline__ void stage_wdn(const bf16_t* __restrict__ w_dn, int hid0, char* lds, int wave, int lane) {
    constexpr int PIECES = MlpCfg<C>::WDN_BYTES / 1024;                                       // 8 rows of 128 B per piece
#pragma unroll
    for (int p = wave; p < PIECES; p += NW) {
        const int row = p * 8 + (lane >> 3), phys = lane & 7;
        const bf16_t* src = w_dn + (long)row * MlpCfg<C>::HID + hid0 + ((phys ^ swz<8>(row)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
    }
}

// LayerNorm (+ affine | per-sample modulate) of a wave's 32 rows, read in the MFMA accumulator layout (lane: row
// rt*16 + lrow, channels n*16 + lq*4 .. +3); h -> bf16 -> the wave's rows of the swizzled LDS image `Hs` -> returned as the
// MFMA operand fragments hf[ks][rt] (8 bf16 of row lrow, k = ks*32 + lq*8 ..).  `xv` keeps the fp32 inputs for the caller.
struct LnSrc {
    const float* x; long ldx; long M;
    const float* ln_w; const float* ln_b; const float* shift; const float* scale; long mod_sample_stride; int rows_per_sample;
};
template <int C, bool LOAD = true, int RT = 2>    // RT = 16-row tiles per wave; LOAD = false: `xv` already holds the rows (the MLP kernel's freshly updated x)
__device__ __forceinline__ void ln_rows_to_frags(const LnSrc& a, long row0, char* Hs, int wave, int lrow, int lq,
                                                 f32x4 (&xv)[C / 16][RT], bf16x8 (&hf)[C / 32][RT], long long* stv = nullptr) {
    using K = MlpCfg<C>;
    constexpr int NN = C / 16;
#ifdef MLP_STAMPS
    const int lane = lq * 16 + lrow;
    long long _stv = stv ? *stv : 0;
#define LN_STAMP(idx) MLP_STAMP(idx)
#else
#define LN_STAMP(idx)
#endif
    // Every request of this phase goes out in ONE flight: the rows of all RT tiles, then the LayerNorm vectors of the lane's channels (they
    // used to be loaded where they are used, inside the per-channel loops under `if (a.ln_w)`: a load + wait per channel group and tile).
    // Worth 0.5 % only: in-kernel stamps (tools/dbg/mlp_stamps.py, profiles/r05_fused_mlp_analysis.txt) show this phase's 10.6 us per
    // workgroup are INSTRUCTION ISSUE beside the co-resident workgroup's chunk loop (4.5 us to get the 32 requests out, by which time
    // the data has landed; 4-5 us of LayerNorm arithmetic), not memory round trips.
    long grow[RT], moff[RT];
    const bool aff = a.ln_w != nullptr, mod = a.shift != nullptr;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        grow[rt] = row0 + rt * 16 + lrow;
        grow[rt] = grow[rt] < a.M ? grow[rt] : a.M - 1;                // tail rows: clamp, never stored
        moff[rt] = !mod ? 0 : a.M < (1L << 31) ? (long)((unsigned)grow[rt] / (unsigned)a.rows_per_sample) * a.mod_sample_stride   // (no 64-bit division routine)
                                               : (grow[rt] / a.rows_per_sample) * a.mod_sample_stride;
        if (LOAD) {
            const float* xr = a.x + grow[rt] * a.ldx + lq * 4;
#pragma unroll
            for (int n = 0; n < NN; ++n) xv[n][rt] = *reinterpret_cast<const f32x4*>(xr + n * 16);
        }
    }
    LN_STAMP(50);
    // va | vb: the affine (w, b); without an affine LayerNorm, tile 0's (shift, scale) — which is every tile's when the wave's rows share a sample
    f32x4 va[NN], vb[NN];
    if (aff) {
#pragma unroll
        for (int n = 0; n < NN; ++n) { va[n] = *reinterpret_cast<const f32x4*>(a.ln_w + n * 16 + lq * 4); vb[n] = *reinterpret_cast<const f32x4*>(a.ln_b + n * 16 + lq * 4); }
    } else if (mod) {
#pragma unroll
        for (int n = 0; n < NN; ++n) { va[n] = *reinterpret_cast<const f32x4*>(a.shift + moff[0] + n * 16 + lq * 4); vb[n] = *reinterpret_cast<const f32x4*>(a.scale + moff[0] + n * 16 + lq * 4); }
    }
    LN_STAMP(51);
#ifdef MLP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LN_STAMP(52);
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NN; ++n) s += (xv[n][rt][0] + xv[n][rt][1]) + (xv[n][rt][2] + xv[n][rt][3]);
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = xv[n][rt][j] - mean; q += d * d; }
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q / (float)C + 1e-6f);
        // (shift, scale) of this tile's rows: tile 0's registers when no lane's row left tile 0's sample (the Compressor: rows_per_sample is a
        // multiple of 32); loaded channel group by channel group otherwise (and behind an affine LayerNorm, which no shipped block combines
        // with a modulation)
        const bool mod_regs = mod && !aff && (rt == 0 || !__any(moff[rt] != moff[0]));
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            f32x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = (xv[n][rt][j] - mean) * rstd;
            if (aff) {
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = h[j] * va[n][j] + vb[n][j];
            }
            if (mod_regs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = h[j] * (1.f + vb[n][j]) + va[n][j];
            } else if (mod) {
                const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + moff[rt] + n * 16 + lq * 4), sc = *reinterpret_cast<const f32x4*>(a.scale + moff[rt] + n * 16 + lq * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = h[j] * (1.f + sc[j]) + sh[j];
            }
            const int hr = wave * (16 * RT) + rt * 16 + lrow;           // (rows of the image are wave-private)
            *reinterpret_cast<bf16x4*>(Hs + hr * K::ROWB + (((n * 2 + (lq >> 1)) ^ swz<K::CB>(hr)) << 4) + (lq & 1) * 8) =
                (bf16x4){(bf16_t)h[0], (bf16_t)h[1], (bf16_t)h[2], (bf16_t)h[3]};
        }
    }
    LN_STAMP(53);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // rows are wave-private: ordering inside the wave suffices
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int hr = wave * (16 * RT) + rt * 16 + lrow;
            hf[ks][rt] = *reinterpret_cast<const bf16x8*>(Hs + hr * K::ROWB + (((ks * 4 + lq) ^ swz<K::CB>(hr)) << 4));
        }
#ifdef MLP_STAMPS
    if (stv) *stv = _stv;
#endif
#undef LN_STAMP
}

// out[rows][N] (bf16) = h . W^T + bias for the wave's 32 rows held as operand fragments `hf`; W streamed in chunks of 64 output
// channels (chunk 0 already requested into W0 by the caller; double-buffered with the h image's LDS), the bf16 result chunk staged
// through the wave's LDS rows so that whole 128-B row pieces are stored.
template <int C, int RT = 2>
__device__ __forceinline__ void linear_chunks(const bf16_t* __restrict__ w, const float* __restrict__ bias, int N, bf16_t* __restrict__ out,
                                              long ldo, long row0, long M, const bf16x8 (&hf)[C / 32][RT], char* Hs, char* Us, char* W0,
                                              int wave, int lane, int lrow, int lq) {
    using K = MlpCfg<C>;
    const int nch = N / 64;
    for (int ch = 0; ch < nch; ++ch) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* Wu = (ch & 1) ? Hs : W0;
        if (ch + 1 < nch) stage_wup<C, 8 / RT>(w, (ch + 1) * 64, (ch & 1) ? W0 : Hs, wave, lane);
        f32x4 uacc[4][RT];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) uacc[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks) {
            const int c = ks * 4 + lq;
            bf16x8 wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int wr = t * 16 + lrow;
                wf[t] = *reinterpret_cast<const bf16x8*>(Wu + wr * K::ROWB + ((c ^ swz<K::CB>(wr)) << 4));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) uacc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], hf[ks][rt], uacc[t][rt], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + ch * 64 + t * 16 + lq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const f32x4 v = uacc[t][rt];
                const bf16x4 pk = {(bf16_t)(v[0] + b4[0]), (bf16_t)(v[1] + b4[1]), (bf16_t)(v[2] + b4[2]), (bf16_t)(v[3] + b4[3])};
                const int ur = wave * (16 * RT) + rt * 16 + lrow;
                *reinterpret_cast<bf16x4*>(Us + ur * 128 + (((t * 2 + (lq >> 1)) ^ swz<8>(ur)) << 4) + (lq & 1) * 8) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2 * RT; ++it) {                          // 8 rows x 128 B per pass, 16 B per lane
            const int r = it * 8 + (lane >> 3), cidx = lane & 7;
            const int ur = wave * (16 * RT) + r;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(Us + ur * 128 + ((cidx ^ swz<8>(ur)) << 4));
            if (row0 + r < M) *reinterpret_cast<bf16x8*>(out + (row0 + r) * ldo + ch * 64 + cidx * 8) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // staging rows are rewritten by the next chunk
    }
}

// RT = 2: 4 waves x 32 rows (two waves per SIMD with two workgroups per CU).  RT = 1: 8 waves x 16 rows — half the accumulator /
// fragment registers per wave, FOUR waves per SIMD: twice as many independent MFMA -> GELU -> MFMA chains to interleave, for twice the
// LDS weight reads per row (not the limiter).  Same arithmetic per row: results are bit-identical.
template <int C, bool GATED, int RT = 2>
__global__ __launch_bounds__(512 / RT, RT == 1 ? 4 : 2) void ln_mlp_resid_kernel(const MlpArgs a) {
    constexpr int NW = 8 / RT;
    using K = MlpCfg<C>;
    extern __shared__ __attribute__((aligned(16))) char mlp_smem[];
    // [ weight set 1 = h image during the LayerNorm | U | weight set 0 ]
    char* Hs = mlp_smem;
    char* Us = Hs + K::H_BYTES;
    char* W0 = Us + K::U_BYTES;
    static_assert(K::H_BYTES >= K::WUP_BYTES + K::WDN_BYTES, "the h image must be able to hold one weight chunk");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * 128 + wave * (16 * RT);              // this wave's 32 rows
    const int lrow = lane & 15, lq = lane >> 4;

    MLP_STAMP_DECL();
    MLP_STAMP(0);
    stage_wup<C, NW>(a.w_up, 0, W0, wave, lane);                           // chunk 0 weights fly under the LayerNorm
    stage_wdn<C, NW>(a.w_dn, 0, W0 + K::WUP_BYTES, wave, lane);

    // ---- 1. LayerNorm -> operand fragments of the wave's rows (kept in VGPRs; the image's LDS becomes weight set 1).
    //         Without a gate the x values just read initialise the output accumulators (x + b_dn): x is never read again.
    f32x4 oacc[C / 16][RT];
    bf16x8 hf[C / 32][RT];
    {
        const LnSrc src{a.x, a.ldx, a.M, a.ln_w, a.ln_b, a.shift, a.scale, a.mod_sample_stride, a.rows_per_sample};
#ifdef MLP_STAMPS
        MLP_STAMP(49);
        ln_rows_to_frags<C, true, RT>(src, row0, Hs, wave, lrow, lq, oacc, hf, &_stv);
#else
        ln_rows_to_frags<C, true, RT>(src, row0, Hs, wave, lrow, lq, oacc, hf);
#endif
        MLP_STAMP(1);
#pragma unroll
        for (int n = 0; n < C / 16; ++n) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b_dn + n * 16 + lq * 4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < 4; ++j) oacc[n][rt][j] = GATED ? 0.f : oacc[n][rt][j] + b4[j];
        }
    }

    // ---- 2. hidden chunks ------------------------------------------------------------------------------------------
#ifdef MLP_DBG_NCH
    for (int ch = 0; ch < MLP_DBG_NCH; ++ch) {
#else
    for (int ch = 0; ch < K::NCH; ++ch) {
#endif
        // chunk ch's weights (issued one chunk ago) have landed for this wave; after the barrier: for every wave, and every
        // wave is past chunk ch-1 (and, at ch = 0, has its h fragments in registers), so the other set may be refilled
        MLP_STAMP(2 + 4 * ch);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        MLP_STAMP(3 + 4 * ch);
        __builtin_amdgcn_s_barrier();
        MLP_STAMP(4 + 4 * ch);
        char* Wu = (ch & 1) ? Hs : W0;
        char* Wd = Wu + K::WUP_BYTES;
        if (ch + 1 < K::NCH) {
            char* Wn = (ch & 1) ? W0 : Hs;
            stage_wup<C, NW>(a.w_up, (ch + 1) * K::HCH, Wn, wave, lane);
            stage_wdn<C, NW>(a.w_dn, (ch + 1) * K::HCH, Wn + K::WUP_BYTES, wave, lane);
        }
        f32x4 uacc[4][RT];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) uacc[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks) {
            const int c = ks * 4 + lq;
            bf16x8 wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int wr = t * 16 + lrow;
                wf[t] = *reinterpret_cast<const bf16x8*>(Wu + wr * K::ROWB + ((c ^ swz<K::CB>(wr)) << 4));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) uacc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], hf[ks][rt], uacc[t][rt], 0, 0, 0);
        }
        // bias + exact-erf GELU -> bf16 -> wave-private rows of U   (lane: hidden t*16 + lq*4 + i, row rt*16 + lrow)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b_up + ch * K::HCH + t * 16 + lq * 4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const f32x4 v = uacc[t][rt];
#ifdef MLP_DBG_NOGELU
                const f32x2 g0 = {v[0] + b4[0], v[1] + b4[1]}, g1 = {v[2] + b4[2], v[3] + b4[3]};
#else
                const f32x2 g0 = gelu_erf_fast2((f32x2){v[0] + b4[0], v[1] + b4[1]});
                const f32x2 g1 = gelu_erf_fast2((f32x2){v[2] + b4[2], v[3] + b4[3]});
#endif
                const bf16x4 pk = {(bf16_t)g0[0], (bf16_t)g0[1], (bf16_t)g1[0], (bf16_t)g1[1]};
                const int ur = wave * (16 * RT) + rt * 16 + lrow;
                *reinterpret_cast<bf16x4*>(Us + ur * 128 + (((t * 2 + (lq >> 1)) ^ swz<8>(ur)) << 4) + (lq & 1) * 8) = pk;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // U rows are wave-private: ordering inside the wave suffices
        MLP_STAMP(5 + 4 * ch);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + lq;
            bf16x8 uf[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int ur = wave * (16 * RT) + rt * 16 + lrow;
                uf[rt] = *reinterpret_cast<const bf16x8*>(Us + ur * 128 + ((c ^ swz<8>(ur)) << 4));
            }
#pragma unroll
            for (int n = 0; n < C / 16; ++n) {
                const int wr = n * 16 + lrow;
                const bf16x8 df = *reinterpret_cast<const bf16x8*>(Wd + wr * 128 + ((c ^ swz<8>(wr)) << 4));
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) oacc[n][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, uf[rt], oacc[n][rt], 0, 0, 0);
            }
        }
    }

    // ---- 3. store: lane holds channels n*16 + lq*4 .. +3 of row rt*16 + lrow ---------------------------------------
    MLP_STAMP(40);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const long grow = row0 + rt * 16 + lrow;
        if (grow >= a.M) continue;
        float* xr = a.x + grow * a.ldx;
        if (GATED) {                                                   // x + gate * (O + b_dn): x is read again (AdaLN blocks)
            const float* g = a.gate + (grow / a.rows_per_sample) * a.mod_sample_stride;
#pragma unroll
            for (int n = 0; n < C / 16; ++n) {
                const int col = n * 16 + lq * 4;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b_dn + col), g4 = *reinterpret_cast<const f32x4*>(g + col);
                f32x4 xo = *reinterpret_cast<const f32x4*>(xr + col);
#pragma unroll
                for (int j = 0; j < 4; ++j) xo[j] = xo[j] + g4[j] * (oacc[n][rt][j] + b4[j]);
                *reinterpret_cast<f32x4*>(xr + col) = xo;
                oacc[n][rt] = xo;
            }
        } else {                                                       // the accumulators started from x + b_dn
#pragma unroll
            for (int n = 0; n < C / 16; ++n) *reinterpret_cast<f32x4*>(xr + n * 16 + lq * 4) = oacc[n][rt];
        }
        if (a.x_bf16) {                                                // bf16 copy for the block that reads x as its K/V source
            bf16_t* xb = a.x_bf16 + grow * a.ldxb;
#pragma unroll
            for (int n = 0; n < C / 16; ++n) {
                bf16x4 o4;
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = (bf16_t)oacc[n][rt][j];
                *reinterpret_cast<bf16x4*>(xb + n * 16 + lq * 4) = o4;
            }
        }
    }

#ifdef MLP_STAMPS
    MLP_STAMP(41);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MLP_STAMP(42);
    MLP_STAMP_FLUSH();
#endif

    // ---- 4. (optional) the NEXT block's LayerNorm + first projection on the rows just produced (model/layers.py:218 / :225 of the
    //         block that follows): its LN(x) . W^T is computed from the accumulators, so that block never reads x for it ----------
    if (a.next.w) {
        __syncthreads();                                               // every wave is done with both weight sets and its U rows
        stage_wup<C, NW>(a.next.w, 0, W0, wave, lane);
        const LnSrc src{nullptr, 0, a.M, a.next.ln_w, a.next.ln_b, a.next.shift, a.next.scale, a.next.mod_sample_stride, a.next.rows_per_sample};
        ln_rows_to_frags<C, false, RT>(src, row0, Hs, wave, lrow, lq, oacc, hf);
        linear_chunks<C, RT>(a.next.w, a.next.bias, a.next.N, a.next.out, a.next.ldo, row0, a.M, hf, Hs, Us, W0, wave, lane, lrow, lq);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm + one linear layer:  out[M][N] (bf16) = LN(x)[affine | modulated] . W^T + b   — the attention half's input side
// (model/layers.py:218 / :225: fc_q, and fc_kv too when the block attends to its own normalised input).  Same skeleton as
// the MLP kernel: 32 rows per wave as register fragments, W streamed in chunks of 64 output channels (double-buffered
// LDS-DMA), the bf16 result chunk staged through the wave's LDS rows so that whole 128-B row pieces are stored.
template <int C>
__global__ __launch_bounds__(256, 2) void ln_linear_kernel(const LnLinArgs a) {
    using K = MlpCfg<C>;
    extern __shared__ __attribute__((aligned(16))) char mlp_smem[];
    char* Hs = mlp_smem;                                               // h image, later weight set 1
    char* Us = Hs + K::H_BYTES;                                        // output staging: 128 rows x 64 bf16
    char* W0 = Us + K::U_BYTES;                                        // weight set 0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * 128 + wave * 32;
    const int lrow = lane & 15, lq = lane >> 4;
    stage_wup<C>(a.w, 0, W0, wave, lane);
    f32x4 xv[C / 16][2];
    bf16x8 hf[C / 32][2];
    {
        const LnSrc src{a.x, a.ldx, a.M, a.ln_w, a.ln_b, a.shift, a.scale, a.mod_sample_stride, a.rows_per_sample};
        ln_rows_to_frags<C>(src, row0, Hs, wave, lrow, lq, xv, hf);
    }
    linear_chunks<C>(a.w, a.bias, a.N, a.out, a.ldo, row0, a.M, hf, Hs, Us, W0, wave, lane, lrow, lq);
}

template <int C>
int launch_ln_linear(const LnLinArgs* a, hipStream_t st) {
    LDT_ENSURE_LDS(&ln_linear_kernel<C>, MlpCfg<C>::LDS, "ln_linear");
    hipLaunchKernelGGL(ln_linear_kernel<C>, dim3((unsigned)((a->M + 127) / 128)), dim3(256), MlpCfg<C>::LDS, st, *a);
    return ldt_check_launch("ln_linear");
}

template <int C, bool GATED>
int launch_mlp(const MlpArgs* a, hipStream_t st) {
    const long blocks = (a->M + 127) / 128;
    static const int rt_env = getenv("LDT_MLP_RT") ? atoi(getenv("LDT_MLP_RT")) : 0;          // tools/dbg: 1 / 2 force
    if ((rt_env ? rt_env : LDT_MLP_RT_DEFAULT) == 1) {
        LDT_ENSURE_LDS((&ln_mlp_resid_kernel<C, GATED, 1>), MlpCfg<C>::LDS, "ln_mlp");
        hipLaunchKernelGGL((ln_mlp_resid_kernel<C, GATED, 1>), dim3((unsigned)blocks), dim3(512), MlpCfg<C>::LDS, st, *a);
        return ldt_check_launch("ln_mlp_resid");
    }
    LDT_ENSURE_LDS((&ln_mlp_resid_kernel<C, GATED, 2>), MlpCfg<C>::LDS, "ln_mlp");
    hipLaunchKernelGGL((ln_mlp_resid_kernel<C, GATED, 2>), dim3((unsigned)blocks), dim3(256), MlpCfg<C>::LDS, st, *a);
    return ldt_check_launch("ln_mlp_resid");
}

}  // namespace

int ldt_ln_mlp_launch(const MlpArgs* a, int C, hipStream_t st) {
    LDT_REQUIRE(a->M > 0 && a->M < (1L << 31) * 128, LDT_ESHAPE, "ln_mlp: bad row count %ld", a->M);
    LDT_REQUIRE(C == 64 || C == 128, LDT_ESHAPE, "ln_mlp: the fused kernel is built for 64 or 128 channels (got %d)", C);
    LDT_REQUIRE(a->ldx >= C && a->ldx % 4 == 0 && ldt_aligned16(a->x), LDT_EALIGN, "ln_mlp: x rows must be 16-byte aligned");
    LDT_REQUIRE((a->ln_w == nullptr) == (a->ln_b == nullptr), LDT_EARG, "ln_mlp: affine weight and bias go together");
    LDT_REQUIRE((a->shift == nullptr) == (a->scale == nullptr), LDT_EARG, "ln_mlp: shift and scale go together");
    LDT_REQUIRE((!a->shift && !a->gate) || (a->rows_per_sample > 0 && a->mod_sample_stride % 4 == 0), LDT_EARG,
                "ln_mlp: modulation / gate need rows_per_sample > 0 and a 16-byte aligned per-sample stride");
    LDT_REQUIRE(ldt_aligned16(a->w_up) && ldt_aligned16(a->w_dn) && ldt_aligned16(a->b_up) && ldt_aligned16(a->b_dn) &&
                (!a->ln_w || (ldt_aligned16(a->ln_w) && ldt_aligned16(a->ln_b))) &&
                (!a->shift || (ldt_aligned16(a->shift) && ldt_aligned16(a->scale))) && (!a->gate || ldt_aligned16(a->gate)),
                LDT_EALIGN, "ln_mlp: operands must be 16-byte aligned");
    LDT_REQUIRE(!a->next.w || (a->next.N > 0 && a->next.N % 64 == 0 && a->next.out && a->next.ldo >= a->next.N && a->next.ldo % 8 == 0 &&
                               ldt_aligned16(a->next.out) && ldt_aligned16(a->next.w) && (!a->next.bias || ldt_aligned16(a->next.bias)) &&
                               (a->next.ln_w == nullptr) == (a->next.ln_b == nullptr) && (a->next.shift == nullptr) == (a->next.scale == nullptr) &&
                               (!a->next.ln_w || (ldt_aligned16(a->next.ln_w) && ldt_aligned16(a->next.ln_b))) &&
                               (!a->next.shift || (a->next.rows_per_sample > 0 && a->next.mod_sample_stride % 4 == 0 &&
                                                   ldt_aligned16(a->next.shift) && ldt_aligned16(a->next.scale)))),
                LDT_EARG, "ln_mlp: the follow-on LN + linear needs N %% 64 == 0, a 16-byte aligned bf16 output and paired, aligned LN vectors");
    LDT_REQUIRE(!a->x_bf16 || (a->ldxb >= C && a->ldxb % 4 == 0 && (reinterpret_cast<uintptr_t>(a->x_bf16) & 7) == 0), LDT_EALIGN,
                "ln_mlp: bf16 mirror rows must be 8-byte aligned");
    if (a->gate) return C == 128 ? launch_mlp<128, true>(a, st) : launch_mlp<64, true>(a, st);
    return C == 128 ? launch_mlp<128, false>(a, st) : launch_mlp<64, false>(a, st);
}

int ldt_ln_linear_launch(const LnLinArgs* a, int C, hipStream_t st) {
    LDT_REQUIRE(a->M > 0 && a->M < (1L << 31) * 128, LDT_ESHAPE, "ln_linear: bad row count %ld", a->M);
    LDT_REQUIRE(C == 64 || C == 128, LDT_ESHAPE, "ln_linear: the fused kernel is built for 64 or 128 channels (got %d)", C);
    LDT_REQUIRE(a->N > 0 && a->N % 64 == 0, LDT_ESHAPE, "ln_linear: N=%d must be a multiple of 64", a->N);
    LDT_REQUIRE(a->ldx >= C && a->ldx % 4 == 0 && ldt_aligned16(a->x) && a->ldo >= a->N && a->ldo % 8 == 0 && ldt_aligned16(a->out) &&
                ldt_aligned16(a->w) && (!a->bias || ldt_aligned16(a->bias)), LDT_EALIGN, "ln_linear: rows must be 16-byte aligned");
    LDT_REQUIRE((a->ln_w == nullptr) == (a->ln_b == nullptr) && (a->shift == nullptr) == (a->scale == nullptr), LDT_EARG,
                "ln_linear: affine / modulation vectors come in pairs");
    LDT_REQUIRE(!a->shift || (a->rows_per_sample > 0 && a->mod_sample_stride % 4 == 0 && ldt_aligned16(a->shift) && ldt_aligned16(a->scale)),
                LDT_EARG, "ln_linear: modulation needs rows_per_sample > 0 and 16-byte aligned vectors");
    LDT_REQUIRE(!a->ln_w || (ldt_aligned16(a->ln_w) && ldt_aligned16(a->ln_b)), LDT_EALIGN, "ln_linear: affine vectors must be 16-byte aligned");
    return C == 128 ? launch_ln_linear<128>(a, st) : launch_ln_linear<64>(a, st);
}
