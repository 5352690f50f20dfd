// Fused multi-head attention forward for the LDT hot path (gfx950 / MI355X):
//   O[b,h] = softmax(Q[b,h] K[b,h]^T * Dh^-0.5) V[b,h]        (reference: model/layers.py:183-197)
// for self-attention over <=512 latent tokens (Score, Dh=64) and the Compressor's cross-attention
// (Dh=32; 2048 point queries x T token keys, or T queries x 2048 keys).  The (B,H,N,N) score tensor the
// reference materialises (layers.py:195-196) never leaves registers.
//
// Output layout is [B][H][Nq][Dh] contiguous — exactly the buffer the reference then reinterprets as
// (B,N,C) without permuting heads back (layers.py:197, quirk Q1), so the consumer GEMM just reads it
// as an [B*Nq, C] matrix.
//
// Structure: one workgroup = 4 waves = 128 query rows of one (b,h); each wave owns 32 rows.  K/V tiles
// of 64 keys are double-buffered in LDS, both ROW-MAJOR with 16-B chunk XOR swizzles (K: conflict-free
// ds_read_b128 rows; V: conflict-free ds_read_b64_tr_b16, the hardware transposed read that yields the
// V^T operand fragments directly, guide T10); the next tile is fetched into registers under the MFMAs and
// written to the other buffer before the single barrier of the iteration (T14).  QK^T is computed SWAPPED
// (S^T = K·Q^T, mfma_f32_32x32x16_bf16) so a query row lives on ONE lane: the online-softmax row
// max/sum are in-register reductions plus a single cross-half (lane^32) exchange, and the S^T accumulator
// is directly the B operand of O^T += V^T·P^T with no LDS round trip (guide §3 "accumulator tile as the
// next MFMA's operand", k order 16s + 8(j>>2) + 4h + (j&3)).
#include <stdlib.h>

#include <type_traits>

#include "kernels.h"
#include "attn_tile.h"

// Round 6, Dh = 32 (the Compressor's cross-attention, BASELINE configs[3]'s microbench): LDS tiles of 128 keys = TWO joint 64-key steps per
// barrier — both steps' S^T MFMAs are issued before the first softmax (four independent score accumulators per wave instead of
// attn_block's one sequential chain, which, not a unit, bounded this kernel), half the barriers and staging round trips per key.
#ifndef ATT_STREAM32_KT
#define ATT_STREAM32_KT 128              /* tools/dbg A/B: 64 = the round-5 form (attn_block chain) */
#endif
#ifndef ATT_ABL
#define ATT_ABL 0                        /* tools/dbg timing-only builds of attn_fwd_kernel (wrong results): bit 1 no K / V fetch behind tile 0, 2 no tile compute, 4 no output store */
#endif
template <int DH, bool OPROJ = false>
__global__ __launch_bounds__(256, (DH == 32 && ATT_STREAM32_KT == 128) ? 3 : 4) void attn_fwd_kernel(const AttnArgs a) {   // (four score accumulators: 168 VGPRs)
    constexpr int KT = (DH == 32) ? ATT_STREAM32_KT : 64;   // keys per LDS tile
    constexpr int ROWB = DH * 2;                // K / V row bytes
    constexpr int CH = ROWB / 16;               // 16-B chunks per row (8 or 4)
    constexpr int NS = DH / 16;                 // k-steps of QK^T
    constexpr int ND = DH / 32;                 // 32-wide d tiles of O^T
    constexpr int NL = KT * CH / 256;           // 16-B pieces of each of K and V per thread and tile (2 or 1)
    constexpr int TILE = KT * ROWB;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE];      // [buf][K | V]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // 1-D grid; the q-blocks of one (b,h) get ids 8 apart => same XCD (blocks are dealt round-robin over the 8
    // XCDs), so the K/V rows they share are served by one L2 instead of two HBM fetches.
    const int nqb = (a.Nq + 127) / 128;
    const int grp8 = blockIdx.x / (8 * nqb), rem = blockIdx.x % (8 * nqb);
    const int bh = grp8 * 8 + (rem & 7), qblk = rem >> 3;
    if (bh >= a.B * a.H) return;                             // whole workgroup (padding of the last group of 8)
    const int b = bh / a.H, head = bh % a.H;
    const int q0 = qblk * 128 + wave * 32;
    const bf16_t* Qb = a.Q + (long)b * a.q_batch_stride + head * DH;
    const bf16_t* Kb = a.K + (long)b * a.kv_batch_stride + head * DH;
    const bf16_t* Vb = a.V + (long)b * a.kv_batch_stride + head * DH;

    // swizzles (16-B chunk index XOR): K for the row-per-lane ds_read_b128 of S^T = K Q^T, V for ds_read_b64_tr_b16
    auto swzK = [](int row) { return (DH == 64) ? ((row >> 1) & 7) : ((row >> 2) & 3); };
    auto swzV = [](int row) { return (DH == 64) ? (((row >> 1) & 1) << 2) : 0; };

    // Q^T fragments (B operand of S^T = K·Q^T): lane (q = r, half hh) holds Q[q][16s + 8hh + j]
    bf16x8 qf[NS];
    {
        int qrow = q0 + r;
        qrow = qrow < a.Nq ? qrow : a.Nq - 1;
        const bf16_t* qp = Qb + (long)qrow * a.ldq + 8 * hh;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }

    // register staging of one K/V tile (issue early, write to LDS late: guide T14).  Thread -> (row, chunk) pieces are
    // fixed, so the global offsets and the swizzled LDS offsets are computed once; a tile's rows are then `uniform tile
    // pointer + per-lane offset`.  Rows past Nk (ragged last tile) re-read row Nk-1: their scores are masked to -inf,
    // P is exactly 0 and the (finite) V values they multiply do not matter.
    bf16x8 sreg[NL];                                         // ONE staging set: K of the next tile, then its V (half the registers)
    // piece l of a thread is 256/CH rows below piece 0 and the swizzles repeat with that period: one offset per operand
    constexpr int RPP = 256 / CH;
    const int prow = tid / CH, pch = tid % CH;
    const int gk_off = prow * (int)a.ldk + pch * 8, gv_off = prow * (int)a.ldv + pch * 8;
    const int sk_off = prow * ROWB + ((pch ^ swzK(prow)) << 4), sv_off = prow * ROWB + ((pch ^ swzV(prow)) << 4);
    auto stage_load = [&](const bf16_t* base, long ld, int goff, int kv0) {
        const bf16_t* T = base + (long)kv0 * ld;
        if (kv0 + KT <= a.Nk) {
#pragma unroll
            for (int l = 0; l < NL; ++l) sreg[l] = *reinterpret_cast<const bf16x8*>(T + (long)l * RPP * ld + goff);
        } else {
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int back = max(kv0 + l * RPP + prow - (a.Nk - 1), 0);   // rows past the end fall back to row Nk-1
                sreg[l] = *reinterpret_cast<const bf16x8*>(T + ((long)l * RPP - back) * ld + goff);
            }
        }
    };
    auto stage_store = [&](char* dst, int soff) {
#pragma unroll
        for (int l = 0; l < NL; ++l) *reinterpret_cast<bf16x8*>(dst + l * RPP * ROWB + soff) = sreg[l];
    };

    f32x16 oacc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c = a.scale_log2e;
    AttnLaneOffs<DH> lo;
    lo.init(lane);

    stage_load(Kb, a.ldk, gk_off, 0); stage_store(smem, sk_off);
    stage_load(Vb, a.ldv, gv_off, 0); stage_store(smem + TILE, sv_off);
    __syncthreads();

    // Two tiles per trip so that the LDS buffer of a tile is a compile-time constant (every LDS address below is then
    // `per-lane offset + immediate`).  The next tile's K is fetched under the first 32-key block and written to the other
    // buffer (consumed an iteration ago) before the second block, under which its V is fetched.
    const int ntiles = (a.Nk + KT - 1) / KT;
    auto do_tile = [&](int t, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        const int kv0 = t * KT;
        const bool more = t + 1 < ntiles;
        char* cur = smem + BUF * 2 * TILE;
        char* nxt = smem + (BUF ^ 1) * 2 * TILE;
        const bool fetch = more && !(ATT_ABL & 1);             // (ATT_ABL: tools/dbg timing-only builds; 0 in the product)
        if (fetch) stage_load(Kb, a.ldk, gk_off, kv0 + KT);
        if constexpr ((ATT_ABL & 2) != 0) {                    // timing-only: the tile's traffic without its compute
            if (fetch) { stage_store(nxt, sk_off); stage_load(Vb, a.ldv, gv_off, kv0 + KT); }
            asm volatile("" : "+v"(oacc[0]));
        } else if constexpr (KT == 128) {
            const bool two = kv0 + 64 < a.Nk;                // (uniform) keys 64.. of the tile exist
            f32x16 sa0, sa1, sb0, sb1;
            attn_scores<DH>(cur, qf, sa0, sa1, lo);
            if (two) attn_scores<DH>(cur + 64 * ROWB, qf, sb0, sb1, lo);
            __builtin_amdgcn_sched_barrier(0);
            attn_softmax_pv<DH>(cur + TILE, sa0, sa1, oacc, m_run, l_run, kv0, a.Nk, hh, c, lo);
            if (fetch) { stage_store(nxt, sk_off); stage_load(Vb, a.ldv, gv_off, kv0 + KT); }
            if (two) attn_softmax_pv<DH>(cur + TILE + 64 * ROWB, sb0, sb1, oacc, m_run, l_run, kv0 + 64, a.Nk, hh, c, lo);
        } else {
            attn_block<DH>(cur, cur + TILE, 0, qf, oacc, m_run, l_run, kv0, a.Nk, hh, c, lo);
            if (fetch) { stage_store(nxt, sk_off); stage_load(Vb, a.ldv, gv_off, kv0 + KT); }
            attn_block<DH>(cur, cur + TILE, 1, qf, oacc, m_run, l_run, kv0, a.Nk, hh, c, lo);
        }
        if (fetch) stage_store(nxt + TILE, sv_off);
        __syncthreads();
    };
    for (int t = 0; t < ntiles; t += 2) {
        do_tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < ntiles) do_tile(t + 1, std::integral_constant<int, 1>{});
    }

    // ---- normalise, stage the wave's 32 x DH output through LDS (the K/V buffers are idle after the last barrier)
    //      and store whole rows, 16 B per lane: the row-per-lane fragment layout would touch 32 lines per store ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (OPROJ) {
        // ---- fused output projection + gated residual (narrow blocks).  The workgroup's 128 x DH outputs are one
        //      contiguous piece of the [B][H][Nq][Dh] buffer = R = 128/H whole rows of the (B*Nq, C) matrix the reference
        //      reinterprets it as (quirk Q1): stage them in LDS as those rows (B operand), Wo fragments straight from
        //      L2 (A operand; every workgroup reads the same 2*C*C bytes), each wave takes C/4 output channels. ----
        const int H = a.H, C = H * DH, R = 128 / H;
        const int rowb = C * 2;
        auto swzO = [&](int row) { return C == 128 ? (row & 15) : ((row >> 1) & 7); };
#pragma unroll
        for (int g = 0; g < 4; ++g) {                        // DH == 32: one 32-wide d tile
            const int flat = (wave * 32 + r) * DH + 8 * g + 4 * hh;
            const int orow = flat / C, ocol = flat % C;
            const bf16x4 pk = {(bf16_t)(oacc[0][4 * g + 0] * inv), (bf16_t)(oacc[0][4 * g + 1] * inv),
                               (bf16_t)(oacc[0][4 * g + 2] * inv), (bf16_t)(oacc[0][4 * g + 3] * inv)};
            *reinterpret_cast<bf16x4*>(smem + orow * rowb + (((ocol >> 3) ^ swzO(orow)) << 4) + (ocol & 7) * 2) = pk;
        }
        __syncthreads();
        const int lrow = lane & 15, lq = lane >> 4;
        const int NT = C / 64, RT = R / 16, n0 = wave * (C / 4);         // (NT, RT) = (2, 2) at C = 128, (1, 4) at C = 64
        f32x4 pacc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) pacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < C / 32; ++ks) {
            bf16x8 of[4], wf[2];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
                if (rt < RT) {
                    const int orow = rt * 16 + lrow;
                    of[rt] = *reinterpret_cast<const bf16x8*>(smem + orow * rowb + (((ks * 4 + lq) ^ swzO(orow)) << 4));
                }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                if (nt < NT) wf[nt] = *reinterpret_cast<const bf16x8*>(a.Wo + (long)(n0 + nt * 16 + lrow) * C + ks * 32 + lq * 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    if (nt < NT && rt < RT) pacc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], of[rt], pacc[nt][rt], 0, 0, 0);
        }
        const int qb0 = qblk * 128;
        const int valid_rows = (min(128, a.Nq - qb0)) / H;               // Nq % H == 0 (checked by the launcher)
        const long xrow0 = (long)b * a.Nq + ((long)head * a.Nq + qb0) / H;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                if (nt >= NT || rt >= RT) continue;
                const int orow = rt * 16 + lrow, ch = n0 + nt * 16 + lq * 4;
                if (orow >= valid_rows) continue;
                float* xp = a.X + (xrow0 + orow) * a.ldx + ch;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bo + ch);
                f32x4 xo = *reinterpret_cast<const f32x4*>(xp);
                if (a.gate) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.gate + (long)b * a.gate_sample_stride + ch);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xo[j] += g4[j] * (pacc[nt][rt][j] + b4[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xo[j] += pacc[nt][rt][j] + b4[j];
                }
                *reinterpret_cast<f32x4*>(xp) = xo;
            }
        return;
    }
    constexpr int ORS = ROWB + 16;                           // staged row stride (pad: conflict-free 8-B column writes)
    char* ost = smem + wave * (32 * ORS);                    // 4 waves x 32 rows x (ROWB+16) <= 4*TILE
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = d * 32 + 8 * g + 4 * hh;
            const bf16x4 pk = {(bf16_t)(oacc[d][4 * g + 0] * inv), (bf16_t)(oacc[d][4 * g + 1] * inv),
                               (bf16_t)(oacc[d][4 * g + 2] * inv), (bf16_t)(oacc[d][4 * g + 3] * inv)};
            *reinterpret_cast<bf16x4*>(ost + r * ORS + col * 2) = pk;
        }
    bf16_t* ob = a.O + (((long)b * a.H + head) * a.Nq + q0) * DH;
    constexpr int LPR = ROWB / 16;                           // lanes per output row (8 or 4)
#pragma unroll
    for (int it = 0; it < (32 * LPR) / 64; ++it) {
        const int row = it * (64 / LPR) + lane / LPR, ch = lane % LPR;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ost + row * ORS + ch * 16);
#if ATT_ABL & 4
        if (q0 + row < a.Nq && v[0] == (bf16_t)12345.0f) *reinterpret_cast<bf16x8*>(ob + (long)row * DH + ch * 8) = v;
#else
        if (q0 + row < a.Nq) *reinterpret_cast<bf16x8*>(ob + (long)row * DH + ch * 8) = v;
#endif
    }
}

// -------------------------------------------------------------------------------------------------
// Resident variant for the self-attention sizes of the path (Nk <= 256 at Dh=64, <= 512 at Dh=32): one
// workgroup per (b,h) loads ALL keys/values of the head into LDS once (64 KiB), then loops over the 128-row
// query blocks with no further workgroup barrier — K/V are fetched once per head instead of once per query
// block, and the per-tile barrier/restage of the streaming kernel disappears.  Same math, same layouts.
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_fwd_resident_kernel(const AttnArgs a, int ntl, int qsplit) {
    constexpr int KT = 64;
    constexpr int ROWB = DH * 2;
    constexpr int CH = ROWB / 16;
    constexpr int NS = DH / 16;
    constexpr int ND = DH / 32;
    constexpr int TILE = KT * ROWB;
    extern __shared__ __attribute__((aligned(16))) char rsmem[];       // [K: ntl tiles][V: ntl tiles][O staging: 4 x 32 rows]
    char* Ks = rsmem;
    char* Vs = rsmem + ntl * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.x / qsplit, part = blockIdx.x % qsplit;
    const int b = bh / a.H, head = bh % a.H;
    char* ost = rsmem + 2 * ntl * TILE + wave * (32 * ROWB);
    const bf16_t* Qb = a.Q + (long)b * a.q_batch_stride + head * DH;
    const bf16_t* Kb = a.K + (long)b * a.kv_batch_stride + head * DH;
    const bf16_t* Vb = a.V + (long)b * a.kv_batch_stride + head * DH;
    auto swzK = [](int row) { return (DH == 64) ? ((row >> 1) & 7) : ((row >> 2) & 3); };
    auto swzV = [](int row) { return (DH == 64) ? (((row >> 1) & 1) << 2) : 0; };

    // ---- load every key / value row of this head: thread -> (row = tid/CH + l*(256/CH), chunk = tid%CH); the
    //      source pointers and LDS offsets advance by constants (the swizzles repeat every 256/CH rows) ----
    {
        constexpr int RPL = 256 / CH;                                   // rows covered per pass (32 or 64)
        const int row0 = tid / CH, ch = tid % CH;
        const int kofs = row0 * ROWB + ((ch ^ swzK(row0)) << 4), vofs = row0 * ROWB + ((ch ^ swzV(row0)) << 4);
        const int npass = ntl * KT / RPL;
        for (int base = 0; base < npass; base += 4) {
            bf16x8 kreg[4], vreg[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int row = row0 + (base + l) * RPL;
                const int krow = row < a.Nk ? row : a.Nk - 1;
                kreg[l] = *reinterpret_cast<const bf16x8*>(Kb + (long)krow * a.ldk + ch * 8);
                vreg[l] = *reinterpret_cast<const bf16x8*>(Vb + (long)krow * a.ldv + ch * 8);   // rows past Nk: P is exactly 0 there
            }
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                if (base + l < npass) {
                    *reinterpret_cast<bf16x8*>(Ks + kofs + (base + l) * RPL * ROWB) = kreg[l];
                    *reinterpret_cast<bf16x8*>(Vs + vofs + (base + l) * RPL * ROWB) = vreg[l];
                }
            }
        }
    }
    __syncthreads();

    const float c = a.scale_log2e;
    AttnLaneOffs<DH> lo;
    lo.init(lane);
    const int nqb = (a.Nq + 127) / 128;
    // `qsplit` workgroups share a head's query blocks (each loads the head's K / V again: 32-64 KB from L2): with few heads in the launch
    // (128 clouds x 4 heads = 512) one workgroup per head left two waves per SIMD, each serialised on its own MFMA -> softmax -> MFMA chain
    // over 16 query blocks (round 5; prefetching the next block's Q instead measured 5 % SLOWER: profiles/r05_attention_resident_ab.txt)
    const int qb_lo = (int)((long)nqb * part / qsplit), qb_hi = (int)((long)nqb * (part + 1) / qsplit);
    for (int qb = qb_lo; qb < qb_hi; ++qb) {
        const int q0 = qb * 128 + wave * 32;
        if (q0 >= a.Nq) continue;                                       // wave-uniform; no barrier below
        bf16x8 qf[NS];
        {
            int qrow = q0 + r;
            qrow = qrow < a.Nq ? qrow : a.Nq - 1;
            const bf16_t* qp = Qb + (long)qrow * a.ldq + 8 * hh;
#pragma unroll
            for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
        }
        f32x16 oacc[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        if (DH == 32) attn_head_pipelined<DH>(Ks, Vs, ntl, qf, oacc, m_run, l_run, a.Nk, hh, c, lo);   // (round 6: two to four chains per wave)
        else
            for (int t = 0; t < ntl; ++t)
                attn_tile<DH>(Ks + t * TILE, Vs + t * TILE, qf, oacc, m_run, l_run, t * KT, a.Nk, hh, c, lo);
        // ---- normalise, stage through the wave's private LDS rows (XOR-swizzled chunks), store whole rows ----
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int chn = d * 4 + g;                                // 16-B chunk; hh picks its 8-B half
                const bf16x4 pk = {(bf16_t)(oacc[d][4 * g + 0] * inv), (bf16_t)(oacc[d][4 * g + 1] * inv),
                                   (bf16_t)(oacc[d][4 * g + 2] * inv), (bf16_t)(oacc[d][4 * g + 3] * inv)};
                *reinterpret_cast<bf16x4*>(ost + r * ROWB + ((chn ^ (r & (CH - 1))) << 4) + hh * 8) = pk;
            }
        bf16_t* ob = a.O + (((long)b * a.H + head) * a.Nq + q0) * DH;
#pragma unroll
        for (int it = 0; it < (32 * CH) / 64; ++it) {
            const int row = it * (64 / CH) + lane / CH, ch = lane % CH;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(ost + row * ROWB + ((ch ^ (row & (CH - 1))) << 4));
            if (q0 + row < a.Nq) *reinterpret_cast<bf16x8*>(ob + (long)row * DH + ch * 8) = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Whole-head form for the Score's self-attention at 129..256 tokens (Dh = 64: T = 256, the headline shape): one workgroup of EIGHT
// waves per (b,h), every query row of the head in flight at once (wave w owns rows 32 w ..).  All of the head's K and V rows go to
// LDS by LDS-DMA (1-KiB pieces, swizzle applied on the source address) behind the wave's Q fragments, requested tile by tile, and the
// workgroup starts on key tile t as soon as ITS pieces are in (counted vmcnt + one barrier per tile, round 4): one memory round trip
// per workgroup of which only the first tile's share is exposed; the output rows are staged through the (then free) K region.
// The streaming kernel pays a round trip per 64-key tile behind a barrier (its tile
// compute, ~0.8 us, is shorter than the load it is supposed to hide), the 4-wave resident kernel two register-staged load rounds
// plus a Q round trip per query block.  64 KiB of LDS and <= 128 VGPRs: two workgroups (16 waves) per CU.  Same math, same layouts,
// the tile is one online-softmax step (attn_tile_joint): results agree with the other two kernels to rounding, not bit for bit.
template <int DH, int NTL>   // NTL = 64-key tiles of the head (compile-time: every counted wait below is then straight-line code)
__global__ __launch_bounds__(512, 4) void attn_fwd_head_kernel(const AttnArgs a) {
    constexpr int ntl = NTL;
    constexpr int KT = 64;
    constexpr int ROWB = DH * 2;
    constexpr int CH = ROWB / 16;
    constexpr int NS = DH / 16;
    constexpr int ND = DH / 32;
    constexpr int TILE = KT * ROWB;
    constexpr int RPP = 64 / CH;                                        // rows per 1-KiB DMA piece
    extern __shared__ __attribute__((aligned(16))) char rsmem[];       // [K: ntl tiles][V: ntl tiles]; later the 8 waves' output rows
    char* Ks = rsmem;
    char* Vs = rsmem + ntl * TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.x;
    const int b = bh / a.H, head = bh % a.H;
    const bf16_t* Qb = a.Q + (long)b * a.q_batch_stride + head * DH;
    const bf16_t* Kb = a.K + (long)b * a.kv_batch_stride + head * DH;
    const bf16_t* Vb = a.V + (long)b * a.kv_batch_stride + head * DH;
    const int q0 = wave * 32;
    const bool active = q0 < a.Nq;                                      // wave-uniform
    // ---- tile-ordered landing: a wave's requests are, oldest first, its Q fragments (asm loads: hipcc drains the whole queue around a
    //      register load it knows of when LDS-DMA is in flight beside it) and then (K, V) piece t * 8 + wave of tile t = 0, 1, ..  VMEM returns
    //      in order, so "tile t is here" = at most 2 (ntl - 1 - t) younger requests outstanding + one barrier: tile t's 16 MFMAs + softmax
    //      run while tiles t + 1.. are still on their way (one round trip per workgroup, but only the first tile's share of it exposed).
    static_assert(DH == 64 && NTL >= 1 && NTL <= 4, "attn_fwd_head_kernel: built for head dim 64 (8 pieces per 64-key tile = one per wave), <= 256 keys");
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 qi[NS];
    {
        int qrow = q0 + r;
        qrow = qrow < a.Nq ? qrow : a.Nq - 1;
        const bf16_t* qp = Qb + (long)qrow * a.ldq + 8 * hh;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#ifdef ATT_DBG_NOLOAD
            qi[s] = (i32x4){lane, 1, 1, 1};
            (void)qp;
#else
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qi[s]) : "v"(qp + 16 * s) : "memory");
#endif
        }
    }
    {
        const int lr = lane / CH, cd = lane % CH;
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Ks;
#ifdef ATT_DBG_NOLOAD
        for (int t = 0; t < 0; ++t) {
#else
        for (int t = 0; t < ntl; ++t) {
#endif
            const int p = t * 8 + wave;
            const int row = p * RPP + lr;
            const int krow = row < a.Nk ? row : a.Nk - 1;               // rows past Nk: P is exactly 0 there
            const int sk = (row >> 1) & 7;
            const int sv = ((row >> 1) & 1) << 2;
            const bf16_t* ks = Kb + (long)krow * a.ldk + ((cd ^ sk) << 3);
            const bf16_t* vs = Vb + (long)krow * a.ldv + ((cd ^ sv) << 3);
            // (asm, not the builtin: with LDS-DMA it knows of in flight hipcc puts vmcnt(0) in front of the first ds_read_b64_tr of V)
            const unsigned kd = lds_base + p * 1024, vd = kd + ntl * TILE;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(ks), "s"(kd) : "memory");   // (m0: hipcc keeps nothing there in this kernel — no builtin DMA, no movrel)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vs), "s"(vd) : "memory");
        }
    }
    // the Q registers become visible to the compiler only through this wait (tile 0's: 2 younger requests per later tile may be out)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(qi[0]), "+v"(qi[1]), "+v"(qi[2]), "+v"(qi[3]) : "n"(2 * (NTL - 1)) : "memory");
    f32x16 oacc[ND];
    float m_run = -INFINITY, l_run = 0.f;
    const float c = a.scale_log2e;
    AttnLaneOffs<DH> lo;
    lo.init(lane);
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
    bf16x8 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = __builtin_bit_cast(bf16x8, qi[s]);
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        if (t > 0) {
            if (t == NTL - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (t == NTL - 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                                    // raw: __syncthreads() would drain vmcnt to 0 (the later tiles' pieces)
        __builtin_amdgcn_sched_barrier(0);
#ifdef ATT_DBG_NOCOMPUTE
        l_run += (float)qf[t & (NS - 1)][0];
#else
        if (active) attn_tile_joint<DH>(Ks + t * TILE, Vs + t * TILE, qf, oacc, m_run, l_run, t * KT, a.Nk, hh, c, lo);
#endif
    }
    __syncthreads();                                                    // every wave is done with K and V: their LDS becomes the output stage
    if (!active) return;
    char* ost = rsmem + wave * (32 * ROWB);
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chn = d * 4 + g;                                  // 16-B chunk; hh picks its 8-B half
            const bf16x4 pk = {(bf16_t)(oacc[d][4 * g + 0] * inv), (bf16_t)(oacc[d][4 * g + 1] * inv),
                               (bf16_t)(oacc[d][4 * g + 2] * inv), (bf16_t)(oacc[d][4 * g + 3] * inv)};
            *reinterpret_cast<bf16x4*>(ost + r * ROWB + ((chn ^ (r & (CH - 1))) << 4) + hh * 8) = pk;
        }
    bf16_t* ob = a.O + (((long)b * a.H + head) * a.Nq + q0) * DH;
#pragma unroll
    for (int it = 0; it < (32 * CH) / 64; ++it) {
        const int row = it * (64 / CH) + lane / CH, ch = lane % CH;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ost + row * ROWB + ((ch ^ (row & (CH - 1))) << 4));
        if (q0 + row < a.Nq) *reinterpret_cast<bf16x8*>(ob + (long)row * DH + ch * 8) = v;
    }
}

// -------------------------------------------------------------------------------------------------
// Resident form of the fused attention + output projection + residual (narrow blocks, Dh = 32) for cross-attention from many
// queries to few keys — the Compressor's decoder levels: 2048 point queries x T <= 512 token keys per cloud.  The streaming kernel
// above gives every 128-query block a workgroup of its own, and each of them re-reads the head's K/V (32 KB) and all of Wo (32 KB)
// from L2: 64 KB of operands for 5 MFLOP, 4.2 GB of L2 traffic per launch at 1024 clouds next to 2.5 GB of HBM data.  Here one
// workgroup owns a (cloud, head): K/V go to LDS once, each wave keeps its C/4 output channels of Wo as register fragments, and
// the workgroup walks the head's query blocks (no barrier inside the attention; two per block around the 8 KB output stage).
// ATT_OPROJ_FORM (tools/dbg A/B): 0 = the sequential 32-key chain of rounds 3-5 (3 waves per SIMD), 1 = the joint 64-key step (3 waves),
// 2 = joint + pipelined across tiles (attn_head_pipelined: four score accumulators, 2 waves per SIMD)
#ifndef ATT_OPROJ_FORM
#define ATT_OPROJ_FORM 2
#endif
template <int H>   // heads: C = 32 H channels (2 or 4)
__global__ __launch_bounds__(256, ATT_OPROJ_FORM == 2 ? 2 : 3) void attn_oproj_resident_kernel(const AttnArgs a, int ntl) {   // (the prefetched Q / residual registers do not fit 128 VGPRs)
    constexpr int DH = 32, KT = 64, ROWB = DH * 2, CH = ROWB / 16, NS = DH / 16, TILE = KT * ROWB;
    constexpr int C = H * DH, R = 128 / H, NT = C / 64, RT = R / 16, rowb = C * 2;
    extern __shared__ __attribute__((aligned(16))) char rsmem[];       // [K: ntl tiles][V: ntl tiles][O stage: R rows x C bf16 = 8 KB]
    char* Ks = rsmem;
    char* Vs = rsmem + ntl * TILE;
    char* Ost = rsmem + 2 * ntl * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.x;
    const int b = bh / H, head = bh % H;
    const bf16_t* Qb = a.Q + (long)b * a.q_batch_stride + head * DH;
    const bf16_t* Kb = a.K + (long)b * a.kv_batch_stride + head * DH;
    const bf16_t* Vb = a.V + (long)b * a.kv_batch_stride + head * DH;
    auto swzK = [](int row) { return (row >> 2) & 3; };
    auto swzO = [](int row) { return C == 128 ? (row & 15) : ((row >> 1) & 7); };
    {   // every key / value row of the head (rows past Nk repeat row Nk-1: their P is exactly 0)
        constexpr int RPL = 256 / CH;
        const int row0 = tid / CH, ch = tid % CH;
        const int kofs = row0 * ROWB + ((ch ^ swzK(row0)) << 4), vofs = row0 * ROWB + (ch << 4);
        const int npass = ntl * KT / RPL;
        for (int base = 0; base < npass; base += 4) {
            bf16x8 kreg[4], vreg[4];
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int row = row0 + (base + l) * RPL;
                const int krow = row < a.Nk ? row : a.Nk - 1;
                kreg[l] = *reinterpret_cast<const bf16x8*>(Kb + (long)krow * a.ldk + ch * 8);
                vreg[l] = *reinterpret_cast<const bf16x8*>(Vb + (long)krow * a.ldv + ch * 8);
            }
#pragma unroll
            for (int l = 0; l < 4; ++l)
                if (base + l < npass) {
                    *reinterpret_cast<bf16x8*>(Ks + kofs + (base + l) * RPL * ROWB) = kreg[l];
                    *reinterpret_cast<bf16x8*>(Vs + vofs + (base + l) * RPL * ROWB) = vreg[l];
                }
        }
    }
    // this wave's output channels of Wo (A operand of out^T = Wo . O'^T), kept for every query block
    const int lrow = lane & 15, lq = lane >> 4, n0 = wave * (C / 4);
    bf16x8 wf[NT][C / 32];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks)
            wf[nt][ks] = *reinterpret_cast<const bf16x8*>(a.Wo + (long)(n0 + nt * 16 + lrow) * C + ks * 32 + lq * 8);
    __syncthreads();

    const float c = a.scale_log2e;
    AttnLaneOffs<DH> lo;
    lo.init(lane);
    const int nqb = (a.Nq + 127) / 128;
    // Round 5: nothing a query block needs from global memory is requested at the point of use any more.  The Q fragments of block
    // qb + 1 and the fp32 residual rows of block qb (read-modify-written by the epilogue) are requested at the head of block qb and land
    // under its attention; the projection bias is loaded once.  Before, each of the up to 16 blocks of a workgroup opened on a
    // dependent Q round trip and closed on a dependent residual round trip (~1.5 us each beside ~3 us of work).
    auto load_q = [&](int qb, bf16x8 (&q)[NS]) {
        int qrow = qb * 128 + wave * 32 + r;
        qrow = qrow < a.Nq ? qrow : a.Nq - 1;                               // rows past the end: computed, never stored
        const bf16_t* qp = Qb + (long)qrow * a.ldq + 8 * hh;
#pragma unroll
        for (int s = 0; s < NS; ++s) q[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    };
    f32x4 b4[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b4[nt] = *reinterpret_cast<const f32x4*>(a.bo + n0 + nt * 16 + lq * 4);
    bf16x8 qf[NS], qn[NS];
    load_q(0, qf);
    for (int qb = 0; qb < nqb; ++qb) {
        const int qb0 = qb * 128;
        const int valid_rows = (min(128, a.Nq - qb0)) / H;                // Nq % H == 0 (checked by the launcher)
        const long xrow0 = (long)b * a.Nq + ((long)head * a.Nq + qb0) / H;
        f32x4 xpre[NT][RT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int orow = rt * 16 + lrow, orc = orow < valid_rows ? orow : valid_rows - 1;   // clamped, never branched; rows past the end are not stored
                xpre[nt][rt] = *reinterpret_cast<const f32x4*>(a.X + (xrow0 + orc) * a.ldx + n0 + nt * 16 + lq * 4);
            }
        load_q(qb + 1 < nqb ? qb + 1 : qb, qn);
        f32x16 oacc[1];
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[0][i] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
#if ATT_OPROJ_FORM == 2
        attn_head_pipelined<DH>(Ks, Vs, ntl, qf, oacc, m_run, l_run, a.Nk, hh, c, lo);
#elif ATT_OPROJ_FORM == 1
        for (int t = 0; t < ntl; ++t)
            attn_tile_joint<DH>(Ks + t * TILE, Vs + t * TILE, qf, oacc, m_run, l_run, t * KT, a.Nk, hh, c, lo);
#else
        for (int t = 0; t < ntl; ++t)
            attn_tile<DH>(Ks + t * TILE, Vs + t * TILE, qf, oacc, m_run, l_run, t * KT, a.Nk, hh, c, lo);
#endif
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        // the workgroup's 128 x Dh outputs = R whole rows of the (B*Nq, C) matrix the reference reinterprets the head-major buffer
        // as (quirk Q1): staged as those rows (B operand of the projection)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int flat = (wave * 32 + r) * DH + 8 * g + 4 * hh;
            const int orow = flat / C, ocol = flat % C;
            const bf16x4 pk = {(bf16_t)(oacc[0][4 * g + 0] * inv), (bf16_t)(oacc[0][4 * g + 1] * inv),
                               (bf16_t)(oacc[0][4 * g + 2] * inv), (bf16_t)(oacc[0][4 * g + 3] * inv)};
            *reinterpret_cast<bf16x4*>(Ost + orow * rowb + (((ocol >> 3) ^ swzO(orow)) << 4) + (ocol & 7) * 2) = pk;
        }
        __syncthreads();
        f32x4 pacc[NT][RT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int jj = 0; jj < RT; ++jj) pacc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C / 32; ++ks) {
            bf16x8 of[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int orow = rt * 16 + lrow;
                of[rt] = *reinterpret_cast<const bf16x8*>(Ost + orow * rowb + (((ks * 4 + lq) ^ swzO(orow)) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) pacc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], of[rt], pacc[nt][rt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int orow = rt * 16 + lrow, ch = n0 + nt * 16 + lq * 4;
                if (orow >= valid_rows) continue;
                float* xp = a.X + (xrow0 + orow) * a.ldx + ch;
                f32x4 xo = xpre[nt][rt];
                if (a.gate) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.gate + (long)b * a.gate_sample_stride + ch);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) xo[jj] += g4[jj] * (pacc[nt][rt][jj] + b4[nt][jj]);
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) xo[jj] += pacc[nt][rt][jj] + b4[nt][jj];
                }
                *reinterpret_cast<f32x4*>(xp) = xo;
            }
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = qn[s];
        __syncthreads();                                                    // the stage is rewritten by the next query block
    }
}

template <int DH>
static int launch_resident(const AttnArgs* a, hipStream_t s) {
    const int ntl = (a->Nk + 63) / 64;
    const size_t lds = (size_t)2 * ntl * 64 * DH * 2 + 4 * 32 * DH * 2;
    LDT_ENSURE_LDS(&attn_fwd_resident_kernel<DH>, 81920, "attention");
    // query blocks of a head over `qsplit` workgroups until the launch has ~4 workgroups per CU (or one block each)
    static const int qs_env = getenv("LDT_ATTN_QSPLIT") ? atoi(getenv("LDT_ATTN_QSPLIT")) : 0;      // tools/dbg
    const long heads = (long)a->B * a->H, nqb = (a->Nq + 127) / 128;
    long qsplit = qs_env > 0 ? qs_env : (4L * LDT_NUM_CUS + heads - 1) / heads;
    qsplit = qsplit < 1 ? 1 : (qsplit > nqb ? nqb : qsplit);
    LDT_REQUIRE(heads * qsplit < (1L << 31), LDT_ESHAPE, "attention: grid too large");
    hipLaunchKernelGGL(attn_fwd_resident_kernel<DH>, dim3((unsigned)(heads * qsplit)), dim3(256), lds, s, *a, ntl, (int)qsplit);
    return ldt_check_launch("attn_fwd_resident");
}

template <int DH, int NTL>
static int launch_head_t(const AttnArgs* a, hipStream_t s) {
    size_t lds = (size_t)2 * NTL * 64 * DH * 2;
    if (lds < (size_t)8 * 32 * DH * 2) lds = (size_t)8 * 32 * DH * 2;
    LDT_ENSURE_LDS((&attn_fwd_head_kernel<DH, NTL>), 65536, "attention");
    hipLaunchKernelGGL((attn_fwd_head_kernel<DH, NTL>), dim3((unsigned)(a->B * a->H)), dim3(512), lds, s, *a);
    return ldt_check_launch("attn_fwd_head");
}
template <int DH>
static int launch_head(const AttnArgs* a, hipStream_t s) {
    switch ((a->Nk + 63) / 64) {
        case 1: return launch_head_t<DH, 1>(a, s);
        case 2: return launch_head_t<DH, 2>(a, s);
        case 3: return launch_head_t<DH, 3>(a, s);
        default: return launch_head_t<DH, 4>(a, s);                      // (caller: Nk <= 256)
    }
}

int ldt_attn_oproj_launch(const AttnArgs* a, int dh, hipStream_t s) {
    LDT_REQUIRE(a->B > 0 && a->H > 0 && a->Nq > 0 && a->Nk > 0, LDT_ESHAPE, "attention_oproj: empty problem");
    LDT_REQUIRE(dh == 32 && (a->H == 2 || a->H == 4), LDT_ESHAPE,
                "attention_oproj: the fused kernel is built for head dim 32 with 2 or 4 heads (C = 64 / 128), got Dh=%d H=%d", dh, a->H);
    LDT_REQUIRE(a->Nq % a->H == 0, LDT_ESHAPE, "attention_oproj: Nq=%d must be a multiple of H=%d (rows of the reinterpreted output)", a->Nq, a->H);
    LDT_REQUIRE(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->q_batch_stride % 8 == 0 && a->kv_batch_stride % 8 == 0 &&
                ldt_aligned16(a->Q) && ldt_aligned16(a->K) && ldt_aligned16(a->V) && ldt_aligned16(a->Wo) && ldt_aligned16(a->bo) &&
                ldt_aligned16(a->X) && a->ldx % 4 == 0 && a->ldx >= a->H * dh && (!a->gate || (ldt_aligned16(a->gate) && a->gate_sample_stride % 4 == 0)),
                LDT_EALIGN, "attention_oproj: operands must be 16-byte aligned");
    const long nqb = (a->Nq + 127) / 128, groups = ((long)a->B * a->H + 7) / 8;
    LDT_REQUIRE(groups * 8 * nqb < (1L << 31), LDT_ESHAPE, "attention_oproj: grid too large");
    // several query blocks per head, few keys, a (cloud, head) pair per CU or more: K/V + Wo resident, one workgroup per pair
    // (config C4, 1024 clouds: decode 16.2 -> 15.3 ms at 128 clouds per call, 16.0 -> 14.5 at 512; encode 34.8 -> 32.9 ms)
    static const int res_env = getenv("LDT_ATTN_OPROJ_RESIDENT") ? atoi(getenv("LDT_ATTN_OPROJ_RESIDENT")) : -1;   // 0 / 1 force (tools/dbg)
    const bool fits = a->Nk <= 512;
    if (fits && (res_env == 1 || (res_env != 0 && nqb >= 2 && (long)a->B * a->H >= 256))) {
        const int ntl = (a->Nk + 63) / 64;
        const size_t lds = (size_t)2 * ntl * 64 * 64 + 8192;
        if (a->H == 4) {
            LDT_ENSURE_LDS(&attn_oproj_resident_kernel<4>, 81920, "attention_oproj");
            hipLaunchKernelGGL(attn_oproj_resident_kernel<4>, dim3((unsigned)(a->B * a->H)), dim3(256), lds, s, *a, ntl);
        } else {
            LDT_ENSURE_LDS(&attn_oproj_resident_kernel<2>, 81920, "attention_oproj");
            hipLaunchKernelGGL(attn_oproj_resident_kernel<2>, dim3((unsigned)(a->B * a->H)), dim3(256), lds, s, *a, ntl);
        }
        return ldt_check_launch("attn_oproj_resident");
    }
    hipLaunchKernelGGL((attn_fwd_kernel<32, true>), dim3((unsigned)(groups * 8 * nqb)), dim3(256), 0, s, *a);
    return ldt_check_launch("attn_oproj");
}

// Which kernel ldt_attn_launch takes for a problem: 0 = streaming (attn_fwd_kernel), 1 = resident (attn_fwd_resident_kernel), 2 = whole-head
// (attn_fwd_head_kernel<64, ceil(Nk / 64)>).  Also exported (ldt_attention_route) so that bench.py names the symbol it timed instead of guessing.
int ldt_attn_route(int B, int H, int Nq, int Nk, int dh) {
    // Short sequences (one 128-row query block, keys/values of a head fit 64 KiB of LDS): resident kernel — K/V
    // loaded once, no per-tile barrier (measured 9.0 vs 9.7 us at T=32).  Longer query sets run the streaming
    // kernel, which spreads (b,h,q-block) over more workgroups (35 vs 37 us at T=256, 46 vs 56 us at 2048x256).
    static const int force = getenv("LDT_ATTN_FORCE") ? atoi(getenv("LDT_ATTN_FORCE")) : 0;   // 1 stream, 2 resident, 3 whole-head (tools/dbg)
    const bool fits = (long)Nk * dh <= 256 * 64;
    if (fits && (force == 2 || (force == 0 && Nq <= 128))) return 1;
    // 129..256 queries of a Dh = 64 head (the Score at T = 256): the whole-head 8-wave kernel
    if (fits && dh == 64 && Nq <= 256 && (long)B * H < (1L << 31) && (force == 3 || (force == 0 && Nq > 128))) return 2;
    return 0;
}

int ldt_attn_launch(const AttnArgs* a, int dh, hipStream_t s) {
    LDT_REQUIRE(a->B > 0 && a->H > 0 && a->Nq > 0 && a->Nk > 0, LDT_ESHAPE, "attention: empty problem B=%d H=%d Nq=%d Nk=%d", a->B, a->H, a->Nq, a->Nk);
    LDT_REQUIRE(dh == 32 || dh == 64, LDT_ESHAPE, "attention: head dim %d not built (32, 64)", dh);
    LDT_REQUIRE(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->q_batch_stride % 8 == 0 && a->kv_batch_stride % 8 == 0 &&
                ldt_aligned16(a->Q) && ldt_aligned16(a->K) && ldt_aligned16(a->V) && ldt_aligned16(a->O), LDT_EALIGN,
                "attention: Q/K/V rows must be 16-byte aligned");
    LDT_REQUIRE(a->H <= 65535 && a->B <= 65535, LDT_ESHAPE, "attention: grid too large");
    const int route = ldt_attn_route(a->B, a->H, a->Nq, a->Nk, dh);
    if (route == 1) return dh == 64 ? launch_resident<64>(a, s) : launch_resident<32>(a, s);
    if (route == 2) return launch_head<64>(a, s);
    const long nqb = (a->Nq + 127) / 128, groups = ((long)a->B * a->H + 7) / 8;
    LDT_REQUIRE(groups * 8 * nqb < (1L << 31), LDT_ESHAPE, "attention: grid too large");
    dim3 grid((unsigned)(groups * 8 * nqb)), block(256);
    if (dh == 64) hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, block, 0, s, *a);
    else hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, block, 0, s, *a);
    return ldt_check_launch("attn_fwd");
}
