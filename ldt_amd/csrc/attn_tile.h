// attn_tile.h — the per-wave attention tile step shared by the attention kernels (attention.hip) and the fused QKV + self-attention
// kernel of the bench shape (gemm_bf16.hip): K / V tiles of 64 keys in LDS, row-major with XOR-swizzled 16-B chunks; a wave owns 32 query
// rows; S^T = K Q^T, online softmax with deferred maximum, O^T += V^T P^T (model/layers.py:183-197).
#pragma once
#include "kernels.h"

// Per-lane LDS offsets of the operand reads inside a K/V tile (the XOR swizzles depend on the lane only: 32- and 16-row
// steps leave them unchanged), so every read below is `tile base + constant + one of these`.
template <int DH>
struct AttnLaneOffs {
    int k0;                  // K row (lane & 31) of a 32-key block, 16-B chunk `half` (k-step 0); k-step s is k0 ^ (s << 5):
                             // row*ROWB and the chunk bits do not overlap, and the swizzle only permutes chunks
    int v[DH / 32];          // V^T fragments by ds_read_b64_tr_b16: row v_row_off of a 16-key step (row + 8: + 8*ROWB), 32-d tile d
    __device__ __forceinline__ void init(int lane) {
        constexpr int ROWB = DH * 2;
        const int r = lane & 31, hh = lane >> 5;
        const int swk = (DH == 64) ? ((r >> 1) & 7) : ((r >> 2) & 3);
        k0 = r * ROWB + ((hh ^ swk) << 4);
        const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
        const int v_row_off = 4 * (tg >> 1) + tq;                // key inside the 16-key k-step: 4*half + q
        const int v_chunk = (tg & 1) * 2 + (tp >> 1);            // 16-B chunk inside a 32-d tile
        const int v_byte = (tp & 1) * 8;
        const int swv = (DH == 64) ? (((v_row_off >> 1) & 1) << 2) : 0;   // unchanged by +8 rows
#pragma unroll
        for (int d = 0; d < DH / 32; ++d) v[d] = v_row_off * ROWB + (((d * 4 + v_chunk) ^ swv) << 4) + v_byte;
    }
};

// One 64-key tile for a wave (32 query rows, one per lane & 31; the keys of a 32-key block split over the two
// half-waves), processed as two 32-key blocks: S^T = K Q^T (4 or 2 MFMAs), online softmax, O^T += V^T P^T.
//   * The running reference m_run is moved (and O, l rescaled) only when some row's block maximum exceeds it by more
//     than 2^ATT_THR (guide T13 "defer-max"): P = 2^((s - m_run) c) then stays <= 2^ATT_THR, exact in fp32 sums and
//     with bf16's full relative precision, and the result O / l is the same quotient.  After the first block the
//     branch is rarely taken, which removes the per-block alpha / rescale work (16 v_pk_mul at Dh = 64).
//   * VALU-lean: packed fp32 FMA/ADD (two scores per instruction), max3 row maxima, masking code only on ragged tiles.
#define ATT_THR 6.0f
#ifndef ATT_SCALAR_SOFTMAX
#define ATT_SCALAR_SOFTMAX 0
#endif
template <int DH>
__device__ __forceinline__ void attn_block(const char* Kt, const char* Vt, const int kt, const bf16x8 (&qf)[DH / 16], f32x16 (&oacc)[DH / 32],
                                           float& m_run, float& l_run, int kv0, int Nk, int hh, float c,
                                           const AttnLaneOffs<DH>& lo) {
    constexpr int ROWB = DH * 2, ND = DH / 32, NS = DH / 16;
    {
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kt + kt * 32 * ROWB + (lo.k0 ^ (s << 5)));
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc, 0, 0, 0);
        }
        if (kv0 + kt * 32 + 32 > Nk) {                           // ragged block: mask keys >= Nk
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kv0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                sacc[i] = (key < Nk) ? sacc[i] : -INFINITY;
            }
        }
        float mx = __builtin_fmaxf(sacc[0], sacc[1]);
#pragma unroll
        for (int i = 2; i < 16; i += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, sacc[i]), sacc[i + 1]);   // -> v_max3_f32
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (__builtin_amdgcn_ballot_w64((mx - m_run) * c > ATT_THR) != 0) {   // wave-uniform; always on the first block (m_run = -inf)
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
            m_run = m_new;
            const f32x2 a2 = {alpha, alpha};
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    f32x2 o = {oacc[d][i], oacc[d][i + 1]};
                    o *= a2;                                     // v_pk_mul_f32
                    oacc[d][i] = o[0]; oacc[d][i + 1] = o[1];
                }
        }
        const f32x2 c2 = {c, c}, nmc2 = {-m_run * c, -m_run * c};
        f32x2 ps2 = {0.f, 0.f};
        bf16x8 pf[2];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 sv = {sacc[i], sacc[i + 1]};
            const f32x2 e = sv * c2 + nmc2;                      // v_pk_fma_f32
            f32x2 pv;
            pv[0] = __builtin_amdgcn_exp2f(e[0]);
            pv[1] = __builtin_amdgcn_exp2f(e[1]);
            ps2 += pv;                                           // v_pk_add_f32
            pf[i >> 3][i & 7] = (bf16_t)pv[0];
            pf[i >> 3][(i & 7) + 1] = (bf16_t)pv[1];
        }
        l_run += ps2[0] + ps2[1];
        // O^T += V^T · P^T : V^T fragments by transposed LDS reads of the row-major V tile
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const char* base = Vt + (kt * 32 + 16 * s2) * ROWB;
                const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + lo.v[d]));
                const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + 8 * ROWB + lo.v[d]));
                const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], oacc[d], 0, 0, 0);
            }
    }
}
template <int DH>
__device__ __forceinline__ void attn_tile(const char* Kt, const char* Vt, const bf16x8 (&qf)[DH / 16], f32x16 (&oacc)[DH / 32],
                                          float& m_run, float& l_run, int kv0, int Nk, int hh, float c,
                                          const AttnLaneOffs<DH>& lo) {
    attn_block<DH>(Kt, Vt, 0, qf, oacc, m_run, l_run, kv0, Nk, hh, c, lo);
    attn_block<DH>(Kt, Vt, 1, qf, oacc, m_run, l_run, kv0, Nk, hh, c, lo);
}

// The 64-key tile as ONE online-softmax step (whole-head kernel): both 32-key blocks' S^T chains are issued interleaved (two
// independent accumulators: the second chain runs in the first one's MFMA latency), one row maximum / one defer-max test per
// tile, 32 exponentials with no dependence between them, then the four 16-key slices of O^T += V^T P^T.  attn_block's
// per-wave dependency chain (LDS read -> 4 dependent MFMAs -> max -> exp -> cvt -> MFMAs, twice per tile) is what bounds the
// other two kernels at 4 waves per SIMD (tools/dbg/attn_ablate.sh: 26.7 us with the loads compiled out, 23.4 us with the
// compute compiled out, 35.9 us together); here the chain per tile is less than half as long.
// The two halves of the joint tile step, separately callable: a caller that keeps TWO score accumulator pairs can issue the S^T MFMAs of key
// tile t + 1 before the softmax of tile t (the matrix pipe then works under that wave's own softmax VALU: gemm_qkv_attn256_kernel).
template <int DH>
__device__ __forceinline__ void attn_scores(const char* Kt, const bf16x8 (&qf)[DH / 16], f32x16& s0, f32x16& s1, const AttnLaneOffs<DH>& lo) {
    constexpr int ROWB = DH * 2, NS = DH / 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(Kt + (lo.k0 ^ (s << 5)));
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(Kt + 32 * ROWB + (lo.k0 ^ (s << 5)));
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[s], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[s], s1, 0, 0, 0);
    }
}
template <int DH>
__device__ __forceinline__ void attn_softmax_pv(const char* Vt, f32x16& s0, f32x16& s1, f32x16 (&oacc)[DH / 32], float& m_run, float& l_run,
                                                int kv0, int Nk, int hh, float c, const AttnLaneOffs<DH>& lo) {
    constexpr int ROWB = DH * 2, ND = DH / 32;
    if (kv0 + 64 > Nk) {                                             // ragged tile: mask keys >= Nk
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = kv0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            s0[i] = (key < Nk) ? s0[i] : -INFINITY;
            s1[i] = (key + 32 < Nk) ? s1[i] : -INFINITY;
        }
    }
    float mx = __builtin_fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s0[i]), s1[i]);   // -> v_max3_f32
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (__builtin_amdgcn_ballot_w64((mx - m_run) * c > ATT_THR) != 0) {   // wave-uniform; always on the first tile (m_run = -inf)
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        l_run *= alpha;
        m_run = m_new;
        const f32x2 a2 = {alpha, alpha};
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 o = {oacc[d][i], oacc[d][i + 1]};
                o *= a2;                                             // v_pk_mul_f32
                oacc[d][i] = o[0]; oacc[d][i + 1] = o[1];
            }
    }
    bf16x8 pf[4];
#if ATT_SCALAR_SOFTMAX                                               /* tools/dbg A/B: scalar v_fma / v_add instead of the packed forms */
    const float nmc = -m_run * c;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        float ea0 = __builtin_fmaf(s0[i], c, nmc), ea1 = __builtin_fmaf(s0[i + 1], c, nmc);
        float eb0 = __builtin_fmaf(s1[i], c, nmc), eb1 = __builtin_fmaf(s1[i + 1], c, nmc);
        asm volatile("" : "+v"(ea0), "+v"(ea1), "+v"(eb0), "+v"(eb1));   // (keeps hipcc's SLP pass from re-packing the four)
        const float pa0 = __builtin_amdgcn_exp2f(ea0), pa1 = __builtin_amdgcn_exp2f(ea1);
        const float pb0 = __builtin_amdgcn_exp2f(eb0), pb1 = __builtin_amdgcn_exp2f(eb1);
        q0 += pa0; q1 += pa1; q2 += pb0; q3 += pb1;
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        pf[i >> 3][i & 7] = (bf16_t)pa0;       pf[i >> 3][(i & 7) + 1] = (bf16_t)pa1;
        pf[2 + (i >> 3)][i & 7] = (bf16_t)pb0; pf[2 + (i >> 3)][(i & 7) + 1] = (bf16_t)pb1;
    }
    l_run += (q0 + q1) + (q2 + q3);
#else
    const f32x2 c2 = {c, c}, nmc2 = {-m_run * c, -m_run * c};
    f32x2 psa = {0.f, 0.f}, psb = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const f32x2 ea = (f32x2){s0[i], s0[i + 1]} * c2 + nmc2;      // v_pk_fma_f32
        const f32x2 eb = (f32x2){s1[i], s1[i + 1]} * c2 + nmc2;
        f32x2 pa, pb;
        pa[0] = __builtin_amdgcn_exp2f(ea[0]); pa[1] = __builtin_amdgcn_exp2f(ea[1]);
        pb[0] = __builtin_amdgcn_exp2f(eb[0]); pb[1] = __builtin_amdgcn_exp2f(eb[1]);
        psa += pa; psb += pb;                                        // v_pk_add_f32
        pf[i >> 3][i & 7] = (bf16_t)pa[0];       pf[i >> 3][(i & 7) + 1] = (bf16_t)pa[1];
        pf[2 + (i >> 3)][i & 7] = (bf16_t)pb[0]; pf[2 + (i >> 3)][(i & 7) + 1] = (bf16_t)pb[1];
    }
    l_run += (psa[0] + psa[1]) + (psb[0] + psb[1]);
#endif
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)                                   // 16-key slices: block s2 >> 1, half s2 & 1
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const char* base = Vt + 16 * s2 * ROWB;
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + lo.v[d]));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + 8 * ROWB + lo.v[d]));
            const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], oacc[d], 0, 0, 0);
        }
}

// A head's resident K / V tiles (ntl x 64 keys in LDS) for one wave's 32 query rows, software-pipelined ACROSS tiles: the S^T MFMAs of tile
// t + 1 are issued before the softmax of tile t (two score accumulator pairs that swap roles every trip: no register copies), so the
// matrix pipe works under the wave's own softmax VALU and a wave carries two to four independent chains instead of attn_block's one —
// the form gemm_qkv_attn256_kernel runs at Dh = 64, here for the Compressor's Dh = 32 kernels (round 6: the sequential chain, not a
// unit, bounded them at 2-3 waves per SIMD).  Same per-tile math and summation order as attn_tile_joint.
template <int DH>
__device__ __forceinline__ void attn_head_pipelined(const char* Ks, const char* Vs, int ntl, const bf16x8 (&qf)[DH / 16], f32x16 (&oacc)[DH / 32],
                                                    float& m_run, float& l_run, int Nk, int hh, float c, const AttnLaneOffs<DH>& lo) {
    constexpr int TILE = 64 * DH * 2;
    f32x16 sa0, sa1, sb0, sb1;
    attn_scores<DH>(Ks, qf, sa0, sa1, lo);
    int t = 0;
    for (; t + 2 <= ntl; t += 2) {
        attn_scores<DH>(Ks + (t + 1) * TILE, qf, sb0, sb1, lo);
        __builtin_amdgcn_sched_barrier(0);
        attn_softmax_pv<DH>(Vs + t * TILE, sa0, sa1, oacc, m_run, l_run, t * 64, Nk, hh, c, lo);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntl) attn_scores<DH>(Ks + (t + 2) * TILE, qf, sa0, sa1, lo);
        __builtin_amdgcn_sched_barrier(0);
        attn_softmax_pv<DH>(Vs + (t + 1) * TILE, sb0, sb1, oacc, m_run, l_run, (t + 1) * 64, Nk, hh, c, lo);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (t < ntl) attn_softmax_pv<DH>(Vs + t * TILE, sa0, sa1, oacc, m_run, l_run, t * 64, Nk, hh, c, lo);   // odd count: the last tile's scores sit in the A pair
}

template <int DH>
__device__ __forceinline__ void attn_tile_joint(const char* Kt, const char* Vt, const bf16x8 (&qf)[DH / 16], f32x16 (&oacc)[DH / 32],
                                                float& m_run, float& l_run, int kv0, int Nk, int hh, float c,
                                                const AttnLaneOffs<DH>& lo) {
    f32x16 s0, s1;
    attn_scores<DH>(Kt, qf, s0, s1, lo);
    attn_softmax_pv<DH>(Vt, s0, s1, oacc, m_run, l_run, kv0, Nk, hh, c, lo);
}


