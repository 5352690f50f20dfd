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
// of 64 keys are staged in LDS (K row-major, XOR-swizzled for ds_read_b128; V transposed [d][key] with a
// +4 pad so the PV operand reads are conflict-free ds_read_b64).  QK^T is computed SWAPPED
// (S^T = K·Q^T, mfma_f32_32x32x16_bf16) so a query row lives on ONE lane: the online-softmax row
// max/sum are in-register reductions plus a single cross-half (lane^32) exchange, and the S^T accumulator
// is directly the B operand of O^T += V^T·P^T with no LDS round trip (guide §3 "accumulator tile as the
// next MFMA's operand", k order 16s + 8(j>>2) + 4h + (j&3)).
#include "kernels.h"

template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
    constexpr int KT = 64;                      // keys per LDS tile
    constexpr int ROWB = DH * 2;                // K row bytes
    constexpr int CH = ROWB / 16;               // 16-B chunks per K row (8 or 4)
    constexpr int VT_LD = KT + 4;               // V^T row stride in elements (136 B)
    constexpr int NS = DH / 16;                 // k-steps of QK^T
    constexpr int ND = DH / 32;                 // 32-wide d tiles of O^T
    __shared__ __attribute__((aligned(16))) char smem[KT * ROWB + DH * VT_LD * 2];
    char* Ks = smem;
    bf16_t* Vt = reinterpret_cast<bf16_t*>(smem + KT * ROWB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bf16_t* Qb = a.Q + (long)b * a.q_batch_stride + head * DH;
    const bf16_t* Kb = a.K + (long)b * a.kv_batch_stride + head * DH;
    const bf16_t* Vb = a.V + (long)b * a.kv_batch_stride + head * DH;

    // Q^T fragments (B operand of S^T = K·Q^T): lane (q = r, half hh) holds Q[q][16s + 8hh + j]
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
    const float c = a.scale_log2e;

    for (int kv0 = 0; kv0 < a.Nk; kv0 += KT) {
        __syncthreads();                         // previous tile fully consumed
        // ---- stage K (swizzled rows) and V^T ----
        for (int idx = tid; idx < KT * CH; idx += 256) {
            const int row = idx / CH, ch = idx % CH;
            int krow = kv0 + row;
            const bool valid = krow < a.Nk;
            krow = valid ? krow : a.Nk - 1;
            const bf16x8 kv = *reinterpret_cast<const bf16x8*>(Kb + (long)krow * a.ldk + ch * 8);
            const int swz = (DH == 64) ? ((row >> 1) & 7) : ((row >> 2) & 3);
            *reinterpret_cast<bf16x8*>(Ks + row * ROWB + ((ch ^ swz) << 4)) = kv;
            bf16x8 vv = *reinterpret_cast<const bf16x8*>(Vb + (long)krow * a.ldv + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) Vt[(ch * 8 + j) * VT_LD + row] = valid ? vv[j] : (bf16_t)0.f;
        }
        __syncthreads();

        // ---- S^T = K·Q^T for the two 32-key sub-tiles ----
        f32x16 sacc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kt][i] = 0.f;
            const int row = kt * 32 + r;
            const int swz = (DH == 64) ? ((row >> 1) & 7) : ((row >> 2) & 3);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + row * ROWB + (((2 * s + hh) ^ swz) << 4));
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kt], 0, 0, 0);
            }
        }
        // ---- online softmax (row = this lane's query; keys split over the two half-waves) ----
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kv0 + kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                const float s = (key < a.Nk) ? sacc[kt][i] : -INFINITY;
                sacc[kt][i] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        float psum = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(sacc[kt][i] * c - mc);
                psum += p;
                pf[kt][i >> 3][i & 7] = (bf16_t)p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
        // ---- O^T += V^T · P^T ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int kbase = kt * 32 + 16 * s2 + 4 * hh;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const bf16_t* vp = Vt + (d * 32 + r) * VT_LD + kbase;
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vp);
                    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vp + 8);
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][s2], oacc[d], 0, 0, 0);
                }
            }
    }

    // ---- normalise and store O[b][h][q][d] ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + r;
    if (q < a.Nq) {
        bf16_t* op = a.O + (((long)b * a.H + head) * a.Nq + q) * DH;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = d * 32 + 8 * g + 4 * hh;
                bf16x4 pk = {(bf16_t)(oacc[d][4 * g + 0] * inv), (bf16_t)(oacc[d][4 * g + 1] * inv),
                             (bf16_t)(oacc[d][4 * g + 2] * inv), (bf16_t)(oacc[d][4 * g + 3] * inv)};
                *reinterpret_cast<bf16x4*>(op + col) = pk;
            }
    }
}

int ldt_attn_launch(const AttnArgs* a, int dh, hipStream_t s) {
    LDT_REQUIRE(a->B > 0 && a->H > 0 && a->Nq > 0 && a->Nk > 0, LDT_ESHAPE, "attention: empty problem B=%d H=%d Nq=%d Nk=%d", a->B, a->H, a->Nq, a->Nk);
    LDT_REQUIRE(dh == 32 || dh == 64, LDT_ESHAPE, "attention: head dim %d not built (32, 64)", dh);
    LDT_REQUIRE(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->q_batch_stride % 8 == 0 && a->kv_batch_stride % 8 == 0 &&
                ldt_aligned16(a->Q) && ldt_aligned16(a->K) && ldt_aligned16(a->V) && ldt_aligned16(a->O), LDT_EALIGN,
                "attention: Q/K/V rows must be 16-byte aligned");
    LDT_REQUIRE(a->H <= 65535 && a->B <= 65535, LDT_ESHAPE, "attention: grid too large");
    dim3 grid((a->Nq + 127) / 128, a->H, a->B), block(256);
    if (dh == 64) hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, block, 0, s, *a);
    else hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, block, 0, s, *a);
    return ldt_check_launch("attn_fwd");
}
