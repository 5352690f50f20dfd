// fp32 SGEMM (NT) on the fp32-input matrix cores:  C[M,N] = act_out( act_in(A[M,K]) · B[N,K]^T + bias[N] ), fp32 or bf16 out.
//
// The fp32-critical linears of the path: the time MLP (model/layers.py:17), every block's AdaLN Linear (layers.py:172,214,
// 238,244) — tabulated for all N steps per sample() call, or per step and per sample in conditional sampling — and the
// LN-folding S / C rows.  v_mfma_f32_32x32x2_f32 multiplies exact fp32 operands and accumulates in fp32 (no bf16/tf32
// truncation), so the 1e-10 / 1e-12 parity bars of the scalar-FMA kernel it replaces hold unchanged; gfx950 issues it at
// 157 TFLOP/s (MI355X_MICROARCH.md), the rate of the fp32 VALU, but one instruction does the work of 64 v_fma and the
// operands come from LDS once per 32x32 block instead of once per 4x4 register tile.
//
// Structure: 256 threads = 4 waves; WG tile TM x TN (128x128: waves 2x2 of 64x64; 32x128: waves 1x4 of 32x32 for the skinny
// per-step conditional AdaLN, M = batch), k-slabs of BK = 16 (32 for the skinny tile), operands staged global -> registers -> LDS
// ([row][k], row stride BK + 4 floats: conflict-free 16-B reads), double-buffered in registers (the next k-slab's global loads are in flight during the
// MFMAs).  Per 8 k's a lane reads ONE 16-B chunk per 32-row block: lanes 0-31 hold k 0..3, lanes 32-63 k 4..7 of that slab —
// the MFMA pairs (k, k+4) instead of (k, k+1), the same permutation on both operands.
#include "kernels.h"

// k-slab depth BK (floats per staged row: BK + 4 pad -> conflict-free 16-B reads): 16 for the square tile; 32 for the skinny one, whose
// launch is pure weight streaming with ONE slab in flight per workgroup — a deeper slab is more bytes per memory latency

typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sg_act(float v, int act) {
    switch (act) {
        case ACT_SILU: return v / (1.0f + expf(-v));
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_GELU: return gelu_erf(v);
        default: return v;
    }
}

template <int TM, int TN, int WGM, int WGN, int SG_BK = 16>    // WG tile, wave grid (WGM x WGN = 4 waves), k-slab depth
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const SgemmArgs a) {
    constexpr int SG_LD = SG_BK + 4, CPR = SG_BK / 4;                  // floats per staged row, 16-B chunks per slab row
    constexpr int WM = TM / WGM, WN = TN / WGN;       // per-wave tile
    constexpr int BM = WM / 32, BN = WN / 32;          // 32x32 MFMA blocks per wave
    constexpr int CA = (TM * CPR + 255) / 256, CB = (TN * CPR + 255) / 256;   // 16-B chunks per thread and slab
    static_assert(WGM * WGN == 4 && BM >= 1 && BN >= 1, "tile shape");
    __shared__ __attribute__((aligned(16))) float As[TM * SG_LD];
    __shared__ __attribute__((aligned(16))) float Bs[TN * SG_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;

    f32x16_t acc[BM][BN];
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging map: chunk c of a slab = (row = c / CPR, k-chunk = c % CPR); thread t takes chunks t, t + 256, ...
    f32x4 ra[CA], rb[CB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            const int ch = tid + c * 256, row = ch / CPR, kc = (ch % CPR) * 4;
            const int m = m0 + row;
            ra[c] = (ch < TM * CPR && m < a.M && k0 + kc < a.K) ? *reinterpret_cast<const f32x4*>(a.A + (long)m * a.lda + k0 + kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int ch = tid + c * 256, row = ch / CPR, kc = (ch % CPR) * 4;
            const int n = n0 + row;
            rb[c] = (ch < TN * CPR && n < a.N && k0 + kc < a.K) ? *reinterpret_cast<const f32x4*>(a.B + (long)n * a.ldb + k0 + kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            const int ch = tid + c * 256, row = ch / CPR, kc = (ch % CPR) * 4;
            f32x4 v = ra[c];
            if (a.act_in) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = sg_act(v[j], a.act_in);
            }
            if (ch < TM * CPR) *reinterpret_cast<f32x4*>(&As[row * SG_LD + kc]) = v;
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int ch = tid + c * 256, row = ch / CPR, kc = (ch % CPR) * 4;
            if (ch < TN * CPR) *reinterpret_cast<f32x4*>(&Bs[row * SG_LD + kc]) = rb[c];
        }
    };

    const int l32 = lane & 31, khalf = (lane >> 5) * 4;   // this lane's row inside a 32-block, its k offset inside an 8-k group
    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += SG_BK) {
        __syncthreads();                                  // previous slab's readers are done
        stage();
        __syncthreads();
        if (k0 + SG_BK < a.K) fetch(k0 + SG_BK);          // in flight during the MFMAs below
#pragma unroll
        for (int kg = 0; kg < SG_BK; kg += 8) {
            f32x4 fa[BM], fb[BN];
#pragma unroll
            for (int i = 0; i < BM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(&As[(wm * WM + i * 32 + l32) * SG_LD + kg + khalf]);
#pragma unroll
            for (int j = 0; j < BN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn * WN + j * 32 + l32) * SG_LD + kg + khalf]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < BM; ++i)
#pragma unroll
                    for (int j = 0; j < BN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
    }

    // D[i][j] of a 32x32 block: lane holds column j = lane & 31 (the B / N side), rows i = 8*(r>>2) + 4*(lane>>5) + (r&3)
#pragma unroll
    for (int bi = 0; bi < BM; ++bi)
#pragma unroll
        for (int bj = 0; bj < BN; ++bj) {
            const int n = n0 + wn * WN + bj * 32 + l32;
            if (n >= a.N) continue;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + bi * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (m >= a.M) continue;
                const float v = sg_act(acc[bi][bj][r] + bv, a.act_out);
                if (a.out_bf16) reinterpret_cast<bf16_t*>(a.C)[(long)m * a.ldc + n] = (bf16_t)v;
                else reinterpret_cast<float*>(a.C)[(long)m * a.ldc + n] = v;
            }
        }
}

// -> true when the MFMA kernel took the problem (K % 4 == 0 — a ragged last slab is zero-filled —, 16-byte aligned rows);
// otherwise the caller falls back to the scalar kernel (the 3-channel input convs, 131/259-channel grouper rows).
bool ldt_sgemm_mfma_try(const SgemmArgs* a, hipStream_t s, int* status) {
    if (a->K % 4 != 0 || a->K < 8 || a->lda % 4 != 0 || a->ldb % 4 != 0 || !ldt_aligned16(a->A) || !ldt_aligned16(a->B)) return false;
    if ((long)a->M * a->N < 64 * 64) return false;       // tiny problems: launch-bound either way, keep the simple kernel
    if (a->M <= 48) {                                     // skinny: per-step per-sample AdaLN rows (M = batch), weight streaming
        // 32 x 128 tiles, 32-deep slabs: M = 32, N = 149,504, K = 1024 in 161 us = 3.8 TB/s of weights with the SiLU of the AdaLN input applied
        // by the caller (176 with it in stage(); 32 x 256 tiles: 196; 16-deep slabs: 215).  Round-4 probes (tools/dbg/skinny_sgemm_bench.py,
        // profiles/r04_skinny_sgemm_probes.txt): a second slab of register prefetch is SLOWER (200 us), rows padded off the 4-KiB stride 165-173,
        // slab-contiguous weights 163-168; one workgroup per CU alone (N = 32,768) runs 1.5 us per slab and the time grows 35 us per further
        // workgroup per CU; an LDS-DMA ring form of this tile (4 stages, counted vmcnt, one barrier per slab) ran 150-158 us stand-alone but
        // 30 us SLOWER per step inside the sampling loop (two 80-KB workgroups per CU = 2.3 rounds with a tail; this kernel's 1168 workgroups
        // are all resident) and was not kept.  A plain read stream of the same bytes gets 6.0 TB/s on the box; the fp32 MFMAs alone are 62 us.
        dim3 grid((a->N + 127) / 128, (a->M + 31) / 32);
        hipLaunchKernelGGL((sgemm_mfma_kernel<32, 128, 1, 4, 32>), grid, dim3(256), 0, s, *a);
    } else {
        dim3 grid((a->N + 127) / 128, (a->M + 127) / 128);
        if (grid.y >= 65536) return false;
        hipLaunchKernelGGL((sgemm_mfma_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, *a);
    }
    *status = ldt_check_launch("sgemm_mfma");
    return true;
}
