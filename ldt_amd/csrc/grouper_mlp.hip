// LocalGrouper rows + PreExtraction + neighbour max in ONE kernel (Compressor/layers.py:288-319 grouping and normalisation,
// :115-160 PreExtraction = Conv+BN+ReLU, residual [Conv+BN+ReLU, Conv] + ReLU, :186 adaptive_max_pool1d over the k neighbours).
//
// The unfused chain writes the grouped rows U [B*S*k][2D+3 -> 320] bf16 (2.7 GB at B = 1024, S*k = 4096 neighbour rows per cloud,
// D = 128), then three [B*S*k][128] bf16 activations and reads them all back: ~13 GB of HBM traffic for 134 MB of result.  Here a
// wave's work item is a tile of 32 neighbour rows — 32 rows ARE the 32 columns of a 32x32 MFMA tile (k = 32 m: m tiles per group,
// each by whichever wave comes to it, combined by atomicMax; k = 16 or 8: 2 or 4 whole groups share a tile):
//
//   layer 1  D1[c][j] = W1[c][:] . U[j][:]     weights = A operand (from LDS), U^T = B operand built in registers from the
//                                              gathered feature rows (lane j gathers the 8 channels its k-slot needs),
//   layer 2  D2[c][j] = W2[c][:] . h1[j][:]    h1 = bf16(relu(D1 + b1)) re-used IN PLACE as the B operand: a lane's 8
//                                              accumulator registers of a 16-channel slab are exactly 8 k-slots of lane j;
//                                              the weight image is stored with the matching k permutation,
//   layer 3  D3[j][c] = r[j][:] . W3[c][:] + h1[j][c]    operands swapped, so a lane now holds ONE output channel for 16 of the
//                                              32 neighbours: the max over neighbours is 15 in-lane max + one cross-half
//                                              exchange.  The residual h1 is added by the matrix core too (x identity).
//
// Numerics are those of the unfused chain (bf16 operands, fp32 accumulation, bf16 rounding of U, h1, r and the result), only
// the k order of the sums differs.  Weights live in LDS for the whole (persistent) kernel as ready-made 1 KB MFMA fragments
// (host-built image: Compressor.pack); the XCD that owns a cloud keeps its 1 MB of features in its own L2.
#include "kernels.h"

namespace {

constexpr int GF_S1 = 17;                                   // layer-1 k-steps: 8 normalised features | 8 anchor features | xyz
constexpr int GF_L2 = GF_S1 * 4, GF_L3 = GF_L2 + 32, GF_NFRAG = GF_L3 + 32;
constexpr int GF_W_BYTES = GF_NFRAG * 1024;
constexpr int GF_BIAS = GF_W_BYTES;                         // b1 | b2 | b3: 3 x 128 floats
constexpr int GF_ALPHA = GF_BIAS + 3 * 128 * 4;             // alpha[0..130] (padded to 144), then beta
constexpr int GF_BETA = GF_ALPHA + 144 * 4;
constexpr int GF_LDS = GF_BETA + 144 * 4;

__device__ __forceinline__ bf16x8 to_bf16x8(const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    return o;
}

// bf16(relu(acc)) of a 32-channel block -> the two 16-channel B/A fragments the next layer consumes (slot e of k-step t = reg 8t + e)
__device__ __forceinline__ void relu_frags(const f32x16& acc, bf16x8& f0, bf16x8& f1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { f0[e] = (bf16_t)fmaxf(acc[e], 0.f); f1[e] = (bf16_t)fmaxf(acc[8 + e], 0.f); }
}

template <int GPT>   // groups per 32-row tile: 1 (k a multiple of 32: k / 32 tiles per group), 2 (k = 16) or 4 (k = 8)
__global__ __launch_bounds__(512) void grouper_mlp_kernel(const GroupMlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    for (int i = tid * 16; i < GF_W_BYTES; i += 512 * 16)
        *reinterpret_cast<f32x4*>(gsm + i) = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.wimg) + i);
    if (tid < 128) {
        float* bs = reinterpret_cast<float*>(gsm + GF_BIAS);
        bs[tid] = a.b1[tid]; bs[128 + tid] = a.b2[tid]; bs[256 + tid] = a.b3[tid];
    } else if (tid < 128 + 144) {
        const int c = tid - 128;
        reinterpret_cast<float*>(gsm + GF_ALPHA)[c] = c < 131 ? a.alpha[c] : 0.f;
        reinterpret_cast<float*>(gsm + GF_BETA)[c] = c < 131 ? a.beta[c] : 0.f;
    }
    __syncthreads();
    const char* Wl = gsm + lane * 16;
    const float* bias = reinterpret_cast<const float*>(gsm + GF_BIAS);
    const float* alpha = reinterpret_cast<const float*>(gsm + GF_ALPHA);
    const float* beta = reinterpret_cast<const float*>(gsm + GF_BETA);

    // identity fragments of the residual MFMA: output channel n (= lane & 31 of its block) picks k-slot e of k-step t
    bf16x8 idf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) idf[t][e] = (bf16_t)((j == 16 * t + 8 * (e >> 2) + 4 * h + (e & 3)) ? 1.f : 0.f);

    // Work items = 32-row tiles.  Clouds b = xcd, xcd + 8, ... belong to the WGs of one XCD (WG x runs on XCD x % 8): its waves
    // walk that list of clouds tile by tile, so an XCD's L2 holds one or two clouds' features at a time.  With few clouds
    // (a.flat) every cloud is spread over all XCDs instead.
    const int xcd = blockIdx.x & 7;
    const int nw = a.flat ? gridDim.x * 8 : (gridDim.x >> 3) * 8;          // waves sharing one work list
    const int w0 = a.flat ? blockIdx.x * 8 + wave : (blockIdx.x >> 3) * 8 + wave;
    const int K = a.k, tpg = GPT == 1 ? K / 32 : 1;                        // tiles per group (GPT == 1) / GPT groups per tile
    const int items = GPT == 1 ? a.S * tpg : (a.S + GPT - 1) / GPT;        // tiles per cloud
    const int nclouds = a.flat ? a.B : (a.B - xcd + 7) / 8;
    const long total = (long)nclouds * items;
    const double cnt = (double)a.S * K * 131.0;
    int b_prev = -1;
    float inv = 0.f;
    for (long f = w0; f < total; f += nw) {
        {
            const int cl = (int)(f / items), item = (int)(f - (long)cl * items);
            const int b = a.flat ? cl : xcd + 8 * cl;
            if (b != b_prev) {                                             // per-sample unbiased std of (g - anchor), layers.py:311-312
                const double mean = fx_load(&a.stats[2 * b]) / cnt;
                const double var = (fx_load(&a.stats[2 * b + 1]) - cnt * mean * mean) / (cnt - 1.0);
                inv = 1.0f / ((float)sqrt(var > 0.0 ? var : 0.0) + 1e-5f);
                b_prev = b;
            }
            const float* featb = a.feat + (long)b * a.n * 128;
            const float* xyzb = a.xyz + (long)b * a.n * 3;
            const int tile = GPT == 1 ? item % tpg : 0;
            int sidx = GPT == 1 ? item / tpg : item * GPT + j / (32 / GPT);   // a ragged last tile repeats the last group (never stored)
            if (GPT > 1 && sidx >= a.S) sidx = a.S - 1;
            const int nb = GPT == 1 ? tile * 32 + j : j % (32 / GPT);
            const long g = (long)b * a.S + sidx;
            const int ci = a.fps_idx[g];
            const int pi = a.knn_idx[g * K + nb];
            const float* fg = featb + (long)pi * 128 + 8 * h;
            const float* fa = featb + (long)ci * 128 + 8 * h;

            // ---- layer 1: U^T built on the fly -------------------------------------------------------------------
            f32x16 acc[4];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + blk * 32 + 8 * q + 4 * h);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[blk][4 * q + r] = b4[r];
                }
            // every gathered row of the tile is requested before the first MFMA: one memory latency per tile, not one per k-step
            f32x4 gq[8][2], aq[8][2];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                gq[s][0] = *reinterpret_cast<const f32x4*>(fg + 16 * s); gq[s][1] = *reinterpret_cast<const f32x4*>(fg + 16 * s + 4);
                aq[s][0] = *reinterpret_cast<const f32x4*>(fa + 16 * s); aq[s][1] = *reinterpret_cast<const f32x4*>(fa + 16 * s + 4);
            }
            float xg[3] = {0.f, 0.f, 0.f}, xa[3] = {0.f, 0.f, 0.f};
            if (h == 0) {
#pragma unroll
                for (int e = 0; e < 3; ++e) { xg[e] = xyzb[(long)pi * 3 + e]; xa[e] = xyzb[(long)ci * 3 + e]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 af[8];                                                  // the anchor's own features (k-steps 8..15), kept as bf16
#pragma unroll
            for (int s = 0; s < 8; ++s) {                                  // alpha * ((g - anchor) * inv) + beta, feature channels
                const f32x4 al0 = *reinterpret_cast<const f32x4*>(alpha + 16 * s + 8 * h), al1 = *reinterpret_cast<const f32x4*>(alpha + 16 * s + 8 * h + 4);
                const f32x4 be0 = *reinterpret_cast<const f32x4*>(beta + 16 * s + 8 * h), be1 = *reinterpret_cast<const f32x4*>(beta + 16 * s + 8 * h + 4);
                float v[8], w[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = al0[e] * ((gq[s][0][e] - aq[s][0][e]) * inv) + be0[e];
                    v[4 + e] = al1[e] * ((gq[s][1][e] - aq[s][1][e]) * inv) + be1[e];
                    w[e] = aq[s][0][e]; w[4 + e] = aq[s][1][e];
                }
                const bf16x8 uf = to_bf16x8(v);
                af[s] = to_bf16x8(w);
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
                    acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Wl + (s * 4 + blk) * 1024), uf, acc[blk], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {                                  // the anchor's own features, repeated on every row
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
                    acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Wl + ((8 + s) * 4 + blk) * 1024), af[s], acc[blk], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            {                                                              // normalised xyz offsets: 3 live k-slots on the lower half
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (h == 0) {
#pragma unroll
                    for (int e = 0; e < 3; ++e) v[e] = alpha[128 + e] * ((xg[e] - xa[e]) * inv) + beta[128 + e];
                }
                const bf16x8 uf = to_bf16x8(v);
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
                    acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Wl + (16 * 4 + blk) * 1024), uf, acc[blk], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            bf16x8 h1[8];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) relu_frags(acc[blk], h1[2 * blk], h1[2 * blk + 1]);

            // ---- layer 2 -------------------------------------------------------------------------------------------
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 128 + blk * 32 + 8 * q + 4 * h);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[blk][4 * q + r] = b4[r];
                }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
                    acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Wl + (GF_L2 + s * 4 + blk) * 1024), h1[s], acc[blk], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            bf16x8 rr[8];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) relu_frags(acc[blk], rr[2 * blk], rr[2 * blk + 1]);

            // ---- layer 3 (+ residual), operands swapped: lane = output channel, registers = neighbours ----------------
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const float b3 = bias[256 + blk * 32 + j];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[blk][r] = b3;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
                    acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rr[s], *reinterpret_cast<const bf16x8*>(Wl + (GF_L3 + s * 4 + blk) * 1024), acc[blk], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1[2 * blk + t], idf[t], acc[blk], 0, 0, 0);

            // ---- max over the neighbours, ReLU, bf16 rounding (both monotone: applied once, after the max) ----------------
            constexpr int RPG = 16 / GPT;                                  // accumulator registers per group and half:
#pragma unroll                                                             // group q = tile rows [q k, (q + 1) k) = registers [q RPG, (q + 1) RPG)
            for (int q = 0; q < GPT; ++q) {
                const int so = GPT == 1 ? sidx : item * GPT + q;
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    float m = acc[blk][q * RPG];
#pragma unroll
                    for (int r = 1; r < RPG; ++r) m = fmaxf(m, acc[blk][q * RPG + r]);
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    m = (float)(bf16_t)fmaxf(m, 0.f);
                    float* o = a.out + ((long)b * a.S + so) * 128 + blk * 32 + j;
                    if (h == 0 && so < a.S) {
                        // several tiles per group: the group's maximum is combined in memory (results are >= 0 after the ReLU:
                        // their bit patterns order like integers; the launcher zeroes `out`)
                        if (GPT == 1 && tpg > 1) atomicMax(reinterpret_cast<int*>(o), __float_as_int(m));
                        else *o = m;
                    }
                }
            }
        }
    }
}

}  // namespace

int ldt_grouper_mlp_launch(const GroupMlpArgs* a, hipStream_t st) {
    LDT_REQUIRE(a->B > 0 && a->n > 0 && a->S > 0, LDT_ESHAPE, "grouper_mlp: bad shape");
    LDT_REQUIRE(a->k == 8 || a->k == 16 || (a->k > 0 && a->k % 32 == 0), LDT_ESHAPE,
                "grouper_mlp: k = %d neighbours (built for 8, 16 and multiples of 32)", a->k);
    LDT_REQUIRE(ldt_aligned16(a->feat) && ldt_aligned16(a->wimg) && ldt_aligned16(a->out), LDT_EALIGN, "grouper_mlp: feat / image / out must be 16-byte aligned");
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    GroupMlpArgs k = *a;
    k.flat = a->B < 16 ? 1 : 0;                                // few clouds: spread every cloud over all XCDs instead
    const int gpt = a->k >= 32 ? 1 : 32 / a->k;
    int grid = cus / 8 * 8;
    if (grid < 8) grid = 8;
    const long tiles_per_cloud = gpt == 1 ? (long)a->S * (a->k / 32) : (a->S + gpt - 1) / gpt;
    if (k.flat) { const long need = (a->B * tiles_per_cloud + 7) / 8; if (grid > need) grid = (int)need; }
    if (a->k > 32) {                                            // tiles of one group meet in memory (atomicMax on results >= 0)
        const hipError_t e = hipMemsetAsync(a->out, 0, sizeof(float) * (size_t)a->B * a->S * 128, st);
        if (e != hipSuccess) { ldt_set_error("grouper_mlp: memset: %s", hipGetErrorString(e)); return (int)e; }
    }
#define GF_LAUNCH(G)                                                                            \
    do {                                                                                        \
        LDT_ENSURE_LDS((grouper_mlp_kernel<G>), GF_LDS, "grouper_mlp");                         \
        hipLaunchKernelGGL((grouper_mlp_kernel<G>), dim3(grid), dim3(512), GF_LDS, st, k);      \
    } while (0)
    if (gpt == 1) GF_LAUNCH(1); else if (gpt == 2) GF_LAUNCH(2); else GF_LAUNCH(4);
#undef GF_LAUNCH
    return ldt_check_launch("grouper_mlp");
}
