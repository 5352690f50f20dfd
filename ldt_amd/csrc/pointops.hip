// Point-cloud front end of the Compressor encoder (gfx950): farthest point sampling, kNN (distance + top-k
// selection), neighbourhood gather + anchor normalisation, max-pooling, ActNorm and the reparameterised
// posterior draw.  Integer/index work is exact; floating point follows the operation order stated per kernel.
// Reference: model/Compressor/layers.py:65-112 (square_distance / knn_point / cluster), :288-319
// (LocalGrouper.forward), model/functional/src/sampling/sampling.cu:86-167 (FPS twin), Network.py:26-29,76.
#include <stdlib.h>

#include "kernels.h"

#define TRY_LAUNCH(what) do { const int _rc = ldt_check_launch(what); if (_rc != LDT_OK) return _rc; } while (0)

// ------------------------------------------------------------------------------------------------
// FPS: one 512-thread workgroup per cloud; points and running min-distances live in registers
// (PPT points per thread, point k = tid + 512*j).  The per-iteration argmax is a wave shuffle reduction + one LDS hop across
// the 8 waves, and the winner's COORDINATES travel with (distance, index) through that reduction, so the next iteration starts
// from LDS — no dependent global load of p[best] on the serial chain (m - 1 iterations; was ~0.6 us of each 1.3 us iteration).
// Several clouds share a CU (40 VGPRs, 112 B of LDS: four workgroups per CU): launch as many clouds at once as the caller has.
// Semantics = the vendored CUDA twin: start at index 0, distances initialised to 1e38, d = (dx*dx + dy*dy) + dz*dz without
// FMA contraction, running min, argmax; a tie goes to the smaller (k % 512, k / 512)  (sampling.cu:141-158: strict '>' per
// thread, then a pairwise tree that keeps the lower thread on equality).
// skip_near_origin (default off): the upstream pointnet2_ops kernel the reference actually calls (not vendored)
// additionally ignores points with |p|^2 <= 1e-3 — they never update their distance and are never selected.
struct Cand { float d; int k; };
__device__ __forceinline__ bool beats(float da, int ka, float db, int kb) {
    // both candidates come from the same 512-stride layout: rank = (k & 511, k >> 9)
    if (da != db) return da > db;
    const int ra = ((ka & 511) << 16) | (ka >> 9), rb = ((kb & 511) << 16) | (kb >> 9);
    return ra < rb;
}

template <int PPT>
__global__ __launch_bounds__(512) void fps_kernel(const float* __restrict__ xyz, int n, int m, int skip_near_origin, int* __restrict__ idx_out) {
    __shared__ float s_d[8];
    __shared__ int s_k[8];
    __shared__ float s_c[8][3];
    __shared__ float s_pt[3];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = xyz + (long)b * n * 3;
    float px[PPT], py[PPT], pz[PPT], dist[PPT];
    bool live[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = tid + 512 * j;
        const bool in = k < n;
        px[j] = in ? p[3 * k] : 0.f; py[j] = in ? p[3 * k + 1] : 0.f; pz[j] = in ? p[3 * k + 2] : 0.f;
        dist[j] = 1e38f;
        float mag;
        {
#pragma clang fp contract(off)
            mag = (px[j] * px[j] + py[j] * py[j]) + pz[j] * pz[j];
        }
        live[j] = in && !(skip_near_origin && mag <= 1e-3f);
    }
    if (tid == 0) { idx_out[(long)b * m] = 0; s_pt[0] = p[0]; s_pt[1] = p[1]; s_pt[2] = p[2]; }
    __syncthreads();
    for (int it = 1; it < m; ++it) {
        const float x1 = s_pt[0], y1 = s_pt[1], z1 = s_pt[2];
        float best = -1.f; int besti = 0;
        float bx = 0.f, by = 0.f, bz = 0.f;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                if (live[j]) {
                    const float dx = px[j] - x1, dy = py[j] - y1, dz = pz[j] - z1;
                    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                    const float s = xx + yy;
                    const float d = s + zz;
                    const float d2 = fminf(d, dist[j]);
                    dist[j] = d2;
                    if (d2 > best) { best = d2; besti = tid + 512 * j; bx = px[j]; by = py[j]; bz = pz[j]; }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(best, o, 64);
            const int ok = __shfl_xor(besti, o, 64);
            const float ox = __shfl_xor(bx, o, 64), oy = __shfl_xor(by, o, 64), oz = __shfl_xor(bz, o, 64);
            if (beats(od, ok, best, besti)) { best = od; besti = ok; bx = ox; by = oy; bz = oz; }
        }
        if (lane == 0) { s_d[wave] = best; s_k[wave] = besti; s_c[wave][0] = bx; s_c[wave][1] = by; s_c[wave][2] = bz; }
        __syncthreads();                     // (also: every thread has read s_pt of this iteration)
        if (tid == 0) {
            float bd = s_d[0]; int bk = s_k[0], bw = 0;
            for (int w = 1; w < 8; ++w) if (beats(s_d[w], s_k[w], bd, bk)) { bd = s_d[w]; bk = s_k[w]; bw = w; }
            idx_out[(long)b * m + it] = bk;
            // a wave with no live point reports (best = -1, index 0, coordinates 0): index 0 then needs its real coordinates
            const bool none = bd < 0.f;
            s_pt[0] = none ? p[0] : s_c[bw][0]; s_pt[1] = none ? p[1] : s_c[bw][1]; s_pt[2] = none ? p[2] : s_c[bw][2];
        }
        __syncthreads();
    }
}

int ldt_fps_launch(const float* xyz, int B, int n, int m, int skip_near_origin, int* idx, hipStream_t s) {
    LDT_REQUIRE(B > 0 && n > 0 && m > 0 && m <= n, LDT_ESHAPE, "fps: B=%d n=%d m=%d", B, n, m);
    LDT_REQUIRE(n <= 512 * 16, LDT_ESHAPE, "fps: n=%d > 8192 points per cloud not built", n);
    static const int wave_env = getenv("LDT_FPS_WAVE") ? atoi(getenv("LDT_FPS_WAVE")) : -1;   // 0 / 1 force (tools/dbg)
    // many clouds: one wave per cloud (all CUs busy from 1024 clouds up; a lone cloud is faster on 512 threads)
    if (n <= 64 * 32 && (wave_env == 1 || (wave_env != 0 && B >= 512))) {
        return ldt_fps_wave_launch(xyz, B, n, m, skip_near_origin, idx, s);
    }
    if (n <= 512 * 4) hipLaunchKernelGGL(fps_kernel<4>, dim3(B), dim3(512), 0, s, xyz, n, m, skip_near_origin, idx);
    else hipLaunchKernelGGL(fps_kernel<16>, dim3(B), dim3(512), 0, s, xyz, n, m, skip_near_origin, idx);
    return ldt_check_launch("fps");
}

// ------------------------------------------------------------------------------------------------
// kNN: one wave per query centre.  d(c,p) = ((-2 * (c.p)) + |c|^2) + |p|^2 (square_distance, layers.py:81-83),
// c.p and the norms accumulated x,y,z in order with FMA.  The k smallest are selected exactly with a 32-step
// radix select over order-preserving integer keys (wave-wide counts), ties at the k-th value resolved towards
// the smaller point index; the result is an UNORDERED index set like topk(sorted=False) (:97).
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int PPL>   // points per lane: n <= 64*PPL
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ xyz, const float* __restrict__ centers,
                                                  int n, int S, long nq, int k, int* __restrict__ out, float* __restrict__ dist_out) {
    const int lane = threadIdx.x & 63;
    const long q = blockIdx.x * 4L + (threadIdx.x >> 6);        // global centre id in [0, B*S)
    if (q >= nq) return;
    const int b = (int)(q / S);
    const float* p = xyz + (long)b * n * 3;
    const float cx = centers[q * 3], cy = centers[q * 3 + 1], cz = centers[q * 3 + 2];
    const float cn = fmaf(cz, cz, fmaf(cy, cy, cx * cx));
    uint32_t key[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int i = lane + 64 * j;
        if (i < n) {
            const float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
            const float dot = fmaf(cz, z, fmaf(cy, y, cx * x));
            const float pn = fmaf(z, z, fmaf(y, y, x * x));
            float d;
            {
#pragma clang fp contract(off)
                d = -2.f * dot;
                d = d + cn;
                d = d + pn;
            }
            key[j] = fkey(d);
            if (dist_out) dist_out[q * n + i] = d;
        } else key[j] = 0xFFFFFFFFu;
    }
    int* o = out + q * (long)k;
    // Fast path (k <= 64): the k-th smallest key cannot exceed tau = the k-th smallest of the 64 per-lane minima (those are k
    // distinct points <= tau), so only points <= tau are candidates — a few more than k, unless many points tie or crowd below
    // tau.  With <= 64 candidates they are compacted to one per lane (in index order) and the exact radix select runs on 64
    // values: 32 + 32 + 32 ballots instead of 32 x PPL.  Same result, same output order as the full select below.
    if (k <= 64) {
        uint32_t m = key[0];
#pragma unroll
        for (int j = 1; j < PPL; ++j) m = min(m, key[j]);
        uint32_t tau = 0;
        int need1 = k;
        for (int bit = 31; bit >= 0; --bit) {
            const bool match = (bit == 31) || (((m ^ tau) >> (bit + 1)) == 0);
            const int c0 = (int)__popcll(__ballot(match && !((m >> bit) & 1u)));
            if (c0 < need1) { tau |= (1u << bit); need1 -= c0; }
        }
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < PPL; ++j) cnt += (int)__popcll(__ballot(key[j] <= tau && lane + 64 * j < n));
        if (cnt <= 64) {
            __shared__ uint2 cand_s[4][64];
            uint2* cand = cand_s[threadIdx.x >> 6];                     // wave-private: no workgroup barrier needed
            const unsigned long long lt = (1ull << lane) - 1ull;
            int base = 0;
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const bool sel = key[j] <= tau && lane + 64 * j < n;
                const unsigned long long bal = __ballot(sel);
                if (sel) cand[base + (int)__popcll(bal & lt)] = make_uint2(key[j], (uint32_t)(lane + 64 * j));
                base += (int)__popcll(bal);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint2 mine = cand[lane];
            const bool live = lane < cnt;
            const uint32_t ck = live ? mine.x : 0xFFFFFFFFu;
            uint32_t kth = 0;
            int need2 = k;
            for (int bit = 31; bit >= 0; --bit) {
                const bool match = (bit == 31) || (((ck ^ kth) >> (bit + 1)) == 0);
                const int c0 = (int)__popcll(__ballot(live && match && !((ck >> bit) & 1u)));
                if (c0 < need2) { kth |= (1u << bit); need2 -= c0; }
            }
            const bool sel = live && ck < kth;
            const unsigned long long bal = __ballot(sel);
            if (sel) o[__popcll(bal & lt)] = (int)mine.y;
            const int nlt = (int)__popcll(bal);
            const bool eq = live && ck == kth;
            const unsigned long long beq = __ballot(eq);
            const int r = (int)__popcll(beq & lt);
            if (eq && r < k - nlt) o[nlt + r] = (int)mine.y;
            return;
        }
    }
    // radix select: the k-th smallest key
    uint32_t prefix = 0;
    int need = k;
    // (counts by ballot + scalar popcount: wave-uniform in SGPRs, no 6-deep shuffle chain per bit — the chain's latency, not
    //  the compares, was the kernel's time: 2.9 ms per 1024 clouds in round 1's profile)
    for (int bit = 31; bit >= 0; --bit) {
        int c0 = 0;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const bool match = (bit == 31) || (((key[j] ^ prefix) >> (bit + 1)) == 0);
            c0 += (int)__popcll(__ballot(match && !((key[j] >> bit) & 1u)));
        }
        if (c0 >= need) { /* k-th has this bit 0 */ }
        else { prefix |= (1u << bit); need -= c0; }
    }
    // emit: all keys < kth, then `need` keys == kth in index order
    int base = 0;
    int lt_total = 0;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const bool sel = key[j] < prefix;
        const unsigned long long bal = __ballot(sel);
        if (sel) o[base + __popcll(bal & ((1ull << lane) - 1ull))] = lane + 64 * j;
        base += __popcll(bal);
    }
    lt_total = base;
    int eq_left = k - lt_total;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const bool eq = (key[j] == prefix) && (lane + 64 * j < n);
        const unsigned long long bal = __ballot(eq);
        const int r = __popcll(bal & ((1ull << lane) - 1ull));
        if (eq && r < eq_left) o[base + r] = lane + 64 * j;
        const int took = min(eq_left, (int)__popcll(bal));
        base += took; eq_left -= took;
    }
}

int ldt_knn_launch(const float* xyz, const float* centers, int B, int n, int S, int k, int* out, float* dist_out, hipStream_t s) {
    LDT_REQUIRE(B > 0 && n > 0 && S > 0 && k > 0 && k <= n, LDT_ESHAPE, "knn: B=%d n=%d S=%d k=%d", B, n, S, k);
    LDT_REQUIRE(n <= 64 * 128, LDT_ESHAPE, "knn: n=%d > 8192 points per cloud not built", n);
    const long nq = (long)B * S;
    dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    if (n <= 64 * 32) hipLaunchKernelGGL(knn_kernel<32>, grid, block, 0, s, xyz, centers, n, S, nq, k, out, dist_out);
    else hipLaunchKernelGGL(knn_kernel<128>, grid, block, 0, s, xyz, centers, n, S, nq, k, out, dist_out);
    return ldt_check_launch("knn");
}

// ------------------------------------------------------------------------------------------------
// LocalGrouper (layers.py:297-315), 'anchor' normalisation.  Two kernels:
//  (1) per-sample sum / sum-of-squares of (g - anchor) over all S*k*(D+3) elements, fp64 accumulation
//      (the unbiased std of torch.std(..., dim=-1), :311);
//  (2) rows U[b,s,j,:] = [ alpha*(g - anchor)/(std+1e-5) + beta | centre feature ] as bf16, K-padded for the
//      MFMA GEMM of PreExtraction (:315, :178-187).
// feat [B*n, D] fp32 (input conv), xyz [B*n, 3]; fps_idx [B,S], knn_idx [B,S,k] int32.
// 'center' mode: gmean[b,s,c] = mean over the k neighbours of the grouped row [feature | xyz]   (layers.py:307-308)
__global__ __launch_bounds__(256) void group_mean_kernel(const float* __restrict__ feat, const float* __restrict__ xyz,
                                                         const int* __restrict__ knn_idx, int n, int S, int k, int D,
                                                         float* __restrict__ gmean) {
    const int b = blockIdx.y;
    const long total = (long)S * (D + 3);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int sidx = (int)(i / (D + 3)), c = (int)(i % (D + 3));
        const int* nb = knn_idx + ((long)b * S + sidx) * k;
        float acc = 0.f;
        for (int j = 0; j < k; ++j) {
            const long pi = (long)b * n + nb[j];
            acc += c < D ? feat[pi * D + c] : xyz[pi * 3 + (c - D)];
        }
        gmean[((long)b * S + sidx) * (D + 3) + c] = acc / (float)k;
    }
}

// The value a grouped element is measured from: the centre point's own [feature | xyz] ('anchor', layers.py:309-311)
// or the group's mean ('center').
template <bool CENTER>
__device__ __forceinline__ float group_origin(const float* __restrict__ fa, const float* __restrict__ xyz_a,
                                              const float* __restrict__ gm, int c, int D) {
    if (CENTER) return gm[c];
    return c < D ? fa[c] : xyz_a[c - D];
}

template <bool CENTER>
__global__ __launch_bounds__(256) void group_stats_kernel(const float* __restrict__ feat, const float* __restrict__ xyz,
                                                          const int* __restrict__ fps_idx, const int* __restrict__ knn_idx,
                                                          const float* __restrict__ gmean,
                                                          int n, int S, int k, int D, double* __restrict__ stats) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const long rows = (long)S * k;
    double s1 = 0.0, s2 = 0.0;
    for (long r = blockIdx.x * 4L + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4L) {
        const int sidx = (int)(r / k);
        const int ci = fps_idx[(long)b * S + sidx];
        const int pi = knn_idx[((long)b * S + sidx) * k + (r % k)];
        const float* fg = feat + ((long)b * n + pi) * D;
        const float* fa = feat + ((long)b * n + ci) * D;
        const float* xa = xyz + ((long)b * n + ci) * 3;
        const float* gm = CENTER ? gmean + ((long)b * S + sidx) * (D + 3) : nullptr;
        for (int c = lane; c < D + 3; c += 64) {
            const float g = c < D ? fg[c] : xyz[((long)b * n + pi) * 3 + (c - D)];
            const float d = g - group_origin<CENTER>(fa, xa, gm, c, D);
            s1 += (double)d; s2 += (double)d * (double)d;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) { fx_atomic_add(&stats[2 * b], s1); fx_atomic_add(&stats[2 * b + 1], s2); }    // order-independent (common.h)
}

// 'anchor' statistics, vector form (D % 4 == 0, D / 4 <= 32: the shipped 128- and 64-channel groupers): one wave per group of
// k neighbour rows; a row is read as 16 B per lane by D/4 lanes (64 / (D/4) rows in flight per wave), the xyz tail by lanes
// 0..2 of each row group; differences and their squares are summed in fp32 over the <= k/rows-in-flight rows a lane sees
// (a few dozen terms) and only the per-group totals go through fp64 — the scalar kernel's per-element fp64 converts / adds
// and its dependent index -> address -> load chain per row were the 611 us (128 clouds) of profiles/r01_c4_*.
template <int D4>
__global__ __launch_bounds__(256) void group_stats_vec_kernel(const float* __restrict__ feat, const float* __restrict__ xyz,
                                                              const int* __restrict__ fps_idx, const int* __restrict__ knn_idx,
                                                              int n, int S, int k, double* __restrict__ stats) {
    constexpr int D = D4 * 4, RPW = 64 / D4;                     // rows in flight per wave
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int sub = lane / D4, l = lane % D4;                    // row slot of this lane, its 4-channel chunk
    double t1 = 0.0, t2 = 0.0;
    for (int sidx = blockIdx.x * 4 + (threadIdx.x >> 6); sidx < S; sidx += gridDim.x * 4) {
        const int ci = fps_idx[(long)b * S + sidx];
        const f32x4 fa = *reinterpret_cast<const f32x4*>(feat + ((long)b * n + ci) * D + l * 4);
        const float xa = l < 3 ? xyz[((long)b * n + ci) * 3 + l] : 0.f;
        const int* nb = knn_idx + ((long)b * S + sidx) * k;
        float s1 = 0.f, s2 = 0.f;
        for (int j = sub; j < k; j += RPW) {
            const int pi = nb[j];
            const f32x4 g = *reinterpret_cast<const f32x4*>(feat + ((long)b * n + pi) * D + l * 4);
            const float gx = l < 3 ? xyz[((long)b * n + pi) * 3 + l] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = g[r] - fa[r]; s1 += d; s2 = fmaf(d, d, s2); }
            const float dx = gx - xa;                            // (0 for lanes >= 3)
            s1 += dx; s2 = fmaf(dx, dx, s2);
        }
        t1 += (double)s1; t2 += (double)s2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
    if (lane == 0) { fx_atomic_add(&stats[2 * b], t1); fx_atomic_add(&stats[2 * b + 1], t2); }    // order-independent (common.h)
}

template <bool CENTER>
__global__ __launch_bounds__(256) void group_build_kernel(const float* __restrict__ feat, const float* __restrict__ xyz,
                                                          const int* __restrict__ fps_idx, const int* __restrict__ knn_idx,
                                                          const float* __restrict__ alpha, const float* __restrict__ beta,
                                                          const double* __restrict__ stats, const float* __restrict__ gmean,
                                                          int n, int S, int k, int D, bf16_t* __restrict__ U, int ldu) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const long rows = (long)S * k;
    const double cnt = (double)rows * (D + 3);
    const double mean = fx_load(&stats[2 * b]) / cnt;
    const double var = (fx_load(&stats[2 * b + 1]) - cnt * mean * mean) / (cnt - 1.0);
    const float inv = 1.0f / ((float)sqrt(var > 0.0 ? var : 0.0) + 1e-5f);
    for (long r = blockIdx.x * 4L + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4L) {
        const int sidx = (int)(r / k);
        const int ci = fps_idx[(long)b * S + sidx];
        const int pi = knn_idx[((long)b * S + sidx) * k + (r % k)];
        const float* fg = feat + ((long)b * n + pi) * D;
        const float* fa = feat + ((long)b * n + ci) * D;
        const float* xa = xyz + ((long)b * n + ci) * 3;
        const float* gm = CENTER ? gmean + ((long)b * S + sidx) * (D + 3) : nullptr;
        bf16_t* u = U + ((long)b * rows + r) * ldu;
        for (int c = lane; c < ldu; c += 64) {
            float v = 0.f;
            if (c < D + 3) {
                const float g = c < D ? fg[c] : xyz[((long)b * n + pi) * 3 + (c - D)];
                v = alpha[c] * ((g - group_origin<CENTER>(fa, xa, gm, c, D)) * inv) + beta[c];
            } else if (c < 2 * D + 3) {
                v = fa[c - (D + 3)];
            }
            u[c] = (bf16_t)v;
        }
    }
}

// per-sample sums of (g - anchor) and its square over all grouped rows ('anchor' normalisation): stats[2b], stats[2b+1]
int ldt_group_stats_launch(const float* feat, const float* xyz, const int* fps_idx, const int* knn_idx, double* stats,
                           int B, int n, int S, int k, int D, hipStream_t s) {
    LDT_REQUIRE(B > 0 && n > 0 && S > 0 && k > 0 && D > 0, LDT_ESHAPE, "group_stats: bad shape");
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 2 * B, s);
    if (e != hipSuccess) { ldt_set_error("group: memset: %s", hipGetErrorString(e)); return (int)e; }
    const long rows = (long)S * k;
    int bx = (int)((rows + 3) / 4); if (bx > 256) bx = 256;
    int gx = (S + 3) / 4; if (gx > 64) gx = 64;
    if (D == 128 && ldt_aligned16(feat)) hipLaunchKernelGGL(group_stats_vec_kernel<32>, dim3(gx, B), dim3(256), 0, s, feat, xyz, fps_idx, knn_idx, n, S, k, stats);
    else if (D == 64 && ldt_aligned16(feat)) hipLaunchKernelGGL(group_stats_vec_kernel<16>, dim3(gx, B), dim3(256), 0, s, feat, xyz, fps_idx, knn_idx, n, S, k, stats);
    else hipLaunchKernelGGL(group_stats_kernel<false>, dim3(bx, B), dim3(256), 0, s, feat, xyz, fps_idx, knn_idx, nullptr, n, S, k, D, stats);
    return ldt_check_launch("group_stats");
}

int ldt_group_launch(const float* feat, const float* xyz, const int* fps_idx, const int* knn_idx, const float* alpha,
                     const float* beta, double* stats, int B, int n, int S, int k, int D, bf16_t* U, int ldu,
                     int center_mode, float* gmean, hipStream_t s) {
    LDT_REQUIRE(B > 0 && n > 0 && S > 0 && k > 0 && D > 0 && ldu >= 2 * D + 3, LDT_ESHAPE, "group: bad shape");
    LDT_REQUIRE(center_mode == 0 || (center_mode == 1 && gmean), LDT_EARG, "group: mode 1 ('center') needs the group-mean workspace");
    const long rows = (long)S * k;
    int bx = (int)((rows + 3) / 4); if (bx > 256) bx = 256;
    if (center_mode) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 2 * B, s);
        if (e != hipSuccess) { ldt_set_error("group: memset: %s", hipGetErrorString(e)); return (int)e; }
        long mb = ((long)S * (D + 3) + 255) / 256; if (mb > 1024) mb = 1024;
        hipLaunchKernelGGL(group_mean_kernel, dim3((unsigned)mb, B), dim3(256), 0, s, feat, xyz, knn_idx, n, S, k, D, gmean);
        TRY_LAUNCH("group_mean");
        hipLaunchKernelGGL(group_stats_kernel<true>, dim3(bx, B), dim3(256), 0, s, feat, xyz, fps_idx, knn_idx, gmean, n, S, k, D, stats);
        TRY_LAUNCH("group_stats");
        hipLaunchKernelGGL(group_build_kernel<true>, dim3(bx, B), dim3(256), 0, s, feat, xyz, fps_idx, knn_idx, alpha, beta, stats, gmean, n, S, k, D, U, ldu);
        return ldt_check_launch("group_build");
    }
    const int st = ldt_group_stats_launch(feat, xyz, fps_idx, knn_idx, stats, B, n, S, k, D, s);
    if (st != LDT_OK) return st;
    hipLaunchKernelGGL(group_build_kernel<false>, dim3(bx, B), dim3(256), 0, s, feat, xyz, fps_idx, knn_idx, alpha, beta, stats, nullptr, n, S, k, D, U, ldu);
    return ldt_check_launch("group_build");
}

// gather rows: out[b,s,:] = src[b, idx[b,s], :]   (index_points, layers.py:46-62) fp32
__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int n, int S, int C,
                                   float* __restrict__ out, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / C; const int c = (int)(i % C);
        const long b = row / S;
        out[i] = src[(b * n + idx[row]) * C + c];
    }
}
int ldt_gather_rows_launch(const float* src, const int* idx, int B, int n, int S, int C, float* out, hipStream_t s) {
    LDT_REQUIRE(B > 0 && n > 0 && S > 0 && C > 0, LDT_ESHAPE, "gather_rows: bad shape");
    const long total = (long)B * S * C;
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, idx, n, S, C, out, total);
    return ldt_check_launch("gather_rows");
}

// max over the middle axis: in [G][n][C] (bf16 or fp32) -> out fp32 [G][C]   (adaptive_max_pool1d :186; MiniPointnet max :97)
template <typename TI>
__global__ void maxpool_kernel(const TI* __restrict__ in, long ld, int n, int C, float* __restrict__ out, long G) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < G * C; i += (long)gridDim.x * blockDim.x) {
        const long g = i / C; const int c = (int)(i % C);
        float m = -INFINITY;
        for (int j = 0; j < n; ++j) m = fmaxf(m, (float)in[(g * n + j) * ld + c]);
        out[i] = m;
    }
}
int ldt_maxpool_launch(const void* in, int in_bf16, long ld, long G, int n, int C, float* out, hipStream_t s) {
    LDT_REQUIRE(G > 0 && n > 0 && C > 0 && ld >= C, LDT_ESHAPE, "maxpool: bad shape");
    long blocks = (G * C + 255) / 256; if (blocks > 4096) blocks = 4096;
    if (in_bf16) hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)in, ld, n, C, out, G);
    else hipLaunchKernelGGL(maxpool_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)in, ld, n, C, out, G);
    return ldt_check_launch("maxpool");
}

// ActNorm (model/layers.py:103-107, eval): y[b,t,c] = (x[b,t,c] - shift[t,c]) * exp(-log_scale[t,c]), in place
__global__ void actnorm_kernel(float* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ log_scale,
                               long total, long per_sample) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long j = i % per_sample;
        x[i] = (x[i] - shift[j]) * expf(-log_scale[j]);
    }
}
int ldt_actnorm_launch(float* x, const float* shift, const float* log_scale, long B, long per_sample, hipStream_t s) {
    LDT_REQUIRE(B > 0 && per_sample > 0, LDT_ESHAPE, "actnorm: bad shape");
    const long total = B * per_sample;
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(actnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, shift, log_scale, total, per_sample);
    return ldt_check_launch("actnorm");
}

// posterior draw (Network.py:26-29,75-77): mu = post[:, :z], logvar = clamp(post[:, z:], lo, hi);
// eps = mu + exp(logvar / 2) * noise   -> written into a strided slice of all_eps
__global__ void reparam_kernel(const float* __restrict__ post, const float* __restrict__ noise, float* __restrict__ out,
                               long ldo, float* __restrict__ mu_out, float* __restrict__ lv_out, long rows, int z, float lo, float hi) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rows * z; i += (long)gridDim.x * blockDim.x) {
        const long r = i / z; const int c = (int)(i % z);
        const float mu = post[r * 2 * z + c];
        const float lv = fminf(fmaxf(post[r * 2 * z + z + c], lo), hi);
        out[r * ldo + c] = mu + expf(lv / 2.f) * noise[i];
        if (mu_out) { mu_out[i] = mu; lv_out[i] = lv; }
    }
}
int ldt_reparam_launch(const float* post, const float* noise, float* out, long ldo, float* mu_out, float* lv_out,
                       long rows, int z, float lo, float hi, hipStream_t s) {
    LDT_REQUIRE(rows > 0 && z > 0 && ldo >= z, LDT_ESHAPE, "reparam: bad shape");
    LDT_REQUIRE((mu_out == nullptr) == (lv_out == nullptr), LDT_EARG, "reparam: mu/logvar outputs go together");
    long blocks = (rows * z + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(reparam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, post, noise, out, ldo, mu_out, lv_out, rows, z, lo, hi);
    return ldt_check_launch("reparam");
}

// Chamfer distance (evaluation/evaluation_metrics.py:23-33,88): dl[b,j] = min_i |a_i - b_j|^2, dr[b,i] = min_j |a_i - b_j|^2
// with the reference's expanded form |a|^2 + |b|^2 - 2 a.b.  One thread per query point.
__global__ void chamfer_min_kernel(const float* __restrict__ q, const float* __restrict__ ref, int nq, int nr, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float* qp = q + ((long)b * nq + i) * 3;
    const float x = qp[0], y = qp[1], z = qp[2];
    const float qn = fmaf(z, z, fmaf(y, y, x * x));
    float m = INFINITY;
    const float* rp = ref + (long)b * nr * 3;
    for (int j = 0; j < nr; ++j) {
        const float rx = rp[3 * j], ry = rp[3 * j + 1], rz = rp[3 * j + 2];
        const float rn = fmaf(rz, rz, fmaf(ry, ry, rx * rx));
        const float dot = fmaf(z, rz, fmaf(y, ry, x * rx));
        m = fminf(m, (qn + rn) - 2.f * dot);
    }
    out[(long)b * nq + i] = m;
}
int ldt_chamfer_launch(const float* a, const float* b, int B, int na, int nb, float* dl, float* dr, hipStream_t s) {
    LDT_REQUIRE(B > 0 && na > 0 && nb > 0, LDT_ESHAPE, "chamfer: bad shape");
    // dl (P.min(1)): for every point of b the nearest a ; dr (P.min(2)): for every point of a the nearest b
    hipLaunchKernelGGL(chamfer_min_kernel, dim3((nb + 255) / 256, B), dim3(256), 0, s, b, a, nb, na, dl);
    TRY_LAUNCH("chamfer_dl");
    hipLaunchKernelGGL(chamfer_min_kernel, dim3((na + 255) / 256, B), dim3(256), 0, s, a, b, na, nb, dr);
    return ldt_check_launch("chamfer_dr");
}

// ------------------------------------------------------------------------------------------------
// Compressor.norm_pts (Network.py:170-174, cfg.norm_input): per cloud and coordinate, (p - mean) / std with the unbiased
// standard deviation over the n points.  One 256-thread workgroup per cloud; sums in fp64 (n is a few thousand).
__global__ __launch_bounds__(256) void norm_points_kernel(const float* __restrict__ xyz, int n, float* __restrict__ out) {
    const float* p = xyz + (long)blockIdx.x * n * 3;
    float* o = out + (long)blockIdx.x * n * 3;
    double s[3] = {0.0, 0.0, 0.0}, q[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
        for (int c = 0; c < 3; ++c) { const double v = p[3 * i + c]; s[c] += v; q[c] += v * v; }
    __shared__ double red[4][6];
    __shared__ float mean_s[3], inv_s[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { s[c] += __shfl_xor(s[c], o2, 64); q[c] += __shfl_xor(q[c], o2, 64); }
    }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[threadIdx.x >> 6][c] = s[c]; red[threadIdx.x >> 6][3 + c] = q[c]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        const double ss = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        const double qq = (red[0][3 + c] + red[1][3 + c]) + (red[2][3 + c] + red[3][3 + c]);
        const double mean = ss / n;
        const double var = (qq - n * mean * mean) / (n - 1.0);
        mean_s[c] = (float)mean;
        inv_s[c] = (float)(1.0 / sqrt(var > 0.0 ? var : 0.0));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * n; i += 256) { const int c = i % 3; o[i] = (p[i] - mean_s[c]) * inv_s[c]; }
}
int ldt_norm_points_launch(const float* xyz, int B, int n, float* out, hipStream_t s) {
    LDT_REQUIRE(B > 0 && n > 1, LDT_ESHAPE, "norm_points: B=%d n=%d", B, n);
    hipLaunchKernelGGL(norm_points_kernel, dim3(B), dim3(256), 0, s, xyz, n, out);
    return ldt_check_launch("norm_points");
}

// InitialSet with max_outputs = None (Compressor/layers.py:17-24,38-42): seed rows drawn from a learned mixture,
//   x[r, d] = sum_m (eps[r, m, d] * sig[m, d] + mu[m, d]) * softmax(logits)[m],   r over B*N rows, m < n_mix (<= 8);
// the two Linear layers of `output` that follow are plain fp32 GEMMs.
__global__ __launch_bounds__(256) void mixture_seed_kernel(const float* __restrict__ eps, const float* __restrict__ sig, const float* __restrict__ mu,
                                                           const float* __restrict__ logits, int n_mix, int D, long rows, float* __restrict__ out) {
    float w[8];
    float mx = -INFINITY, den = 0.f;
    for (int m = 0; m < n_mix; ++m) mx = fmaxf(mx, logits[m]);
    for (int m = 0; m < n_mix; ++m) { w[m] = expf(logits[m] - mx); den += w[m]; }
    for (int m = 0; m < n_mix; ++m) w[m] /= den;
    const long total = rows * D;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / D; const int d = (int)(i % D);
        float acc = 0.f;
        for (int m = 0; m < n_mix; ++m) acc += (eps[(r * n_mix + m) * D + d] * sig[m * D + d] + mu[m * D + d]) * w[m];
        out[i] = acc;
    }
}
int ldt_mixture_seed_launch(const float* eps, const float* sig, const float* mu, const float* logits, int n_mix, int D, long rows, float* out, hipStream_t s) {
    LDT_REQUIRE(n_mix >= 1 && n_mix <= 8 && D > 0 && rows > 0, LDT_ESHAPE, "mixture_seed: n_mix=%d (<= 8) D=%d rows=%ld", n_mix, D, rows);
    long blocks = (rows * D + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mixture_seed_kernel, dim3((unsigned)blocks), dim3(256), 0, s, eps, sig, mu, logits, n_mix, D, rows, out);
    return ldt_check_launch("mixture_seed");
}
