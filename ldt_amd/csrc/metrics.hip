// Generation-quality metrics of the reference's validation loop (evaluation/evaluation_metrics.py:112-277) as gfx950
// kernels: all-pairs Chamfer distance and the approximate-matching EMD.  VALU / LDS-broadcast bound (3-D points: no
// GEMM shape worth MFMA), one workgroup per cloud pair, both clouds resident in LDS, nothing materialised in HBM.
#include "common.h"
#include "kernels.h"

#define TRY_LAUNCH(what) do { const int _rc = ldt_check_launch(what); if (_rc != LDT_OK) return _rc; } while (0)

// ---------------------------------------------------------------------------------------------------------------
// All-pairs Chamfer: cd[s][r] = mean_j min_i P + mean_i min_j P with P[i][j] = |x_i|^2 + |y_j|^2 - 2 x_i.y_j — the
// quantity `_pairwise_CD_` / `_pairwise_EMD_CD_` build from distChamfer (evaluation_metrics.py:23-33,88,141,186):
// (dl.mean(1) + dr.mean(1)) for the sample cloud x[s] against every reference cloud y[r].
// One workgroup per (s, r); each direction streams the "reference" cloud through a 1024-point LDS chunk of
// (x, y, z, |p|^2) that every lane reads as a broadcast, QPT query points per lane in registers.
constexpr int CDP_CHUNK = 1024;
constexpr int CDP_QPT = 4;

__device__ __forceinline__ float cdp_direction(const float* __restrict__ q, int nq, const float* __restrict__ ref, int nr,
                                               float4* __restrict__ buf, float* __restrict__ red) {
    const int tid = threadIdx.x;
    float total = 0.f;
    for (int q0 = 0; q0 < nq; q0 += 256 * CDP_QPT) {
        float qx[CDP_QPT], qy[CDP_QPT], qz[CDP_QPT], qn[CDP_QPT], mn[CDP_QPT];
#pragma unroll
        for (int u = 0; u < CDP_QPT; ++u) {
            const int i = q0 + u * 256 + tid;
            const bool ok = i < nq;
            qx[u] = ok ? q[3 * i] : 0.f; qy[u] = ok ? q[3 * i + 1] : 0.f; qz[u] = ok ? q[3 * i + 2] : 0.f;
            qn[u] = fmaf(qz[u], qz[u], fmaf(qy[u], qy[u], qx[u] * qx[u]));
            mn[u] = INFINITY;
        }
        for (int c0 = 0; c0 < nr; c0 += CDP_CHUNK) {
            const int cn = min(CDP_CHUNK, nr - c0);
            __syncthreads();
            for (int j = tid; j < cn; j += 256) {
                const float rx = ref[3 * (c0 + j)], ry = ref[3 * (c0 + j) + 1], rz = ref[3 * (c0 + j) + 2];
                buf[j] = make_float4(rx, ry, rz, fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
            }
            __syncthreads();
            for (int j = 0; j < cn; ++j) {
                const float4 r = buf[j];
#pragma unroll
                for (int u = 0; u < CDP_QPT; ++u) {
                    const float dot = fmaf(qz[u], r.z, fmaf(qy[u], r.y, qx[u] * r.x));
                    mn[u] = fminf(mn[u], (qn[u] + r.w) - 2.f * dot);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CDP_QPT; ++u)
            if (q0 + u * 256 + tid < nq) total += mn[u];
    }
    total = wave_sum(total);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = total;
    __syncthreads();
    return ((red[0] + red[1]) + (red[2] + red[3])) / (float)nq;
}

__global__ __launch_bounds__(256) void chamfer_pairwise_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               int n, int m, int R, float* __restrict__ cd) {
    __shared__ float4 buf[CDP_CHUNK];
    __shared__ float red[4];
    const int s = blockIdx.y, r = blockIdx.x;
    const float* xs = x + (long)s * n * 3;
    const float* yr = y + (long)r * m * 3;
    const float dl = cdp_direction(yr, m, xs, n, buf, red);      // P.min(1): nearest x for every y_j
    const float dr = cdp_direction(xs, n, yr, m, buf, red);      // P.min(2): nearest y for every x_i
    if (threadIdx.x == 0) cd[(long)s * R + r] = dl + dr;
}

int ldt_chamfer_pairwise_launch(const float* x, const float* y, int S, int R, int n, int m, float* cd, hipStream_t st) {
    LDT_REQUIRE(S > 0 && R > 0 && n > 0 && m > 0 && S <= 65535, LDT_ESHAPE, "chamfer_pairwise: bad shape S=%d R=%d n=%d m=%d", S, R, n, m);
    hipLaunchKernelGGL(chamfer_pairwise_kernel, dim3(R, S), dim3(256), 0, st, x, y, n, m, R, cd);
    return ldt_check_launch("chamfer_pairwise");
}

// ---------------------------------------------------------------------------------------------------------------
// Approximate-matching EMD (evaluation/pytorch_structural_losses/src/approxmatch.cu:3-186 approxmatchkernel followed by
// :188-224 matchcostkernel; Python side emd_approx_cuda, evaluation_metrics.py:40-46).  Per pair (xyz1: n points,
// xyz2: m points), nine annealing levels j = 7..-1 with level = -4^j:
//     ratioL[k] = remainL[k] / (1e-9 + sum_l exp(level d2(k,l)) remainR[l])
//     sumr[l]   = remainR[l] * sum_k exp(level d2) ratioL[k];  ratioR[l] = min(remainR[l]/(sumr+1e-9), 1) * remainR[l];
//     remainR[l] = max(0, remainR[l] - sumr)
//     w(k,l)    = exp(level d2) ratioL[k] ratioR[l];  match[l][k] += w;  remainL[k] = max(0, remainL[k] - sum_l w)
// and cost = sum_{k,l} match[l][k] * sqrt(d2(k,l)).  `match` only ever accumulates, so the cost is accumulated level by
// level here and the n x m match matrix (16 MB per pair at 2048 points, written and re-read by the reference) never
// exists.  One workgroup of 1024 lanes per pair; both clouds and the four marginals live in LDS; a lane owns point k
// (passes 1 and 3) or l (pass 2) and walks the other cloud through LDS broadcasts, in the reference's summation order.
__global__ __launch_bounds__(1024) void emd_approx_kernel(const float* __restrict__ x, const float* __restrict__ y, int S, int R,
                                                          int n, int m, int pairwise, float* __restrict__ out) {
    extern __shared__ float4 emd_smem[];
    float4* P1 = emd_smem;                 // (x, y, z, ratioL)   of xyz1
    float4* P2 = P1 + n;                   // (x, y, z, remainR)  of xyz2
    float* remainL = reinterpret_cast<float*>(P2 + m);
    float* ratioR = remainL + n;
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const long npairs = pairwise ? (long)S * R : S;
    const float multiL = n >= m ? 1.f : (float)(m / n);        // integer ratios, approxmatch.cu:6-12
    const float multiR = n >= m ? (float)(n / m) : 1.f;
    for (long p = blockIdx.x; p < npairs; p += gridDim.x) {
        const float* x1 = x + (pairwise ? p / R : p) * (long)n * 3;
        const float* x2 = y + (pairwise ? p % R : p) * (long)m * 3;
        __syncthreads();
        for (int k = tid; k < n; k += 1024) { P1[k] = make_float4(x1[3 * k], x1[3 * k + 1], x1[3 * k + 2], 0.f); remainL[k] = multiL; }
        for (int l = tid; l < m; l += 1024) { P2[l] = make_float4(x2[3 * l], x2[3 * l + 1], x2[3 * l + 2], multiR); ratioR[l] = 0.f; }
        __syncthreads();
        float cost = 0.f;
        for (int j = 7; j > -2; --j) {
            const float level = -powf(4.0f, (float)j);
            // pass 1 (:27-58): ratioL
            for (int k = tid; k < n; k += 1024) {
                const float4 a = P1[k];
                float suml = 1e-9f;
                for (int l = 0; l < m; ++l) {
                    const float4 b = P2[l];
                    const float d = level * ((b.x - a.x) * (b.x - a.x) + (b.y - a.y) * (b.y - a.y) + (b.z - a.z) * (b.z - a.z));
                    suml += __expf(d) * b.w;
                }
                P1[k].w = remainL[k] / suml;
            }
            __syncthreads();
            // pass 2 (:74-108): ratioR, remainR  (each lane rewrites only its own P2[l].w after reading all of P1)
            for (int l = tid; l < m; l += 1024) {
                const float4 b = P2[l];
                float sumr = 0.f;
                for (int k = 0; k < n; ++k) {
                    const float4 a = P1[k];
                    sumr += __expf(level * ((b.x - a.x) * (b.x - a.x) + (b.y - a.y) * (b.y - a.y) + (b.z - a.z) * (b.z - a.z))) * a.w;
                }
                sumr *= b.w;
                const float consumption = fminf(b.w / (sumr + 1e-9f), 1.0f);
                ratioR[l] = consumption * b.w;
                P2[l].w = fmaxf(0.0f, b.w - sumr);
            }
            __syncthreads();
            // pass 3 (:125-160): the matching mass of this level, its transport cost, remainL
            for (int k = tid; k < n; k += 1024) {
                const float4 a = P1[k];
                float suml = 0.f;
                for (int l = 0; l < m; ++l) {
                    const float4 b = P2[l];
                    const float d2 = (b.x - a.x) * (b.x - a.x) + (b.y - a.y) * (b.y - a.y) + (b.z - a.z) * (b.z - a.z);
                    const float w = __expf(level * d2) * a.w * ratioR[l];
                    cost = fmaf(w, sqrtf(d2), cost);            // matchcostkernel :206-207
                    suml += w;
                }
                remainL[k] = fmaxf(0.0f, remainL[k] - suml);
            }
            __syncthreads();
        }
        cost = wave_sum(cost);
        if ((tid & 63) == 0) red[tid >> 6] = cost;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < 16; ++w) t += red[w];
            out[p] = t;
        }
    }
}

int ldt_emd_approx_launch(const float* x, const float* y, int S, int R, int n, int m, int pairwise, float* out, hipStream_t st) {
    LDT_REQUIRE(S > 0 && R > 0 && n > 0 && m > 0, LDT_ESHAPE, "emd_approx: bad shape");
    LDT_REQUIRE(pairwise || S == R, LDT_ESHAPE, "emd_approx: batched mode pairs cloud b with cloud b (S=%d != R=%d)", S, R);
    const size_t lds = (size_t)(n + m) * (sizeof(float4) + sizeof(float));
    LDT_REQUIRE(lds <= 150 * 1024, LDT_ESHAPE, "emd_approx: n + m = %d points exceed the LDS-resident limit (7680)", n + m);
    LDT_ENSURE_LDS(emd_approx_kernel, 150 * 1024, "emd_approx");
    const long npairs = pairwise ? (long)S * R : S;
    const int grid = (int)(npairs < 4096 ? npairs : 4096);
    hipLaunchKernelGGL(emd_approx_kernel, dim3(grid), dim3(1024), lds, st, x, y, S, R, n, m, pairwise, out);
    return ldt_check_launch("emd_approx");
}
