// Farthest point sampling, one wave per cloud (the many-clouds form of pointops.hip's fps_kernel; same semantics:
// model/functional/src/sampling/sampling.cu:86-167, start index 0, 1e38 initial distances, unfused d = (dx*dx + dy*dy) + dz*dz, ties
// to the smaller (k % 512, k / 512)).  This file is compiled with -fno-slp-vectorize (csrc/build.sh): hipcc otherwise packs the x / y
// halves of every point's distance into v_pk_* pairs, and the even-aligned register pairs that needs turn 128 VGPRs of point state
// into > 256 (8,000 spilled registers at 32 points per lane).
#include "kernels.h"

struct Cand { float d; int k; };
__device__ __forceinline__ bool beats(float da, int ka, float db, int kb) {
    // rank of a point in the twin's 512-thread layout = (k & 511, k >> 9)
    if (da != db) return da > db;
    const int ra = ((ka & 511) << 16) | (ka >> 9), rb = ((kb & 511) << 16) | (kb >> 9);
    return ra < rb;
}

// One WAVE per cloud (n <= 64 * PPL points, point k = lane + 64 j in registers): no LDS, no workgroup barrier on the serial chain
// of m - 1 iterations — the 512-thread form above spends most of an iteration in its two barriers and the LDS hop when many clouds
// are in flight (0.93 ms per 1024 clouds of 2048 points: 3.7 us per iteration for the four clouds of a CU).  Same arithmetic and the
// same tie rule (`beats`: the twin's 512-thread layout ranks a point by (k % 512, k / 512)); the wave maximum is a 6-step butterfly,
// the winner's coordinates are read from its lane by v_readlane.
template <int PPL>
__global__ __launch_bounds__(256) void fps_wave_kernel(const float* __restrict__ xyz, int B, int n, int m, int skip_near_origin, int* __restrict__ idx_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;                                          // whole wave
    const float* p = xyz + (long)b * n * 3;
    float px[PPL], py[PPL], pz[PPL], dist[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int k = lane + 64 * j;
        const bool in = k < n;
        px[j] = in ? p[3 * k] : 0.f; py[j] = in ? p[3 * k + 1] : 0.f; pz[j] = in ? p[3 * k + 2] : 0.f;
        float mag;
        {
#pragma clang fp contract(off)
            mag = (px[j] * px[j] + py[j] * py[j]) + pz[j] * pz[j];
        }
        // a point that is not live (past n, or skipped) carries the running distance -2: min(d, -2) = -2 never beats best = -1
        dist[j] = (in && !(skip_near_origin && mag <= 1e-3f)) ? 1e38f : -2.f;
    }
    const float p0x = p[0], p0y = p[1], p0z = p[2];
    float x1 = p0x, y1 = p0y, z1 = p0z;
    if (lane == 0) idx_out[(long)b * m] = 0;
    for (int it = 1; it < m; ++it) {
        // the lane's own points are visited in the tie rule's rank order ((k % 512, k / 512) for k = lane + 64 j: j & 7 first, then
        // j >> 3), so a strict '>' keeps the right one among equal distances and no per-point rank is needed
        float best = -1.f; int bestj = 0;
        float bx = 0.f, by = 0.f, bz = 0.f;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int a8 = 0; a8 < (PPL < 8 ? PPL : 8); ++a8)
#pragma unroll
                for (int c8 = 0; c8 < (PPL + 7) / 8; ++c8) {
                    const int j = c8 * 8 + a8;
                    if (j >= PPL) continue;
                    const float dx = px[j] - x1, dy = py[j] - y1, dz = pz[j] - z1;
                    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                    const float s2 = xx + yy;
                    const float d = s2 + zz;
                    const float d2 = fminf(d, dist[j]);
                    dist[j] = d2;
                    const bool win = d2 > best;
                    best = win ? d2 : best; bestj = win ? j : bestj;
                    bx = win ? px[j] : bx; by = win ? py[j] : by; bz = win ? pz[j] : bz;
                }
        }
        const int besti = lane + 64 * bestj;
        float wd = best; int wk = besti;                         // wave argmax under `beats`
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(wd, o, 64);
            const int ok = __shfl_xor(wk, o, 64);
            if (beats(od, ok, wd, wk)) { wd = od; wk = ok; }
        }
        const int wlane = __builtin_amdgcn_readfirstlane(wk) & 63;
        const bool none = __builtin_amdgcn_readfirstlane(__float_as_int(wd)) == __float_as_int(-1.f);   // no live point anywhere: index 0
        x1 = none ? p0x : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx), wlane));
        y1 = none ? p0y : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(by), wlane));
        z1 = none ? p0z : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bz), wlane));
        if (lane == 0) idx_out[(long)b * m + it] = __builtin_amdgcn_readfirstlane(wk);
    }
}

int ldt_fps_wave_launch(const float* xyz, int B, int n, int m, int skip_near_origin, int* idx, hipStream_t s) {
    LDT_REQUIRE(B > 0 && n > 0 && n <= 64 * 32 && m > 0 && m <= n, LDT_ESHAPE, "fps_wave: B=%d n=%d m=%d", B, n, m);
    hipLaunchKernelGGL(fps_wave_kernel<32>, dim3((B + 3) / 4), dim3(256), 0, s, xyz, B, n, m, skip_near_origin, idx);
    return ldt_check_launch("fps_wave");
}
