// Probe: does an LDS-DMA load issued AFTER a burst of global stores land in LDS before those stores are acknowledged?
// (`s_waitcnt vmcnt` counts loads and stores together in issue order, so the counter cannot tell.)  Every wave of every CU
// issues NST 16-B-per-lane stores, then one `global_load_lds` into a slot pre-filled with a sentinel, then polls the slot.
// Prints, per stores-per-wave setting, the median cycles until (a) the data is visible in LDS, (b) vmcnt(0) returns.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbg/vmcnt_order_probe.hip -o /tmp/vprobe && /tmp/vprobe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NST>
__global__ __launch_bounds__(512) void probe(float* big, const float* small, long long* out) {
    __shared__ __attribute__((aligned(16))) unsigned slot[8 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned* my = slot + wave * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) my[lane * 4 + i] = 0xFFFFFFFFu;
    __syncthreads();
    float* dst = big + ((long)blockIdx.x * 512 + tid) * 4;
    const f32x4 v = {1.f, 2.f, 3.f, (float)tid};
    const long stride = (long)gridDim.x * 512 * 4;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < NST; ++i) *reinterpret_cast<f32x4*>(dst + i * stride) = v;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(small + (wave * 64 + lane) * 4),
                                     (__attribute__((address_space(3))) void*)my, 16, 0, 0);
    long long t1 = 0;
    for (int it = 0; it < 200000; ++it) {
        const unsigned x = *reinterpret_cast<volatile unsigned*>(my + lane * 4 + 3);
        if (__builtin_amdgcn_ballot_w64(x == 0xFFFFFFFFu) == 0) { t1 = __builtin_amdgcn_s_memtime(); break; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

template <int NST>
void run(float* big, float* small, long long* out, int grid) {
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe<NST>, dim3(grid), dim3(512), 0, 0, big, small, out);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(grid * 8 * 2);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<long long> a, b;
    for (int i = 0; i < grid * 8; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("stores/wave %3d (%4d KB per CU): data visible after %6lld cycles (median; p90 %6lld), vmcnt(0) after %6lld (p90 %6lld)  [shader clock cycles]\n",
           NST, NST * 8, a[a.size() / 2], a[a.size() * 9 / 10], b[b.size() / 2], b[b.size() * 9 / 10]);
}

int main() {
    const int grid = 256;
    float *big, *small; long long* out;
    hipMalloc(&big, (size_t)grid * 512 * 16 * 64);
    hipMalloc(&small, 512 * 16);
    hipMalloc(&out, grid * 8 * 2 * 8);
    std::vector<float> hs(512 * 4, 0.5f);
    hipMemcpy(small, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    run<0>(big, small, out, grid);
    run<4>(big, small, out, grid);
    run<16>(big, small, out, grid);
    run<32>(big, small, out, grid);
    run<64>(big, small, out, grid);
    return 0;
}
