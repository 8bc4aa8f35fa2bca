// Calibration probe: dependent-load latency (pointer chase) at several footprints, shader clock,
// and empty-kernel launch/boundary cost.  Build: hipcc --offload-arch=gfx950 -O3 latency_probe.hip -o latency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void chase(const uint32_t* next, int steps, uint32_t start, uint32_t* out, unsigned long long* cyc, unsigned long long* rt) {
    uint32_t p = start;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < steps; ++i) p = next[p];
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { *out = p; *cyc = c1 - c0; *rt = r1 - r0; }
}
__global__ void empty_kernel(int* x) { if (x && threadIdx.x == 99999) *x = 1; }

int main() {
    uint32_t* d_out; unsigned long long *d_cyc, *d_rt;
    hipMalloc(&d_out, 4); hipMalloc(&d_cyc, 8); hipMalloc(&d_rt, 8);
    for (size_t n : {size_t(1) << 14, size_t(1) << 19, size_t(1) << 23, size_t(1) << 27}) {   // 64 KB, 2 MB, 32 MB, 512 MB
        std::vector<uint32_t> perm(n), next(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
        for (size_t i = 0; i < n; ++i) next[perm[i]] = perm[(i + 1) % n];
        uint32_t* d_next; hipMalloc(&d_next, n * 4); hipMemcpy(d_next, next.data(), n * 4, hipMemcpyHostToDevice);
        const int steps = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d_next, steps, 0u, d_out, d_cyc, d_rt);
            hipDeviceSynchronize();
        }
        unsigned long long cyc, rt; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&rt, d_rt, 8, hipMemcpyDeviceToHost);
        double ns = rt * 10.0;   // s_memrealtime ticks at 100 MHz
        printf("footprint %8.1f MB: %7.1f ns/load, %7.1f cycles/load, shader clock %.2f GHz\n", n * 4 / 1e6, ns / steps, (double)cyc / steps, cyc / ns);
        hipFree(d_next);
    }
    // launch / boundary cost
    hipStream_t st; hipStreamCreate(&st);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, (int*)nullptr);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    const int N = 2000;
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, (int*)nullptr);
    hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("empty kernel, back to back in one stream: %.2f us each\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    return 0;
}
