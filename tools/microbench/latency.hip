// Dependent-load / atomic latency on gfx950 at different memory scopes (developer calibration aid).
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/latency tools/microbench/latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

template <int SCOPE>
__global__ void chase(const unsigned* next, int steps, unsigned* out, long long* cycles) {
    unsigned p = threadIdx.x;
    const long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) {
        if (SCOPE == 0) p = __builtin_nontemporal_load(&next[p]) ;
        else if (SCOPE == 1) p = next[p];
        else if (SCOPE == 2) p = __hip_atomic_load(&next[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 3) p = __hip_atomic_load(&next[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p = __hip_atomic_load(&next[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <int SCOPE>
__global__ void atomchase(unsigned* cells, int steps, unsigned* out, long long* cycles) {
    unsigned p = threadIdx.x;
    const long long t0 = wall_clock64();
    for (int i = 0; i < steps; i++) {
        // returning atomic whose result feeds the next address
        unsigned old;
        if (SCOPE == 2) old = __hip_atomic_fetch_or(&cells[p], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else old = __hip_atomic_fetch_or(&cells[p], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p = old;
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0); printf("wall clock rate %d kHz\n", rate);
    const int steps = 20000;
    for (size_t n : {size_t(1) << 12, size_t(1) << 18, size_t(1) << 22, size_t(1) << 26}) {  // 16 KB, 1 MB, 16 MB, 256 MB
        std::vector<unsigned> perm(n), next(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (size_t i = 0; i < n; i++) next[perm[i]] = perm[(i + 1) % n];
        unsigned *d, *out; long long* cyc;
        hipMalloc(&d, n * 4); hipMalloc(&out, 256); hipMalloc(&cyc, 8);
        hipMemcpy(d, next.data(), n * 4, hipMemcpyHostToDevice);
        long long h; float ms; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const char* names[] = {"nontemporal", "plain", "workgroup", "agent", "system"};
        printf("array %8.1f KB:", n * 4 / 1024.0);
#define RUN(S) hipLaunchKernelGGL(chase<S>, dim3(1), dim3(1), 0, 0, d, steps, out, cyc); hipEventRecord(e0); hipLaunchKernelGGL(chase<S>, dim3(1), dim3(1), 0, 0, d, steps, out, cyc); hipEventRecord(e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); hipEventElapsedTime(&ms, e0, e1); printf("  %s %.0f ns (event %.0f)", names[S], h * 10.0 / steps, ms * 1e6 / steps);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
        hipLaunchKernelGGL(atomchase<2>, dim3(1), dim3(1), 0, 0, d, steps, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("  atomic-wg %.0f ns", h * 10.0 / steps);
        hipLaunchKernelGGL(atomchase<3>, dim3(1), dim3(1), 0, 0, d, steps, out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("  atomic-agent %.0f ns\n", h * 10.0 / steps);
        hipFree(d); hipFree(out); hipFree(cyc);
    }
    return 0;
}
