// read-back cost probe: a small kernel chain, then the host needs 8 bytes the last kernel produced
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_work(unsigned* x, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] += 1u; }
__global__ void k_count(const unsigned* x, unsigned long long* out) { if (threadIdx.x == 0) *out = x[0]; }
__global__ void k_count_host(const unsigned* x, volatile unsigned long long* out, volatile unsigned* flag, unsigned seq) {
    if (threadIdx.x == 0) { *out = x[0]; __threadfence_system(); *flag = seq; }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    unsigned* x; hipMalloc(&x, 1 << 22); hipMemset(x, 0, 1 << 22);
    unsigned long long* d; hipMalloc(&d, 8);
    unsigned long long* hp; hipHostMalloc(&hp, 64, hipHostMallocMapped);
    unsigned* flag = (unsigned*)(hp + 4); *flag = 0;
    unsigned long long hstack = 0;
    const int reps = 2000;
    for (int mode = 0; mode < 4; mode++) {
        double t_total = 0, t_after = 0;
        for (int r = 0; r < reps + 50; r++) {
            const double t0 = now();
            for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, x, 1 << 20);
            if (mode == 0) { hipLaunchKernelGGL(k_count, dim3(1), dim3(64), 0, s, x, d); hipMemcpyAsync(&hstack, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
            if (mode == 1) { hipLaunchKernelGGL(k_count, dim3(1), dim3(64), 0, s, x, d); hipMemcpyAsync(hp, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
            if (mode == 2) { hipLaunchKernelGGL(k_count_host, dim3(1), dim3(64), 0, s, x, hp, flag, (unsigned)(r + 1)); hipStreamSynchronize(s); }
            if (mode == 3) { hipLaunchKernelGGL(k_count_host, dim3(1), dim3(64), 0, s, x, hp, flag, (unsigned)(r + 1)); while (*(volatile unsigned*)flag != (unsigned)(r + 1)) {} }
            const double t1 = now();
            // the next stage's first kernel: how long until the stream is busy again is part of the cost
            hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, x, 1 << 20);
            const double t2 = now();
            if (r >= 50) { t_total += t1 - t0; t_after += t2 - t1; }
        }
        hipStreamSynchronize(s);
        const char* names[] = {"memcpyAsync to pageable + sync", "memcpyAsync to pinned + sync", "kernel stores to mapped host + sync", "kernel stores to mapped host + host polls flag"};
        printf("%-48s: %.1f us per chain+read-back, next launch call %.1f us\n", names[mode], t_total / reps, t_after / reps);
    }
    // the chain alone (no read-back): the floor
    { double t = 0; for (int r = 0; r < reps + 50; r++) { const double t0 = now(); for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, x, 1 << 20); hipStreamSynchronize(s); if (r >= 50) t += now() - t0; }
      printf("%-48s: %.1f us\n", "4 kernels + sync (no copy)", t / reps); }
    return 0;
}
