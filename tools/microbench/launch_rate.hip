// How many small kernels per second does one MI355X retire, from T host threads on T streams?
// (is the multi-stream pipeline rate bounded by launches, whoever issues them?)
//   hipcc --offload-arch=gfx950 -O2 -o launch_rate launch_rate.hip -lpthread && ./launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void k_tiny(unsigned* p, int work) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < work; i++) v = v * 1664525u + 1013904223u;
    if (v == 12345u) p[0] = v;
}

static double run(int T, int N, int blocks, int work, bool dependent_chain) {
    std::vector<hipStream_t> s(T);
    unsigned* buf;
    (void)hipMalloc(&buf, 1 << 20);
    for (auto& x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s[t], buf, work);
            (void)hipStreamSynchronize(s[t]);
        });
    for (auto& x : th) x.join();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (auto& x : s) (void)hipStreamDestroy(x);
    (void)hipFree(buf);
    (void)dependent_chain;
    return dt;
}

int main() {
    run(1, 1000, 1, 0, true);
    for (int blocks : {1, 1024})
        for (int T : {1, 2, 4, 8, 16}) {
            const int N = 20000;
            double dt = run(T, N, blocks, 0, true);
            printf("blocks=%4d threads/streams=%2d: %7.2f us per launch per stream, %8.0f launches/s in total\n", blocks, T,
                   1e6 * dt / N, T * N / dt);
        }
    return 0;
}
