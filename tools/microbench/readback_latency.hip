// Round trip of a count read-back (tiny kernel -> 32-byte device-to-host copy -> wait) on one stream while OTHER streams
// keep the chip busy with kernels of a given length.  (do the ~30 blocking read-backs of a cloud
// wait for the other clouds' chip-filling kernels?)
//   hipcc --offload-arch=gfx950 -O2 -o readback_latency readback_latency.hip -lpthread && ./readback_latency
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void k_busy(unsigned* p, int iters) {
    unsigned v = threadIdx.x + blockIdx.x;
    for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    if (v == 12345u) p[0] = v;
}
__global__ void k_count(unsigned* p) { if (threadIdx.x == 0) p[0] += 1; }

int main() {
    unsigned *busy_buf, *cnt;
    (void)hipMalloc(&busy_buf, 1 << 20);
    (void)hipMalloc(&cnt, 256);
    (void)hipMemset(cnt, 0, 256);
    for (int background : {0, 1, 3, 7})
        for (int iters : {2000, 20000, 100000}) {  // length of one background kernel
            if (background == 0 && iters != 2000) continue;
            std::atomic<bool> stop{false};
            std::vector<std::thread> bg;
            for (int t = 0; t < background; t++)
                bg.emplace_back([&] {
                    hipStream_t s;
                    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                    while (!stop.load()) {
                        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k_busy, dim3(256 * 8), dim3(256), 0, s, busy_buf, iters);
                        (void)hipStreamSynchronize(s);
                    }
                    (void)hipStreamDestroy(s);
                });
            hipStream_t s;
            (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            // length of one background kernel, measured alone
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
            const int N = 300;
            unsigned h[8];
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(k_count, dim3(1), dim3(64), 0, s, cnt);
                (void)hipMemcpyAsync(h, cnt, 32, hipMemcpyDeviceToHost, s);
                (void)hipStreamSynchronize(s);
            }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            stop.store(true);
            for (auto& x : bg) x.join();
            float ms = 0;
            (void)hipEventRecord(e0, s);
            hipLaunchKernelGGL(k_busy, dim3(256 * 8), dim3(256), 0, s, busy_buf, iters);
            (void)hipEventRecord(e1, s);
            (void)hipStreamSynchronize(s);
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("background streams=%d (chip-filling kernels of %7.1f us each): read-back round trip %7.1f us\n", background,
                   1e3 * ms, 1e6 * dt / N);
            (void)hipStreamDestroy(s);
        }
    return 0;
}
