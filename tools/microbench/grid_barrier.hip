// Developer calibration aid: cost of a grid barrier on gfx950 (persistent kernel, all workgroups resident).
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/grid_barrier tools/microbench/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned ld_u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// variant bits: 1 = __threadfence on both sides, 2 = hierarchical (groups of 32), 4 = poll with s_sleep, 8 = s_sleep longer
template <int V>
__global__ void k_bar(unsigned* bar, int rounds) {
    const unsigned G = gridDim.x;
    for (int r = 0; r < rounds; r++) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (V & 1) __threadfence();
            unsigned want;
            if (V & 2) {
                const unsigned ng = (G + 31) / 32, gi = blockIdx.x / 32, gs = gi + 1 == ng ? G - gi * 32 : 32;
                const unsigned old = atomicAdd(&bar[16 * (1 + gi)], 1u);
                if (old % gs == gs - 1) atomicAdd(&bar[0], 1u);
                want = ng * (r + 1);
            } else {
                atomicAdd(&bar[0], 1u);
                want = G * (r + 1);
            }
            while (ld_u(&bar[0]) < want) {
                if (V & 4) __builtin_amdgcn_s_sleep(2);
                if (V & 8) __builtin_amdgcn_s_sleep(16);
            }
            if (V & 1) __threadfence();
        }
        __syncthreads();
    }
}

template <int V>
void run(const char* name, int G, unsigned* bar) {
    const int rounds = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(bar, 0, 16 * 64 * sizeof(unsigned));
    hipLaunchKernelGGL(k_bar<V>, dim3(G), dim3(256), 0, 0, bar, 10);
    hipDeviceSynchronize();
    hipMemset(bar, 0, 16 * 64 * sizeof(unsigned));
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_bar<V>, dim3(G), dim3(256), 0, 0, bar, rounds);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("G=%4d %-40s %.2f us per barrier\n", G, name, ms * 1e3 / rounds);
}

int main() {
    unsigned* bar;
    hipMalloc(&bar, 16 * 64 * sizeof(unsigned));
    for (int G : {64, 256, 1024}) {
        run<0>("flat, no fence, tight poll", G, bar);
        run<4>("flat, no fence, sleep 2", G, bar);
        run<2>("groups, no fence, tight poll", G, bar);
        run<6>("groups, no fence, sleep 2", G, bar);
        run<10>("groups, no fence, sleep 16", G, bar);
        run<7>("groups, fences, sleep 2", G, bar);
        run<3>("groups, fences, tight poll", G, bar);
        run<1>("flat, fences, tight poll", G, bar);
    }
    return 0;
}
