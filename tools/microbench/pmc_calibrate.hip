// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in THIS path's access patterns (MI355X_MICROARCH.md, "HBM":
// FETCH_SIZE reports half the bytes of a wide coalesced streaming read; other access widths and WRITE_SIZE are uncalibrated --
// "calibrate on a known byte count in your own access pattern before trusting an absolute").  Every kernel below touches a
// known number of bytes exactly once, in buffers far larger than the 256 MB Infinity Cache:
//   k_stream16   coalesced 16 B per lane                                  (the guide's own case: expect 0.5)
//   k_stream4    coalesced 4 B per lane
//   k_gather<R>  rows of R bytes (32 / 64 / 128 / 256) by a random permutation, float4 per lane -- the sparse conv's gathers
//   k_scatter<R> the same rows written (the conv's output rows are coalesced; the skeleton's stamps are scattered)
//   k_atomic4    one 4-byte atomicMin per random word -- SSSP / claims / hash inserts
//   hipcc --offload-arch=gfx950 -O2 -o pmc_calibrate pmc_calibrate.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- ./pmc_calibrate
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- ./pmc_calibrate      (tools/pmc_calibrate.sh)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_stream16(const float4* __restrict__ in, int64_t n16, float* sink) {
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_stream4(const float* __restrict__ in, int64_t n4, float* sink) {
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 12345.678f) sink[0] = acc;
}
template <int R>
__global__ void k_gather(const float4* __restrict__ in, const uint32_t* __restrict__ idx, int64_t rows, float* sink) {
    constexpr int L = R / 16;  // lanes per row
    float acc = 0.0f;
    const int64_t total = rows * L;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx[t / L];
        const float4 v = in[row * L + t % L];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int R>
__global__ void k_scatter(float4* __restrict__ out, const uint32_t* __restrict__ idx, int64_t rows) {
    constexpr int L = R / 16;
    const int64_t total = rows * L;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx[t / L];
        out[row * L + t % L] = make_float4((float)t, 1.0f, 2.0f, 3.0f);
    }
}
__global__ void k_atomic4(unsigned* __restrict__ words, const uint32_t* __restrict__ idx, int64_t n, int64_t stride_words) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        atomicMin(&words[(int64_t)idx[t] * stride_words], (unsigned)t);
}
// Fisher-Yates is sequential; a random bijection on 2^k elements from a Feistel-style mix instead
__global__ void k_make_perm(uint32_t* idx, int64_t n, int bits) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)t;
        const uint32_t mask = (1u << bits) - 1u;
        for (int r = 0; r < 4; r++) {  // odd multiply and xor-shift-right are both bijections mod 2^bits
            x = (x * 0x9E3779B1u) & mask;
            x ^= x >> (bits / 2 + 1);
            x = (x * 0x85EBCA6Bu + 0x1234567u) & mask;
            x ^= x >> (bits / 2);
        }
        idx[t] = x;
    }
}

int main() {
    const int64_t BYTES = 2ll << 30;  // 2 GiB buffer: 8x the Infinity Cache
    float4* buf; uint32_t* idx; float* sink;
    CHECK(hipMalloc(&buf, BYTES));
    CHECK(hipMalloc(&idx, (64ll << 20) * 4));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 0, BYTES));
    const dim3 grid(8192), block(256);
    printf("pattern,bytes_touched\n");
    hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, (const float4*)buf, BYTES / 16, sink);
    printf("k_stream16,%lld\n", (long long)BYTES);
    hipLaunchKernelGGL(k_stream4, grid, block, 0, 0, (const float*)buf, BYTES / 4, sink);
    printf("k_stream4,%lld\n", (long long)BYTES);
#define ROWS(R) (BYTES / (R))
#define BITS(R) (31 - __builtin_ctz(R))  /* log2(2 GiB / R) */
#define GATHER(R)                                                                                                  \
    hipLaunchKernelGGL(k_make_perm, grid, block, 0, 0, idx, ROWS(R), BITS(R));                                     \
    hipLaunchKernelGGL((k_gather<R>), grid, block, 0, 0, (const float4*)buf, (const uint32_t*)idx, ROWS(R), sink); \
    printf("k_gather<%d>,%lld (+ %lld of indices)\n", R, (long long)BYTES, (long long)(ROWS(R) * 4));              \
    hipLaunchKernelGGL((k_scatter<R>), grid, block, 0, 0, buf, (const uint32_t*)idx, ROWS(R));                     \
    printf("k_scatter<%d>,%lld written (+ %lld of indices read)\n", R, (long long)BYTES, (long long)(ROWS(R) * 4));
    GATHER(32) GATHER(64) GATHER(128) GATHER(256)
    // 16 M atomics, one per 128-byte line of the buffer (every line touched once)
    hipLaunchKernelGGL(k_make_perm, grid, block, 0, 0, idx, 16ll << 20, 24);
    hipLaunchKernelGGL(k_atomic4, grid, block, 0, 0, (unsigned*)buf, (const uint32_t*)idx, 16ll << 20, 32);
    printf("k_atomic4,%lld useful (4 B each), %lld as 128-byte lines, %lld of indices\n", (long long)(16ll << 20) * 4, (long long)(16ll << 20) * 128, (long long)(16ll << 20) * 4);
    CHECK(hipDeviceSynchronize());
    return 0;
}
