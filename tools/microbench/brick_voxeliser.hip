// Developer measurement (round-3 verdict item 4): the front end of a sort / brick voxeliser -- counting sort of the points by 8x8x8
// voxel brick, first-point-wins per brick in an LDS table of 512 cells, emit -- against the hash insert of csrc/voxelize.hip, on the
// same cloud in whole-cloud mode (ONE membership per point: the lower bound of the blocked mode's ~1.6).  Not part of the library.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/microbench/libbrick_voxeliser.so tools/microbench/brick_voxeliser.hip
#include <hip/hip_runtime.h>
#include <cstdint>

struct BvGrid { float lo[3]; float inv_v; int nb[3]; };

__device__ __forceinline__ void bv_cell(const float* xyz, int64_t i, const BvGrid& g, int* brick, int* cell) {
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) c[a] = (int)floorf((xyz[3 * i + a] - g.lo[a]) * g.inv_v);
    *brick = ((c[0] >> 3) * g.nb[1] + (c[1] >> 3)) * g.nb[2] + (c[2] >> 3);
    *cell = ((c[0] & 7) << 6) | ((c[1] & 7) << 3) | (c[2] & 7);
}

// pass 1: histogram over bricks; every point keeps the rank its atomic returned (as st_grid_build does)
__global__ void __launch_bounds__(256) k_bv_count(const float* xyz, int64_t n, BvGrid g, unsigned* cnt, unsigned* rank) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int b, c;
        bv_cell(xyz, i, g, &b, &c);
        rank[i] = atomicAdd(&cnt[b], 1u);
    }
}

// pass 2: scatter (cell, point) to the brick's segment
__global__ void __launch_bounds__(256) k_bv_scatter(const float* xyz, int64_t n, BvGrid g, const unsigned* base, const unsigned* rank, uint2* sorted) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int b, c;
        bv_cell(xyz, i, g, &b, &c);
        sorted[base[b] + rank[i]] = make_uint2((unsigned)c, (unsigned)i);
    }
}

// pass 3: one wavefront per occupied brick: first point wins per cell (LDS atomicMin), occupied cells emitted
__global__ void __launch_bounds__(256) k_bv_bricks(const long* bricks, int n_bricks, const unsigned* base, const uint2* sorted, unsigned* n_vox,
                                                   unsigned* out_brick, unsigned* out_cell, unsigned* out_rep) {
    __shared__ unsigned cells[4][512];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bi = blockIdx.x * 4 + w;
    for (int j = lane; j < 512; j += 64) cells[w][j] = 0xffffffffu;
    __builtin_amdgcn_wave_barrier();
    if (bi >= n_bricks) return;
    const long b = bricks[bi];
    const unsigned s0 = base[b], s1 = base[b + 1];
    for (unsigned i = s0 + lane; i < s1; i += 64) {
        const uint2 e = sorted[i];
        atomicMin(&cells[w][e.x], e.y);
    }
    __builtin_amdgcn_wave_barrier();
    unsigned mine = 0;
    for (int j = 0; j < 8; j++) mine += cells[w][lane * 8 + j] != 0xffffffffu;
    unsigned incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, (unsigned)d); if (lane >= d) incl += o; }
    const unsigned total = __shfl(incl, 63);
    unsigned at = 0;
    if (lane == 0) at = atomicAdd(n_vox, total);
    at = __shfl(at, 0) + incl - mine;
    for (int j = 0; j < 8; j++) {
        const unsigned r = cells[w][lane * 8 + j];
        if (r != 0xffffffffu) { out_brick[at] = (unsigned)b; out_cell[at] = lane * 8 + j; out_rep[at] = r; at++; }
    }
}

extern "C" void bv_count(const float* xyz, int64_t n, const float* lo, float v, const int* nb, unsigned* cnt, unsigned* rank, void* stream) {
    BvGrid g{{lo[0], lo[1], lo[2]}, 1.0f / v, {nb[0], nb[1], nb[2]}};
    hipLaunchKernelGGL(k_bv_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, n, g, cnt, rank);
}
extern "C" void bv_scatter(const float* xyz, int64_t n, const float* lo, float v, const int* nb, const unsigned* base, const unsigned* rank,
                           void* sorted, void* stream) {
    BvGrid g{{lo[0], lo[1], lo[2]}, 1.0f / v, {nb[0], nb[1], nb[2]}};
    hipLaunchKernelGGL(k_bv_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, n, g, base, rank, (uint2*)sorted);
}
extern "C" void bv_bricks(const long* bricks, int n_bricks, const unsigned* base, const void* sorted, unsigned* n_vox, unsigned* out_brick,
                          unsigned* out_cell, unsigned* out_rep, void* stream) {
    hipLaunchKernelGGL(k_bv_bricks, dim3((unsigned)((n_bricks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, bricks, n_bricks, base,
                       (const uint2*)sorted, n_vox, out_brick, out_cell, out_rep);
}
