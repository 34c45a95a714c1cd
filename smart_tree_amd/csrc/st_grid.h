// Uniform grid over a point set (shared by knn.hip and skeleton.hip).
#pragma once

#include "st_common.h"

#define ST_MAX_SEG 64  // independent clouds ("segments") one batched call may hold

// Batched calls carry several independent clouds in one point array: cloud b owns the index range
// [seg_off[b], seg_off[b + 1]).  The grid keeps them apart in INTEGER cell space -- cloud b's cells are shifted by
// b * seg_dim0 along x, a search never leaves its own slab -- so no float coordinate changes and every per-cloud result
// is bit-identical to the one-cloud call (the result of a search never depends on the grid geometry).
struct StGrid {
    unsigned lo_ord[3], hi_ord[3];  // bounding box as order-preserving uints (atomics)
    float lo[3];
    float cell;
    int dim[3];        // dim[0] = seg_dim0 * nseg
    int seg_dim0;      // cells along x of one cloud's slab
    int nseg;
    int64_t ncell;
    unsigned rmax_ord;  // largest per-query bound (order-preserving bits), when the search radius is taken from the device
    float r;            // search radius the kNN kernels use (one cloud) / largest of seg_r (cell sizing)
    unsigned long long bound_sum_fix;  // sum of the per-query bounds in 2^-16 units (integer: the same in every run)
    unsigned long long bound_cnt;      // how many bounds that sum holds: the VALID, non-NaN ones (a search over a subset counts the subset)
    unsigned seg_rmax_ord[ST_MAX_SEG];
    float seg_r[ST_MAX_SEG];  // per-cloud search radius = max(bound) over THAT cloud, as the one-cloud call computes it
};

static inline int64_t st_min64(int64_t a, int64_t b) { return a < b ? a : b; }

// cloud of element i: seg_off[s] <= i < seg_off[s + 1] (empty clouds are skipped); seg_off == nullptr: one cloud
__device__ __forceinline__ int st_seg_find(const int* seg_off, int nseg, int64_t i) {
    if (seg_off == nullptr || nseg <= 1) return 0;
    int lo = 0, hi = nseg;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)seg_off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

__device__ __forceinline__ int st_grid_axis(const StGrid* g, float v, int a) {
    int c = (int)floorf((v - g->lo[a]) / g->cell);
    const int d = a == 0 ? g->seg_dim0 : g->dim[a];
    if (c < 0) c = 0;
    if (c >= d) c = d - 1;
    return c;
}
__device__ __forceinline__ int64_t st_grid_cell(const StGrid* g, float x, float y, float z, int seg = 0) {
    return ((int64_t)(st_grid_axis(g, x, 0) + seg * g->seg_dim0) * g->dim[1] + st_grid_axis(g, y, 1)) * g->dim[2] + st_grid_axis(g, z, 2);
}

int64_t st_grid_ws_bytes(int64_t n, int64_t max_cells);
// r < 0: the search radius is max(bound[0 .. n_bound)), reduced on the device (no host round trip); cell < 0: the
// cell size is max(r / -cell, 1e-4).  seg_off / nseg: the clouds of a batched call (pts index space; bound_seg_off the
// same for the bound array), nullptr / 1 = one cloud.  mean_mult > 0 (with cell < 0, r < 0): the cell is at most mean_mult x
// the MEAN bound -- one far-too-large bound (a radius the network got wrong) would otherwise make every cell, and with it
// every search, as coarse as that one needs.  The cell size changes the speed of a search, never its result.
int st_grid_build(const float* pts, int64_t n, float cell, int64_t max_cells, StGrid* g, uint32_t* cell_start, float4* recs,
                  void* ws, int64_t ws_bytes, hipStream_t stream, float r = 0.0f, const float* bound = nullptr,
                  int64_t n_bound = 0, const int* seg_off = nullptr, int nseg = 1, const int* bound_seg_off = nullptr,
                  float mean_mult = 0.0f, const uint8_t* valid = nullptr /* [n] optional: only points with valid[i] != 0 enter the table
                  (and the bound reductions): a search over a SUBSET of an array without compacting it first */);
