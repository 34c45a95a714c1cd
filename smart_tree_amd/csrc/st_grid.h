// Uniform grid over a point set (shared by knn.hip and skeleton.hip).
#pragma once

#include "st_common.h"

struct StGrid {
    unsigned lo_ord[3], hi_ord[3];  // bounding box as order-preserving uints (atomics)
    float lo[3];
    float cell;
    int dim[3];
    int64_t ncell;
    unsigned rmax_ord;  // largest per-query bound (order-preserving bits), when the search radius is taken from the device
    float r;            // search radius the kNN kernels use
};

static inline int64_t st_min64(int64_t a, int64_t b) { return a < b ? a : b; }

__device__ __forceinline__ int st_grid_axis(const StGrid* g, float v, int a) {
    int c = (int)floorf((v - g->lo[a]) / g->cell);
    if (c < 0) c = 0;
    if (c >= g->dim[a]) c = g->dim[a] - 1;
    return c;
}
__device__ __forceinline__ int64_t st_grid_cell(const StGrid* g, float x, float y, float z) {
    return ((int64_t)st_grid_axis(g, x, 0) * g->dim[1] + st_grid_axis(g, y, 1)) * g->dim[2] + st_grid_axis(g, z, 2);
}

int64_t st_grid_ws_bytes(int64_t n, int64_t max_cells);
// r < 0: the search radius is max(bound[0 .. n_bound)), reduced on the device (no host round trip); cell < 0: the
// cell size is max(r / -cell, 1e-4).
int st_grid_build(const float* pts, int64_t n, float cell, int64_t max_cells, StGrid* g, uint32_t* cell_start, float4* recs,
                  void* ws, int64_t ws_bytes, hipStream_t stream, float r = 0.0f, const float* bound = nullptr,
                  int64_t n_bound = 0);
