// st_knn_radius: fixed-radius k-nearest-neighbour search on a uniform grid.
//
// Replaces frnn.frnn_grid_points as the reference calls it through
//   knn / nn          smart_tree/skeleton/graph.py:12-33
//   outlier_removal   smart_tree/skeleton/filter.py:6-11   (K = 8)
//   nn_graph          smart_tree/skeleton/graph.py:36-40   (K = 16)
// Semantics (oracle/skeleton_oracle.c so_knn): for each query the <= K nearest points with
// d2 < r*r, ordered by (d2, index) ascending; d2 = (dx*dx + dy*dy) + dz*dz in float32 without
// contraction; idx -1 / dist NaN padding; returned distances are sqrtf(d2).
// An optional per-query bound (bound[i], with strict or non-strict compare on sqrtf(d2)) prunes
// the search to the radius the caller will filter with anyway (nn_graph drops d > r_i,
// outlier_removal needs d < r_i): same result, far fewer cells visited for thin branches.
//
// Grid: dense cell_start[] over the bounding box (cells capped, cell size doubled until it fits),
// points counting-sorted by cell into float4 (x, y, z, index) records; a 4-lane group per query, every
// lane keeps a top-K in registers (fully unrolled insertion), merged by shuffles.
#include "st_common.h"
#include "st_grid.h"

#define KNN_BLOCK 256

// ------------------------------------------------------------------------------- grid build ---
__global__ void k_grid_init(StGrid* g) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int a = 0; a < 3; a++) { g->lo_ord[a] = 0xffffffffu; g->hi_ord[a] = 0u; }
    }
}

__global__ void __launch_bounds__(KNN_BLOCK) k_grid_bbox(const float* pts, int64_t n, StGrid* g) {
    __shared__ unsigned lo[3], hi[3];
    if (threadIdx.x < 3) { lo[threadIdx.x] = 0xffffffffu; hi[threadIdx.x] = 0u; }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int a = 0; a < 3; a++) {
            unsigned o = st_f2ord(pts[3 * i + a]);
            if (o < lo[a]) atomicMin(&lo[a], o);
            if (o > hi[a]) atomicMax(&hi[a], o);
        }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&g->lo_ord[threadIdx.x], lo[threadIdx.x]); atomicMax(&g->hi_ord[threadIdx.x], hi[threadIdx.x]); }
}

__global__ void k_grid_dims(StGrid* g, float cell, int64_t max_cells) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = st_ord2f(g->lo_ord[a]); hi[a] = st_ord2f(g->hi_ord[a]); g->lo[a] = lo[a]; }
    if (!(cell > 0.0f)) cell = 1.0f;
    for (;;) {
        double total = 1.0;
        for (int a = 0; a < 3; a++) {
            float ext = (hi[a] - lo[a]) / cell;
            int d = ext < 2.0e9f ? (int)floorf(ext) + 1 : 0x7fffffff;
            if (d < 1) d = 1;
            g->dim[a] = d;
            total *= (double)d;
        }
        if (total <= (double)max_cells) break;
        cell *= 2.0f;
    }
    g->cell = cell;
    g->ncell = (int64_t)g->dim[0] * g->dim[1] * g->dim[2];
}

__global__ void __launch_bounds__(KNN_BLOCK) k_grid_count(const float* pts, int64_t n, const StGrid* g, uint32_t* counts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&counts[st_grid_cell(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2])], 1u);
}

__global__ void __launch_bounds__(KNN_BLOCK) k_grid_fill(const float* pts, int64_t n, const StGrid* g, uint32_t* cursor,
                                                         float4* recs) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = st_grid_cell(g, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        uint32_t pos = atomicAdd(&cursor[c], 1u);
        recs[pos] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __uint_as_float((unsigned)i));
    }
}

int64_t st_grid_ws_bytes(int64_t n, int64_t max_cells) {
    StArena a(nullptr, 0);
    a.take<uint32_t>(max_cells + 1);               // cursor
    a.take<char>(st_scan_ws_bytes(max_cells + 1)); // scan scratch
    (void)n;
    return a.used;
}

// Builds grid over pts[n]; g (device struct), cell_start[max_cells+1], recs[n] are caller arrays.
int st_grid_build(const float* pts, int64_t n, float cell, int64_t max_cells, StGrid* g, uint32_t* cell_start, float4* recs,
                  void* ws, int64_t ws_bytes, hipStream_t stream) {
    StArena a(ws, ws_bytes);
    uint32_t* cursor = a.take<uint32_t>(max_cells + 1);
    int64_t scan_bytes = st_scan_ws_bytes(max_cells + 1);
    char* scan_ws = a.take<char>(scan_bytes);
    if (!cursor || !scan_ws) {
        st_set_error("grid: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    unsigned gb = (unsigned)st_min64(st_div_up(n > 0 ? n : 1, KNN_BLOCK), 4096);
    hipLaunchKernelGGL(k_grid_init, dim3(1), dim3(64), 0, stream, g);
    hipLaunchKernelGGL(k_grid_bbox, dim3(gb), dim3(KNN_BLOCK), 0, stream, pts, n, g);
    hipLaunchKernelGGL(k_grid_dims, dim3(1), dim3(64), 0, stream, g, cell, max_cells);
    // one small read-back sizes the histogram/scan to the cells actually used instead of the capacity
    StGrid h;
    (void)hipMemcpyAsync(&h, g, sizeof(StGrid), hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    ST_CHECK_LAUNCH();
    const int64_t ncell = h.ncell;
    ST_REQUIRE(ncell >= 1 && ncell <= max_cells, "grid: bad cell count %lld", (long long)ncell);
    (void)hipMemsetAsync(cell_start, 0, (ncell + 1) * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(k_grid_count, dim3(gb), dim3(KNN_BLOCK), 0, stream, pts, n, (const StGrid*)g, cell_start);
    ST_TRY(st_exclusive_scan_u32(cell_start, cell_start, ncell + 1, nullptr, scan_ws, scan_bytes, stream));
    (void)hipMemcpyAsync(cursor, cell_start, (ncell + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
    hipLaunchKernelGGL(k_grid_fill, dim3(gb), dim3(KNN_BLOCK), 0, stream, pts, n, (const StGrid*)g, cursor, recs);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ----------------------------------------------------------------------------------- search ---
__device__ __forceinline__ bool knn_less(float d, int j, float bd, int bj) { return d < bd || (d == bd && j < bj); }

// mode 0: no per-query bound; 1: keep sqrtf(d2) <= bound[i]; 2: keep sqrtf(d2) < bound[i]
// KNN_LANES lanes work on one query: each takes every KNN_LANES-th (x, y) grid row and keeps its own
// sorted top-K in registers; the K smallest of the union are then drawn by K rounds of a group-wide
// minimum (xor shuffles).  With one lane per query a 50k-point cloud is <1 wavefront per SIMD and the
// candidate loop is one long dependent chain.
#define KNN_LANES 4
template <int K>
__global__ void __launch_bounds__(KNN_BLOCK) k_knn(const float* __restrict__ src, int64_t n1, const StGrid* __restrict__ g,
                                                   const uint32_t* __restrict__ cell_start, const float4* __restrict__ recs,
                                                   float r, const float* __restrict__ bound, int mode,
                                                   int64_t* __restrict__ idx_out, float* __restrict__ dist_out) {
    const int64_t gid = (int64_t)blockIdx.x * KNN_BLOCK + threadIdx.x;
    const int sub = (int)(gid & (KNN_LANES - 1));
    int64_t i = gid / KNN_LANES;
    const bool valid = i < n1;
    if (!valid) i = n1 - 1;  // keep the lane in the shuffles below; its result is discarded
    const float px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2];
    const float r2 = r * r;
    float reach_r = r;
    float bnd = 0.0f;
    if (mode != 0) {
        bnd = bound[i];
        if (bnd < reach_r) reach_r = bnd;
    }
    float bd[K];
    int bi[K];
#pragma unroll
    for (int q = 0; q < K; q++) { bd[q] = __uint_as_float(0x7f800000u); bi[q] = 0x7fffffff; }
    const float cell = g->cell;
    int reach = reach_r > 0.0f ? (int)ceilf(reach_r / cell) : 0;
    if (reach < 1) reach = 1;
    int c0[3];
    c0[0] = (int)floorf((px - g->lo[0]) / cell);
    c0[1] = (int)floorf((py - g->lo[1]) / cell);
    c0[2] = (int)floorf((pz - g->lo[2]) / cell);
    const int x0 = st_max(c0[0] - reach, 0), x1 = st_min(c0[0] + reach, g->dim[0] - 1);
    const int y0 = st_max(c0[1] - reach, 0), y1 = st_min(c0[1] + reach, g->dim[1] - 1);
    const int z0 = st_max(c0[2] - reach, 0), z1 = st_min(c0[2] + reach, g->dim[2] - 1);
    // rows (x, y) farther than the search radius in the xy-plane are skipped and the z-range of the others
    // is clipped to the sphere (a slightly inflated radius keeps the pruning conservative)
    const float rs = reach_r * 1.0001f + 1e-7f, rs2 = rs * rs;
    const int ny = y1 - y0 + 1, nrows = (x1 - x0 + 1) * ny;
    for (int rowi = sub; rowi < nrows; rowi += KNN_LANES) {
        const int x = x0 + rowi / ny, y = y0 + rowi % ny;
        const float cx0 = g->lo[0] + (float)x * cell, ex = px < cx0 ? cx0 - px : (px > cx0 + cell ? px - (cx0 + cell) : 0.0f);
        const float cy0 = g->lo[1] + (float)y * cell, ey = py < cy0 ? cy0 - py : (py > cy0 + cell ? py - (cy0 + cell) : 0.0f);
        const float dxy2 = ex * ex + ey * ey;
        if (dxy2 > rs2) continue;
        const float rz = sqrtf(rs2 - dxy2);
        const int za = st_max((int)floorf((pz - rz - g->lo[2]) / cell) - 1, z0), zb = st_min((int)floorf((pz + rz - g->lo[2]) / cell) + 1, z1);
        if (za > zb) continue;
        // cells along z are contiguous: one [start, end) range per (x, y) row
        const int64_t row = ((int64_t)x * g->dim[1] + y) * g->dim[2];
        const uint32_t s = cell_start[row + za], e = cell_start[row + zb + 1];
        for (uint32_t t = s; t < e; t++) {
            const float4 q = recs[t];
            const float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
            float d2 = dx * dx;
            float tt = dy * dy;
            d2 = d2 + tt;
            tt = dz * dz;
            d2 = d2 + tt;
            if (!(d2 < r2)) continue;
            if (mode == 1 && !(sqrtf(d2) <= bnd)) continue;
            if (mode == 2 && !(sqrtf(d2) < bnd)) continue;
            const int j = (int)__float_as_uint(q.w);
            if (!knn_less(d2, j, bd[K - 1], bi[K - 1])) continue;
#pragma unroll
            for (int p = K - 1; p > 0; p--) {
                const bool lt_prev = knn_less(d2, j, bd[p - 1], bi[p - 1]);
                const bool lt_cur = knn_less(d2, j, bd[p], bi[p]);
                const float nd = lt_prev ? bd[p - 1] : (lt_cur ? d2 : bd[p]);
                const int nj = lt_prev ? bi[p - 1] : (lt_cur ? j : bi[p]);
                bd[p] = nd;
                bi[p] = nj;
            }
            if (knn_less(d2, j, bd[0], bi[0])) { bd[0] = d2; bi[0] = j; }
        }
    }
    // merge: K rounds; every lane offers the head of its list, the group minimum is emitted and popped
#pragma unroll
    for (int q = 0; q < K; q++) {
        float md = bd[0];
        int mj = bi[0];
#pragma unroll
        for (int m = 1; m < KNN_LANES; m <<= 1) {
            const float od = __shfl_xor(md, m);
            const int oj = __shfl_xor(mj, m);
            if (knn_less(od, oj, md, mj)) { md = od; mj = oj; }
        }
        if (mj == bi[0] && mj != 0x7fffffff) {  // my head won (indices are unique): pop it
#pragma unroll
            for (int p = 0; p < K - 1; p++) { bd[p] = bd[p + 1]; bi[p] = bi[p + 1]; }
            bd[K - 1] = __uint_as_float(0x7f800000u);
            bi[K - 1] = 0x7fffffff;
        }
        if (sub == 0 && valid) {
            const bool ok = mj != 0x7fffffff;
            idx_out[i * K + q] = ok ? (int64_t)mj : (int64_t)-1;
            dist_out[i * K + q] = ok ? sqrtf(md) : __uint_as_float(0x7fc00000u);
        }
    }
}

#define KNN_MAX_CELLS (1ll << 24)

static void knn_layout(StArena& a, int64_t n2, StGrid** g, uint32_t** cell_start, float4** recs, char** sub, int64_t* sub_bytes) {
    *g = a.take<StGrid>(1);
    *cell_start = a.take<uint32_t>(KNN_MAX_CELLS + 1);
    *recs = a.take<float4>(n2);
    *sub_bytes = st_grid_ws_bytes(n2, KNN_MAX_CELLS);
    *sub = a.take<char>(*sub_bytes);
}

extern "C" int64_t st_knn_workspace_bytes(int64_t n_dst) {
    StArena a(nullptr, 0);
    StGrid* g; uint32_t* cs; float4* recs; char* sub; int64_t sb;
    knn_layout(a, n_dst, &g, &cs, &recs, &sub, &sb);
    return a.used;
}

// idx [n1,K] int64 (-1 pad), dist [n1,K] float32 = sqrtf(d2) (NaN pad).  bound/bound_mode: see above.
// cell_hint: preferred grid cell size (<= 0: use r).
extern "C" int st_knn_radius(const float* src, int64_t n1, const float* dst, int64_t n2, int K, float r, const float* bound,
                             int bound_mode, float cell_hint, int64_t* idx, float* dist, void* ws, int64_t ws_bytes,
                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ST_REQUIRE(K == 1 || K == 8 || K == 16, "knn: K must be 1, 8 or 16 (got %d)", K);
    ST_REQUIRE(bound_mode == 0 || bound != nullptr, "knn: bound_mode needs a bound array");
    ST_REQUIRE(n2 < (1ll << 31), "knn: too many points");
    if (n1 <= 0) return ST_OK;
    StArena a(ws, ws_bytes);
    StGrid* g; uint32_t* cell_start; float4* recs; char* sub; int64_t sub_bytes;
    knn_layout(a, n2, &g, &cell_start, &recs, &sub, &sub_bytes);
    if (!a.ok() || !sub) {
        st_set_error("knn: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    ST_TRY(st_grid_build(dst, n2, cell_hint > 0.0f ? cell_hint : r, KNN_MAX_CELLS, g, cell_start, recs, sub, sub_bytes, stream));
    dim3 grid((unsigned)st_div_up(n1 * KNN_LANES, KNN_BLOCK)), block(KNN_BLOCK);
    if (K == 1)
        hipLaunchKernelGGL((k_knn<1>), grid, block, 0, stream, src, n1, (const StGrid*)g, (const uint32_t*)cell_start,
                           (const float4*)recs, r, bound, bound_mode, idx, dist);
    else if (K == 8)
        hipLaunchKernelGGL((k_knn<8>), grid, block, 0, stream, src, n1, (const StGrid*)g, (const uint32_t*)cell_start,
                           (const float4*)recs, r, bound, bound_mode, idx, dist);
    else
        hipLaunchKernelGGL((k_knn<16>), grid, block, 0, stream, src, n1, (const StGrid*)g, (const uint32_t*)cell_start,
                           (const float4*)recs, r, bound, bound_mode, idx, dist);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
