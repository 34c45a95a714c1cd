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
// points counting-sorted by cell into float4 (x, y, z, index) records; one wavefront per query, the K nearest
// drawn from an LDS candidate list by rank counting.
#include "st_common.h"
#include "st_grid.h"

#define KNN_BLOCK 256

#define KNN_MEAN_MULT 0.45f  // neighbour-search grids: cell <= this x the mean per-query bound (0.45 x the mean radius = the max / 12 cell of a 1M-point
                             // tree at 2 cm: that tuning, without the outliers); a call may override it (cell_mean_mult >= 0: test hook)

// ------------------------------------------------------------------------------- grid build ---
__global__ void k_grid_init(StGrid* g) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int a = 0; a < 3; a++) { g->lo_ord[a] = 0xffffffffu; g->hi_ord[a] = 0u; }
        g->rmax_ord = 0u;
        g->bound_sum_fix = 0ull;
        g->bound_cnt = 0ull;
        g->r = 0.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x < ST_MAX_SEG) { g->seg_rmax_ord[threadIdx.x] = 0u; g->seg_r[threadIdx.x] = 0.0f; }
}

__device__ __forceinline__ unsigned grid_wave_min(unsigned v) { for (int d = 32; d > 0; d >>= 1) { const unsigned o = __shfl_xor(v, d); v = o < v ? o : v; } return v; }
__device__ __forceinline__ unsigned grid_wave_max(unsigned v) { for (int d = 32; d > 0; d >>= 1) { const unsigned o = __shfl_xor(v, d); v = o > v ? o : v; } return v; }

// Both reductions end in atomics on a handful of words of *g: a few hundred workgroups with a private running value per lane,
// one shuffle reduction per wavefront and one global atomic per workgroup and word (thousands of workgroups, each with its
// own atomics on the same six addresses, spent 90 us on a 17 MB array -- the atomics, not the loads).
// blockIdx.y = cloud (gridDim.y = 1, seg_off == nullptr: the whole array is one cloud)
__global__ void __launch_bounds__(KNN_BLOCK) k_bound_max(const float* bound, int64_t n, StGrid* g, const int* seg_off, const uint8_t* valid) {
    __shared__ unsigned m;
    __shared__ unsigned long long sum, cnt;
    if (threadIdx.x == 0) { m = 0u; sum = 0ull; cnt = 0ull; }
    __syncthreads();
    const int seg = blockIdx.y;
    const int64_t i0 = seg_off ? seg_off[seg] : 0, i1 = seg_off ? seg_off[seg + 1] : n;
    unsigned mine = 0u;
    unsigned long long fix = 0ull;  // sum of the bounds in 2^-16 units (a bound is a radius: metres; clamped so 2^40 of them fit)
    unsigned long long seen = 0ull;  // bounds in that sum
    for (int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += (int64_t)gridDim.x * blockDim.x) {
        if (valid && !valid[i]) continue;  // (a point that is not part of the search: see st_grid_build)
        const float b = bound[i];
        if (b != b) continue;  // a NaN bound admits nobody (knn_d2_max) and must not become the search radius of its cloud
        const unsigned o = st_f2ord(b);
        if (o > mine) mine = o;
        fix += (unsigned long long)(fminf(fmaxf(b, 0.0f), 128.0f) * 65536.0f);  // (NaN -> 0)
        seen++;
    }
    mine = grid_wave_max(mine);
    for (int d = 32; d > 0; d >>= 1) { fix += __shfl_xor(fix, d); seen += __shfl_xor(seen, d); }
    if ((threadIdx.x & 63) == 0) { if (mine) atomicMax(&m, mine); atomicAdd(&sum, fix); atomicAdd(&cnt, seen); }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (m) { atomicMax(&g->rmax_ord, m); atomicMax(&g->seg_rmax_ord[seg], m); }
        if (sum) atomicAdd(&g->bound_sum_fix, sum);
        if (cnt) atomicAdd(&g->bound_cnt, cnt);
    }
}

__global__ void __launch_bounds__(KNN_BLOCK) k_grid_bbox(const float* pts, int64_t n, StGrid* g, const uint8_t* valid) {
    __shared__ unsigned lo[3], hi[3];
    if (threadIdx.x < 3) { lo[threadIdx.x] = 0xffffffffu; hi[threadIdx.x] = 0u; }
    __syncthreads();
    unsigned l[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, h[3] = {0u, 0u, 0u};
    // the flat array of 3n floats, one per lane and step (coalesced dwords): element e belongs to axis e % 3
    const int64_t total = 3 * n, step = (int64_t)gridDim.x * blockDim.x;
    const int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int step3 = (int)(step % 3);
    int a = (int)(e0 % 3);
    for (int64_t e = e0; e < total; e += step, a = (a + step3) % 3) {
        const float pv = pts[e];
        if (valid && !valid[e / 3]) continue;  // a point outside the searched subset does not size the grid either
        if (!(fabsf(pv) <= 3.0e38f)) continue;  // NaN / infinity: such a point is not part of the search (k_grid_count), it must not size the grid
        const unsigned o = st_f2ord(pv);
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (a == k) { l[k] = o < l[k] ? o : l[k]; h[k] = o > h[k] ? o : h[k]; }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) { l[a] = grid_wave_min(l[a]); h[a] = grid_wave_max(h[a]); }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) { atomicMin(&lo[a], l[a]); atomicMax(&hi[a], h[a]); }
    __syncthreads();
    if (threadIdx.x < 3 && lo[threadIdx.x] <= hi[threadIdx.x]) {
        atomicMin(&g->lo_ord[threadIdx.x], lo[threadIdx.x]);
        atomicMax(&g->hi_ord[threadIdx.x], hi[threadIdx.x]);
    }
}

__global__ void k_grid_dims(StGrid* g, float cell, int64_t max_cells, float r, int nseg, float mean_mult, int64_t n_bound) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = st_ord2f(g->lo_ord[a]); hi[a] = st_ord2f(g->hi_ord[a]); g->lo[a] = lo[a]; }
    for (int s = 0; s < nseg; s++) g->seg_r[s] = r < 0.0f ? st_ord2f(g->seg_rmax_ord[s]) : r;
    if (r < 0.0f) r = st_ord2f(g->rmax_ord);  // the largest per-query bound, reduced by k_bound_max
    g->r = r;
    if (cell < 0.0f) {
        cell = r / -cell;
        if (mean_mult > 0.0f && n_bound > 0 && g->bound_cnt > 0ull) {  // ... but no coarser than mean_mult x the mean bound
            // (the mean over the bounds k_bound_max summed: the valid, non-NaN ones -- dividing by n_bound made the cap shrink with
            // the valid fraction when the search runs over a subset)
            const float mean = (float)((double)g->bound_sum_fix / 65536.0 / (double)g->bound_cnt);
            if (mean > 0.0f) cell = fminf(cell, mean_mult * mean);
        }
        cell = fmaxf(cell, 1e-4f);
    }
    if (!(cell > 0.0f) || !(cell <= 3.0e38f)) cell = 1.0f;
    g->nseg = nseg;
    for (int a = 0; a < 3; a++)
        if (!(hi[a] >= lo[a])) { lo[a] = hi[a] = 0.0f; g->lo[a] = 0.0f; }  // no finite coordinate at all: an empty one-cell grid
    for (int it = 0;; it++) {
        double total = (double)nseg;
        for (int a = 0; a < 3; a++) {
            float ext = (hi[a] - lo[a]) / cell;
            int d = ext < 2.0e9f ? (int)floorf(ext) + 1 : 0x7fffffff;
            if (d < 1 || it >= 300) d = 1;  // (300 doublings exceed every finite extent; the bound only guards against a loop that cannot end)
            g->dim[a] = d;
            total *= (double)d;
        }
        if (total <= (double)max_cells) break;
        cell *= 2.0f;
    }
    g->seg_dim0 = g->dim[0];
    g->dim[0] = g->seg_dim0 * nseg;  // every cloud gets its own slab of cells along x
    g->cell = cell;
    g->ncell = (int64_t)g->dim[0] * g->dim[1] * g->dim[2];
}

// The counting pass hands every point its rank inside its cell (the value its atomicAdd returns), so the fill pass needs
// neither a second round of atomics nor a cursor copy of the (tens of millions of words long) cell table.
__global__ void k_grid_ncell1(const StGrid* g, int64_t* out) { *out = g->ncell + 1; }

__global__ void __launch_bounds__(KNN_BLOCK) k_grid_count(const float* pts, int64_t n, const StGrid* g, uint32_t* counts,
                                                          const int* seg_off, int nseg, uint32_t* pt_cell, uint32_t* pt_rank,
                                                          const uint8_t* valid) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (valid && !valid[i]) { pt_cell[i] = 0xffffffffu; continue; }  // stays out of the table
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        // a NaN / infinite coordinate: filed under the first cell of its cloud (a cell index cannot be computed from it).  As a
        // candidate it is nobody's neighbour -- every distance to it is NaN or infinite, the compare fails as the oracle's does --
        // and as a query it finds nobody (k_knn)
        const bool finite = fabsf(x) <= 3.0e38f && fabsf(y) <= 3.0e38f && fabsf(z) <= 3.0e38f;
        const int64_t c = st_grid_cell(g, finite ? x : g->lo[0], finite ? y : g->lo[1], finite ? z : g->lo[2], st_seg_find(seg_off, nseg, i));
        pt_cell[i] = (uint32_t)c;
        pt_rank[i] = atomicAdd(&counts[c], 1u);
    }
}

__global__ void __launch_bounds__(KNN_BLOCK) k_grid_fill(const float* pts, int64_t n, const uint32_t* cell_start,
                                                         const uint32_t* pt_cell, const uint32_t* pt_rank, float4* recs) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (pt_cell[i] == 0xffffffffu) continue;
        const uint32_t pos = cell_start[pt_cell[i]] + pt_rank[i];
        recs[pos] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __uint_as_float((unsigned)i));
    }
}

int64_t st_grid_ws_bytes(int64_t n, int64_t max_cells) {
    StArena a(nullptr, 0);
    a.take<uint32_t>(n);                           // cell of every point
    a.take<uint32_t>(n);                           // its rank inside the cell
    a.take<int64_t>(1);                            // cells in use + 1 (device-side length of the table)
    a.take<char>(st_scan_ws_bytes(max_cells + 1)); // scan scratch
    return a.used;
}

// Builds grid over pts[n]; g (device struct), cell_start[max_cells+1], recs[n] are caller arrays.
int st_grid_build(const float* pts, int64_t n, float cell, int64_t max_cells, StGrid* g, uint32_t* cell_start, float4* recs,
                  void* ws, int64_t ws_bytes, hipStream_t stream, float r, const float* bound, int64_t n_bound,
                  const int* seg_off, int nseg, const int* bound_seg_off, float mean_mult, const uint8_t* valid) {
    if (nseg < 1 || nseg > ST_MAX_SEG) { st_set_error("grid: 1 <= clouds per batch <= %d (got %d)", ST_MAX_SEG, nseg); return ST_ERR_INVALID; }
    if (!seg_off) nseg = 1;
    StArena a(ws, ws_bytes);
    uint32_t* pt_cell = a.take<uint32_t>(n);
    uint32_t* pt_rank = a.take<uint32_t>(n);
    int64_t* ncell1_dev = a.take<int64_t>(1);
    int64_t scan_bytes = st_scan_ws_bytes(max_cells + 1);
    char* scan_ws = a.take<char>(scan_bytes);
    if (!pt_cell || !pt_rank || !ncell1_dev || !scan_ws) {
        st_set_error("grid: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    unsigned gb = (unsigned)st_min64(st_div_up(n > 0 ? n : 1, KNN_BLOCK), 4096);
    hipLaunchKernelGGL(k_grid_init, dim3(1), dim3(64), 0, stream, g);
    static_assert(ST_MAX_SEG <= 64, "k_grid_init clears the per-cloud radii with one wavefront");
    hipLaunchKernelGGL(k_grid_bbox, dim3((unsigned)st_min64(st_div_up(3 * (n > 0 ? n : 1), (int64_t)KNN_BLOCK * 8), 512)), dim3(KNN_BLOCK), 0, stream, pts, n, g, valid);
    // No read-back of the cell count (a blocking round trip costs ~1 ms beside other clouds' kernels):
    // the grid is limited to 128 cells per point -- a 2 cm kNN grid over a tree has ~65 -- and histogram, scan and cursor
    // copy run over that bound; cells past the real count stay empty.  A cloud that would need more gets a coarser grid
    // (k_grid_dims doubles the cell), which changes the speed of a search, never its result.
    const int64_t ncell = st_min64(max_cells, 128 * n + 65536);
    if (r < 0.0f && bound && n_bound > 0)
        hipLaunchKernelGGL(k_bound_max, dim3((unsigned)st_min64(st_div_up(n_bound, (int64_t)KNN_BLOCK * nseg * 8), 512 / nseg + 1), (unsigned)nseg),
                           dim3(KNN_BLOCK), 0, stream, bound, n_bound, g, nseg > 1 ? (bound_seg_off ? bound_seg_off : seg_off) : (const int*)nullptr,
                           valid);
    hipLaunchKernelGGL(k_grid_dims, dim3(1), dim3(64), 0, stream, g, cell, ncell, r, nseg, mean_mult,
                       (r < 0.0f && bound) ? n_bound : (int64_t)0);
    if (getenv("ST_GRID_DEBUG")) {
        StGrid h;
        (void)hipMemcpyAsync(&h, g, sizeof(StGrid), hipMemcpyDeviceToHost, stream);
        st_stream_wait(stream);
        fprintf(stderr, "grid: n=%lld cell=%g ncell=%lld of %lld dims=%d,%d,%d\n", (long long)n, h.cell, (long long)h.ncell,
                (long long)ncell, h.dim[0], h.dim[1], h.dim[2]);
    }
    // the cell table is cleared and scanned over the cells the grid really has (g->ncell + 1, a device-side number), not over
    // the host's bound: a batch of 16 clouds is bounded at 134M cells and uses 18M
    hipLaunchKernelGGL(k_grid_ncell1, dim3(1), dim3(1), 0, stream, (const StGrid*)g, ncell1_dev);
    st_fill_u32_dev(cell_start, ncell + 1, ncell1_dev, 0u, stream);
    hipLaunchKernelGGL(k_grid_count, dim3(gb), dim3(KNN_BLOCK), 0, stream, pts, n, (const StGrid*)g, cell_start, seg_off, nseg,
                       pt_cell, pt_rank, valid);
    ST_TRY(st_exclusive_scan_u32(cell_start, cell_start, ncell + 1, nullptr, scan_ws, scan_bytes, stream, ncell1_dev));
    hipLaunchKernelGGL(k_grid_fill, dim3(gb), dim3(KNN_BLOCK), 0, stream, pts, n, (const uint32_t*)cell_start,
                       (const uint32_t*)pt_cell, (const uint32_t*)pt_rank, recs);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ----------------------------------------------------------------------------------- search ---

// mode 0: no per-query bound; 1: keep sqrtf(d2) <= bound[i]; 2: keep sqrtf(d2) < bound[i]
//
// One WAVEFRONT per query.  A 50k-point cloud with a lane (or a few lanes) per query is a handful of
// wavefronts per SIMD, each walking a long chain of dependent loads; with a wavefront per query there are
// 50k of them and every step is 64 wide:
//   1. lanes take the (x, y) rows of grid cells around the query (cells along z are contiguous in memory, so a
//      row is ONE [start, end) range, clipped to the search sphere) and scan their sizes;
//   2. the candidates of all rows are dealt out one per lane (coalesced float4 records), tested against r and
//      the per-query bound, and the survivors appended to a list in LDS as 64-bit keys (d2 bits, index);
//   3. every survivor counts the keys smaller than its own (LDS broadcast reads): that count IS its output
//      slot -- the K nearest come out sorted by (d2, index) without a sort.  If the list fills up it is cut
//      back to its K smallest the same way and the scan goes on.
#define KNN_WAVES (KNN_BLOCK / 64)
#define KNN_MASK_WORDS 64  // candidate -> row by popcount over start marks while a row chunk holds <= 64 * 64 candidates (else: binary search)
#define KNN_CAP 128  // keys per query held in LDS between cuts: a cut happens when a chunk of 64 might not fit any more

__device__ __forceinline__ uint32_t knn_wave_scan(uint32_t v, int lane) {  // inclusive
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, (unsigned)d); if (lane >= d) v += o; }
    return v;
}

// largest d2 >= 0 with sqrtf(d2) <= bound (strict: < bound); -1 if there is none (negative or NaN bound, or bound 0 and strict)
__device__ __forceinline__ float knn_d2_max(float bound, bool strict) {
    if (!(bound >= 0.0f)) return -1.0f;
    if (bound == __uint_as_float(0x7f800000u)) return strict ? __uint_as_float(0x7f7fffffu) : bound;
    auto pass = [&](float x) { const float s = sqrtf(x); return strict ? s < bound : s <= bound; };
    unsigned b = __float_as_uint(bound * bound);  // non-negative floats order like their bit patterns
    if (b > 0x7f7fffffu) b = 0x7f7fffffu;         // bound^2 overflowed: start from the largest finite float
    while (!pass(__uint_as_float(b))) { if (b == 0u) return -1.0f; b--; }
    while (b < 0x7f7fffffu && pass(__uint_as_float(b + 1u))) b++;
    return __uint_as_float(b);
}

struct __attribute__((aligned(16))) KnnPair { unsigned long long x, y; };

// rank of every key = number of smaller keys (keys are unique: the index is part of them); emit(key, rank) for the n keys.
// NCH = 64-key chunks a lane holds (n <= 64 * NCH).  All 64 lanes call.
template <int NCH, class F>
__device__ __forceinline__ void knn_rank(const unsigned long long* keys, int n, int lane, F emit) {
    unsigned long long mine[NCH];
    int rank[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int j = c * 64 + lane;
        mine[c] = j < n ? keys[j] : ~0ull;
        rank[c] = 0;
    }
    int t = 0;
    for (; t + 4 <= n; t += 4) {  // four keys per step, two 16-byte broadcast reads in flight together
        const KnnPair a = *reinterpret_cast<const KnnPair*>(keys + t), b = *reinterpret_cast<const KnnPair*>(keys + t + 2);
#pragma unroll
        for (int c = 0; c < NCH; c++)
            rank[c] += (a.x < mine[c] ? 1 : 0) + (a.y < mine[c] ? 1 : 0) + (b.x < mine[c] ? 1 : 0) + (b.y < mine[c] ? 1 : 0);
    }
    for (; t < n; t++) {
        const unsigned long long kt = keys[t];  // same address in every lane: a broadcast
#pragma unroll
        for (int c = 0; c < NCH; c++) rank[c] += kt < mine[c] ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < NCH; c++)
        if (c * 64 + lane < n) emit(mine[c], rank[c]);
}

// Keep the K smallest of keys[0..n), in ascending order.  Returns the new count.  All 64 lanes call.
template <int K>
__device__ __forceinline__ int knn_cut(unsigned long long* keys, int n, int lane) {
    __builtin_amdgcn_wave_barrier();
    auto put = [&](unsigned long long key, int rank) { if (rank < K) keys[rank] = key; };
    if (n <= 64) knn_rank<1>(keys, n, lane, put); else knn_rank<KNN_CAP / 64>(keys, n, lane, put);
    __builtin_amdgcn_wave_barrier();
    return n < K ? n : K;
}

// COUNT = true: only "does the query have at least K neighbours (itself included) inside its bound?" -- the whole of
// outlier_removal (filter.py:6-11: all nb_points slots filled).  No key list, no ranking, and the scan stops at the K-th hit;
// idx_out is then a byte mask [n1].  The predicates are the search's own, so the mask equals idx[:, K-1] != -1.
template <int K, bool COUNT = false>
__global__ void __launch_bounds__(KNN_BLOCK, 8) k_knn(const float* __restrict__ src, int64_t n1, const StGrid* __restrict__ g,
                                                   const uint32_t* __restrict__ cell_start, const float4* __restrict__ recs,
                                                   float r, const float* __restrict__ bound, int mode,
                                                   int64_t* __restrict__ idx_out, float* __restrict__ dist_out,
                                                   const int* __restrict__ seg_off, int nseg, int cell_order,
                                                   const uint8_t* __restrict__ valid = nullptr) {
    __shared__ uint32_t s_roff[KNN_WAVES][65], s_rfirst[KNN_WAVES][64];
    __shared__ unsigned long long s_rmask[KNN_WAVES][KNN_MASK_WORDS];  // bit p of word w: a (non-empty) row's candidates start at 64 w + p
    __shared__ __attribute__((aligned(16))) unsigned long long s_keys[KNN_WAVES][KNN_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t i = (int64_t)blockIdx.x * KNN_WAVES + wave;
    if (i >= n1) return;  // wave-uniform; the kernel has no workgroup barrier
    // valid (optional, with src == dst): only the points with valid[j] != 0 are part of the search -- the table holds only them, and
    // only they are queried (cell order: the table's total is the number of queries; the caller has cleared the output)
    if (valid && (cell_order ? i >= (int64_t)cell_start[g->ncell] : !valid[i])) return;
    // cell_order (src IS dst): wavefront p takes the p-th record of the cell-sorted point list instead of point p, so
    // neighbouring wavefronts search the same cells (their rows stay in L1 / L2); rows are written by original index, the
    // result does not depend on which wavefront computed it
    // (the record carries the point's coordinates as well: one dependent load less before the cell rows can be requested)
    float px, py, pz;
    if (cell_order) {
        const float4 me = recs[i];
        px = me.x; py = me.y; pz = me.z;
        i = (int64_t)__float_as_uint(me.w);
    } else {
        px = src[3 * i]; py = src[3 * i + 1]; pz = src[3 * i + 2];
    }
    uint32_t* roff = s_roff[wave];
    uint32_t* rfirst = s_rfirst[wave];
    unsigned long long* rmask = s_rmask[wave];
    unsigned long long* keys = s_keys[wave];
    const int seg = st_seg_find(seg_off, nseg, i);  // wave-uniform: one query per wavefront
    if (r < 0.0f) r = g->seg_r[seg];  // radius reduced on the device (st_knn_radius with r < 0): max(bound) over the query's cloud
    const float r2 = r * r;
    float reach_r = r;
    // The per-query bound is defined on the rounded root -- sqrtf(d2) <= bound (mode 1) or < bound (mode 2), graph.py:38-40 /
    // filter.py:9-10 -- and sqrtf is monotone, so it is the same as d2 <= d2_max with d2_max = the largest float whose root
    // still passes: found once per query (a few steps around bound^2), and no candidate needs a root.
    float d2_max = __uint_as_float(0x7f800000u);
    if (mode != 0) {
        const float bnd = bound[i];
        if (bnd < reach_r) reach_r = bnd;
        d2_max = knn_d2_max(bnd, mode == 2);
    }
    const float cell = g->cell;
    const float reach_f = reach_r > 0.0f ? ceilf(reach_r / cell) : 0.0f;
    int reach = reach_f < 1.0e6f ? (int)reach_f : 1000000;  // (an infinite bound reaches every cell; the clamp keeps the integer arithmetic defined)
    if (reach < 1) reach = 1;
    // a query with a NaN / infinite coordinate has no neighbours (every distance is NaN or infinite)
    const bool q_finite = fabsf(px) <= 3.0e38f && fabsf(py) <= 3.0e38f && fabsf(pz) <= 3.0e38f;
    if (!q_finite) { px = g->lo[0]; py = g->lo[1]; pz = g->lo[2]; reach = 0; d2_max = -1.0f; }
    const int cx = (int)floorf((px - g->lo[0]) / cell), cy = (int)floorf((py - g->lo[1]) / cell), cz = (int)floorf((pz - g->lo[2]) / cell);
    const int xoff = seg * g->seg_dim0;  // the cloud's slab of cells
    const int x0 = st_max(cx - reach, 0), x1 = st_min(cx + reach, g->seg_dim0 - 1);
    const int y0 = st_max(cy - reach, 0), y1 = st_min(cy + reach, g->dim[1] - 1);
    const int z0 = st_max(cz - reach, 0), z1 = st_min(cz + reach, g->dim[2] - 1);
    // rows (x, y) farther than the search radius in the xy-plane are skipped and the z-range of the others
    // is clipped to the sphere (a slightly inflated radius keeps the pruning conservative)
    const float rs = reach_r * 1.0001f + 1e-7f, rs2 = rs * rs;
    const int ny = y1 - y0 + 1, nrows = (x1 >= x0 && ny > 0 && z1 >= z0) ? (x1 - x0 + 1) * ny : 0;
    int nkeys = 0;  // wave-uniform
    // once K keys are known, a candidate that is not smaller than the K-th of them cannot be among the K nearest: it is
    // dropped before it reaches the list (around a trunk a query sees thousands of points inside its bound)
    unsigned long long thr = ~0ull;
    for (int rbase = 0; rbase < nrows && !(COUNT && nkeys >= K); rbase += 64) {
        const int rowi = rbase + lane;
        uint32_t first = 0, cnt = 0;
        if (rowi < nrows) {
            const int x = x0 + rowi / ny, y = y0 + rowi % ny;
            const float cx0 = g->lo[0] + (float)x * cell, ex = px < cx0 ? cx0 - px : (px > cx0 + cell ? px - (cx0 + cell) : 0.0f);
            const float cy0 = g->lo[1] + (float)y * cell, ey = py < cy0 ? cy0 - py : (py > cy0 + cell ? py - (cy0 + cell) : 0.0f);
            const float dxy2 = ex * ex + ey * ey;
            if (!(dxy2 > rs2)) {
                const float rz = sqrtf(rs2 - dxy2);
                // (clamped as floats: with an infinite bound rz is infinite and the integer conversion would be undefined)
                const int za = (int)fmaxf(floorf((pz - rz - g->lo[2]) / cell) - 1.0f, (float)z0),
                          zb = (int)fminf(floorf((pz + rz - g->lo[2]) / cell) + 1.0f, (float)z1);
                if (za <= zb) {
                    const int64_t row = ((int64_t)(xoff + x) * g->dim[1] + y) * g->dim[2];
                    first = cell_start[row + za];
                    cnt = cell_start[row + zb + 1] - first;
                }
            }
        }
        const uint32_t incl = knn_wave_scan(cnt, lane);
        const int total = (int)__shfl(incl, 63);
        // The rows of the chunk, EMPTY ONES DROPPED, in LDS (first candidate's number, first record); candidate t belongs to the last
        // row that starts at or before it.  Found by counting start marks: every row sets one bit (its first candidate's number) in a
        // mask of the chunk's candidates, and the 64 candidates [cb, cb + 64) read ONE mask word -- rows before cb + marks at or
        // below the lane.  (A binary search over the row offsets per candidate was six dependent LDS reads; it stays for a chunk
        // with more candidates than the mask holds.)
        const unsigned long long nonempty = __ballot(cnt != 0u);
        const int nr = __popcll(nonempty);  // wave-uniform
        const int slot = __popcll(nonempty & ((1ull << lane) - 1ull));
        const bool marks = total <= 64 * KNN_MASK_WORDS;
        __builtin_amdgcn_wave_barrier();  // the previous chunk's reads of roff / rfirst / rmask are done
        if (marks)
            for (int w = lane; w < (total + 63) / 64; w += 64) rmask[w] = 0ull;
        if (cnt != 0u) { roff[slot] = incl - cnt; rfirst[slot] = first; }
        if (lane == 63) roff[nr] = incl;
        __builtin_amdgcn_wave_barrier();
        if (marks && cnt != 0u) atomicOr(&rmask[(incl - cnt) >> 6], 1ull << ((incl - cnt) & 63u));
        __builtin_amdgcn_wave_barrier();
        int rows_before = 0;  // non-empty rows that start before cb (wave-uniform)
        for (int cb = 0; cb < total && !(COUNT && nkeys >= K); cb += 64) {
            const int t = cb + lane;
            bool ok = false;
            unsigned long long key = 0ull;
            const unsigned long long starts = marks ? rmask[cb >> 6] : 0ull;  // (one word for the whole wavefront: a broadcast)
            if (t < total) {
                int lo;
                if (marks) {
                    lo = rows_before + __popcll(starts & (~0ull >> (63 - lane))) - 1;
                } else {
                    lo = 0;
                    int hi = nr;  // row with roff[row] <= t < roff[row + 1]
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (roff[mid] <= (uint32_t)t) lo = mid; else hi = mid; }
                }
                const float4 q = recs[rfirst[lo] + ((uint32_t)t - roff[lo])];
                const float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
                float d2 = dx * dx;
                float tt = dy * dy;
                d2 = d2 + tt;
                tt = dz * dz;
                d2 = d2 + tt;
                ok = d2 < r2 && d2 <= d2_max;
                key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)__float_as_uint(q.w);  // d2 >= 0: bits order like values
                if (!COUNT) ok = ok && key < thr;
            }
            const unsigned long long bal = __ballot(ok);
            if (!COUNT && ok) keys[nkeys + __popcll(bal & ((1ull << lane) - 1ull))] = key;
            nkeys += __popcll(bal);
            if (!COUNT && nkeys > KNN_CAP - 64) {
                nkeys = knn_cut<K>(keys, nkeys, lane);
                if (nkeys == K) thr = keys[K - 1];
            }
            rows_before += __popcll(starts);
        }
    }
    if (COUNT) {
        if (lane == 0) reinterpret_cast<uint8_t*>(idx_out)[i] = nkeys >= K ? 1 : 0;
        return;
    }
    // output: rank = slot.  (d2, index) ascending, -1 / NaN padding.
    __builtin_amdgcn_wave_barrier();
    {
        auto put = [&](unsigned long long key, int rank) {
            if (rank < K) {
                idx_out[i * K + rank] = (int64_t)(unsigned)(key & 0xffffffffull);
                dist_out[i * K + rank] = sqrtf(__uint_as_float((unsigned)(key >> 32)));
            }
        };
        if (nkeys <= 64) knn_rank<1>(keys, nkeys, lane, put); else knn_rank<KNN_CAP / 64>(keys, nkeys, lane, put);
        const int found = nkeys < K ? nkeys : K;
        if (lane >= found && lane < K) {
            idx_out[i * K + lane] = (int64_t)-1;
            dist_out[i * K + lane] = __uint_as_float(0x7fc00000u);
        }
    }
}

#define KNN_MAX_CELLS (1ll << 24)
// a batch of clouds gets one slab of cells per cloud: the cap grows with the batch (up to 32 x = 2 GiB of cell table; the
// table is cleared and scanned over the cells in use only).  With the cap at 8 x, a batch of 24 clouds searched cells twice
// as coarse -- eight times the candidates -- as one cloud alone.
static inline int64_t knn_max_cells(int nseg) { return KNN_MAX_CELLS * (nseg < 1 ? 1 : (nseg > 32 ? 32 : nseg)); }

// the cell table st_grid_build will actually use: at most 128 cells per point (its own bound) -- the workspace is sized for
// that, not for the batch cap (32 x 2^24 cells = 2 GiB of table for any batch of >= 32 clouds, however few points it has)
static inline int64_t knn_cells(int64_t n2, int nseg) { return st_min64(knn_max_cells(nseg), 128 * (n2 > 0 ? n2 : 1) + 65536); }

static void knn_layout(StArena& a, int64_t n2, int nseg, StGrid** g, uint32_t** cell_start, float4** recs, char** sub, int64_t* sub_bytes) {
    *g = a.take<StGrid>(1);
    *cell_start = a.take<uint32_t>(knn_cells(n2, nseg) + 1);
    *recs = a.take<float4>(n2);
    *sub_bytes = st_grid_ws_bytes(n2, knn_cells(n2, nseg));
    *sub = a.take<char>(*sub_bytes);
}

extern "C" int64_t st_knn_workspace_bytes_seg(int64_t n_dst, int nseg) {
    StArena a(nullptr, 0);
    StGrid* g; uint32_t* cs; float4* recs; char* sub; int64_t sb;
    knn_layout(a, n_dst, nseg, &g, &cs, &recs, &sub, &sb);
    return a.used;
}
extern "C" int64_t st_knn_workspace_bytes(int64_t n_dst) { return st_knn_workspace_bytes_seg(n_dst, 1); }

// idx [n1,K] int64 (-1 pad), dist [n1,K] float32 = sqrtf(d2) (NaN pad).  bound/bound_mode: see above.
// cell_hint: preferred grid cell size (0: use r).  r < 0 (needs `bound`): the search radius is max(bound) -- what the
// callers used to read back to the host just to pass it in again; the reduction stays on the device.  cell_hint < 0
// (with r < 0): cell = max(r / -cell_hint, 1e-4).
// Batched form: src and dst hold `nseg` independent clouds each (cloud b = index ranges [src_seg_off[b], src_seg_off[b+1]) /
// [dst_seg_off[b], dst_seg_off[b+1]), device int32 arrays of nseg + 1 entries); a query only sees the points of its own
// cloud, r < 0 means max(bound) over the query's own cloud, and every cloud's rows equal what the one-cloud call returns
// for it (indices are positions in the batched dst array).
extern "C" int st_knn_radius_seg(const float* src, int64_t n1, const float* dst, int64_t n2, int K, float r, const float* bound,
                                 int bound_mode, float cell_hint, int64_t* idx, float* dist, const int32_t* src_seg_off,
                                 const int32_t* dst_seg_off, int nseg, void* ws, int64_t ws_bytes, void* stream_,
                                 float cell_mean_mult) {
    hipStream_t stream = (hipStream_t)stream_;
    ST_REQUIRE(K == 1 || K == 8 || K == 16 || K == 32 || K == 64, "knn: K must be 1, 8, 16, 32 or 64 (got %d)", K);
    static_assert(KNN_CAP >= 128, "a cut keeps K <= 64 keys and the next chunk adds up to 64");
    ST_REQUIRE(bound_mode == 0 || bound != nullptr, "knn: bound_mode needs a bound array");
    ST_REQUIRE(r >= 0.0f || bound != nullptr, "knn: r < 0 (radius = max(bound)) needs a bound array");
    ST_REQUIRE(cell_hint >= 0.0f || r < 0.0f, "knn: a relative cell size (cell_hint < 0) goes with r < 0");
    ST_REQUIRE(n2 < (1ll << 31), "knn: too many points");
    ST_REQUIRE(nseg >= 1 && nseg <= ST_MAX_SEG, "knn: 1 <= clouds per batch <= %d", ST_MAX_SEG);
    ST_REQUIRE(nseg == 1 || (src_seg_off && dst_seg_off), "knn: a batch needs the cloud offsets of src and dst");
    if (nseg == 1) { src_seg_off = nullptr; dst_seg_off = nullptr; }
    if (n1 <= 0) return ST_OK;
    StArena a(ws, ws_bytes);
    StGrid* g; uint32_t* cell_start; float4* recs; char* sub; int64_t sub_bytes;
    knn_layout(a, n2, nseg, &g, &cell_start, &recs, &sub, &sub_bytes);
    if (!a.ok() || !sub) {
        st_set_error("knn: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    const float cell_arg = cell_hint != 0.0f ? cell_hint : (r >= 0.0f ? r : -1.0f);
    ST_TRY(st_grid_build(dst, n2, cell_arg, knn_cells(n2, nseg), g, cell_start, recs, sub, sub_bytes, stream, r, bound, n1,
                         dst_seg_off, nseg, src_seg_off, cell_hint < 0.0f ? (cell_mean_mult >= 0.0f ? cell_mean_mult : KNN_MEAN_MULT) : 0.0f, nullptr));
    dim3 grid((unsigned)st_div_up(n1, KNN_WAVES)), block(KNN_BLOCK);
    const int cell_order = src == dst && n1 == n2 ? 1 : 0;
    if (K == 1)
        hipLaunchKernelGGL((k_knn<1>), grid, block, 0, stream, src, n1, (const StGrid*)g, (const uint32_t*)cell_start,
                           (const float4*)recs, r, bound, bound_mode, idx, dist, src_seg_off, nseg, cell_order);
    else if (K == 8)
        hipLaunchKernelGGL((k_knn<8>), grid, block, 0, stream, src, n1, (const StGrid*)g, (const uint32_t*)cell_start,
                           (const float4*)recs, r, bound, bound_mode, idx, dist, src_seg_off, nseg, cell_order);
    else if (K == 16)
        hipLaunchKernelGGL((k_knn<16>), grid, block, 0, stream, src, n1, (const StGrid*)g, (const uint32_t*)cell_start,
                           (const float4*)recs, r, bound, bound_mode, idx, dist, src_seg_off, nseg, cell_order);
    else if (K == 32)
        hipLaunchKernelGGL((k_knn<32>), grid, block, 0, stream, src, n1, (const StGrid*)g, (const uint32_t*)cell_start,
                           (const float4*)recs, r, bound, bound_mode, idx, dist, src_seg_off, nseg, cell_order);
    else  // 64: the reference's default widths (graph.py:12 knn K = 50, :36 nn_graph K = 40) are columns of this one
        hipLaunchKernelGGL((k_knn<64>), grid, block, 0, stream, src, n1, (const StGrid*)g, (const uint32_t*)cell_start,
                           (const float4*)recs, r, bound, bound_mode, idx, dist, src_seg_off, nseg, cell_order);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// mask [n1] uint8: 1 iff the query has at least K points of its own cloud (itself included when src == dst) with
// d2 < r*r and, per bound_mode, sqrtf(d2) <= / < bound[i] -- i.e. st_knn_radius_seg(...).idx[:, K-1] != -1 without building
// the neighbour lists.  replaces: skeleton/filter.py:6-11 (outlier_removal's FRNN query + mask arithmetic).
extern "C" int st_radius_count_seg(const float* src, int64_t n1, const float* dst, int64_t n2, int K, float r, const float* bound,
                                   int bound_mode, float cell_hint, uint8_t* mask, const int32_t* src_seg_off,
                                   const int32_t* dst_seg_off, int nseg, void* ws, int64_t ws_bytes, void* stream_,
                                   float cell_mean_mult, const uint8_t* valid) {
    hipStream_t stream = (hipStream_t)stream_;
    ST_REQUIRE(valid == nullptr || (src == dst && n1 == n2), "radius_count: a validity mask needs src == dst");
    ST_REQUIRE(K == 8, "radius_count: K must be 8 (outlier_removal's nb_points; got %d)", K);
    ST_REQUIRE(bound_mode == 0 || bound != nullptr, "radius_count: bound_mode needs a bound array");
    ST_REQUIRE(r >= 0.0f || bound != nullptr, "radius_count: r < 0 (radius = max(bound)) needs a bound array");
    ST_REQUIRE(cell_hint >= 0.0f || r < 0.0f, "radius_count: a relative cell size (cell_hint < 0) goes with r < 0");
    ST_REQUIRE(n2 < (1ll << 31), "radius_count: too many points");
    ST_REQUIRE(nseg >= 1 && nseg <= ST_MAX_SEG, "radius_count: 1 <= clouds per batch <= %d", ST_MAX_SEG);
    ST_REQUIRE(nseg == 1 || (src_seg_off && dst_seg_off), "radius_count: a batch needs the cloud offsets of src and dst");
    if (nseg == 1) { src_seg_off = nullptr; dst_seg_off = nullptr; }
    if (n1 <= 0) return ST_OK;
    StArena a(ws, ws_bytes);
    StGrid* g; uint32_t* cell_start; float4* recs; char* sub; int64_t sub_bytes;
    knn_layout(a, n2, nseg, &g, &cell_start, &recs, &sub, &sub_bytes);
    if (!a.ok() || !sub) {
        st_set_error("radius_count: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    const float cell_arg = cell_hint != 0.0f ? cell_hint : (r >= 0.0f ? r : -1.0f);
    ST_TRY(st_grid_build(dst, n2, cell_arg, knn_cells(n2, nseg), g, cell_start, recs, sub, sub_bytes, stream, r, bound, n1,
                         dst_seg_off, nseg, src_seg_off, cell_hint < 0.0f ? (cell_mean_mult >= 0.0f ? cell_mean_mult : KNN_MEAN_MULT) : 0.0f, valid));
    if (valid) (void)hipMemsetAsync(mask, 0, (size_t)n1, stream);  // (the points outside the search get no query: their answer is "no")
    hipLaunchKernelGGL((k_knn<8, true>), dim3((unsigned)st_div_up(n1, KNN_WAVES)), dim3(KNN_BLOCK), 0, stream, src, n1,
                       (const StGrid*)g, (const uint32_t*)cell_start, (const float4*)recs, r, bound, bound_mode,
                       reinterpret_cast<int64_t*>(mask), (float*)nullptr, src_seg_off, nseg, src == dst && n1 == n2 ? 1 : 0, valid);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

extern "C" int st_knn_radius(const float* src, int64_t n1, const float* dst, int64_t n2, int K, float r, const float* bound,
                             int bound_mode, float cell_hint, int64_t* idx, float* dist, void* ws, int64_t ws_bytes,
                             void* stream_) {
    return st_knn_radius_seg(src, n1, dst, n2, K, r, bound, bound_mode, cell_hint, idx, dist, nullptr, nullptr, 1, ws, ws_bytes,
                             stream_, -1.0f);
}
