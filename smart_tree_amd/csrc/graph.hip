// Graph construction for the skeleton stage: edge list, connected components, component layout.
//
//   st_make_edges            smart_tree/skeleton/graph.py:52-60 (make_edges, incl. the `idx > 0` filter)
//   st_connected_components  smart_tree/data_types/graph.py:32-37 (cugraph.connected_components)
//   st_component_layout      smart_tree/data_types/graph.py:38-51 (size filter, sort by size desc) +
//                            smart_tree/skeleton/skeletonize.py:60-71 / graph.py:94-104 (vertex
//                            renumbering by rank) -- done for ALL components at once, on the device,
//                            instead of a host loop over every label with cudf/pandas round trips
//   st_component_csr         undirected adjacency (both directions) in the renumbered vertex space
//
// Component labels are canonical (smallest member vertex id), component order is size descending
// then label ascending, vertices inside a component keep ascending original id: the same choices
// as oracle/skeleton_oracle.py.
#include "st_common.h"
#include "st_grid.h"  // st_seg_find, ST_MAX_SEG

#define GR_BLOCK 256
static inline unsigned gr_grid(int64_t n) {
    int64_t g = st_div_up(n > 0 ? n : 1, GR_BLOCK);
    return (unsigned)(g < 8192 ? g : 8192);
}
#define GR_LOOP(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// --------------------------------------------------------------- medial points and radii ---
// Cloud.medial_pts / Cloud.radius (smart_tree/data_types/cloud.py:229-231,254-256) evaluated in one
// pass with a fixed float32 operation order: medial = xyz + mv, radius = sqrtf((x*x + y*y) + z*z)
// (correctly rounded sqrt) -- the exact values the oracle uses, independent of torch's math library.
__global__ void __launch_bounds__(GR_BLOCK) k_medial(const float* xyz, const float* mv, int64_t n, float* medial, float* radius) {
    GR_LOOP(i, n) {
        const float x = mv[3 * i], y = mv[3 * i + 1], z = mv[3 * i + 2];
        medial[3 * i] = xyz[3 * i] + x;
        medial[3 * i + 1] = xyz[3 * i + 1] + y;
        medial[3 * i + 2] = xyz[3 * i + 2] + z;
        float s = x * x;
        float t = y * y;
        s = s + t;
        t = z * z;
        s = s + t;
        radius[i] = sqrtf(s);
    }
}

extern "C" int st_medial_points(const float* xyz, const float* mv, int64_t n, float* medial, float* radius, void* stream_) {
    if (n <= 0) return ST_OK;
    hipLaunchKernelGGL(k_medial, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, (hipStream_t)stream_, xyz, mv, n, medial, radius);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ------------------------------------------------------------------------------ CentreCloud ---
// CentreCloud (smart_tree/dataset/augmentations.py:38-41 over Cloud.bbox, data_types/cloud.py:222-227):
// half = (max - min) / 2, centre = min + half, xyz += -centre + (0, half_y, 0) -- same float32 operation
// order as torch evaluates it, but one bounding-box pass (LDS reduction) + one translate pass instead of
// four strided torch reductions.
// blockIdx.y = cloud of a batched call (seg_off == nullptr: one cloud of n points); box [nseg][6] = ord lo, ord hi.
// A cloud is streamed as float4 -- four points per three 16-byte loads -- over the 16-byte aligned body of its range;
// the <= 3 points before and after it (and everything, if the base pointer is not 16-byte aligned) go one float at a time.
struct GrRange {
    int64_t i0, i1;  // the cloud's points
    int64_t a0, a1;  // aligned body [a0, a1): a0 % 4 == 0, (a1 - a0) % 4 == 0
    int64_t nq;      // float4s in the body; float4 q starts at axis q % 3: (x y z x) (y z x y) (z x y z)
};
__device__ __forceinline__ GrRange gr_range(const void* p0, const void* p1, int64_t n, const int* seg_off, int seg) {
    GrRange r;
    r.i0 = seg_off ? seg_off[seg] : 0;
    r.i1 = seg_off ? seg_off[seg + 1] : n;
    const bool vec = ((((uintptr_t)p0) | ((uintptr_t)p1)) & 15) == 0;
    r.a0 = vec ? st_min((r.i0 + 3) & ~(int64_t)3, r.i1) : r.i1;
    r.a1 = r.a0 + ((r.i1 - r.a0) & ~(int64_t)3);
    r.nq = (r.a1 - r.a0) / 4 * 3;
    return r;
}
// j-th point outside the aligned body (j < (a0 - i0) + (i1 - a1))
__device__ __forceinline__ int64_t gr_edge_point(const GrRange& r, int64_t j) { return j < r.a0 - r.i0 ? r.i0 + j : r.a1 + (j - (r.a0 - r.i0)); }

__global__ void __launch_bounds__(GR_BLOCK) k_bbox(const float* xyz, int64_t n, unsigned* box, const int* seg_off) {
    __shared__ unsigned lo[3], hi[3];
    if (threadIdx.x < 3) { lo[threadIdx.x] = 0xffffffffu; hi[threadIdx.x] = 0u; }
    __syncthreads();
    const int seg = blockIdx.y;
    const GrRange r = gr_range(xyz, nullptr, n, seg_off, seg);
    unsigned lx = 0xffffffffu, ly = 0xffffffffu, lz = 0xffffffffu, hx = 0u, hy = 0u, hz = 0u;
    const float4* v = reinterpret_cast<const float4*>(xyz + 3 * r.a0);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < r.nq; q += (int64_t)gridDim.x * blockDim.x) {
        const float4 f = v[q];
        const int ax = (int)(q % 3);
        const unsigned o0 = st_f2ord(f.x), o1 = st_f2ord(f.y), o2 = st_f2ord(f.z), o3 = st_f2ord(f.w);
        const unsigned a = st_min(o0, o3), b = st_max(o0, o3);  // f.x and f.w share an axis
        lx = st_min(lx, ax == 0 ? a : (ax == 1 ? o2 : o1)); hx = st_max(hx, ax == 0 ? b : (ax == 1 ? o2 : o1));
        ly = st_min(ly, ax == 0 ? o1 : (ax == 1 ? a : o2)); hy = st_max(hy, ax == 0 ? o1 : (ax == 1 ? b : o2));
        lz = st_min(lz, ax == 0 ? o2 : (ax == 1 ? o1 : a)); hz = st_max(hz, ax == 0 ? o2 : (ax == 1 ? o1 : b));
    }
    const int64_t ne = (r.a0 - r.i0) + (r.i1 - r.a1);
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ne; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = gr_edge_point(r, j);
        const unsigned ox = st_f2ord(xyz[3 * i]), oy = st_f2ord(xyz[3 * i + 1]), oz = st_f2ord(xyz[3 * i + 2]);
        lx = st_min(lx, ox); hx = st_max(hx, ox); ly = st_min(ly, oy); hy = st_max(hy, oy); lz = st_min(lz, oz); hz = st_max(hz, oz);
    }
    // wavefront reduction first: six LDS atomics per wavefront instead of per lane
    for (int d = 32; d > 0; d >>= 1) {
        lx = st_min(lx, (unsigned)__shfl_xor((int)lx, d)); ly = st_min(ly, (unsigned)__shfl_xor((int)ly, d)); lz = st_min(lz, (unsigned)__shfl_xor((int)lz, d));
        hx = st_max(hx, (unsigned)__shfl_xor((int)hx, d)); hy = st_max(hy, (unsigned)__shfl_xor((int)hy, d)); hz = st_max(hz, (unsigned)__shfl_xor((int)hz, d));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&lo[0], lx); atomicMin(&lo[1], ly); atomicMin(&lo[2], lz);
        atomicMax(&hi[0], hx); atomicMax(&hi[1], hy); atomicMax(&hi[2], hz);
    }
    __syncthreads();
    if (threadIdx.x < 3 && r.i1 > r.i0) { atomicMin(&box[6 * seg + threadIdx.x], lo[threadIdx.x]); atomicMax(&box[6 * seg + 3 + threadIdx.x], hi[threadIdx.x]); }
}
__global__ void __launch_bounds__(GR_BLOCK) k_centre(const float* xyz, int64_t n, const unsigned* box, float* out, const int* seg_off) {
    const int seg = blockIdx.y;
    const GrRange r = gr_range(xyz, out, n, seg_off, seg);
    float shift[3];
    for (int a = 0; a < 3; a++) {
        const float mn = st_ord2f(box[6 * seg + a]), mx = st_ord2f(box[6 * seg + 3 + a]);
        const float half = (mx - mn) / 2.0f;
        const float centre = mn + half;
        shift[a] = -centre + (a == 1 ? half : 0.0f);
    }
    const float4* v = reinterpret_cast<const float4*>(xyz + 3 * r.a0);
    float4* o = reinterpret_cast<float4*>(out + 3 * r.a0);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < r.nq; q += (int64_t)gridDim.x * blockDim.x) {
        float4 f = v[q];
        const int ax = (int)(q % 3);
        const float s0 = ax == 0 ? shift[0] : (ax == 1 ? shift[1] : shift[2]);
        const float s1 = ax == 0 ? shift[1] : (ax == 1 ? shift[2] : shift[0]);
        const float s2 = ax == 0 ? shift[2] : (ax == 1 ? shift[0] : shift[1]);
        f.x = f.x + s0; f.y = f.y + s1; f.z = f.z + s2; f.w = f.w + s0;
        o[q] = f;
    }
    const int64_t ne = (r.a0 - r.i0) + (r.i1 - r.a1);
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ne; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = gr_edge_point(r, j);
        for (int a = 0; a < 3; a++) out[3 * i + a] = xyz[3 * i + a] + shift[a];
    }
}

__global__ void k_box_init(unsigned* box, int nseg) {
    for (int i = threadIdx.x; i < 6 * nseg; i += blockDim.x) box[i] = i % 6 < 3 ? 0xffffffffu : 0u;  // lo = +max, hi = 0 (order-preserving bits)
}

// Batched form: `nseg` clouds in one array, cloud b = points [seg_off[b], seg_off[b+1]) (device int32 [nseg+1]); every
// cloud is centred on ITS OWN bounding box, exactly as the one-cloud call does.  out [n,3]; scratch: 6 x uint32 per cloud.
extern "C" int st_centre_cloud_seg(const float* xyz, int64_t n, const int32_t* seg_off, int nseg, float* out, void* ws,
                                   int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return ST_OK;
    ST_REQUIRE(nseg >= 1 && nseg <= ST_MAX_SEG && (nseg == 1 || seg_off), "centre_cloud: 1 <= clouds per batch <= %d", ST_MAX_SEG);
    if (nseg == 1) seg_off = nullptr;
    StArena a(ws, ws_bytes);
    unsigned* box = a.take<unsigned>(6 * (int64_t)nseg);
    if (!box) { st_set_error("centre_cloud: workspace too small"); return ST_ERR_WORKSPACE; }
    // (one launch: two runtime memsets per cloud were 40 of the ~100 fills of a 20-cloud launch set, 256 us in front of its first kernel)
    hipLaunchKernelGGL(k_box_init, dim3(1), dim3(64), 0, stream, box, nseg);
    // float4 items per cloud ~ 0.75 n / nseg; a few hundred workgroups per cloud keep the final atomics few
    const int64_t per = st_div_up(st_div_up(3 * n, 4), nseg);
    const unsigned gx = (unsigned)st_min64(st_div_up(per > 0 ? per : 1, GR_BLOCK), 2048 / (nseg < 8 ? nseg : 8) + 1);
    hipLaunchKernelGGL(k_bbox, dim3(gx, (unsigned)nseg), dim3(GR_BLOCK), 0, stream, xyz, n, box, seg_off);
    hipLaunchKernelGGL(k_centre, dim3((unsigned)st_min64(st_div_up(per > 0 ? per : 1, GR_BLOCK), 8192), (unsigned)nseg), dim3(GR_BLOCK), 0, stream,
                       xyz, n, (const unsigned*)box, out, seg_off);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
extern "C" int st_centre_cloud(const float* xyz, int64_t n, float* out, void* ws, int64_t ws_bytes, void* stream_) {
    return st_centre_cloud_seg(xyz, n, nullptr, 1, out, ws, ws_bytes, stream_);
}

// ------------------------------------------------------------------------------ make_edges ---
// `idx > 0` (graph.py:59) is a test on the index INSIDE the cloud: in a batched call vertex 0 of cloud b is seg_off[b]
// (a neighbour always belongs to the query's own cloud, so "local index > 0" reads idx > seg_off[cloud of i]).
__global__ void __launch_bounds__(GR_BLOCK) k_edge_count(const int64_t* idx, int64_t n, int K, uint32_t* cnt, const int* seg_off,
                                                         int nseg) {
    GR_LOOP(i, n) {
        const int64_t first = seg_off ? seg_off[st_seg_find(seg_off, nseg, i)] : 0;
        uint32_t c = 0;
        for (int k = 0; k < K; k++) c += idx[i * K + k] > first;
        cnt[i] = c;
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_edge_emit(const int64_t* idx, const float* dist, int64_t n, int K,
                                                        const uint32_t* off, int64_t* edges, float* w, const int* seg_off,
                                                        int nseg) {
    GR_LOOP(i, n) {
        const int64_t first = seg_off ? seg_off[st_seg_find(seg_off, nseg, i)] : 0;
        uint32_t o = off[i];
        for (int k = 0; k < K; k++) {
            int64_t j = idx[i * K + k];
            if (j > first) { edges[2 * (int64_t)o] = i; edges[2 * (int64_t)o + 1] = j; w[o] = dist[i * K + k]; o++; }
        }
    }
}

extern "C" int64_t st_make_edges_workspace_bytes(int64_t n) {
    StArena a(nullptr, 0);
    a.take<uint32_t>(n + 1);
    a.take<char>(st_scan_ws_bytes(n));
    return a.used;
}

// the unused tail of a capacity-sized edge list (a kNN graph fills ~99 % of n*K: only the tail is written)
__global__ void __launch_bounds__(GR_BLOCK) k_edge_pad(int64_t* edges, float* w, const uint32_t* total, int64_t cap) {
    for (int64_t e = (int64_t)*total + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cap; e += (int64_t)gridDim.x * blockDim.x) {
        edges[2 * e] = 0;
        edges[2 * e + 1] = 0;
        w[e] = 0.0f;
    }
}

// idx/dist [n,K] from st_knn_radius (after the caller's radius filter); edges [n*K,2] int64, w [n*K].
// n_edges_host == NULL: no read-back of the edge count (a blocking round trip costs ~1 ms beside other clouds' kernels);
// the n*K - E unused entries are (0, 0) self loops of weight 0, which st_connected_components and st_component_csr ignore
// (a real edge always has dst > 0: the reference's `idx > 0` filter, graph.py:59).
// Batched form: seg_off [nseg + 1] (device) = the clouds' vertex ranges; the padding stays (0, 0).
extern "C" int st_make_edges_seg(const int64_t* idx, const float* dist, int64_t n, int K, int64_t* edges, float* w,
                                 int64_t* n_edges_host, const int32_t* seg_off, int nseg, void* ws, int64_t ws_bytes,
                                 void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges_host) *n_edges_host = 0;
    if (n <= 0) return ST_OK;
    ST_REQUIRE(nseg >= 1 && nseg <= ST_MAX_SEG && (nseg == 1 || seg_off), "make_edges: 1 <= clouds per batch <= %d", ST_MAX_SEG);
    if (nseg == 1) seg_off = nullptr;
    StArena a(ws, ws_bytes);
    uint32_t* cnt = a.take<uint32_t>(n + 1);
    int64_t sb = st_scan_ws_bytes(n);
    char* sw = a.take<char>(sb);
    if (!cnt || !sw) { st_set_error("make_edges: workspace too small"); return ST_ERR_WORKSPACE; }
    hipLaunchKernelGGL(k_edge_count, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, idx, n, K, cnt, seg_off, nseg);
    ST_TRY(st_exclusive_scan_u32(cnt, cnt, n, cnt + n, sw, sb, stream));
    hipLaunchKernelGGL(k_edge_emit, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, idx, dist, n, K, (const uint32_t*)cnt, edges, w,
                       seg_off, nseg);
    if (!n_edges_host) {  // no count on the host: the tail [E, n*K) of the capacity-sized list becomes (0, 0) edges of weight 0
        hipLaunchKernelGGL(k_edge_pad, dim3(64), dim3(GR_BLOCK), 0, stream, edges, w, (const uint32_t*)(cnt + n), n * (int64_t)K);
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    uint32_t total = 0;
    (void)hipMemcpyAsync(&total, cnt + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    st_stream_wait(stream);
    ST_CHECK_LAUNCH();
    *n_edges_host = total;
    return ST_OK;
}

extern "C" int st_make_edges(const int64_t* idx, const float* dist, int64_t n, int K, int64_t* edges, float* w,
                             int64_t* n_edges_host, void* ws, int64_t ws_bytes, void* stream_) {
    return st_make_edges_seg(idx, dist, n, K, edges, w, n_edges_host, nullptr, 1, ws, ws_bytes, stream_);
}

// ---------------------------------------------------------------------------- edge sources ---
// The component / CSR kernels take their edges either from an explicit list (edges [E,2] int64 + w [E], st_make_edges'
// output: what the reference's Graph holds) or straight from the neighbour search (idx / dist [n,K]): edge e = (e / K,
// idx[e]), kept iff idx[e] > first vertex of the source's cloud -- make_edges' own rule (graph.py:59), so both sources
// describe the same edge set and the int64 edge list (20 bytes per edge, written once and read three times) need not exist.
struct GrEdges {
    const int64_t* edges;  // explicit list (or nullptr)
    const float* w;
    const int64_t* idx;    // neighbour table (or nullptr)
    const float* dist;
    int K, k_shift;        // k_shift >= 0: K == 1 << k_shift
    const int* first_of;   // [n] first vertex of every vertex's cloud (batched calls; nullptr: vertex 0)
};
static inline int gr_shift_of(int K) { int sh = 0; while ((1 << sh) < K) sh++; return (1 << sh) == K ? sh : -1; }
// the two ends of edge e; u == v marks "no edge" (padding, self loop, filtered neighbour)
__device__ __forceinline__ void gr_edge(const GrEdges& G, int64_t e, int64_t* u, int64_t* v) {
    if (G.idx == nullptr) { *u = G.edges[2 * e]; *v = G.edges[2 * e + 1]; return; }
    const int64_t i = G.k_shift >= 0 ? (e >> G.k_shift) : e / G.K, j = G.idx[e];
    const int64_t first = G.first_of ? G.first_of[i] : 0;  // (the K lanes of a source read one word)
    *u = i;
    *v = j > first ? j : i;
}
__device__ __forceinline__ float gr_edge_weight(const GrEdges& G, int64_t e) { return G.idx ? G.dist[e] : G.w[e]; }

// ------------------------------------------------------------------ connected components ---
// Union-find over the edge list.  A root with the larger KEY is hooked under the one with the smaller key, where
// key(x) = x * CC_MUL mod 2^32 (a bijection; idx() is its inverse).  With the vertex id itself as key the spatial
// ordering of the ids makes the hooks form chains thousands of roots long (tree k under k-1 under k-2 ...), and
// every find walks them with dependent loads; a scrambled order keeps the forest shallow.  label[] holds the
// parent's key while the forest is built; a last pass rewrites it to the smallest vertex id of the component.
#define CC_MUL 0x9E3779B1u
#define CC_INV 0x0E8B2F51u  // CC_MUL * CC_INV == 1 (mod 2^32)
__device__ __forceinline__ unsigned cc_key(unsigned x) { return x * CC_MUL; }
__device__ __forceinline__ unsigned cc_idx(unsigned k) { return k * CC_INV; }

__device__ __forceinline__ unsigned cc_find(unsigned* label, unsigned k) {
    // chase to the root; path halving (a parent's key is always smaller than the child's, so writing a
    // grandparent is always a valid shortcut, whatever other lanes do concurrently)
    unsigned p = __atomic_load_n(&label[cc_idx(k)], __ATOMIC_RELAXED);
    while (p != k) {
        const unsigned gp = __atomic_load_n(&label[cc_idx(p)], __ATOMIC_RELAXED);
        if (gp != p) atomicMin(&label[cc_idx(k)], gp);
        k = p;
        p = gp;
    }
    return k;
}
__global__ void __launch_bounds__(GR_BLOCK) k_cc_init(unsigned* label, int64_t n) { GR_LOOP(i, n) label[i] = cc_key((unsigned)i); }
// Hooking (ECL-CC style): the root with the larger key is hooked under the other with a CAS that only succeeds
// while it is still a root; on failure the walk continues from whatever it was hooked to, so every
// processed edge ends up with both ends in one tree -- no "until nothing changes" loop, no host sync.
// Afforest-style schedule: a sampled eighth of the edges is linked first, the forest is flattened, and
// the full pass then dismisses almost every edge with two loads (both ends already carry the same root:
// a tree cloud is one giant component) instead of chasing pointers for it.
__global__ void __launch_bounds__(GR_BLOCK) k_cc_hook(GrEdges G, int64_t E, unsigned* label, int sample) {
    GR_LOOP(e, E) {
        // a pseudo-random eighth of the edges (not every eighth: in a neighbour table entries 0 and 8 of every row would be
        // picked, and entry 0 is the vertex itself)
        if (sample && (((uint32_t)e * 2654435761u) >> 29) != 0u) continue;
        int64_t u64, v64;
        gr_edge(G, e, &u64, &v64);
        const unsigned u = (unsigned)u64, v = (unsigned)v64;
        if (u == v) continue;
        if (!sample) {  // labels were flattened by the preceding compress
            // plain (cacheable) loads: a stale pair that still compares equal was and stays in one tree; anything else
            // goes through the coherent walk below
            if (label[u] == label[v]) continue;
        }
        unsigned a = cc_find(label, cc_key(u)), b = cc_find(label, cc_key(v));
        while (a != b) {
            if (a < b) { const unsigned t = a; a = b; b = t; }  // a = larger key
            const unsigned seen = atomicCAS(&label[cc_idx(a)], a, b);
            if (seen == a) break;  // hooked
            a = seen;              // a had already been hooked: continue from there
        }
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_cc_compress(unsigned* label, int64_t n) {
    GR_LOOP(i, n) { const unsigned r = cc_find(label, cc_key((unsigned)i)); label[i] = r; }
}
// canonical labels: smallest member id of every tree (label[] is flat: it holds root keys)
__global__ void __launch_bounds__(GR_BLOCK) k_cc_minid_init(unsigned* minid, int64_t n) { GR_LOOP(i, n) minid[i] = 0xffffffffu; }
__global__ void __launch_bounds__(GR_BLOCK) k_cc_minid(const unsigned* label, int64_t n, unsigned* minid) {
    // lanes of a wave that share a root are represented by their first lane (ids ascend with the lane): a big tree
    // is one component, and per-lane atomics on its one word would serialise the launch
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const bool valid = i < n;
        const unsigned r = valid ? cc_idx(label[i]) : 0xffffffffu;
        unsigned long long todo = __ballot(valid);
        while (todo) {  // wave-uniform
            const int leader = __ffsll(todo) - 1;
            const unsigned lr = __shfl(r, leader);
            const unsigned long long same = __ballot(valid && r == lr);
            if (lane == leader) atomicMin(&minid[lr], (unsigned)i);
            todo &= ~same;
        }
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_cc_relabel(unsigned* label, int64_t n, const unsigned* minid) {
    GR_LOOP(i, n) label[i] = minid[cc_idx(label[i])];
}

extern "C" int64_t st_connected_components_workspace_bytes(int64_t n) { return (n > 0 ? n : 1) * (int64_t)sizeof(unsigned) + 256; }

// labels [n] int32 out: smallest vertex id of each vertex's component.
static int cc_run(GrEdges G, int64_t E, int64_t n, int32_t* labels, void* ws, int64_t ws_bytes, void* stream_);

extern "C" int st_connected_components(const int64_t* edges, int64_t E, int64_t n, int32_t* labels, void* ws,
                                       int64_t ws_bytes, void* stream_) {
    GrEdges G = {edges, nullptr, nullptr, nullptr, 0, -1, nullptr};
    return cc_run(G, E, n, labels, ws, ws_bytes, stream_);
}

// The same components straight from the neighbour search: idx [n,K] (st_knn_radius_seg's output after the caller's radius
// filter); first_of [n] (batched calls) = the first vertex of every vertex's cloud, NULL = one cloud (vertex 0).
extern "C" int st_connected_components_knn(const int64_t* idx, int64_t n, int K, const int32_t* first_of,
                                           int32_t* labels, void* ws, int64_t ws_bytes, void* stream_) {
    ST_REQUIRE(K >= 1, "cc(knn): K must be positive");
    GrEdges G = {nullptr, nullptr, idx, nullptr, K, gr_shift_of(K), first_of};
    return cc_run(G, n * (int64_t)K, n, labels, ws, ws_bytes, stream_);
}

static int cc_run(GrEdges G, int64_t E, int64_t n, int32_t* labels, void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ST_REQUIRE(n < (1ll << 31), "cc: too many vertices");
    if (n <= 0) return ST_OK;
    StArena ar(ws, ws_bytes);
    unsigned* minid = ar.take<unsigned>(n);
    if (!ar.ok() || !minid) {
        st_set_error("cc: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)ar.used);
        return ST_ERR_WORKSPACE;
    }
    unsigned* label = (unsigned*)labels;
    hipLaunchKernelGGL(k_cc_init, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, label, n);
    if (E > 0) {
        hipLaunchKernelGGL(k_cc_hook, dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, G, E, label, 1);
        hipLaunchKernelGGL(k_cc_compress, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, label, n);
        hipLaunchKernelGGL(k_cc_hook, dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, G, E, label, 0);
    }
    hipLaunchKernelGGL(k_cc_compress, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, label, n);
    hipLaunchKernelGGL(k_cc_minid_init, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, minid, n);
    hipLaunchKernelGGL(k_cc_minid, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, (const unsigned*)label, n, minid);
    hipLaunchKernelGGL(k_cc_relabel, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, label, n, (const unsigned*)minid);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ---------------------------------------------------------------------- component layout ---
// component sizes; lanes of a wave that share a label are counted by one atomic (a big tree is one
// component: per-lane atomics on its root word would serialise the whole launch)
__global__ void __launch_bounds__(GR_BLOCK) k_cl_sizes(const int* label, int64_t n, uint32_t* size) {
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const bool valid = i < n;
        const int lab = valid ? label[i] : -1;
        unsigned long long todo = __ballot(valid);
        while (todo) {  // wave-uniform
            const int leader = __ffsll(todo) - 1;
            const int l = __shfl(lab, leader);
            const unsigned long long same = __ballot(valid && lab == l);
            if (lane == leader) atomicAdd(&size[l], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_cl_rootflag(const int* label, const uint32_t* size, int64_t n, uint32_t minv,
                                                          uint32_t* flag, uint32_t* kept) {
    GR_LOOP(i, n) {
        const bool root = label[i] == (int)i && size[i] >= minv;
        flag[i] = root ? 1u : 0u;
        if (root) { atomicAdd(kept, size[i]); atomicMax(kept + 1, size[i]); }  // vertices in kept components and the largest one: known together with their number
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_cl_rootlist(const uint32_t* flag_off, const int* label, const uint32_t* size,
                                                          int64_t n, uint32_t minv, uint32_t* key, uint32_t* root) {
    GR_LOOP(i, n) if (label[i] == (int)i && size[i] >= minv) { key[flag_off[i]] = 0xffffffffu - size[i]; root[flag_off[i]] = (uint32_t)i; }
}
__global__ void __launch_bounds__(GR_BLOCK) k_cl_rank(const uint32_t* root_sorted, const uint32_t* size, int64_t C,
                                                      int* rank_of_root, uint32_t* comp_size) {
    GR_LOOP(c, C) { rank_of_root[root_sorted[c]] = (int)c; comp_size[c] = size[root_sorted[c]]; }
}
// batched call: second (stable) sort key = cloud of the component's root; afterwards comp_seg / the per-cloud ranges
__global__ void __launch_bounds__(GR_BLOCK) k_cl_segkey(const uint32_t* root_sorted, int64_t C, const int* seg_off, int nseg,
                                                        uint32_t* key) {
    GR_LOOP(c, C) key[c] = (uint32_t)st_seg_find(seg_off, nseg, root_sorted[c]);
}
// comp_seg [C]; comp_seg_off [nseg+1] = component ranges of the clouds; vert_seg_off [nseg+1] = their vertex ranges in the
// renumbered space.  One workgroup (C is a few hundred).
__global__ void __launch_bounds__(GR_BLOCK) k_cl_segments(const uint32_t* root_sorted, int C, const int* seg_off, int nseg,
                                                          const int32_t* comp_off, int32_t* comp_seg, int32_t* comp_seg_off,
                                                          int32_t* vert_seg_off) {
    for (int b = threadIdx.x; b <= nseg; b += blockDim.x) comp_seg_off[b] = C;  // clouds after the last component
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int s = st_seg_find(seg_off, nseg, root_sorted[c]);
        comp_seg[c] = s;
        const int prev = c > 0 ? st_seg_find(seg_off, nseg, root_sorted[c - 1]) : -1;
        for (int b = prev + 1; b <= s; b++) comp_seg_off[b] = c;  // clouds without components share their successor's start
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= nseg; b += blockDim.x) vert_seg_off[b] = comp_off[comp_seg_off[b]];
}
__global__ void __launch_bounds__(GR_BLOCK) k_cl_vflag(const int* label, const int* rank_of_root, int64_t n, uint32_t* flag) {
    GR_LOOP(i, n) flag[i] = rank_of_root[label[i]] >= 0 ? 1u : 0u;
}
__global__ void __launch_bounds__(GR_BLOCK) k_cl_vlist(const uint32_t* flag_off, const int* label, const int* rank_of_root,
                                                       int64_t n, uint32_t* vkey, uint32_t* vid) {
    GR_LOOP(i, n) {
        int r = rank_of_root[label[i]];
        if (r >= 0) { vkey[flag_off[i]] = (uint32_t)r; vid[flag_off[i]] = (uint32_t)i; }
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_cl_newid(const uint32_t* vid_sorted, int64_t m, int* new_id, int32_t* vert_order) {
    GR_LOOP(p, m) { new_id[vid_sorted[p]] = (int)p; vert_order[p] = (int32_t)vid_sorted[p]; }
}
__global__ void __launch_bounds__(GR_BLOCK) k_fill_i32(int* p, int64_t n, int v) { GR_LOOP(i, n) p[i] = v; }

extern "C" int64_t st_component_layout_workspace_bytes(int64_t n) {
    StArena a(nullptr, 0);
    a.take<uint32_t>(n + 2);  // size (+ kept-vertex counter, largest kept component)
    a.take<uint32_t>(n + 1);  // flag / offsets
    a.take<uint32_t>(n);      // key
    a.take<uint32_t>(n);      // val
    a.take<int>(n);           // rank_of_root
    a.take<char>(st_sort_ws_bytes(n) > st_scan_ws_bytes(n + 1) ? st_sort_ws_bytes(n) : st_scan_ws_bytes(n + 1));
    return a.used;
}

// Outputs (caller-allocated, capacity n each): comp_size [C], comp_off [C+1], vert_order [m] (original
// vertex ids grouped by component, ascending inside), new_id [n] (-1 for dropped vertices).
//
// Batched form: seg_off [nseg + 1] (device) = the clouds' vertex ranges.  Components are ordered cloud by cloud (inside a
// cloud: size descending, smallest member ascending -- the one-cloud order), so every cloud's components, and its
// vertices in the renumbered space, are contiguous.  Extra outputs (device): comp_seg [C capacity n], comp_seg_off
// [nseg + 1], vert_seg_off [nseg + 1].
extern "C" int st_component_layout_seg(const int32_t* labels, int64_t n, int min_vertices, const int32_t* seg_off, int nseg,
                                       int32_t* comp_size, int32_t* comp_off, int32_t* vert_order, int32_t* new_id,
                                       int32_t* comp_seg, int32_t* comp_seg_off, int32_t* vert_seg_off,
                                       int64_t* n_comp_host, int64_t* n_kept_host, void* ws, int64_t ws_bytes, void* stream_,
                                       int64_t* max_comp_host /*optional: vertices of the largest kept component*/) {
    hipStream_t stream = (hipStream_t)stream_;
    *n_comp_host = 0;
    *n_kept_host = 0;
    if (max_comp_host) *max_comp_host = 0;
    ST_REQUIRE(nseg >= 1 && nseg <= ST_MAX_SEG, "component_layout: 1 <= clouds per batch <= %d", ST_MAX_SEG);
    ST_REQUIRE(nseg == 1 || (seg_off && comp_seg && comp_seg_off && vert_seg_off), "component_layout: a batch needs seg_off and the per-cloud outputs");
    if (n <= 0) {
        if (nseg > 1) { (void)hipMemsetAsync(comp_seg_off, 0, (nseg + 1) * sizeof(int32_t), stream); (void)hipMemsetAsync(vert_seg_off, 0, (nseg + 1) * sizeof(int32_t), stream); }
        return ST_OK;
    }
    StArena a(ws, ws_bytes);
    uint32_t* size = a.take<uint32_t>(n + 2);  // size[n] = number of vertices in kept components, size[n + 1] = the largest one's
    uint32_t* flag = a.take<uint32_t>(n + 1);
    uint32_t* key = a.take<uint32_t>(n);
    uint32_t* val = a.take<uint32_t>(n);
    int* rank_of_root = a.take<int>(n);
    int64_t sb = st_sort_ws_bytes(n) > st_scan_ws_bytes(n + 1) ? st_sort_ws_bytes(n) : st_scan_ws_bytes(n + 1);
    char* sw = a.take<char>(sb);
    if (!a.ok() || !sw) { st_set_error("component_layout: workspace too small"); return ST_ERR_WORKSPACE; }
    const unsigned g = gr_grid(n);
    uint32_t minv = min_vertices > 0 ? (uint32_t)min_vertices : 0u;
    (void)hipMemsetAsync(size, 0, (n + 2) * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(k_cl_sizes, dim3(g), dim3(GR_BLOCK), 0, stream, labels, n, size);
    hipLaunchKernelGGL(k_cl_rootflag, dim3(g), dim3(GR_BLOCK), 0, stream, labels, (const uint32_t*)size, n, minv, flag, size + n);
    ST_TRY(st_exclusive_scan_u32(flag, flag, n, flag + n, sw, sb, stream));
    uint32_t C = 0, mk[2] = {0, 0};  // ONE round trip for the counts (a blocking read-back costs ~1 ms beside other clouds' kernels)
    (void)hipMemcpyAsync(&C, flag + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(mk, size + n, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    st_stream_wait(stream);
    const uint32_t m = mk[0];
    if (max_comp_host) *max_comp_host = mk[1];
    ST_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_fill_i32, dim3(g), dim3(GR_BLOCK), 0, stream, new_id, n, -1);
    if (C == 0) {
        if (nseg > 1) { (void)hipMemsetAsync(comp_seg_off, 0, (nseg + 1) * sizeof(int32_t), stream); (void)hipMemsetAsync(vert_seg_off, 0, (nseg + 1) * sizeof(int32_t), stream); }
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    hipLaunchKernelGGL(k_cl_rootlist, dim3(g), dim3(GR_BLOCK), 0, stream, (const uint32_t*)flag, labels, (const uint32_t*)size, n,
                       minv, key, val);
    ST_TRY(st_radix_sort_pairs_u32(key, val, C, 32, sw, sb, stream));  // size desc; stable => root id asc on ties
    if (nseg > 1) {  // ... then stably by cloud
        int sbits = 1;
        while ((1 << sbits) < nseg) sbits++;
        hipLaunchKernelGGL(k_cl_segkey, dim3(gr_grid(C)), dim3(GR_BLOCK), 0, stream, (const uint32_t*)val, (int64_t)C, seg_off, nseg, key);
        ST_TRY(st_radix_sort_pairs_u32(key, val, C, sbits, sw, sb, stream));
    }
    hipLaunchKernelGGL(k_fill_i32, dim3(g), dim3(GR_BLOCK), 0, stream, rank_of_root, n, -1);
    hipLaunchKernelGGL(k_cl_rank, dim3(gr_grid(C)), dim3(GR_BLOCK), 0, stream, (const uint32_t*)val, (const uint32_t*)size,
                       (int64_t)C, rank_of_root, (uint32_t*)comp_size);
    // comp_off = exclusive scan of comp_size (+ total at [C])
    ST_TRY(st_exclusive_scan_u32((const uint32_t*)comp_size, (uint32_t*)comp_off, C, (uint32_t*)comp_off + C, sw, sb, stream));
    if (nseg > 1)
        hipLaunchKernelGGL(k_cl_segments, dim3(1), dim3(GR_BLOCK), 0, stream, (const uint32_t*)val, (int)C, seg_off, nseg,
                           (const int32_t*)comp_off, comp_seg, comp_seg_off, vert_seg_off);
    // vertices of kept components, ascending id, then stable sort by component rank
    hipLaunchKernelGGL(k_cl_vflag, dim3(g), dim3(GR_BLOCK), 0, stream, labels, (const int*)rank_of_root, n, flag);
    ST_TRY(st_exclusive_scan_u32(flag, flag, n, flag + n, sw, sb, stream));
    hipLaunchKernelGGL(k_cl_vlist, dim3(g), dim3(GR_BLOCK), 0, stream, (const uint32_t*)flag, labels, (const int*)rank_of_root, n,
                       key, val);
    int bits = 1;
    while ((1u << bits) < C && bits < 32) bits++;
    ST_TRY(st_radix_sort_pairs_u32(key, val, m, bits, sw, sb, stream));
    hipLaunchKernelGGL(k_cl_newid, dim3(gr_grid(m)), dim3(GR_BLOCK), 0, stream, (const uint32_t*)val, (int64_t)m, new_id,
                       vert_order);
    ST_CHECK_LAUNCH();
    *n_comp_host = C;
    *n_kept_host = m;
    return ST_OK;
}

extern "C" int st_component_layout(const int32_t* labels, int64_t n, int min_vertices, int32_t* comp_size, int32_t* comp_off,
                                   int32_t* vert_order, int32_t* new_id, int64_t* n_comp_host, int64_t* n_kept_host,
                                   void* ws, int64_t ws_bytes, void* stream_) {
    return st_component_layout_seg(labels, n, min_vertices, nullptr, 1, comp_size, comp_off, vert_order, new_id, nullptr, nullptr,
                                   nullptr, n_comp_host, n_kept_host, ws, ws_bytes, stream_, nullptr);
}

// ---------------------------------------------------------------------------- component CSR ---
// make_edges emits the edge list grouped by source vertex (up to K consecutive edges share `u`): a plain atomicAdd per
// edge end would serialise ~16 lanes on one counter.  Consecutive lanes with the same key elect their first lane, which
// adds the run length once; the others take their offset inside the run.  key < 0: lane takes no part.
// Must be reached by all 64 lanes.  Returns the lane's slot (leader's old counter value + rank in the run).
__device__ __forceinline__ uint32_t gr_run_add(uint32_t* ctr, int key) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1);
    const bool head = key >= 0 && (lane == 0 || prev != key);
    const unsigned long long heads = __ballot(head), active = __ballot(key >= 0);
    const unsigned long long upto = heads & (~0ull >> (63 - lane));     // heads at or below this lane
    const int h = 63 - __clzll((long long)(upto | 1ull));                // my run's first lane (lane 0 if none: inactive)
    const unsigned long long stop = (heads | ~active) & ~(~0ull >> (63 - h));  // first lane above h that is not in the run
    const int end = stop ? __ffsll((long long)stop) - 1 : 64;
    uint32_t base = 0;
    if (head) base = atomicAdd(&ctr[key], (uint32_t)(end - lane));
    base = __shfl(base, h);
    return base + (uint32_t)(lane - h);
}

// the two ends of edge e in the renumbered vertex space, -1/-1 when the edge is dropped
__device__ __forceinline__ void csr_edge_ends(const GrEdges& G, int64_t E, const int* new_id, int64_t e, int* a, int* b) {
    *a = -1; *b = -1;
    if (e >= E) return;
    int64_t u, v;
    gr_edge(G, e, &u, &v);
    if (u == v) return;  // self loops (every vertex but 0 has one) never matter for paths
    const int x = new_id[u], y = new_id[v];
    if (x < 0 || y < 0) return;
    *a = x; *b = y;
}

__global__ void __launch_bounds__(GR_BLOCK) k_csr_count(GrEdges G, int64_t E, const int* new_id, uint32_t* deg) {
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); base < E; base += (int64_t)gridDim.x * blockDim.x) {
        int a, b;
        csr_edge_ends(G, E, new_id, base + lane, &a, &b);
        gr_run_add(deg, a);
        if (b >= 0) atomicAdd(&deg[b], 1u);  // targets are scattered: no runs to aggregate
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_csr_fill(GrEdges G, int64_t E, const int* new_id,
                                                       uint32_t* cursor, uint32_t* col, float* wgt) {
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); base < E; base += (int64_t)gridDim.x * blockDim.x) {
        int a, b;
        csr_edge_ends(G, E, new_id, base + lane, &a, &b);
        const uint32_t pa = gr_run_add(cursor, a);
        if (a < 0) continue;
        const float we = gr_edge_weight(G, base + lane);
        col[pa] = (uint32_t)b; wgt[pa] = we;
        const uint32_t pb = atomicAdd(&cursor[b], 1u);
        col[pb] = (uint32_t)a; wgt[pb] = we;
    }
}

// ---- adjacency straight from the neighbour tables, WITHOUT the duplicate of a mutual pair (round 3) ----
// In a kNN graph most pairs are mutual (v in kNN(u) and u in kNN(v)); the edge-list build above stores such a pair twice in
// both rows (the reference's cugraph graph is undirected and keeps one edge per pair).  Here row(x) = the valid forward
// neighbours of x, in table order, followed by the "reverse-only" ones (w lists x, x does not list w):
//   * forward entries need no atomics at all: their count is a ballot over the K lanes of the source's table row and their
//     slot is row_off[x] + rank;
//   * only a reverse-only edge costs a scattered atomic (count) and a returning one (fill) -- about one edge in five;
//   * whether u -> v is mutual is ONE 8K-byte read of v's table row (K = 16: a 128-byte line), done in the count pass; the
//     fill pass gets the renumbered target and the flag back as one int32 per edge.
// The same set of (neighbour, weight) pairs per row as the edge-list build, each pair once (d(u,v) is computed from the same
// squares on either side, so the two copies of a mutual pair carry the same float): distances, predecessors and the tree
// distance do not change, the SSSP relaxes ~40 % fewer entries.
#define CSRK_MUTUAL 0x40000000
struct __attribute__((aligned(16))) GrI64x2 { int64_t x, y; };  // one 16-byte load of two table entries
template <int K>
__global__ void __launch_bounds__(GR_BLOCK) k_csrk_count(const int64_t* __restrict__ idx, int64_t n, const int* __restrict__ first_of,
                                                         const int* __restrict__ new_id, uint32_t* deg, uint32_t* fwdc,
                                                         int32_t* __restrict__ tgt) {
    const int64_t E = n * K;
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); base < E; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = base + lane;
        const int64_t i = e / K;
        int x = -1, y = -1;
        bool mutual = false;
        if (e < E) {
            x = new_id[i];  // (the K lanes of a source read one word)
            const int64_t j = idx[e], first = first_of ? first_of[i] : 0;
            if (x >= 0 && j > first && j != i) {
                y = new_id[j];
                if (y >= 0 && i > first) {  // does j list i?  (that entry is valid by the same rule: i > first, i != j, both kept)
                    const int64_t* row = idx + j * K;
                    if (K % 2 == 0) {
#pragma unroll
                        for (int k = 0; k < K; k += 2) {
                            const GrI64x2 r = *reinterpret_cast<const GrI64x2*>(row + k);
                            mutual = mutual || r.x == i || r.y == i;
                        }
                    } else {
                        for (int k = 0; k < K; k++) mutual = mutual || row[k] == i;
                    }
                }
            }
        }
        const bool fwd = y >= 0;
        const unsigned long long fb = __ballot(fwd);
        if (e < E) tgt[e] = fwd ? (y | (mutual ? CSRK_MUTUAL : 0)) : -1;
        if (fwd && !mutual) atomicAdd(&deg[y], 1u);  // reverse-only entry of row y
        if ((lane % K) == 0 && x >= 0) {  // leader of the table row (every kept vertex has exactly one): its forward entries at once
            const unsigned cnt = (unsigned)__popcll(K == 64 ? fb : (fb >> lane) & ((1ull << (K & 63)) - 1ull));
            fwdc[x] = cnt;
            if (cnt) atomicAdd(&deg[x], cnt);
        }
    }
}
template <int K>
__global__ void __launch_bounds__(GR_BLOCK) k_csrk_fill(const int64_t n, const float* __restrict__ dist, const int* __restrict__ new_id,
                                                        const int32_t* __restrict__ tgt, const uint32_t* __restrict__ row_off,
                                                        uint32_t* cursor, uint32_t* __restrict__ col, float* __restrict__ wgt) {
    const int64_t E = n * K;
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); base < E; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = base + lane;
        const int32_t t = e < E ? tgt[e] : -1;
        const bool fwd = t >= 0;
        const unsigned long long fb = __ballot(fwd);
        if (!fwd) continue;
        const int x = new_id[e / K], y = t & ~CSRK_MUTUAL;
        const float we = dist[e];
        const int k = lane % K;
        const unsigned long long mine = K == 64 ? fb : (fb >> (lane - k)) & ((1ull << (K & 63)) - 1ull);
        const uint32_t pa = row_off[x] + (uint32_t)__popcll(mine & ((1ull << k) - 1ull));
        col[pa] = (uint32_t)y; wgt[pa] = we;
        if (!(t & CSRK_MUTUAL)) {
            const uint32_t pb = atomicAdd(&cursor[y], 1u);
            col[pb] = (uint32_t)x; wgt[pb] = we;
        }
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_csrk_cursor(const uint32_t* __restrict__ row_off, const uint32_t* __restrict__ fwdc, int64_t m,
                                                          uint32_t* __restrict__ cursor) {
    GR_LOOP(v, m) cursor[v] = row_off[v] + fwdc[v];
}

extern "C" int64_t st_component_csr_workspace_bytes(int64_t m) {
    StArena a(nullptr, 0);
    a.take<uint32_t>(m + 1);
    a.take<char>(st_scan_ws_bytes(m + 1));
    return a.used;
}
// the table form (st_component_csr_knn) also keeps a forward count per vertex and one int32 per table entry
extern "C" int64_t st_component_csr_knn_workspace_bytes(int64_t m, int64_t n, int K) {
    StArena a(nullptr, 0);
    a.take<uint32_t>(m + 1);
    a.take<char>(st_scan_ws_bytes(m + 1));
    a.take<uint32_t>(m + 1);
    a.take<int32_t>(n * (int64_t)(K > 0 ? K : 1));
    return a.used;
}

// row_off [m+1] uint32, col / wgt capacity 2*E.  Adjacency order inside a row is unspecified
// (atomic cursors); every consumer is order independent.
static int csr_run(GrEdges G, int64_t E, const int32_t* new_id, int64_t m, uint32_t* row_off, uint32_t* col, float* wgt, void* ws,
                   int64_t ws_bytes, void* stream_);

extern "C" int st_component_csr(const int64_t* edges, const float* w, int64_t E, const int32_t* new_id, int64_t m,
                                uint32_t* row_off, uint32_t* col, float* wgt, void* ws, int64_t ws_bytes, void* stream_) {
    GrEdges G = {edges, w, nullptr, nullptr, 0, -1, nullptr};
    return csr_run(G, E, new_id, m, row_off, col, wgt, ws, ws_bytes, stream_);
}

// The same adjacency straight from the neighbour search (idx / dist [n,K]); col / wgt capacity 2 * n * K.  Every
// (neighbour, weight) pair of a row ONCE (see k_csrk_count above): forward neighbours in table order, then the reverse-only
// ones in unspecified order.  Workspace: st_component_csr_knn_workspace_bytes; with the smaller st_component_csr_workspace_bytes
// (or a K that is not a power of two <= 64) the call falls back to the edge-list build, which keeps both copies of a mutual pair.
template <int K>
static void csrk_launch(const int64_t* idx, const float* dist, int64_t n, const int32_t* first_of, const int32_t* new_id, int64_t m,
                        uint32_t* row_off, uint32_t* col, float* wgt, uint32_t* cursor, uint32_t* fwdc, int32_t* tgt, char* sw, int64_t sb,
                        hipStream_t stream, int* rc) {
    const int64_t E = n * K;
    (void)hipMemsetAsync(row_off, 0, (m + 1) * sizeof(uint32_t), stream);
    hipLaunchKernelGGL((k_csrk_count<K>), dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, idx, n, first_of, new_id, row_off, fwdc, tgt);
    *rc = st_exclusive_scan_u32(row_off, row_off, m + 1, nullptr, sw, sb, stream);
    if (*rc != ST_OK) return;
    hipLaunchKernelGGL(k_csrk_cursor, dim3(gr_grid(m)), dim3(GR_BLOCK), 0, stream, (const uint32_t*)row_off, (const uint32_t*)fwdc, m, cursor);
    hipLaunchKernelGGL((k_csrk_fill<K>), dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, n, dist, new_id, (const int32_t*)tgt,
                       (const uint32_t*)row_off, cursor, col, wgt);
}
extern "C" int st_component_csr_knn(const int64_t* idx, const float* dist, int64_t n, int K, const int32_t* first_of,
                                    const int32_t* new_id, int64_t m, uint32_t* row_off, uint32_t* col, float* wgt, void* ws,
                                    int64_t ws_bytes, void* stream_) {
    ST_REQUIRE(K >= 1, "csr(knn): K must be positive");
    hipStream_t stream = (hipStream_t)stream_;
    if (m > 0 && n > 0 && gr_shift_of(K) >= 0 && K <= 64 && ws_bytes >= st_component_csr_knn_workspace_bytes(m, n, K)) {
        ST_REQUIRE(m < CSRK_MUTUAL, "csr(knn): too many vertices");
        StArena a(ws, ws_bytes);
        uint32_t* cursor = a.take<uint32_t>(m + 1);
        const int64_t sb = st_scan_ws_bytes(m + 1);
        char* sw = a.take<char>(sb);
        uint32_t* fwdc = a.take<uint32_t>(m + 1);
        int32_t* tgt = a.take<int32_t>(n * (int64_t)K);
        if (!cursor || !sw || !fwdc || !tgt) { st_set_error("component_csr(knn): workspace too small"); return ST_ERR_WORKSPACE; }
        int rc = ST_OK;
#define CSRK_CASE(K_) case K_: csrk_launch<K_>(idx, dist, n, first_of, new_id, m, row_off, col, wgt, cursor, fwdc, tgt, sw, sb, stream, &rc); break;
        switch (K) { CSRK_CASE(1) CSRK_CASE(2) CSRK_CASE(4) CSRK_CASE(8) CSRK_CASE(16) CSRK_CASE(32) CSRK_CASE(64) }
#undef CSRK_CASE
        if (rc != ST_OK) return rc;
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    GrEdges G = {nullptr, nullptr, idx, dist, K, gr_shift_of(K), first_of};
    return csr_run(G, n * (int64_t)K, new_id, m, row_off, col, wgt, ws, ws_bytes, stream_);
}

static int csr_run(GrEdges G, int64_t E, const int32_t* new_id, int64_t m, uint32_t* row_off, uint32_t* col, float* wgt, void* ws,
                   int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (m <= 0) return ST_OK;
    StArena a(ws, ws_bytes);
    uint32_t* cursor = a.take<uint32_t>(m + 1);
    int64_t sb = st_scan_ws_bytes(m + 1);
    char* sw = a.take<char>(sb);
    if (!cursor || !sw) { st_set_error("component_csr: workspace too small"); return ST_ERR_WORKSPACE; }
    (void)hipMemsetAsync(row_off, 0, (m + 1) * sizeof(uint32_t), stream);
    if (E > 0) hipLaunchKernelGGL(k_csr_count, dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, G, E, new_id, row_off);
    ST_TRY(st_exclusive_scan_u32(row_off, row_off, m + 1, nullptr, sw, sb, stream));
    (void)hipMemcpyAsync(cursor, row_off, (m + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
    if (E > 0) hipLaunchKernelGGL(k_csr_fill, dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, G, E, new_id, cursor, col, wgt);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
