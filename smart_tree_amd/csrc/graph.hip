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
__global__ void __launch_bounds__(GR_BLOCK) k_bbox(const float* xyz, int64_t n, unsigned* box /*[6] ord lo, ord hi*/) {
    __shared__ unsigned lo[3], hi[3];
    if (threadIdx.x < 3) { lo[threadIdx.x] = 0xffffffffu; hi[threadIdx.x] = 0u; }
    __syncthreads();
    unsigned l[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, h[3] = {0u, 0u, 0u};
    GR_LOOP(i, n)
        for (int a = 0; a < 3; a++) {
            const unsigned o = st_f2ord(xyz[3 * i + a]);
            l[a] = o < l[a] ? o : l[a];
            h[a] = o > h[a] ? o : h[a];
        }
    for (int a = 0; a < 3; a++) { atomicMin(&lo[a], l[a]); atomicMax(&hi[a], h[a]); }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&box[threadIdx.x], lo[threadIdx.x]); atomicMax(&box[3 + threadIdx.x], hi[threadIdx.x]); }
}
__global__ void __launch_bounds__(GR_BLOCK) k_centre(const float* xyz, int64_t n, const unsigned* box, float* out) {
    float shift[3];
    for (int a = 0; a < 3; a++) {
        const float mn = st_ord2f(box[a]), mx = st_ord2f(box[3 + a]);
        const float half = (mx - mn) / 2.0f;
        const float centre = mn + half;
        shift[a] = -centre + (a == 1 ? half : 0.0f);
    }
    GR_LOOP(i, n)
        for (int a = 0; a < 3; a++) out[3 * i + a] = xyz[3 * i + a] + shift[a];
}

// out [n,3]; scratch: 6 x uint32
extern "C" int st_centre_cloud(const float* xyz, int64_t n, float* out, void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return ST_OK;
    StArena a(ws, ws_bytes);
    unsigned* box = a.take<unsigned>(6);
    if (!box) { st_set_error("centre_cloud: workspace too small"); return ST_ERR_WORKSPACE; }
    (void)hipMemsetAsync(box, 0xff, 3 * sizeof(unsigned), stream);
    (void)hipMemsetAsync(box + 3, 0, 3 * sizeof(unsigned), stream);
    hipLaunchKernelGGL(k_bbox, dim3(gr_grid(n) < 1024 ? gr_grid(n) : 1024), dim3(GR_BLOCK), 0, stream, xyz, n, box);
    hipLaunchKernelGGL(k_centre, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, xyz, n, (const unsigned*)box, out);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ------------------------------------------------------------------------------ make_edges ---
__global__ void __launch_bounds__(GR_BLOCK) k_edge_count(const int64_t* idx, int64_t n, int K, uint32_t* cnt) {
    GR_LOOP(i, n) {
        uint32_t c = 0;
        for (int k = 0; k < K; k++) c += idx[i * K + k] > 0;
        cnt[i] = c;
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_edge_emit(const int64_t* idx, const float* dist, int64_t n, int K,
                                                        const uint32_t* off, int64_t* edges, float* w) {
    GR_LOOP(i, n) {
        uint32_t o = off[i];
        for (int k = 0; k < K; k++) {
            int64_t j = idx[i * K + k];
            if (j > 0) { edges[2 * (int64_t)o] = i; edges[2 * (int64_t)o + 1] = j; w[o] = dist[i * K + k]; o++; }
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
extern "C" int st_make_edges(const int64_t* idx, const float* dist, int64_t n, int K, int64_t* edges, float* w,
                             int64_t* n_edges_host, void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges_host) *n_edges_host = 0;
    if (n <= 0) return ST_OK;
    StArena a(ws, ws_bytes);
    uint32_t* cnt = a.take<uint32_t>(n + 1);
    int64_t sb = st_scan_ws_bytes(n);
    char* sw = a.take<char>(sb);
    if (!cnt || !sw) { st_set_error("make_edges: workspace too small"); return ST_ERR_WORKSPACE; }
    hipLaunchKernelGGL(k_edge_count, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, idx, n, K, cnt);
    ST_TRY(st_exclusive_scan_u32(cnt, cnt, n, cnt + n, sw, sb, stream));
    hipLaunchKernelGGL(k_edge_emit, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, idx, dist, n, K, (const uint32_t*)cnt, edges, w);
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
__global__ void __launch_bounds__(GR_BLOCK) k_cc_hook(const int64_t* edges, int64_t E, unsigned* label, int sample) {
    GR_LOOP(e, E) {
        if (sample && (e & 7) != 0) continue;
        const unsigned u = (unsigned)edges[2 * e], v = (unsigned)edges[2 * e + 1];
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
extern "C" int st_connected_components(const int64_t* edges, int64_t E, int64_t n, int32_t* labels, void* ws,
                                       int64_t ws_bytes, void* stream_) {
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
        hipLaunchKernelGGL(k_cc_hook, dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, edges, E, label, 1);
        hipLaunchKernelGGL(k_cc_compress, dim3(gr_grid(n)), dim3(GR_BLOCK), 0, stream, label, n);
        hipLaunchKernelGGL(k_cc_hook, dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, edges, E, label, 0);
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
        if (root) atomicAdd(kept, size[i]);  // vertices in kept components: known together with their number
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_cl_rootlist(const uint32_t* flag_off, const int* label, const uint32_t* size,
                                                          int64_t n, uint32_t minv, uint32_t* key, uint32_t* root) {
    GR_LOOP(i, n) if (label[i] == (int)i && size[i] >= minv) { key[flag_off[i]] = 0xffffffffu - size[i]; root[flag_off[i]] = (uint32_t)i; }
}
__global__ void __launch_bounds__(GR_BLOCK) k_cl_rank(const uint32_t* root_sorted, const uint32_t* key_sorted, int64_t C,
                                                      int* rank_of_root, uint32_t* comp_size) {
    GR_LOOP(c, C) { rank_of_root[root_sorted[c]] = (int)c; comp_size[c] = 0xffffffffu - key_sorted[c]; }
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
    a.take<uint32_t>(n + 1);  // size (+ kept-vertex counter)
    a.take<uint32_t>(n + 1);  // flag / offsets
    a.take<uint32_t>(n);      // key
    a.take<uint32_t>(n);      // val
    a.take<int>(n);           // rank_of_root
    a.take<char>(st_sort_ws_bytes(n) > st_scan_ws_bytes(n + 1) ? st_sort_ws_bytes(n) : st_scan_ws_bytes(n + 1));
    return a.used;
}

// Outputs (caller-allocated, capacity n each): comp_size [C], comp_off [C+1], vert_order [m] (original
// vertex ids grouped by component, ascending inside), new_id [n] (-1 for dropped vertices).
extern "C" int st_component_layout(const int32_t* labels, int64_t n, int min_vertices, int32_t* comp_size, int32_t* comp_off,
                                   int32_t* vert_order, int32_t* new_id, int64_t* n_comp_host, int64_t* n_kept_host,
                                   void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    *n_comp_host = 0;
    *n_kept_host = 0;
    if (n <= 0) return ST_OK;
    StArena a(ws, ws_bytes);
    uint32_t* size = a.take<uint32_t>(n + 1);  // size[n] = number of vertices in kept components
    uint32_t* flag = a.take<uint32_t>(n + 1);
    uint32_t* key = a.take<uint32_t>(n);
    uint32_t* val = a.take<uint32_t>(n);
    int* rank_of_root = a.take<int>(n);
    int64_t sb = st_sort_ws_bytes(n) > st_scan_ws_bytes(n + 1) ? st_sort_ws_bytes(n) : st_scan_ws_bytes(n + 1);
    char* sw = a.take<char>(sb);
    if (!a.ok() || !sw) { st_set_error("component_layout: workspace too small"); return ST_ERR_WORKSPACE; }
    const unsigned g = gr_grid(n);
    uint32_t minv = min_vertices > 0 ? (uint32_t)min_vertices : 0u;
    (void)hipMemsetAsync(size, 0, (n + 1) * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(k_cl_sizes, dim3(g), dim3(GR_BLOCK), 0, stream, labels, n, size);
    hipLaunchKernelGGL(k_cl_rootflag, dim3(g), dim3(GR_BLOCK), 0, stream, labels, (const uint32_t*)size, n, minv, flag, size + n);
    ST_TRY(st_exclusive_scan_u32(flag, flag, n, flag + n, sw, sb, stream));
    uint32_t C = 0, m = 0;  // ONE round trip for both counts (a blocking read-back costs ~1 ms beside other clouds' kernels)
    (void)hipMemcpyAsync(&C, flag + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(&m, size + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    st_stream_wait(stream);
    ST_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_fill_i32, dim3(g), dim3(GR_BLOCK), 0, stream, new_id, n, -1);
    if (C == 0) { ST_CHECK_LAUNCH(); return ST_OK; }
    hipLaunchKernelGGL(k_cl_rootlist, dim3(g), dim3(GR_BLOCK), 0, stream, (const uint32_t*)flag, labels, (const uint32_t*)size, n,
                       minv, key, val);
    ST_TRY(st_radix_sort_pairs_u32(key, val, C, 32, sw, sb, stream));  // size desc; stable => root id asc on ties
    hipLaunchKernelGGL(k_fill_i32, dim3(g), dim3(GR_BLOCK), 0, stream, rank_of_root, n, -1);
    hipLaunchKernelGGL(k_cl_rank, dim3(gr_grid(C)), dim3(GR_BLOCK), 0, stream, (const uint32_t*)val, (const uint32_t*)key,
                       (int64_t)C, rank_of_root, (uint32_t*)comp_size);
    // comp_off = exclusive scan of comp_size (+ total at [C])
    ST_TRY(st_exclusive_scan_u32((const uint32_t*)comp_size, (uint32_t*)comp_off, C, (uint32_t*)comp_off + C, sw, sb, stream));
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
__device__ __forceinline__ void csr_edge_ends(const int64_t* edges, int64_t E, const int* new_id, int64_t e, int* a, int* b) {
    *a = -1; *b = -1;
    if (e >= E) return;
    const int64_t u = edges[2 * e], v = edges[2 * e + 1];
    if (u == v) return;  // self loops (every vertex but 0 has one) never matter for paths
    const int x = new_id[u], y = new_id[v];
    if (x < 0 || y < 0) return;
    *a = x; *b = y;
}

__global__ void __launch_bounds__(GR_BLOCK) k_csr_count(const int64_t* edges, int64_t E, const int* new_id, uint32_t* deg) {
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); base < E; base += (int64_t)gridDim.x * blockDim.x) {
        int a, b;
        csr_edge_ends(edges, E, new_id, base + lane, &a, &b);
        gr_run_add(deg, a);
        if (b >= 0) atomicAdd(&deg[b], 1u);  // targets are scattered: no runs to aggregate
    }
}
__global__ void __launch_bounds__(GR_BLOCK) k_csr_fill(const int64_t* edges, const float* w, int64_t E, const int* new_id,
                                                       uint32_t* cursor, uint32_t* col, float* wgt) {
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); base < E; base += (int64_t)gridDim.x * blockDim.x) {
        int a, b;
        csr_edge_ends(edges, E, new_id, base + lane, &a, &b);
        const uint32_t pa = gr_run_add(cursor, a);
        if (a < 0) continue;
        const float we = w[base + lane];
        col[pa] = (uint32_t)b; wgt[pa] = we;
        const uint32_t pb = atomicAdd(&cursor[b], 1u);
        col[pb] = (uint32_t)a; wgt[pb] = we;
    }
}

extern "C" int64_t st_component_csr_workspace_bytes(int64_t m) {
    StArena a(nullptr, 0);
    a.take<uint32_t>(m + 1);
    a.take<char>(st_scan_ws_bytes(m + 1));
    return a.used;
}

// row_off [m+1] uint32, col / wgt capacity 2*E.  Adjacency order inside a row is unspecified
// (atomic cursors); every consumer is order independent.
extern "C" int st_component_csr(const int64_t* edges, const float* w, int64_t E, const int32_t* new_id, int64_t m,
                                uint32_t* row_off, uint32_t* col, float* wgt, void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (m <= 0) return ST_OK;
    StArena a(ws, ws_bytes);
    uint32_t* cursor = a.take<uint32_t>(m + 1);
    int64_t sb = st_scan_ws_bytes(m + 1);
    char* sw = a.take<char>(sb);
    if (!cursor || !sw) { st_set_error("component_csr: workspace too small"); return ST_ERR_WORKSPACE; }
    (void)hipMemsetAsync(row_off, 0, (m + 1) * sizeof(uint32_t), stream);
    if (E > 0) hipLaunchKernelGGL(k_csr_count, dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, edges, E, new_id, row_off);
    ST_TRY(st_exclusive_scan_u32(row_off, row_off, m + 1, nullptr, sw, sb, stream));
    (void)hipMemcpyAsync(cursor, row_off, (m + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
    if (E > 0) hipLaunchKernelGGL(k_csr_fill, dim3(gr_grid(E)), dim3(GR_BLOCK), 0, stream, edges, w, E, new_id, cursor, col, wgt);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
