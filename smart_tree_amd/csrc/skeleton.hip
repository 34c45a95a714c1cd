// Per-component skeleton extraction: SSSP, predecessor tree, tree distance, greedy sample_tree.
//
// Reference (smart_tree/skeleton): process_subgraph skeletonize.py:57-95, shortest_paths
// shortest_path.py:12-21 (cugraph.sssp), pred_graph :46-55 + second sssp skeletonize.py:80-85,
// sample_tree / trace_route / select_path_points path.py:9-140.  There every component costs
// several cugraph builds, a cudf->pandas->torch round trip, and per branch a host-synchronising
// argmax, a Python pointer chase with an O(|T|) tensor membership test per step and an ALL-points
// FRNN query.  Here ONE persistent workgroup owns a component from root search to the last branch:
// no host round trip, the termination set is a flag array, and the K=1 "nearest path vertex" query
// is inverted (path vertices scan the uniform grid around themselves and race with a packed
// atomicMin(d2 | path position)), which visits only points near the path and yields the same
// assignment.  Components run concurrently, one workgroup each (largest first).
// Semantics and tie-breaks: oracle/skeleton_oracle.c (so_sssp, so_tree_distance, so_sample_tree).
//
// Everything a workgroup shares through global memory is touched with relaxed atomic
// loads/stores or atomic RMWs (L2 coherent); workgroups never talk to each other.
#include "st_common.h"
#include "st_grid.h"

#define SK_MAX_WAVES 16
#define SK_LIFT 24  // 2^24 hops bound the deepest predecessor chain
#define SK_EMPTY64 0xffffffffffffffffull

struct SkArgs {
    int C;
    const int* comp_off;    // [C+1] offsets into the renumbered vertex space
    const float* pts;       // [m,3] medial points
    const float* rad;       // [m] raw radius (cloud.radius)
    const float* ysurf;     // [m] y of the surface point (root = lowest surface point)
    const uint32_t* row_off;  // [m+1]
    const uint32_t* col;      // renumbered neighbour ids
    const float* wgt;
    const StGrid* grid;
    const uint32_t* cell_start;
    const float4* recs;
    // outputs
    float* dist;       // [m]
    int* pred;         // [m] component-local predecessor, -1 at the root
    int* root_local;   // [C]
    int* branch_parent;  // [m] (component slice)
    int* branch_off;     // [m] offset into the component's path_verts slice
    int* branch_len;     // [m]
    int* n_branches;     // [C]
    int* path_verts;     // [m] component-local vertex ids, root side first
    int* branch_of;      // [m] final branch id per point (-1: none)
    // scratch [m]
    unsigned* dist_ord;
    unsigned* stamp;     // SSSP: queued-in-round marker; preds: resolution round
    unsigned* q0;
    unsigned* q1;
    float* alloc;        // sample_tree's `distances` (-1 once allocated)
    unsigned* term;      // termination set
    unsigned long long* best;  // claim race: (d2 bits << 32) | path position
    unsigned* touched;
    int* anc;            // [SK_LIFT][m] binary-lifting table over the predecessor tree (component-local ids)
    int64_t m;
    long long* ticks;    // optional [C][8] phase timestamps (wall_clock64), NULL = off
};

// agent scope: served by the L2 (never a stale per-CU L1 line, never a trip to HBM for a hot word)
template <class T>
__device__ __forceinline__ T ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ void st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld(const float* p) { return __uint_as_float(ld((const unsigned*)p)); }
__device__ __forceinline__ void st(float* p, float v) { st((unsigned*)p, __float_as_uint(v)); }

// workgroup-wide max of a 64-bit key; every thread must call; lds needs SK_MAX_WAVES words
__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, unsigned long long* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    for (int d = 32; d > 0; d >>= 1) {
        unsigned long long o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    unsigned long long r = lds[0];
    for (int w = 1; w < nw; w++) r = lds[w] > r ? lds[w] : r;
    return r;
}

__device__ __forceinline__ float sk_dist(const float* a, const float* b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float s = dx * dx;
    float t = dy * dy;
    s = s + t;
    t = dz * dz;
    s = s + t;
    return s;
}

// ------------------------------------------------------------------------------------ SSSP ---
__device__ void sk_sssp(const SkArgs& A, int base, int n, int root) {
    __shared__ unsigned s_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    const unsigned inf = st_f2ord(__uint_as_float(0x7f800000u));
    for (int v = tid; v < n; v += blockDim.x) { st(&A.dist_ord[base + v], v == root ? st_f2ord(0.0f) : inf); st(&A.stamp[base + v], 0u); }
    unsigned* q = A.q0 + base;
    unsigned* qn = A.q1 + base;
    if (tid == 0) { q[0] = (unsigned)root; s_next = 0; }
    __syncthreads();
    unsigned count = 1, round = 1;
    unsigned long long visits = 0;
    while (count > 0) {
        visits += count;
        if (count >= 2u * (unsigned)nw) {
            // wide frontier: one lane per frontier vertex, its edges relaxed back to back
            for (unsigned f = tid; f < count; f += blockDim.x) {
                const unsigned u = q[f];
                const float du = st_ord2f(ld(&A.dist_ord[base + u]));
                const uint32_t s = A.row_off[base + u], e = A.row_off[base + u + 1];
                for (uint32_t t = s; t < e; t++) {
                    const unsigned v = A.col[t] - (unsigned)base;
                    const unsigned o = st_f2ord(du + A.wgt[t]);
                    const unsigned old = atomicMin(&A.dist_ord[base + v], o);
                    if (o < old && atomicExch(&A.stamp[base + v], round) != round) qn[atomicAdd(&s_next, 1u)] = v;
                }
            }
        } else {
            // narrow frontier: one wave per frontier vertex, lanes over its edges
            for (unsigned f = wave; f < count; f += nw) {
                const unsigned u = q[f];
                const float du = st_ord2f(ld(&A.dist_ord[base + u]));
                const uint32_t s = A.row_off[base + u], e = A.row_off[base + u + 1];
                for (uint32_t t = s + lane; t < e; t += 64) {
                    const unsigned v = A.col[t] - (unsigned)base;
                    const unsigned o = st_f2ord(du + A.wgt[t]);
                    const unsigned old = atomicMin(&A.dist_ord[base + v], o);
                    if (o < old && atomicExch(&A.stamp[base + v], round) != round) qn[atomicAdd(&s_next, 1u)] = v;
                }
            }
        }
        __syncthreads();
        count = s_next;
        __syncthreads();
        if (tid == 0) s_next = 0;
        unsigned* tq = q; q = qn; qn = tq;
        round++;
        __syncthreads();
    }
    for (int v = tid; v < n; v += blockDim.x) A.dist[base + v] = st_ord2f(ld(&A.dist_ord[base + v]));
    if (A.ticks && tid == 0) { A.ticks[(int64_t)blockIdx.x * 8 + 6] = (long long)round; A.ticks[(int64_t)blockIdx.x * 8 + 7] = (long long)visits; }
    __syncthreads();
}

// canonical predecessors (see oracle so_sssp): smallest tight in-neighbour with smaller distance;
// plateau members are resolved in synchronous rounds from already resolved plateau neighbours
__device__ void sk_preds(const SkArgs& A, int base, int n, int root) {
    __shared__ unsigned s_unres, s_prog;
    const int tid = threadIdx.x;
    if (tid == 0) s_unres = 0;
    __syncthreads();
    for (int v = tid; v < n; v += blockDim.x) {
        const float dv = A.dist[base + v];
        int best = 0x7fffffff;
        if (v != root) {
            for (uint32_t t = A.row_off[base + v]; t < A.row_off[base + v + 1]; t++) {
                const int u = (int)(A.col[t] - (unsigned)base);
                if (u == v) continue;
                const float du = A.dist[base + u];
                if (du < dv && du + A.wgt[t] == dv && u < best) best = u;
            }
        }
        const bool ok = v == root || best != 0x7fffffff;
        A.pred[base + v] = v == root ? -1 : (ok ? best : -1);
        st(&A.stamp[base + v], ok ? 1u : 0u);
        if (!ok) atomicAdd(&s_unres, 1u);
    }
    __syncthreads();
    unsigned unres = s_unres, round = 2;
    while (unres > 0) {
        __syncthreads();
        if (tid == 0) s_prog = 0;
        __syncthreads();
        for (int v = tid; v < n; v += blockDim.x) {
            if (ld(&A.stamp[base + v]) != 0u) continue;
            const float dv = A.dist[base + v];
            int best = 0x7fffffff;
            for (uint32_t t = A.row_off[base + v]; t < A.row_off[base + v + 1]; t++) {
                const int u = (int)(A.col[t] - (unsigned)base);
                if (u == v) continue;
                const unsigned su = ld(&A.stamp[base + u]);
                if (su == 0u || su >= round) continue;  // only vertices resolved in EARLIER rounds
                const float du = A.dist[base + u];
                if (du == dv && du + A.wgt[t] == dv && u < best) best = u;
            }
            if (best != 0x7fffffff) { A.pred[base + v] = best; st(&A.stamp[base + v], round); atomicAdd(&s_prog, 1u); }
        }
        __syncthreads();
        const unsigned prog = s_prog;
        if (prog == 0) break;  // unreachable leftovers (cannot happen inside one component)
        unres -= prog;
        round++;
    }
    __syncthreads();
}

// second SSSP of the reference on the predecessor tree (recomputed Euclidean edge lengths):
// breadth-first down the tree, td[child] = td[parent] + |p_child - p_parent|
__device__ void sk_tree_distance(const SkArgs& A, int base, int n, int root, float* td) {
    __shared__ unsigned s_next;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    unsigned* q = A.q0 + base;
    unsigned* qn = A.q1 + base;
    for (int v = tid; v < n; v += blockDim.x) td[base + v] = __uint_as_float(0x7f800000u);
    __syncthreads();
    if (tid == 0) { q[0] = (unsigned)root; s_next = 0; td[base + root] = 0.0f; }
    __syncthreads();
    unsigned count = 1;
    while (count > 0) {
        for (unsigned f = wave; f < count; f += nw) {
            const unsigned u = q[f];
            const float du = ld(&td[base + u]);
            for (uint32_t t = A.row_off[base + u] + lane; t < A.row_off[base + u + 1]; t += 64) {
                const unsigned v = A.col[t] - (unsigned)base;
                if (A.pred[base + v] != (int)u) continue;
                // duplicate (u,v) edges may exist: only the first writer enqueues
                if (atomicExch(&A.stamp[base + v], 0xfffffffeu) == 0xfffffffeu) continue;
                st(&td[base + v], du + sqrtf(sk_dist(A.pts + 3 * (int64_t)(base + v), A.pts + 3 * (int64_t)(base + u))));
                qn[atomicAdd(&s_next, 1u)] = v;
            }
        }
        __syncthreads();
        count = s_next;
        __syncthreads();
        if (tid == 0) s_next = 0;
        unsigned* tq = q; q = qn; qn = tq;
        __syncthreads();
    }
}

// anc[k][v] = 2^k-th ancestor of v in the predecessor tree (-1 past the root); returns levels built
__device__ int sk_build_lifting(const SkArgs& A, int base, int n) {
    __shared__ unsigned s_any;
    const int tid = threadIdx.x;
    for (int v = tid; v < n; v += blockDim.x) A.anc[base + v] = A.pred[base + v];
    int levels = 1;
    for (int k = 1; k < SK_LIFT; k++) {
        __syncthreads();
        if (tid == 0) s_any = 0;
        __syncthreads();
        const int* prev = A.anc + (int64_t)(k - 1) * A.m + base;
        int* cur = A.anc + (int64_t)k * A.m + base;
        unsigned any = 0;
        for (int v = tid; v < n; v += blockDim.x) {
            const int h = prev[v];
            const int a = h >= 0 ? prev[h] : -1;
            cur[v] = a;
            any |= a >= 0;
        }
        if (any) s_any = 1;
        __syncthreads();
        levels = k + 1;
        if (!s_any) break;
    }
    __syncthreads();
    return levels;
}

// j-th ancestor of v (j = 0: v itself); -1 once the chain passes the root
__device__ __forceinline__ int sk_ancestor(const SkArgs& A, int base, int v, unsigned j, int levels) {
    for (int k = 0; j != 0 && v >= 0; k++, j >>= 1) {
        if (k >= levels) return -1;
        if (j & 1u) v = A.anc[(int64_t)k * A.m + base + v];
    }
    return v;
}

// ----------------------------------------------------------------------------- sample_tree ---
// `distances` = per-vertex tree distance (path.py:53); results in branch_* / path_verts / branch_of.
__device__ void sk_sample_tree(const SkArgs& A, int base, int n, const float* distances, int comp) {
    __shared__ unsigned long long s_red[SK_MAX_WAVES];
    __shared__ int s_len, s_term, s_parent, s_nb, s_total;
    __shared__ unsigned s_ntouched;
    const int tid = threadIdx.x;
    const StGrid* g = A.grid;
    for (int v = tid; v < n; v += blockDim.x) {
        st(&A.alloc[base + v], A.pred[base + v] > 0 ? distances[base + v] : -1.0f);  // path.py:71-72
        st(&A.term[base + v], 0u);
        st(&A.branch_of[base + v], -1);
        st(&A.best[base + v], (unsigned long long)SK_EMPTY64);
    }
    if (tid == 0) { s_nb = 0; s_total = 0; }
    __syncthreads();
    unsigned* tmp = A.q0 + base;
    int* path_out = A.path_verts + base;
    const int levels = sk_build_lifting(A, base, n);
    for (;;) {
        // 1. farthest unallocated vertex, first maximum (path.py:92).  `alloc` is only written with
        //    write-through stores (st) before a barrier; one L1 invalidate makes plain, pipelined
        //    loads of each lane's contiguous slice safe.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned long long key = 0;
        {
            const int per = (n + (int)blockDim.x - 1) / (int)blockDim.x;
            const int v0 = tid * per, v1 = st_min(v0 + per, n);
            const float* al = A.alloc + base;
            for (int v = v0; v < v1; v++) {
                const unsigned long long k = ((unsigned long long)st_f2ord(al[v]) << 32) | (0xffffffffu - (unsigned)v);
                key = k > key ? k : key;
            }
        }
        key = block_max_u64(key, s_red);
        const int far = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
        const float dfar = st_ord2f((unsigned)(key >> 32));
        if (!(dfar > 0.0f)) break;  // path.py:94-95 (uniform: every thread holds the same key)
        // 2. trace_route (path.py:9-16): walk the predecessors until an allocated vertex / past the root.
        //    Lane j looks at the j-th ancestor (binary lifting) so a whole chunk of the chain is
        //    inspected per round instead of one dependent load per hop.
        {
            int found = -1;
            for (unsigned chunk = 0; found < 0; chunk += blockDim.x) {
                const unsigned j = chunk + tid;
                const int node = sk_ancestor(A, base, far, j, levels);
                const bool end = node < 0 || ld(&A.term[base + node]) != 0u;
                if (!end) tmp[j] = (unsigned)node;
                // smallest j that ends the walk: max over (~j) of the lanes that see an end
                unsigned long long k = end ? ((unsigned long long)(0xffffffffu - j) << 32) | (unsigned)(node + 1) : 0ull;
                k = block_max_u64(k, s_red);
                if (k != 0ull) {
                    found = (int)(0xffffffffu - (unsigned)(k >> 32));
                    if (tid == 0) { s_len = found; s_term = (int)(unsigned)(k & 0xffffffffu) - 1; s_ntouched = 0; }
                }
            }
        }
        __syncthreads();
        const int len = s_len, total = s_total, nb = s_nb;
        const bool keep = len >= 2;  // path.py:125-126
        // 3. path stored root side first; r = max radius on the path (path.py:31)
        unsigned long long rk = 0;
        for (int qi = tid; qi < len; qi += blockDim.x) {
            const int v = (int)tmp[len - 1 - qi];
            path_out[total + qi] = v;
            const unsigned long long k = (unsigned long long)st_f2ord(A.rad[base + v]) << 32;
            rk = k > rk ? k : rk;
        }
        rk = block_max_u64(rk, s_red);  // (barriers inside also publish path_out)
        const float rp = st_ord2f((unsigned)(rk >> 32));
        const float rp2 = rp * rp;
        // 4. claim race: every path vertex offers (d2, position) to the points within r of it
        int reach = rp > 0.0f ? (int)ceilf(rp / g->cell) : 0;
        if (reach < 1) reach = 1;
        {
            const int side = 2 * reach + 1, nrow = side * side;
            const int64_t items = (int64_t)len * nrow;
            for (int64_t it = tid; it < items; it += blockDim.x) {  // one lane per (path vertex, x/y grid row)
                const int qi = (int)(it / nrow), rr = (int)(it % nrow);
                const float* pv = A.pts + 3 * (int64_t)(base + path_out[total + qi]);
                const int x = (int)floorf((pv[0] - g->lo[0]) / g->cell) - reach + rr / side;
                const int y = (int)floorf((pv[1] - g->lo[1]) / g->cell) - reach + rr % side;
                if (x < 0 || x >= g->dim[0] || y < 0 || y >= g->dim[1]) continue;
                const int cz = (int)floorf((pv[2] - g->lo[2]) / g->cell);
                const int z0 = st_max(cz - reach, 0), z1 = st_min(cz + reach, g->dim[2] - 1);
                if (z0 > z1) continue;
                const int64_t row = ((int64_t)x * g->dim[1] + y) * g->dim[2];
                const uint32_t s = A.cell_start[row + z0], e = A.cell_start[row + z1 + 1];
                for (uint32_t t = s; t < e; t++) {
                    const float4 r4 = A.recs[t];
                    const int p = (int)__float_as_uint(r4.w) - base;
                    if (p < 0 || p >= n) continue;  // other component
                    const float pp[3] = {r4.x, r4.y, r4.z};
                    const float d2 = sk_dist(pp, pv);
                    if (!(d2 < rp2)) continue;
                    const unsigned long long pk = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)qi;
                    const unsigned long long old = atomicMin(&A.best[base + p], pk);
                    if (old == SK_EMPTY64) A.touched[base + atomicAdd(&s_ntouched, 1u)] = (unsigned)p;
                }
            }
        }
        __syncthreads();
        // 5. parent id is read BEFORE this branch stamps anything (path.py:128-136);
        //    termination -1 reads branch_ids[-1] = the last vertex (quirk kept)
        if (tid == 0) s_parent = ld(&A.branch_of[base + (s_term < 0 ? n - 1 : s_term)]);
        __syncthreads();
        // 6. on-path test (path.py:35-40) and bookkeeping (:112-122,135-136)
        const unsigned nt = s_ntouched;
        for (unsigned t = tid; t < nt; t += blockDim.x) {
            const int p = (int)A.touched[base + t];
            const unsigned long long pk = ld(&A.best[base + p]);
            st(&A.best[base + p], (unsigned long long)SK_EMPTY64);
            const float d2 = __uint_as_float((unsigned)(pk >> 32));
            const int qi = (int)(pk & 0xffffffffu);
            if (sqrtf(d2) < A.rad[base + path_out[total + qi]]) {
                st(&A.alloc[base + p], -1.0f);
                st(&A.term[base + p], 1u);
                if (keep) st(&A.branch_of[base + p], nb);
            }
        }
        for (int qi = tid; qi < len; qi += blockDim.x) {
            const int v = path_out[total + qi];
            st(&A.alloc[base + v], -1.0f);
            st(&A.term[base + v], 1u);
            if (keep) st(&A.branch_of[base + v], nb);
        }
        __syncthreads();
        if (tid == 0 && keep) {
            A.branch_parent[base + nb] = s_parent;
            A.branch_off[base + nb] = total;
            A.branch_len[base + nb] = len;
            s_nb = nb + 1;
            s_total = total + len;
        }
        __syncthreads();
    }
    if (tid == 0) A.n_branches[comp] = s_nb;
    __syncthreads();
}

// stages: bit0 root+sssp+preds, bit1 literal tree distance (into `alloc`-independent td buffer), bit2 sample_tree
__global__ void __launch_bounds__(1024) k_skeleton_components(SkArgs A, int stages, float* tree_dist) {
    __shared__ unsigned long long s_red[SK_MAX_WAVES];
    const int c = blockIdx.x;
    const int base = A.comp_off[c], n = A.comp_off[c + 1] - base;
    if (n <= 0) { if (threadIdx.x == 0 && (stages & 4)) A.n_branches[c] = 0; return; }
#define SK_TICK(i) if (A.ticks && threadIdx.x == 0) A.ticks[(int64_t)c * 8 + (i)] = wall_clock64()
    SK_TICK(0);
    int root;
    if (stages & 1) {
        // root = first minimum of the surface y (cloud.py:204-206)
        unsigned long long key = 0;
        for (int v = threadIdx.x; v < n; v += blockDim.x) {
            const unsigned long long k = ((unsigned long long)(0xffffffffu - st_f2ord(A.ysurf[base + v])) << 32) | (0xffffffffu - (unsigned)v);
            key = k > key ? k : key;
        }
        key = block_max_u64(key, s_red);
        root = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
        if (threadIdx.x == 0) A.root_local[c] = root;
        SK_TICK(1);
        sk_sssp(A, base, n, root);
        SK_TICK(2);
        sk_preds(A, base, n, root);
        SK_TICK(3);
    } else {
        root = A.root_local[c];
    }
    if (stages & 2) sk_tree_distance(A, base, n, root, tree_dist);
    SK_TICK(4);
    if (stages & 4) sk_sample_tree(A, base, n, (stages & 2) ? tree_dist : A.dist, c);
    SK_TICK(5);
}

// ------------------------------------------------------------------------------- host side ---
struct SkScratch {
    unsigned *dist_ord, *stamp, *q0, *q1, *term, *touched;
    float* alloc;
    unsigned long long* best;
    int* anc;
};
static void sk_scratch(StArena& a, int64_t m, SkScratch* s) {
    s->dist_ord = a.take<unsigned>(m);
    s->stamp = a.take<unsigned>(m);
    s->q0 = a.take<unsigned>(m);
    s->q1 = a.take<unsigned>(m);
    s->term = a.take<unsigned>(m);
    s->touched = a.take<unsigned>(m);
    s->alloc = a.take<float>(m);
    s->best = a.take<unsigned long long>(m);
    s->anc = a.take<int>((int64_t)SK_LIFT * m);
}

#define SK_GRID_CELLS (1ll << 24)

extern "C" int64_t st_skeleton_workspace_bytes(int64_t m) {
    StArena a(nullptr, 0);
    SkScratch s;
    sk_scratch(a, m, &s);
    a.take<StGrid>(1);
    a.take<uint32_t>(SK_GRID_CELLS + 1);
    a.take<float4>(m);
    a.take<char>(st_grid_ws_bytes(m, SK_GRID_CELLS));
    a.take<int>(4);
    return a.used;
}

// All components of one cloud: root search, SSSP, predecessors, (optional literal tree distance),
// sample_tree.  Vertex arrays are in the renumbered space of st_component_layout.
// stages: 1 = sssp+preds, 2 = literal second SSSP (tree_dist must be given), 4 = sample_tree.
extern "C" int st_skeleton_components(int n_comp, const int32_t* comp_off, int64_t m, const float* pts, const float* rad,
                                      const float* ysurf, const uint32_t* row_off, const uint32_t* col, const float* wgt,
                                      float grid_cell, int stages, int block_threads, float* dist, int32_t* pred,
                                      int32_t* root_local, float* tree_dist, int32_t* branch_parent, int32_t* branch_off,
                                      int32_t* branch_len, int32_t* n_branches, int32_t* path_verts, int32_t* branch_of,
                                      long long* phase_ticks /* optional [n_comp][8] device buffer, NULL = off */, void* ws,
                                      int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_comp <= 0 || m <= 0) return ST_OK;
    ST_REQUIRE(!(stages & 2) || tree_dist != nullptr, "skeleton: stage 2 needs a tree_dist buffer");
    if (block_threads <= 0) block_threads = 512;
    ST_REQUIRE(block_threads % 64 == 0 && block_threads <= 1024, "skeleton: block_threads must be a multiple of 64, <= 1024");
    StArena a(ws, ws_bytes);
    SkScratch s;
    sk_scratch(a, m, &s);
    StGrid* g = a.take<StGrid>(1);
    uint32_t* cell_start = a.take<uint32_t>(SK_GRID_CELLS + 1);
    float4* recs = a.take<float4>(m);
    int64_t gb = st_grid_ws_bytes(m, SK_GRID_CELLS);
    char* gws = a.take<char>(gb);
    if (!a.ok() || !gws) {
        st_set_error("skeleton: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    if (stages & 4) ST_TRY(st_grid_build(pts, m, grid_cell, SK_GRID_CELLS, g, cell_start, recs, gws, gb, stream));
    SkArgs A;
    A.C = n_comp; A.comp_off = comp_off; A.pts = pts; A.rad = rad; A.ysurf = ysurf;
    A.row_off = row_off; A.col = col; A.wgt = wgt; A.grid = g; A.cell_start = cell_start; A.recs = recs;
    A.dist = dist; A.pred = pred; A.root_local = root_local;
    A.branch_parent = branch_parent; A.branch_off = branch_off; A.branch_len = branch_len; A.n_branches = n_branches;
    A.path_verts = path_verts; A.branch_of = branch_of;
    A.dist_ord = s.dist_ord; A.stamp = s.stamp; A.q0 = s.q0; A.q1 = s.q1; A.alloc = s.alloc; A.term = s.term;
    A.best = s.best; A.touched = s.touched; A.anc = s.anc; A.m = m; A.ticks = phase_ticks;
    hipLaunchKernelGGL(k_skeleton_components, dim3((unsigned)n_comp), dim3((unsigned)block_threads), 0, stream, A, stages,
                       tree_dist);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
