// Skeleton extraction for ALL components of a cloud at once: SSSP, predecessor tree, (literal
// second SSSP), greedy sample_tree.
//
// Reference (smart_tree/skeleton): process_subgraph skeletonize.py:57-95, shortest_paths
// shortest_path.py:12-21 (cugraph.sssp), pred_graph :46-55 + second sssp skeletonize.py:80-85,
// sample_tree / trace_route / select_path_points path.py:9-140.  There every component costs
// several cugraph builds, a cudf->pandas->torch round trip, and per branch a host-synchronising
// argmax, a Python pointer chase with an O(|T|) tensor membership test per step and an ALL-points
// FRNN query, one component after the other.
//
// Here every phase is a chip-wide, level-synchronous kernel over the renumbered vertex space of
// st_component_layout (components contiguous), and all components advance in lockstep:
//   * SSSP: frontier rounds (one wavefront per frontier vertex, lanes over its edges, atomicMin on
//     order-preserving float bits, wave-aggregated queue push); every component's root is a source.
//     The distances are the least fixed point of d[v] = min fl32(d[u]+w) -- schedule independent.
//   * predecessors: canonical choice (smallest tight in-neighbour, plateau rounds), one lane/vertex.
//   * sample_tree, one iteration = the next branch of EVERY unfinished component:
//       select   (one workgroup per component) argmax of the remaining distances, path trace by
//                binary lifting over the predecessor tree (lane j inspects the j-th ancestor),
//                path radius, parent lookup, branch record
//       claim    (workgroups proportional to component size) the K=1 "nearest path vertex" query
//                inverted: every (path vertex, grid row) pair scans the uniform grid and races
//                with a packed atomicMin(d2 bits | path position) per point
//       (the on-path test and the allocation / termination / branch-id stamps of iteration i run at
//        the head of select i+1, inside the component's own workgroup: two launches per iteration)
//   Launches are enqueued in batches; the host reads one small counter block per batch.
// Semantics and tie-breaks: oracle/skeleton_oracle.c (so_sssp, so_tree_distance, so_sample_tree).
#include "st_common.h"
#include "st_grid.h"
#include "smarttree_hip.h"  // the public declarations of this file's entry points (checked against the definitions by the compiler)

#include <atomic>

#define SK_MAX_WAVES 16
#define SK_EMPTY64 0xffffffffffffffffull
#define SK_WIDE_BLOCK 256  // block size of the vertex / frontier kernels
#define SK_SSSP_BLOCKS 2048  // upper bound; small graphs launch fewer (one wave per frontier vertex).  A batch of clouds has a frontier of ~10k vertices:
                             // with 256 workgroups a launch took 81 us, with 2048 (and the look before the atomic) 40 us (24 clouds)
#define SK_MARK 0xfffffffeu
#define SK_ANC 64  // direct ancestors kept per vertex; longer walks hop 64 levels at a time (16 measured slower: every lane
                   // of a 1024-wide chunk hops j/SK_ANC times, so the chunk costs as much as its farthest lane)

// A long-path claim handed to a component's helper workgroups: written by the component's own workgroup with agent-scope
// stores (the helpers run on other compute units, possibly other XCDs), then `seq` is bumped.
struct SkJob {
    unsigned seq, done, quit;
    int len, id, csz, nrows, ny, x0, y0, z0, z1, path_off;
    float rp;
    unsigned gone;  // a helper left because its life time ran out: the component's workgroup posts no more jobs
    unsigned alive;  // the component's workgroup is resident (a helper that never sees it gives up after `help_announce`)
};

struct SkArgs {
    int C;
    int64_t m;
    const int* comp_off;  // [C+1] offsets into the renumbered vertex space
    const int* comp_seg;  // [C] cloud of each component in a batched call (nullptr: one cloud) -- selects the slab of grid cells
    int* comp_of;         // [m] component of each vertex
    const float* pts;     // [m,3] medial points
    const float* rad;     // [m] raw radius (cloud.radius)
    const float* ysurf;   // [m] y of the surface point (root = lowest surface point)
    const uint32_t* row_off;  // [m+1]
    const uint32_t* col;      // renumbered neighbour ids
    const float* wgt;
    const StGrid* grid;
    const uint32_t* cell_start;
    const float4* recs;
    // outputs
    float* dist;         // [m]
    int* pred;           // [m] component-local predecessor, -1 at the root
    int* root_local;     // [C]
    float* tree_dist;    // [m] optional
    int* branch_parent;  // [m] (component slice)
    int* branch_off;     // [m] offset into the component's path_verts slice
    int* branch_len;     // [m]
    int* n_branches;     // [C]
    int* path_verts;     // [m] component-local vertex ids, root side first
    int* branch_of;      // [m] final branch id per point (-1: none)
    // scratch [m]
    unsigned* dist_ord;
    unsigned* stamp;  // SSSP: queued-in-round marker; preds: resolution round; tree distance: visited
    unsigned* q0;
    unsigned* q1;
    unsigned* term_bits;  // sample_tree's termination set, one bit per vertex; component c owns the words from (comp_off[c] >> 5) + c
    unsigned* help_bits;  // the same layout: what helper workgroups terminated (agent scope), folded into term_bits / the LDS copy
    float4* pr;       // [m] (x, y, z, radius) of every vertex in one 16-byte record (k_sk_lift_init): one gather instead of two
    unsigned long long* best;  // claim race: (d2 bits << 32) | path position
    unsigned* touched;
    int* anc;         // [m][SK_ANC] direct ancestor table (component-local ids)
    // global counters: [0..2] rotating frontier counts (tree-distance rounds), [3] unresolved, [4] plateau progress, [5] components done
    unsigned* cnt;
    unsigned* fcnt;   // [3][SK_FS + 1] SSSP frontier counters per generation: SK_FS shards + the overflow area (see sk_q_reserve)
    unsigned fseg;    // entries per shard of a frontier queue
    // per-component sample_tree state [C]
    int* s_done;
    int* s_len;
    int* s_cur_id;   // branch id being stamped (-1: path shorter than 2, nothing stamped)
    int* s_cur_off;  // where this iteration's path sits in the component's path_verts slice
    int* s_nb;
    int* s_total;
    float* s_rp;
    unsigned* s_ntouched;
    int* s_cursor;   // position in `order` where the search for the next farthest vertex resumes
    int* s_wide;     // this launch's path is long: the chip-wide claim kernel handles it
    const unsigned* order;  // [m] vertices of each component by distance descending (ties: index ascending)
    const float* order_init;  // [m] initial distance of order[j] (<= 0 marks the tail of never-selectable vertices)
    // claim grid: workgroup b works for component blk_comp[b], as slice (b - blk_first[c]) of blk_count[c]
    const int* blk_comp;
    const int* blk_first;
    const int* blk_count;
    float prune_factor;  // select: an entry within prune_factor x radius of an earlier chosen tip gets no wavefront
    int small_work;      // select: candidate points x path vertices one workgroup takes on itself (beyond: k_sk_claim)
    int iters_per_launch;
    int wave_work;       // select: candidate points x path vertices of one speculative branch (beyond: whole workgroup)
    int local_items;     // select: (path vertex, cell row) pairs one workgroup claims path-centric by itself
    int long_mode;       // select: paths that fit the LDS path buffer but are too much work for the plain point-centric claim are
                         // claimed by the workgroup itself with chunk-pruned distance tests (below), not handed to k_sk_claim
    // helper workgroups of the long-path claims (k_sk_select): the first n_helpers workgroups of the launch serve the large components
    int n_helpers;
    const int* hl_comp;   // [n_helpers] component a helper serves (-1: none)
    const int* hl_rank;   // [n_helpers] its rank among that component's helpers (1 ..)
    const int* c_nhelp;   // [C] helpers of a component
    struct SkJob* jobs;   // [C]
    long long help_lifetime, help_timeout, help_announce;  // time-outs of the helper protocol, ticks of the 100 MHz wall clock
    long long* ticks;  // optional developer aid: per-phase wall_clock64 sums of k_sk_select (tuning[15])
};

// agent-scope (L2) accesses for data that the SAME launch wrote earlier (never a stale L1 line)
template <class T>
__device__ __forceinline__ T ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld(const float* p) { return __uint_as_float(ld((const unsigned*)p)); }
// workgroup-scope load: may be served by this CU's L1 / this XCD's L2
__device__ __forceinline__ unsigned ld_wg(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int ld_wg(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ float ld_wg(const float* p) { return __uint_as_float(ld_wg((const unsigned*)p)); }
// Workgroup-scope read-modify-writes.  The selection state of a component (termination bits, branch ids) is touched by ONE workgroup
// per launch: its atomics need not be coherent across the eight XCDs' L2s.  A device-scope atomic is executed on the memory
// side of the fabric (calibration: 32 bytes of WRITE_SIZE each, ~12 G/s) and every agent-scope load goes there too; at
// workgroup scope both stay in this XCD's L2.  Visibility to the NEXT launch comes with the kernel boundary.
__device__ __forceinline__ void wg_or(unsigned* p, unsigned v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wg_and(unsigned* p, unsigned v) { (void)__hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wg_max(int* p, int v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

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

#define SK_VERTEX_LOOP(v) \
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < A.m; v += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------- set-up ---
// SSSP frontier queues.  Every workgroup of a frontier launch appends what it improved to the next frontier with a
// RETURNING atomicAdd on the queue's counter; on one word those retire at ~90 per microsecond chip-wide
// (MI355X_MICROARCH.md "dequeue"): the ~2000 workgroups of a batched launch spent 20+ us of a 45 us launch queueing on it
// (one cloud: 190 workgroups, 2 us).  The queue is therefore cut into SK_FS shards (workgroup b appends to shard b % SK_FS,
// its own counter, its own segment of `fseg` entries); a reservation that does not fit its segment goes to the overflow
// area behind the segments (one more counter, room for every vertex), the slots it leaves unused in the segment are
// filled with SK_Q_HOLE, which readers skip.  A frontier is read as the concatenation shard 0 .. SK_FS-1, overflow.
#define SK_FS 16
#define SK_Q_HOLE 0xffffffffu
// where `n` entries of workgroup-shard `shard` go in queue `q`; fc = the generation's SK_FS + 1 counters
__device__ __forceinline__ unsigned sk_q_reserve(unsigned* q, unsigned* fc, unsigned seg, unsigned shard, unsigned n) {
    const unsigned at = atomicAdd(&fc[shard], n);
    if (at + n <= seg) return shard * seg + at;
    for (unsigned i = at; i < seg; i++) q[shard * seg + i] = SK_Q_HOLE;  // (at most n - 1 slots, once per shard and launch)
    return SK_FS * seg + atomicAdd(&fc[SK_FS], n);
}

// the same for the persistent form: the holes are written where another workgroup's agent-scope read finds them
__device__ __forceinline__ unsigned sk_q_reserve_coop(unsigned* q, unsigned* fc, unsigned seg, unsigned shard, unsigned n) {
    const unsigned at = atomicAdd(&fc[shard], n);
    if (at + n <= seg) return shard * seg + at;
    for (unsigned i = at; i < seg; i++) __hip_atomic_store(&q[shard * seg + i], SK_Q_HOLE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return SK_FS * seg + atomicAdd(&fc[SK_FS], n);
}

__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_init(SkArgs A) {
    const unsigned inf = st_f2ord(__uint_as_float(0x7f800000u));
    SK_VERTEX_LOOP(v) { A.dist_ord[v] = inf; A.stamp[v] = 0u; }
    if (blockIdx.x == 0 && threadIdx.x < 8) A.cnt[threadIdx.x] = 0u;
    if (blockIdx.x == 0)  // round 0's frontier = every component's root, in the overflow area of q0 (k_sk_roots)
        for (int i = threadIdx.x; i < 3 * (SK_FS + 1); i += blockDim.x) A.fcnt[i] = i == SK_FS ? (unsigned)A.C : 0u;
}

// one workgroup per component: comp_of[], root = first minimum of the surface y (cloud.py:204-206)
__global__ void __launch_bounds__(1024) k_sk_roots(SkArgs A) {
    __shared__ unsigned long long s_red[SK_MAX_WAVES];
    const int c = blockIdx.x, base = A.comp_off[c], n = A.comp_off[c + 1] - base;
    unsigned long long key = 0;
    for (int v = threadIdx.x; v < n; v += blockDim.x) {
        A.comp_of[base + v] = c;
        const unsigned long long k = ((unsigned long long)(0xffffffffu - st_f2ord(A.ysurf[base + v])) << 32) | (0xffffffffu - (unsigned)v);
        key = k > key ? k : key;
    }
    key = block_max_u64(key, s_red);
    if (threadIdx.x == 0) {
        const int root = n > 0 ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffu)) : 0;
        A.root_local[c] = root;
        A.q0[(int64_t)SK_FS * A.fseg + c] = (unsigned)(base + root);  // round 0 frontier = every component's root
        if (n > 0) A.dist_ord[base + root] = st_f2ord(0.0f);
    }
}

// comp_of[] alone (callers that bring their own predecessors skip the root / SSSP stage)
__global__ void __launch_bounds__(1024) k_sk_fill_comp_of(SkArgs A) {
    const int c = blockIdx.x, base = A.comp_off[c], n = A.comp_off[c + 1] - base;
    for (int v = threadIdx.x; v < n; v += blockDim.x) A.comp_of[base + v] = c;
}

// ------------------------------------------------------------------------------------ SSSP ---
// Launch r: global frontier r%2 -> (r+1)%2; counts rotate through cnt[0..2].  The frontier of a tree-shaped graph
// is small (a few hundred vertices) and hundreds of levels deep, so a launch per level is bound by launch latency.
// Each workgroup therefore keeps relaxing what IT improved for up to `hops` further levels from a queue in LDS
// before handing the rest to the next launch.  The distances are the least fixed point of d[v] = min(d[u] + w)
// whatever the order of relaxations (atomicMin; every improvement is queued again), so the result is the same
// as one level per launch -- at 1/hops of the launches.
#define SK_LQ 1024  // entries per local generation; overflow goes straight to the global queue
__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_sssp_round(SkArgs A, int r, int hops, int glanes, int lcap) {
    __shared__ unsigned lq[2][SK_LQ];
    __shared__ unsigned ln[3], lq_base;  // generation h fills lq[h & 1], counted by ln[h % 3]
    __shared__ unsigned fpre[SK_FS + 2];  // this launch's frontier: entries in front of each shard / of the overflow area
    const unsigned* fc_in = A.fcnt + (r % 3) * (SK_FS + 1);
    unsigned* fc_out = A.fcnt + ((r + 1) % 3) * (SK_FS + 1);
    if (blockIdx.x == 0 && threadIdx.x <= SK_FS) A.fcnt[((r + 2) % 3) * (SK_FS + 1) + threadIdx.x] = 0u;
    const unsigned seg = A.fseg, shard = blockIdx.x % SK_FS;
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int s_ = 0; s_ <= SK_FS; s_++) {
            fpre[s_] = run;
            const unsigned c_ = fc_in[s_];
            run += s_ < SK_FS && c_ > seg ? seg : c_;  // a shard holds at most `seg` entries (the rest went to the overflow area)
        }
        fpre[SK_FS + 1] = run;
    }
    if (threadIdx.x < 3) ln[threadIdx.x] = 0;
    __syncthreads();
    const unsigned count = fpre[SK_FS + 1];
    if (count == 0) return;  // (uniform)
    const unsigned* q = (r & 1) ? A.q1 : A.q0;
    unsigned* qn = (r & 1) ? A.q0 : A.q1;
    const bool lookfirst = lcap >= 0;
    if (lcap < 0) lcap = -lcap;
    const unsigned round = (unsigned)r + 1u;  // stamp[v] == round: v already sits in the next global frontier
    const int lane = threadIdx.x & (glanes - 1);  // `glanes` lanes share one vertex (rows hold ~16 edges)
    const unsigned wave = threadIdx.x / glanes, nwv = blockDim.x / glanes;
    const unsigned gw = blockIdx.x * nwv + wave, tw = gridDim.x * nwv;
    // one lane group per vertex: relax its edges; improved neighbours go to the local generation `out`
#define SK_RELAX(u, du, out, out_n)                                                                        \
    {                                                                                                      \
        const uint32_t s_ = A.row_off[u], e_ = A.row_off[(u) + 1];                                         \
        for (uint32_t t = s_ + lane; t < e_; t += glanes) {                                                    \
            const unsigned v = A.col[t];                                                                   \
            const unsigned o = st_f2ord((du) + A.wgt[t]);                                                  \
            if (lookfirst && o >= A.dist_ord[v]) continue; /* plain load: a stale line holds an OLDER, larger value */ \
            const unsigned old = atomicMin(&A.dist_ord[v], o);                                             \
            if (o < old) {                                                                                 \
                const unsigned slot = atomicAdd(out_n, 1u);                                                \
                if (slot < SK_LQ) (out)[slot] = v;                                                         \
                else if (atomicExch(&A.stamp[v], round) != round) qn[sk_q_reserve(qn, fc_out, seg, shard, 1u)] = v; \
            }                                                                                              \
        }                                                                                                  \
    }
    for (unsigned f = gw; f < count; f += tw) {
        int sh = 0;  // the f-th entry of the frontier sits in shard `sh` (SK_FS: the overflow area)
#pragma unroll
        for (int s_ = 1; s_ <= SK_FS; s_++) sh = f >= fpre[s_] ? s_ : sh;
        const unsigned u = q[(int64_t)sh * seg + (f - fpre[sh])];
        if (u == SK_Q_HOLE) continue;  // (wave-uniform: one vertex per lane group)
        const float du = st_ord2f(A.dist_ord[u]);  // written before this launch
        SK_RELAX(u, du, lq[0], &ln[0]);
    }
    int h = 1;  // generation being read: lq[(h - 1) & 1], ln[(h - 1) % 3]
    for (; h < hops; h++) {
        __syncthreads();  // generation h-1 is complete; nobody still reads generation h-2
        const unsigned have = ln[(h - 1) % 3];
        unsigned ncur = have < SK_LQ ? have : SK_LQ;
        if (ncur == 0) break;  // uniform
        if (threadIdx.x == 0) ln[(h + 1) % 3] = 0;  // counter of the generation after this one (last read two barriers ago)
        const unsigned* in = lq[(h - 1) & 1];
        unsigned* out = lq[h & 1];
        if (ncur > (unsigned)lcap) {
            // a launch lasts as long as its busiest workgroup: what exceeds `lcap` vertices per level goes back to the global
            // frontier, where the next launch deals it out over all workgroups
            for (unsigned i = lcap + threadIdx.x; i < ncur; i += blockDim.x) {
                const unsigned v = in[i];
                if (atomicExch(&A.stamp[v], round) != round) qn[sk_q_reserve(qn, fc_out, seg, shard, 1u)] = v;
            }
            ncur = (unsigned)lcap;
        }
        for (unsigned i = wave; i < ncur; i += nwv) {
            const unsigned u = in[i];
            const float du = st_ord2f(ld(&A.dist_ord[u]));  // improved during this launch: read it where the atomics act
            SK_RELAX(u, du, out, &ln[h % 3]);
        }
    }
#undef SK_RELAX
    __syncthreads();
    // what is left joins the next global frontier (once per vertex: stamp)
    const unsigned have = ln[(h - 1) % 3];
    const unsigned nrem = have < SK_LQ ? have : SK_LQ;
    const unsigned* rem = lq[(h - 1) & 1];
    unsigned* keep = lq[h & 1];
    unsigned* keep_n = &ln[h % 3];
    if (threadIdx.x == 0) *keep_n = 0;
    __syncthreads();
    for (unsigned i = threadIdx.x; i < nrem; i += blockDim.x) {
        const unsigned v = rem[i];
        if (atomicExch(&A.stamp[v], round) != round) keep[atomicAdd(keep_n, 1u)] = v;
    }
    __syncthreads();
    const unsigned nloc = *keep_n;
    if (threadIdx.x == 0 && nloc) lq_base = sk_q_reserve(qn, fc_out, seg, shard, nloc);
    __syncthreads();
    for (unsigned i = threadIdx.x; i < nloc; i += blockDim.x) qn[lq_base + i] = keep[i];
}


// ---- the same rounds in ONE launch: a persistent grid with a barrier between the rounds -----------------------------------
// A launch per round costs its launch latency ~100 times over (a 1M-point tree: 96 rounds of 22-39 us, most of it the
// launch) and a host read-back per batch of rounds.  Here the workgroups stay: round r ends at a grid barrier (one arrival
// per workgroup on its group's counter, the last of a group on the top counter, everybody polls the top counter), the
// frontier counters tell every workgroup whether another round is due.  Everything that crosses workgroups inside the
// launch -- distances, queue entries, counters -- is read and written at agent scope (the L2s of the eight XCDs are not
// coherent with each other for plain accesses); the adjacency is read-only.  The grid must be resident as a whole: the host
// launches at most SK_COOP_BLOCKS workgroups of 256 lanes (four per compute unit) and a workgroup that waits longer than
// ~2 s at a barrier gives up and flags the launch (the host then falls back to a launch per round).
#define SK_COOP_BLOCKS 1024
#define SK_COOP_GROUP 32
#define SK_COOP_STRIDE 16  // words between the barrier counters (one 64-byte line each)
__device__ __forceinline__ unsigned ld_u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// bar[0] top counter, bar[SK_COOP_STRIDE * (1 + g)] counter of group g, bar[1] "gave up" flag.  Returns false on a time-out.
__device__ __forceinline__ bool sk_grid_barrier(unsigned* bar, unsigned round) {
    __shared__ int ok_;
    __syncthreads();
    if (threadIdx.x == 0) {
        // No __threadfence() on either side (it writes back / invalidates the L2: measured 35 us per barrier at 1024 workgroups
        // against 4.3 without, tools/microbench/grid_barrier.hip): everything that crosses workgroups is an agent-scope atomic
        // access, coherent by itself; the __syncthreads() above has waited for this workgroup's accesses to complete.
        const unsigned G = gridDim.x, ngroups = (G + SK_COOP_GROUP - 1) / SK_COOP_GROUP, gidx = blockIdx.x / SK_COOP_GROUP;
        const unsigned gsize = gidx + 1 == ngroups ? G - gidx * SK_COOP_GROUP : SK_COOP_GROUP;
        const unsigned old = atomicAdd(&bar[SK_COOP_STRIDE * (1 + gidx)], 1u);
        if (old % gsize == gsize - 1) atomicAdd(&bar[0], 1u);  // the last of its group in this round
        const unsigned want = ngroups * (round + 1u);
        int ok = 1;
        const long long t0 = wall_clock64();
        while (ld_u(&bar[0]) < want) {
            __builtin_amdgcn_s_sleep(2);
            if (ld_u(&bar[1]) != 0u || wall_clock64() - t0 > 200000000ll) { st_u(&bar[1], 1u); ok = 0; break; }  // (100 MHz clock)
        }
        ok_ = ok;
    }
    __syncthreads();
    return ok_ != 0;
}

__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_sssp_coop(SkArgs A, int hops, int glanes, int lcap, unsigned* bar, int max_rounds) {
    __shared__ unsigned lq[2][SK_LQ];
    __shared__ unsigned ln[3], lq_base;
    __shared__ unsigned fpre[SK_FS + 2];
    const unsigned seg = A.fseg, shard = blockIdx.x % SK_FS;
    const bool lookfirst = lcap >= 0;
    if (lcap < 0) lcap = -lcap;
    const int lane = threadIdx.x & (glanes - 1);
    const unsigned wave = threadIdx.x / glanes, nwv = blockDim.x / glanes;
    const unsigned gw = blockIdx.x * nwv + wave, tw = gridDim.x * nwv;
    for (int r = 0; r < max_rounds; r++) {
        const unsigned* fc_in = A.fcnt + (r % 3) * (SK_FS + 1);
        unsigned* fc_out = A.fcnt + ((r + 1) % 3) * (SK_FS + 1);
        if (blockIdx.x == 0 && threadIdx.x <= SK_FS) st_u(&A.fcnt[((r + 2) % 3) * (SK_FS + 1) + threadIdx.x], 0u);
        if (threadIdx.x == 0) {
            unsigned run = 0;
            for (int s_ = 0; s_ <= SK_FS; s_++) {
                fpre[s_] = run;
                const unsigned c_ = ld_u(&fc_in[s_]);
                run += s_ < SK_FS && c_ > seg ? seg : c_;
            }
            fpre[SK_FS + 1] = run;
        }
        if (threadIdx.x < 3) ln[threadIdx.x] = 0;
        __syncthreads();
        const unsigned count = fpre[SK_FS + 1];
        if (count == 0) {  // (the same for every workgroup: the counters were final at the barrier)
            if (blockIdx.x == 0 && threadIdx.x == 0) st_u(&bar[2], (unsigned)r);  // rounds used (statistics)
            return;
        }
        const unsigned* q = (r & 1) ? A.q1 : A.q0;
        unsigned* qn = (r & 1) ? A.q0 : A.q1;
        const unsigned round = (unsigned)r + 1u;
#define SK_QPUSH(v_) st_u(&qn[sk_q_reserve_coop(qn, fc_out, seg, shard, 1u)], (v_))
#define SK_RELAX_C(u, du, out, out_n)                                                                      \
    {                                                                                                      \
        const uint32_t s_ = A.row_off[u], e_ = A.row_off[(u) + 1];                                         \
        for (uint32_t t = s_ + lane; t < e_; t += glanes) {                                                \
            const unsigned v = A.col[t];                                                                   \
            const unsigned o = st_f2ord((du) + A.wgt[t]);                                                  \
            if (lookfirst && o >= ld_u(&A.dist_ord[v])) continue;                                          \
            const unsigned old = atomicMin(&A.dist_ord[v], o);                                             \
            if (o < old) {                                                                                 \
                const unsigned slot = atomicAdd(out_n, 1u);                                                \
                if (slot < SK_LQ) (out)[slot] = v;                                                         \
                else if (atomicExch(&A.stamp[v], round) != round) SK_QPUSH(v);                             \
            }                                                                                              \
        }                                                                                                  \
    }
        for (unsigned f = gw; f < count; f += tw) {
            int sh = 0;
#pragma unroll
            for (int s_ = 1; s_ <= SK_FS; s_++) sh = f >= fpre[s_] ? s_ : sh;
            const unsigned u = ld_u(&q[(int64_t)sh * seg + (f - fpre[sh])]);
            if (u == SK_Q_HOLE) continue;
            const float du = st_ord2f(ld_u(&A.dist_ord[u]));
            SK_RELAX_C(u, du, lq[0], &ln[0]);
        }
        int h = 1;
        for (; h < hops; h++) {
            __syncthreads();
            const unsigned have = ln[(h - 1) % 3];
            unsigned ncur = have < SK_LQ ? have : SK_LQ;
            if (ncur == 0) break;
            if (threadIdx.x == 0) ln[(h + 1) % 3] = 0;
            const unsigned* in = lq[(h - 1) & 1];
            unsigned* out = lq[h & 1];
            if (ncur > (unsigned)lcap) {
                for (unsigned i = lcap + threadIdx.x; i < ncur; i += blockDim.x) {
                    const unsigned v = in[i];
                    if (atomicExch(&A.stamp[v], round) != round) SK_QPUSH(v);
                }
                ncur = (unsigned)lcap;
            }
            for (unsigned i = wave; i < ncur; i += nwv) {
                const unsigned u = in[i];
                const float du = st_ord2f(ld_u(&A.dist_ord[u]));
                SK_RELAX_C(u, du, out, &ln[h % 3]);
            }
        }
#undef SK_RELAX_C
        __syncthreads();
        const unsigned have = ln[(h - 1) % 3];
        const unsigned nrem = have < SK_LQ ? have : SK_LQ;
        const unsigned* rem = lq[(h - 1) & 1];
        unsigned* keep = lq[h & 1];
        unsigned* keep_n = &ln[h % 3];
        if (threadIdx.x == 0) *keep_n = 0;
        __syncthreads();
        for (unsigned i = threadIdx.x; i < nrem; i += blockDim.x) {
            const unsigned v = rem[i];
            if (atomicExch(&A.stamp[v], round) != round) keep[atomicAdd(keep_n, 1u)] = v;
        }
        __syncthreads();
        const unsigned nloc = *keep_n;
        if (threadIdx.x == 0 && nloc) lq_base = sk_q_reserve_coop(qn, fc_out, seg, shard, nloc);
        __syncthreads();
        for (unsigned i = threadIdx.x; i < nloc; i += blockDim.x) st_u(&qn[lq_base + i], keep[i]);
#undef SK_QPUSH
        if (!sk_grid_barrier(bar, (unsigned)r)) return;
    }
}

__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_dist_out(SkArgs A) {
    SK_VERTEX_LOOP(v) A.dist[v] = st_ord2f(A.dist_ord[v]);
}

// canonical predecessors (oracle so_sssp): smallest tight in-neighbour with smaller distance.
// Eight lanes per vertex: a row (~32 entries) is read as coalesced 32-byte pieces and its distance gathers are in flight
// together (one lane per vertex walked its row alone: 64 rows per wavefront, one dependent gather after the other).
#define SK_PRED_LANES 8
__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_preds(SkArgs A) {
    const int sub = threadIdx.x & (SK_PRED_LANES - 1);
    const int64_t groups = ((int64_t)gridDim.x * blockDim.x) / SK_PRED_LANES;
    // (every lane of a group runs the same number of iterations: v depends on the group only)
    for (int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SK_PRED_LANES; v < A.m; v += groups) {
        const int c = A.comp_of[v], base = A.comp_off[c];
        const bool is_root = (int)(v - base) == A.root_local[c];
        const float dv = A.dist[v];
        unsigned best = 0xffffffffu;
        if (!is_root)
            for (uint32_t t = A.row_off[v] + sub, e = A.row_off[v + 1]; t < e; t += SK_PRED_LANES) {
                const unsigned u = A.col[t];
                if (u == (unsigned)v) continue;
                const float du = A.dist[u];
                if (du < dv && du + A.wgt[t] == dv && u < best) best = u;
            }
        for (int d = SK_PRED_LANES / 2; d > 0; d >>= 1) { const unsigned o = __shfl_xor(best, d); best = o < best ? o : best; }
        if (sub != 0) continue;
        const bool ok = is_root || best != 0xffffffffu;
        A.pred[v] = (is_root || !ok) ? -1 : (int)best - base;
        A.stamp[v] = ok ? 1u : 0u;
        if (!ok) atomicAdd(&A.cnt[3], 1u);
    }
}

// plateau members: resolved in synchronous rounds from plateau neighbours resolved in EARLIER rounds
__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_preds_plateau(SkArgs A, unsigned round) {
    SK_VERTEX_LOOP(v) {
        if (A.stamp[v] != 0u) continue;
        const int base = A.comp_off[A.comp_of[v]];
        const float dv = A.dist[v];
        unsigned best = 0xffffffffu;
        for (uint32_t t = A.row_off[v]; t < A.row_off[v + 1]; t++) {
            const unsigned u = A.col[t];
            if (u == (unsigned)v) continue;
            const unsigned su = __hip_atomic_load(&A.stamp[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (su == 0u || su >= round) continue;
            const float du = A.dist[u];
            if (du == dv && du + A.wgt[t] == dv && u < best) best = u;
        }
        if (best != 0xffffffffu) {
            A.pred[v] = (int)best - base;
            __hip_atomic_store(&A.stamp[v], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&A.cnt[4], 1u);
        }
    }
}

// ------------------------------------------------- literal second SSSP on the predecessor tree ---
__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_td_init(SkArgs A) {
    SK_VERTEX_LOOP(v) { A.tree_dist[v] = __uint_as_float(0x7f800000u); A.stamp[v] = 0u; }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < A.C; c += blockDim.x) A.q0[c] = (unsigned)(A.comp_off[c] + A.root_local[c]);
        if (threadIdx.x < 3) A.cnt[threadIdx.x] = threadIdx.x == 0 ? (unsigned)A.C : 0u;
    }
}
__global__ void k_sk_td_roots(SkArgs A) {
    for (int c = threadIdx.x; c < A.C; c += blockDim.x) A.tree_dist[A.comp_off[c] + A.root_local[c]] = 0.0f;
}

// breadth first down the tree: td[child] = td[parent] + |p_child - p_parent| (skeletonize.py:80-85)
__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_td_round(SkArgs A, int r) {
    const unsigned count = A.cnt[r % 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) A.cnt[(r + 2) % 3] = 0u;
    if (count == 0) return;
    const unsigned* q = (r & 1) ? A.q1 : A.q0;
    unsigned* qn = (r & 1) ? A.q0 : A.q1;
    unsigned* cnt_out = &A.cnt[(r + 1) % 3];
    const int lane = threadIdx.x & 63;
    const unsigned gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, tw = (gridDim.x * blockDim.x) >> 6;
    for (unsigned f = gw; f < count; f += tw) {
        const unsigned u = q[f];
        const int base = A.comp_off[A.comp_of[u]];
        const float du = A.tree_dist[u];
        for (uint32_t t = A.row_off[u] + lane; t < A.row_off[u + 1]; t += 64) {
            const unsigned v = A.col[t];
            if (A.pred[v] != (int)u - base) continue;
            if (atomicExch(&A.stamp[v], SK_MARK) == SK_MARK) continue;  // duplicate (u,v) edges: first one enqueues
            A.tree_dist[v] = du + sqrtf(sk_dist(A.pts + 3 * (int64_t)v, A.pts + 3 * (int64_t)u));
            qn[atomicAdd(cnt_out, 1u)] = v;
        }
    }
}

// ----------------------------------------------------------------------------- sample_tree ---
__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_lift_init(SkArgs A) {
    SK_VERTEX_LOOP(v) {
        A.anc[v * SK_ANC] = A.pred[v];
        A.branch_of[v] = -1;
        A.best[v] = SK_EMPTY64;
        A.pr[v] = make_float4(A.pts[3 * v], A.pts[3 * v + 1], A.pts[3 * v + 2], A.rad[v]);
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < A.C; c += blockDim.x) { A.s_done[c] = 0; A.s_nb[c] = 0; A.s_total[c] = 0; A.s_len[c] = 0; A.s_cursor[c] = 0; A.s_wide[c] = 0; }
}

// sample_tree's initial `distances` (path.py:71-72): vertices whose predecessor is not > 0 are never selectable
__device__ __forceinline__ float sk_initial_distance(const SkArgs& A, const float* distances, int64_t v) {
    return A.pred[v] > 0 ? distances[v] : -1.0f;
}

// sort keys: distance descending (masked vertices, -1, go last); second pass groups by component
__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_sort_keys(SkArgs A, const float* distances, uint32_t* keys, uint32_t* vals, int pass) {
    SK_VERTEX_LOOP(v) {
        if (pass == 0) { keys[v] = ~st_f2ord(sk_initial_distance(A, distances, v)); vals[v] = (uint32_t)v; }
        else keys[v] = (uint32_t)A.comp_of[vals[v]];
    }
}

__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_order_init(SkArgs A, const float* distances, float* order_init) {
    SK_VERTEX_LOOP(j) order_init[j] = sk_initial_distance(A, distances, A.order[j]);
}

// Ancestor table: anc[v*SK_ANC + k] = (k+1)-th ancestor of v (component-local id, -1 past the root).
// k = 0 is written by k_sk_lift_init; pass `span` (1, 2, 4, ..) fills entries [span, 2*span) from
// the table of the span-th ancestor.  A lane then reads "my j-th ancestor" with ONE load for j <= SK_ANC
// and hops SK_ANC levels at a time beyond (the trace of trace_route, path.py:9-16, without a pointer chase).
__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_anc_pass(SkArgs A, int span) {
    const int64_t total = A.m * span;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = i / span;
        const int k = (int)(i % span);
        const int base = A.comp_off[A.comp_of[v]];
        const int h = A.anc[v * SK_ANC + span - 1];  // span-th ancestor
        A.anc[v * SK_ANC + span + k] = h >= 0 ? A.anc[(int64_t)(base + h) * SK_ANC + k] : -1;
    }
}

// j-th ancestor of `far` (j = 0: itself); lanes of one SK_ANC-chunk share the hop chain (broadcast loads)
__device__ __forceinline__ int sk_ancestor(const SkArgs& A, int base, int far, unsigned j) {
    int cur = far;
    for (unsigned h = j / SK_ANC; h > 0 && cur >= 0; h--) cur = A.anc[(int64_t)(base + cur) * SK_ANC + (SK_ANC - 1)];
    const unsigned rem = j % SK_ANC;
    if (rem == 0u || cur < 0) return cur;
    return A.anc[(int64_t)(base + cur) * SK_ANC + rem - 1];
}

// claim: every (path vertex, x/y grid row) pair offers (d2, position) to the points within r of it.
// A 16-lane group per pair: the records of its cells are raced in parallel (a lone lane would chain
// one returning atomic per record).  First touches are staged in LDS and flushed with one reservation.
#define SK_LQ_CLAIM 2048
// keep_local: the caller is the component's own workgroup and will finish the branch itself -- if the
// staged list did not overflow it stays in LDS (returns true) instead of being flushed to `touched`.
__device__ __forceinline__ bool sk_claim_items(const SkArgs& A, int c, int base, int n, int len, float rp, const int* path,
                                               bool path_in_lds, int slice, int nslice, unsigned* lq, unsigned* lq_n,
                                               unsigned* lq_base, bool keep_local) {
    const StGrid* g = A.grid;
    const int xoff = A.comp_seg ? A.comp_seg[c] * g->seg_dim0 : 0;  // this cloud's slab of cells
    const float rp2 = rp * rp;
    int reach = rp > 0.0f ? (int)ceilf(rp / g->cell) : 0;
    if (reach < 1) reach = 1;
    const int side = 2 * reach + 1, nrow = side * side;
    const int64_t items = (int64_t)len * nrow;
    if (threadIdx.x == 0) *lq_n = 0;
    __syncthreads();
    const int sub = threadIdx.x & 15;
    const int64_t group = ((int64_t)slice * blockDim.x + threadIdx.x) >> 4, ngroup = ((int64_t)nslice * blockDim.x) >> 4;
    for (int64_t it = group; it < items; it += ngroup) {
        const int qi = (int)(it / nrow), rr = (int)(it % nrow);
        const float* pv = A.pts + 3 * (int64_t)(base + (path_in_lds ? path[qi] : ld(&path[qi])));
        const int x = (int)floorf((pv[0] - g->lo[0]) / g->cell) - reach + rr / side;
        const int y = (int)floorf((pv[1] - g->lo[1]) / g->cell) - reach + rr % side;
        if (x < 0 || x >= g->seg_dim0 || y < 0 || y >= g->dim[1]) continue;
        const int cz = (int)floorf((pv[2] - g->lo[2]) / g->cell);
        const int z0 = st_max(cz - reach, 0), z1 = st_min(cz + reach, g->dim[2] - 1);
        if (z0 > z1) continue;
        const int64_t row = ((int64_t)(xoff + x) * g->dim[1] + y) * g->dim[2];
        const uint32_t s = A.cell_start[row + z0], e = A.cell_start[row + z1 + 1];
        for (uint32_t t = s + sub; t < e; t += 16) {
            const float4 r4 = A.recs[t];
            const int p = (int)__float_as_uint(r4.w) - base;
            if (p < 0 || p >= n) continue;  // other component
            const float pp[3] = {r4.x, r4.y, r4.z};
            const float d2 = sk_dist(pp, pv);
            if (!(d2 < rp2)) continue;
            const unsigned long long pk = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)qi;
            const unsigned long long old = atomicMin(&A.best[base + p], pk);
            if (old == SK_EMPTY64) {
                const unsigned slot = atomicAdd(lq_n, 1u);
                if (slot < SK_LQ_CLAIM) lq[slot] = (unsigned)p; else A.touched[base + atomicAdd(&A.s_ntouched[c], 1u)] = (unsigned)p;
            }
        }
    }
    __syncthreads();
    const unsigned staged = *lq_n;
    if (keep_local && staged <= SK_LQ_CLAIM) return true;  // uniform
    const unsigned nloc = staged < SK_LQ_CLAIM ? staged : SK_LQ_CLAIM;
    if (threadIdx.x == 0 && nloc) *lq_base = atomicAdd(&A.s_ntouched[c], nloc);
    __syncthreads();
    for (unsigned i = threadIdx.x; i < nloc; i += blockDim.x) A.touched[base + *lq_base + i] = lq[i];
    __syncthreads();
    return false;
}

// ------------------------------------------------------------------------------ select ---
// One workgroup per component; speculative rounds of up to one branch per wavefront with an in-order replay (k_sk_select
// below).  The selection state is the termination set, a bitmap in LDS (SkBm), and the branch ids, stamped straight into
// the caller's array with a max (branch_ids keeps the last writer, ids grow with the order of the loop).
#define SK_HELP_MIN 8192
#define SK_HELP_PER 4096
#define SK_HELP_MAX 12
#define SK_HELP_POOL 160
#define SK_HELP_SEGS 32
#define SK_SMALL_WORK (1 << 18)  // candidate points x path vertices one workgroup takes on point-centric, unpruned
#define SK_ITERS_PER_LAUNCH 32
#define SK_LPATH 1024
#define SK_CHUNK 32  // path vertices per bounding box in the long-path claim

// The termination set of a component (sample_tree's `termination_pts`; `distances == -1` is the same set plus the vertices
// masked at the start, which sit at the tail of the distance order and are never looked at): one bit per vertex.  In LDS
// while the workgroup runs (loaded from / flushed to the component's words of A.term_bits at the launch boundaries), in
// global memory for a component too large for the LDS words.
#define SK_BM_WORDS 8192  // 262,144 vertices
struct SkBm {
    unsigned* lds;
    unsigned* glb;
    bool in_lds;
    bool agent;  // the component has helper workgroups on other compute units: branch ids are stamped at agent scope
};
// (a component too large for the LDS words keeps the set in global memory, at workgroup scope -- this XCD's L2: only its own
// workgroup reads and writes it.  Helper workgroups stamp a SECOND set of words at agent scope, which the component's workgroup
// folds into its own after every job.)
__device__ __forceinline__ bool bm_test(const SkBm& B, int v) {
    const unsigned w = B.in_lds ? B.lds[v >> 5] : ld_wg(&B.glb[v >> 5]);
    return (w >> (v & 31)) & 1u;
}
__device__ __forceinline__ void bm_set(const SkBm& B, int v) {
    if (B.in_lds) atomicOr(&B.lds[v >> 5], 1u << (v & 31));
    else wg_or(&B.glb[v >> 5], 1u << (v & 31));
}
// allocation / termination / branch-id stamp of a point (path.py:112-122,135-136).  branch_ids keeps the LAST writer; ids
// grow with the order of the loop, so "last" is a max -- commutative, which lets the lanes of a commit race.
__device__ __forceinline__ void sk_mark(const SkArgs& A, const SkBm& B, int base, int p, int id) {
    bm_set(B, p);
    if (id >= 0) { if (B.agent) (void)atomicMax(&A.branch_of[base + p], id); else wg_max(&A.branch_of[base + p], id); }
}

// on-path test of the points a chip-wide claim touched (path.py:35-40) + their stamps
__device__ __forceinline__ void sk_finish_branch(const SkArgs& A, const SkBm& B, int base, int len, int id, const int* path,
                                                 bool path_in_lds, const unsigned* touched, bool touched_in_lds, unsigned nt) {
    for (unsigned t = threadIdx.x; t < nt; t += blockDim.x) {
        const int p = (int)(touched_in_lds ? touched[t] : ld(&touched[t]));
        const unsigned long long pk = ld(&A.best[base + p]);
        A.best[base + p] = SK_EMPTY64;
        const float d2 = __uint_as_float((unsigned)(pk >> 32));
        const int qi = (int)(pk & 0xffffffffu);
        if (sqrtf(d2) < A.rad[base + (path_in_lds ? path[qi] : ld(&path[qi]))]) sk_mark(A, B, base, p, id);
    }
    for (int qi = threadIdx.x; qi < len; qi += blockDim.x) sk_mark(A, B, base, path_in_lds ? path[qi] : ld(&path[qi]), id);
    __syncthreads();
}

#define SK_TICK(i) do { if (A.ticks && tid == 0) { const long long now_ = wall_clock64(); tk[i] += now_ - t_last; t_last = now_; } } while (0)
#define SK_TICK_FLUSH() do { if (A.ticks && tid == 0) for (int i_ = 0; i_ < 8; i_++) A.ticks[i_] += tk[i_]; } while (0)

// ---- wavefront helpers (all 64 lanes must call) ----
__device__ __forceinline__ float wave_readlane_f(float v, int src) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), src)); }
__device__ __forceinline__ int wave_min_i(int v) { for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(v, d); v = o < v ? o : v; } return v; }
__device__ __forceinline__ int wave_max_i(int v) { for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(v, d); v = o > v ? o : v; } return v; }
__device__ __forceinline__ unsigned wave_max_u(unsigned v) { for (int d = 32; d > 0; d >>= 1) { const unsigned o = __shfl_xor(v, d); v = o > v ? o : v; } return v; }
__device__ __forceinline__ unsigned wave_or_u(unsigned v) { for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d); return v; }
__device__ __forceinline__ unsigned wave_incl_scan_u(unsigned v, int lane) {
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(v, (unsigned)d); if (lane >= d) v += o; }
    return v;
}

// LDS of k_sk_select.  Two modes share the space: `one` = a single branch worked on by the whole workgroup
// (paths up to SK_LPATH vertices), `slot[]` = one speculative branch per wavefront (short paths).
#define SK_WSLOTS 16   // = SK_MAX_WAVES
#define SK_WENT 32     // window entries looked at per round (those predicted to be swallowed get no slot)
#define SK_WPATH 64    // a speculative walk is ONE row of the ancestor table
#define SK_WCHUNK 8    // path vertices per bounding box in the claim of a speculative path
#define SK_WROWS 256   // (x, y) cell rows around a speculative path: up to four per lane
#define SK_WAVE_WORK (1 << 20)  // candidate points x path vertices per speculative branch
#define SK_WAVE_CAND 16384
#define SK_ROUND_ITEMS 32       // candidates per thread and round
#define SK_CL_KEEP 4            // claimed points a thread remembers in LDS; further ones are found again through a bit mask
struct SkSelOne {
    int lpath[SK_LPATH];
    float lpx[SK_LPATH], lpy[SK_LPATH], lpz[SK_LPATH], lpr[SK_LPATH];
    uint32_t row_off[1025], row_first[1024];
};
struct SkSelSlot {
    int path[SK_WPATH];   // root side first
    float4 p[SK_WPATH];   // position, radius
    uint32_t row_off[SK_WROWS + 1], row_first[SK_WROWS];
    int lo[3], hi[3];  // cell bounding box of the path (LDS min / max)
    unsigned rk;       // ordered bits of the largest radius
    float4 blo[SK_WPATH / SK_WCHUNK], bhi[SK_WPATH / SK_WCHUNK];  // bounding boxes of SK_WCHUNK consecutive path vertices
};
union SkSelLds {
    SkSelOne one;
    SkSelSlot slot[SK_WSLOTS];
};

// row r with row_off[r] <= t < row_off[r+1]
__device__ __forceinline__ int sk_find_row(const uint32_t* row_off, int nrows, uint32_t t) {
    int lo = 0, hi = nrows;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (row_off[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}

// Watch table of a speculative round (LDS, open addressing): the points whose marks the replay will ask for -- the tips of
// the round's entries, every slot's walk and the vertex its parent id is read from: at most SK_WENT + SK_WSLOTS x
// (SK_WPATH + 1) keys, so the table (SK_WT slots) never fills.  A claim of point p by slot s ORs bit s into p's word IF p is
// watched; nobody ever asks about the other claimed points.
#define SK_WT 4096
#define SK_WT_EMPTY 0xffffffffu
__device__ __forceinline__ unsigned sk_wt_hash(unsigned key) { return (key * 0x9E3779B1u) >> 20; }  // 12 bits
__device__ __forceinline__ void sk_wt_insert(unsigned* wkey, unsigned* wmask, unsigned key, unsigned bits) {
    for (unsigned h = sk_wt_hash(key);; h = (h + 1u) & (SK_WT - 1u)) {
        const unsigned old = atomicCAS(&wkey[h], SK_WT_EMPTY, key);
        if (old == SK_WT_EMPTY || old == key) { if (bits) atomicOr(&wmask[h], bits); return; }
    }
}
__device__ __forceinline__ int sk_wt_find(const unsigned* wkey, unsigned key) {
    for (unsigned h = sk_wt_hash(key);; h = (h + 1u) & (SK_WT - 1u)) {
        const unsigned k = wkey[h];
        if (k == key) return (int)h;
        if (k == SK_WT_EMPTY) return -1;
    }
}

// select: one workgroup per component.  sample_tree (path.py:49-140) is a sequential greedy loop -- take the
// farthest unallocated vertex, walk to the skeleton, claim the points within the path's radius -- but
// branches far apart do not interact, so each ROUND speculates: wavefront s takes the s-th farthest
// unallocated vertex, walks it and marks (bit s of the point's watch-table word) every watched point its branch would allocate, all
// against the state at the start of the round.  A scan in order then replays the sequential semantics from
// the marks: a tip already marked by an accepted earlier slot would never have been selected (skipped); a
// walk (or the vertex the parent id is read from) touched by an accepted earlier slot would have come out
// differently -- the round stops there and the rest is retried next round; everything else is exactly what the
// sequential loop produces and is committed (ids / offsets by prefix over the accepted slots;
// branch_of = max id = last writer).  Slot 0 is always accepted, so every round makes progress.  A path
// too long for one wavefront is worked on by the whole workgroup (`one` mode); one too long even for that
// is handed to the chip-wide k_sk_claim and finished at the head of the next launch.
// The long-path claim, point-centric and exact (see k_sk_select, "A long path"): rows [r_begin, r_end) of the (x, y) cell rows
// around the path are taken W at a time; every candidate point of those rows walks the path's chunks.  The path (positions,
// radii) and its chunk boxes sit in LDS.  HELPER = false: the component's own workgroup (termination bits in its LDS bitmap);
// HELPER = true: a helper workgroup on another compute unit -- bits and branch ids go to global memory at agent scope, the
// component's workgroup folds the bits into its bitmap when the job is done.
template <bool HELPER>
__device__ __forceinline__ void sk_long_rows(const SkArgs& A, const SkBm& B, SkSelOne& O, const float (*cb_lo)[SK_LPATH / SK_CHUNK],
                                             const float (*cb_hi)[SK_LPATH / SK_CHUNK], uint32_t* s_scan, int base, int n, int xoff,
                                             int x0, int y0, int z0, int z1, int ny, int r_begin, int r_end, int len, int csz, float rp2,
                                             int id) {
    const StGrid* g = A.grid;
    const float4* __restrict__ recs = A.recs;
    const int tid = threadIdx.x, W = (int)blockDim.x;
    const int nch = (len + csz - 1) / csz;
    for (int r0 = r_begin; r0 < r_end; r0 += W) {
        const int rr = r0 + tid, nr = r_end - r0 < W ? r_end - r0 : W;
        uint32_t cnt = 0, first = 0;
        if (rr < r_end) {
            const int64_t row = ((int64_t)(xoff + x0 + rr / ny) * g->dim[1] + (y0 + rr % ny)) * g->dim[2];
            first = A.cell_start[row + z0];
            cnt = A.cell_start[row + z1 + 1] - first;
        }
        uint32_t tot;
        const uint32_t off = block_exclusive_scan(cnt, s_scan, &tot);  // (its first barrier: the previous rows' tables are dead)
        if (tid < nr) { O.row_off[tid] = off; O.row_first[tid] = first; }
        if (tid == 0) O.row_off[nr] = tot;
        __syncthreads();
        for (uint32_t t = (uint32_t)tid; t < tot; t += (uint32_t)W) {
            const int row = sk_find_row(O.row_off, nr, t);
            const float4 r4 = recs[O.row_first[row] + (t - O.row_off[row])];
            const int p = (int)__float_as_uint(r4.w) - base;
            if (p < 0 || p >= n) continue;  // other component
            float bd2 = __uint_as_float(0x7f800000u), bw = 0.0f;
            for (int ch = 0; ch < nch; ch++) {
                float e[3];
                const float pv[3] = {r4.x, r4.y, r4.z};
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const float below = cb_lo[a][ch] - pv[a], above = pv[a] - cb_hi[a][ch];
                    const float m = below > above ? below : above;
                    e[a] = m > 0.0f ? m : 0.0f;
                }
                float lb = e[0] * e[0];
                float tt = e[1] * e[1];
                lb = lb + tt;
                tt = e[2] * e[2];
                lb = lb + tt;
                if (lb >= bd2 || lb >= rp2) continue;  // nothing in this chunk can be strictly nearer / inside the radius
                const int q1 = (ch + 1) * csz < len ? (ch + 1) * csz : len;
                for (int qi = ch * csz; qi < q1; qi++) {  // ascending: ties keep the first path vertex
                    const float dx = r4.x - O.lpx[qi], dy = r4.y - O.lpy[qi], dz = r4.z - O.lpz[qi];
                    float d2 = dx * dx;
                    float t2 = dy * dy;
                    d2 = d2 + t2;
                    t2 = dz * dz;
                    d2 = d2 + t2;
                    if (d2 < bd2) { bd2 = d2; bw = O.lpr[qi]; }
                }
            }
            if (bd2 < rp2 && sqrtf(bd2) < bw) {  // path.py:35-40
                if (HELPER) {
                    (void)atomicOr(&B.glb[p >> 5], 1u << (p & 31));
                    if (id >= 0) (void)atomicMax(&A.branch_of[base + p], id);
                } else {
                    sk_mark(A, B, base, p, id);
                }
            }
        }
    }
}

// chunk boxes of the path in O (every thread calls; a barrier must follow before they are read)
__device__ __forceinline__ void sk_chunk_boxes(const SkSelOne& O, float (*cb_lo)[SK_LPATH / SK_CHUNK], float (*cb_hi)[SK_LPATH / SK_CHUNK], int len,
                                               int csz) {
    const int tid = threadIdx.x, nch = (len + csz - 1) / csz;
    if (tid < nch) {
        float lo[3] = {__uint_as_float(0x7f800000u), __uint_as_float(0x7f800000u), __uint_as_float(0x7f800000u)};
        float hi[3] = {__uint_as_float(0xff800000u), __uint_as_float(0xff800000u), __uint_as_float(0xff800000u)};
        for (int qi = tid * csz; qi < len && qi < (tid + 1) * csz; qi++) {
            const float v[3] = {O.lpx[qi], O.lpy[qi], O.lpz[qi]};
#pragma unroll
            for (int a = 0; a < 3; a++) { lo[a] = v[a] < lo[a] ? v[a] : lo[a]; hi[a] = v[a] > hi[a] ? v[a] : hi[a]; }
        }
#pragma unroll
        for (int a = 0; a < 3; a++) { cb_lo[a][tid] = lo[a]; cb_hi[a][tid] = hi[a]; }
    }
}

// agent-scope stores / loads of the job words (see SkJob)
__device__ __forceinline__ void st_ai(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_ai(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_au(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_au(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Time-outs of the helper protocol (100 MHz wall clock; SkArgs.help_*; tuning codes 16-18 in microseconds).  They exist so that a
// workgroup that never becomes resident cannot hang the GPU; none is an error: a helper whose life time ran out -- or whose component's
// workgroup has not announced itself (`alive`) within `help_announce`: the chip is full of other work, the helper only takes a
// compute unit away from it -- says so (`gone`) and leaves; a component's workgroup that waits too long for an answer claims the
// helpers' shares itself (the stamps are idempotent: bits are ORed, branch ids are a maximum) and goes on without helpers.  A select
// launch of a very large cloud may legitimately last seconds: the life time is generous.  tests/test_helpers.py shortens all three
// to exercise every fall-back against the shipped library.
#define SK_HELP_LIFETIME_US 30000000ll  // 30 s
#define SK_HELP_TIMEOUT_US 5000000ll    // 5 s waiting for one job's answers
#define SK_HELP_ANNOUNCE_US 50000ll     // 50 ms for the component's workgroup to become resident

// Ordering across compute units (helper protocol).  Everything a helper and its component's workgroup exchange is written and read
// with agent-scope atomics (served at the memory side, never from a stale L2 line of another XCD), but a RELAXED atomic only says
// where an access is performed, not when: stores and no-return atomics of a wavefront may still be in flight when a later store of
// ANOTHER wavefront (the `done` count, the `seq` word) becomes visible -- s_barrier does not wait for vector memory.  So: every
// wavefront that wrote payload issues an agent-scope release fence (s_waitcnt vmcnt(0) + write-back) BEFORE the workgroup barrier in
// front of the flag, the flag is written with release semantics, and the reader fences with acquire after it has seen the flag.
__device__ __forceinline__ void sk_release_agent() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ void sk_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

__global__ void __launch_bounds__(1024) k_sk_select(SkArgs A) {
    long long t_last = A.ticks ? wall_clock64() : 0;
    __shared__ long long tk[8];  // phase timers (developer aid), touched by thread 0 only
    if (A.ticks && threadIdx.x == 0) for (int i_ = 0; i_ < 8; i_++) tk[i_] = 0;
    __shared__ unsigned long long s_red[SK_MAX_WAVES];
    __shared__ SkSelLds L;
    __shared__ unsigned cl_list[SK_CL_KEEP][1024];  // claimed points of this round, SK_CL_KEEP private entries per thread
    __shared__ uint32_t s_scan[SK_MAX_WAVES + 1];
    __shared__ unsigned wt_key[SK_WT], wt_mask[SK_WT];  // watch table (see above)
    __shared__ float cb_lo[3][SK_LPATH / SK_CHUNK], cb_hi[3][SK_LPATH / SK_CHUNK];  // chunk boxes of a long path
    __shared__ unsigned bm_words[SK_BM_WORDS];  // termination set (see SkBm)
    __shared__ int s_lo[3], s_hi[3];
    __shared__ int s_term;
    __shared__ int w_cnt[SK_WSLOTS], w_tail[SK_WSLOTS], cand_v[SK_WENT];  // the round's entries: first live window vertices
    __shared__ float4 cand_p[SK_WENT];
    __shared__ int sl_ent[SK_WSLOTS], s_nb2, s_tot2;
    __shared__ unsigned s_alive;
    __shared__ int sl_len[SK_WSLOTS], sl_term[SK_WSLOTS], sl_parent[SK_WSLOTS], sl_nrows[SK_WSLOTS], sl_ncand[SK_WSLOTS],
        sl_big[SK_WSLOTS], sl_id[SK_WSLOTS], sl_off[SK_WSLOTS];
    __shared__ float sl_rp[SK_WSLOTS];
    __shared__ unsigned sl_walkm[SK_WSLOTS];
    const int tid = threadIdx.x, W = (int)blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = (W + 63) >> 6;
    if ((int)blockIdx.x < A.n_helpers) {
        // ------------------------------------------------------------------ a helper workgroup ---
        // serves ONE large component: waits for its workgroup to post a long-path claim (SkJob), takes its share of the cell
        // rows around that path, stamps what it claims at agent scope, reports, waits for the next one -- until `quit`.
        __shared__ unsigned h_seq, h_quit;
        const int hc = A.hl_comp[blockIdx.x];
        if (hc < 0 || A.s_done[hc]) return;  // (a finished component's workgroup leaves at once, too: nobody would send `quit`)
        const int rank = A.hl_rank[blockIdx.x], nranks = A.c_nhelp[hc] + 1;
        const int hbase = A.comp_off[hc], hn = A.comp_off[hc + 1] - hbase;
        const int hxoff = A.comp_seg ? A.comp_seg[hc] * A.grid->seg_dim0 : 0;
        SkBm HB;
        HB.lds = nullptr; HB.glb = A.help_bits + (hbase >> 5) + hc; HB.in_lds = false; HB.agent = true;  // (the helpers' own words)
        SkJob* J = &A.jobs[hc];
        unsigned seen = 0u;
        const long long t_start = wall_clock64();
        for (;;) {
            if (tid == 0) {
                h_seq = ld_au(&J->seq);
                unsigned q_ = ld_au(&J->quit);
                if (!q_) {
                    const long long waited = wall_clock64() - t_start;
                    if (waited > A.help_lifetime || (waited > A.help_announce && !ld_au(&J->alive))) { st_au(&J->gone, 1u); q_ = 1u; }
                }
                h_quit = q_;
            }
            __syncthreads();
            const unsigned sq = h_seq, qt = h_quit;
            __syncthreads();
            if (qt != 0u) return;
            if (sq == seen) { __builtin_amdgcn_s_sleep(16); continue; }
            seen = sq;
            sk_acquire_agent();  // pairs with the release store of `seq`: the job words and the path are visible
            const int len = ld_ai(&J->len), id = ld_ai(&J->id), csz = ld_ai(&J->csz), nrows = ld_ai(&J->nrows), ny = ld_ai(&J->ny);
            const int x0 = ld_ai(&J->x0), y0 = ld_ai(&J->y0), z0 = ld_ai(&J->z0), z1 = ld_ai(&J->z1), poff = ld_ai(&J->path_off);
            const float rp = __uint_as_float((unsigned)ld_ai((const int*)&J->rp));
            for (int qi = tid; qi < len; qi += W) {  // the path (root side first) and its positions / radii
                const int v = ld_ai(&A.path_verts[hbase + poff + qi]);
                const float4 q4 = A.pr[hbase + v];
                L.one.lpx[qi] = q4.x; L.one.lpy[qi] = q4.y; L.one.lpz[qi] = q4.z; L.one.lpr[qi] = q4.w;
            }
            __syncthreads();
            sk_chunk_boxes(L.one, cb_lo, cb_hi, len, csz);
            __syncthreads();
            const int per = (nrows + nranks - 1) / nranks;
            const int rb = rank * per, re = rb + per < nrows ? rb + per : nrows;
            sk_long_rows<true>(A, HB, L.one, cb_lo, cb_hi, s_scan, hbase, hn, hxoff, x0, y0, z0, z1, ny, rb, re, len, csz, rp * rp, id);
            sk_release_agent();  // EVERY wavefront: its stamps (no-return atomics) have been performed ...
            __syncthreads();     // ... before any wavefront is past this barrier
            if (tid == 0) (void)__hip_atomic_fetch_add(&J->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int c = (int)blockIdx.x - A.n_helpers;
    if (A.s_done[c]) return;
    const int base = A.comp_off[c], n = A.comp_off[c + 1] - base;
    const unsigned* order = A.order + base;
    unsigned* tmp = A.q0 + base;
    const float4* __restrict__ recs = A.recs;
    const StGrid* g = A.grid;
    const int xoff = A.comp_seg ? A.comp_seg[c] * g->seg_dim0 : 0;  // this cloud's slab of grid cells (batched call)
    // the termination set: this component's words of A.term_bits ((base >> 5) + c: no two components share a word)
    SkBm B;
    B.lds = bm_words;
    B.glb = A.term_bits + (base >> 5) + c;
    const int nwords = (n + 31) >> 5;
    B.in_lds = nwords <= SK_BM_WORDS;
    int nhelp = A.n_helpers > 0 ? A.c_nhelp[c] : 0;  // helper workgroups of this component (long-path claims); 0 once one was lost
    __shared__ unsigned s_lost;
    B.agent = nhelp > 0;
    SkJob* J = &A.jobs[c];
    unsigned job_seq = 0u;
    if (B.in_lds) for (int i = tid; i < nwords; i += W) bm_words[i] = B.glb[i];
    if (nhelp > 0 && tid == 0) st_au(&J->alive, 1u);  // (a helper that does not see this within `help_announce` gives up)
    __syncthreads();
#define SK_QUIT_HELPERS() do { if (A.n_helpers > 0 && A.c_nhelp[c] > 0 && tid == 0) st_au(&J->quit, 1u); } while (0)  // (also when this workgroup does not use them: a component too large for the LDS bitmap)
#define SK_FLUSH_BM()                                                             \
    do {                                                                          \
        __syncthreads();                                                          \
        if (B.in_lds) for (int i_ = tid; i_ < nwords; i_ += W) B.glb[i_] = bm_words[i_]; \
    } while (0)
    // a long path left over from the previous launch: k_sk_claim filled `touched`
    {
        const int plen = A.s_len[c];
        if (plen > 0)
            sk_finish_branch(A, B, base, plen, A.s_cur_id[c], A.path_verts + base + A.s_cur_off[c], false, A.touched + base, false,
                             A.s_ntouched[c]);
    }
    SK_TICK(0);
    int win_base = A.s_cursor[c], total = A.s_total[c], nb = A.s_nb[c];  // per-component state lives in registers
    int wv = -1;         // my window entry: component-local vertex, -1 = none
    float wx = 0.0f, wy = 0.0f, wz = 0.0f, wr = 0.0f;  // ... its position and radius (for step 1b)
    bool wtail = false;  // my entry marks the end of the selectable vertices (initial distance <= 0, or end of the list)
    bool need_fill = true;
    for (int iter = 0; iter < A.iters_per_launch; iter++) {
        // 1. the farthest unallocated vertices (path.py:92) = the first live entries of the distance-sorted order.
        //    A window of W entries is held in registers / LDS; its flags are cleared as points get allocated, so
        //    finding the next tips costs no global access until the window is used up.
        int nc = 0, ne = 0;
        bool exhausted = false;
        for (;;) {
            if (need_fill) {
                const int j = win_base + tid;
                wv = -1; wtail = false;
                if (j < n) {
                    wv = (int)order[j] - base;
                    wtail = !(A.order_init[base + j] > 0.0f);
                    if (!wtail) {
                        const float4 q = A.pr[base + wv];
                        wx = q.x; wy = q.y; wz = q.z; wr = q.w;
                    }
                } else if (j == n) {
                    wtail = true;
                }
                need_fill = false;
            }
            // "still unallocated" = not in the termination set (the vertices masked at the start sit behind the tail)
            const bool live = wv >= 0 && !wtail && !bm_test(B, wv);
            const unsigned long long lb = __ballot(live), tb = __ballot(wtail);
            __syncthreads();  // (w_cnt / cand_v of the previous pass have been read)
            if (lane == 0) { w_cnt[wave] = __popcll(lb); w_tail[wave] = tb != 0ull; }
            for (int i = tid; i < SK_WT; i += W) { wt_key[i] = SK_WT_EMPTY; wt_mask[i] = 0u; }  // this round's watch table
            __syncthreads();
            int before = 0, tot = 0, anytail = 0;
            for (int w = 0; w < nw; w++) { const int k = w_cnt[w]; before += w < wave ? k : 0; tot += k; anytail |= w_tail[w]; }
            if (tot == 0) {
                if (anytail) { exhausted = true; break; }
                win_base += W; need_fill = true;  // nothing left in this window
                continue;
            }
            const int rank = before + __popcll(lb & ((1ull << lane) - 1ull));
            if (live && rank < SK_WENT) { cand_v[rank] = wv; cand_p[rank] = make_float4(wx, wy, wz, wr); }
            ne = tot < SK_WENT ? tot : SK_WENT;
            __syncthreads();
            break;
        }
        if (exhausted) {  // path.py:94-95 (uniform)
            SK_QUIT_HELPERS();
            SK_FLUSH_BM();
            if (tid == 0) {
                A.s_done[c] = 1; A.s_len[c] = 0; A.s_wide[c] = 0; A.n_branches[c] = nb; A.s_nb[c] = nb; A.s_total[c] = total;
                atomicAdd(&A.cnt[6], (unsigned)nb);     // cloud totals: branches and path vertices arrive with the progress
                atomicAdd(&A.cnt[7], (unsigned)total);  // read-back, so assembling the skeleton needs no count read-back
                atomicAdd(&A.cnt[5], 1u);
            }
            SK_TICK_FLUSH();
            return;
        }
        SK_TICK(7);
        // 1b. which entries get a wavefront?  A tip within the radius of an earlier chosen tip will almost surely
        //     be swallowed by that branch: it gets no slot (if the guess is wrong the replay below simply stops
        //     there).  Every wavefront runs this little greedy pass itself -- no barrier, no LDS.
        int my_slot = -1;  // lane e < ne: slot of entry e (-1: none)
        int my_ent = 0;    // the entry this wavefront speculates on (wave < nc)
        int ent_v = -1;
        {
            float ex = 0.0f, ey = 0.0f, ez = 0.0f, er = 0.0f;
            if (lane < ne) {
                ent_v = cand_v[lane];
                const float4 e4 = cand_p[lane];
                ex = e4.x; ey = e4.y; ez = e4.z; er = e4.w * A.prune_factor;
                if (wave == 0) sk_wt_insert(wt_key, wt_mask, (unsigned)ent_v, 0u);  // the replay asks who claimed this tip
            }
            int shadowed = 0, chosen = 0;
            for (int u = 0; u < ne; u++) {
                if (__builtin_amdgcn_readlane(shadowed, u)) continue;
                if (chosen == nw) { ne = u; break; }  // out of wavefronts: the round ends before this entry
                chosen++;
                const float ux = wave_readlane_f(ex, u), uy = wave_readlane_f(ey, u), uz = wave_readlane_f(ez, u),
                            ur = wave_readlane_f(er, u);
                const float dx = ex - ux, dy = ey - uy, dz = ez - uz;
                if (lane > u && dx * dx + dy * dy + dz * dz < ur * ur) shadowed = 1;
            }
            const unsigned long long cb = __ballot(lane < ne && !shadowed);
            nc = __popcll(cb);
            if (lane < ne && !shadowed) my_slot = __popcll(cb & ((1ull << lane) - 1ull));
            unsigned long long rest = cb;
            for (int k = 0; k < wave && rest; k++) rest &= rest - 1ull;
            my_ent = rest ? __ffsll(rest) - 1 : 0;
        }
        SK_TICK(1);
        // 2. speculative walks: wavefront s traces its entry through ONE ancestor-table row (trace_route,
        //    path.py:9-16: lane j inspects the j-th ancestor; the first allocated one, or the step past the root,
        //    ends the walk), then gathers radius / position of its path, the (x, y) rows of grid cells around it
        //    and how many candidate points they hold.
        if (wave < nc) {
            SkSelSlot& S = L.slot[wave];
            const int tip = cand_v[my_ent];
            const int node = lane == 0 ? tip : A.anc[(int64_t)(base + tip) * SK_ANC + lane - 1];
            const bool end = node < 0 || bm_test(B, node);
            const unsigned long long eb = __ballot(end);
            int big = eb == 0ull, len = 0, termv = -1, nrows = 0, ncand = 0, parent = -1;
            float rp = 0.0f;
            if (!big) {
                len = __ffsll(eb) - 1;
                termv = __shfl(node, len);
                const int qi = len - 1 - lane;  // walk order -> root side first
                if (lane < 3) { S.lo[lane] = 0x7fffffff; S.hi[lane] = (int)0x80000000; }
                if (lane == 3) S.rk = 0u;
                __builtin_amdgcn_wave_barrier();
                // the parent id is read BEFORE anything is stamped (path.py:128-136); termination -1 reads
                // branch_ids[-1] = the last vertex (quirk kept)
                if (lane == 0 && len >= 2) parent = ld(&A.branch_of[base + (termv < 0 ? n - 1 : termv)]);
                // watched: the walk (marked by its own slot) and the vertex the parent id is read from
                if (lane < len) sk_wt_insert(wt_key, wt_mask, (unsigned)node, 1u << wave);
                else if (lane == len) sk_wt_insert(wt_key, wt_mask, (unsigned)(termv < 0 ? n - 1 : termv), 0u);
                if (lane < len) {
                    const float4 q4 = A.pr[base + node];
                    const float x = q4.x, y = q4.y, z = q4.z, r = q4.w;
                    S.path[qi] = node; S.p[qi] = q4;
                    const int cx = (int)floorf((x - g->lo[0]) / g->cell), cy = (int)floorf((y - g->lo[1]) / g->cell),
                              cz = (int)floorf((z - g->lo[2]) / g->cell);
                    atomicMax(&S.rk, st_f2ord(r));  // path.py:31
                    atomicMin(&S.lo[0], cx); atomicMin(&S.lo[1], cy); atomicMin(&S.lo[2], cz);
                    atomicMax(&S.hi[0], cx); atomicMax(&S.hi[1], cy); atomicMax(&S.hi[2], cz);
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < (len + SK_WCHUNK - 1) / SK_WCHUNK) {  // chunk boxes for the claim
                    float4 lo = S.p[lane * SK_WCHUNK], hi = lo;
                    for (int qj = lane * SK_WCHUNK + 1; qj < len && qj < (lane + 1) * SK_WCHUNK; qj++) {
                        const float4 q = S.p[qj];
                        lo.x = q.x < lo.x ? q.x : lo.x; lo.y = q.y < lo.y ? q.y : lo.y; lo.z = q.z < lo.z ? q.z : lo.z;
                        hi.x = q.x > hi.x ? q.x : hi.x; hi.y = q.y > hi.y ? q.y : hi.y; hi.z = q.z > hi.z ? q.z : hi.z;
                    }
                    S.blo[lane] = lo; S.bhi[lane] = hi;
                }
                rp = st_ord2f(S.rk);
                int reach = rp > 0.0f ? (int)ceilf(rp / g->cell) : 0;
                if (reach < 1) reach = 1;
                const int x0 = st_max(S.lo[0] - reach, 0), x1 = st_min(S.hi[0] + reach, g->seg_dim0 - 1);
                const int y0 = st_max(S.lo[1] - reach, 0), y1 = st_min(S.hi[1] + reach, g->dim[1] - 1);
                const int z0 = st_max(S.lo[2] - reach, 0), z1 = st_min(S.hi[2] + reach, g->dim[2] - 1);
                const int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
                nrows = (nx > 0 && ny > 0 && z0 <= z1) ? nx * ny : 0;
                big = nrows > SK_WROWS;
                if (!big) {
                    uint32_t cnt[SK_WROWS / 64], first[SK_WROWS / 64];
#pragma unroll
                    for (int ch = 0; ch < SK_WROWS / 64; ch++) {  // all loads first, then the scans
                        const int rr = ch * 64 + lane;
                        cnt[ch] = 0u; first[ch] = 0u;
                        if (rr < nrows) {
                            const int64_t row = ((int64_t)(xoff + x0 + rr / ny) * g->dim[1] + (y0 + rr % ny)) * g->dim[2];
                            first[ch] = A.cell_start[row + z0];
                            cnt[ch] = A.cell_start[row + z1 + 1] - first[ch];
                        }
                    }
                    uint32_t run = 0u;
#pragma unroll
                    for (int ch = 0; ch < SK_WROWS / 64; ch++) {
                        if (ch * 64 >= nrows) break;  // wave-uniform
                        const uint32_t inc = wave_incl_scan_u(cnt[ch], lane);
                        const int rr = ch * 64 + lane;
                        if (rr < nrows) { S.row_off[rr] = run + inc - cnt[ch]; S.row_first[rr] = first[ch]; }
                        run += __shfl(inc, 63);
                    }
                    ncand = (int)run;
                    if (lane == 0) S.row_off[nrows] = (uint32_t)ncand;
                    big = ncand > SK_WAVE_CAND || ncand > SK_ROUND_ITEMS * W || (int64_t)ncand * len > A.wave_work;
                }
            }
            if (lane == 0) {
                sl_len[wave] = len; sl_term[wave] = termv; sl_parent[wave] = parent; sl_nrows[wave] = nrows; sl_ncand[wave] = ncand;
                sl_big[wave] = big; sl_rp[wave] = rp; sl_ent[wave] = my_ent;
            }
        }
        __syncthreads();
        SK_TICK(2);
        if (!sl_big[0]) {
            // slots after the first oversized one (or past the per-round item budget) wait for a later round.
            // pre[k] = candidates of the slots before k: workgroup-uniform, so it lives in scalar registers.
            int pre[SK_WSLOTS + 1];
            pre[0] = 0;
            {
                // lane k reads slot k's numbers once (one LDS round trip), the scan below takes them out with readlane
                int cand_l = 0, big_l = 0, ent_l = 0;
                if (lane < SK_WSLOTS && lane < nc) { cand_l = sl_ncand[lane]; big_l = sl_big[lane]; ent_l = sl_ent[lane]; }
                bool open = true;
                int k_cut = nc;
#pragma unroll
                for (int k = 0; k < SK_WSLOTS; k++) {
                    int add = 0;
                    if (k < nc && open) {
                        const int cand_k = __builtin_amdgcn_readlane(cand_l, k), big_k = __builtin_amdgcn_readlane(big_l, k);
                        if (big_k || pre[k] + cand_k > SK_ROUND_ITEMS * W) { open = false; k_cut = k; }
                        else add = cand_k;
                    }
                    pre[k + 1] = pre[k] + add;
                }
                if (k_cut < nc) { ne = __builtin_amdgcn_readlane(ent_l, k_cut); nc = k_cut; }
            }
            const int T = pre[SK_WSLOTS];
            // 3. claims (select_path_points, path.py:19-46), point-centric: the candidates of ALL slots are dealt
            //    out over the workgroup; each finds ITS nearest path vertex from LDS -- no atomics but the mark.
            unsigned cl_bits = 0u;  // bit k: my k-th item was claimed but did not fit cl_list
            int cl_n = 0;
            const int nround = (T + W - 1) / W;
            for (int k0 = 0; k0 < nround; k0 += 4) {
                float4 r4[4];
                int ss[4];
                {
                    // the rows of four candidates at once: the four binary searches advance in lockstep (a fixed number of
                    // steps), so that their dependent LDS reads overlap instead of queueing behind each other
                    const uint32_t* ro[4];
                    int lo[4], hi[4];
                    uint32_t tt[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int gi = (k0 + u) * W + tid;
                        ss[u] = -1;
                        ro[u] = L.slot[0].row_off; lo[u] = 0; hi[u] = 1; tt[u] = 0u;
                        if (k0 + u < nround && gi < T) {
                            int sidx = 0, acc = 0;
#pragma unroll
                            for (int k = 1; k < SK_WSLOTS; k++)
                                if (gi >= pre[k]) { sidx = k; acc = pre[k]; }  // pre[] is non-decreasing and ends at T > gi
                            ss[u] = sidx;
                            ro[u] = L.slot[sidx].row_off; hi[u] = sl_nrows[sidx]; tt[u] = (uint32_t)(gi - acc);
                        }
                    }
#pragma unroll
                    for (int step = 0; step < 8; step++) {  // SK_WROWS = 2^8 rows at most; a converged search repeats its last step
                        uint32_t v[4];
                        int mid[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { mid[u] = (lo[u] + hi[u]) >> 1; v[u] = ro[u][mid[u]]; }
#pragma unroll
                        for (int u = 0; u < 4; u++) { if (v[u] <= tt[u]) lo[u] = mid[u]; else hi[u] = mid[u]; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        r4[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (ss[u] >= 0) r4[u] = recs[L.slot[ss[u]].row_first[lo[u]] + (tt[u] - ro[u][lo[u]])];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (ss[u] < 0) continue;
                    const int p = (int)__float_as_uint(r4[u].w) - base;
                    if (p < 0 || p >= n) continue;  // other component
                    const SkSelSlot& S = L.slot[ss[u]];
                    const int len = sl_len[ss[u]];
                    const float rp = sl_rp[ss[u]], rp2 = rp * rp;
                    float bd2 = __uint_as_float(0x7f800000u);
                    float bw = 0.0f;
                    // ascending: ties keep the first path vertex.  The path is cut into chunks of SK_WCHUNK vertices with their
                    // bounding boxes; a chunk whose box is no closer than the best vertex so far (or than the path radius) is
                    // skipped -- exact for the reason given at the long-path claim below (same float32 operations, monotone
                    // rounding); most candidates reject most chunks.
#define SK_NEAREST(q)                                                                         \
    {                                                                                          \
        const float dx = r4[u].x - (q).x, dy = r4[u].y - (q).y, dz = r4[u].z - (q).z;          \
        float d2 = dx * dx;                                                                    \
        float tq = dy * dy;                                                                    \
        d2 = d2 + tq;                                                                          \
        tq = dz * dz;                                                                          \
        d2 = d2 + tq;                                                                          \
        if (d2 < bd2) { bd2 = d2; bw = (q).w; }                                                \
    }
                    for (int ch = 0; ch * SK_WCHUNK < len; ch++) {
                        const float4 lo = S.blo[ch], hi = S.bhi[ch];
                        float e0 = lo.x - r4[u].x, e1 = lo.y - r4[u].y, e2 = lo.z - r4[u].z;
                        const float a0 = r4[u].x - hi.x, a1 = r4[u].y - hi.y, a2 = r4[u].z - hi.z;
                        e0 = e0 > a0 ? e0 : a0; e1 = e1 > a1 ? e1 : a1; e2 = e2 > a2 ? e2 : a2;
                        e0 = e0 > 0.0f ? e0 : 0.0f; e1 = e1 > 0.0f ? e1 : 0.0f; e2 = e2 > 0.0f ? e2 : 0.0f;
                        float lb = e0 * e0;
                        float tb = e1 * e1;
                        lb = lb + tb;
                        tb = e2 * e2;
                        lb = lb + tb;
                        if (lb >= bd2 || lb >= rp2) continue;
                        const int q0 = ch * SK_WCHUNK;
                        if (q0 + SK_WCHUNK <= len) {
#pragma unroll
                            for (int j = 0; j < SK_WCHUNK; j += 4) {  // four path vertices per step: their LDS reads are in flight together
                                const float4 p0 = S.p[q0 + j], p1 = S.p[q0 + j + 1], p2 = S.p[q0 + j + 2], p3 = S.p[q0 + j + 3];
                                SK_NEAREST(p0) SK_NEAREST(p1) SK_NEAREST(p2) SK_NEAREST(p3)
                            }
                        } else {
                            for (int qi = q0; qi < len; qi++) {
                                const float4 q = S.p[qi];
                                SK_NEAREST(q)
                            }
                        }
                    }
#undef SK_NEAREST
                    if (bd2 < rp2 && sqrtf(bd2) < bw) {  // path.py:35-40
                        const int h = sk_wt_find(wt_key, (unsigned)p);
                        if (h >= 0) atomicOr(&wt_mask[h], 1u << ss[u]);
                        if (cl_n < SK_CL_KEEP) cl_list[cl_n][tid] = (unsigned)p | ((unsigned)ss[u] << 28);
                        else cl_bits |= 1u << (k0 + u);
                        cl_n++;
                    }
                }
            }
            __syncthreads();  // all marks are in the table
            SK_TICK(3);
            // 4. what did the earlier slots touch?  (walk + the vertex the parent id was read from; tip of every entry)
            if (wave < nc) {
                const SkSelSlot& S = L.slot[wave];
                const int len = sl_len[wave], termv = sl_term[wave];
                unsigned mk = 0u;
                if (lane <= len) mk = wt_mask[sk_wt_find(wt_key, (unsigned)(lane < len ? S.path[len - 1 - lane] : (termv < 0 ? n - 1 : termv)))];
                const unsigned walkm = wave_or_u(mk);
                if (lane == 0) sl_walkm[wave] = walkm;
            }
            unsigned etip = 0u;
            if (wave == 0 && lane < ne) etip = wt_mask[sk_wt_find(wt_key, (unsigned)ent_v)];
            __syncthreads();
            if (wave == 0) {  // the sequential replay over the entries, lane e holding entry e
                if (my_slot >= nc) my_slot = -1;
                const unsigned walkm = my_slot >= 0 ? sl_walkm[my_slot] : 0u;
                const int mylen = my_slot >= 0 ? sl_len[my_slot] : 0;
                unsigned alive = 0u;
                int nb2 = nb, tot2 = total, commits = 0, my_id = -2, my_off = 0;
                for (int e = 0; e < ne; e++) {
                    if ((unsigned)__builtin_amdgcn_readlane((int)etip, e) & alive) continue;  // never selected
                    const int sl = __builtin_amdgcn_readlane(my_slot, e);
                    if (sl < 0) break;                                                         // guessed wrong: it lives
                    if ((unsigned)__builtin_amdgcn_readlane((int)walkm, e) & alive) break;     // depends on an accepted slot
                    alive |= 1u << sl;
                    const int l = __builtin_amdgcn_readlane(mylen, e);
                    const bool keep = l >= 2;  // path.py:125-126: shorter paths still consume their points
                    if (lane == e) { my_id = keep ? nb2 : -1; my_off = tot2; }
                    if (keep) { nb2++; tot2 += l; }
                    commits++;
                }
                if (my_slot >= 0) { sl_id[my_slot] = my_id; sl_off[my_slot] = my_off; }
                if (lane == 0) {
                    s_alive = alive; s_nb2 = nb2; s_tot2 = tot2;
                    if (A.ticks) { A.ticks[8] += 1; A.ticks[12] += commits; A.ticks[13] += nc; A.ticks[11] += T; }
                }
            }
            __syncthreads();
            const unsigned alive = s_alive;
            nb = s_nb2; total = s_tot2;
            // 5. commit the accepted slots (path.py:112-136), wipe every mark
            if (wave < nc) {
                const SkSelSlot& S = L.slot[wave];
                const int len = sl_len[wave], id = sl_id[wave];
                if (lane < len) {
                    const int v = S.path[lane];
                    if (id != -2) {
                        if (id >= 0) A.path_verts[base + sl_off[wave] + lane] = v;  // (a dropped path shares its offset with the next one)
                        sk_mark(A, B, base, v, id);
                    }
                }
                if (lane == 0 && id >= 0) {
                    A.branch_parent[base + id] = sl_parent[wave];
                    A.branch_off[base + id] = sl_off[wave];
                    A.branch_len[base + id] = len;
                    if (A.ticks) A.ticks[10] += len;
                }
            }
            const int kept_n = cl_n < SK_CL_KEEP ? cl_n : SK_CL_KEEP;
            for (int j = 0; j < kept_n; j++) {
                const unsigned e = cl_list[j][tid];
                const int p = (int)(e & 0x0fffffffu), sidx = (int)(e >> 28);
                if ((alive >> sidx) & 1u) {
                    const int id = sl_id[sidx];
                    sk_mark(A, B, base, p, id);
                }
            }
            while (cl_bits) {  // the overflow: find the point again
                const int k = __ffs(cl_bits) - 1;
                cl_bits &= cl_bits - 1u;
                const int gi = k * W + tid;
                int sidx = 0, acc = 0;
#pragma unroll
                for (int k2 = 1; k2 < SK_WSLOTS; k2++)
                    if (gi >= pre[k2]) { sidx = k2; acc = pre[k2]; }
                if (!((alive >> sidx) & 1u)) continue;
                const SkSelSlot& S = L.slot[sidx];
                const uint32_t t = (uint32_t)(gi - acc);
                const int row = sk_find_row(S.row_off, sl_nrows[sidx], t);
                const int p = (int)__float_as_uint(recs[S.row_first[row] + (t - S.row_off[row])].w) - base;
                {
                    const int id = sl_id[sidx];
                    sk_mark(A, B, base, p, id);
                }
            }
            __syncthreads();  // (also: LDS of this round is dead)
            SK_TICK(4);
            continue;
        }
        // ---- `one` mode: candidate 0 needs the whole workgroup ----
        const int far = cand_v[0];
        __syncthreads();  // slot data is dead; its space is reused below
        int len = -1;
        for (unsigned chunk = 0; len < 0; chunk += blockDim.x) {
            const unsigned j = chunk + tid;
            const int node = sk_ancestor(A, base, far, j);
            const bool end = node < 0 || bm_test(B, node);
            if (!end) { if (j < SK_LPATH) L.one.lpath[j] = node; else tmp[j] = (unsigned)node; }
            unsigned long long k = end ? ((unsigned long long)(0xffffffffu - j) << 32) | (unsigned)(node + 1) : 0ull;
            k = block_max_u64(k, s_red);
            if (k != 0ull) {
                len = (int)(0xffffffffu - (unsigned)(k >> 32));
                if (tid == 0) s_term = (int)(unsigned)(k & 0xffffffffu) - 1;
            }
        }
        if (tid < 3) { s_lo[tid] = 0x7fffffff; s_hi[tid] = (int)0x80000000; }
        __syncthreads();
        // one pass over the path: output (root side first), radius maximum (path.py:31), and -- when it fits --
        // coordinates / radii / cell bounding box into LDS for the claim below.
        const bool keep = len >= 2;
        int parent = -1;
        if (tid == 0 && keep) parent = ld(&A.branch_of[base + (s_term < 0 ? n - 1 : s_term)]);
        int* path_out = A.path_verts + base + total;
        const bool fits = len <= SK_LPATH;
        unsigned long long rk = 0;
        for (int qi = tid; qi < len; qi += blockDim.x) {
            const int w = len - 1 - qi;  // walk order -> root side first
            const int v = w < SK_LPATH ? L.one.lpath[w] : (int)ld_wg(&tmp[w]);
            st_ai(&path_out[qi], v);  // (agent scope: the component's helper workgroups read the path)
            const float4 q4 = A.pr[base + v];
            const float r = q4.w;
            const unsigned long long k = (unsigned long long)st_f2ord(r) << 32;
            rk = k > rk ? k : rk;
            if (fits) {
                const float x = q4.x, y = q4.y, z = q4.z;
                L.one.lpx[qi] = x; L.one.lpy[qi] = y; L.one.lpz[qi] = z; L.one.lpr[qi] = r;
                const int cx = (int)floorf((x - g->lo[0]) / g->cell), cy = (int)floorf((y - g->lo[1]) / g->cell),
                          cz = (int)floorf((z - g->lo[2]) / g->cell);
                atomicMin(&s_lo[0], cx); atomicMin(&s_lo[1], cy); atomicMin(&s_lo[2], cz);
                atomicMax(&s_hi[0], cx); atomicMax(&s_hi[1], cy); atomicMax(&s_hi[2], cz);
            }
        }
        rk = block_max_u64(rk, s_red);  // (its barriers also publish lp*, s_lo / s_hi)
        const float rp = st_ord2f((unsigned)(rk >> 32)), rp2 = rp * rp;
        int reach = rp > 0.0f ? (int)ceilf(rp / g->cell) : 0;
        if (reach < 1) reach = 1;
        bool small = fits;
        int nrows_s = 0, ncand = 0;
        if (small) {
            const int x0 = st_max(s_lo[0] - reach, 0), x1 = st_min(s_hi[0] + reach, g->seg_dim0 - 1);
            const int y0 = st_max(s_lo[1] - reach, 0), y1 = st_min(s_hi[1] + reach, g->dim[1] - 1);
            const int z0 = st_max(s_lo[2] - reach, 0), z1 = st_min(s_hi[2] + reach, g->dim[2] - 1);
            const int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
            const int nrows = (nx > 0 && ny > 0 && z0 <= z1) ? nx * ny : 0;
            small = nrows <= (int)blockDim.x;  // one lane per (x, y) row of cells; z is contiguous in memory
            uint32_t cnt = 0, first = 0;
            if (small && tid < nrows) {
                const int64_t row = ((int64_t)(xoff + x0 + tid / ny) * g->dim[1] + (y0 + tid % ny)) * g->dim[2];
                first = A.cell_start[row + z0];
                cnt = A.cell_start[row + z1 + 1] - first;
            }
            uint32_t tot;
            const uint32_t off = block_exclusive_scan(cnt, s_scan, &tot);
            ncand = (int)tot;
            small = small && (int64_t)ncand * len <= A.small_work;
            if (small && tid < nrows) { L.one.row_off[tid] = off; L.one.row_first[tid] = first; }
            if (small && tid == 0) L.one.row_off[nrows] = tot;
            __syncthreads();
            nrows_s = nrows;
        }
        const int id = keep ? nb : -1;
        if (tid == 0 && keep) {
            A.branch_parent[base + nb] = parent;
            A.branch_off[base + nb] = total;
            A.branch_len[base + nb] = len;
        }
        const int cur_off = total;
        if (keep) { nb++; total += len; }
        // A long path (the trunk, a main limb: hundreds of vertices, tens of thousands of candidates) inside the workgroup, still
        // point-centric and exact: the path is cut into chunks of SK_CHUNK consecutive vertices with their bounding boxes; a
        // candidate walks the chunks in path order and skips one whose box is no closer than its best vertex so far (or than the
        // path radius).  The box distance is evaluated with the SAME float32 operations, in the same order, as the vertex
        // distance ((dx*dx + dy*dy) + dz*dz): rounding is monotone, so it never exceeds the distance to any vertex inside --
        // the nearest vertex (ties: the first on the path) is the one the full scan finds.  The rows of grid cells around the
        // path are taken W at a time.  No hand-over to k_sk_claim, no launch boundary: in a batch of clouds every tree keeps
        // going at its own pace instead of waiting for the slowest one at every long path.
        // (a path of more than a few vertices that the plain scan below could take goes this way, too, in chunks of eight: 2-3x fewer
        // distance evaluations)
        // (a SHORT path that is not small -- a radius outlier: a few vertices, a metre of reach, more candidates than the plain scan's
        // budget -- goes this way as well: handing it to k_sk_claim would end the launch for this tree, and in a batch the tree would
        // wait for every other tree of the launch; configs[3], two canopies side by side: 68 -> one launch)
        if (fits && (len > 16 ? (small || A.long_mode) : (!small && A.long_mode))) {
            const int csz = len <= 8 * (SK_LPATH / SK_CHUNK) ? 8 : SK_CHUNK;  // vertices per chunk: at most SK_LPATH / SK_CHUNK boxes
            sk_chunk_boxes(L.one, cb_lo, cb_hi, len, csz);
            const int x0 = st_max(s_lo[0] - reach, 0), x1 = st_min(s_hi[0] + reach, g->seg_dim0 - 1);
            const int y0 = st_max(s_lo[1] - reach, 0), y1 = st_min(s_hi[1] + reach, g->dim[1] - 1);
            const int z0 = st_max(s_lo[2] - reach, 0), z1 = st_min(s_hi[2] + reach, g->dim[2] - 1);
            const int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
            const int nrows = (nx > 0 && ny > 0 && z0 <= z1) ? nx * ny : 0;
            if (A.ticks && tid == 0) { A.ticks[9] += 1; A.ticks[10] += len; A.ticks[15] += 1; }
            // a heavy claim is shared with the component's helper workgroups (other compute units): each takes an equal share of
            // the cell rows; this workgroup takes the first, waits for the others and folds their termination bits into its bitmap
            bool help = nhelp > 0 && (nrows > 128 || len > 64);  // (a job costs ~5-10 us of hand-shake; such a claim > 50 us alone)
            int my_end = nrows;
            if (help) {
                if (tid == 0) s_lost = ld_au(&J->gone);
                sk_release_agent();  // EVERY wavefront: its agent-scope stores of the path above have been performed ...
                __syncthreads();     // ... before thread 0 publishes the job below
                if (s_lost) {  // (uniform) a helper's life time ran out: alone from here on
                    if (tid == 0) { st_au(&J->quit, 1u); st_au(&A.fcnt[0], 1u); }
                    help = false; nhelp = 0;
                }
            }
            if (help) {
                if (tid == 0) {
                    st_ai(&J->len, len); st_ai(&J->id, id); st_ai(&J->csz, csz); st_ai(&J->nrows, nrows); st_ai(&J->ny, ny);
                    st_ai(&J->x0, x0); st_ai(&J->y0, y0); st_ai(&J->z0, z0); st_ai(&J->z1, z1); st_ai(&J->path_off, cur_off);
                    st_ai((int*)&J->rp, (int)__float_as_uint(rp));
                    st_au(&J->done, 0u);
                    // release: the job words (and, through the barrier above, the path) are performed before `seq` can be seen
                    __hip_atomic_store(&J->seq, ++job_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    if (A.ticks) A.ticks[14] += 1;
                }
                my_end = (nrows + nhelp) / (nhelp + 1);
                if (my_end > nrows) my_end = nrows;
            }
            __syncthreads();  // chunk boxes published
            sk_long_rows<false>(A, B, L.one, cb_lo, cb_hi, s_scan, base, n, xoff, x0, y0, z0, z1, ny, 0, my_end, len, csz, rp2, id);
            for (int qi = tid; qi < len; qi += blockDim.x) sk_mark(A, B, base, L.one.lpath[len - 1 - qi], id);  // path.py:112-113,135
            if (help) {
                if (tid == 0) {
                    const long long t0 = wall_clock64();
                    unsigned lost = 0u;
                    while (ld_au(&J->done) < (unsigned)nhelp) {
                        __builtin_amdgcn_s_sleep(12);  // (the helpers write to this line: do not hammer it)
                        if (ld_au(&J->gone) != 0u || wall_clock64() - t0 > A.help_timeout) { lost = 1u; break; }
                    }
                    s_lost = lost;
                    if (lost) { st_au(&J->quit, 1u); st_au(&A.fcnt[0], 1u); }  // (fcnt[0]: a statistic -- helpers were lost in this call)
                }
                __syncthreads();
                sk_acquire_agent();  // pairs with the helpers' release of `done`: their termination bits and branch ids are visible
                if (s_lost) {  // (uniform) the answers did not come: claim the helpers' shares here -- whatever a late helper still
                    // stamps is what this workgroup stamps, too -- and go on without helpers
                    sk_long_rows<false>(A, B, L.one, cb_lo, cb_hi, s_scan, base, n, xoff, x0, y0, z0, z1, ny, my_end, nrows, len, csz, rp2, id);
                    nhelp = 0;
                    __syncthreads();
                }
                const unsigned* hw = A.help_bits + (base >> 5) + c;  // what the helpers terminated
                for (int i = tid; i < nwords; i += W) {
                    const unsigned w = ld_au(&hw[i]);
                    if (B.in_lds) bm_words[i] |= w; else if (w) wg_or(&B.glb[i], w);
                }
            }
            __syncthreads();
            SK_TICK(6);
            continue;
        }
        // too much work point-centric (every candidate against every path vertex)?  Path-centric then: every
        // (path vertex, cell row) pair offers itself to the points of its cells -- here if the path is of moderate
        // length, chip-wide (k_sk_claim) otherwise.
        const int side = 2 * reach + 1;
        const bool local = !small && (int64_t)len * side * side <= A.local_items;
        if (A.ticks && tid == 0) { A.ticks[9] += 1; A.ticks[10] += len; A.ticks[14] += (small || local) ? 0 : 1; A.ticks[15] += local ? 1 : 0; }
        if (local) {
            if (tid == 0) A.s_ntouched[c] = 0u;
            __syncthreads();  // (LDS path coordinates are dead: the staging list takes their place)
            unsigned* lq = (unsigned*)L.one.lpx;  // SK_LQ_CLAIM words = lpx + lpy
            unsigned* lq_ctl = (unsigned*)L.one.lpz;
            const bool in_lds = sk_claim_items(A, c, base, n, len, rp, path_out, false, 0, 1, lq, &lq_ctl[0], &lq_ctl[1], true);
            __syncthreads();
            sk_finish_branch(A, B, base, len, id, path_out, false, in_lds ? lq : A.touched + base, in_lds,
                             in_lds ? lq_ctl[0] : ld(&A.s_ntouched[c]));
            need_fill = true;  // the window flags are rebuilt from the allocation state
            SK_TICK(6);
            continue;
        }
        if (!small) {  // hand the path to k_sk_claim; its points are finished at the next launch (uniform)
            SK_QUIT_HELPERS();
            SK_FLUSH_BM();
            if (tid == 0) {
                A.s_len[c] = len; A.s_rp[c] = rp; A.s_ntouched[c] = 0u; A.s_cur_off[c] = cur_off; A.s_cur_id[c] = id;
                A.s_wide[c] = 1; A.s_cursor[c] = win_base; A.s_total[c] = total; A.s_nb[c] = nb;
            }
            SK_TICK_FLUSH();
            return;
        }
        for (int t = tid; t < ncand; t += blockDim.x) {
            const int row = sk_find_row(L.one.row_off, nrows_s, (uint32_t)t);
            const float4 r4 = recs[L.one.row_first[row] + ((uint32_t)t - L.one.row_off[row])];
            const int p = (int)__float_as_uint(r4.w) - base;
            if (p < 0 || p >= n) continue;  // other component
            float bd2 = __uint_as_float(0x7f800000u);
            int bq = 0;
            for (int qi = 0; qi < len; qi++) {  // LDS broadcast reads; ascending: ties keep the first path vertex
                const float dx = r4.x - L.one.lpx[qi], dy = r4.y - L.one.lpy[qi], dz = r4.z - L.one.lpz[qi];
                float d2 = dx * dx;
                float tt = dy * dy;
                d2 = d2 + tt;
                tt = dz * dz;
                d2 = d2 + tt;
                if (d2 < bd2) { bd2 = d2; bq = qi; }
            }
            if (bd2 < rp2 && sqrtf(bd2) < L.one.lpr[bq]) {  // path.py:35-40
                sk_mark(A, B, base, p, id);
            }
        }
        for (int qi = tid; qi < len; qi += blockDim.x) {  // path.py:112-113,135
            const int v = L.one.lpath[len - 1 - qi];
            sk_mark(A, B, base, v, id);
        }
        __syncthreads();
        SK_TICK(5);
    }
    SK_QUIT_HELPERS();
    SK_FLUSH_BM();
    if (tid == 0) { A.s_len[c] = 0; A.s_wide[c] = 0; A.s_cursor[c] = win_base; A.s_total[c] = total; A.s_nb[c] = nb; }
    SK_TICK_FLUSH();
#undef SK_FLUSH_BM
#undef SK_QUIT_HELPERS
}


__global__ void __launch_bounds__(SK_WIDE_BLOCK) k_sk_claim(SkArgs A) {
    __shared__ unsigned lq[SK_LQ_CLAIM];
    __shared__ unsigned lq_n, lq_base;
    const int c = A.blk_comp[blockIdx.x];
    if (c < 0 || A.s_done[c] || !A.s_wide[c]) return;  // c < 0: padding of the bound-sized grid
    const int base = A.comp_off[c], n = A.comp_off[c + 1] - base;
    sk_claim_items(A, c, base, n, A.s_len[c], A.s_rp[c], A.path_verts + base + A.s_cur_off[c], false,
                   blockIdx.x - A.blk_first[c], A.blk_count[c], lq, &lq_n, &lq_base, false);
}

#define SK_MAX_CLAIM_BLOCKS 256
// Claim grid laid out on the device (one workgroup): component c gets clamp(ceil(size / 1024), 1, SK_MAX_CLAIM_BLOCKS)
// consecutive workgroups from blk_first[c]; the host launches the bound n_comp + ceil(m / 1024) and the rest is padding
// (-1).  The component sizes therefore never travel to the host.
__global__ void __launch_bounds__(1024) k_sk_blk_tables(SkArgs A, int* blk_comp, int* blk_first, int* blk_count, int nblk_bound) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (A.C + 1023) / 1024, c0 = tid * per, c1 = min(A.C, c0 + per);
    int mine = 0;
    for (int c = c0; c < c1; c++) {
        const int k = (A.comp_off[c + 1] - A.comp_off[c] + 1023) / 1024;
        mine += k < 1 ? 1 : (k > SK_MAX_CLAIM_BLOCKS ? SK_MAX_CLAIM_BLOCKS : k);
    }
    part[tid] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan of the per-lane totals
        const int o = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += o;
        __syncthreads();
    }
    int at = part[tid] - mine;
    const int total = part[1023];
    for (int c = c0; c < c1; c++) {
        int k = (A.comp_off[c + 1] - A.comp_off[c] + 1023) / 1024;
        k = k < 1 ? 1 : (k > SK_MAX_CLAIM_BLOCKS ? SK_MAX_CLAIM_BLOCKS : k);
        blk_first[c] = at;
        blk_count[c] = k;
        for (int j = 0; j < k; j++) blk_comp[at++] = c;
    }
    for (int b = total + tid; b < nblk_bound; b += 1024) blk_comp[b] = -1;
}

// Helper workgroups of the long-path claims: a component of >= SK_HELP_MIN vertices gets one helper per SK_HELP_PER vertices
// (at most SK_HELP_MAX), in component order until the launch's n_helpers are used up.  One thread: a few thousand components.
__global__ void k_sk_helper_tables(SkArgs A, int* hl_comp, int* hl_rank, int* c_nhelp) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int at = 0;
    for (int c = 0; c < A.C; c++) {
        const int n = A.comp_off[c + 1] - A.comp_off[c];
        int k = n >= SK_HELP_MIN ? n / SK_HELP_PER : 0;
        k = k > SK_HELP_MAX ? SK_HELP_MAX : k;
        if (k > A.n_helpers - at) k = A.n_helpers - at;
        c_nhelp[c] = k;
        for (int j = 0; j < k; j++) { hl_comp[at] = c; hl_rank[at] = j + 1; at++; }
    }
    for (; at < A.n_helpers; at++) { hl_comp[at] = -1; hl_rank[at] = 0; }
}

// ------------------------------------------------------------------------------- host side ---
#define SK_GRID_CELLS (1ll << 24)

struct SkLayout {
    unsigned *dist_ord, *stamp, *q0, *q1, *touched, *cnt, *s_ntouched, *sort_keys, *order, *coop_bar;
    float *s_rp, *order_init;
    unsigned* term_bits;
    float4* pr;
    int *s_cursor, *s_wide;
    char* sort_ws;
    int64_t sort_bytes;
    unsigned long long* best;
    int *anc, *comp_of, *s_done, *s_len, *s_cur_id, *s_cur_off, *s_nb, *s_total, *blk_comp, *blk_first, *blk_count, *hl_comp, *hl_rank, *c_nhelp;
    SkJob* jobs;
    StGrid* g;
    uint32_t* cell_start;
    float4* recs;
    char* gws;
    int64_t gws_bytes;
};

static inline int64_t sk_grid_cells(int nseg, int64_t m) {  // (st_grid_build uses at most 128 cells per point: size for that)
    return st_min64(SK_GRID_CELLS * (nseg < 1 ? 1 : (nseg > 32 ? 32 : nseg)), 128 * (m > 0 ? m : 1) + 65536);
}

// frontier queue geometry, in one place: SK_FS shard segments of sk_fseg entries + an overflow area with room for every entry
static inline int64_t sk_fseg(int64_t m, int64_t C) { return (m + C) / SK_FS > 0 ? (m + C) / SK_FS : 1; }
static inline int64_t sk_queue_words(int64_t m, int64_t C) { return SK_FS * sk_fseg(m, C) + (m + C); }
// termination bits: component c owns the words from (comp_off[c] >> 5) + c, ceil(size / 32) of them
static inline int64_t sk_term_words(int64_t m, int64_t C) { return m / 32 + C + 2; }

static void sk_layout(StArena& a, int64_t m, int64_t C, SkLayout* s, int nseg = 1) {
    s->dist_ord = a.take<unsigned>(m);
    s->stamp = a.take<unsigned>(m);
    s->q0 = a.take<unsigned>(sk_queue_words(m, C));  // SSSP frontier: SK_FS shard segments + the overflow area (sk_q_reserve)
    s->q1 = a.take<unsigned>(sk_queue_words(m, C));
    s->touched = a.take<unsigned>(m);
    s->term_bits = a.take<unsigned>(2 * sk_term_words(m, C));  // the termination set, then the helpers' words
    s->pr = a.take<float4>(m);
    s->best = a.take<unsigned long long>(m);
    s->anc = a.take<int>((int64_t)SK_ANC * m);
    s->comp_of = a.take<int>(m);
    s->cnt = a.take<unsigned>(8 + 3 * (SK_FS + 1));  // [8] counters, then the frontier counters of three generations
    s->coop_bar = a.take<unsigned>(SK_COOP_STRIDE * (2 + SK_COOP_BLOCKS / SK_COOP_GROUP));  // grid barrier of the persistent SSSP
    s->s_done = a.take<int>(C);
    s->s_len = a.take<int>(C);
    s->s_cur_id = a.take<int>(C);
    s->s_cur_off = a.take<int>(C);
    s->s_nb = a.take<int>(C);
    s->s_total = a.take<int>(C);
    s->s_rp = a.take<float>(C);
    s->s_ntouched = a.take<unsigned>(C);
    s->s_cursor = a.take<int>(C);
    s->s_wide = a.take<int>(C);
    s->sort_keys = a.take<unsigned>(m);
    s->order = a.take<unsigned>(m);
    s->order_init = a.take<float>(m);
    s->sort_bytes = st_sort_ws_bytes(m);
    s->sort_ws = a.take<char>(s->sort_bytes);
    s->blk_first = a.take<int>(C);
    s->blk_count = a.take<int>(C);
    s->blk_comp = a.take<int>(C + st_div_up(m, 1024));
    s->hl_comp = a.take<int>(SK_HELP_POOL);
    s->hl_rank = a.take<int>(SK_HELP_POOL);
    s->c_nhelp = a.take<int>(C);
    s->jobs = a.take<SkJob>(C);
    s->g = a.take<StGrid>(1);
    s->cell_start = a.take<uint32_t>(sk_grid_cells(nseg, m) + 1);
    s->recs = a.take<float4>(m);
    s->gws_bytes = st_grid_ws_bytes(m, sk_grid_cells(nseg, m));
    s->gws = a.take<char>(s->gws_bytes);
}

// Tuning of one call (st_skeleton_components_seg's `tuning` argument: 16 x int64, entry = ST_TUNE_DEFAULT or NULL array =
// the default).  Test hook (tests/test_skeleton.py forces every claim strategy and every SSSP form) and sweep aid
// (tools/); nothing is process-global, so concurrent calls cannot see each other's settings.
//   0 prune factor (x1000)   1 small_work   2 rounds per select launch   3 select launches per host read-back
//   4 local_items   5 wave_work   6 SSSP levels per frontier launch   7 frontier launches per read-back   8 lanes per vertex
//   9 length of the first frontier batch (in batches)   10 frontier workgroups   11 claim-grid cell cap (hundredths of the
//   mean radius)   12 SSSP rounds in one persistent launch with grid barriers (1) or one launch per round (0, default: measured equal for
//   one cloud, 1-3 % slower per step for a launch set of 20, 2.4 % with two batches in flight -- the rounds are bound by their ~6 us
//   per level, not by the launches);
//   bits 8 ..: 1 + the number of helper workgroups of the branch selection's long-path claims (0 = by size)
//   13 frontier: 0 = no look before the atomic, else vertices a workgroup relaxes per local level
//   14 long-path claim inside the workgroup (1, default) or by the local / chip-wide path-centric claims (0)
//   15 device pointer of 32 int64 phase timers / counters of k_sk_select
//   16 / 17 / 18 time-outs of the helper protocol in microseconds: a helper's life time, a component's wait for one job's answers,
//   a helper's wait for its component's workgroup to announce itself (tests/test_helpers.py drives every fall-back with them)
#define ST_TUNE_DEFAULT INT64_MIN
#define ST_TUNE_ENTRIES 24
#define SK_MAX_LAUNCH_BATCH 32
struct SkTuning {
    float prune_factor = 1.0f, grid_mean_mult = 1.0f;
    int small_work = SK_SMALL_WORK, iters_per_launch = SK_ITERS_PER_LAUNCH, launch_batch = 24, local_items = 0, wave_work = SK_WAVE_WORK, long_mode = 1;
    int sssp_hops = 6 /* round 6: 6 levels per launch instead of 4 -- the same on the network's 360-level graph (skeleton kernels 0.548 against 0.537 ms
    per cloud in a set of twenty, 657-660 against 655-662 M points/s; 8 levels: 0.562 ms, 645-655 M), 8 % less on the 2400-level graph of exact medial
    vectors (a set of twenty 36.2 -> 33.3 ms, one cloud 16.5 -> 15.5; 8 levels: 32.9 / 15.4): profiles/r06_ab_hops2.txt, r06_sweep_gt_sssp.txt */, sssp_batch = 36 /* launches per read-back; the first batch is sssp_first x this = 72 launches = 432 levels: the 1M-point trees' graphs are
    260-384 levels deep, so one batch ends them (with 32 a set of twenty needed a second batch of 32 mostly empty launches) */, sssp_lanes = 32, sssp_first = 2, sssp_blocks = SK_SSSP_BLOCKS, sssp_lcap = SK_LQ;
    int sssp_coop = 0, helpers = -1;  // helpers: -1 = by size, else the number of helper workgroups of a select launch
    long long help_lifetime_us = SK_HELP_LIFETIME_US, help_timeout_us = SK_HELP_TIMEOUT_US, help_announce_us = SK_HELP_ANNOUNCE_US;
    bool small_work_set = false, iters_set = false, long_set = false, launch_set = false;
    long long* ticks = nullptr;
    explicit SkTuning(const int64_t* t) {
        if (!t) return;
        auto has = [&](int i) { return t[i] != ST_TUNE_DEFAULT; };
        if (has(0)) prune_factor = (float)t[0] / 1000.0f;
        if (has(1)) { small_work = (int)t[1]; small_work_set = true; }
        if (has(2)) { iters_per_launch = (int)t[2]; iters_set = true; }
        if (has(3)) { launch_batch = t[3] < 1 ? 1 : (t[3] > SK_MAX_LAUNCH_BATCH ? SK_MAX_LAUNCH_BATCH : (int)t[3]); launch_set = true; }
        if (has(4)) local_items = (int)t[4];
        if (has(5)) wave_work = (int)t[5];
        if (has(6)) sssp_hops = t[6] < 1 ? 1 : (int)t[6];
        if (has(7)) sssp_batch = t[7] < 1 ? 1 : (t[7] > 64 ? 64 : (int)t[7]);
        if (has(8)) sssp_lanes = t[8] == 64 ? 64 : (t[8] == 32 ? 32 : 16);
        if (has(9)) sssp_first = t[9] < 1 ? 1 : (t[9] > 8 ? 8 : (int)t[9]);
        if (has(10)) sssp_blocks = t[10] < 1 ? 1 : (t[10] > 8192 ? 8192 : (int)t[10]);
        if (has(11)) grid_mean_mult = (float)t[11] / 100.0f;
        if (has(12)) { sssp_coop = (t[12] & 1) != 0; if (t[12] >= 256) helpers = (int)(t[12] >> 8) - 1; }
        if (has(13)) sssp_lcap = t[13] == 0 ? -SK_LQ : (t[13] > SK_LQ ? SK_LQ : (int)t[13]);
        if (has(14)) { long_mode = t[14] != 0; long_set = true; }
        if (has(15)) ticks = (long long*)(intptr_t)t[15];
        if (has(16)) help_lifetime_us = t[16] < 0 ? 0 : t[16];
        if (has(17)) help_timeout_us = t[17] < 0 ? 0 : t[17];
        if (has(18)) help_announce_us = t[18] < 0 ? 0 : t[18];
    }
};

// Helper workgroups spin on a compute unit each (k_sk_select holds 142 KB of LDS: one workgroup per compute unit), so their number
// is bounded by the DEVICE -- at most 5/8 of its compute units (160 of an MI355X's 256; a partition or a smaller part gets fewer, a
// device of one compute unit none) -- and by what other calls of this process have out at the moment: two helper-enabled calls on
// different streams share the budget instead of filling the chip with waiting workgroups between them.
static std::atomic<int> g_helpers_out{0};
static int sk_cu_count() {  // compute units of the calling thread's current device (0 if it cannot be asked)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return cus;
}
static int sk_helper_cap() {
    const int cap = sk_cu_count() * 5 / 8;
    return cap < SK_HELP_POOL ? cap : SK_HELP_POOL;
}
struct SkHelperLease {  // returned on every way out of the call
    int n = 0;
    int take(int want) {
        const int cap = sk_helper_cap();
        int cur = g_helpers_out.load();
        do { n = want < cap - cur ? want : cap - cur; if (n <= 0) { n = 0; return 0; } } while (!g_helpers_out.compare_exchange_weak(cur, cur + n));
        return n;
    }
    ~SkHelperLease() { if (n > 0) g_helpers_out.fetch_sub(n); }
};

// lengths of the caller-provided int64 arrays this build reads / writes (include/smarttree_hip.h: st_abi_entries)
static_assert(ST_TUNE_ENTRIES == ST_SKELETON_TUNING_ENTRIES, "header and library disagree about the tuning array");
extern "C" int st_abi_entries(int what) {
    return what == 0 ? ST_SKELETON_STATS_ENTRIES : (what == 1 ? ST_SKELETON_TUNING_ENTRIES : (what == 2 ? ST_MAX_SEG : -1));
}

extern "C" int64_t st_skeleton_workspace_bytes_seg(int64_t m, int64_t n_comp, int nseg) {
    StArena a(nullptr, 0);
    SkLayout s;
    sk_layout(a, m > 0 ? m : 1, n_comp > 0 ? n_comp : 1, &s, nseg);
    return a.used;
}
extern "C" int64_t st_skeleton_workspace_bytes(int64_t m, int64_t n_comp) { return st_skeleton_workspace_bytes_seg(m, n_comp, 1); }

static inline unsigned sk_vgrid(int64_t m) {
    int64_t g = st_div_up(m > 0 ? m : 1, SK_WIDE_BLOCK);
    return (unsigned)(g < 2048 ? g : 2048);
}

static int sk_read(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
    st_stream_wait(stream);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// All components of one cloud.  Vertex arrays are in the renumbered space of st_component_layout;
// comp_size_host: unused (may be NULL) -- the claim grid is laid out on the device from comp_off.
// grid_cell > 0: cell size of the claim grid; < 0: max(rad) / -grid_cell, reduced on the device.
// stages: 1 = roots + SSSP + predecessors, 2 = literal second SSSP into tree_dist, 4 = sample_tree
// (on tree_dist if stage 2 ran, else on dist -- the two are bit-identical).
// block_threads: lanes of the per-component select workgroup (0 = 1024).
// stats_host (optional, 16 x int64): [0] SSSP rounds, [1] plateau rounds, [2] select/claim launch pairs, [3] lifting
// levels; if stats_host[7] != 0 on entry, every k_sk_select launch is bracketed by HIP events on `stream` and
// [4] = their summed duration in ns, [5] = number of launches (profiling aid for bench.py's roofline block);
// [6] = branches of the cloud | path vertices << 32 (sizes st_assemble_branches' outputs without a read-back of its own);
// [8] = 1 if helper workgroups were lost in this call (time-outs: their component's workgroup did the work), [9] = helper workgroups launched,

//
// Batched form (st_skeleton_components_seg): the components of `nseg` independent clouds in one call.  comp_seg [C] = cloud
// of each component, vert_seg_off [nseg + 1] = the clouds' ranges in the renumbered vertex space (both device arrays,
// from st_component_layout_seg).  Components never interact, so the only thing the batch shares is the claim grid, where
// every cloud has its own slab of cells: each component's outputs are those of the one-cloud call.
extern "C" int st_skeleton_components_seg(int n_comp, const int32_t* comp_off, const int32_t* comp_seg,
                                          const int32_t* vert_seg_off, int nseg, int64_t m,
                                      const float* pts, const float* rad, const float* ysurf, const uint32_t* row_off,
                                      const uint32_t* col, const float* wgt, float grid_cell, int stages, int block_threads,
                                      float* dist, int32_t* pred, int32_t* root_local, float* tree_dist,
                                      int32_t* branch_parent, int32_t* branch_off, int32_t* branch_len, int32_t* n_branches,
                                      int32_t* path_verts, int32_t* branch_of, int64_t* stats_host, void* ws,
                                      int64_t ws_bytes, void* stream_, const int64_t* tuning) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool time_select = stats_host && stats_host[7] != 0;
    if (stats_host) for (int i = 0; i < ST_SKELETON_STATS_ENTRIES; i++) if (i != 7) stats_host[i] = 0;
    if (n_comp <= 0 || m <= 0) return ST_OK;
    ST_REQUIRE(!(stages & 2) || tree_dist != nullptr, "skeleton: stage 2 needs a tree_dist buffer");
    if (block_threads <= 0) block_threads = 1024;
    ST_REQUIRE(block_threads % 64 == 0 && block_threads <= 1024, "skeleton: block_threads must be a multiple of 64, <= 1024");
    ST_REQUIRE(m < (1ll << 28), "skeleton: at most 2^28 graph vertices");
    ST_REQUIRE(nseg >= 1 && nseg <= ST_MAX_SEG, "skeleton: 1 <= clouds per batch <= %d", ST_MAX_SEG);
    ST_REQUIRE(nseg == 1 || (comp_seg && vert_seg_off), "skeleton: a batch needs comp_seg and vert_seg_off");
    if (nseg == 1) { comp_seg = nullptr; vert_seg_off = nullptr; }
    StArena a(ws, ws_bytes);
    SkLayout s;
    sk_layout(a, m, n_comp, &s, nseg);
    if (!a.ok() || !s.gws) {
        st_set_error("skeleton: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    SkArgs A;
    memset(&A, 0, sizeof(A));
    A.C = n_comp; A.m = m; A.comp_off = comp_off; A.comp_seg = comp_seg; A.comp_of = s.comp_of; A.pts = pts; A.rad = rad; A.ysurf = ysurf;
    A.row_off = row_off; A.col = col; A.wgt = wgt; A.grid = s.g; A.cell_start = s.cell_start; A.recs = s.recs;
    A.dist = dist; A.pred = pred; A.root_local = root_local; A.tree_dist = tree_dist;
    A.branch_parent = branch_parent; A.branch_off = branch_off; A.branch_len = branch_len; A.n_branches = n_branches;
    A.path_verts = path_verts; A.branch_of = branch_of;
    A.dist_ord = s.dist_ord; A.stamp = s.stamp; A.q0 = s.q0; A.q1 = s.q1; A.term_bits = s.term_bits; A.help_bits = s.term_bits + sk_term_words(m, n_comp); A.pr = s.pr;
    A.best = s.best; A.touched = s.touched; A.anc = s.anc; A.cnt = s.cnt; A.fcnt = s.cnt + 8;
    A.fseg = (unsigned)sk_fseg(m, n_comp);
    A.s_done = s.s_done; A.s_len = s.s_len; A.s_cur_id = s.s_cur_id; A.s_cur_off = s.s_cur_off; A.s_nb = s.s_nb;
    A.s_total = s.s_total; A.s_rp = s.s_rp; A.s_ntouched = s.s_ntouched;
    A.blk_comp = s.blk_comp; A.blk_first = s.blk_first; A.blk_count = s.blk_count;
    A.s_cursor = s.s_cursor; A.s_wide = s.s_wide; A.order = s.order; A.order_init = s.order_init;
    A.hl_comp = s.hl_comp; A.hl_rank = s.hl_rank; A.c_nhelp = s.c_nhelp; A.jobs = s.jobs; A.n_helpers = 0;
    const SkTuning T(tuning);
    A.ticks = T.ticks;
    A.help_lifetime = T.help_lifetime_us * 100; A.help_timeout = T.help_timeout_us * 100; A.help_announce = T.help_announce_us * 100;
    A.prune_factor = T.prune_factor; A.small_work = T.small_work; A.iters_per_launch = T.iters_per_launch; A.local_items = T.local_items; A.wave_work = T.wave_work; A.long_mode = T.long_mode;
    // A batch of clouds advances in lockstep: a launch lasts as long as its slowest component, and a component that hands a
    // long path to the chip-wide claim kernel waits for everybody else's rounds.  Fewer hand-overs (the workgroup claims
    // paths up to 16x larger by itself) and shorter launches measured 2.98 -> 2.48 ms of skeleton stage per cloud at 8 clouds
    // per batch (tools/sweep_select.sh, profiles/r02_sweep_select.txt); one cloud alone keeps the round-1 optimum.
    // Round 3: in a batch the workgroup also claims its long paths itself (chunk-pruned, k_sk_select "long path") and a launch
    // runs until the tree is done -- no tree waits for another one at a hand-over (profiles/r03_sweep_select.txt: 1.204 -> 1.176 ms
    // per cloud at 64 clouds per launch set, 2.00 -> 1.95 at 10).  One cloud alone keeps the chip-wide claim for its long paths
    // (the chip is idle then: 11.2 against 12.0 ms).
    int first_launches = T.launch_batch;
    // Helper workgroups spin while they wait: fine when the call has the chip to itself (one cloud -4 %, a launch set of 20 clouds
    // -3 to -6 % per cloud), a loss when another batch's chip-filling kernels want the compute units (two 64-cloud batches in
    // flight: +3.6 % per cloud; profiles/r04_sweep_code12.txt).  By default a call of up to SK_HELP_SEGS clouds uses them; tuning
    // code 12 overrides.
    // (the CPU emulator reports one compute unit and runs workgroups one after the other: sk_helper_cap() == 0, no helpers there)
    SkHelperLease lease;
    int helpers_want = 0;
    if ((stages & 4) && T.helpers != 0 && block_threads >= 256 && m >= SK_HELP_MIN && (T.helpers > 0 || nseg <= SK_HELP_SEGS))
        helpers_want = T.helpers > 0 ? T.helpers : (int)st_min64(SK_HELP_POOL, m / SK_HELP_PER);
    const bool helpers_avail = lease.take(helpers_want) > 0;
    // Round 4: with helper workgroups for the long-path claims, one cloud alone runs like a batch as well -- one launch to the
    // end, long paths claimed inside the launch (6.3 ms against 6.6 with the chip-wide claim at the launch boundaries).
    if (nseg > 1 || helpers_avail) {
        if (!T.small_work_set) A.small_work = 1 << 20;
        if (!T.iters_set) A.iters_per_launch = 1 << 20;
        // ... so ONE launch pair normally ends every tree (a path too long for the workgroup's LDS still goes to k_sk_claim and
        // costs another pair): the 23 further pairs of the first batch were empty launches, ~19 us each = 0.45 ms at the end of
        // every launch set (kernel trace of a 24-cloud batch, round 3)
        if (!T.launch_set && A.long_mode && A.iters_per_launch >= (1 << 20)) first_launches = 1;
    } else if (!T.long_set) {
        A.long_mode = 0;
    }

    const unsigned vg = sk_vgrid(m);
    const unsigned fg = (unsigned)st_min64(st_div_up(m, SK_WIDE_BLOCK), T.sssp_blocks);
    unsigned h[8 + 3 * (SK_FS + 1)];
    int64_t sssp_rounds = 0;
    const bool defer_plateaus = (stages & 1) && (stages & 4) && !(stages & 2);
    auto resolve_plateaus = [&](unsigned unresolved) -> int {
        unsigned round = 2;
        while (unresolved > 0) {
            (void)hipMemsetAsync(&s.cnt[4], 0, sizeof(unsigned), stream);
            hipLaunchKernelGGL(k_sk_preds_plateau, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A, round);
            ST_TRY(sk_read(h, s.cnt, sizeof(unsigned) * 8, stream));
            if (h[4] == 0) break;  // unreachable leftovers (cannot happen inside a component)
            unresolved -= h[4];
            round++;
        }
        if (stats_host) stats_host[1] = round - 2;
        return ST_OK;
    };
    // Roots, SSSP, canonical predecessors (not the plateau rounds).
    auto run_sssp = [&]() -> int {
        hipLaunchKernelGGL(k_sk_init, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A);
        hipLaunchKernelGGL(k_sk_roots, dim3((unsigned)n_comp), dim3((unsigned)block_threads), 0, stream, A);
        bool coop_done = false;

        if (T.sssp_coop) {
            // every round in ONE launch (k_sk_sssp_coop); one read-back tells whether a workgroup gave up at a barrier
            // the grid must be resident as a whole: at most four workgroups of 256 lanes per compute unit (1024 on an MI355X; a device
            // of ONE compute unit -- the CPU emulator, which runs the workgroups of a launch one after the other -- gets one workgroup)
            const int cus = sk_cu_count();
            const unsigned resident = cus > 1 ? (unsigned)st_min64(4ll * cus, SK_COOP_BLOCKS) : 1u;
            const unsigned cg = fg < resident ? fg : resident;
            (void)hipMemsetAsync(s.coop_bar, 0, SK_COOP_STRIDE * (2 + SK_COOP_BLOCKS / SK_COOP_GROUP) * sizeof(unsigned), stream);
            hipLaunchKernelGGL(k_sk_sssp_coop, dim3(cg), dim3(SK_WIDE_BLOCK), 0, stream, A, T.sssp_hops, T.sssp_lanes, T.sssp_lcap, s.coop_bar, 1 << 24);
            unsigned hb[4];
            ST_TRY(sk_read(hb, s.coop_bar, sizeof(hb), stream));
            if (hb[1] == 0u) { coop_done = true; sssp_rounds = hb[2]; }
            else {  // start over, one launch per round
                hipLaunchKernelGGL(k_sk_init, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A);
                hipLaunchKernelGGL(k_sk_roots, dim3((unsigned)n_comp), dim3((unsigned)block_threads), 0, stream, A);
            }
        }
        for (int r = 0; !coop_done;) {  // frontier rounds in batches, one counter read-back per batch.  The first batch is twice as long:
            // a tree of a million points needs 65-96 launches, an empty round costs ~4 us, a read-back beside other clouds ~1 ms
            const int batch = r == 0 ? T.sssp_first * T.sssp_batch : T.sssp_batch;
            for (int b = 0; b < batch; b++, r++)
                hipLaunchKernelGGL(k_sk_sssp_round, dim3(fg), dim3(SK_WIDE_BLOCK), 0, stream, A, r, T.sssp_hops, T.sssp_lanes, T.sssp_lcap);
            ST_TRY(sk_read(h, s.cnt, sizeof(h), stream));
            sssp_rounds = r;
            unsigned left = 0;  // entries of the frontier the next launch would read
            for (int i = 0; i <= SK_FS; i++) left |= h[8 + (r % 3) * (SK_FS + 1) + i];
            if (left == 0) break;
            ST_REQUIRE(r < (1 << 24), "skeleton: SSSP did not converge");
        }
        hipLaunchKernelGGL(k_sk_dist_out, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A);
        hipLaunchKernelGGL(k_sk_preds, dim3((unsigned)st_min64(st_div_up(m * SK_PRED_LANES, SK_WIDE_BLOCK), 8192)), dim3(SK_WIDE_BLOCK), 0, stream, A);
        if (stats_host) stats_host[0] = sssp_rounds;
        return ST_OK;
    };
    if (stages & 1) {
        ST_TRY(run_sssp());
        // Vertices whose tight in-neighbours all sit on their own distance plateau (cnt[3]) are rare; with sample_tree
        // next, their count is not read back here (a blocking round trip costs ~1 ms beside other clouds' kernels,
        // DESIGN.md section 5) but arrives with the first progress read-back of the select loop, which is then redone.
        if (!defer_plateaus) {
            ST_TRY(sk_read(h, s.cnt, sizeof(unsigned) * 8, stream));
            ST_TRY(resolve_plateaus(h[3]));
        }
    }
    if (!(stages & 1)) hipLaunchKernelGGL(k_sk_fill_comp_of, dim3((unsigned)n_comp), dim3((unsigned)block_threads), 0, stream, A);
    if (stages & 2) {
        hipLaunchKernelGGL(k_sk_td_init, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A);
        hipLaunchKernelGGL(k_sk_td_roots, dim3(1), dim3(SK_WIDE_BLOCK), 0, stream, A);
        for (int r = 0;;) {
            for (int b = 0; b < 32; b++, r++)
                hipLaunchKernelGGL(k_sk_td_round, dim3(fg), dim3(SK_WIDE_BLOCK), 0, stream, A, r);
            ST_TRY(sk_read(h, s.cnt, sizeof(unsigned) * 8, stream));
            if (h[r % 3] == 0) break;
            ST_REQUIRE(r < (1 << 24), "skeleton: tree distance did not converge");
        }
    }
    if (stages & 4) {
        // claim / finalize grid: workgroups per component proportional to its size, laid out by k_sk_blk_tables
        const int nblk = (int)st_min64((int64_t)n_comp + st_div_up(m, 1024), (int64_t)n_comp * SK_MAX_CLAIM_BLOCKS);
        hipLaunchKernelGGL(k_sk_blk_tables, dim3(1), dim3(1024), 0, stream, A, s.blk_comp, s.blk_first, s.blk_count, nblk);
        // helper workgroups of the long-path claims (they sit in front of the components' workgroups in the grid, so they are
        // normally resident before any component can wait for them; if not, the time-outs above apply)
        A.n_helpers = lease.n;
        hipLaunchKernelGGL(k_sk_helper_tables, dim3(1), dim3(64), 0, stream, A, s.hl_comp, s.hl_rank, s.c_nhelp);
        // grid_cell < 0: cell = max(rad) / -grid_cell with the maximum reduced on the device (no host round trip)
        ST_TRY(st_grid_build(pts, m, grid_cell, sk_grid_cells(nseg, m), s.g, s.cell_start, s.recs, s.gws, s.gws_bytes, stream,
                             grid_cell < 0.0f ? -1.0f : 0.0f, grid_cell < 0.0f ? rad : nullptr, grid_cell < 0.0f ? m : 0,
                             vert_seg_off, nseg, vert_seg_off, T.grid_mean_mult));
        int64_t iters = 0;
        struct EventSet {  // destroyed on every way out of the select loop (early error returns included)
            hipEvent_t e[2 * SK_MAX_LAUNCH_BATCH];
            int n = 0;
            bool create() { for (; n < 2 * SK_MAX_LAUNCH_BATCH; n++) if (hipEventCreate(&e[n]) != hipSuccess) return false; return true; }
            ~EventSet() { for (int i = 0; i < n; i++) (void)hipEventDestroy(e[i]); }
        } evs;
        hipEvent_t* ev = evs.e;
        bool timing_ok = true;  // a failed event call must not put garbage into the roofline numbers
        if (time_select && !evs.create()) { (void)hipGetLastError(); timing_ok = false; }
        const bool time_sel = time_select && timing_ok;
        double select_ms = 0.0;
        bool plateaus_pending = defer_plateaus;
        for (;;) {  // second pass only if the deferred check found plateau vertices: predecessors completed, selection redone
            (void)hipMemsetAsync(&s.cnt[5], 0, 4 * sizeof(unsigned), stream);  // finished components, branches, path vertices; fcnt[0] = helper time-out flag
            const float* distances = (stages & 2) ? tree_dist : dist;
            (void)hipMemsetAsync(s.term_bits, 0, 2 * sk_term_words(m, n_comp) * sizeof(unsigned), stream);
            hipLaunchKernelGGL(k_sk_lift_init, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A);
            for (int span = 1; span < SK_ANC; span *= 2)  // direct ancestor table by doubling
                hipLaunchKernelGGL(k_sk_anc_pass, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A, span);
            // order the vertices of every component by distance, once: the per-branch argmax becomes a cursor
            hipLaunchKernelGGL(k_sk_sort_keys, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A, distances, s.sort_keys, s.order, 0);
            ST_TRY(st_radix_sort_pairs_u32(s.sort_keys, s.order, m, 32, s.sort_ws, s.sort_bytes, stream));
            if (n_comp > 1) {
                int bits = 1;
                while ((1ll << bits) < n_comp) bits++;
                hipLaunchKernelGGL(k_sk_sort_keys, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A, distances, s.sort_keys, s.order, 1);
                ST_TRY(st_radix_sort_pairs_u32(s.sort_keys, s.order, m, bits, s.sort_ws, s.sort_bytes, stream));
            }
            hipLaunchKernelGGL(k_sk_order_init, dim3(vg), dim3(SK_WIDE_BLOCK), 0, stream, A, distances, s.order_init);
            iters = 0;
            select_ms = 0.0;
            bool redo = false;
            // the 1M-point synthetic trees need 10-21 launch pairs (tools/round_counts.py): a first batch of 24 ends them with ONE
            // progress read-back; A/B on one box, 8 clouds in flight: 5.08 ms per cloud against 5.66 with 16 (tools/sweep_batches.sh)
            for (int batch = first_launches;; batch = batch > 4 ? 4 : batch) {  // launch pairs per counter read-back: a long first
                // batch, short ones for the stragglers (a finished launch pair still costs its ~10 us of launch latency)
                for (int b = 0; b < batch; b++, iters++) {
                    if (A.n_helpers > 0) (void)hipMemsetAsync(s.jobs, 0, (size_t)n_comp * sizeof(SkJob), stream);
                    if (time_sel) (void)hipEventRecord(ev[2 * b], stream);
                    hipLaunchKernelGGL(k_sk_select, dim3((unsigned)(A.n_helpers + n_comp)), dim3((unsigned)block_threads), 0, stream, A);
                    if (time_sel) (void)hipEventRecord(ev[2 * b + 1], stream);
                    hipLaunchKernelGGL(k_sk_claim, dim3((unsigned)nblk), dim3(SK_WIDE_BLOCK), 0, stream, A);
                }
                ST_TRY(sk_read(h, s.cnt, sizeof(unsigned) * 9, stream));  // (h[8] = fcnt[0]: helper workgroups were lost -- their
                // component's workgroup did the work itself: slower, not wrong)
                if (time_sel)
                    for (int b = 0; b < batch; b++) {
                        float ms = 0.0f;
                        if (hipEventElapsedTime(&ms, ev[2 * b], ev[2 * b + 1]) == hipSuccess) select_ms += ms;
                        else { (void)hipGetLastError(); timing_ok = false; }
                    }
                if (plateaus_pending) {  // first read-back after the deferred checks of the SSSP stage
                    plateaus_pending = false;
                    if (h[3] > 0) { redo = true; break; }
                }
                if (h[5] >= (unsigned)n_comp) break;
                ST_REQUIRE(iters <= m + 64 && iters < (1 << 26), "skeleton: sample_tree did not terminate");
            }
            if (!redo) break;
            // plateau vertices left unresolved: redo the predecessor pass with its resolved / unresolved marks, then the plateaus
            (void)hipMemsetAsync(&s.cnt[3], 0, sizeof(unsigned), stream);
            hipLaunchKernelGGL(k_sk_preds, dim3((unsigned)st_min64(st_div_up(m * SK_PRED_LANES, SK_WIDE_BLOCK), 8192)), dim3(SK_WIDE_BLOCK), 0, stream, A);
            ST_TRY(sk_read(h, s.cnt, sizeof(unsigned) * 8, stream));
            ST_TRY(resolve_plateaus(h[3]));
        }
        if (time_select) {  // [5] = 0 tells the caller that no (trustworthy) timing was taken
            stats_host[4] = timing_ok ? (int64_t)(select_ms * 1e6) : 0;
            stats_host[5] = timing_ok ? iters : 0;
        }
        if (stats_host) {
            stats_host[2] = iters; stats_host[3] = SK_ANC; stats_host[6] = ((int64_t)h[7] << 32) | (int64_t)h[6];
            stats_host[8] = h[8] != 0u;  // helper workgroups were lost in this call (a fall-back ran: slower, not wrong)
            stats_host[9] = A.n_helpers;
        }
    }
    ST_CHECK_LAUNCH();
    return ST_OK;
}

extern "C" int st_skeleton_components(int n_comp, const int32_t* comp_off, const int32_t* comp_size_host, int64_t m,
                                      const float* pts, const float* rad, const float* ysurf, const uint32_t* row_off,
                                      const uint32_t* col, const float* wgt, float grid_cell, int stages, int block_threads,
                                      float* dist, int32_t* pred, int32_t* root_local, float* tree_dist,
                                      int32_t* branch_parent, int32_t* branch_off, int32_t* branch_len, int32_t* n_branches,
                                      int32_t* path_verts, int32_t* branch_of, int64_t* stats_host, void* ws,
                                      int64_t ws_bytes, void* stream_) {
    (void)comp_size_host;
    return st_skeleton_components_seg(n_comp, comp_off, nullptr, nullptr, 1, m, pts, rad, ysurf, row_off, col, wgt, grid_cell,
                                      stages, block_threads, dist, pred, root_local, tree_dist, branch_parent, branch_off,
                                      branch_len, n_branches, path_verts, branch_of, stats_host, ws, ws_bytes, stream_, nullptr);
}

// Named single-stage entry points: the same call with one stage selected.
// st_tree_distance / st_sample_tree expect dist / pred / root_local from an earlier st_sssp call.
#define SK_STAGE_ENTRY(name, stage_bits)                                                                                  \
    extern "C" int name(int n_comp, const int32_t* comp_off, const int32_t* comp_size_host, int64_t m, const float* pts,   \
                        const float* rad, const float* ysurf, const uint32_t* row_off, const uint32_t* col, const float* wgt, \
                        float grid_cell, int block_threads, float* dist, int32_t* pred, int32_t* root_local, float* tree_dist, \
                        int32_t* branch_parent, int32_t* branch_off, int32_t* branch_len, int32_t* n_branches,             \
                        int32_t* path_verts, int32_t* branch_of, int64_t* stats_host, void* ws, int64_t ws_bytes,          \
                        void* stream) {                                                                                    \
        return st_skeleton_components(n_comp, comp_off, comp_size_host, m, pts, rad, ysurf, row_off, col, wgt, grid_cell,   \
                                      stage_bits, block_threads, dist, pred, root_local, tree_dist, branch_parent,         \
                                      branch_off, branch_len, n_branches, path_verts, branch_of, stats_host, ws, ws_bytes, \
                                      stream);                                                                             \
    }
SK_STAGE_ENTRY(st_sssp, 1)
SK_STAGE_ENTRY(st_tree_distance, 2)
SK_STAGE_ENTRY(st_sample_tree, 4)
