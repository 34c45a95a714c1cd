// st_post_process: prune / repair / smooth of the extracted skeletons on the device, before the one
// device-to-host copy that materialises the BranchSkeleton objects.
//
// Reference (host loops over small tensors, one GPU einsum per branch):
//   Pipeline.post_process            smart_tree/pipeline.py:95-106
//   DisjointTreeSkeleton.prune/...   smart_tree/data_types/tree.py:164-176 (only skeleton 0 is pruned)
//   TreeSkeleton.prune               tree.py:94-121  (+ BranchSkeleton.length / initial_radius, branch.py:61-67)
//   TreeSkeleton.repair              tree.py:73-92   (+ pts_to_nearest_tube_gpu, util/queries.py:89-133)
//   TreeSkeleton.smooth              tree.py:123-134 (zero-padded box filter, only when len > kernel)
// Branches are visited in the order of the branch hierarchy (a parent always has a smaller id than its
// children, so a child sees its parent already repaired, exactly like the reference's dict walk).
// Float32 operation order is fixed and mirrored by oracle/pipeline_oracle.py:
//   dot(a,b) = (ax*bx + ay*by) + az*bz, no contraction; length = sequential sum of segment norms;
//   box filter = sequential sum of r*w with w = fl32(1/k).
//
// Geometry layout: branch b owns slots [start[b], start[b] + len[b] + 1); slot start[b] is reserved for
// the connection point `repair` prepends (its radius slot must already hold a copy of the first
// radius, tree.py:92), the extracted path sits in start[b]+1 ...
#include "st_common.h"

#define PP_BLOCK 1024  // 16 wavefronts: the per-branch work is a chain of dependent loads, more waves = more branches in flight
#define PP_WAVES (PP_BLOCK / 64)

struct PpArgs {
    int n_trees;
    const int* tree_off;  // [T+1] branch ranges
    const int* parent;    // [B] parent branch id inside the tree (-1 / any id outside [0,nb): none)
    const int* start;     // [B]
    const int* len;       // [B] extracted path length (>= 2)
    float* xyz;           // [P,3]
    const float* rad_in;  // [P]
    float* rad_out;       // [P]
    uint8_t* keep;        // [B]
    uint8_t* repaired;    // [B]
    uint8_t* smoothed;    // [B]
    int* depth;           // [B] scratch: level of each branch in the repair order
    int do_prune, do_repair, do_smooth, kernel;
    float min_radius, min_length;
    const int* first_tree;  // batched call: [n_first] the first tree of every cloud (the one `prune` works on); nullptr: tree 0
    int n_first;
};

// L2 (agent scope) load of a float another wavefront of this workgroup wrote earlier in the launch
__device__ __forceinline__ float pp_ldf(const float* p) {
    return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ float pp_dot(const float* a, const float* b) {
    float s = a[0] * b[0];
    float t = a[1] * b[1];
    s = s + t;
    t = a[2] * b[2];
    return s + t;
}

// Score of one tube (a, b, radii ra, rb) for the connection point of a branch whose first vertex is pt: |distance to the
// segment - interpolated radius| (pts_to_nearest_tube_gpu, util/queries.py:107-133), as the ordered bit pattern the argmin
// works on (NaN counts as minimal).  t comes back for the caller that has to place the projection.
__device__ __forceinline__ unsigned pp_tube_bits(const float* a, const float* bb, float ra, float rb, const float* pt, float* t_out) {
    const float ab[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]};
    const float ap[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
    float t = pp_dot(ap, ab) / pp_dot(ab, ab);
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    const float proj[3] = {a[0] + t * ab[0], a[1] + t * ab[1], a[2] + t * ab[2]};
    const float r = (1.0f - t) * ra + t * rb;
    const float d[3] = {proj[0] - pt[0], proj[1] - pt[1], proj[2] - pt[2]};
    const float score = fabsf(sqrtf(pp_dot(d, d)) - r);
    *t_out = t;
    return score != score ? 0u : __float_as_uint(score) + 1u;  // >= 0: bits are ordered
}
// (score, tube index) as one key whose MAXIMUM is the minimum score at the smallest index
__device__ __forceinline__ unsigned long long pp_key(unsigned bits, unsigned i) {
    return ((unsigned long long)(0xffffffffu - bits) << 32) | (0xffffffffu - i);
}

// Three launches (round 5; it was one workgroup per tree doing everything, 251 us for one cloud's tree of ~290 branches, every
// step a chain of dependent loads on ONE compute unit):
//   k_pp_branch  one wavefront per branch of every tree, the whole chip: the branch's own prune tests (length, initial radius)
//                and the best tube among its parent's EXTRACTED path -- everything that does not depend on another branch's result
//   k_pp_tree    one workgroup per tree: the keep chain, the repair levels.  A level's branches are one lane each: only the
//                parent's first tube (from its connection point, placed one level earlier) is still to be scored against the
//                key k_pp_branch left
//   k_pp_smooth  one wavefront per branch: the box filter
// The key of k_pp_branch travels in the branch's own first two rad_out slots (rad_out is written by k_pp_smooth, last).
#define PP_WIDE_BLOCK 256
#define PP_WIDE_GRID 1024
__global__ void __launch_bounds__(PP_WIDE_BLOCK) k_pp_branch(PpArgs A) {
    const int B = A.tree_off[A.n_trees], lane = threadIdx.x & 63;
    const int nw = (int)gridDim.x * (PP_WIDE_BLOCK / 64);
    for (int g = (int)blockIdx.x * (PP_WIDE_BLOCK / 64) + ((int)threadIdx.x >> 6); g < B; g += nw) {  // wave-uniform
        int tree = 0, hi = A.n_trees;  // the last tree whose range starts at or before g (empty trees in between are skipped)
        while (hi - tree > 1) {
            const int mid = (tree + hi) >> 1;
            if (A.tree_off[mid] <= g) tree = mid; else hi = mid;
        }
        const int b0 = A.tree_off[tree], nb = A.tree_off[tree + 1] - b0;
        // ---- prune, own tests (skeleton 0 of its cloud only: tree.py:164-168)
        bool prune_this = tree == 0;
        if (A.first_tree) {
            prune_this = false;
            for (int k = 0; k < A.n_first; k++) prune_this = prune_this || A.first_tree[k] == tree;  // uniform: n_first <= 64
        }
        if (A.do_prune && prune_this) {
            const int s = A.start[g] + 1, n = A.len[g];
            float length = 0.0f;  // lanes evaluate 64 segment norms at a time; they are added in path order (sequential float32 sum)
            for (int i0 = 0; i0 + 1 < n; i0 += 64) {
                const int i = i0 + lane;
                float seg = 0.0f;
                if (i + 1 < n) {
                    const float d[3] = {A.xyz[3 * (s + i + 1)] - A.xyz[3 * (s + i)], A.xyz[3 * (s + i + 1) + 1] - A.xyz[3 * (s + i) + 1],
                                        A.xyz[3 * (s + i + 1) + 2] - A.xyz[3 * (s + i) + 2]};
                    seg = sqrtf(pp_dot(d, d));
                }
                const int cnt = (n - 1 - i0) < 64 ? (n - 1 - i0) : 64;
                for (int j = 0; j < cnt; j++)  // j is wave-uniform: v_readlane, not a cross-lane permute through LDS
                    length = length + __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(seg), j));
            }
            if (lane == 0) {
                const float r0 = A.rad_in[s], r1 = A.rad_in[s + n - 1];
                const float initial = r0 > r1 ? r0 : r1;
                A.keep[g] = !(length < A.min_length) && !(initial < A.min_radius);  // own tests (tree.py:113-116)
            }
        } else if (lane == 0) {
            A.keep[g] = 1;
        }
        // ---- repair: the best tube of the parent's extracted path (tube j = path vertices j, j + 1; immutable input)
        if (A.do_repair) {
            const int p = A.parent[g];
            unsigned long long key = 0;
            if (p >= 0 && p < nb) {
                const int ps = A.start[b0 + p] + 1, pn = A.len[b0 + p], s = A.start[g];
                const float pt[3] = {A.xyz[3 * (s + 1)], A.xyz[3 * (s + 1) + 1], A.xyz[3 * (s + 1) + 2]};
                for (int j = lane; j + 1 < pn; j += 64) {
                    float a[3], bb[3], t;
                    for (int k = 0; k < 3; k++) {
                        a[k] = A.xyz[3 * (ps + j) + k];
                        bb[k] = A.xyz[3 * (ps + j + 1) + k];
                    }
                    const unsigned long long k2 = pp_key(pp_tube_bits(a, bb, A.rad_in[ps + j], A.rad_in[ps + j + 1], pt, &t), (unsigned)j);
                    key = k2 > key ? k2 : key;
                }
                for (int d = 32; d > 0; d >>= 1) {  // wave max of the inverted key = minimum score, smallest index
                    const unsigned long long o = __shfl_xor(key, d);
                    key = o > key ? o : key;
                }
            }
            if (lane == 0) {
                unsigned* slot = (unsigned*)(A.rad_out + A.start[g]);
                slot[0] = (unsigned)(key >> 32);
                slot[1] = (unsigned)(key & 0xffffffffu);
            }
        }
    }
}

#define PP_LDS_BRANCHES 8192  // branch tables of this size run their chains out of LDS
__global__ void __launch_bounds__(PP_BLOCK) k_pp_tree(PpArgs A) {
    __shared__ int l_parent[PP_LDS_BRANCHES];
    __shared__ short l_depth[PP_LDS_BRANCHES];
    __shared__ uint8_t l_keep[PP_LDS_BRANCHES];
    __shared__ int s_maxdepth, s_flag;
    const int tree = blockIdx.x, tid = threadIdx.x;
    const int b0 = A.tree_off[tree], nb = A.tree_off[tree + 1] - b0;
    const bool in_lds = nb <= PP_LDS_BRANCHES;
    if (in_lds) for (int b = tid; b < nb; b += PP_BLOCK) l_parent[b] = A.parent[b0 + b];
    // ---- prune (skeleton 0 of its cloud only): the keep chain over the own tests k_pp_branch left in `keep`.
    //      keep[b] = own[b] and keep[parent[b]] (tree.py:111-112), the root always stays (tree.py:101-103,120).  A parent has a
    //      smaller id than its children, so the reference's walk in id order computes the one solution of that recurrence; in LDS
    //      it is reached by passes over all branches at once (as many passes as the hierarchy is deep) instead of one lane's walk.
    bool prune_this = tree == 0;  // tree.py:164-168
    if (A.first_tree) {
        prune_this = false;
        for (int k = 0; k < A.n_first; k++) prune_this = prune_this || A.first_tree[k] == tree;  // uniform: n_first <= 64
    }
    if (A.do_prune && prune_this && nb > 0) {
        if (in_lds) {
            for (int b = tid; b < nb; b += PP_BLOCK) l_keep[b] = b == 0 ? 1 : A.keep[b0 + b];
            do {
                __syncthreads();
                if (tid == 0) s_flag = 0;
                __syncthreads();
                for (int b = tid; b < nb; b += PP_BLOCK) {
                    if (b == 0 || !l_keep[b]) continue;
                    const int p = l_parent[b];
                    if (!(p >= 0 && p < nb && l_keep[p])) { l_keep[b] = 0; s_flag = 1; }
                }
                __syncthreads();
            } while (s_flag);
            for (int b = tid; b < nb; b += PP_BLOCK) A.keep[b0 + b] = l_keep[b];
        } else if (tid == 0) {
            A.keep[b0] = 1;
            for (int b = 1; b < nb; b++) {
                const int p = A.parent[b0 + b];
                if (!(p >= 0 && p < nb && A.keep[b0 + p])) A.keep[b0 + b] = 0;
            }
        }
        __syncthreads();
    }
    // ---- repair: nearest point on the (already repaired) parent's tube chain.  A branch only needs its
    //      parent finished, so branches are processed level by level of the branch hierarchy, a lane per branch.
    //      Level of a branch that is repaired (it and its parent are kept, tree.py:80-82): one more than its parent's, 1 under a
    //      parent that is not repaired itself; -1 = not repaired.
    int* depth = A.depth + b0;
    for (int b = tid; b < nb; b += PP_BLOCK) A.repaired[b0 + b] = 0;
    if (in_lds) {
        for (int b = tid; b < nb; b += PP_BLOCK) l_keep[b] = A.keep[b0 + b];
        if (tid == 0) s_maxdepth = 0;
        __syncthreads();
        for (int b = tid; b < nb; b += PP_BLOCK) {
            const int p = l_parent[b];
            l_depth[b] = (A.do_repair && l_keep[b] && p >= 0 && p < nb && l_keep[p]) ? 1 : -1;
        }
        do {
            __syncthreads();
            if (tid == 0) s_flag = 0;
            __syncthreads();
            for (int b = tid; b < nb; b += PP_BLOCK) {
                const int d = l_depth[b];
                if (d < 0) continue;
                const int dp = l_depth[l_parent[b]];
                int nd = dp < 0 ? 1 : dp + 1;
                nd = nd > 32767 ? 32767 : nd;
                if (nd != d) { l_depth[b] = (short)nd; s_flag = 1; }
            }
            __syncthreads();
        } while (s_flag);
        for (int b = tid; b < nb; b += PP_BLOCK) {
            depth[b] = l_depth[b];
            if (l_depth[b] > 0) atomicMax(&s_maxdepth, (int)l_depth[b]);
        }
    } else if (tid == 0) {
        int md = 0;
        for (int b = 0; b < nb; b++) {  // parents have smaller ids: one forward pass
            const int p = A.parent[b0 + b];
            const bool go = A.do_repair && A.keep[b0 + b] && p >= 0 && p < nb && A.keep[b0 + p];
            const int dp = go ? depth[p] : -1;
            const int d = go ? (dp < 0 ? 1 : dp + 1) : -1;
            depth[b] = d;
            if (d > md) md = d;
        }
        s_maxdepth = md > 32767 ? 32767 : md;
    }
    __syncthreads();
    const int maxdepth = s_maxdepth;
    if (nb <= PP_BLOCK) {
        // one branch per lane: everything but the parent's connection point (placed one level earlier) is loaded before the
        // level loop, so a level costs one round trip through L2 instead of a chain of them
        const int b = tid;
        const int lev = b < nb ? (int)l_depth[b] : -1;
        int s = 0, ps = 0, prep = 0;
        float pt[3] = {0, 0, 0}, b0v[3] = {0, 0, 0}, ai[3] = {0, 0, 0}, bi[3] = {0, 0, 0}, ra0 = 0, rb0 = 0;
        unsigned long long key_imm = 0;
        if (lev > 0) {
            const int p = l_parent[b];
            prep = l_depth[p] > 0;  // the parent is repaired (at an earlier level) exactly when it has a level
            ps = A.start[b0 + p] + (prep ? 0 : 1);
            s = A.start[b0 + b];
            for (int k = 0; k < 3; k++) pt[k] = A.xyz[3 * (s + 1) + k];
            const unsigned* slot = (const unsigned*)(A.rad_out + s);
            key_imm = ((unsigned long long)slot[0] << 32) | slot[1];
            if (prep) {
                if (key_imm) key_imm -= 1;  // index j -> j + 1 (the low word holds 0xffffffff - j)
                for (int k = 0; k < 3; k++) b0v[k] = A.xyz[3 * (ps + 1) + k];
                ra0 = A.rad_in[ps];
                rb0 = A.rad_in[ps + 1];
            }
            if (key_imm || !prep) {  // the best tube of the extracted path (without one: the slot in front of it, as the chain walk read)
                const int i = (int)(0xffffffffu - (unsigned)(key_imm & 0xffffffffu));
                for (int k = 0; k < 3; k++) {
                    ai[k] = A.xyz[3 * (ps + i) + k];
                    bi[k] = A.xyz[3 * (ps + i + 1) + k];
                }
            }
        }
        for (int level = 1; level <= maxdepth; level++) {
            if (lev == level) {
                unsigned long long key = key_imm;
                float a[3] = {ai[0], ai[1], ai[2]}, bb[3] = {bi[0], bi[1], bi[2]}, t;
                if (prep) {
                    float a0[3];
                    for (int k = 0; k < 3; k++) a0[k] = pp_ldf(&A.xyz[3 * ps + k]);  // written one level earlier by another lane: through L2
                    const unsigned long long k0 = pp_key(pp_tube_bits(a0, b0v, ra0, rb0, pt, &t), 0u);
                    if (k0 > key) {
                        for (int k = 0; k < 3; k++) { a[k] = a0[k]; bb[k] = b0v[k]; }
                    }
                }
                const float ab[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]};
                const float ap[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
                t = pp_dot(ap, ab) / pp_dot(ab, ab);
                t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
                for (int k = 0; k < 3; k++) {
                    const float proj = a[k] + t * ab[k];
                    A.xyz[3 * s + k] = pt[k] + (proj - pt[k]);  // tree.py:89: tip + vector to the projection
                }
                A.repaired[b0 + b] = 1;
            }
            __threadfence();
            __syncthreads();  // the next level reads this level's connection points (through L2)
        }
        return;
    }
    for (int level = 1; level <= maxdepth; level++) {
        for (int b = tid; b < nb; b += PP_BLOCK) {
            if ((in_lds ? (int)l_depth[b] : depth[b]) != level) continue;
            const int p = in_lds ? l_parent[b] : A.parent[b0 + b];
            const int prep = __hip_atomic_load(&A.repaired[b0 + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int ps = A.start[b0 + p] + (prep ? 0 : 1);
            const int s = A.start[b0 + b];
            const float pt[3] = {A.xyz[3 * (s + 1)], A.xyz[3 * (s + 1) + 1], A.xyz[3 * (s + 1) + 2]};
            // argmin over the parent's tubes of |dist - radius| (first minimum; NaN counts as minimal): k_pp_branch's key over the
            // extracted path (its tube j is tube j + 1 of a repaired parent), and the tube from the parent's connection point
            const unsigned* slot = (const unsigned*)(A.rad_out + s);
            unsigned long long key = ((unsigned long long)slot[0] << 32) | slot[1];
            float t;
            if (prep) {
                if (key) key -= 1;  // index j -> j + 1 (the low word holds 0xffffffff - j)
                float a[3], bb[3];
                for (int k = 0; k < 3; k++) {
                    a[k] = pp_ldf(&A.xyz[3 * ps + k]);  // written one level earlier by another lane: through L2
                    bb[k] = A.xyz[3 * (ps + 1) + k];
                }
                const unsigned long long k0 = pp_key(pp_tube_bits(a, bb, A.rad_in[ps], A.rad_in[ps + 1], pt, &t), 0u);
                key = k0 > key ? k0 : key;
            }
            const int i = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
            float a[3], bb[3];
            for (int k = 0; k < 3; k++) {
                a[k] = (prep && i == 0) ? pp_ldf(&A.xyz[3 * ps + k]) : A.xyz[3 * (ps + i) + k];
                bb[k] = A.xyz[3 * (ps + i + 1) + k];
            }
            const float ab[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]};
            const float ap[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
            t = pp_dot(ap, ab) / pp_dot(ab, ab);
            t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
            for (int k = 0; k < 3; k++) {
                const float proj = a[k] + t * ab[k];
                A.xyz[3 * s + k] = pt[k] + (proj - pt[k]);  // tree.py:89: tip + vector to the projection
            }
            A.repaired[b0 + b] = 1;
        }
        __threadfence();
        __syncthreads();  // the next level reads this level's connection points (through L2)
    }
}

// ---- radii: box filter over the (possibly prepended) radius array, one wavefront per branch of every tree
__global__ void __launch_bounds__(PP_WIDE_BLOCK) k_pp_smooth(PpArgs A) {
    const int B = A.tree_off[A.n_trees], lane = threadIdx.x & 63;
    const int nw = (int)gridDim.x * (PP_WIDE_BLOCK / 64);
    const float w = A.kernel > 0 ? 1.0f / (float)A.kernel : 0.0f;
    const int half = (A.kernel - 1) / 2;  // F.conv1d(padding="same"): left pad (k-1)/2, the extra sample of an even kernel goes right
    for (int g = (int)blockIdx.x * (PP_WIDE_BLOCK / 64) + ((int)threadIdx.x >> 6); g < B; g += nw) {
        const int rep = A.repaired[g];
        const int s = A.start[g] + (rep ? 0 : 1), n = A.len[g] + rep;
        const bool sm = A.do_smooth && A.keep[g] && n > A.kernel;  // tree.py:129
        if (lane == 0) A.smoothed[g] = sm;
        for (int i = lane; i < n; i += 64) {
            if (!sm) { A.rad_out[s + i] = A.rad_in[s + i]; continue; }
            float acc = 0.0f;
            for (int j = 0; j < A.kernel; j++) {
                const int q = i - half + j;
                const float v = (q >= 0 && q < n) ? A.rad_in[s + q] : 0.0f;
                acc = acc + v * w;
            }
            A.rad_out[s + i] = acc;
        }
    }
}

// tree_off [T+1], parent/start/len [B], xyz [P,3] (in/out), rad_in/rad_out [P], keep/repaired/smoothed [B],
// depth_scratch [B] int32
// Batched form: the trees of several clouds in one call; first_tree [n_first] (device) = the first tree of every cloud that
// has one -- `prune` applies to those (DisjointTreeSkeleton.prune touches skeletons[0] of ITS cloud only).
extern "C" int st_post_process_seg(int n_trees, const int32_t* tree_off, const int32_t* parent, const int32_t* start,
                               const int32_t* len, float* xyz, const float* rad_in, float* rad_out, uint8_t* keep,
                               uint8_t* repaired, uint8_t* smoothed, int32_t* depth_scratch, int do_prune, float min_radius,
                               float min_length, int do_repair, int do_smooth, int kernel_size, const int32_t* first_tree,
                               int n_first, void* stream_) {
    if (n_trees <= 0) return ST_OK;
    ST_REQUIRE(n_first >= 0 && n_first <= 64, "post_process: at most 64 clouds per batch");
    ST_REQUIRE(!do_smooth || kernel_size > 0, "post_process: smoothing needs kernel_size > 0");
    PpArgs A;
    A.n_trees = n_trees; A.tree_off = tree_off; A.parent = parent; A.start = start; A.len = len; A.xyz = xyz;
    A.rad_in = rad_in; A.rad_out = rad_out; A.keep = keep; A.repaired = repaired; A.smoothed = smoothed; A.depth = depth_scratch;
    A.do_prune = do_prune; A.do_repair = do_repair; A.do_smooth = do_smooth; A.kernel = kernel_size;
    A.min_radius = min_radius; A.min_length = min_length;
    A.first_tree = first_tree; A.n_first = first_tree ? n_first : 0;
    hipLaunchKernelGGL(k_pp_branch, dim3(PP_WIDE_GRID), dim3(PP_WIDE_BLOCK), 0, (hipStream_t)stream_, A);
    hipLaunchKernelGGL(k_pp_tree, dim3((unsigned)n_trees), dim3(PP_BLOCK), 0, (hipStream_t)stream_, A);
    hipLaunchKernelGGL(k_pp_smooth, dim3(PP_WIDE_GRID), dim3(PP_WIDE_BLOCK), 0, (hipStream_t)stream_, A);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

extern "C" int st_post_process(int n_trees, const int32_t* tree_off, const int32_t* parent, const int32_t* start,
                               const int32_t* len, float* xyz, const float* rad_in, float* rad_out, uint8_t* keep,
                               uint8_t* repaired, uint8_t* smoothed, int32_t* depth_scratch, int do_prune, float min_radius,
                               float min_length, int do_repair, int do_smooth, int kernel_size, void* stream_) {
    return st_post_process_seg(n_trees, tree_off, parent, start, len, xyz, rad_in, rad_out, keep, repaired, smoothed,
                               depth_scratch, do_prune, min_radius, min_length, do_repair, do_smooth, kernel_size, nullptr, 0,
                               stream_);
}

// ------------------------------------------------------------------ branch assembly ---
// sample_tree's per-component output (branch table + path vertex lists, skeleton.hip) -> the flat branch
// layout st_post_process works on: BranchSkeleton construction of skeleton/path.py:128-133 for every tree
// at once.  Branch k owns geometry slots [start[k], start[k] + len[k] + 1); slot start[k] is reserved for the
// connection point `repair` prepends and is pre-filled with the branch's first vertex (tree.py:92 reads
// its radius).
struct AsmArgs {
    int C;
    const int32_t *comp_off, *n_branches, *branch_parent, *branch_off, *branch_len, *path_verts, *vert_order;
    const float *medial, *radius;
    int32_t *tree_off, *parent, *start, *length;
    float *xyz, *rad;
    uint32_t *len1, *src0, *vbase;  // scratch [cap_b]
    int64_t cap_b, cap_p;
    int64_t* counts;  // device: B, P
};

__global__ void __launch_bounds__(1024) k_asm_trees(AsmArgs A) {  // one workgroup: tree_off = scan of the branch counts
    __shared__ uint32_t s_scan[17];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (int c0 = 0; c0 < A.C; c0 += (int)blockDim.x) {
        const int c = c0 + (int)threadIdx.x;
        const uint32_t v = c < A.C ? (uint32_t)A.n_branches[c] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(v, s_scan, &tot);
        const uint32_t carry = s_carry;
        if (c < A.C) A.tree_off[c] = (int32_t)(carry + ex);
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { A.tree_off[A.C] = (int32_t)s_carry; A.counts[0] = (int64_t)s_carry; }
}

__global__ void __launch_bounds__(256) k_asm_branches(AsmArgs A) {
    const int64_t B = A.tree_off[A.C];
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < A.cap_b; b += (int64_t)gridDim.x * blockDim.x) {
        if (b >= B) { A.len1[b] = 0u; continue; }
        int lo = 0, hi = A.C;  // tree with tree_off[c] <= b < tree_off[c + 1] (empty trees have equal offsets: take the last)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)A.tree_off[mid] <= b) lo = mid; else hi = mid; }
        const int base = A.comp_off[lo];
        const int slot = base + (int)(b - A.tree_off[lo]);
        const int len = A.branch_len[slot];
        A.parent[b] = A.branch_parent[slot];
        A.length[b] = len;
        A.len1[b] = (uint32_t)(len + 1);
        A.src0[b] = (uint32_t)(base + A.branch_off[slot]);
        A.vbase[b] = (uint32_t)base;
    }
}

__global__ void __launch_bounds__(256) k_asm_geometry(AsmArgs A) {
    const int64_t B = A.tree_off[A.C];
    const int64_t P = B > 0 ? (int64_t)A.start[B - 1] + A.length[B - 1] + 1 : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) A.counts[1] = P;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < P && k < A.cap_p; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = B;  // branch with start[b] <= k
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)A.start[mid] <= k) lo = mid; else hi = mid; }
        const int64_t j = k - A.start[lo];  // 0 = the reserved slot
        const int64_t src = (int64_t)A.src0[lo] + (j > 0 ? j - 1 : 0);
        const int64_t id = A.vert_order[(int64_t)A.path_verts[src] + A.vbase[lo]];
        A.xyz[3 * k] = A.medial[3 * id]; A.xyz[3 * k + 1] = A.medial[3 * id + 1]; A.xyz[3 * k + 2] = A.medial[3 * id + 2];
        A.rad[k] = A.radius[id];
    }
}

static void asm_layout(StArena& a, int64_t cap_b, uint32_t** len1, uint32_t** src0, uint32_t** vbase, int64_t** counts, char** scan_ws,
                       int64_t* scan_bytes) {
    *len1 = a.take<uint32_t>(cap_b);
    *src0 = a.take<uint32_t>(cap_b);
    *vbase = a.take<uint32_t>(cap_b);
    *counts = a.take<int64_t>(2);
    *scan_bytes = st_scan_ws_bytes(cap_b);
    *scan_ws = a.take<char>(*scan_bytes);
}

extern "C" int64_t st_assemble_workspace_bytes(int64_t cap_b) {
    StArena a(nullptr, 0);
    uint32_t *l, *s0, *vb; int64_t* cn; char* sw; int64_t sb;
    asm_layout(a, cap_b, &l, &s0, &vb, &cn, &sw, &sb);
    return a.used;
}

// Outputs are sized by the caller to the capacities cap_b (branches; every component-local branch slot is enough:
// cap_b = m) and cap_p (geometry slots; path vertices + branches <= 2 m); counts_host receives the numbers used
// (NULL: no read-back -- branches and path vertices + branches are known from st_skeleton_components' stats_host[6]).
extern "C" int st_assemble_branches(int n_comp, const int32_t* comp_off, const int32_t* n_branches, const int32_t* branch_parent,
                                    const int32_t* branch_off, const int32_t* branch_len, const int32_t* path_verts,
                                    const int32_t* vert_order, const float* medial, const float* radius, int32_t* tree_off,
                                    int32_t* parent, int32_t* start, int32_t* length, float* xyz, float* rad, int64_t cap_b,
                                    int64_t cap_p, int64_t* counts_host, void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (counts_host) counts_host[0] = counts_host[1] = 0;
    if (n_comp <= 0) return ST_OK;
    ST_REQUIRE(cap_b >= 1 && cap_p >= 1, "assemble: empty capacity");
    StArena a(ws, ws_bytes);
    AsmArgs A;
    char* scan_ws; int64_t scan_bytes;
    asm_layout(a, cap_b, &A.len1, &A.src0, &A.vbase, &A.counts, &scan_ws, &scan_bytes);
    if (!a.ok() || !scan_ws) {
        st_set_error("assemble: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    A.C = n_comp; A.comp_off = comp_off; A.n_branches = n_branches; A.branch_parent = branch_parent; A.branch_off = branch_off;
    A.branch_len = branch_len; A.path_verts = path_verts; A.vert_order = vert_order; A.medial = medial; A.radius = radius;
    A.tree_off = tree_off; A.parent = parent; A.start = start; A.length = length; A.xyz = xyz; A.rad = rad;
    A.cap_b = cap_b; A.cap_p = cap_p;
    hipLaunchKernelGGL(k_asm_trees, dim3(1), dim3(1024), 0, stream, A);
    const unsigned gb = (unsigned)(st_div_up(cap_b, 256) < 1024 ? st_div_up(cap_b, 256) : 1024),
                   gp = (unsigned)(st_div_up(cap_p, 256) < 2048 ? st_div_up(cap_p, 256) : 2048);
    hipLaunchKernelGGL(k_asm_branches, dim3(gb), dim3(256), 0, stream, A);
    ST_TRY(st_exclusive_scan_u32(A.len1, (uint32_t*)start, cap_b, nullptr, scan_ws, scan_bytes, stream));
    hipLaunchKernelGGL(k_asm_geometry, dim3(gp), dim3(256), 0, stream, A);
    if (!counts_host) { ST_CHECK_LAUNCH(); return ST_OK; }  // the caller has the counts from st_skeleton_components' stats[6]
    (void)hipMemcpyAsync(counts_host, A.counts, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, stream);
    st_stream_wait(stream);
    ST_CHECK_LAUNCH();
    ST_REQUIRE(counts_host[0] <= cap_b && counts_host[1] <= cap_p, "assemble: capacity exceeded (%lld branches, %lld slots)",
               (long long)counts_host[0], (long long)counts_host[1]);
    return ST_OK;
}
