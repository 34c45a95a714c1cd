// st_post_process: prune / repair / smooth of the extracted skeletons on the device, before the one
// device-to-host copy that materialises the BranchSkeleton objects.
//
// Reference (host loops over small tensors, one GPU einsum per branch):
//   Pipeline.post_process            smart_tree/pipeline.py:95-106
//   DisjointTreeSkeleton.prune/...   smart_tree/data_types/tree.py:164-176 (only skeleton 0 is pruned)
//   TreeSkeleton.prune               tree.py:94-121  (+ BranchSkeleton.length / initial_radius, branch.py:61-67)
//   TreeSkeleton.repair              tree.py:73-92   (+ pts_to_nearest_tube_gpu, util/queries.py:89-133)
//   TreeSkeleton.smooth              tree.py:123-134 (zero-padded box filter, only when len > kernel)
// One workgroup per tree.  Branches are visited in id order (a parent always has a smaller id than its
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

#define PP_LDS_BRANCHES 8192  // branch tables of this size run their sequential chains out of LDS
__global__ void __launch_bounds__(PP_BLOCK) k_post_process(PpArgs A) {
    __shared__ int l_parent[PP_LDS_BRANCHES];
    __shared__ short l_depth[PP_LDS_BRANCHES];
    __shared__ uint8_t l_keep[PP_LDS_BRANCHES];
    const int tree = blockIdx.x, tid = threadIdx.x;
    const int b0 = A.tree_off[tree], nb = A.tree_off[tree + 1] - b0;
    const bool in_lds = nb <= PP_LDS_BRANCHES;
    if (in_lds) for (int b = tid; b < nb; b += PP_BLOCK) l_parent[b] = A.parent[b0 + b];
    // ---- prune (tree 0 only): length / initial radius per branch in parallel, then the keep chain
    for (int b = tid; b < nb; b += PP_BLOCK) A.keep[b0 + b] = 1;
    __syncthreads();
    if (A.do_prune && tree == 0 && nb > 0) {
        for (int b = tid >> 6; b < nb; b += PP_WAVES) {  // one wavefront per branch
            const int s = A.start[b0 + b] + 1, n = A.len[b0 + b], ln = tid & 63;
            float length = 0.0f;  // lanes evaluate 64 segment norms at a time; they are added in path order (sequential float32 sum)
            for (int i0 = 0; i0 + 1 < n; i0 += 64) {
                const int i = i0 + ln;
                float seg = 0.0f;
                if (i + 1 < n) {
                    const float d[3] = {A.xyz[3 * (s + i + 1)] - A.xyz[3 * (s + i)], A.xyz[3 * (s + i + 1) + 1] - A.xyz[3 * (s + i) + 1],
                                        A.xyz[3 * (s + i + 1) + 2] - A.xyz[3 * (s + i) + 2]};
                    seg = sqrtf(pp_dot(d, d));
                }
                const int cnt = (n - 1 - i0) < 64 ? (n - 1 - i0) : 64;
                for (int j = 0; j < cnt; j++) length = length + __shfl(seg, j);
            }
            if (ln == 0) {
                const float r0 = A.rad_in[s], r1 = A.rad_in[s + n - 1];
                const float initial = r0 > r1 ? r0 : r1;
                A.keep[b0 + b] = !(length < A.min_length) && !(initial < A.min_radius);  // own tests (tree.py:113-116)
            }
        }
        __syncthreads();
        if (in_lds) for (int b = tid; b < nb; b += PP_BLOCK) l_keep[b] = A.keep[b0 + b];
        __syncthreads();
        if (tid == 0) {  // the keep chain is sequential (a child needs its parent's verdict): LDS-resident when it fits
            if (in_lds) {
                l_keep[0] = 1;  // the root (smallest id) always stays (tree.py:101-103,120)
                for (int b = 1; b < nb; b++) {
                    const int p = l_parent[b];
                    if (!(p >= 0 && p < nb && l_keep[p])) l_keep[b] = 0;  // tree.py:111-112
                }
            } else {
                A.keep[b0] = 1;
                for (int b = 1; b < nb; b++) {
                    const int p = A.parent[b0 + b];
                    if (!(p >= 0 && p < nb && A.keep[b0 + p])) A.keep[b0 + b] = 0;
                }
            }
        }
        __syncthreads();
        if (in_lds) for (int b = tid; b < nb; b += PP_BLOCK) A.keep[b0 + b] = l_keep[b];
        __syncthreads();
    }
    // ---- repair: nearest point on the (already repaired) parent's tube chain.  A branch only needs its
    //      parent finished, so branches are processed level by level of the branch hierarchy (one wavefront
    //      per branch, lanes over the parent's tubes) instead of one after the other.
    __shared__ int s_maxdepth;
    int* depth = A.depth + b0;
    for (int b = tid; b < nb; b += PP_BLOCK) A.repaired[b0 + b] = 0;
    if (in_lds) for (int b = tid; b < nb; b += PP_BLOCK) l_keep[b] = A.keep[b0 + b];
    __syncthreads();
    if (tid == 0) {
        int md = 0;
        for (int b = 0; b < nb; b++) {  // parents have smaller ids: one forward pass
            const int p = in_lds ? l_parent[b] : A.parent[b0 + b];
            const bool kb = in_lds ? l_keep[b] : A.keep[b0 + b];
            const bool kp = p >= 0 && p < nb && (in_lds ? l_keep[p] : A.keep[b0 + p]);
            const bool go = A.do_repair && kb && kp;  // tree.py:80-82
            const int dp = go ? (in_lds ? (int)l_depth[p] : depth[p]) : -1;
            const int d = go ? (dp < 0 ? 1 : dp + 1) : -1;  // -1: not repaired; a parent without repair is ready at once
            if (in_lds) l_depth[b] = (short)(d > 32767 ? 32767 : d); else depth[b] = d;
            if (d > md) md = d;
        }
        s_maxdepth = md > 32767 ? 32767 : md;
    }
    __syncthreads();
    if (in_lds) for (int b = tid; b < nb; b += PP_BLOCK) depth[b] = l_depth[b];
    __syncthreads();
    const int maxdepth = s_maxdepth, lane = tid & 63, wave = tid >> 6;
    for (int level = 1; level <= maxdepth; level++) {
        for (int b = wave; b < nb; b += PP_WAVES) {  // wave-uniform
            if ((in_lds ? (int)l_depth[b] : depth[b]) != level) continue;
            const int p = A.parent[b0 + b];
            const int prep = __hip_atomic_load(&A.repaired[b0 + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int ps = A.start[b0 + p] + (prep ? 0 : 1);
            const int pn = A.len[b0 + p] + (prep ? 1 : 0);
            const int s = A.start[b0 + b];
            const float pt[3] = {A.xyz[3 * (s + 1)], A.xyz[3 * (s + 1) + 1], A.xyz[3 * (s + 1) + 2]};
            // argmin over the parent's tubes of |dist - radius| (first minimum; NaN counts as minimal)
            unsigned long long key = 0;
            for (int i = lane; i + 1 < pn; i += 64) {
                float a[3], bb[3];
                for (int k = 0; k < 3; k++) {
                    // only the parent's connection point (slot 0, written one level earlier by another wavefront)
                    // needs the L2-coherent load; the extracted path is immutable input
                    a[k] = (prep && i == 0) ? pp_ldf(&A.xyz[3 * ps + k]) : A.xyz[3 * (ps + i) + k];
                    bb[k] = A.xyz[3 * (ps + i + 1) + k];
                }
                const float ab[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]};
                const float ap[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
                float t = pp_dot(ap, ab) / pp_dot(ab, ab);
                t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
                const float proj[3] = {a[0] + t * ab[0], a[1] + t * ab[1], a[2] + t * ab[2]};
                const float r = (1.0f - t) * A.rad_in[ps + i] + t * A.rad_in[ps + i + 1];
                const float d[3] = {proj[0] - pt[0], proj[1] - pt[1], proj[2] - pt[2]};
                const float score = fabsf(sqrtf(pp_dot(d, d)) - r);
                const unsigned bits = score != score ? 0u : __float_as_uint(score) + 1u;  // >= 0: bits are ordered
                const unsigned long long k = ((unsigned long long)(0xffffffffu - bits) << 32) | (0xffffffffu - (unsigned)i);
                key = k > key ? k : key;
            }
            for (int d = 32; d > 0; d >>= 1) {  // wave max of the inverted key = minimum score, smallest index
                const unsigned long long o = __shfl_xor(key, d);
                key = o > key ? o : key;
            }
            if (lane == 0) {
                const int i = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
                float a[3], bb[3];
                for (int k = 0; k < 3; k++) {
                    a[k] = (prep && i == 0) ? pp_ldf(&A.xyz[3 * ps + k]) : A.xyz[3 * (ps + i) + k];
                    bb[k] = A.xyz[3 * (ps + i + 1) + k];
                }
                const float ab[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]};
                const float ap[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
                float t = pp_dot(ap, ab) / pp_dot(ab, ab);
                t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
                for (int k = 0; k < 3; k++) {
                    const float proj = a[k] + t * ab[k];
                    A.xyz[3 * s + k] = pt[k] + (proj - pt[k]);  // tree.py:89: tip + vector to the projection
                }
                A.repaired[b0 + b] = 1;
            }
        }
        __syncthreads();  // the next level reads this level's connection points (through L2)
    }
    // ---- radii: box filter over the (possibly prepended) radius array
    const float w = A.kernel > 0 ? 1.0f / (float)A.kernel : 0.0f;
    const int half = A.kernel / 2;
    for (int b = wave; b < nb; b += PP_WAVES) {  // one wavefront per branch
        const int rep = __hip_atomic_load(&A.repaired[b0 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int s = A.start[b0 + b] + (rep ? 0 : 1), n = A.len[b0 + b] + rep;
        const bool sm = A.do_smooth && A.keep[b0 + b] && n > A.kernel;  // tree.py:129
        if (lane == 0) A.smoothed[b0 + b] = sm;
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
extern "C" int st_post_process(int n_trees, const int32_t* tree_off, const int32_t* parent, const int32_t* start,
                               const int32_t* len, float* xyz, const float* rad_in, float* rad_out, uint8_t* keep,
                               uint8_t* repaired, uint8_t* smoothed, int32_t* depth_scratch, int do_prune, float min_radius,
                               float min_length, int do_repair, int do_smooth, int kernel_size, void* stream_) {
    if (n_trees <= 0) return ST_OK;
    ST_REQUIRE(!do_smooth || kernel_size > 0, "post_process: smoothing needs kernel_size > 0");
    PpArgs A;
    A.n_trees = n_trees; A.tree_off = tree_off; A.parent = parent; A.start = start; A.len = len; A.xyz = xyz;
    A.rad_in = rad_in; A.rad_out = rad_out; A.keep = keep; A.repaired = repaired; A.smoothed = smoothed; A.depth = depth_scratch;
    A.do_prune = do_prune; A.do_repair = do_repair; A.do_smooth = do_smooth; A.kernel = kernel_size;
    A.min_radius = min_radius; A.min_length = min_length;
    hipLaunchKernelGGL(k_post_process, dim3((unsigned)n_trees), dim3(PP_BLOCK), 0, (hipStream_t)stream_, A);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
