// st_voxelize_blocks: 4 m blocks + halo, per-block first-point-wins voxelisation, collate.
//
// Replaces, for the inference path, the reference's host loop
//   SingleTreeInference.compute_blocks   smart_tree/dataset/dataset.py:166-190
//   SingleTreeInference.__getitem__      smart_tree/dataset/dataset.py:192-226 (spconv PointToVoxel, CPU)
//   batch_collate                        smart_tree/model/sparse.py:40-61
// with one device-resident pass over the cloud: block histogram -> kept blocks (count > min,
// lexicographic order) -> per-block bounding boxes over halo members -> hash insert with
// atomicMin(point index) (deterministic "first point wins") -> the occupied slots ARE the voxels: streamed out of the table
// and put in (block, representative point index) order, which is exactly the order a sequential voxeliser emits.
// All float arithmetic that decides an integer (block id, membership, voxel coordinate, grid
// size, inner mask) is float32 in the same operation order as the oracle (oracle/voxel_oracle.py).
#include "st_common.h"
#include "st_grid.h"  // ST_MAX_SEG

#define VX_BLOCK 256
#define VX_TABLE_CAP 32768  // least capacity of the block table: cells of the block-id bounding box of ONE cloud (32^3 blocks
                            // of 4 m = 128 m per axis); the table grows with max_blocks (vx_table_cells) for larger plots
static inline int64_t vx_table_cells(int max_blocks, int nseg) {  // per cloud: max_blocks is the capacity of the whole batch
    const int64_t want = 8ll * ((max_blocks + nseg - 1) / (nseg > 0 ? nseg : 1));
    return want > VX_TABLE_CAP ? want : VX_TABLE_CAP;
}
#define VX_LDS_BLOCKS 128

// Batched calls: `nseg` independent clouds in one point array, cloud s = points [seg_off[s], seg_off[s+1]).  Every
// streaming kernel runs with blockIdx.y = cloud; blocks are numbered cloud by cloud (inside a cloud in torch.unique
// order), so the collated batch is the concatenation of the one-cloud batches with the block index shifted.
struct VxState {
    int lo[3], hi[3];    // block-id bounding box over ALL clouds of the call
    uint32_t n_blocks;
    uint32_t n_vox;
    uint32_t overflow;   // bit0: block table, bit1: max_blocks, bit2: hash full, bit3: a block's voxel grid exceeds the 16-bit key fields, bit4: non-finite coordinates
    int table_cells;     // capacity of the block table per cloud (vx_table_cells)
    uint32_t seg_blk_off[ST_MAX_SEG + 1];  // first block of every cloud
    uint32_t seg_vox_off[ST_MAX_SEG + 1];  // first voxel of every cloud (written by the gather pass)
};

struct VxParams {
    float bs;          // block size
    float bs_inv;      // 1 / bs when bs is a power of two (then v * bs_inv == v / bs exactly), else 0
    float half_outer;  // (block + 2*buffer)/2 rounded to float
    float half_inner;  // block/2
    float bs_half;     // block/2 (centre offset)
    float vs;          // voxel size
    float vs_inv;      // fl(1 / vs) (vx_floor_div)
    int min_points;
    int max_blocks;
    int nseg;
    int whole;         // st_voxelize_cloud: no blocks -- every cloud is ONE block holding all of its points (TreeDataset.process_cloud)
};

__device__ __forceinline__ int vx_block_id(float v, const VxParams& p) {
    if (p.whole) return 0;
    return (int)floorf(p.bs_inv != 0.0f ? v * p.bs_inv : v / p.bs);
}

// Deals the points [i0, i1) of a cloud over the launch, FOUR CONSECUTIVE POINTS PER LANE per step: three 16-byte loads
// (48 contiguous bytes) instead of twelve dword loads, all issued before the first point is looked at.
// fn(i, x, y, z) is called for every point of the range.
template <class F>
__device__ __forceinline__ void vx_stream_points(const float* __restrict__ xyz, int64_t n_total, int64_t i0, int64_t i1, F fn) {
    const bool vec = (((uintptr_t)xyz) & 15) == 0;
    const int64_t step = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t base = (i0 & ~(int64_t)3) + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; base < i1; base += step) {
        float c[12];
        if (vec && base + 4 <= n_total) {
            const float4* v = reinterpret_cast<const float4*>(xyz + 3 * base);
            const float4 f0 = v[0], f1 = v[1], f2 = v[2];
            c[0] = f0.x; c[1] = f0.y; c[2] = f0.z; c[3] = f0.w; c[4] = f1.x; c[5] = f1.y; c[6] = f1.z; c[7] = f1.w;
            c[8] = f2.x; c[9] = f2.y; c[10] = f2.z; c[11] = f2.w;
        } else {
#pragma unroll
            for (int j = 0; j < 12; j++) c[j] = base + j / 3 < n_total ? xyz[3 * base + j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t i = base + j;
            if (i >= i0 && i < i1) fn(i, c[3 * j], c[3 * j + 1], c[3 * j + 2]);
        }
    }
}

__global__ void k_vx_init(VxState* st, int table_cells) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->table_cells = table_cells;
        for (int a = 0; a < 3; a++) { st->lo[a] = 0x7fffffff; st->hi[a] = (int)0x80000000; }
        st->n_blocks = 0; st->n_vox = 0; st->overflow = 0;
    }
}

__device__ __forceinline__ int vx_wave_min(int v) { for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(v, d); v = o < v ? o : v; } return v; }
__device__ __forceinline__ int vx_wave_max(int v) { for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(v, d); v = o > v ? o : v; } return v; }

__global__ void __launch_bounds__(VX_BLOCK) k_vx_bbox(const float* xyz, int64_t n, const int* seg_off, VxParams p, VxState* st) {
    __shared__ int lo[3], hi[3];
    if (threadIdx.x < 3) { lo[threadIdx.x] = 0x7fffffff; hi[threadIdx.x] = (int)0x80000000; }
    __syncthreads();
    const int seg = blockIdx.y;
    const int64_t i0 = seg_off ? seg_off[seg] : 0, i1 = seg_off ? seg_off[seg + 1] : n;
    int l[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, h[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    bool finite = true;  // NaN / infinite coordinates have no block: the call fails with a message that says so
    vx_stream_points(xyz, n, i0, i1, [&](int64_t, float x, float y, float z) {
        if (!(fabsf(x) <= 3.0e38f && fabsf(y) <= 3.0e38f && fabsf(z) <= 3.0e38f)) { finite = false; return; }  // (no integer is computed from it)
        const int q[3] = {vx_block_id(x, p), vx_block_id(y, p), vx_block_id(z, p)};
#pragma unroll
        for (int a = 0; a < 3; a++) { l[a] = q[a] < l[a] ? q[a] : l[a]; h[a] = q[a] > h[a] ? q[a] : h[a]; }
    });
    if (!finite) atomicOr(&st->overflow, 16u);
#pragma unroll
    for (int a = 0; a < 3; a++) { l[a] = vx_wave_min(l[a]); h[a] = vx_wave_max(h[a]); }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) { atomicMin(&lo[a], l[a]); atomicMax(&hi[a], h[a]); }
    __syncthreads();
    if (threadIdx.x < 3 && lo[threadIdx.x] <= hi[threadIdx.x]) { atomicMin(&st->lo[threadIdx.x], lo[threadIdx.x]); atomicMax(&st->hi[threadIdx.x], hi[threadIdx.x]); }
}

__device__ __forceinline__ bool vx_dims(const VxState* st, int* d) {
    int64_t total = 1;
    for (int a = 0; a < 3; a++) { d[a] = st->hi[a] - st->lo[a] + 1; if (d[a] < 1) return false; total *= d[a]; }
    return total <= st->table_cells;
}

#define VX_LDS_CELLS 2048
__global__ void __launch_bounds__(VX_BLOCK) k_vx_hist(const float* xyz, int64_t n, const int* seg_off, VxParams p, VxState* st,
                                                      int* table) {
    __shared__ int h[VX_LDS_CELLS];
    int d[3];
    if (st->overflow & 16u) return;  // non-finite coordinates: the call fails (vx_voxelize), nothing is computed from them
    if (!vx_dims(st, d)) { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) atomicOr(&st->overflow, 1u); return; }
    // a tree spans a few dozen blocks: count in LDS, flush once per workgroup (global atomics on a
    // handful of words would serialise a million points)
    const int ncell = d[0] * d[1] * d[2];
    const int seg = blockIdx.y;
    int* tab = table + (int64_t)seg * ncell;
    const int64_t i0 = seg_off ? seg_off[seg] : 0, i1 = seg_off ? seg_off[seg + 1] : n;
    const bool use_lds = ncell <= VX_LDS_CELLS;
    if (use_lds) for (int c = threadIdx.x; c < ncell; c += blockDim.x) h[c] = 0;
    __syncthreads();
    const int l0 = st->lo[0], l1 = st->lo[1], l2 = st->lo[2];
    vx_stream_points(xyz, n, i0, i1, [&](int64_t, float x, float y, float z) {
        const int c = ((vx_block_id(x, p) - l0) * d[1] + (vx_block_id(y, p) - l1)) * d[2] + (vx_block_id(z, p) - l2);
        if (use_lds) atomicAdd(&h[c], 1); else atomicAdd(&tab[c], 1);
    });
    __syncthreads();
    if (use_lds) for (int c = threadIdx.x; c < ncell; c += blockDim.x) if (h[c]) atomicAdd(&tab[c], h[c]);
}

// single workgroup: counts -> block rank (or -1), centres, bbox init; cells are walked cloud by cloud
__global__ void __launch_bounds__(VX_BLOCK) k_vx_blocks(VxState* st, int* table, VxParams p, float* centres, unsigned* blk_lo,
                                                        unsigned* blk_hi, int32_t* blk_seg) {
    __shared__ uint32_t lds[VX_BLOCK / 64 + 1];
    int d[3];
    if (!vx_dims(st, d)) return;
    const int ncell = d[0] * d[1] * d[2];
    const int64_t total_cells = (int64_t)ncell * p.nseg;
    uint32_t carry = 0;
    for (int64_t base = 0; base < total_cells; base += VX_BLOCK) {
        const int64_t g = base + threadIdx.x;
        const int seg = (int)(g / ncell), c = (int)(g % ncell);
        uint32_t keep = (g < total_cells && table[g] > p.min_points) ? 1u : 0u;
        uint32_t total;
        uint32_t rank = block_exclusive_scan(keep, lds, &total) + carry;
        if (g < total_cells) {
            if (c == 0) st->seg_blk_off[seg] = rank < (uint32_t)p.max_blocks ? rank : (uint32_t)p.max_blocks;
            if (keep && (int)rank < p.max_blocks) {
                table[g] = (int)rank;
                int cz = c % d[2], cy = (c / d[2]) % d[1], cx = c / (d[2] * d[1]);
                float id[3] = {(float)(cx + st->lo[0]), (float)(cy + st->lo[1]), (float)(cz + st->lo[2])};
                for (int a = 0; a < 3; a++) {
                    centres[3 * rank + a] = id[a] * p.bs + p.bs_half;
                    blk_lo[3 * rank + a] = 0xffffffffu;
                    blk_hi[3 * rank + a] = 0u;
                }
                if (blk_seg) blk_seg[rank] = seg;
            } else {
                table[g] = -1;
            }
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        st->n_blocks = carry;
        st->seg_blk_off[p.nseg] = carry < (uint32_t)p.max_blocks ? carry : (uint32_t)p.max_blocks;
        if ((int)carry > p.max_blocks) atomicOr(&st->overflow, 2u);
    }
}

// Calls fn(b) for every kept block whose halo cube [c - half_outer, c + half_outer) holds p, in (x, y, z) block order.
// The test is separable: per axis, which of the three neighbouring block columns hold the coordinate (the centre is
// recomputed with the expression k_vx_blocks stores, so the decision is bit-identical to testing `centres`).  With a halo
// narrower than half a block at most TWO columns per axis qualify, i.e. at most eight blocks: their table entries are
// requested together (eight independent loads), then fn runs for the hits -- eight uniform steps per point.  (Walking the 27
// combinations and calling fn inside cost a wavefront 27 serialised bodies, each with its own chain of dependent loads: some
// lane of 64 scattered points sits near every face.)  `table` = the slice of the point's own cloud.
template <class F>
__device__ __forceinline__ void vx_for_each_block(const float* pt, const VxState* st, const int* d, const int* table,
                                                  const VxParams& p, F fn) {
    if (p.whole) {  // the cloud's only block
        if (table[0] >= 0) fn(table[0]);
        return;
    }
    int q[3];
    unsigned ok[3];
    for (int a = 0; a < 3; a++) {
        q[a] = vx_block_id(pt[a], p) - st->lo[a];
        ok[a] = 0;
        for (int o = -1; o <= 1; o++) {
            const int c = q[a] + o;
            if (c < 0 || c >= d[a]) continue;
            const float ctr = (float)(c + st->lo[a]) * p.bs + p.bs_half;
            if (pt[a] >= ctr - p.half_outer && pt[a] < ctr + p.half_outer) ok[a] |= 1u << (o + 1);
        }
    }
    if (!(ok[0] && ok[1] && ok[2])) return;
    if (__popc(ok[0]) <= 2 && __popc(ok[1]) <= 2 && __popc(ok[2]) <= 2) {
        int lo_opt[3], hi_opt[3];  // column offsets (-1, 0, 1) of the first / second qualifying column, hi = 2: none
        for (int a = 0; a < 3; a++) {
            lo_opt[a] = __ffs(ok[a]) - 2;
            const unsigned rest = ok[a] & (ok[a] - 1u);
            hi_opt[a] = rest ? __ffs(rest) - 2 : 2;
        }
        int blk[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {  // j = (x choice, y choice, z choice): ascending (x, y, z) order
            const int ox = (j & 4) ? hi_opt[0] : lo_opt[0], oy = (j & 2) ? hi_opt[1] : lo_opt[1], oz = (j & 1) ? hi_opt[2] : lo_opt[2];
            blk[j] = (ox == 2 || oy == 2 || oz == 2) ? -1 : table[((q[0] + ox) * d[1] + (q[1] + oy)) * d[2] + (q[2] + oz)];
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (blk[j] >= 0) fn(blk[j]);
        return;
    }
    for (int dx = -1; dx <= 1; dx++) {  // halo >= half a block: up to 27 blocks
        if (!((ok[0] >> (dx + 1)) & 1u)) continue;
        for (int dy = -1; dy <= 1; dy++) {
            if (!((ok[1] >> (dy + 1)) & 1u)) continue;
            for (int dz = -1; dz <= 1; dz++) {
                if (!((ok[2] >> (dz + 1)) & 1u)) continue;
                const int b = table[((q[0] + dx) * d[1] + (q[1] + dy)) * d[2] + (q[2] + dz)];
                if (b >= 0) fn(b);
            }
        }
    }
}

__global__ void __launch_bounds__(VX_BLOCK) k_vx_minmax(const float* xyz, int64_t n, const int* seg_off, const VxState* st,
                                                        const int* table, VxParams p, unsigned* blk_lo, unsigned* blk_hi) {
    __shared__ unsigned slo[VX_LDS_BLOCKS * 3], shi[VX_LDS_BLOCKS * 3];
    int d[3];
    if ((st->overflow & 16u) || !vx_dims(st, d)) return;
    const int seg = blockIdx.y;
    const int* tab = table + (int64_t)seg * (d[0] * d[1] * d[2]);
    const int64_t i0 = seg_off ? seg_off[seg] : 0, i1 = seg_off ? seg_off[seg + 1] : n;
    const int b0 = (int)st->seg_blk_off[seg], nb = (int)st->seg_blk_off[seg + 1] - b0;  // this cloud's blocks
    const bool use_lds = nb <= VX_LDS_BLOCKS;
    if (use_lds)
        for (int i = threadIdx.x; i < nb * 3; i += blockDim.x) { slo[i] = 0xffffffffu; shi[i] = 0u; }
    __syncthreads();
    vx_stream_points(xyz, n, i0, i1, [&](int64_t, float x, float y, float z) {
        const float pt[3] = {x, y, z};
        vx_for_each_block(pt, st, d, tab, p, [&](int b) {
            for (int a = 0; a < 3; a++) {
                unsigned o = st_f2ord(pt[a]);
                if (use_lds) {
                    if (o < slo[3 * (b - b0) + a]) atomicMin(&slo[3 * (b - b0) + a], o);
                    if (o > shi[3 * (b - b0) + a]) atomicMax(&shi[3 * (b - b0) + a], o);
                } else {
                    atomicMin(&blk_lo[3 * b + a], o);
                    atomicMax(&blk_hi[3 * b + a], o);
                }
            }
        });
    });
    __syncthreads();
    if (use_lds)
        for (int i = threadIdx.x; i < nb * 3; i += blockDim.x) {
            if (slo[i] != 0xffffffffu) atomicMin(&blk_lo[3 * b0 + i], slo[i]);
            if (shi[i] != 0u) atomicMax(&blk_hi[3 * b0 + i], shi[i]);
        }
}

// Per block, once (instead of per point and block): the voxel origin as a float and the grid size roundf((hi - lo) / v).
__global__ void __launch_bounds__(VX_BLOCK) k_vx_block_grid(VxState* st, int max_blocks, const unsigned* blk_lo,
                                                            const unsigned* blk_hi, float vs, float* blk_lof, int* blk_grid) {
    const int nb = (int)st_min<uint32_t>(st->n_blocks, (uint32_t)max_blocks);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * nb; i += gridDim.x * blockDim.x) {
        const float lo = st_ord2f(blk_lo[i]), hi = st_ord2f(blk_hi[i]);
        blk_lof[i] = lo;
        const float cells = roundf((hi - lo) / vs);
        // st_pack_key gives z / y / x 16 bits each: a grid beyond that (whole-cloud mode: 1 mm voxels over 65 m) would alias keys
        // and merge distinct voxels silently -- flagged, the host call fails with a clear message
        if (!(cells <= 65535.0f)) atomicOr(&st->overflow, 8u);
        blk_grid[i] = cells <= 65535.0f ? (int)cells : 65535;
    }
}

// (int)floorf(q / v), bit for bit, mostly without the division: t = q * fl(1/v) differs from the correctly rounded
// quotient by < 2e-7 |t| (three roundings), so unless an integer lies within 1e-6 |t| of t both have the same floor; the
// rare lane that is that close to an integer evaluates the exact expression.
__device__ __forceinline__ int vx_floor_div(float q, float v, float inv_v) {
    const float t = q * inv_v;
    const float f = floorf(t);
    const float frac = t - f, tol = fabsf(t) * 1e-6f + 1e-30f;
    if (frac < tol || 1.0f - frac < tol) return (int)floorf(q / v);
    return (int)f;
}

// voxel coordinate of p inside block b: floorf((p - lo) / v), valid iff 0 <= c < roundf((hi - lo) / v)
__device__ __forceinline__ bool vx_coord(const float* pt, int b, const float* blk_lof, const int* blk_grid, float vs, float inv_vs,
                                         int* c) {
    bool ok = true;
    for (int a = 0; a < 3; a++) {
        c[a] = vx_floor_div(pt[a] - blk_lof[3 * b + a], vs, inv_vs);
        ok = ok && c[a] >= 0 && c[a] < blk_grid[3 * b + a];
    }
    return ok;
}

// The voxeliser's own hash table keeps key and value in ONE 16-byte slot: a probe is one random cache line instead of two
// (the passes are bound by exactly those line fetches: ~2.6 block memberships per point, every one a look-up in a table of
// a hundred megabytes).  Slot order and probe sequence are the shared ones (st_hash_slot / st_hash_next).
struct __attribute__((aligned(16))) VxSlot {
    unsigned long long key;
    unsigned val;
    unsigned pad;
};
struct __attribute__((aligned(16))) VxRaw { unsigned long long x, y; };  // one 16-byte load of a slot: key | (val, pad)
// returns 0: table full / gave up; 1: an earlier point already holds the voxel (this point can never win it); 2: this point
// raced for the voxel with an atomic -- only such points can be the final winner, so pass 1 looks up nothing else
__device__ __forceinline__ int vx_insert_min(VxSlot* slots, unsigned long long cap, unsigned long long key, unsigned val,
                                             const unsigned* give_up) {
    unsigned long long slot = st_hash_slot(key, cap);
    for (unsigned long long probe = 0; probe < cap && probe < ST_HASH_MAX_PROBE; probe++) {
        if ((probe & 63ull) == 63ull && (__hip_atomic_load(give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u)) return 0;
        // look before touching the slot with atomics (~9 points per voxel: most find an earlier winner).  A key never changes
        // once written and a value only decreases, so a stale read can only send us down the atomic path needlessly.
        const VxRaw seen = *reinterpret_cast<const VxRaw*>(&slots[slot]);
        unsigned long long prev = seen.x;
        if (prev == ST_EMPTY_KEY) prev = atomicCAS(&slots[slot].key, (unsigned long long)ST_EMPTY_KEY, key);
        if (prev == ST_EMPTY_KEY || prev == key) {
            if (seen.x != key || (unsigned)(seen.y & 0xffffffffull) > val) { atomicMin(&slots[slot].val, val); return 2; }
            return 1;
        }
        slot = st_hash_next(slot, key, cap);
    }
    return 0;
}
__device__ __forceinline__ int vx_find(const VxSlot* slots, unsigned long long cap, unsigned long long key) {
    unsigned long long slot = st_hash_slot(key, cap);
    for (unsigned long long probe = 0; probe < cap && probe < ST_HASH_MAX_PROBE; probe++) {
        const VxRaw seen = *reinterpret_cast<const VxRaw*>(&slots[slot]);
        if (seen.x == key) return (int)(unsigned)(seen.y & 0xffffffffull);
        if (seen.x == ST_EMPTY_KEY) return -1;
        slot = st_hash_next(slot, key, cap);
    }
    return -1;
}

// Insert pass: key (block, voxel) -> min point index ("first point wins", deterministic), one look-up per block membership.
// The voxels are then read off the TABLE, not off the points (round 3): every occupied slot is one voxel and holds its
// representative, so k_vx_emit streams the slots once (16 bytes each, coalesced) instead of sending every point that raced for
// a voxel back into the table (one more random probe for a quarter of the 2.6 memberships per point), scanning a count per point
// and walking the points a third time.  Emission order is the table's; the (block, representative) order of a sequential
// voxeliser comes from the two radix sorts in vx_voxelize.
__global__ void __launch_bounds__(VX_BLOCK) k_vx_insert(const float* xyz, int64_t n, const int* seg_off, VxState* st,
                                                        const int* table, VxParams p, const float* blk_lof,
                                                        const int* blk_grid, VxSlot* slots, unsigned long long cap) {
    int d[3];
    if ((st->overflow & 16u) || !vx_dims(st, d)) return;
    const int seg = blockIdx.y;
    const int* tab = table + (int64_t)seg * (d[0] * d[1] * d[2]);
    const int64_t i0 = seg_off ? seg_off[seg] : 0, i1 = seg_off ? seg_off[seg + 1] : n;
    // a table that filled up (the host's capacity guess was too small: it retries with a larger one): stop inserting as soon as
    // the flag is seen (checked where a probe sequence gets long, vx_insert_min)
    // ONE point per lane and step (the other streaming passes take four: three 16-byte loads): a point's memberships are a
    // chain of dependent random accesses (slot look-up, compare-and-swap, minimum: ~4 us), and four points per lane in a row
    // left half the chip's lanes empty while every busy lane walked four such chains (131 -> 100 us per 1M points; what is left
    // is the pass's ~1M device-scope atomics, ~3 contended look-ups per voxel: cutting the cloud into four launches so that
    // later points find settled voxels costs more in launches (4 x 27-40 us) than it saves)
    for (int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += (int64_t)gridDim.x * blockDim.x) {
        const float pt[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        vx_for_each_block(pt, st, d, tab, p, [&](int b) {
            int c[3];
            if (!vx_coord(pt, b, blk_lof, blk_grid, p.vs, p.vs_inv, c)) return;
            if (vx_insert_min(slots, cap, st_pack_key(b, c[2], c[1], c[0]), (unsigned)i, &st->overflow) == 0) atomicOr(&st->overflow, 4u);
        });
    }
}

// every occupied slot -> one record (representative point, block).  A workgroup stages the records of a tile of VX_EMIT_TILE
// slots in LDS and reserves their place with ONE returning atomic on the voxel counter: returning atomics on one word retire at
// ~90 per microsecond chip-wide, so a reservation per wavefront (262 k of them for a 16 M-slot table) made this pass 3 ms long.
#define VX_EMIT_TILE 4096
__global__ void __launch_bounds__(VX_BLOCK) k_vx_emit(const VxSlot* __restrict__ slots, unsigned long long cap, VxState* st,
                                                      uint32_t* __restrict__ rec_b, uint32_t* __restrict__ rec_pt, int64_t max_voxels) {
    __shared__ uint32_t l_pt[VX_EMIT_TILE], l_b[VX_EMIT_TILE];
    __shared__ unsigned l_n, l_base;
    const int lane = threadIdx.x & 63;
    const unsigned long long ntile = (cap + VX_EMIT_TILE - 1) / VX_EMIT_TILE;
    for (unsigned long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        if (threadIdx.x == 0) l_n = 0u;
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < VX_EMIT_TILE / VX_BLOCK; k++) {
            const unsigned long long s = tile * VX_EMIT_TILE + (unsigned long long)k * VX_BLOCK + threadIdx.x;
            VxRaw seen = {ST_EMPTY_KEY, 0ull};
            if (s < cap) seen = *reinterpret_cast<const VxRaw*>(&slots[s]);
            const bool occ = seen.x != ST_EMPTY_KEY;
            const unsigned long long ob = __ballot(occ);
            if (ob == 0ull) continue;  // (wave-uniform)
            unsigned at = 0;
            if (lane == 0) at = atomicAdd(&l_n, (unsigned)__popcll(ob));
            at = __shfl(at, 0) + (unsigned)__popcll(ob & ((1ull << lane) - 1ull));
            if (occ) { l_pt[at] = (uint32_t)(seen.y & 0xffffffffull); l_b[at] = (uint32_t)(seen.x >> 48); }
        }
        __syncthreads();
        const unsigned cnt = l_n;
        if (threadIdx.x == 0 && cnt) l_base = atomicAdd(&st->n_vox, cnt);
        __syncthreads();
        for (unsigned i = threadIdx.x; i < cnt; i += VX_BLOCK)
            if ((int64_t)l_base + i < max_voxels) { rec_pt[l_base + i] = l_pt[i]; rec_b[l_base + i] = l_b[i]; }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(VX_BLOCK) k_vx_iota(uint32_t* v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        v[i] = (uint32_t)i;
}

// blk_seg (batched calls): cloud of every block -> the first voxel of every cloud lands in st->seg_vox_off
__global__ void __launch_bounds__(VX_BLOCK) k_vx_gather(const float* xyz, const float* rgb, int64_t m, const uint32_t* sorted_b,
                                                        const uint32_t* order, const uint32_t* rec_pt, const float* centres,
                                                        VxParams p, const float* blk_lof, const int* blk_grid,
                                                        float* feats, int32_t* coords, uint8_t* mask, int64_t* point_index,
                                                        const int32_t* blk_seg, VxState* st) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
        int b = (int)sorted_b[j];
        int64_t i = order ? rec_pt[order[j]] : rec_pt[j];
        float pt[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        int c[3];
        vx_coord(pt, b, blk_lof, blk_grid, p.vs, p.vs_inv, c);
        bool inner = true;
        for (int a = 0; a < 3; a++) {
            float ctr = centres[3 * b + a];
            feats[6 * j + a] = pt[a];
            feats[6 * j + 3 + a] = rgb ? rgb[3 * i + a] : 0.0f;
            inner = inner && pt[a] >= ctr - p.half_inner && pt[a] < ctr + p.half_inner;
        }
        if (p.whole) inner = true;  // dataset.py:131: loss_mask = ones
        coords[4 * j] = b;
        coords[4 * j + 1] = c[2];
        coords[4 * j + 2] = c[1];
        coords[4 * j + 3] = c[0];
        mask[j] = inner ? 1 : 0;
        point_index[j] = i;
        if (blk_seg) {  // cloud boundaries: voxel j opens every cloud after its predecessor's, up to its own
            const int s = blk_seg[b], prev = j > 0 ? blk_seg[sorted_b[j - 1]] : -1;
            for (int t = prev + 1; t <= s; t++) st->seg_vox_off[t] = (uint32_t)j;
        }
    }
}
__global__ void k_vx_seg_vox_init(VxState* st, int nseg, const uint32_t* n_vox) {
    for (int t = threadIdx.x; t <= nseg; t += blockDim.x) st->seg_vox_off[t] = *n_vox;  // clouds after the last voxel (and the end)
}
__global__ void k_vx_seg_out(const VxState* st, int nseg, int32_t* seg_vox_off, int32_t* seg_blk_off) {
    for (int t = threadIdx.x; t <= nseg; t += blockDim.x) { seg_vox_off[t] = (int32_t)st->seg_vox_off[t]; seg_blk_off[t] = (int32_t)st->seg_blk_off[t]; }
}

static inline unsigned vx_grid(int64_t n) {
    int64_t g = st_div_up(n > 0 ? n : 1, VX_BLOCK);
    return (unsigned)(g < 4096 ? g : 4096);
}

// streaming kernels: four points per lane and step, blockIdx.y = cloud; `cap` bounds the workgroups per cloud (the
// reductions flush a few words per workgroup with global atomics: fewer, longer-running workgroups)
static inline dim3 vx_grid_seg(int64_t n, int nseg, int64_t cap) {
    int64_t per = st_div_up(n > 0 ? n : 1, nseg);
    int64_t g = st_div_up(st_div_up(per, 4), VX_BLOCK) + 1;  // + 1: the range of a cloud starts at its offset rounded down to 4
    if (g > cap) g = cap;
    return dim3((unsigned)g, (unsigned)nseg);
}

static int64_t vx_layout(StArena& a, int64_t n, int max_blocks, int64_t max_voxels, int nseg, VxState** st, int** table,
                         unsigned** blk_lo, unsigned** blk_hi, float** blk_lof, int** blk_grid, VxSlot** slots, uint32_t** cnt, uint32_t** win,
                         uint32_t** rec_b, uint32_t** rec_pt, uint32_t** order, char** sub, int64_t* sub_bytes,
                         int64_t* cap, float** spare_centres = nullptr, int32_t** spare_i32 = nullptr) {
    *cap = st_next_pow2((max_voxels > 8 ? max_voxels : 8) + (max_voxels > 8 ? max_voxels : 8) / 2);  // load <= 2/3 at max_voxels
    *st = a.take<VxState>(1);
    *table = a.take<int>(vx_table_cells(max_blocks, nseg) * nseg);
    *blk_lo = a.take<unsigned>(3 * (int64_t)max_blocks);
    *blk_hi = a.take<unsigned>(3 * (int64_t)max_blocks);
    *blk_lof = a.take<float>(3 * (int64_t)max_blocks);
    *blk_grid = a.take<int>(3 * (int64_t)max_blocks);
    *slots = a.take<VxSlot>(*cap);
    *cnt = a.take<uint32_t>(n);
    *win = a.take<uint32_t>(n);
    *rec_b = a.take<uint32_t>(max_voxels);
    *rec_pt = a.take<uint32_t>(max_voxels);
    *order = a.take<uint32_t>(max_voxels);
    int64_t s1 = st_scan_ws_bytes(n), s2 = st_sort_ws_bytes(max_voxels);
    *sub_bytes = s1 > s2 ? s1 : s2;
    *sub = a.take<char>(*sub_bytes);
    float* sc = a.take<float>(3 * (int64_t)max_blocks);   // st_voxelize_cloud_seg: block centres and per-cloud block tables nobody asked for
    int32_t* si = a.take<int32_t>(max_blocks + 2 * ((int64_t)nseg + 1));
    if (spare_centres) *spare_centres = sc;
    if (spare_i32) *spare_i32 = si;
    return a.used;
}

extern "C" int64_t st_voxelize_workspace_bytes_seg(int64_t n_points, int max_blocks, int64_t max_voxels, int nseg) {
    StArena a(nullptr, 0);
    VxState* st; int* table; unsigned *lo, *hi; float* lof; int* grd; VxSlot* slots;
    uint32_t *cnt, *win, *rb, *rp, *ord; char* sub; int64_t sb, cap;
    return vx_layout(a, n_points, max_blocks, max_voxels, nseg < 1 ? 1 : nseg, &st, &table, &lo, &hi, &lof, &grd, &slots, &cnt,
                     &win, &rb, &rp, &ord, &sub, &sb, &cap);
}
extern "C" int64_t st_voxelize_workspace_bytes(int64_t n_points, int max_blocks, int64_t max_voxels) {
    return st_voxelize_workspace_bytes_seg(n_points, max_blocks, max_voxels, 1);
}

// Batched form: `nseg` clouds in one array (seg_off [nseg + 1], device int32).  Blocks are numbered cloud by cloud, the
// voxels come out cloud by cloud; extra outputs (device, optional for nseg == 1): blk_seg [max_blocks] = cloud of every
// block, seg_vox_off / seg_blk_off [nseg + 1] = first voxel / block of every cloud.  point_index holds positions in the
// batched point array.  The part of cloud s equals what the one-cloud call returns for it (block ids shifted).
static int vx_voxelize(const float* xyz, const float* rgb, int64_t n, const int32_t* seg_off, int nseg, double voxel_size,
                       double block_size, double buffer_size, int min_points, int max_blocks, int64_t max_voxels, float* feats,
                       int32_t* coords, uint8_t* mask, int64_t* point_index, float* block_centres, int32_t* blk_seg,
                       int32_t* seg_vox_off, int32_t* seg_blk_off, int64_t* n_voxels_out, int64_t* n_blocks_out, void* ws,
                       int64_t ws_bytes, void* stream_, int whole) {
    hipStream_t stream = (hipStream_t)stream_;
    ST_REQUIRE(n >= 0 && n < (1ll << 31), "voxelize: n_points out of range");
    ST_REQUIRE(voxel_size > 0 && block_size > 0 && buffer_size >= 0 && buffer_size < block_size,
               "voxelize: need voxel_size > 0, 0 <= buffer_size < block_size");
    ST_REQUIRE(max_blocks > 0 && max_blocks < 65536 && max_voxels > 0, "voxelize: bad capacities");
    ST_REQUIRE(nseg >= 1 && nseg <= ST_MAX_SEG, "voxelize: 1 <= clouds per batch <= %d", ST_MAX_SEG);
    ST_REQUIRE(nseg == 1 || (seg_off && seg_vox_off && (whole || (blk_seg && seg_blk_off))), "voxelize: a batch needs seg_off and the per-cloud outputs");
    if (nseg == 1) seg_off = nullptr;
    *n_voxels_out = 0;
    *n_blocks_out = 0;
    if (n == 0) {
        if (seg_vox_off) (void)hipMemsetAsync(seg_vox_off, 0, (nseg + 1) * sizeof(int32_t), stream);
        if (seg_blk_off) (void)hipMemsetAsync(seg_blk_off, 0, (nseg + 1) * sizeof(int32_t), stream);
        return ST_OK;
    }

    StArena a(ws, ws_bytes);
    VxState* st; int* table; unsigned *blk_lo, *blk_hi; float* blk_lof; int* blk_grid; VxSlot* slots;
    uint32_t *cnt, *win, *rec_b, *rec_pt, *order; char* sub; int64_t sub_bytes, cap;
    float* spare_centres = nullptr; int32_t* spare_i32 = nullptr;
    vx_layout(a, n, max_blocks, max_voxels, nseg, &st, &table, &blk_lo, &blk_hi, &blk_lof, &blk_grid, &slots, &cnt, &win, &rec_b,
              &rec_pt, &order, &sub, &sub_bytes, &cap, &spare_centres, &spare_i32);
    if (!a.ok() || !sub || !spare_centres || !spare_i32) {
        st_set_error("voxelize: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    if (!block_centres) block_centres = spare_centres;
    if (nseg > 1 && !blk_seg) blk_seg = spare_i32;
    if (nseg > 1 && !seg_blk_off) seg_blk_off = spare_i32 + max_blocks;
    VxParams p;
    p.whole = whole;
    p.bs = (float)block_size;
    {
        int e = 0;
        const float mant = frexpf(p.bs, &e);
        p.bs_inv = mant == 0.5f ? 1.0f / p.bs : 0.0f;  // power of two: multiplying by the reciprocal IS the division
    }
    p.half_outer = (float)((block_size + buffer_size * 2) / 2);
    p.half_inner = (float)(block_size / 2);
    p.bs_half = (float)(block_size / 2);
    p.vs = (float)voxel_size;
    p.vs_inv = 1.0f / p.vs;
    p.min_points = min_points;
    p.max_blocks = max_blocks;
    p.nseg = nseg;

    const dim3 gr = vx_grid_seg(n, nseg, 512 / (nseg < 8 ? nseg : 8) + 8);
    hipLaunchKernelGGL(k_vx_init, dim3(1), dim3(64), 0, stream, st, (int)vx_table_cells(max_blocks, nseg));
    (void)hipMemsetAsync(table, 0, vx_table_cells(max_blocks, nseg) * nseg * sizeof(int), stream);
    (void)hipMemsetAsync(slots, 0xff, cap * sizeof(VxSlot), stream);  // empty key, value = "no point yet"
    hipLaunchKernelGGL(k_vx_bbox, gr, dim3(VX_BLOCK), 0, stream, xyz, n, seg_off, p, st);
    hipLaunchKernelGGL(k_vx_hist, gr, dim3(VX_BLOCK), 0, stream, xyz, n, seg_off, p, st, table);
    hipLaunchKernelGGL(k_vx_blocks, dim3(1), dim3(VX_BLOCK), 0, stream, st, table, p, block_centres, blk_lo, blk_hi, blk_seg);
    hipLaunchKernelGGL(k_vx_minmax, gr, dim3(VX_BLOCK), 0, stream, xyz, n, seg_off, (const VxState*)st, (const int*)table, p, blk_lo,
                       blk_hi);
    hipLaunchKernelGGL(k_vx_block_grid, dim3((unsigned)st_min64(st_div_up(3 * (int64_t)max_blocks, VX_BLOCK), 64)), dim3(VX_BLOCK), 0, stream,
                       st, max_blocks, (const unsigned*)blk_lo, (const unsigned*)blk_hi, p.vs, blk_lof, blk_grid);
    {
        const int64_t per = st_div_up(n > 0 ? n : 1, nseg);
        // (every point in flight at once is the fastest: 95 us; 2048 / 1024 / 512 workgroups per cloud: 112 / 127 / 177 us)
        const dim3 gi((unsigned)st_min64(st_div_up(per, VX_BLOCK) + 1, 8192), (unsigned)nseg);
        hipLaunchKernelGGL(k_vx_insert, gi, dim3(VX_BLOCK), 0, stream, xyz, n, seg_off, st, (const int*)table, p, (const float*)blk_lof,
                           (const int*)blk_grid, slots, (unsigned long long)cap);
    }
    hipLaunchKernelGGL(k_vx_emit, dim3((unsigned)st_min64(st_div_up(cap, VX_EMIT_TILE), 4096)), dim3(VX_BLOCK), 0, stream,
                       (const VxSlot*)slots, (unsigned long long)cap, st, rec_b, rec_pt, max_voxels);
    if (nseg > 1) hipLaunchKernelGGL(k_vx_seg_vox_init, dim3(1), dim3(128), 0, stream, st, nseg, (const uint32_t*)&st->n_vox);
    ST_CHECK_LAUNCH();

    VxState h;
    (void)hipMemcpyAsync(&h, st, sizeof(VxState), hipMemcpyDeviceToHost, stream);
    st_stream_wait(stream);
    ST_CHECK_LAUNCH();
    ST_REQUIRE(!(h.overflow & 16u), "voxelize: the cloud holds non-finite coordinates (NaN or infinity)");
    ST_REQUIRE(!(h.overflow & 1u), "voxelize: the block-id bounding box of a cloud has more than %lld cells (raise max_blocks)",
               (long long)vx_table_cells(max_blocks, nseg));
    ST_REQUIRE(!(h.overflow & 2u), "voxelize: %u blocks exceed max_blocks=%d", h.n_blocks, max_blocks);
    ST_REQUIRE(!(h.overflow & 8u), "voxelize: a voxel grid spans more than 65535 cells along an axis (voxel size %g): use blocks, or a larger voxel", voxel_size);
    ST_REQUIRE(!(h.overflow & 4u) && (int64_t)h.n_vox <= max_voxels, "voxelize: %u voxels exceed max_voxels=%lld",
               h.n_vox, (long long)max_voxels);
    const int64_t m = h.n_vox;
    *n_voxels_out = m;
    *n_blocks_out = h.n_blocks;
    if (m > 0) {
        // the order of a sequential voxeliser: by block, inside a block by representative point index (ascending = order of first
        // appearance).  Two stable radix sorts of the (representative, block) records: by point index, then by block id.
        int bits = 1, pbits = 1;
        while ((1u << bits) < h.n_blocks) bits++;
        while ((1ll << pbits) < n) pbits++;
        ST_TRY(st_radix_sort_pairs_u32(rec_pt, rec_b, m, pbits, sub, sub_bytes, stream));
        ST_TRY(st_radix_sort_pairs_u32(rec_b, rec_pt, m, bits, sub, sub_bytes, stream));
        hipLaunchKernelGGL(k_vx_gather, dim3(vx_grid(m)), dim3(VX_BLOCK), 0, stream, xyz, rgb, m, (const uint32_t*)rec_b,
                           (const uint32_t*)nullptr, (const uint32_t*)rec_pt, (const float*)block_centres, p,
                           (const float*)blk_lof, (const int*)blk_grid, feats, coords, mask, point_index,
                           (const int32_t*)(nseg > 1 ? blk_seg : nullptr), st);
    }
    if (nseg > 1) hipLaunchKernelGGL(k_vx_seg_out, dim3(1), dim3(128), 0, stream, (const VxState*)st, nseg, seg_vox_off, seg_blk_off);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

extern "C" int st_voxelize_blocks_seg(const float* xyz, const float* rgb, int64_t n, const int32_t* seg_off, int nseg,
                                      double voxel_size, double block_size, double buffer_size, int min_points, int max_blocks,
                                      int64_t max_voxels, float* feats, int32_t* coords, uint8_t* mask, int64_t* point_index,
                                      float* block_centres, int32_t* blk_seg, int32_t* seg_vox_off, int32_t* seg_blk_off,
                                      int64_t* n_voxels_out, int64_t* n_blocks_out, void* ws, int64_t ws_bytes, void* stream_) {
    ST_REQUIRE(block_centres != nullptr, "voxelize: block_centres is an output of the blocked call");
    ST_REQUIRE(nseg == 1 || (blk_seg && seg_blk_off), "voxelize: a batch needs the per-cloud block outputs");
    return vx_voxelize(xyz, rgb, n, seg_off, nseg, voxel_size, block_size, buffer_size, min_points, max_blocks, max_voxels, feats,
                       coords, mask, point_index, block_centres, blk_seg, seg_vox_off, seg_blk_off, n_voxels_out, n_blocks_out, ws,
                       ws_bytes, stream_, 0);
}

// Whole-cloud voxelisation, the training / evaluation side's data path: replaces TreeDataset.process_cloud's PointToVoxel call
// (smart_tree/dataset/dataset.py:103-131: range = the cloud's own bounding box, one point per voxel) and batch_collate's
// batch column (model/sparse.py:40-61) for `nseg` clouds at once.  Same kernels as the blocked call with every cloud as
// its own single block: coords[:, 0] = cloud, (z, y, x) = floorf((p - min) / v) inside roundf((max - min) / v) cells, first
// point in input order represents a voxel, voxels in order of first appearance; mask = 1 (loss_mask, dataset.py:131).
// seg_vox_off [nseg + 1] (device, may be null for one cloud) = first voxel of every cloud.
extern "C" int64_t st_voxelize_cloud_workspace_bytes(int64_t n_points, int64_t max_voxels, int nseg) {
    return st_voxelize_workspace_bytes_seg(n_points, nseg < 1 ? 1 : nseg, max_voxels, nseg);
}
extern "C" int st_voxelize_cloud_seg(const float* xyz, const float* rgb, int64_t n, const int32_t* seg_off, int nseg,
                                     double voxel_size, int64_t max_voxels, float* feats, int32_t* coords, uint8_t* mask,
                                     int64_t* point_index, int32_t* seg_vox_off, int64_t* n_voxels_out, void* ws, int64_t ws_bytes,
                                     void* stream_) {
    int64_t n_blocks = 0;
    return vx_voxelize(xyz, rgb, n, seg_off, nseg, voxel_size, 1.0, 0.0, -1, nseg < 1 ? 1 : nseg, max_voxels, feats, coords, mask,
                       point_index, nullptr, nullptr, seg_vox_off, nullptr, n_voxels_out, &n_blocks, ws, ws_bytes, stream_, 1);
}

extern "C" int st_voxelize_blocks(const float* xyz, const float* rgb, int64_t n, double voxel_size, double block_size,
                                  double buffer_size, int min_points, int max_blocks, int64_t max_voxels, float* feats,
                                  int32_t* coords, uint8_t* mask, int64_t* point_index, float* block_centres,
                                  int64_t* n_voxels_out, int64_t* n_blocks_out, void* ws, int64_t ws_bytes,
                                  void* stream_) {
    return st_voxelize_blocks_seg(xyz, rgb, n, nullptr, 1, voxel_size, block_size, buffer_size, min_points, max_blocks, max_voxels,
                                  feats, coords, mask, point_index, block_centres, nullptr, nullptr, nullptr, n_voxels_out,
                                  n_blocks_out, ws, ws_bytes, stream_);
}
