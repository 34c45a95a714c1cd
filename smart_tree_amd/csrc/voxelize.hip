// st_voxelize_blocks: 4 m blocks + halo, per-block first-point-wins voxelisation, collate.
//
// Replaces, for the inference path, the reference's host loop
//   SingleTreeInference.compute_blocks   smart_tree/dataset/dataset.py:166-190
//   SingleTreeInference.__getitem__      smart_tree/dataset/dataset.py:192-226 (spconv PointToVoxel, CPU)
//   batch_collate                        smart_tree/model/sparse.py:40-61
// with one device-resident pass over the cloud: block histogram -> kept blocks (count > min,
// lexicographic order) -> per-block bounding boxes over halo members -> hash insert with
// atomicMin(point index) (deterministic "first point wins") -> ordered compaction by
// (block, representative point index), which is exactly the order a sequential voxeliser emits.
// All float arithmetic that decides an integer (block id, membership, voxel coordinate, grid
// size, inner mask) is float32 in the same operation order as the oracle (oracle/voxel_oracle.py).
#include "st_common.h"

#define VX_BLOCK 256
#define VX_TABLE_CAP 32768  // max cells of the block-id bounding box (32^3 blocks of 4 m = 128 m)
#define VX_LDS_BLOCKS 128

struct VxState {
    int lo[3], hi[3];
    uint32_t n_blocks;
    uint32_t n_vox;
    uint32_t overflow;  // bit0: block table, bit1: max_blocks, bit2: hash full
};

struct VxParams {
    float bs;          // block size
    float half_outer;  // (block + 2*buffer)/2 rounded to float
    float half_inner;  // block/2
    float bs_half;     // block/2 (centre offset)
    float vs;          // voxel size
    int min_points;
    int max_blocks;
};

__device__ __forceinline__ int vx_block_id(float v, float bs) { return (int)floorf(v / bs); }

__global__ void k_vx_init(VxState* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int a = 0; a < 3; a++) { st->lo[a] = 0x7fffffff; st->hi[a] = (int)0x80000000; }
        st->n_blocks = 0; st->n_vox = 0; st->overflow = 0;
    }
}

__global__ void __launch_bounds__(VX_BLOCK) k_vx_bbox(const float* xyz, int64_t n, float bs, VxState* st) {
    __shared__ int lo[3], hi[3];
    if (threadIdx.x < 3) { lo[threadIdx.x] = 0x7fffffff; hi[threadIdx.x] = (int)0x80000000; }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int a = 0; a < 3; a++) {
            int q = vx_block_id(xyz[3 * i + a], bs);
            if (q < lo[a]) atomicMin(&lo[a], q);
            if (q > hi[a]) atomicMax(&hi[a], q);
        }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&st->lo[threadIdx.x], lo[threadIdx.x]); atomicMax(&st->hi[threadIdx.x], hi[threadIdx.x]); }
}

__device__ __forceinline__ bool vx_dims(const VxState* st, int* d) {
    int64_t total = 1;
    for (int a = 0; a < 3; a++) { d[a] = st->hi[a] - st->lo[a] + 1; if (d[a] < 1) return false; total *= d[a]; }
    return total <= VX_TABLE_CAP;
}

#define VX_LDS_CELLS 2048
__global__ void __launch_bounds__(VX_BLOCK) k_vx_hist(const float* xyz, int64_t n, float bs, VxState* st, int* table) {
    __shared__ int h[VX_LDS_CELLS];
    int d[3];
    if (!vx_dims(st, d)) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(&st->overflow, 1u); return; }
    // a tree spans a few dozen blocks: count in LDS, flush once per workgroup (global atomics on a
    // handful of words would serialise a million points)
    const int ncell = d[0] * d[1] * d[2];
    const bool use_lds = ncell <= VX_LDS_CELLS;
    if (use_lds) for (int c = threadIdx.x; c < ncell; c += blockDim.x) h[c] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int cx = vx_block_id(xyz[3 * i], bs) - st->lo[0], cy = vx_block_id(xyz[3 * i + 1], bs) - st->lo[1],
            cz = vx_block_id(xyz[3 * i + 2], bs) - st->lo[2];
        const int c = (cx * d[1] + cy) * d[2] + cz;
        if (use_lds) atomicAdd(&h[c], 1); else atomicAdd(&table[c], 1);
    }
    __syncthreads();
    if (use_lds) for (int c = threadIdx.x; c < ncell; c += blockDim.x) if (h[c]) atomicAdd(&table[c], h[c]);
}

// single workgroup: counts -> block rank (or -1), centres, bbox init
__global__ void __launch_bounds__(VX_BLOCK) k_vx_blocks(VxState* st, int* table, VxParams p, float* centres, unsigned* blk_lo,
                                                        unsigned* blk_hi) {
    __shared__ uint32_t lds[VX_BLOCK / 64 + 1];
    int d[3];
    if (!vx_dims(st, d)) return;
    int ncell = d[0] * d[1] * d[2];
    uint32_t carry = 0;
    for (int base = 0; base < ncell; base += VX_BLOCK) {
        int c = base + threadIdx.x;
        uint32_t keep = (c < ncell && table[c] > p.min_points) ? 1u : 0u;
        uint32_t total;
        uint32_t rank = block_exclusive_scan(keep, lds, &total) + carry;
        if (c < ncell) {
            if (keep && (int)rank < p.max_blocks) {
                table[c] = (int)rank;
                int cz = c % d[2], cy = (c / d[2]) % d[1], cx = c / (d[2] * d[1]);
                float id[3] = {(float)(cx + st->lo[0]), (float)(cy + st->lo[1]), (float)(cz + st->lo[2])};
                for (int a = 0; a < 3; a++) {
                    centres[3 * rank + a] = id[a] * p.bs + p.bs_half;
                    blk_lo[3 * rank + a] = 0xffffffffu;
                    blk_hi[3 * rank + a] = 0u;
                }
            } else {
                table[c] = -1;
            }
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        st->n_blocks = carry;
        if ((int)carry > p.max_blocks) atomicOr(&st->overflow, 2u);
    }
}

// Calls fn(b) for every kept block whose halo cube [c - half_outer, c + half_outer) holds p, in (x, y, z) block order.
// The test is separable: per axis, which of the three neighbouring block columns hold the coordinate (the centre is
// recomputed with the expression k_vx_blocks stores, so the decision is bit-identical to testing `centres`); only the
// surviving combinations -- one for an interior point, up to eight near a block corner -- touch the block table.
template <class F>
__device__ __forceinline__ void vx_for_each_block(const float* pt, const VxState* st, const int* d, const int* table,
                                                  const VxParams& p, F fn) {
    int q[3];
    unsigned ok[3];
    for (int a = 0; a < 3; a++) {
        q[a] = vx_block_id(pt[a], p.bs) - st->lo[a];
        ok[a] = 0;
        for (int o = -1; o <= 1; o++) {
            const int c = q[a] + o;
            if (c < 0 || c >= d[a]) continue;
            const float ctr = (float)(c + st->lo[a]) * p.bs + p.bs_half;
            if (pt[a] >= ctr - p.half_outer && pt[a] < ctr + p.half_outer) ok[a] |= 1u << (o + 1);
        }
    }
    if (!(ok[0] && ok[1] && ok[2])) return;
    for (int dx = -1; dx <= 1; dx++) {
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

__global__ void __launch_bounds__(VX_BLOCK) k_vx_minmax(const float* xyz, int64_t n, const VxState* st, const int* table,
                                                        const float* centres, VxParams p, unsigned* blk_lo,
                                                        unsigned* blk_hi) {
    __shared__ unsigned slo[VX_LDS_BLOCKS * 3], shi[VX_LDS_BLOCKS * 3];
    int d[3];
    if (!vx_dims(st, d)) return;
    const int nb = (int)st_min<uint32_t>(st->n_blocks, (uint32_t)p.max_blocks);
    const bool use_lds = nb <= VX_LDS_BLOCKS;
    if (use_lds)
        for (int i = threadIdx.x; i < nb * 3; i += blockDim.x) { slo[i] = 0xffffffffu; shi[i] = 0u; }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pt[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        vx_for_each_block(pt, st, d, table, p, [&](int b) {
            for (int a = 0; a < 3; a++) {
                unsigned o = st_f2ord(pt[a]);
                if (use_lds) {
                    if (o < slo[3 * b + a]) atomicMin(&slo[3 * b + a], o);
                    if (o > shi[3 * b + a]) atomicMax(&shi[3 * b + a], o);
                } else {
                    atomicMin(&blk_lo[3 * b + a], o);
                    atomicMax(&blk_hi[3 * b + a], o);
                }
            }
        });
    }
    __syncthreads();
    if (use_lds)
        for (int i = threadIdx.x; i < nb * 3; i += blockDim.x) {
            if (slo[i] != 0xffffffffu) atomicMin(&blk_lo[i], slo[i]);
            if (shi[i] != 0u) atomicMax(&blk_hi[i], shi[i]);
        }
}

// voxel coordinate of p inside block b: floorf((p - lo) / v), valid iff 0 <= c < roundf((hi - lo) / v)
__device__ __forceinline__ bool vx_coord(const float* pt, int b, const unsigned* blk_lo, const unsigned* blk_hi, float vs,
                                         int* c) {
    bool ok = true;
    for (int a = 0; a < 3; a++) {
        float lo = st_ord2f(blk_lo[3 * b + a]), hi = st_ord2f(blk_hi[3 * b + a]);
        int grid = (int)roundf((hi - lo) / vs);
        c[a] = (int)floorf((pt[a] - lo) / vs);
        ok = ok && c[a] >= 0 && c[a] < grid;
    }
    return ok;
}

// pass 0: insert (key -> min point index); pass 1: count the voxels a point won and remember WHICH of its blocks
// (bit j of win[i] = the j-th block vx_for_each_block visits; at most 27); pass 2: emit winners from that mask -- no
// second round of hash look-ups.
template <int PASS>
__global__ void __launch_bounds__(VX_BLOCK) k_vx_pass(const float* xyz, int64_t n, VxState* st, const int* table,
                                                      VxParams p, const unsigned* blk_lo, const unsigned* blk_hi,
                                                      unsigned long long* keys, unsigned* vals, unsigned long long cap,
                                                      uint32_t* cnt_or_off, uint32_t* win, uint32_t* rec_b,
                                                      uint32_t* rec_pt, int64_t max_voxels) {
    int d[3];
    if (!vx_dims(st, d)) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t mine = 0, won = PASS == 2 ? win[i] : 0u;
        if (PASS == 2 && won == 0u) continue;
        float pt[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        uint32_t off = PASS == 2 ? cnt_or_off[i] : 0u;
        int j = 0;
        vx_for_each_block(pt, st, d, table, p, [&](int b) {
            const uint32_t bit = 1u << j++;
            if (PASS == 2) {
                if (!(won & bit)) return;
                if ((int64_t)(off + mine) < max_voxels) {
                    rec_b[off + mine] = (uint32_t)b;
                    rec_pt[off + mine] = (uint32_t)i;
                }
                mine++;
                return;
            }
            int c[3];
            if (!vx_coord(pt, b, blk_lo, blk_hi, p.vs, c)) return;
            unsigned long long key = st_pack_key(b, c[2], c[1], c[0]);
            if (PASS == 0) {
                if (!st_hash_insert_min_dup(keys, vals, cap, key, (unsigned)i)) atomicOr(&st->overflow, 4u);
            } else if (st_hash_find(keys, vals, cap, key) == (int)i) {
                won |= bit;
                mine++;
            }
        });
        if (PASS == 1) { cnt_or_off[i] = mine; win[i] = won; }
    }
}

__global__ void __launch_bounds__(VX_BLOCK) k_vx_iota(uint32_t* v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        v[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(VX_BLOCK) k_vx_gather(const float* xyz, const float* rgb, int64_t m, const uint32_t* sorted_b,
                                                        const uint32_t* order, const uint32_t* rec_pt, const float* centres,
                                                        VxParams p, const unsigned* blk_lo, const unsigned* blk_hi,
                                                        float* feats, int32_t* coords, uint8_t* mask, int64_t* point_index) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (int64_t)gridDim.x * blockDim.x) {
        int b = (int)sorted_b[j];
        int64_t i = rec_pt[order[j]];
        float pt[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        int c[3];
        vx_coord(pt, b, blk_lo, blk_hi, p.vs, c);
        bool inner = true;
        for (int a = 0; a < 3; a++) {
            float ctr = centres[3 * b + a];
            feats[6 * j + a] = pt[a];
            feats[6 * j + 3 + a] = rgb ? rgb[3 * i + a] : 0.0f;
            inner = inner && pt[a] >= ctr - p.half_inner && pt[a] < ctr + p.half_inner;
        }
        coords[4 * j] = b;
        coords[4 * j + 1] = c[2];
        coords[4 * j + 2] = c[1];
        coords[4 * j + 3] = c[0];
        mask[j] = inner ? 1 : 0;
        point_index[j] = i;
    }
}

static inline unsigned vx_grid(int64_t n) {
    int64_t g = st_div_up(n > 0 ? n : 1, VX_BLOCK);
    return (unsigned)(g < 4096 ? g : 4096);
}

// reductions flush a few words per workgroup with global atomics: fewer, longer-running workgroups
static inline unsigned vx_grid_reduce(int64_t n) {
    const unsigned g = vx_grid(n);
    return g < 512 ? g : 512;
}

static int64_t vx_layout(StArena& a, int64_t n, int max_blocks, int64_t max_voxels, VxState** st, int** table,
                         unsigned** blk_lo, unsigned** blk_hi, unsigned long long** keys, unsigned** vals, uint32_t** cnt, uint32_t** win,
                         uint32_t** rec_b, uint32_t** rec_pt, uint32_t** order, char** sub, int64_t* sub_bytes,
                         int64_t* cap) {
    *cap = st_next_pow2(2 * (max_voxels > 8 ? max_voxels : 8));
    *st = a.take<VxState>(1);
    *table = a.take<int>(VX_TABLE_CAP);
    *blk_lo = a.take<unsigned>(3 * (int64_t)max_blocks);
    *blk_hi = a.take<unsigned>(3 * (int64_t)max_blocks);
    *keys = a.take<unsigned long long>(*cap);
    *vals = a.take<unsigned>(*cap);
    *cnt = a.take<uint32_t>(n);
    *win = a.take<uint32_t>(n);
    *rec_b = a.take<uint32_t>(max_voxels);
    *rec_pt = a.take<uint32_t>(max_voxels);
    *order = a.take<uint32_t>(max_voxels);
    int64_t s1 = st_scan_ws_bytes(n), s2 = st_sort_ws_bytes(max_voxels);
    *sub_bytes = s1 > s2 ? s1 : s2;
    *sub = a.take<char>(*sub_bytes);
    return a.used;
}

extern "C" int64_t st_voxelize_workspace_bytes(int64_t n_points, int max_blocks, int64_t max_voxels) {
    StArena a(nullptr, 0);
    VxState* st; int* table; unsigned *lo, *hi; unsigned long long* keys; unsigned* vals;
    uint32_t *cnt, *win, *rb, *rp, *ord; char* sub; int64_t sb, cap;
    return vx_layout(a, n_points, max_blocks, max_voxels, &st, &table, &lo, &hi, &keys, &vals, &cnt, &win, &rb, &rp, &ord, &sub,
                     &sb, &cap);
}

extern "C" int st_voxelize_blocks(const float* xyz, const float* rgb, int64_t n, double voxel_size, double block_size,
                                  double buffer_size, int min_points, int max_blocks, int64_t max_voxels, float* feats,
                                  int32_t* coords, uint8_t* mask, int64_t* point_index, float* block_centres,
                                  int64_t* n_voxels_out, int64_t* n_blocks_out, void* ws, int64_t ws_bytes,
                                  void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ST_REQUIRE(n >= 0 && n < (1ll << 31), "voxelize: n_points out of range");
    ST_REQUIRE(voxel_size > 0 && block_size > 0 && buffer_size >= 0 && buffer_size < block_size,
               "voxelize: need voxel_size > 0, 0 <= buffer_size < block_size");
    ST_REQUIRE(max_blocks > 0 && max_blocks < 65536 && max_voxels > 0, "voxelize: bad capacities");
    *n_voxels_out = 0;
    *n_blocks_out = 0;
    if (n == 0) return ST_OK;

    StArena a(ws, ws_bytes);
    VxState* st; int* table; unsigned *blk_lo, *blk_hi; unsigned long long* keys; unsigned* vals;
    uint32_t *cnt, *win, *rec_b, *rec_pt, *order; char* sub; int64_t sub_bytes, cap;
    vx_layout(a, n, max_blocks, max_voxels, &st, &table, &blk_lo, &blk_hi, &keys, &vals, &cnt, &win, &rec_b, &rec_pt, &order, &sub,
              &sub_bytes, &cap);
    if (!a.ok() || !sub) {
        st_set_error("voxelize: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    VxParams p;
    p.bs = (float)block_size;
    p.half_outer = (float)((block_size + buffer_size * 2) / 2);
    p.half_inner = (float)(block_size / 2);
    p.bs_half = (float)(block_size / 2);
    p.vs = (float)voxel_size;
    p.min_points = min_points;
    p.max_blocks = max_blocks;

    const unsigned g = vx_grid(n);
    hipLaunchKernelGGL(k_vx_init, dim3(1), dim3(64), 0, stream, st);
    (void)hipMemsetAsync(table, 0, VX_TABLE_CAP * sizeof(int), stream);
    (void)hipMemsetAsync(keys, 0xff, cap * sizeof(unsigned long long), stream);
    (void)hipMemsetAsync(vals, 0xff, cap * sizeof(unsigned), stream);
    hipLaunchKernelGGL(k_vx_bbox, dim3(vx_grid_reduce(n)), dim3(VX_BLOCK), 0, stream, xyz, n, p.bs, st);
    hipLaunchKernelGGL(k_vx_hist, dim3(vx_grid_reduce(n)), dim3(VX_BLOCK), 0, stream, xyz, n, p.bs, st, table);
    hipLaunchKernelGGL(k_vx_blocks, dim3(1), dim3(VX_BLOCK), 0, stream, st, table, p, block_centres, blk_lo, blk_hi);
    hipLaunchKernelGGL(k_vx_minmax, dim3(g), dim3(VX_BLOCK), 0, stream, xyz, n, (const VxState*)st, (const int*)table,
                       (const float*)block_centres, p, blk_lo, blk_hi);
    hipLaunchKernelGGL((k_vx_pass<0>), dim3(g), dim3(VX_BLOCK), 0, stream, xyz, n, st, (const int*)table, p,
                       (const unsigned*)blk_lo, (const unsigned*)blk_hi, keys, vals, (unsigned long long)cap, cnt, win, rec_b,
                       rec_pt, max_voxels);
    hipLaunchKernelGGL((k_vx_pass<1>), dim3(g), dim3(VX_BLOCK), 0, stream, xyz, n, st, (const int*)table, p,
                       (const unsigned*)blk_lo, (const unsigned*)blk_hi, keys, vals, (unsigned long long)cap, cnt, win, rec_b,
                       rec_pt, max_voxels);
    ST_TRY(st_exclusive_scan_u32(cnt, cnt, n, &st->n_vox, sub, sub_bytes, stream));
    hipLaunchKernelGGL((k_vx_pass<2>), dim3(g), dim3(VX_BLOCK), 0, stream, xyz, n, st, (const int*)table, p,
                       (const unsigned*)blk_lo, (const unsigned*)blk_hi, keys, vals, (unsigned long long)cap, cnt, win, rec_b,
                       rec_pt, max_voxels);
    ST_CHECK_LAUNCH();

    VxState h;
    (void)hipMemcpyAsync(&h, st, sizeof(VxState), hipMemcpyDeviceToHost, stream);
    st_stream_wait(stream);
    ST_CHECK_LAUNCH();
    ST_REQUIRE(!(h.overflow & 1u), "voxelize: cloud spans more than %d blocks", VX_TABLE_CAP);
    ST_REQUIRE(!(h.overflow & 2u), "voxelize: %u blocks exceed max_blocks=%d", h.n_blocks, max_blocks);
    ST_REQUIRE(!(h.overflow & 4u) && (int64_t)h.n_vox <= max_voxels, "voxelize: %u voxels exceed max_voxels=%lld",
               h.n_vox, (long long)max_voxels);
    const int64_t m = h.n_vox;
    *n_voxels_out = m;
    *n_blocks_out = h.n_blocks;
    if (m == 0) return ST_OK;

    // stable sort by block id: records are already ascending in point index
    int bits = 1;
    while ((1u << bits) < h.n_blocks) bits++;
    hipLaunchKernelGGL(k_vx_iota, dim3(vx_grid(m)), dim3(VX_BLOCK), 0, stream, order, m);
    ST_TRY(st_radix_sort_pairs_u32(rec_b, order, m, bits, sub, sub_bytes, stream));
    hipLaunchKernelGGL(k_vx_gather, dim3(vx_grid(m)), dim3(VX_BLOCK), 0, stream, xyz, rgb, m, (const uint32_t*)rec_b,
                       (const uint32_t*)order, (const uint32_t*)rec_pt, (const float*)block_centres, p,
                       (const unsigned*)blk_lo, (const unsigned*)blk_hi, feats, coords, mask, point_index);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
