// Rulebooks (neighbour tables) for the sparse convolutions, built on the GPU.
//
// Replaces what spconv builds inside every conv call of the reference network
// (smart_tree/model/model_blocks.py:135-142 constructs all 14 SubMConv3d with indice_key=None, so
// spconv rebuilds the same 27-offset rulebook 14 times; the strided pairs are stored under
// indice_key 1..3 by SparseConv3d, :58-67, and re-used by SparseInverseConv3d, :91-98).
// Here each level's tables are built once and shared by every conv of that level.
//
// All tables are OUTPUT-stationary: nbr[k * n_out + o] = input row feeding output row o through
// kernel offset k (k = (kz*3+ky)*3+kx), or -1.  Orientation follows the oracle
// (oracle/unet_oracle.py): subm  in = out + (k-1);  strided (k3,s2,p1)  in = 2*out - 1 + k;
// inverse: the strided pairs with roles swapped, same k.
// The strided output set is emitted in canonical first-appearance order (inputs in row order,
// k ascending) via hash insert with atomicMin(candidate id) + ordered compaction -- no sort, no
// atomics-order dependence, bit-reproducible.
#include "st_common.h"
#include "st_grid.h"  // ST_MAX_SEG

#define RB_BLOCK 256

static inline unsigned rb_grid(int64_t n) {
    int64_t g = st_div_up(n > 0 ? n : 1, RB_BLOCK);
    return (unsigned)(g < 8192 ? g : 8192);
}

__global__ void __launch_bounds__(RB_BLOCK) k_rb_insert(const int32_t* coords, int64_t n, unsigned long long* keys,
                                                        unsigned* vals, unsigned long long cap, unsigned* fail) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long key = st_pack_key(coords[4 * i], coords[4 * i + 1], coords[4 * i + 2], coords[4 * i + 3]);
        if (!st_hash_insert_min(keys, vals, cap, key, (unsigned)i) && fail) atomicOr(fail, 1u);
    }
}

__global__ void __launch_bounds__(RB_BLOCK) k_rb_subm(const int32_t* coords, int64_t n, const unsigned long long* keys,
                                                      const unsigned* vals, unsigned long long cap, int32_t* nbr) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (int64_t)gridDim.x * blockDim.x) {
        const int b = coords[4 * o], z = coords[4 * o + 1], y = coords[4 * o + 2], x = coords[4 * o + 3];
#pragma unroll
        for (int dz = -1; dz <= 1; dz++) {  // one z-plane of nine neighbours per batch of overlapped look-ups (all 27 at
            unsigned long long key[9];      // once was measured and is slower: register pressure)
            bool valid[9];
            int r[9];
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const int zz = z + dz, yy = y + j / 3 - 1, xx = x + j % 3 - 1;
                valid[j] = !(dz == 0 && j == 4) && zz >= 0 && yy >= 0 && xx >= 0 && zz < 65535 && yy < 65535 && xx < 65535;
                key[j] = st_pack_key(b, zz, yy, xx);
            }
            st_hash_find_batch<9>(keys, vals, cap, key, valid, r);
#pragma unroll
            for (int j = 0; j < 9; j++) nbr[(int64_t)((dz + 1) * 9 + j) * n + o] = (dz == 0 && j == 4) ? (int)o : r[j];
        }
    }
}

// The spatial extent that clips the strided output set is the extent of the voxel's own CLOUD: a one-cloud call has one
// extent for the whole batch of blocks (DESIGN.md "canonical choices"); a batched call keeps one per cloud (blk_seg maps
// the block index coords[:,0] to its cloud), so a cloud's coarse sets do not depend on what else is in the batch.
struct RbState {
    int ext[ST_MAX_SEG][3];  // max coordinate per axis (z,y,x) over the blocks of each cloud
    uint32_t n_out;
    uint32_t fail;
};

__global__ void k_rb_state_init(RbState* st) {
    for (int i = threadIdx.x; i < ST_MAX_SEG * 3; i += blockDim.x) (&st->ext[0][0])[i] = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->n_out = 0; st->fail = 0; }
}

__global__ void __launch_bounds__(RB_BLOCK) k_rb_extent(const int32_t* coords, int64_t n, RbState* st, const int32_t* blk_seg,
                                                        int nseg) {
    __shared__ int m[ST_MAX_SEG * 3];
    for (int i = threadIdx.x; i < nseg * 3; i += blockDim.x) m[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = blk_seg ? blk_seg[coords[4 * i]] : 0;
        for (int a = 0; a < 3; a++) {
            int c = coords[4 * i + 1 + a];
            if (c > m[3 * s + a]) atomicMax(&m[3 * s + a], c);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nseg * 3; i += blockDim.x) if (m[i]) atomicMax(&(&st->ext[0][0])[i], m[i]);
}

// candidate outputs of input voxel i: o = (c + 1 - k) / 2 per axis when even and inside out_shape
// PASS 0: insert(o -> min candidate id i*27+k); PASS 1: count the candidates that won and remember which (bit k of
// win[i]); PASS 2: emit rows from that mask (no hash access).  Passes 1 and 2 do nothing once pass 0 has flagged a full
// table (the host sees the flag after pass 2 and retries with a larger capacity).
template <int PASS>
__global__ void __launch_bounds__(RB_BLOCK) k_rb_down_pass(const int32_t* coords, int64_t n, RbState* st,
                                                           unsigned long long* keys, unsigned* vals, unsigned long long cap,
                                                           uint32_t* cnt_or_off, uint32_t* win, int32_t* out_coords,
                                                           int64_t max_out, const int32_t* blk_seg) {
    if (PASS != 0 && st->fail) return;
    int oshape[3];
    for (int a = 0; a < 3; a++) oshape[a] = st->ext[0][a] / 2 + 1;  // ((ext+1) - 1)/2 + 1
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t mine = 0, won = PASS == 2 ? win[i] : 0u;
        if (PASS == 2 && won == 0u) continue;
        int b = coords[4 * i], c[3] = {coords[4 * i + 1], coords[4 * i + 2], coords[4 * i + 3]};
        if (blk_seg) { const int* e = st->ext[blk_seg[b]]; for (int a = 0; a < 3; a++) oshape[a] = e[a] / 2 + 1; }
        uint32_t off = PASS == 2 ? cnt_or_off[i] : 0u;
        if (PASS == 1) {  // which of my candidates is the stored winner: nine overlapped look-ups per z-plane
#pragma unroll
            for (int kz = 0; kz < 3; kz++) {
                unsigned long long key[9];
                bool valid[9];
                int r[9];
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    const int nz = c[0] + 1 - kz, ny = c[1] + 1 - j / 3, nx = c[2] + 1 - j % 3;
                    const int oz = nz >> 1, oy = ny >> 1, ox = nx >> 1;
                    valid[j] = !((nz | ny | nx) & 1) && nz >= 0 && ny >= 0 && nx >= 0 && oz < oshape[0] && oy < oshape[1] && ox < oshape[2];
                    key[j] = st_pack_key(b, oz, oy, ox);
                }
                st_hash_find_batch<9>(keys, vals, cap, key, valid, r);
#pragma unroll
                for (int j = 0; j < 9; j++)
                    if (valid[j] && (unsigned)r[j] == (unsigned)(i * 27 + kz * 9 + j)) { mine++; won |= 1u << (kz * 9 + j); }
            }
            cnt_or_off[i] = mine;
            win[i] = won;
            continue;
        }
        int k = 0;
        for (int kz = 0; kz < 3; kz++)
            for (int ky = 0; ky < 3; ky++)
                for (int kx = 0; kx < 3; kx++, k++) {
                    if (PASS == 2 && !((won >> k) & 1u)) continue;
                    int nz = c[0] + 1 - kz, ny = c[1] + 1 - ky, nx = c[2] + 1 - kx;
                    if ((nz | ny | nx) & 1) continue;
                    if (nz < 0 || ny < 0 || nx < 0) continue;
                    int oz = nz >> 1, oy = ny >> 1, ox = nx >> 1;
                    if (oz >= oshape[0] || oy >= oshape[1] || ox >= oshape[2]) continue;
                    if (PASS == 2) {
                        uint32_t row = off + mine;
                        if ((int64_t)row < max_out) {
                            out_coords[4 * row] = b;
                            out_coords[4 * row + 1] = oz;
                            out_coords[4 * row + 2] = oy;
                            out_coords[4 * row + 3] = ox;
                        }
                        mine++;
                        continue;
                    }
                    unsigned long long key = st_pack_key(b, oz, oy, ox);
                    unsigned cand = (unsigned)(i * 27 + k);
                    // ~6 fine voxels offer themselves for every coarse one: look before touching the slot with atomics
                    if (!st_hash_insert_min_dup(keys, vals, cap, key, cand, &st->fail, 1u)) atomicOr(&st->fail, 1u);
                }
    }
}

// after compaction: table value := output row (was: winning candidate id)
__global__ void __launch_bounds__(RB_BLOCK) k_rb_relabel(const int32_t* out_coords, int64_t m, const unsigned long long* keys,
                                                         unsigned* vals, unsigned long long cap) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < m; o += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long key = st_pack_key(out_coords[4 * o], out_coords[4 * o + 1], out_coords[4 * o + 2], out_coords[4 * o + 3]);
        unsigned long long slot = st_hash_slot(key, cap);
        while (keys[slot] != key) slot = st_hash_next(slot, key, cap);
        vals[slot] = (unsigned)o;
    }
}

struct RbShape { int v[3]; };

// parity class of a fine voxel: which of the 27 offsets of the strided pair set can have a partner depends only on it
// (per axis an even coordinate pairs through k = 1, an odd one through k = 0 and k = 2): 1, 2, 4 or 8 live offsets.
__device__ __forceinline__ int rb_parity_class(const int* c) { return ((c[0] & 1) << 2) | ((c[1] & 1) << 1) | (c[2] & 1); }

// Both tables of the strided pair set from ONE round of look-ups: fine voxel i reaches coarse row r through offset k
// (nbr_up[k][i] = r), and that same pair read from the coarse side is nbr_down[k][r] = i -- (k, r) determines the fine
// coordinate 2*o - 1 + k, so every entry of nbr_down has exactly one writer; the rest keeps the -1 it was filled with.
__global__ void __launch_bounds__(RB_BLOCK) k_rb_up_nbr(const int32_t* coords, int64_t n, RbShape osh,
                                                        const unsigned long long* ckeys, const unsigned* cvals,
                                                        unsigned long long ccap, int32_t* nbr, int32_t* nbr_down, int64_t m,
                                                        uint32_t* parity_count, const int32_t* blk_seg, const int32_t* ext_dev) {
    __shared__ uint32_t hist[8];
    if (threadIdx.x < 8) hist[threadIdx.x] = 0;
    __syncthreads();
    int oshape[3] = {osh.v[0], osh.v[1], osh.v[2]};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int b = coords[4 * i], c[3] = {coords[4 * i + 1], coords[4 * i + 2], coords[4 * i + 3]};
        if (blk_seg) { const int32_t* e = ext_dev + 3 * blk_seg[b]; for (int a = 0; a < 3; a++) oshape[a] = e[a] / 2 + 1; }
        if (parity_count) atomicAdd(&hist[rb_parity_class(c)], 1u);
#pragma unroll
        for (int kz = 0; kz < 3; kz++) {  // nine offsets per batch of overlapped look-ups (at most four of them live)
            unsigned long long key[9];
            bool valid[9];
            int r[9];
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const int nz = c[0] + 1 - kz, ny = c[1] + 1 - j / 3, nx = c[2] + 1 - j % 3;
                const int oz = nz >> 1, oy = ny >> 1, ox = nx >> 1;
                valid[j] = !((nz | ny | nx) & 1) && nz >= 0 && ny >= 0 && nx >= 0 && oz < oshape[0] && oy < oshape[1] && ox < oshape[2];
                key[j] = st_pack_key(b, oz, oy, ox);
            }
            st_hash_find_batch<9>(ckeys, cvals, ccap, key, valid, r);
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const int k = kz * 9 + j;
                nbr[(int64_t)k * n + i] = r[j];
                if (r[j] >= 0 && r[j] < m) nbr_down[(int64_t)k * m + r[j]] = (int32_t)i;
            }
        }
    }
    __syncthreads();
    if (parity_count && threadIdx.x < 8 && hist[threadIdx.x]) atomicAdd(&parity_count[threadIdx.x], hist[threadIdx.x]);
}

// order[p] = fine row | (8 + class) << 28, rows grouped by parity class (class-major; inside a class in whatever order the
// workgroups reach the cursors -- the convolution's result does not depend on it, see st_sparse_conv_fwd; the tag in
// the top four bits tells the conv kernel which offsets can be live for the row).  count[8] from k_rb_up_nbr, cursor[8]
// zeroed.  A workgroup owns a contiguous chunk of rows: it counts its chunk, reserves one range per class with ONE global
// atomic each (a few thousand atomics on eight words instead of one per wavefront and class), then deals the rows out
// through cursors in LDS.
#define RB_ORDER_BLOCKS 512
__global__ void __launch_bounds__(RB_BLOCK) k_rb_parity_order(const int32_t* coords, int64_t n, const uint32_t* count,
                                                              uint32_t* cursor, int32_t* order) {
    __shared__ uint32_t hist[8], gbase[8], lcur[8];
    const int lane = threadIdx.x & 63;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, chunk = (per + RB_BLOCK - 1) / RB_BLOCK * RB_BLOCK;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (threadIdx.x < 8) { hist[threadIdx.x] = 0; lcur[threadIdx.x] = 0; }
    __syncthreads();
    for (int64_t i0 = lo + (threadIdx.x - lane); i0 < hi; i0 += RB_BLOCK) {
        const int64_t i = i0 + lane;
        int cls = -1;
        if (i < hi) { const int c[3] = {coords[4 * i + 1], coords[4 * i + 2], coords[4 * i + 3]}; cls = rb_parity_class(c); }
        for (int q = 0; q < 8; q++) {
            const unsigned long long mask = __ballot(cls == q);
            if (mask != 0ull && lane == __ffsll((long long)mask) - 1) atomicAdd(&hist[q], (uint32_t)__popcll(mask));
        }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        uint32_t base = 0;
        for (int q = 0; q < (int)threadIdx.x; q++) base += count[q];
        gbase[threadIdx.x] = base + (hist[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], hist[threadIdx.x]) : 0u);
    }
    __syncthreads();
    for (int64_t i0 = lo + (threadIdx.x - lane); i0 < hi; i0 += RB_BLOCK) {
        const int64_t i = i0 + lane;
        int cls = -1;
        if (i < hi) { const int c[3] = {coords[4 * i + 1], coords[4 * i + 2], coords[4 * i + 3]}; cls = rb_parity_class(c); }
        for (int q = 0; q < 8; q++) {
            const unsigned long long mask = __ballot(cls == q);
            if (mask == 0ull) continue;
            const int leader = __ffsll((long long)mask) - 1;
            uint32_t at = 0;
            if (lane == leader) at = atomicAdd(&lcur[q], (uint32_t)__popcll(mask));
            at = __shfl(at, leader);
            if (cls == q)
                order[gbase[q] + at + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)((uint32_t)i | ((8u + (uint32_t)q) << 28));
        }
    }
}

extern "C" int64_t st_hash_capacity(int64_t n) { return st_next_pow2(2 * (n > 8 ? n : 8)); }

extern "C" int st_build_coord_hash(const int32_t* coords, int64_t n, unsigned long long* keys, unsigned* vals, int64_t cap,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ST_REQUIRE(cap >= 2 * n && (cap & (cap - 1)) == 0, "coord hash: capacity must be a power of two >= 2n");
    (void)hipMemsetAsync(keys, 0xff, cap * sizeof(unsigned long long), stream);
    (void)hipMemsetAsync(vals, 0xff, cap * sizeof(unsigned), stream);
    if (n > 0) {
        // capacity >= 2n: an insert can never run out of slots, the flag word is only a guard
        hipLaunchKernelGGL(k_rb_insert, dim3(rb_grid(n)), dim3(RB_BLOCK), 0, stream, coords, n, keys, vals,
                           (unsigned long long)cap, (unsigned*)nullptr);
    }
    ST_CHECK_LAUNCH();
    return ST_OK;
}

extern "C" int st_build_subm_rulebook(const int32_t* coords, int64_t n, const unsigned long long* keys, const unsigned* vals,
                                      int64_t cap, int32_t* nbr, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return ST_OK;
    hipLaunchKernelGGL(k_rb_subm, dim3(rb_grid(n)), dim3(RB_BLOCK), 0, stream, coords, n, keys, vals, (unsigned long long)cap,
                       nbr);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

extern "C" int64_t st_strided_workspace_bytes(int64_t n_fine) {
    StArena a(nullptr, 0);
    a.take<RbState>(1);
    a.take<uint32_t>(n_fine);
    a.take<uint32_t>(n_fine);
    a.take<char>(st_scan_ws_bytes(n_fine));
    return a.used;
}

// Phase 1: discover the coarse active set (canonical order), build its hash.  Returns n_out on the host.
// Batched form (blk_seg [n_blocks], nseg): one extent per cloud; ext_dev [nseg * 3] (device, out) is what phase 2 reads.
extern "C" int st_build_strided_outputs_seg(const int32_t* coords, int64_t n, int64_t max_out, int32_t* out_coords,
                                        unsigned long long* ckeys, unsigned* cvals, int64_t ccap, int64_t* n_out_host,
                                        int32_t* extent_host /*[3] max (z,y,x) coordinate (cloud 0)*/, const int32_t* blk_seg,
                                        int nseg, int32_t* ext_dev, void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    *n_out_host = 0;
    extent_host[0] = extent_host[1] = extent_host[2] = 0;
    ST_REQUIRE(nseg >= 1 && nseg <= ST_MAX_SEG && (nseg == 1 || (blk_seg && ext_dev)), "strided: 1 <= clouds per batch <= %d", ST_MAX_SEG);
    if (nseg == 1) blk_seg = nullptr;
    ST_REQUIRE(ccap >= 2 * max_out && (ccap & (ccap - 1)) == 0, "strided: coarse hash capacity must be a power of two >= 2*max_out");
    ST_REQUIRE(n < (1ll << 31) / 27, "strided: too many voxels for 32-bit candidate ids");
    StArena a(ws, ws_bytes);
    RbState* st = a.take<RbState>(1);
    uint32_t* cnt = a.take<uint32_t>(n);
    uint32_t* win = a.take<uint32_t>(n);
    int64_t scan_bytes = st_scan_ws_bytes(n);
    char* scan_ws = a.take<char>(scan_bytes);
    if (!st || !cnt || !win || !scan_ws) {
        st_set_error("strided: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    (void)hipMemsetAsync(ckeys, 0xff, ccap * sizeof(unsigned long long), stream);
    (void)hipMemsetAsync(cvals, 0xff, ccap * sizeof(unsigned), stream);
    hipLaunchKernelGGL(k_rb_state_init, dim3(1), dim3(64), 0, stream, st);
    if (n == 0) {
        if (ext_dev) (void)hipMemsetAsync(ext_dev, 0, 3 * nseg * sizeof(int32_t), stream);
        return ST_OK;
    }
    const unsigned g = rb_grid(n);
    hipLaunchKernelGGL(k_rb_extent, dim3(g), dim3(RB_BLOCK), 0, stream, coords, n, st, blk_seg, nseg);
    hipLaunchKernelGGL((k_rb_down_pass<0>), dim3(g), dim3(RB_BLOCK), 0, stream, coords, n, st, ckeys, cvals,
                       (unsigned long long)ccap, cnt, win, out_coords, max_out, blk_seg);
    hipLaunchKernelGGL((k_rb_down_pass<1>), dim3(g), dim3(RB_BLOCK), 0, stream, coords, n, st, ckeys, cvals,
                       (unsigned long long)ccap, cnt, win, out_coords, max_out, blk_seg);
    ST_TRY(st_exclusive_scan_u32(cnt, cnt, n, &st->n_out, scan_ws, scan_bytes, stream));
    hipLaunchKernelGGL((k_rb_down_pass<2>), dim3(g), dim3(RB_BLOCK), 0, stream, coords, n, st, ckeys, cvals,
                       (unsigned long long)ccap, cnt, win, out_coords, max_out, blk_seg);
    if (ext_dev) (void)hipMemcpyAsync(ext_dev, &st->ext[0][0], 3 * nseg * sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
    struct { uint32_t n_out, fail; } hc;
    int hext[3];
    (void)hipMemcpyAsync(hext, &st->ext[0][0], sizeof(hext), hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(&hc, &st->n_out, sizeof(hc), hipMemcpyDeviceToHost, stream);
    struct { int ext[3]; uint32_t n_out, fail; } h;
    st_stream_wait(stream);  // the ONE read-back of this stage: n_out sizes everything downstream
    ST_CHECK_LAUNCH();
    for (int a = 0; a < 3; a++) h.ext[a] = hext[a];
    h.n_out = hc.n_out; h.fail = hc.fail;
    ST_REQUIRE(!h.fail, "strided: more than max_out=%lld output voxels", (long long)max_out);
    ST_REQUIRE((int64_t)h.n_out <= max_out, "strided: %u output voxels exceed max_out=%lld", h.n_out, (long long)max_out);
    *n_out_host = h.n_out;
    for (int a = 0; a < 3; a++) extent_host[a] = h.ext[a];
    if (h.n_out)
        hipLaunchKernelGGL(k_rb_relabel, dim3(rb_grid(h.n_out)), dim3(RB_BLOCK), 0, stream, (const int32_t*)out_coords,
                           (int64_t)h.n_out, (const unsigned long long*)ckeys, cvals, (unsigned long long)ccap);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

extern "C" int st_build_strided_outputs(const int32_t* coords, int64_t n, int64_t max_out, int32_t* out_coords,
                                        unsigned long long* ckeys, unsigned* cvals, int64_t ccap, int64_t* n_out_host,
                                        int32_t* extent_host /*[3] max (z,y,x) coordinate*/, void* ws, int64_t ws_bytes,
                                        void* stream_) {
    return st_build_strided_outputs_seg(coords, n, max_out, out_coords, ckeys, cvals, ccap, n_out_host, extent_host, nullptr, 1,
                                        nullptr, ws, ws_bytes, stream_);
}

// Phase 2: the two neighbour tables of the pair set: nbr_down [27][n_out] (fine rows), nbr_up [27][n] (coarse rows);
// up_order (optional, n + 16 words: the tail is scratch): the fine rows grouped by coordinate parity and tagged with
// their class, the row order the inverse convolution should be launched with (st_sparse_conv_fwd's row_order).
// Batched form: blk_seg / ext_dev as given to / filled by st_build_strided_outputs_seg (NULL: one cloud, extent_host).
extern "C" int st_build_strided_rulebook_seg(const int32_t* coords, int64_t n, const unsigned long long* fkeys,
                                         const unsigned* fvals, int64_t fcap, const int32_t* out_coords, int64_t n_out,
                                         const unsigned long long* ckeys, const unsigned* cvals, int64_t ccap,
                                         const int32_t* extent_host, int32_t* nbr_down, int32_t* nbr_up,
                                         int32_t* up_order /*[n + 16] or NULL*/, const int32_t* blk_seg, const int32_t* ext_dev,
                                         void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n == 0) return ST_OK;
    ST_REQUIRE((blk_seg == nullptr) == (ext_dev == nullptr), "strided: blk_seg and ext_dev go together");
    ST_REQUIRE(up_order == nullptr || n < (1ll << 28), "strided: the tagged row order holds at most 2^28 rows");
    RbShape osh;
    for (int a = 0; a < 3; a++) osh.v[a] = extent_host[a] / 2 + 1;
    (void)fkeys; (void)fvals; (void)fcap;  // the fine hash is no longer consulted (kept in the signature)
    if (n_out) (void)hipMemsetAsync(nbr_down, 0xff, 27 * n_out * sizeof(int32_t), stream);
    uint32_t* pc = up_order ? reinterpret_cast<uint32_t*>(up_order + n) : nullptr;  // 8 class counts + 8 cursors
    if (pc) (void)hipMemsetAsync(pc, 0, 16 * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(k_rb_up_nbr, dim3(rb_grid(n)), dim3(RB_BLOCK), 0, stream, coords, n, osh, ckeys, cvals,
                       (unsigned long long)ccap, nbr_up, nbr_down, n_out, pc, blk_seg, ext_dev);
    if (pc)
        hipLaunchKernelGGL(k_rb_parity_order, dim3(rb_grid(n) < RB_ORDER_BLOCKS ? rb_grid(n) : RB_ORDER_BLOCKS), dim3(RB_BLOCK), 0, stream, coords,
                           n, (const uint32_t*)pc, pc + 8, up_order);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

extern "C" int st_build_strided_rulebook(const int32_t* coords, int64_t n, const unsigned long long* fkeys,
                                         const unsigned* fvals, int64_t fcap, const int32_t* out_coords, int64_t n_out,
                                         const unsigned long long* ckeys, const unsigned* cvals, int64_t ccap,
                                         const int32_t* extent_host, int32_t* nbr_down, int32_t* nbr_up,
                                         int32_t* up_order /*[n + 16] or NULL*/, void* stream_) {
    return st_build_strided_rulebook_seg(coords, n, fkeys, fvals, fcap, out_coords, n_out, ckeys, cvals, ccap, extent_host,
                                         nbr_down, nbr_up, up_order, nullptr, nullptr, stream_);
}

// ------------------------------------------------------------------------ spatial row order ---
// The network does not care in which order the voxels are stored: every output row is computed on its own, in k order,
// from the rows its neighbour table names.  The order decides speed, though: voxels arrive in "first point of the cloud"
// order (spatially incoherent), so the 27 rows a voxel gathers are scattered over the feature array and the 16 rows of a
// matrix-core tile have little in common (every tile needs nearly all 27 offsets, most of its rows padded with zeros).
// st_spatial_order returns the permutation that sorts the voxels by (batch index, Morton code of z, y, x): the caller runs
// the network on the permuted set -- coarser levels inherit the order, their sets being built in first-appearance order --
// and scatters the outputs back.  Same values, bit for bit (tests/test_unet.py::test_spatial_order_is_invisible).
__device__ __forceinline__ uint32_t rb_spread3(uint32_t v) {  // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ void __launch_bounds__(RB_BLOCK) k_rb_morton_keys(const int32_t* coords, int64_t n, uint32_t* keys, uint32_t* vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = (rb_spread3((uint32_t)coords[4 * i + 1]) << 2) | (rb_spread3((uint32_t)coords[4 * i + 2]) << 1) |
                  rb_spread3((uint32_t)coords[4 * i + 3]);
        vals[i] = (uint32_t)i;
    }
}
__global__ void __launch_bounds__(RB_BLOCK) k_rb_block_keys(const int32_t* coords, int64_t n, const uint32_t* vals, uint32_t* keys) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        keys[i] = (uint32_t)coords[4 * (int64_t)vals[i]];
}

extern "C" int64_t st_spatial_order_workspace_bytes(int64_t n) {
    StArena a(nullptr, 0);
    a.take<uint32_t>(n);
    a.take<char>(st_sort_ws_bytes(n));
    return a.used;
}

// order [n] int32: position p of the spatial order holds input row order[p].  Stable: equal keys keep their input order.
extern "C" int st_spatial_order(const int32_t* coords, int64_t n, int32_t* order, void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return ST_OK;
    ST_REQUIRE(n < (1ll << 31), "spatial_order: too many voxels");
    StArena a(ws, ws_bytes);
    uint32_t* keys = a.take<uint32_t>(n);
    const int64_t sb = st_sort_ws_bytes(n);
    char* sw = a.take<char>(sb);
    if (!keys || !sw) { st_set_error("spatial_order: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used); return ST_ERR_WORKSPACE; }
    uint32_t* vals = reinterpret_cast<uint32_t*>(order);
    hipLaunchKernelGGL(k_rb_morton_keys, dim3(rb_grid(n)), dim3(RB_BLOCK), 0, stream, coords, n, keys, vals);
    ST_TRY(st_radix_sort_pairs_u32(keys, vals, n, 30, sw, sb, stream));
    hipLaunchKernelGGL(k_rb_block_keys, dim3(rb_grid(n)), dim3(RB_BLOCK), 0, stream, coords, n, (const uint32_t*)vals, keys);
    ST_TRY(st_radix_sort_pairs_u32(keys, vals, n, 16, sw, sb, stream));  // stable: Morton order inside every block
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// dst[p] = src[order[p]] (scatter == 0) or dst[order[p]] = src[p] (scatter != 0) for rows of `row_words` 32-bit words:
// the permutation of the voxel rows into / out of the spatial order (torch's index_select needs an int64 index and ran the
// 12- and 16-byte rows at a tenth of the HBM rate).
template <int VEC>
__global__ void __launch_bounds__(RB_BLOCK) k_rb_move_rows(const uint32_t* __restrict__ src, int row_words, const int32_t* __restrict__ order,
                                                           int64_t n, uint32_t* __restrict__ dst, int scatter) {
    const int per_row = row_words / VEC;
    const int64_t total = n * per_row;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = t / per_row;
        const int c = (int)(t % per_row) * VEC;
        const int64_t o = order[p];
        const int64_t from = (scatter ? p : o) * row_words + c, to = (scatter ? o : p) * row_words + c;
        if (VEC == 4) *reinterpret_cast<uint4*>(dst + to) = *reinterpret_cast<const uint4*>(src + from);
        else dst[to] = src[from];
    }
}

extern "C" int st_move_rows(const void* src, int row_words, const int32_t* order, int64_t n, void* dst, int scatter, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return ST_OK;
    ST_REQUIRE(row_words >= 1 && row_words <= 256, "move_rows: rows of 1..256 words");
    const bool vec = row_words % 4 == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    const int64_t total = n * (vec ? row_words / 4 : row_words);
    if (vec)
        hipLaunchKernelGGL((k_rb_move_rows<4>), dim3(rb_grid(total)), dim3(RB_BLOCK), 0, stream, (const uint32_t*)src, row_words, order, n,
                           (uint32_t*)dst, scatter);
    else
        hipLaunchKernelGGL((k_rb_move_rows<1>), dim3(rb_grid(total)), dim3(RB_BLOCK), 0, stream, (const uint32_t*)src, row_words, order, n,
                           (uint32_t*)dst, scatter);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
