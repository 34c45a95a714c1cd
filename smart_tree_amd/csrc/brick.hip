// Rulebooks without hash probes: occupancy bricks + popcount ranks (round 3).
//
// Replaces, for the network's own forward pass, the hash-table builders of rulebook.hip (which stay the generic entry
// points of the C ABI and the fallback for coordinate ranges this structure is not sized for).  What spconv does inside
// every conv call of the reference (smart_tree/model/model_blocks.py:57-70,90-101,134-143: a GPU hash table + 27 probes per
// voxel, per conv) is done here ONCE per level, and a neighbour look-up is two cache-friendly loads instead of a probe chain:
//
//   * every block (batch index) is cut into 8 x 8 x 8 BRICKS; a dense table per block, indexed by the Morton code of the
//     brick coordinate, says which bricks hold voxels (flag -> exclusive prefix = the brick's slot);
//   * an occupied brick has a 512-bit occupancy mask (eight 64-bit words, one per z), the running popcount of its words
//     and `base` = the number of voxels in the bricks in front of it;
//   * the level's voxels are ORDERED by (block, brick Morton code, z, y, x inside the brick), so
//         row(voxel) = base[brick] + cum[z & 7] + popcount(mask[z & 7] below bit (y & 7) * 8 + (x & 7))
//     -- "is there a voxel at (b, z, y, x), and which row is it" needs the brick's table entry and one mask word; the 27
//     neighbours of a voxel sit in at most eight bricks, and neighbouring rows (adjacent lanes) read the same lines.
//   * the same order is the spatially coherent one the convolutions want (st_spatial_order sorted by Morton code for that:
//     two radix sorts, 26 launches): here it falls out of the structure -- order0[] is computed, not sorted.
//   * coarse active sets: a stencil on the fine level's bit planes (k_bk_coarse_*: per coarse brick 27 fine-brick look-ups, not
//     3.4 look-ups per fine voxel); the coarse coordinates are then ENUMERATED from the masks.  The coarse rows come out in
//     brick order, not in spconv's (hash) or the oracle's (first appearance) order: an internal row numbering -- every
//     output row of a convolution is computed on its own, so the features of a voxel do not depend on it
//     (tests/test_unet.py::test_brick_pyramid_*: same sets, same pairs, same network outputs bit for bit).
//   * all levels are built by ONE call with the counts kept on the device; the host reads them back once at the end
//     (rounds 1-2: one blocking read-back per strided level).  Tables are laid out with a row stride = the level's capacity.
#include "st_common.h"
#include "st_grid.h"  // ST_MAX_SEG

#define BK_BLOCK 256
#define BK_MAX_DEPTH 6

struct alignas(16) BkRec {
    unsigned long long mask[8];  // word w = z & 7, bit (y & 7) * 8 + (x & 7)
    unsigned char cum[8];        // voxels of the brick in the words before w (a brick holds <= 512: the last prefix is <= 448...
    unsigned tidx;               // dense table index of the brick (block * mt + Morton code): the way back to coordinates
    unsigned cum_hi;             // bit w: cum[w] has overflowed 255 (cum is kept modulo 256, see bk_cum)
};

struct BkLevel {
    int mb;            // Morton bits per axis of the brick coordinate
    unsigned mt;       // dense table entries per block = 1 << (3 * mb)
    int64_t ntab;      // blocks * mt
    uint32_t* p;       // [ntab + 1] occupancy flags, then their exclusive prefix; p[t + 1] > p[t] <=> brick t is occupied, slot p[t]
    BkRec* rec;        // [slot_cap]
    uint32_t* base;    // [slot_cap] voxels in the bricks in front (exclusive prefix of the bricks' counts)
    int64_t slot_cap;
    int64_t* n_slots;  // device: bricks in use
    int64_t* n_vox;    // device: voxels of the level
};

struct BkState {
    int ext[ST_MAX_SEG][3];  // level-0 extent (largest z, y, x) of every cloud; level l clips at ext >> l
    unsigned fail;           // bit 0: a capacity was exceeded, bit 1: a coordinate outside the declared bound
};

__device__ __forceinline__ unsigned bk_spread3(unsigned v) {  // 7 bits -> every third bit
    v &= 0x7fu;
    v = (v | (v << 8)) & 0x0000700fu;
    v = (v | (v << 4)) & 0x000430c3u;
    v = (v | (v << 2)) & 0x00049249u;
    return v;
}
__device__ __forceinline__ unsigned bk_compact3(unsigned v) {
    v &= 0x00049249u;
    v = (v | (v >> 2)) & 0x000430c3u;
    v = (v | (v >> 4)) & 0x0000700fu;
    v = (v | (v >> 8)) & 0x0000007fu;
    return v;
}
__device__ __forceinline__ unsigned bk_morton(int bz, int by, int bx) { return (bk_spread3((unsigned)bz) << 2) | (bk_spread3((unsigned)by) << 1) | bk_spread3((unsigned)bx); }
__device__ __forceinline__ unsigned bk_cum(const BkRec& r, int w) { return (unsigned)r.cum[w] + (((r.cum_hi >> w) & 1u) << 8); }

// dense table index of the brick that holds (b, z, y, x); coordinates must be inside [0, 8 << mb)
__device__ __forceinline__ int64_t bk_tab(const BkLevel& L, int b, int z, int y, int x) {
    return (int64_t)b * L.mt + bk_morton(z >> 3, y >> 3, x >> 3);
}
// row of the voxel at (b, z, y, x) in the level's order, or -1
__device__ __forceinline__ int bk_lookup(const BkLevel& L, int b, int z, int y, int x) {
    const int lim = 8 << L.mb;
    if ((unsigned)z >= (unsigned)lim || (unsigned)y >= (unsigned)lim || (unsigned)x >= (unsigned)lim) return -1;
    const int64_t t = bk_tab(L, b, z, y, x);
    const uint32_t s = L.p[t];
    if (L.p[t + 1] == s) return -1;
    const BkRec& r = L.rec[s];
    const int w = z & 7, bit = ((y & 7) << 3) | (x & 7);
    const unsigned long long m = r.mask[w];
    if (!((m >> bit) & 1ull)) return -1;
    return (int)(L.base[s] + bk_cum(r, w) + (unsigned)__popcll(m & ((1ull << bit) - 1ull)));
}

#define BK_LOOP(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

static inline unsigned bk_grid(int64_t n) {
    int64_t g = st_div_up(n > 0 ? n : 1, BK_BLOCK);
    return (unsigned)(g < 16384 ? g : 16384);
}

// ---- level 0 -----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BK_BLOCK) k_bk_mark0(const int32_t* coords, int64_t n, BkLevel L, BkState* st, const int32_t* blk_seg,
                                                       int nseg, int nblk) {
    __shared__ int m[ST_MAX_SEG * 3];
    for (int i = threadIdx.x; i < nseg * 3; i += blockDim.x) m[i] = 0;
    __syncthreads();
    const int lim = 8 << L.mb;
    BK_LOOP(i, n) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];  // (b, z, y, x)
        if ((unsigned)c.x >= (unsigned)nblk || (unsigned)c.y >= (unsigned)lim || (unsigned)c.z >= (unsigned)lim || (unsigned)c.w >= (unsigned)lim) {
            atomicOr(&st->fail, 2u);
            continue;
        }
        L.p[bk_tab(L, c.x, c.y, c.z, c.w)] = 1u;
        const int s = blk_seg ? blk_seg[c.x] : 0;
        if (c.y > m[3 * s]) atomicMax(&m[3 * s], c.y);
        if (c.z > m[3 * s + 1]) atomicMax(&m[3 * s + 1], c.z);
        if (c.w > m[3 * s + 2]) atomicMax(&m[3 * s + 2], c.w);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nseg * 3; i += blockDim.x) if (m[i]) atomicMax(&(&st->ext[0][0])[i], m[i]);
}

// occupied bricks get an empty record that remembers its table index (after the scan of the flags)
__global__ void __launch_bounds__(BK_BLOCK) k_bk_init_rec(BkLevel L, BkState* st) {
    BK_LOOP(t, L.ntab) {
        const uint32_t s = L.p[t];
        if (L.p[t + 1] == s) continue;
        if ((int64_t)s >= L.slot_cap) { atomicOr(&st->fail, 1u); continue; }
        BkRec r;
#pragma unroll
        for (int w = 0; w < 8; w++) { r.mask[w] = 0ull; r.cum[w] = 0; }
        r.tidx = (unsigned)t;
        r.cum_hi = 0u;
        L.rec[s] = r;
    }
}

__device__ __forceinline__ void bk_set_bit(const BkLevel& L, int b, int z, int y, int x) {
    const uint32_t s = L.p[bk_tab(L, b, z, y, x)];
    if ((int64_t)s >= L.slot_cap) return;  // flagged by k_bk_init_rec
    const int bit = ((y & 7) << 3) | (x & 7);
    unsigned* half = reinterpret_cast<unsigned*>(&L.rec[s].mask[z & 7]) + (bit >> 5);
    const unsigned v = 1u << (bit & 31);
    if (!(*half & v)) atomicOr(half, v);  // (a stale read can only send us to the atomic needlessly)
}

__global__ void __launch_bounds__(BK_BLOCK) k_bk_bits0(const int32_t* coords, int64_t n, BkLevel L, const BkState* st) {
    if (st->fail) return;  // a coordinate outside the declared bounds or a level over its capacity (slots >= slot_cap): nothing below may index the tables
    const int lim = 8 << L.mb;
    BK_LOOP(i, n) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        if ((unsigned)c.y >= (unsigned)lim || (unsigned)c.z >= (unsigned)lim || (unsigned)c.w >= (unsigned)lim) continue;
        bk_set_bit(L, c.x, c.y, c.z, c.w);
    }
}

// per brick: running popcounts of its words, total -> base[] (scanned afterwards)
__global__ void __launch_bounds__(BK_BLOCK) k_bk_count(BkLevel L) {
    const int64_t ns = *L.n_slots < L.slot_cap ? *L.n_slots : L.slot_cap;
    BK_LOOP(s, ns) {
        BkRec& r = L.rec[s];
        unsigned run = 0, hi = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            r.cum[w] = (unsigned char)(run & 255u);
            hi |= (run >> 8) << w;
            run += (unsigned)__popcll(r.mask[w]);
        }
        r.cum_hi = hi;
        L.base[s] = run;
    }
}

// order0[row] = input voxel, coords_sorted[row] = its coordinates
__global__ void __launch_bounds__(BK_BLOCK) k_bk_order0(const int32_t* coords, int64_t n, BkLevel L, int32_t* order0, int32_t* sorted,
                                                        const BkState* st) {
    if (st->fail) return;
    BK_LOOP(i, n) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        const int r = bk_lookup(L, c.x, c.y, c.z, c.w);
        if (r < 0 || r >= n) continue;  // (cannot happen for in-bound coordinates; the fail flag tells the host otherwise)
        order0[r] = (int32_t)i;
        reinterpret_cast<int4*>(sorted)[r] = c;
    }
}

// ---- submanifold neighbour table of a level (rows in the level's order) ------------------------------------------------------
// A 3 x 3 x 3 neighbourhood meets at most 2 x 2 x 2 bricks: their table entries are fetched once, then per z-plane the
// (up to four) mask words; the 27 ranks are popcounts.
__global__ void __launch_bounds__(BK_BLOCK) k_bk_subm(const int32_t* coords, const int64_t* n_dev, int64_t cap, BkLevel L, int32_t* nbr,
                                                      const BkState* st) {
    if (st->fail) return;
    const int64_t n = *n_dev < cap ? *n_dev : cap;
    const int lim = 8 << L.mb;
    BK_LOOP(o, n) {
        const int4 c = reinterpret_cast<const int4*>(coords)[o];
        const int b = c.x, z = c.y, y = c.z, x = c.w;
        // brick coordinate of the low / high side per axis (equal unless the voxel sits on a brick face)
        int slot[8];
        uint32_t bs[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int zz = z + ((q & 4) ? 1 : -1), yy = y + ((q & 2) ? 1 : -1), xx = x + ((q & 1) ? 1 : -1);
            slot[q] = -1;
            bs[q] = 0u;
            if ((unsigned)zz < (unsigned)lim && (unsigned)yy < (unsigned)lim && (unsigned)xx < (unsigned)lim) {
                const int64_t t = bk_tab(L, b, zz, yy, xx);
                const uint32_t s = L.p[t];
                if (L.p[t + 1] != s) { slot[q] = (int)s; bs[q] = L.base[s]; }
            }
        }
#pragma unroll
        for (int dz = -1; dz <= 1; dz++) {
            const int zz = z + dz, w = zz & 7;
            const int qz = (zz >> 3) == ((z - 1) >> 3) ? 0 : 4;  // which of the two brick layers this plane lies in
            unsigned long long m[4];
            unsigned cu[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int s = slot[qz | q];
                m[q] = 0ull;
                cu[q] = 0u;
                if (s >= 0 && (unsigned)zz < (unsigned)lim) { const BkRec& r = L.rec[s]; m[q] = r.mask[w]; cu[q] = bs[qz | q] + bk_cum(r, w); }
            }
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const int dy = j / 3 - 1, dx = j % 3 - 1;
                const int yy = y + dy, xx = x + dx, k = (dz + 1) * 9 + j;
                int r = -1;
                if (dz == 0 && j == 4) r = (int)o;
                else if ((unsigned)yy < (unsigned)lim && (unsigned)xx < (unsigned)lim) {
                    const int q = ((yy >> 3) == ((y - 1) >> 3) ? 0 : 2) | ((xx >> 3) == ((x - 1) >> 3) ? 0 : 1);
                    const int bit = ((yy & 7) << 3) | (xx & 7);
                    const unsigned long long mm = m[q];
                    if ((mm >> bit) & 1ull) r = (int)(cu[q] + (unsigned)__popcll(mm & ((1ull << bit) - 1ull)));
                }
                nbr[(int64_t)k * cap + o] = r;
            }
        }
    }
}

// ---- coarse active set of the strided convolution (k3 s2 p1): coarse o is active iff a fine voxel sits at 2o - 1 + k, k in 0..2 per
// axis, and 0 <= o <= ext / 2.  Computed brick by brick on the MASKS (a stencil on bit planes), not voxel by voxel: a fine
// voxel offers itself to up to 8 outputs and ~6 fine voxels share an output -- per voxel that was 3.4 look-ups + bit-sets each
// (27 us per cloud at 24 clouds per launch set); per coarse brick it is 27 fine-brick look-ups for 512 outputs.
// PASS A: every occupied fine brick fb flags the coarse bricks it can reach: fb >> 1 per axis, and (fb >> 1) + 1 when fb is odd
// and the brick has a voxel on its last layer of that axis (fine c = 8 fb + 7 -> o = 4 fb + 4).
__global__ void __launch_bounds__(BK_BLOCK) k_bk_coarse_mark(BkLevel Lf, BkLevel Lc, const BkState* st, int level, const int32_t* blk_seg) {
    if (st->fail) return;
    const int64_t ns = *Lf.n_slots < Lf.slot_cap ? *Lf.n_slots : Lf.slot_cap;
    const int nbc = 1 << Lc.mb;
    BK_LOOP(s, ns) {
        const BkRec& r = Lf.rec[s];
        const unsigned t = r.tidx, mc = t & (Lf.mt - 1u);
        const int b = (int)(t >> (3 * Lf.mb));
        const int fb[3] = {(int)bk_compact3(mc >> 2), (int)bk_compact3(mc >> 1), (int)bk_compact3(mc)};
        const int* e = st->ext[blk_seg ? blk_seg[b] : 0];
        unsigned long long all = 0ull;
#pragma unroll
        for (int w = 0; w < 8; w++) all |= r.mask[w];
        for (int q = 0; q < 8; q++) {  // q bit 2 / 1 / 0: the next coarse brick along z / y / x
            const int d[3] = {(q >> 2) & 1, (q >> 1) & 1, q & 1};
            unsigned long long m = d[0] ? r.mask[7] : all;
            if (d[1]) m &= 0xff00000000000000ull;
            if (d[2]) m &= 0x8080808080808080ull;
            bool ok = m != 0ull;
            int cb[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                cb[a] = (fb[a] >> 1) + d[a];
                ok = ok && (!d[a] || (fb[a] & 1)) && cb[a] < nbc && 8 * cb[a] <= ((e[a] >> level) >> 1);
            }
            if (ok) Lc.p[(int64_t)b * Lc.mt + bk_morton(cb[0], cb[1], cb[2])] = 1u;
        }
    }
}

// PASS B: one thread per (coarse brick, z word): OR of the three fine z planes 2 oz - 1 .. 2 oz + 1, each a 17 x 17 bit window
// (fine y, x from 16 cb - 1 to 16 cb + 15: the brick pair 2 cb, 2 cb + 1 and the last row / column of brick 2 cb - 1), then
// "dilate by one and take every second bit" along x and y, then the clip at the cloud's extent.
__global__ void __launch_bounds__(BK_BLOCK) k_bk_coarse_stencil(BkLevel Lf, BkLevel Lc, const BkState* st, int level, const int32_t* blk_seg) {
    if (st->fail) return;
    const int64_t ns = *Lc.n_slots < Lc.slot_cap ? *Lc.n_slots : Lc.slot_cap;
    const int limf = 8 << Lf.mb, nbf = 1 << Lf.mb;
    BK_LOOP(idx, ns * 8) {
        const int64_t s = idx >> 3;
        const int w = (int)(idx & 7);
        const unsigned t = Lc.rec[s].tidx, mc = t & (Lc.mt - 1u);
        const int b = (int)(t >> (3 * Lc.mb));
        const int cb[3] = {(int)bk_compact3(mc >> 2), (int)bk_compact3(mc >> 1), (int)bk_compact3(mc)};
        const int* e = st->ext[blk_seg ? blk_seg[b] : 0];
        const int omax[3] = {(e[0] >> level) >> 1, (e[1] >> level) >> 1, (e[2] >> level) >> 1};
        const int oz = 8 * cb[0] + w;
        unsigned long long word = 0ull;
        if (oz <= omax[0]) {
            unsigned rows[17];
#pragma unroll
            for (int r = 0; r < 17; r++) rows[r] = 0u;
            for (int dz = -1; dz <= 1; dz++) {
                const int fz = 2 * oz + dz;
                if (fz < 0 || fz >= limf) continue;
                const int fbz = fz >> 3, wz = fz & 7;
                for (int iy = 0; iy < 3; iy++) {
                    const int fby = 2 * cb[1] - 1 + iy;
                    if (fby < 0 || fby >= nbf) continue;
                    for (int ix = 0; ix < 3; ix++) {
                        const int fbx = 2 * cb[2] - 1 + ix;
                        if (fbx < 0 || fbx >= nbf) continue;
                        const int64_t tf = (int64_t)b * Lf.mt + bk_morton(fbz, fby, fbx);
                        const uint32_t sf = Lf.p[tf];
                        if (Lf.p[tf + 1] == sf) continue;
                        const unsigned long long m = Lf.rec[sf].mask[wz];
                        if (m == 0ull) continue;
                        for (int yy = iy == 0 ? 7 : 0; yy < 8; yy++) {
                            const unsigned rb = (unsigned)(m >> (8 * yy)) & 0xffu;
                            const int r = iy == 0 ? 0 : (iy == 1 ? 1 + yy : 9 + yy);
                            rows[r] |= ix == 0 ? (rb >> 7) : (ix == 1 ? rb << 1 : rb << 9);
                        }
                    }
                }
            }
            unsigned c8[17];
#pragma unroll
            for (int r = 0; r < 17; r++) {
                unsigned x = rows[r] | (rows[r] >> 1) | (rows[r] >> 2);  // bit 2 ox: any of columns 2 ox, 2 ox + 1, 2 ox + 2
                x &= 0x5555u;
                x = (x | (x >> 1)) & 0x3333u;
                x = (x | (x >> 2)) & 0x0f0fu;
                x = (x | (x >> 4)) & 0x00ffu;
                c8[r] = x;
            }
            unsigned xclip = 0xffu;  // coarse x beyond the extent
            if (8 * cb[2] + 7 > omax[2]) xclip = omax[2] >= 8 * cb[2] ? (0xffu >> (7 - (omax[2] - 8 * cb[2]))) : 0u;
#pragma unroll
            for (int oy = 0; oy < 8; oy++) {
                if (8 * cb[1] + oy > omax[1]) break;
                const unsigned rb = (c8[2 * oy] | c8[2 * oy + 1] | c8[2 * oy + 2]) & xclip;
                word |= (unsigned long long)rb << (8 * oy);
            }
        }
        Lc.rec[s].mask[w] = word;
    }
}

// coordinates of a level enumerated from its masks (rows in the level's order)
__global__ void __launch_bounds__(BK_BLOCK) k_bk_emit(BkLevel L, int32_t* coords, int64_t cap, BkState* st) {
    if (st->fail) return;
    const int64_t ns = *L.n_slots < L.slot_cap ? *L.n_slots : L.slot_cap;
    BK_LOOP(idx, ns * 8) {
        const int64_t s = idx >> 3;
        const int w = (int)(idx & 7);
        const BkRec& r = L.rec[s];
        unsigned long long m = r.mask[w];
        if (m == 0ull) continue;
        const unsigned t = r.tidx, mc = t & (L.mt - 1u);
        const int b = (int)(t >> (3 * L.mb));
        const int z = (int)(bk_compact3(mc >> 2) << 3) | w, y0 = (int)(bk_compact3(mc >> 1) << 3), x0 = (int)(bk_compact3(mc) << 3);
        int64_t row = (int64_t)L.base[s] + bk_cum(r, w);
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            if (row < cap) reinterpret_cast<int4*>(coords)[row] = make_int4(b, z, y0 + (bit >> 3), x0 + (bit & 7));
            else atomicOr(&st->fail, 1u);
            row++;
        }
    }
}

// both tables of the strided pair set: nbr_up[k][i] = coarse row reached from fine row i through offset k, nbr_down[k][r] = i
// (every (k, r) has one writer; the rest keeps the -1 it was filled with); class histogram of the fine rows for the parity order
__device__ __forceinline__ int bk_parity_class(int z, int y, int x) { return ((z & 1) << 2) | ((y & 1) << 1) | (x & 1); }
__global__ void __launch_bounds__(BK_BLOCK) k_bk_updown(const int32_t* coords, const int64_t* n_dev, int64_t cap_f, BkLevel Lc, const BkState* st,
                                                        int level, const int32_t* blk_seg, const int64_t* m_dev, int64_t cap_c,
                                                        int32_t* nbr_up, int32_t* nbr_down, uint32_t* parity_count) {
    __shared__ uint32_t hist[8];
    if (st->fail) return;
    if (threadIdx.x < 8) hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t n = *n_dev < cap_f ? *n_dev : cap_f;
    const int64_t m = *m_dev < cap_c ? *m_dev : cap_c;
    BK_LOOP(i, n) {
        const int4 c4 = reinterpret_cast<const int4*>(coords)[i];
        const int b = c4.x, c[3] = {c4.y, c4.z, c4.w};
        const int* e = st->ext[blk_seg ? blk_seg[b] : 0];
        const int omax[3] = {(e[0] >> level) >> 1, (e[1] >> level) >> 1, (e[2] >> level) >> 1};
        atomicAdd(&hist[bk_parity_class(c[0], c[1], c[2])], 1u);
        // a fine voxel reaches at most 2 x 2 x 2 outputs (o = c >> 1, and o + 1 when c is odd), nearly always in ONE coarse brick:
        // the brick's table entry and the mask word of a z plane are fetched when they change, not once per offset
        int64_t t_have = -1;
        int slot = -1, w_have = -1;
        unsigned bs = 0u, cu = 0u;
        unsigned long long mw = 0ull;
        const int limc = 8 << Lc.mb;
#pragma unroll
        for (int kz = 0; kz < 3; kz++)
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const int k = kz * 9 + j;
                const int nz = c[0] + 1 - kz, ny = c[1] + 1 - j / 3, nx = c[2] + 1 - j % 3;
                int r = -1;
                if (!((nz | ny | nx) & 1) && nz >= 0 && ny >= 0 && nx >= 0 && (nz >> 1) <= omax[0] && (ny >> 1) <= omax[1] && (nx >> 1) <= omax[2]) {
                    const int oz = nz >> 1, oy = ny >> 1, ox = nx >> 1;
                    if (oz < limc && oy < limc && ox < limc) {
                        const int64_t tc = bk_tab(Lc, b, oz, oy, ox);
                        if (tc != t_have) {
                            t_have = tc;
                            w_have = -1;
                            const uint32_t sc = Lc.p[tc];
                            slot = Lc.p[tc + 1] != sc ? (int)sc : -1;
                            bs = slot >= 0 ? Lc.base[slot] : 0u;
                        }
                        if (slot >= 0) {
                            const int wz = oz & 7;
                            if (wz != w_have) { w_have = wz; const BkRec& rc = Lc.rec[slot]; mw = rc.mask[wz]; cu = bs + bk_cum(rc, wz); }
                            const int bit = ((oy & 7) << 3) | (ox & 7);
                            if ((mw >> bit) & 1ull) r = (int)(cu + (unsigned)__popcll(mw & ((1ull << bit) - 1ull)));
                        }
                    }
                }
                nbr_up[(int64_t)k * cap_f + i] = r;
                if (r >= 0 && r < m) nbr_down[(int64_t)k * cap_c + r] = (int32_t)i;
            }
    }
    __syncthreads();
    if (threadIdx.x < 8 && hist[threadIdx.x]) atomicAdd(&parity_count[threadIdx.x], hist[threadIdx.x]);
}

// order[p] = fine row | (8 + class) << 28, rows grouped by coordinate parity class (see rulebook.hip k_rb_parity_order: the
// launch order of the inverse convolution; never changes a result).  count[8] from k_bk_updown, cursor[8] zeroed.
#define BK_ORDER_BLOCKS 512
__global__ void __launch_bounds__(BK_BLOCK) k_bk_parity_order(const int32_t* coords, const int64_t* n_dev, int64_t cap, const uint32_t* count,
                                                              uint32_t* cursor, int32_t* order, const BkState* st) {
    __shared__ uint32_t hist[8], gbase[8], lcur[8];
    if (st->fail) return;
    const int64_t n = *n_dev < cap ? *n_dev : cap;
    const int lane = threadIdx.x & 63;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, chunk = (per + BK_BLOCK - 1) / BK_BLOCK * BK_BLOCK;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (threadIdx.x < 8) { hist[threadIdx.x] = 0; lcur[threadIdx.x] = 0; }
    __syncthreads();
    for (int64_t i0 = lo + (threadIdx.x - lane); i0 < hi; i0 += BK_BLOCK) {
        const int64_t i = i0 + lane;
        int cls = -1;
        if (i < hi) cls = bk_parity_class(coords[4 * i + 1], coords[4 * i + 2], coords[4 * i + 3]);
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
    for (int64_t i0 = lo + (threadIdx.x - lane); i0 < hi; i0 += BK_BLOCK) {
        const int64_t i = i0 + lane;
        int cls = -1;
        if (i < hi) cls = bk_parity_class(coords[4 * i + 1], coords[4 * i + 2], coords[4 * i + 3]);
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

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct BkPlan {
    int depth;
    int mb[BK_MAX_DEPTH + 1];
    int64_t ntab[BK_MAX_DEPTH + 1], slot_cap[BK_MAX_DEPTH + 1];
};

static bool bk_plan(int64_t n0, int n_blocks, int coord_bound, int depth, const int64_t* caps, BkPlan* P) {
    if (depth < 0 || depth > BK_MAX_DEPTH || n_blocks < 1 || coord_bound < 1) return false;
    P->depth = depth;
    int bound = coord_bound;  // coordinates of level l are < bound_l
    for (int l = 0; l <= depth; l++) {
        int nb = (bound + 7) / 8, mb = 0;
        while ((1 << mb) < nb) mb++;
        if (mb > 7) return false;
        P->mb[l] = mb;
        P->ntab[l] = (int64_t)n_blocks << (3 * mb);
        if (P->ntab[l] >= (1ll << 31)) return false;  // tidx is 32 bit
        const int64_t cap = l == 0 ? n0 : caps[l];
        P->slot_cap[l] = cap < P->ntab[l] ? cap : P->ntab[l];
        bound = (bound - 1) / 2 + 1;
    }
    return true;
}

struct BkLayout {
    BkState* st;
    int64_t* counts;    // [2 * (depth + 1)]: n_slots, n_vox per level
    uint32_t* p[BK_MAX_DEPTH + 1];
    BkRec* rec[BK_MAX_DEPTH + 1];
    uint32_t* base[BK_MAX_DEPTH + 1];
    char* scan_ws;
    int64_t scan_bytes;
};

static void bk_layout(StArena& a, const BkPlan& P, BkLayout* Y) {
    Y->st = a.take<BkState>(1);
    Y->counts = a.take<int64_t>(2 * (BK_MAX_DEPTH + 1));
    int64_t scan_n = 1;
    for (int l = 0; l <= P.depth; l++) {
        Y->p[l] = a.take<uint32_t>(P.ntab[l] + 1);
        Y->rec[l] = a.take<BkRec>(P.slot_cap[l]);
        Y->base[l] = a.take<uint32_t>(P.slot_cap[l] + 1);
        if (P.ntab[l] + 1 > scan_n) scan_n = P.ntab[l] + 1;
        if (P.slot_cap[l] + 1 > scan_n) scan_n = P.slot_cap[l] + 1;
    }
    Y->scan_bytes = st_scan_ws_bytes(scan_n);
    Y->scan_ws = a.take<char>(Y->scan_bytes);
}

// 0 if the structure cannot be sized for these arguments (the caller then uses the hash-table builders of rulebook.hip)
extern "C" int64_t st_brick_pyramid_workspace_bytes(int64_t n0, int n_blocks, int coord_bound, int depth, const int64_t* caps) {
    BkPlan P;
    if (!bk_plan(n0 > 0 ? n0 : 1, n_blocks, coord_bound, depth, caps, &P)) return 0;
    StArena a(nullptr, 0);
    BkLayout Y;
    bk_layout(a, P, &Y);
    return a.used;
}

// All levels of the rulebook pyramid of one voxel batch.
//   coords0 [n0,4] int32 (batch index, z, y, x), any order; batch index < n_blocks, z / y / x < coord_bound (both are host-side
//   bounds the caller knows from how the voxels were made; violated -> ST_ERR_INVALID); blk_seg / nseg: cloud of every batch
//   index (NULL / 1: one cloud) -- the coarse sets of a cloud are clipped to that cloud's own extent.
//   caps [depth + 1]: row capacity of every level's arrays (caps[0] = n0); exceeded -> ST_ERR_WORKSPACE with counts_host filled
//   with what WOULD be needed is not possible (the enumeration stops), so the caller retries with the worst case 8 x.
// out: order0 [n0] (row p of level 0 holds input voxel order0[p]); coords_out[l] [caps[l],4]; subm[l] [27][caps[l]];
//   down[l] [27][caps[l+1]]; up[l] [27][caps[l]]; up_order[l] [caps[l] + 16]; counts_host [depth + 1] rows per level (ONE
//   device-to-host copy at the end of the call).
extern "C" int st_brick_pyramid(const int32_t* coords0, int64_t n0, int n_blocks, int coord_bound, int depth, const int32_t* blk_seg,
                                int nseg, const int64_t* caps, int32_t* order0, int32_t* const* coords_out, int32_t* const* subm,
                                int32_t* const* down, int32_t* const* up, int32_t* const* up_order, int64_t* counts_host, void* ws,
                                int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    for (int l = 0; l <= depth && l <= BK_MAX_DEPTH; l++) counts_host[l] = 0;
    if (n0 <= 0) return ST_OK;
    ST_REQUIRE(nseg >= 1 && nseg <= ST_MAX_SEG && (nseg == 1 || blk_seg), "brick pyramid: 1 <= clouds per batch <= %d", ST_MAX_SEG);
    ST_REQUIRE(caps && caps[0] >= n0, "brick pyramid: caps[0] must hold the input voxels");
    if (nseg == 1) blk_seg = nullptr;
    BkPlan P;
    ST_REQUIRE(bk_plan(n0, n_blocks, coord_bound, depth, caps, &P), "brick pyramid: cannot be sized for %d blocks of %d^3 cells", n_blocks, coord_bound);
    StArena a(ws, ws_bytes);
    BkLayout Y;
    bk_layout(a, P, &Y);
    if (!a.ok() || !Y.scan_ws) {
        st_set_error("brick pyramid: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    (void)hipMemsetAsync(Y.st, 0, sizeof(BkState), stream);
    (void)hipMemsetAsync(Y.counts, 0, 2 * (BK_MAX_DEPTH + 1) * sizeof(int64_t), stream);
    BkLevel L[BK_MAX_DEPTH + 1];
    for (int l = 0; l <= depth; l++) {
        L[l].mb = P.mb[l]; L[l].mt = 1u << (3 * P.mb[l]); L[l].ntab = P.ntab[l]; L[l].p = Y.p[l]; L[l].rec = Y.rec[l];
        L[l].base = Y.base[l]; L[l].slot_cap = P.slot_cap[l]; L[l].n_slots = &Y.counts[2 * l]; L[l].n_vox = &Y.counts[2 * l + 1];
        (void)hipMemsetAsync(Y.p[l], 0, (P.ntab[l] + 1) * sizeof(uint32_t), stream);
    }
    // after the flags / bits of level l are set: slots, records, counts, bases
    auto close_flags = [&](int l) -> int {  // flags -> slots + empty records
        ST_TRY(st_exclusive_scan_u32(L[l].p, L[l].p, L[l].ntab + 1, reinterpret_cast<uint32_t*>(L[l].n_slots), Y.scan_ws, Y.scan_bytes, stream));
        hipLaunchKernelGGL(k_bk_init_rec, dim3(bk_grid(L[l].ntab)), dim3(BK_BLOCK), 0, stream, L[l], Y.st);
        return ST_OK;
    };
    auto close_bits = [&](int l) -> int {  // bits -> running popcounts, brick bases, the level's voxel count
        hipLaunchKernelGGL(k_bk_count, dim3(bk_grid(L[l].slot_cap)), dim3(BK_BLOCK), 0, stream, L[l]);
        ST_TRY(st_exclusive_scan_u32(L[l].base, L[l].base, L[l].slot_cap, reinterpret_cast<uint32_t*>(L[l].n_vox), Y.scan_ws, Y.scan_bytes,
                                     stream, L[l].n_slots));
        return ST_OK;
    };
    // ---- level 0: structure, order, sorted coordinates
    hipLaunchKernelGGL(k_bk_mark0, dim3(bk_grid(n0)), dim3(BK_BLOCK), 0, stream, coords0, n0, L[0], Y.st, blk_seg, nseg, n_blocks);
    ST_TRY(close_flags(0));
    hipLaunchKernelGGL(k_bk_bits0, dim3(bk_grid(n0)), dim3(BK_BLOCK), 0, stream, coords0, n0, L[0], (const BkState*)Y.st);
    ST_TRY(close_bits(0));
    hipLaunchKernelGGL(k_bk_order0, dim3(bk_grid(n0)), dim3(BK_BLOCK), 0, stream, coords0, n0, L[0], order0, coords_out[0], (const BkState*)Y.st);
    for (int l = 0;; l++) {
        const int64_t cap = caps[l];
        hipLaunchKernelGGL(k_bk_subm, dim3(bk_grid(cap)), dim3(BK_BLOCK), 0, stream, (const int32_t*)coords_out[l], (const int64_t*)L[l].n_vox, cap,
                           L[l], subm[l], (const BkState*)Y.st);
        if (l == depth) break;
        const int64_t cap_c = caps[l + 1];
        hipLaunchKernelGGL(k_bk_coarse_mark, dim3(bk_grid(L[l].slot_cap)), dim3(BK_BLOCK), 0, stream, L[l], L[l + 1], (const BkState*)Y.st, l, blk_seg);
        ST_TRY(close_flags(l + 1));
        hipLaunchKernelGGL(k_bk_coarse_stencil, dim3(bk_grid(L[l + 1].slot_cap * 8)), dim3(BK_BLOCK), 0, stream, L[l], L[l + 1], (const BkState*)Y.st, l,
                           blk_seg);
        ST_TRY(close_bits(l + 1));
        hipLaunchKernelGGL(k_bk_emit, dim3(bk_grid(L[l + 1].slot_cap * 8)), dim3(BK_BLOCK), 0, stream, L[l + 1], coords_out[l + 1], cap_c, Y.st);
        (void)hipMemsetAsync(down[l], 0xff, 27 * cap_c * sizeof(int32_t), stream);
        uint32_t* pc = reinterpret_cast<uint32_t*>(up_order[l] + cap);  // 8 class counts + 8 cursors
        (void)hipMemsetAsync(pc, 0, 16 * sizeof(uint32_t), stream);
        hipLaunchKernelGGL(k_bk_updown, dim3(bk_grid(cap)), dim3(BK_BLOCK), 0, stream, (const int32_t*)coords_out[l], (const int64_t*)L[l].n_vox, cap,
                           L[l + 1], (const BkState*)Y.st, l, blk_seg, (const int64_t*)L[l + 1].n_vox, cap_c, up[l], down[l], pc);
        const unsigned og = bk_grid(cap) < BK_ORDER_BLOCKS ? bk_grid(cap) : BK_ORDER_BLOCKS;
        hipLaunchKernelGGL(k_bk_parity_order, dim3(og), dim3(BK_BLOCK), 0, stream, (const int32_t*)coords_out[l], (const int64_t*)L[l].n_vox, cap,
                           (const uint32_t*)pc, pc + 8, up_order[l], (const BkState*)Y.st);
    }
    // the ONE read-back: rows per level + the fail flags
    struct { int64_t counts[2 * (BK_MAX_DEPTH + 1)]; } hc;
    unsigned hfail = 0;
    (void)hipMemcpyAsync(&hc, Y.counts, sizeof(hc), hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(&hfail, &Y.st->fail, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
    st_stream_wait(stream);
    ST_CHECK_LAUNCH();
    ST_REQUIRE(!(hfail & 2u), "brick pyramid: a coordinate lies outside the declared bounds (%d blocks, %d cells per axis)", n_blocks, coord_bound);
    for (int l = 0; l <= depth; l++) counts_host[l] = hc.counts[2 * l + 1];
    bool over = (hfail & 1u) != 0;
    for (int l = 0; l <= depth; l++) over = over || hc.counts[2 * l + 1] > caps[l] || hc.counts[2 * l] > P.slot_cap[l];
    if (over) {
        st_set_error("brick pyramid: a level needs more rows than its capacity (caps)");
        return ST_ERR_WORKSPACE;
    }
    ST_REQUIRE(hc.counts[1] == n0, "brick pyramid: %lld distinct voxels among %lld input rows (duplicate coordinates)", (long long)hc.counts[1],
               (long long)n0);
    return ST_OK;
}
