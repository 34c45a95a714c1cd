// Device-wide exclusive scan and stable LSD radix sort (wave64 / LDS based), used by the
// voxeliser, the rulebook builder and the graph stage for order-preserving compaction.
// No inter-workgroup communication inside a launch: scan = reduce / scan-of-sums / apply.
#include <cstdarg>

#include "st_common.h"

// ------------------------------------------------------------------------------ error text ---
static thread_local char g_err[512] = "";

void st_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* st_last_error(void) { return g_err; }
void st_stream_wait(hipStream_t stream) { (void)hipStreamSynchronize(stream); }

extern "C" int st_version(void) { return 101; }

// ------------------------------------------------------------------------------------ scan ---
// 16 items per lane, read as four 16-byte loads where the arrays allow it (the grid builders scan tens of millions of
// mostly empty cells: at 8 scalar items per lane the scan ran at a tenth of the HBM rate).
#define SCAN_BLOCK 256
#define SCAN_ITEMS 16
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)
#define SCAN_DIRECT_BLOCKS 1024  // up to this many tiles every workgroup sums the tile totals in front of it itself

__device__ __forceinline__ void scan_load(const uint32_t* in, int64_t base, int64_t n, uint32_t (&v)[SCAN_ITEMS]) {
    if (base + SCAN_ITEMS <= n && ((((uintptr_t)in) & 15) == 0)) {  // base is a multiple of 16 items
        const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS / 4; q++) {
            const uint4 t = p[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) v[i] = base + i < n ? in[base + i] : 0u;
    }
}

// n_dev (optional): the number of elements actually in use, known on the device only (e.g. the cell count of a grid
// whose bounding box never visits the host); the launch is sized for the bound n, tiles past *n_dev do (almost) nothing.
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_reduce(const uint32_t* in, uint32_t* block_sums, int64_t n, const int64_t* n_dev) {
    __shared__ uint32_t lds[SCAN_BLOCK / 64 + 1];
    if (n_dev) {
        const int64_t m = *n_dev;
        n = m < n ? m : n;
        if ((int64_t)blockIdx.x * SCAN_TILE >= n) { if (threadIdx.x == 0) block_sums[blockIdx.x] = 0u; return; }
    }
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    scan_load(in, base, n, v);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    uint32_t total;
    block_exclusive_scan(s, lds, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Last pass.  sums_scanned == false: every workgroup sums the tile totals in front of its own tile itself (at most
// SCAN_DIRECT_BLOCKS words, L2 resident) -- one launch less per scan, and a pipeline pass runs ~20 scans; true: block_sums
// has been scanned (recursively) and holds each tile's offset, block_sums[gridDim.x] the grand total.
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const uint32_t* in, uint32_t* out, const uint32_t* block_sums,
                                                           int64_t n, uint32_t* total_out, int sums_scanned, const int64_t* n_dev) {
    __shared__ uint32_t lds[SCAN_BLOCK / 64 + 1];
    if (n_dev) {  // elements past *n_dev are neither read nor written; the total still lands in total_out
        const int64_t m = *n_dev;
        n = m < n ? m : n;
        const bool last_live = (int64_t)blockIdx.x * SCAN_TILE < n && ((int64_t)blockIdx.x + 1) * SCAN_TILE >= n;
        if ((int64_t)blockIdx.x * SCAN_TILE >= n && !(n == 0 && blockIdx.x == 0)) return;
        if (total_out && !last_live && !(n == 0)) total_out = nullptr;
    } else if (blockIdx.x != gridDim.x - 1) {
        total_out = nullptr;
    }
    uint32_t carry;
    if (sums_scanned) {
        carry = block_sums[blockIdx.x];
    } else {
        uint32_t pre = 0;
        for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += SCAN_BLOCK) pre += block_sums[j];
        block_exclusive_scan(pre, lds, &carry);
    }
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    scan_load(in, base, n, v);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    uint32_t total;
    uint32_t ex = block_exclusive_scan(s, lds, &total) + carry;
    if (base + SCAN_ITEMS <= n && ((((uintptr_t)out) & 15) == 0)) {
        uint4* p = reinterpret_cast<uint4*>(out + base);
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS / 4; q++) {
            uint4 t;
            t.x = ex; ex += v[4 * q];
            t.y = ex; ex += v[4 * q + 1];
            t.z = ex; ex += v[4 * q + 2];
            t.w = ex; ex += v[4 * q + 3];
            p[q] = t;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            if (base + i < n) out[base + i] = ex;
            ex += v[i];
        }
    }
    if (total_out && threadIdx.x == 0) *total_out = carry + total;
}

int64_t st_scan_ws_bytes(int64_t n) {
    StArena a(nullptr, 0);
    for (int64_t nb = st_div_up(n > 0 ? n : 1, SCAN_TILE);; nb = st_div_up(nb, SCAN_TILE)) {  // tile totals of every level
        a.take<uint32_t>(nb + 1);
        if (nb <= SCAN_DIRECT_BLOCKS) break;
    }
    return a.used;
}

static int scan_level(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total, StArena& a, hipStream_t stream,
                      const int64_t* n_dev = nullptr) {
    const int64_t nb = st_div_up(n, SCAN_TILE);
    uint32_t* sums = a.take<uint32_t>(nb + 1);
    if (!sums) {
        st_set_error("scan: workspace too small (%lld < %lld)", (long long)a.size, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, stream, in, sums, n, n_dev);
    const bool deep = nb > SCAN_DIRECT_BLOCKS;
    if (deep) ST_TRY(scan_level(sums, sums, nb, sums + nb, a, stream));  // tile totals -> tile offsets, in place
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, stream, in, out, (const uint32_t*)sums, n, total,
                       deep ? 1 : 0, n_dev);
    return ST_OK;
}

int st_exclusive_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total, void* ws, int64_t ws_bytes,
                          hipStream_t stream, const int64_t* n_dev) {
    if (n <= 0) {
        if (total) (void)hipMemsetAsync(total, 0, sizeof(uint32_t), stream);
        return ST_OK;
    }
    StArena a(ws, ws_bytes);
    ST_TRY(scan_level(in, out, n, total, a, stream, n_dev));
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// p[0 .. min(n, *n_dev)) = value (a memset whose length lives on the device)
__global__ void __launch_bounds__(SCAN_BLOCK) k_fill_u32_dev(uint32_t* p, int64_t n, const int64_t* n_dev, uint32_t value) {
    const int64_t m = *n_dev < n ? *n_dev : n;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < m; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= m && ((((uintptr_t)p) & 15) == 0)) *reinterpret_cast<uint4*>(p + i) = make_uint4(value, value, value, value);
        else for (int64_t j = i; j < m && j < i + 4; j++) p[j] = value;
    }
}
void st_fill_u32_dev(uint32_t* p, int64_t n, const int64_t* n_dev, uint32_t value, hipStream_t stream) {
    if (n <= 0) return;
    const int64_t g = st_div_up(st_div_up(n, 4), SCAN_BLOCK);
    hipLaunchKernelGGL(k_fill_u32_dev, dim3((unsigned)(g < 16384 ? g : 16384)), dim3(SCAN_BLOCK), 0, stream, p, n, n_dev, value);
}

// ------------------------------------------------------------------------------ radix sort ---
#define SORT_BLOCK 256
#define SORT_ITEMS 4
#define SORT_TILE (SORT_BLOCK * SORT_ITEMS)
#define SORT_WAVES (SORT_BLOCK / 64)

__global__ void __launch_bounds__(SORT_BLOCK) k_sort_hist(const uint32_t* keys, int64_t n, int shift, uint32_t* hist,
                                                          int64_t nb) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * SORT_TILE;
    for (int r = 0; r < SORT_ITEMS; r++) {
        int64_t e = base + (int64_t)r * SORT_BLOCK + threadIdx.x;
        if (e < n) atomicAdd(&h[(keys[e] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(SORT_BLOCK) k_sort_scatter(const uint32_t* keys, const uint32_t* vals, uint32_t* okeys,
                                                             uint32_t* ovals, int64_t n, int shift, const uint32_t* hist,
                                                             int64_t nb) {
    __shared__ uint32_t running[256];
    __shared__ uint32_t wd[SORT_WAVES][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    running[tid] = hist[(int64_t)tid * nb + blockIdx.x];
    for (int w = 0; w < SORT_WAVES; w++) wd[w][tid] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * SORT_TILE;
    for (int r = 0; r < SORT_ITEMS; r++) {
        int64_t e = base + (int64_t)r * SORT_BLOCK + tid;
        bool valid = e < n;
        uint32_t key = valid ? keys[e] : 0u;
        uint32_t val = valid ? vals[e] : 0u;
        uint32_t digit = (key >> shift) & 255u;
        // peers = valid lanes of this wave holding the same digit
        unsigned long long peers = __ballot(valid);
        for (int b = 0; b < 8; b++) {
            bool bit = (digit >> b) & 1u;
            unsigned long long bal = __ballot(valid && bit);
            peers &= bit ? bal : ~bal;
        }
        unsigned long long below = peers & ((1ull << lane) - 1ull);
        uint32_t rank = (uint32_t)__popcll(below);
        if (valid && rank == 0) wd[wave][digit] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t off = running[digit] + rank;
            for (int w = 0; w < wave; w++) off += wd[w][digit];
            okeys[off] = key;
            ovals[off] = val;
        }
        __syncthreads();
        uint32_t add = 0;
        for (int w = 0; w < SORT_WAVES; w++) {
            add += wd[w][tid];
            wd[w][tid] = 0;
        }
        running[tid] += add;
        __syncthreads();
    }
}

int64_t st_sort_ws_bytes(int64_t n) {
    if (n <= 0) n = 1;
    StArena a(nullptr, 0);
    int64_t nb = st_div_up(n, SORT_TILE);
    a.take<uint32_t>(n);
    a.take<uint32_t>(n);
    a.take<uint32_t>(256 * nb);
    a.take<char>(st_scan_ws_bytes(256 * nb));
    return a.used;
}

int st_radix_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int key_bits, void* ws, int64_t ws_bytes,
                            hipStream_t stream) {
    if (n <= 1 || key_bits <= 0) return ST_OK;
    StArena a(ws, ws_bytes);
    int64_t nb = st_div_up(n, SORT_TILE);
    uint32_t* tk = a.take<uint32_t>(n);
    uint32_t* tv = a.take<uint32_t>(n);
    uint32_t* hist = a.take<uint32_t>(256 * nb);
    int64_t scan_bytes = st_scan_ws_bytes(256 * nb);
    char* scan_ws = a.take<char>(scan_bytes);
    if (!tk || !tv || !hist || !scan_ws) {
        st_set_error("sort: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    uint32_t *ik = keys, *iv = vals, *ok = tk, *ov = tv;
    for (int shift = 0; shift < key_bits; shift += 8) {
        hipLaunchKernelGGL(k_sort_hist, dim3((unsigned)nb), dim3(SORT_BLOCK), 0, stream, (const uint32_t*)ik, n, shift,
                           hist, nb);
        ST_TRY(st_exclusive_scan_u32(hist, hist, 256 * nb, nullptr, scan_ws, scan_bytes, stream));
        hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)nb), dim3(SORT_BLOCK), 0, stream, (const uint32_t*)ik,
                           (const uint32_t*)iv, ok, ov, n, shift, (const uint32_t*)hist, nb);
        uint32_t* t;
        t = ik; ik = ok; ok = t;
        t = iv; iv = ov; ov = t;
    }
    if (ik != keys) {
        (void)hipMemcpyAsync(keys, ik, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        (void)hipMemcpyAsync(vals, iv, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
    }
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// exported for the primitive-level tests
extern "C" int64_t st_scan_workspace_bytes(int64_t n) { return st_scan_ws_bytes(n); }
extern "C" int st_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total, void* ws, int64_t ws_bytes,
                           void* stream) {
    return st_exclusive_scan_u32(in, out, n, total, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int64_t st_sort_workspace_bytes(int64_t n) { return st_sort_ws_bytes(n); }
extern "C" int st_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int key_bits, void* ws, int64_t ws_bytes,
                                 void* stream) {
    return st_radix_sort_pairs_u32(keys, vals, n, key_bits, ws, ws_bytes, (hipStream_t)stream);
}

// One query for every workspace size (SURVEY.md section 8b `st_query_workspace`).
extern "C" int64_t st_voxelize_workspace_bytes(int64_t, int, int64_t);
extern "C" int64_t st_strided_workspace_bytes(int64_t);
extern "C" int64_t st_knn_workspace_bytes(int64_t);
extern "C" int64_t st_make_edges_workspace_bytes(int64_t);
extern "C" int64_t st_component_layout_workspace_bytes(int64_t);
extern "C" int64_t st_component_csr_workspace_bytes(int64_t);
extern "C" int64_t st_skeleton_workspace_bytes(int64_t, int64_t);
extern "C" int64_t st_query_workspace(int op, int64_t a, int64_t b, int64_t c) {
    switch (op) {
        case 0: return st_scan_ws_bytes(a);
        case 1: return st_sort_ws_bytes(a);
        case 2: return st_voxelize_workspace_bytes(a, (int)b, c);
        case 3: return st_strided_workspace_bytes(a);
        case 4: return st_knn_workspace_bytes(a);
        case 5: return st_make_edges_workspace_bytes(a);
        case 6: return st_component_layout_workspace_bytes(a);
        case 7: return st_component_csr_workspace_bytes(a);
        case 8: return st_skeleton_workspace_bytes(a, b);
        default: st_set_error("st_query_workspace: unknown op %d", op); return -1;
    }
}
