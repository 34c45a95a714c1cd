// Shared helpers for the smart-tree HIP kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#define ST_WAVE 64

// ---- status / error text (C ABI: every entry point returns int, 0 = ok) -------------------
enum StStatus : int {
    ST_OK = 0,
    ST_ERR_INVALID = -1,    // bad argument
    ST_ERR_WORKSPACE = -2,  // caller workspace too small
    ST_ERR_LAUNCH = -3,     // HIP launch / runtime error
};

void st_set_error(const char* fmt, ...);
// Wait until everything enqueued on `stream` has finished (every count read-back of the library goes through this).
void st_stream_wait(hipStream_t stream);

#define ST_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            st_set_error(__VA_ARGS__);   \
            return ST_ERR_INVALID;       \
        }                                \
    } while (0)

#define ST_CHECK_LAUNCH()                                                  \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            st_set_error("%s:%d HIP error: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return ST_ERR_LAUNCH;                                          \
        }                                                                  \
    } while (0)

#define ST_TRY(expr)              \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != ST_OK) return rc__; \
    } while (0)

// ---- bump allocator over the caller-provided workspace --------------------------------------
struct StArena {
    char* base;
    int64_t size;
    int64_t used;
    bool dry;  // size query: no pointers handed out
    StArena(void* p, int64_t n) : base((char*)p), size(n), used(0), dry(p == nullptr) {}
    template <class T>
    T* take(int64_t count) {
        int64_t bytes = ((int64_t)sizeof(T) * (count > 0 ? count : 1) + 255) & ~(int64_t)255;
        int64_t at = used;
        used += bytes;
        if (dry || used > size) return nullptr;
        return (T*)(base + at);
    }
    bool ok() const { return dry || used <= size; }
};

static inline int64_t st_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t st_next_pow2(int64_t v) {
    int64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ---- device helpers ---------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T st_min(T a, T b) { return a < b ? a : b; }
template <class T>
__device__ __forceinline__ T st_max(T a, T b) { return a > b ? a : b; }

// order-preserving float <-> uint map (for atomicMin/Max on floats)
__device__ __forceinline__ unsigned st_f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float st_ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// 64-bit voxel key: block (16 bit) | z | y | x (16 bit each); EMPTY = all ones
#define ST_EMPTY_KEY 0xffffffffffffffffull
__device__ __forceinline__ unsigned long long st_pack_key(int b, int z, int y, int x) {
    return ((unsigned long long)(unsigned)b << 48) | ((unsigned long long)(unsigned)z << 32) |
           ((unsigned long long)(unsigned)y << 16) | (unsigned long long)(unsigned)x;
}
__device__ __forceinline__ unsigned long long st_hash64(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// Home slot of a key.  The eight keys that differ only in the low three bits of x share one aligned group of eight
// slots (= one 64-byte line of keys[]): a 3x3x3 neighbourhood probe touches 9-12 lines instead of 27 scattered
// ones.  Collisions still resolve by linear probing.
__device__ __forceinline__ unsigned long long st_hash_slot(unsigned long long key, unsigned long long cap) {
    const unsigned long long hg = st_hash64(key >> 3);
    // the lane inside the group is x & 7 permuted by three hash bits: a wall of constant x must not load one lane only
    return ((hg << 3) | ((key ^ (hg >> 40)) & 7ull)) & (cap - 1);
}
// Next slot of the probe sequence: the key stays in its lane and jumps whole groups by a key-dependent odd
// stride (double hashing) -- with +1 steps a displaced x-run would pile onto its neighbours' groups (primary
// clustering: twice the probe length at 50 % load).  cap must be a power of two >= 16.
__device__ __forceinline__ unsigned long long st_hash_next(unsigned long long slot, unsigned long long key, unsigned long long cap) {
    const unsigned long long stride = ((st_hash64(key >> 3) >> 29) | 8ull) & ~7ull;  // odd number of groups
    return (slot + stride) & (cap - 1);
}

// Open-addressing table: keys[cap] (u64), vals[cap] (u32).  cap is a power of two (>= 8).
// insert-with-min: the smallest value ever offered for a key wins (deterministic).
// An insert gives up (returns false = "table full", the caller flags it and the host retries with a larger table) after
// ST_HASH_MAX_PROBE slots: below the 50 % load the callers size for, probe sequences are a handful of slots long, and
// without the limit every insert into a table that did fill up would walk ALL of it.
#define ST_HASH_MAX_PROBE 4096ull
__device__ __forceinline__ bool st_hash_insert_min(unsigned long long* keys, unsigned* vals, unsigned long long cap,
                                                   unsigned long long key, unsigned val) {
    unsigned long long slot = st_hash_slot(key, cap);
    for (unsigned long long probe = 0; probe < cap && probe < ST_HASH_MAX_PROBE; probe++) {
        unsigned long long prev = atomicCAS(&keys[slot], (unsigned long long)ST_EMPTY_KEY, key);
        if (prev == ST_EMPTY_KEY || prev == key) {
            atomicMin(&vals[slot], val);
            return true;
        }
        slot = st_hash_next(slot, key, cap);
    }
    return false;
}
// Same contract for a stream with MANY duplicates per key (voxelisation: ~9 points per voxel): look before
// touching the slot with atomics.  A key never changes once written and a value only decreases, so a stale read
// can only send us down the atomic path needlessly, never skip a needed update.
// give_up / give_up_bit (optional): a flag word another insert sets when the table is full -- a probe sequence that has
// grown long looks at it every 64 slots and stops (a full table would otherwise cost every insert its whole probe limit).
__device__ __forceinline__ bool st_hash_insert_min_dup(unsigned long long* keys, unsigned* vals, unsigned long long cap,
                                                       unsigned long long key, unsigned val, const unsigned* give_up = nullptr,
                                                       unsigned give_up_bit = 0u) {
    unsigned long long slot = st_hash_slot(key, cap);
    for (unsigned long long probe = 0; probe < cap && probe < ST_HASH_MAX_PROBE; probe++) {
        if (give_up && (probe & 63ull) == 63ull &&
            (__hip_atomic_load(give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & give_up_bit)) return false;
        unsigned long long prev = keys[slot];
        if (prev == ST_EMPTY_KEY) prev = atomicCAS(&keys[slot], (unsigned long long)ST_EMPTY_KEY, key);
        if (prev == ST_EMPTY_KEY || prev == key) {
            if (vals[slot] > val) atomicMin(&vals[slot], val);
            return true;
        }
        slot = st_hash_next(slot, key, cap);
    }
    return false;
}
__device__ __forceinline__ int st_hash_find(const unsigned long long* keys, const unsigned* vals, unsigned long long cap,
                                            unsigned long long key) {
    unsigned long long slot = st_hash_slot(key, cap);
    for (unsigned long long probe = 0; probe < cap && probe < ST_HASH_MAX_PROBE; probe++) {
        unsigned long long k = keys[slot];
        if (k == key) return (int)vals[slot];
        if (k == ST_EMPTY_KEY) return -1;
        slot = st_hash_next(slot, key, cap);
    }
    return -1;
}

// N look-ups whose memory round trips overlap: all home-slot keys are requested before the first is inspected, then
// all values of the hits (a thread that walks 26 neighbours one st_hash_find after the other waits for ~50 dependent
// loads; this way it waits for two).  Keys displaced from their home slot (rare below 50 % load) take the ordinary
// probe loop.  out[j] = value or -1; valid[j] == false -> -1 without touching memory.
template <int N>
__device__ __forceinline__ void st_hash_find_batch(const unsigned long long* keys, const unsigned* vals, unsigned long long cap,
                                                   const unsigned long long (&key)[N], const bool (&valid)[N], int (&out)[N]) {
    unsigned long long slot[N], got[N];
#pragma unroll
    for (int j = 0; j < N; j++) slot[j] = st_hash_slot(key[j], cap);
#pragma unroll
    for (int j = 0; j < N; j++) got[j] = valid[j] ? keys[slot[j]] : (unsigned long long)ST_EMPTY_KEY;
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (got[j] != key[j] && got[j] != ST_EMPTY_KEY) {  // displaced: follow the probe sequence
            unsigned long long s = slot[j], g = got[j];
            for (unsigned long long probe = 1; probe < cap && probe < ST_HASH_MAX_PROBE && g != key[j] && g != ST_EMPTY_KEY; probe++) {
                s = st_hash_next(s, key[j], cap);
                g = keys[s];
            }
            slot[j] = s;
            got[j] = g;
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = (valid[j] && got[j] == key[j]) ? (int)vals[slot[j]] : -1;
}

// exclusive scan of one value per thread across the workgroup; *total = workgroup sum.
// lds needs blockDim.x/64 + 1 words.  Must be reached by every thread of the workgroup.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();  // lds may still be read from a previous call
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t t = lane < nw ? lds[lane] : 0u;
        uint32_t s = t;
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(s, d);
            if (lane >= d) s += y;
        }
        if (lane < nw) lds[lane] = s - t;
        if (lane == nw - 1) lds[nw] = s;
    }
    __syncthreads();
    *total = lds[nw];
    return x - v + lds[wave];
}

// ---- internal primitives (prims.hip) --------------------------------------------------------------
int64_t st_scan_ws_bytes(int64_t n);
// out[i] = sum_{j<i} in[j]; if total != nullptr, *total = sum of all (device pointer). in may alias out.
// n_dev (optional, device): only the first min(n, *n_dev) elements are in use -- the launch is sized for n, the rest is skipped.
int st_exclusive_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total, void* ws, int64_t ws_bytes,
                          hipStream_t stream, const int64_t* n_dev = nullptr);
void st_fill_u32_dev(uint32_t* p, int64_t n, const int64_t* n_dev, uint32_t value, hipStream_t stream);
int64_t st_sort_ws_bytes(int64_t n);
// Stable LSD radix sort of (key, val) pairs on key bits [0, key_bits).  Result lands in keys/vals.
int st_radix_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int key_bits, void* ws, int64_t ws_bytes,
                            hipStream_t stream);
