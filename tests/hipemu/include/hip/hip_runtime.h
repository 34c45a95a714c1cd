// TEST INFRASTRUCTURE ONLY -- a CPU "sanitizer build" shim for the HIP kernel sources.
//
// tests/hipemu/build.py compiles smart_tree_amd/csrc/*.hip UNCHANGED with the host clang++ and
// this directory first on the include path, so `#include <hip/hip_runtime.h>` resolves here.
// Every workgroup runs as a set of cooperative fibers (ucontext) inside one OS thread:
// __syncthreads() and the wavefront collectives (__ballot/__shfl*/readfirstlane/MFMA) are
// rendezvous points.  This lets the `-m "not gpu"` suite execute the real kernel logic against
// the oracle (and under ASan/UBSan, which the GPU pool does not offer) before GPU minutes are
// spent.  It is NOT a product path: smart_tree_amd/_lib.py loads only libsmarttree_hip.so and
// raises if it is missing; nothing under smart_tree_amd/ references this directory.
//
// Limits (kernels are written to respect them): 1-D grids/blocks, static __shared__ only, no
// inter-workgroup communication inside a launch, wave collectives only in wave-convergent code.
#pragma once
#define ST_HIPEMU 1  // lets a source file skip host-only runtime calls the shim does not model

#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
typedef int hipEvent_t;
// one compute unit, workgroups one after the other: the library sizes its resident-workgroup schemes (helper workgroups) to zero
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 1; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline int2 make_int2(int x, int y) { return {x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

namespace hipemu {

enum State { READY = 0, AT_BARRIER = 1, AT_WAVEOP = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    int state;
    dim3 tid;
    uint64_t deposit[4];
};

struct Runtime {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    std::vector<char> stacks;
    Fiber* cur = nullptr;
    dim3 bid, bdim, gdim;
    std::function<void()> body;
    // snapshot of the last resolved wave op, per wave
    std::vector<uint64_t> snap;      // [nwaves][64][4]
    std::vector<uint64_t> snap_mask; // [nwaves]
    static constexpr size_t kStack = 256 * 1024;
};

inline Runtime& rt() {
    static Runtime r;
    return r;
}

inline void yield_to_sched(int state) {
    Runtime& r = rt();
    Fiber* f = r.cur;
    f->state = state;
    swapcontext(&f->ctx, &r.sched);
}

inline void fiber_entry() {
    Runtime& r = rt();
    r.body();
    r.cur->state = DONE;
    swapcontext(&r.cur->ctx, &r.sched);
}

inline int sched_order() {
    static const int v = [] { const char* e = getenv("HIPEMU_ORDER"); return e ? atoi(e) : 0; }();
    return v;
}

[[noreturn]] inline void die(const char* msg) {
    fprintf(stderr, "hipemu: %s\n", msg);
    abort();
}

inline void run_block(unsigned nthreads) {
    Runtime& r = rt();
    if (r.fibers.size() < nthreads) r.fibers.resize(nthreads);
    if (r.stacks.size() < (size_t)nthreads * Runtime::kStack) r.stacks.resize((size_t)nthreads * Runtime::kStack);
    unsigned nwaves = (nthreads + 63) / 64;
    r.snap.assign((size_t)nwaves * 64 * 4, 0);
    r.snap_mask.assign(nwaves, 0);
    for (unsigned t = 0; t < nthreads; t++) {
        Fiber& f = r.fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = r.stacks.data() + (size_t)t * Runtime::kStack;
        f.ctx.uc_stack.ss_size = Runtime::kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        f.state = READY;
        f.tid = dim3(t);
    }
    const int order = sched_order();
    uint64_t lcg = 0x9e3779b97f4a7c15ull * (uint64_t)(order + 1) + r.bid.x;
    for (;;) {
        // HIPEMU_ORDER: 0 ascending (default), 1 descending, >= 2 a pseudo-random rotation + direction per pass.
        // Results must not depend on it -- a cheap detector for races between lanes / waves of a workgroup.
        unsigned start = 0;
        bool down = order == 1;
        if (order >= 2) {
            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
            start = (unsigned)((lcg >> 33) % nthreads);
            down = (lcg >> 20) & 1;
        }
        for (unsigned i = 0; i < nthreads; i++) {
            const unsigned t = down ? (start + nthreads - i) % nthreads : (start + i) % nthreads;
            Fiber& f = r.fibers[t];
            if (f.state != READY) continue;
            r.cur = &f;
            swapcontext(&r.sched, &f.ctx);
        }
        // everyone is now parked or finished
        bool released = false;
        unsigned at_barrier = 0, done = 0;
        for (unsigned w = 0; w < nwaves; w++) {
            unsigned lo = w * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads;
            unsigned nop = 0, nbar = 0, ndone = 0;
            for (unsigned t = lo; t < hi; t++) {
                int s = r.fibers[t].state;
                nop += s == AT_WAVEOP;
                nbar += s == AT_BARRIER;
                ndone += s == DONE;
            }
            at_barrier += nbar;
            done += ndone;
            if (nop == 0) continue;
            if (nbar != 0) die("wave collective reached by some lanes while others wait at __syncthreads (divergent wave op)");
            uint64_t mask = 0;
            for (unsigned t = lo; t < hi; t++) {
                Fiber& f = r.fibers[t];
                if (f.state != AT_WAVEOP) continue;
                unsigned lane = t - lo;
                mask |= 1ull << lane;
                for (int d = 0; d < 4; d++) r.snap[((size_t)w * 64 + lane) * 4 + d] = f.deposit[d];
                f.state = READY;
            }
            r.snap_mask[w] = mask;
            released = true;
        }
        if (released) continue;
        if (done == nthreads) break;
        if (at_barrier + done == nthreads) {
            for (unsigned t = 0; t < nthreads; t++)
                if (r.fibers[t].state == AT_BARRIER) r.fibers[t].state = READY;
            continue;
        }
        die("deadlock: no runnable fiber");
    }
    r.cur = nullptr;
}

template <class F>
inline void launch(dim3 grid, dim3 block, F&& body) {
    Runtime& r = rt();
    if (grid.z != 1 || block.y != 1 || block.z != 1) die("only launches with 1-D workgroups and 2-D grids are emulated");
    r.gdim = grid;
    r.bdim = block;
    r.body = body;
    for (unsigned j = 0; j < grid.y; j++)
        for (unsigned i = 0; i < grid.x; i++) {
            r.bid = dim3(sched_order() == 0 ? i : grid.x - 1 - i, sched_order() == 0 ? j : grid.y - 1 - j);
            run_block(block.x);
        }
}

// wave rendezvous: deposit up to four 64-bit words, get everyone's words back
struct WaveView {
    const uint64_t* data;
    uint64_t mask;
    unsigned lane;
    uint64_t lo(unsigned l) const { return data[l * 4]; }
    uint64_t hi(unsigned l) const { return data[l * 4 + 1]; }
    uint64_t word(unsigned l, unsigned d) const { return data[l * 4 + d]; }
};

inline WaveView wave_exchange(uint64_t a, uint64_t b = 0, uint64_t c = 0, uint64_t d = 0) {
    Runtime& r = rt();
    Fiber* f = r.cur;
    f->deposit[0] = a;
    f->deposit[1] = b;
    f->deposit[2] = c;
    f->deposit[3] = d;
    yield_to_sched(AT_WAVEOP);
    unsigned t = f->tid.x, w = t / 64;
    return WaveView{r.snap.data() + (size_t)w * 64 * 4, r.snap_mask[w], t % 64};
}

template <class T>
inline uint64_t to_bits(T v) {
    static_assert(sizeof(T) <= 8, "");
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    return u;
}
template <class T>
inline T from_bits(uint64_t u) {
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
}

}  // namespace hipemu

#define threadIdx (hipemu::rt().cur->tid)
#define blockIdx (hipemu::rt().bid)
#define blockDim (hipemu::rt().bdim)
#define gridDim (hipemu::rt().gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ((void)(stream), (void)(shmem), hipemu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); }))

inline void __syncthreads() { hipemu::yield_to_sched(hipemu::AT_BARRIER); }
inline void __threadfence() {}
inline void __threadfence_block() {}

// ---- wave collectives ------------------------------------------------------------------
inline unsigned long long __ballot(int pred) {
    auto v = hipemu::wave_exchange(pred ? 1 : 0);
    unsigned long long m = 0;
    for (unsigned l = 0; l < 64; l++)
        if ((v.mask >> l) & 1 && v.lo(l)) m |= 1ull << l;
    return m;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) {
    auto v = hipemu::wave_exchange(pred ? 1 : 0);
    for (unsigned l = 0; l < 64; l++)
        if ((v.mask >> l) & 1 && !v.lo(l)) return 0;
    return 1;
}
template <class T>
inline T __shfl(T x, int src, int width = 64) {
    auto v = hipemu::wave_exchange(hipemu::to_bits(x));
    unsigned base = v.lane & ~(unsigned)(width - 1);
    unsigned s = base + ((unsigned)src & (unsigned)(width - 1));
    return hipemu::from_bits<T>(v.lo(s));
}
template <class T>
inline T __shfl_up(T x, unsigned delta, int width = 64) {
    auto v = hipemu::wave_exchange(hipemu::to_bits(x));
    unsigned in = v.lane & (unsigned)(width - 1);
    return in >= delta ? hipemu::from_bits<T>(v.lo(v.lane - delta)) : x;
}
template <class T>
inline T __shfl_down(T x, unsigned delta, int width = 64) {
    auto v = hipemu::wave_exchange(hipemu::to_bits(x));
    unsigned in = v.lane & (unsigned)(width - 1);
    return in + delta < (unsigned)width ? hipemu::from_bits<T>(v.lo(v.lane + delta)) : x;
}
template <class T>
inline T __shfl_xor(T x, int m, int width = 64) {
    auto v = hipemu::wave_exchange(hipemu::to_bits(x));
    unsigned s = v.lane ^ (unsigned)m;
    (void)width;
    return hipemu::from_bits<T>(v.lo(s));
}
inline int __builtin_amdgcn_readfirstlane(int x) {
    auto v = hipemu::wave_exchange((uint64_t)(uint32_t)x);
    return (int)(uint32_t)v.lo(__builtin_ctzll(v.mask));
}
inline int __builtin_amdgcn_readlane(int x, int src) {
    auto v = hipemu::wave_exchange((uint64_t)(uint32_t)x);
    return (int)(uint32_t)v.lo((unsigned)src & 63u);
}
// hardware: a scheduling fence (lanes run in lockstep); here lanes are fibers, so it must be a rendezvous
inline void __builtin_amdgcn_wave_barrier() { (void)hipemu::wave_exchange((uint64_t)0); }
inline unsigned __lane_id() { return threadIdx.x % 64; }

typedef float hipemu_v4f __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: A[i][k] held by lane k*16+i, B[k][j] by lane k*16+j,
// D[row][col]: col = lane&15, row = (lane>>4)*4 + reg; k-ordered fmaf chain (guide section 3).
inline hipemu_v4f __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_v4f c, int, int, int) {
    auto v = hipemu::wave_exchange(hipemu::to_bits(a), hipemu::to_bits(b));
    unsigned col = v.lane & 15;
    for (int reg = 0; reg < 4; reg++) {
        unsigned row = (v.lane >> 4) * 4 + reg;
        float acc = c[reg];
        for (unsigned k = 0; k < 4; k++) {
            float av = hipemu::from_bits<float>(v.lo(k * 16 + row));
            float bv = hipemu::from_bits<float>(v.hi(k * 16 + col));
            acc = fmaf(av, bv, acc);
        }
        c[reg] = acc;
    }
    return c;
}

// v_mfma_f32_16x16x16_f16: A[i][k] = element k%4 of lane (k/4)*16 + i, B[k][j] = element k%4 of lane (k/4)*16 + j, D as above.
// Products of two halves are exact in float32; the hardware's summation order is not documented, this emulation adds in
// k order -- tests of the fp16 path compare against a float64 oracle with a tolerance, never bit for bit.
typedef _Float16 hipemu_v4h __attribute__((ext_vector_type(4)));
inline hipemu_v4f __builtin_amdgcn_mfma_f32_16x16x16f16(hipemu_v4h a, hipemu_v4h b, hipemu_v4f c, int, int, int) {
    uint64_t ab, bb;
    memcpy(&ab, &a, 8);
    memcpy(&bb, &b, 8);
    auto v = hipemu::wave_exchange(ab, bb);
    auto half_at = [](uint64_t word, unsigned e) {
        uint16_t h = (uint16_t)(word >> (16 * e));
        _Float16 f;
        memcpy(&f, &h, 2);
        return (float)f;
    };
    unsigned col = v.lane & 15;
    for (int reg = 0; reg < 4; reg++) {
        unsigned row = (v.lane >> 4) * 4 + reg;
        float acc = c[reg];
        for (unsigned k = 0; k < 16; k++)
            acc = fmaf(half_at(v.lo((k / 4) * 16 + row), k % 4), half_at(v.hi((k / 4) * 16 + col), k % 4), acc);
        c[reg] = acc;
    }
    return c;
}

// v_mfma_f32_16x16x32_bf16 (gfx950): lane (i = l & 15, g = l >> 4) holds eight bf16 of A row i / B column i: k = 8g + e.
// Products of two bf16 are exact in float32; added in k order (tests compare with a float64 oracle under a tolerance).
typedef __bf16 hipemu_bf8 __attribute__((ext_vector_type(8)));
inline hipemu_v4f __builtin_amdgcn_mfma_f32_16x16x32_bf16(hipemu_bf8 a, hipemu_bf8 b, hipemu_v4f c, int, int, int) {
    uint64_t aw[2], bw[2];
    memcpy(aw, &a, 16);
    memcpy(bw, &b, 16);
    auto v = hipemu::wave_exchange(aw[0], aw[1], bw[0], bw[1]);  // words 0-1: A elements 0-3 / 4-7, words 2-3: B
    auto bf_at = [](uint64_t word, unsigned e) {
        const uint32_t bits = (uint32_t)((word >> (16 * e)) & 0xffffu) << 16;
        float f;
        memcpy(&f, &bits, 4);
        return f;
    };
    unsigned col = v.lane & 15;
    for (int reg = 0; reg < 4; reg++) {
        unsigned row = (v.lane >> 4) * 4 + reg;
        float acc = c[reg];
        for (unsigned k = 0; k < 32; k++) {
            const unsigned g = k / 8, e = k % 8;
            acc = fmaf(bf_at(v.word(g * 16 + row, e / 4), e % 4), bf_at(v.word(g * 16 + col, 2 + e / 4), e % 4), acc);
        }
        c[reg] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x32_f16 (gfx950): as the bf16 form, eight halves per lane.
typedef _Float16 hipemu_v8h __attribute__((ext_vector_type(8)));
inline hipemu_v4f __builtin_amdgcn_mfma_f32_16x16x32_f16(hipemu_v8h a, hipemu_v8h b, hipemu_v4f c, int, int, int) {
    uint64_t aw[2], bw[2];
    memcpy(aw, &a, 16);
    memcpy(bw, &b, 16);
    auto v = hipemu::wave_exchange(aw[0], aw[1], bw[0], bw[1]);
    auto half_at = [](uint64_t word, unsigned e) {
        uint16_t h = (uint16_t)(word >> (16 * e));
        _Float16 f;
        memcpy(&f, &h, 2);
        return (float)f;
    };
    unsigned col = v.lane & 15;
    for (int reg = 0; reg < 4; reg++) {
        unsigned row = (v.lane >> 4) * 4 + reg;
        float acc = c[reg];
        for (unsigned k = 0; k < 32; k++) {
            const unsigned g = k / 8, e = k % 8;
            acc = fmaf(half_at(v.word(g * 16 + row, e / 4), e % 4), half_at(v.word(g * 16 + col, 2 + e / 4), e % 4), acc);
        }
        c[reg] = acc;
    }
    return c;
}
// v_perm_b32: byte i of the result = byte sel[i] of the 8-byte value {s0 (bytes 4-7), s1 (bytes 0-3)}; 0x0c = 0x00
inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
    const uint64_t v = ((uint64_t)s0 << 32) | (uint64_t)s1;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned b = (sel >> (8 * i)) & 0xffu;
        const unsigned byte = b < 8u ? (unsigned)((v >> (8 * b)) & 0xffu) : 0u;
        r |= byte << (8 * i);
    }
    return r;
}

// ---- bit / math helpers ----------------------------------------------------------------
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
inline unsigned __float_as_uint(float f) { return hipemu::from_bits<unsigned>(hipemu::to_bits(f)); }
inline int __float_as_int(float f) { return hipemu::from_bits<int>(hipemu::to_bits(f)); }
inline float __uint_as_float(unsigned u) { return hipemu::from_bits<float>((uint64_t)u); }
inline float __int_as_float(int u) { return hipemu::from_bits<float>((uint64_t)(uint32_t)u); }
inline float __expf(float x) { return expf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }

#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_exchange(p, v, order, scope) __atomic_exchange_n((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), (order))
#define __hip_atomic_fetch_and(p, v, order, scope) __atomic_fetch_and((p), (v), (order))
#define __hip_atomic_fetch_max(p, v, order, scope) hipemu_fetch_max((p), (v))
template <class T> inline T hipemu_fetch_max(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
#define __hip_atomic_fetch_min(p, v, order, scope) hipemu_fetch_min((p), (v))
template <class T> inline T hipemu_fetch_min(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
inline long long wall_clock64() { return 0; }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)

// ---- atomics (fibers never run concurrently, plain RMW is exact) -------------------------
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
