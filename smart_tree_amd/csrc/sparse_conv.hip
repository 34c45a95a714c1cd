// st_sparse_conv_fwd / st_pointwise_mlp_heads: the network's arithmetic.
//
// One kernel family serves every conv of the reference graph:
//   SubMConv3d k3   (ResBlock,       smart_tree/model/model_blocks.py:134-143)  nbr = subm table, K = 27
//   SparseConv3d    (EncoderBlock,   :57-70)                                    nbr = nbr_down,   K = 27
//   SparseInverse   (DecoderBlock,   :90-101)                                   nbr = nbr_up,     K = 27
//   SubMConv3d k1   (input conv :23-35, ResBlock identity :123-131)             nbr = NULL,       K = 1
// Output-stationary: one lane owns one output voxel and a tile of COT output channels, walks the
// K kernel offsets in ascending order and accumulates with fmaf -- no scatter, no atomics, so
// results are bit-reproducible run to run.  Fused around the accumulation:
//   prologue  channel concat cat(skip, decoded) (UBlock.forward, :238-240) read from two tensors
//   epilogue  eval-mode BatchNorm1d as y*scale+shift (:33,68,99,138,143), residual add and ReLU
//             (ResBlock.forward, :149-156)
// Weights are pre-laid out [K][Cin][Cout]; a workgroup handles ONE cout tile, so every weight
// address is wave-uniform and the compiler feeds the FMAs from scalar loads (SGPR operands).
#include "st_common.h"

#define CONV_BLOCK 256

// fp16 storage (config 5: levels whose channel count is a multiple of 16 keep their features in half precision):
// TIN / TOUT = float or st_h; the arithmetic is always float32.
typedef _Float16 st_h;
typedef _Float16 st_v4h __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 conv_load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 conv_load4(const st_h* p) {
    const st_v4h v = *reinterpret_cast<const st_v4h*>(p);
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
__device__ __forceinline__ void conv_store4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
__device__ __forceinline__ void conv_store4(st_h* p, float a, float b, float c, float d) {
    *reinterpret_cast<st_v4h*>(p) = st_v4h{(st_h)a, (st_h)b, (st_h)c, (st_h)d};  // round to nearest even
}

// row_order entries: bits 0-27 = output row.  Top four bits 0 = nothing known; 8 + c = the row belongs to coordinate
// parity class c = (z&1)<<2 | (y&1)<<1 | (x&1) of an inverse k3-s2-p1 convolution: per axis an even coordinate pairs
// through k = 1 only, an odd one through k = 0 and 2, so at most 8 of the 27 table entries can be >= 0 and the rest
// need not be read.  Returns the 27-bit mask of offsets worth reading.
__device__ __forceinline__ uint32_t conv_live_offsets(int32_t entry, int K) {
    const uint32_t tag = (uint32_t)entry >> 28;
    if (!(tag & 8u) || K != 27) return 0xffffffffu;
    const uint32_t ax[3] = {(tag & 4u) ? 5u : 2u, (tag & 2u) ? 5u : 2u, (tag & 1u) ? 5u : 2u};  // z, y, x: {0,2} or {1}
    uint32_t live = 0;
#pragma unroll
    for (int k = 0; k < 27; k++)
        if (((ax[0] >> (k / 9)) & (ax[1] >> ((k / 3) % 3)) & (ax[2] >> (k % 3))) & 1u) live |= 1u << k;
    return live;
}
#define CONV_ROW_MASK 0x0fffffff

template <int CIN, int COT, class TIN = float, class TOUT = float>
__global__ void __launch_bounds__(CONV_BLOCK) k_sparse_conv(const TIN* __restrict__ x0, int c0,
                                                            const TIN* __restrict__ x1, const int32_t* __restrict__ nbr,
                                                            int K, int64_t n_out, int64_t nstride, const float* __restrict__ w, int cout,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ residual, int relu,
                                                            TOUT* __restrict__ y, const int32_t* __restrict__ row_order) {
    const int co_tiles = cout / COT;
    const int co0 = (int)(blockIdx.x % co_tiles) * COT;
    const int64_t pos = (int64_t)(blockIdx.x / co_tiles) * CONV_BLOCK + threadIdx.x;
    const bool active = pos < n_out;
    // row_order (optional): the output rows in an order that makes the live offsets wave-uniform (see st_sparse_conv_fwd)
    const int32_t entry = active && row_order ? row_order[pos] : 0;
    const int64_t o = active && row_order ? (int64_t)(entry & CONV_ROW_MASK) : pos;
    const uint32_t live = conv_live_offsets(entry, K);
    const int c1 = CIN - c0;
    float acc[COT];
#pragma unroll
    for (int c = 0; c < COT; c++) acc[c] = 0.0f;

    if (CIN % 4 == 0 && CIN <= 16) {
        // Offsets in groups of G: all neighbour indices of the group first, then all row gathers, then the FMAs (still in
        // ascending k, ci order -- bit-identical sums).  A lane that walks 27 offsets one by one waits for 27 dependent
        // index -> row round trips; at the finest level there are only 2-3 wavefronts per SIMD to hide them.
        constexpr int G = CIN <= 8 ? 9 : 3, Q = CIN / 4;
        for (int k0 = 0; k0 < K; k0 += G) {
            int idx[G];
            float4 v[G][Q];
#pragma unroll
            for (int j = 0; j < G; j++) {
                const int k = k0 + j;
                idx[j] = -1;
                if (k < K && active && ((live >> k) & 1u)) idx[j] = nbr ? nbr[(int64_t)k * nstride + o] : (int)o;
            }
#pragma unroll
            for (int j = 0; j < G; j++)
                if (idx[j] >= 0) {
#pragma unroll
                    for (int q = 0; q < Q; q++) {
                        const int ci = 4 * q;
                        const TIN* row = ci < c0 ? x0 + (int64_t)idx[j] * c0 + ci : x1 + (int64_t)idx[j] * c1 + (ci - c0);
                        v[j][q] = conv_load4(row);
                    }
                }
#pragma unroll
            for (int j = 0; j < G; j++) {
                if (idx[j] < 0) continue;
                const float* __restrict__ wk = w + (int64_t)(k0 + j) * CIN * cout + co0;
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const float xs[4] = {v[j][q].x, v[j][q].y, v[j][q].z, v[j][q].w};
#pragma unroll
                    for (int jj = 0; jj < 4; jj++)
#pragma unroll
                        for (int c = 0; c < COT; c++) acc[c] = fmaf(xs[jj], wk[(4 * q + jj) * cout + c], acc[c]);
                }
            }
        }
    } else
    for (int k = 0; k < K; k++) {
        int idx = -1;
        if (active && ((live >> k) & 1u)) idx = nbr ? nbr[(int64_t)k * nstride + o] : (int)o;
        if (idx < 0) continue;
        const float* __restrict__ wk = w + (int64_t)k * CIN * cout + co0;
        if (CIN % 4 == 0) {
#pragma unroll 2
            for (int ci = 0; ci < CIN; ci += 4) {
                const TIN* row = ci < c0 ? x0 + (int64_t)idx * c0 + ci : x1 + (int64_t)idx * c1 + (ci - c0);
                const float4 v = conv_load4(row);
                const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int c = 0; c < COT; c++) acc[c] = fmaf(xs[j], wk[(ci + j) * cout + c], acc[c]);
            }
        } else {
            for (int ci = 0; ci < CIN; ci++) {
                const float xv = (float)x0[(int64_t)idx * CIN + ci];
#pragma unroll
                for (int c = 0; c < COT; c++) acc[c] = fmaf(xv, wk[ci * cout + c], acc[c]);
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int c = 0; c < COT; c++) {
        float v = acc[c];
        if (scale) v = fmaf(v, scale[co0 + c], shift[co0 + c]);
        if (residual) v += residual[o * cout + co0 + c];
        if (relu) v = v > 0.0f ? v : 0.0f;
        acc[c] = v;
    }
    TOUT* out = y + o * cout + co0;
    if (COT % 4 == 0) {
#pragma unroll
        for (int c = 0; c < COT; c += 4) conv_store4(out + c, acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    } else {
#pragma unroll
        for (int c = 0; c < COT; c++) out[c] = (TOUT)acc[c];
    }
}

// Any (Cin, Cout): one lane per output row and four output channels, plain loops.  Same order of operations per output element
// as the instantiated kernels (offsets ascending, input channels ascending, one fmaf each): the same bits where both exist.
__global__ void __launch_bounds__(CONV_BLOCK) k_sparse_conv_any(const float* __restrict__ x0, int c0, const float* __restrict__ x1, int cin,
                                                                const int32_t* __restrict__ nbr, int K, int64_t n_out, int64_t nstride,
                                                                const float* __restrict__ w, int cout, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ residual, int relu,
                                                                float* __restrict__ y, const int32_t* __restrict__ row_order) {
    const int64_t pos = (int64_t)blockIdx.x * CONV_BLOCK + threadIdx.x;
    if (pos >= n_out) return;
    const int co0 = (int)blockIdx.y * 4, nco = cout - co0 < 4 ? cout - co0 : 4;
    const int32_t entry = row_order ? row_order[pos] : 0;
    const int64_t o = row_order ? (int64_t)(entry & CONV_ROW_MASK) : pos;
    const uint32_t live = conv_live_offsets(entry, K);
    const int c1 = cin - c0;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < K; k++) {
        if (K <= 32 && !((live >> k) & 1u)) continue;
        const int idx = nbr ? nbr[(int64_t)k * nstride + o] : (int)o;
        if (idx < 0) continue;
        const float* __restrict__ wk = w + (int64_t)k * cin * cout + co0;
        for (int ci = 0; ci < cin; ci++) {
            const float xv = ci < c0 ? x0[(int64_t)idx * c0 + ci] : x1[(int64_t)idx * c1 + (ci - c0)];
            for (int c = 0; c < nco; c++) acc[c] = fmaf(xv, wk[(int64_t)ci * cout + c], acc[c]);
        }
    }
    for (int c = 0; c < nco; c++) {
        float v = acc[c];
        if (scale) v = fmaf(v, scale[co0 + c], shift[co0 + c]);
        if (residual) v += residual[o * cout + co0 + c];
        if (relu) v = v > 0.0f ? v : 0.0f;
        y[o * cout + co0 + c] = v;
    }
}

template <int CIN, int COT, class TIN = float, class TOUT = float>
static int conv_launch(const TIN* x0, int c0, const TIN* x1, const int32_t* nbr, int K, int64_t n_out, int64_t nstride, const float* w,
                       int cout, const float* scale, const float* shift, const float* residual, int relu, TOUT* y,
                       hipStream_t stream, const int32_t* row_order = nullptr) {
    int64_t blocks = st_div_up(n_out, CONV_BLOCK) * (cout / COT);
    hipLaunchKernelGGL((k_sparse_conv<CIN, COT, TIN, TOUT>), dim3((unsigned)blocks), dim3(CONV_BLOCK), 0, stream, x0, c0, x1, nbr, K,
                       n_out, nstride, w, cout, scale, shift, residual, relu, y, row_order);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ------------------------------------------------------------------------- MFMA rule-GEMM ---
// For Cin, Cout multiples of 16 the per-offset contraction [16 voxels x Cin] . [Cin x Cout] runs on the
// matrix cores with v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate; NOT bit-identical to the vector kernel's fmaf chain -- measured in round 2: last-bit differences --
// so a layer must use the same kernel family at every size,
// cdna_hip_programming.md section 3), still output-stationary and atomics-free:
//   wave   = MF_RT row tiles of 16 output voxels x all Cout (Cout/16 column tiles), accumulators in VGPRs
//   A      = gathered input rows: lane (i = l&15, kg = l>>4) loads ONE float4 = channels 16c+4kg .. +3 of row i
//            (a 64 B contiguous piece per row and instruction); register s of it feeds MFMA step s
//   B      = W_k staged once per offset and workgroup in LDS in the matching order
//            wp[k][c][kg][co][s] = W[k][16c + 4kg + s][co]  (host pre-permuted: a straight float4 copy),
//            read with one ds_read_b128 per (c, column tile)
//   D      = lane l holds rows (l>>4)*4 + r, column l&15 of each 16x16 tile: epilogue = BN affine, residual,
//            ReLU, 64 B row segments stored per instruction.
typedef float st_v4f __attribute__((ext_vector_type(4)));
#define MF_BLOCK 256

// RT = row tiles (of 16 output voxels) per wave; LDSW = stage W_k through LDS (one copy per workgroup and
// offset, two barriers) or let every lane fetch its B float4 straight from the L2-resident weights (no
// barrier, loads free to run ahead of the MFMAs).
template <int CIN, int COUT, int RT, bool LDSW>
__global__ void __launch_bounds__(MF_BLOCK) k_sparse_conv_mfma(const float* __restrict__ x0, int c0, const float* __restrict__ x1,
                                                               const int32_t* __restrict__ nbr, int K, int64_t n_out, int64_t nstride,
                                                               const float* __restrict__ wp, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ residual,
                                                               int relu, float* __restrict__ y, const int32_t* __restrict__ row_order) {
    constexpr int CT = COUT / 16, NC = CIN / 16;
    __shared__ float4 wl[LDSW ? CIN * COUT / 4 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kg = lane >> 4;
    const int64_t obase = ((int64_t)blockIdx.x * (MF_BLOCK / 64) + wave) * (16 * RT);
    const int c1 = CIN - c0;
    st_v4f acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[t][ct] = st_v4f{0.0f, 0.0f, 0.0f, 0.0f};

    int64_t orow[RT];  // output row of tile position i16 (identity unless row_order is given)
    uint32_t live[RT];
#pragma unroll
    for (int t = 0; t < RT; t++) {
        const int64_t pos = obase + t * 16 + i16;
        const int32_t entry = pos < n_out && row_order ? row_order[pos] : 0;
        orow[t] = pos < n_out ? (row_order ? (int64_t)(entry & CONV_ROW_MASK) : pos) : -1;
        live[t] = conv_live_offsets(entry, K);
    }
    for (int k = 0; k < K; k++) {
        const float4* wsrc = reinterpret_cast<const float4*>(wp + (int64_t)k * CIN * COUT);
        if (LDSW) {
            __syncthreads();  // everyone is done with the previous offset's weights
            for (int i = tid; i < CIN * COUT / 4; i += MF_BLOCK) wl[i] = wsrc[i];
        }
        int idx[RT];
        bool any = false;
#pragma unroll
        for (int t = 0; t < RT; t++) {
            idx[t] = orow[t] >= 0 && ((live[t] >> k) & 1u) ? (nbr ? nbr[(int64_t)k * nstride + orow[t]] : (int)orow[t]) : -1;
            any = any || idx[t] >= 0;
        }
        if (LDSW) __syncthreads();
        if (__ballot(any) == 0ull) continue;  // no voxel of this wave has a neighbour at offset k (wave-uniform)
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int ci = 16 * c + 4 * kg;
            float4 av[RT];
#pragma unroll
            for (int t = 0; t < RT; t++) {
                av[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (idx[t] >= 0) {
                    const float* row = ci < c0 ? x0 + (int64_t)idx[t] * c0 + ci : x1 + (int64_t)idx[t] * c1 + (ci - c0);
                    av[t] = *reinterpret_cast<const float4*>(row);
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const int wi = (c * 4 + kg) * COUT + ct * 16 + i16;
                const float4 bv = LDSW ? wl[wi] : wsrc[wi];
#pragma unroll
                for (int t = 0; t < RT; t++) {
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].x, bv.x, acc[t][ct], 0, 0, 0);
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].y, bv.y, acc[t][ct], 0, 0, 0);
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].z, bv.z, acc[t][ct], 0, 0, 0);
                    acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].w, bv.w, acc[t][ct], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int ch = ct * 16 + i16;
        const float sc = scale ? scale[ch] : 1.0f, sh = scale ? shift[ch] : 0.0f;
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t pos = obase + t * 16 + kg * 4 + r;
                if (pos >= n_out) continue;
                const int64_t o = row_order ? (int64_t)(row_order[pos] & CONV_ROW_MASK) : pos;
                float v = acc[t][ct][r];
                if (scale) v = fmaf(v, sc, sh);
                if (residual) v += residual[o * COUT + ch];
                if (relu) v = v > 0.0f ? v : 0.0f;
                y[o * COUT + ch] = v;
            }
    }
}

template <int CIN, int COUT, int RT, bool LDSW>
static void conv_launch_mfma_v(const float* x0, int c0, const float* x1, const int32_t* nbr, int K, int64_t n_out, int64_t nstride, const float* wp,
                               const float* scale, const float* shift, const float* residual, int relu, float* y,
                               hipStream_t stream, const int32_t* row_order) {
    const int64_t blocks = st_div_up(n_out, (MF_BLOCK / 64) * 16 * RT);
    hipLaunchKernelGGL((k_sparse_conv_mfma<CIN, COUT, RT, LDSW>), dim3((unsigned)blocks), dim3(MF_BLOCK), 0, stream, x0, c0, x1,
                       nbr, K, n_out, nstride, wp, scale, shift, residual, relu, y, row_order);
}

template <int CIN, int COUT>
static int conv_launch_mfma(const float* x0, int c0, const float* x1, const int32_t* nbr, int K, int64_t n_out, int64_t nstride, const float* wp,
                            const float* scale, const float* shift, const float* residual, int relu, float* y,
                            hipStream_t stream, const int32_t* row_order, int variant) {
    int v = variant;  // 0: by size (below); else RT | (LDSW << 4): tools/bench_conv.py times the variants
    // measured on MI355X (tools/bench_conv.py, profiles/r02_conv_variants_batch8.txt): one row tile per wave with the weights
    // straight from L2 wins while a level has too few rows to fill the chip (one cloud: <= 90k rows below level 0) and for the
    // parity-ordered inverse convs; from ~150k rows on (a batch of clouds, the 5M-point cloud) two row tiles per wave with
    // W_k staged once per workgroup in LDS is 5-20 % faster (B fragments reused, a quarter of the weight traffic from L2)
    if (v == 0) v = (row_order == nullptr && n_out >= 150000) ? 18 : 1;
#define MFMA_V(RT_, L_) conv_launch_mfma_v<CIN, COUT, RT_, L_>(x0, c0, x1, nbr, K, n_out, nstride, wp, scale, shift, residual, relu, y, stream, row_order)
    switch (v) {
        case 1: MFMA_V(1, false); break;
        case 2: MFMA_V(2, false); break;
        case 4: MFMA_V(4, false); break;
        case 17: MFMA_V(1, true); break;
        default: MFMA_V(2, true); break;
    }
#undef MFMA_V
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ------------------------------------------------------------- split-bf16 rule-GEMM (round 3) ---
// The f32 matrix instruction runs at the vector rate (v_mfma_f32_16x16x4_f32: 32 cycles for 2048 flops), and the 32- and
// 64-channel levels keep that pipe 49-68 % busy.  Here the SAME float32 features and weights are contracted on the bf16 pipe
// (v_mfma_f32_16x16x32_bf16: ~17 cycles for 16384 flops) without giving up float32 accuracy: a float32 is cut into three bf16
// pieces by TRUNCATION -- hi = the upper 16 bits, mid = the upper 16 bits of a - hi, lo = those of a - hi - mid -- which splits
// its 24-bit significand into 8 + 8 + 8 bits, so a = hi + mid + lo EXACTLY.  A product a*b is the sum of nine bf16 products
// (each exact in float32); the six largest are issued -- hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -- and the three that are
// dropped (mid*lo, lo*mid, lo*lo) are below 2^-24 of |a*b|: the result differs from a float32 FMA chain by rounding-order noise
// only (measured against the float64 oracle: the same 5e-7 as the f32 matrix-core kernel).  Six instructions of the 15x faster
// kind per 32 channels instead of eight of the slow kind: 2.5x less matrix-pipe time; the split costs ~44 vector instructions
// per eight gathered channels and row tile, amortised over the Cout / 16 column tiles.  Features stay float32 in memory: no new
// tensor format, every other kernel is untouched.  Weights: host-side split into the same three planes, operand order
//   wq[k][c][plane][g][co][e] = piece `plane` of W[k][32c + 8g + e][co]   (bf16, a 16-byte vector per (k, c, plane, g, co))
// lane (i = l & 15, g = l >> 4) feeds channels 32c + 8g .. +7 of row i (A: two float4 loads) and of output column i (B).
typedef __bf16 st_bf8 __attribute__((ext_vector_type(8)));
struct StB3 { uint4 h, m, l; };
// (upper halves of a1, a0) packed as two bf16: v_perm_b32 {a1.b3, a1.b2, a0.b3, a0.b2}
__device__ __forceinline__ unsigned b3_pack_hi(unsigned u0, unsigned u1) { return __builtin_amdgcn_perm(u1, u0, 0x07060302u); }
__device__ __forceinline__ void b3_split_pair(float a0, float a1, unsigned* h, unsigned* m, unsigned* l) {
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);
    *h = b3_pack_hi(u0, u1);
    const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u);  // exact
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    *m = b3_pack_hi(v0, v1);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);  // exact
    *l = b3_pack_hi(__float_as_uint(s0), __float_as_uint(s1));
}
__device__ __forceinline__ StB3 b3_split8(float4 a, float4 b) {
    StB3 o;
    b3_split_pair(a.x, a.y, &o.h.x, &o.m.x, &o.l.x);
    b3_split_pair(a.z, a.w, &o.h.y, &o.m.y, &o.l.y);
    b3_split_pair(b.x, b.y, &o.h.z, &o.m.z, &o.l.z);
    b3_split_pair(b.z, b.w, &o.h.w, &o.m.w, &o.l.w);
    return o;
}
__device__ __forceinline__ st_bf8 b3_bf8(uint4 v) { return __builtin_bit_cast(st_bf8, v); }

template <int CIN, int COUT, int RT>
__global__ void __launch_bounds__(MF_BLOCK) k_sparse_conv_mfma_b3(const float* __restrict__ x0, int c0, const float* __restrict__ x1,
                                                                  const int32_t* __restrict__ nbr, int K, int64_t n_out, int64_t nstride,
                                                                  const uint4* __restrict__ wq, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ residual,
                                                                  int relu, float* __restrict__ y, const int32_t* __restrict__ row_order) {
    constexpr int CT = COUT / 16, NC = CIN / 32, WK = NC * 3 * 4 * COUT;  // 16-byte vectors of one offset's weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int64_t obase = ((int64_t)blockIdx.x * (MF_BLOCK / 64) + wave) * (16 * RT);
    const int c1 = CIN - c0;
    st_v4f acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[t][ct] = st_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    int64_t orow[RT];
    uint32_t live[RT];
#pragma unroll
    for (int t = 0; t < RT; t++) {
        const int64_t pos = obase + t * 16 + i16;
        const int32_t entry = pos < n_out && row_order ? row_order[pos] : 0;
        orow[t] = pos < n_out ? (row_order ? (int64_t)(entry & CONV_ROW_MASK) : pos) : -1;
        live[t] = conv_live_offsets(entry, K);
    }
    for (int k = 0; k < K; k++) {
        int idx[RT];
        bool any = false;
#pragma unroll
        for (int t = 0; t < RT; t++) {
            idx[t] = orow[t] >= 0 && ((live[t] >> k) & 1u) ? (nbr ? nbr[(int64_t)k * nstride + orow[t]] : (int)orow[t]) : -1;
            any = any || idx[t] >= 0;
        }
        if (__ballot(any) == 0ull) continue;  // no voxel of this wave has a neighbour at offset k (wave-uniform)
        // B fragments straight from the L2-resident planes (staging an offset's planes in LDS once per workgroup, two barriers per
        // offset, measured no faster at 16 clouds per launch set and 1.5x slower for the parity-ordered inverse convs)
        const uint4* wk = wq + (int64_t)k * WK;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int ci = 32 * c + 8 * g;
            StB3 a[RT];
#pragma unroll
            for (int t = 0; t < RT; t++) {
                float4 lo4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), hi4 = lo4;
                if (idx[t] >= 0) {
                    const float* row = ci < c0 ? x0 + (int64_t)idx[t] * c0 + ci : x1 + (int64_t)idx[t] * c1 + (ci - c0);
                    lo4 = *reinterpret_cast<const float4*>(row);
                    hi4 = *reinterpret_cast<const float4*>(row + 4);
                }
                a[t] = b3_split8(lo4, hi4);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const uint4* wc = wk + ((int64_t)(c * 3) * 4 + g) * COUT + ct * 16 + i16;  // plane stride = 4 * COUT
                const st_bf8 bh = b3_bf8(wc[0]), bm = b3_bf8(wc[4 * COUT]), bl = b3_bf8(wc[8 * COUT]);
#pragma unroll
                for (int t = 0; t < RT; t++) {
                    const st_bf8 ah = b3_bf8(a[t].h), am = b3_bf8(a[t].m), al = b3_bf8(a[t].l);
                    st_v4f d = acc[t][ct];
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d, 0, 0, 0);
                    acc[t][ct] = d;
                }
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int ch = ct * 16 + i16;
        const float sc = scale ? scale[ch] : 1.0f, sh = scale ? shift[ch] : 0.0f;
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t pos = obase + t * 16 + g * 4 + r;
                if (pos >= n_out) continue;
                const int64_t o = row_order ? (int64_t)(row_order[pos] & CONV_ROW_MASK) : pos;
                float v = acc[t][ct][r];
                if (scale) v = fmaf(v, sc, sh);
                if (residual) v += residual[o * COUT + ch];
                if (relu) v = v > 0.0f ? v : 0.0f;
                y[o * COUT + ch] = v;
            }
    }
}

// Sixteen input channels (the submanifold and strided convs of level 1): a 32-deep instruction takes TWO kernel offsets at once --
// lanes g = 0, 1 feed channels 8g .. +7 of the row at offset 2j, lanes g = 2, 3 those of the row at offset 2j + 1, and the B operand
// stacks W_{2j} on W_{2j+1} (zeros behind an odd last offset): wq[j][plane][g][co][e] = piece of W[2j + (g >> 1)][8 (g & 1) + e][co].
template <int COUT, int RT>
__global__ void __launch_bounds__(MF_BLOCK) k_sparse_conv_mfma_b3_c16(const float* __restrict__ x, const int32_t* __restrict__ nbr, int K,
                                                                      int64_t n_out, int64_t nstride, const uint4* __restrict__ wq,
                                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                                      const float* __restrict__ residual, int relu, float* __restrict__ y,
                                                                      const int32_t* __restrict__ row_order) {
    constexpr int CT = COUT / 16, WK = 3 * 4 * COUT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int64_t obase = ((int64_t)blockIdx.x * (MF_BLOCK / 64) + wave) * (16 * RT);
    st_v4f acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[t][ct] = st_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    int64_t orow[RT];
    uint32_t live[RT];
#pragma unroll
    for (int t = 0; t < RT; t++) {
        const int64_t pos = obase + t * 16 + i16;
        const int32_t entry = pos < n_out && row_order ? row_order[pos] : 0;
        orow[t] = pos < n_out ? (row_order ? (int64_t)(entry & CONV_ROW_MASK) : pos) : -1;
        live[t] = conv_live_offsets(entry, K);
    }
    const int ch = 8 * (g & 1);
    for (int j = 0; 2 * j < K; j++) {
        const int k = 2 * j + (g >> 1);  // this lane's offset of the pair
        int idx[RT];
        bool any = false;
#pragma unroll
        for (int t = 0; t < RT; t++) {
            idx[t] = k < K && orow[t] >= 0 && ((live[t] >> k) & 1u) ? (nbr ? nbr[(int64_t)k * nstride + orow[t]] : (int)orow[t]) : -1;
            any = any || idx[t] >= 0;
        }
        if (__ballot(any) == 0ull) continue;  // neither offset of the pair has a neighbour in this wave (wave-uniform)
        StB3 a[RT];
#pragma unroll
        for (int t = 0; t < RT; t++) {
            float4 lo4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), hi4 = lo4;
            if (idx[t] >= 0) {
                const float* row = x + (int64_t)idx[t] * 16 + ch;
                lo4 = *reinterpret_cast<const float4*>(row);
                hi4 = *reinterpret_cast<const float4*>(row + 4);
            }
            a[t] = b3_split8(lo4, hi4);
        }
        const uint4* wk = wq + (int64_t)j * WK;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const uint4* wc = wk + (int64_t)g * COUT + ct * 16 + i16;  // plane stride = 4 * COUT
            const st_bf8 bh = b3_bf8(wc[0]), bm = b3_bf8(wc[4 * COUT]), bl = b3_bf8(wc[8 * COUT]);
#pragma unroll
            for (int t = 0; t < RT; t++) {
                const st_bf8 ah = b3_bf8(a[t].h), am = b3_bf8(a[t].m), al = b3_bf8(a[t].l);
                st_v4f d = acc[t][ct];
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d, 0, 0, 0);
                acc[t][ct] = d;
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int co = ct * 16 + i16;
        const float sc = scale ? scale[co] : 1.0f, sh = scale ? shift[co] : 0.0f;
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t pos = obase + t * 16 + g * 4 + r;
                if (pos >= n_out) continue;
                const int64_t o = row_order ? (int64_t)(row_order[pos] & CONV_ROW_MASK) : pos;
                float v = acc[t][ct][r];
                if (scale) v = fmaf(v, sc, sh);
                if (residual) v += residual[o * COUT + co];
                if (relu) v = v > 0.0f ? v : 0.0f;
                y[o * COUT + co] = v;
            }
    }
}

// Same contract as st_sparse_conv_mfma_fwd with the weights as three bf16 planes (see above; smart_tree_amd/model/sparse_ops.py
// b3_weight).  Needs Cin % 32 == 0, Cout % 16 == 0 and a concat split that is a multiple of 8.  variant: 0 = by size, 1 / 2 = row
// tiles per wavefront.
extern "C" int st_sparse_conv_b3_fwd(const float* x0, int c0, const float* x1, int cin, const int32_t* nbr, int K, int64_t n_out,
                                     const void* wq, int cout, const float* scale, const float* shift, const float* residual,
                                     int relu, float* y, const int32_t* row_order, void* stream_, int64_t nbr_stride, int variant) {
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t nstride = nbr_stride > 0 ? nbr_stride : n_out;
    ST_REQUIRE(K >= 1 && (nbr != nullptr || K == 1), "conv: a NULL neighbour table means pointwise (K = 1)");
    ST_REQUIRE(c0 > 0 && c0 <= cin && (c0 == cin || x1 != nullptr), "conv: bad concat split");
    ST_REQUIRE((scale == nullptr) == (shift == nullptr), "conv: scale and shift go together");
    ST_REQUIRE((cin % 32 == 0 || (cin == 16 && c0 == cin)) && cout % 16 == 0 && c0 % 8 == 0,
               "conv(b3): Cin % 32 (or Cin = 16 without concat), Cout % 16 and a concat split % 8 are required");
    if (n_out <= 0) return ST_OK;
    if (cin == 16) {  // two kernel offsets per instruction; wq = b3_weight's pair layout [ceil(K/2)][3][4][Cout][8]
        const int rt16 = variant == 1 || variant == 2 ? variant : (n_out >= (row_order == nullptr ? 56000 : 300000) ? 2 : 1);
#define B3_LAUNCH16(CO, RT_)                                                                                                            \
    hipLaunchKernelGGL((k_sparse_conv_mfma_b3_c16<CO, RT_>), dim3((unsigned)st_div_up(n_out, (MF_BLOCK / 64) * 16 * RT_)), dim3(MF_BLOCK), 0, \
                       stream, x0, nbr, K, n_out, nstride, (const uint4*)wq, scale, shift, residual, relu, y, row_order)
        if (cout == 16) { if (rt16 == 2) B3_LAUNCH16(16, 2); else B3_LAUNCH16(16, 1); }
        else if (cout == 32) { if (rt16 == 2) B3_LAUNCH16(32, 2); else B3_LAUNCH16(32, 1); }
        else { st_set_error("conv(b3): no kernel instance for cin=16 cout=%d", cout); return ST_ERR_INVALID; }
#undef B3_LAUNCH16
        ST_CHECK_LAUNCH();
        return ST_OK;
    }
    // measured on MI355X (tools/bench_conv.py at 1 / 4 / 16 clouds per launch set, profiles/r03_conv_layers_b3.txt): two row tiles per
    // wavefront (each B fragment feeds two tiles) win from ~56k rows on, for the parity-ordered inverse convs from ~300k; below, one
    // tile per wavefront gives the chip twice the wavefronts.  Both compute a row with the same instruction sequence: same bits.
    const int rt = variant == 1 || variant == 2 ? variant : (n_out >= (row_order == nullptr ? 56000 : 300000) ? 2 : 1);
#define B3_LAUNCH(CI, CO, RT_)                                                                                                          \
    hipLaunchKernelGGL((k_sparse_conv_mfma_b3<CI, CO, RT_>), dim3((unsigned)st_div_up(n_out, (MF_BLOCK / 64) * 16 * RT_)), dim3(MF_BLOCK), 0, \
                       stream, x0, c0, x1, nbr, K, n_out, nstride, (const uint4*)wq, scale, shift, residual, relu, y, row_order)
#define B3_CASE(CI, CO)                                                  \
    if (cin == CI && cout == CO) {                                       \
        if (rt == 2) B3_LAUNCH(CI, CO, 2); else B3_LAUNCH(CI, CO, 1);    \
        ST_CHECK_LAUNCH();                                               \
        return ST_OK;                                                    \
    }
    B3_CASE(32, 16)
    B3_CASE(32, 32)
    B3_CASE(32, 64)
    B3_CASE(64, 32)
    B3_CASE(64, 64)
#undef B3_CASE
#undef B3_LAUNCH
    st_set_error("conv(b3): no kernel instance for cin=%d cout=%d", cin, cout);
    return ST_ERR_INVALID;
}

// fp16 rule-GEMM (config 5): features and weights in half precision, v_mfma_f32_16x16x16_f16 with float32
// accumulation.  The operand layout is the f32 kernel's with four channels per lane and ONE instruction per
// 16-channel chunk: lane (i = l & 15, kg = l >> 4) feeds channels 16c + 4kg .. +3 of row i (A, an 8-byte load) and of
// output column i (B: the same host permutation wp[k][c][kg][co][s], stored as half).  Half the gather bytes, a
// quarter of the matrix instructions.  BatchNorm affine / residual / ReLU in float32, one rounding on the store.
template <int CIN, int COUT>
__global__ void __launch_bounds__(MF_BLOCK) k_sparse_conv_mfma_f16(const st_h* __restrict__ x0, int c0, const st_h* __restrict__ x1,
                                                                   const int32_t* __restrict__ nbr, int K, int64_t n_out, int64_t nstride,
                                                                   const st_h* __restrict__ wp, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const st_h* __restrict__ residual,
                                                                   int relu, st_h* __restrict__ y, const int32_t* __restrict__ row_order) {
    constexpr int CT = COUT / 16, NC = CIN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kg = lane >> 4;
    const int64_t obase = ((int64_t)blockIdx.x * (MF_BLOCK / 64) + wave) * 16;
    const int c1 = CIN - c0;
    st_v4f acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) acc[ct] = st_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    const int32_t entry = obase + i16 < n_out && row_order ? row_order[obase + i16] : 0;
    const int64_t orow = obase + i16 < n_out ? (row_order ? (int64_t)(entry & CONV_ROW_MASK) : obase + i16) : -1;
    const uint32_t live = conv_live_offsets(entry, K);
    for (int k = 0; k < K; k++) {
        const st_v4h* wsrc = reinterpret_cast<const st_v4h*>(wp + (int64_t)k * CIN * COUT);
        const int idx = orow >= 0 && ((live >> k) & 1u) ? (nbr ? nbr[(int64_t)k * nstride + orow] : (int)orow) : -1;
        if (__ballot(idx >= 0) == 0ull) continue;  // no voxel of this wave has a neighbour at offset k (wave-uniform)
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int ci = 16 * c + 4 * kg;
            st_v4h av = st_v4h{(st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f};
            if (idx >= 0) {
                const st_h* row = ci < c0 ? x0 + (int64_t)idx * c0 + ci : x1 + (int64_t)idx * c1 + (ci - c0);
                av = *reinterpret_cast<const st_v4h*>(row);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const st_v4h bv = wsrc[(c * 4 + kg) * COUT + ct * 16 + i16];
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(av, bv, acc[ct], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int ch = ct * 16 + i16;
        const float sc = scale ? scale[ch] : 1.0f, sh = scale ? shift[ch] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t pos = obase + kg * 4 + r;
            if (pos >= n_out) continue;
            const int64_t o = row_order ? (int64_t)(row_order[pos] & CONV_ROW_MASK) : pos;
            float v = acc[ct][r];
            if (scale) v = fmaf(v, sc, sh);
            if (residual) v += (float)residual[o * COUT + ch];
            if (relu) v = v > 0.0f ? v : 0.0f;
            y[o * COUT + ch] = (st_h)v;
        }
    }
}

// The same on gfx950's 32-deep instruction (v_mfma_f32_16x16x32_f16, twice the rate per instruction of the 16-deep one) for
// Cin % 32 == 0, with one or two row tiles per wavefront: lane (i, g) feeds channels 32c + 8g .. +7 of row i (ONE 16-byte load) and
// of output column i; weights in the order wp[k][c][g][co][e] = W[k][32c + 8g + e][co] (half; sparse_ops.mfma_weight32).
typedef _Float16 st_v8h __attribute__((ext_vector_type(8)));
template <int CIN, int COUT, int RT>
__global__ void __launch_bounds__(MF_BLOCK) k_sparse_conv_mfma_f16x(const st_h* __restrict__ x0, int c0, const st_h* __restrict__ x1,
                                                                    const int32_t* __restrict__ nbr, int K, int64_t n_out, int64_t nstride,
                                                                    const st_v8h* __restrict__ wp, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, const st_h* __restrict__ residual,
                                                                    int relu, st_h* __restrict__ y, const int32_t* __restrict__ row_order) {
    constexpr int CT = COUT / 16, NC = CIN / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int64_t obase = ((int64_t)blockIdx.x * (MF_BLOCK / 64) + wave) * (16 * RT);
    const int c1 = CIN - c0;
    st_v4f acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[t][ct] = st_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    int64_t orow[RT];
    uint32_t live[RT];
#pragma unroll
    for (int t = 0; t < RT; t++) {
        const int64_t pos = obase + t * 16 + i16;
        const int32_t entry = pos < n_out && row_order ? row_order[pos] : 0;
        orow[t] = pos < n_out ? (row_order ? (int64_t)(entry & CONV_ROW_MASK) : pos) : -1;
        live[t] = conv_live_offsets(entry, K);
    }
    for (int k = 0; k < K; k++) {
        int idx[RT];
        bool any = false;
#pragma unroll
        for (int t = 0; t < RT; t++) {
            idx[t] = orow[t] >= 0 && ((live[t] >> k) & 1u) ? (nbr ? nbr[(int64_t)k * nstride + orow[t]] : (int)orow[t]) : -1;
            any = any || idx[t] >= 0;
        }
        if (__ballot(any) == 0ull) continue;  // no voxel of this wave has a neighbour at offset k (wave-uniform)
        const st_v8h* wk = wp + (int64_t)k * NC * 4 * COUT;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int ci = 32 * c + 8 * g;
            st_v8h a[RT];
#pragma unroll
            for (int t = 0; t < RT; t++) {
                a[t] = st_v8h{(st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f};
                if (idx[t] >= 0) {
                    const st_h* row = ci < c0 ? x0 + (int64_t)idx[t] * c0 + ci : x1 + (int64_t)idx[t] * c1 + (ci - c0);
                    a[t] = *reinterpret_cast<const st_v8h*>(row);
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const st_v8h b = wk[((int64_t)c * 4 + g) * COUT + ct * 16 + i16];
#pragma unroll
                for (int t = 0; t < RT; t++) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b, acc[t][ct], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int ch = ct * 16 + i16;
        const float sc = scale ? scale[ch] : 1.0f, sh = scale ? shift[ch] : 0.0f;
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t pos = obase + t * 16 + g * 4 + r;
                if (pos >= n_out) continue;
                const int64_t o = row_order ? (int64_t)(row_order[pos] & CONV_ROW_MASK) : pos;
                float v = acc[t][ct][r];
                if (scale) v = fmaf(v, sc, sh);
                if (residual) v += (float)residual[o * COUT + ch];
                if (relu) v = v > 0.0f ? v : 0.0f;
                y[o * COUT + ch] = (st_h)v;
            }
    }
}

// ... and for 16 input channels two kernel offsets per instruction (lanes g = 0, 1: the row at offset 2j, g = 2, 3: at 2j + 1; B stacks
// W_2j on W_2j+1, zeros behind an odd last offset: mfma_weight16_half lays the pairs out as 32-channel chunks).
template <int COUT, int RT>
__global__ void __launch_bounds__(MF_BLOCK) k_sparse_conv_mfma_f16x_c16(const st_h* __restrict__ x, const int32_t* __restrict__ nbr, int K,
                                                                        int64_t n_out, int64_t nstride, const st_v8h* __restrict__ wp,
                                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                                        const st_h* __restrict__ residual, int relu, st_h* __restrict__ y,
                                                                        const int32_t* __restrict__ row_order) {
    constexpr int CT = COUT / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int64_t obase = ((int64_t)blockIdx.x * (MF_BLOCK / 64) + wave) * (16 * RT);
    st_v4f acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[t][ct] = st_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    int64_t orow[RT];
    uint32_t live[RT];
#pragma unroll
    for (int t = 0; t < RT; t++) {
        const int64_t pos = obase + t * 16 + i16;
        const int32_t entry = pos < n_out && row_order ? row_order[pos] : 0;
        orow[t] = pos < n_out ? (row_order ? (int64_t)(entry & CONV_ROW_MASK) : pos) : -1;
        live[t] = conv_live_offsets(entry, K);
    }
    const int ch = 8 * (g & 1);
    for (int j = 0; 2 * j < K; j++) {
        const int k = 2 * j + (g >> 1);  // this lane's offset of the pair
        bool any = false;
        st_v8h a[RT];
#pragma unroll
        for (int t = 0; t < RT; t++) {
            const int idx = k < K && orow[t] >= 0 && ((live[t] >> k) & 1u) ? (nbr ? nbr[(int64_t)k * nstride + orow[t]] : (int)orow[t]) : -1;
            any = any || idx >= 0;
            a[t] = st_v8h{(st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f, (st_h)0.0f};
            if (idx >= 0) a[t] = *reinterpret_cast<const st_v8h*>(x + (int64_t)idx * 16 + ch);
        }
        if (__ballot(any) == 0ull) continue;  // neither offset of the pair has a neighbour in this wave (wave-uniform)
        const st_v8h* wk = wp + (int64_t)j * 4 * COUT;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const st_v8h b = wk[(int64_t)g * COUT + ct * 16 + i16];
#pragma unroll
            for (int t = 0; t < RT; t++) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b, acc[t][ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int co = ct * 16 + i16;
        const float sc = scale ? scale[co] : 1.0f, sh = scale ? shift[co] : 0.0f;
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t pos = obase + t * 16 + g * 4 + r;
                if (pos >= n_out) continue;
                const int64_t o = row_order ? (int64_t)(row_order[pos] & CONV_ROW_MASK) : pos;
                float v = acc[t][ct][r];
                if (scale) v = fmaf(v, sc, sh);
                if (residual) v += (float)residual[o * COUT + co];
                if (relu) v = v > 0.0f ? v : 0.0f;
                y[o * COUT + co] = (st_h)v;
            }
    }
}

// Half-precision storage variants of the two calls above (config 5).  in_half / out_half say which side is fp16:
//   both      -> the f16 matrix-core kernel; weights = the MFMA order as half, residual half, channels % 16 == 0
//   exactly one -> the float32 kernel with a converting load or store (the 8 -> 16 "down" and 16 -> 8 "up" convs
//                between the float32 level 0 and the half-precision levels below); weights [K][cin][cout] float32,
//                no residual
extern "C" int st_sparse_conv_f16_fwd(const void* x0, int c0, const void* x1, int cin, const int32_t* nbr, int K, int64_t n_out,
                                      const void* w, int cout, const float* scale, const float* shift, const void* residual,
                                      int relu, void* y, int in_half, int out_half, const int32_t* row_order, void* stream_,
                                      int64_t nbr_stride) {
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t nstride = nbr_stride > 0 ? nbr_stride : n_out;
    ST_REQUIRE(K >= 1 && (nbr != nullptr || K == 1), "conv: a NULL neighbour table means pointwise (K = 1)");
    ST_REQUIRE(c0 > 0 && c0 <= cin && (c0 == cin || x1 != nullptr), "conv: bad concat split");
    ST_REQUIRE((scale == nullptr) == (shift == nullptr), "conv: scale and shift go together");
    ST_REQUIRE(in_half || out_half, "conv(f16): neither side is half precision -- use st_sparse_conv_fwd");
    if (n_out <= 0) return ST_OK;
    if (in_half && out_half) {
        ST_REQUIRE(cin % 16 == 0 && cout % 16 == 0 && c0 % 16 == 0, "conv(f16): channels and concat split must be multiples of 16");
        if (cin == 16 && c0 == cin && (cout == 16 || cout == 32)) {  // two kernel offsets per 32-deep instruction (weights: mfma_weight16_half)
            const int rt16 = n_out >= (row_order == nullptr ? 56000 : 300000) ? 2 : 1;
#define F16C_LAUNCH(CO, RT_)                                                                                                                  \
    hipLaunchKernelGGL((k_sparse_conv_mfma_f16x_c16<CO, RT_>), dim3((unsigned)st_div_up(n_out, (MF_BLOCK / 64) * 16 * RT_)), dim3(MF_BLOCK), 0, \
                       stream, (const st_h*)x0, nbr, K, n_out, nstride, (const st_v8h*)w, scale, shift, (const st_h*)residual, relu, (st_h*)y, row_order)
            if (cout == 16) { if (rt16 == 2) F16C_LAUNCH(16, 2); else F16C_LAUNCH(16, 1); }
            else { if (rt16 == 2) F16C_LAUNCH(32, 2); else F16C_LAUNCH(32, 1); }
#undef F16C_LAUNCH
            ST_CHECK_LAUNCH();
            return ST_OK;
        }
        if (cin % 32 == 0) {  // 32-deep instruction; weights in the 32-channel operand order (mfma_weight32)
            const int rt = n_out >= (row_order == nullptr ? 56000 : 300000) ? 2 : 1;
#define F16X_LAUNCH(CI, CO, RT_)                                                                                                          \
    hipLaunchKernelGGL((k_sparse_conv_mfma_f16x<CI, CO, RT_>), dim3((unsigned)st_div_up(n_out, (MF_BLOCK / 64) * 16 * RT_)), dim3(MF_BLOCK), 0, \
                       stream, (const st_h*)x0, c0, (const st_h*)x1, nbr, K, n_out, nstride, (const st_v8h*)w, scale, shift,               \
                       (const st_h*)residual, relu, (st_h*)y, row_order)
#define F16X_CASE(CI, CO)                                                    \
    if (cin == CI && cout == CO) {                                           \
        if (rt == 2) F16X_LAUNCH(CI, CO, 2); else F16X_LAUNCH(CI, CO, 1);    \
        ST_CHECK_LAUNCH();                                                   \
        return ST_OK;                                                        \
    }
            F16X_CASE(32, 16)
            F16X_CASE(32, 32)
            F16X_CASE(32, 64)
            F16X_CASE(64, 32)
            F16X_CASE(64, 64)
#undef F16X_CASE
#undef F16X_LAUNCH
            st_set_error("conv(f16): no kernel instance for cin=%d cout=%d", cin, cout);
            return ST_ERR_INVALID;
        }
        const int64_t blocks = st_div_up(n_out, (MF_BLOCK / 64) * 16);
#define F16_CASE(CI, CO)                                                                                                        \
    if (cin == CI && cout == CO) {                                                                                              \
        hipLaunchKernelGGL((k_sparse_conv_mfma_f16<CI, CO>), dim3((unsigned)blocks), dim3(MF_BLOCK), 0, stream, (const st_h*)x0, c0, \
                           (const st_h*)x1, nbr, K, n_out, nstride, (const st_h*)w, scale, shift, (const st_h*)residual, relu, (st_h*)y, row_order); \
        ST_CHECK_LAUNCH();                                                                                                      \
        return ST_OK;                                                                                                           \
    }
        F16_CASE(16, 16)
        F16_CASE(16, 32)
        F16_CASE(32, 16)
        F16_CASE(32, 32)
        F16_CASE(32, 64)
        F16_CASE(64, 32)
        F16_CASE(64, 64)
#undef F16_CASE
        st_set_error("conv(f16): no kernel instance for cin=%d cout=%d", cin, cout);
        return ST_ERR_INVALID;
    }
    ST_REQUIRE(residual == nullptr && c0 == cin, "conv(f16): the converting kernels take no residual and no concat");
    ST_REQUIRE(cin % 4 == 0 && cout % 4 == 0, "conv(f16): channels must be multiples of 4");
#define CAST_CASE(CI, CO, COT_)                                                                                                 \
    if (cin == CI && cout == CO) {                                                                                              \
        if (in_half) return conv_launch<CI, COT_, st_h, float>((const st_h*)x0, c0, (const st_h*)x1, nbr, K, n_out, nstride, (const float*)w, \
                                                               cout, scale, shift, nullptr, relu, (float*)y, stream, row_order); \
        return conv_launch<CI, COT_, float, st_h>((const float*)x0, c0, (const float*)x1, nbr, K, n_out, nstride, (const float*)w, cout,   \
                                                  scale, shift, nullptr, relu, (st_h*)y, stream, row_order);                     \
    }
    CAST_CASE(8, 16, 16)
    CAST_CASE(16, 8, 8)
    CAST_CASE(16, 32, 16)
    CAST_CASE(32, 16, 16)
#undef CAST_CASE
    st_set_error("conv(f16): no converting kernel instance for cin=%d cout=%d", cin, cout);
    return ST_ERR_INVALID;
}

// Same contract as st_sparse_conv_fwd, weights in the MFMA order wp[K][Cin/16][4][Cout][4] (see above).
// Needs Cin, Cout multiples of 16 and a concat split that is a multiple of 16 (or no concat).
extern "C" int st_sparse_conv_mfma_fwd(const float* x0, int c0, const float* x1, int cin, const int32_t* nbr, int K,
                                       int64_t n_out, const float* wp, int cout, const float* scale, const float* shift,
                                       const float* residual, int relu, float* y, const int32_t* row_order, void* stream_,
                                       int64_t nbr_stride, int variant) {
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t nstride = nbr_stride > 0 ? nbr_stride : n_out;
    ST_REQUIRE(K >= 1 && (nbr != nullptr || K == 1), "conv: a NULL neighbour table means pointwise (K = 1)");
    ST_REQUIRE(c0 > 0 && c0 <= cin && (c0 == cin || x1 != nullptr), "conv: bad concat split");
    ST_REQUIRE((scale == nullptr) == (shift == nullptr), "conv: scale and shift go together");
    ST_REQUIRE(cin % 16 == 0 && cout % 16 == 0 && c0 % 16 == 0, "conv(mfma): channels and concat split must be multiples of 16");
    if (n_out <= 0) return ST_OK;
#define MFMA_CASE(CI, CO) \
    if (cin == CI && cout == CO) return conv_launch_mfma<CI, CO>(x0, c0, x1, nbr, K, n_out, nstride, wp, scale, shift, residual, relu, y, stream, row_order, variant);
    MFMA_CASE(16, 16)
    MFMA_CASE(16, 32)
    MFMA_CASE(32, 16)
    MFMA_CASE(32, 32)
    MFMA_CASE(32, 64)
    MFMA_CASE(64, 32)
    MFMA_CASE(64, 64)
#undef MFMA_CASE
    st_set_error("conv(mfma): no kernel instance for cin=%d cout=%d", cin, cout);
    return ST_ERR_INVALID;
}

// x = cat(x0[:, :c0], x1[:, :cin-c0]) (x1 may be NULL when c0 == cin); w [K][cin][cout];
// nbr [K][n_out] or NULL (K must be 1: pointwise); scale/shift/residual may be NULL.  nbr_stride: elements between the rows of
// the table (0 = n_out; st_brick_pyramid lays its tables out with the level's capacity as the stride).
// row_order (may be NULL): a permutation of the output rows (bits 0-27; the top four bits may carry the parity tag
// described at conv_live_offsets, 0 = none); wave / tile position p works on output row_order[p].
// Every row is still computed by one lane (or one MFMA tile row) with the same k-ordered arithmetic, so the result
// does not depend on it -- it only decides which rows share a wavefront.  The inverse ("up") convs pass the fine
// rows grouped by coordinate parity (st_build_strided_rulebook): inside a parity class only 1, 2, 4 or 8 of the 27
// offsets can have a partner, and the wave-uniform "nobody has offset k" skip then drops the other 19-26.
extern "C" int st_sparse_conv_fwd(const float* x0, int c0, const float* x1, int cin, const int32_t* nbr, int K,
                                  int64_t n_out, const float* w, int cout, const float* scale, const float* shift,
                                  const float* residual, int relu, float* y, const int32_t* row_order, void* stream_,
                                  int64_t nbr_stride) {
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t nstride = nbr_stride > 0 ? nbr_stride : n_out;
    ST_REQUIRE(K >= 1 && (nbr != nullptr || K == 1), "conv: a NULL neighbour table means pointwise (K = 1)");
    ST_REQUIRE(c0 > 0 && c0 <= cin && (c0 == cin || x1 != nullptr), "conv: bad concat split");
    ST_REQUIRE((scale == nullptr) == (shift == nullptr), "conv: scale and shift go together");
    ST_REQUIRE(cout >= 1, "conv: cout must be positive");
    if (n_out <= 0) return ST_OK;
    // the instantiated kernels read the concatenated row in float4 pieces: their split must fall on a multiple of 4 channels
    // (a concat of a 3-channel input is refused); the generic kernel below takes any split 0 < c0 <= cin
#define CONV_CASE(CI, CO, COT_)                                                                                  \
    if (cin == CI && cout == CO) {                                                                               \
        ST_REQUIRE(CI % 4 != 0 || c0 % 4 == 0, "conv: concat split must be a multiple of 4 channels");           \
        ST_REQUIRE(CI % 4 == 0 || c0 == cin, "conv: concat needs cin % 4 == 0");                                  \
        return conv_launch<CI, COT_>(x0, c0, x1, nbr, K, n_out, nstride, w, cout, scale, shift, residual, relu, y, stream, row_order); \
    }
    CONV_CASE(3, 8, 8)
    CONV_CASE(8, 8, 8)
    CONV_CASE(8, 16, 16)
    CONV_CASE(16, 8, 8)
    CONV_CASE(16, 16, 16)
    CONV_CASE(16, 32, 16)
    CONV_CASE(32, 16, 16)
    CONV_CASE(32, 32, 16)
    CONV_CASE(32, 64, 16)
    CONV_CASE(64, 32, 16)
    CONV_CASE(64, 64, 16)
#undef CONV_CASE
    // any other channel counts (a model config away from the shipped planes, e.g. colour as input channels 4-6): the generic kernel
    hipLaunchKernelGGL(k_sparse_conv_any, dim3((unsigned)st_div_up(n_out, CONV_BLOCK), (unsigned)st_div_up(cout, 4)), dim3(CONV_BLOCK), 0, stream, x0, c0,
                       x1, cin, nbr, K, n_out, nstride, w, cout, scale, shift, residual, relu, y, row_order);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// ----------------------------------------------------------------------------------- heads ---
// Three SparseFC heads (model_blocks.py:246-285 as the checkpoints hold them): per voxel
//   8 -> 8 (+BN+ReLU) -> 4 (+BN+ReLU) -> {1, 3, 2}, no bias; then F.normalize on direction
// (model.py:84, eps 1e-12) and, for ModelInference.forward's tail (model_inference.py:87-88),
// medial_vector = exp(radius) * direction and class = argmax(class_l) (first maximum).
// Packed parameter block per head h (float): W1[8][8] (in-major: [ci][co]), s1[8], t1[8],
// W2[8][4], s2[4], t2[4], W3[4][nout_h]  ->  see HEAD_STRIDE.
#define HEAD_W1 0
#define HEAD_S1 64
#define HEAD_T1 72
#define HEAD_W2 80
#define HEAD_S2 112
#define HEAD_T2 116
#define HEAD_W3 120
#define HEAD_STRIDE 132

template <int NOUT>
__device__ __forceinline__ void head_eval(const float* __restrict__ p, const float* x, float* out) {
    float h1[8], h2[4];
#pragma unroll
    for (int co = 0; co < 8; co++) {
        float a = 0.0f;
#pragma unroll
        for (int ci = 0; ci < 8; ci++) a = fmaf(x[ci], p[HEAD_W1 + ci * 8 + co], a);
        a = fmaf(a, p[HEAD_S1 + co], p[HEAD_T1 + co]);
        h1[co] = a > 0.0f ? a : 0.0f;
    }
#pragma unroll
    for (int co = 0; co < 4; co++) {
        float a = 0.0f;
#pragma unroll
        for (int ci = 0; ci < 8; ci++) a = fmaf(h1[ci], p[HEAD_W2 + ci * 4 + co], a);
        a = fmaf(a, p[HEAD_S2 + co], p[HEAD_T2 + co]);
        h2[co] = a > 0.0f ? a : 0.0f;
    }
#pragma unroll
    for (int co = 0; co < NOUT; co++) {
        float a = 0.0f;
#pragma unroll
        for (int ci = 0; ci < 4; ci++) a = fmaf(h2[ci], p[HEAD_W3 + ci * NOUT + co], a);
        out[co] = a;
    }
}

__global__ void __launch_bounds__(CONV_BLOCK) k_heads(const float* __restrict__ x, int64_t n, const float* __restrict__ params,
                                                      float* __restrict__ radius, float* __restrict__ direction,
                                                      float* __restrict__ class_l, float* __restrict__ medial_vector,
                                                      int64_t* __restrict__ class_idx) {
    int64_t i = (int64_t)blockIdx.x * CONV_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 a = *reinterpret_cast<const float4*>(x + i * 8);
    const float4 b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
    const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float r[1], d[3], c[2];
    head_eval<1>(params, xv, r);
    head_eval<3>(params + HEAD_STRIDE, xv, d);
    head_eval<2>(params + 2 * HEAD_STRIDE, xv, c);
    float norm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float den = norm > 1e-12f ? norm : 1e-12f;
    d[0] /= den; d[1] /= den; d[2] /= den;
    radius[i] = r[0];
    direction[3 * i] = d[0]; direction[3 * i + 1] = d[1]; direction[3 * i + 2] = d[2];
    class_l[2 * i] = c[0]; class_l[2 * i + 1] = c[1];
    if (medial_vector) {
        float e = expf(r[0]);
        medial_vector[3 * i] = e * d[0]; medial_vector[3 * i + 1] = e * d[1]; medial_vector[3 * i + 2] = e * d[2];
    }
    if (class_idx) class_idx[i] = c[1] > c[0] ? 1 : 0;
}

extern "C" int st_head_param_floats(void) { return 3 * HEAD_STRIDE; }

extern "C" int st_pointwise_mlp_heads(const float* x, int64_t n, const float* params, float* radius, float* direction,
                                      float* class_l, float* medial_vector, int64_t* class_idx, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return ST_OK;
    hipLaunchKernelGGL(k_heads, dim3((unsigned)st_div_up(n, CONV_BLOCK)), dim3(CONV_BLOCK), 0, stream, x, n, params, radius,
                       direction, class_l, medial_vector, class_idx);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
