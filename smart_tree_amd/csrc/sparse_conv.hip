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

template <int CIN, int COT>
__global__ void __launch_bounds__(CONV_BLOCK) k_sparse_conv(const float* __restrict__ x0, int c0,
                                                            const float* __restrict__ x1, const int32_t* __restrict__ nbr,
                                                            int K, int64_t n_out, const float* __restrict__ w, int cout,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ residual, int relu,
                                                            float* __restrict__ y) {
    const int co_tiles = cout / COT;
    const int co0 = (int)(blockIdx.x % co_tiles) * COT;
    const int64_t o = (int64_t)(blockIdx.x / co_tiles) * CONV_BLOCK + threadIdx.x;
    const bool active = o < n_out;
    const int c1 = CIN - c0;
    float acc[COT];
#pragma unroll
    for (int c = 0; c < COT; c++) acc[c] = 0.0f;

    for (int k = 0; k < K; k++) {
        int idx = -1;
        if (active) idx = nbr ? nbr[(int64_t)k * n_out + o] : (int)o;
        if (idx < 0) continue;
        const float* __restrict__ wk = w + (int64_t)k * CIN * cout + co0;
        if (CIN % 4 == 0) {
#pragma unroll 2
            for (int ci = 0; ci < CIN; ci += 4) {
                const float* row = ci < c0 ? x0 + (int64_t)idx * c0 + ci : x1 + (int64_t)idx * c1 + (ci - c0);
                const float4 v = *reinterpret_cast<const float4*>(row);
                const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int c = 0; c < COT; c++) acc[c] = fmaf(xs[j], wk[(ci + j) * cout + c], acc[c]);
            }
        } else {
            for (int ci = 0; ci < CIN; ci++) {
                const float xv = x0[(int64_t)idx * CIN + ci];
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
    float* out = y + o * cout + co0;
    if (COT % 4 == 0) {
#pragma unroll
        for (int c = 0; c < COT; c += 4)
            *reinterpret_cast<float4*>(out + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    } else {
#pragma unroll
        for (int c = 0; c < COT; c++) out[c] = acc[c];
    }
}

template <int CIN, int COT>
static int conv_launch(const float* x0, int c0, const float* x1, const int32_t* nbr, int K, int64_t n_out, const float* w,
                       int cout, const float* scale, const float* shift, const float* residual, int relu, float* y,
                       hipStream_t stream) {
    int64_t blocks = st_div_up(n_out, CONV_BLOCK) * (cout / COT);
    hipLaunchKernelGGL((k_sparse_conv<CIN, COT>), dim3((unsigned)blocks), dim3(CONV_BLOCK), 0, stream, x0, c0, x1, nbr, K,
                       n_out, w, cout, scale, shift, residual, relu, y);
    ST_CHECK_LAUNCH();
    return ST_OK;
}

// x = cat(x0[:, :c0], x1[:, :cin-c0]) (x1 may be NULL when c0 == cin); w [K][cin][cout];
// nbr [K][n_out] or NULL (K must be 1: pointwise); scale/shift/residual may be NULL.
extern "C" int st_sparse_conv_fwd(const float* x0, int c0, const float* x1, int cin, const int32_t* nbr, int K,
                                  int64_t n_out, const float* w, int cout, const float* scale, const float* shift,
                                  const float* residual, int relu, float* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ST_REQUIRE(K >= 1 && (nbr != nullptr || K == 1), "conv: a NULL neighbour table means pointwise (K = 1)");
    ST_REQUIRE(c0 > 0 && c0 <= cin && (c0 == cin || x1 != nullptr), "conv: bad concat split");
    ST_REQUIRE((scale == nullptr) == (shift == nullptr), "conv: scale and shift go together");
    ST_REQUIRE(cin % 4 != 0 || c0 % 4 == 0, "conv: concat split must be a multiple of 4 channels");
    ST_REQUIRE(cin % 4 == 0 || c0 == cin, "conv: concat needs cin % 4 == 0");
    if (n_out <= 0) return ST_OK;
#define CONV_CASE(CI, CO, COT_)                                                                                  \
    if (cin == CI && cout == CO)                                                                                 \
        return conv_launch<CI, COT_>(x0, c0, x1, nbr, K, n_out, w, cout, scale, shift, residual, relu, y, stream);
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
    st_set_error("conv: no kernel instance for cin=%d cout=%d", cin, cout);
    return ST_ERR_INVALID;
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
