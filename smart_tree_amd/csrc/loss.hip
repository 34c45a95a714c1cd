// st_loss_forward: the three training / evaluation losses of one batch in ONE pass over the voxels.
//
// Replaces (forward only -- SURVEY.md section 8f.4; there is no backward pass in this package)
//   compute_loss             smart_tree/model/loss.py:7-51   (loss mask, "vector" class mask, log of the target radius)
//   L1Loss                   smart_tree/model/loss.py:54-56  (mean |radius - log target radius| over the vector rows)
//   cosine_similarity_loss   smart_tree/model/loss.py:59-61  (mean 1 - cos over the vector rows; torch's CosineSimilarity:
//                                                             each vector divided by max(|v|, 1e-8), then the dot product)
//   focal_loss               smart_tree/model/loss.py:81-97  (gamma = 2: mean -(1 - p_t)^2 log p_t over the masked rows)
//   dice_loss                smart_tree/model/loss.py:64-78  (1 - (2 sum(softmax * onehot) + 1) / (sum softmax + sum onehot + 1))
// which the reference evaluates as a dozen torch kernels with three boolean-mask compactions in between.  Here a row is read
// once (9 + C floats and a mask byte), every term is computed in float32 like torch does, and the sums are carried in float64:
// lane -> wavefront (shuffles) -> workgroup (LDS) -> one slot per workgroup in the workspace -> a single-workgroup pass adds the
// slots in index order.  No float atomics: the result is a function of (inputs, n) alone, run to run.
// HBM-bound: (10 + C) * 4 + 1 bytes per voxel, nothing written.
#include "st_common.h"

#define LS_BLOCK 256
#define LS_MAX_BLOCKS 1024
#define LS_TERMS 8  // sum |dr|, vector rows, sum (1 - cos), sum focal, class rows, dice intersection, sum softmax, bad class ids
#define LS_MAX_CLASSES 16

struct LsArgs {
    const float* radius;     // [n]     predicted log radius
    const float* direction;  // [n, 3]  predicted direction
    const float* class_l;    // [n, C]  class logits
    const float* targets;    // [n, 5]  radius, direction xyz, class id (loss.py:19-21)
    const uint8_t* mask;     // [n] or null (loss.py:23-29)
    int64_t n;
    int n_classes;
    int vector_class;        // -1: every masked row is a vector row (loss.py:32-38)
    int target_radius_log;   // loss.py:40-41
};

__device__ __forceinline__ double ls_wave_sum(double v) {
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ void __launch_bounds__(LS_BLOCK) k_loss_partial(LsArgs A, double* partial) {
    __shared__ double s[LS_BLOCK / 64][LS_TERMS];
    double acc[LS_TERMS];
#pragma unroll
    for (int k = 0; k < LS_TERMS; k++) acc[k] = 0.0;
    const int C = A.n_classes;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * blockDim.x) {
        if (A.mask && !A.mask[i]) continue;
        const float* t = A.targets + 5 * i;
        const float tcf = t[4];
        // .long(): truncation.  A NaN / infinite class id has no defined conversion: it is reported like an id outside the logits
        const bool tc_ok = tcf > -1.0f && tcf < (float)C;
        const long long tc = tc_ok ? (long long)tcf : -1ll;
        // class terms (every masked row)
        const float* z = A.class_l + (int64_t)C * i;
        float zmax = z[0];
        for (int k = 1; k < C; k++) zmax = z[k] > zmax ? z[k] : zmax;
        float se = 0.0f;
        for (int k = 0; k < C; k++) se += expf(z[k] - zmax);
        const float lse = logf(se);
        acc[4] += 1.0;
        acc[6] += 1.0;  // a softmax row sums to one
        if (tc_ok) {
            const float logpt = (z[tc] - zmax) - lse;  // log_softmax
            const float pt = expf(logpt);
            const float om = 1.0f - pt;
            acc[3] += (double)(-1.0f * (om * om) * logpt);
            acc[5] += (double)pt;
        } else {
            acc[7] += 1.0;  // torch's gather / one_hot would raise: reported to the host
        }
        // vector terms
        if (A.vector_class >= 0 && tc != (long long)A.vector_class) continue;
        float tr = t[0];
        if (A.target_radius_log) tr = logf(tr);
        acc[0] += (double)fabsf(A.radius[i] - tr);
        acc[1] += 1.0;
        const float* d = A.direction + 3 * i;
        const float px = d[0], py = d[1], pz = d[2], qx = t[1], qy = t[2], qz = t[3];
        const float np = fmaxf(sqrtf(px * px + py * py + pz * pz), 1e-8f), nq = fmaxf(sqrtf(qx * qx + qy * qy + qz * qz), 1e-8f);
        const float cs = (px / np) * (qx / nq) + (py / np) * (qy / nq) + (pz / np) * (qz / nq);
        acc[2] += (double)(1.0f - cs);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < LS_TERMS; k++) {
        const double w = ls_wave_sum(acc[k]);
        if (lane == 0) s[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < LS_TERMS) {
        double v = 0.0;
        for (int w = 0; w < LS_BLOCK / 64; w++) v += s[w][threadIdx.x];
        partial[(int64_t)blockIdx.x * LS_TERMS + threadIdx.x] = v;
    }
}

// out[0..3] = radius, direction, focal, dice losses; out[4] = vector rows, out[5] = class rows, out[6] = rows with a class id
// outside [0, C); out[7] = 0.  An empty selection gives NaN, like the mean of an empty tensor.
__global__ void k_loss_final(const double* partial, int nblocks, double* out) {
    __shared__ double tot[LS_TERMS];
    if (threadIdx.x < LS_TERMS) {
        double v = 0.0;
        for (int b = 0; b < nblocks; b++) v += partial[(int64_t)b * LS_TERMS + threadIdx.x];
        tot[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nan = (double)__uint_as_float(0x7fc00000u);
        const double nv = tot[1], nc = tot[4];
        out[0] = nv > 0.0 ? tot[0] / nv : nan;
        out[1] = nv > 0.0 ? tot[2] / nv : nan;
        out[2] = nc > 0.0 ? tot[3] / nc : nan;
        out[3] = 1.0 - (2.0 * tot[5] + 1.0) / (tot[6] + nc + 1.0);  // sum one_hot = one per row
        out[4] = nv; out[5] = nc; out[6] = tot[7]; out[7] = 0.0;
    }
}

extern "C" int64_t st_loss_workspace_bytes(void) { return (int64_t)(LS_MAX_BLOCKS * LS_TERMS + 8) * sizeof(double) + 256; }

extern "C" int st_loss_forward(const float* radius, const float* direction, const float* class_l, int n_classes,
                               const float* targets, int target_cols, const uint8_t* mask, int64_t n, int vector_class,
                               int target_radius_log, double* out_host, void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    ST_REQUIRE(n >= 0, "loss: n < 0");
    ST_REQUIRE(target_cols == 5, "loss: targets must be [n, 5] = radius, direction xyz, class (got %d columns)", target_cols);
    ST_REQUIRE(n_classes >= 1 && n_classes <= LS_MAX_CLASSES, "loss: 1 <= classes <= %d (got %d)", LS_MAX_CLASSES, n_classes);
    ST_REQUIRE(out_host != nullptr, "loss: out_host is null");
    ST_REQUIRE(n == 0 || (radius && direction && class_l && targets), "loss: null input");
    StArena a(ws, ws_bytes);
    double* partial = a.take<double>((int64_t)LS_MAX_BLOCKS * LS_TERMS);
    double* out = a.take<double>(8);
    if (!a.ok() || !partial || !out) {
        st_set_error("loss: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)a.used);
        return ST_ERR_WORKSPACE;
    }
    LsArgs A{radius, direction, class_l, targets, mask, n, n_classes, vector_class, target_radius_log};
    int64_t nb = st_div_up(n > 0 ? n : 1, (int64_t)LS_BLOCK * 4);
    if (nb > LS_MAX_BLOCKS) nb = LS_MAX_BLOCKS;
    hipLaunchKernelGGL(k_loss_partial, dim3((unsigned)nb), dim3(LS_BLOCK), 0, stream, A, partial);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, stream, (const double*)partial, (int)nb, out);
    ST_CHECK_LAUNCH();
    (void)hipMemcpyAsync(out_host, out, 8 * sizeof(double), hipMemcpyDeviceToHost, stream);
    st_stream_wait(stream);
    ST_CHECK_LAUNCH();
    ST_REQUIRE(out_host[6] == 0.0, "loss: %lld target class ids outside [0, %d)", (long long)out_host[6], n_classes);
    return ST_OK;
}
