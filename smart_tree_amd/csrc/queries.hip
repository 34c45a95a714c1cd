// st_points_to_nearest_tube: every point against every tapered tube of a skeleton.
//
// Replaces  pts_to_nearest_tube_gpu   smart_tree/util/queries.py:107-133  (N x M dense torch expression: project on
//           the tube axis with t clipped to [0,1], interpolate the radius, pick the tube minimising |distance - radius|)
//      and  skeleton_to_points        smart_tree/util/queries.py:139-166  (the same in host chunks of 4096 points)
// -- the evaluation utility that labels a cloud with its skeleton.  N x M is 5e9 pairs for a
// 1M-point cloud and a 5000-tube skeleton, so this is the one COMPUTE-bound kernel of the package: a lane owns two
// points (packed float32 adds / multiplies), the tubes stream through LDS in tiles (every lane reads the same tube: LDS broadcast), ~45 float32 operations
// per pair including one IEEE division and one square root.
// Float32 operation order is fixed and mirrored by oracle/queries_oracle.py (the same as st_post_process's repair):
//   dot(a,b) = (ax*bx + ay*by) + az*bz, no contraction; t = dot(ap,ab) / dot(ab,ab) (NaN for a zero-length tube:
//   the score is NaN and NaN counts as minimal, as torch.argmin does); first minimum wins.
#include "st_common.h"

#define QT_BLOCK 256
#define QT_TILE 512
typedef float qt_f2 __attribute__((ext_vector_type(2)));  // two points per lane: v_pk_{add,mul}_f32 do both at once

__device__ __forceinline__ float qt_dot(float ax, float ay, float az, float bx, float by, float bz) {
    float s = ax * bx;
    float t = ay * by;
    s = s + t;
    t = az * bz;
    return s + t;
}
__device__ __forceinline__ qt_f2 qt_dot2(qt_f2 ax, qt_f2 ay, qt_f2 az, qt_f2 bx, qt_f2 by, qt_f2 bz) {
    qt_f2 s = ax * bx;
    qt_f2 t = ay * by;
    s = s + t;
    t = az * bz;
    return s + t;
}
__device__ __forceinline__ float qt_clip01(float t) { return t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t); }  // NaN stays NaN

__global__ void __launch_bounds__(QT_BLOCK) k_nearest_tube(const float* __restrict__ pts, int64_t n, const float* __restrict__ a,
                                                           const float* __restrict__ b, const float* __restrict__ r1,
                                                           const float* __restrict__ r2, int64_t m, float* __restrict__ vec,
                                                           int64_t* __restrict__ idx, float* __restrict__ rad) {
    // per tube: (a.xyz, r1), (ab.xyz, r2), ab.ab -- three LDS reads per tube, the same address in every lane (broadcast)
    __shared__ float4 ta[QT_TILE], tb[QT_TILE];
    __shared__ float tab2[QT_TILE];
    const int64_t i0 = ((int64_t)blockIdx.x * QT_BLOCK + threadIdx.x) * 2, i1 = i0 + 1;
    const bool live0 = i0 < n, live1 = i1 < n;
    const qt_f2 px = {live0 ? pts[3 * i0] : 0.0f, live1 ? pts[3 * i1] : 0.0f};
    const qt_f2 py = {live0 ? pts[3 * i0 + 1] : 0.0f, live1 ? pts[3 * i1 + 1] : 0.0f};
    const qt_f2 pz = {live0 ? pts[3 * i0 + 2] : 0.0f, live1 ? pts[3 * i1 + 2] : 0.0f};
    unsigned key0 = 0xffffffffu, key1 = 0xffffffffu;  // 0 = NaN score (minimal), else score bits + 1
    int best0 = 0, best1 = 0;
    float v0x = 0.0f, v0y = 0.0f, v0z = 0.0f, r0 = 0.0f, v1x = 0.0f, v1y = 0.0f, v1z = 0.0f, rr1 = 0.0f;
    for (int64_t base = 0; base < m; base += QT_TILE) {
        const int cnt = (int)(m - base < QT_TILE ? m - base : QT_TILE);
        __syncthreads();  // the previous tile is no longer read
        for (int j = threadIdx.x; j < cnt; j += QT_BLOCK) {
            const int64_t g = base + j;
            const float ax = a[3 * g], ay = a[3 * g + 1], az = a[3 * g + 2];
            const float dx = b[3 * g] - ax, dy = b[3 * g + 1] - ay, dz = b[3 * g + 2] - az;
            ta[j] = make_float4(ax, ay, az, r1[g]);
            tb[j] = make_float4(dx, dy, dz, r2[g]);
            tab2[j] = qt_dot(dx, dy, dz, dx, dy, dz);
        }
        __syncthreads();
        if (!live0) continue;
        for (int j = 0; j < cnt; j++) {
            const float4 A = ta[j], B = tb[j];
            const float ab2 = tab2[j];
            const qt_f2 num = qt_dot2(px - A.x, py - A.y, pz - A.z, qt_f2{B.x, B.x}, qt_f2{B.y, B.y}, qt_f2{B.z, B.z});
            const qt_f2 t = {qt_clip01(num.x / ab2), qt_clip01(num.y / ab2)};
            qt_f2 qx = t * B.x, qy = t * B.y, qz = t * B.z;
            qx = A.x + qx; qy = A.y + qy; qz = A.z + qz;
            qt_f2 r = (1.0f - t) * A.w;
            const qt_f2 r_hi = t * B.w;
            r = r + r_hi;
            const qt_f2 vx = qx - px, vy = qy - py, vz = qz - pz;
            const qt_f2 d2 = qt_dot2(vx, vy, vz, vx, vy, vz);
            const float s0 = fabsf(sqrtf(d2.x) - r.x), s1 = fabsf(sqrtf(d2.y) - r.y);
            const unsigned k0 = s0 != s0 ? 0u : __float_as_uint(s0) + 1u;  // score >= 0: bits are ordered
            const unsigned k1 = s1 != s1 ? 0u : __float_as_uint(s1) + 1u;
            if (k0 < key0) { key0 = k0; best0 = (int)(base + j); v0x = vx.x; v0y = vy.x; v0z = vz.x; r0 = r.x; }
            if (k1 < key1) { key1 = k1; best1 = (int)(base + j); v1x = vx.y; v1y = vy.y; v1z = vz.y; rr1 = r.y; }
        }
    }
    if (live0) { vec[3 * i0] = v0x; vec[3 * i0 + 1] = v0y; vec[3 * i0 + 2] = v0z; idx[i0] = best0; rad[i0] = r0; }
    if (live1) { vec[3 * i1] = v1x; vec[3 * i1 + 1] = v1y; vec[3 * i1 + 2] = v1z; idx[i1] = best1; rad[i1] = rr1; }
}

// pts [n,3]; a, b [m,3] tube end points, r1, r2 [m] end radii; vec [n,3] (projection - point), idx [n] int64, rad [n].
extern "C" int st_points_to_nearest_tube(const float* pts, int64_t n, const float* a, const float* b, const float* r1,
                                         const float* r2, int64_t m, float* vec, int64_t* idx, float* rad, void* stream_) {
    ST_REQUIRE(m >= 1 && m < (1ll << 31), "nearest tube: the skeleton needs 1 .. 2^31 tubes");
    if (n <= 0) return ST_OK;
    hipLaunchKernelGGL(k_nearest_tube, dim3((unsigned)st_div_up(n, 2 * QT_BLOCK)), dim3(QT_BLOCK), 0, (hipStream_t)stream_, pts, n, a, b,
                       r1, r2, m, vec, idx, rad);
    ST_CHECK_LAUNCH();
    return ST_OK;
}
