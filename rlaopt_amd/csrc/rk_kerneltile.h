// Tile code shared by the kernel-matrix kernels (rk_kernelmat.hip, rk_symm.hip): 64 x 64 pair
// tiles, panels staged feature-major in LDS, direct (x - y) distance statistics.
#pragma once

#include "rk_common.h"

namespace rk {

constexpr int TM = 64;   // tile rows  (points of X1)
constexpr int TN = 64;   // tile cols  (points of X2)
// feature chunk staged per pass: template parameter DCV (32 when d <= 32, else 64)
constexpr int LDP = 68;  // padded panel stride (multiple of 4 for 16-B LDS reads)

template <typename T>
__device__ __forceinline__ T rk_exp(T x);
template <>
__device__ __forceinline__ float rk_exp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double rk_exp<double>(double x) { return exp(x); }
template <typename T>
__device__ __forceinline__ T rk_sqrt(T x);
template <>
__device__ __forceinline__ float rk_sqrt<float>(float x) { return sqrtf(x); }
template <>
__device__ __forceinline__ double rk_sqrt<double>(double x) { return sqrt(x); }

// distance statistic s (sum of squares, or sum of abs for Laplace) -> kernel value.
// `w` = 1/lengthscale for a scalar lengthscale, 1 when the panels were pre-scaled (ARD).
template <typename T>
__device__ __forceinline__ T kernel_value(int kind, T s, T w) {
    switch (kind) {
        case RK_RBF: return rk_exp<T>(T(-0.5) * s * w * w);
        case RK_LAPLACE: return rk_exp<T>(-s * w);
        case RK_MATERN12: return rk_exp<T>(-rk_sqrt<T>(s) * w);
        case RK_MATERN32: {
            const T D = T(1.7320508075688772) * rk_sqrt<T>(s) * w;
            return (T(1) + D) * rk_exp<T>(-D);
        }
        default: {  // RK_MATERN52
            const T D0 = rk_sqrt<T>(s) * w;
            const T D = T(2.23606797749979) * D0;
            return (T(1) + D + T(5.0 / 3.0) * D0 * D0) * rk_exp<T>(-D);
        }
    }
}

// Stage a 64-point x DC-feature panel into LDS, feature-major, optionally gathered through
// idx and pre-scaled by 1/ell[c].  Lane -> point, wave -> 8-feature group: the LDS writes of
// a wave are 64 consecutive words (conflict-free).
template <typename T, int DCV>
__device__ __forceinline__ void stage_panel(T (*dst)[LDP], const T* __restrict__ X, int64_t n,
                                            int64_t ldx, const int64_t* __restrict__ idx,
                                            int64_t p0, int64_t c0, int64_t d,
                                            const T* __restrict__ ell_vec) {
    const int pt = threadIdx.x & 63;
    const int cg = threadIdx.x >> 6;  // 0..3
    const int64_t p = p0 + pt;
    const T* row = nullptr;
    if (p < n) row = X + (idx ? idx[p] : p) * ldx;
    // all global loads first, then the LDS stores (a load -> store chain per element serialises
    // the memory latencies: measured 10 % on the fused product)
    T v[DCV / 4];
#pragma unroll
    for (int u = 0; u < DCV / 4; ++u) {
        const int c = cg * (DCV / 4) + u;
        v[u] = T(0);
        if (row != nullptr && c0 + c < d) {
            v[u] = row[c0 + c];
            if (ell_vec) v[u] = v[u] / ell_vec[c0 + c];
        }
    }
#pragma unroll
    for (int u = 0; u < DCV / 4; ++u) dst[cg * (DCV / 4) + u][pt] = v[u];
}

// acc[r][q] += stat(x_r - y_q) over the first `nc` features of the staged chunk (only the
// features that exist are visited: d = 50 costs 50 steps, not a padded 64)
template <typename T, bool L1>
__device__ __forceinline__ void tile_step(const T (*xs)[LDP], const T (*ys)[LDP], int ty, int tx,
                                          T acc[4][4], int c) {
    T xa[4], yb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { xa[e] = xs[c][ty * 4 + e]; yb[e] = ys[c][tx * 4 + e]; }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const T df = xa[r] - yb[q];
            if constexpr (L1) acc[r][q] += fabs(df);
            else acc[r][q] += df * df;
        }
}

template <typename T, bool L1, int DCV>
__device__ __forceinline__ void tile_accumulate(const T (*xs)[LDP], const T (*ys)[LDP], int ty,
                                                int tx, T acc[4][4], int nc) {
    if (nc == DCV) {  // full chunk: compile-time trip count
#pragma unroll 8
        for (int c = 0; c < DCV; ++c) tile_step<T, L1>(xs, ys, ty, tx, acc, c);
    } else {
#pragma unroll 2
        for (int c = 0; c < nc; ++c) tile_step<T, L1>(xs, ys, ty, tx, acc, c);
    }
}

}  // namespace rk
