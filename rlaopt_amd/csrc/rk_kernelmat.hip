// Kernel-matrix entries on gfx950: dense materialisation and the fused matrix-free product.
//
// Both kernels evaluate the SAME tile code: a 64 x 64 tile of pairs per 256-thread
// workgroup, 4 x 4 pairs per lane, with the two 64-point panels staged through LDS in
// feature-major order (xs[c][i]) so that a lane reads its 4 points of either panel with one
// ds_read_b128 (broadcast along the other tile dimension, conflict-free).  Distances use
// the direct form sum(((x-y)/l)^2) / sum(|x-y|/l) - never |x|^2+|y|^2-2xy, whose
// cancellation breaks the reference's 1e-8 fp64 tolerance for near-coincident points.
//
//   rk_kernel_materialize : writes the tile (scaled) to HBM as float4 rows
//   rk_kernel_matmat      : multiplies the tile with the matching 64 x k slice of V held in
//                           LDS and keeps 4 x k running sums per lane; the 16 lanes that
//                           share a row combine with a 4-step butterfly at the end.  K is
//                           never written.  Column segments (gridDim.y) give short-and-wide
//                           oracle blocks (5 000 x 500 000) enough workgroups; their
//                           partials are summed in fixed order by the finishing kernel.
#include "rk_common.h"
#include "rk_kerneltile.h"

namespace rk {

template <typename T, bool L1, int DCV>
__global__ void __launch_bounds__(256)
kmat_materialize_kernel(int kind, const T* __restrict__ X1, int64_t n1, int64_t ldx1,
                        const int64_t* __restrict__ idx1, const T* __restrict__ X2, int64_t n2,
                        int64_t ldx2, const int64_t* __restrict__ idx2, int64_t d, T w,
                        const T* __restrict__ ell_vec, T scale, T* __restrict__ K, int64_t ldk,
                        int vec_store) {
    __shared__ T xs[DCV][LDP];
    __shared__ T ys[DCV][LDP];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t i0 = (int64_t)blockIdx.y * TM, j0 = (int64_t)blockIdx.x * TN;
    T acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] = T(0);
    for (int64_t c0 = 0; c0 < d; c0 += DCV) {
        __syncthreads();
        stage_panel<T, DCV>(xs, X1, n1, ldx1, idx1, i0, c0, d, ell_vec);
        stage_panel<T, DCV>(ys, X2, n2, ldx2, idx2, j0, c0, d, ell_vec);
        __syncthreads();
        tile_accumulate<T, L1, DCV>(xs, ys, ty, tx, acc, (int)((d - c0) < DCV ? (d - c0) : DCV));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + ty * 4 + r;
        if (i >= n1) continue;
        T o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = scale * kernel_value<T>(kind, acc[r][q], w);
        const int64_t j = j0 + tx * 4;
        T* dst = K + i * ldk + j;
        if (vec_store && j + 4 <= n2) {
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                *reinterpret_cast<double2*>(dst) = make_double2(o[0], o[1]);
                *reinterpret_cast<double2*>(dst + 2) = make_double2(o[2], o[3]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j + q < n2) dst[q] = o[q];
        }
    }
}

template <typename T, bool L1, int KB, int DCV>
__global__ void __launch_bounds__(256)
kmat_matmat_kernel(int kind, const T* __restrict__ X1, int64_t n1, int64_t ldx1,
                   const int64_t* __restrict__ idx1, const T* __restrict__ X2, int64_t n2,
                   int64_t ldx2, const int64_t* __restrict__ idx2, int64_t d, T w,
                   const T* __restrict__ ell_vec, T scale, const T* __restrict__ V, int k,
                   int64_t ldv, int64_t tiles_per_seg, T* __restrict__ part, T* __restrict__ Y,
                   int64_t ldy, int accumulate) {
    __shared__ T xs[DCV][LDP];
    __shared__ T ys[DCV][LDP];
    __shared__ T vs[TN][KB];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * TM;
    const int64_t ntile = (n2 + TN - 1) / TN;
    const int64_t t_lo = (int64_t)blockIdx.y * tiles_per_seg;
    const int64_t t_hi = (t_lo + tiles_per_seg < ntile) ? t_lo + tiles_per_seg : ntile;

    T y[4][KB];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < KB; ++c) y[r][c] = T(0);

    // the X1 panel is staged once when all features fit one chunk, else per (tile, chunk)
    const bool one_chunk = (d <= DCV);
    if (one_chunk) stage_panel<T, DCV>(xs, X1, n1, ldx1, idx1, i0, 0, d, ell_vec);
    for (int64_t t = t_lo; t < t_hi; ++t) {
        const int64_t j0 = t * TN;
        T acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] = T(0);
        for (int64_t c0 = 0; c0 < d; c0 += DCV) {
            __syncthreads();  // readers of the previous panels / V slice are done
            if (!one_chunk) stage_panel<T, DCV>(xs, X1, n1, ldx1, idx1, i0, c0, d, ell_vec);
            stage_panel<T, DCV>(ys, X2, n2, ldx2, idx2, j0, c0, d, ell_vec);
            if (c0 == 0) {
                // V slice for this column tile, zero padded to KB columns / past n2
                for (int e = threadIdx.x; e < TN * KB; e += 256) {
                    const int jj = e / KB, c = e - jj * KB;
                    const int64_t j = j0 + jj;
                    vs[jj][c] = (j < n2 && c < k) ? V[j * ldv + c] : T(0);
                }
            }
            __syncthreads();
            tile_accumulate<T, L1, DCV>(xs, ys, ty, tx, acc, (int)((d - c0) < DCV ? (d - c0) : DCV));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            T kv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) kv[r] = kernel_value<T>(kind, acc[r][q], w);
#pragma unroll
            for (int c = 0; c < KB; ++c) {
                const T v = vs[tx * 4 + q][c];  // zero rows past n2 kill padded pairs
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r][c] += kv[r] * v;
            }
        }
    }
    // combine the 16 lanes (tx) that share each row: lanes differ in bits 0..3
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < KB; ++c) {
            T s = y[r][c];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 8, 64);
            y[r][c] = s;
        }
    if (tx == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = i0 + ty * 4 + r;
            if (i >= n1) continue;
#pragma unroll
            for (int c = 0; c < KB; ++c) {
                if (c >= k) continue;
                if (part != nullptr) {
                    part[((int64_t)blockIdx.y * n1 + i) * k + c] = y[r][c];
                } else {
                    const T o = scale * y[r][c];
                    if (accumulate) Y[i * ldy + c] += o;
                    else Y[i * ldy + c] = o;
                }
            }
        }
    }
}

// Y[i][c] = (accumulate ? Y : 0) + scale * sum_s part[s][i][c]
template <typename T>
__global__ void __launch_bounds__(256)
kmat_matmat_finish(const T* __restrict__ part, int S, int64_t n1, int k, T scale,
                   T* __restrict__ Y, int64_t ldy, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n1 * k) return;
    const int64_t i = e / k;
    const int c = (int)(e - i * k);
    T s = T(0);
    for (int t = 0; t < S; ++t) s += part[(int64_t)t * n1 * k + e];
    s *= scale;
    if (accumulate) Y[i * ldy + c] += s;
    else Y[i * ldy + c] = s;
}

struct KMPlan {
    int64_t S, tiles_per_seg;
};

static KMPlan plan_kmatmat(int64_t n1, int64_t n2) {
    const int64_t row_blocks = ceil_div(n1, TM);
    const int64_t ntile = ceil_div(n2, TN);
    int64_t S = ceil_div(4096, row_blocks);  // >= ~16 workgroups per CU in flight
    if (S > ntile / 8) S = ntile / 8;        // >= 8 column tiles per workgroup
    if (S < 1) S = 1;
    KMPlan p;
    p.tiles_per_seg = ceil_div(ntile, S);
    p.S = ceil_div(ntile, p.tiles_per_seg);
    return p;
}

template <typename T>
static int materialize_impl(int kind, const T* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                            const T* X2, int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d,
                            double ell, const T* ell_vec, double scale, T* K, int64_t ldk,
                            hipStream_t st) {
    const T w = ell_vec ? T(1) : T(1.0 / ell);
    const int vec = aligned16(K) && (ldk % (16 / sizeof(T)) == 0);
    dim3 grid((unsigned)ceil_div(n2, TN), (unsigned)ceil_div(n1, TM));
#define RK_MAT(L1V, DCVV)                                                                         \
    hipLaunchKernelGGL((kmat_materialize_kernel<T, L1V, DCVV>), grid, dim3(256), 0, st, kind, X1,  \
                       n1, ldx1, idx1, X2, n2, ldx2, idx2, d, w, ell_vec, (T)scale, K, ldk, vec)
    // 32-feature chunks: the 17 KB LDS footprint keeps 7 workgroups per CU resident (measured
    // faster than one 64-feature chunk for the store-heavy materialisation)
    if (kind == RK_LAPLACE) RK_MAT(true, 32);
    else RK_MAT(false, 32);
#undef RK_MAT
    return launch_status("rk_kernel_materialize");
}

template <typename T, bool L1>
static void launch_kmatmat(int KB, dim3 grid, hipStream_t st, int kind, const T* X1, int64_t n1,
                           int64_t ldx1, const int64_t* idx1, const T* X2, int64_t n2, int64_t ldx2,
                           const int64_t* idx2, int64_t d, T w, const T* ell_vec, T scale,
                           const T* V, int k, int64_t ldv, int64_t tps, T* part, T* Y, int64_t ldy,
                           int accumulate) {
#define RK_KM(KBV)                                                                                 \
    do {                                                                                           \
        if (d <= 32)                                                                               \
            hipLaunchKernelGGL((kmat_matmat_kernel<T, L1, KBV, 32>), grid, dim3(256), 0, st, kind,  \
                               X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, w, ell_vec, scale, V, k,  \
                               ldv, tps, part, Y, ldy, accumulate);                                \
        else                                                                                       \
            hipLaunchKernelGGL((kmat_matmat_kernel<T, L1, KBV, 64>), grid, dim3(256), 0, st, kind,  \
                               X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, w, ell_vec, scale, V, k,  \
                               ldv, tps, part, Y, ldy, accumulate);                                \
    } while (0)
    switch (KB) {
        case 1: RK_KM(1); break;
        case 2: RK_KM(2); break;
        case 4: RK_KM(4); break;
        case 8: RK_KM(8); break;
        default: RK_KM(16); break;
    }
#undef RK_KM
}

template <typename T>
static int kmatmat_impl(int kind, const T* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                        const T* X2, int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d,
                        double ell, const T* ell_vec, double scale, const T* V, int64_t k,
                        int64_t ldv, T* Y, int64_t ldy, int accumulate, void* ws, size_t ws_bytes,
                        hipStream_t st) {
    const T w = ell_vec ? T(1) : T(1.0 / ell);
    const KMPlan p = plan_kmatmat(n1, n2);
    T* part = nullptr;
    if (p.S > 1) {
        const size_t need = (size_t)p.S * n1 * k * sizeof(T);
        if (ws == nullptr || ws_bytes < need) {
            set_error("rk_kernel_matmat: workspace %zu B < required %zu B", ws_bytes, need);
            return RK_ERR_WORKSPACE;
        }
        part = static_cast<T*>(ws);
    }
    const int KB = k <= 1 ? 1 : (k <= 2 ? 2 : (k <= 4 ? 4 : (k <= 8 ? 8 : 16)));
    dim3 grid((unsigned)ceil_div(n1, TM), (unsigned)p.S);
    if (kind == RK_LAPLACE)
        launch_kmatmat<T, true>(KB, grid, st, kind, X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, w,
                                ell_vec, (T)scale, V, (int)k, ldv, p.tiles_per_seg, part, Y, ldy,
                                accumulate);
    else
        launch_kmatmat<T, false>(KB, grid, st, kind, X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, w,
                                 ell_vec, (T)scale, V, (int)k, ldv, p.tiles_per_seg, part, Y, ldy,
                                 accumulate);
    int rc = launch_status("rk_kernel_matmat");
    if (rc != RK_OK || part == nullptr) return rc;
    hipLaunchKernelGGL((kmat_matmat_finish<T>), dim3((unsigned)ceil_div(n1 * k, 256)), dim3(256), 0,
                       st, part, (int)p.S, n1, (int)k, (T)scale, Y, ldy, accumulate);
    return launch_status("rk_kernel_matmat(finish)");
}

static int check_kernel_args(const char* fn, int dtype, int kind, const void* X1, int64_t n1,
                             int64_t ldx1, const void* X2, int64_t n2, int64_t ldx2, int64_t d,
                             double ell, const void* ell_vec) {
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "%s: bad dtype %d", fn, dtype);
    RK_REQUIRE(kind >= RK_RBF && kind <= RK_MATERN52, "%s: unknown kernel kind %d", fn, kind);
    RK_REQUIRE(X1 && X2, "%s: null data pointer", fn);
    RK_REQUIRE(n1 > 0 && n2 > 0 && d > 0, "%s: empty input (n1=%lld n2=%lld d=%lld)", fn,
               (long long)n1, (long long)n2, (long long)d);
    RK_REQUIRE(ldx1 >= d && ldx2 >= d, "%s: leading dimension smaller than d", fn);
    RK_REQUIRE(ell_vec != nullptr || ell > 0.0, "%s: lengthscale must be positive", fn);
    return RK_OK;
}

}  // namespace rk

using namespace rk;

extern "C" int rk_kernel_materialize(int dtype, int kind, const void* X1, int64_t n1, int64_t ldx1,
                                     const int64_t* idx1, const void* X2, int64_t n2, int64_t ldx2,
                                     const int64_t* idx2, int64_t d, double ell_scalar,
                                     const void* ell_vec, double scale, void* K, int64_t ldk,
                                     void* stream) {
    clear_error();
    int rc = check_kernel_args("rk_kernel_materialize", dtype, kind, X1, n1, ldx1, X2, n2, ldx2, d,
                               ell_scalar, ell_vec);
    if (rc != RK_OK) return rc;
    RK_REQUIRE(K != nullptr && ldk >= n2, "rk_kernel_materialize: bad output (ldk=%lld < n2=%lld?)",
               (long long)ldk, (long long)n2);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        return materialize_impl<float>(kind, (const float*)X1, n1, ldx1, idx1, (const float*)X2, n2,
                                       ldx2, idx2, d, ell_scalar, (const float*)ell_vec, scale,
                                       (float*)K, ldk, st);
    return materialize_impl<double>(kind, (const double*)X1, n1, ldx1, idx1, (const double*)X2, n2,
                                    ldx2, idx2, d, ell_scalar, (const double*)ell_vec, scale,
                                    (double*)K, ldk, st);
}

extern "C" size_t rk_kernel_matmat_ws_bytes(int dtype, int64_t n1, int64_t n2, int64_t k) {
    if (n1 <= 0 || n2 <= 0 || k <= 0 || k > 16) return 0;
    const KMPlan p = plan_kmatmat(n1, n2);
    return p.S > 1 ? (size_t)p.S * n1 * k * (dtype == RK_F32 ? 4 : 8) : 0;
}

extern "C" int rk_kernel_matmat(int dtype, int kind, const void* X1, int64_t n1, int64_t ldx1,
                                const int64_t* idx1, const void* X2, int64_t n2, int64_t ldx2,
                                const int64_t* idx2, int64_t d, double ell_scalar,
                                const void* ell_vec, double scale, const void* V, int64_t k,
                                int64_t ldv, void* Y, int64_t ldy, int accumulate, void* ws,
                                size_t ws_bytes, void* stream) {
    clear_error();
    int rc = check_kernel_args("rk_kernel_matmat", dtype, kind, X1, n1, ldx1, X2, n2, ldx2, d,
                               ell_scalar, ell_vec);
    if (rc != RK_OK) return rc;
    RK_REQUIRE(V && Y, "rk_kernel_matmat: null pointer");
    RK_REQUIRE(k >= 1 && k <= 16, "rk_kernel_matmat: k=%lld outside [1,16]", (long long)k);
    RK_REQUIRE(ldv >= k && ldy >= k, "rk_kernel_matmat: leading dimension too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        return kmatmat_impl<float>(kind, (const float*)X1, n1, ldx1, idx1, (const float*)X2, n2,
                                   ldx2, idx2, d, ell_scalar, (const float*)ell_vec, scale,
                                   (const float*)V, k, ldv, (float*)Y, ldy, accumulate, ws, ws_bytes,
                                   st);
    return kmatmat_impl<double>(kind, (const double*)X1, n1, ldx1, idx1, (const double*)X2, n2, ldx2,
                                idx2, d, ell_scalar, (const double*)ell_vec, scale, (const double*)V,
                                k, ldv, (double*)Y, ldy, accumulate, ws, ws_bytes, st);
}
