/*
 * C restatement (oracle) of the kernel mat-mat  Y = scale * K(X1, X2) @ V  - TEST / BASELINE
 * INFRASTRUCTURE ONLY (see oracle/__init__.py).  Never linked into the product.
 *
 * Follows the closed forms of the reference (rlaopt/kernels/standard.py:31-85; pinned by
 * tests/kernels/utils.py:4-60): D = (x - y) / lengthscale per feature, then
 *   rbf exp(-sum D^2 / 2), laplace exp(-sum |D|), matern12/32/52 on dist = sqrt(sum D^2).
 * Matrix-free, one row of K at a time, OpenMP over rows: this is also the CPU baseline that
 * bench.py times on the GPU box's host cores (the reference's CPU path is pykeops' matrix-free
 * CPU backend, which is not available offline).
 *
 * kind: 0 rbf, 1 laplace, 2 matern12, 3 matern32, 4 matern52.  ell_vec may be NULL (scalar).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DEFINE_ORACLE(NAME, T, EXP, SQRT, FABS)                                                  \
    int NAME(int kind, const T* X1, int64_t n1, const T* X2, int64_t n2, int64_t d, double ell,  \
             const T* ell_vec, double scale, const T* V, int64_t k, T* Y) {                      \
        if (kind < 0 || kind > 4 || n1 <= 0 || n2 <= 0 || d <= 0 || k <= 0) return -1;          \
        const T s3 = (T)1.7320508075688772, s5 = (T)2.23606797749979;                           \
        _Pragma("omp parallel") {                                                                \
            T* w = (T*)malloc((size_t)d * sizeof(T));                                           \
            T* xi = (T*)malloc((size_t)d * sizeof(T));                                          \
            T* acc = (T*)malloc((size_t)k * sizeof(T));                                         \
            for (int64_t c = 0; c < d; ++c) w[c] = (T)1 / (ell_vec ? ell_vec[c] : (T)ell);      \
            _Pragma("omp for schedule(dynamic, 8)") for (int64_t i = 0; i < n1; ++i) {          \
                for (int64_t c = 0; c < d; ++c) xi[c] = X1[i * d + c];                          \
                for (int64_t c = 0; c < k; ++c) acc[c] = 0;                                     \
                for (int64_t j = 0; j < n2; ++j) {                                              \
                    const T* y = X2 + j * d;                                                    \
                    T s = 0;                                                                    \
                    if (kind == 1) {                                                            \
                        _Pragma("omp simd reduction(+ : s)")                                    \
                        for (int64_t c = 0; c < d; ++c) s += FABS((xi[c] - y[c]) * w[c]);       \
                    } else {                                                                    \
                        _Pragma("omp simd reduction(+ : s)")                                    \
                        for (int64_t c = 0; c < d; ++c) {                                       \
                            const T df = (xi[c] - y[c]) * w[c];                                 \
                            s += df * df;                                                       \
                        }                                                                       \
                    }                                                                           \
                    T kv;                                                                       \
                    if (kind == 0) kv = EXP(-s / 2);                                            \
                    else if (kind == 1) kv = EXP(-s);                                           \
                    else {                                                                      \
                        const T D = SQRT(s);                                                    \
                        if (kind == 2) kv = EXP(-D);                                            \
                        else if (kind == 3) kv = ((T)1 + s3 * D) * EXP(-s3 * D);                \
                        else kv = ((T)1 + s5 * D + (T)(5.0 / 3.0) * D * D) * EXP(-s5 * D);      \
                    }                                                                           \
                    for (int64_t c = 0; c < k; ++c) acc[c] += kv * V[j * k + c];                \
                }                                                                               \
                for (int64_t c = 0; c < k; ++c) Y[i * k + c] = (T)scale * acc[c];               \
            }                                                                                   \
            free(w); free(xi); free(acc);                                                       \
        }                                                                                       \
        return 0;                                                                               \
    }

/* The same entries written out: K (n1 x n2, row-major) = scale * K(X1, X2).  Used by the parity tests that
 * need the dense operator of a mid-size system on the host (the reference's composite oracle: a dense torch
 * K behind its own LinOp). */
#define DEFINE_ORACLE_MATRIX(NAME, T, EXP, SQRT, FABS)                                           \
    int NAME(int kind, const T* X1, int64_t n1, const T* X2, int64_t n2, int64_t d, double ell,  \
             const T* ell_vec, double scale, T* K) {                                             \
        if (kind < 0 || kind > 4 || n1 <= 0 || n2 <= 0 || d <= 0) return -1;                    \
        const T s3 = (T)1.7320508075688772, s5 = (T)2.23606797749979;                           \
        _Pragma("omp parallel") {                                                                \
            T* w = (T*)malloc((size_t)d * sizeof(T));                                           \
            for (int64_t c = 0; c < d; ++c) w[c] = (T)1 / (ell_vec ? ell_vec[c] : (T)ell);      \
            _Pragma("omp for schedule(static)") for (int64_t i = 0; i < n1; ++i) {              \
                const T* xi = X1 + i * d;                                                       \
                for (int64_t j = 0; j < n2; ++j) {                                              \
                    const T* y = X2 + j * d;                                                    \
                    T s = 0;                                                                    \
                    if (kind == 1) {                                                            \
                        _Pragma("omp simd reduction(+ : s)")                                    \
                        for (int64_t c = 0; c < d; ++c) s += FABS((xi[c] - y[c]) * w[c]);       \
                    } else {                                                                    \
                        _Pragma("omp simd reduction(+ : s)")                                    \
                        for (int64_t c = 0; c < d; ++c) {                                       \
                            const T df = (xi[c] - y[c]) * w[c];                                 \
                            s += df * df;                                                       \
                        }                                                                       \
                    }                                                                           \
                    T kv;                                                                       \
                    if (kind == 0) kv = EXP(-s / 2);                                            \
                    else if (kind == 1) kv = EXP(-s);                                           \
                    else {                                                                      \
                        const T D = SQRT(s);                                                    \
                        if (kind == 2) kv = EXP(-D);                                            \
                        else if (kind == 3) kv = ((T)1 + s3 * D) * EXP(-s3 * D);                \
                        else kv = ((T)1 + s5 * D + (T)(5.0 / 3.0) * D * D) * EXP(-s5 * D);      \
                    }                                                                           \
                    K[i * n2 + j] = (T)scale * kv;                                              \
                }                                                                               \
            }                                                                                   \
            free(w);                                                                            \
        }                                                                                       \
        return 0;                                                                               \
    }

DEFINE_ORACLE_MATRIX(rk_oracle_kernel_matrix_f32, float, expf, sqrtf, fabsf)
DEFINE_ORACLE_MATRIX(rk_oracle_kernel_matrix_f64, double, exp, sqrt, fabs)
DEFINE_ORACLE(rk_oracle_kernel_matmat_f32, float, expf, sqrtf, fabsf)
DEFINE_ORACLE(rk_oracle_kernel_matmat_f64, double, exp, sqrt, fabs)

/* Y (n1 x k) = K (n1 x n2, row-major) @ V (n2 x k), k <= 16: the host's product with a kernel matrix that was
 * written out once (rows are dealt to the threads exactly as rk_oracle_kernel_matrix_* wrote them, so a thread
 * streams the pages it touched first).  Accumulation in the working type, 16 partial sums per row (omp simd). */
#define DEFINE_DENSE_MATMAT(NAME, T)                                                             \
    int NAME(const T* K, int64_t n1, int64_t n2, const T* V, int64_t k, T* Y) {                  \
        if (n1 <= 0 || n2 <= 0 || k <= 0 || k > 16) return -1;                                   \
        _Pragma("omp parallel for schedule(static)") for (int64_t i = 0; i < n1; ++i) {          \
            const T* row = K + i * n2;                                                           \
            if (k == 1) {                                                                        \
                T s = 0;                                                                         \
                _Pragma("omp simd reduction(+ : s)")                                             \
                for (int64_t j = 0; j < n2; ++j) s += row[j] * V[j];                             \
                Y[i] = s;                                                                        \
            } else {                                                                             \
                T acc[16];                                                                       \
                for (int64_t c = 0; c < k; ++c) acc[c] = 0;                                      \
                for (int64_t j = 0; j < n2; ++j) {                                               \
                    const T a = row[j];                                                          \
                    for (int64_t c = 0; c < k; ++c) acc[c] += a * V[j * k + c];                  \
                }                                                                                \
                for (int64_t c = 0; c < k; ++c) Y[i * k + c] = acc[c];                           \
            }                                                                                    \
        }                                                                                        \
        return 0;                                                                                \
    }

DEFINE_DENSE_MATMAT(rk_oracle_dense_matmat_f32, float)
DEFINE_DENSE_MATMAT(rk_oracle_dense_matmat_f64, double)

void rk_oracle_set_num_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#endif
}

int rk_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
