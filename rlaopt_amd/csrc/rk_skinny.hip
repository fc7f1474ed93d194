// HBM-streaming skinny products for gfx950 (MI355X).
//
//   rk_matmat_skinny    Y = alpha A V + beta C + shift V[row0+i]   (A m x n row-major, k <= 16)
//   rk_matmat_skinny_t  Y = alpha A^T V + beta C
//
// Design (DESIGN.md "Kernels"): these products are HBM-bound (k <= 16 gives <= 8 flop/B
// against a machine balance of ~26 flop/B), so the only thing that matters is streaming A
// exactly once with wide coalesced loads and enough bytes in flight per CU.
//   * one 64-lane wave owns R rows of A and walks a column segment in steps of 64 x 16 B
//     (1 KiB per row per step, `global_load_dwordx4`), so every row read is a full-line
//     coalesced burst; the next step's R loads are issued before the current step's FMAs
//     (explicit register double buffer: 2 x R KiB in flight per wave);
//   * the matching slice of V is read once per step per wave (1/R of the A traffic, served
//     by the XCD L2 - V is at most a few MB) and reused across the R rows in registers;
//   * each lane keeps R x k partial sums; one butterfly (DPP/ds_swizzle shuffles) per wave
//     at the end of the segment turns them into row sums - no LDS, no atomics;
//   * columns are cut into S segments so that the grid has >= ~20 rounds of resident waves
//     (tail < 5 %); per-segment partials are summed in a fixed order by a tiny second
//     kernel, which also applies the epilogue -> bitwise reproducible results.
#include <atomic>
#include <stdlib.h>

#include "rk_common.h"

namespace rk {

// ------------------------------------------------------------------------------------------
// Y = A V  (row-streaming)
// ------------------------------------------------------------------------------------------
template <typename T, int K, int R, bool VEC>
__global__ void __launch_bounds__(256)
skinny_nn_kernel(const T* __restrict__ A, int64_t m, int64_t n, int64_t lda,
                 const T* __restrict__ V, int64_t ldv, T* __restrict__ part, int64_t seg_cols,
                 // direct epilogue (used when gridDim.y == 1 and part == nullptr)
                 T alpha, T beta, const T* __restrict__ C, int64_t ldc, T shift, int64_t row0,
                 T* __restrict__ Y, int64_t ldy) {
    using V16 = typename Vec16<T>::type;
    constexpr int VW = Vec16<T>::N;
    constexpr int STEP = kWave * VW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // row pointers in SGPRs
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * R;
    if (r0 >= m) return;  // whole wave leaves; no block-level synchronisation below
    const int64_t c_lo = (int64_t)blockIdx.y * seg_cols;
    const int64_t c_hi = (c_lo + seg_cols < n) ? c_lo + seg_cols : n;

    const T* arow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t row = r0 + r;
        if (row > m - 1) row = m - 1;  // clamp: duplicate reads, never written
        arow[r] = A + row * lda;
    }
    T acc[R][K];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < K; ++c) acc[r][c] = T(0);

    int64_t jb = c_lo;
    if constexpr (VEC) {
        // Two register stages: the loads of step jb + STEP (V slice first, then the R rows) are
        // issued unconditionally in the same basic block that consumes step jb, so the compiler
        // waits for the older step only (vmcnt(R + K)) instead of draining the queue.
        V16 aA[R], aB[R], vA[K], vB[K];
        // addresses = wave-uniform pointer (SGPR pair) + 32-bit lane offset: no per-row 64-bit
        // VGPR address arithmetic, and no address temporaries aliasing the stage registers
        const unsigned lane_a = (unsigned)lane * 16u, lane_v = (unsigned)lane * (16u * K);
        auto fetch = [&](V16 a[R], V16 vv[K], int64_t jstep) {
            // V[j .. j+VW) x K: VW*K contiguous elements (ldv == K on this path)
            const char* vp = reinterpret_cast<const char*>(V + jstep * K) + lane_v;
#pragma unroll
            for (int t = 0; t < K; ++t) vv[t] = *reinterpret_cast<const V16*>(vp + 16 * t);
#pragma unroll
            for (int r = 0; r < R; ++r)
                a[r] = load_stream<V16>(reinterpret_cast<const char*>(arow[r] + jstep) + lane_a);
        };
        auto consume = [&](const V16 a[R], const V16 vv[K]) {
            const T* v = reinterpret_cast<const T*>(vv);  // v[q * K + c]
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T* ae = reinterpret_cast<const T*>(&a[r]);
#pragma unroll
                for (int q = 0; q < VW; ++q)
#pragma unroll
                    for (int c = 0; c < K; ++c) acc[r][c] += ae[q] * v[q * K + c];
            }
        };
        if (jb + STEP <= c_hi) {
            fetch(aA, vA, jb);
            for (;;) {
                if (jb + 2 * STEP > c_hi) { consume(aA, vA); jb += STEP; break; }
                fetch(aB, vB, jb + STEP);
                consume(aA, vA);
                jb += STEP;
                if (jb + 2 * STEP > c_hi) { consume(aB, vB); jb += STEP; break; }
                fetch(aA, vA, jb + STEP);
                consume(aB, vB);
                jb += STEP;
            }
        }
    }
    // scalar path: the ragged tail of the segment, or everything when !VEC
    for (; jb < c_hi; jb += STEP) {
        const int64_t j = jb + lane * VW;
#pragma unroll
        for (int q = 0; q < VW; ++q) {
            if (j + q < c_hi) {
                T v[K];
#pragma unroll
                for (int c = 0; c < K; ++c) v[c] = V[(j + q) * ldv + c];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T a = arow[r][j + q];
#pragma unroll
                    for (int c = 0; c < K; ++c) acc[r][c] += a * v[c];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < K; ++c) acc[r][c] = wave_total_lane63(acc[r][c]);

    if (lane == 63) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = r0 + r;
            if (row < m) {
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    if (part != nullptr) {
                        part[((int64_t)blockIdx.y * m + row) * K + c] = acc[r][c];
                    } else {
                        T y = alpha * acc[r][c];
                        if (beta != T(0)) y += beta * C[row * ldc + c];
                        if (shift != T(0)) y += shift * V[(row0 + row) * ldv + c];
                        Y[row * ldy + c] = y;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Y = A V for 5 <= k <= 16: the V slice of each column step is shared by the 4 waves of the
// workgroup through LDS (rhs-major, so a lane's 4 columns of one right-hand side are one
// ds_read_b128) instead of being re-read k times per lane from L2.  Per column step a
// workgroup moves 4 R KiB of A against one 1 x k KiB V slice; the next slice and the next A
// rows are fetched into registers before the FMAs of the current step (one barrier per step).
// ------------------------------------------------------------------------------------------
template <typename T, int K, int R>
__global__ void __launch_bounds__(256)
skinny_nn_lds_kernel(const T* __restrict__ A, int64_t m, int64_t n, int64_t lda,
                     const T* __restrict__ V, int64_t ldv, T* __restrict__ part, int64_t seg_cols,
                     T alpha, T beta, const T* __restrict__ C, int64_t ldc, T shift, int64_t row0,
                     T* __restrict__ Y, int64_t ldy, int vecA) {
    using V16 = typename Vec16<T>::type;
    constexpr int VW = Vec16<T>::N;
    constexpr int STEP = kWave * VW;
    constexpr int LDV = STEP + 4;            // 16-B aligned rows, <= 2-way store conflicts
    constexpr int NST = (STEP * K + 255) / 256;  // V elements staged per thread per step
    __shared__ T vs[2][K][LDV];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * R;
    const int64_t c_lo = (int64_t)blockIdx.y * seg_cols;
    const int64_t c_hi = (c_lo + seg_cols < n) ? c_lo + seg_cols : n;

    const T* arow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t row = r0 + r;
        if (row > m - 1) row = m - 1;  // clamp (all waves must reach the barriers)
        arow[r] = A + row * lda;
    }
    T acc[R][K];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < K; ++c) acc[r][c] = T(0);

    T sreg[NST];
    auto fetch_v = [&](int64_t jb) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int e = threadIdx.x + u * 256;
            const int col = e / K, c = e - col * K;
            const int64_t j = jb + col;
            sreg[u] = (e < STEP * K && j < c_hi) ? V[j * ldv + c] : T(0);
        }
    };
    auto store_v = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int e = threadIdx.x + u * 256;
            const int col = e / K, c = e - col * K;
            if (e < STEP * K) vs[buf][c][col] = sreg[u];
        }
    };
    auto load_a = [&](V16 dst[R], int64_t jb) {
        const int64_t j = jb + lane * VW;
        if (vecA && j + VW <= c_hi) {
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = load_stream<V16>(arow[r] + j);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                T* de = reinterpret_cast<T*>(&dst[r]);
#pragma unroll
                for (int q = 0; q < VW; ++q) de[q] = (j + q < c_hi) ? arow[r][j + q] : T(0);
            }
        }
    };

    V16 cur[R], nxt[R];
    fetch_v(c_lo);
    store_v(0);
    load_a(cur, c_lo);
    __syncthreads();
    int buf = 0;
    for (int64_t jb = c_lo; jb < c_hi; jb += STEP) {
        const bool more = jb + STEP < c_hi;  // uniform over the workgroup
        if (more) {
            fetch_v(jb + STEP);
            load_a(nxt, jb + STEP);
        }
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const V16 vq = *reinterpret_cast<const V16*>(&vs[buf][c][lane * VW]);
            const T* ve = reinterpret_cast<const T*>(&vq);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T* ae = reinterpret_cast<const T*>(&cur[r]);
#pragma unroll
                for (int q = 0; q < VW; ++q) acc[r][c] += ae[q] * ve[q];
            }
        }
        if (more) store_v(buf ^ 1);
        __syncthreads();
        buf ^= 1;
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = nxt[r];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < K; ++c) acc[r][c] = wave_total_lane63(acc[r][c]);
    if (lane == 63) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = r0 + r;
            if (row < m) {
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    if (part != nullptr) {
                        part[((int64_t)blockIdx.y * m + row) * K + c] = acc[r][c];
                    } else {
                        T y = alpha * acc[r][c];
                        if (beta != T(0)) y += beta * C[row * ldc + c];
                        if (shift != T(0)) y += shift * V[(row0 + row) * ldv + c];
                        Y[row * ldy + c] = y;
                    }
                }
            }
        }
    }
}

// Y[i][c] = alpha * sum_s part[s][i][c] + beta * C[i][c] + shift * V[row0+i][c]
template <typename T>
__global__ void __launch_bounds__(256)
skinny_nn_finish(const T* __restrict__ part, int S, int64_t m, int K, T alpha, T beta,
                 const T* __restrict__ C, int64_t ldc, T shift, const T* __restrict__ V,
                 int64_t ldv, int64_t row0, T* __restrict__ Y, int64_t ldy, int KT = 0) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * K) return;
    const int64_t i = e / K;
    const int c = (int)(e - i * K);
    if (KT == 0) KT = K;  // entries per table row (the matrix-core kernel carries 16)
    T s = T(0);
    for (int t = 0; t < S; ++t) s += part[((int64_t)t * m + i) * KT + c];
    T y = alpha * s;
    if (beta != T(0)) y += beta * C[i * ldc + c];
    if (shift != T(0)) y += shift * V[(row0 + i) * ldv + c];
    Y[i * ldy + c] = y;
}

// ------------------------------------------------------------------------------------------
// Y = A^T V  (column-owning lanes, rows streamed)
// ------------------------------------------------------------------------------------------
template <typename T, int K, bool VEC>
__global__ void __launch_bounds__(256)
skinny_t_kernel(const T* __restrict__ A, int64_t m, int64_t n, int64_t lda,
                const T* __restrict__ V, int64_t ldv, T* __restrict__ part, int64_t ncc,
                int64_t seg_rows, int64_t nwork) {
    using V16 = typename Vec16<T>::type;
    constexpr int VW = Vec16<T>::N;
    constexpr int STEP = kWave * VW;
    constexpr int UNR = 8;
    const int lane = threadIdx.x & 63;
    // wave index as a scalar: row addresses of V below become wave-uniform (scalar loads)
    const int64_t wid = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wid >= nwork) return;
    const int64_t cc = wid % ncc;
    const int64_t seg = wid / ncc;
    const int64_t j = cc * STEP + lane * VW;
    const int64_t i_lo = seg * seg_rows;
    const int64_t i_hi = (i_lo + seg_rows < m) ? i_lo + seg_rows : m;

    T acc[VW][K];
#pragma unroll
    for (int q = 0; q < VW; ++q)
#pragma unroll
        for (int c = 0; c < K; ++c) acc[q][c] = T(0);

    const bool full = VEC && (j + VW <= n);
    if (full) {
        int64_t i = i_lo;
        for (; i + UNR <= i_hi; i += UNR) {
            V16 a[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                a[u] = load_stream<V16>(A + (i + u) * lda + j);
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const T* ae = reinterpret_cast<const T*>(&a[u]);
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const T v = V[(i + u) * ldv + c];  // wave-uniform -> scalar load
#pragma unroll
                    for (int q = 0; q < VW; ++q) acc[q][c] += ae[q] * v;
                }
            }
        }
        for (; i < i_hi; ++i) {
            V16 a = load_stream<V16>(A + i * lda + j);
            const T* ae = reinterpret_cast<const T*>(&a);
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const T v = V[i * ldv + c];
#pragma unroll
                for (int q = 0; q < VW; ++q) acc[q][c] += ae[q] * v;
            }
        }
    } else {
        for (int64_t i = i_lo; i < i_hi; ++i) {
#pragma unroll
            for (int q = 0; q < VW; ++q) {
                if (j + q < n) {
                    const T a = A[i * lda + j + q];
#pragma unroll
                    for (int c = 0; c < K; ++c) acc[q][c] += a * V[i * ldv + c];
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < VW; ++q) {
        if (j + q < n) {
#pragma unroll
            for (int c = 0; c < K; ++c) part[(seg * n + j + q) * K + c] = acc[q][c];
        }
    }
}

// out[e] = alpha * sum_s part[s][e] + beta * C[e]; the S partials are summed in float64 by
// 16 slices per output (64 outputs x 16 slices per workgroup), combined in a fixed order
// through LDS.
template <typename T, typename TO>
__global__ void __launch_bounds__(1024)
skinny_t_finish(const T* __restrict__ part, int64_t S, int64_t n, int K, double alpha, double beta,
                const T* __restrict__ C, int64_t ldc, TO* __restrict__ Y, int64_t ldy) {
    constexpr int NS = 16;
    __shared__ double red[NS][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + tx;
    const int64_t total = n * K;
    double s = 0.0;
    if (e < total) {
        for (int64_t t = ty; t < S; t += NS) s += (double)part[t * total + e];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && e < total) {
        s = 0.0;
#pragma unroll
        for (int y = 0; y < NS; ++y) s += red[y][tx];
        const int64_t jrow = e / K;
        const int c = (int)(e - jrow * K);
        double y = alpha * s;
        if (beta != 0.0) y += beta * (double)C[jrow * ldc + c];
        Y[jrow * ldy + c] = (TO)y;
    }
}

// ------------------------------------------------------------------------------------------
// block-PCG vector updates and the k x k solve
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
pcg_update_wr_kernel(int64_t n, int k, T* __restrict__ W, int64_t ldw, T* __restrict__ R,
                     int64_t ldr, const T* __restrict__ P, int64_t ldp,
                     const T* __restrict__ AP, int64_t ldap, const double* __restrict__ alpha) {
    __shared__ T a[16 * 16];
    if (threadIdx.x < k * k) a[threadIdx.x] = (T)alpha[threadIdx.x];
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * k) return;
    const int64_t i = e / k;
    const int c = (int)(e - i * k);
    T dw = T(0), dr = T(0);
    for (int j = 0; j < k; ++j) {
        dw += P[i * ldp + j] * a[j * k + c];
        dr += AP[i * ldap + j] * a[j * k + c];
    }
    W[i * ldw + c] += dw;
    R[i * ldr + c] -= dr;
}

template <typename T>
__global__ void __launch_bounds__(256)
pcg_update_p_kernel(int64_t n, int k, T* __restrict__ P, int64_t ldp, const T* __restrict__ Z,
                    int64_t ldz, const double* __restrict__ beta) {
    __shared__ T b[16 * 16];
    if (threadIdx.x < k * k) b[threadIdx.x] = (T)beta[threadIdx.x];
    __syncthreads();
    // one thread per ROW: the whole old row of P is needed before any entry is overwritten
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T oldp[16];
    for (int j = 0; j < k; ++j) oldp[j] = P[i * ldp + j];
    for (int c = 0; c < k; ++c) {
        T s = Z[i * ldz + c];
        for (int j = 0; j < k; ++j) s += oldp[j] * b[j * k + c];
        P[i * ldp + c] = s;
    }
}

// ------------------------------------------------------------------------------------------
// G (k1 x k2, float64) = X^T Y for two tall row-major blocks with k1, k2 <= 4: the block inner
// products of PCG (P^T AP, R^T Z), column norms, LSQR norms.  Stage 1: grid-stride over rows,
// k1 k2 float64 accumulators per thread, wave total by DPP, 4 waves through LDS -> one partial
// per workgroup; stage 2: one workgroup adds the partials in a fixed order.
// ------------------------------------------------------------------------------------------
template <typename T, int K1, int K2>
__global__ void __launch_bounds__(256)
gram_small_kernel(const T* __restrict__ X, int64_t ldx, const T* __restrict__ Y, int64_t ldy,
                  int64_t n, double* __restrict__ part) {
    __shared__ double red[4][K1 * K2];
    double acc[K1][K2];
#pragma unroll
    for (int a = 0; a < K1; ++a)
#pragma unroll
        for (int b = 0; b < K2; ++b) acc[a][b] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        T x[K1], y[K2];
#pragma unroll
        for (int a = 0; a < K1; ++a) x[a] = X[i * ldx + a];
#pragma unroll
        for (int b = 0; b < K2; ++b) y[b] = Y[i * ldy + b];
#pragma unroll
        for (int a = 0; a < K1; ++a)
#pragma unroll
            for (int b = 0; b < K2; ++b) acc[a][b] += (double)x[a] * (double)y[b];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < K1; ++a)
#pragma unroll
        for (int b = 0; b < K2; ++b) {
            const double tot = wave_total_lane63(acc[a][b]);
            if (lane == 63) red[wave][a * K2 + b] = tot;
        }
    __syncthreads();
    if (threadIdx.x < K1 * K2)
        part[(int64_t)blockIdx.x * (K1 * K2) + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// one wave per output entry: lane l adds the partials l, l + 64, ... (independent loads in
// flight), then one DPP wave total - fixed order, reproducible
__global__ void __launch_bounds__(64)
gram_small_finish(const double* __restrict__ part, int nblocks, int kk, double* __restrict__ G) {
    const int e = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += part[(int64_t)b * kk + e];
    s = wave_total_lane63(s);
    if (threadIdx.x == 63) G[e] = s;
}

// ------------------------------------------------------------------------------------------
// Sparse-sign sketch apply  Y (s x d) = scale * Omega A  for Omega (s x m) with <= zeta signed
// entries per COLUMN (rlaopt/sketches/sparse.py:52-76), given in row-sorted (CSR) form:
// Y[r, :] = scale * sum_{e in row r} sign[e] * A[col[e], :].  A gather of whole rows (1 KiB
// per wave load), no atomics, fixed summation order: each A row is read zeta times instead of
// multiplying by a dense s x m matrix (zeta / s of the flops, none of its 8 s m bytes).
// One workgroup per (sketch row, 64 x 16 B column chunk); its 4 waves split the row's entries.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
sparse_sign_apply_kernel(const T* __restrict__ A, int64_t d, int64_t lda,
                         const int* __restrict__ col, const signed char* __restrict__ sgn,
                         const int64_t* __restrict__ row_ptr, T scale, T* __restrict__ Y,
                         int64_t ldy) {
    using V16 = typename Vec16<T>::type;
    constexpr int VW = Vec16<T>::N;
    constexpr int UNR = 8;
    __shared__ T red[4][kWave * VW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r = blockIdx.y;
    const int64_t j = ((int64_t)blockIdx.x * kWave + lane) * VW;
    const bool vec = (j + VW <= d) && (lda % VW == 0);
    const int64_t e_lo = row_ptr[r], e_hi = row_ptr[r + 1];
    const int64_t per = (e_hi - e_lo + 3) / 4;
    const int64_t lo = e_lo + wave * per;
    const int64_t hi = (lo + per < e_hi) ? lo + per : e_hi;
    T acc[VW];
#pragma unroll
    for (int q = 0; q < VW; ++q) acc[q] = T(0);
    int64_t e = lo;
    if (vec) {
        for (; e + UNR <= hi; e += UNR) {
            V16 a[UNR];
            T sg[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                a[u] = *reinterpret_cast<const V16*>(A + (int64_t)col[e + u] * lda + j);
                sg[u] = (T)sgn[e + u];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const T* ae = reinterpret_cast<const T*>(&a[u]);
#pragma unroll
                for (int q = 0; q < VW; ++q) acc[q] += sg[u] * ae[q];
            }
        }
    }
    for (; e < hi; ++e) {
        const T* row = A + (int64_t)col[e] * lda;
        const T sg = (T)sgn[e];
#pragma unroll
        for (int q = 0; q < VW; ++q)
            if (j + q < d) acc[q] += sg * row[j + q];
    }
#pragma unroll
    for (int q = 0; q < VW; ++q) red[wave][lane * VW + q] = acc[q];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < VW; ++q) {
            const int idx = lane * VW + q;
            if (j + q < d)
                Y[r * ldy + j + q] = scale * (((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx]);
        }
    }
}

// Y[:, c] = a[c] * X[:, c] + b[c] * Y[:, c] with device-resident float64 coefficients
template <typename T>
__global__ void __launch_bounds__(256)
axpby_cols_kernel(int64_t n, int k, const T* __restrict__ X, int64_t ldx,
                  const double* __restrict__ a, T* __restrict__ Y, int64_t ldy,
                  const double* __restrict__ b) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * k) return;
    const int64_t i = e / k;
    const int c = (int)(e - i * k);
    T y = (a ? (T)a[c] : T(1)) * X[i * ldx + c];
    if (b) y += (T)b[c] * Y[i * ldy + c];
    Y[i * ldy + c] = y;
}

// Gaussian elimination with partial pivoting on the augmented matrix [A | B] in LDS.
__global__ void __launch_bounds__(512)
small_solve_kernel(int k, int nrhs, const double* __restrict__ A, const double* __restrict__ B,
                   double* __restrict__ X) {
    __shared__ double M[16][33];
    __shared__ int piv;
    const int r = threadIdx.x >> 5, c = threadIdx.x & 31;  // 16 x 32 threads
    const int w = k + nrhs;
    if (r < k && c < w) M[r][c] = (c < k) ? A[r * k + c] : B[r * nrhs + (c - k)];
    __syncthreads();
    for (int p = 0; p < k; ++p) {
        if (threadIdx.x == 0) {
            int best = p;
            double bv = fabs(M[p][p]);
            for (int q = p + 1; q < k; ++q) {
                const double v = fabs(M[q][p]);
                if (v > bv) { bv = v; best = q; }
            }
            piv = best;
        }
        __syncthreads();
        const int bp = piv;
        if (bp != p && r == 0 && c < w) {
            const double t = M[p][c];
            M[p][c] = M[bp][c];
            M[bp][c] = t;
        }
        __syncthreads();
        double f = 0.0, pr = 0.0;
        const bool act = (r > p && r < k && c > p && c < w);
        if (act) { f = M[r][p] / M[p][p]; pr = M[p][c]; }
        __syncthreads();
        if (act) M[r][c] -= f * pr;
        __syncthreads();
    }
    for (int p = k - 1; p >= 0; --p) {
        if (r == 0 && c >= k && c < w) M[p][c] /= M[p][p];
        __syncthreads();
        if (r < p && c >= k && c < w) M[r][c] -= M[r][p] * M[p][c];
        __syncthreads();
    }
    if (r < k && c >= k && c < w) X[r * nrhs + (c - k)] = M[r][c];
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename T>
__global__ void __launch_bounds__(256)
randn_kernel(T* __restrict__ out, int64_t rows, int64_t cols, int64_t ld, uint64_t seed,
             uint64_t offset, double scale) {
    // one Philox call -> 4 uniforms -> 2 Box-Muller pairs -> logical elements 4g .. 4g+3
    const int64_t total = rows * cols;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g * 4 >= total) return;
    uint32_t u[4];
    philox4x32_10((uint32_t)g, (uint32_t)((uint64_t)g >> 32), (uint32_t)offset,
                  (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), u);
    double z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double u1 = ((double)u[2 * h] + 1.0) * (1.0 / 4294967296.0);  // (0, 1]
        const double u2 = (double)u[2 * h + 1] * (1.0 / 4294967296.0);      // [0, 1)
        const double rad = sqrt(-2.0 * log(u1));
        double s, co;
        sincospi(2.0 * u2, &s, &co);
        z[2 * h] = rad * co;
        z[2 * h + 1] = rad * s;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t e = g * 4 + q;
        if (e < total) {
            const int64_t i = e / cols;
            out[i * ld + (e - i * cols)] = (T)(scale * z[q]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host-side planning + dispatch
// ------------------------------------------------------------------------------------------
struct NNPlan {
    int R;             // rows per wave
    int64_t S;         // column segments
    int64_t seg_cols;  // multiple of the wave step
};

// ------------------------------------------------------------------------------------------
// Y = A V for 9 .. 16 right-hand sides on the matrix cores, operands straight from global memory
// ------------------------------------------------------------------------------------------
// At 16 right-hand sides the vector units cannot keep up with the HBM stream (16 FMAs per loaded
// element: the LDS-shared kernel above reaches 2.2 TB/s in float32).  v_mfma_*_16x16x4 wants its A
// operand as lane (row = lane % 16, k = lane / 16): a lane loads 16 bytes of row lane % 16 at byte
// offset 16 (lane / 16) of the step, so one load instruction fetches 16 rows x 64 bytes and each of its
// components feeds one MFMA - no LDS, no barrier.  A wave owns 64 rows (four 16-row blocks share the B
// operands: V of the step's columns, from a re-packed copy in which a lane's values are contiguous)
// and walks its column segment in steps of 256 bytes per row, three steps in flight in registers
// (240 of the wave's 512 registers: one wave per SIMD).  The loads are ordinary cached ones: the two
// halves of a 128-byte line arrive in consecutive instructions and must meet in the vector cache
// (non-temporal loads: 4.5 TB/s instead of 5.5).  Measured on the 25 000 x 200 000 float32 shard, any
// 9 <= k <= 16: 5.97 TB/s (2 row blocks per wave: 5.4; 8: 5.8; two steps in flight: 5.7; the previous
// kernel of this path - 32 x 256 tiles staged in LDS, XOR-swizzled, two tiles in flight - 5.0, and
// 5.5 with its MFMAs and V loads removed: the global -> register -> LDS -> register hop was the bound).
constexpr int MM_K = 16;     // right-hand sides carried (k is padded up to it)
constexpr int MD_RB = 4;     // 16-row blocks per wave
constexpr int MD_DEPTH = 3;  // steps in flight
constexpr int MD_ROWS = 64 * MD_RB;  // rows per workgroup (4 waves)

template <typename T>
struct MdAcc;
template <>
struct MdAcc<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ type mfma(float a, float b, type c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // accumulator element v of lane group lq = lane / 16 sits in row `row(lq, v)` of the 16 x 16 result
    static __device__ __forceinline__ int row(int lq, int v) { return 4 * lq + v; }
};
template <>
struct MdAcc<double> {
    typedef double type __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ type mfma(double a, double b, type c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lq, int v) { return lq + 4 * v; }
};

// a step = 4 loads of 16 bytes per lane = SC columns; element q of load s of lane (lr, lq) is column
// SC T + 4 VW s + VW lq + q:  Vd[((T * 4 + s) * 64 + lane) * VW + q] = V[that column][rhs lr]
template <typename T>
__global__ void __launch_bounds__(256)
md_pack_v_kernel(const T* __restrict__ V, int64_t n, int64_t nstep, int k, int64_t ldv, T* __restrict__ Vd) {
    constexpr int VW = Vec16<T>::N;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nstep * 256 * VW) return;
    const int q = (int)(e % VW);
    const int64_t g = e / VW;
    const int lane = (int)(g & 63), s4 = (int)((g >> 6) & 3);
    const int64_t t = g >> 8;
    const int64_t col = (int64_t)16 * VW * t + 4 * VW * s4 + VW * (lane >> 4) + q;
    const int c = lane & 15;
    Vd[e] = (col < n && c < k) ? V[col * ldv + c] : T(0);
}

template <typename T>
__global__ void __launch_bounds__(256, 1)
skinny_nn_mfma_kernel(const T* __restrict__ A, int64_t m, int64_t n, int64_t lda, const T* __restrict__ Vd,
                      T* __restrict__ part, int64_t seg_steps, int64_t nstep) {
    using V16 = typename Vec16<T>::type;
    using Acc = typename MdAcc<T>::type;
    constexpr int VW = Vec16<T>::N;  // elements per 16-byte load: 4 / 2
    constexpr int SC = 16 * VW;      // columns per step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * MD_ROWS + 16 * MD_RB * wave;
    if (i0 >= m) return;  // (no barriers in this kernel)
    const int64_t t_lo = (int64_t)blockIdx.y * seg_steps;
    const int64_t t_hi = t_lo + seg_steps < nstep ? t_lo + seg_steps : nstep;
    if (t_lo >= t_hi) return;
    const T* ar[MD_RB];  // rows past the matrix re-read its last row (their results are not stored)
#pragma unroll
    for (int b = 0; b < MD_RB; ++b) {
        const int64_t r = i0 + 16 * b + lr < m ? i0 + 16 * b + lr : m - 1;
        ar[b] = A + r * lda + VW * lq;
    }
    const V16* vd = reinterpret_cast<const V16*>(Vd) + lane;
    // full steps: SC (T + 1) <= n.  The last step of the matrix may be ragged (n % VW == 0): it runs
    // after the pipelined loop with clamped, zeroed loads.
    const int64_t t_full = n / SC < t_hi ? n / SC : t_hi;
    Acc c[MD_RB];
#pragma unroll
    for (int b = 0; b < MD_RB; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) c[b][v] = T(0);
    struct Step {
        V16 v[4], x[MD_RB][4];
    };
    // every load of the pipelined loop is unconditional (a step past the end re-reads the last full
    // one): the compiler then waits with partial vmcnt counts and the queue never drains
    auto load = [&](int64_t t, Step& st) {
        t = t < t_full ? t : t_full - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) st.v[s] = vd[(t * 4 + s) * 64];
#pragma unroll
        for (int b = 0; b < MD_RB; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) st.x[b][s] = *reinterpret_cast<const V16*>(ar[b] + SC * t + 4 * VW * s);
    };
    auto fma = [&](const Step& st) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const T* ve = reinterpret_cast<const T*>(&st.v[s]);
#pragma unroll
            for (int q = 0; q < VW; ++q)
#pragma unroll
                for (int b = 0; b < MD_RB; ++b)
                    c[b] = MdAcc<T>::mfma(reinterpret_cast<const T*>(&st.x[b][s])[q], ve[q], c[b]);
        }
    };
    if (t_full > t_lo) {
        Step st[MD_DEPTH];
#pragma unroll
        for (int j = 0; j < MD_DEPTH; ++j) load(t_lo + j, st[j]);
        for (int64_t t = t_lo; t < t_full; t += MD_DEPTH) {
#pragma unroll
            for (int j = 0; j < MD_DEPTH; ++j) {
                if (j == 0 || t + j < t_full) fma(st[j]);
                load(t + MD_DEPTH + j, st[j]);
            }
        }
    }
    if (t_full < t_hi) {  // the ragged step (last segment only, n % SC != 0)
        const int64_t t = t_full;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t col = SC * t + 4 * VW * s + VW * lq;
            const bool in = col < n;
            const int64_t cc = (in ? col : n - VW) - VW * lq;  // (ar[] already carries VW lq)
            const T msk = in ? T(1) : T(0);
            const V16 v = vd[(t * 4 + s) * 64];
            const T* ve = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int b = 0; b < MD_RB; ++b) {
                const V16 x = *reinterpret_cast<const V16*>(ar[b] + cc);
                const T* xe = reinterpret_cast<const T*>(&x);
#pragma unroll
                for (int q = 0; q < VW; ++q) c[b] = MdAcc<T>::mfma(xe[q] * msk, ve[q], c[b]);
            }
        }
    }
    // c[b][v] = Y[row i0 + 16 b + row(lq, v)][rhs lr] over this workgroup's column segment
#pragma unroll
    for (int b = 0; b < MD_RB; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t i = i0 + 16 * b + MdAcc<T>::row(lq, v);
            if (i < m) part[((int64_t)blockIdx.y * m + i) * MM_K + lr] = c[b][v];
        }
}

struct MMPlan {
    int64_t nstep, S, seg_steps;
};
template <typename T>
static MMPlan plan_mm(int64_t m, int64_t n) {
    MMPlan p;
    p.nstep = ceil_div(n, (int64_t)16 * Vec16<T>::N);
    const int64_t row_blocks = ceil_div(m, MD_ROWS);
    int64_t S = 4096 / row_blocks;           // one workgroup per CU is resident: ~16 rounds
    if (S > p.nstep / 24) S = p.nstep / 24;  // >= 24 steps per wave
    if (S < 1) S = 1;
    p.seg_steps = ceil_div(p.nstep, S);
    p.S = ceil_div(p.nstep, p.seg_steps);
    return p;
}
template <typename T>
static size_t mm_ws_bytes(int64_t m, int64_t n) {
    const MMPlan p = plan_mm<T>(m, n);
    return ((size_t)p.nstep * 256 * Vec16<T>::N + (size_t)p.S * m * MM_K) * sizeof(T);
}
// RK_SKINNY_MFMA: 0 never, 2 always, otherwise by size: the matrix-core path is three launches (re-pack V,
// product, fixed-order sum of the column segments), the LDS-shared kernel one or two - below ~16 MB of
// matrix the launches outweigh the streaming rate (SAP's 5 000 x 100 basis, the 100 x 100 apply factor)
static bool mm_enabled(int64_t m, int64_t n, size_t elem) {
    const char* e = getenv("RK_SKINNY_MFMA");
    if (e != nullptr && e[0] == '0') return false;
    if (e != nullptr && e[0] == '2') return true;
    return (size_t)m * (size_t)n * elem >= ((size_t)16 << 20);
}
// smallest k that takes this kernel: float32 9 (5.9 against 5.1 TB/s of the LDS-shared kernel; at k = 8 that
// one streams 6.3), float64 7 (5.8 against 4.5 at k = 7, 8; at k = 6: 6.6)
template <typename T>
constexpr int64_t mm_min_k() { return sizeof(T) == 4 ? 9 : 7; }

// k <= 2: register kernel (V slice in registers); k >= 3: LDS-shared V slice (25 000 x 200 000 float32, k = 3 / 4:
// 6.64 / 6.67 TB/s against 6.46 / 6.05 of the register kernel; float64 alike).  RK_SKINNY_LDS_MIN = 3 .. 5
// moves the switch for experiments.
static int64_t lds_min_k() {
    const char* e = getenv("RK_SKINNY_LDS_MIN");
    const int64_t v = e ? atoll(e) : 0;
    return (v >= 3 && v <= 5) ? v : 3;
}
static int rows_per_wave(int64_t k) {
    if (k >= lds_min_k()) return k <= 8 ? 8 : 4;
    return k <= 2 ? 8 : 4;
}

static NNPlan plan_nn(int dtype, int64_t m, int64_t n, int64_t k) {
    const int64_t step = (dtype == RK_F32) ? 256 : 128;
    NNPlan p;
    p.R = rows_per_wave(k);
    const int64_t row_waves = ceil_div(m, p.R);
    // >= ~80k waves in total (20 rounds of 4096 resident waves) but >= 16 steps per wave
    int64_t S = ceil_div(81920, row_waves);
    const int64_t max_S = n / (16 * step) > 0 ? n / (16 * step) : 1;
    if (S > max_S) S = max_S;
    if (S < 1) S = 1;
    p.seg_cols = round_up(ceil_div(n, S), step);
    p.S = ceil_div(n, p.seg_cols);
    return p;
}

template <typename T, int K, int R>
static void launch_nn(bool vec, dim3 grid, hipStream_t st, const T* A, int64_t m, int64_t n,
                      int64_t lda, const T* V, int64_t ldv, T* part, int64_t seg_cols, T alpha,
                      T beta, const T* C, int64_t ldc, T shift, int64_t row0, T* Y, int64_t ldy) {
    if (vec)
        hipLaunchKernelGGL((skinny_nn_kernel<T, K, R, true>), grid, dim3(256), 0, st, A, m, n, lda,
                           V, ldv, part, seg_cols, alpha, beta, C, ldc, shift, row0, Y, ldy);
    else
        hipLaunchKernelGGL((skinny_nn_kernel<T, K, R, false>), grid, dim3(256), 0, st, A, m, n,
                           lda, V, ldv, part, seg_cols, alpha, beta, C, ldc, shift, row0, Y, ldy);
}

template <typename T>
static int matmat_skinny_impl(const T* A, int64_t m, int64_t n, int64_t lda, const T* V, int64_t k,
                              int64_t ldv, double alpha, double beta, const T* C, int64_t ldc,
                              double shift, int64_t row0, T* Y, int64_t ldy, void* ws,
                              size_t ws_bytes, hipStream_t st, int dtype) {
    constexpr int VW = Vec16<T>::N;
    if (k >= mm_min_k<T>() && aligned16(A) && lda % VW == 0 && n % VW == 0 && n >= 16 * VW && mm_enabled(m, n, sizeof(T))) {
        const MMPlan q = plan_mm<T>(m, n);
        const size_t need = mm_ws_bytes<T>(m, n);
        if (ws == nullptr || ws_bytes < need) {
            set_error("rk_matmat_skinny: workspace %zu B < required %zu B", ws_bytes, need);
            return RK_ERR_WORKSPACE;
        }
        T* Vd = static_cast<T*>(ws);
        T* partq = Vd + (size_t)q.nstep * 256 * VW;
        hipLaunchKernelGGL((md_pack_v_kernel<T>), dim3((unsigned)ceil_div(q.nstep * 256 * VW, 256)), dim3(256), 0, st, V,
                           n, q.nstep, (int)k, ldv, Vd);
        dim3 gridq((unsigned)ceil_div(m, MD_ROWS), (unsigned)q.S);
        hipLaunchKernelGGL((skinny_nn_mfma_kernel<T>), gridq, dim3(256), 0, st, A, m, n, lda, Vd, partq, q.seg_steps,
                           q.nstep);
        int rcq = launch_status("rk_matmat_skinny(mfma)");
        if (rcq != RK_OK) return rcq;
        hipLaunchKernelGGL((skinny_nn_finish<T>), dim3((unsigned)ceil_div(m * k, 256)), dim3(256), 0, st, partq, (int)q.S,
                           m, (int)k, (T)alpha, (T)beta, C, ldc, (T)shift, V, ldv, row0, Y, ldy, MM_K);
        return launch_status("rk_matmat_skinny(finish)");
    }
    const NNPlan p = plan_nn(dtype, m, n, k);
    const bool vec = aligned16(A) && aligned16(V) && (lda % VW == 0) && (ldv == k);
    T* part = nullptr;
    if (p.S > 1) {
        const size_t need = (size_t)p.S * m * k * sizeof(T);
        if (ws == nullptr || ws_bytes < need) {
            set_error("rk_matmat_skinny: workspace %zu B < required %zu B", ws_bytes, need);
            return RK_ERR_WORKSPACE;
        }
        part = static_cast<T*>(ws);
    }
    dim3 grid((unsigned)ceil_div(ceil_div(m, p.R), 4), (unsigned)p.S);
#define RK_NN_CASE(KK, RR)                                                                       \
    case KK:                                                                                     \
        launch_nn<T, KK, RR>(vec, grid, st, A, m, n, lda, V, ldv, part, p.seg_cols, (T)alpha,   \
                             (T)beta, C, ldc, (T)shift, row0, Y, ldy);                          \
        break;
#define RK_NN_LDS_CASE(KK, RR)                                                                   \
    case KK:                                                                                     \
        hipLaunchKernelGGL((skinny_nn_lds_kernel<T, KK, RR>), grid, dim3(256), 0, st, A, m, n,   \
                           lda, V, ldv, part, p.seg_cols, (T)alpha, (T)beta, C, ldc, (T)shift,   \
                           row0, Y, ldy, vecA);                                                  \
        break;
    const int vecA = aligned16(A) && (lda % VW == 0);
    if (k >= lds_min_k() && k < 5) {  // k = 3, 4: the LDS-shared kernel (unless RK_SKINNY_LDS_MIN moves the switch)
        switch (k) {
            RK_NN_LDS_CASE(3, 8) RK_NN_LDS_CASE(4, 8)
            default: break;
        }
    } else
    switch (k) {
        RK_NN_CASE(1, 8) RK_NN_CASE(2, 8) RK_NN_CASE(3, 4) RK_NN_CASE(4, 4)
        RK_NN_LDS_CASE(5, 8) RK_NN_LDS_CASE(6, 8) RK_NN_LDS_CASE(7, 8) RK_NN_LDS_CASE(8, 8)
        RK_NN_LDS_CASE(9, 4) RK_NN_LDS_CASE(10, 4) RK_NN_LDS_CASE(11, 4) RK_NN_LDS_CASE(12, 4)
        RK_NN_LDS_CASE(13, 4) RK_NN_LDS_CASE(14, 4) RK_NN_LDS_CASE(15, 4) RK_NN_LDS_CASE(16, 4)
        default:
            set_error("rk_matmat_skinny: k=%lld outside [1,16]", (long long)k);
            return RK_ERR_INVALID;
    }
#undef RK_NN_LDS_CASE
#undef RK_NN_CASE
    int rc = launch_status("rk_matmat_skinny");
    if (rc != RK_OK) return rc;
    if (part != nullptr) {
        const int64_t total = m * k;
        hipLaunchKernelGGL((skinny_nn_finish<T>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                           st, part, (int)p.S, m, (int)k, (T)alpha, (T)beta, C, ldc, (T)shift, V,
                           ldv, row0, Y, ldy);
        rc = launch_status("rk_matmat_skinny(finish)");
    }
    return rc;
}

struct TPlan {
    int64_t ncc, S, seg_rows;
};

static TPlan plan_t(int dtype, int64_t m, int64_t n) {
    const int64_t step = (dtype == RK_F32) ? 256 : 128;
    TPlan p;
    p.ncc = ceil_div(n, step);
    int64_t S = ceil_div(16384, p.ncc);  // ~16k waves (4 rounds of the resident set)
    // >= 256 rows per wave; a short operand (the 5 000 x 100 Nystrom factor of a SAP block: 20 waves
    // at 256 rows, 37 us) is cut finer - it is latency-, not bandwidth-bound
    const int64_t max_S = (m >= 2048 && m < 65536) ? ceil_div(m, 32) : ceil_div(m, 256);
    if (S > max_S) S = max_S;
    if (S < 1) S = 1;
    p.seg_rows = ceil_div(m, S);
    p.S = ceil_div(m, p.seg_rows);
    return p;
}

template <typename T>
static int matmat_skinny_t_impl(const T* A, int64_t m, int64_t n, int64_t lda, const T* V,
                                int64_t k, int64_t ldv, double alpha, double beta, const T* C,
                                int64_t ldc, void* Y, int64_t ldy, int out_f64, void* ws,
                                size_t ws_bytes, hipStream_t st, int dtype) {
    constexpr int VW = Vec16<T>::N;
    const TPlan p = plan_t(dtype, m, n);
    const size_t need = (size_t)p.S * n * k * sizeof(T);
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_matmat_skinny_t: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    T* part = static_cast<T*>(ws);
    const bool vec = aligned16(A) && (lda % VW == 0);
    const int64_t nwork = p.ncc * p.S;
    dim3 grid((unsigned)ceil_div(nwork, 4));
#define RK_T_CASE(KK)                                                                            \
    case KK:                                                                                     \
        if (vec)                                                                                 \
            hipLaunchKernelGGL((skinny_t_kernel<T, KK, true>), grid, dim3(256), 0, st, A, m, n,  \
                               lda, V, ldv, part, p.ncc, p.seg_rows, nwork);                     \
        else                                                                                     \
            hipLaunchKernelGGL((skinny_t_kernel<T, KK, false>), grid, dim3(256), 0, st, A, m, n, \
                               lda, V, ldv, part, p.ncc, p.seg_rows, nwork);                     \
        break;
    switch (k) {
        RK_T_CASE(1) RK_T_CASE(2) RK_T_CASE(3) RK_T_CASE(4) RK_T_CASE(5) RK_T_CASE(6)
        RK_T_CASE(7) RK_T_CASE(8) RK_T_CASE(9) RK_T_CASE(10) RK_T_CASE(11) RK_T_CASE(12)
        RK_T_CASE(13) RK_T_CASE(14) RK_T_CASE(15) RK_T_CASE(16)
        default:
            set_error("rk_matmat_skinny_t: k=%lld outside [1,16]", (long long)k);
            return RK_ERR_INVALID;
    }
#undef RK_T_CASE
    int rc = launch_status("rk_matmat_skinny_t");
    if (rc != RK_OK) return rc;
    const int64_t total = n * k;
    dim3 fgrid((unsigned)ceil_div(total, 64));
    if (out_f64)
        hipLaunchKernelGGL((skinny_t_finish<T, double>), fgrid, dim3(1024), 0, st, part, p.S, n,
                           (int)k, alpha, beta, C, ldc, static_cast<double*>(Y), ldy);
    else
        hipLaunchKernelGGL((skinny_t_finish<T, T>), fgrid, dim3(1024), 0, st, part, p.S, n, (int)k,
                           alpha, beta, C, ldc, static_cast<T*>(Y), ldy);
    return launch_status("rk_matmat_skinny_t(finish)");
}

// ------------------------------------------------------------------------------------------
// SAP / ASkotch iterate update (rlaopt/solvers/sap.py:139-175) in two launches
// ------------------------------------------------------------------------------------------
// every row: W <- Y;  V <- beta V + (1 - beta) Y;  Y <- alpha V + (1 - alpha) W
template <typename T>
__global__ void __launch_bounds__(256)
sap_dense_kernel(int64_t total, T* __restrict__ W, T* __restrict__ V, T* __restrict__ Y, T beta, T alpha) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const T y = Y[e];
    const T v = beta * V[e] + (T(1) - beta) * y;
    W[e] = y;
    V[e] = v;
    Y[e] = alpha * v + (T(1) - alpha) * y;
}

// rows of the block (distinct): W -= eta d;  V -= (eta gamma) d;  Y <- alpha V + (1 - alpha) W
template <typename T>
__global__ void __launch_bounds__(256)
sap_block_kernel(int64_t nb, int k, const int64_t* __restrict__ blk, const T* __restrict__ D, int64_t ldd,
                 const T* __restrict__ eta_dev, T eta_host, T* __restrict__ W, T* __restrict__ V,
                 T* __restrict__ Y, int64_t ld, int accel, T gamma, T alpha) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nb * k) return;
    const int64_t p = e / k;
    const int c = (int)(e - p * k);
    const T eta = eta_dev ? eta_dev[0] : eta_host;
    const T d = D[p * ldd + c];
    const int64_t at = blk[p] * ld + c;
    const T w = W[at] - eta * d;
    W[at] = w;
    if (accel) {
        const T v = V[at] - (eta * gamma) * d;
        V[at] = v;
        Y[at] = alpha * v + (T(1) - alpha) * w;
    }
}

// ------------------------------------------------------------------------------------------
// power iteration for the SAP step size: lambda_max(Pinv (K + reg I)), device-resident
// ------------------------------------------------------------------------------------------
// t = M c for the r x r float64 apply factor of the Nystrom inverse (or its diagonal form), cast to T
// (16 lanes per row of M, columns strided over them: coalesced 128-byte runs, 64 rows per pass)
template <typename T>
__global__ void __launch_bounds__(1024)
pi_factor_kernel(const double* __restrict__ M, int mdiag, const double* __restrict__ c, int r,
                 T* __restrict__ t) {
    if (mdiag) {
        for (int i = threadIdx.x; i < r; i += 1024) t[i] = (T)(M[i] * c[i]);
        return;
    }
    const int l = threadIdx.x & 15;
    for (int i = threadIdx.x >> 4; i < r; i += 64) {  // (uniform trip count within a 16-lane row)
        const double* row = M + (int64_t)i * r;
        double s = 0.0;
        for (int j = l; j < r; j += 16) s += row[j] * c[j];
        s += dpp_fetch<0xB1, 0xF>(s);
        s += dpp_fetch<0x4E, 0xF>(s);
        s += dpp_fetch<0x141, 0xF>(s);
        s += dpp_fetch<0x140, 0xF>(s);
        if (l == 0) t[i] = (T)s;
    }
}

// the same for k right-hand sides: t (r x k, row-major, T) = M c,  c (r x k, row-major float64)
template <typename T>
__global__ void __launch_bounds__(256)
nys_factor_kernel(const double* __restrict__ M, int mdiag, const double* __restrict__ c, int r, int k,
                  T* __restrict__ t) {
    // 16 rows of M per workgroup (16 lanes per row): r = 512 spreads its 2 MB over 32 CUs
    if (mdiag) {
        for (int e = blockIdx.x * 256 + threadIdx.x; e < r * k; e += gridDim.x * 256) t[e] = (T)(M[e / k] * c[e]);
        return;
    }
    const int l = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= r) return;  // (whole 16-lane rows leave together)
    const double* row = M + (int64_t)i * r;
    for (int j = 0; j < k; ++j) {
        double s = 0.0;
        for (int q = l; q < r; q += 16) s += row[q] * c[(int64_t)q * k + j];
        s += dpp_fetch<0xB1, 0xF>(s);
        s += dpp_fetch<0x4E, 0xF>(s);
        s += dpp_fetch<0x141, 0xF>(s);
        s += dpp_fetch<0x140, 0xF>(s);
        if (l == 0) t[(int64_t)i * k + j] = (T)s;
    }
}

// block-wide sum of a double (1024 threads), result in every thread
__device__ __forceinline__ double pi_block_sum(double x, double* red) {
    x = wave_total_lane63(x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // (red may still be read from a previous call)
    if (lane == 63) red[wave] = x;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w];
    return s;
}

// v <- v / |v|, state <- {sig = 0, done = 0, iterations = 0}
template <typename T>
__global__ void __launch_bounds__(1024)
pi_start_kernel(T* __restrict__ v, int64_t m, double* __restrict__ state) {
    __shared__ double red[16];
    double ss = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 1024) ss += (double)v[i] * (double)v[i];
    const double nrm = sqrt(pi_block_sum(ss, red));
    const T inv = (T)(1.0 / nrm);
    for (int64_t i = threadIdx.x; i < m; i += 1024) v[i] *= inv;
    if (threadIdx.x == 0) { state[0] = 0.0; state[1] = 0.0; state[2] = 0.0; }
}

// One step of rlaopt/spectral_estimators/spectral_norm.py:22-29 with z = S v already formed:
//     sig_new = v . z;  v = z / |z|;  err = |sig_new - sig|;  sig = sig_new;  i += 1
// and the loop test "i < max_iters and err > rtol * sig" evaluated on the device: once it fails the
// state is frozen (later launches of this kernel leave sig and the count alone), so the value left
// in state[0] is the reference's return value although the host never looks at the iteration.
// Inner products accumulate in float64 and are rounded to the working type like torch.dot's result.
template <typename T>
__global__ void __launch_bounds__(1024)
pi_update_kernel(T* __restrict__ v, const T* __restrict__ z, int64_t m, double rtol,
                 double* __restrict__ state) {
    __shared__ double red[16];
    double dot = 0.0, ss = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 1024) {
        const double zi = (double)z[i];
        dot += (double)v[i] * zi;
        ss += zi * zi;
    }
    dot = pi_block_sum(dot, red);
    ss = pi_block_sum(ss, red);
    const T inv = (T)(1.0 / (double)(T)sqrt(ss));
    for (int64_t i = threadIdx.x; i < m; i += 1024) v[i] = z[i] * inv;
    if (threadIdx.x == 0 && state[1] == 0.0) {
        const double sig_new = (double)(T)dot;
        const double err = fabs(sig_new - state[0]);
        state[0] = sig_new;
        state[2] += 1.0;
        if (!(err > rtol * sig_new)) state[1] = 1.0;
    }
}

struct PIPlan {
    size_t off_y, off_z, off_t, off_c, off_nn, nn_bytes, off_tt, tt_bytes, total;
};

static PIPlan plan_pi(int dtype, int64_t m, int64_t r) {
    const size_t es = dtype == RK_F32 ? 4 : 8;
    PIPlan p;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += round_up((int64_t)bytes, 256); return at; };
    p.off_y = take(m * es);
    p.off_z = take(m * es);
    p.off_t = take((r > 0 ? r : 1) * es);
    p.off_c = take((r > 0 ? r : 1) * sizeof(double));
    const NNPlan a = plan_nn(dtype, m, m, 1);
    size_t nn = a.S > 1 ? (size_t)a.S * m * es : 0;
    if (r > 0) {
        const NNPlan b = plan_nn(dtype, m, r, 1);
        const size_t nb = b.S > 1 ? (size_t)b.S * m * es : 0;
        if (nb > nn) nn = nb;
    }
    p.nn_bytes = nn;
    p.off_nn = take(nn ? nn : 1);
    p.tt_bytes = r > 0 ? (size_t)plan_t(dtype, m, r).S * r * es : 0;
    p.off_tt = take(p.tt_bytes ? p.tt_bytes : 1);
    p.total = o;
    return p;
}

template <typename T>
static int power_iteration_impl(int dtype, const T* K, int64_t m, int64_t ldk, double reg, const T* U,
                                int64_t r, int64_t ldu, const double* M, int mdiag, double rho, T* v,
                                int64_t max_iters, double rtol, double* state, void* ws, hipStream_t st) {
    const PIPlan p = plan_pi(dtype, m, r);
    char* base = static_cast<char*>(ws);
    T* y = reinterpret_cast<T*>(base + p.off_y);
    T* z = reinterpret_cast<T*>(base + p.off_z);
    T* t = reinterpret_cast<T*>(base + p.off_t);
    double* c = reinterpret_cast<double*>(base + p.off_c);
    void* nn_ws = p.nn_bytes ? base + p.off_nn : nullptr;
    void* tt_ws = base + p.off_tt;
    hipLaunchKernelGGL((pi_start_kernel<T>), dim3(1), dim3(1024), 0, st, v, m, state);
    int rc = launch_status("rk_power_iteration(start)");
    const double inv_rho = 1.0 / rho;
    for (int64_t it = 0; it < max_iters && rc == RK_OK; ++it) {
        // y = K v + reg v
        rc = matmat_skinny_impl<T>(K, m, m, ldk, v, 1, 1, 1.0, 0.0, nullptr, 1, reg, 0, y, 1, nn_ws,
                                   p.nn_bytes, st, dtype);
        if (rc != RK_OK) break;
        const T* zin = y;
        if (U != nullptr) {
            // z = (y - U (M (U^T y))) / rho
            rc = matmat_skinny_t_impl<T>(U, m, r, ldu, y, 1, 1, 1.0, 0.0, nullptr, 1, c, 1, 1, tt_ws,
                                         p.tt_bytes, st, dtype);
            if (rc != RK_OK) break;
            hipLaunchKernelGGL((pi_factor_kernel<T>), dim3(1), dim3(1024), 0, st, M, mdiag, c, (int)r, t);
            rc = matmat_skinny_impl<T>(U, m, r, ldu, t, 1, 1, -inv_rho, inv_rho, y, 1, 0.0, 0, z, 1, nn_ws,
                                       p.nn_bytes, st, dtype);
            if (rc != RK_OK) break;
            zin = z;
        } else if (rho != 1.0) {
            set_error("rk_power_iteration: without U the preconditioner is the identity (rho must be 1)");
            return RK_ERR_INVALID;
        }
        hipLaunchKernelGGL((pi_update_kernel<T>), dim3(1), dim3(1024), 0, st, v, zin, m, rtol, state);
        rc = launch_status("rk_power_iteration(update)");
    }
    return rc;
}

struct NysPlan {
    size_t off_c, off_t, off_nn, nn_bytes, off_tt, tt_bytes, total;
};
static NysPlan plan_nys(int dtype, int64_t n, int64_t r, int64_t k) {
    const size_t es = dtype == RK_F32 ? 4 : 8;
    NysPlan p;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += round_up((int64_t)(bytes ? bytes : 1), 256); return at; };
    p.off_c = take((size_t)r * k * sizeof(double));
    p.off_t = take((size_t)r * k * es);
    const NNPlan a = plan_nn(dtype, n, r, k);
    p.nn_bytes = a.S > 1 ? (size_t)a.S * n * k * es : 0;
    p.off_nn = take(p.nn_bytes);
    p.tt_bytes = (size_t)plan_t(dtype, n, r).S * r * k * es;
    p.off_tt = take(p.tt_bytes);
    p.total = o;
    return p;
}

template <typename T>
static int nystrom_inv_apply_impl(int dtype, const T* U, int64_t n, int64_t r, int64_t ldu, const double* M, int mdiag,
                                  double rho, const T* X, int64_t k, int64_t ldx, T* Y, int64_t ldy, void* ws,
                                  hipStream_t st) {
    const NysPlan p = plan_nys(dtype, n, r, k);
    char* base = static_cast<char*>(ws);
    double* c = reinterpret_cast<double*>(base + p.off_c);
    T* t = reinterpret_cast<T*>(base + p.off_t);
    // c = U^T X (float64), t = M c, Y = (X - U t) / rho
    int rc = matmat_skinny_t_impl<T>(U, n, r, ldu, X, k, ldx, 1.0, 0.0, nullptr, k, c, k, 1, base + p.off_tt, p.tt_bytes,
                                     st, dtype);
    if (rc != RK_OK) return rc;
    hipLaunchKernelGGL((nys_factor_kernel<T>), dim3((unsigned)ceil_div(r, 16)), dim3(256), 0, st, M, mdiag, c, (int)r, (int)k, t);
    rc = launch_status("rk_nystrom_inv_apply(factor)");
    if (rc != RK_OK) return rc;
    const double inv_rho = 1.0 / rho;
    return matmat_skinny_impl<T>(U, n, r, ldu, t, k, k, -inv_rho, inv_rho, X, ldx, 0.0, 0, Y, ldy,
                                 p.nn_bytes ? base + p.off_nn : nullptr, p.nn_bytes, st, dtype);
}

}  // namespace rk

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace rk;

extern "C" int rk_sap_update(int dtype, int64_t n, int64_t k, void* W, void* V, void* Y, int64_t ld,
                             const int64_t* blk, int64_t nb, const void* D, int64_t ldd,
                             const void* eta_dev, double eta_host, int accel, double beta, double gamma,
                             double alpha, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_sap_update: bad dtype %d", dtype);
    RK_REQUIRE(W && blk && D, "rk_sap_update: null pointer");
    RK_REQUIRE(!accel || (V && Y), "rk_sap_update: the accelerated update needs V and Y");
    RK_REQUIRE(n > 0 && k >= 1 && nb >= 1 && nb <= n && ldd >= k, "rk_sap_update: bad sizes");
    RK_REQUIRE(ld == k, "rk_sap_update: iterates must be contiguous (ld == k)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 gd((unsigned)ceil_div(n * k, 256)), gb((unsigned)ceil_div(nb * k, 256));
    if (dtype == RK_F32) {
        if (accel)
            hipLaunchKernelGGL((sap_dense_kernel<float>), gd, dim3(256), 0, st, n * k, (float*)W, (float*)V,
                               (float*)Y, (float)beta, (float)alpha);
        hipLaunchKernelGGL((sap_block_kernel<float>), gb, dim3(256), 0, st, nb, (int)k, blk, (const float*)D,
                           ldd, (const float*)eta_dev, (float)eta_host, (float*)W, (float*)V, (float*)Y, ld,
                           accel, (float)gamma, (float)alpha);
    } else {
        if (accel)
            hipLaunchKernelGGL((sap_dense_kernel<double>), gd, dim3(256), 0, st, n * k, (double*)W, (double*)V,
                               (double*)Y, beta, alpha);
        hipLaunchKernelGGL((sap_block_kernel<double>), gb, dim3(256), 0, st, nb, (int)k, blk, (const double*)D,
                           ldd, (const double*)eta_dev, eta_host, (double*)W, (double*)V, (double*)Y, ld, accel,
                           gamma, alpha);
    }
    return launch_status("rk_sap_update");
}

extern "C" size_t rk_power_iteration_ws_bytes(int dtype, int64_t m, int64_t r) {
    if (m <= 0 || r < 0) return 0;
    return plan_pi(dtype, m, r).total;
}

extern "C" int rk_power_iteration(int dtype, const void* K, int64_t m, int64_t ldk, double reg,
                                  const void* U, int64_t r, int64_t ldu, const double* M, int mdiag,
                                  double rho, void* v, int64_t max_iters, double rtol, double* state,
                                  void* ws, size_t ws_bytes, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_power_iteration: bad dtype %d", dtype);
    RK_REQUIRE(K && v && state, "rk_power_iteration: null pointer");
    RK_REQUIRE(m > 0 && ldk >= m && max_iters >= 0 && max_iters <= 1000, "rk_power_iteration: bad sizes");
    RK_REQUIRE(U == nullptr || (r > 0 && r <= 4096 && ldu >= r && M != nullptr && rho > 0.0),
               "rk_power_iteration: bad preconditioner factors");
    const int64_t rr = U ? r : 0;
    const size_t need = plan_pi(dtype, m, rr).total;
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_power_iteration: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        return power_iteration_impl<float>(dtype, (const float*)K, m, ldk, reg, (const float*)U, rr, ldu, M,
                                           mdiag, rho, (float*)v, max_iters, rtol, state, ws, st);
    return power_iteration_impl<double>(dtype, (const double*)K, m, ldk, reg, (const double*)U, rr, ldu, M,
                                        mdiag, rho, (double*)v, max_iters, rtol, state, ws, st);
}

extern "C" size_t rk_matmat_skinny_ws_bytes(int dtype, int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0 || k > 16) return 0;
    const NNPlan p = plan_nn(dtype, m, n, k);
    const size_t es = dtype == RK_F32 ? 4 : 8;
    size_t bytes = p.S > 1 ? (size_t)p.S * m * k * es : 0;
    if (k >= (dtype == RK_F32 ? mm_min_k<float>() : mm_min_k<double>())) {  // the matrix-core kernel: re-packed V + partial sums
        const size_t b = dtype == RK_F32 ? mm_ws_bytes<float>(m, n) : mm_ws_bytes<double>(m, n);
        if (b > bytes) bytes = b;
    }
    return bytes;
}

extern "C" int rk_matmat_skinny(int dtype, const void* A, int64_t m, int64_t n, int64_t lda,
                                const void* V, int64_t k, int64_t ldv, double alpha, double beta,
                                const void* C, int64_t ldc, double diag_shift, int64_t row0,
                                void* Y, int64_t ldy, void* ws, size_t ws_bytes, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_matmat_skinny: bad dtype %d", dtype);
    RK_REQUIRE(A && V && Y, "rk_matmat_skinny: null pointer");
    RK_REQUIRE(m > 0 && n > 0, "rk_matmat_skinny: empty operand (m=%lld n=%lld)", (long long)m,
               (long long)n);
    RK_REQUIRE(k >= 1 && k <= 16, "rk_matmat_skinny: k=%lld outside [1,16]", (long long)k);
    RK_REQUIRE(lda >= n && ldv >= k && ldy >= k, "rk_matmat_skinny: leading dimension too small");
    RK_REQUIRE(beta == 0.0 || (C != nullptr && ldc >= k), "rk_matmat_skinny: beta != 0 needs C");
    RK_REQUIRE(row0 >= 0, "rk_matmat_skinny: negative row0");
    RK_REQUIRE(diag_shift == 0.0 || row0 + m <= n,
               "rk_matmat_skinny: diag_shift needs rows [row0, row0+m) inside V (n=%lld)",
               (long long)n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        return matmat_skinny_impl<float>((const float*)A, m, n, lda, (const float*)V, k, ldv, alpha,
                                         beta, (const float*)C, ldc, diag_shift, row0, (float*)Y,
                                         ldy, ws, ws_bytes, st, dtype);
    return matmat_skinny_impl<double>((const double*)A, m, n, lda, (const double*)V, k, ldv, alpha,
                                      beta, (const double*)C, ldc, diag_shift, row0, (double*)Y, ldy,
                                      ws, ws_bytes, st, dtype);
}

extern "C" size_t rk_matmat_skinny_t_ws_bytes(int dtype, int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0 || k > 16) return 0;
    const TPlan p = plan_t(dtype, m, n);
    const size_t es = dtype == RK_F32 ? 4 : 8;
    return (size_t)p.S * n * k * es;
}

extern "C" int rk_matmat_skinny_t(int dtype, const void* A, int64_t m, int64_t n, int64_t lda,
                                  const void* V, int64_t k, int64_t ldv, double alpha, double beta,
                                  const void* C, int64_t ldc, void* Y, int64_t ldy, int out_f64,
                                  void* ws, size_t ws_bytes, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_matmat_skinny_t: bad dtype %d", dtype);
    RK_REQUIRE(A && V && Y, "rk_matmat_skinny_t: null pointer");
    RK_REQUIRE(m > 0 && n > 0, "rk_matmat_skinny_t: empty operand");
    RK_REQUIRE(k >= 1 && k <= 16, "rk_matmat_skinny_t: k=%lld outside [1,16]", (long long)k);
    RK_REQUIRE(lda >= n && ldv >= k && ldy >= k, "rk_matmat_skinny_t: leading dimension too small");
    RK_REQUIRE(beta == 0.0 || (C != nullptr && ldc >= k), "rk_matmat_skinny_t: beta != 0 needs C");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        return matmat_skinny_t_impl<float>((const float*)A, m, n, lda, (const float*)V, k, ldv,
                                           alpha, beta, (const float*)C, ldc, Y, ldy, out_f64, ws,
                                           ws_bytes, st, dtype);
    return matmat_skinny_t_impl<double>((const double*)A, m, n, lda, (const double*)V, k, ldv, alpha,
                                        beta, (const double*)C, ldc, Y, ldy, out_f64, ws, ws_bytes,
                                        st, dtype);
}

extern "C" int rk_pcg_update_wr(int dtype, int64_t n, int64_t k, void* W, int64_t ldw, void* R,
                                int64_t ldr, const void* P, int64_t ldp, const void* AP,
                                int64_t ldap, const double* alpha, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_pcg_update_wr: bad dtype %d", dtype);
    RK_REQUIRE(W && R && P && AP && alpha, "rk_pcg_update_wr: null pointer");
    RK_REQUIRE(n > 0 && k >= 1 && k <= 16, "rk_pcg_update_wr: bad sizes n=%lld k=%lld",
               (long long)n, (long long)k);
    RK_REQUIRE(ldw >= k && ldr >= k && ldp >= k && ldap >= k, "rk_pcg_update_wr: bad ld");
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)ceil_div(n * k, 256));
    if (dtype == RK_F32)
        hipLaunchKernelGGL((pcg_update_wr_kernel<float>), grid, dim3(256), 0, st, n, (int)k,
                           (float*)W, ldw, (float*)R, ldr, (const float*)P, ldp, (const float*)AP,
                           ldap, alpha);
    else
        hipLaunchKernelGGL((pcg_update_wr_kernel<double>), grid, dim3(256), 0, st, n, (int)k,
                           (double*)W, ldw, (double*)R, ldr, (const double*)P, ldp,
                           (const double*)AP, ldap, alpha);
    return launch_status("rk_pcg_update_wr");
}

extern "C" int rk_pcg_update_p(int dtype, int64_t n, int64_t k, void* P, int64_t ldp, const void* Z,
                               int64_t ldz, const double* beta, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_pcg_update_p: bad dtype %d", dtype);
    RK_REQUIRE(P && Z && beta, "rk_pcg_update_p: null pointer");
    RK_REQUIRE(n > 0 && k >= 1 && k <= 16, "rk_pcg_update_p: bad sizes");
    RK_REQUIRE(ldp >= k && ldz >= k, "rk_pcg_update_p: bad ld");
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)ceil_div(n, 256));
    if (dtype == RK_F32)
        hipLaunchKernelGGL((pcg_update_p_kernel<float>), grid, dim3(256), 0, st, n, (int)k,
                           (float*)P, ldp, (const float*)Z, ldz, beta);
    else
        hipLaunchKernelGGL((pcg_update_p_kernel<double>), grid, dim3(256), 0, st, n, (int)k,
                           (double*)P, ldp, (const double*)Z, ldz, beta);
    return launch_status("rk_pcg_update_p");
}

static int gram_small_blocks(int64_t n) {
    int64_t b = ceil_div(n, 256 * 8);  // >= 8 rows per thread
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" size_t rk_gram_small_ws_bytes(int64_t n, int64_t k1, int64_t k2) {
    if (n <= 0 || k1 < 1 || k2 < 1 || k1 > 4 || k2 > 4) return 0;
    return (size_t)gram_small_blocks(n) * k1 * k2 * sizeof(double);
}

template <typename T>
static int gram_small_dispatch(const T* X, int64_t ldx, const T* Y, int64_t ldy, int64_t n, int k1,
                               int k2, double* part, int nb, hipStream_t st) {
#define RK_GS(A, B)                                                                              \
    case (A * 8 + B):                                                                            \
        hipLaunchKernelGGL((gram_small_kernel<T, A, B>), dim3(nb), dim3(256), 0, st, X, ldx, Y,  \
                           ldy, n, part);                                                        \
        break;
    switch (k1 * 8 + k2) {
        RK_GS(1, 1) RK_GS(1, 2) RK_GS(1, 3) RK_GS(1, 4) RK_GS(2, 1) RK_GS(2, 2) RK_GS(2, 3)
        RK_GS(2, 4) RK_GS(3, 1) RK_GS(3, 2) RK_GS(3, 3) RK_GS(3, 4) RK_GS(4, 1) RK_GS(4, 2)
        RK_GS(4, 3) RK_GS(4, 4)
        default: return RK_ERR_INVALID;
    }
#undef RK_GS
    return launch_status("rk_gram_small");
}

extern "C" int rk_gram_small(int dtype, const void* X, int64_t ldx, const void* Y, int64_t ldy,
                             int64_t n, int64_t k1, int64_t k2, double* G, void* ws,
                             size_t ws_bytes, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_gram_small: bad dtype %d", dtype);
    RK_REQUIRE(X && Y && G, "rk_gram_small: null pointer");
    RK_REQUIRE(n > 0 && k1 >= 1 && k1 <= 4 && k2 >= 1 && k2 <= 4, "rk_gram_small: sizes outside [1,4]");
    RK_REQUIRE(ldx >= k1 && ldy >= k2, "rk_gram_small: leading dimension too small");
    const int nb = gram_small_blocks(n);
    const size_t need = (size_t)nb * k1 * k2 * sizeof(double);
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_gram_small: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(ws);
    int rc = dtype == RK_F32
                 ? gram_small_dispatch<float>((const float*)X, ldx, (const float*)Y, ldy, n, (int)k1,
                                              (int)k2, part, nb, st)
                 : gram_small_dispatch<double>((const double*)X, ldx, (const double*)Y, ldy, n,
                                               (int)k1, (int)k2, part, nb, st);
    if (rc != RK_OK) return rc;
    hipLaunchKernelGGL(gram_small_finish, dim3((unsigned)(k1 * k2)), dim3(64), 0, st, part, nb,
                       (int)(k1 * k2), G);
    return launch_status("rk_gram_small(finish)");
}

extern "C" int rk_sparse_sign_apply(int dtype, const void* A, int64_t m, int64_t d, int64_t lda,
                                    const int* col, const signed char* sgn, const int64_t* row_ptr,
                                    int64_t s, double scale, void* Y, int64_t ldy, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_sparse_sign_apply: bad dtype %d", dtype);
    RK_REQUIRE(A && col && sgn && row_ptr && Y, "rk_sparse_sign_apply: null pointer");
    RK_REQUIRE(m > 0 && d > 0 && s > 0 && s <= 65535, "rk_sparse_sign_apply: bad sizes (s <= 65535)");
    RK_REQUIRE(lda >= d && ldy >= d, "rk_sparse_sign_apply: leading dimension too small");
    RK_REQUIRE(aligned16(A), "rk_sparse_sign_apply: A must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vw = dtype == RK_F32 ? 4 : 2;
    dim3 grid((unsigned)ceil_div(d, (int64_t)kWave * vw), (unsigned)s);
    if (dtype == RK_F32)
        hipLaunchKernelGGL((sparse_sign_apply_kernel<float>), grid, dim3(256), 0, st, (const float*)A,
                           d, lda, col, sgn, row_ptr, (float)scale, (float*)Y, ldy);
    else
        hipLaunchKernelGGL((sparse_sign_apply_kernel<double>), grid, dim3(256), 0, st,
                           (const double*)A, d, lda, col, sgn, row_ptr, scale, (double*)Y, ldy);
    return launch_status("rk_sparse_sign_apply");
}

extern "C" int rk_axpby_cols(int dtype, int64_t n, int64_t k, const void* X, int64_t ldx,
                             const double* a, void* Y, int64_t ldy, const double* b, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_axpby_cols: bad dtype %d", dtype);
    RK_REQUIRE(X && Y, "rk_axpby_cols: null pointer");
    RK_REQUIRE(n > 0 && k >= 1 && ldx >= k && ldy >= k, "rk_axpby_cols: bad sizes");
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)ceil_div(n * k, 256));
    if (dtype == RK_F32)
        hipLaunchKernelGGL((axpby_cols_kernel<float>), grid, dim3(256), 0, st, n, (int)k,
                           (const float*)X, ldx, a, (float*)Y, ldy, b);
    else
        hipLaunchKernelGGL((axpby_cols_kernel<double>), grid, dim3(256), 0, st, n, (int)k,
                           (const double*)X, ldx, a, (double*)Y, ldy, b);
    return launch_status("rk_axpby_cols");
}

extern "C" int rk_small_solve(int64_t k, int64_t nrhs, const double* A, const double* B, double* X,
                              void* stream) {
    clear_error();
    RK_REQUIRE(A && B && X, "rk_small_solve: null pointer");
    RK_REQUIRE(k >= 1 && k <= 16 && nrhs >= 1 && nrhs <= 16, "rk_small_solve: sizes outside [1,16]");
    hipLaunchKernelGGL(small_solve_kernel, dim3(1), dim3(512), 0, static_cast<hipStream_t>(stream),
                       (int)k, (int)nrhs, A, B, X);
    return launch_status("rk_small_solve");
}

// ------------------------------------------------------------------------------------------
// the n-space half of an LSQR step in one launch
// ------------------------------------------------------------------------------------------
// Per column c (one workgroup each; the columns of a block solve are independent recurrences):
//     v <- g - beta v;  alpha = |v|;  v <- v / alpha                    (Golub-Kahan, step 3(b))
//     rho = hypot(rhobar, beta); cs = rhobar / rho; sn = beta / rho; theta = sn alpha
//     rhobar <- -cs alpha;  phi = cs phibar;  phibar <- sn phibar       (plane rotation, step 4)
//     y <- y + (phi / rho) w;  w <- v - (theta / rho) w                 (step 5)
// The vectors have n entries (the columns of A: 512 in BASELINE config 4), so this is pure launch
// latency: the same arithmetic as ~20 element-wise / reduction launches of 4.5 us each.
template <typename T>
__global__ void __launch_bounds__(1024)
lsqr_tail_kernel(int64_t n, int k, const T* __restrict__ G, int64_t ldg, T* __restrict__ V, int64_t ldv,
                 T* __restrict__ W, int64_t ldw, T* __restrict__ Y, int64_t ldy, const double* __restrict__ beta,
                 double* __restrict__ rhobar, double* __restrict__ phibar, double* __restrict__ alpha) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    const double b = beta[c];
    double ss = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const T x = (T)((double)G[i * ldg + c] - b * (double)V[i * ldv + c]);
        V[i * ldv + c] = x;  // (rounded to the working dtype before the norm, like the separate launches)
        ss += (double)x * (double)x;
    }
    const double tot = pi_block_sum(ss, red);  // (fixed order; the same value in every thread)
    const double a = sqrt(tot > 0.0 ? tot : 0.0);
    const double rb = rhobar[c], pb = phibar[c];
    const double rho = sqrt(rb * rb + b * b);
    const double cs = rb / rho, sn = b / rho;
    const double theta = sn * a, phi = cs * pb;
    const double cy = phi / rho, cw = -(theta / rho), inva = 1.0 / a;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const T v = (T)((double)V[i * ldv + c] * inva);
        const T w = W[i * ldw + c];
        V[i * ldv + c] = v;
        Y[i * ldy + c] = (T)((double)Y[i * ldy + c] + cy * (double)w);
        W[i * ldw + c] = (T)((double)v + cw * (double)w);
    }
    if (threadIdx.x == 0) {
        alpha[c] = a;
        rhobar[c] = -cs * a;
        phibar[c] = sn * pb;
    }
}

extern "C" int rk_lsqr_tail(int dtype, int64_t n, int64_t k, const void* G, int64_t ldg, void* V, int64_t ldv,
                            void* W, int64_t ldw, void* Y, int64_t ldy, const double* beta, double* rhobar,
                            double* phibar, double* alpha, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_lsqr_tail: bad dtype %d", dtype);
    RK_REQUIRE(G && V && W && Y && beta && rhobar && phibar && alpha, "rk_lsqr_tail: null pointer");
    RK_REQUIRE(n >= 1 && k >= 1 && k <= 65535 && ldg >= k && ldv >= k && ldw >= k && ldy >= k, "rk_lsqr_tail: bad shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        hipLaunchKernelGGL((lsqr_tail_kernel<float>), dim3((unsigned)k), dim3(1024), 0, st, n, (int)k, (const float*)G, ldg,
                           (float*)V, ldv, (float*)W, ldw, (float*)Y, ldy, beta, rhobar, phibar, alpha);
    else
        hipLaunchKernelGGL((lsqr_tail_kernel<double>), dim3((unsigned)k), dim3(1024), 0, st, n, (int)k, (const double*)G,
                           ldg, (double*)V, ldv, (double*)W, ldw, (double*)Y, ldy, beta, rhobar, phibar, alpha);
    return launch_status("rk_lsqr_tail");
}

// ------------------------------------------------------------------------------------------
// k distinct uniform draws from [0, n): the head of a pseudo-random permutation
// ------------------------------------------------------------------------------------------
// out[i] = pi(i), pi a keyed bijection of [0, n): an alternating Feistel network on the b = ceil(log2 n)
// bits (left half b / 2 bits, right half the rest; round r XORs one half with a hash of the other - each
// round is invertible whatever the hash), walked until the image falls below n (a bijection of [0, 2^b)
// restricted this way is a bijection of [0, n); 2^b < 2 n: fewer than two evaluations on average, and
// the walk ends because the orbit of i returns to i < n).  One launch of k threads replaces the
// sort-based `randperm(n)[:k]` (27 kernels, 0.19 - 0.35 ms at n = 500 000) in SAP's block draw.
__device__ __forceinline__ uint64_t sd_mix(uint64_t x) {  // murmur3's 64-bit finaliser
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
constexpr int SD_ROUNDS = 8;

__global__ void __launch_bounds__(256)
sample_distinct_kernel(int64_t n, int64_t k, int bits, uint64_t seed, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= k) return;
    const int cb = bits - bits / 2;  // right half (the larger one when b is odd)
    const uint64_t mask_l = (1ull << (bits / 2)) - 1, mask_r = (1ull << cb) - 1;
    uint64_t x = (uint64_t)i;
    do {
        uint64_t l = x >> cb, r = x & mask_r;
#pragma unroll
        for (int q = 0; q < SD_ROUNDS; ++q) {
            const uint64_t key = seed + 0x9e3779b97f4a7c15ull * (uint64_t)(q + 1);
            if (q & 1)
                r = (r ^ sd_mix(l + key)) & mask_r;
            else
                l = (l ^ sd_mix(r + key)) & mask_l;
        }
        x = (l << cb) | r;
    } while (x >= (uint64_t)n);
    out[i] = (int64_t)x;
}

extern "C" int rk_sample_distinct(int64_t n, int64_t k, uint64_t seed, int64_t* out, void* stream) {
    clear_error();
    RK_REQUIRE(out != nullptr, "rk_sample_distinct: null pointer");
    RK_REQUIRE(n >= 1 && k >= 1 && k <= n, "rk_sample_distinct: need 1 <= k <= n (k=%lld, n=%lld)", (long long)k,
               (long long)n);
    RK_REQUIRE(n <= (1ll << 62), "rk_sample_distinct: n too large");
    int bits = 2;  // (at least one bit per half)
    while ((1ll << bits) < n) ++bits;
    hipLaunchKernelGGL(sample_distinct_kernel, dim3((unsigned)ceil_div(k, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), n, k, bits, seed, out);
    return launch_status("rk_sample_distinct");
}

extern "C" int rk_randn(int dtype, void* out, int64_t rows, int64_t cols, int64_t ld, uint64_t seed,
                        uint64_t offset, double scale, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_randn: bad dtype %d", dtype);
    RK_REQUIRE(out != nullptr, "rk_randn: null pointer");
    RK_REQUIRE(rows > 0 && cols > 0 && ld >= cols, "rk_randn: bad shape");
    const int64_t groups = ceil_div(rows * cols, 4);
    dim3 grid((unsigned)ceil_div(groups, 256));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        hipLaunchKernelGGL((randn_kernel<float>), grid, dim3(256), 0, st, (float*)out, rows, cols, ld,
                           seed, offset, scale);
    else
        hipLaunchKernelGGL((randn_kernel<double>), grid, dim3(256), 0, st, (double*)out, rows, cols,
                           ld, seed, offset, scale);
    return launch_status("rk_randn");
}

extern "C" size_t rk_nystrom_inv_apply_ws_bytes(int dtype, int64_t n, int64_t r, int64_t k) {
    if (n <= 0 || r <= 0 || k <= 0 || k > 16 || r > 4096) return 0;
    return plan_nys(dtype, n, r, k).total;
}

extern "C" int rk_nystrom_inv_apply(int dtype, const void* U, int64_t n, int64_t r, int64_t ldu, const double* M,
                                    int mdiag, double rho, const void* X, int64_t k, int64_t ldx, void* Y, int64_t ldy,
                                    void* ws, size_t ws_bytes, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_nystrom_inv_apply: bad dtype %d", dtype);
    RK_REQUIRE(U && M && X && Y, "rk_nystrom_inv_apply: null pointer");
    RK_REQUIRE(n > 0 && r > 0 && r <= 4096 && k >= 1 && k <= 16 && ldu >= r && ldx >= k && ldy >= k && rho > 0.0,
               "rk_nystrom_inv_apply: bad sizes (n=%lld r=%lld k=%lld rho=%g)", (long long)n, (long long)r, (long long)k, rho);
    const size_t need = plan_nys(dtype, n, r, k).total;
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_nystrom_inv_apply: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        return nystrom_inv_apply_impl<float>(dtype, (const float*)U, n, r, ldu, M, mdiag, rho, (const float*)X, k, ldx,
                                             (float*)Y, ldy, ws, st);
    return nystrom_inv_apply_impl<double>(dtype, (const double*)U, n, r, ldu, M, mdiag, rho, (const double*)X, k, ldx,
                                          (double*)Y, ldy, ws, st);
}
