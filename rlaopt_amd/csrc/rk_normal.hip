// One pass over a tall matrix for both products of a Golub-Kahan (LSQR) step.
//
//     U <- A Z - U diag(coef)         (m x k, in place)
//     G  = A^T U_new                  (n x k, float64)
//     ss = column sums of U_new^2     (k,     float64)
//
// LSQR needs A z and then A^T u with u = (A z - alpha u_old) / beta; beta is a norm that is only
// known after the first product, which is why implementations read A twice per iteration.  But
// A^T u = (A^T u_unnormalised) / beta, and u_unnormalised[i] depends on row i alone - so while a
// wave still holds row i in registers it can finish the row's dot product, form u[i], and add
// u[i] * a_i to its share of G.  A is streamed ONCE per iteration: algorithmic bytes m n s instead
// of 2 m n s (BASELINE config 4: 8.2 GB instead of 16.4 GB).
//
// Layout: a wave owns a contiguous run of rows and all n columns of them, lane l holding columns
// (step * 64 + l) * VW + e for step < NS: a row of n <= 64 * VW * NS elements sits in NS 16-byte
// registers per lane.  R = 4 or 8 rows are in flight per stage, two stages (non-temporal loads issued
// one group ahead).  Per row: K dot products (4 v_add_dpp + 2 row_bcast each, total read back with
// v_readlane), u[i] stored by one lane, then NS * VW * K FMAs per lane into the G accumulators,
// which stay in registers for the whole run; the 4 waves of a workgroup add theirs through LDS and
// write one float64 (workgroup, column) table; a finishing kernel adds the tables in fixed order
// (bitwise reproducible).
#include "rk_common.h"

namespace rk {

// rows per register stage: 8 while a row is one or two 16-byte steps per lane, 4 for four steps
// (measured within 5 % of the best of {2, 4, 8} for every shape tried; 8 x 4 steps would not fit)
constexpr int np_rows(int ns) { return ns >= 4 ? 4 : 8; }
constexpr int NP_ROW_ALIGN = 8;

template <typename T, int K, int NS>
__global__ void __launch_bounds__(256)
normal_pass_kernel(const T* __restrict__ A, int64_t m, int64_t n, int64_t lda,
                   const T* __restrict__ Z, int64_t ldz, T* __restrict__ U, int64_t ldu,
                   const double* __restrict__ coef, double* __restrict__ part_g,
                   double* __restrict__ part_ss, int64_t rows_per_wave, int64_t nwaves) {
    using V16 = typename Vec16<T>::type;
    constexpr int VW = Vec16<T>::N;
    constexpr int R = np_rows(NS);
    const int lane = threadIdx.x & 63;
    const int64_t wid = uniform64((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    // (a wave past the last run of rows has an empty range; it still joins the barrier below)
    const int64_t i_lo = wid * rows_per_wave < m ? wid * rows_per_wave : m;
    const int64_t i_hi = (i_lo + rows_per_wave < m) ? i_lo + rows_per_wave : m;

    // this lane's columns, its slice of Z, and its G accumulators
    bool col_ok[NS];
    T z[NS][VW][K], g[NS][VW][K];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int64_t j0 = ((int64_t)s * 64 + lane) * VW;
        col_ok[s] = j0 + VW <= n;  // (n is a multiple of VW on this path)
#pragma unroll
        for (int e = 0; e < VW; ++e)
#pragma unroll
            for (int c = 0; c < K; ++c) {
                z[s][e][c] = col_ok[s] ? Z[(j0 + e) * ldz + c] : T(0);
                g[s][e][c] = T(0);
            }
    }
    T cf[K];
#pragma unroll
    for (int c = 0; c < K; ++c) cf[c] = (T)coef[c];
    double ss[K];
#pragma unroll
    for (int c = 0; c < K; ++c) ss[c] = 0.0;

    V16 stA[R][NS], stB[R][NS];
    T uA, uB;  // lane r*K + c holds U[i0 + r][c] of the stage's group
    const unsigned lane_off = (unsigned)lane * 16u;
    auto fetch = [&](V16 a[R][NS], T& uo, int64_t i0) {
        {
            const int e = lane < R * K ? lane : R * K - 1;
            int64_t i = i0 + e / K;
            if (i > m - 1) i = m - 1;
            uo = U[i * ldu + e % K];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int64_t i = i0 + r;
            if (i > m - 1) i = m - 1;  // clamp; masked below
            // (unconditional: a lane whose columns lie past n re-reads the first 16 bytes of its column
            // chunk and is masked when the data is used; a chunk that starts at or past the end of the
            // row - three steps of columns run the NS = 4 instantiation - falls back to chunk 0, so no
            // load ever leaves the row; a conditional load makes the compiler drain the whole prefetch
            // queue at the branch join, once per group of rows)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t c0 = (int64_t)s * 64 * VW;
                a[r][s] = load_stream_at<V16>(A + i * lda + (c0 < n ? c0 : 0), col_ok[s] ? lane_off : 0u);
            }
        }
    };
    auto consume = [&](const V16 a[R][NS], T uo, int64_t i0) {
        T ubuf = T(0);  // lane r*K + c collects U_new[i0 + r][c]: one coalesced store per group
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool live = i0 + r < i_hi;
            T ae[NS][VW];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const T* p = reinterpret_cast<const T*>(&a[r][s]);
#pragma unroll
                for (int e = 0; e < VW; ++e) ae[s][e] = col_ok[s] ? p[e] : T(0);
            }
            T unew[K];
#pragma unroll
            for (int c = 0; c < K; ++c) {
                T dot = T(0);
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int e = 0; e < VW; ++e) dot += ae[s][e] * z[s][e][c];
                const T tot = lane_value(wave_total_lane63(dot), 63);
                unew[c] = tot - cf[c] * lane_value(uo, r * K + c);
            }
#pragma unroll
            for (int c = 0; c < K; ++c)
                if (lane == r * K + c) ubuf = unew[c];
            if (live) {
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    ss[c] += (double)unew[c] * (double)unew[c];
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int e = 0; e < VW; ++e) g[s][e][c] += ae[s][e] * unew[c];
                }
            }
        }
        // (a plain store: L2 merges the 16 - 64 byte pieces of neighbouring groups)
        if (lane < R * K && i0 + lane / K < i_hi) U[(i0 + lane / K) * ldu + lane % K] = ubuf;
    };
    if (i_lo < i_hi) {
        int64_t i0 = i_lo;
        fetch(stA, uA, i0);
        for (;;) {
            if (i0 + R >= i_hi) { consume(stA, uA, i0); break; }
            fetch(stB, uB, i0 + R);
            consume(stA, uA, i0);
            i0 += R;
            if (i0 + R >= i_hi) { consume(stB, uB, i0); break; }
            fetch(stA, uA, i0 + R);
            consume(stB, uB, i0);
            i0 += R;
        }
    }
    // the workgroup's 4 waves add their tables in LDS (fixed order) and write one: the finishing
    // kernel then reads a quarter of the partials
    __shared__ double lg[4][64 * VW * NS * K];
    __shared__ double lss[4][K];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int e = 0; e < VW; ++e)
#pragma unroll
            for (int c = 0; c < K; ++c) lg[wv][((s * 64 + lane) * VW + e) * K + c] = (double)g[s][e][c];
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < K; ++c) lss[wv][c] = ss[c];
    }
    __syncthreads();
    double* pg = part_g + (int64_t)blockIdx.x * n * K;
    for (int64_t e = threadIdx.x; e < n * K; e += 256)
        pg[e] = ((lg[0][e] + lg[1][e]) + lg[2][e]) + lg[3][e];
    if (threadIdx.x < K)
        part_ss[(int64_t)blockIdx.x * K + threadIdx.x] =
            ((lss[0][threadIdx.x] + lss[1][threadIdx.x]) + lss[2][threadIdx.x]) + lss[3][threadIdx.x];
}

// G[e] = sum_w part_g[w][e] (e < n*k), ss[c] = sum_w part_ss[w][c]: 64 outputs x 16 slices per
// workgroup, slices combined through LDS in a fixed order
__global__ void __launch_bounds__(1024)
normal_pass_finish(const double* __restrict__ part_g, const double* __restrict__ part_ss,
                   int64_t nwaves, int64_t nk, int k, double* __restrict__ G, double* __restrict__ ss) {
    constexpr int NSL = 16;
    __shared__ double red[NSL][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + tx;
    const int64_t total = nk + k;  // the k sums of squares ride behind the n*k entries of G
    double sum = 0.0;
    if (e < total) {
        // four independent chains (the loads are latency-bound), combined in a fixed order
        const double* src = e < nk ? part_g + e : part_ss + (e - nk);
        const int64_t ld = e < nk ? nk : k;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        int64_t w = ty;
        for (; w + 3 * NSL < nwaves; w += 4 * NSL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] += src[(w + q * NSL) * ld];
        }
        for (int q = 0; w < nwaves; w += NSL, ++q) s4[q] += src[w * ld];
        sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    red[ty][tx] = sum;
    __syncthreads();
    if (ty == 0 && e < total) {
        sum = 0.0;
#pragma unroll
        for (int y = 0; y < NSL; ++y) sum += red[y][tx];
        if (e < nk) G[e] = sum;
        else ss[e - nk] = sum;
    }
}

struct NormalPlan {
    int ns;                 // 16-byte steps per lane per row (0: unsupported shape)
    int64_t rows_per_wave;  // multiple of NP_ROW_ALIGN
    int64_t nwaves;
};

static NormalPlan plan_normal(int dtype, int64_t m, int64_t n, int64_t lda, int64_t k) {
    NormalPlan p{0, 0, 0};
    const int vw = dtype == RK_F32 ? 4 : 2;
    if (m <= 0 || n <= 0 || k < 1 || k > 2 || n % vw != 0 || lda % vw != 0) return p;
    const int64_t steps = ceil_div(n, (int64_t)64 * vw);
    p.ns = steps <= 1 ? 1 : (steps <= 2 ? 2 : (steps <= 4 ? 4 : 0));
    if (p.ns == 0) return p;
    // ~4096 waves (two rounds of the resident ones), at least 64 rows each
    int64_t rpw = ceil_div(m, 4096);
    if (rpw < 64) rpw = 64;
    p.rows_per_wave = round_up(rpw, NP_ROW_ALIGN);
    p.nwaves = ceil_div(m, p.rows_per_wave);
    return p;
}

template <typename T>
static int normal_pass_impl(const NormalPlan& p, const T* A, int64_t m, int64_t n, int64_t lda,
                            const T* Z, int64_t k, int64_t ldz, T* U, int64_t ldu,
                            const double* coef, double* G, double* ss, void* ws, hipStream_t st) {
    const int64_t ntab = ceil_div(p.nwaves, 4);  // one partial table per workgroup
    double* part_g = static_cast<double*>(ws);
    double* part_ss = part_g + ntab * n * k;
    dim3 grid((unsigned)ntab);
#define RK_NP(KK, NSV)                                                                           \
    hipLaunchKernelGGL((normal_pass_kernel<T, KK, NSV>), grid, dim3(256), 0, st, A, m, n, lda, Z,  \
                       ldz, U, ldu, coef, part_g, part_ss, p.rows_per_wave, p.nwaves)
#define RK_NP_K(KK)                                                                              \
    do {                                                                                         \
        if (p.ns == 1) RK_NP(KK, 1);                                                             \
        else if (p.ns == 2) RK_NP(KK, 2);                                                        \
        else RK_NP(KK, 4);                                                                       \
    } while (0)
    if (k == 1) RK_NP_K(1);
    else RK_NP_K(2);
#undef RK_NP_K
#undef RK_NP
    int rc = launch_status("rk_normal_pass");
    if (rc != RK_OK) return rc;
    hipLaunchKernelGGL(normal_pass_finish, dim3((unsigned)ceil_div(n * k + k, 64)), dim3(1024), 0, st,
                       part_g, part_ss, ntab, n * k, (int)k, G, ss);
    return launch_status("rk_normal_pass(finish)");
}

}  // namespace rk

using namespace rk;

extern "C" int rk_normal_pass_supported(int dtype, int64_t m, int64_t n, int64_t lda, int64_t k) {
    return plan_normal(dtype, m, n, lda, k).ns != 0;
}

extern "C" size_t rk_normal_pass_ws_bytes(int dtype, int64_t m, int64_t n, int64_t k) {
    const NormalPlan p = plan_normal(dtype, m, n, n, k);
    return p.ns ? (size_t)ceil_div(p.nwaves, 4) * (n * k + k) * sizeof(double) : 0;
}

extern "C" int rk_normal_pass(int dtype, const void* A, int64_t m, int64_t n, int64_t lda,
                              const void* Z, int64_t k, int64_t ldz, void* U, int64_t ldu,
                              const double* coef, double* G, double* ss, void* ws, size_t ws_bytes,
                              void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_normal_pass: bad dtype %d", dtype);
    RK_REQUIRE(A && Z && U && coef && G && ss, "rk_normal_pass: null pointer");
    RK_REQUIRE(ldz >= k && ldu >= k && lda >= n, "rk_normal_pass: leading dimension too small");
    const NormalPlan p = plan_normal(dtype, m, n, lda, k);
    RK_REQUIRE(p.ns != 0,
               "rk_normal_pass: unsupported shape (k=%lld in [1,2], n=%lld a multiple of the vector "
               "width with n*sizeof <= 4 KiB); use rk_matmat_skinny + rk_matmat_skinny_t",
               (long long)k, (long long)n);
    RK_REQUIRE(aligned16(A), "rk_normal_pass: A must be 16-byte aligned");
    const size_t need = (size_t)ceil_div(p.nwaves, 4) * (n * k + k) * sizeof(double);
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_normal_pass: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        return normal_pass_impl<float>(p, (const float*)A, m, n, lda, (const float*)Z, k, ldz, (float*)U,
                                       ldu, coef, G, ss, ws, st);
    return normal_pass_impl<double>(p, (const double*)A, m, n, lda, (const double*)Z, k, ldz,
                                    (double*)U, ldu, coef, G, ss, ws, st);
}
