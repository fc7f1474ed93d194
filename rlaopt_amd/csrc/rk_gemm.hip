// Dense GEMMs on the gfx950 matrix cores (exact-f32 and f64 MFMA), row-major.
//
//   rk_gemm     C = alpha op(A) op(B) + beta C          (sketch product Y = K Omega, U = Y M)
//   rk_gram_tn  C(f64) = A^T B, f64 accumulation         (Omega^T Y, Y^T Y : reduce over n)
//
// Shape: 128 x 128 block tile, 256 threads = 4 waves in 2 x 2, each wave a 64 x 64
// sub-tile: 2 x 2 v_mfma_f32_32x32x2_f32 tiles (f32) or 4 x 4 v_mfma_f64_16x16x4_f64
// tiles (f64).  Both operands are staged k-major in LDS (S[k][i]) so that the MFMA operand
// fetch - lane l reads element [k0 + l/32][i0 + l%32] (f32) or [k0 + l/16][i0 + l%16]
// (f64) - is a run of consecutive words per lane group: conflict-free ds_read_b32/b64.
// Global->LDS goes through registers (the transposing store cannot be an LDS-DMA, whose
// destination is lane-linear); tile t+1 is fetched into registers before the MFMAs of
// tile t and stored to the other LDS buffer after them: one barrier per k-tile.
// Workgroup ids are remapped so that the column tiles of one row panel run on one XCD
// (shared L2) - the big operand (the kernel shard) is then fetched from HBM once.
#include "rk_common.h"

namespace rk {

constexpr int BM = 128, BN = 128, BK = 16;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename TC>
struct MfmaTraits;
template <>
struct MfmaTraits<float> {
    static constexpr int MT = 32;    // tile edge
    static constexpr int KM = 2;     // k per instruction
    static constexpr int NT = 2;     // tiles per wave edge (64 / MT)
    static constexpr int LDT = 130;  // LDS row stride (elements): conflict-free transposing store
    using Acc = f32x16;
};
template <>
struct MfmaTraits<double> {
    static constexpr int MT = 16;
    static constexpr int KM = 4;
    static constexpr int NT = 4;
    static constexpr int LDT = 144;  // rows k, k+1 land 32 banks apart for ds_read_b64
    using Acc = f64x4;
};

// 8 consecutive elements per thread of a (128 x BK) operand tile.
//  KCONTIG: memory is [mn][k] (k contiguous): thread -> (mn = t/2, k = (t%2)*8 ..)
//  else   : memory is [k][mn] (mn contiguous): thread -> (k = t/16, mn = (t%16)*8 ..)
template <typename TI, typename TC, bool KCONTIG>
__device__ __forceinline__ void tile_fetch(TC regs[8], const TI* __restrict__ G, int64_t ld,
                                           int64_t dim_mn, int64_t k_end, int64_t mn0, int64_t k0,
                                           bool vec) {
    const int t = threadIdx.x;
    int64_t row, col, row_lim, col_lim;
    if constexpr (KCONTIG) {
        row = mn0 + (t >> 1); col = k0 + (t & 1) * 8; row_lim = dim_mn; col_lim = k_end;
    } else {
        row = k0 + (t >> 4); col = mn0 + (t & 15) * 8; row_lim = k_end; col_lim = dim_mn;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) regs[u] = TC(0);
    if (row >= row_lim) return;
    const TI* p = G + row * ld + col;
    if (vec && col + 8 <= col_lim) {
        constexpr int VW = 16 / sizeof(TI);
        using V16 = typename Vec16<TI>::type;
#pragma unroll
        for (int v = 0; v < 8 / VW; ++v) {
            V16 x = reinterpret_cast<const V16*>(p)[v];
            const TI* xe = reinterpret_cast<const TI*>(&x);
#pragma unroll
            for (int e = 0; e < VW; ++e) regs[v * VW + e] = (TC)xe[e];
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (col + u < col_lim) regs[u] = (TC)p[u];
    }
}

template <typename TC, bool KCONTIG, int LDT>
__device__ __forceinline__ void tile_store(TC (*S)[LDT], const TC regs[8]) {
    const int t = threadIdx.x;
    if constexpr (KCONTIG) {
        const int i = t >> 1, kk = (t & 1) * 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) S[kk + u][i] = regs[u];
    } else {
        const int k = t >> 4, ii = (t & 15) * 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) S[k][ii + u] = regs[u];
    }
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// bijective XCD-aware remap of a linear workgroup id (guide T1): ids that are consecutive
// AFTER the remap run on one XCD
__device__ __forceinline__ int64_t xcd_remap(int64_t id, int64_t nwg) {
    const int64_t q = nwg / 8, r = nwg % 8, x = id % 8;
    const int64_t base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + id / 8;
}

// AK: A stored [m][kd] (not transposed)  -> k contiguous.   else stored [kd][m].
// BK_: B stored [n][kd] (transposed)     -> k contiguous.   else stored [kd][n].
template <typename TI, typename TC, bool AKC, bool BKC>
__global__ void __launch_bounds__(256)
gemm_kernel(int64_t M, int64_t N, int64_t Kd, const TI* __restrict__ A, int64_t lda,
            const TI* __restrict__ B, int64_t ldb, TC alpha, TC beta, TI* __restrict__ C,
            int64_t ldc, TC* __restrict__ part, int64_t kslice, int vecA, int vecB) {
    using TR = MfmaTraits<TC>;
    constexpr int MT = TR::MT, KM = TR::KM, NT = TR::NT, LDT = TR::LDT;
    __shared__ TC As[2][BK][LDT];
    __shared__ TC Bs[2][BK][LDT];

    const int64_t tiles_n = (N + BN - 1) / BN;
    const int64_t tiles_m = (M + BM - 1) / BM;
    const int64_t id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int64_t m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * BN;
    const int64_t k_lo = (int64_t)blockIdx.z * kslice;
    const int64_t k_hi = (k_lo + kslice < Kd) ? k_lo + kslice : Kd;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lk = lane / MT, li = lane % MT;  // operand k-row and index within the tile

    typename TR::Acc acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int v = 0; v < (int)(sizeof(typename TR::Acc) / sizeof(TC)); ++v) acc[a][b][v] = TC(0);

    TC ra[8], rb[8];
    tile_fetch<TI, TC, AKC>(ra, A, lda, M, k_hi, m0, k_lo, vecA);
    tile_fetch<TI, TC, BKC>(rb, B, ldb, N, k_hi, n0, k_lo, vecB);
    tile_store<TC, AKC, LDT>(As[0], ra);
    tile_store<TC, BKC, LDT>(Bs[0], rb);
    __syncthreads();

    int buf = 0;
    for (int64_t k0 = k_lo; k0 < k_hi; k0 += BK) {
        const bool more = (k0 + BK < k_hi);
        if (more) {
            tile_fetch<TI, TC, AKC>(ra, A, lda, M, k_hi, m0, k0 + BK, vecA);
            tile_fetch<TI, TC, BKC>(rb, B, ldb, N, k_hi, n0, k0 + BK, vecB);
        }
#pragma unroll
        for (int ks = 0; ks < BK; ks += KM) {
            TC af[NT], bf[NT];
#pragma unroll
            for (int a = 0; a < NT; ++a) af[a] = As[buf][ks + lk][wm + a * MT + li];
#pragma unroll
            for (int b = 0; b < NT; ++b) bf[b] = Bs[buf][ks + lk][wn + b * MT + li];
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = mfma(af[a], bf[b], acc[a][b]);
        }
        if (more) {
            tile_store<TC, AKC, LDT>(As[buf ^ 1], ra);
            tile_store<TC, BKC, LDT>(Bs[buf ^ 1], rb);
        }
        __syncthreads();
        buf ^= 1;
    }

    // epilogue: C/D lane maps (guide section 3): f32 32x32: col = lane%32,
    // row = (v%4) + 8*(v/4) + 4*(lane/32); f64 16x16: col = lane%16, row = lane/16 + 4*v
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int64_t col = n0 + wn + b * MT + li;
#pragma unroll
            for (int v = 0; v < (int)(sizeof(typename TR::Acc) / sizeof(TC)); ++v) {
                int rloc;
                if constexpr (sizeof(TC) == 4) rloc = (v & 3) + 8 * (v >> 2) + 4 * lk;
                else rloc = lk + 4 * v;
                const int64_t row = m0 + wm + a * MT + rloc;
                if (row < M && col < N) {
                    if (part != nullptr) {
                        part[((int64_t)blockIdx.z * M + row) * N + col] = acc[a][b][v];
                    } else {
                        TC o = alpha * acc[a][b][v];
                        if (beta != TC(0)) o += beta * (TC)C[row * ldc + col];
                        C[row * ldc + col] = (TI)o;
                    }
                }
            }
        }
}

// C[e] = alpha * sum_z part[z][e] + beta * C[e]   (fixed order -> reproducible)
template <typename TC, typename TO>
__global__ void __launch_bounds__(256)
splitk_reduce(const TC* __restrict__ part, int S, int64_t M, int64_t N, TC alpha, TC beta,
              TO* __restrict__ C, int64_t ldc) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * N) return;
    const int64_t i = e / N, j = e - i * N;
    TC s = TC(0);
    for (int z = 0; z < S; ++z) s += part[(int64_t)z * M * N + e];
    TC o = alpha * s;
    if (beta != TC(0)) o += beta * (TC)C[i * ldc + j];
    C[i * ldc + j] = (TO)o;
}

template <typename TI, typename TC>
static int gemm_dispatch(int transA, int transB, int64_t M, int64_t N, int64_t Kd, double alpha,
                         const TI* A, int64_t lda, const TI* B, int64_t ldb, double beta, TI* C,
                         int64_t ldc, TC* part, int64_t S, int64_t kslice, hipStream_t st) {
    constexpr int VW = 16 / sizeof(TI);
    const int vecA = aligned16(A) && (lda % VW == 0);
    const int vecB = aligned16(B) && (ldb % VW == 0);
    const int64_t tiles = ceil_div(M, BM) * ceil_div(N, BN);
    dim3 grid((unsigned)tiles, 1, (unsigned)S);
#define RK_GEMM_LAUNCH(AKC, BKC)                                                                  \
    hipLaunchKernelGGL((gemm_kernel<TI, TC, AKC, BKC>), grid, dim3(256), 0, st, M, N, Kd, A, lda, \
                       B, ldb, (TC)alpha, (TC)beta, C, ldc, part, kslice, vecA, vecB)
    // A not transposed -> k contiguous; B transposed -> k contiguous
    if (!transA && !transB) RK_GEMM_LAUNCH(true, false);
    else if (!transA && transB) RK_GEMM_LAUNCH(true, true);
    else if (transA && !transB) RK_GEMM_LAUNCH(false, false);
    else RK_GEMM_LAUNCH(false, true);
#undef RK_GEMM_LAUNCH
    return launch_status("rk_gemm");
}

static void gram_plan(int64_t rows, int64_t m, int64_t n, int64_t* S, int64_t* kslice) {
    const int64_t tiles = ceil_div(m, BM) * ceil_div(n, BN);
    int64_t s = ceil_div(512, tiles);
    const int64_t max_s = ceil_div(rows, 8 * BK);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    *kslice = round_up(ceil_div(rows, s), BK);
    *S = ceil_div(rows, *kslice);
}

}  // namespace rk

using namespace rk;

// split the reduction dimension when the output has too few tiles to fill 256 CUs
static void gemm_plan(int64_t m, int64_t n, int64_t kd, int64_t* S, int64_t* kslice) {
    const int64_t tiles = ceil_div(m, BM) * ceil_div(n, BN);
    int64_t s = 1;
    if (tiles < 256) {
        s = ceil_div(512, tiles);
        const int64_t max_s = ceil_div(kd, 16 * BK);
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
    }
    *kslice = round_up(ceil_div(kd, s), BK);
    *S = ceil_div(kd, *kslice);
}

extern "C" size_t rk_gemm_ws_bytes(int dtype, int64_t m, int64_t n, int64_t kd) {
    if (m <= 0 || n <= 0 || kd <= 0) return 0;
    int64_t S, ks;
    gemm_plan(m, n, kd, &S, &ks);
    return S > 1 ? (size_t)S * m * n * (dtype == RK_F32 ? 4 : 8) : 0;
}

extern "C" int rk_gemm(int dtype, int transA, int transB, int64_t m, int64_t n, int64_t kd,
                       double alpha, const void* A, int64_t lda, const void* B, int64_t ldb,
                       double beta, void* C, int64_t ldc, void* ws, size_t ws_bytes,
                       void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_gemm: bad dtype %d", dtype);
    RK_REQUIRE(A && B && C, "rk_gemm: null pointer");
    RK_REQUIRE(m > 0 && n > 0 && kd > 0, "rk_gemm: empty operand");
    RK_REQUIRE(lda >= (transA ? m : kd) && ldb >= (transB ? kd : n) && ldc >= n,
               "rk_gemm: leading dimension too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int64_t S, ks;
    gemm_plan(m, n, kd, &S, &ks);
    if (S > 1) {
        const size_t need = (size_t)S * m * n * (dtype == RK_F32 ? 4 : 8);
        if (ws == nullptr || ws_bytes < need) {
            set_error("rk_gemm: workspace %zu B < required %zu B", ws_bytes, need);
            return RK_ERR_WORKSPACE;
        }
    }
    int rc;
    if (dtype == RK_F32) {
        float* part = S > 1 ? static_cast<float*>(ws) : nullptr;
        rc = gemm_dispatch<float, float>(transA, transB, m, n, kd, alpha, (const float*)A, lda,
                                         (const float*)B, ldb, beta, (float*)C, ldc, part, S, ks, st);
        if (rc == RK_OK && S > 1) {
            hipLaunchKernelGGL((splitk_reduce<float, float>), dim3((unsigned)ceil_div(m * n, 256)),
                               dim3(256), 0, st, part, (int)S, m, n, (float)alpha, (float)beta,
                               (float*)C, ldc);
            rc = launch_status("rk_gemm(reduce)");
        }
    } else {
        double* part = S > 1 ? static_cast<double*>(ws) : nullptr;
        rc = gemm_dispatch<double, double>(transA, transB, m, n, kd, alpha, (const double*)A, lda,
                                           (const double*)B, ldb, beta, (double*)C, ldc, part, S, ks,
                                           st);
        if (rc == RK_OK && S > 1) {
            hipLaunchKernelGGL((splitk_reduce<double, double>), dim3((unsigned)ceil_div(m * n, 256)),
                               dim3(256), 0, st, part, (int)S, m, n, alpha, beta, (double*)C, ldc);
            rc = launch_status("rk_gemm(reduce)");
        }
    }
    return rc;
}

extern "C" size_t rk_gram_tn_ws_bytes(int64_t rows, int64_t m, int64_t n) {
    if (rows <= 0 || m <= 0 || n <= 0) return 0;
    int64_t S, ks;
    gram_plan(rows, m, n, &S, &ks);
    return (size_t)S * m * n * sizeof(double);
}

extern "C" int rk_gram_tn(int dtype, const void* A, int64_t lda, const void* B, int64_t ldb,
                          int64_t rows, int64_t m, int64_t n, double* C, int64_t ldc, void* ws,
                          size_t ws_bytes, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_gram_tn: bad dtype %d", dtype);
    RK_REQUIRE(A && B && C, "rk_gram_tn: null pointer");
    RK_REQUIRE(rows > 0 && m > 0 && n > 0, "rk_gram_tn: empty operand");
    RK_REQUIRE(lda >= m && ldb >= n && ldc >= n, "rk_gram_tn: leading dimension too small");
    int64_t S, ks;
    gram_plan(rows, m, n, &S, &ks);
    const size_t need = (size_t)S * m * n * sizeof(double);
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_gram_tn: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(ws);
    int rc;
    // op(A) = A^T : A stored [rows][m] -> "transposed" storage, mn contiguous
    if (dtype == RK_F32)
        rc = gemm_dispatch<float, double>(1, 0, m, n, rows, 1.0, (const float*)A, lda,
                                          (const float*)B, ldb, 0.0, nullptr, 0, part, S, ks, st);
    else
        rc = gemm_dispatch<double, double>(1, 0, m, n, rows, 1.0, (const double*)A, lda,
                                           (const double*)B, ldb, 0.0, nullptr, 0, part, S, ks, st);
    if (rc != RK_OK) return rc;
    hipLaunchKernelGGL((splitk_reduce<double, double>), dim3((unsigned)ceil_div(m * n, 256)),
                       dim3(256), 0, st, part, (int)S, m, n, 1.0, 0.0, C, ldc);
    return launch_status("rk_gram_tn(reduce)");
}
