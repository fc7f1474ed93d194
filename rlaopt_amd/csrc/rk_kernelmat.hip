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
#include <stdlib.h>

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

// ------------------------------------------------------------------------------------------
// fused product, float32: 64 x 128 tile, 4 x 8 pairs per lane, packed fp32 math
// ------------------------------------------------------------------------------------------
// A lane owns 4 rows x (4 + 4) columns (columns tx*4.. and 64 + tx*4.., so that both panel reads
// stay conflict-free 16-byte accesses): 3/32 of a ds_read_b128 per pair and feature instead of
// the 2/16 of the 4 x 4 tile above, and every arithmetic instruction is a v_pk_add_f32 /
// v_pk_fma_f32 on two pairs.  Measured: the packing itself is worth only 3 % (packed fp32 issues
// at about half rate on gfx950: RK_KMAT_SCALAR_MATH=1 doubles the instruction count and loses
// 3 %); the gain over the 4 x 4 kernel - 9 to 35 % for long rows or many right-hand sides - comes
// from the wider register tile, from issuing all global loads of a panel before its LDS stores,
// and from the V slice sitting rhs-major in LDS (vs[c][column]: the 16 lanes of a row read 16
// consecutive float4; the point-major layout of the 4 x 4 kernel is a 16-way bank conflict for
// KB = 16) with the y update packed over row pairs.  The 16 lanes that share a row are one DPP
// row: their sums combine with 4 v_add_f32_dpp, no LDS.
#ifndef RK_KMAT_PREFETCH
// 1: load the next tile's panels while the current tile is computed.  Measured slower (10.1 vs
// 8.9 ms on the ASkotch row oracle): the 38 extra live registers push the kernel over the 256
// that two waves per SIMD allow.  What did pay is issuing all global loads of a panel before the
// first LDS store (panel_load / panel_store): 10.0 -> 8.9 ms.
#define RK_KMAT_PREFETCH 0
#endif
#ifndef RK_KMAT_SCALAR_MATH
#define RK_KMAT_SCALAR_MATH 0
#endif
#ifndef RK_KMAT_OCC3_KB
#define RK_KMAT_OCC3_KB 4
#endif
constexpr int TNW = 128;   // tile columns of the wide kernel
constexpr int LDPW = 132;  // padded stride of its column panel

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Staging of NP points x DCV features (chunk c0) of a row-major X into a feature-major LDS panel,
// split into a global -> register load and a register -> LDS store so that the loads of the NEXT
// tile can be in flight while the current tile is being computed.  Two lane mappings:
//  VLD = 4 / 2 : coalesced - a point's chunk is contiguous, so lanes are laid along the features
//            (VLD floats per lane, DCV / VLD lanes per point); a wave instruction touches 8 - 16
//            cache lines.  Needs 16- / 8-byte aligned rows.
//  VLD = 0 : point per lane (any alignment): lane -> point, wave -> DCV/4 consecutive features.
// Either way a thread holds NP * DCV / 256 values.
template <int NP, int DCV, int VLD>
__device__ __forceinline__ void panel_load_f32(float (&v)[NP * DCV / 256], const float* __restrict__ X,
                                               int64_t n, int64_t ldx,
                                               const int64_t* __restrict__ idx, int64_t p0,
                                               int64_t c0, int64_t d,
                                               const float* __restrict__ ell_vec) {
    if constexpr (VLD == 0) {
        const int pt = threadIdx.x & 63;
        const int cg = threadIdx.x >> 6;  // 0..3
#pragma unroll
        for (int half = 0; half < NP / 64; ++half) {
            const int64_t p = p0 + half * 64 + pt;
            const float* row = nullptr;
            if (p < n) row = X + (idx ? idx[p] : p) * ldx;
#pragma unroll
            for (int u = 0; u < DCV / 4; ++u) {
                const int c = cg * (DCV / 4) + u;
                float x = 0.f;
                if (row != nullptr && c0 + c < d) {
                    x = row[c0 + c];
                    if (ell_vec) x = x / ell_vec[c0 + c];
                }
                v[half * (DCV / 4) + u] = x;
            }
        }
    } else {
        constexpr int LPP = DCV / VLD;   // lanes per point
        constexpr int PPP = 256 / LPP;   // points per pass of the workgroup
        const int f = (threadIdx.x % LPP) * VLD;
        const int pl = threadIdx.x / LPP;
#pragma unroll
        for (int u = 0; u < NP / PPP; ++u) {
            const int64_t p = p0 + pl + u * PPP;
            float x[VLD];
#pragma unroll
            for (int e = 0; e < VLD; ++e) x[e] = 0.f;
            if (p < n) {
                const float* src = X + (idx ? idx[p] : p) * ldx + c0 + f;
                if (c0 + f + VLD <= d) {
                    if constexpr (VLD == 4) {
                        const float4 q = *reinterpret_cast<const float4*>(src);
                        x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
                    } else {
                        const float2 q = *reinterpret_cast<const float2*>(src);
                        x[0] = q.x; x[1] = q.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VLD; ++e)
                        if (c0 + f + e < d) x[e] = src[e];
                }
                if (ell_vec) {
#pragma unroll
                    for (int e = 0; e < VLD; ++e)
                        if (c0 + f + e < d) x[e] = x[e] / ell_vec[c0 + f + e];
                }
            }
#pragma unroll
            for (int e = 0; e < VLD; ++e) v[u * VLD + e] = x[e];
        }
    }
}

template <int NP, int DCV, int LD, int VLD>
__device__ __forceinline__ void panel_store_f32(float (*dst)[LD], const float (&v)[NP * DCV / 256]) {
    if constexpr (VLD == 0) {
        const int pt = threadIdx.x & 63;
        const int cg = threadIdx.x >> 6;
#pragma unroll
        for (int half = 0; half < NP / 64; ++half)
#pragma unroll
            for (int u = 0; u < DCV / 4; ++u)
                dst[cg * (DCV / 4) + u][half * 64 + pt] = v[half * (DCV / 4) + u];
    } else {
        constexpr int LPP = DCV / VLD;
        constexpr int PPP = 256 / LPP;
        const int f = (threadIdx.x % LPP) * VLD;
        const int pl = threadIdx.x / LPP;
#pragma unroll
        for (int u = 0; u < NP / PPP; ++u)
#pragma unroll
            for (int e = 0; e < VLD; ++e) dst[f + e][pl + u * PPP] = v[u * VLD + e];
    }
}

__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row
    v += dpp_fetch<0xB1, 0xF>(v);
    v += dpp_fetch<0x4E, 0xF>(v);
    v += dpp_fetch<0x141, 0xF>(v);
    v += dpp_fetch<0x140, 0xF>(v);
    return v;
}

// (few right-hand sides: 3 waves per SIMD - the LDS footprint allows 3 workgroups per CU then;
// +14 % at k = 4, neutral at k <= 2)
template <bool L1, int KB, int VLD, int DCV>
__global__ void __launch_bounds__(256, (KB <= RK_KMAT_OCC3_KB ? 3 : 2))
kmat_matmat_wide_kernel(int kind, const float* __restrict__ X1, int64_t n1, int64_t ldx1,
                        const int64_t* __restrict__ idx1, const float* __restrict__ X2, int64_t n2,
                        int64_t ldx2, const int64_t* __restrict__ idx2, int64_t d, float w,
                        const float* __restrict__ ell_vec, float scale, const float* __restrict__ V,
                        int k, int64_t ldv, int64_t tiles_per_seg, float* __restrict__ part,
                        float* __restrict__ Y, int64_t ldy, int accumulate) {
    __shared__ __attribute__((aligned(16))) float xs[DCV][LDP];
    __shared__ __attribute__((aligned(16))) float ys[DCV][LDPW];
    __shared__ __attribute__((aligned(16))) float vs[KB][LDPW];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * TM;
    const int64_t ntile = (n2 + TNW - 1) / TNW;
    const int64_t t_lo = (int64_t)blockIdx.y * tiles_per_seg;
    const int64_t t_hi = (t_lo + tiles_per_seg < ntile) ? t_lo + tiles_per_seg : ntile;

    // y2[h][c] = running sums of rows (2h, 2h + 1) for right-hand side c
    f32x2 y2[2][KB];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < KB; ++c) y2[h][c] = (f32x2)(0.f);

    const bool one_chunk = (d <= DCV);
    constexpr int VR = (TNW * KB + 255) / 256;  // V-slice values per thread
    float xreg[TM * DCV / 256], yreg[TNW * DCV / 256], vreg[VR];
    auto load_v = [&](int64_t j0) {  // V slice of a column tile, zero past k / n2
#pragma unroll
        for (int u = 0; u < VR; ++u) {
            const int e = threadIdx.x + u * 256;
            const int jj = e / KB, c = e - jj * KB;
            const int64_t j = j0 + jj;
            vreg[u] = (e < TNW * KB && j < n2 && c < k) ? V[j * ldv + c] : 0.f;
        }
    };
    auto store_v = [&]() {  // rhs-major in LDS
#pragma unroll
        for (int u = 0; u < VR; ++u) {
            const int e = threadIdx.x + u * 256;
            if (e < TNW * KB) vs[e % KB][e / KB] = vreg[u];
        }
    };
    const bool pipelined = one_chunk && RK_KMAT_PREFETCH;
    if (one_chunk) {
        panel_load_f32<TM, DCV, VLD>(xreg, X1, n1, ldx1, idx1, i0, 0, d, ell_vec);
        panel_store_f32<TM, DCV, LDP, VLD>(xs, xreg);
        if (pipelined && t_lo < t_hi) {  // first tile: nothing to overlap with yet
            panel_load_f32<TNW, DCV, VLD>(yreg, X2, n2, ldx2, idx2, t_lo * TNW, 0, d, ell_vec);
            load_v(t_lo * TNW);
        }
    }
    for (int64_t t = t_lo; t < t_hi; ++t) {
        const int64_t j0 = t * TNW;
        // acc[r][g][h]: row r, column group g (tx*4.. / 64 + tx*4..), columns (2h, 2h + 1) of it
        f32x2 acc[4][2][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[r][g][h] = (f32x2)(0.f);
        for (int64_t c0 = 0; c0 < d; c0 += DCV) {
            __syncthreads();  // readers of the previous panels / V slice are done
            if (pipelined) {
                // this tile's panel and V slice were loaded while the previous tile was computed;
                // the next tile's loads are issued now and land during this tile's arithmetic
                panel_store_f32<TNW, DCV, LDPW, VLD>(ys, yreg);
                store_v();
                __syncthreads();
                if (t + 1 < t_hi) {
                    panel_load_f32<TNW, DCV, VLD>(yreg, X2, n2, ldx2, idx2, j0 + TNW, 0, d, ell_vec);
                    load_v(j0 + TNW);
                }
            } else {
                if (!one_chunk) {
                    panel_load_f32<TM, DCV, VLD>(xreg, X1, n1, ldx1, idx1, i0, c0, d, ell_vec);
                    panel_store_f32<TM, DCV, LDP, VLD>(xs, xreg);
                }
                panel_load_f32<TNW, DCV, VLD>(yreg, X2, n2, ldx2, idx2, j0, c0, d, ell_vec);
                panel_store_f32<TNW, DCV, LDPW, VLD>(ys, yreg);
                if (c0 == 0) {
                    load_v(j0);
                    store_v();
                }
                __syncthreads();
            }
            const int nc = (int)((d - c0) < DCV ? (d - c0) : DCV);
            // operands of feature c + 1 are fetched from LDS while feature c is being used
            float4 xa = *reinterpret_cast<const float4*>(&xs[0][ty * 4]);
            float4 ya = *reinterpret_cast<const float4*>(&ys[0][tx * 4]);
            float4 yb = *reinterpret_cast<const float4*>(&ys[0][64 + tx * 4]);
            auto step = [&](int c) {
                const int cn = c + 1 < DCV ? c + 1 : DCV - 1;  // (stays inside the panel)
                const float4 xa_n = *reinterpret_cast<const float4*>(&xs[cn][ty * 4]);
                const float4 ya_n = *reinterpret_cast<const float4*>(&ys[cn][tx * 4]);
                const float4 yb_n = *reinterpret_cast<const float4*>(&ys[cn][64 + tx * 4]);
                const f32x2 yv[2][2] = {{{ya.x, ya.y}, {ya.z, ya.w}}, {{yb.x, yb.y}, {yb.z, yb.w}}};
                const float xr[4] = {xa.x, xa.y, xa.z, xa.w};
                // differences of two rows first, then their FMAs (keeps a v_pk_fma_f32 from issuing
                // right behind the v_pk_add_f32 that feeds it; measured neutral - the PMC profile,
                // VALU 58 % active / 37 % issue-stalled per 2 waves, is not explained by that)
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    f32x2 df[2][2][2];
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const f32x2 xv = (f32x2)(xr[rp * 2 + r2]);
#pragma unroll
                        for (int g = 0; g < 2; ++g)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
#if RK_KMAT_SCALAR_MATH  // experiment: two scalar ops instead of one packed op
                                df[r2][g][h] = (f32x2){xv.x - yv[g][h].x, xv.y - yv[g][h].y};
#else
                                df[r2][g][h] = xv - yv[g][h];
#endif
                            }
                    }
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int g = 0; g < 2; ++g)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                f32x2& a = acc[rp * 2 + r2][g][h];
                                if constexpr (L1) a += __builtin_elementwise_abs(df[r2][g][h]);
#if RK_KMAT_SCALAR_MATH
                                else a = (f32x2){__builtin_fmaf(df[r2][g][h].x, df[r2][g][h].x, a.x),
                                                 __builtin_fmaf(df[r2][g][h].y, df[r2][g][h].y, a.y)};
#else
                                else a = __builtin_elementwise_fma(df[r2][g][h], df[r2][g][h], a);
#endif
                            }
                }
                xa = xa_n; ya = ya_n; yb = yb_n;
            };
            if (nc == DCV) {
#pragma unroll 4
                for (int c = 0; c < DCV; ++c) step(c);
            } else {
#pragma unroll 2
                for (int c = 0; c < nc; ++c) step(c);
            }
        }
        // kernel values packed over row pairs, kv[q][h] = (K[2h][col], K[2h+1][col]), one column
        // group at a time (halves the registers the 16 + 16 values would hold at once)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x2 kv[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float a = kernel_value<float>(kind, acc[2 * h][g][q >> 1][q & 1], w);
                    const float b = kernel_value<float>(kind, acc[2 * h + 1][g][q >> 1][q & 1], w);
                    kv[q][h] = (f32x2){a, b};
                }
#pragma unroll
            for (int c = 0; c < KB; ++c) {
                // zero entries past n2 kill padded pairs
                const float4 v4 = *reinterpret_cast<const float4*>(&vs[c][g * 64 + tx * 4]);
                const float vq[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 vv = (f32x2)(vq[q]);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        y2[h][c] = __builtin_elementwise_fma(kv[q][h], vv, y2[h][c]);
                }
            }
        }
    }
    // combine the 16 lanes (tx) that share each row: they form one DPP row
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + ty * 4 + r;
#pragma unroll
        for (int c = 0; c < KB; ++c) {
            const float sum = row16_sum(y2[r >> 1][c][r & 1]);
            if (tx == 0 && i < n1 && c < k) {
                if (part != nullptr) {
                    part[((int64_t)blockIdx.y * n1 + i) * k + c] = sum;
                } else {
                    const float o = scale * sum;
                    if (accumulate) Y[i * ldy + c] += o;
                    else Y[i * ldy + c] = o;
                }
            }
        }
    }
}

struct KMPlan {
    int64_t S, tiles_per_seg;
};

static KMPlan plan_kmatmat(int64_t n1, int64_t n2, int tile_cols = TN) {
    const int64_t row_blocks = ceil_div(n1, TM);
    const int64_t ntile = ceil_div(n2, tile_cols);
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

#ifndef RK_KMAT_WIDE
#define RK_KMAT_WIDE 1
#endif
#ifndef RK_KMAT_KB12
#define RK_KMAT_KB12 1
#endif
#ifndef RK_KMAT_COAL_MIN
#define RK_KMAT_COAL_MIN 4  // coalesced staging only for 16-byte aligned rows (measured)
#endif
#ifndef RK_KMAT_WIDE_DCV64
#define RK_KMAT_WIDE_DCV64 1
#endif
// column-tile width of the fused product: the packed 64 x 128 kernel for float32
// (measured with tools/ab.py --kernel kmatmat, 200 000 - 500 000 columns: the packed kernel wins
// by 9 - 35 % for long rows (d >= 64 at k <= 2, d > 64 at any k) and for many right-hand sides
// (k >= 5 from d = 44 on); the 4 x 4 kernel - 4 workgroups per CU instead of 2 - wins below that)
static int kmat_tile_cols(bool is_f32, int64_t d, int64_t k) {
    if (!is_f32 || !RK_KMAT_WIDE) return TN;
    const bool wide = d > 64 || (d == 64 && k <= 2) || (d > 40 && k >= 5);
    return wide ? TNW : TN;
}

template <bool L1>
static void launch_kmatmat_wide(int KB, dim3 grid, hipStream_t st, int kind, const float* X1,
                                int64_t n1, int64_t ldx1, const int64_t* idx1, const float* X2,
                                int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d, float w,
                                const float* ell_vec, float scale, const float* V, int k,
                                int64_t ldv, int64_t tps, float* part, float* Y, int64_t ldy,
                                int accumulate) {
    // widest aligned load of a row chunk (chunks start at multiples of 32 features)
    auto row_align = [](const float* X, int64_t ldx) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(X);
        if (a % 16 == 0 && ldx % 4 == 0) return 4;
        if (a % 8 == 0 && ldx % 2 == 0) return 2;
        return 1;
    };
    const int v1 = row_align(X1, ldx1), v2 = row_align(X2, ldx2);
    int vld = v1 < v2 ? v1 : v2;
    if (vld < RK_KMAT_COAL_MIN) vld = 0;  // point-per-lane staging
    // one staged chunk holds 32 features (d <= 32) or 64 (fewer barriers and stagings per tile)
    const bool dc64 = RK_KMAT_WIDE_DCV64 && d > 32;
#define RK_KMW_D(KBV, VLDV, DCVV)                                                                 \
    hipLaunchKernelGGL((kmat_matmat_wide_kernel<L1, KBV, VLDV, DCVV>), grid, dim3(256), 0, st, kind, \
                       X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, w, ell_vec, scale, V, k, ldv,    \
                       tps, part, Y, ldy, accumulate)
#define RK_KMW_V(KBV, VLDV)                                                                       \
    do {                                                                                          \
        if (dc64) RK_KMW_D(KBV, VLDV, 64);                                                        \
        else RK_KMW_D(KBV, VLDV, 32);                                                             \
    } while (0)
#define RK_KMW(KBV)                                                                               \
    do {                                                                                          \
        if (vld == 4) RK_KMW_V(KBV, 4);                                                           \
        else if (vld == 2) RK_KMW_V(KBV, 2);                                                      \
        else RK_KMW_V(KBV, 0);                                                                    \
    } while (0)
    switch (KB) {
        case 1: RK_KMW(1); break;
        case 2: RK_KMW(2); break;
        case 4: RK_KMW(4); break;
        case 8: RK_KMW(8); break;
        case 12: RK_KMW(12); break;
        default: RK_KMW(16); break;
    }
#undef RK_KMW
#undef RK_KMW_V
#undef RK_KMW_D
}

// rk_kernelmfma.hip: the attention-shaped formulation on the matrix cores (fp32, L2 kernels)
int launch_kmatmat_mfma(int kind, const float* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                        const float* X2, int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d,
                        float w, const float* ell_vec, const float* center, float scale, const float* V,
                        int k, int64_t ldv, int64_t S,
                        int64_t tiles_per_seg, float* norms, float* part, float* Y, int64_t ldy,
                        int accumulate, hipStream_t st);

// float32, kernels smooth in the squared distance (RBF, Matern-3/2, -5/2), d <= 128 (bf16) / 64 (fp32), and a
// `gram_center` from the caller (who thereby vouches for the norm bound, see rk.h): the matrix-core
// formulation (rk_kernelmfma.hip; 1.4 - 2.2x the VALU kernels: 711 vs 330 Gpair/s at d = 64,
// 470 vs 281 at d = 50, k = 10).  Scalar and per-feature lengthscales both.
// RK_KMAT_MFMA=0 in the environment (read per call) selects the direct-difference kernels.
#ifndef RK_KMAT_MFMA_DMIN
#define RK_KMAT_MFMA_DMIN 1
#endif
static bool kmat_use_gram(int kind, const void* center, int64_t d, int64_t dmax, int64_t n1, int64_t n2) {
    if (!(center != nullptr && kind != RK_LAPLACE && kind != RK_MATERN12 && d >= RK_KMAT_MFMA_DMIN &&
          d <= dmax && n2 >= 128 && n1 >= 1))
        return false;
    const char* e = getenv("RK_KMAT_MFMA");
    return !(e != nullptr && e[0] == '0');
}

// rk_kernelbf16.hip: the same formulation with the distance contraction on the bf16 matrix cores
// (three-way bf16 split of every coordinate, six partial products: float32 accuracy at 2.7 x less
// matrix-core time).  RK_KMAT_BF16X3=0 (read per call) keeps the float32 MFMA kernel.
size_t bx_aux_bytes(int64_t n1, int64_t n2);
size_t bx_x2_image_bytes(int64_t n2, int64_t d);
int64_t bx_tile_points(int64_t d);
int launch_kmatmat_bf16x3(int kind, const float* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                          const float* X2, int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d, float w,
                          const float* ell_vec, const float* center, float scale, const float* V, int k,
                          int64_t ldv, int64_t S, int64_t tiles_per_seg, void* aux, float* part, float* Y,
                          int64_t ldy, int accumulate, void* x2_image, int x2_ready, hipStream_t st);

size_t bx_sym_ws_bytes(int64_t n, int64_t k);
int launch_kmatmat_bf16x3_sym(int kind, const float* X, int64_t n, int64_t ldx, const int64_t* idx, int64_t d,
                              float w, const float* ell_vec, const float* center, float scale, const float* V,
                              int k, int64_t ldv, void* aux, float* Y, int64_t ldy, int accumulate, hipStream_t st);
// symmetric operands (same points, same gather), k <= 4, d <= 64: half the tiles (RK_KMAT_SYM=0: all)
static bool kmat_use_sym() {
    const char* e = getenv("RK_KMAT_SYM");
    return !(e != nullptr && e[0] == '0');
}

// (measured, n = 200 000, d = 64: k = 1 18.3 vs 26.1 ms, k = 2 20.0 vs 27.0, k = 4 24.5 vs 29.4; d = 32 does not
// gain - the column sums are vector work and one k-step leaves little matrix-core time to save -, and below
// n ~ 65 000 the work list and the finishing kernel cost more than the saved tiles)
// (the symmetric kernel has instantiations for k <= 4 only: the knob can lower the limit, not raise it)
static int64_t sym_max_k() {
    const char* e = getenv("RK_KMAT_SYM_MAXK");
    const int64_t v = e ? atoll(e) : 4;
    return v < 4 ? v : 4;
}
static int64_t sym_min_n() { const char* e = getenv("RK_KMAT_SYM_MINN"); return e ? atoll(e) : 65536; }

static bool kmat_use_bf16x3() {
    const char* e = getenv("RK_KMAT_BF16X3");
    return !(e != nullptr && e[0] == '0');
}

static KMPlan plan_kmatmat_bf16(int64_t n1, int64_t n2, int64_t tile_points) {
    const int64_t row_blocks = ceil_div(n1, 128);
    const int64_t ntile = ceil_div(n2, tile_points);
    int64_t S = ceil_div(4096, row_blocks);  // 512 workgroups are resident: >= 8 rounds (short tail)
    const int64_t min_tiles = 512 / tile_points;  // >= 512 columns per workgroup
    if (S > ntile / min_tiles) S = ntile / min_tiles;
    if (S < 1) S = 1;
    KMPlan p;
    p.tiles_per_seg = ceil_div(ntile, S);
    p.S = ceil_div(ntile, p.tiles_per_seg);
    return p;
}

static KMPlan plan_kmatmat_mfma(int64_t n1, int64_t n2) {
    const int64_t row_blocks = ceil_div(n1, 128);
    const int64_t ntile = ceil_div(n2, 128);
    int64_t S = ceil_div(1024, row_blocks);  // one workgroup per CU is resident: >= 4 rounds
    if (S > ntile / 4) S = ntile / 4;        // >= 4 column tiles per workgroup
    if (S < 1) S = 1;
    KMPlan p;
    p.tiles_per_seg = ceil_div(ntile, S);
    p.S = ceil_div(ntile, p.tiles_per_seg);
    return p;
}

static size_t kmat_norms_offset(int64_t S, int64_t n1, int64_t k) {  // norms sit behind the partials
    return round_up((size_t)(S > 1 ? S : 0) * n1 * k * sizeof(float), 256);
}
static size_t kmat_norms_bytes(int64_t n1, int64_t n2) {  // affine map (2 x 128) + |x'|^2 of both sets
    return (size_t)(2 * 128 + n1 + n2) * sizeof(float);
}

template <typename T>
static int kmatmat_impl(int kind, const T* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                        const T* X2, int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d,
                        double ell, const T* ell_vec, double scale, const T* V, int64_t k,
                        int64_t ldv, T* Y, int64_t ldy, int accumulate, const T* gram_center, void* ws,
                        size_t ws_bytes, hipStream_t st, void* x2_image = nullptr, size_t x2_image_bytes = 0,
                        int x2_ready = 0, int* x2_valid = nullptr) {
    if (x2_valid != nullptr) *x2_valid = 0;  // (set where the image is written or used)
    const T w = ell_vec ? T(1) : T(1.0 / ell);
    if constexpr (sizeof(T) == 4) {
        if (kmat_use_gram(kind, gram_center, d, 64, n1, n2) && kmat_use_bf16x3() && kmat_use_sym() && X1 == X2 &&
            idx1 == idx2 && n1 == n2 && ldx1 == ldx2 && k <= sym_max_k() && d > 32 && n1 >= sym_min_n() && ws != nullptr &&
            ws_bytes >= bx_sym_ws_bytes(n1, k))
            return launch_kmatmat_bf16x3_sym(kind, X1, n1, ldx1, idx1, d, (float)w, ell_vec, gram_center,
                                             (float)scale, V, (int)k, ldv, ws, Y, ldy, accumulate, st);
        if (kmat_use_gram(kind, gram_center, d, 128, n1, n2) && kmat_use_bf16x3()) {
            const KMPlan pb = plan_kmatmat_bf16(n1, n2, bx_tile_points(d));
            const size_t off = kmat_norms_offset(pb.S, n1, k);
            const size_t need = off + bx_aux_bytes(n1, n2);
            if (ws == nullptr || ws_bytes < need) {
                set_error("rk_kernel_matmat: workspace %zu B < required %zu B", ws_bytes, need);
                return RK_ERR_WORKSPACE;
            }
            float* partb = pb.S > 1 ? static_cast<float*>(ws) : nullptr;
            if (x2_image != nullptr && x2_image_bytes < bx_x2_image_bytes(n2, d)) {
                set_error("rk_kernel_matmat_cached: image buffer %zu B < required %zu B", x2_image_bytes,
                          bx_x2_image_bytes(n2, d));
                return RK_ERR_WORKSPACE;
            }
            int rcb = launch_kmatmat_bf16x3(kind, X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, (float)w, ell_vec,
                                            gram_center, (float)scale, V, (int)k, ldv, pb.S, pb.tiles_per_seg,
                                            static_cast<char*>(ws) + off, partb, Y, ldy, accumulate, x2_image,
                                            x2_ready, st);
            if (x2_valid != nullptr && x2_image != nullptr && rcb == RK_OK) *x2_valid = 1;
            if (rcb != RK_OK || partb == nullptr) return rcb;
            hipLaunchKernelGGL((kmat_matmat_finish<T>), dim3((unsigned)ceil_div(n1 * k, 256)), dim3(256),
                               0, st, partb, (int)pb.S, n1, (int)k, (T)scale, Y, ldy, accumulate);
            return launch_status("rk_kernel_matmat(finish)");
        }
        if (kmat_use_gram(kind, gram_center, d, 64, n1, n2)) {
            const KMPlan pm = plan_kmatmat_mfma(n1, n2);
            const size_t off = kmat_norms_offset(pm.S, n1, k);
            const size_t need = off + kmat_norms_bytes(n1, n2);
            if (ws == nullptr || ws_bytes < need) {
                set_error("rk_kernel_matmat: workspace %zu B < required %zu B", ws_bytes, need);
                return RK_ERR_WORKSPACE;
            }
            float* partm = pm.S > 1 ? static_cast<float*>(ws) : nullptr;
            float* norms = reinterpret_cast<float*>(static_cast<char*>(ws) + off);
            int rcm = launch_kmatmat_mfma(kind, X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, (float)w,
                                          ell_vec, gram_center, (float)scale, V, (int)k, ldv, pm.S,
                                          pm.tiles_per_seg, norms,
                                          partm, Y, ldy, accumulate, st);
            if (rcm != RK_OK || partm == nullptr) return rcm;
            hipLaunchKernelGGL((kmat_matmat_finish<T>), dim3((unsigned)ceil_div(n1 * k, 256)), dim3(256),
                               0, st, partm, (int)pm.S, n1, (int)k, (T)scale, Y, ldy, accumulate);
            return launch_status("rk_kernel_matmat(finish)");
        }
    }
    const KMPlan p = plan_kmatmat(n1, n2, kmat_tile_cols(sizeof(T) == 4, d, k));
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
    if constexpr (sizeof(T) == 4) {
        if (kmat_tile_cols(true, d, k) == TNW) {
            const int KBW = (RK_KMAT_KB12 && k > 8 && k <= 12) ? 12 : KB;  // ASkotch's k = 10
            if (kind == RK_LAPLACE)
                launch_kmatmat_wide<true>(KBW, grid, st, kind, X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d,
                                          w, ell_vec, (T)scale, V, (int)k, ldv, p.tiles_per_seg, part,
                                          Y, ldy, accumulate);
            else
                launch_kmatmat_wide<false>(KBW, grid, st, kind, X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d,
                                           w, ell_vec, (T)scale, V, (int)k, ldv, p.tiles_per_seg, part,
                                           Y, ldy, accumulate);
            int rcw = launch_status("rk_kernel_matmat");
            if (rcw != RK_OK || part == nullptr) return rcw;
            hipLaunchKernelGGL((kmat_matmat_finish<T>), dim3((unsigned)ceil_div(n1 * k, 256)), dim3(256),
                               0, st, part, (int)p.S, n1, (int)k, (T)scale, Y, ldy, accumulate);
            return launch_status("rk_kernel_matmat(finish)");
        }
    }
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
    // the tile shape depends on d (not an argument here): size for the larger of the two plans
    const KMPlan pa = plan_kmatmat(n1, n2, TN), pb = plan_kmatmat(n1, n2, TNW);
    const int64_t S = pa.S > pb.S ? pa.S : pb.S;
    size_t bytes = S > 1 ? (size_t)S * n1 * k * (dtype == RK_F32 ? 4 : 8) : 0;
    if (dtype == RK_F32) {  // the matrix-core path: its partials + the |x|^2 of both point sets
        const KMPlan pm = plan_kmatmat_mfma(n1, n2);
        const size_t m = kmat_norms_offset(pm.S, n1, k) + kmat_norms_bytes(n1, n2);
        if (m > bytes) bytes = m;
        for (int64_t tp = 32; tp <= 64; tp *= 2) {  // (d is not an argument: both tile geometries)
            const KMPlan pb = plan_kmatmat_bf16(n1, n2, tp);
            const size_t b = kmat_norms_offset(pb.S, n1, k) + bx_aux_bytes(n1, n2);
            if (b > bytes) bytes = b;
        }
    }
    return bytes;
}

/* workspace that additionally enables the symmetric evaluation (X1 == X2, idx1 == idx2, k <= 4) */
extern "C" size_t rk_kernel_matmat_sym_ws_bytes(int dtype, int64_t n, int64_t k) {
    if (dtype != RK_F32 || n < 256 || k < 1 || k > 4) return 0;
    return bx_sym_ws_bytes(n, k);
}

extern "C" int rk_kernel_matmat(int dtype, int kind, const void* X1, int64_t n1, int64_t ldx1,
                                const int64_t* idx1, const void* X2, int64_t n2, int64_t ldx2,
                                const int64_t* idx2, int64_t d, double ell_scalar,
                                const void* ell_vec, double scale, const void* V, int64_t k,
                                int64_t ldv, void* Y, int64_t ldy, int accumulate,
                                const void* gram_center, void* ws, size_t ws_bytes, void* stream) {
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
                                   (const float*)V, k, ldv, (float*)Y, ldy, accumulate,
                                   (const float*)gram_center, ws, ws_bytes, st);
    return kmatmat_impl<double>(kind, (const double*)X1, n1, ldx1, idx1, (const double*)X2, n2, ldx2,
                                idx2, d, ell_scalar, (const double*)ell_vec, scale, (const double*)V,
                                k, ldv, (double*)Y, ldy, accumulate, (const double*)gram_center, ws,
                                ws_bytes, st);
}

extern "C" size_t rk_kernel_x2_image_bytes(int dtype, int kind, int64_t n1, int64_t n2, int64_t d, int64_t k,
                                           int symmetric_operands) {
    // only the non-symmetric bf16 matrix-core formulation keeps an image of its X2 side
    if (dtype != RK_F32 || n2 <= 0 || symmetric_operands) return 0;
    if (!(kind == RK_RBF || kind == RK_MATERN32 || kind == RK_MATERN52)) return 0;
    if (d < RK_KMAT_MFMA_DMIN || d > 128 || n2 < 128 || n1 < 1 || k < 1 || k > 16) return 0;
    return bx_x2_image_bytes(n2, d);
}

extern "C" int rk_kernel_matmat_cached(int dtype, int kind, const void* X1, int64_t n1, int64_t ldx1,
                                       const int64_t* idx1, const void* X2, int64_t n2, int64_t ldx2,
                                       const int64_t* idx2, int64_t d, double ell_scalar, const void* ell_vec,
                                       double scale, const void* V, int64_t k, int64_t ldv, void* Y, int64_t ldy,
                                       int accumulate, const void* gram_center, void* ws, size_t ws_bytes,
                                       void* x2_image, size_t x2_image_bytes, int x2_image_ready,
                                       int* x2_image_valid, void* stream) {
    if (x2_image_valid != nullptr) *x2_image_valid = 0;
    if (dtype != RK_F32 || x2_image == nullptr)
        return rk_kernel_matmat(dtype, kind, X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, ell_scalar, ell_vec, scale,
                                V, k, ldv, Y, ldy, accumulate, gram_center, ws, ws_bytes, stream);
    clear_error();
    int rc = check_kernel_args("rk_kernel_matmat_cached", dtype, kind, X1, n1, ldx1, X2, n2, ldx2, d, ell_scalar,
                               ell_vec);
    if (rc != RK_OK) return rc;
    RK_REQUIRE(V && Y, "rk_kernel_matmat_cached: null pointer");
    RK_REQUIRE(k >= 1 && k <= 16, "rk_kernel_matmat_cached: k=%lld outside [1,16]", (long long)k);
    RK_REQUIRE(ldv >= k && ldy >= k, "rk_kernel_matmat_cached: leading dimension too small");
    RK_REQUIRE(!(X1 == X2 && idx1 == idx2 && n1 == n2),
               "rk_kernel_matmat_cached: symmetric operands keep no image (use rk_kernel_matmat)");
    return kmatmat_impl<float>(kind, (const float*)X1, n1, ldx1, idx1, (const float*)X2, n2, ldx2, idx2, d,
                               ell_scalar, (const float*)ell_vec, scale, (const float*)V, k, ldv, (float*)Y, ldy,
                               accumulate, (const float*)gram_center, ws, ws_bytes, static_cast<hipStream_t>(stream),
                               x2_image, x2_image_bytes, x2_image_ready, x2_image_valid);
}
