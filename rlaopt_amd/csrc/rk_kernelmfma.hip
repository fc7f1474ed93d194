// Fused kernel-matrix product on the matrix cores (float32, L2 kernels, d <= 64, k <= 16):
//
//     Y[i][c] = scale * sum_j kappa(|x_i - y_j|^2) V[j][c]
//
// as two chained exact-fp32 MFMA products per 128 x 128 tile - the shape of an attention block:
//     S^T = X2_tile X1_block^T                 (v_mfma_f32_32x32x2_f32 over the d features)
//     P   = kappa(|x_i|^2 + |y_j|^2 - 2 S)     (on the accumulator registers)
//     Y^T += V_tile^T P^T                      (second MFMA; P is fed from those same registers:
//                                               with S computed transposed, a lane's accumulator
//                                               element v is exactly the B operand B[k = lane/32]
//                                               [n = lane%32] of step v of the second product)
// The VALU-formulated kernels (rk_kernelmat.hip) are bound by 2 vector instructions per pair and
// feature; here the d-long contraction runs on the MFMA pipe (64 flop/clk/SIMD, twice the scalar
// FMA rate, and 2 instead of 3 flops per pair and feature) and the VALU only evaluates kappa.
// Accuracy.  s = |x|^2 + |y|^2 - 2 x.y is formed from three rounded float32 sums, so its absolute
// error is ~ eps32 * (|x|^2 + |y|^2) / ell^2 whatever the true distance is.  The kernels are
// translation invariant, so both panels are CENTRED on the way into LDS (x' = (x - c) / ell, c
// given by the caller: rk_kernel_center) and the caller vouches for max |x'|^2 <= RK_GRAM_MAX_NORM2
// (rk_kernel_gram_bound): the error of s is then <= ~1e-5 and the entry-wise error of a kernel
// that is smooth in s (RBF, Matern-3/2, -5/2: |d kappa / d s| <= 3/2) stays an order of magnitude
// inside the float32 tolerance of the reference tests (1e-4) - including the diagonal of
// K(X, X), where s should be 0 and is ~1e-6 instead.  Matern-1/2 = exp(-sqrt(s)) is NOT smooth
// at s = 0 (sqrt(1e-6) = 1e-3): it keeps the direct (x - y)^2 kernels, like Laplace (L1) and
// float64 (1e-8 tolerance).  Without a centre from the caller the direct kernels run.
//
// Workgroup = 4 waves (2 x 2), tile = 128 points of X2 (j) x 128 points of X1 (i); a wave owns
// 64 j x 64 i = 2 x 2 MFMA blocks.  The X1 block stays in LDS (k-major) for the whole kernel; the
// X2 tile, V tile and |y|^2 of the next tile are fetched into registers during the MFMAs of the
// current one and stored to the (single) LDS buffers between two barriers: 76 KB of LDS, two
// workgroups per CU, so one workgroup's MFMAs overlap the other's kappa evaluation and waits.
// Per wave Y^T lives in 2 accumulators (32 rhs x 32 i each) for all tiles; at the end the two
// waves that share an i-range add theirs through LDS (fixed order).
#include <atomic>

#include "rk_common.h"
#include "rk_kerneltile.h"

namespace rk {

constexpr int MF_T = 128;    // tile edge (points)
constexpr int MF_D = 64;     // features staged (d <= MF_D)
constexpr int MF_LD = 130;   // LDS stride of the k-major panels (conflict-free transposing store)
constexpr int MF_LDV = 20;   // stride of the V tile rows (16 rhs, contiguous; 16-byte aligned rows)

typedef float mf_acc __attribute__((ext_vector_type(16)));

// per-feature affine map of the staged panels: x' = x * iw[c] - muw[c] with iw = 1/ell_c (1 for a
// scalar lengthscale, which kappa applies) and muw = centre_c * iw[c]; entries c >= d are zero
constexpr int GRAM_DMAX = 128;  // features the map covers (rk_kernelbf16.hip goes up to d = 128)
__global__ void gram_prep_kernel(const float* __restrict__ center, const float* __restrict__ ell_vec,
                                 int64_t d, float* __restrict__ iw, float* __restrict__ muw) {
    const int c = threadIdx.x;  // GRAM_DMAX threads
    float a = 0.f, b = 0.f;
    if (c < d) {
        a = ell_vec ? 1.f / ell_vec[c] : 1.f;
        b = center[c] * a;
    }
    iw[c] = a;
    muw[c] = b;
}

void gram_prep_launch(const float* center, const float* ell_vec, int64_t d, float* iw, float* muw,
                      hipStream_t st) {
    hipLaunchKernelGGL(gram_prep_kernel, dim3(1), dim3(GRAM_DMAX), 0, st, center, ell_vec, d, iw, muw);
}

// |row'|^2 of gathered, centred, scaled rows: out[p] = sum_c (X[row(p)][c] iw[c] - muw[c])^2.
// 16 lanes (one DPP row) per point, the features strided over them: a wave reads 4 rows in
// 64-byte runs instead of 64 rows 4 bytes each
__global__ void __launch_bounds__(256)
row_norms_kernel(const float* __restrict__ X, int64_t ldx, const int64_t* __restrict__ idx, int64_t n,
                 int64_t d, const float* __restrict__ iw, const float* __restrict__ muw,
                 float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    float s = 0.f;
    if (p < n) {
        const float* row = X + (idx ? idx[p] : p) * ldx;
        for (int64_t c = l; c < d; c += 16) {
            const float x = fmaf(row[c], iw[c], -muw[c]);
            s += x * x;
        }
    }
    s += dpp_fetch<0xB1, 0xF>(s);   // quad_perm [1,0,3,2]
    s += dpp_fetch<0x4E, 0xF>(s);   // quad_perm [2,3,0,1]
    s += dpp_fetch<0x141, 0xF>(s);  // row_half_mirror
    s += dpp_fetch<0x140, 0xF>(s);  // row_mirror: every lane of the row holds the sum
    if (p < n && l == 0) out[p] = s;
}

// element v of a 32x32 accumulator sits in row (v&3) + 8*(v>>2) + 4*(lane/32), column lane%32
__device__ __forceinline__ int mf_row(int v, int lk) { return (v & 3) + 8 * (v >> 2) + 4 * lk; }

// kappa on the accumulators: the RBF instantiation is one multiply and one v_exp_f32
template <bool RBF>
__device__ __forceinline__ float mf_kappa(int kind, float s, float w, float rbf_c) {
    if constexpr (RBF) return __builtin_amdgcn_exp2f(s * rbf_c);  // exp(-s w^2 / 2), c = -w^2 log2(e) / 2
    else return kernel_value<float>(kind, s, w);
}

// KV = 0: the second product on the matrix cores (rhs padded to 32).  KV = 1 / 2 / 4 (k <= KV):
// the second product on the VALU - per lane 64 * KV FMAs per tile into 2 * KV running sums, a
// sixteenth of the MFMA work the padded form would spend on so few right-hand sides.  (KV = 8 / 16
// were tried: the 2 * KV running sums spill, 6 - 14 % slower than the matrix cores at k = 10 / 16.)
template <bool RBF, int VLD, int KV>
__global__ void __launch_bounds__(256, 2)
kmat_matmat_mfma_kernel(int kind, const float* __restrict__ X1, int64_t n1, int64_t ldx1,
                        const int64_t* __restrict__ idx1, const float* __restrict__ X2, int64_t n2,
                        int64_t ldx2, const int64_t* __restrict__ idx2, int64_t d, float w,
                        const float* __restrict__ iw, const float* __restrict__ muw, float scale,
                        const float* __restrict__ nx, const float* __restrict__ ny,
                        const float* __restrict__ V, int k, int64_t ldv, int64_t tiles_per_seg,
                        float* __restrict__ part, float* __restrict__ Y, int64_t ldy, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*S1)[MF_LD] = reinterpret_cast<float (*)[MF_LD]>(smem);                   // [MF_D][MF_LD]
    float (*S2)[MF_LD] = reinterpret_cast<float (*)[MF_LD]>(smem + MF_D * MF_LD);     // [MF_D][MF_LD]
    float (*VS)[MF_LDV] = reinterpret_cast<float (*)[MF_LDV]>(smem + 2 * MF_D * MF_LD);  // [MF_T][MF_LDV]
    float* NY = smem + 2 * MF_D * MF_LD + MF_T * MF_LDV;                              // [MF_T]
    float* IW = NY + MF_T;                                                            // [MF_D]
    float* MUW = IW + MF_D;                                                           // [MF_D]
    const float rbf_c = -0.5f * w * w * 1.4426950408889634f;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;  // j- and i-offset of this wave
    const int lk = lane >> 5, li = lane & 31;
    const int64_t i0 = (int64_t)blockIdx.x * MF_T;
    const int64_t ntile = (n2 + MF_T - 1) / MF_T;
    const int64_t t_lo = (int64_t)blockIdx.y * tiles_per_seg;
    const int64_t t_hi = (t_lo + tiles_per_seg < ntile) ? t_lo + tiles_per_seg : ntile;
    const int dpad = (int)((d + 1) & ~(int64_t)1);  // MFMA k step is 2; features >= d are zero in LDS

    // ---- staging helpers: thread -> point tid/2, features (tid%2)*32 .. +31 -------------------
    const int sp = tid >> 1, sf = (tid & 1) * 32;
    float xr[32];
    auto load_panel = [&](const float* __restrict__ X, int64_t n, int64_t ldx,
                          const int64_t* __restrict__ idx, int64_t p0) {
        const int64_t p = p0 + sp;
        const float* row = (p < n) ? X + (idx ? idx[p] : p) * ldx : nullptr;
        if constexpr (VLD == 1) {
#pragma unroll
            for (int u = 0; u < 32; ++u) xr[u] = (row != nullptr && sf + u < d) ? row[sf + u] : 0.f;
        } else {  // rows are VLD-float aligned and d is a multiple of VLD
#pragma unroll
            for (int u = 0; u < 32; u += VLD) {
                if (row != nullptr && sf + u < d) {
                    if constexpr (VLD == 4) {
                        const float4 q = *reinterpret_cast<const float4*>(row + sf + u);
                        xr[u] = q.x; xr[u + 1] = q.y; xr[u + 2] = q.z; xr[u + 3] = q.w;
                    } else {
                        const float2 q = *reinterpret_cast<const float2*>(row + sf + u);
                        xr[u] = q.x; xr[u + 1] = q.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VLD; ++e) xr[u + e] = 0.f;
                }
            }
        }
        // centre and scale on the way in (staged once per workgroup in LDS; the two feature
        // halves read two addresses per wave: broadcasts).  Padded points stay zero: their V rows
        // are zero, and padded features have iw = muw = 0.
        if (row != nullptr) {
#pragma unroll
            for (int u = 0; u < 32; u += 4) {
                const float4 a = *reinterpret_cast<const float4*>(IW + sf + u);
                const float4 b = *reinterpret_cast<const float4*>(MUW + sf + u);
                xr[u] = fmaf(xr[u], a.x, -b.x);
                xr[u + 1] = fmaf(xr[u + 1], a.y, -b.y);
                xr[u + 2] = fmaf(xr[u + 2], a.z, -b.z);
                xr[u + 3] = fmaf(xr[u + 3], a.w, -b.w);
            }
        }
    };
    auto store_panel = [&](float (*dst)[MF_LD]) {
#pragma unroll
        for (int u = 0; u < 32; ++u) dst[sf + u][sp] = xr[u];
    };
    float vr[8], nyr;
    auto load_v = [&](int64_t j0) {  // thread -> point tid/2, rhs (tid%2)*8 .. +7 (zero past k)
        const int64_t j = j0 + sp;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = (tid & 1) * 8 + u;
            vr[u] = (j < n2 && c < k) ? V[j * ldv + c] : 0.f;
        }
        nyr = (tid < MF_T && j0 + tid < n2) ? ny[j0 + tid] : 0.f;
    };
    auto store_v = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) VS[sp][(tid & 1) * 8 + u] = vr[u];
        if (tid < MF_T) NY[tid] = nyr;
    };

    if (tid < MF_D) {
        IW[tid] = iw[tid];
        MUW[tid] = muw[tid];
    }
    __syncthreads();
    // X1 block: resident for the whole kernel
    load_panel(X1, n1, ldx1, idx1, i0);
    store_panel(S1);
    float nxr[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t i = i0 + wn + b * 32 + li;
        nxr[b] = i < n1 ? nx[i] : 0.f;
    }
    // two-level sums: the MFMA accumulators are folded into ytot every 16 tiles (2 048 columns), so
    // the float32 rounding of a 30 000-column segment grows like a pairwise, not a running, sum
    mf_acc yacc[2], ytot[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            yacc[b][v] = 0.f;
            ytot[b][v] = 0.f;
        }

    constexpr int KVN = KV > 0 ? KV : 1;
    float yv[2][KVN], yvt[2][KVN];  // (KV > 0) running sums of this lane's (i, rhs) pairs, two-level
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < KVN; ++c) yv[b][c] = yvt[b][c] = 0.f;

    if (t_lo < t_hi) {
        load_panel(X2, n2, ldx2, idx2, t_lo * MF_T);
        load_v(t_lo * MF_T);
    }
    for (int64_t t = t_lo; t < t_hi; ++t) {
        if (((t - t_lo) & 15) == 15) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < KVN; ++c) {
                    yvt[b][c] += yv[b][c];
                    yv[b][c] = 0.f;
                }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                ytot[b] += yacc[b];
#pragma unroll
                for (int v = 0; v < 16; ++v) yacc[b][v] = 0.f;
            }
        }
        __syncthreads();  // every wave is done reading the previous tile
        store_panel(S2);
        store_v();
        __syncthreads();  // tile t is in LDS (and, first time round, the X1 block)
        if (t + 1 < t_hi) {  // next tile: global loads in flight during this tile's MFMAs
            load_panel(X2, n2, ldx2, idx2, (t + 1) * MF_T);
            load_v((t + 1) * MF_T);
        }
        // ---- S^T = X2_tile X1^T over the features ------------------------------------------
        mf_acc acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
#pragma unroll 2
        for (int ks = 0; ks < dpad; ks += 2) {
            float af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = S2[ks + lk][wm + a * 32 + li];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = S1[ks + lk][wn + b * 32 + li];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        // ---- P = kappa(|x|^2 + |y|^2 - 2 S) in place; Y^T += V^T P^T ---------------------------
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float nyv = NY[wm + a * 32 + mf_row(v, lk)];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float s = nxr[b] + nyv - 2.f * acc[a][b][v];
                    s = s > 0.f ? s : 0.f;
                    acc[a][b][v] = mf_kappa<RBF>(kind, s, w, rbf_c);
                }
            }
            if constexpr (KV == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    // A operand of step v: V[j = row v of this lane half][c = lane%32] (zero for c >= 16)
                    const float av = li < 16 ? VS[wm + a * 32 + mf_row(v, lk)][li] : 0.f;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        yacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, acc[a][b][v], yacc[b], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float* vj = VS[wm + a * 32 + mf_row(v, lk)];  // same address in a half-wave: broadcast
                    float vv[KV];
                    if constexpr (KV >= 4) {
#pragma unroll
                        for (int c = 0; c < KV; c += 4) {
                            const float4 q = *reinterpret_cast<const float4*>(vj + c);
                            vv[c] = q.x; vv[c + 1] = q.y; vv[c + 2] = q.z; vv[c + 3] = q.w;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < KV; ++c) vv[c] = vj[c];
                    }
#pragma unroll
                    for (int c = 0; c < KV; ++c)
#pragma unroll
                        for (int b = 0; b < 2; ++b) yv[b][c] += acc[a][b][v] * vv[c];
                }
            }
        }
    }
    // ---- combine the partial sums of each i (two j-half waves; for KV > 0 also two lane halves) ------
    __syncthreads();  // every wave is done with the panels: reuse S2 as [8 slots][16 c][64 i]
    float* yl = &S2[0][0];
    constexpr int NSLOT = KV == 0 ? 2 : 4;  // contributions per (i, c)
    if constexpr (KV == 0) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 8; ++v) {  // rhs c = mf_row(v, lk) < 16 for v < 8
                const int c = mf_row(v, lk);
                yl[(wave * 16 + c) * 64 + b * 32 + li] = ytot[b][v] + yacc[b][v];
            }
    } else {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < KV; ++c)
                yl[((wave * 2 + lk) * 16 + c) * 64 + b * 32 + li] = yvt[b][c] + yv[b][c];
    }
    __syncthreads();
    for (int e = tid; e < MF_T * k; e += 256) {
        const int il = e / k, c = e - il * k;  // il: row within the block
        const int wcol = il >> 6;              // waves wcol (j half 0) and wcol + 2 (j half 1)
        float sum;
        if constexpr (KV == 0) {
            sum = yl[(wcol * 16 + c) * 64 + (il & 63)] + yl[((wcol + 2) * 16 + c) * 64 + (il & 63)];
        } else {
            const float s0 = yl[((wcol * 2 + 0) * 16 + c) * 64 + (il & 63)] + yl[((wcol * 2 + 1) * 16 + c) * 64 + (il & 63)];
            const float s1 = yl[(((wcol + 2) * 2 + 0) * 16 + c) * 64 + (il & 63)] + yl[(((wcol + 2) * 2 + 1) * 16 + c) * 64 + (il & 63)];
            sum = s0 + s1;
        }
        (void)NSLOT;
        const int64_t i = i0 + il;
        if (i >= n1) continue;
        if (part != nullptr) {
            part[((int64_t)blockIdx.y * n1 + i) * k + c] = sum;
        } else {
            const float o = scale * sum;
            if (accumulate) Y[i * ldy + c] += o;
            else Y[i * ldy + c] = o;
        }
    }
}

constexpr size_t MF_SMEM_BYTES =
    sizeof(float) * (2 * MF_D * MF_LD + MF_T * MF_LDV + MF_T + 2 * MF_D);  // S1 + S2 + VS + NY + IW + MUW

// launch (float32 only): the affine map (2 * GRAM_DMAX floats) and the norms of both point sets go into
// `norms` (2 * GRAM_DMAX + n1 + n2 floats), then the tiles
int launch_kmatmat_mfma(int kind, const float* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                        const float* X2, int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d,
                        float w, const float* ell_vec, const float* center, float scale, const float* V,
                        int k, int64_t ldv, int64_t S,
                        int64_t tiles_per_seg, float* norms, float* part, float* Y, int64_t ldy,
                        int accumulate, hipStream_t st) {
    float* iw = norms;
    float* muw = norms + GRAM_DMAX;
    float* nxp = norms + 2 * GRAM_DMAX;
    float* nyp = nxp + n1;
    gram_prep_launch(center, ell_vec, d, iw, muw, st);
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)ceil_div(n1, 16)), dim3(256), 0, st, X1, ldx1,
                       idx1, n1, d, iw, muw, nxp);
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)ceil_div(n2, 16)), dim3(256), 0, st, X2, ldx2,
                       idx2, n2, d, iw, muw, nyp);
    auto align = [&](const float* X, int64_t ldx) {  // widest vector load of a row piece
        const uintptr_t a = reinterpret_cast<uintptr_t>(X);
        if (a % 16 == 0 && ldx % 4 == 0 && d % 4 == 0) return 4;
        if (a % 8 == 0 && ldx % 2 == 0 && d % 2 == 0) return 2;
        return 1;
    };
    const int v1 = align(X1, ldx1), v2 = align(X2, ldx2);
    const int vld = v1 < v2 ? v1 : v2;
    dim3 grid((unsigned)ceil_div(n1, MF_T), (unsigned)S);
#define RK_MF(RBFV, VLDV)                                                                          \
    do {                                                                                           \
        if (k == 1) RK_MF_K(RBFV, VLDV, 1);                                                        \
        else if (k == 2) RK_MF_K(RBFV, VLDV, 2);                                                   \
        else if (k <= 4) RK_MF_K(RBFV, VLDV, 4);                                                   \
        else RK_MF_K(RBFV, VLDV, 0);                                                               \
    } while (0)
#define RK_MF_K(RBFV, VLDV, KVV)                                                                   \
    do {                                                                                           \
        static std::atomic<unsigned long long> attr_devs{0}; /* one bit per device ordinal */        \
        int dev_ = 0;                                                                              \
        (void)hipGetDevice(&dev_);                                                                 \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                       \
        if (!(attr_devs.load(std::memory_order_acquire) & bit_)) {                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kmat_matmat_mfma_kernel<RBFV, VLDV, KVV>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,                    \
                                    (int)MF_SMEM_BYTES) != hipSuccess) {                           \
                (void)hipGetLastError();                                                           \
                set_error("rk_kernel_matmat: cannot reserve %zu bytes of LDS", MF_SMEM_BYTES);     \
                return RK_ERR_LAUNCH;                                                              \
            }                                                                                      \
            attr_devs.fetch_or(bit_, std::memory_order_release);                                   \
        }                                                                                          \
        hipLaunchKernelGGL((kmat_matmat_mfma_kernel<RBFV, VLDV, KVV>), grid, dim3(256), MF_SMEM_BYTES, st, \
                           kind, X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, w, iw, muw, scale, nxp, nyp, V, \
                           k, ldv, tiles_per_seg, part, Y, ldy, accumulate);                       \
    } while (0)
    if (kind == RK_RBF) {
        if (vld == 4) RK_MF(true, 4);
        else if (vld == 2) RK_MF(true, 2);
        else RK_MF(true, 1);
    } else {
        if (vld == 4) RK_MF(false, 4);
        else if (vld == 2) RK_MF(false, 2);
        else RK_MF(false, 1);
    }
#undef RK_MF
#undef RK_MF_K
    return launch_status("rk_kernel_matmat(mfma)");
}

// ---- centre and norm bound of a point set (the caller-side guard of the formulation above) ------

// centre[c] = mean over <= 4096 evenly strided rows (any point near the data works as the origin;
// a fixed sample keeps this a one-workgroup, fixed-order - bitwise reproducible - reduction)
template <typename T>
__global__ void __launch_bounds__(256)
center_sample_kernel(const T* __restrict__ X, int64_t n, int64_t ldx, int64_t d, T* __restrict__ center) {
    __shared__ double red[4][64];
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const int64_t stride = n > 4096 ? n / 4096 : 1;
    const int64_t ns = (n + stride - 1) / stride;
    for (int64_t c0 = 0; c0 < d; c0 += 64) {
        const int64_t c = c0 + cl;
        double s = 0.0;
        if (c < d)
            for (int64_t r = g; r < ns; r += 4) s += (double)X[r * stride * ldx + c];
        red[g][cl] = s;
        __syncthreads();
        if (g == 0 && c < d) center[c] = (T)((red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / (double)ns);
        __syncthreads();
    }
}

// out[0] = max(out[0], max_p sum_c ((X[p][c] - centre[c]) / ell_c)^2): the maximum of non-negative
// floats is the maximum of their bit patterns, and a maximum does not depend on the order
template <typename T>
__global__ void __launch_bounds__(256)
gram_bound_kernel(const T* __restrict__ X, int64_t n, int64_t ldx, int64_t d, T w,
                  const T* __restrict__ ell_vec, const T* __restrict__ center, T* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    T s = T(0);
    if (p < n) {
        const T* row = X + p * ldx;
        for (int64_t c = l; c < d; c += 16) {
            const T x = (row[c] - center[c]) * (ell_vec ? T(1) / ell_vec[c] : w);
            s += x * x;
        }
    }
    s += dpp_fetch<0xB1, 0xF>(s);
    s += dpp_fetch<0x4E, 0xF>(s);
    s += dpp_fetch<0x141, 0xF>(s);
    s += dpp_fetch<0x140, 0xF>(s);
    if (!(s == s)) s = T(INFINITY);  // NaN data: never vouch for it
    if (p >= n) s = T(0);
    // one atomic per workgroup (a per-point atomic on one address serialises: 1.4 ms at n = 500 000)
    __shared__ T red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + h] ? red[threadIdx.x] : red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const T m = red[0];
        if constexpr (sizeof(T) == 4) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
        else atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(m));
    }
}

}  // namespace rk

using namespace rk;

extern "C" int rk_kernel_center(int dtype, const void* X, int64_t n, int64_t ldx, int64_t d,
                                void* center, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_kernel_center: bad dtype %d", dtype);
    RK_REQUIRE(X && center && n > 0 && d > 0 && ldx >= d, "rk_kernel_center: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        hipLaunchKernelGGL(center_sample_kernel<float>, dim3(1), dim3(256), 0, st, (const float*)X, n, ldx, d,
                           (float*)center);
    else
        hipLaunchKernelGGL(center_sample_kernel<double>, dim3(1), dim3(256), 0, st, (const double*)X, n, ldx,
                           d, (double*)center);
    return launch_status("rk_kernel_center");
}

extern "C" int rk_kernel_gram_bound(int dtype, const void* X, int64_t n, int64_t ldx, int64_t d,
                                    double ell_scalar, const void* ell_vec, const void* center,
                                    void* bound, void* stream) {
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_kernel_gram_bound: bad dtype %d", dtype);
    RK_REQUIRE(X && center && bound && n > 0 && d > 0 && ldx >= d, "rk_kernel_gram_bound: bad arguments");
    RK_REQUIRE(ell_vec != nullptr || ell_scalar > 0.0, "rk_kernel_gram_bound: lengthscale must be positive");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)ceil_div(n, 16));
    if (dtype == RK_F32)
        hipLaunchKernelGGL(gram_bound_kernel<float>, grid, dim3(256), 0, st, (const float*)X, n, ldx, d,
                           (float)(1.0 / ell_scalar), (const float*)ell_vec, (const float*)center, (float*)bound);
    else
        hipLaunchKernelGGL(gram_bound_kernel<double>, grid, dim3(256), 0, st, (const double*)X, n, ldx, d,
                           1.0 / ell_scalar, (const double*)ell_vec, (const double*)center, (double*)bound);
    return launch_status("rk_kernel_gram_bound");
}
