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
// |x|^2 + |y|^2 - 2 x.y cancels to ~1e-7 absolute in fp32 - inside the fp32 tolerance (1e-4);
// float64 keeps the direct (x - y)^2 kernels for its 1e-8 tolerance, and so does Laplace (L1).
//
// Workgroup = 4 waves (2 x 2), tile = 128 points of X2 (j) x 128 points of X1 (i); a wave owns
// 64 j x 64 i = 2 x 2 MFMA blocks.  The X1 block stays in LDS (k-major) for the whole kernel; the
// X2 tile, V tile and |y|^2 of the next tile are fetched into registers during the MFMAs of the
// current one and stored to the (single) LDS buffers between two barriers: 76 KB of LDS, two
// workgroups per CU, so one workgroup's MFMAs overlap the other's kappa evaluation and waits.
// Per wave Y^T lives in 2 accumulators (32 rhs x 32 i each) for all tiles; at the end the two
// waves that share an i-range add theirs through LDS (fixed order).
#include "rk_common.h"
#include "rk_kerneltile.h"

namespace rk {

constexpr int MF_T = 128;    // tile edge (points)
constexpr int MF_D = 64;     // features staged (d <= MF_D)
constexpr int MF_LD = 130;   // LDS stride of the k-major panels (conflict-free transposing store)
constexpr int MF_LDV = 20;   // stride of the V tile rows (16 rhs, contiguous; 16-byte aligned rows)

typedef float mf_acc __attribute__((ext_vector_type(16)));

// |row|^2 of gathered rows: out[p] = sum_c X[row(p)][c]^2.  16 lanes (one DPP row) per point, the
// features strided over them: a wave reads 4 rows in 64-byte runs instead of 64 rows 4 bytes each
__global__ void __launch_bounds__(256)
row_norms_kernel(const float* __restrict__ X, int64_t ldx, const int64_t* __restrict__ idx, int64_t n,
                 int64_t d, const float* __restrict__ ell_vec, float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    float s = 0.f;
    if (p < n) {
        const float* row = X + (idx ? idx[p] : p) * ldx;
        for (int64_t c = l; c < d; c += 16) {
            const float x = ell_vec ? row[c] / ell_vec[c] : row[c];
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
                        const float* __restrict__ ell_vec, float scale, const float* __restrict__ nx,
                        const float* __restrict__ ny,
                        const float* __restrict__ V, int k, int64_t ldv, int64_t tiles_per_seg,
                        float* __restrict__ part, float* __restrict__ Y, int64_t ldy, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*S1)[MF_LD] = reinterpret_cast<float (*)[MF_LD]>(smem);                   // [MF_D][MF_LD]
    float (*S2)[MF_LD] = reinterpret_cast<float (*)[MF_LD]>(smem + MF_D * MF_LD);     // [MF_D][MF_LD]
    float (*VS)[MF_LDV] = reinterpret_cast<float (*)[MF_LDV]>(smem + 2 * MF_D * MF_LD);  // [MF_T][MF_LDV]
    float* NY = smem + 2 * MF_D * MF_LD + MF_T * MF_LDV;                              // [MF_T]
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
        if (ell_vec != nullptr) {  // per-feature lengthscales (ARD): features are divided on the way in
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (sf + u < d) xr[u] = xr[u] / ell_vec[sf + u];
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
    sizeof(float) * (2 * MF_D * MF_LD + MF_T * MF_LDV + MF_T);  // S1 + S2 + VS + NY

// launch (float32 only): norms of both point sets into `norms` (n1 + n2 floats), then the tiles
int launch_kmatmat_mfma(int kind, const float* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                        const float* X2, int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d,
                        float w, const float* ell_vec, float scale, const float* V, int k, int64_t ldv,
                        int64_t S,
                        int64_t tiles_per_seg, float* norms, float* part, float* Y, int64_t ldy,
                        int accumulate, hipStream_t st) {
    float* nxp = norms;
    float* nyp = norms + n1;
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)ceil_div(n1, 16)), dim3(256), 0, st, X1, ldx1,
                       idx1, n1, d, ell_vec, nxp);
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)ceil_div(n2, 16)), dim3(256), 0, st, X2, ldx2,
                       idx2, n2, d, ell_vec, nyp);
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
        static bool attr_set = false;                                                              \
        if (!attr_set) {                                                                           \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kmat_matmat_mfma_kernel<RBFV, VLDV, KVV>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,                    \
                                    (int)MF_SMEM_BYTES) != hipSuccess) {                           \
                (void)hipGetLastError();                                                           \
                set_error("rk_kernel_matmat: cannot reserve %zu bytes of LDS", MF_SMEM_BYTES);     \
                return RK_ERR_LAUNCH;                                                              \
            }                                                                                      \
            attr_set = true;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL((kmat_matmat_mfma_kernel<RBFV, VLDV, KVV>), grid, dim3(256), MF_SMEM_BYTES, st, \
                           kind, X1, n1, ldx1, idx1, X2, n2, ldx2, idx2, d, w, ell_vec, scale, nxp, nyp, V, \
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

}  // namespace rk
