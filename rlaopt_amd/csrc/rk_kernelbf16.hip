// Fused kernel-matrix product with the distance contraction on the DOUBLE-RATE bf16 matrix cores of
// gfx950, at float32 accuracy (float32 data, kernels smooth in the squared distance, d <= 64):
//
//     Y[i][c] = scale * sum_j kappa(|x'_i|^2 + |y'_j|^2 - 2 x'_i . y'_j) V[j][c]
//
// Every centred, scaled coordinate x' (see rk_kernelmfma.hip for the centring and the norm bound
// the caller vouches for) is split EXACTLY into three bf16 numbers x' = h + m + l (8 significant
// bits each: h = bf16(x'), m = bf16(x' - h), l = bf16(x' - h - m); the remainder is below 2^-24
// |x'|).  The dot product is the sum of the six partial products
//     h.h + (h.m + m.h) + (m.m + h.l + l.h)
// - products of bf16 numbers are exact in float32 and the matrix cores accumulate in float32; the
// three dropped terms (m.l, l.m, l.l) are below 2^-23 |x'||y'|, the rounding level of a float32 dot
// product.  d <= 128 (longer rows take the direct-difference kernels).  v_mfma_f32_16x16x32_bf16 retires 16 384 flop in 16 cycles per SIMD - 16 x the rate of
// v_mfma_f32_32x32x2_f32 (4 096 flop in 64 cycles) - so six of them still cost 2.7 x less matrix-
// core time than the float32 formulation (rk_kernelmfma.hip), and a d <= 32 system needs one k-step.
//
// Prologue (gram_split_kernel): both point sets are gathered, centred, scaled, split and written
// once per call as 64-point "images" - per image 3 splits x 64 points x 64 features (zero padded)
// of bf16 = 24 KiB, rows XOR-swizzled in 16-byte chunks so that a wave's ds_read_b128 of an MFMA
// operand (16 points x 8 features per lane group) is bank-conflict free - together with |x'|^2.
// Main kernel: workgroup = 4 waves, each owns 32 points of X1 (two 16-wide blocks) whose B
// operands stay in 48 registers for the whole kernel; the workgroup walks 64-point tiles of X2
// through a double-buffered LDS image (register-staged prefetch two tiles ahead, one barrier per
// tile).  Per tile and wave: S^T = X2 X1^T (4 x 2 blocks of 16 x 16, 6 MFMAs per block and
// k-step), P = kappa(|x|^2 + |y|^2 - 2 S) on the accumulators, then Y^T += V^T P^T either on the
// VALU (k <= 4) or on the float32 matrix cores (v_mfma_f32_16x16x4_f32: P is the B operand as it
// sits in the accumulators - element v of a lane is P[j = 4 (lane / 16) + v][i = lane % 16] - and
// 16 right-hand sides fill the instruction).  No cross-wave reduction: a wave owns its i-range.
#include <stdlib.h>

#include "rk_common.h"
#include "rk_kerneltile.h"

namespace rk {

// RK_BX_F16X2 = 1 (default): TWO float16 pieces per coordinate instead of three bfloat16 ones.  The scaled,
// centred coordinates obey |x'|^2 <= 32 (the norm bound), so x' 2^10 sits in float16's normal range; with
// h = f16(x' 2^10), m = f16(x' 2^10 - h) the remainder is below 2^-22 |x'| and the dot product is the sum of
// THREE exact partial products h.h + h.m + m.h (the dropped m.m is below 2^-22 |x'||y'|) - half the matrix-core
// work of the six bfloat16 products, for twice their (already negligible) contraction error: 2^-22 instead of
// 2^-23 relative to |x'||y'|, against the 2^-24 (|x'|^2 + |y'|^2) that the float32 norms carry anyway.
// Measured (n = 200 000, d = 64, k = 1): see DESIGN.md.  RK_BX_F16X2 = 0 builds the bfloat16 x 3 kernel.
#ifndef RK_BX_F16X2
#define RK_BX_F16X2 1
#endif
constexpr int BX_NSPLIT = RK_BX_F16X2 ? 2 : 3;  // pieces per coordinate
constexpr float BX_PRESCALE = RK_BX_F16X2 ? 1024.f : 1.f;              // pieces hold x' times this
constexpr float BX_UNSCALE = RK_BX_F16X2 ? (1.f / 1048576.f) : 1.f;    // accumulated dot products times this = x'.y'
constexpr int BX_TI = 128;                      // X1 points per workgroup (4 waves x 32)
constexpr int BX_SPLIT = 8192;                  // bytes of one split of one image
constexpr int BX_IMG = BX_NSPLIT * BX_SPLIT;    // bytes of one image (16 KiB / 24 KiB)
constexpr int BX_LDV = 20;                      // stride of the V tile rows (conflict-free A operand reads)
constexpr int BX_VROWS = 64;                    // rows of the V / |y|^2 tiles in LDS (largest tile)
constexpr int BX_FOLD_COLS = 2048;              // columns between folds of the running sums

// Image geometry: FEAT = 64 features per point (d <= 64: 64 points per image = X2 points per tile) or
// 128 (d <= 128: 32 points per image), either way 24 KiB per image.  A point's split is one row of
// FEAT bf16; its 16-byte chunks are XOR-swizzled with the row index so that the 16 rows of a
// ds_read_b128 lane group fall into 16 distinct 16-byte slots of the 256-byte bank row.
template <int FEAT>
struct BxGeo {
    static constexpr int TJ = BX_SPLIT / (FEAT * 2);  // points per image: 64 or 32
    static constexpr int ROWB = FEAT * 2;             // bytes of one point in one split
    static constexpr int CPR = FEAT / 8;              // 16-byte chunks per row: 8 or 16
    static constexpr int JB = TJ / 16;                // 16-point block rows per tile
    __host__ __device__ static int chunk(int r, int c) {
        return r * ROWB + ((c ^ (FEAT == 64 ? ((r >> 1) & 7) : (r & 15))) << 4);
    }
};

#if RK_BX_F16X2
typedef _Float16 bx_frag __attribute__((ext_vector_type(8)));
#else
typedef __bf16 bx_frag __attribute__((ext_vector_type(8)));
#endif
typedef float bx_acc __attribute__((ext_vector_type(4)));
typedef unsigned int bx_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bx_bf16_rn(float x) {  // round to nearest even, finite x
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// One thread per (point, 8-feature chunk): gather, centre, scale, split, swizzle, |x'|^2.
// Points n .. npad - 1 (padding of the last image) are written as zeros; `norms` has npad entries.
template <int FEAT>
__global__ void __launch_bounds__(256)
gram_split_kernel(const float* __restrict__ X, int64_t ldx, const int64_t* __restrict__ idx, int64_t n,
                  int64_t npad, int64_t d, const float* __restrict__ iw, const float* __restrict__ muw,
                  unsigned char* __restrict__ img, float* __restrict__ norms) {
    using G = BxGeo<FEAT>;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p = g / G::CPR;
    const int c = (int)(g % G::CPR);
    if (p >= npad) return;  // (whole groups of CPR lanes: npad * CPR is a multiple of 64)
    const bool live = p < n;
    const float* row = live ? X + (idx ? idx[p] : p) * ldx : nullptr;
    unsigned hs[8], ms[8], ls[8];
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int f = c * 8 + u;
        const float x = (live && f < d) ? fmaf(row[f], iw[f], -muw[f]) : 0.f;
        ss += x * x;
#if RK_BX_F16X2
        const float xs = x * BX_PRESCALE;
        const _Float16 h = (_Float16)xs;          // round to nearest even
        const float r1 = xs - (float)h;           // exact
        const _Float16 m = (_Float16)r1;
        hs[u] = __builtin_bit_cast(unsigned short, h); ms[u] = __builtin_bit_cast(unsigned short, m); ls[u] = 0;
#else
        const unsigned h = bx_bf16_rn(x);
        const float r1 = x - __uint_as_float(h << 16);   // exact
        const unsigned m = bx_bf16_rn(r1);
        const float r2 = r1 - __uint_as_float(m << 16);  // exact
        hs[u] = h; ms[u] = m; ls[u] = bx_bf16_rn(r2);
#endif
    }
#pragma unroll
    for (int sh = 1; sh < G::CPR; sh <<= 1) ss += __shfl_xor(ss, sh);
    // padding points get a huge |x'|^2: their kernel value is exactly 0 whatever V holds there
    if (c == 0) norms[p] = live ? ss : 1e24f;
    const int r = (int)(p % G::TJ);
    unsigned char* dst = img + (p / G::TJ) * BX_IMG + G::chunk(r, c);
    auto pack = [](const unsigned* e) {
        bx_u4 o;
        o.x = e[0] | (e[1] << 16); o.y = e[2] | (e[3] << 16); o.z = e[4] | (e[5] << 16); o.w = e[6] | (e[7] << 16);
        return o;
    };
    *reinterpret_cast<bx_u4*>(dst) = pack(hs);
    *reinterpret_cast<bx_u4*>(dst + BX_SPLIT) = pack(ms);
    if constexpr (BX_NSPLIT == 3) *reinterpret_cast<bx_u4*>(dst + 2 * BX_SPLIT) = pack(ls);
}

// ---- RK_BX_F16X2, k > 4: the SECOND product Y^T += V^T P^T on the 16-bit matrix cores too ------------------
// (it was 4 v_mfma_f32_16x16x4_f32 = 128 matrix-core cycles per 16 x 16 block of P - more than the three float16
// products of the first one.)  v_mfma_f32_16x16x16_f16 takes P as it sits in the accumulators - element v of a lane
// is P[j = 4 (lane / 16) + v][i = lane % 16], exactly its B operand (k = 4 (lane / 16) + v) - after a split into
// two float16 pieces (P 2^10 = h + m, kernel values are <= 1), and V^T as A operand from a pre-split image:
// per X2 tile [piece][rhs 16][TJ points] float16, every column scaled by a power of two from its largest entry
// (found on the device), rows padded so that the 8-byte operand reads are bank-conflict free.  Three products
// (m.h + h.m + h.h) of 16 cycles per block instead of four of 32.
constexpr int bx_vrowb(int tj) { return tj == 64 ? 136 : 80; }           // bytes of one rhs row of a V tile
constexpr int bx_vtile_bytes(int tj) { return 2 * 16 * bx_vrowb(tj); }   // 4 352 / 2 560
constexpr float BX_PSCALE = 1024.f;
typedef _Float16 bx_h4 __attribute__((ext_vector_type(4)));
typedef unsigned int bx_u2 __attribute__((ext_vector_type(2)));

// vinfo: [16 absmax bits][16 x 2^eb][16 x 2^-(eb + 10)]
__global__ void __launch_bounds__(256)
bx_v_absmax_kernel(const float* __restrict__ V, int64_t n2, int k, int64_t ldv, unsigned* __restrict__ vinfo) {
    // thread (column c = tid & 15, row group tid >> 4); one atomic per column and WORKGROUP (a per-thread atomic on
    // 10 addresses serialised: 167 us for a 500 000 x 10 block)
    __shared__ float red[16][16];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    float m = 0.f;
    if (c < k) {
        for (int64_t r = (int64_t)blockIdx.x * 16 + g; r < n2; r += (int64_t)gridDim.x * 16) {
            const float v = fabsf(V[r * ldv + c]);
            m = (v == v && v > m) ? v : m;
        }
    }
    red[g][c] = m;
    __syncthreads();
    if (threadIdx.x < 16 && threadIdx.x < k) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t = red[q][threadIdx.x] > t ? red[q][threadIdx.x] : t;
        if (t > 0.f) atomicMax(vinfo + threadIdx.x, __float_as_uint(t));
    }
}
__global__ void bx_v_scale_kernel(unsigned* __restrict__ vinfo) {
    const int c = threadIdx.x;
    if (c >= 16) return;
    int eb = 0;
    const float m = __uint_as_float(vinfo[c]);
    if (m > 0.f && m < 3.0e38f) {
        int ex;
        frexpf(m, &ex);
        eb = 14 - ex;
        eb = eb > 100 ? 100 : (eb < -100 ? -100 : eb);
    }
    vinfo[16 + c] = __float_as_uint(ldexpf(1.f, eb));
    vinfo[32 + c] = __float_as_uint(ldexpf(1.f, -eb) / BX_PSCALE);
}
// one thread per (point-quad, rhs): 4 consecutive points of one column -> 8 bytes per piece
template <int TJ>
__global__ void __launch_bounds__(256)
bx_v_split_kernel(const float* __restrict__ V, int64_t n2, int64_t n2pad, int k, int64_t ldv,
                  const unsigned* __restrict__ vinfo, unsigned char* __restrict__ vimg) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(g & 15);
    const int64_t quad = g >> 4;
    if (quad >= n2pad / 4) return;
    const float sb = __uint_as_float(vinfo[16 + c]);
    unsigned short h[4], m[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t j = quad * 4 + e;
        const float x = (j < n2 && c < k) ? V[j * ldv + c] * sb : 0.f;
        const _Float16 hh = (_Float16)x;
        const _Float16 mm = (_Float16)(x - (float)hh);
        h[e] = __builtin_bit_cast(unsigned short, hh);
        m[e] = __builtin_bit_cast(unsigned short, mm);
    }
    const int64_t tile = (quad * 4) / TJ;
    const int jl = (int)((quad * 4) % TJ);
    unsigned char* dst = vimg + tile * bx_vtile_bytes(TJ) + c * bx_vrowb(TJ) + jl * 2;
    bx_u2 oh, om;
    oh.x = h[0] | ((unsigned)h[1] << 16); oh.y = h[2] | ((unsigned)h[3] << 16);
    om.x = m[0] | ((unsigned)m[1] << 16); om.y = m[2] | ((unsigned)m[3] << 16);
    *reinterpret_cast<bx_u2*>(dst) = oh;
    *reinterpret_cast<bx_u2*>(dst + 16 * bx_vrowb(TJ)) = om;
}
static size_t bx_vimg_bytes(int64_t n2) {  // vinfo (256 B) + the split image of V (64-point tiles: 68 B per point)
    return 256 + (size_t)(round_up(n2, 64) / 32) * bx_vtile_bytes(32);
}

// k <= 4 (full evaluation): a 4-column V tile (50.5 KiB of LDS) and 138 registers admit 3 workgroups per
// CU: +2 % at d = 64, +4.5 % at d = 32 (tools/ab.py --kernel kmatmat), bitwise identical results.  (The
// symmetric evaluation at 3 per CU - k = 1 fits the LDS - spills 10 registers: 19.5 instead of 18.1 ms.)
#ifndef RK_BX_WG3
#define RK_BX_WG3 1
#endif
#ifndef RK_BX_SYM_WG3
#define RK_BX_SYM_WG3 0  // (measured per build: see below)
#endif
constexpr int bx_ldv(int kv) { return (RK_BX_WG3 && kv > 0) ? 4 : BX_LDV; }
constexpr size_t bx_smem_bytes(int kv) {
    return 2 * BX_IMG + 2 * BX_VROWS * bx_ldv(kv) * sizeof(float) + 2 * BX_VROWS * sizeof(float);
}
constexpr size_t bx_sym_smem_bytes(int kv) {  // + column sums [2][4 waves][64][kv] + 64 dummy slots
    return bx_smem_bytes(kv) + (size_t)(2 * 4 * 64 * kv + 64) * sizeof(float);
}

// KV = 0: second product on the float32 matrix cores (k <= 16); KV = 1 / 2 / 4: on the VALU (k <= KV)
// NKS = k-steps of 32 features (1: d <= 32, 2: d <= 64, 4: d <= 128 - half-height tiles)
// (A hand-ordered instruction stream - operand reads one step ahead, kappa of block row jb - 1 placed in
// the shadow of the MFMAs of block row jb with sched_barrier - was 3 - 11 % SLOWER than the compiler's
// order: two workgroups per CU already overlap one wave's kappa with the other's MFMAs.)
// SYM (X1 is X2, k <= 4): only the tiles on and above the diagonal are evaluated; a tile beyond the
// diagonal 128 x 128 block feeds y[i] += P v[j] AND y[j] += P^T v[i] (the column sums of the tile:
// per lane 2 FMAs and a 16-lane DPP reduction per row of P, added over the 4 waves through LDS and
// written to a (work item, tile) table).  Work items (row block, tile segment) are listed by
// bx_sym_items_kernel; a finishing kernel adds, in a fixed order, the row parts of a row block's
// items and the column parts of every row block above.  Half the matrix-core and exponential work.
struct BxSym {
    const int* item_rb;        // row block of an item
    const int64_t* item_tlo;   // its first tile
    const int64_t* item_coff;  // tiles before it in the column table
    int64_t segt;              // tiles per item
    float* prow;               // [item][128][KV]
    float* pcol;               // [column-table tile][64][KV]
};

// KIND: the kernel function at compile time (with a run-time switch in the tile loop the Matern
// instantiations spilled up to 280 bytes per lane and ran 22 - 93 % slower)
template <int KIND, int KV, int NKS, bool SYM>
__global__ void __launch_bounds__(256, ((RK_BX_WG3 && (KV > 0 || RK_BX_F16X2) && (!SYM || RK_BX_SYM_WG3)) ? 3 : 2))
kmat_matmat_bf16x3_kernel(int kind, const unsigned char* __restrict__ img1, const unsigned char* __restrict__ img2,
                          int64_t n1, int64_t n2, float w, float scale, const float* __restrict__ nx,
                          const float* __restrict__ ny, const float* __restrict__ V, int k, int64_t ldv,
                          int64_t tiles_per_seg, float* __restrict__ part, float* __restrict__ Y, int64_t ldy,
                          int accumulate, BxSym sym, const unsigned char* __restrict__ vimg,
                          const unsigned* __restrict__ vinfo) {
    using G = BxGeo<(NKS <= 2 ? 64 : 128)>;
    constexpr int TJ = G::TJ;
    constexpr bool RBF = KIND == RK_RBF;
    constexpr bool VH = RK_BX_F16X2 && KV == 0;   // second product on the 16-bit matrix cores (V from its split image)
    constexpr int VROWB = bx_vrowb(TJ);
    constexpr int VCH = bx_vtile_bytes(TJ) / 16;   // 16-byte chunks of a V tile: 272 / 160
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* XB = smem;                                                  // [2][BX_IMG]
    constexpr int LDV = bx_ldv(KV);
    float* VS = reinterpret_cast<float*>(smem + 2 * BX_IMG);                   // [2][BX_VROWS][LDV]
    float* NY = VS + 2 * BX_VROWS * LDV;                                       // [2][BX_VROWS]
    float* CS = NY + 2 * BX_VROWS;                                             // SYM: [2][4][64][KV] + dummy slots
    const float rbf_c = -0.5f * w * w * 1.4426950408889634f;  // exp(-s w^2 / 2) = exp2(rbf_c s)
    const float m2c = -2.f * rbf_c * BX_UNSCALE;  // (the accumulators hold the dot product of the pre-scaled pieces)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t ntile = (n2 + TJ - 1) / TJ;
    const int64_t rblk = SYM ? (int64_t)sym.item_rb[blockIdx.x] : (int64_t)blockIdx.x;
    const int64_t i0 = rblk * BX_TI;
    const int64_t seg_len = SYM ? sym.segt : tiles_per_seg;
    const int64_t t_lo = SYM ? sym.item_tlo[blockIdx.x] : (int64_t)blockIdx.y * tiles_per_seg;
    const int64_t t_hi = (t_lo + seg_len < ntile) ? t_lo + seg_len : ntile;
    const int64_t t_sym = 2 * rblk + 2;  // SYM: tiles from here on lie beyond the diagonal block

    // ---- this wave's X1 operands: 2 blocks of 16 points x 2 k-steps x 3 splits, resident ----------
    bx_frag bh[2][NKS], bm[2][NKS];
#if !RK_BX_F16X2
    bx_frag bl[2][NKS];
#endif
    float nxr[2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
        const int64_t ip = i0 + wave * 32 + ib * 16 + lr;  // < round_up(n1, 128): the images exist
        const int r = (int)(ip % TJ);
        const unsigned char* base = img1 + (ip / TJ) * BX_IMG;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const unsigned char* src = base + G::chunk(r, 4 * ks + lq);
            bh[ib][ks] = __builtin_bit_cast(bx_frag, *reinterpret_cast<const bx_u4*>(src));
            bm[ib][ks] = __builtin_bit_cast(bx_frag, *reinterpret_cast<const bx_u4*>(src + BX_SPLIT));
#if !RK_BX_F16X2
            bl[ib][ks] = __builtin_bit_cast(bx_frag, *reinterpret_cast<const bx_u4*>(src + 2 * BX_SPLIT));
#endif
        }
        // (nx has round_up(n1, 128) entries.)  RBF: the exponent is formed directly in base 2,
        // exp2(rbf_c |x|^2 + rbf_c |y|^2 - 2 rbf_c S): the norms are kept pre-multiplied by rbf_c
        nxr[ib] = nx[ip] * (RBF ? rbf_c : 1.f);
        if constexpr (RBF && RK_BX_F16X2 && KV == 0) nxr[ib] += 10.f;  // the kernel values come out times 2^10 (see the second product)
    }

    constexpr int KVS = (SYM && KV > 0) ? KV : 1;
    float vi[2][KVS];  // SYM: V of this lane's two X1 points (B side of the column sums)
    if constexpr (SYM) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int64_t ip = i0 + wave * 32 + ib * 16 + lr;
#pragma unroll
            for (int c = 0; c < KVS; ++c) vi[ib][c] = (ip < n1 && c < k) ? V[ip * ldv + c] : 0.f;
        }
    }

    // ---- tile staging: registers -> LDS (the image is copied linearly: it is stored swizzled) -------
    constexpr int NPRE = BX_IMG / 16 / 256;  // 16-byte chunks of an image per thread: 4 / 6
    bx_u4 pre[NPRE];
    float vr[4], nyr;
    bx_u4 vq[2];  // VH: this thread's 16-byte chunks of the V tile
    // Every load is unconditional (a conditional load makes the compiler wait for the whole
    // prefetch at the branch join, in front of the arithmetic): rows past n2 re-read row n2 - 1 and
    // columns past k re-read column k - 1 - finite stand-ins that never reach the output, because the
    // padding points carry |y'|^2 = 1e24 (kernel value exactly 0) and columns >= k are not written.
    auto load_tile = [&](int64_t t) {
        const bx_u4* src = reinterpret_cast<const bx_u4*>(img2 + t * BX_IMG);
#pragma unroll
        for (int q = 0; q < NPRE; ++q) pre[q] = src[tid + 256 * q];
        if constexpr (VH) {
            const bx_u4* vsrc = reinterpret_cast<const bx_u4*>(vimg + t * bx_vtile_bytes(TJ));
            vq[0] = vsrc[tid < VCH ? tid : VCH - 1];
            if constexpr (VCH > 256) vq[1] = vsrc[256 + (tid < VCH - 256 ? tid : 0)];
        } else {
            const int64_t j = t * TJ + ((tid >> 2) & (TJ - 1));
            const float* vrow = V + (j < n2 ? j : n2 - 1) * ldv;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = (tid & 3) * 4 + u;
                vr[u] = vrow[c < k ? c : k - 1];
            }
        }
        nyr = ny[t * TJ + (tid & (TJ - 1))];  // (ny has round_up(n2, 64) entries)
    };
    auto store_tile = [&](int buf) {
        bx_u4* dst = reinterpret_cast<bx_u4*>(XB + buf * BX_IMG);
#pragma unroll
        for (int q = 0; q < NPRE; ++q) dst[tid + 256 * q] = pre[q];
        if constexpr (VH) {
            bx_u4* vdst = reinterpret_cast<bx_u4*>(VS + buf * BX_VROWS * LDV);  // (5 120 B per buffer >= the tile)
            if (tid < VCH) vdst[tid] = vq[0];
            if constexpr (VCH > 256) {
                if (tid < VCH - 256) vdst[256 + tid] = vq[1];
            }
        } else if constexpr (LDV == 4) {  // only columns 0 .. 3 are kept (k <= 4)
            if ((tid & 3) == 0 && tid < 4 * TJ)
                *reinterpret_cast<float4*>(VS + buf * BX_VROWS * LDV + (tid >> 2) * LDV) = make_float4(vr[0], vr[1], vr[2], vr[3]);
        } else {
            float* vs = VS + buf * BX_VROWS * LDV + (tid >> 2) * LDV + (tid & 3) * 4;
            if (tid < 4 * TJ) *reinterpret_cast<float4*>(vs) = make_float4(vr[0], vr[1], vr[2], vr[3]);
        }
        if (tid < TJ) NY[buf * BX_VROWS + tid] = nyr * (RBF ? rbf_c : 1.f);
    };

    // running sums, two-level (folded every BX_FOLD_COLS columns)
    bx_acc yacc[2], ytot[2];
    constexpr int KVN = KV > 0 ? KV : 1;
    float yv[2][KVN], yvt[2][KVN];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
        for (int v = 0; v < 4; ++v) { yacc[ib][v] = 0.f; ytot[ib][v] = 0.f; }
#pragma unroll
        for (int c = 0; c < KVN; ++c) yv[ib][c] = yvt[ib][c] = 0.f;
    }

    if (t_lo >= t_hi) return;  // (uniform; the launch plan never produces an empty segment)
    const int64_t t_last = t_hi - 1;
    load_tile(t_lo);
    store_tile(0);
    load_tile(t_lo + 1 < t_hi ? t_lo + 1 : t_last);
    __syncthreads();
    for (int64_t t = t_lo; t < t_hi; ++t) {
        const int buf = (int)((t - t_lo) & 1);
        // tile t + 1 (loaded one iteration ago) -> the other buffer; tile t + 2 in flight during this
        // tile's arithmetic.  Past the end the last tile is staged again (unused): no branches, the
        // body is one basic block and the scheduler can mix matrix-core and vector work
        store_tile(buf ^ 1);
        load_tile(t + 2 < t_hi ? t + 2 : t_last);
        if constexpr (SYM) {  // column sums of tile t - 1 (complete since the last barrier): 4 waves -> table
            if (t > t_lo && t - 1 >= t_sym && tid < 64 * KVS) {
                const float* cs = CS + (buf ^ 1) * 4 * 64 * KVS;
                const float sum = ((cs[tid] + cs[64 * KVS + tid]) + cs[2 * 64 * KVS + tid]) + cs[3 * 64 * KVS + tid];
                __builtin_nontemporal_store(sum, &sym.pcol[(sym.item_coff[blockIdx.x] + (t - 1 - t_lo)) * 64 * KVS + tid]);
            }
        }
        const bool sym_tile = SYM && t >= t_sym;
        float* csw = CS + (buf * 4 + wave) * 64 * KVS;
        const unsigned char* xb = XB + buf * BX_IMG;
        const float* vsb = VS + buf * BX_VROWS * LDV;
        const float* nyb = NY + buf * BX_VROWS;
#pragma unroll
        for (int jb = 0; jb < G::JB; ++jb) {
            // ---- S^T block row jb: 16 points of X2 against this wave's 2 x 16 points of X1 -------
            bx_acc acc[2];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[ib][v] = 0.f;
            const int r = jb * 16 + lr;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const unsigned char* src = xb + G::chunk(r, 4 * ks + lq);
                const bx_frag ah = *reinterpret_cast<const bx_frag*>(src);
                const bx_frag am = *reinterpret_cast<const bx_frag*>(src + BX_SPLIT);
#if RK_BX_F16X2
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {  // small terms first
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bh[ib][ks], acc[ib], 0, 0, 0);
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bm[ib][ks], acc[ib], 0, 0, 0);
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ib][ks], acc[ib], 0, 0, 0);
                }
#else
                const bx_frag al = *reinterpret_cast<const bx_frag*>(src + 2 * BX_SPLIT);
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {  // small terms first
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[ib][ks], acc[ib], 0, 0, 0);
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[ib][ks], acc[ib], 0, 0, 0);
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[ib][ks], acc[ib], 0, 0, 0);
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[ib][ks], acc[ib], 0, 0, 0);
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[ib][ks], acc[ib], 0, 0, 0);
                    acc[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[ib][ks], acc[ib], 0, 0, 0);
                }
#endif
            }
            // ---- P = kappa(|x|^2 + |y|^2 - 2 S); Y^T += V^T P^T ------------------------------------
            const float4 ny4 = *reinterpret_cast<const float4*>(nyb + jb * 16 + 4 * lq);
            const float nyv[4] = {ny4.x, ny4.y, ny4.z, ny4.w};
            float qv[4][KVS];  // SYM: this lane's share p[0] v_i0 + p[1] v_i1 of the column sums of rows v = 0 .. 3
            float pb[2][4];  // VH: the block's kernel values, for the split after the loop
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float p[2];
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    if constexpr (RBF) {
                        // add, fma, exp2 per pair (no clamp: a squared distance that rounds to
                        // -1e-6 gives 1 + 7e-7 instead of 1)
                        p[ib] = __builtin_amdgcn_exp2f(fmaf(acc[ib][v], m2c, nxr[ib] + nyv[v]));
                    } else {
                        float s = nxr[ib] + nyv[v] - (2.f * BX_UNSCALE) * acc[ib][v];
                        s = s > 0.f ? s : 0.f;
                        p[ib] = kernel_value<float>(KIND, s, w);
                    }
                }
                const float* vj = vsb + (jb * 16 + 4 * lq + v) * LDV;
                if constexpr (VH) {
                    pb[0][v] = p[0];
                    pb[1][v] = p[1];
                } else if constexpr (KV == 0) {
                    const float av = vj[lr];  // A operand: V[j = 4 (lane / 16) + v][rhs = lane % 16]
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
                        yacc[ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, p[ib], yacc[ib], 0, 0, 0);
                } else {
                    float vv[KV];
                    if constexpr (KV == 4) {
                        const float4 q = *reinterpret_cast<const float4*>(vj);
                        vv[0] = q.x; vv[1] = q.y; vv[2] = q.z; vv[3] = q.w;
                    } else {
#pragma unroll
                        for (int c = 0; c < KV; ++c) vv[c] = vj[c];
                    }
#pragma unroll
                    for (int c = 0; c < KV; ++c)
#pragma unroll
                        for (int ib = 0; ib < 2; ++ib) yv[ib][c] += p[ib] * vv[c];
                    if constexpr (SYM) {
#pragma unroll
                        for (int c = 0; c < KV; ++c) qv[v][c] = fmaf(p[0], vi[0][c], p[1] * vi[1][c]);
                    }
                }
            }
            if constexpr (VH) {
                // A = V^T pieces: rhs lr, points jb * 16 + 4 lq .. + 4 (8 bytes per piece)
                const unsigned char* va = reinterpret_cast<const unsigned char*>(vsb) + lr * VROWB + (jb * 16 + 4 * lq) * 2;
                const bx_h4 vh = __builtin_bit_cast(bx_h4, *reinterpret_cast<const bx_u2*>(va));
                const bx_h4 vm = __builtin_bit_cast(bx_h4, *reinterpret_cast<const bx_u2*>(va + 16 * VROWB));
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    // split of P 2^10 (RBF: the factor rides in the exponent) into two float16 pieces, two values per
                    // conversion: h = toward-zero rounding (v_cvt_pkrtz_f16_f32), m = the exact remainder rounded the
                    // same way: P 2^10 = h + m + O(2^-21 P)
                    typedef __fp16 bx_p2 __attribute__((ext_vector_type(2)));
                    bx_u2 phu, pmu;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float x0 = RBF ? pb[ib][2 * e] : pb[ib][2 * e] * BX_PSCALE;
                        const float x1 = RBF ? pb[ib][2 * e + 1] : pb[ib][2 * e + 1] * BX_PSCALE;
                        const bx_p2 h2 = __builtin_amdgcn_cvt_pkrtz(x0, x1);
                        const bx_p2 m2 = __builtin_amdgcn_cvt_pkrtz(x0 - (float)h2[0], x1 - (float)h2[1]);
                        if (e == 0) { phu.x = __builtin_bit_cast(unsigned, h2); pmu.x = __builtin_bit_cast(unsigned, m2); }
                        else { phu.y = __builtin_bit_cast(unsigned, h2); pmu.y = __builtin_bit_cast(unsigned, m2); }
                    }
                    const bx_h4 ph = __builtin_bit_cast(bx_h4, phu), pm = __builtin_bit_cast(bx_h4, pmu);
                    yacc[ib] = __builtin_amdgcn_mfma_f32_16x16x16f16(vm, ph, yacc[ib], 0, 0, 0);
                    yacc[ib] = __builtin_amdgcn_mfma_f32_16x16x16f16(vh, pm, yacc[ib], 0, 0, 0);
                    yacc[ib] = __builtin_amdgcn_mfma_f32_16x16x16f16(vh, ph, yacc[ib], 0, 0, 0);
                }
            }
            if constexpr (SYM && KV > 0) {
                if (sym_tile) {  // (tile-uniform)
                    // 4 values per lane x 16 lanes -> 4 sums: a reduce-scatter butterfly (lane pairs trade
                    // two of their four values, then one of two) and two in-row rotations: 11 vector
                    // instructions per right-hand side instead of 16 for four separate 16-lane sums.
                    // Afterwards lane lr holds the sum of row v = 2 (lr & 1) + ((lr >> 1) & 1).
                    const bool b0 = lr & 1, b1 = lr & 2;
#pragma unroll
                    for (int c = 0; c < KV; ++c) {
                        const float k0 = b0 ? qv[2][c] : qv[0][c], k1 = b0 ? qv[3][c] : qv[1][c];
                        const float s0 = b0 ? qv[0][c] : qv[2][c], s1 = b0 ? qv[1][c] : qv[3][c];
                        const float r0 = k0 + dpp_fetch<0xB1, 0xF>(s0);  // quad_perm [1,0,3,2]: lane ^ 1
                        const float r1 = k1 + dpp_fetch<0xB1, 0xF>(s1);
                        const float kk = b1 ? r1 : r0, ss = b1 ? r0 : r1;
                        float u = kk + dpp_fetch<0x4E, 0xF>(ss);         // quad_perm [2,3,0,1]: lane ^ 2
                        u += dpp_fetch<0x124, 0xF>(u);                   // row_ror:4
                        u += dpp_fetch<0x128, 0xF>(u);                   // row_ror:8: every lane, its row's total
                        // lanes lr < 4 store (the others into dummy slots: one unconditional ds_write)
                        const int vrow = 2 * (lr & 1) + ((lr >> 1) & 1);
                        float* dst = lr < 4 ? csw + (jb * 16 + 4 * lq + vrow) * KVS + c : CS + 2 * 4 * 64 * KVS + lane;
                        *dst = u;
                    }
                }
            }
        }
        // ask the scheduler to alternate matrix-core and vector instructions over the tile's basic block
        // (the block rows are independent: +2 - 3 % measured; writing the software pipeline out by hand -
        // the MFMAs of row jb + 1 ahead of the vector half of row jb - gave the same and is not kept)
        // (not for d > 64: there the hint costs registers the three-workgroup variants do not have)
        if constexpr (NKS <= 2) {
#pragma unroll
            // (three float16 products per block instead of six: half as many matrix-core instructions per vector one)
            for (int q = 0; q < (RK_BX_F16X2 ? 64 : 128); ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, RK_BX_F16X2 ? 3 : 1, 0);
            }
        }
        if (((t - t_lo) & (BX_FOLD_COLS / TJ - 1)) == BX_FOLD_COLS / TJ - 1) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                ytot[ib] += yacc[ib];
#pragma unroll
                for (int v = 0; v < 4; ++v) yacc[ib][v] = 0.f;
#pragma unroll
                for (int c = 0; c < KVN; ++c) { yvt[ib][c] += yv[ib][c]; yv[ib][c] = 0.f; }
            }
        }
        __syncthreads();  // tile t + 1 is in LDS; every wave is done with tile t's buffer
    }
    if constexpr (SYM) {  // column sums of the last tile
        if (t_last >= t_sym && tid < 64 * KVS) {
            const float* cs = CS + (int)((t_last - t_lo) & 1) * 4 * 64 * KVS;
            const float sum = ((cs[tid] + cs[64 * KVS + tid]) + cs[2 * 64 * KVS + tid]) + cs[3 * 64 * KVS + tid];
            __builtin_nontemporal_store(sum, &sym.pcol[(sym.item_coff[blockIdx.x] + (t_last - t_lo)) * 64 * KVS + tid]);
        }
    }
    // ---- this wave's 32 rows of Y: through LDS as [i (128)][rhs (16)], then coalesced out ---------------
    float* yl = reinterpret_cast<float*>(XB);  // (all reads of the last tile are behind the barrier above)
    if constexpr (KV == 0) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int v = 0; v < 4; ++v)  // element v: rhs 4 (lane / 16) + v, i = lane % 16
                yl[(wave * 32 + ib * 16 + lr) * 16 + 4 * lq + v] = ytot[ib][v] + yacc[ib][v];
    } else {
        // the 4 lane groups hold partial sums over their j's: [i][rhs][group]
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int c = 0; c < KV; ++c)
                yl[((wave * 32 + ib * 16 + lr) * KV + c) * 4 + lq] = yvt[ib][c] + yv[ib][c];
    }
    __syncthreads();
    for (int e = tid; e < BX_TI * k; e += 256) {
        const int il = e / k, c = e - il * k;
        float sum;
        if constexpr (KV == 0) {
            sum = yl[il * 16 + c];
            if constexpr (VH) sum *= __uint_as_float(vinfo[32 + c]);  // undo the column's scale and the 2^10 of P
        } else {
            const float4 q = *reinterpret_cast<const float4*>(yl + (il * KV + c) * 4);
            sum = (q.x + q.y) + (q.z + q.w);
        }
        const int64_t i = i0 + il;
        if constexpr (SYM) {
            sym.prow[((int64_t)blockIdx.x * BX_TI + il) * KVS + c] = sum;  // (rows past n1 are never read)
            continue;
        }
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

size_t bx_aux_bytes(int64_t n1, int64_t n2);
size_t bx_x2_image_bytes(int64_t n2, int64_t d);

// ---- symmetric products: work list and finishing kernel ------------------------------------------------
// Row block rb (128 points) owns the 64-point tiles 2 rb .. T - 1, cut into items of `segt` tiles.
__global__ void bx_sym_items_kernel(int64_t T, int64_t nrb, int64_t segt, int* __restrict__ item_rb,
                                    int64_t* __restrict__ item_tlo, int64_t* __restrict__ item_coff,
                                    int64_t* __restrict__ rb_base) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t items = 0, coff = 0;
    for (int64_t rb = 0; rb < nrb; ++rb) {
        rb_base[rb] = items;
        for (int64_t t = 2 * rb; t < T; t += segt) {
            item_rb[items] = (int)rb;
            item_tlo[items] = t;
            item_coff[items] = coff;
            coff += (t + segt < T ? segt : T - t);
            ++items;
        }
    }
    rb_base[nrb] = items;
}
static void bx_sym_counts(int64_t n, int64_t segt, int64_t* items, int64_t* col_tiles) {
    const int64_t T = ceil_div(n, 64), nrb = ceil_div(n, 128);
    int64_t it = 0, ct = 0;
    for (int64_t rb = 0; rb < nrb; ++rb) {
        const int64_t nt = T - 2 * rb;
        it += ceil_div(nt, segt);
        ct += nt;
    }
    *items = it;
    *col_tiles = ct;
}
static int64_t bx_sym_segt(int64_t n) {
    const int64_t T = ceil_div(n, 64);
    int64_t segt = (T * T / 4) / 4096;  // ~4 096 items: 8 rounds of the 512 resident workgroups
    if (segt < 8) segt = 8;
    if (segt > 1024) segt = 1024;
    return segt;
}

// Y[r][c] = scale * (row parts of the items of r's row block + column parts of every row block above)
template <int KV>
__global__ void __launch_bounds__(256)
bx_sym_finish_kernel(int64_t n, int k, int64_t segt, const int64_t* __restrict__ rb_base,
                     const int64_t* __restrict__ item_coff, const float* __restrict__ prow,
                     const float* __restrict__ pcol, float scale, float* __restrict__ Y, int64_t ldy, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * k) return;
    const int64_t r = e / k;
    const int c = (int)(e - r * k);
    const int64_t rb = r >> 7, t = r >> 6;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t it = rb_base[rb], u = 0; it < rb_base[rb + 1]; ++it, ++u)
        s4[u & 3] += prow[(it * BX_TI + (r & 127)) * KV + c];
    const int64_t above = t >= 2 ? (t - 2) / 2 + 1 : 0;  // row blocks rb' with 2 rb' + 2 <= t
    for (int64_t b = 0; b < above; ++b) {
        const int64_t rel = t - 2 * b;
        const int64_t it = rb_base[b] + rel / segt;
        s4[b & 3] += pcol[((item_coff[it] + rel % segt) * 64 + (r & 63)) * KV + c];
    }
    const float o = scale * ((s4[0] + s4[1]) + (s4[2] + s4[3]));
    if (accumulate) Y[r * ldy + c] += o;
    else Y[r * ldy + c] = o;
}

size_t bx_sym_ws_bytes(int64_t n, int64_t k) {
    if (k < 1 || k > 4) return 0;
    const int kv = k <= 1 ? 1 : (k <= 2 ? 2 : 4);
    const int64_t segt = bx_sym_segt(n);
    int64_t items = 0, col_tiles = 0;
    bx_sym_counts(n, segt, &items, &col_tiles);
    const int64_t nrb = ceil_div(n, 128);
    size_t b = bx_aux_bytes(n, 0);                                   // map, norms, ONE set of images
    b += (size_t)round_up(items * (int64_t)(sizeof(int) + 2 * sizeof(int64_t)) + (nrb + 1) * (int64_t)sizeof(int64_t), 256);
    b += (size_t)round_up(items * BX_TI * kv * (int64_t)sizeof(float), 256);
    b += (size_t)col_tiles * 64 * kv * sizeof(float);
    return b;
}

// workspace behind the partial sums: affine map (2 x 128 floats), |x'|^2 of both sets (padded), the
// images (sized for the 128-feature geometry: 768 bytes per point)
static size_t bx_float_bytes(int64_t n1, int64_t n2) {
    return (size_t)round_up((2 * 128 + round_up(n1, BX_TI) + round_up(n2, 64)) * (int64_t)sizeof(float), 256);
}
static size_t bx_img_bytes(int64_t n1, int64_t n2) {
    return (size_t)(round_up(n1, BX_TI) / 32 + round_up(n2, 64) / 32) * BX_IMG;
}
size_t bx_aux_bytes(int64_t n1, int64_t n2) {  // floats | images of both point sets | vinfo + split image of V
    return bx_float_bytes(n1, n2) + bx_img_bytes(n1, n2) + bx_vimg_bytes(n2);
}
int64_t bx_tile_points(int64_t d) { return d <= 64 ? 64 : 32; }
// a caller-kept image of the X2 side of a product: |y'|^2 of the padded points, then their split images
size_t bx_x2_image_bytes(int64_t n2, int64_t d) {
    const int64_t n2pad = round_up(n2, 64);
    return (size_t)round_up(n2pad * (int64_t)sizeof(float), 256) + (size_t)(n2pad / bx_tile_points(d)) * BX_IMG;
}

void gram_prep_launch(const float* center, const float* ell_vec, int64_t d, float* iw, float* muw, hipStream_t st);

// the split images of a point set (d <= 64; 64 points per 16 KiB image) and |x'|^2, for other translation units
// (rk_symgemm.hip: the fused sketch product of matrix-free operators)
void gram_split64_launch(const float* X, int64_t ldx, const int64_t* idx, int64_t n, int64_t npad, int64_t d,
                         const float* iw, const float* muw, unsigned char* img, float* norms, hipStream_t st) {
    static_assert(RK_BX_F16X2 == 1 && BX_IMG == 16384 && BX_SPLIT == 8192, "rk_symgemm.hip reads this image format");
    hipLaunchKernelGGL(gram_split_kernel<64>, dim3((unsigned)ceil_div(npad * 8, 256)), dim3(256), 0, st, X, ldx, idx, n,
                       npad, d, iw, muw, img, norms);
}

int launch_kmatmat_bf16x3(int kind, const float* X1, int64_t n1, int64_t ldx1, const int64_t* idx1,
                          const float* X2, int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d, float w,
                          const float* ell_vec, const float* center, float scale, const float* V, int k,
                          int64_t ldv, int64_t S, int64_t tiles_per_seg, void* aux, float* part, float* Y,
                          int64_t ldy, int accumulate, void* x2_image, int x2_ready, hipStream_t st) {
    const int64_t tj = bx_tile_points(d);
    float* iw = static_cast<float*>(aux);
    float* muw = iw + 128;
    float* nxp = muw + 128;
    const int64_t n1pad = round_up(n1, BX_TI), n2pad = round_up(n2, 64);
    float* nyp = nxp + n1pad;
    unsigned char* img1 = static_cast<unsigned char*>(aux) + bx_float_bytes(n1, n2);
    unsigned char* img2 = img1 + (n1pad / tj) * BX_IMG;
    if (x2_image != nullptr) {  // caller-kept image of the X2 side: [ |y'|^2 (n2pad) ][ split images ]
        nyp = static_cast<float*>(x2_image);
        img2 = static_cast<unsigned char*>(x2_image) + round_up(n2pad * (int64_t)sizeof(float), 256);
    }
    const bool split2 = !(x2_image != nullptr && x2_ready);
    gram_prep_launch(center, ell_vec, d, iw, muw, st);
    if (tj == 64) {
        hipLaunchKernelGGL(gram_split_kernel<64>, dim3((unsigned)ceil_div(n1pad * 8, 256)), dim3(256), 0, st, X1,
                           ldx1, idx1, n1, n1pad, d, iw, muw, img1, nxp);
        if (split2)
            hipLaunchKernelGGL(gram_split_kernel<64>, dim3((unsigned)ceil_div(n2pad * 8, 256)), dim3(256), 0, st, X2,
                               ldx2, idx2, n2, n2pad, d, iw, muw, img2, nyp);
    } else {
        hipLaunchKernelGGL(gram_split_kernel<128>, dim3((unsigned)ceil_div(n1pad * 16, 256)), dim3(256), 0, st, X1,
                           ldx1, idx1, n1, n1pad, d, iw, muw, img1, nxp);
        if (split2)
            hipLaunchKernelGGL(gram_split_kernel<128>, dim3((unsigned)ceil_div(n2pad * 16, 256)), dim3(256), 0, st, X2,
                               ldx2, idx2, n2, n2pad, d, iw, muw, img2, nyp);
    }
    const int nks = d <= 32 ? 1 : (d <= 64 ? 2 : 4);
    // k > 4 (float16 build): the split image of V for the second product, behind the images of the point sets
    unsigned* vinfo = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(aux) + bx_float_bytes(n1, n2) +
                                                  bx_img_bytes(n1, n2));
    unsigned char* vimg = reinterpret_cast<unsigned char*>(vinfo) + 256;
    if (RK_BX_F16X2 && k > 4) {
        if (hipMemsetAsync(vinfo, 0, 256, st) != hipSuccess) return launch_status("rk_kernel_matmat(bf16x3 memset)");
        hipLaunchKernelGGL(bx_v_absmax_kernel, dim3(256), dim3(256), 0, st, V, n2, k, ldv, vinfo);
        hipLaunchKernelGGL(bx_v_scale_kernel, dim3(1), dim3(64), 0, st, vinfo);
        const dim3 vg((unsigned)ceil_div((n2pad / 4) * 16, 256));
        if (tj == 64)
            hipLaunchKernelGGL(bx_v_split_kernel<64>, vg, dim3(256), 0, st, V, n2, n2pad, k, ldv, vinfo, vimg);
        else
            hipLaunchKernelGGL(bx_v_split_kernel<32>, vg, dim3(256), 0, st, V, n2, n2pad, k, ldv, vinfo, vimg);
    }
    dim3 grid((unsigned)ceil_div(n1, BX_TI), (unsigned)S);
    // RK_BX_LDS_PAD: extra dynamic LDS per workgroup (bytes, < 20 000) - lowers the number of resident workgroups
    // per CU, which matters when this product shares the device with a latency-bound chain on another stream
    // A caller-kept X2 image marks the SAP row oracle (rk_kernel_matmat_cached): that product runs BESIDE the
    // step's latency-bound chain (block preconditioner, power iteration: ~100 small kernels on a high-priority
    // stream).  At three workgroups per CU it leaves the chain no slots until it has finished (measured: no chain
    // kernel starts inside the product, step 3.3 ms); at two the chain runs beside it (2.8 ms) although the
    // product itself is slower (1.8 instead of 1.6 ms) - the step is bound by the chain, not by the product.
    size_t lds_pad = x2_image != nullptr ? 12288 : 0;
    if (const char* e = getenv("RK_BX_LDS_PAD")) {
        const long v = atol(e);
        if (v >= 0 && v < 20000) lds_pad = (size_t)v & ~(size_t)15;
    }
#define RK_BX_N(RBFV, KVV, NKSV)                                                                           \
    hipLaunchKernelGGL((kmat_matmat_bf16x3_kernel<RBFV, KVV, NKSV, false>), grid, dim3(256), bx_smem_bytes(KVV) + lds_pad, \
                       st, kind, img1, img2, n1, n2, w, scale, nxp, nyp, V, k, ldv, tiles_per_seg, part, Y, ldy, \
                       accumulate, BxSym{}, vimg, vinfo)
#define RK_BX(RBFV, KVV)                                                                                   \
    do {                                                                                                   \
        if (nks == 1) RK_BX_N(RBFV, KVV, 1);                                                               \
        else if (nks == 2) RK_BX_N(RBFV, KVV, 2);                                                          \
        else RK_BX_N(RBFV, KVV, 4);                                                                        \
    } while (0)
#define RK_BX_K(RBFV)                                                                                      \
    do {                                                                                                   \
        if (k == 1) RK_BX(RBFV, 1);                                                                        \
        else if (k == 2) RK_BX(RBFV, 2);                                                                   \
        else if (k <= 4) RK_BX(RBFV, 4);                                                        \
        else RK_BX(RBFV, 0);                                                                               \
    } while (0)
    if (kind == RK_RBF) RK_BX_K(RK_RBF);
    else if (kind == RK_MATERN32) RK_BX_K(RK_MATERN32);
    else RK_BX_K(RK_MATERN52);
#undef RK_BX
#undef RK_BX_N
#undef RK_BX_K
    return launch_status("rk_kernel_matmat(bf16x3)");
}

// Symmetric product (X1 is X2, same gather, k <= 4, d <= 64): `aux` = bx_sym_ws_bytes(n, k) bytes.
int launch_kmatmat_bf16x3_sym(int kind, const float* X, int64_t n, int64_t ldx, const int64_t* idx, int64_t d,
                              float w, const float* ell_vec, const float* center, float scale, const float* V,
                              int k, int64_t ldv, void* aux, float* Y, int64_t ldy, int accumulate, hipStream_t st) {
    const int kv = k <= 1 ? 1 : (k <= 2 ? 2 : 4);
    const int64_t segt = bx_sym_segt(n);
    int64_t items = 0, col_tiles = 0;
    bx_sym_counts(n, segt, &items, &col_tiles);
    const int64_t T = ceil_div(n, 64), nrb = ceil_div(n, 128);
    const int64_t npad = round_up(n, BX_TI);
    char* base = static_cast<char*>(aux);
    float* iw = reinterpret_cast<float*>(base);
    float* muw = iw + 128;
    float* nrm = muw + 128;
    unsigned char* img = reinterpret_cast<unsigned char*>(base) + bx_float_bytes(n, 0);
    char* lists = base + bx_aux_bytes(n, 0);
    int64_t* item_tlo = reinterpret_cast<int64_t*>(lists);
    int64_t* item_coff = item_tlo + items;
    int64_t* rb_base = item_coff + items;
    int* item_rb = reinterpret_cast<int*>(rb_base + nrb + 1);
    float* prow = reinterpret_cast<float*>(
        lists + round_up(items * (int64_t)(sizeof(int) + 2 * sizeof(int64_t)) + (nrb + 1) * (int64_t)sizeof(int64_t), 256));
    float* pcol = prow + round_up(items * BX_TI * kv * (int64_t)sizeof(float), 256) / sizeof(float);
    gram_prep_launch(center, ell_vec, d, iw, muw, st);
    hipLaunchKernelGGL(gram_split_kernel<64>, dim3((unsigned)ceil_div(npad * 8, 256)), dim3(256), 0, st, X, ldx, idx,
                       n, npad, d, iw, muw, img, nrm);
    hipLaunchKernelGGL(bx_sym_items_kernel, dim3(1), dim3(1), 0, st, T, nrb, segt, item_rb, item_tlo, item_coff,
                       rb_base);
    BxSym sym{item_rb, item_tlo, item_coff, segt, prow, pcol};
    const int nks = d <= 32 ? 1 : 2;
    dim3 grid((unsigned)items);
#define RK_BXS_N(RBFV, KVV, NKSV)                                                                          \
    hipLaunchKernelGGL((kmat_matmat_bf16x3_kernel<RBFV, KVV, NKSV, true>), grid, dim3(256),               \
                       bx_sym_smem_bytes(KVV), st, kind, img, img, n, n, w, scale, nrm, nrm, V, k, ldv,    \
                       (int64_t)0, (float*)nullptr, (float*)nullptr, (int64_t)0, 0, sym,                   \
                       (const unsigned char*)nullptr, (const unsigned*)nullptr)
#define RK_BXS(RBFV, KVV)                                                                                  \
    do {                                                                                                   \
        if (nks == 1) RK_BXS_N(RBFV, KVV, 1);                                                              \
        else RK_BXS_N(RBFV, KVV, 2);                                                                       \
    } while (0)
#define RK_BXS_K(RBFV)                                                                                     \
    do {                                                                                                   \
        if (kv == 1) RK_BXS(RBFV, 1);                                                                      \
        else if (kv == 2) RK_BXS(RBFV, 2);                                                                 \
        else RK_BXS(RBFV, 4);                                                                              \
    } while (0)
    if (kind == RK_RBF) RK_BXS_K(RK_RBF);
    else if (kind == RK_MATERN32) RK_BXS_K(RK_MATERN32);
    else RK_BXS_K(RK_MATERN52);
#undef RK_BXS_K
#undef RK_BXS
#undef RK_BXS_N
    int rc = launch_status("rk_kernel_matmat(bf16x3, symmetric)");
    if (rc != RK_OK) return rc;
    const dim3 fg((unsigned)ceil_div(n * k, 256));
    if (kv == 1)
        hipLaunchKernelGGL(bx_sym_finish_kernel<1>, fg, dim3(256), 0, st, n, k, segt, rb_base, item_coff, prow, pcol,
                           scale, Y, ldy, accumulate);
    else if (kv == 2)
        hipLaunchKernelGGL(bx_sym_finish_kernel<2>, fg, dim3(256), 0, st, n, k, segt, rb_base, item_coff, prow, pcol,
                           scale, Y, ldy, accumulate);
    else
        hipLaunchKernelGGL(bx_sym_finish_kernel<4>, fg, dim3(256), 0, st, n, k, segt, rb_base, item_coff, prow, pcol,
                           scale, Y, ldy, accumulate);
    return launch_status("rk_kernel_matmat(bf16x3, symmetric finish)");
}

}  // namespace rk
