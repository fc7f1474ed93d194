// Wide products with a strip-packed symmetric kernel matrix, straight from the resident strips:
//
//     Y (n x r) = K V,   K = K(X, X) stored as the packed upper triangle of rk_symm.hip (float32),
//                        r = 17 .. 1024 columns (the Nystrom sketch  Y = K Omega,
//                        rlaopt/preconditioners/nystrom.py:79-80 - the one n x n x r product of a solve)
//
// on the double-rate 16-bit matrix cores at float32 accuracy.  Before this kernel the sketch product wrote
// temporary row-major panels of K (a second evaluation of every entry: 182 ms at n = 200 000) and
// multiplied them on the float32 matrix cores (358 ms at 115 TFLOP/s, 73 % of their peak).
//
// Arithmetic.  Both factors are split into 16-bit pieces whose products the matrix cores form exactly and
// accumulate in float32:
//   * default: TWO float16 pieces (round to nearest), x 2^e = h + m + O(2^-22 |x|); three products
//     h.h + h.m + m.h per pair (the dropped m.m is below 2^-22 relative).  Both factors are first scaled by
//     a power of two (K: from the caller's bound on its entries; V: from its largest entry, found on the
//     device) so that the pieces sit in float16's normal range; the result is unscaled exactly.  Entry error
//     <= 2.4e-7 |x| - four float32 ulps, far inside what a float32 accumulation over n = 200 000 terms
//     carries - for half the matrix-core time of
//   * RK_SYMGEMM_SPLIT=bf16x3: THREE bfloat16 pieces, six products (the split of rk_kernelbf16.hip:
//     below 2^-23 relative, the rounding level of float32 itself).
//
// Work.  A workgroup owns 128 rows of Y and TN = 256 (or 128) of its columns and walks the whole
// contraction j = 0 .. n in steps of 32, ascending - every workgroup in the same order, so the V tiles of a
// step are shared through L2 by everything resident.  K[i][j] for the rows of row block I comes
//   * j in the strip that holds I's own columns, rows above its diagonal block: TRANSPOSED tiles of that
//     strip (lane <-> row i, 8 dword loads down the strip's rows: 256 contiguous bytes per load and wave);
//   * j from I's diagonal block on: DIRECT tiles, rows I of every later strip (128 contiguous bytes per row).
// Every stored entry is read twice over the whole launch (once per use; the two workgroups that share a
// tile's rows sit on one XCD), nothing is accumulated across workgroups: no atomics, no partial tables,
// bitwise reproducible.  With cyclically dealt strips (one rank per GPU) a rank forms the contributions of
// ITS strips to every row block: the partials of the ranks add up to K V.
//
// Per step and workgroup: the K tile is loaded to registers, scaled, split and written to LDS in operand layout
// (row-major 64-byte rows, 16-byte chunks XOR-swizzled: conflict-free ds_read_b128 for the 16-row x 4-chunk operand
// fetch); the V tile is a linear copy of a pre-split, pre-swizzled image written once per call.  Waves: 4 x 2 (eight per
// workgroup, two per SIMD), each 32 rows x TN / 2 columns = 2 x NCB accumulator blocks of v_mfma_f32_16x16x32_{f16,bf16}.
//
// Two kernels:
//   * symgemm_ph_kernel (default for the float16 split): two phases per step, the two waves of a SIMD half a step apart,
//     every wait behind the matrix-core work, the V tiles by LDS-DMA - see the comment in front of it.  104 ms at
//     n = 200 000, r = 512 against 117 ms on the same box for
//   * symgemm_kernel (RK_SYMGEMM_IMPL=lockstep, and the three-bfloat16-piece split): two LDS buffers, loads two and three
//     steps ahead in two register sets, one barrier per step, every wave in the same part of the step at the same time
//     (with four waves, one per SIMD, the matrix cores were busy 39 % of the time).  Measured on it and not kept (A/B on
//     one box, 121.0 ms): s_setprio 1 for waves 4 - 7 (120.8 ms), sched_group_barrier interleaving of operand reads and
//     matrix-core instructions (122.2 ms); round 4: re-ordering the step's code for waves 4 - 7 - as a branch inside the
//     loop 294 ms (the compiler drains the prefetch queue at the joins), as two copies of the loop 223 ms
//     (profiles/r04_symgemm_stagger.log): the stagger that works keeps ONE instruction stream and delays half the waves
//     by a barrier (symgemm_ph_kernel).
// A V on the float16 grid (rk_symm_gemm_rhs16: the Gaussian test matrix is drawn on it) has ONE piece: two products per
// pair, half the V bytes: 84 ms.
#include <atomic>
#include <stdlib.h>
#include <type_traits>

#include "rk_common.h"
#include "rk_kerneltile.h"

namespace rk {

struct SgLayout {  // the packed layout of rk_symm.hip for float32 storage (W = 256)
    int64_t n, first, stride;
    static constexpr int W = 256;
    __host__ __device__ int64_t strips_total() const { return (n + W - 1) / W; }
    __host__ __device__ int64_t strips_local() const {
        const int64_t S = strips_total();
        return first < S ? (S - first + stride - 1) / stride : 0;
    }
    __host__ __device__ int64_t row_offset(int64_t t) const {
        return (int64_t)W * (t * (first + 1) + stride * (t * (t - 1) / 2));
    }
};

typedef unsigned int sg_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int sg_u2 __attribute__((ext_vector_type(2)));
typedef float sg_acc __attribute__((ext_vector_type(4)));
typedef float sg_acc16 __attribute__((ext_vector_type(16)));
// RK_SG_M32 = 1: the main product on v_mfma_f32_32x32x16_* (a wave's 32 x 128 part of Y = 4 blocks of 32 x 32) instead of
// 16x16x32 (2 x 8 blocks of 16 x 16): the same matrix-core cycles and LDS reads, HALF the matrix-core instructions - each
// holds the SIMD's vector issue port for 8 cycles, and with two waves per SIMD the port (matrix-core issue + ~100 vector +
// ~26 LDS instructions per wave and step) looked like what the step waits for (SQ_WAIT_INST_ANY 40 % of wave cycles).
// Measured (profiles/r04_symgemm_m32.log): SLOWER - 131.7 against 122.1 ms (three bfloat16 pieces: 242 against 207 ms),
// same results: the port is not the bound (and the 32 x 32 form holds a lower clock under load: MI355X_MICROARCH.md,
// DVFS item 7).  Kept as a build option.
#ifndef RK_SG_M32
#define RK_SG_M32 0
#endif
typedef _Float16 sg_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 sg_b8 __attribute__((ext_vector_type(8)));

constexpr int SG_TM = 128;   // rows of Y per workgroup
constexpr int SG_THREADS = 512;  // 8 waves: 4 (rows) x 2 (columns)
constexpr int SG_RB = 2;     // 16-row accumulator blocks per wave
constexpr int SG_KT = 32;    // contraction per step
constexpr int SG_ROWB = 64;  // bytes of one operand row in LDS (32 x 16 bit)
constexpr int SG_FOLD = 16;  // steps between folds of the running sums (a power of two; RK_SYMGEMM_FOLD overrides)

// swizzle of the 16-byte chunks of an operand row: chunk q of row r sits at q ^ ((-(r >> 2)) & 3).  For the
// four 16-lane groups in which the hardware serves a wave's ds_read_b128 (lanes {0-3, 12-15, 20-27}, ...)
// the 16 (row, chunk) pairs of an MFMA operand fetch then fall into 16 distinct 16-byte slots of the
// 256-byte bank row.
__host__ __device__ __forceinline__ int sg_swz(int r, int q) { return q ^ ((0 - (r >> 2)) & 3); }

// ---- piece formats ----------------------------------------------------------------------------------
struct SgF16 {  // two float16 pieces, three products
    static constexpr int NS = 2;
    static constexpr bool SCALED = true;
    // pieces of two values, packed (low half = first value)
    __device__ static __forceinline__ void split2(float a, float b, unsigned* out) {
        const _Float16 ha = (_Float16)a, hb = (_Float16)b;  // round to nearest even
        const float ra = a - (float)ha, rb = b - (float)hb;  // exact
        const _Float16 ma = (_Float16)ra, mb = (_Float16)rb;
        out[0] = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
        out[1] = (unsigned)__builtin_bit_cast(unsigned short, ma) | ((unsigned)__builtin_bit_cast(unsigned short, mb) << 16);
    }
    __device__ static __forceinline__ sg_acc mfma(sg_u4 a, sg_u4 b, sg_acc c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sg_h8, a), __builtin_bit_cast(sg_h8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ sg_acc16 mfma32(sg_u4 a, sg_u4 b, sg_acc16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sg_h8, a), __builtin_bit_cast(sg_h8, b), c, 0, 0, 0);
    }
};
struct SgBf16 {  // three bfloat16 pieces, six products
    static constexpr int NS = 3;
    static constexpr bool SCALED = false;
    __device__ static __forceinline__ unsigned rn(float x) {
        unsigned u = __float_as_uint(x);
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    }
    __device__ static __forceinline__ void split1(float x, unsigned* p) {
        p[0] = rn(x);
        const float r1 = x - __uint_as_float(p[0] << 16);
        p[1] = rn(r1);
        const float r2 = r1 - __uint_as_float(p[1] << 16);
        p[2] = rn(r2);
    }
    __device__ static __forceinline__ void split2(float a, float b, unsigned* out) {
        unsigned pa[3], pb[3];
        split1(a, pa);
        split1(b, pb);
#pragma unroll
        for (int p = 0; p < 3; ++p) out[p] = pa[p] | (pb[p] << 16);
    }
    __device__ static __forceinline__ sg_acc mfma(sg_u4 a, sg_u4 b, sg_acc c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sg_b8, a), __builtin_bit_cast(sg_b8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ sg_acc16 mfma32(sg_u4 a, sg_u4 b, sg_acc16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sg_b8, a), __builtin_bit_cast(sg_b8, b), c, 0, 0, 0);
    }
};

// ---- prologue 1: largest |V[:, c]| per column (device array, as float bits: non-negative floats order like
// integers).  info = [ absmax bits (kpad) | 2^eB per column (kpad) | alpha 2^-(eA + eB) per column (kpad) ] ----
constexpr int SG_INFO_BYTES = 3 * 1024 * 4;  // kpad <= 1024
__global__ void __launch_bounds__(256)
sg_absmax_kernel(const float* __restrict__ V, int64_t n, int64_t k, int64_t ldv, unsigned* __restrict__ out) {
    __shared__ float red[4][64];  // one atomic per column and workgroup
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + cx;
    float m = 0.f;
    if (c < k) {
        for (int64_t r = (int64_t)blockIdx.y * 4 + ry; r < n; r += (int64_t)gridDim.y * 4) {
            const float v = fabsf(V[r * ldv + c]);
            m = (v == v && v > m) ? v : m;
        }
    }
    red[ry][cx] = m;
    __syncthreads();
    if (ry == 0 && c < k) {
        float t = red[0][cx];
#pragma unroll
        for (int q = 1; q < 4; ++q) t = red[q][cx] > t ? red[q][cx] : t;
        if (t > 0.f) atomicMax(out + c, __float_as_uint(t));
    }
}

// scale factors per column: info[kpad + c] <- 2^eB, info[2 kpad + c] <- alpha 2^-(eA + eB)
__global__ void sg_scale_kernel(unsigned* __restrict__ info, int64_t kpad, int ea, float alpha, int scaled) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= kpad) return;
    int eb = 0;
    if (scaled) {
        const float m = __uint_as_float(info[c]);
        if (m > 0.f && m < 3.0e38f) {
            int ex;
            frexpf(m, &ex);  // m = f 2^ex, f in [0.5, 1): m <= 2^ex
            eb = 14 - ex;
            eb = eb > 100 ? 100 : (eb < -100 ? -100 : eb);
        }
    }
    info[kpad + c] = __float_as_uint(ldexpf(1.f, eb));
    info[2 * kpad + c] = __float_as_uint(alpha * ldexpf(1.f, -eb) * ldexpf(1.f, -ea));
}

// ---- prologue 2: the split image of V -----------------------------------------------------------------
// image[step = j / 32][column tile][piece][column in tile][32 halves, chunks swizzled]; rows >= n and
// columns >= k are zero.  One thread per (column, 8-row chunk).
template <typename F, int NSB = F::NS>  // NSB < F::NS: only the leading pieces are kept (V on the grid of the first piece)
__global__ void __launch_bounds__(256)
sg_split_rhs_kernel(const float* __restrict__ V, int64_t n, int64_t npad, int64_t k, int64_t kpad, int64_t ldv,
                    int tn, const unsigned* __restrict__ info, unsigned char* __restrict__ img) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t col = g % kpad;
    const int64_t chunk_all = g / kpad;  // 8-row chunk index over npad / 8
    if (chunk_all >= npad / 8) return;
    const float sb = F::SCALED ? __uint_as_float(info[kpad + col]) : 1.f;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t j = chunk_all * 8 + e;
        x[e] = (j < n && col < k) ? V[j * ldv + col] * sb : 0.f;
    }
    unsigned pk[4][F::NS];
#pragma unroll
    for (int e = 0; e < 4; ++e) F::split2(x[2 * e], x[2 * e + 1], pk[e]);
    const int64_t step = chunk_all / 4;
    const int q = (int)(chunk_all % 4);
    const int64_t ct = col / tn;
    const int cl = (int)(col % tn);
    const int64_t nct = kpad / tn;
    unsigned char* base = img + ((step * nct + ct) * NSB) * (int64_t)tn * SG_ROWB + cl * SG_ROWB + (sg_swz(cl, q) << 4);
#pragma unroll
    for (int p = 0; p < NSB; ++p) {
        sg_u4 o;
        o.x = pk[0][p]; o.y = pk[1][p]; o.z = pk[2][p]; o.w = pk[3][p];
        *reinterpret_cast<sg_u4*>(base + (int64_t)p * tn * SG_ROWB) = o;
    }
}

// ---- the product ----------------------------------------------------------------------------------------
template <typename F, int NCB>
__global__ void __launch_bounds__(SG_THREADS, 2)
symgemm_kernel(const float* __restrict__ Kp, SgLayout L, const unsigned char* __restrict__ img, int64_t nct,
               float sa, const unsigned* __restrict__ info, int64_t k, float* __restrict__ Y, int64_t ldy,
               int fold_mask) {
    constexpr int NS = F::NS;
    constexpr int TN = 32 * NCB;
    constexpr int W = SgLayout::W;
    constexpr int ABYTES = SG_TM * SG_ROWB;     // one piece of the K tile: 8 KiB
    constexpr int BBYTES = TN * SG_ROWB;        // one piece of the V tile
    constexpr int NBQ = NS * BBYTES / 16 / SG_THREADS;  // 16-byte chunks of the V tile per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* AL = smem;                       // [2][NS][ABYTES]
    unsigned char* BL = smem + 2 * NS * ABYTES;     // [2][NS][BBYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;  // wr: 32-row slab, wc: column half
    // workgroup id -> (row block, column tile): the column tiles of a row block are 8 ids apart - one XCD
    // under round-robin placement (speed only) - and start together
    const int64_t grp = (int64_t)blockIdx.x / (8 * nct);
    const int64_t rem = (int64_t)blockIdx.x % (8 * nct);
    // (the slot inside a group of 8 row blocks is rotated by the group index: with cyclically dealt strips the row
    // blocks whose own strip is local - the ones with the long transposed phase - are every (2 stride)-th pair; an
    // un-rotated map put ALL of them on two of the eight XCDs at stride 8: 57 ms for an eighth of the work)
    // Last row blocks first: with cyclically dealt strips a row block whose own strip is local walks 8 b transposed
    // steps on top of its direct ones - the later the block, the longer its workgroup: longest first shortens the tail
    // (one GPU: every workgroup does the same n / 32 steps and the order does not matter).
    const int64_t rb = ((int64_t)gridDim.x / nct - 1) - (grp * 8 + (((rem & 7) + grp) & 7));
    const int64_t ct = rem >> 3;
    const int64_t i0 = rb * SG_TM;
    if (i0 >= L.n) return;  // (uniform, before any barrier)
    const int64_t b = i0 / W;            // the strip that holds this row block's columns
    const int ioff = (int)(i0 - b * W);  // 0 or 128
    const int64_t T = L.strips_local();
    const bool own = b >= L.first && (b - L.first) % L.stride == 0;
    const int64_t tb = own ? (b - L.first) / L.stride : 0;
    const int64_t t_min = b <= L.first ? 0 : (b - L.first + L.stride - 1) / L.stride;
    const int64_t nT = own ? b * (W / SG_KT) : 0;
    const int64_t nD = t_min < T ? (T - t_min) * (W / SG_KT) : 0;

#if RK_SG_M32
    constexpr int NC32 = NCB / 2;  // 32 x 32 blocks of this wave's 32 rows x TN / 2 columns
    sg_acc16 acc[NC32], tot[NC32];
#pragma unroll
    for (int c = 0; c < NC32; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc[c][v] = 0.f; tot[c][v] = 0.f; }
    auto fold = [&]() {
#pragma unroll
        for (int c = 0; c < NC32; ++c) {
            tot[c] += acc[c];
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
        }
    };
#else
    sg_acc acc[SG_RB][NCB];
#pragma unroll
    for (int r = 0; r < SG_RB; ++r)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[r][c][v] = 0.f;

    // two-level running sums: the matrix cores add 3 (or 6) x 32 products per step to one float32 accumulator;
    // over n / 32 = 6 250 steps the rounding of that one long chain reached 6e-5 relative (measured at
    // n = 200 000).  The accumulators are folded into a second set every SG_FOLD steps: two short chains.
    sg_acc tot[SG_RB][NCB];
#pragma unroll
    for (int r = 0; r < SG_RB; ++r)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) tot[r][c][v] = 0.f;
    auto fold = [&]() {
#pragma unroll
        for (int r = 0; r < SG_RB; ++r)
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                tot[r][c] += acc[r][c];
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[r][c][v] = 0.f;
            }
    };
#endif

    // two register sets: the loads of steps q + 2 and q + 3 are in flight while step q is computed (with one set -
    // one step of cover - a rank's partial over cyclically dealt strips, where both operands come from HBM, ran at
    // a quarter of the single-GPU rate)
    struct Stage {
        float a[8];
        sg_u4 b[NBQ];
    };
    Stage sA, sB;

    // ---- loads of one step into registers ----------------------------------------------------------
    auto load_b = [&](int64_t jstep, Stage& st) {
        const sg_u4* src = reinterpret_cast<const sg_u4*>(img + ((jstep * nct + ct) * NS) * (int64_t)BBYTES);
#pragma unroll
        for (int q = 0; q < NBQ; ++q) st.b[q] = src[tid + SG_THREADS * q];
    };
    // transposed step q: rows j = 32 q .. + 32 of strip b, columns ioff + (0 .. 127): lane <-> column,
    // thread group tid >> 7 <-> 8-row chunk
    auto load_t = [&](int64_t q, Stage& st) {
        const float* base = Kp + (L.row_offset(tb) + q * SG_KT + (tid >> 7) * 8) * W + ioff + (tid & 127);
#pragma unroll
        for (int e = 0; e < 8; ++e) st.a[e] = __builtin_nontemporal_load(base + (int64_t)e * W);
        load_b(q, st);
    };
    // direct step q: local strip t_min + q / 8, its columns 32 (q % 8) .. + 32, rows i0 .. i0 + 127
    auto load_d = [&](int64_t q, Stage& st) {
        const int64_t t = t_min + q / (W / SG_KT);
        const int kk = (int)(q % (W / SG_KT)) * SG_KT;
        const int64_t s = L.first + t * L.stride;
        const float* base = Kp + L.row_offset(t) * W + kk + (tid & 7) * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int64_t row = i0 + u * 64 + (tid >> 3);
            row = row < L.n ? row : L.n - 1;  // (rows past n: finite stand-ins, never stored)
            const sg_u4 v = __builtin_nontemporal_load(reinterpret_cast<const sg_u4*>(base + row * W));
            st.a[u * 4 + 0] = __uint_as_float(v.x); st.a[u * 4 + 1] = __uint_as_float(v.y);
            st.a[u * 4 + 2] = __uint_as_float(v.z); st.a[u * 4 + 3] = __uint_as_float(v.w);
        }
        load_b(s * (W / SG_KT) + kk / SG_KT, st);
    };
    // ---- registers -> LDS (scale, split, operand layout) ------------------------------------------------
    auto store_b = [&](int buf, const Stage& st) {
        sg_u4* dst = reinterpret_cast<sg_u4*>(BL + buf * NS * BBYTES);
#pragma unroll
        for (int q = 0; q < NBQ; ++q) dst[tid + SG_THREADS * q] = st.b[q];
    };
    auto store_t = [&](int buf, const Stage& st) {
        unsigned char* al = AL + buf * NS * ABYTES;
        const int i = tid & 127, c = tid >> 7;
        unsigned pk[4][NS];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            F::split2(st.a[2 * e] * (F::SCALED ? sa : 1.f), st.a[2 * e + 1] * (F::SCALED ? sa : 1.f), pk[e]);
        unsigned char* dst = al + i * SG_ROWB + (sg_swz(i, c) << 4);
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            sg_u4 o;
            o.x = pk[0][p]; o.y = pk[1][p]; o.z = pk[2][p]; o.w = pk[3][p];
            *reinterpret_cast<sg_u4*>(dst + p * ABYTES) = o;
        }
        store_b(buf, st);
    };
    auto store_d = [&](int buf, const Stage& st) {
        unsigned char* al = AL + buf * NS * ABYTES;
        const int c16 = tid & 7;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = u * 64 + (tid >> 3);
            unsigned pk[2][NS];
#pragma unroll
            for (int e = 0; e < 2; ++e)
                F::split2(st.a[u * 4 + 2 * e] * (F::SCALED ? sa : 1.f), st.a[u * 4 + 2 * e + 1] * (F::SCALED ? sa : 1.f), pk[e]);
            unsigned char* dst = al + row * SG_ROWB + (sg_swz(row, c16 >> 1) << 4) + (c16 & 1) * 8;
#pragma unroll
            for (int p = 0; p < NS; ++p) {
                sg_u2 o;
                o.x = pk[0][p]; o.y = pk[1][p];
                *reinterpret_cast<sg_u2*>(dst + p * ABYTES) = o;
            }
        }
        store_b(buf, st);
    };
    // ---- one step of matrix-core work out of LDS ----------------------------------------------------------
#if RK_SG_M32
    // operands of v_mfma_f32_32x32x16: lane (row or column lane % 32, 8 contraction indices 16 kk + 8 (lane / 32) .. + 8)
    const int l32 = lane & 31, lh = lane >> 5;
    const int a_off = (wr * 32 + l32) * SG_ROWB;
    const int b_off = (wc * (TN / 2) + l32) * SG_ROWB;
    auto compute = [&](int buf) {
        const unsigned char* al = AL + buf * NS * ABYTES + a_off;
        const unsigned char* bl = BL + buf * NS * BBYTES + b_off;
        sg_u4 af[2][NS];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int p = 0; p < NS; ++p)
                af[kk][p] = *reinterpret_cast<const sg_u4*>(al + (sg_swz(l32, 2 * kk + lh) << 4) + p * ABYTES);
#pragma unroll
        for (int c = 0; c < NC32; ++c) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                sg_u4 bf[NS];
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    bf[p] = *reinterpret_cast<const sg_u4*>(bl + c * 32 * SG_ROWB + (sg_swz(l32, 2 * kk + lh) << 4) + p * BBYTES);
#pragma unroll
                for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
                    for (int pa = 0; pa <= sum; ++pa) acc[c] = F::mfma32(af[kk][pa], bf[sum - pa], acc[c]);
            }
        }
    };
#else
    const int a_off = (wr * 16 * SG_RB + lr) * SG_ROWB + (sg_swz(lr, lq) << 4);       // + rbk * 16 rows
    const int b_off = (wc * (TN / 2) + lr) * SG_ROWB + (sg_swz(lr, lq) << 4);         // + cb * 16 columns
    auto compute = [&](int buf) {
        const unsigned char* al = AL + buf * NS * ABYTES + a_off;
        const unsigned char* bl = BL + buf * NS * BBYTES + b_off;
        sg_u4 af[SG_RB][NS];
#pragma unroll
        for (int r = 0; r < SG_RB; ++r)
#pragma unroll
            for (int p = 0; p < NS; ++p)
                af[r][p] = *reinterpret_cast<const sg_u4*>(al + r * 16 * SG_ROWB + p * ABYTES);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            sg_u4 bf[NS];
#pragma unroll
            for (int p = 0; p < NS; ++p) bf[p] = *reinterpret_cast<const sg_u4*>(bl + c * 16 * SG_ROWB + p * BBYTES);
#pragma unroll
            for (int r = 0; r < SG_RB; ++r) {
                // small terms first: piece indices (pa, pb) with pa + pb = NS - 1 down to 0
#pragma unroll
                for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
                    for (int pa = 0; pa <= sum; ++pa) acc[r][c] = F::mfma(af[r][pa], bf[sum - pa], acc[r][c]);
            }
        }
    };
#endif

    // ---- the two phases: transposed tiles, then direct tiles (same pipeline, different loads / stores) ----------
    auto run_phase = [&](auto load, auto store, int64_t nst) {
        if (nst <= 0) return;
        auto at = [&](int64_t q) { return q < nst ? q : nst - 1; };  // (past the end: the last step again, unused)
        load(0, sA);
        store(0, sA);
        load(at(1), sA);
        load(at(2), sB);
        __syncthreads();
        for (int64_t q = 0; q < nst; q += 2) {
            store(1, sA);          // step q + 1
            load(at(q + 3), sA);
            compute(0);
            if (((int)q & fold_mask) == fold_mask) fold();
            __syncthreads();
            if (q + 1 < nst) {
                store(0, sB);      // step q + 2
                load(at(q + 4), sB);
                compute(1);
                if (((int)(q + 1) & fold_mask) == fold_mask) fold();
                __syncthreads();
            }
        }
    };
    run_phase(load_t, store_t, nT);
    run_phase(load_d, store_d, nD);
    // ---- epilogue: element v of a lane is Y[row 4 (lane / 16) + v][column lane % 16] of its block -----------
    const int64_t kpad = nct * TN;
#if RK_SG_M32
    // element v of a 32 x 32 block: row (v & 3) + 8 (v >> 2) + 4 (lane / 32), column lane % 32
#pragma unroll
    for (int c = 0; c < NC32; ++c) {
        const int64_t col = ct * TN + wc * (TN / 2) + c * 32 + l32;
        const float unscale = __uint_as_float(info[2 * kpad + col]);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int64_t row = i0 + wr * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            if (row < L.n && col < k) Y[row * ldy + col] = (tot[c][v] + acc[c][v]) * unscale;
        }
    }
}

#else
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int64_t col = ct * TN + wc * (TN / 2) + c * 16 + lr;
        const float unscale = __uint_as_float(info[2 * kpad + col]);
#pragma unroll
        for (int r = 0; r < SG_RB; ++r)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t row = i0 + wr * 16 * SG_RB + r * 16 + 4 * lq + v;
                if (row < L.n && col < k) Y[row * ldy + col] = (tot[r][c][v] + acc[r][c][v]) * unscale;
            }
    }
}
#endif

// ------------------------------------------------------------------------------------------------------------------
// symgemm_ph_kernel: the same product (same tiles, same order of additions per accumulator: bitwise the results of
// symgemm_kernel<SgF16>) with the step cut into TWO phases and the eight waves in two groups HALF A STEP APART.
//
// symgemm_kernel's waves run in lockstep: one barrier per step, every wave stages (split + LDS stores), then every wave
// reads operands, then every wave issues its 48 matrix-core instructions.  Probes of this kernel's first form (four
// phases per step; profiles/r04_symgemm_phases.log) showed what a step costs: the matrix-core work alone 63 ms, everything
// else alone 77 ms - per step a chain of latencies (LDS-DMA issue ~75 - 150 cycles a piece, LDS round trips, barriers),
// not throughput - and the two ADD UP unless one wave's chain runs beside its SIMD partner's matrix-core work.
// Here every wave runs THE SAME code
//     L(q): issue only: LDS store of K tile q + 1 (split in M(q - 1)), operand reads (A, two quarters of B), DMA of a V
//           tile, loads of K tile q + 4                                                                 | barrier |
//     M(q): 48 matrix-core instructions; between them the other two quarters of B (into the register sets of the
//           first two); the split of K tile q + 2 among the last of them (two pieces of V) or spread over all of them
//           (one piece) - measured, see RK_SG_FILL; then the step's two waits                            | barrier |
// and waves 4 - 7 pass ONE barrier more before their first step (waves 0 - 3 one more at the end): a SIMD holds one wave
// of each group, so one is in a phase M while the other is in a phase L.
//   * V tiles: global_load_lds_dwordx4 (LDS-DMA) straight from the pre-split image into a ring of FOUR LDS slots (tile t
//     in slot t % 4): no registers, no ds_write.  With the K-tile buffers that is all 160 KiB.
//   * K tiles: registers (they are split on the way; tile t in register set t % 4, loaded four steps ahead), by inline
//     asm with a counted s_waitcnt: beside an LDS-DMA in flight the compiler waits vmcnt(0) for every ordinary load.
//     Every wait sits at the END of a phase M, behind its matrix-core instructions: queue of a wave there, oldest first,
//     .. | V K(q+3) | V K(q+4) (what L(q - 1) and L(q) issued): vmcnt(NA + NBQ) = K tile q + 3 (split in M(q + 1)) and the
//     older V tile have landed, what L(q) issued stays in flight.  A phase L only issues.
//   * a group stages (splits, stores) the 64 rows of a K tile that its own waves read: the K tiles never cross groups.
//   * hazards (g0 / g1 = the groups, I_k = the interval between barriers k - 1 and k; L(q) / M(q) are phases 2q / 2q + 1;
//     g0 runs phase k in I_k, g1 in I_k+1; every wave waits lgkmcnt(0) before the barrier that ends a phase L, and all
//     reads of a phase M are consumed inside it):
//       K tile q + 2 -> buffer q % 2 in L(q + 1) (I_2q+2 / I_2q+3); that buffer was last read in L(q) (I_2q / I_2q+1) and
//       is next read in L(q + 2) (I_2q+4 / I_2q+5).
//       V tile t -> slot t % 4 by DMA: group 0 issues its share in L(t - 2) (I_2t-4), group 1 - one interval behind in
//       everything else - in ITS L(t - 3) (I_2t-5): the slot's tile t - 4 was last read in M(t - 4) (I_2t-7 / I_2t-6).  The
//       counted wait at the end of M(t - 1) (group 0, I_2t-1) / M(t - 2) (group 1, I_2t-2) covers it - three intervals of
//       flight for both - and the first read is L(t) (I_2t / I_2t+1).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sg_bar() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void sg_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// timing probes (wrong results): RK_PH_PROBE bit 0: no DMA, 1: no matrix-core instructions, 2: no split / LDS store of the K
// tile, 3: no stagger, 4: no K-tile loads, 5: K-tile loads of the same eight tiles over and over (cache hits), 6: the
// transposed steps with the direct steps' loads (2 x dwordx4 instead of 8 x dword per thread)
#ifndef RK_PH_PROBE
#define RK_PH_PROBE 0
#endif
#ifndef RK_SG_VMEM_FIRST
#define RK_SG_VMEM_FIRST 0  // phase L: the DMA and the K-tile loads in front of the LDS instructions
#endif
#ifndef RK_SG_LATE
#define RK_SG_LATE 0  // > 0: no vector instructions in the first half of a phase M, this many per matrix-core instruction in the second
#endif
#ifndef RK_SG_SPLIT_IN_L
#define RK_SG_SPLIT_IN_L 0  // the split of the K tile in the phase L (in front of its barrier) instead of between the phase M's matrix-core instructions
#endif
#ifndef RK_SG_SGB
#define RK_SG_SGB 1  // sched_group_barrier order of a phase M (0: the compiler's own)
#endif
#ifndef RK_SG_PRIO
#define RK_SG_PRIO 1  // s_setprio 1 around the matrix-core work of a phase M
#endif
#ifndef RK_SG_PF
#define RK_SG_PF 0  // > 0: every wave touches 16 lines of the K tile this many steps ahead (one dword per line, result unused):
                    // measured 4 / 8 / 16 steps: 110.5 / 116.8 / 117.8 against 110.4 ms (profiles/r04_symgemm_phases.log): off
#endif
#ifndef RK_SG_FILL
#define RK_SG_FILL (-1)  // vector instructions (the split of a K tile) the scheduler places behind each matrix-core instruction
#endif                   // of a phase M; -1: 0 with two pieces of V (the split then follows the 48 matrix-core instructions:
                         // 102.6 against 112.3 ms with 1), 1 with one piece (84.6 against 86.3 ms with 0);
                         // profiles/r04_symgemm_phases.log (also: no s_setprio, the compiler's own order, the split in phase L)

struct SgStageT { float a[8]; };   // transposed phase: 8 rows j of one column i
struct SgStageD { sg_u4 d[2]; };   // direct phase: 4 columns j of two rows i
struct SgPkT { sg_u4 o[2]; };      // the split pieces of a thread's part of a K tile, as they go to LDS
struct SgPkD { sg_u2 o[2][2]; };
// the pieces count as written HERE (the compiler would otherwise sink the split to its use, behind the next barrier)
__device__ __forceinline__ void sg_pin(SgPkT& pk) { asm volatile("" : "+v"(pk.o[0]), "+v"(pk.o[1])); }
__device__ __forceinline__ void sg_pin(SgPkD& pk) {
    asm volatile("" : "+v"(pk.o[0][0]), "+v"(pk.o[0][1]), "+v"(pk.o[1][0]), "+v"(pk.o[1][1]));
}
struct SgWaitT {  // vmcnt(N) naming a register set: the set counts as written here, not at its asm load
    template <int N> static __device__ __forceinline__ void at(SgStageT& st) {
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(st.a[0]), "+v"(st.a[1]), "+v"(st.a[2]), "+v"(st.a[3]), "+v"(st.a[4]), "+v"(st.a[5]),
                       "+v"(st.a[6]), "+v"(st.a[7])
                     : "n"(N)
                     : "memory");
    }
    static constexpr int NA = 8;  // loads per K tile and thread
};
struct SgWaitD {
    template <int N> static __device__ __forceinline__ void at(SgStageD& st) {
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(st.d[0]), "+v"(st.d[1]) : "n"(N) : "memory");
    }
    static constexpr int NA = 2;
};

#ifndef RK_SG_TRACE
#define RK_SG_TRACE 0
#endif
#if RK_SG_TRACE  // tuning builds only (one piece of V: LDS to spare): cycle stamps of 16 steps of one workgroup (tools/ab_symgemm.py --trace)
__device__ unsigned long long sg_trace_buf[256];
constexpr int SG_TRACE_BLOCK = 300, SG_TRACE_Q0 = 100, SG_TRACE_PTS = 6;
#endif

template <int NCB, int NSB>  // NSB: pieces of V (2, or 1 for a V on the float16 grid: two exact products per pair, half the V bytes)
__global__ void __launch_bounds__(SG_THREADS, 2)
symgemm_ph_kernel(const float* __restrict__ Kp, SgLayout L, const unsigned char* __restrict__ img, int64_t nct,
                  float sa, const unsigned* __restrict__ info, int64_t k, float* __restrict__ Y, int64_t ldy,
                  int fold_mask) {
    using F = SgF16;
    constexpr int NS = 2;
    constexpr int TN = 32 * NCB;
    constexpr int QC = NCB / 4;                 // 16-column blocks per wave and quarter of a phase M
    constexpr int W = SgLayout::W;
    constexpr int ABYTES = SG_TM * SG_ROWB;     // one piece of the K tile: 8 KiB
    constexpr int BBYTES = TN * SG_ROWB;        // one piece of the V tile
    constexpr int BTILE = NSB * BBYTES;         // a V tile: 32 KiB (TN = 256, two pieces)
    constexpr int NBQ = BTILE / 16 / SG_THREADS;  // DMAs per wave and tile: wave w copies bytes [w NBQ KiB, (w + 1) NBQ KiB)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // (the only LDS object of the kernel)
    unsigned char* AL = smem;                       // [2][NS][ABYTES]
    unsigned char* BL = smem + 2 * NS * ABYTES;     // [4][BTILE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int grp = wave >> 2;
    const int64_t grpi = (int64_t)blockIdx.x / (8 * nct);
    const int64_t rem = (int64_t)blockIdx.x % (8 * nct);
    const int64_t rb = ((int64_t)gridDim.x / nct - 1) - (grpi * 8 + (((rem & 7) + grpi) & 7));
    const int64_t ct = rem >> 3;
    const int64_t i0 = rb * SG_TM;
    if (i0 >= L.n) return;  // (uniform, before any barrier)
    const int64_t b = i0 / W;
    const int ioff = (int)(i0 - b * W);
    const int64_t T = L.strips_local();
    const bool own = b >= L.first && (b - L.first) % L.stride == 0;
    const int64_t tb = own ? (b - L.first) / L.stride : 0;
    const int64_t t_min = b <= L.first ? 0 : (b - L.first + L.stride - 1) / L.stride;
    const int64_t nT = own ? b * (W / SG_KT) : 0;
    const int64_t nD = t_min < T ? (T - t_min) * (W / SG_KT) : 0;

    sg_acc acc[SG_RB][NCB], tot[SG_RB][NCB];
#pragma unroll
    for (int r = 0; r < SG_RB; ++r)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) { acc[r][c][v] = 0.f; tot[r][c][v] = 0.f; }
    auto fold = [&]() {
#pragma unroll
        for (int r = 0; r < SG_RB; ++r)
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                tot[r][c] += acc[r][c];
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[r][c][v] = 0.f;
            }
    };

    // ---- V tiles: LDS-DMA, NBQ pieces of 1 KiB per wave and tile -----------------------------------------------------
    const unsigned char* bsrc0 = img + (ct * NSB) * (int64_t)BBYTES + wave * (NBQ * 1024) + lane * 16;
    auto dma_b = [&](int64_t jstep, int slot) {
        if constexpr (!(RK_PH_PROBE & 1)) {
            const unsigned char* src = bsrc0 + jstep * nct * (int64_t)BTILE;
            auto* dst = (__attribute__((address_space(3))) void*)(BL + slot * BTILE + wave * (NBQ * 1024));
            __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
            if constexpr (NBQ > 1) __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
            if constexpr (NBQ > 2) __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0);
            if constexpr (NBQ > 3) __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0);
        }
    };
    // ---- K tiles: asm loads, counted waits ---------------------------------------------------------------------------
    using StageT = SgStageT;
    using StageD = SgStageD;
    // a group stages the 64 rows of the K tile that its own waves read (rows 64 grp .. + 64): no LDS traffic between the
    // groups for the K tiles.  tg = thread in the group.  Transposed tiles: column ti of the strip, 8-row chunk tch;
    // direct tiles: rows td and td + 32 of the group's 64, columns 4 (tg % 8) .. + 4
    const int tg = tid & 255;
    const int ti = grp * 64 + (tg & 63), tch = tg >> 6;
    const int td = grp * 64 + (tg >> 3);
    using PkT = SgPkT;
    using PkD = SgPkD;
    auto load_t = [&](int64_t q, StageT& st) {
        const float* p0 = Kp + (L.row_offset(tb) + q * SG_KT + tch * 8) * W + ioff + ti;
        const float* p1 = p0 + 4 * W;
        asm volatile(
            "global_load_dword %0, %8, off nt\n\t"
            "global_load_dword %1, %8, off offset:1024 nt\n\t"
            "global_load_dword %2, %8, off offset:2048 nt\n\t"
            "global_load_dword %3, %8, off offset:3072 nt\n\t"
            "global_load_dword %4, %9, off nt\n\t"
            "global_load_dword %5, %9, off offset:1024 nt\n\t"
            "global_load_dword %6, %9, off offset:2048 nt\n\t"
            "global_load_dword %7, %9, off offset:3072 nt"
            : "=&v"(st.a[0]), "=&v"(st.a[1]), "=&v"(st.a[2]), "=&v"(st.a[3]), "=&v"(st.a[4]), "=&v"(st.a[5]),
              "=&v"(st.a[6]), "=&v"(st.a[7])
            : "v"(p0), "v"(p1)
            : "memory");
    };
    unsigned pfreg = 0;  // (destination of the prefetch touches: stays allocated from touch to touch)
    auto touch_t = [&](int64_t q) {  // wave w: rows 4 w .. + 4 of the 32-row tile, 4 lines of 128 B each
        const float* pa = Kp + (L.row_offset(tb) + q * SG_KT + wave * 4 + ((lane & 15) >> 2)) * W + ioff + (lane & 3) * 32;
        asm volatile("global_load_dword %0, %1, off" : "+v"(pfreg) : "v"(pa) : "memory");
    };
    auto jstep_t = [&](int64_t q) { return q; };
    auto load_d = [&](int64_t q, StageD& st) {
        const int64_t t = t_min + q / (W / SG_KT);
        const int kk = (int)(q % (W / SG_KT)) * SG_KT;
        const float* base = Kp + L.row_offset(t) * W + kk + (tid & 7) * 4;
        int64_t r0 = i0 + td, r1 = r0 + 32;
        r0 = r0 < L.n ? r0 : L.n - 1;  // (rows past n: finite stand-ins, never stored)
        r1 = r1 < L.n ? r1 : L.n - 1;
        const float* p0 = base + r0 * W;
        const float* p1 = base + r1 * W;
        asm volatile(
            "global_load_dwordx4 %0, %2, off nt\n\t"
            "global_load_dwordx4 %1, %3, off nt"
            : "=&v"(st.d[0]), "=&v"(st.d[1])
            : "v"(p0), "v"(p1)
            : "memory");
    };
    auto touch_d = [&](int64_t q) {  // wave w: rows 16 w .. + 16 of the tile, one line each
        const int64_t t = t_min + q / (W / SG_KT);
        const int kk = (int)(q % (W / SG_KT)) * SG_KT;
        int64_t r = i0 + wave * 16 + (lane & 15);
        r = r < L.n ? r : L.n - 1;
        const float* pa = Kp + (L.row_offset(t) + r) * W + kk;
        asm volatile("global_load_dword %0, %1, off" : "+v"(pfreg) : "v"(pa) : "memory");
    };
    auto jstep_d = [&](int64_t q) {
        const int64_t t = t_min + q / (W / SG_KT);
        return (L.first + t * L.stride) * (W / SG_KT) + q % (W / SG_KT);
    };
    // ---- registers -> pieces (scale, split), pieces -> LDS (operand layout) -------------------------------------------
    auto split_t = [&](const StageT& st) {
        unsigned pk[4][NS];
#pragma unroll
        for (int e = 0; e < 4; ++e) F::split2(st.a[2 * e] * sa, st.a[2 * e + 1] * sa, pk[e]);
        PkT out;
#pragma unroll
        for (int p = 0; p < NS; ++p) { out.o[p].x = pk[0][p]; out.o[p].y = pk[1][p]; out.o[p].z = pk[2][p]; out.o[p].w = pk[3][p]; }
        return out;
    };
    const int st_off_t = ti * SG_ROWB + (sg_swz(ti, tch) << 4);
    auto store_t = [&](int buf, const PkT& pk) {
        unsigned char* dst = AL + buf * NS * ABYTES + st_off_t;
#pragma unroll
        for (int p = 0; p < NS; ++p) *reinterpret_cast<sg_u4*>(dst + p * ABYTES) = pk.o[p];
    };
    auto split_d = [&](const StageD& st) {
        PkD out;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            unsigned pk[2][NS];
            F::split2(__uint_as_float(st.d[u].x) * sa, __uint_as_float(st.d[u].y) * sa, pk[0]);
            F::split2(__uint_as_float(st.d[u].z) * sa, __uint_as_float(st.d[u].w) * sa, pk[1]);
#pragma unroll
            for (int p = 0; p < NS; ++p) { out.o[u][p].x = pk[0][p]; out.o[u][p].y = pk[1][p]; }
        }
        return out;
    };
    const int st_off_d = td * SG_ROWB + (sg_swz(td, (tid & 7) >> 1) << 4) + (tid & 1) * 8;  // (+ 32 rows: same swizzle)
    auto store_d = [&](int buf, const PkD& pk) {
        unsigned char* dst = AL + buf * NS * ABYTES + st_off_d;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < NS; ++p) *reinterpret_cast<sg_u2*>(dst + u * 32 * SG_ROWB + p * ABYTES) = pk.o[u][p];
    };
    // ---- operand reads and matrix-core work --------------------------------------------------------------------------
    const int a_off = (wr * 16 * SG_RB + lr) * SG_ROWB + (sg_swz(lr, lq) << 4);
    const int b_off = (wc * (TN / 2) + lr) * SG_ROWB + (sg_swz(lr, lq) << 4);
    sg_u4 af[SG_RB][NS], bfa[QC][NSB], bfb[QC][NSB];
    auto read_a = [&](int buf) {
        const unsigned char* al = AL + buf * NS * ABYTES + a_off;
#pragma unroll
        for (int r = 0; r < SG_RB; ++r)
#pragma unroll
            for (int p = 0; p < NS; ++p) af[r][p] = *reinterpret_cast<const sg_u4*>(al + r * 16 * SG_ROWB + p * ABYTES);
    };
    auto read_bq = [&](int slot, int quarter, sg_u4 (&bf)[QC][NSB]) {
        const unsigned char* bl = BL + slot * BTILE + b_off + quarter * QC * 16 * SG_ROWB;
#pragma unroll
        for (int c = 0; c < QC; ++c)
#pragma unroll
            for (int p = 0; p < NSB; ++p) bf[c][p] = *reinterpret_cast<const sg_u4*>(bl + c * 16 * SG_ROWB + p * BBYTES);
    };
    auto mma_q = [&](int quarter, const sg_u4 (&bf)[QC][NSB]) {
        if constexpr (RK_PH_PROBE & 2) {  // (keep the operand reads alive)
#pragma unroll
            for (int c = 0; c < QC; ++c)
#pragma unroll
                for (int r = 0; r < SG_RB; ++r)
                    asm volatile("" : : "v"(af[r][0]), "v"(af[r][1]), "v"(bf[c][0]), "v"(bf[c][NSB - 1]));
            return;
        }
        // small terms first (the order of symgemm_kernel per accumulator): h.m (two pieces of V only), m.h, then h.h
        if constexpr (NSB == 2) {
#pragma unroll
            for (int c = 0; c < QC; ++c)
#pragma unroll
                for (int r = 0; r < SG_RB; ++r)
                    acc[r][quarter * QC + c] = F::mfma(af[r][0], bf[c][NSB - 1], acc[r][quarter * QC + c]);
        }
#pragma unroll
        for (int c = 0; c < QC; ++c)
#pragma unroll
            for (int r = 0; r < SG_RB; ++r) acc[r][quarter * QC + c] = F::mfma(af[r][1], bf[c][0], acc[r][quarter * QC + c]);
#pragma unroll
        for (int c = 0; c < QC; ++c)
#pragma unroll
            for (int r = 0; r < SG_RB; ++r) acc[r][quarter * QC + c] = F::mfma(af[r][0], bf[c][0], acc[r][quarter * QC + c]);
    };
    // the order of a quarter for the scheduler: matrix-core instructions with RK_SG_FILL vector instructions (the split
    // of a K tile) behind each - a matrix-core instruction holds the issue port for 8 of its 16 cycles
    auto quarter_order = [&](int first, int last, auto fill_tag) {
        if constexpr (!RK_SG_SGB) return;
#pragma unroll
        for (int g = first; g < last; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, decltype(fill_tag)::value, 0);
        }
    };
    // vector instructions (the split of a K tile) behind each matrix-core instruction of the first / second half of a phase M
    constexpr int FILL_Av = RK_SG_LATE ? 0 : (RK_SG_FILL >= 0 ? RK_SG_FILL : (NSB == 2 ? 0 : 1));
    constexpr int FILL_Bv = RK_SG_LATE ? RK_SG_LATE : FILL_Av;
    using FILL_A = std::integral_constant<int, FILL_Av>;
    using FILL_B = std::integral_constant<int, FILL_Bv>;

#if RK_SG_TRACE
    unsigned long long* tr = reinterpret_cast<unsigned long long*>(smem + 2 * NS * ABYTES + 4 * BTILE);
    const bool tracing = NSB == 1 && NCB == 8 && blockIdx.x == SG_TRACE_BLOCK && lane == 0 && (wave & 3) == 0;
    bool trace_phase = false;
    auto stamp = [&](int64_t q, int point) {
        if (tracing && trace_phase && q >= SG_TRACE_Q0 && q < SG_TRACE_Q0 + 16)
            tr[((q - SG_TRACE_Q0) * SG_TRACE_PTS + point) * 2 + grp] = clock64();
    };
#else
    auto stamp = [&](int64_t, int) {};
#endif
    if (grp == 1 && !(RK_PH_PROBE & 8)) sg_bar();  // the second group runs one barrier behind the first from here on

    auto run_phase = [&](auto load, auto touch, auto jstep, auto waiter, auto split, auto store, auto& s0, auto& s1, auto& s2,
                         auto& s3, int64_t nst) {
        if (nst <= 0) return;
        using Wt = decltype(waiter);
        constexpr int NA = Wt::NA;
        auto at = [&](int64_t q) { return q < nst ? q : nst - 1; };  // (past the end: the last step again, unused)
        // K tile t lives in register set t % 4, V tile t in ring slot t % 4
        dma_b(jstep(0), 0);
        dma_b(jstep(at(1)), 1);
        if (grp == 1) dma_b(jstep(at(2)), 2);
        load(0, s0);
        load(at(1), s1);
        load(at(2), s2);
        load(at(3), s3);
        Wt::template at<2 * NA>(s0);
        store(0, split(s0));
        Wt::template at<2 * NA>(s1);         // V tiles 0, 1 and K tiles 0, 1 (K tiles 2, 3 stay in flight)
        auto pk = split(s1);                 // (goes to LDS in L(0))
        Wt::template at<0>(s2);
        Wt::template at<0>(s3);
        sg_lgkm0();
        sg_bar();
        sg_bar();
        // one step: tile q (K buffer q % 2, V slot q % 4); st_load <- K tile q + 4 (the set of tile q), st_split = K tile
        // q + 2, st_next = K tile q + 3 (certified by this step's wait, split in the next step)
        auto step = [&](int64_t q, auto qtag, auto& st_load, auto& st_split, auto& st_next) {
            constexpr int qm = decltype(qtag)::value;  // q % 4
            constexpr int buf = qm & 1, slot = qm;
            const int slot2 = (qm + 2 + grp) & 3;   // the V tile this wave's DMAs fetch: q + 2 (group 0), q + 3 (group 1)
            // ---- L(q): issue only
            stamp(q, 0);
            if constexpr (RK_SG_PF > 0) touch(at(q + RK_SG_PF));
            if constexpr (RK_SG_VMEM_FIRST) {
                dma_b(jstep(at(q + 2 + grp)), slot2);
                if constexpr (!(RK_PH_PROBE & 16)) load((RK_PH_PROBE & 32) ? (q & 7) : at(q + 4), st_load);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!(RK_PH_PROBE & 4)) store(buf ^ 1, pk);     // K tile q + 1 (own rows)
            read_a(buf);
            read_bq(slot, 0, bfa);
            read_bq(slot, 1, bfb);
            if constexpr (!RK_SG_VMEM_FIRST) {
                dma_b(jstep(at(q + 2 + grp)), slot2);
                if constexpr (!(RK_PH_PROBE & 16)) load((RK_PH_PROBE & 32) ? (q & 7) : at(q + 4), st_load);  // (bit 5: cached tiles)
            }
            if constexpr (RK_SG_SPLIT_IN_L) {  // the split of K tile q + 2 in the wave's idle time in front of the barrier
                __builtin_amdgcn_sched_barrier(0);
                pk = split(st_split);
                if constexpr (RK_PH_PROBE & 4) pk = decltype(pk){};
                sg_pin(pk);
            }
            sg_bar();
            // ---- M(q): quarter 0 | reads of quarter 2 | quarter 1 | part of quarter 2 | reads of quarter 3 | rest
#if RK_SG_TRACE
            stamp(q, 1);
            sg_lgkm0();
            stamp(q, 2);
            __builtin_amdgcn_sched_barrier(0);
#endif
            if constexpr (RK_SG_PRIO) __builtin_amdgcn_s_setprio(RK_SG_PRIO);
            mma_q(0, bfa);
            if constexpr (!RK_SG_SPLIT_IN_L) {
                pk = split(st_split);
                if constexpr (RK_PH_PROBE & 4) pk = decltype(pk){};
            }
            read_bq(slot, 2, bfa);
            mma_q(1, bfb);
            mma_q(2, bfa);
            read_bq(slot, 3, bfb);
            mma_q(3, bfb);
            // the order for the scheduler (the reads of quarter 3 a few instructions into quarter 2: the wait in front of
            // quarter 2 then covers quarter 2's own reads only)
            constexpr int QM = (NSB + 1) * QC * SG_RB;  // matrix-core instructions of a quarter
            constexpr int QH = QM / (NSB + 1);          // (the first of its NSB + 1 passes)
            quarter_order(0, QM, FILL_A{});
            if constexpr (RK_SG_SGB) __builtin_amdgcn_sched_group_barrier(0x100, QC * NSB, 0);
            quarter_order(0, QM, FILL_A{});
            quarter_order(0, QH, FILL_B{});
            if constexpr (RK_SG_SGB) __builtin_amdgcn_sched_group_barrier(0x100, QC * NSB, 0);
            quarter_order(QH, QM, FILL_B{});
            quarter_order(0, QM, FILL_B{});
            sg_pin(pk);
            if constexpr (RK_SG_PRIO) __builtin_amdgcn_s_setprio(0);
            if (((int)q & fold_mask) == fold_mask) fold();
            // the waits of the step, behind its matrix-core work: K tile q + 3 and the DMAs issued in L(q - 1) have landed
            // (what L(q) issued stays in flight), this wave's LDS stores of L(q) are done
            __builtin_amdgcn_sched_barrier(0);
            stamp(q, 3);
            Wt::template at<NA + NBQ + (RK_SG_PF > 0 ? 1 : 0)>(st_next);
            stamp(q, 4);
            sg_lgkm0();
            stamp(q, 5);
            sg_bar();
        };
        using Q0 = std::integral_constant<int, 0>;
        using Q1 = std::integral_constant<int, 1>;
        using Q2 = std::integral_constant<int, 2>;
        using Q3 = std::integral_constant<int, 3>;
        for (int64_t q = 0; q < nst; q += 4) {
            step(q, Q0{}, s0, s2, s3);
            if (q + 1 < nst) step(q + 1, Q1{}, s1, s3, s0);
            if (q + 2 < nst) step(q + 2, Q2{}, s2, s0, s1);
            if (q + 3 < nst) step(q + 3, Q3{}, s3, s1, s2);
        }
        // the loads past the end: the next phase counts from zero, no register is reused under a load
        Wt::template at<0>(s0);
        Wt::template at<0>(s1);
        Wt::template at<0>(s2);
        Wt::template at<0>(s3);
    };
    {
        if constexpr (RK_PH_PROBE & 64) {  // (bit 6: the transposed steps with the direct phase's loads and stores: their cost)
            StageD s0, s1, s2, s3;
            auto load_dp = [&](int64_t q, StageD& st) { load_d(nD > 0 ? q % nD : 0, st); };  // (in-bounds addresses only)
            if (nD > 0) run_phase(load_dp, touch_d, jstep_t, SgWaitD{}, split_d, store_d, s0, s1, s2, s3, nT);
        } else {
            StageT s0, s1, s2, s3;
            run_phase(load_t, touch_t, jstep_t, SgWaitT{}, split_t, store_t, s0, s1, s2, s3, nT);
        }
        asm volatile("" : "+v"(pfreg));
    }
    {
        StageD s0, s1, s2, s3;
#if RK_SG_TRACE
        trace_phase = true;
#endif
        run_phase(load_d, touch_d, jstep_d, SgWaitD{}, split_d, store_d, s0, s1, s2, s3, nD);
        asm volatile("" : "+v"(pfreg));
    }
    if (grp == 0 && !(RK_PH_PROBE & 8)) sg_bar();
#if RK_SG_TRACE
    if (NSB == 1 && NCB == 8 && blockIdx.x == SG_TRACE_BLOCK) {
        sg_lgkm0();
        sg_bar();
        if (tid < 16 * SG_TRACE_PTS * 2) sg_trace_buf[tid] = tr[tid];
    }
#endif

    const int64_t kpad = nct * TN;
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int64_t col = ct * TN + wc * (TN / 2) + c * 16 + lr;
        const float unscale = __uint_as_float(info[2 * kpad + col]);
#pragma unroll
        for (int r = 0; r < SG_RB; ++r)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t row = i0 + wr * 16 * SG_RB + r * 16 + 4 * lq + v;
                if (row < L.n && col < k) Y[row * ldy + col] = (tot[r][c][v] + acc[r][c][v]) * unscale;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same wide product with the K tile COMPUTED in the kernel (matrix-free operators: no n x n storage, no
// temporary strips):   Y (n1 x r) = scale * K(X1, X2) V,   float32, d <= 64, kernels smooth in the squared distance.
//
// Structure of symgemm_kernel (128 rows x TN columns of Y per workgroup, contraction in steps of 32 points of X2,
// V tiles from the pre-split image, two LDS buffers, one barrier per step, two-level accumulation) with the staging
// of a K tile replaced by its evaluation, flash-style:
//   * X1 and X2 are centred, scaled, split into two float16 pieces and written once per call as the 64-point images
//     of rk_kernelbf16.hip (gram_split_kernel: |x'|^2 <= 32 is the caller's bound, the same guard as the matrix-free
//     product);
//   * per step a wave owns 16 rows i of the tile: S^T[j][i] = X2_tile X1_rows^T by 12 v_mfma_f32_16x16x32_f16 (two
//     16-point blocks x two 32-feature steps x the three exact piece products; the X1 operand stays in 16 registers,
//     the X2 tile comes from a three-slot LDS ring that is filled one step ahead), kappa(|x|^2 + |y|^2 - 2 S) on the
//     accumulators - element v of a lane is (j = 4 (lane / 16) + v, i = lane % 16): four consecutive j of one row -
//     then scale, split and ONE 8-byte LDS store per piece and block into the A-operand image the main product reads;
//   * 12 + 48 matrix-core instructions per wave and step instead of 48, ~60 more vector instructions than the split of
//     loaded entries.  Every workgroup evaluates its own tiles: with r = 512 (two column tiles) every entry is
//     evaluated twice - still far cheaper than writing it (the strip-subset route: 80 ms of evaluation + five partial
//     products with a tail each).
// ------------------------------------------------------------------------------------------------------------------
constexpr int KG_IMG = 16384, KG_SPLIT = 8192;  // one 64-point image: two pieces of 64 points x 64 features (float16)
constexpr int KG_TILE = 2 * 4096;               // one 32-point tile of it: two pieces x 32 points x 128 B
constexpr int KG_SLOT = KG_TILE + 128;          // + |y'|^2 of the 32 points
constexpr float KG_UNSCALE = 1.f / 1048576.f;   // the pieces hold x' 2^10
__host__ __device__ __forceinline__ int kg_chunk(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

void gram_prep_launch(const float* center, const float* ell_vec, int64_t d, float* iw, float* muw, hipStream_t st);
void gram_split64_launch(const float* X, int64_t ldx, const int64_t* idx, int64_t n, int64_t npad, int64_t d,
                         const float* iw, const float* muw, unsigned char* img, float* norms, hipStream_t st);

template <int NCB, int KIND, int NSB>  // NSB: pieces of V (1: V on the float16 grid, two products per pair)
__global__ void __launch_bounds__(SG_THREADS, 2)
kgemm_kernel(const unsigned char* __restrict__ ximg1, const float* __restrict__ xn1, int64_t n1,
             const unsigned char* __restrict__ ximg2, const float* __restrict__ xn2, int64_t n2, float w, float kscale,
             float sa, const unsigned char* __restrict__ img, int64_t nct, const unsigned* __restrict__ info, int64_t k,
             float* __restrict__ Y, int64_t ldy, int fold_mask) {
    using F = SgF16;
    constexpr int NS = 2;
    constexpr int TN = 32 * NCB;
    constexpr int ABYTES = SG_TM * SG_ROWB;
    constexpr int BBYTES = TN * SG_ROWB;
    constexpr int NBQ = NSB * BBYTES / 16 / SG_THREADS;
    constexpr bool RBF = KIND == RK_RBF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* AL = smem;                               // [2][NS][ABYTES]
    unsigned char* BL = smem + 2 * NS * ABYTES;             // [2][NSB][BBYTES]
    unsigned char* XJ = BL + 2 * NSB * BBYTES;              // [3][KG_SLOT]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t grp = (int64_t)blockIdx.x / (8 * nct);
    const int64_t rem = (int64_t)blockIdx.x % (8 * nct);
    const int64_t rb = grp * 8 + (rem & 7);  // (column tiles of a row block 8 ids apart: one XCD, they share the X2 / V tiles)
    const int64_t ct = rem >> 3;
    const int64_t i0 = rb * SG_TM;
    if (i0 >= n1) return;  // (uniform, before any barrier)
    const int64_t nst = (n2 + SG_KT - 1) / SG_KT;
    const float rbf_c = -0.5f * w * w * 1.4426950408889634f;  // exp(-s w^2 / 2) = exp2(rbf_c s)
    const float m2c = -2.f * rbf_c * KG_UNSCALE;
    const float pscale = kscale * sa;  // (the image of K is scaled below 2^14 by sa = 2^ea, like the stored strips)

    // this wave's 16 rows of X1 as B operand of S^T (2 feature steps x 2 pieces), and their |x'|^2
    sg_u4 xi[2][2];
    float ni;
    {
        const int64_t ip = i0 + 16 * wave + lr;  // < round_up(n1, 128): the image exists
        const unsigned char* base = ximg1 + (ip >> 6) * KG_IMG;
        const int r = (int)(ip & 63);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* src = base + kg_chunk(r, 4 * ks + lq);
            xi[ks][0] = *reinterpret_cast<const sg_u4*>(src);
            xi[ks][1] = *reinterpret_cast<const sg_u4*>(src + KG_SPLIT);
        }
        ni = xn1[ip] * (RBF ? rbf_c : 1.f);
    }

    sg_acc acc[SG_RB][NCB], tot[SG_RB][NCB];
#pragma unroll
    for (int r = 0; r < SG_RB; ++r)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int v = 0; v < 4; ++v) { acc[r][c][v] = 0.f; tot[r][c][v] = 0.f; }
    auto fold = [&]() {
#pragma unroll
        for (int r = 0; r < SG_RB; ++r)
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                tot[r][c] += acc[r][c];
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[r][c][v] = 0.f;
            }
    };

    struct Stage {
        sg_u4 b[NBQ];
        sg_u4 xj;   // this thread's 16 bytes of the X2 tile of the step AFTER the stage's
        float nj;   // |y'|^2 of point tid & 31 of that tile
    };
    Stage sA, sB;
    auto clampq = [&](int64_t q) { return q < nst ? q : nst - 1; };
    // X2 tile t: image t / 2, its points 32 (t % 2) .. + 32; piece tid / 256, 16-byte chunk tid % 256 of the piece's 4 KiB
    auto load_xj = [&](int64_t t, sg_u4& xj, float& nj) {
        t = clampq(t);
        const unsigned char* src = ximg2 + (t >> 1) * KG_IMG + (tid >> 8) * KG_SPLIT + (int)(t & 1) * 4096 + (tid & 255) * 16;
        xj = *reinterpret_cast<const sg_u4*>(src);
        nj = xn2[t * SG_KT + (tid & 31)];
    };
    auto store_xj = [&](int64_t t, const sg_u4& xj, float nj) {
        unsigned char* slot = XJ + (int)(t % 3) * KG_SLOT;
        *reinterpret_cast<sg_u4*>(slot + (tid >> 8) * 4096 + (tid & 255) * 16) = xj;
        if (tid < 32) reinterpret_cast<float*>(slot + KG_TILE)[tid] = nj;
    };
    auto load_c = [&](int64_t q, Stage& st) {
        q = clampq(q);
        const sg_u4* src = reinterpret_cast<const sg_u4*>(img + ((q * nct + ct) * NSB) * (int64_t)BBYTES);
#pragma unroll
        for (int e = 0; e < NBQ; ++e) st.b[e] = src[tid + SG_THREADS * e];
        load_xj(q + 1, st.xj, st.nj);
    };
    // stage step q: X2 tile q + 1 into the ring, the V tile into BL[buf], and the K tile of step q - evaluated from ring
    // slot q % 3, which an earlier stage filled before the last barrier - into AL[buf]
    auto store_c = [&](int buf, int64_t q, const Stage& st) {
        store_xj(q + 1, st.xj, st.nj);
        sg_u4* dstb = reinterpret_cast<sg_u4*>(BL + buf * NSB * BBYTES);
#pragma unroll
        for (int e = 0; e < NBQ; ++e) dstb[tid + SG_THREADS * e] = st.b[e];
        const unsigned char* slot = XJ + (int)(q % 3) * KG_SLOT;
        unsigned char* al = AL + buf * NS * ABYTES;
        const int row = 16 * wave + lr;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            sg_acc sj;
#pragma unroll
            for (int v = 0; v < 4; ++v) sj[v] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char* src = slot + kg_chunk(16 * jb + lr, 4 * ks + lq);
                const sg_u4 ah = *reinterpret_cast<const sg_u4*>(src);
                const sg_u4 am = *reinterpret_cast<const sg_u4*>(src + 4096);
                sj = F::mfma(am, xi[ks][0], sj);
                sj = F::mfma(ah, xi[ks][1], sj);
                sj = F::mfma(ah, xi[ks][0], sj);
            }
            const float4 nj4 = *reinterpret_cast<const float4*>(slot + KG_TILE + (16 * jb + 4 * lq) * 4);
            const float njv[4] = {nj4.x, nj4.y, nj4.z, nj4.w};
            float p[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if constexpr (RBF) {
                    p[v] = __builtin_amdgcn_exp2f(fmaf(sj[v], m2c, ni + njv[v] * rbf_c));
                } else {
                    float s = ni + njv[v] - (2.f * KG_UNSCALE) * sj[v];
                    s = s > 0.f ? s : 0.f;
                    p[v] = kernel_value<float>(KIND, s, w);
                }
                p[v] *= pscale;
            }
            unsigned pk[2][NS];
            F::split2(p[0], p[1], pk[0]);
            F::split2(p[2], p[3], pk[1]);
            // columns j = 16 jb + 4 lq .. + 3 of tile row `row`: 16-byte chunk 2 jb + lq / 2, its half lq % 2
            unsigned char* dst = al + row * SG_ROWB + (sg_swz(row, 2 * jb + (lq >> 1)) << 4) + (lq & 1) * 8;
#pragma unroll
            for (int pc = 0; pc < NS; ++pc) {
                sg_u2 o;
                o.x = pk[0][pc]; o.y = pk[1][pc];
                *reinterpret_cast<sg_u2*>(dst + pc * ABYTES) = o;
            }
        }
    };
    const int a_off = (wr * 16 * SG_RB + lr) * SG_ROWB + (sg_swz(lr, lq) << 4);
    const int b_off = (wc * (TN / 2) + lr) * SG_ROWB + (sg_swz(lr, lq) << 4);
    auto compute = [&](int buf) {
        const unsigned char* al = AL + buf * NS * ABYTES + a_off;
        const unsigned char* bl = BL + buf * NSB * BBYTES + b_off;
        sg_u4 af[SG_RB][NS];
#pragma unroll
        for (int r = 0; r < SG_RB; ++r)
#pragma unroll
            for (int pc = 0; pc < NS; ++pc)
                af[r][pc] = *reinterpret_cast<const sg_u4*>(al + r * 16 * SG_ROWB + pc * ABYTES);
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            sg_u4 bf[NSB];
#pragma unroll
            for (int pc = 0; pc < NSB; ++pc) bf[pc] = *reinterpret_cast<const sg_u4*>(bl + c * 16 * SG_ROWB + pc * BBYTES);
#pragma unroll
            for (int r = 0; r < SG_RB; ++r) {
                acc[r][c] = F::mfma(af[r][1], bf[0], acc[r][c]);
                if constexpr (NSB == 2) acc[r][c] = F::mfma(af[r][0], bf[NSB - 1], acc[r][c]);
                acc[r][c] = F::mfma(af[r][0], bf[0], acc[r][c]);
            }
        }
    };

    // ---- pipeline (run_phase of symgemm_kernel with the step index passed to the staging) ------------------------------
    {
        sg_u4 x0;
        float n0;
        load_xj(0, x0, n0);
        store_xj(0, x0, n0);
    }
    __syncthreads();  // ring slot 0 holds X2 tile 0
    load_c(0, sA);
    store_c(0, 0, sA);
    load_c(1, sA);
    load_c(2, sB);
    __syncthreads();
    for (int64_t q = 0; q < nst; q += 2) {
        store_c(1, q + 1, sA);
        load_c(q + 3, sA);
        compute(0);
        if (((int)q & fold_mask) == fold_mask) fold();
        __syncthreads();
        if (q + 1 < nst) {
            store_c(0, q + 2, sB);
            load_c(q + 4, sB);
            compute(1);
            if (((int)(q + 1) & fold_mask) == fold_mask) fold();
            __syncthreads();
        }
    }
    const int64_t kpad = nct * TN;
    // (the unscaling alpha 2^-(ea + eb) rides in info[2 kpad + col])
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int64_t col = ct * TN + wc * (TN / 2) + c * 16 + lr;
        const float unscale = __uint_as_float(info[2 * kpad + col]);
#pragma unroll
        for (int r = 0; r < SG_RB; ++r)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t row = i0 + wr * 16 * SG_RB + r * 16 + 4 * lq + v;
                if (row < n1 && col < k) Y[row * ldy + col] = (tot[r][c][v] + acc[r][c][v]) * unscale;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The packed triangle WRITTEN from the matrix cores (float32, d <= 64, kernels smooth in the squared distance).
//
// sympack_materialize_kernel (rk_symm.hip) forms the squared distances feature by feature on the vector units: 250 G entries
// a second whatever the kernel function and whether the float32 operations are packed or not (profiles/r04_materialize.log) -
// 80 ms for the 80 GB triangle of n = 200 000, 1 TB/s of writes.  Here the entries come as kgemm_kernel evaluates them: the
// points are centred, scaled, split into two float16 pieces once (the 64-point images of rk_kernelbf16.hip, |x'|^2 <= 32
// vouched for by the caller's centre), S^T = X_cols X_rows^T by three exact v_mfma_f32_16x16x32_f16 per 16 x 16 block and
// 32-feature step, the kernel function on the accumulators - element v of a lane is (column 4 (lane / 16) + v, row lane % 16):
// four consecutive columns of one row, one 16-byte store.  A workgroup keeps the four images of a strip's 256 columns in LDS
// (64 KiB) and walks up to RI rows of the strip, eight waves x 16 rows at a time.  Entry error: that of the matrix-free
// product (2^-22 |x'||y'| in the dot product), at the float32 rounding level for |x'|^2 of order 1.
// ------------------------------------------------------------------------------------------------------------------
constexpr int MC_RI = 1024;  // rows of a strip per workgroup
template <int KIND>
__global__ void __launch_bounds__(SG_THREADS, 4)
sympack_mc_kernel(const unsigned char* __restrict__ ximg, const float* __restrict__ xn, SgLayout L, float w, float kscale,
                  float* __restrict__ Kp) {
    constexpr int W = SgLayout::W;
    constexpr bool RBF = KIND == RK_RBF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* XJ = smem;                                      // [4][KG_IMG]: the strip's 256 points
    float* NJ = reinterpret_cast<float*>(smem + 4 * KG_IMG);       // their |x'|^2 (RBF: times rbf_c)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t t = blockIdx.y;
    const int64_t sg = L.first + t * L.stride;                     // global strip
    const int64_t nrows = (sg + 1) * W < L.n ? (sg + 1) * W : L.n;
    const int64_t ibeg = (int64_t)blockIdx.x * MC_RI;
    if (ibeg >= nrows) return;  // (uniform, before any barrier)
    const int64_t iend = ibeg + MC_RI < nrows ? ibeg + MC_RI : nrows;
    const float rbf_c = -0.5f * w * w * 1.4426950408889634f;       // exp(-s w^2 / 2) = exp2(rbf_c s)
    const float m2c = -2.f * rbf_c * KG_UNSCALE;
    {
        const sg_u4* src = reinterpret_cast<const sg_u4*>(ximg + sg * 4 * (int64_t)KG_IMG);
        sg_u4* dst = reinterpret_cast<sg_u4*>(XJ);
#pragma unroll
        for (int e = 0; e < 4 * KG_IMG / 16 / SG_THREADS; ++e) dst[tid + SG_THREADS * e] = src[tid + SG_THREADS * e];
        if (tid < W) NJ[tid] = xn[sg * W + tid] * (RBF ? rbf_c : 1.f);
    }
    __syncthreads();
    float* strip = Kp + L.row_offset(t) * W;
    for (int64_t i0 = ibeg; i0 < iend; i0 += SG_TM) {
        const int64_t ip = i0 + 16 * wave + lr;  // (< round_up(n, 256): the image exists)
        sg_u4 xi[2][2];
        {
            const unsigned char* base = ximg + (ip >> 6) * KG_IMG;
            const int r = (int)(ip & 63);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char* src = base + kg_chunk(r, 4 * ks + lq);
                xi[ks][0] = *reinterpret_cast<const sg_u4*>(src);
                xi[ks][1] = *reinterpret_cast<const sg_u4*>(src + KG_SPLIT);
            }
        }
        const float ni = xn[ip] * (RBF ? rbf_c : 1.f);
        const bool live = ip < nrows;
        float* drow = strip + ip * W + 4 * lq;
#pragma unroll 4
        for (int jb = 0; jb < 16; ++jb) {
            sg_acc sj;
#pragma unroll
            for (int v = 0; v < 4; ++v) sj[v] = 0.f;
            const unsigned char* img = XJ + (jb >> 2) * KG_IMG;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char* src = img + kg_chunk(16 * (jb & 3) + lr, 4 * ks + lq);
                const sg_u4 ah = *reinterpret_cast<const sg_u4*>(src);
                const sg_u4 am = *reinterpret_cast<const sg_u4*>(src + KG_SPLIT);
                sj = SgF16::mfma(am, xi[ks][0], sj);
                sj = SgF16::mfma(ah, xi[ks][1], sj);
                sj = SgF16::mfma(ah, xi[ks][0], sj);
            }
            const float4 nj4 = *reinterpret_cast<const float4*>(NJ + 16 * jb + 4 * lq);
            const float njv[4] = {nj4.x, nj4.y, nj4.z, nj4.w};
            float p[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if constexpr (RBF) {
                    p[v] = __builtin_amdgcn_exp2f(fmaf(sj[v], m2c, ni + njv[v]));
                } else {
                    float s = ni + njv[v] - (2.f * KG_UNSCALE) * sj[v];
                    s = s > 0.f ? s : 0.f;
                    p[v] = kernel_value<float>(KIND, s, w);
                }
                p[v] = (sg * W + 16 * jb + 4 * lq + v < L.n) ? p[v] * kscale : 0.f;  // (columns past n: zeros, as the vector-unit kernel writes)
            }
            if (live) {
                sg_acc o;
                o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3];
                __builtin_nontemporal_store(o, reinterpret_cast<sg_acc*>(drow + 16 * jb));
            }
        }
    }
}

// The same evaluation into a ROW-MAJOR block K (n1 x n2, leading dimension ldk): K(X1[idx1], X2[idx2]) for the ASkotch
// blocks (rlaopt/solvers/sap.py:64-66) and row-major shards.  A workgroup keeps 256 points of X2 in LDS and walks `ri`
// rows of X1.  16-byte stores where the row alignment allows, element stores otherwise (and at the ragged edge).
template <int KIND>
__global__ void __launch_bounds__(SG_THREADS, 4)
kmat_mc_kernel(const unsigned char* __restrict__ ximg1, const float* __restrict__ xn1, int64_t n1,
               const unsigned char* __restrict__ ximg2, const float* __restrict__ xn2, int64_t n2, float w, float kscale,
               float* __restrict__ K, int64_t ldk, int64_t ri, int vec_ok) {
    constexpr int W = 256;
    constexpr bool RBF = KIND == RK_RBF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* XJ = smem;
    float* NJ = reinterpret_cast<float*>(smem + 4 * KG_IMG);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t cg = blockIdx.y;
    const int64_t ibeg = (int64_t)blockIdx.x * ri;
    if (ibeg >= n1) return;  // (uniform, before any barrier)
    const int64_t iend = ibeg + ri < n1 ? ibeg + ri : n1;
    const float rbf_c = -0.5f * w * w * 1.4426950408889634f;
    const float m2c = -2.f * rbf_c * KG_UNSCALE;
    {
        const sg_u4* src = reinterpret_cast<const sg_u4*>(ximg2 + cg * 4 * (int64_t)KG_IMG);
        sg_u4* dst = reinterpret_cast<sg_u4*>(XJ);
#pragma unroll
        for (int e = 0; e < 4 * KG_IMG / 16 / SG_THREADS; ++e) dst[tid + SG_THREADS * e] = src[tid + SG_THREADS * e];
        if (tid < W) NJ[tid] = xn2[cg * W + tid] * (RBF ? rbf_c : 1.f);
    }
    __syncthreads();
    for (int64_t i0 = ibeg; i0 < iend; i0 += SG_TM) {
        const int64_t ip = i0 + 16 * wave + lr;  // (< round_up(n1, 256): the image exists)
        sg_u4 xi[2][2];
        {
            const unsigned char* base = ximg1 + (ip >> 6) * KG_IMG;
            const int r = (int)(ip & 63);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char* src = base + kg_chunk(r, 4 * ks + lq);
                xi[ks][0] = *reinterpret_cast<const sg_u4*>(src);
                xi[ks][1] = *reinterpret_cast<const sg_u4*>(src + KG_SPLIT);
            }
        }
        const float ni = xn1[ip] * (RBF ? rbf_c : 1.f);
        const bool live = ip < n1;
        float* drow = K + ip * ldk + cg * W + 4 * lq;
#pragma unroll 4
        for (int jb = 0; jb < 16; ++jb) {
            sg_acc sj;
#pragma unroll
            for (int v = 0; v < 4; ++v) sj[v] = 0.f;
            const unsigned char* img = XJ + (jb >> 2) * KG_IMG;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char* src = img + kg_chunk(16 * (jb & 3) + lr, 4 * ks + lq);
                const sg_u4 ah = *reinterpret_cast<const sg_u4*>(src);
                const sg_u4 am = *reinterpret_cast<const sg_u4*>(src + KG_SPLIT);
                sj = SgF16::mfma(am, xi[ks][0], sj);
                sj = SgF16::mfma(ah, xi[ks][1], sj);
                sj = SgF16::mfma(ah, xi[ks][0], sj);
            }
            const float4 nj4 = *reinterpret_cast<const float4*>(NJ + 16 * jb + 4 * lq);
            const float njv[4] = {nj4.x, nj4.y, nj4.z, nj4.w};
            sg_acc o;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float pv;
                if constexpr (RBF) {
                    pv = __builtin_amdgcn_exp2f(fmaf(sj[v], m2c, ni + njv[v]));
                } else {
                    float s = ni + njv[v] - (2.f * KG_UNSCALE) * sj[v];
                    s = s > 0.f ? s : 0.f;
                    pv = kernel_value<float>(KIND, s, w);
                }
                o[v] = pv * kscale;
            }
            const int64_t j = cg * W + 16 * jb + 4 * lq;
            if (live) {
                if (vec_ok && j + 3 < n2) {
                    __builtin_nontemporal_store(o, reinterpret_cast<sg_acc*>(drow + 16 * jb));
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (j + v < n2) drow[16 * jb + v] = o[v];
                }
            }
        }
    }
}

static bool sg_use_bf16x3() {
    const char* e = getenv("RK_SYMGEMM_SPLIT");
    return e != nullptr && (e[0] == 'b' || e[0] == '3');
}
static bool sg_use_phases() {  // symgemm_ph_kernel (two wave groups half a step apart) unless RK_SYMGEMM_IMPL=lockstep
    const char* e = getenv("RK_SYMGEMM_IMPL");
    return e == nullptr || e[0] != 'l';
}
static bool sg_force_phases() {  // RK_SYMGEMM_IMPL=ph: the phase kernel for every shape (tests)
    const char* e = getenv("RK_SYMGEMM_IMPL");
    return e != nullptr && e[0] == 'p';
}
static int sg_tn(int64_t k) { return k <= 128 ? 128 : 256; }
static size_t sg_image_bytes(int64_t n, int64_t k, int ns) {
    const int64_t npad = round_up(n, SgLayout::W);
    const int64_t kpad = round_up(k, sg_tn(k));
    return (size_t)(npad / SG_KT) * (size_t)kpad * SG_ROWB * ns;
}

}  // namespace rk

using namespace rk;

extern "C" size_t rk_symm_gemm_ws_bytes(int64_t n, int64_t k) {
    if (n <= 0 || k <= 0) return 0;
    return SG_INFO_BYTES + sg_image_bytes(n, k, sg_use_bf16x3() ? 3 : 2);
}

static int symm_gemm_impl(const void* Kp, int64_t n, int64_t strip_first, int64_t strip_stride, double kmax, const void* V,
                          int64_t k, int64_t ldv, double alpha, void* Y, int64_t ldy, void* ws, size_t ws_bytes, void* stream,
                          int rhs_pieces) {
    clear_error();
    RK_REQUIRE(Kp && V && Y, "rk_symm_gemm: null pointer");
    RK_REQUIRE(k <= 1024, "rk_symm_gemm: at most 1024 columns per call (got %lld)", (long long)k);
    RK_REQUIRE(n > 0 && k > 0 && ldv >= k && ldy >= k, "rk_symm_gemm: bad sizes (n=%lld k=%lld ldv=%lld ldy=%lld)",
               (long long)n, (long long)k, (long long)ldv, (long long)ldy);
    RK_REQUIRE(strip_stride >= 1 && strip_first >= 0 && strip_first < strip_stride, "rk_symm_gemm: bad strip distribution");
    RK_REQUIRE(kmax > 0.0 && kmax < 1e30, "rk_symm_gemm: kmax must bound the stored entries (got %g)", kmax);
    RK_REQUIRE(aligned16(Kp), "rk_symm_gemm: the packed matrix must be 16-byte aligned");
    const bool bf = sg_use_bf16x3() && rhs_pieces != 1;
    const size_t need = rk_symm_gemm_ws_bytes(n, k);
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_symm_gemm: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned* info = static_cast<unsigned*>(ws);
    unsigned char* img = static_cast<unsigned char*>(ws) + SG_INFO_BYTES;
    const int tn = sg_tn(k);
    const int64_t npad = round_up(n, SgLayout::W), kpad = round_up(k, tn), nct = kpad / tn;
    // K entries <= kmax: scaled by 2^ea they stay below 2^14 (float16 pieces in their normal range)
    int ea = 0;
    if (!bf) {
        int ex;
        frexp(kmax, &ex);
        ea = 14 - ex;
    }
    if (hipMemsetAsync(info, 0, SG_INFO_BYTES, st) != hipSuccess) return launch_status("rk_symm_gemm(memset)");
    if (!bf)
        hipLaunchKernelGGL(sg_absmax_kernel, dim3((unsigned)ceil_div(k, 64), 512), dim3(256), 0, st, (const float*)V, n,
                           k, ldv, info);
    hipLaunchKernelGGL(sg_scale_kernel, dim3((unsigned)ceil_div(kpad, 256)), dim3(256), 0, st, info, kpad, ea,
                       (float)alpha, bf ? 0 : 1);
    const dim3 sgrid((unsigned)ceil_div((npad / 8) * kpad, 256));
    if (bf)
        hipLaunchKernelGGL(sg_split_rhs_kernel<SgBf16>, sgrid, dim3(256), 0, st, (const float*)V, n, npad, k, kpad, ldv, tn,
                           info, img);
    else if (rhs_pieces == 1)
        hipLaunchKernelGGL((sg_split_rhs_kernel<SgF16, 1>), sgrid, dim3(256), 0, st, (const float*)V, n, npad, k, kpad, ldv,
                           tn, info, img);
    else
        hipLaunchKernelGGL(sg_split_rhs_kernel<SgF16>, sgrid, dim3(256), 0, st, (const float*)V, n, npad, k, kpad, ldv, tn,
                           info, img);
    int rc = launch_status("rk_symm_gemm(split)");
    if (rc != RK_OK) return rc;
    SgLayout L{n, strip_first, strip_stride};
    const int64_t nrb = ceil_div(n, SG_TM);
    const dim3 grid((unsigned)(ceil_div(nrb, 8) * 8 * nct));
    const float sa = (float)ldexp(1.0, ea);
    int fold_steps = SG_FOLD;
    if (const char* e = getenv("RK_SYMGEMM_FOLD")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 4096 && (v & (v - 1)) == 0) fold_steps = v;
    }
    const int fold_mask = fold_steps - 1;
#define RK_SG(FMT, NCBV)                                                                                         \
    do {                                                                                                         \
        constexpr size_t smem = 2 * FMT::NS * (SG_TM * SG_ROWB + 32 * NCBV * SG_ROWB);                           \
        static std::atomic<unsigned long long> attr_devs{0}; /* one bit per device ordinal */                      \
        int dev_ = 0;                                                                                            \
        (void)hipGetDevice(&dev_);                                                                               \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                     \
        if (!(attr_devs.load(std::memory_order_acquire) & bit_)) {                                               \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(symgemm_kernel<FMT, NCBV>),                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {      \
                (void)hipGetLastError();                                                                         \
                set_error("rk_symm_gemm: cannot reserve %zu bytes of LDS", smem);                                \
                return RK_ERR_LAUNCH;                                                                            \
            }                                                                                                    \
            attr_devs.fetch_or(bit_, std::memory_order_release);                                                 \
        }                                                                                                        \
        hipLaunchKernelGGL((symgemm_kernel<FMT, NCBV>), grid, dim3(SG_THREADS), smem, st, (const float*)Kp, L, img, nct, \
                           sa, info, k, (float*)Y, ldy, fold_mask);                                                      \
    } while (0)
#define RK_SGPH(NCBV, NSBV)                                                                                      \
    do {                                                                                                         \
        constexpr size_t smem = 2 * 2 * SG_TM * SG_ROWB + 4 * NSBV * 32 * NCBV * SG_ROWB + (RK_SG_TRACE && NSBV == 1 ? 4096 : 0); \
        static std::atomic<unsigned long long> attr_devs{0};                                                     \
        int dev_ = 0;                                                                                            \
        (void)hipGetDevice(&dev_);                                                                               \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                     \
        if (!(attr_devs.load(std::memory_order_acquire) & bit_)) {                                               \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(symgemm_ph_kernel<NCBV, NSBV>),                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {      \
                (void)hipGetLastError();                                                                         \
                set_error("rk_symm_gemm: cannot reserve %zu bytes of LDS", smem);                                \
                return RK_ERR_LAUNCH;                                                                            \
            }                                                                                                    \
            attr_devs.fetch_or(bit_, std::memory_order_release);                                                 \
        }                                                                                                        \
        hipLaunchKernelGGL((symgemm_ph_kernel<NCBV, NSBV>), grid, dim3(SG_THREADS), smem, st, (const float*)Kp, L, img, nct, \
                           sa, info, k, (float*)Y, ldy, fold_mask);                                                    \
    } while (0)
    if (!bf && rhs_pieces == 1) {
        if (tn == 256) RK_SGPH(8, 1);
        else RK_SGPH(4, 1);
    } else if (!bf && sg_use_phases() && (tn == 256 || sg_force_phases())) {
        // (128-column tiles, two pieces: the lockstep kernel is 3 % faster - 43.4 against 44.8 ms at r = 100)
        if (tn == 256) RK_SGPH(8, 2);
        else RK_SGPH(4, 2);
    } else if (bf) {
        if (tn == 256) RK_SG(SgBf16, 8);
        else RK_SG(SgBf16, 4);
    } else {
        if (tn == 256) RK_SG(SgF16, 8);
        else RK_SG(SgF16, 4);
    }
#undef RK_SG
#undef RK_SGPH
    return launch_status("rk_symm_gemm");
}

extern "C" int rk_symm_gemm(const void* Kp, int64_t n, int64_t strip_first, int64_t strip_stride, double kmax,
                            const void* V, int64_t k, int64_t ldv, double alpha, void* Y, int64_t ldy, void* ws,
                            size_t ws_bytes, void* stream) {
    return symm_gemm_impl(Kp, n, strip_first, strip_stride, kmax, V, k, ldv, alpha, Y, ldy, ws, ws_bytes, stream, 2);
}

extern "C" int rk_symm_gemm_rhs16(const void* Kp, int64_t n, int64_t strip_first, int64_t strip_stride, double kmax,
                                  const void* V, int64_t k, int64_t ldv, double alpha, void* Y, int64_t ldy, void* ws,
                                  size_t ws_bytes, void* stream) {
    return symm_gemm_impl(Kp, n, strip_first, strip_stride, kmax, V, k, ldv, alpha, Y, ldy, ws, ws_bytes, stream, 1);
}

#if RK_SG_TRACE
extern "C" int rk_debug_symgemm_trace(unsigned long long* out) {  // [step 16][point][group]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sg_trace_buf), 16 * SG_TRACE_PTS * 2 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

// ---- the fused evaluate-and-multiply product (matrix-free operators) -------------------------------------------------
static size_t kg_x_bytes(int64_t n) {  // |x'|^2 of the padded points + their split images
    const int64_t npad = round_up(n, 256);
    return (size_t)round_up(npad * (int64_t)sizeof(float), 256) + (size_t)(npad / 64) * KG_IMG;
}

static size_t mc_ws_bytes(int64_t n) { return 1024 + kg_x_bytes(n); }  // centre / scale vectors + |x'|^2 + split images

extern "C" size_t rk_kernel_materialize_sympack_mc_ws_bytes(int64_t n) { return n > 0 ? mc_ws_bytes(n) : 0; }

extern "C" int rk_kernel_materialize_sympack_mc(int kind, const void* X, int64_t n, int64_t ldx, int64_t d, double ell_scalar,
                                                const void* ell_vec, const void* center, double scale,
                                                int64_t strip_first, int64_t strip_stride, void* Kp, void* ws,
                                                size_t ws_bytes, void* stream) {
    clear_error();
    RK_REQUIRE(X && Kp && center, "rk_kernel_materialize_sympack_mc: null pointer (the centre is required)");
    RK_REQUIRE(kind == RK_RBF || kind == RK_MATERN32 || kind == RK_MATERN52,
               "rk_kernel_materialize_sympack_mc: kernels smooth in the squared distance only (rbf, matern32, matern52)");
    RK_REQUIRE(n > 0 && d > 0 && d <= 64 && ldx >= d, "rk_kernel_materialize_sympack_mc: n > 0, 1 <= d <= 64");
    RK_REQUIRE(strip_stride >= 1 && strip_first >= 0 && strip_first < strip_stride,
               "rk_kernel_materialize_sympack_mc: bad strip distribution");
    RK_REQUIRE(ell_vec != nullptr || ell_scalar > 0.0, "rk_kernel_materialize_sympack_mc: lengthscale must be positive");
    RK_REQUIRE(aligned16(Kp), "rk_kernel_materialize_sympack_mc: output must be 16-byte aligned");
    const size_t need = mc_ws_bytes(n);
    if (ws == nullptr || ws_bytes < need || !aligned16(ws)) {
        set_error("rk_kernel_materialize_sympack_mc: workspace %zu B < required %zu B (16-byte aligned)", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    SgLayout L{n, strip_first, strip_stride};
    const int64_t T_loc = L.strips_local();
    if (T_loc == 0) return RK_OK;
    float* iw = static_cast<float*>(ws);
    float* muw = iw + 128;
    const int64_t npad = round_up(n, 256);
    float* xn = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + 1024);
    unsigned char* ximg = reinterpret_cast<unsigned char*>(xn) + round_up(npad * (int64_t)sizeof(float), 256);
    gram_prep_launch((const float*)center, (const float*)ell_vec, d, iw, muw, st);
    gram_split64_launch((const float*)X, ldx, nullptr, n, npad, d, iw, muw, ximg, xn, st);
    int rc = launch_status("rk_kernel_materialize_sympack_mc(split)");
    if (rc != RK_OK) return rc;
    const int64_t s_last = L.first + (T_loc - 1) * L.stride;
    const int64_t max_rows = (s_last + 1) * SgLayout::W < n ? (s_last + 1) * SgLayout::W : n;
    const dim3 grid((unsigned)ceil_div(max_rows, MC_RI), (unsigned)T_loc);
    const float w = ell_vec ? 1.f : (float)(1.0 / ell_scalar);
    constexpr size_t smem = 4 * KG_IMG + SgLayout::W * sizeof(float);
#define RK_MC(KINDV)                                                                                              \
    do {                                                                                                          \
        static std::atomic<unsigned long long> attr_devs{0};                                                      \
        int dev_ = 0;                                                                                             \
        (void)hipGetDevice(&dev_);                                                                                \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                      \
        if (!(attr_devs.load(std::memory_order_acquire) & bit_)) {                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(sympack_mc_kernel<KINDV>),                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {       \
                (void)hipGetLastError();                                                                          \
                set_error("rk_kernel_materialize_sympack_mc: cannot reserve %zu bytes of LDS", smem);             \
                return RK_ERR_LAUNCH;                                                                             \
            }                                                                                                     \
            attr_devs.fetch_or(bit_, std::memory_order_release);                                                  \
        }                                                                                                         \
        hipLaunchKernelGGL((sympack_mc_kernel<KINDV>), grid, dim3(SG_THREADS), smem, st, ximg, xn, L, w,          \
                           (float)scale, (float*)Kp);                                                             \
    } while (0)
    if (kind == RK_RBF) RK_MC(RK_RBF);
    else if (kind == RK_MATERN32) RK_MC(RK_MATERN32);
    else RK_MC(RK_MATERN52);
#undef RK_MC
    return launch_status("rk_kernel_materialize_sympack_mc");
}

extern "C" size_t rk_kernel_materialize_mc_ws_bytes(int64_t n1, int64_t n2) {
    return (n1 > 0 && n2 > 0) ? 1024 + kg_x_bytes(n1) + kg_x_bytes(n2) : 0;
}

extern "C" int rk_kernel_materialize_mc(int kind, const void* X1, int64_t n1, int64_t ldx1, const int64_t* idx1, const void* X2,
                                        int64_t n2, int64_t ldx2, const int64_t* idx2, int64_t d, double ell_scalar,
                                        const void* ell_vec, const void* center, double scale, void* K, int64_t ldk,
                                        void* ws, size_t ws_bytes, void* stream) {
    clear_error();
    RK_REQUIRE(X1 && X2 && K && center, "rk_kernel_materialize_mc: null pointer (the centre is required)");
    RK_REQUIRE(kind == RK_RBF || kind == RK_MATERN32 || kind == RK_MATERN52,
               "rk_kernel_materialize_mc: kernels smooth in the squared distance only (rbf, matern32, matern52)");
    RK_REQUIRE(n1 > 0 && n2 > 0 && d > 0 && d <= 64 && ldx1 >= d && ldx2 >= d && ldk >= n2,
               "rk_kernel_materialize_mc: bad sizes (n1=%lld n2=%lld d=%lld ldk=%lld)", (long long)n1, (long long)n2,
               (long long)d, (long long)ldk);
    RK_REQUIRE(ell_vec != nullptr || ell_scalar > 0.0, "rk_kernel_materialize_mc: lengthscale must be positive");
    const size_t need = rk_kernel_materialize_mc_ws_bytes(n1, n2);
    if (ws == nullptr || ws_bytes < need || !aligned16(ws)) {
        set_error("rk_kernel_materialize_mc: workspace %zu B < required %zu B (16-byte aligned)", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* iw = static_cast<float*>(ws);
    float* muw = iw + 128;
    const bool same = X1 == X2 && idx1 == idx2 && n1 == n2 && ldx1 == ldx2;
    const int64_t n1pad = round_up(n1, 256), n2pad = round_up(n2, 256);
    float* xn1 = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + 1024);
    unsigned char* ximg1 = reinterpret_cast<unsigned char*>(xn1) + round_up(n1pad * (int64_t)sizeof(float), 256);
    float* xn2 = same ? xn1 : reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(xn1) + kg_x_bytes(n1));
    unsigned char* ximg2 = same ? ximg1 : reinterpret_cast<unsigned char*>(xn2) + round_up(n2pad * (int64_t)sizeof(float), 256);
    gram_prep_launch((const float*)center, (const float*)ell_vec, d, iw, muw, st);
    gram_split64_launch((const float*)X1, ldx1, idx1, n1, n1pad, d, iw, muw, ximg1, xn1, st);
    if (!same) gram_split64_launch((const float*)X2, ldx2, idx2, n2, n2pad, d, iw, muw, ximg2, xn2, st);
    int rc = launch_status("rk_kernel_materialize_mc(split)");
    if (rc != RK_OK) return rc;
    // rows per workgroup: 1024, fewer for small blocks (at least ~1 000 workgroups in the grid)
    const int64_t cgs = ceil_div(n2, 256);
    int64_t ri = 1024;
    while (ri > 128 && ceil_div(n1, ri) * cgs < 1024) ri /= 2;
    const dim3 grid((unsigned)ceil_div(n1, ri), (unsigned)cgs);
    const float w = ell_vec ? 1.f : (float)(1.0 / ell_scalar);
    const int vec_ok = (ldk % 4 == 0) && aligned16(K);
    constexpr size_t smem = 4 * KG_IMG + 256 * sizeof(float);
#define RK_MC(KINDV)                                                                                              \
    do {                                                                                                          \
        static std::atomic<unsigned long long> attr_devs{0};                                                      \
        int dev_ = 0;                                                                                             \
        (void)hipGetDevice(&dev_);                                                                                \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                      \
        if (!(attr_devs.load(std::memory_order_acquire) & bit_)) {                                                \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kmat_mc_kernel<KINDV>),                         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {       \
                (void)hipGetLastError();                                                                          \
                set_error("rk_kernel_materialize_mc: cannot reserve %zu bytes of LDS", smem);                     \
                return RK_ERR_LAUNCH;                                                                             \
            }                                                                                                     \
            attr_devs.fetch_or(bit_, std::memory_order_release);                                                  \
        }                                                                                                         \
        hipLaunchKernelGGL((kmat_mc_kernel<KINDV>), grid, dim3(SG_THREADS), smem, st, ximg1, xn1, n1, ximg2, xn2, \
                           n2, w, (float)scale, (float*)K, ldk, ri, vec_ok);                                      \
    } while (0)
    if (kind == RK_RBF) RK_MC(RK_RBF);
    else if (kind == RK_MATERN32) RK_MC(RK_MATERN32);
    else RK_MC(RK_MATERN52);
#undef RK_MC
    return launch_status("rk_kernel_materialize_mc");
}

extern "C" size_t rk_kernel_sketch_ws_bytes(int64_t n1, int64_t n2, int64_t k) {
    if (n1 <= 0 || n2 <= 0 || k <= 0) return 0;
    return SG_INFO_BYTES + sg_image_bytes(n2, k, 2) + 1024 + kg_x_bytes(n1) + kg_x_bytes(n2);
}

static int kernel_sketch_impl(int kind, const void* X1, int64_t n1, int64_t ldx1, const void* X2, int64_t n2, int64_t ldx2,
                              int64_t d, double ell_scalar, const void* ell_vec, const void* center, double scale,
                              const void* V, int64_t k, int64_t ldv, double alpha, void* Y, int64_t ldy, void* ws,
                              size_t ws_bytes, void* stream, int rhs_pieces) {
    clear_error();
    RK_REQUIRE(X1 && X2 && V && Y && center, "rk_kernel_sketch_matmat: null pointer (the centre is required)");
    RK_REQUIRE(kind == RK_RBF || kind == RK_MATERN32 || kind == RK_MATERN52,
               "rk_kernel_sketch_matmat: kernels smooth in the squared distance only (rbf, matern32, matern52)");
    RK_REQUIRE(d > 0 && d <= 64 && ldx1 >= d && ldx2 >= d, "rk_kernel_sketch_matmat: 1 <= d <= 64");
    RK_REQUIRE(n1 > 0 && n2 > 0 && k > 0 && k <= 1024 && ldv >= k && ldy >= k,
               "rk_kernel_sketch_matmat: bad sizes (n1=%lld n2=%lld k=%lld)", (long long)n1, (long long)n2, (long long)k);
    RK_REQUIRE(ell_vec != nullptr || ell_scalar > 0.0, "rk_kernel_sketch_matmat: lengthscale must be positive");
    RK_REQUIRE(scale != 0.0 && fabs(scale) < 1e30, "rk_kernel_sketch_matmat: scale must be finite and non-zero");
    const size_t need = rk_kernel_sketch_ws_bytes(n1, n2, k);
    if (ws == nullptr || ws_bytes < need || !aligned16(ws)) {
        set_error("rk_kernel_sketch_matmat: workspace %zu B < required %zu B (16-byte aligned)", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char* base = static_cast<unsigned char*>(ws);
    unsigned* info = reinterpret_cast<unsigned*>(base);
    unsigned char* img = base + SG_INFO_BYTES;
    float* iw = reinterpret_cast<float*>(img + sg_image_bytes(n2, k, 2));
    float* muw = iw + 128;
    const bool same = X1 == X2 && n1 == n2 && ldx1 == ldx2;
    const int64_t n1pad = round_up(n1, 256), n2pad = round_up(n2, 256);
    float* xn1 = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(iw) + 1024);
    unsigned char* ximg1 = reinterpret_cast<unsigned char*>(xn1) + round_up(n1pad * (int64_t)sizeof(float), 256);
    float* xn2 = same ? xn1 : reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(xn1) + kg_x_bytes(n1));
    unsigned char* ximg2 = same ? ximg1 : reinterpret_cast<unsigned char*>(xn2) + round_up(n2pad * (int64_t)sizeof(float), 256);
    gram_prep_launch((const float*)center, (const float*)ell_vec, d, iw, muw, st);
    gram_split64_launch((const float*)X1, ldx1, nullptr, n1, n1pad, d, iw, muw, ximg1, xn1, st);
    if (!same) gram_split64_launch((const float*)X2, ldx2, nullptr, n2, n2pad, d, iw, muw, ximg2, xn2, st);
    // the split image of V and the column scales: as rk_symm_gemm (entries of K are bounded by |scale|)
    const int tn = sg_tn(k);
    const int64_t npadv = round_up(n2, SgLayout::W), kpad = round_up(k, tn), nct = kpad / tn;
    int ex;
    frexp(fabs(scale), &ex);
    const int ea = 14 - ex;
    if (hipMemsetAsync(info, 0, SG_INFO_BYTES, st) != hipSuccess) return launch_status("rk_kernel_sketch_matmat(memset)");
    hipLaunchKernelGGL(sg_absmax_kernel, dim3((unsigned)ceil_div(k, 64), 512), dim3(256), 0, st, (const float*)V, n2, k,
                       ldv, info);
    hipLaunchKernelGGL(sg_scale_kernel, dim3((unsigned)ceil_div(kpad, 256)), dim3(256), 0, st, info, kpad, ea, (float)alpha, 1);
    if (rhs_pieces == 1)
        hipLaunchKernelGGL((sg_split_rhs_kernel<SgF16, 1>), dim3((unsigned)ceil_div((npadv / 8) * kpad, 256)), dim3(256), 0,
                           st, (const float*)V, n2, npadv, k, kpad, ldv, tn, info, img);
    else
        hipLaunchKernelGGL(sg_split_rhs_kernel<SgF16>, dim3((unsigned)ceil_div((npadv / 8) * kpad, 256)), dim3(256), 0, st,
                           (const float*)V, n2, npadv, k, kpad, ldv, tn, info, img);
    int rc = launch_status("rk_kernel_sketch_matmat(split)");
    if (rc != RK_OK) return rc;
    const int64_t nrb = ceil_div(n1, SG_TM);
    const dim3 grid((unsigned)(ceil_div(nrb, 8) * 8 * nct));
    const float w = ell_vec ? 1.f : (float)(1.0 / ell_scalar);
    const float sa = (float)ldexp(1.0, ea);
    int fold_steps = SG_FOLD;
    if (const char* e = getenv("RK_SYMGEMM_FOLD")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 4096 && (v & (v - 1)) == 0) fold_steps = v;
    }
    const int fold_mask = fold_steps - 1;
#define RK_KG(NCBV, KINDV, NSBV)                                                                                    \
    do {                                                                                                            \
        constexpr size_t smem = 2 * (2 * SG_TM * SG_ROWB + NSBV * 32 * NCBV * SG_ROWB) + 3 * KG_SLOT;               \
        static std::atomic<unsigned long long> attr_devs{0};                                                        \
        int dev_ = 0;                                                                                               \
        (void)hipGetDevice(&dev_);                                                                                  \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                        \
        if (!(attr_devs.load(std::memory_order_acquire) & bit_)) {                                                  \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kgemm_kernel<NCBV, KINDV, NSBV>),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {         \
                (void)hipGetLastError();                                                                            \
                set_error("rk_kernel_sketch_matmat: cannot reserve %zu bytes of LDS", smem);                        \
                return RK_ERR_LAUNCH;                                                                               \
            }                                                                                                       \
            attr_devs.fetch_or(bit_, std::memory_order_release);                                                    \
        }                                                                                                           \
        hipLaunchKernelGGL((kgemm_kernel<NCBV, KINDV, NSBV>), grid, dim3(SG_THREADS), smem, st, ximg1, xn1, n1, ximg2, xn2, n2, \
                           w, (float)scale, sa, img, nct, info, k, (float*)Y, ldy, fold_mask);                      \
    } while (0)
#define RK_KG_KIND(NCBV, NSBV)                                                \
    do {                                                                      \
        if (kind == RK_RBF) RK_KG(NCBV, RK_RBF, NSBV);                        \
        else if (kind == RK_MATERN32) RK_KG(NCBV, RK_MATERN32, NSBV);         \
        else RK_KG(NCBV, RK_MATERN52, NSBV);                                  \
    } while (0)
    if (rhs_pieces == 1) {
        if (tn == 256) RK_KG_KIND(8, 1);
        else RK_KG_KIND(4, 1);
    } else {
        if (tn == 256) RK_KG_KIND(8, 2);
        else RK_KG_KIND(4, 2);
    }
#undef RK_KG_KIND
#undef RK_KG
    return launch_status("rk_kernel_sketch_matmat");
}

extern "C" int rk_kernel_sketch_matmat(int kind, const void* X1, int64_t n1, int64_t ldx1, const void* X2, int64_t n2,
                                       int64_t ldx2, int64_t d, double ell_scalar, const void* ell_vec,
                                       const void* center, double scale, const void* V, int64_t k, int64_t ldv,
                                       double alpha, void* Y, int64_t ldy, void* ws, size_t ws_bytes, void* stream) {
    return kernel_sketch_impl(kind, X1, n1, ldx1, X2, n2, ldx2, d, ell_scalar, ell_vec, center, scale, V, k, ldv, alpha, Y,
                              ldy, ws, ws_bytes, stream, 2);
}

extern "C" int rk_kernel_sketch_matmat_rhs16(int kind, const void* X1, int64_t n1, int64_t ldx1, const void* X2, int64_t n2,
                                             int64_t ldx2, int64_t d, double ell_scalar, const void* ell_vec,
                                             const void* center, double scale, const void* V, int64_t k, int64_t ldv,
                                             double alpha, void* Y, int64_t ldy, void* ws, size_t ws_bytes,
                                             void* stream) {
    return kernel_sketch_impl(kind, X1, n1, ldx1, X2, n2, ldx2, d, ell_scalar, ell_vec, center, scale, V, k, ldv, alpha, Y,
                              ldy, ws, ws_bytes, stream, 1);
}
