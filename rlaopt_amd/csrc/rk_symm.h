// Shared pieces of the strip-packed symmetric kernels (rk_symm.hip, rk_symmblk.hip): the packed layout and the
// float16-split helpers of the 16-bit matrix-core products.
#pragma once

#include "rk_common.h"

namespace rk {

struct SymLayout {
    int64_t n;
    int W;
    int64_t first, stride;

    __host__ __device__ int64_t strips_total() const { return (n + W - 1) / W; }
    __host__ __device__ int64_t strips_local() const {
        const int64_t S = strips_total();
        return first < S ? (S - first + stride - 1) / stride : 0;
    }
    __host__ __device__ int64_t strip(int64_t t) const { return first + t * stride; }
    __host__ __device__ int64_t rows(int64_t t) const {
        const int64_t r = (strip(t) + 1) * W;
        return r < n ? r : n;
    }
    // stored rows before local strip t (only the globally last strip is clipped, and it can
    // only be the last local one, so every earlier strip s' has (s'+1) W rows)
    __host__ __device__ int64_t row_offset(int64_t t) const {
        return (int64_t)W * (t * (first + 1) + stride * (t * (t - 1) / 2));
    }
    __host__ __device__ int64_t total_rows() const {
        const int64_t T = strips_local();
        return T ? row_offset(T - 1) + rows(T - 1) : 0;
    }
};

template <typename T>
constexpr int strip_width() { return kWave * Vec16<T>::N; }

constexpr int SMK = 16;  // right-hand sides carried by the matrix-core strip kernels (k is padded up to it)

template <typename T>
inline SymLayout make_layout(int64_t n, int64_t first, int64_t stride) {
    SymLayout L;
    L.n = n;
    L.W = strip_width<T>();
    L.first = first;
    L.stride = stride;
    return L;
}

// ---- float16 split (x 2^e = h + m + O(2^-22 |x|)) and the 16-bit matrix-core products built on it ----------------
constexpr int H2_TR = 32;                 // rows per tile
constexpr int H2_ROWB = 544;              // bytes of one image row (256 halves + 32 bytes of padding)
// byte offset of 16-column block c16 (0 .. 15) of image row `row`: the blocks of rows with bit 2 != bit 3 are
// XOR-ed with 4 (found by exhaustive search over strides and XOR masks: with it the operand fetch of the row part
// (ds_read_b128, 16 rows x 4 chunks per lane group), the transposed fetch of the column part (ds_read_b64_tr_b16, rows
// q and 8 + q in one 32-lane half) and the 8-byte stores are ALL bank-conflict free)
__host__ __device__ __forceinline__ int h2_off(int row, int c16) {
    return row * H2_ROWB + ((c16 ^ (4 * (((row >> 3) ^ (row >> 2)) & 1))) << 5);
}
constexpr int H2_PIECE = H2_TR * H2_ROWB;  // one piece of one tile: 16 896 bytes
constexpr int H2_FOLD = 8;                // tiles between folds of the column accumulators
typedef unsigned int h2_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int h2_u2 __attribute__((ext_vector_type(2)));
typedef float h2_acc __attribute__((ext_vector_type(4)));
typedef _Float16 h2_h8 __attribute__((ext_vector_type(8)));
typedef short h2_s4 __attribute__((ext_vector_type(4)));
constexpr size_t h2_smem_bytes() {  // 2 buffers x 2 pieces + row parts of the 2 column halves (2 buffers)
    return (size_t)(2 * 2 * H2_PIECE) + (size_t)(2 * 2 * H2_TR * SMK) * sizeof(float);
}

__device__ __forceinline__ void h2_split2(float a, float b, unsigned& h, unsigned& m) {
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;   // round to nearest even
    const float ra = a - (float)ha, rb = b - (float)hb;  // exact
    const _Float16 ma = (_Float16)ra, mb = (_Float16)rb;
    h = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
    m = (unsigned)__builtin_bit_cast(unsigned short, ma) | ((unsigned)__builtin_bit_cast(unsigned short, mb) << 16);
}
__device__ __forceinline__ h2_acc h2_mfma(h2_u4 a, h2_u4 b, h2_acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h2_h8, a), __builtin_bit_cast(h2_h8, b), c, 0, 0, 0);
}
// h.h + h.m + m.h, small terms first
__device__ __forceinline__ h2_acc h2_mfma3(const h2_u4 (&a)[2], const h2_u4 (&b)[2], h2_acc c) {
    c = h2_mfma(a[1], b[0], c);
    c = h2_mfma(a[0], b[1], c);
    return h2_mfma(a[0], b[0], c);
}


// scale / split V for the float16-split products (rk_symm.hip): info[0..15] column maxima (scratch), info[16 + c] =
// 2^eb(c), info[32 + c] = the unscaling 2^-(ea + eb(c)); Vt[piece][rhs (16)][j (npad)] float16.  Returns ea through
// *ea_out (stored entries <= kmax are scaled by 2^ea).
int h2_prepare_v(const float* V, int64_t n, int64_t npad, int64_t k, int64_t ldv, double kmax, unsigned* info,
                 unsigned short* Vt, int* ea_out, hipStream_t st);

// the block form of the float16-split product (rk_symmblk.hip); same contract as symm_matmat_h2_impl
size_t symm_h2blk_ws_bytes(int64_t n, int64_t k, int64_t first, int64_t stride);
int symm_matmat_h2blk_impl(const float* Kp, int64_t n, int64_t first, int64_t stride, double kmax, const float* V,
                           int64_t k, int64_t ldv, double alpha, double shift, float* Y, int64_t ldy, void* ws,
                           size_t ws_bytes, hipStream_t st);

// the vector-unit block kernel (rk_symmblk.hip): storage TS, vectors T (float/float, float/double, double/double), k <= 2
template <typename TS, typename T>
size_t symm_blkreg_ws_bytes(int64_t n, int64_t k, int64_t first, int64_t stride);
template <typename TS, typename T>
int symm_matmat_blkreg_impl(const TS* Kp, int64_t n, int64_t first, int64_t stride, const T* V, int64_t k, int64_t ldv,
                            double alpha, double shift, T* Y, int64_t ldy, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace rk
