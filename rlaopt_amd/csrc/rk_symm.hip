// Symmetric kernel matrices K(X, X): strip-packed upper-triangular storage and the product
// that reads every stored entry once and uses it twice.
//
// Layout ("sympack").  Columns are cut into strips of W = 64 x 16 B / sizeof(T) columns
// (256 fp32 / 128 fp64).  Strip s stores rows 0 .. min(n, (s+1) W) - 1 of its W columns - the
// part on and above its diagonal block, the diagonal block in full - row-major with leading
// dimension W (every stored row is exactly one 1 KiB wave load; columns past n are zero).
// A rank stores the strips s = first, first + stride, ... (cyclic distribution: balanced to
// within one strip); total elements ~ n^2 / (2 stride): HALF of the row-major shard.
//
// Product y = K v for symmetric K.  A wave walks down a strip: for a stored row i of strip s
//     y[i]       += sum_j K[i][j] v[j]      (j in the strip)   - a wave reduction per row
//     y[j]       += K[i][j] v[i]            (i above the diagonal block only) - per-lane sums
// The per-row wave reduction is 4 v_add_f32_dpp + 2 row_bcast steps (no LDS traffic), the
// column sums stay in registers for the whole item.  Row results go to a (strip, row) table,
// column results to an (item, column) table; a finishing kernel adds them in a fixed order
// (bitwise reproducible, no atomics) and applies alpha and the "+ reg v" term.
// HBM traffic: the packed triangle once (+ ~1 % of partial tables) instead of the full square.
#include <atomic>
#include <stdlib.h>

#include "rk_common.h"
#include "rk_kerneltile.h"
#include "rk_symm.h"

namespace rk {

// tuning knobs (overridable at build time for experiments: tools/variants.sh + tools/ab.py)
#ifndef RK_SYMM_ITEM_ROWS
#define RK_SYMM_ITEM_ROWS 1024
#endif
#ifndef RK_SYMM_RB
#define RK_SYMM_RB 8
#endif
#ifndef RK_SYMM_RB_WIDE
#define RK_SYMM_RB_WIDE 8  // rows per batch when the per-lane accumulators are wide (K >= 2 ...)
#endif
#ifndef RK_SYMM_WG_PER_CU
#define RK_SYMM_WG_PER_CU 6  // persistent grid = CUs x this (0: ask the occupancy calculator)
#endif
constexpr int ITEM_ROWS = RK_SYMM_ITEM_ROWS;  // rows of a strip handled by one workgroup (4 waves)
constexpr int SYMM_THREADS = 4 * kWave;
static_assert(ITEM_ROWS % (4 * RK_SYMM_RB) == 0, "a wave's rows must be whole batches");
static_assert(ITEM_ROWS % 256 == 0, "items must be whole multiples of the strip widths");

// ------------------------------------------------------------------------------------------
// materialise the packed triangle
// ------------------------------------------------------------------------------------------
template <typename T, typename TS, bool L1, int DCV>
__global__ void __launch_bounds__(256)
sympack_materialize_kernel(int kind, const T* __restrict__ X, int64_t ldx, int64_t d, T w,
                           const T* __restrict__ ell_vec, T scale, SymLayout L,
                           TS* __restrict__ Kp) {
    constexpr int W = strip_width<TS>();
    constexpr int CT = W / TN;  // column tiles per strip
    __shared__ T xs[DCV][LDP];
    __shared__ T ys[DCV][LDP];
    const int64_t t = blockIdx.y;
    const int64_t nrows = L.rows(t);
    const int64_t i0 = ((int64_t)blockIdx.x / CT) * TM;
    if (i0 >= nrows) return;  // uniform: before any barrier
    const int jt = (int)(blockIdx.x % CT);
    const int64_t j0 = L.strip(t) * W + (int64_t)jt * TN;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    T acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] = T(0);
    for (int64_t c0 = 0; c0 < d; c0 += DCV) {
        __syncthreads();
        stage_panel<T, DCV>(xs, X, L.n, ldx, nullptr, i0, c0, d, ell_vec);
        stage_panel<T, DCV>(ys, X, L.n, ldx, nullptr, j0, c0, d, ell_vec);
        __syncthreads();
        tile_accumulate<T, L1, DCV>(xs, ys, ty, tx, acc, (int)((d - c0) < DCV ? (d - c0) : DCV));
    }
    TS* base = Kp + L.row_offset(t) * W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + ty * 4 + r;
        if (i >= nrows) continue;
        TS o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t j = j0 + tx * 4 + q;
            o[q] = (j < L.n) ? (TS)(scale * kernel_value<T>(kind, acc[r][q], w)) : TS(0);
        }
        TS* dst = base + i * W + jt * TN + tx * 4;
        if constexpr (sizeof(TS) == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            *reinterpret_cast<double2*>(dst) = make_double2(o[0], o[1]);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(o[2], o[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// the strip product
// ------------------------------------------------------------------------------------------
// TS = storage type of the packed matrix, T = type of the vectors and of all arithmetic
// (TS = float with T = double: "fp32 storage, fp64 iteration" - fp32 HBM traffic, fp64 accuracy)
//
// What was measured on MI355X while tuning this kernel (tools/ab.py, n = 200000, k = 1, fp32):
//  * with the partial-table stores removed it streams at 6.8 TB/s, the rate of the row-major
//    kernel; the stores (0.5 % of the bytes) cost 5-8 % of the time, and the cost is not
//    proportional to their volume (a quarter of the stores still costs 3 %), nor to which wave
//    issues them (a dedicated writer wave changed nothing): reads and writes interleave badly
//    at the HBM controller.  Larger per-workgroup tiles would shrink the tables, but 1 MiB
//    items are already what the tail of a 1280-workgroup grid allows; a design with a fifth of
//    the stores (4 strips per workgroup, column sums carried down a strip) was emulated by
//    dropping the other stores: 2.1 % before its own costs - not pursued.
//  * item order matters: ids interleaved over the persistent workgroups (the resident grid
//    sweeps one contiguous window) beat a contiguous id range per workgroup by 1.2 % and a
//    row-block-major order by 5 %.
//  * real double buffering (below) and the removal of the FMA / DPP work are both within 1 %.
//  * writing an item's tables at the START of the next item (second set of LDS buffers, the stores in
//    flight together with that item's first loads, one barrier per item instead of two) changes
//    nothing either: 12.45 vs 12.47 ms (round 2, tools/ab.py) - the cost of the stores is not a wait
//    for their completion at the item boundary.  Float atomics into an n-vector instead of the tables
//    (no HBM writes at all; timing probe): 12.53 vs 12.45 ms - not the write traffic either.
template <typename TS, typename T, int K>
__global__ void __launch_bounds__(SYMM_THREADS)
symm_strip_kernel(const TS* __restrict__ Kp, SymLayout L, const T* __restrict__ V, int64_t ldv,
                  T* __restrict__ Prow, T* __restrict__ Pcol, int64_t max_items,
                  int64_t total_items) {
    using V16 = typename Vec16<TS>::type;
    constexpr int VW = Vec16<TS>::N;
    constexpr int W = strip_width<TS>();
    // rows per batch (loads in flight per stage)
    constexpr int RB = (VW * K * (int)sizeof(T) / 4 <= 4) ? RK_SYMM_RB : RK_SYMM_RB_WIDE;
    __shared__ T rowbuf[ITEM_ROWS * K];
    __shared__ T colbuf[4][W * K];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane_off = (unsigned)lane * 16u;
    // Persistent workgroups: items (strip, 1024-row chunk) are numbered strip by strip; this
    // workgroup takes the ids blockIdx.x, blockIdx.x + gridDim.x, ... (equal-sized items, no
    // empty workgroups for the triangle's short strips).  (t, before) walks forward with the id.
    int64_t t = 0, before = 0;
  for (int64_t id = blockIdx.x; id < total_items; id += gridDim.x) {
    for (;;) {
        const int64_t cnt = (L.rows(t) + ITEM_ROWS - 1) / ITEM_ROWS;
        if (id < before + cnt) break;
        before += cnt;
        ++t;
    }
    const int64_t chunk = id - before;
    const int64_t nrows = uniform64(L.rows(t));
    const int64_t item_lo = uniform64(chunk * ITEM_ROWS);
    const int64_t item_hi = item_lo + ITEM_ROWS < nrows ? item_lo + ITEM_ROWS : nrows;
    const int64_t s = L.strip(t);
    const int64_t diag_lo = uniform64(s * W);  // rows >= diag_lo: the (fully stored) diagonal block
    const TS* base = Kp + uniform64(L.row_offset(t) * W);  // (offset, not the pointer: keeps it global)
    // the item's batches of RB rows (8 KiB each) are dealt round-robin to the 4 waves: the
    // workgroup sweeps its item front to back as one 4-batch-wide window
    const int nb_item = (int)((item_hi - item_lo + RB - 1) / RB);
    const int nbw = wave < nb_item ? (nb_item - wave + 3) / 4 : 0;
    auto batch_of = [&](int b) -> int { return b * 4 + wave; };

    // this lane's VW columns of the strip and their right-hand-side entries
    T vj[VW][K];
#pragma unroll
    for (int q = 0; q < VW; ++q) {
        const int64_t j = diag_lo + lane * VW + q;
#pragma unroll
        for (int c = 0; c < K; ++c) vj[q][c] = (j < L.n) ? V[j * ldv + c] : T(0);
    }
    T col[VW][K];
#pragma unroll
    for (int q = 0; q < VW; ++q)
#pragma unroll
        for (int c = 0; c < K; ++c) col[q][c] = T(0);

    // Two register stages (A, B): while one batch is consumed from one stage the loads of the
    // wave's next batch are already in flight into the other.  The fetch is unconditional inside
    // its basic block, so the compiler waits for the older batch only instead of draining the
    // queue.  V[i] of the batch rides in one vector load (lane r*K + c holds V[i0 + r][c]; read
    // back with readlane), keeping every wait on the in-order vmcnt counter.
    V16 stA[RB], stB[RB];
    T vA, vB;
    auto fetch = [&](V16 a[RB], T& vi, int b) {
        const int64_t i0 = item_lo + (int64_t)batch_of(b) * RB;
        {   // every lane loads (lanes >= RB*K repeat the last entry): no divergent branch
            const int e = lane < RB * K ? lane : RB * K - 1;
            int64_t i = i0 + e / K;
            if (i > nrows - 1) i = nrows - 1;
            vi = V[i * ldv + e % K];
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            int64_t i = i0 + r;
            if (i > nrows - 1) i = nrows - 1;  // clamp; contribution masked below
            a[r] = load_stream_at<V16>(base + i * W, lane_off);
        }
    };
    auto consume = [&](const V16 a[RB], T vi, int b) {
        const int pb = batch_of(b);
        const int64_t i0 = item_lo + (int64_t)pb * RB;
        const bool above = i0 < diag_lo;  // batch-uniform: diag_lo and i0 are multiples of RB
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const bool live = i0 + r < item_hi;
            const TS* as = reinterpret_cast<const TS*>(&a[r]);
            T ae[VW];
#pragma unroll
            for (int q = 0; q < VW; ++q) ae[q] = (T)as[q];
            T dot[K];
#pragma unroll
            for (int c = 0; c < K; ++c) dot[c] = T(0);
#pragma unroll
            for (int q = 0; q < VW; ++q)
#pragma unroll
                for (int c = 0; c < K; ++c) dot[c] += ae[q] * vj[q][c];
            if (above && live) {
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const T vic = lane_value(vi, r * K + c);
#pragma unroll
                    for (int q = 0; q < VW; ++q) col[q][c] += ae[q] * vic;
                }
            }
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const T tot = wave_total_lane63(dot[c]);
                if (lane == 63 && live) rowbuf[(pb * RB + r) * K + c] = tot;
            }
        }
    };
    if (nbw > 0) {
        int b = 0;
        fetch(stA, vA, b);
        for (;;) {
            if (b + 1 >= nbw) { consume(stA, vA, b); break; }
            fetch(stB, vB, b + 1);
            consume(stA, vA, b);
            ++b;
            if (b + 1 >= nbw) { consume(stB, vB, b); break; }
            fetch(stA, vA, b + 1);
            consume(stB, vB, b);
            ++b;
        }
    }
#pragma unroll
    for (int q = 0; q < VW; ++q)
#pragma unroll
        for (int c = 0; c < K; ++c) colbuf[wave][(lane * VW + q) * K + c] = col[q][c];
    __syncthreads();  // rowbuf (written by lane 63) and colbuf are complete
    // row sums of the item's rows -> Prow, coalesced
    T* pr = Prow + (L.row_offset(t) + item_lo) * K;
    // (non-temporal stores: the tables are written once and read once by the finishing kernel;
    // measured 1.4 % faster than plain stores)
    for (int e = threadIdx.x; e < (int)(item_hi - item_lo) * K; e += SYMM_THREADS)
        __builtin_nontemporal_store(rowbuf[e], &pr[e]);
    // column sums of the 4 waves -> Pcol[item]
    T* pc = Pcol + ((t * max_items + chunk) * W) * K;
    for (int e = threadIdx.x; e < W * K; e += SYMM_THREADS) {
        const T sum = ((colbuf[0][e] + colbuf[1][e]) + colbuf[2][e]) + colbuf[3][e];
        __builtin_nontemporal_store(sum, &pc[e]);
    }
    __syncthreads();  // rowbuf / colbuf are reused by the next item
  }
}

// ------------------------------------------------------------------------------------------
// the strip product for 5 .. 16 right-hand sides (float32): both uses of a stored entry on the
// float32 matrix cores
// ------------------------------------------------------------------------------------------
// At 16 right-hand sides every loaded element feeds 32 FMAs: 104 TFLOP/s at the HBM rate - more
// than the vector units deliver, 66 % of the fp32 matrix peak.  A workgroup stages 32-row tiles of
// its item (32 x 256 floats = 32 KiB, double-buffered, register-staged two tiles ahead like
// rk_kernelbf16.hip) in LDS and both products read their v_mfma_f32_16x16x4_f32 A operands from
// there with ds_read_b128 (4 MFMAs per read):
//   row part  Yrow[32 x 16] = T[32 x 256] V[256 x 16]:  lane (row = lane % 16, 4 columns at
//             16 s + 4 (lane / 16)); wave (rb, half) owns 16 rows x 128 columns, its 32 B operands
//             (V of those columns) stay in registers for the whole item; the two column halves are
//             added through LDS and go to the (strip, row) table;
//   col part  Ycol[256 x 16] += T^T[256 x 32] V_i[32 x 16]:  lane (column group = lane % 16, row
//             4 s + lane / 16); wave w owns columns 64 w .. 64 w + 63, 16 accumulator registers for
//             the whole item, written to the (item, column) table at its end.
// The 16-byte chunks of a staged row are XOR-swizzled with the row index: both read patterns (16
// rows x one chunk, one row x 16 chunks) are bank-conflict free.  V is first copied into a zero-
// padded [round_up(n, 256)][16] array so that every operand load is an aligned, unconditional one.
template <typename T, int TR>
constexpr size_t sm_smem_bytes() {  // staged tiles (TR KiB each) + V rows of the tile + row parts of the column slices
    return (size_t)(2 * TR * 1024) + (size_t)(2 * TR * SMK + 2 * (4 / (TR / 16)) * TR * SMK) * sizeof(T);
}

template <typename T>
struct SmAcc;
template <>
struct SmAcc<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ type mfma(float a, float b, type c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // accumulator element v of lane (lq = lane / 16) sits in row m of the 16 x 16 result
    static __device__ __forceinline__ int row(int lq, int v) { return 4 * lq + v; }
};
template <>
struct SmAcc<double> {
    typedef double type __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ type mfma(double a, double b, type c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lq, int v) { return lq + 4 * v; }  // (the f64 layout differs)
};

template <typename T>
__global__ void __launch_bounds__(256)
symm_pad_v_kernel(const T* __restrict__ V, int64_t n, int64_t npad, int k, int64_t ldv, T* __restrict__ Vp) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npad * SMK) return;
    const int64_t i = e / SMK;
    const int c = (int)(e - i * SMK);
    Vp[e] = (i < n && c < k) ? V[i * ldv + c] : T(0);
}

// element index of 16-byte chunk `chunk` (0 .. 63) of staged row `row` (rows are 1 KiB)
template <typename T>
__device__ __forceinline__ int sm_at(int row, int chunk) {
    return (row * 64 + (chunk ^ (row & 15))) * Vec16<T>::N;
}

// TR = rows per tile: 32 (2 row blocks x 2 column halves over the waves) or 16 (1 row block x 4 column
// quarters).  float: TR = 32 at 76 KiB of LDS, 2 workgroups per CU (TR = 16: 4 per CU, 1 % slower);
// double: TR = 16 at 52 KiB (the float64 row parts do not fit beside 32-row tiles twice per CU).
template <typename T, int TR>
__global__ void __launch_bounds__(SYMM_THREADS, (TR == 32 ? 2 : (sizeof(T) == 4 ? 4 : 3)))
symm_strip_mfma_kernel(const T* __restrict__ Kp, SymLayout L, const T* __restrict__ Vp,
                       T* __restrict__ Prow, T* __restrict__ Pcol, int64_t max_items,
                       int64_t total_items) {
    using V16 = typename Vec16<T>::type;
    using Acc = typename SmAcc<T>::type;
    constexpr int VW = Vec16<T>::N;       // elements per 16-byte chunk: 4 / 2
    constexpr int W = 64 * VW;            // strip width: 256 / 128 columns
    constexpr int NRB = TR / 16;          // 16-row blocks per tile
    constexpr int NCS = 4 / NRB;          // column slices of the row part (one per wave and row block)
    constexpr int CSW = W / NCS;          // columns per slice
    constexpr int NS = CSW / (4 * VW);    // ds_read_b128 steps per slice (4 lane groups x VW columns each)
    constexpr int CW = W / 4;             // columns per wave in the column part
    constexpr int RPW = TR / 4;           // staged rows per wave
    constexpr int VPT = TR * SMK / SYMM_THREADS;  // V-tile elements per thread: 2 or 1
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    T* TB = reinterpret_cast<T*>(sm_raw);        // [2][TR][W]          staged tiles (swizzled)
    T* VI = TB + 2 * TR * W;                     // [2][TR][SMK]        V rows of the tile
    T* RP = VI + 2 * TR * SMK;                   // [2][NCS][TR][SMK]   row parts of the column slices
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int rb = wave % NRB, cs = wave / NRB;  // row part: 16 rows x CSW columns per wave
    int64_t t = 0, before = 0;
    for (int64_t id = blockIdx.x; id < total_items; id += gridDim.x) {
        for (;;) {
            const int64_t cnt = (L.rows(t) + ITEM_ROWS - 1) / ITEM_ROWS;
            if (id < before + cnt) break;
            before += cnt;
            ++t;
        }
        const int64_t chunk = id - before;
        const int64_t nrows = uniform64(L.rows(t));
        const int64_t item_lo = uniform64(chunk * ITEM_ROWS);
        const int64_t item_hi = item_lo + ITEM_ROWS < nrows ? item_lo + ITEM_ROWS : nrows;
        const int64_t diag_lo = uniform64(L.strip(t) * W);
        const T* base = Kp + uniform64(L.row_offset(t) * W);
        const int ntile = (int)((item_hi - item_lo + TR - 1) / TR);

        // B operands of the row part: V[diag_lo + CSW cs + 4 VW s + VW lq + q][rhs = lr]
        T vB[NS][VW];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int q = 0; q < VW; ++q)
                vB[s][q] = Vp[(diag_lo + CSW * cs + 4 * VW * s + VW * lq + q) * SMK + lr];
        Acc ccol[VW];
#pragma unroll
        for (int q = 0; q < VW; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v) ccol[q][v] = T(0);

        // staging: per load instruction a wave fetches one whole stored row (1 KiB).  TWO tiles are in
        // flight per workgroup (register sets A / B, the tile loop is unrolled by two): with one, the
        // loads of a tile land in the first half of the previous tile's arithmetic and nothing is
        // outstanding for the rest of it
        V16 preA[RPW], preB[RPW];
        T vpreA[VPT], vpreB[VPT];
        auto load_tile = [&](int tl, V16 (&pre)[RPW], T (&vpre)[VPT]) {
            const int64_t r0 = item_lo + (int64_t)(tl < ntile ? tl : ntile - 1) * TR;
#pragma unroll
            for (int u = 0; u < RPW; ++u) {
                const int64_t i = r0 + wave + 4 * u;
                pre[u] = load_stream_at<V16>(base + (i < nrows ? i : nrows - 1) * W, (unsigned)lane * 16u);
            }
#pragma unroll
            for (int e = 0; e < VPT; ++e) {
                const int at = tid * VPT + e;  // element (row at / 16, rhs at % 16) of the tile's V rows
                const int64_t iv = r0 + at / SMK;
                vpre[e] = Vp[(iv < nrows ? iv : nrows - 1) * SMK + at % SMK];
            }
        };
        auto store_tile = [&](int buf, int tl, const V16 (&pre)[RPW], const T (&vpre)[VPT]) {
            T* tb = TB + buf * TR * W;
            const int64_t r0 = item_lo + (int64_t)(tl < ntile ? tl : ntile - 1) * TR;
#pragma unroll
            for (int u = 0; u < RPW; ++u) {
                const T m = r0 + wave + 4 * u < item_hi ? T(1) : T(0);  // rows past the item: exact zeros
                V16 x = pre[u];
                T* xe = reinterpret_cast<T*>(&x);
#pragma unroll
                for (int e = 0; e < VW; ++e) xe[e] *= m;
                *reinterpret_cast<V16*>(tb + sm_at<T>(wave + 4 * u, lane)) = x;
            }
#pragma unroll
            for (int e = 0; e < VPT; ++e) VI[buf * TR * SMK + tid * VPT + e] = vpre[e];
        };
        auto flush_rows = [&](int buf, int tl) {  // add the column slices, store to the row table
            const T* rp = RP + buf * NCS * TR * SMK;
            T* pr = Prow + (L.row_offset(t) + item_lo + (int64_t)tl * TR) * SMK;
            const int64_t live = (item_hi - item_lo - (int64_t)tl * TR) * SMK;
#pragma unroll
            for (int e = tid; e < TR * SMK; e += SYMM_THREADS) {
                T sum = rp[e];
#pragma unroll
                for (int c = 1; c < NCS; ++c) sum += rp[c * TR * SMK + e];
                if (e < live) __builtin_nontemporal_store(sum, &pr[e]);
            }
        };
        auto compute = [&](int buf, int tl) {
            if (tl > 0) flush_rows(buf ^ 1, tl - 1);  // (RP[buf ^ 1] was completed before the last barrier)
            const T* tb = TB + buf * TR * W;
            // ---- row part: lane (row lr, column group lq), VW columns per read ---------------------
            Acc crow;
#pragma unroll
            for (int v = 0; v < 4; ++v) crow[v] = T(0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const V16 a = *reinterpret_cast<const V16*>(tb + sm_at<T>(16 * rb + lr, (CSW / VW) * cs + 4 * s + lq));
                const T* ae = reinterpret_cast<const T*>(&a);
#pragma unroll
                for (int q = 0; q < VW; ++q) crow = SmAcc<T>::mfma(ae[q], vB[s][q], crow);
            }
            T* rp = RP + (buf * NCS + cs) * TR * SMK;
#pragma unroll
            for (int v = 0; v < 4; ++v) rp[(16 * rb + SmAcc<T>::row(lq, v)) * SMK + lr] = crow[v];
            // ---- column part (rows above the diagonal block only; tile-uniform): lane (column group
            // lr, row lq) ------------------------------------------------------------------------
            if (item_lo + (int64_t)tl * TR < diag_lo) {
                const T* vi = VI + buf * TR * SMK;
#pragma unroll
                for (int s = 0; s < TR / 4; ++s) {
                    const V16 a = *reinterpret_cast<const V16*>(tb + sm_at<T>(4 * s + lq, 16 * wave + lr));
                    const T* ae = reinterpret_cast<const T*>(&a);
                    const T bq = vi[(4 * s + lq) * SMK + lr];
#pragma unroll
                    for (int q = 0; q < VW; ++q) ccol[q] = SmAcc<T>::mfma(ae[q], bq, ccol[q]);
                }
            }
        };

        load_tile(0, preA, vpreA);
        store_tile(0, 0, preA, vpreA);
        load_tile(1, preA, vpreA);
        load_tile(2, preB, vpreB);
        __syncthreads();
        for (int tl = 0; tl < ntile; tl += 2) {
            store_tile(1, tl + 1, preA, vpreA);
            load_tile(tl + 3, preA, vpreA);
            compute(0, tl);
            __syncthreads();  // tile tl + 1 staged, RP[0] complete, everyone done with TB[0]
            if (tl + 1 < ntile) {
                store_tile(0, tl + 2, preB, vpreB);
                load_tile(tl + 4, preB, vpreB);
                compute(1, tl + 1);
                __syncthreads();
            }
        }
        flush_rows((ntile - 1) & 1, ntile - 1);
        // column sums of the item: ccol[q][v] = Ycol[column CW w + VW m + q][rhs = lr], m = row(lq, v)
        T* pc = Pcol + ((t * max_items + chunk) * W) * SMK;
#pragma unroll
        for (int q = 0; q < VW; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                __builtin_nontemporal_store(ccol[q][v], &pc[(CW * wave + VW * SmAcc<T>::row(lq, v) + q) * SMK + lr]);
        __syncthreads();  // RP / TB are reused by the next item
    }
}

// Y[i][c] = alpha * (row parts of every local strip that stores row i
//                    + column parts of the local strip that owns column i) + shift * V[i][c]
// 64 outputs x 16 slices per workgroup: slice y adds the terms y, y + 16, ... (row parts first,
// then column parts, one combined index); slices are combined in a fixed order through LDS.
template <typename T>
__global__ void __launch_bounds__(1024)
symm_finish_kernel(SymLayout L, int K, int KT, const T* __restrict__ Prow, const T* __restrict__ Pcol,
                   int64_t max_items, T alpha, T shift, const T* __restrict__ V, int64_t ldv,
                   T* __restrict__ Y, int64_t ldy,
                   const float* __restrict__ colscale = nullptr) {  // KT: entries per table row (>= K)
    constexpr int NS = 16;
    __shared__ T red[NS][64];
    const int W = L.W;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + tx;
    const bool live = e < L.n * K;
    T sum = T(0);
    int64_t i = 0;
    int c = 0;
    if (live) {
        i = e / K;
        c = (int)(e - i * K);
        const int64_t blk = i / W;
        const int64_t T_loc = L.strips_local();
        const int64_t t0 = blk <= L.first ? 0 : (blk - L.first + L.stride - 1) / L.stride;
        const int64_t nrow = T_loc > t0 ? T_loc - t0 : 0;
        int64_t tc = -1, items = 0;
        if (blk >= L.first && (blk - L.first) % L.stride == 0) {
            tc = (blk - L.first) / L.stride;
            items = (L.rows(tc) + ITEM_ROWS - 1) / ITEM_ROWS;
        }
        const int64_t jloc = i - blk * W;
        // four independent chains (the loads are latency-bound), combined in a fixed order
        auto term = [&](int64_t u) -> T {
            return u < nrow ? Prow[(L.row_offset(t0 + u) + i) * KT + c]
                            : Pcol[((tc * max_items + (u - nrow)) * W + jloc) * KT + c];
        };
        T s4[4] = {T(0), T(0), T(0), T(0)};
        int64_t u = ty;
        for (; u + 3 * NS < nrow + items; u += 4 * NS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] += term(u + q * NS);
        }
        for (int q = 0; u < nrow + items; u += NS, ++q) s4[q] += term(u);
        sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    red[ty][tx] = sum;
    __syncthreads();
    if (ty == 0 && live) {
        sum = T(0);
#pragma unroll
        for (int y = 0; y < NS; ++y) sum += red[y][tx];
        T y = alpha * sum;
        if (colscale != nullptr) y *= (T)colscale[c];  // (tables of the float16-split kernel hold scaled sums)
        if (shift != T(0)) y += shift * V[i * ldv + c];
        Y[i * ldy + c] = y;
    }
}

// ------------------------------------------------------------------------------------------
// the strip product for up to 16 right-hand sides on the 16-BIT matrix cores (float32 storage and vectors)
// ------------------------------------------------------------------------------------------
// symm_strip_mfma_kernel above is bound by the float32 matrix cores: 64 v_mfma_f32_16x16x4_f32 (32 cycles
// each) per wave and 32 KiB tile = 2 048 cycles, 54 % busy at 4.3 - 4.6 TB/s.  Here every stored entry and
// every entry of V is split into two float16 pieces (x 2^e = h + m + O(2^-22 |x|), see rk_symgemm.hip) and
// both uses of a tile run as three exact v_mfma_f32_16x16x32_f16 products (h.h + h.m + m.h): 24 MFMAs of
// 16 cycles per wave and tile = 384 cycles.
//   * staging: a wave loads whole stored rows (lane <-> 4 consecutive columns), scales, splits and writes
//     8 bytes per piece into two row-major float16 images [32 rows][256 columns];
//   * row part  Yrow[32 x 16] = T[32 x 256] V[256 x 16]: contraction along a stored row - the A operand
//     (row = lane % 16, 8 consecutive columns) is one ds_read_b128 of the row-major image; B = the strip's
//     V columns, pre-split rhs-major (16 bytes = 8 consecutive j of one right-hand side), in registers for
//     the whole item;
//   * column part  Ycol[256 x 16] += T^T[256 x 32] V_i[32 x 16]: contraction DOWN the tile's 32 rows - the
//     A operand (column = lane % 16, 8 consecutive rows) comes out of the SAME image through the transposing
//     LDS read ds_read_b64_tr_b16 (two per operand: lane 4q + p of a 16-lane group addresses row q, columns
//     4p .. 4p + 3 and lane i receives column i of the four rows); B = V of the tile's rows, a 16-byte global
//     load per piece and tile.  Row stride 544 bytes + an XOR of the 32-byte column blocks (h2_off): both read
//     patterns and the 8-byte writes are bank-conflict free.
// Accumulation: the row part's chains are 12 MFMAs long (then float32 adds); the column part's accumulators
// are folded into a second set every 8 tiles (the matrix cores' float32 accumulation truncates, ~1e-8 per
// step of a chain).  Tables, work items and the finishing kernel are those of the kernels above; the
// per-column unscaling rides in the finishing kernel.
// largest |V[:, c]| per column (float bits; non-negative floats order like integers)
__global__ void __launch_bounds__(256)
h2_absmax_kernel(const float* __restrict__ V, int64_t n, int k, int64_t ldv, unsigned* __restrict__ out) {
    __shared__ float red[16][16];  // one atomic per column and workgroup
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    float m = 0.f;
    if (c < k) {
        for (int64_t r = (int64_t)blockIdx.x * 16 + g; r < n; r += (int64_t)gridDim.x * 16) {
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
        if (t > 0.f) atomicMax(out + threadIdx.x, __float_as_uint(t));
    }
}
// info[16 + c] <- 2^eb(c) (V is scaled by it), info[32 + c] <- 2^-(ea + eb(c)) (the unscaling)
__global__ void h2_scale_kernel(unsigned* __restrict__ info, int ea) {
    const int c = threadIdx.x;
    if (c >= SMK) return;
    int eb = 0;
    const float m = __uint_as_float(info[c]);
    if (m > 0.f && m < 3.0e38f) {
        int ex;
        frexpf(m, &ex);
        eb = 14 - ex;
        eb = eb > 100 ? 100 : (eb < -100 ? -100 : eb);
    }
    info[16 + c] = __float_as_uint(ldexpf(1.f, eb));
    info[32 + c] = __float_as_uint(ldexpf(1.f, -eb) * ldexpf(1.f, -ea));
}
// Vt[piece][rhs (16)][j (npad)] float16: one thread per (rhs, 8-row chunk); rows >= n and rhs >= k are zero
__global__ void __launch_bounds__(256)
h2_split_v_kernel(const float* __restrict__ V, int64_t n, int64_t npad, int k, int64_t ldv,
                  const unsigned* __restrict__ info, unsigned short* __restrict__ Vt) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(g & 15);
    const int64_t chunk = g >> 4;
    if (chunk >= npad / 8) return;
    const float sb = __uint_as_float(info[16 + c]);
    unsigned h[4], m[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t j = chunk * 8 + 2 * e;
        const float a = (j < n && c < k) ? V[j * ldv + c] * sb : 0.f;
        const float b = (j + 1 < n && c < k) ? V[(j + 1) * ldv + c] * sb : 0.f;
        h2_split2(a, b, h[e], m[e]);
    }
    h2_u4 oh, om;
    oh.x = h[0]; oh.y = h[1]; oh.z = h[2]; oh.w = h[3];
    om.x = m[0]; om.y = m[1]; om.z = m[2]; om.w = m[3];
    *reinterpret_cast<h2_u4*>(Vt + ((int64_t)c * npad + chunk * 8)) = oh;
    *reinterpret_cast<h2_u4*>(Vt + ((int64_t)(SMK + c) * npad + chunk * 8)) = om;
}

// (A wave-private variant - each wave walks its own 16-row slabs, row-part operands straight from the load
// registers, the column part through one ds_read_b64_tr_b16 per block and v_mfma_f32_16x16x16_f16, two barriers
// per 1 MiB item instead of one per tile - was built and measured: 18.6 - 19.3 ms against 16.6 - 17.8 ms for the
// tile-staged kernel below; its loads touch 16 rows x 64 bytes per instruction and stream at 4.3 TB/s.  What
// bounds the tile-staged kernel is the traffic of the partial tables, not the matrix cores: see KTAB.)
//
// KTAB: the (strip, row) and (item, column) tables hold KTAB = k entries per row / column, not 16: at k = 5 that is
// 3.9 GB of table traffic (written here, read by the finishing kernel) instead of 12.6 GB beside the 80 GB triangle.
__global__ void __launch_bounds__(SYMM_THREADS, 2)
symm_strip_h2_kernel(const float* __restrict__ Kp, SymLayout L, const unsigned short* __restrict__ Vt, int64_t npad,
                     float sa, int ktab, float* __restrict__ Prow, float* __restrict__ Pcol, int64_t max_items,
                     int64_t total_items) {
    constexpr int W = 256;
    constexpr int TR = H2_TR;
    constexpr int RPW = TR / 4;  // staged rows per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    unsigned char* TB = sm_raw;                                               // [2 buffers][2 pieces][TR][544 B]
    float* RP = reinterpret_cast<float*>(sm_raw + 2 * 2 * H2_PIECE);          // [2][2 halves][TR][SMK]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int rb = wave & 1, cs = wave >> 1;  // row part: 16 rows x 128 columns per wave
    const unsigned short* Vh = Vt;                          // piece 0: [16][npad]
    const unsigned short* Vm = Vt + (int64_t)SMK * npad;    // piece 1
    // transposing read: lane 4q + p of a 16-lane group addresses (row q, columns 4p .. 4p + 3) of its block
    const int trq = (lane >> 2) & 3, trp = lane & 3;
    // (the XOR of h2_off moves whole groups of four blocks: block 4 wave + cb of rows 8 lq + q / + 4 sits at
    // h2_off(row, 4 wave) + 32 cb)
    const int tr_lo = h2_off(8 * lq + trq, 4 * wave) + 8 * trp;
    const int tr_hi = h2_off(8 * lq + trq + 4, 4 * wave) + 8 * trp;
    int aoff[4];  // row part: row 16 rb + lr, columns 128 cs + 32 s + 8 lq .. + 8
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = h2_off(16 * rb + lr, 8 * cs + 2 * s + (lq >> 1)) + (lq & 1) * 16;
    int64_t t = 0, before = 0;
    for (int64_t id = blockIdx.x; id < total_items; id += gridDim.x) {
        for (;;) {
            const int64_t cnt = (L.rows(t) + ITEM_ROWS - 1) / ITEM_ROWS;
            if (id < before + cnt) break;
            before += cnt;
            ++t;
        }
        const int64_t chunk = id - before;
        const int64_t nrows = uniform64(L.rows(t));
        const int64_t item_lo = uniform64(chunk * ITEM_ROWS);
        const int64_t item_hi = item_lo + ITEM_ROWS < nrows ? item_lo + ITEM_ROWS : nrows;
        const int64_t diag_lo = uniform64(L.strip(t) * W);
        const float* base = Kp + uniform64(L.row_offset(t) * W);
        const int ntile = (int)((item_hi - item_lo + TR - 1) / TR);

        // B operands of the row part: V[j = diag_lo + 128 cs + 32 s + 8 lq .. + 8][rhs = lr], both pieces
        h2_u4 vB[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t j = diag_lo + 128 * cs + 32 * s + 8 * lq;  // < npad
            vB[s][0] = *reinterpret_cast<const h2_u4*>(Vh + (int64_t)lr * npad + j);
            vB[s][1] = *reinterpret_cast<const h2_u4*>(Vm + (int64_t)lr * npad + j);
        }
        h2_acc ccol[4], ctot[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int v = 0; v < 4; ++v) { ccol[cb][v] = 0.f; ctot[cb][v] = 0.f; }

        float4 preA[RPW], preB[RPW];
        h2_u2 viA, viB;
        // V of the tile's rows (B operand of the column part: 2 pieces x 16 rhs x 32 rows = 2 KiB per tile): ONE 8-byte
        // load per thread (piece tid >> 7, rhs (tid >> 3) & 15, rows 4 (tid & 7) .. + 4), parked in the 32 padding bytes
        // of the image rows (h2_off uses bytes 0 .. 511 of a 544-byte row) - each wave loading all of it from global
        // was 8 KiB of L2 / Infinity-Cache reads per 32 KiB tile (PMC: fetches 1.11 x the packed bytes)
        const unsigned short* vsrc = Vt + ((int64_t)(tid >> 7) * SMK + ((tid >> 3) & 15)) * npad + 4 * (tid & 7);
        const int vch = 4 * ((tid >> 3) & 15) + ((tid & 7) >> 1);  // 16-byte chunk (rhs, 8-row group) this thread half-fills
        const int vi_st = (tid >> 7) * H2_PIECE + (vch >> 1) * H2_ROWB + 512 + (vch & 1) * 16 + (tid & 1) * 8;
        const int vi_ld = ((4 * lr + lq) >> 1) * H2_ROWB + 512 + ((4 * lr + lq) & 1) * 16;  // chunk (rhs lr, rows 8 lq .. + 8)
        auto load_tile = [&](int tl, float4 (&pre)[RPW], h2_u2& vi) {
            const int64_t r0 = item_lo + (int64_t)(tl < ntile ? tl : ntile - 1) * TR;
#pragma unroll
            for (int u = 0; u < RPW; ++u) {
                const int64_t i = r0 + wave + 4 * u;
                pre[u] = load_stream_at<float4>(base + (i < nrows ? i : nrows - 1) * W, (unsigned)lane * 16u);
            }
            vi = *reinterpret_cast<const h2_u2*>(vsrc + r0);  // (r0 + 32 <= npad)
        };
        auto store_tile = [&](int buf, int tl, const float4 (&pre)[RPW], const h2_u2& vi) {
            unsigned char* tb = TB + buf * 2 * H2_PIECE;
            const int64_t r0 = item_lo + (int64_t)(tl < ntile ? tl : ntile - 1) * TR;
#pragma unroll
            for (int u = 0; u < RPW; ++u) {
                const float m = r0 + wave + 4 * u < item_hi ? sa : 0.f;  // rows past the item: exact zeros
                unsigned h0, h1, m0, m1;
                h2_split2(pre[u].x * m, pre[u].y * m, h0, m0);
                h2_split2(pre[u].z * m, pre[u].w * m, h1, m1);
                h2_u2 h, mm;
                h.x = h0; h.y = h1; mm.x = m0; mm.y = m1;
                unsigned char* dst = tb + h2_off(wave + 4 * u, lane >> 2) + (lane & 3) * 8;
                *reinterpret_cast<h2_u2*>(dst) = h;
                *reinterpret_cast<h2_u2*>(dst + H2_PIECE) = mm;
            }
            *reinterpret_cast<h2_u2*>(tb + vi_st) = vi;
        };
        auto flush_rows = [&](int buf, int tl) {  // add the two column halves, store to the row table
            const float* rp = RP + buf * 2 * TR * SMK;
            float* pr = Prow + (L.row_offset(t) + item_lo + (int64_t)tl * TR) * ktab;
            const int64_t live = item_hi - item_lo - (int64_t)tl * TR;  // rows
#pragma unroll
            for (int e = tid; e < TR * SMK; e += SYMM_THREADS) {
                const int r = e >> 4, c = e & 15;
                const float sum = rp[e] + rp[TR * SMK + e];
#if defined(RK_H2_PROBE_NOTABLE)
                if (r < live && c < ktab && sum == 123.456f) __builtin_nontemporal_store(sum, &pr[r * ktab + c]);
#else
                if (r < live && c < ktab) __builtin_nontemporal_store(sum, &pr[r * ktab + c]);
#endif
            }
        };
        auto compute = [&](int buf, int tl) {
            if (tl > 0) flush_rows(buf ^ 1, tl - 1);  // (RP[buf ^ 1] was completed before the last barrier)
            const unsigned char* tb = TB + buf * 2 * H2_PIECE;
#if defined(RK_H2_PROBE_NOMFMA)
            if (sa != 0.f) return;
#endif
            // ---- row part: A = rows 16 rb + lr, columns 128 cs + 32 s + 8 lq .. + 8 -----------------------
            h2_acc crow;
#pragma unroll
            for (int v = 0; v < 4; ++v) crow[v] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                h2_u4 a[2];
                a[0] = *reinterpret_cast<const h2_u4*>(tb + aoff[s]);
                a[1] = *reinterpret_cast<const h2_u4*>(tb + aoff[s] + H2_PIECE);
                crow = h2_mfma3(a, vB[s], crow);
            }
            float* rp = RP + (buf * 2 + cs) * TR * SMK;
#pragma unroll
            for (int v = 0; v < 4; ++v) rp[(16 * rb + 4 * lq + v) * SMK + lr] = crow[v];
            // ---- column part (rows above the diagonal block only; tile-uniform): A = T^T through the
            // transposing read: columns 64 wave + 16 cb + (0 .. 15), rows 8 lq .. 8 lq + 7 ---------------------
            if (item_lo + (int64_t)tl * TR < diag_lo) {
                h2_u4 vi[2];
                vi[0] = *reinterpret_cast<const h2_u4*>(tb + vi_ld);
                vi[1] = *reinterpret_cast<const h2_u4*>(tb + vi_ld + H2_PIECE);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    h2_u4 a[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        typedef __attribute__((address_space(3))) h2_s4* lds_s4;
                        const unsigned char* src = tb + cb * 32 + p * H2_PIECE;
                        const h2_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(src + tr_lo));
                        const h2_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(src + tr_hi));
                        const h2_u2 l2 = __builtin_bit_cast(h2_u2, lo), h2v = __builtin_bit_cast(h2_u2, hi);
                        a[p].x = l2.x; a[p].y = l2.y; a[p].z = h2v.x; a[p].w = h2v.y;
                    }
                    ccol[cb] = h2_mfma3(a, vi, ccol[cb]);
                }
                if ((tl & (H2_FOLD - 1)) == H2_FOLD - 1) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        ctot[cb] += ccol[cb];
#pragma unroll
                        for (int v = 0; v < 4; ++v) ccol[cb][v] = 0.f;
                    }
                }
            }
        };

        load_tile(0, preA, viA);
        store_tile(0, 0, preA, viA);
        load_tile(1, preA, viA);
        load_tile(2, preB, viB);
        __syncthreads();
        for (int tl = 0; tl < ntile; tl += 2) {
            store_tile(1, tl + 1, preA, viA);
            load_tile(tl + 3, preA, viA);
            compute(0, tl);
            __syncthreads();  // tile tl + 1 staged, RP[0] complete, everyone done with TB[0]
            if (tl + 1 < ntile) {
                store_tile(0, tl + 2, preB, viB);
                load_tile(tl + 4, preB, viB);
                compute(1, tl + 1);
                __syncthreads();
            }
        }
        flush_rows((ntile - 1) & 1, ntile - 1);
        // column sums of the item: element v of ccol[cb] = Ycol[column 64 wave + 16 cb + 4 lq + v][rhs = lr]
        float* pc = Pcol + ((t * max_items + chunk) * W) * ktab;
        if (lr < ktab) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    __builtin_nontemporal_store(ctot[cb][v] + ccol[cb][v],
                                                &pc[(64 * wave + 16 * cb + 4 * lq + v) * ktab + lr]);
        }
        __syncthreads();  // RP / TB are reused by the next item
    }
}

template <typename T, typename TS>
static int sympack_materialize_impl(int kind, const T* X, int64_t n, int64_t ldx, int64_t d,
                                    double ell, const T* ell_vec, double scale, int64_t first,
                                    int64_t stride, TS* Kp, hipStream_t st) {
    const SymLayout L = make_layout<TS>(n, first, stride);
    const int64_t T_loc = L.strips_local();
    if (T_loc == 0) return RK_OK;
    constexpr int CT = strip_width<TS>() / TN;
    const int64_t max_rows = L.rows(T_loc - 1);
    dim3 grid((unsigned)(ceil_div(max_rows, TM) * CT), (unsigned)T_loc);
    const T w = ell_vec ? T(1) : T(1.0 / ell);
#define RK_SMAT(L1V, DCVV)                                                                        \
    hipLaunchKernelGGL((sympack_materialize_kernel<T, TS, L1V, DCVV>), grid, dim3(256), 0, st, kind, X, \
                       ldx, d, w, ell_vec, (T)scale, L, Kp)
    if (kind == RK_LAPLACE) RK_SMAT(true, 32);
    else RK_SMAT(false, 32);
#undef RK_SMAT
    return launch_status("rk_kernel_materialize_sympack");
}

template <typename TS, typename T>
static size_t symm_ws_bytes(int64_t n, int64_t k, int64_t first, int64_t stride) {
    const SymLayout L = make_layout<TS>(n, first, stride);
    const int64_t max_items = ceil_div(n, ITEM_ROWS);
    const size_t prow = (size_t)round_up(L.total_rows() * k, 16);  // keeps Pcol 16-byte aligned
    const size_t pcol = (size_t)L.strips_local() * max_items * L.W * k;
    return (prow + pcol) * sizeof(T);
}

// resident workgroups of the persistent strip kernel on the CURRENT device (cached per
// instantiation and device ordinal; the slots are written with the same value by any thread)
template <typename KernelT>
static int64_t symm_resident_workgroups(KernelT kernel, int wg_per_cu) {
    static std::atomic<int64_t> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::atomic<int64_t>& slot = cached[dev & 63];
    int64_t v = slot.load(std::memory_order_acquire);
    if (v == 0) {
        int per_cu = wg_per_cu, cus = 0;
        if (per_cu <= 0 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, SYMM_THREADS, 0) != hipSuccess)
            per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        (void)hipGetLastError();
        v = (int64_t)(per_cu > 0 ? per_cu : 3) * (cus > 0 ? cus : 256);
        slot.store(v, std::memory_order_release);
    }
    return v;
}

template <typename TS, typename T>
static int symm_matmat_impl(const TS* Kp, int64_t n, int64_t first, int64_t stride, const T* V,
                            int64_t k, int64_t ldv, double alpha, double shift, T* Y, int64_t ldy,
                            void* ws, size_t ws_bytes, hipStream_t st) {
    const SymLayout L = make_layout<TS>(n, first, stride);
    const int64_t T_loc = L.strips_local();
    const int64_t max_items = ceil_div(n, ITEM_ROWS);
    const size_t need = symm_ws_bytes<TS, T>(n, k, first, stride);
    if (need > 0 && (ws == nullptr || ws_bytes < need)) {
        set_error("rk_symm_matmat: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    T* Prow = static_cast<T*>(ws);
    T* Pcol = Prow + (size_t)round_up(L.total_rows() * k, 16);
    if (T_loc > 0) {
        int64_t total_items = 0;
        for (int64_t t = 0; t < T_loc; ++t) total_items += ceil_div(L.rows(t), ITEM_ROWS);
#define RK_SYMM(KK)                                                                             \
    {                                                                                           \
        /* measured (tools/ab.py): float64 storage, k = 1 prefers 5 workgroups per CU (+3 %) */   \
        const int wg = (RK_SYMM_WG_PER_CU == 6 && sizeof(TS) == 8 && KK == 1) ? 5 : RK_SYMM_WG_PER_CU; \
        const int64_t resident = symm_resident_workgroups(symm_strip_kernel<TS, T, KK>, wg);    \
        dim3 grid((unsigned)(total_items < resident ? total_items : resident));                 \
        hipLaunchKernelGGL((symm_strip_kernel<TS, T, KK>), grid, dim3(SYMM_THREADS), 0, st, Kp, L, V, \
                           ldv, Prow, Pcol, max_items, total_items);                            \
    }
        switch (k) {
            case 1: RK_SYMM(1); break;
            case 2: RK_SYMM(2); break;
            case 3: RK_SYMM(3); break;
            default: RK_SYMM(4); break;
        }
#undef RK_SYMM
        int rc = launch_status("rk_symm_matmat");
        if (rc != RK_OK) return rc;
    }
    hipLaunchKernelGGL((symm_finish_kernel<T>), dim3((unsigned)ceil_div(n * k, 64)), dim3(1024), 0,
                       st, L, (int)k, (int)k, Prow, Pcol, max_items, (T)alpha, (T)shift, V, ldv, Y, ldy);
    return launch_status("rk_symm_matmat(finish)");
}

// ---- 5 <= k <= 16 right-hand sides: the matrix-core strip kernel -------------------------------------
template <typename T, int TR>
static int launch_symm_mfma(const T* Kp, const SymLayout& L, const T* Vp, T* Prow, T* Pcol,
                            int64_t max_items, int64_t total_items, hipStream_t st) {
    constexpr size_t smem = sm_smem_bytes<T, TR>();
    static std::atomic<unsigned long long> attr_devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (smem > 65536 && !(attr_devs.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(symm_strip_mfma_kernel<T, TR>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
            (void)hipGetLastError();
            set_error("rk_symm_matmat: cannot reserve %zu bytes of LDS", smem);
            return RK_ERR_LAUNCH;
        }
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    const int per_cu = TR == 32 ? 2 : (sizeof(T) == 4 ? 4 : 3);
    const int64_t resident = symm_resident_workgroups(symm_strip_mfma_kernel<T, TR>, per_cu);
    dim3 grid((unsigned)(total_items < resident ? total_items : resident));
    hipLaunchKernelGGL((symm_strip_mfma_kernel<T, TR>), grid, dim3(SYMM_THREADS), smem, st, Kp, L, Vp, Prow, Pcol,
                       max_items, total_items);
    return RK_OK;
}

template <typename T>
static size_t symm_wide_ws_bytes(int64_t n, int64_t first, int64_t stride) {
    const SymLayout L = make_layout<T>(n, first, stride);
    const int64_t max_items = ceil_div(n, ITEM_ROWS);
    const size_t vp = (size_t)round_up(n, 256) * SMK;
    const size_t prow = (size_t)L.total_rows() * SMK;
    const size_t pcol = (size_t)L.strips_local() * max_items * L.W * SMK;
    return (vp + prow + pcol) * sizeof(T);
}

template <typename T>
static int symm_matmat_wide_impl(const T* Kp, int64_t n, int64_t first, int64_t stride, const T* V,
                                 int64_t k, int64_t ldv, double alpha, double shift, T* Y, int64_t ldy,
                                 void* ws, size_t ws_bytes, hipStream_t st) {
    const SymLayout L = make_layout<T>(n, first, stride);
    const int64_t T_loc = L.strips_local();
    const int64_t max_items = ceil_div(n, ITEM_ROWS);
    const size_t need = symm_wide_ws_bytes<T>(n, first, stride);
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_symm_matmat: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    const int64_t npad = round_up(n, 256);
    T* Vp = static_cast<T*>(ws);
    T* Prow = Vp + (size_t)npad * SMK;
    T* Pcol = Prow + (size_t)L.total_rows() * SMK;
    hipLaunchKernelGGL(symm_pad_v_kernel<T>, dim3((unsigned)ceil_div(npad * SMK, 256)), dim3(256), 0, st, V, n,
                       npad, (int)k, ldv, Vp);
    if (T_loc > 0) {
        int64_t total_items = 0;
        for (int64_t t = 0; t < T_loc; ++t) total_items += ceil_div(L.rows(t), ITEM_ROWS);
        int rcl;
        if constexpr (sizeof(T) == 4) {
            // rows per tile: 32 (default; 17.9 ms at n = 200 000, k = 10) or RK_SYMM_TR=16 (4 workgroups per
            // CU instead of 2: 18.7 ms - the barrier per 16 KiB costs more than the occupancy gains)
            const char* e = getenv("RK_SYMM_TR");
            const bool tr16 = e != nullptr && e[0] == '1';
            rcl = tr16 ? launch_symm_mfma<T, 16>(Kp, L, Vp, Prow, Pcol, max_items, total_items, st)
                       : launch_symm_mfma<T, 32>(Kp, L, Vp, Prow, Pcol, max_items, total_items, st);
        } else {
            rcl = launch_symm_mfma<T, 16>(Kp, L, Vp, Prow, Pcol, max_items, total_items, st);
        }
        if (rcl != RK_OK) return rcl;
        int rc = launch_status("rk_symm_matmat(wide)");
        if (rc != RK_OK) return rc;
    }
    hipLaunchKernelGGL((symm_finish_kernel<T>), dim3((unsigned)ceil_div(n * k, 64)), dim3(1024), 0, st, L,
                       (int)k, SMK, Prow, Pcol, max_items, (T)alpha, (T)shift, V, ldv, Y, ldy);
    return launch_status("rk_symm_matmat(finish)");
}

// ---- up to 16 right-hand sides on the 16-bit matrix cores (float32) -------------------------------------------
static size_t symm_h2_ws_bytes(int64_t n, int64_t first, int64_t stride) {
    const SymLayout L = make_layout<float>(n, first, stride);
    const int64_t max_items = ceil_div(n, ITEM_ROWS);
    const size_t npad = (size_t)round_up(n, 256);
    const size_t vt = 2 * SMK * npad * sizeof(unsigned short);            // both pieces, rhs-major
    const size_t prow = (size_t)L.total_rows() * SMK * sizeof(float);
    const size_t pcol = (size_t)L.strips_local() * max_items * L.W * SMK * sizeof(float);
    return 256 + vt + prow + pcol;
}

int h2_prepare_v(const float* V, int64_t n, int64_t npad, int64_t k, int64_t ldv, double kmax, unsigned* info,
                 unsigned short* Vt, int* ea_out, hipStream_t st) {
    int ex;
    frexp(kmax, &ex);
    const int ea = 14 - ex;  // stored entries <= kmax < 2^ex: scaled below 2^14
    *ea_out = ea;
    if (hipMemsetAsync(info, 0, 256, st) != hipSuccess) return launch_status("rk_symm_matmat_f16x2(memset)");
    hipLaunchKernelGGL(h2_absmax_kernel, dim3(256), dim3(256), 0, st, V, n, (int)k, ldv, info);
    hipLaunchKernelGGL(h2_scale_kernel, dim3(1), dim3(64), 0, st, info, ea);
    hipLaunchKernelGGL(h2_split_v_kernel, dim3((unsigned)ceil_div((npad / 8) * SMK, 256)), dim3(256), 0, st, V, n, npad,
                       (int)k, ldv, info, Vt);
    return launch_status("rk_symm_matmat_f16x2(split)");
}

static int symm_matmat_h2_impl(const float* Kp, int64_t n, int64_t first, int64_t stride, double kmax, const float* V,
                               int64_t k, int64_t ldv, double alpha, double shift, float* Y, int64_t ldy, void* ws,
                               size_t ws_bytes, hipStream_t st) {
    const SymLayout L = make_layout<float>(n, first, stride);
    const int64_t T_loc = L.strips_local();
    const int64_t max_items = ceil_div(n, ITEM_ROWS);
    const size_t need = symm_h2_ws_bytes(n, first, stride);
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_symm_matmat_f16x2: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    const int64_t npad = round_up(n, 256);
    unsigned* info = static_cast<unsigned*>(ws);
    unsigned short* Vt = reinterpret_cast<unsigned short*>(static_cast<char*>(ws) + 256);
    float* Prow = reinterpret_cast<float*>(Vt + 2 * SMK * npad);
    float* Pcol = Prow + (size_t)round_up(L.total_rows() * k, 4);
    int ea = 0;
    int rc = h2_prepare_v(V, n, npad, k, ldv, kmax, info, Vt, &ea, st);
    if (rc != RK_OK) return rc;
    if (T_loc > 0) {
        int64_t total_items = 0;
        for (int64_t t = 0; t < T_loc; ++t) total_items += ceil_div(L.rows(t), ITEM_ROWS);
        constexpr size_t smem = h2_smem_bytes();
        static std::atomic<unsigned long long> attr_devs{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(symm_strip_h2_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
                (void)hipGetLastError();
                set_error("rk_symm_matmat_f16x2: cannot reserve %zu bytes of LDS", smem);
                return RK_ERR_LAUNCH;
            }
            attr_devs.fetch_or(bit, std::memory_order_release);
        }
        const int64_t resident = symm_resident_workgroups(symm_strip_h2_kernel, 2);
        dim3 grid((unsigned)(total_items < resident ? total_items : resident));
        hipLaunchKernelGGL(symm_strip_h2_kernel, grid, dim3(SYMM_THREADS), smem, st, Kp, L, Vt, npad,
                           (float)ldexp(1.0, ea), (int)k, Prow, Pcol, max_items, total_items);
        rc = launch_status("rk_symm_matmat_f16x2");
        if (rc != RK_OK) return rc;
    }
    hipLaunchKernelGGL((symm_finish_kernel<float>), dim3((unsigned)ceil_div(n * k, 64)), dim3(1024), 0, st, L, (int)k,
                       (int)k, Prow, Pcol, max_items, (float)alpha, (float)shift, V, ldv, Y, ldy,
                       reinterpret_cast<const float*>(info + 32));
    return launch_status("rk_symm_matmat_f16x2(finish)");
}

}  // namespace rk

using namespace rk;

// the matrix-core strip kernel takes 5 .. 16 right-hand sides, and float64 from 4 on (the float64 register
// kernel streams 3.7 TB/s at k = 4: 21.3 ms against 19.4 ms at n = 140 000)
static bool symm_use_wide(int dtype, int store_dtype, int64_t k) {
    if (store_dtype != dtype) return false;
    if (k > 4) return true;
    if (k < 4 || dtype != RK_F64) return false;
    const char* e = getenv("RK_SYMM_WIDE");  // (0: the register kernel also at float64 k = 4)
    return !(e != nullptr && e[0] == '0');
}

static bool symm_use_blkreg() {
    const char* e = getenv("RK_SYMM_BLKREG");
    return !(e != nullptr && e[0] == '0');
}

static int check_layout(const char* fn, int dtype, int store_dtype, int64_t n, int64_t first,
                        int64_t stride) {
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "%s: bad dtype %d", fn, dtype);
    RK_REQUIRE(store_dtype == dtype || (dtype == RK_F64 && store_dtype == RK_F32),
               "%s: storage dtype must equal dtype, or be float32 under float64 arithmetic", fn);
    RK_REQUIRE(n > 0, "%s: n must be positive", fn);
    RK_REQUIRE(stride >= 1 && first >= 0 && first < stride, "%s: need 0 <= strip_first < strip_stride", fn);
    return RK_OK;
}

extern "C" size_t rk_sympack_elems(int store_dtype, int64_t n, int64_t strip_first,
                                   int64_t strip_stride) {
    if (n <= 0 || strip_stride < 1 || strip_first < 0 || strip_first >= strip_stride) return 0;
    const SymLayout L = store_dtype == RK_F32 ? make_layout<float>(n, strip_first, strip_stride)
                                              : make_layout<double>(n, strip_first, strip_stride);
    return (size_t)L.total_rows() * L.W;
}

extern "C" int rk_kernel_materialize_sympack(int dtype, int store_dtype, int kind, const void* X,
                                             int64_t n, int64_t ldx, int64_t d, double ell_scalar,
                                             const void* ell_vec, double scale, int64_t strip_first,
                                             int64_t strip_stride, void* Kp, void* stream) {
    clear_error();
    int rc = check_layout("rk_kernel_materialize_sympack", dtype, store_dtype, n, strip_first,
                          strip_stride);
    if (rc != RK_OK) return rc;
    RK_REQUIRE(kind >= RK_RBF && kind <= RK_MATERN52, "rk_kernel_materialize_sympack: bad kind %d", kind);
    RK_REQUIRE(X && Kp, "rk_kernel_materialize_sympack: null pointer");
    RK_REQUIRE(d > 0 && ldx >= d, "rk_kernel_materialize_sympack: bad feature dimension");
    RK_REQUIRE(ell_vec != nullptr || ell_scalar > 0.0, "rk_kernel_materialize_sympack: lengthscale must be positive");
    RK_REQUIRE(aligned16(Kp), "rk_kernel_materialize_sympack: output must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        return sympack_materialize_impl<float, float>(kind, (const float*)X, n, ldx, d, ell_scalar,
                                                      (const float*)ell_vec, scale, strip_first,
                                                      strip_stride, (float*)Kp, st);
    if (store_dtype == RK_F32)
        return sympack_materialize_impl<double, float>(kind, (const double*)X, n, ldx, d, ell_scalar,
                                                       (const double*)ell_vec, scale, strip_first,
                                                       strip_stride, (float*)Kp, st);
    return sympack_materialize_impl<double, double>(kind, (const double*)X, n, ldx, d, ell_scalar,
                                                    (const double*)ell_vec, scale, strip_first,
                                                    strip_stride, (double*)Kp, st);
}

extern "C" size_t rk_symm_matmat_ws_bytes(int dtype, int store_dtype, int64_t n, int64_t k,
                                          int64_t strip_first, int64_t strip_stride) {
    if (n <= 0 || k < 1 || strip_stride < 1 || strip_first < 0 || strip_first >= strip_stride) return 0;
    if (symm_use_wide(dtype, store_dtype, k)) {
        if (k > SMK) return 0;
        return dtype == RK_F32 ? symm_wide_ws_bytes<float>(n, strip_first, strip_stride)
                               : symm_wide_ws_bytes<double>(n, strip_first, strip_stride);
    }
    if (k > 4) return 0;
    auto both = [](size_t a, size_t b) { return a > b ? a : b; };  // (either implementation can run in it)
    if (dtype == RK_F32)
        return both(symm_ws_bytes<float, float>(n, k, strip_first, strip_stride),
                    symm_blkreg_ws_bytes<float, float>(n, k, strip_first, strip_stride));
    return store_dtype == RK_F32 ? both(symm_ws_bytes<float, double>(n, k, strip_first, strip_stride),
                                        symm_blkreg_ws_bytes<float, double>(n, k, strip_first, strip_stride))
                                 : both(symm_ws_bytes<double, double>(n, k, strip_first, strip_stride),
                                        symm_blkreg_ws_bytes<double, double>(n, k, strip_first, strip_stride));
}

extern "C" int rk_symm_matmat(int dtype, int store_dtype, const void* Kp, int64_t n,
                              int64_t strip_first, int64_t strip_stride, const void* V, int64_t k,
                              int64_t ldv, double alpha, double diag_shift, void* Y, int64_t ldy,
                              void* ws, size_t ws_bytes, void* stream) {
    clear_error();
    int rc = check_layout("rk_symm_matmat", dtype, store_dtype, n, strip_first, strip_stride);
    if (rc != RK_OK) return rc;
    RK_REQUIRE(Kp && V && Y, "rk_symm_matmat: null pointer");
    const bool wide_ok = store_dtype == dtype;
    RK_REQUIRE(k >= 1 && k <= (wide_ok ? SMK : 4), "rk_symm_matmat: k=%lld outside [1,%d]", (long long)k,
               wide_ok ? SMK : 4);
    RK_REQUIRE(ldv >= k && ldy >= k, "rk_symm_matmat: leading dimension too small");
    RK_REQUIRE(aligned16(Kp), "rk_symm_matmat: packed matrix must be 16-byte aligned");
    RK_REQUIRE(aligned16(ws), "rk_symm_matmat: workspace must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (symm_use_wide(dtype, store_dtype, k)) {
        if (dtype == RK_F32)
            return symm_matmat_wide_impl<float>((const float*)Kp, n, strip_first, strip_stride, (const float*)V, k,
                                                ldv, alpha, diag_shift, (float*)Y, ldy, ws, ws_bytes, st);
        return symm_matmat_wide_impl<double>((const double*)Kp, n, strip_first, strip_stride, (const double*)V, k,
                                             ldv, alpha, diag_shift, (double*)Y, ldy, ws, ws_bytes, st);
    }
    // one or two columns of the storage type, one float64 column over float32 storage: the block decomposition on the
    // vector units (rk_symmblk.hip; RK_SYMM_BLKREG=0: the strip kernel).  Beyond that the per-lane column sums of four
    // strips do not fit the registers at a useful occupancy.
    if (symm_use_blkreg() && k <= ((dtype == RK_F32 || store_dtype == dtype) ? 2 : 1)) {
        if (dtype == RK_F32)
            return symm_matmat_blkreg_impl<float, float>((const float*)Kp, n, strip_first, strip_stride, (const float*)V,
                                                         k, ldv, alpha, diag_shift, (float*)Y, ldy, ws, ws_bytes, st);
        if (store_dtype == RK_F32)
            return symm_matmat_blkreg_impl<float, double>((const float*)Kp, n, strip_first, strip_stride, (const double*)V,
                                                          k, ldv, alpha, diag_shift, (double*)Y, ldy, ws, ws_bytes, st);
        return symm_matmat_blkreg_impl<double, double>((const double*)Kp, n, strip_first, strip_stride, (const double*)V, k,
                                                       ldv, alpha, diag_shift, (double*)Y, ldy, ws, ws_bytes, st);
    }
    if (dtype == RK_F32)
        return symm_matmat_impl<float, float>((const float*)Kp, n, strip_first, strip_stride,
                                              (const float*)V, k, ldv, alpha, diag_shift, (float*)Y,
                                              ldy, ws, ws_bytes, st);
    if (store_dtype == RK_F32)
        return symm_matmat_impl<float, double>((const float*)Kp, n, strip_first, strip_stride,
                                               (const double*)V, k, ldv, alpha, diag_shift, (double*)Y,
                                               ldy, ws, ws_bytes, st);
    return symm_matmat_impl<double, double>((const double*)Kp, n, strip_first, strip_stride,
                                            (const double*)V, k, ldv, alpha, diag_shift, (double*)Y, ldy,
                                            ws, ws_bytes, st);
}

extern "C" size_t rk_symm_matmat_f16x2_ws_bytes(int64_t n, int64_t k, int64_t strip_first, int64_t strip_stride) {
    if (n <= 0 || k < 1 || k > SMK || strip_stride < 1 || strip_first < 0 || strip_first >= strip_stride) return 0;
    const size_t a = symm_h2_ws_bytes(n, strip_first, strip_stride);
    const size_t b = symm_h2blk_ws_bytes(n, k, strip_first, strip_stride);
    return a > b ? a : b;  // (either implementation can run in it)
}

extern "C" int rk_symm_matmat_f16x2(const void* Kp, int64_t n, int64_t strip_first, int64_t strip_stride, double kmax,
                                    const void* V, int64_t k, int64_t ldv, double alpha, double diag_shift, void* Y,
                                    int64_t ldy, void* ws, size_t ws_bytes, void* stream) {
    clear_error();
    int rc = check_layout("rk_symm_matmat_f16x2", RK_F32, RK_F32, n, strip_first, strip_stride);
    if (rc != RK_OK) return rc;
    RK_REQUIRE(Kp && V && Y, "rk_symm_matmat_f16x2: null pointer");
    RK_REQUIRE(k >= 1 && k <= SMK, "rk_symm_matmat_f16x2: k=%lld outside [1,%d]", (long long)k, SMK);
    RK_REQUIRE(ldv >= k && ldy >= k, "rk_symm_matmat_f16x2: leading dimension too small");
    RK_REQUIRE(kmax > 0.0 && kmax < 1e30, "rk_symm_matmat_f16x2: kmax must bound the stored entries (got %g)", kmax);
    RK_REQUIRE(aligned16(Kp) && aligned16(ws), "rk_symm_matmat_f16x2: packed matrix and workspace must be 16-byte aligned");
    // default: the block form (rk_symmblk.hip: four strips side by side per work item, a quarter of the partial
    // tables); RK_SYMM_H2_IMPL=strip keeps the round-3 strip kernel (A/B partner)
    const char* impl = getenv("RK_SYMM_H2_IMPL");
    if (impl != nullptr && impl[0] == 's')
        return symm_matmat_h2_impl((const float*)Kp, n, strip_first, strip_stride, kmax, (const float*)V, k, ldv, alpha,
                                   diag_shift, (float*)Y, ldy, ws, ws_bytes, static_cast<hipStream_t>(stream));
    return symm_matmat_h2blk_impl((const float*)Kp, n, strip_first, strip_stride, kmax, (const float*)V, k, ldv, alpha,
                                  diag_shift, (float*)Y, ldy, ws, ws_bytes, static_cast<hipStream_t>(stream));
}
