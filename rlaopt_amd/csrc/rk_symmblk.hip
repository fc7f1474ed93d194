// The strip-packed symmetric product for 1 .. 16 right-hand sides, BLOCK form (round 4).
//
// What bounded the round-3 strip kernel (symm_strip_h2_kernel, rk_symm.hip), measured with build-time probes
// (tools/ab.py --kernel symmh2, n = 200 000, k = 16: 19.2 ms; without its table stores 16.4; without its matrix-core
// work 16.6; without both 14.7) and with tools/wr_probe.hip (a 7.1 TB/s read stream loses 9 % to 0.3 GB and 23 - 43 % to
// 2.5 GB of stores issued beside it - a written byte costs 4 - 7 read bytes, whatever the store width, policy or burst
// size):
//   1. the partial tables: k floats per stored ROW of every strip (4 k bytes per KiB read) + k per column and item;
//   2. every wave of a workgroup serialises "wait for the tile - split - barrier - LDS reads - 24 dependent matrix-core
//      instructions" with two tiles in flight, so the time from a tile's arrival to the re-issue of its registers'
//      next loads (Little's law: rate = bytes in flight / (latency + that time)) includes the arithmetic.
//
// Here a work item is a BLOCK of G = 4 strips side by side x H rows (H = 2048: 8 MiB; chosen per call, 2048 .. 256:
// make_blk_layout), walked row-tile by row-tile:
// the four 32 x 256 tiles of a row-tile are consumed back to back, so
//   * a row's partial sum is carried across the four strips in the accumulators and written ONCE per row-tile: the
//     (group, row) table has k floats per 4 KiB read - a quarter of the strip kernel's;
//   * the column sums of all four strips stay in registers for the whole item (8 accumulators of 4 registers per wave;
//     folded into a second set every 8 row-tiles like the strip kernel's): k floats per column and H rows.
//   Tables per element read: k (1/1024 + 1/2048) instead of k (1/256 + 1/1024): 3.3 x fewer bytes written here and
//   read by the finishing kernel.
// One workgroup of 8 waves per CU owns the CU's LDS: two float16 images of a tile (hi / lo pieces, the conflict-free
// layout of the strip kernel, double-buffered: one barrier per tile), the split V of the group's 4 x 256 columns in
// matrix-core operand order (64 KiB, loaded once per item: the strip kernel kept 32 registers of it per wave), the
// row-part exchange buffers and V of the tile's rows.  Per tile a wave stages 4 rows, then issues 3 + 3 matrix-core
// instructions for the row part (16 rows x 64 columns, two independent chains) and 3 + 3 for the column part (two
// 16-column blocks): half the chain length of the strip kernel.  S tiles (RK_B2_S, default 4 = 128 KiB per CU) are in
// flight in registers; the loads of tile t + S are issued right behind the split of tile t, BEFORE the barrier and
// the arithmetic of tile t.
// Strips of a group that do not reach a row-tile (the triangle's diagonal blocks; strips dealt cyclically over ranks
// differ by `stride` blocks in height) keep their slot in the tile sequence - the sequence stays regular, every
// register array statically indexed - but load one cached row and are skipped by the arithmetic.
#include <atomic>
#include <stdlib.h>

#include "rk_common.h"
#include "rk_symm.h"

namespace rk {

#ifndef RK_B2_G
#define RK_B2_G 4
#endif
#ifndef RK_B2_H
#define RK_B2_H 0  // rows per item: 0 = chosen per call (2048 .. 256, see make_blk_layout), else fixed (a multiple of 32)
#endif
#ifndef RK_B2_S
#define RK_B2_S 4
#endif
#ifndef RK_B2_FOLD
#define RK_B2_FOLD 8  // row-tiles between folds of the column accumulators (0: never)
#endif
constexpr int B2_G = RK_B2_G;        // strips per group
constexpr int B2_H_FIXED = RK_B2_H;
constexpr int B2_S = RK_B2_S;        // tiles in flight per workgroup
constexpr int B2_FOLD = RK_B2_FOLD;
constexpr int B2_THREADS = 512;
constexpr int B2_W = 256;
constexpr int B2_TR = H2_TR;         // rows per tile (32)
static_assert(B2_S % B2_G == 0 && B2_S % 2 == 0, "the unrolled tile loop keeps the strip index and the buffer parity static");
static_assert(B2_H_FIXED % B2_TR == 0 && B2_G == 4, "items are whole row-tiles; the LDS plan assumes 4 strips");

// LDS plan (bytes)
constexpr int B2_TB = 0;                                   // [2 buffers][2 pieces][32 rows][544 B]
constexpr int B2_VS = B2_TB + 2 * 2 * H2_PIECE;            // [G][2 pieces][4 column slices][2 steps][64 lanes][16 B]
constexpr int B2_RP = B2_VS + B2_G * 2 * 4 * 2 * 1024;     // [2][4 slices][32 rows][16 rhs] float
constexpr int B2_VI = B2_RP + 2 * 4 * B2_TR * SMK * 4;     // [2][2 pieces][64 lanes][16 B]
constexpr int B2_NEXT = B2_VI + 2 * 2 * 1024;              // next work item (dynamic scheduling)
constexpr int B2_SMEM = B2_NEXT + 16;
static_assert(B2_SMEM <= 160 * 1024, "one workgroup per CU owns the LDS");

struct BlkLayout {
    SymLayout L;
    int64_t T_loc, Q, max_chunks;
    int H;  // rows per item
    // rows of group q = rows of its last strip (strip heights grow with the strip index)
    __host__ __device__ int64_t grp_rows(int64_t q) const {
        int64_t t = q * B2_G + B2_G - 1;
        if (t > T_loc - 1) t = T_loc - 1;
        return L.rows(t);
    }
    // table rows before group q (every earlier group is complete and unclipped)
    __host__ __device__ int64_t prow_off(int64_t q) const {
        return (int64_t)L.W * (q * (L.first + 1 + (B2_G - 1) * L.stride) + B2_G * L.stride * (q * (q - 1) / 2));
    }
    __host__ __device__ int64_t prow_rows() const { return Q ? prow_off(Q - 1) + grp_rows(Q - 1) : 0; }
};

static int64_t blk_items(const BlkLayout& B) {
    int64_t total = 0;
    for (int64_t q = 0; q < B.Q; ++q) total += ceil_div(B.grp_rows(q), B.H);
    return total;
}

// Rows per item.  Taller items shrink the (item, column) table (k floats per column and item) and amortise the per-item
// fixed costs (the group's V into LDS, one pipeline fill, the table flush: about 40 rows' worth of streaming), but the
// persistent grid runs ceil(items / CUs) rounds of items and the last round is rarely full: H is the multiple of 32 in
// [256, 2048] that minimises rounds x (H + 40) among those that leave every CU at least 12 items (items are claimed
// dynamically, so rounds are approximate: the tail is at most one item long).  n = 200 000 on one GPU (256 CUs): see
// profiles/r04_ab_symm.log.  The choice is cached per (n, first, stride, CUs).
static int pick_item_rows(BlkLayout B, int64_t cus, int hmax) {
    struct Key { int64_t n, first, stride, cus; int h; };
    static thread_local Key cache[8] = {};
    for (const Key& c : cache)
        if (c.h && c.n == B.L.n && c.first == B.L.first && c.stride == B.L.stride && c.cus == cus + 100000 * (int64_t)hmax + 50000 * (B.L.W == 128)) return c.h;
    double best = 1e300;
    int best_h = 256;
    for (int h = hmax; h >= 256; h -= 32) {
        B.H = h;
        const int64_t items = blk_items(B);
        if (items < 12 * cus && h > 256) continue;
        const double cost = (double)ceil_div(items, cus) * (h + 40);
        if (cost < best) { best = cost; best_h = h; }
    }
    static thread_local int next = 0;
    cache[next] = Key{B.L.n, B.L.first, B.L.stride, cus + 100000 * (int64_t)hmax + 50000 * (B.L.W == 128), best_h};
    next = (next + 1) % 8;
    return best_h;
}

static BlkLayout make_blk_layout(int64_t n, int64_t first, int64_t stride, int64_t cus = 256, int hmax = 2048,
                                 bool f64_storage = false) {
    BlkLayout B;
    B.L = f64_storage ? make_layout<double>(n, first, stride) : make_layout<float>(n, first, stride);
    B.T_loc = B.L.strips_local();
    B.Q = ceil_div(B.T_loc, B2_G);
    B.H = B2_H_FIXED;
    B.max_chunks = 0;
    if (B2_H_FIXED == 0 || hmax != 2048) B.H = B.Q > 0 ? pick_item_rows(B, cus, hmax) : 256;
    B.max_chunks = ceil_div(n, B.H);
    return B;
}

__global__ void __launch_bounds__(B2_THREADS, 2)
symm_blk_h2_kernel(const float* __restrict__ Kp, BlkLayout B, const unsigned short* __restrict__ Vt, int64_t npad,
                   float sa, int ktab, float* __restrict__ Prow, float* __restrict__ Pcol, int64_t total_items,
                   unsigned* __restrict__ counter) {
    constexpr int G = B2_G, S = B2_S, TR = B2_TR, W = B2_W;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    const SymLayout& L = B.L;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int rb = wave & 1, cs = wave >> 1;  // row part: 16 rows x 64 columns per wave
    // transposing read of the column part (see symm_strip_h2_kernel): this wave's two 16-column blocks 2 wave, 2 wave + 1
    const int trq = (lane >> 2) & 3, trp = lane & 3;
    const int tr_lo = h2_off(8 * lq + trq, 2 * wave) + 8 * trp;
    const int tr_hi = h2_off(8 * lq + trq + 4, 2 * wave) + 8 * trp;
    int aoff[2];  // row part: row 16 rb + lr, columns 64 cs + 32 s + 8 lq .. + 8
#pragma unroll
    for (int s = 0; s < 2; ++s) aoff[s] = h2_off(16 * rb + lr, 4 * cs + 2 * s + (lq >> 1)) + (lq & 1) * 16;
    // V of a row-tile's rows: thread -> (piece, rhs) = tid >> 4, rows 2 (tid & 15), 2 (tid & 15) + 1
    const unsigned short* vsrc = Vt + (int64_t)(tid >> 4) * npad + 2 * (tid & 15);
    const int vi_st = (tid >> 8) * 1024 + ((((tid & 15) >> 2) * 16 + ((tid >> 4) & 15)) * 16) + ((tid & 15) & 3) * 4;

    // Work items are claimed from a device counter (zeroed by the caller) in ascending order - the resident grid sweeps
    // ONE contiguous window of the packed triangle (a contiguous range of row-tiles per workgroup - equal work, one
    // pipeline fill, column sums flushed once per group - was built and measured: 13.3 against 12.1 ms at k = 1, the 256
    // far-apart streams advance in lockstep and collide in the memory system).  The first item is blockIdx.x, every later
    // one gridDim.x + (ticket); the ticket for the NEXT item is drawn when an item starts, so its latency is hidden.  Ids
    // only grow per workgroup, so (q, before) walk forward; results do not depend on who computes what (tables are
    // indexed by item).
    int64_t q = 0, before = 0;
    int64_t id = blockIdx.x;
    while (id < total_items) {
        unsigned ticket = 0;
        if (tid == 0) ticket = atomicAdd(counter, 1u);
        for (;;) {
            const int64_t cnt = (B.grp_rows(q) + B.H - 1) / B.H;
            if (id < before + cnt) break;
            before += cnt;
            ++q;
        }
        const int64_t chunk = id - before;
        const int64_t grows = uniform64(B.grp_rows(q));
        const int64_t r_lo = uniform64(chunk * (int64_t)B.H);
        const int64_t r_hi = r_lo + B.H < grows ? r_lo + B.H : grows;
        const int ntr = (int)((r_hi - r_lo + TR - 1) / TR);  // row-tiles of the item
        // the group's strips
        const float* base[G];
        int64_t lim[G], diag[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int64_t t = q * G + g;
            const bool valid = t < B.T_loc;
            const int64_t rows = valid ? L.rows(t) : 0;
            lim[g] = uniform64(rows < r_hi ? rows : r_hi);                 // rows of strip g inside this item end here
            diag[g] = uniform64(valid ? L.strip(t) * W : 0);               // rows >= diag: the strip's diagonal block
            base[g] = Kp + uniform64(valid ? L.row_offset(t) * W : 0);
        }
        // split V of the group's columns -> LDS in operand order (chunk index = ((((g 2 + p) 4 + cs) 2 + s) 64 + lane))
#pragma unroll
        for (int it = 0; it < G * 2; ++it) {
            const int ch = it * B2_THREADS + tid;
            const int g = it >> 1;  // (= ch >> 10)
            const int l_ = ch & 63, s_ = (ch >> 6) & 1, cs_ = (ch >> 7) & 3, p_ = (ch >> 9) & 1;
            const int64_t j = diag[g] + 64 * cs_ + 32 * s_ + 8 * (l_ >> 4);
            const h2_u4 v = *reinterpret_cast<const h2_u4*>(Vt + (int64_t)(p_ * SMK + (l_ & 15)) * npad + j);
            *reinterpret_cast<h2_u4*>(sm_raw + B2_VS + ch * 16) = v;
        }
        h2_acc ccol[G][2], ctot[G][2], crow[2];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int v = 0; v < 4; ++v) { ccol[g][cb][v] = 0.f; ctot[g][cb][v] = 0.f; }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int v = 0; v < 4; ++v) crow[s][v] = 0.f;

        float4 pre[S][4];
        unsigned vi[S / G];
        auto load_tile = [&](int g, int64_t r0, float4 (&p4)[4]) {
            const bool part = r0 < lim[g];  // wave-uniform; absent tiles read one cached row
            const float* b = part ? base[g] : Kp;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int64_t i = r0 + wave + 8 * u;
                i = part ? (i < lim[g] ? i : lim[g] - 1) : 0;
                p4[u] = load_stream_at<float4>(b + i * W, (unsigned)lane * 16u);
            }
        };
        auto load_vi = [&](int64_t r0, unsigned& v) {
            v = *reinterpret_cast<const unsigned*>(vsrc + (r0 < r_hi ? r0 : r_lo));  // (row-tile start + 32 <= npad)
        };
        auto store_tile = [&](int parity, int g, int64_t rtl, int64_t r0, const float4 (&p4)[4], unsigned v) {
            unsigned char* tb = sm_raw + B2_TB + parity * 2 * H2_PIECE;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = wave + 8 * u;
                const float m = r0 + row < lim[g] ? sa : 0.f;  // rows the strip does not have: exact zeros
                unsigned h0, h1, m0, m1;
                h2_split2(p4[u].x * m, p4[u].y * m, h0, m0);
                h2_split2(p4[u].z * m, p4[u].w * m, h1, m1);
                h2_u2 h, mm;
                h.x = h0; h.y = h1; mm.x = m0; mm.y = m1;
                unsigned char* dst = tb + h2_off(row, lane >> 2) + (lane & 3) * 8;
                *reinterpret_cast<h2_u2*>(dst) = h;
                *reinterpret_cast<h2_u2*>(dst + H2_PIECE) = mm;
            }
            if (g == 0) *reinterpret_cast<unsigned*>(sm_raw + B2_VI + (int)(rtl & 1) * 2048 + vi_st) = v;
        };
        auto flush_rows = [&](int64_t rtl) {  // add the four column slices of row-tile rtl, store to the (group, row) table
            const float* rp = reinterpret_cast<const float*>(sm_raw + B2_RP) + (int)(rtl & 1) * 4 * TR * SMK;
            const int r = tid >> 4, c = tid & 15;
            const float sum = (rp[tid] + rp[TR * SMK + tid]) + (rp[2 * TR * SMK + tid] + rp[3 * TR * SMK + tid]);
            const int64_t i = r_lo + rtl * TR + r;
#if defined(RK_B2_PROBE_NOTABLE)
            if (i < r_hi && c < ktab && sum == 123.456f)
#else
            if (i < r_hi && c < ktab)
#endif
                __builtin_nontemporal_store(sum, &Prow[(B.prow_off(q) + i) * ktab + c]);
        };
        auto compute = [&](int parity, int g, int64_t rtl, int64_t r0) {
            if (g == 0 && rtl > 0) flush_rows(rtl - 1);  // (its row parts were complete before the last barrier)
            const unsigned char* tb = sm_raw + B2_TB + parity * 2 * H2_PIECE;
#if defined(RK_B2_PROBE_NOMFMA)
            if (sa == 0.f)
#endif
            if (r0 < lim[g]) {
                // ---- row part: A = rows 16 rb + lr, columns 64 cs + 32 s + 8 lq .. + 8; B = V of those columns ---------
                const unsigned char* vs = sm_raw + B2_VS + (g * 2 * 4 + cs) * 2 * 1024 + lane * 16;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    h2_u4 a[2], b[2];
                    a[0] = *reinterpret_cast<const h2_u4*>(tb + aoff[s]);
                    a[1] = *reinterpret_cast<const h2_u4*>(tb + aoff[s] + H2_PIECE);
                    b[0] = *reinterpret_cast<const h2_u4*>(vs + s * 1024);
                    b[1] = *reinterpret_cast<const h2_u4*>(vs + 4 * 2 * 1024 + s * 1024);
                    crow[s] = h2_mfma3(a, b, crow[s]);
                }
                // ---- column part (rows above the strip's diagonal block; tile-uniform): A = T^T through the
                // transposing read: columns 32 wave + 16 cb + (0 .. 15), rows 8 lq .. 8 lq + 7; B = V of the tile's rows
                if (r0 < diag[g]) {
                    const unsigned char* vp = sm_raw + B2_VI + (int)(rtl & 1) * 2048 + lane * 16;
                    h2_u4 vv[2];
                    vv[0] = *reinterpret_cast<const h2_u4*>(vp);
                    vv[1] = *reinterpret_cast<const h2_u4*>(vp + 1024);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
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
                        ccol[g][cb] = h2_mfma3(a, vv, ccol[g][cb]);
                    }
                }
            }
            if (B2_FOLD > 0 && (rtl & (B2_FOLD - 1)) == B2_FOLD - 1) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    ctot[g][cb] += ccol[g][cb];
#pragma unroll
                    for (int v = 0; v < 4; ++v) ccol[g][cb][v] = 0.f;
                }
            }
            if (g == G - 1) {  // the row-tile is complete: this wave's 16 rows x 64 columns part -> exchange buffer
                float* rp = reinterpret_cast<float*>(sm_raw + B2_RP) + ((int)(rtl & 1) * 4 + cs) * TR * SMK;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    rp[(16 * rb + 4 * lq + v) * SMK + lr] = crow[0][v] + crow[1][v];
                    crow[0][v] = 0.f;
                    crow[1][v] = 0.f;
                }
            }
        };

#pragma unroll
        for (int u = 0; u < S; ++u) {
            load_tile(u % G, r_lo + (int64_t)TR * (u / G), pre[u]);
            if (u % G == 0) load_vi(r_lo + (int64_t)TR * (u / G), vi[u / G]);
        }
        constexpr int RTS = S / G;  // row-tiles per trip of the unrolled loop
        int64_t rt0 = 0;
        for (; rt0 < ntr; rt0 += RTS) {
#pragma unroll
            for (int u = 0; u < S; ++u) {
                const int g = u % G;
                const int64_t rtl = rt0 + u / G;
                const int64_t r0 = r_lo + rtl * TR;
                store_tile(u & 1, g, rtl, r0, pre[u], vi[u / G]);
                load_tile(g, r0 + (int64_t)TR * RTS, pre[u]);
                if (g == 0) load_vi(r0 + (int64_t)TR * RTS, vi[u / G]);
                __syncthreads();  // image u & 1 (and, for g = 0, V of the rows) staged; everyone is done with the other image's
                                  // previous tile
                compute(u & 1, g, rtl, r0);
            }
        }
        __syncthreads();  // the last row parts are in the exchange buffer
        if (rt0 == ntr) flush_rows(ntr - 1);  // (a trip that ran past the item flushed it already)
        // column sums of the item: element v of ccol[g][cb] = Ycol[column 32 wave + 16 cb + 4 lq + v][rhs = lr] of strip g
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (r_lo < diag[g] && lr < ktab) {  // (strips without a column part in this item have no slot to fill)
                float* pc = Pcol + (((q * B.max_chunks + chunk) * G + g) * W) * ktab;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float val = ctot[g][cb][v] + ccol[g][cb][v];
#if defined(RK_B2_PROBE_NOTABLE)
                        if (val == 123.456f)
#endif
                        __builtin_nontemporal_store(val, &pc[(32 * wave + 16 * cb + 4 * lq + v) * ktab + lr]);
                    }
            }
        }
        if (tid == 0) *reinterpret_cast<unsigned*>(sm_raw + B2_NEXT) = ticket;
        __syncthreads();  // LDS is reused by the next item
        id = (int64_t)gridDim.x + __builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned*>(sm_raw + B2_NEXT));
    }
}

// Y[i][c] = alpha * unscale[c] * (row parts of every local group that stores row i
//                                 + column parts of the items of the group that owns column i) + shift * V[i][c]
// 64 outputs x 16 slices per workgroup, terms in a fixed order (bitwise reproducible), like symm_finish_kernel.
__global__ void __launch_bounds__(1024)
symm_blk_finish_kernel(BlkLayout B, int K, const float* __restrict__ Prow, const float* __restrict__ Pcol, float alpha,
                       float shift, const float* __restrict__ V, int64_t ldv, float* __restrict__ Y, int64_t ldy,
                       const float* __restrict__ colscale) {
    constexpr int NS = 16;
    __shared__ float red[NS][64];
    const SymLayout& L = B.L;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + tx;
    const bool live = e < L.n * K;
    float sum = 0.f;
    int64_t i = 0;
    int c = 0;
    if (live) {
        i = e / K;
        c = (int)(e - i * K);
        const int64_t blk = i / B2_W;
        const int64_t t0 = blk <= L.first ? 0 : (blk - L.first + L.stride - 1) / L.stride;  // first local strip holding row i
        const int64_t q0 = t0 / B2_G;
        const int64_t nrow = (t0 < B.T_loc && B.Q > q0) ? B.Q - q0 : 0;
        int64_t qc = 0, gc = 0, ncol = 0;
        if (blk >= L.first && (blk - L.first) % L.stride == 0) {
            const int64_t tc = (blk - L.first) / L.stride;
            qc = tc / B2_G;
            gc = tc - qc * B2_G;
            ncol = (blk * B2_W + B.H - 1) / B.H;  // items whose rows start above the strip's diagonal block
        }
        const int64_t jloc = i - blk * B2_W;
        auto term = [&](int64_t u) -> float {
            return u < nrow ? Prow[(B.prow_off(q0 + u) + i) * K + c]
                            : Pcol[(((qc * B.max_chunks + (u - nrow)) * B2_G + gc) * B2_W + jloc) * K + c];
        };
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        int64_t u = ty;
        for (; u + 3 * NS < nrow + ncol; u += 4 * NS) {
#pragma unroll
            for (int w = 0; w < 4; ++w) s4[w] += term(u + w * NS);
        }
        for (int w = 0; u < nrow + ncol; u += NS, ++w) s4[w] += term(u);
        sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    red[ty][tx] = sum;
    __syncthreads();
    if (ty == 0 && live) {
        sum = 0.f;
#pragma unroll
        for (int y = 0; y < NS; ++y) sum += red[y][tx];
        float y = alpha * sum * colscale[c];
        if (shift != 0.f) y += shift * V[i * ldv + c];
        Y[i * ldy + c] = y;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same block decomposition on the VECTOR units, for what the 16-bit matrix cores cannot do: float64 vectors over the
// float32 triangle (the converging configuration of the KRR solve) and float32 products of 1 .. 4 columns over shards
// too small to amortise the split prologue.  A wave owns batches of RB rows of the item; per row it loads the SAME row of
// the group's four strips (4 x 1 KiB, non-temporal), forms the row's partial dot products over all 1024 columns in its
// lanes and reduces them ONCE (symm_strip_kernel reduces once per strip and writes a table row per strip: here the
// (group, row) table is a quarter of that), and adds the column sums of the four strips in registers for the whole
// item.  Items, tickets, tables and the finishing kernel are the block kernel's.
// ------------------------------------------------------------------------------------------------------------------
constexpr int BR_RB = 2;        // rows per batch: 8 loads of 1 KiB per stage, two stages per wave
constexpr int BR_HMAX = 1024;   // rows per item at most (the row sums of an item wait in LDS)
constexpr int BR_THREADS = 256;

template <typename TS, typename T, int K>
__global__ void __launch_bounds__(BR_THREADS)
symm_blk_reg_kernel(const TS* __restrict__ Kp, BlkLayout B, const T* __restrict__ V, int64_t ldv,
                    T* __restrict__ Prow, T* __restrict__ Pcol, int64_t total_items, unsigned* __restrict__ counter) {
    using V16 = typename Vec16<TS>::type;
    constexpr int VW = Vec16<TS>::N;  // stored entries per 16-byte load: 4 / 2
    constexpr int G = B2_G, W = 64 * VW, RB = BR_RB;
    __shared__ T rowbuf[BR_HMAX * K];
    __shared__ T colacc[B2_G * 64 * (16 / sizeof(TS)) * K];
    __shared__ unsigned next_ticket;
    const SymLayout& L = B.L;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane_off = (unsigned)lane * 16u;
    int64_t q = 0, before = 0;
    int64_t id = blockIdx.x;
    while (id < total_items) {
        unsigned ticket = 0;
        if (tid == 0) ticket = atomicAdd(counter, 1u);
        for (;;) {
            const int64_t cnt = (B.grp_rows(q) + B.H - 1) / B.H;
            if (id < before + cnt) break;
            before += cnt;
            ++q;
        }
        const int64_t chunk = id - before;
        const int64_t grows = uniform64(B.grp_rows(q));
        const int64_t r_lo = uniform64(chunk * (int64_t)B.H);
        const int64_t r_hi = r_lo + B.H < grows ? r_lo + B.H : grows;
        const TS* base[G];
        int64_t lim[G], diag[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int64_t t = q * G + g;
            const bool valid = t < B.T_loc;
            const int64_t rows = valid ? L.rows(t) : 0;
            lim[g] = uniform64(rows < r_hi ? rows : r_hi);
            diag[g] = uniform64(valid ? L.strip(t) * W : 0);
            base[g] = Kp + uniform64(valid ? L.row_offset(t) * W : 0);
        }
        // this lane's 4 columns of every strip and their right-hand-side entries
        T vj[G][VW][K], col[G][VW][K];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const int64_t j = diag[g] + lane * VW + e;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    vj[g][e][c] = (lim[g] > 0 && j < L.n) ? V[j * ldv + c] : T(0);
                    col[g][e][c] = T(0);
                }
            }
        const int nb_item = (int)((r_hi - r_lo + RB - 1) / RB);
        const int nbw = wave < nb_item ? (nb_item - wave + 3) / 4 : 0;
        V16 stA[RB][G], stB[RB][G];
        T vA, vB;
        auto fetch = [&](V16 (&a)[RB][G], T& vi, int b) {
            const int64_t i0 = r_lo + (int64_t)(b * 4 + wave) * RB;
            {
                const int e = lane < RB * K ? lane : RB * K - 1;
                int64_t i = i0 + e / K;
                if (i > r_hi - 1) i = r_hi - 1;
                vi = V[i * ldv + e % K];
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int64_t i = i0 + r;
                    const bool part = i < lim[g];  // wave-uniform; absent rows read one cached row and are skipped below
                    a[r][g] = load_stream_at<V16>((part ? base[g] : Kp) + (part ? i : 0) * W, lane_off);
                }
        };
        auto consume = [&](const V16 (&a)[RB][G], T vi, int b) {
            const int64_t i0 = r_lo + (int64_t)(b * 4 + wave) * RB;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int64_t i = i0 + r;
                T dot[K];
#pragma unroll
                for (int c = 0; c < K; ++c) dot[c] = T(0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (i < lim[g]) {  // (wave-uniform)
                        const TS* as = reinterpret_cast<const TS*>(&a[r][g]);
                        T ae[VW];
#pragma unroll
                        for (int e = 0; e < VW; ++e) ae[e] = (T)as[e];
#pragma unroll
                        for (int e = 0; e < VW; ++e)
#pragma unroll
                            for (int c = 0; c < K; ++c) dot[c] += ae[e] * vj[g][e][c];
                        if (i < diag[g]) {
#pragma unroll
                            for (int c = 0; c < K; ++c) {
                                const T vic = lane_value(vi, r * K + c);
#pragma unroll
                                for (int e = 0; e < VW; ++e) col[g][e][c] += ae[e] * vic;
                            }
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const T tot = wave_total_lane63(dot[c]);
                    if (lane == 63 && i < r_hi) rowbuf[(i - r_lo) * K + c] = tot;
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
        __syncthreads();  // the item's row sums are in rowbuf
        T* pr = Prow + (B.prow_off(q) + r_lo) * K;
        for (int e = tid; e < (int)(r_hi - r_lo) * K; e += BR_THREADS) __builtin_nontemporal_store(rowbuf[e], &pr[e]);
        // column sums: the four waves add theirs in a fixed order (bitwise reproducible)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int e = 0; e < VW; ++e)
#pragma unroll
                        for (int c = 0; c < K; ++c) {
                            const int at = ((g * W) + lane * VW + e) * K + c;
                            colacc[at] = w == 0 ? col[g][e][c] : colacc[at] + col[g][e][c];
                        }
            }
            __syncthreads();
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (r_lo < diag[g]) {
                T* pc = Pcol + (((q * B.max_chunks + chunk) * G + g) * W) * K;
                for (int e = tid; e < W * K; e += BR_THREADS) __builtin_nontemporal_store(colacc[g * W * K + e], &pc[e]);
            }
        }
        if (tid == 0) next_ticket = ticket;
        __syncthreads();  // rowbuf / colacc are reused by the next item
        id = (int64_t)gridDim.x + __builtin_amdgcn_readfirstlane(next_ticket);
    }
}

template <typename T>
__global__ void __launch_bounds__(1024)
symm_blk_finish_t_kernel(BlkLayout B, int K, const T* __restrict__ Prow, const T* __restrict__ Pcol, T alpha, T shift,
                         const T* __restrict__ V, int64_t ldv, T* __restrict__ Y, int64_t ldy) {
    constexpr int NS = 16;
    __shared__ T red[NS][64];
    const SymLayout& L = B.L;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + tx;
    const bool live = e < L.n * K;
    T sum = T(0);
    int64_t i = 0;
    int c = 0;
    if (live) {
        i = e / K;
        c = (int)(e - i * K);
        const int W = L.W;
        const int64_t blk = i / W;
        const int64_t t0 = blk <= L.first ? 0 : (blk - L.first + L.stride - 1) / L.stride;
        const int64_t q0 = t0 / B2_G;
        const int64_t nrow = (t0 < B.T_loc && B.Q > q0) ? B.Q - q0 : 0;
        int64_t qc = 0, gc = 0, ncol = 0;
        if (blk >= L.first && (blk - L.first) % L.stride == 0) {
            const int64_t tc = (blk - L.first) / L.stride;
            qc = tc / B2_G;
            gc = tc - qc * B2_G;
            ncol = (blk * W + B.H - 1) / B.H;
        }
        const int64_t jloc = i - blk * W;
        auto term = [&](int64_t u) -> T {
            return u < nrow ? Prow[(B.prow_off(q0 + u) + i) * K + c]
                            : Pcol[(((qc * B.max_chunks + (u - nrow)) * B2_G + gc) * W + jloc) * K + c];
        };
        T s4[4] = {T(0), T(0), T(0), T(0)};
        int64_t u = ty;
        for (; u + 3 * NS < nrow + ncol; u += 4 * NS) {
#pragma unroll
            for (int w = 0; w < 4; ++w) s4[w] += term(u + w * NS);
        }
        for (int w = 0; u < nrow + ncol; u += NS, ++w) s4[w] += term(u);
        sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    red[ty][tx] = sum;
    __syncthreads();
    if (ty == 0 && live) {
        sum = T(0);
#pragma unroll
        for (int y = 0; y < NS; ++y) sum += red[y][tx];
        T y = alpha * sum;
        if (shift != T(0)) y += shift * V[i * ldv + c];
        Y[i * ldy + c] = y;
    }
}

static int64_t device_cus() {
    static std::atomic<int> cu_count[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int v = cu_count[dev & 63].load(std::memory_order_acquire);
    if (v == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        (void)hipGetLastError();
        v = cus > 0 ? cus : 256;
        cu_count[dev & 63].store(v, std::memory_order_release);
    }
    return v;
}

size_t symm_h2blk_ws_bytes(int64_t n, int64_t k, int64_t first, int64_t stride) {
    const BlkLayout B = make_blk_layout(n, first, stride, device_cus());
    const size_t npad = (size_t)round_up(n, 256);
    const size_t vt = 2 * SMK * npad * sizeof(unsigned short);
    const size_t prow = (size_t)round_up(B.prow_rows() * k, 4) * sizeof(float);
    const size_t pcol = (size_t)B.Q * B.max_chunks * B2_G * B2_W * k * sizeof(float);
    return 256 + vt + prow + pcol;
}

int symm_matmat_h2blk_impl(const float* Kp, int64_t n, int64_t first, int64_t stride, double kmax, const float* V,
                           int64_t k, int64_t ldv, double alpha, double shift, float* Y, int64_t ldy, void* ws,
                           size_t ws_bytes, hipStream_t st) {
    const int64_t cus = device_cus();
    const BlkLayout B = make_blk_layout(n, first, stride, cus);
    const size_t need = symm_h2blk_ws_bytes(n, k, first, stride);
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_symm_matmat_f16x2: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    const int64_t npad = round_up(n, 256);
    unsigned* info = static_cast<unsigned*>(ws);
    unsigned short* Vt = reinterpret_cast<unsigned short*>(static_cast<char*>(ws) + 256);
    float* Prow = reinterpret_cast<float*>(Vt + 2 * SMK * npad);
    float* Pcol = Prow + (size_t)round_up(B.prow_rows() * k, 4);
    int ea = 0;
    int rc = h2_prepare_v(V, n, npad, k, ldv, kmax, info, Vt, &ea, st);
    if (rc != RK_OK) return rc;
    if (B.T_loc > 0) {
        const int64_t total_items = blk_items(B);
        static std::atomic<unsigned long long> attr_devs{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(symm_blk_h2_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, B2_SMEM) != hipSuccess) {
                (void)hipGetLastError();
                set_error("rk_symm_matmat_f16x2: cannot reserve %d bytes of LDS", B2_SMEM);
                return RK_ERR_LAUNCH;
            }
            attr_devs.fetch_or(bit, std::memory_order_release);
        }
        const int64_t resident = cus;  // one workgroup per CU
        dim3 grid((unsigned)(total_items < resident ? total_items : resident));
        hipLaunchKernelGGL(symm_blk_h2_kernel, grid, dim3(B2_THREADS), B2_SMEM, st, Kp, B, Vt, npad,
                           (float)ldexp(1.0, ea), (int)k, Prow, Pcol, total_items, info + 60);  // (info is zeroed per call)
        rc = launch_status("rk_symm_matmat_f16x2(block)");
        if (rc != RK_OK) return rc;
    }
    hipLaunchKernelGGL(symm_blk_finish_kernel, dim3((unsigned)ceil_div(n * k, 64)), dim3(1024), 0, st, B, (int)k, Prow,
                       Pcol, (float)alpha, (float)shift, V, ldv, Y, ldy, reinterpret_cast<const float*>(info + 32));
    return launch_status("rk_symm_matmat_f16x2(finish)");
}

// ---- the vector-unit block kernel: float32 or float64 storage, vectors of the storage type or float64 over float32 -----
// resident workgroups per CU by register count: one column of float / float and double / double 141 - 144 registers
// (three waves per SIMD), everything else 201 - 211 (two)
template <typename TS, typename T>
static int blkreg_wgs(int64_t k) { return (k == 1 && sizeof(TS) == sizeof(T)) ? 3 : 2; }

template <typename TS, typename T>
static BlkLayout blkreg_layout(int64_t n, int64_t k, int64_t first, int64_t stride) {
    const int wg = blkreg_wgs<TS, T>(k);
    return make_blk_layout(n, first, stride, device_cus() * wg, BR_HMAX, sizeof(TS) == 8);
}

template <typename TS, typename T>
size_t symm_blkreg_ws_bytes(int64_t n, int64_t k, int64_t first, int64_t stride) {
    const BlkLayout B = blkreg_layout<TS, T>(n, k, first, stride);
    const size_t prow = (size_t)round_up(B.prow_rows() * k, 4) * sizeof(T);
    const size_t pcol = (size_t)B.Q * B.max_chunks * B2_G * B.L.W * k * sizeof(T);
    return 256 + prow + pcol;
}
template size_t symm_blkreg_ws_bytes<float, float>(int64_t, int64_t, int64_t, int64_t);
template size_t symm_blkreg_ws_bytes<float, double>(int64_t, int64_t, int64_t, int64_t);
template size_t symm_blkreg_ws_bytes<double, double>(int64_t, int64_t, int64_t, int64_t);

template <typename TS, typename T>
int symm_matmat_blkreg_impl(const TS* Kp, int64_t n, int64_t first, int64_t stride, const T* V, int64_t k, int64_t ldv,
                            double alpha, double shift, T* Y, int64_t ldy, void* ws, size_t ws_bytes, hipStream_t st) {
    const BlkLayout B = blkreg_layout<TS, T>(n, k, first, stride);
    const size_t need = symm_blkreg_ws_bytes<TS, T>(n, k, first, stride);
    if (ws == nullptr || ws_bytes < need) {
        set_error("rk_symm_matmat: workspace %zu B < required %zu B", ws_bytes, need);
        return RK_ERR_WORKSPACE;
    }
    unsigned* counter = static_cast<unsigned*>(ws);
    T* Prow = reinterpret_cast<T*>(static_cast<char*>(ws) + 256);
    T* Pcol = Prow + (size_t)round_up(B.prow_rows() * k, 4);
    if (hipMemsetAsync(counter, 0, 256, st) != hipSuccess) return launch_status("rk_symm_matmat(memset)");
    if (B.T_loc > 0) {
        const int64_t total_items = blk_items(B);
        const int wg = blkreg_wgs<TS, T>(k);
        const int64_t resident = device_cus() * wg;
        dim3 grid((unsigned)(total_items < resident ? total_items : resident));
        if (k == 1)
            hipLaunchKernelGGL((symm_blk_reg_kernel<TS, T, 1>), grid, dim3(BR_THREADS), 0, st, Kp, B, V, ldv, Prow, Pcol,
                               total_items, counter);
        else
            hipLaunchKernelGGL((symm_blk_reg_kernel<TS, T, 2>), grid, dim3(BR_THREADS), 0, st, Kp, B, V, ldv, Prow, Pcol,
                               total_items, counter);
        int rc = launch_status("rk_symm_matmat(block)");
        if (rc != RK_OK) return rc;
    }
    hipLaunchKernelGGL((symm_blk_finish_t_kernel<T>), dim3((unsigned)ceil_div(n * k, 64)), dim3(1024), 0, st, B, (int)k,
                       Prow, Pcol, (T)alpha, (T)shift, V, ldv, Y, ldy);
    return launch_status("rk_symm_matmat(finish)");
}
template int symm_matmat_blkreg_impl<float, float>(const float*, int64_t, int64_t, int64_t, const float*, int64_t, int64_t,
                                                   double, double, float*, int64_t, void*, size_t, hipStream_t);
template int symm_matmat_blkreg_impl<float, double>(const float*, int64_t, int64_t, int64_t, const double*, int64_t,
                                                    int64_t, double, double, double*, int64_t, void*, size_t, hipStream_t);
template int symm_matmat_blkreg_impl<double, double>(const double*, int64_t, int64_t, int64_t, const double*, int64_t,
                                                     int64_t, double, double, double*, int64_t, void*, size_t, hipStream_t);

}  // namespace rk
