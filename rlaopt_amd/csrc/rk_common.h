// Internal helpers shared by the HIP translation units of librk_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rk.h"

namespace rk {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char* fmt, ...);
void clear_error();

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return RK_ERR_LAUNCH;
    }
    return RK_OK;
}

#define RK_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            rk::set_error(__VA_ARGS__);  \
            return RK_ERR_INVALID;       \
        }                                \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

template <typename T>
struct Vec16;  // 16-byte vector of T
template <>
struct Vec16<float> {
    using type = float4;
    static constexpr int N = 4;
};
template <>
struct Vec16<double> {
    using type = double2;
    static constexpr int N = 2;
};

// 16-byte load of data that is streamed exactly once (the operator): non-temporal, so that it
// does not evict the small re-read operands (vectors, partial tables) from L2 / MALL
template <typename V16>
__device__ __forceinline__ V16 load_stream(const void* p) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return __builtin_bit_cast(V16, raw);
}

// wave total in lane 63 without LDS: 4 DPP adds inside each 16-lane row, then row broadcasts
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fetch(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <typename T>
__device__ __forceinline__ T wave_total_lane63(T v) {
    v += dpp_fetch<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_fetch<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_fetch<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_fetch<0x140, 0xF>(v);  // row_mirror: every lane holds its 16-lane row sum
    v += dpp_fetch<0x142, 0xA>(v);  // row_bcast15 -> rows 1 and 3
    v += dpp_fetch<0x143, 0xC>(v);  // row_bcast31 -> rows 2 and 3: lane 63 holds the total
    return v;
}

// tell the compiler a value is wave-uniform (keeps it - and what is derived from it - in SGPRs)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    const int lo = __builtin_amdgcn_readfirstlane((int)v);
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((int64_t)hi << 32) | (unsigned int)lo;
}

// Streaming 16-byte load from (wave-uniform base) + (32-bit per-lane byte offset).  The base is
// pinned in an SGPR pair and how it was computed is hidden from the optimiser, so the load is
// selected in its saddr form: no 64-bit VGPR address temporaries (those get allocated on top of
// in-flight load destinations and make the compiler drain the queue before every batch).
template <typename V16>
__device__ __forceinline__ V16 load_stream_at(const void* uniform_base, unsigned lane_off) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) char* gchar_ptr;
    typedef const __attribute__((address_space(1))) u32x4* gvec_ptr;
    gchar_ptr g = (gchar_ptr)uniform_base;
    asm volatile("" : "+s"(g));
    const u32x4 raw = __builtin_nontemporal_load((gvec_ptr)(g + lane_off));
    return __builtin_bit_cast(V16, raw);
}

// wave-uniform copy of lane `src`'s value (src is a compile-time-uniform index): lands in SGPRs
__device__ __forceinline__ float lane_value(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ double lane_value(double v, int src) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)bits, src);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

}  // namespace rk
