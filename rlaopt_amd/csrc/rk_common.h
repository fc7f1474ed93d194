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

// wave-wide sum (all 64 lanes end with the total)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace rk
