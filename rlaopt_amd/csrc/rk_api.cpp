// Error state and version of librk_hip.so (see include/rk.h for the conventions).
#include <stdarg.h>
#include <stdio.h>

#include "rk_common.h"

namespace rk {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void clear_error() { g_err[0] = '\0'; }

}  // namespace rk

extern "C" const char* rk_last_error(void) { return rk::g_err; }
extern "C" int rk_abi_version(void) { return 23; }

// ---- wire buffer of the reduce-scatter that sums the ranks' partial products (rlaopt_amd/dist.py) ----------
namespace rk {
template <typename T>
__global__ void __launch_bounds__(256)
spmd_pack_kernel(const T* __restrict__ part, int64_t n, int64_t k, int64_t ldp, const double* __restrict__ tail,
                 int64_t t, int64_t step, int64_t slots, T* __restrict__ out) {
    const int64_t slot = step * k + t;
    const int64_t total = slots * slot;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / slot, o = e - r * slot;
        T v;
        if (o < step * k) {
            const int64_t row = r * step + o / k;
            v = row < n ? part[row * ldp + o % k] : T(0);
        } else {
            v = (T)tail[o - step * k];
        }
        out[e] = v;
    }
}
}  // namespace rk

extern "C" int rk_spmd_pack(int dtype, const void* part, int64_t n, int64_t k, int64_t ldp, const double* tail,
                            int64_t t, int64_t step, int64_t slots, void* out, void* stream) {
    using namespace rk;
    clear_error();
    RK_REQUIRE(dtype == RK_F32 || dtype == RK_F64, "rk_spmd_pack: bad dtype %d", dtype);
    RK_REQUIRE(part && out && (t == 0 || tail), "rk_spmd_pack: null pointer");
    RK_REQUIRE(n > 0 && k > 0 && ldp >= k && t >= 0 && step > 0 && slots > 0 && step * slots >= n,
               "rk_spmd_pack: bad sizes (n=%lld k=%lld step=%lld slots=%lld)", (long long)n, (long long)k,
               (long long)step, (long long)slots);
    const int64_t total = slots * (step * k + t);
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == RK_F32)
        hipLaunchKernelGGL(spmd_pack_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)part, n, k, ldp, tail, t,
                           step, slots, (float*)out);
    else
        hipLaunchKernelGGL(spmd_pack_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double*)part, n, k, ldp, tail,
                           t, step, slots, (double*)out);
    return launch_status("rk_spmd_pack");
}
