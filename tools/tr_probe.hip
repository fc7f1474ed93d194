// Probe of ds_read_b64_tr_b16 (gfx950): which element lands in which lane.  LDS holds a [64 rows][64 cols] image of
// 16-bit values v = row * 64 + col; lane 4q+p of each 16-lane group g reads at row (4g + q), cols 4p .. 4p+3
// (byte address (row * 64 + 4p) * 2).  Prints, per lane, the four values received.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long* o) {
    __shared__ unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + (4 * g + q) * 64 + 4 * p));
    o[lane] = __builtin_bit_cast(unsigned long long, v);
}
int main() {
    unsigned long long* d;
    unsigned long long h[64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            const unsigned v = (unsigned)((h[l] >> (16 * e)) & 0xffff);
            printf("  (r%2u,c%2u)", v / 64, v % 64);
        }
        printf("\n");
    }
    return 0;
}
