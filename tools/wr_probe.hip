// What do small writes cost beside a streaming read on MI355X?  (Round 4: the packed symmetric multi-RHS product loses
// 2 ms of 15 to 2 GB of partial-table stores beside its 80 GB read stream - far more than their volume.)
//
// Persistent workgroups (256 threads, WGPC per CU) stream a large buffer in 32 KiB tiles exactly like the strip kernels
// (a wave loads whole 1 KiB rows with non-temporal dwordx4, two register stages of 8 rows), and every `every` tiles a
// workgroup writes `bytes` bytes to its own advancing region of a table.  Store forms:
//   form 0: dword per lane (tid-contiguous)      form 1: dwordx4 per lane (tid-contiguous)
//   policy 0: non-temporal   1: plain   2: sc1 (write-through, via atomic relaxed agent store for dword only)
// Build: hipcc -O3 --offload-arch=gfx950 tools/wr_probe.hip -o tools/_build/wr_probe
// Run:   tools/_build/wr_probe [GB=40]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld_nt(const f4* p) { return __builtin_nontemporal_load(p); }

template <int FORM, int POLICY>
__global__ void __launch_bounds__(256)
stream_kernel(const f4* __restrict__ src, long long tiles, float* __restrict__ table, long long table_floats,
              int every, int bytes, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4 stA[8], stB[8];
    // tile = 32 rows x 64 f4 (32 KiB); wave w takes rows w, w + 4, ...
    auto fetch = [&](f4 (&st)[8], long long tile) {
        const f4* base = src + tile * 2048;
#pragma unroll
        for (int u = 0; u < 8; ++u) st[u] = ld_nt(base + (wave + 4 * u) * 64 + lane);
    };
    auto consume = [&](const f4 (&st)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += st[u];
    };
    long long region = (long long)blockIdx.x * (table_floats / gridDim.x);
    const long long region_end = region + table_floats / gridDim.x;
    long long wpos = region;
    int count = 0;
    auto maybe_write = [&]() {
        if (every <= 0) return;
        if (++count < every) return;
        count = 0;
        const int nfl = bytes / 4;
        if (wpos + nfl > region_end) wpos = region;
        float* dst = table + wpos;
        if (FORM == 0) {
            for (int e = threadIdx.x; e < nfl; e += 256) {
                const float v = acc.x + (float)e;
                if (POLICY == 0) __builtin_nontemporal_store(v, dst + e);
                else if (POLICY == 1) dst[e] = v;
                else __hip_atomic_store(dst + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (FORM == 2) {  // float atomics (no return), dword per lane
            for (int e = threadIdx.x; e < nfl; e += 256) atomicAdd(dst + e, acc.x + 1.0f);
        } else {
            for (int e = threadIdx.x * 4; e < nfl; e += 1024) {
                f4 v = acc;
                v.x += (float)e;
                if (POLICY == 0) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst + e));
                else *reinterpret_cast<f4*>(dst + e) = v;
            }
        }
        wpos += nfl;
    };
    long long tl = blockIdx.x;
    if (tl < tiles) {
        fetch(stA, tl);
        for (;;) {
            long long nx = tl + gridDim.x;
            if (nx >= tiles) { consume(stA); maybe_write(); break; }
            fetch(stB, nx);
            consume(stA);
            maybe_write();
            tl = nx;
            nx = tl + gridDim.x;
            if (nx >= tiles) { consume(stB); maybe_write(); break; }
            fetch(stA, nx);
            consume(stB);
            maybe_write();
            tl = nx;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) sink[threadIdx.x] = acc.x;
}

template <int FORM, int POLICY>
static float run(const f4* src, long long tiles, float* table, long long table_floats, int every, int bytes, float* sink,
                 int grid, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((stream_kernel<FORM, POLICY>), dim3(grid), dim3(256), 0, 0, src, tiles, table, table_floats, every,
                       bytes, sink);
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_kernel<FORM, POLICY>), dim3(grid), dim3(256), 0, 0, src, tiles, table, table_floats,
                           every, bytes, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float t;
        CK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    float best = ms[0];
    for (float t : ms) best = t < best ? t : best;
    return best;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 40.0;
    const long long tiles = (long long)(gb * 1e9 / 32768.0);
    f4* src;
    float *table, *sink;
    const long long table_floats = 2ll << 30;  // 8 GiB of table space
    CK(hipMalloc(&src, tiles * 32768ll));
    CK(hipMalloc(&table, table_floats * 4));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(src, 0x11, tiles * 32768ll));
    CK(hipMemset(table, 0, table_floats * 4));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs, read stream %.1f GB\n", p.name, cus, tiles * 32768.0 / 1e9);
    struct Cfg { const char* name; int form, policy, every, bytes; double table_mb; };
    const Cfg cfgs[] = {
        {"no writes", 0, 0, 0, 0, 8192},
        {"2 KiB / tile dword nt, 8 GiB table (k=16 today)", 0, 0, 1, 2048, 8192},
        {"1 KiB / tile x4 nt, 8 GiB", 1, 0, 1, 1024, 8192},
        {"512 B / tile x4 nt, 8 GiB", 1, 0, 1, 512, 8192},
        {"256 B / tile x4 nt, 8 GiB", 1, 0, 1, 256, 8192},
        {"4 KiB / 32 tiles x4 nt, 8 GiB", 1, 0, 32, 4096, 8192},
        {"2 KiB / 32 tiles x4 nt, 8 GiB", 1, 0, 32, 2048, 8192},
        {"2 KiB / tile x4 nt, table 1 GiB (rewritten)", 1, 0, 1, 2048, 1024},
        {"2 KiB / tile x4 nt, table 128 MiB", 1, 0, 1, 2048, 128},
        {"2 KiB / tile x4 nt, table 32 MiB", 1, 0, 1, 2048, 32},
        {"2 KiB / tile x4 nt, table 8 MiB", 1, 0, 1, 2048, 8},
        {"2 KiB / tile x4 plain, table 128 MiB", 1, 1, 1, 2048, 128},
        {"2 KiB / tile x4 plain, table 32 MiB", 1, 1, 1, 2048, 32},
        {"2 KiB / tile x4 plain, table 8 MiB", 1, 1, 1, 2048, 8},
        {"2 KiB / tile atomics f32, table 8 GiB", 2, 0, 1, 2048, 8192},
        {"2 KiB / tile atomics f32, table 128 MiB", 2, 0, 1, 2048, 128},
        {"2 KiB / tile atomics f32, table 16 MiB", 2, 0, 1, 2048, 16},
        {"512 B / tile atomics f32, table 16 MiB", 2, 0, 1, 512, 16},
        {"512 B / tile atomics f32, table 128 MiB", 2, 0, 1, 512, 128},
    };
    for (int wgpc : {2}) {
        const int grid = cus * wgpc;
        printf("## %d workgroups per CU\n", wgpc);
        for (const Cfg& c : cfgs) {
            float ms;
#define RUN(F, P) ms = run<F, P>(src, tiles, table, (long long)(c.table_mb * 262144.0), c.every, c.bytes, sink, grid, 3)
            if (c.form == 0 && c.policy == 0) RUN(0, 0);
            else if (c.form == 0 && c.policy == 1) RUN(0, 1);
            else if (c.form == 0 && c.policy == 2) RUN(0, 2);
            else if (c.form == 1 && c.policy == 0) RUN(1, 0);
            else if (c.form == 2) RUN(2, 0);
            else RUN(1, 1);
            const double wbytes = c.every ? (double)tiles / c.every * c.bytes : 0.0;
            printf("%-48s %8.3f ms  read %.0f GB/s  (+%.2f GB written)\n", c.name, ms, tiles * 32768.0 / ms / 1e6,
                   wbytes / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
