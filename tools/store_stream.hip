// Plain store streams on gfx950: what does the PATTERN of a full-rate write stream change?  (the write kernel k_relabel_v5 is measured against
// k_stream_store -- variant A here -- in bench.py's stream_ceiling)
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/store_stream tools/store_stream.hip && tools/exp/store_stream [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// chunk of 256 * U * 16 bytes per workgroup; MAP: 0 linear, 1 one contiguous eighth of the buffer per XCD (blockIdx % 8)
template <int U, bool NT, int MAP, int THREADS>
__global__ __launch_bounds__(THREADS) void k_store(i32x4 *__restrict__ dst, int64_t n16, int64_t nchunks)
{
    int64_t c = blockIdx.x;
    if (MAP == 1) { const int64_t per = (nchunks + 7) / 8; c = (c & 7) * per + (c >> 3); if (c >= nchunks) return; }
    const int64_t base = c * (THREADS * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t i = base + (int64_t)u * THREADS;
        if (i < n16) { if (NT) __builtin_nontemporal_store((i32x4)(0), dst + i); else dst[i] = (i32x4)(0); }
    }
}
template <int U, bool NT, int MAP>
__global__ __launch_bounds__(256) void k_load(const i32x4 *__restrict__ src, int64_t n16, int64_t nchunks, int *__restrict__ sink)
{
    int64_t c = blockIdx.x;
    if (MAP == 1) { const int64_t per = (nchunks + 7) / 8; c = (c & 7) * per + (c >> 3); if (c >= nchunks) return; }
    const int64_t base = c * (256 * U) + threadIdx.x;
    i32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int64_t i = min(base + (int64_t)u * 256, n16 - 1); v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i]; }
    int acc = 0;
#pragma unroll
    for (int u = 0; u < U; u++) acc |= v[u].x | v[u].y | v[u].z | v[u].w;
    if (acc == 0x5a5a5a5a) *sink = acc;
}
// a fixed grid walking the chunks
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_store_persistent(i32x4 *__restrict__ dst, int64_t n16, int64_t nchunks)
{
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t base = c * (256 * U) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + (int64_t)u * 256;
            if (i < n16) { if (NT) __builtin_nontemporal_store((i32x4)(0), dst + i); else dst[i] = (i32x4)(0); }
        }
    }
}
// every lane 64 contiguous bytes (four 16-byte stores side by side)
template <bool NT>
__global__ __launch_bounds__(256) void k_store_lane64(i32x4 *__restrict__ dst, int64_t n16)
{
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int u = 0; u < 4; u++) { const int64_t i = base + u; if (i < n16) { if (NT) __builtin_nontemporal_store((i32x4)(0), dst + i); else dst[i] = (i32x4)(0); } }
}
// 8-byte stores
template <bool NT>
__global__ __launch_bounds__(256) void k_store8(i32x2 *__restrict__ dst, int64_t n8)
{
    const int64_t base = (int64_t)blockIdx.x * (256 * 16) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; u++) { const int64_t i = base + u * 256; if (i < n8) { if (NT) __builtin_nontemporal_store((i32x2)(0), dst + i); else dst[i] = (i32x2)(0); } }
}

template <typename F>
static double best_ms(F launch, int reps = 7)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        CHK(hipEventRecord(a, 0));
        launch();
        CHK(hipEventRecord(b, 0));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, (double)ms);
    }
    return best;
}
int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 705;
    const size_t bytes = mb << 20;
    char *p;
    CHK(hipMalloc(&p, bytes));
    CHK(hipMemset(p, 1, bytes));
    int *sink; CHK(hipMalloc(&sink, 64));
    const int64_t n16 = (int64_t)(bytes / 16);
    auto report = [&](const char *name, double ms) { printf("%-64s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, bytes / ms / 1e6); };
#define RUN(name, U, NT, MAP, TH) do { const int64_t nch = (n16 + (TH) * (U) - 1) / ((TH) * (U)); const int64_t g = (MAP) ? ((nch + 7) / 8) * 8 : nch; \
        report(name, best_ms([&]() { hipLaunchKernelGGL((k_store<U, NT, MAP, TH>), dim3((unsigned)g), dim3(TH), 0, 0, (i32x4 *)p, n16, nch); })); } while (0)
    printf("buffer %zu MB\n", mb);
    for (int pass = 0; pass < 2; pass++) {
        RUN("A  nt, 32 KB per workgroup (k_stream_store)", 8, true, 0, 256);
        RUN("B  plain stores, 32 KB per workgroup", 8, false, 0, 256);
        RUN("C  nt, 64 KB per workgroup", 16, true, 0, 256);
        RUN("C2 nt, 16 KB per workgroup", 4, true, 0, 256);
        RUN("D  nt, 4 KB per workgroup", 1, true, 0, 256);
        RUN("E  nt, 32 KB per workgroup, one eighth of the buffer per XCD", 8, true, 1, 256);
        RUN("E2 plain, 32 KB per workgroup, one eighth per XCD", 8, false, 1, 256);
        RUN("G  nt, 1024 threads, 32 KB per workgroup", 2, true, 0, 1024);
        RUN("G2 nt, 512 threads, 32 KB per workgroup", 4, true, 0, 512);
        RUN("G3 nt, 64 threads, 8 KB per workgroup", 8, true, 0, 64);
        { const int64_t nch = (n16 + 2047) / 2048;
          report("F  nt, fixed grid 2048 workgroups walking 32 KB chunks", best_ms([&]() { hipLaunchKernelGGL((k_store_persistent<8, true>), dim3(2048), dim3(256), 0, 0, (i32x4 *)p, n16, nch); }));
          report("F2 nt, fixed grid 1024 workgroups", best_ms([&]() { hipLaunchKernelGGL((k_store_persistent<8, true>), dim3(1024), dim3(256), 0, 0, (i32x4 *)p, n16, nch); }));
          report("F3 nt, fixed grid 4096 workgroups", best_ms([&]() { hipLaunchKernelGGL((k_store_persistent<8, true>), dim3(4096), dim3(256), 0, 0, (i32x4 *)p, n16, nch); })); }
        report("H  nt, 64 contiguous bytes per lane", best_ms([&]() { hipLaunchKernelGGL((k_store_lane64<true>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (i32x4 *)p, n16); }));
        report("H2 plain, 64 contiguous bytes per lane", best_ms([&]() { hipLaunchKernelGGL((k_store_lane64<false>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, (i32x4 *)p, n16); }));
        report("J  nt, 8-byte stores, 32 KB per workgroup", best_ms([&]() { hipLaunchKernelGGL((k_store8<true>), dim3((unsigned)((n16 * 2 + 4095) / 4096)), dim3(256), 0, 0, (i32x2 *)p, n16 * 2); }));
        report("I  hipMemsetAsync", best_ms([&]() { CHK(hipMemsetAsync(p, 0, bytes, 0)); }));
#define RUNL(name, U, NT, MAP) do { const int64_t nch = (n16 + 256 * (U) - 1) / (256 * (U)); const int64_t g = (MAP) ? ((nch + 7) / 8) * 8 : nch; \
        report(name, best_ms([&]() { hipLaunchKernelGGL((k_load<U, NT, MAP>), dim3((unsigned)g), dim3(256), 0, 0, (const i32x4 *)p, n16, nch, sink); })); } while (0)
        RUNL("LA nt loads, 32 KB per workgroup (k_stream_load)", 8, true, 0);
        RUNL("LB plain loads, 32 KB per workgroup", 8, false, 0);
        RUNL("LE nt loads, one eighth of the buffer per XCD", 8, true, 1);
        RUNL("LE2 plain loads, one eighth per XCD", 8, false, 1);
        RUNL("LC nt loads, 16 KB per workgroup", 4, true, 0);
        RUNL("LC2 nt loads, 16 KB per workgroup, one eighth per XCD", 4, true, 1);
        printf("\n");
    }
    return 0;
}
