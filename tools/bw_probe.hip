// bw_probe.hip -- measures what a pure float4 read stream and a pure int4 write stream reach on this GPU,
// as the practical ceilings for k_threshold_v4 (read 4 B/px) and k_relabel_v4 (write 4 B/px).
// build: hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o /tmp/bw_probe ; run: /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ in, int64_t n4, unsigned *sink)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * UNROLL;
    unsigned acc = 0;
    for (; i < n4; i += stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            int64_t j = i + (int64_t)u * blockDim.x;
            if (j < n4) v[u] = NT ? __builtin_nontemporal_load(&in[j]) : in[j]; else v[u] = (f4)(0.f);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += (v[u].x >= 160.f) + (v[u].y >= 160.f) + (v[u].z >= 160.f) + (v[u].w >= 160.f);
    }
    if (acc == 0xffffffffu) *sink = acc;
}

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_write(i4 *__restrict__ out, int64_t n4)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * UNROLL;
    const i4 z = (i4)(0);
    for (; i < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            int64_t j = i + (int64_t)u * blockDim.x;
            if (j < n4) { if (NT) __builtin_nontemporal_store(z, &out[j]); else out[j] = z; }
        }
    }
}

// relabel-shaped write: a one-shot grid, every workgroup writes S consecutive int4 per thread (its own contiguous block), optionally
// after a "table" phase like k_relabel_v4's (1 KB per workgroup from a table in HBM -> LDS -> barrier; the stored value depends on it)
template <int S, bool TABLES>
__global__ __launch_bounds__(256) void k_write_blocks(i4 *__restrict__ out, int64_t n4, const int *__restrict__ table)
{
    __shared__ int lds[256];
    int add = 0;
    if (TABLES) {
        lds[threadIdx.x] = table[((int64_t)blockIdx.x * 256 + threadIdx.x) & ((16 << 20) - 1)];
        __syncthreads();
        add = lds[(threadIdx.x * 7) & 255];
    }
    const int64_t base = (int64_t)blockIdx.x * 256 * S;
    const i4 z = (i4)(add);
#pragma unroll
    for (int u = 0; u < S; u++) {
        const int64_t j = base + (int64_t)u * 256 + threadIdx.x;
        if (j < n4) __builtin_nontemporal_store(z, &out[j]);
    }
}

template <typename F>
double time_ms(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; r++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const int64_t n4 = 2707ll * 181 * 360 / 4;          // the bench slab: 705.6 MB
    void *buf = nullptr;
    unsigned *sink = nullptr;
    hipMalloc(&buf, n4 * 16);
    hipMalloc((void **)&sink, 4);
    hipMemset(buf, 0, n4 * 16);
    const double gb = n4 * 16 / 1e9;
    int grids[] = {1024, 2048, 4096, 8192, 16384, 65536};
    for (int g : grids) {
        double r1 = time_ms([&] { k_read<1, false><<<g, 256>>>((f4 *)buf, n4, sink); }, 20);
        double r4 = time_ms([&] { k_read<4, false><<<g, 256>>>((f4 *)buf, n4, sink); }, 20);
        double r8 = time_ms([&] { k_read<8, false><<<g, 256>>>((f4 *)buf, n4, sink); }, 20);
        double r4n = time_ms([&] { k_read<4, true><<<g, 256>>>((f4 *)buf, n4, sink); }, 20);
        double w1 = time_ms([&] { k_write<1, false><<<g, 256>>>((i4 *)buf, n4); }, 20);
        double w4 = time_ms([&] { k_write<4, false><<<g, 256>>>((i4 *)buf, n4); }, 20);
        double w4n = time_ms([&] { k_write<4, true><<<g, 256>>>((i4 *)buf, n4); }, 20);
        printf("grid %6d  read x1 %.0f x4 %.0f x8 %.0f x4nt %.0f GB/s | write x1 %.0f x4 %.0f x4nt %.0f GB/s\n", g, gb / r1 * 1e3, gb / r4 * 1e3,
               gb / r8 * 1e3, gb / r4n * 1e3, gb / w1 * 1e3, gb / w4 * 1e3, gb / w4n * 1e3);
    }
    // one-shot grid (one float4 per thread)
    {
        int g = (int)((n4 + 255) / 256);
        double r = time_ms([&] { k_read<1, false><<<g, 256>>>((f4 *)buf, n4, sink); }, 20);
        double w = time_ms([&] { k_write<1, false><<<g, 256>>>((i4 *)buf, n4); }, 20);
        printf("one-shot grid %d: read %.0f GB/s write %.0f GB/s\n", g, gb / r * 1e3, gb / w * 1e3);
    }
    // how many stores per thread a short-lived workgroup should issue, and what a table phase in front costs
    {
        int *table = nullptr;
        hipMalloc((void **)&table, (size_t)(16 << 20) * 4);
        hipMemset(table, 0, (size_t)(16 << 20) * 4);
#define WB(S) do { const int g = (int)((n4 + 256 * S - 1) / (256 * S)); \
        double a = time_ms([&] { k_write_blocks<S, false><<<g, 256>>>((i4 *)buf, n4, table); }, 20); \
        double b = time_ms([&] { k_write_blocks<S, true><<<g, 256>>>((i4 *)buf, n4, table); }, 20); \
        printf("write blocks, %2d stores/thread (%6d workgroups): %.0f GB/s; with a table phase in front: %.0f GB/s\n", S, g, gb / a * 1e3, gb / b * 1e3); } while (0)
        WB(1); WB(2); WB(3); WB(4); WB(6); WB(8); WB(16); WB(32);
        hipFree(table);
    }
    hipFree(buf);
    return 0;
}
