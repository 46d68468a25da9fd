// gridbar2.hip -- round 6: what does a device-wide barrier INSIDE a resident kernel cost on MI355X, against the kernel boundary it
// would replace?  (Verdict item 1a asked for one persistent "tables" kernel over the seven launch-bound kernels of the pass' tail.)
//
//   hipcc --offload-arch=gfx950 -O3 tools/gridbar2.hip -o tools/exp/gridbar2 && tools/exp/gridbar2
//
// Every variant runs K "phases" of trivial work (one dependent load + one store per thread: the shape of the tail kernels, which are
// chains of dependent loads) separated by a barrier over G workgroups of 256 threads:
//   launches   K kernel launches back to back on one stream (what the pass does now)
//   one        one counter, device-scope atomics, every workgroup polls it            (NOTES round 3: ~78 ns per workgroup)
//   xcd        two levels: arrive on the counter of the workgroup's XCD with an atomic that is performed in that XCD's L2
//              (workgroup-scope instruction: no sc1 -- all workgroups of an XCD share the L2), the last arriver of an XCD adds to the
//              device-scope counter; release the same way back (one poller per XCD on the device word, the rest on their XCD's word)
//   flags      every workgroup stores its epoch to its own 128-byte line; workgroup 0 gathers them with one load per lane and stores
//              eight per-XCD release words
// Prints microseconds per phase.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define LINE 32      // uint32 per 128-byte line
// every wait is bounded (a barrier over workgroups that are not all resident must not hang the box): ~0.3 s, then the kernel gives up
#define WAIT_WHILE(cond) do { uint32_t spins_ = 0; while (cond) { __builtin_amdgcn_s_sleep(1); if (++spins_ > 4000000u) { *g_fail = 1u; return; } } } while (0)
__device__ uint32_t g_fail_word;

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xfu; }      // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ void work(uint32_t *buf, int n, int phase)
{
    // a dependent pair: index from one load, value from the next, one store
    const int i = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)n);
    const uint32_t j = __hip_atomic_load(&buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % (unsigned)n;
    const uint32_t v = __hip_atomic_load(&buf[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&buf[i], (v + (uint32_t)phase) % (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_phase(uint32_t *buf, int n, int phase) { work(buf, n, phase); }

// mode 0: one counter
__global__ __launch_bounds__(256) void k_bar_one(uint32_t *buf, int n, int K, uint32_t *cnt)
{
    uint32_t *g_fail = &g_fail_word;
    for (int k = 0; k < K; k++) {
        work(buf, n, k);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = (uint32_t)(k + 1) * gridDim.x;
            WAIT_WHILE(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want);
        }
        __syncthreads();
    }
}

// mode 1: per-XCD counters in the XCD's own L2 + one device-scope counter
__global__ __launch_bounds__(256) void k_bar_xcd(uint32_t *buf, int n, int K, uint32_t *xc /* [8][LINE]: arrive */, uint32_t *xr /* [8][LINE]: release */,
                                                 uint32_t *gc /* device-scope counter */, uint32_t *reg /* [8][LINE] + [LINE]: census of THIS launch */)
{
    uint32_t *g_fail = &g_fail_word;
    const uint32_t x = xcc_id();
    // census of this very launch: every workgroup registers with its XCD, a one-counter barrier makes the counts final
    __shared__ uint32_t s_mine, s_nx;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&reg[x * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&reg[8 * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        WAIT_WHILE(__hip_atomic_load(&reg[8 * LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x);
        uint32_t nx = 0;
        for (int q = 0; q < 8; q++) nx += __hip_atomic_load(&reg[q * LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
        s_mine = __hip_atomic_load(&reg[x * LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_nx = nx;
    }
    __syncthreads();
    const uint32_t mine = s_mine, nx_active = s_nx;
    for (int k = 0; k < K; k++) {
        work(buf, n, k);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            // (the work's stores were device-scope: visible at the coherence point once acknowledged)
            const uint32_t old = __hip_atomic_fetch_add(&xc[x * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old + 1u == (uint32_t)(k + 1) * mine) {
                // last of this XCD: arrive on the device word, wait for all XCDs, release the XCD
                __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t want = (uint32_t)(k + 1) * nx_active;
                WAIT_WHILE(__hip_atomic_load(gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want);
                // (read-modify-write instructions are performed in the L2; a workgroup-scope LOAD may be served by the CU's own L1)
                __hip_atomic_fetch_max(&xr[x * LINE], (uint32_t)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                WAIT_WHILE(__hip_atomic_fetch_add(&xr[x * LINE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)(k + 1));
            }
        }
        __syncthreads();
    }
}

// mode 2: flag gather by workgroup 0, per-XCD release words (device scope)
__global__ __launch_bounds__(256) void k_bar_flags(uint32_t *buf, int n, int K, uint32_t *flags /* [G][LINE] */, uint32_t *rel /* [8][LINE] */)
{
    uint32_t *g_fail = &g_fail_word;
    const uint32_t x = blockIdx.x & 7u;
    for (int k = 0; k < K; k++) {
        work(buf, n, k);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        const uint32_t ep = (uint32_t)(k + 1);
        if (blockIdx.x == 0) {
            for (uint32_t g = threadIdx.x + 1; g < gridDim.x; g += 256)
                WAIT_WHILE(__hip_atomic_load(&flags[g * LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ep);
            __syncthreads();
            if (threadIdx.x < 8) __hip_atomic_store(&rel[threadIdx.x * LINE], ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (threadIdx.x == 0) {
            __hip_atomic_store(&flags[blockIdx.x * LINE], ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            WAIT_WHILE(__hip_atomic_load(&rel[x * LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ep);
        }
        __syncthreads();
    }
}

__global__ void k_xcc_census(uint32_t *per_xcd, uint32_t *modmatch)
{
    if (threadIdx.x == 0) {
        const uint32_t x = xcc_id();
        atomicAdd(&per_xcd[x], 1u);
        if (x == (blockIdx.x & 7u)) atomicAdd(modmatch, 1u);
    }
}

int main()
{
    const int n = 1 << 20, K = 200;
    uint32_t *buf, *scratch;
    CHK(hipMalloc(&buf, (size_t)n * 4));
    CHK(hipMalloc(&scratch, (size_t)(4096 + 64) * LINE * 4));
    std::vector<uint32_t> init((size_t)n);
    for (int i = 0; i < n; i++) init[(size_t)i] = (uint32_t)((i * 2654435761u) % (unsigned)n);
    CHK(hipMemcpy(buf, init.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int G : {64, 128, 256, 512, 1024}) {
        float ms[4] = {0, 0, 0, 0};
        // census: which XCD do the workgroups of a G-launch land on (is it blockIdx % 8?)
        CHK(hipMemsetAsync(scratch, 0, (size_t)(4096 + 64) * LINE * 4, s));
        uint32_t *per_xcd = scratch, *modmatch = scratch + 16;
        k_xcc_census<<<G, 64, 0, s>>>(per_xcd, modmatch);
        uint32_t census[17];
        CHK(hipMemcpyAsync(census, scratch, sizeof(census), hipMemcpyDeviceToHost, s));
        CHK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 3; rep++) {
            // launches
            CHK(hipEventRecord(e0, s));
            for (int k = 0; k < K; k++) k_phase<<<G, 256, 0, s>>>(buf, n, k);
            CHK(hipEventRecord(e1, s));
            CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms[0], e0, e1));
            uint32_t *cnt = scratch + 64 * LINE, *xc = scratch + 65 * LINE, *xr = scratch + 80 * LINE, *gc = scratch + 96 * LINE, *rel = scratch + 100 * LINE, *reg = scratch + 110 * LINE, *flags = scratch + 128 * LINE;
            CHK(hipMemsetAsync(scratch + 64 * LINE, 0, (size_t)(4096 - 64) * LINE * 4, s));
            CHK(hipEventRecord(e0, s));
            k_bar_one<<<G, 256, 0, s>>>(buf, n, K, cnt);
            CHK(hipEventRecord(e1, s));
            CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms[1], e0, e1));
            CHK(hipEventRecord(e0, s));
            k_bar_xcd<<<G, 256, 0, s>>>(buf, n, K, xc, xr, gc, reg);
            CHK(hipEventRecord(e1, s));
            CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms[2], e0, e1));
            CHK(hipEventRecord(e0, s));
            k_bar_flags<<<G, 256, 0, s>>>(buf, n, K, flags, rel);
            CHK(hipEventRecord(e1, s));
            CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms[3], e0, e1));
        }
        uint32_t failed = 0;
        CHK(hipMemcpyFromSymbol(&failed, HIP_SYMBOL(g_fail_word), 4));
        if (failed) printf("(a bounded wait gave up for G = %d: the numbers of that variant mean nothing)\n", G);
        printf("G %4d workgroups (census per XCD %u %u %u %u %u %u %u %u; %u of %d on XCD blockIdx %% 8): us per phase: launches %.2f | one counter %.2f | per-XCD L2 + device %.2f | flag gather %.2f\n",
               G, census[0], census[1], census[2], census[3], census[4], census[5], census[6], census[7], census[16], G,
               ms[0] * 1e3 / K, ms[1] * 1e3 / K, ms[2] * 1e3 / K, ms[3] * 1e3 / K);
        fflush(stdout);
    }
    return 0;
}
