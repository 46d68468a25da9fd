// How many workgroups of 256 threads with X KB of dynamic LDS does a CU of this board hold at once?  Every workgroup stamps its start and
// spins for ~40 us; the workgroups that start within the first 10 us are the resident ones.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/occupancy_probe tools/occupancy_probe.hip && tools/exp/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k_occ(unsigned long long *start, int spin_ticks)
{
    extern __shared__ int lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; lds[0] = 1; }
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(32);
}
// the same with a high SGPR number in use (the clobber makes the compiler count it)
#define OCC_S(NAME, REG) __global__ __launch_bounds__(256) void NAME(unsigned long long *start, int spin_ticks) { \
    extern __shared__ int lds[]; const unsigned long long t0 = wall_clock64(); asm volatile("" ::: REG); \
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; lds[0] = 1; } \
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(32); }
OCC_S(k_occ_s79, "s79")
OCC_S(k_occ_s87, "s87")
OCC_S(k_occ_s95, "s95")
OCC_S(k_occ_s101, "s101")
// ... and with many VGPRs
#define OCC_V(NAME, REG) __global__ __launch_bounds__(256) void NAME(unsigned long long *start, int spin_ticks) { \
    extern __shared__ int lds[]; const unsigned long long t0 = wall_clock64(); asm volatile("" ::: REG); \
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; lds[0] = 1; } \
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(32); }
OCC_V(k_occ_v55, "v55")
OCC_V(k_occ_v63, "v63")
OCC_V(k_occ_v71, "v71")
template <typename K>
static void run(const char *name, K kern, unsigned long long *d, int ncu)
{
    const int grid = ncu * 16;
    hipMemset(d, 0, 8 * 16384);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 1024, 0, d, 4000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
    const unsigned long long t0 = *std::min_element(h.begin(), h.end());
    int first = 0;
    for (auto v : h) if (v - t0 < 1000) first++;
    printf("%-28s %5d of %d workgroups started in the first 10 us = %.2f per CU\n", name, first, grid, (double)first / ncu);
}
int main()
{
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    int maxlds = 0; hipDeviceGetAttribute(&maxlds, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
    printf("CUs %d, max LDS per block %d\n", ncu, maxlds);
    unsigned long long *d; hipMalloc(&d, 8 * 16384);
    hipFuncSetAttribute((const void *)k_occ, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int kb : {1, 8, 12, 16, 18, 20, 24, 28, 32, 40, 64}) {
        const int grid = ncu * 16;
        hipMemset(d, 0, 8 * 16384);
        hipLaunchKernelGGL(k_occ, dim3(grid), dim3(256), kb * 1024, 0, d, 4000);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(h.begin(), h.end());
        int first = 0;
        for (auto v : h) if (v - t0 < 1000) first++;
        printf("%2d KB of LDS: %5d of %d workgroups started in the first 10 us = %.2f per CU  (LDS alone would allow %d)\n", kb, first, grid, (double)first / ncu, 160 / kb);
    }
    run("1 KB LDS, s79 in use", k_occ_s79, d, ncu);
    run("1 KB LDS, s87 in use", k_occ_s87, d, ncu);
    run("1 KB LDS, s95 in use", k_occ_s95, d, ncu);
    run("1 KB LDS, s101 in use", k_occ_s101, d, ncu);
    run("1 KB LDS, v55 in use", k_occ_v55, d, ncu);
    run("1 KB LDS, v63 in use", k_occ_v63, d, ncu);
    run("1 KB LDS, v71 in use", k_occ_v71, d, ncu);
    return 0;
}
