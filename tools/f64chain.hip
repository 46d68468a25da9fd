// Latency of a chain of dependent float64 adds in ONE lane of a wave on gfx950 (what bounds k_life_exact's sequential sums).
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/f64chain tools/f64chain.hip && tools/exp/f64chain
// variants: values in registers | from LDS as k_life_exact reads them (two per load, sixteen ahead) | one / three / sixteen waves busy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k_chain_reg(double *out, unsigned long long *ticks, int iters, int nwaves, double x0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = x0 * (i + 1);
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (lane == 0 && wave < nwaves) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc += x[i];
            asm volatile("" : "+v"(acc));
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0 && wave < nwaves) { out[wave] = acc; ticks[wave] = t1 - t0; }
}
// all 64 lanes active (does the exec mask matter?)
__global__ __launch_bounds__(1024) void k_chain_reg_all(double *out, unsigned long long *ticks, int iters, int nwaves, double x0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = lane;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = x0 * (i + 1);
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (wave < nwaves) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc += x[i];
            asm volatile("" : "+v"(acc));
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (wave < nwaves) { out[threadIdx.x] = acc; if (lane == 0) ticks[wave] = t1 - t0; }
}
__global__ __launch_bounds__(1024) void k_chain_lds(double *out, unsigned long long *ticks, int iters, int nwaves, double x0)
{
    __shared__ __attribute__((aligned(16))) double stage[3][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3 * 1024; i += blockDim.x) (&stage[0][0])[i] = x0 * (i & 15);
    __syncthreads();
    double acc = 0.0;
    const unsigned long long t0 = wall_clock64();
    if (lane == 0 && wave < nwaves) {
        const double *sv = stage[wave % 3];
        for (int it = 0; it < iters; it++) {
            const int cn = 1024;
            int i;
#define LX_LD(o) (*reinterpret_cast<const d2 *>(sv + (o)))
#define LX_ADD(q0, q1, q2, q3) do { acc += q0.x; acc += q0.y; acc += q1.x; acc += q1.y; acc += q2.x; acc += q2.y; acc += q3.x; acc += q3.y; } while (0)
            d2 a0 = LX_LD(0), a1 = LX_LD(2), a2 = LX_LD(4), a3 = LX_LD(6);
            for (i = 8; i + 16 <= cn; i += 16) {
                const d2 b0 = LX_LD(i), b1 = LX_LD(i + 2), b2 = LX_LD(i + 4), b3 = LX_LD(i + 6);
                asm volatile("" : "+v"(acc) : : "memory");
                LX_ADD(a0, a1, a2, a3);
                a0 = LX_LD(i + 8); a1 = LX_LD(i + 10); a2 = LX_LD(i + 12); a3 = LX_LD(i + 14);
                asm volatile("" : "+v"(acc) : : "memory");
                LX_ADD(b0, b1, b2, b3);
            }
            LX_ADD(a0, a1, a2, a3);
            for (i += 8; i < cn; ++i) acc += sv[i];
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0 && wave < nwaves) { out[wave] = acc; ticks[wave] = t1 - t0; }
}
// the three chains of k_life_exact in the lanes of ONE wave: lane l walks list l % 3 (per-lane LDS address; lists padded apart by 16 bytes)
template <int AHEAD>
__global__ __launch_bounds__(1024) void k_chain_lds_lanes(double *out, unsigned long long *ticks, int iters, int lanes_on, double x0)
{
    __shared__ __attribute__((aligned(16))) double stage[3][1024 + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3 * 1026; i += blockDim.x) (&stage[0][0])[i] = x0 * (i & 15);
    __syncthreads();
    double acc = 0.0;
    const unsigned long long t0 = wall_clock64();
    if (wave == 0 && lane < lanes_on) {
        const double *sv = stage[lane % 3];
        for (int it = 0; it < iters; it++) {
            const int cn = 1024;
            int i;
#define LX_LD(o) (*reinterpret_cast<const d2 *>(sv + (o)))
#define LX_ADD(q0, q1, q2, q3) do { acc += q0.x; acc += q0.y; acc += q1.x; acc += q1.y; acc += q2.x; acc += q2.y; acc += q3.x; acc += q3.y; } while (0)
            if (AHEAD == 16) {
                d2 a0 = LX_LD(0), a1 = LX_LD(2), a2 = LX_LD(4), a3 = LX_LD(6);
                for (i = 8; i + 16 <= cn; i += 16) {
                    const d2 b0 = LX_LD(i), b1 = LX_LD(i + 2), b2 = LX_LD(i + 4), b3 = LX_LD(i + 6);
                    asm volatile("" : "+v"(acc) : : "memory");
                    LX_ADD(a0, a1, a2, a3);
                    a0 = LX_LD(i + 8); a1 = LX_LD(i + 10); a2 = LX_LD(i + 12); a3 = LX_LD(i + 14);
                    asm volatile("" : "+v"(acc) : : "memory");
                    LX_ADD(b0, b1, b2, b3);
                }
                LX_ADD(a0, a1, a2, a3);
                for (i += 8; i < cn; ++i) acc += sv[i];
            } else if (AHEAD == 0) {
                // the whole block unrolled into one basic block: no load is carried around a loop edge (hipcc waits for ALL outstanding LDS
                // loads -- lgkmcnt(0) -- in front of the first use of a value loaded in the previous iteration, i.e. also for the ones just issued)
                d2 a0 = LX_LD(0), a1 = LX_LD(2), a2 = LX_LD(4), a3 = LX_LD(6), a4 = LX_LD(8), a5 = LX_LD(10), a6 = LX_LD(12), a7 = LX_LD(14);
#pragma unroll
                for (int j = 16; j < 1024; j += 16) {
                    const d2 b0 = LX_LD(j), b1 = LX_LD(j + 2), b2 = LX_LD(j + 4), b3 = LX_LD(j + 6), b4 = LX_LD(j + 8), b5 = LX_LD(j + 10), b6 = LX_LD(j + 12), b7 = LX_LD(j + 14);
                    asm volatile("" : "+v"(acc) : : "memory");
                    LX_ADD(a0, a1, a2, a3); LX_ADD(a4, a5, a6, a7);
                    a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
                }
                LX_ADD(a0, a1, a2, a3); LX_ADD(a4, a5, a6, a7);
            } else {
                // 32 ahead: two sets of eight d2
                d2 a0 = LX_LD(0), a1 = LX_LD(2), a2 = LX_LD(4), a3 = LX_LD(6), a4 = LX_LD(8), a5 = LX_LD(10), a6 = LX_LD(12), a7 = LX_LD(14);
                for (i = 16; i + 32 <= cn; i += 32) {
                    const d2 b0 = LX_LD(i), b1 = LX_LD(i + 2), b2 = LX_LD(i + 4), b3 = LX_LD(i + 6), b4 = LX_LD(i + 8), b5 = LX_LD(i + 10), b6 = LX_LD(i + 12), b7 = LX_LD(i + 14);
                    asm volatile("" : "+v"(acc) : : "memory");
                    LX_ADD(a0, a1, a2, a3); LX_ADD(a4, a5, a6, a7);
                    a0 = LX_LD(i + 16); a1 = LX_LD(i + 18); a2 = LX_LD(i + 20); a3 = LX_LD(i + 22); a4 = LX_LD(i + 24); a5 = LX_LD(i + 26); a6 = LX_LD(i + 28); a7 = LX_LD(i + 30);
                    asm volatile("" : "+v"(acc) : : "memory");
                    LX_ADD(b0, b1, b2, b3); LX_ADD(b4, b5, b6, b7);
                }
                LX_ADD(a0, a1, a2, a3); LX_ADD(a4, a5, a6, a7);
                for (i += 16; i < cn; ++i) acc += sv[i];
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (wave == 0) { out[lane] = acc; if (lane == 0) ticks[0] = t1 - t0; }
}
int main()
{
    double *out; unsigned long long *ticks;
    hipMalloc(&out, 1024 * 8); hipMalloc(&ticks, 16 * 8);
    std::vector<unsigned long long> h(16);
    for (int rep = 0; rep < 2; rep++)
        for (int nw : {1, 3, 4, 16}) {
            const int iters = 20000;
            hipLaunchKernelGGL(k_chain_reg, dim3(1), dim3(1024), 0, 0, out, ticks, iters, nw, 1.0000001);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), ticks, 16 * 8, hipMemcpyDeviceToHost);
            printf("registers, lane 0 only, %2d waves: %.2f ns per dependent add (wave 0), %.2f (last wave)\n", nw, h[0] * 10.0 / (iters * 16.0), h[nw - 1] * 10.0 / (iters * 16.0));
            hipLaunchKernelGGL(k_chain_reg_all, dim3(1), dim3(1024), 0, 0, out, ticks, iters, nw, 1.0000001);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), ticks, 16 * 8, hipMemcpyDeviceToHost);
            printf("registers, all 64 lanes,  %2d waves: %.2f ns per dependent add (wave 0), %.2f (last wave)\n", nw, h[0] * 10.0 / (iters * 16.0), h[nw - 1] * 10.0 / (iters * 16.0));
            const int it2 = 400;
            hipLaunchKernelGGL(k_chain_lds, dim3(1), dim3(1024), 0, 0, out, ticks, it2, nw, 1.0000001);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), ticks, 16 * 8, hipMemcpyDeviceToHost);
            printf("LDS as k_life_exact,      %2d waves: %.2f ns per dependent add (wave 0), %.2f (last wave)\n", nw, h[0] * 10.0 / (it2 * 1024.0), h[nw - 1] * 10.0 / (it2 * 1024.0));
        }
    for (int lanes_on : {3, 64}) {
        const int it2 = 400;
        hipLaunchKernelGGL(k_chain_lds_lanes<16>, dim3(1), dim3(1024), 0, 0, out, ticks, it2, lanes_on, 1.0000001);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), ticks, 16 * 8, hipMemcpyDeviceToHost);
        printf("LDS, three lists in the lanes of one wave (%2d lanes on), 16 ahead: %.2f ns per step of the three chains\n", lanes_on, h[0] * 10.0 / (it2 * 1024.0));
        hipLaunchKernelGGL(k_chain_lds_lanes<0>, dim3(1), dim3(1024), 0, 0, out, ticks, it2, lanes_on, 1.0000001);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), ticks, 16 * 8, hipMemcpyDeviceToHost);
        printf("LDS, three lists in the lanes of one wave (%2d lanes on), block unrolled: %.2f ns per step of the three chains\n", lanes_on, h[0] * 10.0 / (it2 * 1024.0));
        hipLaunchKernelGGL(k_chain_lds_lanes<32>, dim3(1), dim3(1024), 0, 0, out, ticks, it2, lanes_on, 1.0000001);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), ticks, 16 * 8, hipMemcpyDeviceToHost);
        printf("LDS, three lists in the lanes of one wave (%2d lanes on), 32 ahead: %.2f ns per step of the three chains\n", lanes_on, h[0] * 10.0 / (it2 * 1024.0));
    }
    return 0;
}
