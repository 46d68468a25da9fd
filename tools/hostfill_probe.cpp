// Probe: how fast can N host threads write a 705.6 MB int32 result (zeros + sparse runs) into
// (a) fresh pageable memory, (b) the same memory again (touched)?  Compared with a D2H copy at 47-57 GB/s.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <immintrin.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void fill(int* p, size_t n, int mode) {
    if (mode == 0) { memset(p, 0, n * 4); return; }
    // non-temporal zero stores
    __m128i z = _mm_setzero_si128();
    size_t i = 0;
    for (; i + 4 <= n; i += 4) _mm_stream_si128((__m128i*)(p + i), z);
    for (; i < n; ++i) p[i] = 0;
    _mm_sfence();
}
int main(int argc, char** argv) {
    size_t n = (size_t)2707 * 181 * 360;
    if (argc > 1) n = strtoull(argv[1], 0, 10);
    printf("hw threads %u, elements %zu (%.1f MB)\n", std::thread::hardware_concurrency(), n, n * 4 / 1e6);
    for (int mode = 0; mode < 2; ++mode)
    for (int nt : {4, 8, 16, 32, 64}) {
        for (int fresh = 1; fresh >= 0; --fresh) {
            double best = 1e9, first = 0;
            int* p = nullptr;
            for (int rep = 0; rep < 4; ++rep) {
                if (fresh || !p) {
                    if (p) munmap(p, n * 4);
                    p = (int*)mmap(nullptr, n * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                    if (!fresh) memset(p, 1, n * 4);
                }
                double t0 = now();
                std::vector<std::thread> th;
                size_t per = (n / nt + 1023) & ~(size_t)1023;
                for (int k = 0; k < nt; ++k) {
                    size_t a = (size_t)k * per, b = a + per > n ? n : a + per;
                    if (a >= n) break;
                    th.emplace_back(fill, p + a, b - a, mode);
                }
                for (auto& t : th) t.join();
                double dt = now() - t0;
                if (rep == 0) first = dt;
                if (dt < best) best = dt;
            }
            munmap(p, n * 4);
            printf("mode %s threads %2d %s: best %.2f ms (%.1f GB/s), first %.2f ms\n", mode ? "nt-store" : "memset", nt,
                   fresh ? "fresh  " : "touched", best * 1e3, n * 4 / best / 1e9, first * 1e3);
        }
    }
    return 0;
}
