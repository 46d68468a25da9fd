// rw_mix_probe.hip -- round 5: why does a 4 B/px read stream with a 1/32 write stream (k_threshold) run in "modes" that depend on the
// ALLOCATION the small write stream lands in?  A stand-alone model of the kernel: every workgroup reads 24 KB with six non-temporal
// float4 loads per lane and writes 768 B of "mask" (8-byte stores from every 16th lane).
// build: hipcc --offload-arch=gfx950 -O3 tools/rw_mix_probe.hip -o /tmp/rw_mix_probe ; run: /tmp/rw_mix_probe [test]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <chrono>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: read + write; 1: read only; 2: write only
__device__ __forceinline__ unsigned xcd_chunk(unsigned b, unsigned n, int on)
{
    if (!on) return b;
    const unsigned k = (unsigned)on, g = 8u * k, grp = b / g;
    if ((grp + 1u) * g > n) return b;
    const unsigned r = b - grp * g;
    return grp * g + (r & 7u) * k + (r >> 3);
}
static int g_tile = 0;
__global__ __launch_bounds__(256) void k_mix(const f4 *__restrict__ in, uint64_t *__restrict__ mask, int mode, int tile)
{
    const int tid = threadIdx.x;
    const unsigned bid = xcd_chunk(blockIdx.x, gridDim.x, tile);
    const f4 *p = in + (int64_t)bid * 1536 + tid;
    uint64_t *m = mask + (int64_t)bid * 96;
    f4 v[6];
    if (mode != 2) {
#pragma unroll
        for (int u = 0; u < 6; u++) v[u] = __builtin_nontemporal_load(p + u * 256);
    } else {
#pragma unroll
        for (int u = 0; u < 6; u++) v[u] = (f4)((float)tid);
    }
#pragma unroll
    for (int u = 0; u < 6; u++) {
        uint32_t nib = (v[u].x >= 160.f ? 1u : 0u) | (v[u].y >= 160.f ? 2u : 0u) | (v[u].z >= 160.f ? 4u : 0u) | (v[u].w >= 160.f ? 8u : 0u);
        uint32_t x = nib << (4 * (tid & 7));
        x |= __shfl_xor(x, 1); x |= __shfl_xor(x, 2); x |= __shfl_xor(x, 4);
        const uint32_t hi = __shfl_down(x, 8);
        if ((tid & 15) == 0 && mode != 1) m[u * 16 + (tid >> 4)] = ((uint64_t)hi << 32) | x;
        else if (mode == 1 && x == 0x12345678u) m[0] = x;
    }
}

static float time_it(const f4 *in, uint64_t *mask, int64_t nwg, int mode, int reps = 5)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    k_mix<<<(unsigned)nwg, 256>>>(in, mask, mode, g_tile);
    std::vector<float> t;
    for (int r = 0; r < reps; r++) {
        CHK(hipEventRecord(e0));
        k_mix<<<(unsigned)nwg, 256>>>(in, mask, mode, g_tile);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float f; CHK(hipEventElapsedTime(&f, e0, e1)); t.push_back(f);
    }
    std::sort(t.begin(), t.end());
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

int main(int argc, char **argv)
{
    const int64_t T = 2707, ny = 181, nx = 360;
    const int64_t bytes = T * ny * nx * 4, nwg = bytes / (1536 * 16);          // 28 708 workgroups of 24 KB
    const size_t mbytes = (size_t)nwg * 96 * 8;
    const char *test = argc > 1 ? argv[1] : "allocs";
    float *slab = nullptr, *out = nullptr;
    const size_t extra = (size_t)64 << 20;
    CHK(hipMalloc(&slab, bytes + extra));
    CHK(hipMalloc(&out, bytes));                                              // (the pass' output slab: same allocation history as the library)
    CHK(hipMemset(slab, 0, bytes + extra));
    printf("slab %p (%.1f MB), mask %.1f MB, %lld workgroups\n", (void *)slab, bytes / 1e6, mbytes / 1e6, (long long)nwg);
    {
        uint64_t *m0; CHK(hipMalloc(&m0, mbytes));
        printf("read only  %.4f ms   write only %.4f ms\n", time_it((const f4 *)slab, m0, nwg, 1), time_it((const f4 *)slab, m0, nwg, 2));
        CHK(hipFree(m0));
    }
    if (!strcmp(test, "tiles")) {
        std::vector<uint64_t *> keep;
        for (int k = 0; k < 12; k++) {
            uint64_t *m; CHK(hipMalloc(&m, mbytes)); keep.push_back(m);
            printf("T alloc %2d  mask %p ", k, (void *)m);
            for (int tile : {0, 4, 16, 64, 256}) { g_tile = tile; printf(" tile %3d: %.4f", tile, time_it((const f4 *)slab, m, nwg, 0)); }
            printf("\n");
        }
        for (int tile : {0, 4, 16, 64, 256}) { g_tile = tile; printf("read only, tile %3d: %.4f ms\n", tile, time_it((const f4 *)slab, keep[0], nwg, 1)); }
        g_tile = 0;
    }
    if (!strcmp(test, "map")) {
        // many allocations, revisited: is the mode a property of the allocation (stable over rounds) and where do the fast ones lie?
        const int N = argc > 2 ? atoi(argv[2]) : 80;
        std::vector<uint64_t *> keep;
        std::vector<std::vector<float>> ms(N);
        for (int k = 0; k < N; k++) { uint64_t *m; CHK(hipMalloc(&m, mbytes)); keep.push_back(m); }
        for (int round = 0; round < 3; round++)
            for (int k = 0; k < N; k++) ms[k].push_back(time_it((const f4 *)slab, keep[k], nwg, 0, 3));
        for (int k = 0; k < N; k++) printf("M alloc %3d  mask %p  %.4f %.4f %.4f  %s\n", k, (void *)keep[k], ms[k][0], ms[k][1], ms[k][2], ms[k][1] < 0.111 ? "FAST" : "");
    }
    if (!strcmp(test, "imap")) {
        // the same, but every allocation is timed right after it was made (round 0), then all of them are revisited twice
        const int N = argc > 2 ? atoi(argv[2]) : 80;
        std::vector<uint64_t *> keep;
        std::vector<std::vector<float>> ms(N);
        for (int k = 0; k < N; k++) { uint64_t *m; CHK(hipMalloc(&m, mbytes)); keep.push_back(m); ms[k].push_back(time_it((const f4 *)slab, m, nwg, 0, 3)); }
        for (int round = 0; round < 2; round++)
            for (int k = 0; k < N; k++) ms[k].push_back(time_it((const f4 *)slab, keep[k], nwg, 0, 3));
        for (int k = 0; k < N; k++) printf("I alloc %3d  mask %p  %.4f %.4f %.4f  %s\n", k, (void *)keep[k], ms[k][0], ms[k][1], ms[k][2], ms[k][1] < 0.111 ? "FAST" : "");
    }
    if (!strcmp(test, "rel")) {
        // is the mode a property of the mask's region alone, or of its relation to where the slab lies?  One 8 GB allocation, the 705 MB
        // window read at sixteen offsets; twelve masks (allocated and classified first against the original slab)
        char *big; CHK(hipMalloc(&big, (size_t)8 << 30)); CHK(hipMemset(big, 0, (size_t)8 << 30));
        const int N = 12;
        std::vector<uint64_t *> keep;
        for (int k = 0; k < N; k++) { uint64_t *m; CHK(hipMalloc(&m, mbytes)); keep.push_back(m); }
        printf("big %p\n%-22s", (void *)big, "slab window at");
        for (int k = 0; k < N; k++) printf(" m%-2d   ", k);
        printf("\n%-22s", "original slab");
        for (int k = 0; k < N; k++) printf(" %.4f", time_it((const f4 *)slab, keep[k], nwg, 0, 3));
        printf("\n");
        for (int o = 0; o < 28; o++) {
            const size_t off = (size_t)o * ((size_t)256 << 20);
            if (off + bytes > ((size_t)8 << 30)) break;
            char lab[64]; snprintf(lab, sizeof lab, "big + %5zu MB", off >> 20);
            printf("%-22s", lab);
            for (int k = 0; k < N; k++) printf(" %.4f", time_it((const f4 *)(big + off), keep[k], nwg, 0, 3));
            printf("\n");
        }
        // and a mask INSIDE the big allocation, 64 positions 128 MB apart, against the original slab
        printf("mask inside big, every 128 MB, original slab:\n");
        for (int o = 0; o < 64; o++) {
            uint64_t *m = (uint64_t *)(big + (size_t)o * ((size_t)128 << 20));
            printf(" %.4f%s", time_it((const f4 *)slab, m, nwg, 0, 3), (o & 15) == 15 ? "\n" : "");
        }
    }
    if (!strcmp(test, "arena")) {
        // the structure of slow and fast regions along one (presumably physically contiguous) 6 GB allocation: the mask every 48 MB;
        // then the same after the arena has been memset
        const size_t A = (size_t)6 << 30, step = (size_t)48 << 20;
        char *arena; CHK(hipMalloc(&arena, A));
        printf("arena %p\n", (void *)arena);
        for (int pass = 0; pass < 3; pass++) {
            if (pass == 2) { CHK(hipMemset(arena, 0, A)); CHK(hipDeviceSynchronize()); printf("after memset:\n"); }
            int n = 0;
            for (size_t off = 0; off + mbytes <= A; off += step, n++) {
                const float f = time_it((const f4 *)slab, (uint64_t *)(arena + off), nwg, 0, 3);
                printf("%c", f < 0.1105 ? 'F' : (f < 0.1135 ? 'm' : 's'));
                if ((n & 63) == 63) printf("\n");
            }
            printf("\n");
        }
    }
    if (!strcmp(test, "matrix")) {
        // regions: 1 GB pieces of up to four 6 GB arenas, classified by a mask at their start against the original slab; then
        // slab window x mask over a selection of slow and fast regions (the slab window needs 705 MB: one piece)
        std::vector<char *> pieces; std::vector<int> cls;
        for (int a = 0; a < 4; a++) {
            char *arena; if (hipMalloc(&arena, (size_t)6 << 30) != hipSuccess) break;
            for (int p = 0; p < 6; p++) {
                char *q = arena + ((size_t)p << 30);
                const float f = time_it((const f4 *)slab, (uint64_t *)q, nwg, 0, 3);
                pieces.push_back(q); cls.push_back(f < 0.111 ? 1 : 0);
                printf("arena %d piece %d  %p  %.4f %s\n", a, p, (void *)q, f, f < 0.111 ? "FAST" : "slow");
            }
        }
        std::vector<int> sel;
        for (int want = 0; want < 2; want++) { int n = 0; for (size_t i = 0; i < pieces.size() && n < 4; i++) if (cls[i] == want) { sel.push_back((int)i); n++; } }
        printf("rows: slab window in piece r (s/F = its class as a mask region); columns: mask at piece c + 800 MB\n        ");
        for (int c : sel) printf("  %c%-5d", cls[c] ? 'F' : 's', c);
        printf("\n%-8s", "orig");
        for (int c : sel) printf(" %.4f", time_it((const f4 *)slab, (uint64_t *)(pieces[c] + ((size_t)800 << 20)), nwg, 0, 3));
        printf("\n");
        for (int r : sel) {
            printf("%c%-7d", cls[r] ? 'F' : 's', r);
            for (int c : sel) printf(" %.4f", time_it((const f4 *)pieces[r], (uint64_t *)(pieces[c] + ((size_t)800 << 20)), nwg, 0, 3));
            printf("   read-only %.4f\n", time_it((const f4 *)pieces[r], (uint64_t *)(pieces[r] + ((size_t)800 << 20)), nwg, 1, 3));
        }
    }
    if (!strcmp(test, "spacers")) {
        // how far does one have to go to find memory of the other class?  candidates with spacers of growing size held in between
        // (sizes in MB from argv), every hipMalloc timed
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const float ro = time_it((const f4 *)slab, (uint64_t *)out, nwg, 1, 5);
        printf("read only %.4f\n", ro);
        size_t held = 0;
        for (int i = 2; i <= argc; i++) {
            uint64_t *m; double t0 = now(); CHK(hipMalloc(&m, mbytes)); double t1 = now();
            const float f = time_it((const f4 *)slab, m, nwg, 0, 3);
            printf("after %6zu MB of spacers: candidate %p  %.4f (x%.3f) %s   [hipMalloc 24 MB: %.2f ms]\n", held >> 20, (void *)m, f, f / ro, f < ro * 1.085 ? "FAST" : "slow", t1 - t0);
            if (i < argc) {
                const size_t sp = (size_t)atoll(argv[i]) << 20; void *q; t0 = now(); CHK(hipMalloc(&q, sp)); t1 = now();
                printf("   spacer %zu MB: hipMalloc %.2f ms\n", sp >> 20, t1 - t0);
                held += sp;
            }
        }
    }
    if (!strcmp(test, "bigmap")) {
        // the large-scale structure: ONE allocation of argv[2] GB (default 48), the mask every 256 MB against the original slab; a
        // physically contiguous block would show the address pattern of the classes, if there is one
        const size_t gb = argc > 2 ? (size_t)atoll(argv[2]) : 48;
        const size_t A = gb << 30, step = (size_t)(argc > 3 ? atoll(argv[3]) : 256) << 20;
        char *arena; CHK(hipMalloc(&arena, A));
        printf("arena %p, %zu GB, one letter per %zu MB (F fast / m / s slow), 32 per line\n", (void *)arena, gb, step >> 20);
        int n = 0;
        for (size_t off = 0; off + mbytes <= A; off += step, n++) {
            const float f = time_it((const f4 *)slab, (uint64_t *)(arena + off), nwg, 0, 3);
            printf("%c", f < 0.1105 ? 'F' : (f < 0.1135 ? 'm' : 's'));
            if ((n & 31) == 31) printf("\n");
        }
        printf("\n");
    }
    if (!strcmp(test, "allocs")) {
        // (A) separate allocations, kept alive
        std::vector<uint64_t *> keep;
        for (int k = 0; k < 24; k++) {
            uint64_t *m; CHK(hipMalloc(&m, mbytes)); keep.push_back(m);
            printf("A alloc %2d  mask %p  mix %.4f ms  write-only %.4f\n", k, (void *)m, time_it((const f4 *)slab, m, nwg, 0), time_it((const f4 *)slab, m, nwg, 2));
        }
        // (A2) again, every one of them: is the mode stable?
        for (int k = 0; k < 24; k += 3) printf("A2 alloc %2d  mix %.4f ms\n", k, time_it((const f4 *)slab, keep[k], nwg, 0));
        for (auto m : keep) CHK(hipFree(m));
        // (B) alloc / free / alloc: the allocator's two regions in turn
        for (int k = 0; k < 8; k++) {
            uint64_t *m; CHK(hipMalloc(&m, mbytes));
            printf("B realloc %d  mask %p  mix %.4f ms\n", k, (void *)m, time_it((const f4 *)slab, m, nwg, 0));
            CHK(hipFree(m));
        }
        // (C) the mask inside the slab's own allocation (behind the slab)
        for (int k = 0; k < 4; k++) {
            uint64_t *m = (uint64_t *)((char *)slab + bytes + ((size_t)k << 20) * 8);
            printf("C inside slab alloc +%d MB  mask %p  mix %.4f ms\n", k * 8, (void *)m, time_it((const f4 *)slab, m, nwg, 0));
        }
        // (D) one 1 GB arena, the mask at sixteen 64 MB steps
        char *arena; CHK(hipMalloc(&arena, (size_t)1 << 30));
        for (int k = 0; k < 16; k++) {
            uint64_t *m = (uint64_t *)(arena + ((size_t)k << 26));
            printf("D arena +%4d MB  mask %p  mix %.4f ms\n", k * 64, (void *)m, time_it((const f4 *)slab, m, nwg, 0));
        }
        CHK(hipFree(arena));
        // (E) sizes: does the size class of the allocation matter?
        for (size_t mb : {24, 32, 48, 64, 128, 256, 24, 32}) {
            uint64_t *m; CHK(hipMalloc(&m, mb << 20));
            printf("E size %3zu MB  mask %p  mix %.4f ms\n", mb, (void *)m, time_it((const f4 *)slab, m, nwg, 0));
            keep.push_back(m);
        }
    }
    return 0;
}
