#include <utility>
// ctk_kernels.hip -- gfx950 (MI355X, wave64) kernels of the run_contrack hot path.
//
// Data layout in HBM (one shard = T consecutive timesteps of a (ny, nx) grid, W = ceil(nx/64)):
//   anom      float32 [T][ny][nx]        input slab, read ONCE (k_threshold)
//   mask      uint64  [T][ny][W]         1 bit per pixel (bit x&63 of word x>>6), 1/32 of the slab
//   rowcnt    uint16  [T][ny]            foreground runs per row
//   rowstart  uint32  [T][ny]            first run of the row, relative to the timestep
//   run_base  uint32  [T+1]              first run of the timestep (exclusive scan of runs/timestep)
//   run_comp  uint32  [runs]             id of the no-wrap 2-D component of each run (raster order ids)
//   comp_*            [components]       per component: merged-representative, bbox, exact area limbs
//   pairs / seams                        co-occurrence records / seam rows (ctk_tables.h)
//   run_val   int32   [runs]             final flag value of each run (after resolve + persistence)
//   flag      int32   [T][ny][nx]        output, written ONCE (k_relabel)
// A run is a maximal horizontal segment of foreground pixels; all labelling works on runs (about 1 % of
// the pixel count on Z500 anomaly fields), the only per-pixel passes are the first and the last kernel.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "ctk_tables.h"
#include "ctk_device.h"

#define WAVE 64
#define FULL64 0xffffffffffffffffull

// ------------------------------------------------------------------------------------------------
// small wave / block helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }

// "a background value was written": one of CTK_ZF_SLOTS words per workgroup (ctk_device.h), plain stores
__device__ __forceinline__ void ctk_zf_set(uint32_t *counters, uint32_t key) { counters[CTK_ZF_OFF + (key & (CTK_ZF_SLOTS - 1)) * CTK_ZF_STRIDE] = 1u; }
__device__ __forceinline__ void ctk_zf_reset(uint32_t *counters, int64_t i) { if (i < CTK_ZF_SLOTS) counters[CTK_ZF_OFF + i * CTK_ZF_STRIDE] = 0u; }
// any slot set?  called by all threads of a workgroup of `nthreads`; the result is valid for thread 0 .. 63 after a wave OR
__device__ __forceinline__ uint32_t ctk_zf_mine(const uint32_t *counters, int tid, int nthreads)
{
    uint32_t v = 0;
    for (int i = tid; i < CTK_ZF_SLOTS; i += nthreads) v |= __hip_atomic_load(&counters[CTK_ZF_OFF + i * CTK_ZF_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    int lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d)
{
    uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_down_u64(uint64_t v, int d)
{
    uint32_t lo = __shfl_down((uint32_t)v, d), hi = __shfl_down((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// Block-wide exclusive scan of one uint32 per thread.  `sm` holds (blockDim.x/64 + 1) words.
// Returns the exclusive prefix; *total receives the block sum.  Contains two __syncthreads().
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *sm, uint32_t *total)
{
    int lane = lane_id(), wv = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    uint32_t inc = wave_incl_scan_u32(v);
    __syncthreads();                       // protect sm against the previous use
    if (lane == WAVE - 1) sm[wv] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
        uint32_t s = sm[i];
        if (i < wv) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// lock-free union-find with "smaller index wins" hooking (root = smallest index of the set)
// find with intermediate pointer jumping: every visited node is re-pointed to its grandparent.  Parents only
// ever decrease and always stay inside the set, so the racing writes of other lanes are harmless.
__device__ __forceinline__ uint32_t uf_find(uint32_t *p, uint32_t i)
{
    uint32_t cur = __hip_atomic_load(&p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (cur == i) return i;
    uint32_t prev = i;
    for (;;) {
        const uint32_t next = __hip_atomic_load(&p[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (next == cur) return cur;
        __hip_atomic_store(&p[prev], next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prev = cur;
        cur = next;
    }
}
__device__ __forceinline__ void uf_unite(uint32_t *p, uint32_t a, uint32_t b)
{
    for (;;) {
        a = uf_find(p, a);
        b = uf_find(p, b);
        if (a == b) return;
        if (a < b) { uint32_t s = a; a = b; b = s; }          // a > b: hook a below b
        uint32_t old = atomicMin(&p[a], b);
        if (old == a) return;                                  // a was still a root
        a = old;                                               // somebody re-hooked a: unite that with b
    }
}

// ------------------------------------------------------------------------------------------------
// K0  threshold -> bit mask                          (contrack/contrack.py:646-674, NaN -> 0)
// The only kernel that reads the float slab: pure streaming, 4 B/pixel in, 1/8 B/pixel out.
// thr[t] is the threshold the host derived so that the compare in the slab's dtype equals the
// reference's compare (ctk_api: adjust_threshold).
//
// k_threshold_v4 (float32, nx % 4 == 0, 16-byte aligned slab): one workgroup per (timestep, 16 rows);
// every lane issues 8 independent non-temporal float4 loads, the 4 compare bits of a lane are ORed across
// its 16-lane group (64 pixels = one mask word) with 4 cross-lane steps, lane 0 of the group stores the word.
// k_threshold (generic: any nx, float32 / float64): lane l tests pixel 64k+l, __ballot packs a word.
// ------------------------------------------------------------------------------------------------
template <int OP, typename TIN>
__device__ __forceinline__ bool cmp_op(TIN v, TIN th)
{
    if (OP == 0) return v >= th;
    if (OP == 1) return v <= th;
    if (OP == 2) return v > th;
    return v < th;
}

// OR of v across the 16 lanes of a DPP row, left in every lane: quad swaps, then half-row and row mirrors
// (full-rate VALU with DPP operands; no LDS crossbar traffic, unlike __shfl_xor = ds_bpermute)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_or(uint32_t v)
{
    return v | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t row16_or(uint32_t v)
{
    v = dpp_or<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_or<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_or<0x141>(v);     // row_half_mirror
    v = dpp_or<0x140>(v);     // row_mirror
    return v;
}

// Workgroup b of a launch lands on XCD b % 8 (round-robin dispatch).  With `on`, the chunk a workgroup takes is remapped so that every
// XCD streams ONE contiguous eighth of the slab instead of every eighth chunk (experiment, CTK_XCD_REMAP).
__device__ __forceinline__ unsigned xcd_chunk(unsigned b, unsigned n, int on)
{
    if (!on) return b;
    if (on == 1) {                                         // one contiguous eighth of the launch per XCD
        const unsigned per = n >> 3;
        if (b >= (per << 3)) return b;                     // the n % 8 last chunks stay where they are
        return (b & 7u) * per + (b >> 3);
    }
    // tiles of `on` consecutive chunks per XCD: 8 * on chunks form a group inside which XCD x takes chunks [x * on, (x + 1) * on)
    const unsigned k = (unsigned)on, g = 8u * k, grp = b / g;
    if ((grp + 1u) * g > n) return b;                      // the incomplete last group stays where it is
    const unsigned r = b - grp * g;                        // position in the group: XCD r & 7, its (r >> 3)-th chunk
    return grp * g + (r & 7u) * k + (r >> 3);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// SGPRs cost occupancy on this board (round 6, tools/occupancy_probe.hip: workgroups of 256 threads per CU with 16 SGPRs in use 8, with 86 or
// 94: 7, with 102 or 108: 6 -- whatever the guides say about a fixed SGPR allotment per wave).  Kernels that take a struct of a dozen pointers by
// value and hold it in SGPRs end at 100-106: six waves per SIMD, i.e. six instead of eight 256-thread workgroups per CU, ONE instead of two
// 1024-thread workgroups.  With the limit below the compiler keeps what does not fit in lanes of a VGPR (v_writelane / v_readlane), which
// costs nothing measurable: k_relabel_v5 -3 ... -5 % in the same process (tools/relabel_variants.py), 18.0 -> 15.2 ms at 438 000 x 192 x 288;
// k_rs_pass_blk there 2.5 -> 1.9 ms (its second build, k_rs_pass_blk_2pc; short launches keep their SGPRs: ctk_resolve_dev.hip).
#define CTK_SGPR_8WAVES __attribute__((amdgpu_num_sgpr(80)))
#define CTK_RB 16                  // rows per workgroup in the two streaming kernels

template <int OP, int U = 8 /* independent 16-byte loads in flight per lane (8 vs 4: 2.5 % at 1 deg, equal at 0.25 deg) */>
__global__ __launch_bounds__(256) void k_threshold_v4(const float *__restrict__ anom, const float *__restrict__ thr32,
                                                      int ny, int nx, int W, uint64_t *__restrict__ mask, int rb,
                                                      uint32_t *__restrict__ zero_counters /* the pass' device counters start at zero (or nullptr) */)
{
    if (zero_counters && blockIdx.x == 0 && threadIdx.x < CTK_CNT_ZEROED) zero_counters[threadIdx.x] = 0u;
    const int nchunk = (ny + rb - 1) / rb;
    const int t = (int)(blockIdx.x / (unsigned)nchunk), y0 = (int)(blockIdx.x - (unsigned)t * nchunk) * rb, tid = (int)threadIdx.x;
    const int rows = min(rb, ny - y0);
    const float th = thr32[t];
    const int n4 = nx >> 2, n4p = (n4 + 15) & ~15;         // float4 slots per row, padded to whole 16-lane groups (= words)
    const int total = rows * n4p;
    const int64_t row0 = (int64_t)t * ny + y0;
    const float *base = anom + row0 * (int64_t)nx;
    const int sub = tid & 15;
    // (row, slot) of the lane's next load, advanced by 256 slots at a time without dividing
    int nr = tid / n4p, nc = tid - nr * n4p;
    const int dr = 256 / n4p, dc = 256 - dr * n4p;
    for (int i0 = 0; i0 < total; i0 += 256 * U) {
        f32x4 v[U];
        int rr[U], cc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * 256 + tid;
            const int r = nr, c = nc;
            rr[u] = r; cc[u] = c;
            if (i < total && c < n4) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (int64_t)r * nx) + c);
            else v[u] = (f32x4)(__builtin_nanf(""));
            nr += dr; nc += dc;
            if (nc >= n4p) { nc -= n4p; nr++; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t nib = (cmp_op<OP, float>(v[u].x, th) ? 1u : 0u) | (cmp_op<OP, float>(v[u].y, th) ? 2u : 0u) |
                                 (cmp_op<OP, float>(v[u].z, th) ? 4u : 0u) | (cmp_op<OP, float>(v[u].w, th) ? 8u : 0u);
            // lanes 16g .. 16g+15 hold the 64 pixels of one word: place the nibble, OR across the group
            uint32_t lo = (sub < 8) ? (nib << (4 * sub)) : 0u, hi = (sub >= 8) ? (nib << (4 * (sub - 8))) : 0u;
            lo = row16_or(lo);
            hi = row16_or(hi);
            if (sub == 0 && i0 + u * 256 + tid < total) mask[(row0 + rr[u]) * W + (cc[u] >> 4)] = ((uint64_t)hi << 32) | lo;
        }
    }
}

// k_threshold_v7: k_threshold_v4's decomposition (one workgroup per (timestep, rb rows), 16-byte non-temporal loads, a 16-lane
// DPP row = 64 pixels = one mask word) with a third of its VALU work.  SQ counters (profiles/r04_sq_1deg.md) showed v4 to be
// VALU-bound, not memory-bound: 446 VALU instructions per wave = 85 % of every SIMD's issue slots for the 125 us the kernel ran --
// 58 per float4 load, most of them 64-bit address arithmetic, lane predicates and the two-register (lo / hi) 16-lane reduction.
// Here: 32-bit byte offsets from a wave-uniform base (saddr form), (row, slot) advanced without divisions, the four compare
// bits of a lane assembled by v_cmp + v_addc_co (nib = 2 nib + vcc: 8 instructions), ONE register reduced over the 8 lanes that
// share a 32-bit half of the word (3 DPP ORs), the upper half fetched by a row mirror.  ~20 VALU per float4.
template <int OP>
__device__ __forceinline__ uint32_t thr_nibble(const f32x4 v, const float th)
{
    uint32_t nib = 0;
    // nib = (((w) 2 + z) 2 + y) 2 + x : bit j = pixel j of the lane's four; NaN compares false in all four forms
#define CTK_CMP_ADDC(INS)                                                                                                         \
    asm volatile(INS " vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t" INS " vcc, %2, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t" \
                 INS " vcc, %3, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t" INS " vcc, %4, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"      \
                 : "+v"(nib) : "v"(v.w), "v"(v.z), "v"(v.y), "v"(v.x), "v"(th) : "vcc")
    if (OP == 0) CTK_CMP_ADDC("v_cmp_ge_f32");
    else if (OP == 1) CTK_CMP_ADDC("v_cmp_le_f32");
    else if (OP == 2) CTK_CMP_ADDC("v_cmp_gt_f32");
    else CTK_CMP_ADDC("v_cmp_lt_f32");
#undef CTK_CMP_ADDC
    return nib;
}

// U = loads per lane and step, picked by the host so that the steps of a chunk are (nearly) full: no predicates around the loads
// (a conditional load made hipcc wait for every load before issuing the next one), lanes beyond the chunk re-read its last row.
template <int OP, int U>
__device__ __forceinline__ void threshold_v7_body(const float *__restrict__ anom, const float *__restrict__ thr32,
                                                  int ny, int nx, int W, uint64_t *__restrict__ mask, int rb,
                                                  uint32_t *__restrict__ zero_counters, int xcd, int nostore)
{
    if (zero_counters && blockIdx.x == 0 && threadIdx.x < CTK_CNT_ZEROED) zero_counters[threadIdx.x] = 0u;
    const int nchunk = (ny + rb - 1) / rb;
    const unsigned bid = xcd_chunk(blockIdx.x, gridDim.x, xcd);
    const int t = (int)(bid / (unsigned)nchunk), y0 = (int)(bid - (unsigned)t * nchunk) * rb, tid = (int)threadIdx.x;
    const int rows = min(rb, ny - y0);
    const float th = thr32[t];
    const int n4 = nx >> 2, n4p = W << 4;                  // float4 slots per row, padded to whole 16-lane groups (= words)
    const int total = rows * n4p;
    const int64_t row0 = (int64_t)t * ny + y0;
    const char *base = (const char *)(anom + row0 * (int64_t)nx);      // wave-uniform; the lanes add 32-bit byte offsets
    char *mbase = (char *)(mask + row0 * W);
    const int sub = tid & 15;
    const uint32_t shift = (uint32_t)(4 * (sub & 7));
    // (row, slot) of the lane's next load, advanced by 256 slots at a time without dividing
    int nr = tid / n4p, nc = tid - nr * n4p;
    const int dr = 256 / n4p, dc = 256 - dr * n4p;
    const uint32_t pitch = (uint32_t)nx * 4u, wpitch = (uint32_t)W * 8u;
    for (int i0 = 0; i0 < total; i0 += 256 * U) {
        f32x4 v[U];
        uint32_t moff[U], vm[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int cl = min(nc, n4 - 1);                // padding slots re-read the row's last quad; their bits are cleared
            const int rl = min(nr, rows - 1);              // lanes beyond the chunk re-read its last row and store nothing
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + ((uint32_t)rl * pitch + (uint32_t)cl * 16u)));
            vm[u] = nc < n4 ? 0xfu : 0u;
            moff[u] = nr < rows ? (uint32_t)nr * wpitch + (uint32_t)(nc >> 4) * 8u : 0xffffffffu;
            nr += dr; nc += dc;
            if (nc >= n4p) { nc -= n4p; nr++; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t nib = thr_nibble<OP>(v[u], th) & vm[u];
            uint32_t x = nib << shift;                     // lanes 0-7 of the row build the low half of the word, lanes 8-15 the high half
            x = dpp_or<0xB1>(x);                           // quad_perm [1,0,3,2]
            x = dpp_or<0x4E>(x);                           // quad_perm [2,3,0,1]
            x = dpp_or<0x141>(x);                          // row_half_mirror: every lane of a half row holds its half
            const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false);      // row_mirror: lane 0 <- lane 15
            // (nostore: the kernel without its 1/32 write stream -- the yardstick of the mask placement check, ctk_api.hip)
            if (sub == 0 && moff[u] != 0xffffffffu && !nostore) *reinterpret_cast<uint64_t *>(mbase + moff[u]) = ((uint64_t)hi << 32) | x;
        }
    }
}
template <int OP, int U>
__global__ __launch_bounds__(256) void k_threshold_v7(const float *__restrict__ anom, const float *__restrict__ thr32,
                                                      int ny, int nx, int W, uint64_t *__restrict__ mask, int rb,
                                                      uint32_t *__restrict__ zero_counters, int xcd)
{
    threshold_v7_body<OP, U>(anom, thr32, ny, nx, W, mask, rb, zero_counters, xcd, 0);
}
// the same code under another name: the launches of the mask placement check (with and without the stores; ctk_api.hip) -- kept apart so
// that a profile's statistics of k_threshold_v7 are those of the passes
template <int OP, int U>
__global__ __launch_bounds__(256) void k_threshold_probe(const float *__restrict__ anom, const float *__restrict__ thr32,
                                                         int ny, int nx, int W, uint64_t *__restrict__ mask, int rb,
                                                         uint32_t *__restrict__ zero_counters, int xcd, int nostore)
{
    threshold_v7_body<OP, U>(anom, thr32, ny, nx, W, mask, rb, zero_counters, xcd, nostore);
}

// k_threshold_v6 (float32, nx <= 4096, any alignment): the mask word of 64 pixels IS the ballot of one compare when lane l holds
// pixel 64k + l -- dword loads with the row pointer in scalar registers and the lane offset in one VGPR, one v_cmp per 256 bytes,
// the 64-bit result moved from the scalar pair into lane k of two VGPRs (v_writelane), so that a wave leaves its chunk of
// R = 64 / W consecutive rows (R*W words, contiguous in the mask) with ONE coalesced store.  ~10 VALU instructions per KB read
// (k_threshold_v4: 71 -- nibbles placed, ORed across 16 lanes with 8 DPP steps, per-lane 64-bit addresses).
// Row by row: the W - 1 full words of a row in batches of U loads (immediate offsets from one row pointer; the remainder batch
// is picked by a switch on its size, so that no load or compare sits behind a predicate), the row's last word on its own (lanes
// past the row re-read its last pixel, their bits are cleared).  Word k of the chunk goes to lane k.
template <int OP, int N>
__device__ __forceinline__ void thr_batch(const float *p /* per lane */, float th, int lane, int k, uint32_t &mlo, uint32_t &mhi)
{
    float v[N > 0 ? N : 1];
#pragma unroll
    for (int u = 0; u < N; u++) v[u] = __builtin_nontemporal_load(p + u * 64);
#pragma unroll
    for (int u = 0; u < N; u++) {
        const uint64_t b = __ballot(cmp_op<OP, float>(v[u], th));
        if (lane == k + u) { mlo = (uint32_t)b; mhi = (uint32_t)(b >> 32); }
    }
}
template <int OP, int U /* loads in flight per lane and batch: 8 */>
__global__ __launch_bounds__(256) void k_threshold_v6(const float *__restrict__ anom, const float *__restrict__ thr32, int ny, int nx, int W,
                                                      uint64_t *__restrict__ mask, int R, int nchunk_t, int64_t nchunks,
                                                      uint32_t *__restrict__ zero_counters)
{
    static_assert(U == 8, "the remainder switch below lists 1..7");
    if (zero_counters && blockIdx.x == 0 && threadIdx.x < CTK_CNT_ZEROED) zero_counters[threadIdx.x] = 0u;
    const int lane = (int)(threadIdx.x & 63);
    // (the wave's index through readfirstlane: the compiler then knows that everything derived from it is wave-uniform and keeps
    // rows, words and pointers in scalar registers)
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = (int64_t)gridDim.x * 4;
    const int tail = nx - 64 * (W - 1);                               // valid pixels of a row's last word (1 .. 64)
    const uint64_t tail_mask = tail == 64 ? ~0ull : ((1ull << tail) - 1ull);
    const unsigned off_tail = (unsigned)((W - 1) * 64 + min(lane, tail - 1));
    const int nbatch = (W - 1) / U, rem = (W - 1) - nbatch * U;
    for (int64_t ch = wave; ch < nchunks; ch += nwaves) {
        const int t = (int)(ch / nchunk_t), y0 = (int)(ch - (int64_t)t * nchunk_t) * R;
        const int64_t row0 = (int64_t)t * ny + y0;
        const int rows = min(R, ny - y0);
        const float th = thr32[t];
        const float *rowp = anom + row0 * (int64_t)nx;
        uint32_t mlo = 0, mhi = 0;
        int k = 0;                                                     // lane that takes the next word
        for (int r = 0; r < rows; r++, rowp += nx) {
            const float vt = __builtin_nontemporal_load(rowp + off_tail);
            const float *p = rowp + lane;
            for (int q = 0; q < nbatch; q++, p += U * 64, k += U) thr_batch<OP, U>(p, th, lane, k, mlo, mhi);
            switch (rem) {
            case 1: thr_batch<OP, 1>(p, th, lane, k, mlo, mhi); break;
            case 2: thr_batch<OP, 2>(p, th, lane, k, mlo, mhi); break;
            case 3: thr_batch<OP, 3>(p, th, lane, k, mlo, mhi); break;
            case 4: thr_batch<OP, 4>(p, th, lane, k, mlo, mhi); break;
            case 5: thr_batch<OP, 5>(p, th, lane, k, mlo, mhi); break;
            case 6: thr_batch<OP, 6>(p, th, lane, k, mlo, mhi); break;
            case 7: thr_batch<OP, 7>(p, th, lane, k, mlo, mhi); break;
            default: break;
            }
            k += rem;
            const uint64_t b = __ballot(cmp_op<OP, float>(vt, th)) & tail_mask;
            if (lane == k) { mlo = (uint32_t)b; mhi = (uint32_t)(b >> 32); }
            k++;
        }
        if (lane < k) mask[row0 * W + lane] = ((uint64_t)mhi << 32) | mlo;
    }
}

template <int OP, typename TIN>
__global__ __launch_bounds__(256) void k_threshold(const TIN *__restrict__ anom, const TIN *__restrict__ thr32,
                                                   int64_t nrows, int ny, int nx, int W, uint64_t *__restrict__ mask,
                                                   uint32_t *__restrict__ zero_counters)
{
    if (zero_counters && blockIdx.x == 0 && threadIdx.x < CTK_CNT_ZEROED) zero_counters[threadIdx.x] = 0u;
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const TIN qnan = (TIN)__builtin_nanf("");
    for (int64_t row = wave; row < nrows; row += nwaves) {
        const TIN th = thr32[row / ny];
        const TIN *src = anom + row * (int64_t)nx;
        for (int w0 = 0; w0 < W; w0 += WAVE) {
            const int wn = min(WAVE, W - w0);
            uint64_t mine = 0;
            for (int k = 0; k < wn; k += 8) {                  // 8 independent loads in flight per lane
                TIN v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    int x = (w0 + k + j) * 64 + lane;
                    v[j] = (x < nx) ? src[x] : qnan;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    uint64_t b = __ballot(cmp_op<OP, TIN>(v[j], th));
                    if (lane == k + j) mine = b;
                }
            }
            if (lane < wn) mask[row * W + w0 + lane] = mine;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K0b runs per row / per word from the mask: one workgroup per timestep, one thread per row.
//   wstart[row][w] = run starts in words < w of the row      rowstart[t][y] = first run of row y
//   tcount[t]      = runs of the timestep
// ------------------------------------------------------------------------------------------------
#define RC_ROWS 2048
__global__ __launch_bounds__(1024) CTK_SGPR_8WAVES void k_rowcount(const uint64_t *__restrict__ mask, int ny, int W, uint16_t *__restrict__ wstart,
                                                  uint32_t *__restrict__ rowstart, uint32_t *__restrict__ tcount)
{
    const int t = (int)blockIdx.x, tid = (int)threadIdx.x;
    __shared__ uint32_t sm[17];
    const int nthr = (int)blockDim.x, nwv = nthr >> 6;       // 256 threads, or up to 1024 in the first form when the timesteps alone do not fill the chip
    uint32_t carry_rows = 0;
    if (W <= 64 && ny <= RC_ROWS) {
        // Rows to waves: a wave takes floor(64 / W) whole rows per step, RCU steps' words in flight; the prefix inside a row is a
        // segmented wave scan (no barrier), the row totals go to LDS and ONE block scan over them gives the row starts.  (The
        // form below walks the rows 256 / W at a time with a block scan -- two barriers -- per step: a chain of 65 of them for a
        // 721 x 1440 plane, 68 us; 0.8 ms on the 14 600-step slab.)
        __shared__ uint32_t rowtot[RC_ROWS];
        constexpr int RCU = 8;
        const int lane = tid & 63, wv = tid >> 6;
        const int rpw = 64 / W;                                    // rows per wave step
        const int seg = lane / W, wl = lane - seg * W;             // row of the step and word of the row this lane holds
        const bool lane_used = seg < rpw;
        const int64_t base = (int64_t)t * ny;
        for (int y0 = wv * rpw; y0 < ny; y0 += nwv * rpw * RCU) {
            uint64_t m[RCU], pv[RCU];
#pragma unroll
            for (int u = 0; u < RCU; u++) {
                const int y = y0 + u * nwv * rpw + seg;
                m[u] = 0; pv[u] = 0;
                if (lane_used && y < ny) {
                    const uint64_t *mw = mask + (base + y) * W;
                    m[u] = mw[wl];
                }
            }
#pragma unroll
            for (int u = 0; u < RCU; u++) {                         // the word to the left: the lane to the left holds it (same row)
                const uint64_t left = shfl_up_u64(m[u], 1);
                pv[u] = wl > 0 ? left : 0ull;
            }
#pragma unroll
            for (int u = 0; u < RCU; u++) {
                const int y = y0 + u * nwv * rpw + seg;
                const uint32_t c = (uint32_t)__popcll(m[u] & ~((m[u] << 1) | (pv[u] >> 63)));
                uint32_t inc = c;                                  // inclusive scan inside the row's W lanes
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(inc, d);
                    if (wl >= d) inc += o;
                }
                if (lane_used && y < ny) {
                    wstart[(base + y) * W + wl] = (uint16_t)(inc - c);
                    if (wl == W - 1) rowtot[y] = inc;
                }
            }
        }
        __syncthreads();
        for (int yb = 0; yb < ny; yb += nthr) {
            const int y = yb + tid;
            uint32_t tot;
            const uint32_t ex = block_excl_scan(y < ny ? rowtot[y] : 0u, sm, &tot);
            if (y < ny) rowstart[base + y] = carry_rows + ex;
            carry_rows += tot;
        }
    } else if (W <= 256) {
        // one thread per WORD, k = 256 / W whole rows per step: the mask is read and the per-word prefixes are written
        // as contiguous streams (a thread per row walked its words with a stride of W words: 4.2 ms for the
        // 14 600 x 721 x 1440 slab, this form 0.5)
        __shared__ uint32_t pl[256];
        const int k = 256 / W, r = tid / W, w = tid - r * W;
        const int64_t base = (int64_t)t * ny;
        for (int y0 = 0; y0 < ny; y0 += k) {
            const int y = y0 + r;
            const bool live = r < k && y < ny;
            uint32_t c = 0;
            if (live) {
                const uint64_t *mw = mask + (base + y) * W;
                const uint64_t m = mw[w], prev = w > 0 ? mw[w - 1] >> 63 : 0ull;
                c = (uint32_t)__popcll(m & ~((m << 1) | prev));
            }
            uint32_t tot;
            const uint32_t ex = block_excl_scan(c, sm, &tot);
            pl[tid] = ex;
            __syncthreads();
            if (live) {
                const uint32_t rs = pl[r * W];                    // prefix at the row's first word
                wstart[(base + y) * W + w] = (uint16_t)(ex - rs);
                if (w == 0) rowstart[base + y] = carry_rows + rs;
            }
            carry_rows += tot;
            __syncthreads();                                      // pl is rewritten in the next step
        }
    } else {
        for (int y0 = 0; y0 < ny; y0 += 256) {                    // very wide grids (nx > 16384): one thread per row
            const int y = y0 + tid;
            uint32_t n = 0;
            if (y < ny) {
                const int64_t row = (int64_t)t * ny + y;
                const uint64_t *mw = mask + row * W;
                uint16_t *ws = wstart + row * W;
                uint64_t carry = 0;
                for (int w = 0; w < W; w++) {
                    const uint64_t m = mw[w];
                    ws[w] = (uint16_t)n;
                    n += (uint32_t)__popcll(m & ~((m << 1) | carry));
                    carry = m >> 63;
                }
            }
            uint32_t tot;
            const uint32_t ex = block_excl_scan(n, sm, &tot);
            if (y < ny) rowstart[(int64_t)t * ny + y] = carry_rows + ex;
            carry_rows += tot;
        }
    }
    if (tid == 0) tcount[t] = carry_rows;
}

// ------------------------------------------------------------------------------------------------
// K1  exclusive scan of a uint32 vector by ONE workgroup (n is the number of timesteps: small)
// out[0..n] (n+1 entries, out[n] = total).  Totals beyond 2^32-1 set *ovf.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v)
{
    int lane = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint64_t o = shfl_up_u64(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// base_ptr (time shards): the scan starts at *base_ptr (the number of halo components) and out[-1] = 0
__global__ __launch_bounds__(1024) void k_scan_u32(const uint32_t *__restrict__ in, int64_t n,
                                                   uint32_t *__restrict__ out, uint32_t *ovf, uint32_t *mail = nullptr,
                                                   const uint32_t *base_ptr = nullptr, uint32_t stamp = 0 /* written to mail[4] after the rest: the host polls for it */)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint32_t wmax[16];
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = min(n, (int64_t)tid * per), e = min(n, b + per);
    uint64_t s = 0;
    uint32_t mx = 0;
    for (int64_t i = b; i < e; i++) { const uint32_t v = in[i]; s += v; mx = max(mx, v); }
    uint64_t inc = wave_incl_scan_u64(s);
    if (lane == WAVE - 1) wsum[wv] = inc;
    if (mail) {
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
        if (lane == 0) wmax[wv] = mx;
    }
    __syncthreads();
    uint64_t base = base_ptr ? (uint64_t)*base_ptr : 0ull;
    if (base_ptr && tid == 0) out[-1] = 0u;
    for (int i = 0; i < wv; i++) base += wsum[i];
    uint64_t run = base + inc - s;
    for (int64_t i = b; i < e; i++) { out[i] = (uint32_t)run; run += in[i]; }
    if (tid == 1023) {
        out[n] = (uint32_t)run;
        if (run > 0xffffffffull) atomicOr(ovf, CTK_OVF_RUNS);
        if (mail) {
            // the host's view of the scan, written straight into pinned host memory: total, largest item, overflow,
            // and the last item (total - last = out[n-1])
            uint32_t m = 0;
            for (int i = 0; i < 16; i++) m = max(m, wmax[i]);
            mail[0] = (uint32_t)run; mail[1] = m; mail[2] = run > 0xffffffffull ? CTK_OVF_RUNS : 0u; mail[3] = n > 0 ? in[n - 1] : 0u;
            if (stamp) { __threadfence_system(); __hip_atomic_store(&mail[4], stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
}

// Long shards (438 000 timesteps: the one-workgroup scan above takes 0.87 ms, each thread walking 428 items twice, uncoalesced): block
// sums and maxima first (k_scan_blocks_sum), then every workgroup scans its 1024 items behind the sum of the blocks in front of it.
// The mail (total, largest item, overflow, last item, stamp) is written by the workgroup of the last block, from the block sums alone.
#define CTK_SCAN_BLOCK 1024
__global__ __launch_bounds__(CTK_SCAN_BLOCK) void k_scan_blocks_sum(const uint32_t *__restrict__ in, int64_t n, uint64_t *__restrict__ bsum, uint32_t *__restrict__ bmax)
{
    __shared__ uint64_t ws[CTK_SCAN_BLOCK / 64];
    __shared__ uint32_t wm[CTK_SCAN_BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * CTK_SCAN_BLOCK + threadIdx.x;
    const uint32_t v = i < n ? in[i] : 0u;
    uint64_t s = v;
    uint32_t m = v;
    for (int o = 32; o > 0; o >>= 1) { s += (uint64_t)__shfl_xor((unsigned long long)s, o); m = max(m, (uint32_t)__shfl_xor((int)m, o)); }
    if (lane_id() == 0) { ws[threadIdx.x >> 6] = s; wm[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0; uint32_t mm = 0;
        for (int k = 0; k < CTK_SCAN_BLOCK / 64; k++) { t += ws[k]; mm = max(mm, wm[k]); }
        bsum[blockIdx.x] = t; bmax[blockIdx.x] = mm;
    }
}
__global__ __launch_bounds__(CTK_SCAN_BLOCK) void k_scan_blocks(const uint32_t *__restrict__ in, int64_t n, uint32_t *__restrict__ out, uint32_t *ovf,
                                                                const uint64_t *__restrict__ bsum, const uint32_t *__restrict__ bmax, uint32_t *mail, uint32_t stamp)
{
    __shared__ uint64_t ws[CTK_SCAN_BLOCK / 64];
    __shared__ uint32_t wm[CTK_SCAN_BLOCK / 64];
    __shared__ uint64_t sbase;
    const int tid = (int)threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    // sum (and, in the last workgroup, maximum) over the blocks in front
    uint64_t s = 0;
    uint32_t m = 0;
    for (int k = tid; k < b; k += CTK_SCAN_BLOCK) s += bsum[k];
    if (b == nb - 1 && mail) for (int k = tid; k < nb; k += CTK_SCAN_BLOCK) m = max(m, bmax[k]);
    for (int o = 32; o > 0; o >>= 1) { s += (uint64_t)__shfl_xor((unsigned long long)s, o); m = max(m, (uint32_t)__shfl_xor((int)m, o)); }
    if (lane == 0) { ws[wv] = s; wm[wv] = m; }
    __syncthreads();
    if (tid == 0) { uint64_t t = 0; for (int k = 0; k < CTK_SCAN_BLOCK / 64; k++) t += ws[k]; sbase = t; }
    __syncthreads();
    const uint64_t base = sbase;
    uint32_t mall = 0;
    for (int k = 0; k < CTK_SCAN_BLOCK / 64; k++) mall = max(mall, wm[k]);
    __syncthreads();
    const int64_t i = (int64_t)b * CTK_SCAN_BLOCK + tid;
    const uint32_t v = i < n ? in[i] : 0u;
    const uint64_t inc = wave_incl_scan_u64((uint64_t)v);
    if (lane == WAVE - 1) ws[wv] = inc;
    __syncthreads();
    uint64_t run = base + inc - v;
    for (int k = 0; k < wv; k++) run += ws[k];
    if (i < n) out[i] = (uint32_t)run;
    if (b == nb - 1 && i == n - 1) {
        const uint64_t total = run + v;
        out[n] = (uint32_t)total;
        if (total > 0xffffffffull) atomicOr(ovf, CTK_OVF_RUNS);
        if (mail) {
            mail[0] = (uint32_t)total; mail[1] = mall; mail[2] = total > 0xffffffffull ? CTK_OVF_RUNS : 0u; mail[3] = v;
            if (stamp) { __threadfence_system(); __hip_atomic_store(&mail[4], stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2  2-D connected-component labelling of one timestep on RUNS, 8-connectivity, plus the
//     longitude seam merge.           (contrack/contrack.py:684-698; scipy.ndimage.label numbering)
//
// One workgroup per timestep.  Run arrays + union-find parents live in LDS (k_label2d_lds) or, for
// timesteps with more runs than the LDS variant carries, in a global scratch area (k_label2d_glb).
//   phase 1  rowstart of the timestep (k_rowcount) -> LDS
//   phase 2  run extraction from the mask words (thread per word, ctz over start/end bits)
//   phase 3  union every run with the runs of the previous row it touches (8-connectivity: x0-1..x1+1)
//   phase 4  flatten -> root of every run; root = smallest run index = the run holding the component's
//            first raster pixel, so ranking the roots reproduces scipy's label order
//   phase 5  seam: rows whose first run starts at x=0 and last run ends at x=nx-1 unite those two runs
//            (same-row pairs only, contrack.py:693-698); second flatten -> merged component
//   phase 6  component ids, per-component bbox and exact area limbs, seam-row records
// ------------------------------------------------------------------------------------------------
struct Label2dArgs {
    const uint64_t *mask;
    const uint16_t *wstart;        // [T][ny][W] run starts left of each word (k_rowcount)
    const uint32_t *rowstart;      // [T][ny]
    const uint32_t *run_base;      // [T+1]
    uint32_t *run_comp;
    uint32_t *ncomp;               // [T]
    // per-component tables in run-indexed scratch slots (slot = run_base[t] + c)
    uint32_t *cs_mrep;
    uint32_t *cs_box;              // 4 x uint32 per slot: y0, y1, x0, x1
    int64_t *cs_area;              // 2 per slot
    CtkSeam *seams;                // row-indexed scratch [T][ny]
    uint32_t *seam_cnt;            // [T]
    uint32_t *counters;            // CTK_CNT_*
    const int64_t *wlo, *whi;      // [ny] weight limbs
    int ny, nx, W;
    uint32_t lds_cap;              // runs the LDS variant carries
    uint32_t cap_runs;             // runs the run-indexed buffers hold (a speculative launch may precede the size check)
    // global scratch for the fallback variant (indexed by run_base[t] + r)
    uint16_t *g_x0, *g_x1, *g_y;
    uint32_t *g_parent, *g_root, *g_idmap;
};

#ifdef CTK_PHASE_TIMING
__device__ unsigned long long g_phase_t[16];
#define PHASE_MARK(k) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) g_phase_t[k] = wall_clock64(); } while (0)
#else
#define PHASE_MARK(k) do { } while (0)
#endif

// What k_label2d_lds fetched for a timestep whose mask it stages in LDS, before it knew anything but t: the run prefixes of the
// thread's words (word k = tid + u * THREADS).  NW = 0: nothing was fetched.
template <int NW> struct L2dPrefetch { uint16_t w[NW > 0 ? NW : 1]; };

template <int THREADS, typename IT /* uint16_t when run / component indices fit, else uint32_t */, int LDS_COMPS, int NW = 0>
__device__ __forceinline__ void label2d_body(const Label2dArgs &a, const int t, const uint32_t nruns,
                                             uint16_t *x0, uint16_t *x1, uint16_t *yrow, uint32_t *parent,
                                             IT *root, IT *idmap, IT *rs /* rowstart within the timestep, ny+1 */,
                                             uint32_t *sm_scan, const uint64_t *mrow /* the timestep's mask words (LDS or global) */,
                                             const bool mrow_staged /* mrow is the whole timestep in LDS */,
                                             void *lds_tab /* LDS_COMPS x 32 B of LDS: band staging, then the component tables; or nullptr */,
                                             const uint32_t rbase, const L2dPrefetch<NW> &pf)
{
    const int tid = (int)threadIdx.x;
    const int ny = a.ny, nx = a.nx, W = a.W;

    PHASE_MARK(0);
    // ---- phase 1: rowstart (computed by k_rowcount) -> LDS/scratch -------------------------------
    if (NW == 0) {                                                      // (else: the caller did, together with the mask words)
        for (int y = tid; y < ny; y += THREADS) rs[y] = (IT)a.rowstart[(int64_t)t * ny + y];
        if (tid == 0) rs[ny] = (IT)nruns;
    }
    __syncthreads();

    PHASE_MARK(1);
    // ---- phase 2: run extraction, one thread per mask word -------------------------------------
    // the i-th start bit and the i-th end bit of a row delimit its i-th run; the number of starts / ends in the
    // words to the left comes from k_rowcount's per-word prefix (an open run at the word boundary has started
    // but not ended yet).  The mask words are read from LDS: either the whole timestep was staged by the caller
    // (mrow_staged) or bands of rows are staged here through the table area.
    {
        const uint16_t *wglob = a.wstart + (int64_t)t * ny * W;
        const uint64_t *mglob = a.mask + (int64_t)t * ny * W;
        const int nwords = ny * W;
        // one word: the i-th start bit and the i-th end bit of a row delimit its i-th run
        // (y, w) = row and word-in-row of word k: the callers advance them without dividing (an integer division by a run-time W is ~25
        // VALU instructions, five of them per thread of the staged form: a tenth of this kernel's instructions -- it is issue-bound)
        auto extract = [&](int k, int y, int w, uint64_t m, uint64_t left, uint64_t right, uint32_t wpre) {
            (void)k;
            const uint64_t cin = (w > 0) ? (left >> 63) : 0ull;
            const uint64_t nin = (w + 1 < W) ? (right & 1ull) : 0ull;
            uint64_t starts = m & ~((m << 1) | cin);
            uint64_t ends = m & ~((m >> 1) | (nin << 63));
            const uint32_t nbefore = rs[y] + wpre;
            uint32_t si = nbefore, ei = nbefore - (uint32_t)(cin & m & 1ull);      // open run: started, not yet ended
            const int xb = w * 64;
            while (starts) {
                int b = __builtin_ctzll(starts);
                starts &= starts - 1;
                x0[si] = (uint16_t)(xb + b);
                yrow[si] = (uint16_t)y;
                parent[si] = si;
                si++;
            }
            while (ends) {
                int b = __builtin_ctzll(ends);
                ends &= ends - 1;
                x1[ei] = (uint16_t)(xb + b);
                ei++;
            }
        };
        const int dq = THREADS / W, dr = THREADS - dq * W;           // word k + THREADS lies dq rows and dr words further (+ a carry)
        int yk = tid / W, wk = tid - yk * W;                         // (one division per thread)
        if (NW > 0) {                                                // the timestep's words are in LDS, the prefixes of this thread's in registers
#pragma unroll
            for (int u = 0; u < (NW > 0 ? NW : 1); u++) {
                const int k = tid + u * THREADS;
                if (k >= nwords) break;
                const int y = yk, w = wk;
                yk += dq; wk += dr;
                if (wk >= W) { wk -= W; yk++; }
                const uint64_t m = mrow[k];
                if (m == 0ull) continue;
                extract(k, y, w, m, k > 0 ? mrow[k - 1] : 0ull, k + 1 < nwords ? mrow[k + 1] : 0ull, pf.w[u]);
            }
        } else if (mrow_staged) {                                    // the timestep's words are in LDS
            const uint16_t *ws = wglob;
            for (int k = tid; k < nwords; k += THREADS) {
                const int y = yk, w = wk;
                yk += dq; wk += dr;
                if (wk >= W) { wk -= W; yk++; }
                const uint64_t m = mrow[k];
                if (m == 0ull) continue;
                extract(k, y, w, m, k > 0 ? mrow[k - 1] : 0ull, k + 1 < nwords ? mrow[k + 1] : 0ull, ws[k]);
            }
        } else {
            // Straight from global memory (L2), P2B words per thread at a time: every load is issued before the first word is
            // looked at.  (Staging bands of rows through LDS first cost a dependent round trip and two barriers per band: eleven
            // bands = 29 of the 54 us a 721 x 1440 timestep took.)
            constexpr int P2B = THREADS == 1024 ? 4 : 8;          // (the 1024-thread variants run at 64 VGPRs: eight words in flight spilled 20 B per lane)
            for (int k0 = tid; k0 < nwords; k0 += THREADS * P2B) {
                uint64_t m[P2B], ml[P2B], mr[P2B];
                uint32_t wp[P2B];
                int yy[P2B], ww[P2B];
#pragma unroll
                for (int u = 0; u < P2B; u++) {
                    const int k = k0 + u * THREADS, kc = min(k, nwords - 1);
                    m[u] = k < nwords ? mglob[kc] : 0ull;
                    ml[u] = mglob[max(kc - 1, 0)];
                    mr[u] = mglob[min(kc + 1, nwords - 1)];
                    wp[u] = wglob[kc];
                    yy[u] = yk; ww[u] = wk;
                    yk += dq; wk += dr;
                    if (wk >= W) { wk -= W; yk++; }
                }
#pragma unroll
                for (int u = 0; u < P2B; u++) if (m[u] != 0ull) extract(k0 + u * THREADS, yy[u], ww[u], m[u], ml[u], mr[u], wp[u]);
            }
        }
    }
    __syncthreads();

    PHASE_MARK(2);
    // ---- phase 3: unions with the previous row ------------------------------------------------
    for (uint32_t r = tid; r < nruns; r += THREADS) {
        const int y = yrow[r];
        if (y == 0) continue;
        uint32_t lo = rs[y - 1], hi = rs[y];
        if (lo == hi) continue;
        const int xa = (int)x0[r] - 1, xb = (int)x1[r] + 1;
        // first run s in [lo,hi) with x1[s] >= xa
        uint32_t l = lo, h = hi;
        while (l < h) {
            uint32_t m = (l + h) >> 1;
            if ((int)x1[m] < xa) l = m + 1; else h = m;
        }
        for (uint32_t s = l; s < hi && (int)x0[s] <= xb; s++) uf_unite(parent, r, s);
    }
    __syncthreads();

    PHASE_MARK(3);
    // ---- phase 4: flatten (no-wrap components) + ids ---------------------------------------------------
    // Round 6: every thread takes `per` CONSECUTIVE runs (raster order is kept: thread by thread, run by run), finds their roots and
    // counts its roots on the way; ONE block scan numbers them (was: a strided flatten pass, then one scan per THREADS runs -- two at
    // 1 degree, four barriers -- over the roots read back from LDS).
    uint32_t ncomp = 0;
    {
        const uint32_t per = (nruns + THREADS - 1) / THREADS, rb = min(nruns, (uint32_t)tid * per), re = min(nruns, rb + per);
        uint32_t v = 0, tot;
        for (uint32_t r = rb; r < re; r++) { const uint32_t q = uf_find(parent, r); root[r] = (IT)q; v += (q == r) ? 1u : 0u; }
        uint32_t ex = block_excl_scan(v, sm_scan, &tot);                 // (its first barrier: every root is in LDS)
        for (uint32_t r = rb; r < re; r++) if (root[r] == r) idmap[r] = (IT)(ex++);      // (meaningful at roots only)
        ncomp = tot;
    }
    PHASE_MARK(4);
    if (tid == 0) a.ncomp[t] = ncomp;
    uint32_t *cmrep = a.cs_mrep + rbase;
    uint32_t *gbox = a.cs_box + (int64_t)rbase * 4;
    int64_t *garea = a.cs_area + (int64_t)rbase * 2;
    // per-component bbox / exact area: accumulated with LDS atomics when the timestep's components fit the
    // LDS table (the staged mask words are dead by now and lend their space), else with global atomics
    const bool tab_lds = lds_tab != nullptr && ncomp <= (uint32_t)LDS_COMPS;
    int64_t *carea = tab_lds ? (int64_t *)lds_tab : garea;
    uint32_t *cbox = tab_lds ? (uint32_t *)((int64_t *)lds_tab + 2 * LDS_COMPS) : gbox;
    // Round 6: the seam merge (contrack.py:693-698) is a union-find over COMPONENTS -- a few dozen per plane, a handful of seam rows --
    // in the run-level parent array, which is dead once the roots are known.  Component ids follow the raster order of the root
    // runs, so "smallest id of the set" is the component of the smallest run: the representative the run-level form found by
    // uniting the seam runs on top of the flattened forest and flattening ALL runs a second time (three more passes over the runs
    // and five barriers).
    for (uint32_t c = tid; c < ncomp; c += THREADS) {
        parent[c] = c;
        cbox[c * 4 + 0] = 0xffffu; cbox[c * 4 + 1] = 0u; cbox[c * 4 + 2] = 0xffffu; cbox[c * 4 + 3] = 0u;
        carea[c * 2] = 0; carea[c * 2 + 1] = 0;
    }
    __syncthreads();

    PHASE_MARK(5);
    // ---- phase 6: seam unions between components, and the seam-row records ---------------------------------------------
    // seam rows: both seam pixels set (also when they are one run / one component: chain events can still
    // split them).  Written in y order into the timestep's slice of a row-indexed
    // scratch (at most ny records per timestep).  (Round 6: one pass over the rows for both; the scan's barriers are the ones the
    // unions need in front of the tables.)
    {
        uint32_t carry = 0;
        CtkSeam *out = a.seams + (int64_t)t * ny;
        for (int y0 = 0; y0 < ny; y0 += THREADS) {
            const int y = y0 + tid;
            uint32_t f = 0, l = 0, v = 0, cf = 0, cl = 0;
            if (y < ny) {
                f = rs[y]; l = rs[y + 1];
                if (f != l) {
                    l--;
                    v = (x0[f] == 0 && x1[l] == (uint16_t)(nx - 1)) ? 1u : 0u;
                    if (v) {
                        cf = idmap[root[f]]; cl = idmap[root[l]];
                        if (cf != cl) uf_unite(parent, cf, cl);
                    }
                }
            }
            uint32_t tot;
            const uint32_t ex = block_excl_scan(v, sm_scan, &tot);
            if (v) {
                CtkSeam q;
                q.t = (uint32_t)t; q.y = (uint32_t)y; q.cl = cf; q.cr = cl;
                out[carry + ex] = q;
            }
            carry += tot;
        }
        if (tid == 0) a.seam_cnt[t] = carry;
    }
    PHASE_MARK(6);
    // (the weight limbs of four runs' rows are requested before the first is used: a plane's ~500 runs would otherwise walk
    // row -> weights -> atomics two or three times in a row)
    constexpr int TBN = THREADS == 1024 ? 2 : 4;                  // (the 1024-thread variants run at 64 VGPRs)
    for (uint32_t r0 = tid; r0 < nruns; r0 += TBN * THREADS) {
        int yy[TBN];
        int64_t wl[TBN], wh[TBN];
#pragma unroll
        for (int j = 0; j < TBN; j++) yy[j] = yrow[min(r0 + (uint32_t)j * THREADS, nruns - 1u)];
#pragma unroll
        for (int j = 0; j < TBN; j++) { wl[j] = a.wlo[yy[j]]; wh[j] = a.whi[yy[j]]; }
#pragma unroll
        for (int j = 0; j < TBN; j++) {
            const uint32_t r = r0 + (uint32_t)j * THREADS;
            if (r >= nruns) break;
            const uint32_t rt = root[r];
            const uint32_t c = idmap[rt];
            a.run_comp[rbase + r] = c;
            const int y = yy[j];
            const int64_t len = (int64_t)x1[r] - (int64_t)x0[r] + 1;
            const uint32_t cm = uf_find(parent, c);            // seam-merged component: its area is what contrack.py:717 sums
            atomicAdd((unsigned long long *)&carea[cm * 2], (unsigned long long)(len * wl[j]));
            atomicAdd((unsigned long long *)&carea[cm * 2 + 1], (unsigned long long)(len * wh[j]));
            atomicMin(&cbox[c * 4 + 0], (uint32_t)y);
            atomicMax(&cbox[c * 4 + 1], (uint32_t)y);
            atomicMin(&cbox[c * 4 + 2], (uint32_t)x0[r]);
            atomicMax(&cbox[c * 4 + 3], (uint32_t)x1[r]);
            if (rt == r) cmrep[c] = cm;                        // (the representative: smallest component id of the merged set)
        }
    }
    if (tab_lds) {
        __syncthreads();
        for (uint32_t i = tid; i < ncomp * 4; i += THREADS) gbox[i] = cbox[i];
        for (uint32_t i = tid; i < ncomp * 2; i += THREADS) garea[i] = carea[i];
    }
    PHASE_MARK(7);
    PHASE_MARK(8);
}

#define CTK_LDS_RUNS 4096
#define CTK_LDS_NY 1024

// LDS variants: RUNS = most runs per timestep carried, COMPS = components whose bbox/area tables live in LDS
// (the table area doubles as the staging area of the timestep's mask words, COMPS*4 words).
//   <1024, 288>: 25.6 KB -> 6 workgroups per CU (typical 1 deg Z500 timestep: 500 runs, 40 components)
//   <2048, 512>: 46 KB -> 3 workgroups per CU
//   <4096, 512>: 76 KB -> 2 workgroups per CU (0.25 deg timesteps: ~1600 runs, mask words read through L2)
// (1024 threads: two workgroups per CU need 8 waves per SIMD, i.e. at most 64 VGPRs -- the compiler takes 71 when left alone, and
// 480 planes of a 0.25 deg grid then run as two rounds of one workgroup per CU: 86 us instead of ~45)
//   <832, 240, ..., 256, 256>: 19.9 KB -> 8 workgroups per CU: small planes (ny <= 256, at most 960 mask words: 192 x 288) in long shards,
//   where the kernel is bound by the planes in flight, not by a plane's chain (round 5; the planes with 833 .. 1024 runs go to
//   <1024, 288, 832, 256>); <768, 272, ..., 256, 256>: 20.3 KB, the same for planes of up to 1088 mask words (181 x 360; round 6)
template <int RUNS, int COMPS, int RUNS_BELOW, int THREADS, int NYCAP = CTK_LDS_NY>
__global__ __launch_bounds__(THREADS, (THREADS == 1024 || NYCAP < CTK_LDS_NY) ? 8 : 1) void k_label2d_lds(Label2dArgs a)
{
    const int t = (int)blockIdx.x;
#ifdef CTK_PHASE_TIMING
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) g_phase_t[15] = wall_clock64();      // kernel entry of the probed workgroup
    if (threadIdx.x == 0 && blockIdx.x == 0) g_phase_t[14] = wall_clock64();                  // ... and of the first one
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) g_phase_t[13] = wall_clock64();      // ... and of the last one
#endif
    __shared__ uint16_t x0[RUNS], x1[RUNS], yrow[RUNS], root[RUNS], idmap[RUNS];
    __shared__ uint32_t parent[RUNS];
    __shared__ uint16_t rs[NYCAP + 2];
    __shared__ uint64_t mlds[COMPS * 4];
    __shared__ uint32_t sm_scan[THREADS / 64 + 1];
    const int nwords = a.ny * a.W;
    const uint64_t *mg = a.mask + (int64_t)t * nwords;
    constexpr int NW = (COMPS * 4 + THREADS - 1) / THREADS;        // words per thread of a staged timestep
    if (THREADS < 1024 && nwords <= COMPS * 4 && NW <= 8) {     // (the 1024-thread variants serve planes far beyond their table area and run at 64 VGPRs)
        // A timestep whose mask is staged in LDS (8.7 KB at 1 deg).  Everything that depends on t alone is requested at once -- the
        // run range, the thread's mask words and their run prefixes, its rows' first runs -- and waited for once: as a chain
        // (range -> words -> first runs -> a prefix per word inside the extraction loop) these were up to eight dependent trips to
        // L2 in front of and inside phase 2.  Indices are clamped, nothing is conditional; a plane another variant takes
        // returns with its loads in flight.
        const uint32_t rb0 = a.run_base[t], rb1 = a.run_base[t + 1];
        const uint16_t *wg = a.wstart + (int64_t)t * nwords;
        const uint32_t *rg = a.rowstart + (int64_t)t * a.ny;
        uint64_t mreg[NW];
        uint32_t rreg[NW];
        L2dPrefetch<NW> pf;
#pragma unroll
        for (int u = 0; u < NW; u++) {
            const int k = min((int)threadIdx.x + u * THREADS, nwords - 1);
            mreg[u] = mg[k];
            pf.w[u] = wg[k];
            rreg[u] = rg[min((int)threadIdx.x + u * THREADS, a.ny - 1)];       // (ny <= nwords: NW rows per thread cover them)
        }
        const uint32_t nruns = rb1 - rb0;
        if (nruns > RUNS || (int64_t)nruns <= (int64_t)RUNS_BELOW || a.ny > NYCAP) return;    // another variant takes it
        if (rb1 > a.cap_runs) return;                                  // buffers too small: the host relaunches after growing them
        if (nruns == 0) {
            if (threadIdx.x == 0) { a.ncomp[t] = 0; a.seam_cnt[t] = 0; }
            return;
        }
#pragma unroll
        for (int u = 0; u < NW; u++) {
            const int k = (int)threadIdx.x + u * THREADS;
            if (k < nwords) mlds[k] = mreg[u];
            if (k < a.ny) rs[k] = (uint16_t)rreg[u];
        }
        if (threadIdx.x == 0) rs[a.ny] = (uint16_t)nruns;
        label2d_body<THREADS, uint16_t, COMPS, NW>(a, t, nruns, x0, x1, yrow, parent, root, idmap, rs, sm_scan, mlds, true, mlds, rb0, pf);
        return;
    }
    const uint32_t nruns = a.run_base[t + 1] - a.run_base[t];
    if (nruns > RUNS || (int64_t)nruns <= (int64_t)RUNS_BELOW || a.ny > NYCAP) return;        // another variant takes it
    if (a.run_base[t + 1] > a.cap_runs) return;                    // buffers too small: the host relaunches after growing them
    if (nruns == 0) {
        if (threadIdx.x == 0) { a.ncomp[t] = 0; a.seam_cnt[t] = 0; }
        return;
    }
    label2d_body<THREADS, uint16_t, COMPS>(a, t, nruns, x0, x1, yrow, parent, root, idmap, rs, sm_scan, mg, false, mlds, a.run_base[t], L2dPrefetch<0>());
}

__global__ __launch_bounds__(256) void k_label2d_glb(Label2dArgs a, uint32_t *g_rs /* [T][ny+1] scratch */)
{
    const int t = (int)blockIdx.x;
    const uint32_t rb = a.run_base[t];
    const uint32_t nruns = a.run_base[t + 1] - rb;
    if (!(nruns > CTK_LDS_RUNS || a.ny > CTK_LDS_NY)) return;
    if (a.run_base[t + 1] > a.cap_runs) return;
    __shared__ uint32_t sm_scan[8];
    label2d_body<256, uint32_t, 1>(a, t, nruns, a.g_x0 + rb, a.g_x1 + rb, a.g_y + rb, a.g_parent + rb, a.g_root + rb,
                                   a.g_idmap + rb, g_rs + (int64_t)t * (a.ny + 1), sm_scan, a.mask + (int64_t)t * a.ny * a.W, false, nullptr, rb, L2dPrefetch<0>());
}

// ------------------------------------------------------------------------------------------------
// K3  compaction of the per-component scratch slots into dense (t, c) order
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact_comps(const uint32_t *__restrict__ run_base, const uint32_t *__restrict__ ncomp,
                                                       const uint32_t *__restrict__ cprefix, const uint32_t *__restrict__ cs_mrep,
                                                       const uint32_t *__restrict__ cs_box, const int64_t *__restrict__ cs_area,
                                                       uint32_t *__restrict__ d_mrep, uint16_t *__restrict__ d_box,
                                                       int64_t *__restrict__ d_area, uint32_t *__restrict__ d_comp_t)
{
    const int t = (int)blockIdx.x;
    const uint32_t n = ncomp[t], rb = run_base[t], cb = cprefix[t];
    for (uint32_t c = threadIdx.x; c < n; c += blockDim.x) {
        d_mrep[cb + c] = cs_mrep[rb + c];
        for (int k = 0; k < 4; k++) d_box[(int64_t)(cb + c) * 4 + k] = (uint16_t)cs_box[(int64_t)(rb + c) * 4 + k];
        d_area[(int64_t)(cb + c) * 2] = cs_area[(int64_t)(rb + c) * 2];
        d_area[(int64_t)(cb + c) * 2 + 1] = cs_area[(int64_t)(rb + c) * 2 + 1];
        d_comp_t[cb + c] = (uint32_t)t;
    }
}

// Fused one-call path: the same compaction with (a) the component prefix computed by the workgroup itself -- the sum of the
// component counts of the timesteps in front of it, T^2 / 2 cached loads in all instead of a scan launch -- and (b) the resolver's
// per-component work arrays initialised on the way (k_rs_init and the second half of k_rs_pairs): one launch instead of four.
struct CompInit {
    int64_t *F, *B;                 // [NC][2]
    uint8_t *keep0, *keep1;
    uint32_t *touch, *parent;
    uint32_t *changed;              // [(passes + 1) * 64]
    uint32_t *ambig;
    uint32_t *pstate;               // [(T + 1) * stride] or nullptr
    const int32_t *next_tiny;       // [ny + 1]
    int nchanged, pstride;
    int64_t T;
    // time shards: the components of the previous shard's last timestep (the halo, *base_ptr of them) come first -- "timestep -1",
    // cprefix[-1] = 0, cprefix[0] = *base_ptr -- and are initialised here as their own representatives; nullptr: no halo
    const uint32_t *base_ptr;
    uint32_t *ovr_slot;             // [NC] or nullptr
    uint32_t *amb_cnt, *dcount;     // scalars reset with *ambig, or nullptr
    // long shards: sums of ncomp over blocks of CTK_CI_BLOCK timesteps (k_sum_blocks), so that the workgroup of timestep t adds up
    // t / CTK_CI_BLOCK block sums + < CTK_CI_BLOCK counts instead of t counts (438 000 steps: 10^11 loads otherwise); nullptr: short shard
    const uint32_t *bsum;
};
#define CTK_CI_BLOCK 1024
__global__ __launch_bounds__(CTK_CI_BLOCK) void k_sum_blocks(const uint32_t *__restrict__ in, int64_t n, uint32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * CTK_CI_BLOCK + threadIdx.x;
    __shared__ uint32_t sm[CTK_CI_BLOCK / 64];
    const uint32_t s = wave_sum_u32(i < n ? in[i] : 0u);
    if (lane_id() == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < CTK_CI_BLOCK / 64; k++) t += sm[k]; out[blockIdx.x] = t; }
}
__global__ __launch_bounds__(256) void k_compact_init(const uint32_t *__restrict__ run_base, const uint32_t *__restrict__ ncomp,
                                                      uint32_t *__restrict__ cprefix, const uint32_t *__restrict__ cs_mrep,
                                                      const uint32_t *__restrict__ cs_box, const int64_t *__restrict__ cs_area,
                                                      uint32_t *__restrict__ d_mrep, uint16_t *__restrict__ d_box,
                                                      int64_t *__restrict__ d_area, uint32_t *__restrict__ d_comp_t, CompInit ci)
{
    const int t = (int)blockIdx.x, tid = (int)threadIdx.x;
    __shared__ uint32_t sm[8];
    uint32_t s = 0;
    int u0 = 0;
    if (ci.bsum) { const int nb = t / CTK_CI_BLOCK; for (int u = tid; u < nb; u += blockDim.x) s += ci.bsum[u]; u0 = nb * CTK_CI_BLOCK; }
    for (int u = u0 + tid; u < t; u += blockDim.x) s += ncomp[u];
    uint32_t cb;
    (void)block_excl_scan(s, sm, &cb);
    const uint32_t nh = ci.base_ptr ? *ci.base_ptr : 0u;
    cb += nh;
    const uint32_t n = ncomp[t], rb = run_base[t];
    if (tid == 0) { cprefix[t] = cb; if (t == ci.T - 1) cprefix[ci.T] = cb + n; if (t == 0 && ci.base_ptr) cprefix[-1] = 0u; }
    if (t == 0)
        for (uint32_t g = tid; g < nh; g += blockDim.x) {              // halo components: each its own (already seam-resolved) representative
            d_mrep[g] = g; d_comp_t[g] = 0xffffffffu;
            for (int k = 0; k < 4; k++) d_box[(int64_t)g * 4 + k] = 0;
            d_area[(int64_t)g * 2] = 0; d_area[(int64_t)g * 2 + 1] = 0;
            ci.F[2 * (int64_t)g] = 0; ci.F[2 * (int64_t)g + 1] = 0; ci.B[2 * (int64_t)g] = 0; ci.B[2 * (int64_t)g + 1] = 0;
            ci.keep0[g] = 1; ci.keep1[g] = 1; ci.touch[g] = 0; ci.parent[g] = g;
            if (ci.ovr_slot) ci.ovr_slot[g] = 0;
        }
    for (uint32_t c = tid; c < n; c += blockDim.x) {
        const uint32_t g = cb + c;
        d_mrep[g] = cs_mrep[rb + c];
        for (int k = 0; k < 4; k++) d_box[(int64_t)g * 4 + k] = (uint16_t)cs_box[(int64_t)(rb + c) * 4 + k];
        d_area[(int64_t)g * 2] = cs_area[(int64_t)(rb + c) * 2];
        d_area[(int64_t)g * 2 + 1] = cs_area[(int64_t)(rb + c) * 2 + 1];
        d_comp_t[g] = (uint32_t)t;
        ci.F[2 * (int64_t)g] = 0; ci.F[2 * (int64_t)g + 1] = 0;
        ci.B[2 * (int64_t)g] = 0; ci.B[2 * (int64_t)g + 1] = 0;
        ci.keep0[g] = 1; ci.keep1[g] = 1;
        ci.touch[g] = 0;
        ci.parent[g] = g;
        if (ci.ovr_slot) ci.ovr_slot[g] = 0;
    }
    __syncthreads();
    // seam-merged components that hold a row with very low weight bits (ResolveDev::next_tiny): flag at the representative
    for (uint32_t c = tid; c < n; c += blockDim.x) {
        const uint32_t y0 = cs_box[(int64_t)(rb + c) * 4], y1 = cs_box[(int64_t)(rb + c) * 4 + 1];
        if (ci.next_tiny[y0] <= (int32_t)y1) ci.touch[cb + cs_mrep[rb + c]] = 1u;
    }
    if (t == 0) {
        for (int i = tid; i < ci.nchanged; i += blockDim.x) ci.changed[i] = 0u;
        if (tid == 0) { *ci.ambig = 0u; if (ci.amb_cnt) *ci.amb_cnt = 0u; if (ci.dcount) *ci.dcount = 0u; }
    }
    if (ci.pstate && tid == 0) { ci.pstate[(size_t)t * ci.pstride] = 0u; if (t == ci.T - 1) ci.pstate[(size_t)ci.T * ci.pstride] = 0u; }
}

// dense (t, y)-ordered seam records from the row-indexed scratch
__global__ __launch_bounds__(256) void k_compact_seams(const CtkSeam *__restrict__ scratch, const uint32_t *__restrict__ seam_cnt,
                                                       const uint32_t *__restrict__ seam_off, int ny, CtkSeam *__restrict__ out)
{
    const int t = (int)blockIdx.x;
    const uint32_t n = seam_cnt[t], o = seam_off[t];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) out[o + i] = scratch[(int64_t)t * ny + i];
}

// ------------------------------------------------------------------------------------------------
// K4  label co-occurrence histogram between timestep t and t-1     (contrack/contrack.py:718-719 areas,
//     and the temporal links of the 3-D structure at contrack.py:748-750)
// One workgroup per timestep; wave per row, lane per mask word; o = m[t] & m[t-1]; every maximal piece
// of o inside a word belongs to exactly one (run@t, run@t-1) pair, whose ranks follow from popcounts
// of the start bits.  (component@t, component@t-1) -> exact area limbs are accumulated in an
// LDS-resident hash table with LDS atomics; entries that do not find a slot are emitted directly
// (records are additive, duplicates are fine).
// ------------------------------------------------------------------------------------------------
#define CTK_HASH_SLOTS 512              // (12 KB: twelve workgroups per CU; a 1 degree timestep has ~40 pairs, a 0.25 degree one ~100)
#define CTK_HASH_PROBES 12

struct OverlapArgs {
    const uint64_t *mask;
    const uint16_t *wstart;
    const uint32_t *rowstart;
    const uint32_t *run_base;
    const uint32_t *run_comp;
    // halo = last timestep of the previous shard (used for t == 0 when has_prev)
    const uint64_t *halo_mask;
    const uint16_t *halo_wstart;
    const uint32_t *halo_rowstart;
    const uint32_t *halo_run_comp;
    int has_prev;
    CtkPair *pairs;                // grouped records: pairs[pair_base[t] .. +pair_cnt[t]) belong to timestep t
    uint32_t pair_cap;
    uint32_t *pair_base, *pair_cnt; // [T]
    uint32_t *counters;            // records that found no hash slot are stored from the END of `pairs` downwards
                                   // (pairs[pair_cap-1-i]), counted by CTK_CNT_UPAIRS
    const int64_t *wlo, *whi;
    int ny, nx, W;
    // fused one-call path: what k_rs_pairs derives from every record is written with the record -- dense ids of both components
    // and of their seam-merged representatives, and the forward overlap of the EARLIER component (contrack.py:718) added to F
    // (zeroed by k_compact_init).  nullptr: k_rs_pairs does it.
    const uint32_t *cprefix, *mrep;
    uint32_t *p_rc, *p_rd, *p_gc, *p_gd;
    int64_t *F;
    // ... and the grouped records of timestep t go to the FIXED slots pairs[t * pslot .. + pslot) (entries beyond: the ungrouped
    // path) instead of a range taken from one counter: 2707 workgroups adding to one address was most of this kernel's time
    // (~20 ns per add, serialised: 54 us).  0: ranges from the counter (the synchronous resolver wants them contiguous).
    uint32_t pslot;
    uint32_t upair_cap;            // ungrouped records the end of the buffer may hold
};

__device__ __forceinline__ void pair_prepare(const OverlapArgs &a, uint32_t slot, uint32_t cb, uint32_t db, uint32_t c, uint32_t d, int64_t lo, int64_t hi)
{
    const uint32_t gc = cb + c, gd = db + d;
    const uint32_t rd = db + a.mrep[gd];
    a.p_gc[slot] = gc; a.p_gd[slot] = gd; a.p_rc[slot] = cb + a.mrep[gc]; a.p_rd[slot] = rd;
    atomicAdd((unsigned long long *)&a.F[2 * (int64_t)rd], (unsigned long long)lo);
    atomicAdd((unsigned long long *)&a.F[2 * (int64_t)rd + 1], (unsigned long long)hi);
}

__device__ __forceinline__ void emit_pair(const OverlapArgs &a, uint32_t t, uint32_t c, uint32_t d, int64_t lo, int64_t hi, uint32_t cb = 0, uint32_t db = 0)
{
    uint32_t i = atomicAdd(&a.counters[CTK_CNT_UPAIRS], 1u);
    if (i < a.upair_cap) {
        CtkPair p;
        p.t = t; p.c = c; p.d = d; p.pad = 0; p.lo = lo; p.hi = hi;
        a.pairs[a.pair_cap - 1u - i] = p;
        if (a.p_rc) pair_prepare(a, a.pair_cap - 1u - i, cb, db, c, d, lo, hi);
    } else atomicOr(&a.counters[CTK_CNT_OVERFLOW], CTK_OVF_PAIRS);
}

// OVB: mask words per thread and step -- chosen by the host so that ONE step covers the timestep where it can (1086 words at
// 181 x 360: five per thread; with four a second step ran for the last 62 words)
// THREADS: 256, or 1024 when the timesteps alone leave the chip empty (480 x 721 x 1440: 16 steps of three round trips per plane
// with 256 threads, 4 with 1024)
template <int OVB, int THREADS = 256, int WPE = 1 /* waves per SIMD the compiler has to make room for (experiment: occupancy against spills) */>
__global__ __launch_bounds__(THREADS, WPE) void k_overlap(OverlapArgs a)
{
    const int t = (int)blockIdx.x;
    if (t == 0 && !a.has_prev) {
        if (threadIdx.x == 0) { a.pair_base[t] = 0; a.pair_cnt[t] = 0; }
        return;
    }
    const int tid = (int)threadIdx.x;
    const int ny = a.ny, W = a.W;
    // (fused path: the component prefixes of t and t-1, needed when the table is flushed -- requested now, used then)
    const uint32_t cbc = a.p_rc ? a.cprefix[t] : 0u, cbd = a.p_rc ? a.cprefix[t - 1] : 0u;
    __shared__ unsigned long long hkey[CTK_HASH_SLOTS];
    __shared__ long long hlo[CTK_HASH_SLOTS], hhi[CTK_HASH_SLOTS];
    __shared__ uint32_t sm_scan[THREADS / 64 + 1];
    __shared__ uint32_t out_base;
    for (int i = tid; i < CTK_HASH_SLOTS; i += THREADS) { hkey[i] = FULL64; hlo[i] = 0; hhi[i] = 0; }
    __syncthreads();

    const int nwords = ny * W;
    const uint64_t *mc = a.mask + (int64_t)t * nwords;
    const uint16_t *wsc = a.wstart + (int64_t)t * nwords;
    const uint32_t *rsc = a.rowstart + (int64_t)t * ny;
    const uint32_t *rcc = a.run_comp + a.run_base[t];
    const uint64_t *mp;
    const uint16_t *wsp;
    const uint32_t *rsp, *rcp;
    if (t == 0) { mp = a.halo_mask; wsp = a.halo_wstart; rsp = a.halo_rowstart; rcp = a.halo_run_comp; }
    else {
        mp = a.mask + (int64_t)(t - 1) * nwords; wsp = a.wstart + (int64_t)(t - 1) * nwords;
        rsp = a.rowstart + (int64_t)(t - 1) * ny; rcp = a.run_comp + a.run_base[t - 1];
    }
    // One thread per mask word, OVB words per thread and step.  The work on a word is a chain of dependent loads (the two mask
    // words -> neighbours, run prefixes, row starts, weights -> the two runs' components); written word by word that is three
    // round trips to L2 per word, most of this kernel's time.  Here every load of a level is issued for all OVB words before the
    // first is used: three round trips per OVB words.  (Words without common pixels -- nine of ten -- load their tables in vain:
    // L2 hits next to data that is read anyway.)
    auto insert = [&](uint32_t cc, uint32_t cd, int64_t lo, int64_t hi) {
        const unsigned long long key = ((unsigned long long)cc << 32) | cd;
        uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (CTK_HASH_SLOTS - 1);
        for (int q = 0; q < CTK_HASH_PROBES; q++) {
            const unsigned long long old = atomicCAS(&hkey[h], FULL64, key);
            if (old == FULL64 || old == key) {
                atomicAdd((unsigned long long *)&hlo[h], (unsigned long long)lo);
                atomicAdd((unsigned long long *)&hhi[h], (unsigned long long)hi);
                return;
            }
            h = (h + 1) & (CTK_HASH_SLOTS - 1);
        }
        emit_pair(a, (uint32_t)t, cc, cd, lo, hi, cbc, cbd);
    };
    const int lane = tid & 63;
    // (row, word-in-row) of the thread's next word, advanced by THREADS words at a time without dividing (a division by a run-time W is
    // ~25 VALU instructions; OVB of them per thread were a tenth of this kernel's instructions -- it is issue-bound: NOTES round 6)
    const int dq = THREADS / W, dr = THREADS - dq * W;
    int yk = tid / W, wk = tid - yk * W;
    for (int i0 = tid; i0 < nwords; i0 += THREADS * OVB) {
        // Registers decide how many of these workgroups a CU holds (every load in flight owns its destination): the words to the
        // left are not loaded -- consecutive threads hold consecutive words, so the carry-in bit comes from the lane to the left
        // (lane 0 of a wave loads the upper half of its left neighbours) -- and the row weights are requested with the components,
        // for the one word in ten that has common pixels.
        uint64_t c[OVB], p[OVB];
        uint32_t ec[OVB], ep[OVB], lt[OVB];                              // lt: bit 0 / 1 = carry-in of c / p (top bit of the word to the left, same row)
        uint32_t ltc[OVB], ltp[OVB];
        int yy[OVB];
        uint32_t first = 0;                                              // bit u: word u is the first of its row
#pragma unroll
        for (int u = 0; u < OVB; u++) {                                  // level 1 (and everything whose address is known already)
            const int idx = i0 + u * THREADS, ii = min(idx, nwords - 1), im = max(ii - 1, 0);
            const int y = min(yk, ny - 1);                               // (= ii / W: past the end the last row's tables are loaded in vain)
            if (wk == 0) first |= 1u << u;
            yk += dq; wk += dr;
            if (wk >= W) { wk -= W; yk++; }
            yy[u] = idx < nwords ? y : -1;
            c[u] = mc[ii]; p[u] = mp[ii];
            // lane 0's left neighbours: requested by EVERY lane (the others ask for the first word of the plane, one cached line) --
            // a load under `if (lane == 0)` made hipcc wait for every outstanding load right behind it, word after word
            const int il = lane == 0 ? 2 * im + 1 : 1;
            ltc[u] = reinterpret_cast<const uint32_t *>(mc)[il];
            ltp[u] = reinterpret_cast<const uint32_t *>(mp)[il];
            ec[u] = rsc[y] + wsc[ii]; ep[u] = rsp[y] + wsp[ii];          // runs started left of this word
        }
#pragma unroll
        for (int u = 0; u < OVB; u++) {
            const uint32_t mine = (uint32_t)(c[u] >> 63) | ((uint32_t)(p[u] >> 63) << 1);
            const uint32_t left = (uint32_t)__shfl_up((int)mine, 1);
            lt[u] = lane != 0 ? left : ((ltc[u] >> 31) | ((ltp[u] >> 31) << 1));
            if ((first >> u) & 1u) lt[u] = 0;                            // first word of a row: nothing to the left
            if (yy[u] < 0) c[u] = 0ull;                                  // past the end: no common pixels
        }
        // what the table phase needs of a word: c, p, lt (pieces are re-derived from them), the run prefixes, the components of
        // the first piece and the row weights
        uint32_t cc0[OVB], cd0[OVB];
        int64_t wl[OVB], wh[OVB];
        auto starts = [&](int u, uint64_t &sc, uint64_t &sp) {
            sc = c[u] & ~((c[u] << 1) | (uint64_t)(lt[u] & 1u)); sp = p[u] & ~((p[u] << 1) | (uint64_t)(lt[u] >> 1));
        };
#pragma unroll
        for (int u = 0; u < OVB; u++) {                                  // level 2: the components of the first common piece
            const uint64_t o = c[u] & p[u];
            cc0[u] = 0; cd0[u] = 0; wl[u] = 0; wh[u] = 0;
            if (o == 0ull) continue;
            uint64_t sc, sp;
            starts(u, sc, sp);
            const int b = __builtin_ctzll(o);
            const uint64_t below = (b + 1 >= 64) ? FULL64 : ((1ull << (b + 1)) - 1ull);   // bits 0..b
            cc0[u] = rcc[ec[u] + (uint32_t)__popcll(sc & below) - 1u];
            cd0[u] = rcp[ep[u] + (uint32_t)__popcll(sp & below) - 1u];
            wl[u] = a.wlo[yy[u]]; wh[u] = a.whi[yy[u]];
        }
#pragma unroll
        for (int u = 0; u < OVB; u++) {                                  // the table; further pieces of a word (rare) one by one
            uint64_t ou = c[u] & p[u];
            if (ou == 0ull) continue;
            int b = __builtin_ctzll(ou);
            uint64_t sh = ou >> b;
            int n = (~sh == 0ull) ? 64 : __builtin_ctzll(~sh);
            insert(cc0[u], cd0[u], (int64_t)n * wl[u], (int64_t)n * wh[u]);
            ou = (n >= 64 - b) ? 0ull : (ou & ~(((1ull << n) - 1ull) << b));
            if (ou == 0ull) continue;
            uint64_t sc, sp;
            starts(u, sc, sp);
            while (ou) {
                b = __builtin_ctzll(ou);
                sh = ou >> b;
                n = (~sh == 0ull) ? 64 : __builtin_ctzll(~sh);
                const uint64_t below = (b + 1 >= 64) ? FULL64 : ((1ull << (b + 1)) - 1ull);
                const uint32_t cc = rcc[ec[u] + (uint32_t)__popcll(sc & below) - 1u], cd = rcp[ep[u] + (uint32_t)__popcll(sp & below) - 1u];
                insert(cc, cd, (int64_t)n * wl[u], (int64_t)n * wh[u]);
                ou = (n >= 64 - b) ? 0ull : (ou & ~(((1ull << n) - 1ull) << b));
            }
        }
    }
    __syncthreads();
    // flush the table: ONE contiguous block of records per timestep (pair_base[t], pair_cnt[t])
    {
        uint32_t mine = 0;
        for (int i = tid; i < CTK_HASH_SLOTS; i += THREADS) mine += (hkey[i] != FULL64) ? 1u : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(mine, sm_scan, &tot);
        if (tid == 0) {
            if (a.pslot) {
                out_base = (uint32_t)t * a.pslot;
                a.pair_base[t] = out_base;
                a.pair_cnt[t] = min(tot, a.pslot);
            } else {
                out_base = tot ? atomicAdd(&a.counters[CTK_CNT_PAIRS], tot) : 0u;
                a.pair_base[t] = out_base;
                a.pair_cnt[t] = (tot && out_base + tot <= a.pair_cap) ? tot : 0u;
                if (tot && out_base + tot > a.pair_cap) atomicOr(&a.counters[CTK_CNT_OVERFLOW], CTK_OVF_PAIRS);
            }
        }
        __syncthreads();
        uint32_t j = out_base + ex;
        const uint32_t jend = a.pslot ? out_base + a.pslot : a.pair_cap;
        for (int i = tid; i < CTK_HASH_SLOTS; i += THREADS) {
            if (hkey[i] == FULL64) continue;
            if (a.pslot && j >= jend) {                                         // more entries than the timestep's slots: ungrouped
                emit_pair(a, (uint32_t)t, (uint32_t)(hkey[i] >> 32), (uint32_t)hkey[i], hlo[i], hhi[i], cbc, cbd);
                j++;
                continue;
            }
            if (j < jend) {
                CtkPair p;
                p.t = (uint32_t)t; p.c = (uint32_t)(hkey[i] >> 32); p.d = (uint32_t)hkey[i]; p.pad = 0;
                p.lo = hlo[i]; p.hi = hhi[i];
                a.pairs[j] = p;
                if (a.p_rc) pair_prepare(a, j, cbc, cbd, p.c, p.d, p.lo, p.hi);
            }
            j++;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-pixel fold of the ordered seam operations (contrack/contrack.py:753-763 semantics, see
// ctk_resolve.cpp): only used for the rare "complex" components.
//   first / next : per label, the chain of ops that have it as `hi`, in execution order
// ------------------------------------------------------------------------------------------------
#define CTK_CV 64                       // run values per chunk in the chunk-ordered copy (more runs in a chunk: staged from run_val)
#define CTK_CV_MAXCHUNK 1024            // chunks per timestep the copy is built for
struct FoldArgs {
    const CtkOp *ops;          // execution order
    const int32_t *first;      // [n_labels + 1] first op that has the label as `hi` (-1: none)
    const int32_t *next;       // [nops] next op with the same `hi`
    int32_t nops;
};

__device__ inline int32_t fold_pixel(const FoldArgs &f, int32_t l, int32_t t, int32_t y, int32_t x)
{
    int32_t s = 0;
    for (;;) {
        bool moved = false;
        for (int32_t idx = f.first[l]; idx >= 0; idx = f.next[idx]) {
            if (idx < s) continue;
            const CtkOp o = f.ops[idx];
            if (t >= o.t0 && t <= o.t1 && y >= o.y0 && y <= o.y1 && x >= o.x0 && x <= o.x1) {
                l = o.lo; s = idx + 1; moved = true; break;
            }
        }
        if (!moved) return l;
    }
}

// Visit every foreground pixel of row (t, y) with its run index (relative to the timestep).
// F(x, run) is called by the lane that owns pixel x (lane = x & 63).
template <typename F>
__device__ __forceinline__ void for_each_fg_pixel_in_row(const uint64_t *mw, int W, uint32_t rowstart, F f)
{
    const int lane = lane_id();
    uint32_t base = rowstart;
    uint64_t carry = 0;
    for (int w0 = 0; w0 < W; w0 += WAVE) {
        const int wn = min(WAVE, W - w0);
        uint64_t m = (lane < wn) ? mw[w0 + lane] : 0ull;
        uint64_t pm = shfl_up_u64(m, 1);
        uint64_t st = m & ~((m << 1) | ((lane == 0) ? carry : (pm >> 63)));
        uint32_t n = (uint32_t)__popcll(st);
        uint32_t inc = wave_incl_scan_u32(n);
        uint32_t before = base + inc - n;
        for (int k = 0; k < wn; k++) {
            uint64_t mk = shfl_u64(m, k);
            if (mk == 0ull) continue;                                  // wave-uniform
            uint64_t sk = shfl_u64(st, k);
            uint32_t bk = __shfl(before, k);
            if ((mk >> lane) & 1ull) {
                uint64_t below = (lane == 63) ? FULL64 : ((1ull << (lane + 1)) - 1ull);
                f((w0 + k) * 64 + lane, bk + (uint32_t)__popcll(sk & below) - 1u);
            }
        }
        base += __shfl(inc, WAVE - 1);
        carry = shfl_u64(m, wn - 1) >> 63;
    }
}

// ------------------------------------------------------------------------------------------------
// K5  time extents of the final ids (persistence, contrack/contrack.py:765-772).  One workgroup per
//     timestep; one atomic min/max per component; complex components fold pixel by pixel.
//     ext[0..n] = min t, ext[n+1 .. 2n+1] = max t  (global timestep numbers)
// ------------------------------------------------------------------------------------------------
// R5: final id of a component with fresh label l (> 0), box q = {y0, y1, x0, x1}, at timestep t.  All of a component's
// pixels move together through an op whose box contains the component's box, none moves when the boxes are disjoint;
// anything else is resolved per pixel (returns -l).
__device__ inline int32_t comp_final_label(const FoldArgs &f, int32_t l, int32_t t, int q0, int q1, int q2, int q3 /* the box, in registers */)
{
    if (f.nops == 0) return l;
    int32_t cur = l, s = 0;
    for (;;) {
        bool again = false;
        for (int32_t idx = f.first[cur]; idx >= 0; idx = f.next[idx]) {
            if (idx < s) continue;
            const CtkOp o = f.ops[idx];
            const bool t_in = t >= o.t0 && t <= o.t1;
            const bool inside = t_in && q0 >= o.y0 && q1 <= o.y1 && q2 >= o.x0 && q3 <= o.x1;
            const bool disjoint = !t_in || q1 < o.y0 || q0 > o.y1 || q3 < o.x0 || q2 > o.x1;
            if (inside) { cur = o.lo; s = idx + 1; again = true; break; }
            if (!disjoint) return -l;
        }
        if (!again) return cur;
    }
}

// fused one-call path: the kernels behind the resolver run without the host having looked at the tables.  `guard` = the device
// counters: a co-occurrence table that overflowed or a poisoned resolution (CTK_CNT_POISON) means the tables hold nothing
// usable -- the kernel returns at once and the host repeats the resolution on its synchronous path.
__device__ __forceinline__ bool ctk_guard_bad(const uint32_t *guard)
{
    return guard && ((guard[CTK_CNT_OVERFLOW] & CTK_OVF_PAIRS) != 0u || guard[CTK_CNT_POISON] != 0u);
}

struct ExtentArgs {
    const uint32_t *guard;
    const uint64_t *mask;
    const uint32_t *rowstart;
    const uint32_t *run_base;
    const uint32_t *run_comp;
    const uint32_t *ncomp;
    const uint32_t *cprefix;
    const int32_t *comp_label;     // dense (t,c) order
    const int32_t *lab;            // if set: fresh labels -- the final id is computed here and written to comp_label_w
    int32_t *comp_label_w;
    const uint16_t *box;           // [NC][4] y0, y1, x0, x1 of every component (rows of the per-pixel pass)
    int32_t *ext;
    int64_t n_labels;
    int64_t t_begin;
    FoldArgs fold;
    int ny, nx, W;
    int64_t T;                     // timesteps of the shard (k_extent_blk: several per workgroup)
};

// time extent of an id: look before the atomic (a long-lived id gets one update per timestep -- or per pixel -- and
// same-address atomics serialise)
__device__ inline void ext_update(int32_t *tmin, int32_t *tmax, int32_t l, int32_t tg)
{
    // a stale bound is looser than the true one -- at worst a superfluous atomic.  Device-scope loads: the L2s of the eight XCDs
    // are not coherent with each other, a plain load would keep returning the bound this XCD saw first.
    if (tg < __hip_atomic_load(&tmin[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&tmin[l], tg);
    if (tg > __hip_atomic_load(&tmax[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&tmax[l], tg);
}

__global__ __launch_bounds__(256) CTK_SGPR_8WAVES void k_extent(ExtentArgs a)
{
    if (ctk_guard_bad(a.guard)) return;
    const int t = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    const uint32_t n = a.ncomp[t], cb = a.cprefix[t];
    const int32_t tg = (int32_t)(a.t_begin + t);
    int32_t *tmin = a.ext, *tmax = a.ext + a.n_labels + 1;        // (n_labels + 1: the offset of the second half; an upper bound in the fused path)
    __shared__ int ylo, yhi;                               // rows that hold pixels of complex components
    if (tid == 0) { ylo = 0x7fffffff; yhi = -1; }
    __syncthreads();
    for (uint32_t c = tid; c < n; c += blockDim.x) {
        int32_t l;
        if (a.lab) {                                       // single-GPU path: k_rs_final's work, one launch less
            // (the box travels with the label, as one 8-byte load: read element by element inside the short-circuit comparisons
            // of the fold it was up to eight more dependent trips to L2 for every component of a contour that crosses the seam)
            l = a.lab[cb + c];
            const uint2 bq = *reinterpret_cast<const uint2 *>(a.box + 4 * (int64_t)(cb + c));
            if (l > 0) l = comp_final_label(a.fold, l, tg, (int)(bq.x & 0xffffu), (int)(bq.x >> 16), (int)(bq.y & 0xffffu), (int)(bq.y >> 16));
            a.comp_label_w[cb + c] = l;
        } else l = a.comp_label[cb + c];
        if (l > 0) { ext_update(tmin, tmax, l, tg); }
        else if (l < 0) { atomicMin(&ylo, (int)a.box[4 * (int64_t)(cb + c)]); atomicMax(&yhi, (int)a.box[4 * (int64_t)(cb + c) + 1]); }
    }
    __syncthreads();
    if (yhi < 0) return;
    const uint32_t *rc = a.run_comp + a.run_base[t];
    const int wv = tid >> 6, nwv = (int)(blockDim.x >> 6);
    for (int y = ylo + wv; y <= yhi && y < a.ny; y += nwv) {
        const uint64_t *mw = a.mask + ((int64_t)t * a.ny + y) * a.W;
        for_each_fg_pixel_in_row(mw, a.W, a.rowstart[(int64_t)t * a.ny + y], [&](int x, uint32_t run) {
            int32_t l = a.comp_label[cb + rc[run]];
            if (l < 0) {
                int32_t fl = fold_pixel(a.fold, -l, tg, y, x);
                ext_update(tmin, tmax, fl, tg);
            }
        });
    }
}

// k_extent with EX_TW consecutive timesteps per workgroup, one wave each (round 6).  The time extents are min / max reductions onto
// HOT addresses: an id that lives for hundreds of timesteps is updated by every one of them, from all eight XCDs, and each update is
// a device-scope look (the L2s of the XCDs are not coherent with each other) plus, sometimes, an atomic -- 2-5 us apiece under load.
// With one wave per timestep that chain is the kernel: 2.0 ms at 438 000 x 192 x 288 (93 % of the wave cycles waiting).  Here the
// waves of a workgroup reduce (id -> first / last timestep) in an LDS hash first and the workgroup touches global memory once per
// id: EX_TW times fewer hot operations (the scheme of k_fz_groups).  Complex components (per-pixel folds) update directly, as before.
#define EX_TW 16
#define EX_HS 512
#define EX_PROBES 8
__global__ __launch_bounds__(64 * EX_TW) void k_extent_blk(ExtentArgs a)
{
    if (ctk_guard_bad(a.guard)) return;
    __shared__ int32_t hk[EX_HS], hlo[EX_HS], hhi[EX_HS];
    for (int s = (int)threadIdx.x; s < EX_HS; s += 64 * EX_TW) { hk[s] = 0; hlo[s] = INT32_MAX; hhi[s] = INT32_MIN; }
    __syncthreads();
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int64_t T = a.T;
    const int64_t t = (int64_t)blockIdx.x * EX_TW + wv;
    int32_t *tmin = a.ext, *tmax = a.ext + a.n_labels + 1;
    if (t < T) {
        const uint32_t n = a.ncomp[t], cb = a.cprefix[t];
        const int32_t tg = (int32_t)(a.t_begin + t);
        int ylo = 0x7fffffff, yhi = -1;                     // rows that hold pixels of complex components
        for (uint32_t c = lane; c < n; c += 64) {
            int32_t l;
            if (a.lab) {
                l = a.lab[cb + c];
                const uint2 bq = *reinterpret_cast<const uint2 *>(a.box + 4 * (int64_t)(cb + c));
                if (l > 0) l = comp_final_label(a.fold, l, tg, (int)(bq.x & 0xffffu), (int)(bq.x >> 16), (int)(bq.y & 0xffffu), (int)(bq.y >> 16));
                a.comp_label_w[cb + c] = l;
                if (l < 0) { ylo = min(ylo, (int)(bq.x & 0xffffu)); yhi = max(yhi, (int)(bq.x >> 16)); }
            } else {
                l = a.comp_label[cb + c];
                if (l < 0) { ylo = min(ylo, (int)a.box[4 * (int64_t)(cb + c)]); yhi = max(yhi, (int)a.box[4 * (int64_t)(cb + c) + 1]); }
            }
            if (l > 0) {
                uint32_t s = ((uint32_t)l * 2654435761u) >> 16;
                bool placed = false;
                for (int p = 0; p < EX_PROBES; p++, s++) {
                    const int32_t old = atomicCAS(&hk[s & (EX_HS - 1)], 0, l);
                    if (old == 0 || old == l) { atomicMin(&hlo[s & (EX_HS - 1)], tg); atomicMax(&hhi[s & (EX_HS - 1)], tg); placed = true; break; }
                }
                if (!placed) ext_update(tmin, tmax, l, tg);                      // (a crowded hash: straight to memory)
            }
        }
        for (int o = 32; o > 0; o >>= 1) { ylo = min(ylo, __shfl_xor(ylo, o)); yhi = max(yhi, __shfl_xor(yhi, o)); }
        if (yhi >= 0) {                                                          // (wave-uniform) pixels of complex components, folded one by one
            __threadfence_block();                                               // (the labels this wave just stored are read back below, by other lanes)
            const uint32_t *rc = a.run_comp + a.run_base[t];
            for (int y = ylo; y <= yhi && y < a.ny; y++) {
                const uint64_t *mw = a.mask + ((int64_t)t * a.ny + y) * a.W;
                for_each_fg_pixel_in_row(mw, a.W, a.rowstart[(int64_t)t * a.ny + y], [&](int x, uint32_t run) {
                    const int32_t l = a.comp_label_w[cb + rc[run]];
                    if (l < 0) {
                        const int32_t fl = fold_pixel(a.fold, -l, tg, y, x);
                        ext_update(tmin, tmax, fl, tg);
                    }
                });
            }
        }
    }
    __syncthreads();
    for (int s = (int)threadIdx.x; s < EX_HS; s += 64 * EX_TW) {
        const int32_t l = hk[s];
        if (l == 0) continue;
        const int32_t lo = hlo[s], hi = hhi[s];
        if (lo < __hip_atomic_load(&tmin[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&tmin[l], lo);
        if (hi > __hip_atomic_load(&tmax[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&tmax[l], hi);
    }
}

// ------------------------------------------------------------------------------------------------
// K6  final value of every run: id if its label survives persistence, 0 otherwise; complex
//     components keep the negative fresh label (folded per pixel in k_relabel).
//     mode 0: final values; mode 1/2: debug -- global 2-D ids before / after the seam merge.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_run_values(const uint32_t *__restrict__ run_base, const uint32_t *__restrict__ run_comp,
                                                    const uint32_t *__restrict__ cprefix, const int32_t *__restrict__ comp_label,
                                                    const int32_t *__restrict__ ext, int64_t n_labels, int persistence,
                                                    const uint32_t *__restrict__ d_mrep, int64_t comp_id_base, int mode,
                                                    int32_t *__restrict__ run_val,
                                                    // optional: the values again, grouped the way k_relabel_v4 consumes them -- CTK_CV slots per chunk of
                                                    // `rows` rows, so that its workgroups can load them together with their tables (no dependent load)
                                                    const uint32_t *__restrict__ rowstart = nullptr, int ny = 0, int rows = 0,
                                                    int32_t *__restrict__ chunk_vals = nullptr, const uint32_t *guard = nullptr,
                                                    // fused path: the ids that are present and survive persistence are counted here, a slice of
                                                    // the ids per workgroup (t_alive[t]); the counting kernel only adds the T numbers up
                                                    const uint32_t *__restrict__ nlab_ptr = nullptr, uint32_t *__restrict__ t_alive = nullptr)
{
    if (ctk_guard_bad(guard)) return;
    const int t = (int)blockIdx.x;
    if (t_alive) {
        const uint32_t nl = *nlab_ptr, per = (nl + gridDim.x - 1) / gridDim.x;
        const uint32_t l0 = 1u + (uint32_t)t * per, l1 = min(nl + 1u, l0 + per);
        uint32_t v = 0;
        for (uint32_t l = l0 + threadIdx.x; l < l1; l += blockDim.x) {
            const int64_t lo = ext[l], hi = ext[n_labels + 1 + l];
            v += (hi >= lo && hi - lo + 1 >= persistence) ? 1u : 0u;
        }
        const int tot = __syncthreads_count((int)(v != 0u));             // (per <= 256 in practice: one id per thread; else add up)
        if (per > blockDim.x) {
            __shared__ uint32_t sacc;
            if (threadIdx.x == 0) sacc = 0;
            __syncthreads();
            if (v) atomicAdd(&sacc, v);
            __syncthreads();
            if (threadIdx.x == 0) t_alive[t] = sacc;
        } else if (threadIdx.x == 0) t_alive[t] = (uint32_t)tot;
    }
    const uint32_t rb = run_base[t], n = run_base[t + 1] - rb, cb = cprefix[t];
    __shared__ uint32_t qb[CTK_CV_MAXCHUNK + 1];                        // first run of every chunk
    const int nchunk = chunk_vals ? (ny + rows - 1) / rows : 0;
    if (chunk_vals) {
        for (int q = threadIdx.x; q <= nchunk; q += blockDim.x) qb[q] = q < nchunk ? rowstart[(int64_t)t * ny + (int64_t)q * rows] : n;
        __syncthreads();
    }
    auto emit = [&](uint32_t r, int32_t v) {
        run_val[rb + r] = v;
        if (chunk_vals) {
            int lo = 0, hi = nchunk - 1;                                // last chunk whose first run is <= r
            while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (qb[m] <= r) lo = m; else hi = m - 1; }
            const uint32_t pos = r - qb[lo];
            if (pos < CTK_CV) chunk_vals[((int64_t)t * nchunk + lo) * CTK_CV + pos] = v;
        }
    };
    if (mode == 0) {
        // Four runs per thread and step, level by level: run -> component -> label -> time extent is a chain of dependent loads,
        // and a plane's ~500 runs would walk it two or three times in a row with one run per thread and step.  Nothing around the
        // loads is conditional (indices clamped instead).
        if (n == 0u) return;
        constexpr int RV = 4;
        for (uint32_t r0 = threadIdx.x; r0 < n; r0 += RV * blockDim.x) {
            uint32_t c[RV];
            int32_t l[RV], e0[RV], e1[RV];
#pragma unroll
            for (int j = 0; j < RV; j++) c[j] = run_comp[rb + min(r0 + (uint32_t)j * blockDim.x, n - 1u)];
#pragma unroll
            for (int j = 0; j < RV; j++) l[j] = comp_label[cb + c[j]];
#pragma unroll
            for (int j = 0; j < RV; j++) { const int32_t lc = max(l[j], 0); e0[j] = ext[lc]; e1[j] = ext[n_labels + 1 + lc]; }
#pragma unroll
            for (int j = 0; j < RV; j++) {
                const uint32_t r = r0 + (uint32_t)j * blockDim.x;
                if (r >= n) break;
                int32_t v = l[j];
                if (v > 0 && (int64_t)e1[j] - (int64_t)e0[j] + 1 < persistence) v = 0;
                emit(r, v);
            }
        }
        return;
    }
    for (uint32_t r = threadIdx.x; r < n; r += blockDim.x) {
        const uint32_t c = run_comp[rb + r];
        emit(r, mode == 1 ? (int32_t)(comp_id_base + cb + c + 1) : (int32_t)(comp_id_base + cb + d_mrep[cb + c] + 1));
    }
}

// ------------------------------------------------------------------------------------------------
// K7  relabel pass: writes the int32 flag slab ONCE.       (contrack/contrack.py:776-791 payload)
// One wave per row.  Lane l writes pixel 64k+l (256 B coalesced stores); a pixel's run index is
// rowstart + (number of run starts at or left of it) - 1, from popcounts on the mask words.
// ------------------------------------------------------------------------------------------------
struct RelabelArgs {
    const uint64_t *mask;
    const uint16_t *wstart;
    const uint32_t *rowstart;
    const uint32_t *run_base;
    const int32_t *run_val;
    const int32_t *ext;
    int64_t n_labels;
    int persistence;
    int64_t t_begin;
    FoldArgs fold;
    int32_t *flag;
    uint32_t *counters;
    int64_t nrows;
    int ny, nx, W;
    const int32_t *chunk_vals;     // [T][nchunk][CTK_CV] (k_run_values) or nullptr
    const uint32_t *guard;         // see ctk_guard_bad
    int plain_stores;              // experiment: plain instead of non-temporal stores
    int xcd_remap;                 // experiment: every XCD streams one contiguous eighth of the slab (xcd_chunk)
    int fast_zero;                 // k_relabel_v5: a chunk without a run is written as zeros straight from registers (no LDS image)
    int tab_batched;               // k_relabel_v5: the chunk's tables in one round of unconditional loads (launches below ~200 000 workgroups)
};

// fast path (nx % 4 == 0, 16-byte aligned flag): one workgroup per (timestep, 16 rows).  The rows' mask
// words, per-word run prefixes, row starts and the final values of their runs are staged in LDS first, so the
// store stream has no dependent global loads: one lane per 4 consecutive pixels, non-temporal int4 stores.
// rb rows per workgroup and rvcap staged run values are launch parameters (dynamic LDS:
// rb*W*8 + rb*W*2 (padded to 8) + (rb+1)*4 (padded to 8) + rvcap*4 bytes).
__global__ __launch_bounds__(256) void k_relabel_v4(RelabelArgs a, int rb, int rvcap)
{
    if (ctk_guard_bad(a.guard)) return;
    const int ny = a.ny, nx = a.nx, W = a.W;
    const int nchunk = (ny + rb - 1) / rb;
    const int t = (int)(blockIdx.x / (unsigned)nchunk), y0 = (int)(blockIdx.x - (unsigned)t * nchunk) * rb, tid = (int)threadIdx.x;
    const int rows = min(rb, ny - y0);
    const int64_t row0 = (int64_t)t * ny + y0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *mrow = reinterpret_cast<uint64_t *>(smem);
    uint16_t *wst = reinterpret_cast<uint16_t *>(smem + (size_t)rb * W * 8);
    uint32_t *rst = reinterpret_cast<uint32_t *>(smem + (size_t)rb * W * 8 + (((size_t)rb * W * 2 + 7) & ~(size_t)7));
    int32_t *rvs = reinterpret_cast<int32_t *>(reinterpret_cast<unsigned char *>(rst) + ((((size_t)rb + 1) * 4 + 7) & ~(size_t)7));
    const uint32_t trun = a.run_base[t + 1] - a.run_base[t];
    for (int i = tid; i < rows * W; i += 256) { mrow[i] = a.mask[row0 * W + i]; wst[i] = a.wstart[row0 * W + i]; }
    for (int i = tid; i <= rows; i += 256) rst[i] = (y0 + i < ny) ? a.rowstart[row0 + i] : trun;
    // the chunk's run values in chunk order (k_run_values): loaded together with the tables -- one round trip, one barrier
    if (a.chunk_vals && tid >= 256 - CTK_CV) rvs[tid - (256 - CTK_CV)] = a.chunk_vals[(int64_t)blockIdx.x * CTK_CV + (tid - (256 - CTK_CV))];
    __syncthreads();
    const uint32_t r0 = rst[0], nr = rst[rows] - r0;
    const int32_t *rvg = a.run_val + a.run_base[t] + r0;
    const bool staged = nr <= (uint32_t)rvcap;
    if (!(a.chunk_vals && nr <= (uint32_t)CTK_CV)) {                    // more runs than the chunk-ordered copy holds (or no copy)
        if (staged) for (uint32_t i = tid; i < nr; i += 256) rvs[i] = rvg[i];
        __syncthreads();
    }
    const int n4 = nx >> 2, total = rows * n4;
    i32x4 *dst = reinterpret_cast<i32x4 *>(a.flag + row0 * (int64_t)nx);
    bool z = false;
    // (row, column) of this lane's slot, advanced by 256 slots per step without dividing (integer division costs more than
    // the rest of the loop body)
    int r = tid / n4, c = tid - r * n4;
    const int dr = 256 / n4, dc = 256 - dr * n4;
    for (int i = tid; i < total; i += 256, r += dr, c += dc) {
        if (c >= n4) { c -= n4; r++; }
        const int x = c << 2, w = x >> 6, xb = x & 63;
        const uint64_t m = mrow[r * W + w];
        const uint32_t nib = (uint32_t)(m >> xb) & 0xfu;
        i32x4 out = (i32x4)(0);
        if (nib) {
            const uint64_t cin = (w > 0) ? (mrow[r * W + w - 1] >> 63) : 0ull;
            const uint64_t st = m & ~((m << 1) | cin);
            const uint32_t base = rst[r] - r0 + wst[r * W + w];
            int32_t v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[j] = 0;
                if ((nib >> j) & 1u) {
                    const int bit = xb + j;
                    const uint64_t below = (bit == 63) ? FULL64 : ((1ull << (bit + 1)) - 1ull);
                    const uint32_t k = base + (uint32_t)__popcll(st & below) - 1u;
                    int32_t val = staged ? rvs[k] : rvg[k];
                    if (val < 0) {                                              // complex component: fold this pixel
                        const int32_t fl = fold_pixel(a.fold, -val, (int32_t)(a.t_begin + t), y0 + r, x + j);
                        val = ((int64_t)a.ext[a.n_labels + 1 + fl] - (int64_t)a.ext[fl] + 1 < a.persistence) ? 0 : fl;
                    }
                    v[j] = val;
                }
            }
            out.x = v[0]; out.y = v[1]; out.z = v[2]; out.w = v[3];
        }
        __builtin_nontemporal_store(out, dst + i);                               // (the chunk's rows are contiguous: slot i)
        z |= (out.x == 0) | (out.y == 0) | (out.z == 0) | (out.w == 0);
    }
    if (__ballot(z) && lane_id() == 0) ctk_zf_set(a.counters, blockIdx.x * 4u + (threadIdx.x >> 6));
}

// Word-centric form of the fast path: the same tables and launch geometry as k_relabel_v4 above, but the chunk's flag values are
// assembled in LDS (dynamic LDS: the tables + rvcap * 4 + rows * nx * 4 bytes) and stored from there.  Used while that image
// leaves room for eight workgroups per CU (1 degree: 11 rows = 15.8 KB; 0.25 degree: 2 rows = 11.5 KB); the 8-row chunks of
// slabs with millions of chunks stay with k_relabel_v4.
#ifdef CTK_PHASE_TIMING
__device__ unsigned long long g_rel_t[16];
#define REL_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 + 8) g_rel_t[k] = wall_clock64(); } while (0)
__device__ unsigned long long g_rel_acc[4096];       // [0..1023] sums of (end - entry) of the workgroups b with b % 1024 == slot, [1024..] their maxima
#define REL_IN() const unsigned long long rel_t_in = wall_clock64()
#define REL_OUT() do { if (threadIdx.x == 0) { const unsigned long long d_ = wall_clock64() - rel_t_in; atomicAdd(&g_rel_acc[blockIdx.x & 1023u], d_); atomicMax(&g_rel_acc[1024 + (blockIdx.x & 1023u)], d_); } } while (0)
#else
#define REL_IN() do { } while (0)
#define REL_OUT() do { } while (0)
#define REL_MARK(k) do { } while (0)
#endif
template <int TH /* threads: 256; 512 / 1024 for tall chunks (fewer stores per lane at the same number of workgroups) */>
__device__ __forceinline__ void relabel_v5_body(const RelabelArgs &a, int rb, int rvcap, int sub /* rows per LDS image: rb, or less for tall chunks */)
{
    REL_IN();
    REL_MARK(0);
    if (ctk_guard_bad(a.guard)) return;
    REL_MARK(1);
    const int ny = a.ny, nx = a.nx, W = a.W;
    const int nchunk = (ny + rb - 1) / rb;
    const unsigned bid = xcd_chunk(blockIdx.x, gridDim.x, a.xcd_remap);
    const int t = (int)(bid / (unsigned)nchunk), y0 = (int)(bid - (unsigned)t * nchunk) * rb, tid = (int)threadIdx.x;
    const int rows = min(rb, ny - y0);
    const int64_t row0 = (int64_t)t * ny + y0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *mrow = reinterpret_cast<uint64_t *>(smem);
    uint16_t *wst = reinterpret_cast<uint16_t *>(smem + (size_t)rb * W * 8);
    uint32_t *rst = reinterpret_cast<uint32_t *>(smem + (size_t)rb * W * 8 + (((size_t)rb * W * 2 + 7) & ~(size_t)7));
    int32_t *rvs = reinterpret_cast<int32_t *>(reinterpret_cast<unsigned char *>(rst) + ((((size_t)rb + 1) * 4 + 7) & ~(size_t)7));
    const uint32_t trun = a.run_base[t + 1] - a.run_base[t];
    // The tables of the chunk: mask words, run prefixes, row starts and the chunk's run values in chunk order (k_run_values) --
    // ONE round trip, one barrier.  Every load is issued before the first is used and none stands under a per-lane condition
    // (indices clamped, the LDS stores masked): written as three guarded loops, hipcc waited for all outstanding loads behind
    // each of them -- three trips to L2 in a row at the start of every workgroup.
    if (a.tab_batched) {
        const int nw = rows * W;
        const int iw = min(tid, nw - 1), ir = min(tid, rows), irc = min(ir, ny - 1 - y0);
        const uint64_t m0 = a.mask[row0 * W + iw];
        const uint16_t w0 = a.wstart[row0 * W + iw];
        const uint32_t rs0 = a.rowstart[row0 + irc];
        int32_t cv0 = 0;
        if (a.chunk_vals) cv0 = a.chunk_vals[(int64_t)bid * CTK_CV + max(tid - (TH - CTK_CV), 0)];       // (uniform condition, last load)
        if (tid < nw) { mrow[tid] = m0; wst[tid] = w0; }
        if (tid <= rows) rst[tid] = (y0 + ir < ny) ? rs0 : trun;
        if (a.chunk_vals && tid >= TH - CTK_CV) rvs[tid - (TH - CTK_CV)] = cv0;
        for (int i = tid + TH; i < nw; i += TH) { mrow[i] = a.mask[row0 * W + i]; wst[i] = a.wstart[row0 * W + i]; }      // (chunks of more than TH words)
        for (int i = tid + TH; i <= rows; i += TH) rst[i] = (y0 + i < ny) ? a.rowstart[row0 + i] : trun;
    } else {
        // (launches of some hundred thousand workgroups and more -- BASELINE configs[2]: 1.3 M chunks of eight rows -- are 2.7 % faster
        // with the three guarded loops: measured, alternating, NOTES round 4)
        for (int i = tid; i < rows * W; i += TH) { mrow[i] = a.mask[row0 * W + i]; wst[i] = a.wstart[row0 * W + i]; }
        for (int i = tid; i <= rows; i += TH) rst[i] = (y0 + i < ny) ? a.rowstart[row0 + i] : trun;
        if (a.chunk_vals && tid >= TH - CTK_CV) rvs[tid - (TH - CTK_CV)] = a.chunk_vals[(int64_t)bid * CTK_CV + (tid - (TH - CTK_CV))];
    }
    __syncthreads();
    const uint32_t r0 = rst[0], nr = rst[rows] - r0;
    if (nr == 0u && a.fast_zero) {
        // no run in the chunk (a third of the 6-row chunks of a 0.25 deg slab): zeros straight from registers -- no LDS image, no
        // further barrier
        i32x4 *dst = reinterpret_cast<i32x4 *>(a.flag + row0 * (int64_t)nx);
        const int total = rows * (nx >> 2);
        for (int i = tid; i < total; i += TH) __builtin_nontemporal_store((i32x4)(0), dst + i);
        if (lane_id() == 0) ctk_zf_set(a.counters, blockIdx.x * (unsigned)(TH / 64) + (threadIdx.x >> 6));
        return;
    }
    REL_MARK(2);
    const int32_t *rvg = a.run_val + a.run_base[t] + r0;
    const bool staged = nr <= (uint32_t)rvcap;
    if (!(a.chunk_vals && nr <= (uint32_t)CTK_CV)) {                    // more runs than the chunk-ordered copy holds (or no copy)
        if (staged) for (uint32_t i = tid; i < nr; i += TH) rvs[i] = rvg[i];
        __syncthreads();
    }
    // Word-centric: the chunk's flag values are assembled in LDS and stored from there, `sub` rows at a time.
    //   A  zero the LDS image (16-byte LDS stores, no mask decoding)
    //   B  one thread per mask word that holds foreground (a tenth of the words): every maximal piece of set bits belongs to one
    //      run, whose value is written over the piece's pixels
    //   C  the store stream: LDS -> 16 bytes per lane, 1 KB contiguous per wave instruction, non-temporal
    // Decoding the mask per four-pixel slot instead (k_relabel_v4) costs some 500 VALU instructions per wave, the foreground branch
    // being taken by a whole wave whenever one of its 64 slots needs it: that kernel is bound by those, not by HBM.  The barriers
    // wait for LDS traffic only (no global store precedes the first two; the third lets the image be zeroed again).
    const int n4 = nx >> 2;
    int32_t *outv = reinterpret_cast<int32_t *>(smem + ((((size_t)(reinterpret_cast<unsigned char *>(rvs) - smem) + (size_t)rvcap * 4) + 15) & ~(size_t)15));
    i32x4 *outv4 = reinterpret_cast<i32x4 *>(outv);
    bool z = false;
    const int tail = nx - (W - 1) * 64;                                       // valid bits of a row's last word
    for (int s0 = 0; s0 < rows; s0 += sub) {
        const int srows = min(sub, rows - s0), total = srows * n4;
        i32x4 *dst = reinterpret_cast<i32x4 *>(a.flag + (row0 + s0) * (int64_t)nx);
        if (s0 == 0) REL_MARK(3);
        for (int i = tid; i < total; i += TH) outv4[i] = (i32x4)(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s0 == 0) REL_MARK(4);
        const int nw = srows * W;
        auto decode = [&](auto staged_c) {
        constexpr bool STAGED = decltype(staged_c)::value;
        for (int k = tid; k < nw; k += TH) {
            const int idx = s0 * W + k;                                         // word of the chunk
            const uint64_t m = mrow[idx];
            const int rr = k / W, w = k - rr * W, r = s0 + rr;
            const uint64_t valid = (w == W - 1 && tail < 64) ? ((1ull << tail) - 1ull) : FULL64;
            if (m != valid) z = true;                                           // a background pixel in this word
            if (m == 0ull) continue;
            const uint64_t cin = (w > 0) ? (mrow[idx - 1] >> 63) : 0ull;
            const uint64_t st = m & ~((m << 1) | cin);
            const uint32_t base = rst[r] - r0 + wst[idx];
            int32_t *orow = outv + rr * nx + w * 64;
            uint64_t mm = m;
            while (mm) {
                const int b = __builtin_ctzll(mm);
                const uint64_t sh = mm >> b;
                const int n = (~sh == 0ull) ? 64 : __builtin_ctzll(~sh);
                const uint64_t below = (b + 1 >= 64) ? FULL64 : ((1ull << (b + 1)) - 1ull);     // bits 0..b
                const uint32_t kk = base + (uint32_t)__popcll(st & below) - 1u;
                int32_t val;
                if constexpr (STAGED) val = rvs[kk]; else val = rvg[kk];
                if (val > 0) {                                                      // head up to a multiple of four, 16-byte LDS stores, tail
                    int q = b;
                    const int e = b + n;
                    for (; q < e && (q & 3); q++) orow[q] = val;
                    const i32x4 v4 = (i32x4)(val);
                    for (; q + 4 <= e; q += 4) *reinterpret_cast<i32x4 *>(orow + q) = v4;
                    for (; q < e; q++) orow[q] = val;
                } else if (val == 0) {
                    z = true;                                                       // filtered out: the zeros are there already
                } else {                                                            // complex component: fold pixel by pixel
                    for (int q = 0; q < n; q++) {
                        const int32_t fl = fold_pixel(a.fold, -val, (int32_t)(a.t_begin + t), y0 + r, w * 64 + b + q);
                        const int32_t v = ((int64_t)a.ext[a.n_labels + 1 + fl] - (int64_t)a.ext[fl] + 1 < a.persistence) ? 0 : fl;
                        z |= v == 0;
                        orow[b + q] = v;
                    }
                }
                mm = (n >= 64 - b) ? 0ull : (mm & ~(((1ull << n) - 1ull) << b));
            }
        }
        };
        // (two copies of the loop: with `staged ? rvs[kk] : rvg[kk]` in one loop hipcc selects the ADDRESS and reads it with a flat load, which
        // counts in vmcnt too -- the decode of a chunk's second image then waited for the stores of its first to drain)
        if (staged) decode(std::true_type{}); else decode(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s0 == 0) REL_MARK(5);
        if (a.plain_stores) { for (int i = tid; i < total; i += TH) dst[i] = outv4[i]; }
        else for (int i = tid; i < total; i += TH) __builtin_nontemporal_store(outv4[i], dst + i);    // (the rows are contiguous: slot i)
        if (s0 == 0) REL_MARK(6);
        if (s0 + sub < rows) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // the image is zeroed again
        if (s0 == 0) REL_MARK(7);
    }
    REL_OUT();
    REL_MARK(8);
    if (__ballot(z) && lane_id() == 0) ctk_zf_set(a.counters, blockIdx.x * (unsigned)(TH / 64) + (threadIdx.x >> 6));
}

// (80 SGPRs instead of 101: eight workgroups per CU instead of six -- CTK_SGPR_8WAVES)
template <int TH>
__global__ __launch_bounds__(TH) CTK_SGPR_8WAVES void k_relabel_v5(RelabelArgs a, int rb, int rvcap, int sub) { relabel_v5_body<TH>(a, rb, rvcap, sub); }
template <int TH>
__global__ __launch_bounds__(TH) void k_relabel_v5_allsgpr(RelabelArgs a, int rb, int rvcap, int sub) { relabel_v5_body<TH>(a, rb, rvcap, sub); }      // (A/B: CTK_RELABEL_SGPR=0)
// the same code under another name: the launches that time the chunk -> XCD mapping once per shape (tune_relabel, ctk_api.hip) -- kept
// apart so that a profile's statistics of k_relabel_v5 are those of the passes (as k_threshold_probe does for the mask placement check)
template <int TH>
__global__ __launch_bounds__(TH) CTK_SGPR_8WAVES void k_relabel_probe(RelabelArgs a, int rb, int rvcap, int sub) { relabel_v5_body<TH>(a, rb, rvcap, sub); }

__global__ __launch_bounds__(256) CTK_SGPR_8WAVES void k_relabel(RelabelArgs a)
{
    if (ctk_guard_bad(a.guard)) return;
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int W = a.W, nx = a.nx;
    uint32_t wrote_zero = 0;
    for (int64_t row = wave; row < a.nrows; row += nwaves) {
        const int64_t t = row / a.ny;
        const int y = (int)(row - t * a.ny);
        const uint64_t *mw = a.mask + row * W;
        int32_t *dst = a.flag + row * (int64_t)nx;
        const int32_t *rv = a.run_val + a.run_base[t];
        uint32_t base = a.rowstart[row];
        uint64_t carry = 0;
        for (int w0 = 0; w0 < W; w0 += WAVE) {
            const int wn = min(WAVE, W - w0);
            uint64_t m = (lane < wn) ? mw[w0 + lane] : 0ull;
            uint64_t pm = shfl_up_u64(m, 1);
            uint64_t st = m & ~((m << 1) | ((lane == 0) ? carry : (pm >> 63)));
            uint32_t n = (uint32_t)__popcll(st);
            uint32_t inc = wave_incl_scan_u32(n);
            uint32_t before = base + inc - n;
            for (int k = 0; k < wn; k++) {
                const int x = (w0 + k) * 64 + lane;
                uint64_t mk = shfl_u64(m, k);
                int32_t v = 0;
                if (mk != 0ull) {                                       // wave-uniform
                    uint64_t sk = shfl_u64(st, k);
                    uint32_t bk = __shfl(before, k);
                    if ((mk >> lane) & 1ull) {
                        uint64_t below = (lane == 63) ? FULL64 : ((1ull << (lane + 1)) - 1ull);
                        v = rv[bk + (uint32_t)__popcll(sk & below) - 1u];
                        if (v < 0) {                                    // complex component: fold this pixel
                            int32_t fl = fold_pixel(a.fold, -v, (int32_t)(a.t_begin + t), y, x);
                            v = ((int64_t)a.ext[a.n_labels + 1 + fl] - (int64_t)a.ext[fl] + 1 < a.persistence) ? 0 : fl;
                        }
                    }
                }
                if (x < nx) { dst[x] = v; wrote_zero |= (v == 0); }
            }
            base += __shfl(inc, WAVE - 1);
            carry = shfl_u64(m, wn - 1) >> 63;
        }
    }
    if (__ballot(wrote_zero != 0) && lane == 0) ctk_zf_set(a.counters, blockIdx.x * 4u + (threadIdx.x >> 6));
}

// ------------------------------------------------------------------------------------------------
// K8  number of ids that are present and survive persistence
// ------------------------------------------------------------------------------------------------
// Fused one-call path: what the host wants to know about the pass it did not watch, written by the counting kernel into pinned
// host memory (scal[CTK_AM_*]); the host validates it after its only synchronisation.
struct AsyncMail {
    uint32_t *scal;                 // nullptr: not the fused path
    uint32_t stamp;                 // written to scal[CTK_AM_DONE] after everything else: the host polls for it
    const uint32_t *nlab_ptr;       // number of fresh 3-D labels (= ids to count; n_labels is then only the offset of ext's second half)
    const uint32_t *nc_ptr;         // cprefix[T]
    const uint32_t *changed;        // [passes][CTK_CHG_SLOTS]
    const uint32_t *ambig;
    const uint32_t *rec_cnt;        // [T] candidate group records per timestep
    const uint32_t *t_nops;         // [T] relabel operations of the clusters that start in timestep t
    const uint32_t *pair_cnt;       // [T] grouped co-occurrence records per timestep
    const uint32_t *t_alive;        // [T] surviving ids counted by k_run_values (nullptr: the counting kernel walks the ids itself)
    int64_t T;
    int passes;
};
#define CTK_AM_COUNTERS 0           // .. + CTK_CNT_N
#define CTK_AM_NC       16
#define CTK_AM_NLAB     17
#define CTK_AM_NPAIRS   18          // grouped co-occurrence records (sum of pair_cnt)
#define CTK_AM_NCAND    19
#define CTK_AM_AMBIG    20
#define CTK_AM_CONV     21          // first filter pass that changed nothing, + 1; 0 = none of the passes launched
#define CTK_AM_DONE     22          // written last: the block is complete
#define CTK_AM_WORDS    32

// called by the first 64 threads of a workgroup (one word per lane: the stores to host memory leave together)
__device__ inline void async_mail_write(const AsyncMail &m, const uint32_t *counters, uint32_t ncand, uint32_t nops, uint32_t npairs = 0)
{
    const int lane = (int)threadIdx.x;
    if (lane >= 64) return;
    uint32_t v = 0;
    bool has = false;
    if (lane == CTK_CNT_NOPS) { v = nops; has = true; }
    else if (lane < CTK_CNT_N) { v = __hip_atomic_load(&counters[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); has = true; }
    else if (lane == CTK_AM_NC) { v = *m.nc_ptr; has = true; }
    else if (lane == CTK_AM_NLAB) { v = *m.nlab_ptr; has = true; }
    else if (lane == CTK_AM_NCAND) { v = ncand; has = true; }
    else if (lane == CTK_AM_NPAIRS) { v = npairs; has = true; }
    else if (lane == CTK_AM_AMBIG) { v = *m.ambig; has = true; }

    // first filter pass that changed nothing: lane k looks at pass k (and k + 64, ...)
    uint32_t conv = 0xffffffffu;
    for (int k = lane; k < m.passes; k += 64) {
        bool any = false;
        for (int j = 0; j < 64; j++) any |= m.changed[k * 64 + j] != 0u;
        if (!any) { conv = (uint32_t)k; break; }
    }
    for (int o = 32; o > 0; o >>= 1) conv = min(conv, (uint32_t)__shfl_xor((int)conv, o));
    if (lane == CTK_AM_CONV) { v = conv == 0xffffffffu ? 0u : conv + 1u; has = true; }
    if (has) m.scal[lane] = v;
    // the block is complete (this wave wrote all of it, and the two words of the older mailbox before): then the stamp
    __threadfence_system();
    if (lane == CTK_AM_DONE) __hip_atomic_store(&m.scal[CTK_AM_DONE], m.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_count_alive(const int32_t *__restrict__ ext, int64_t n_labels, int persistence,
                                                     uint32_t *counters, uint32_t *mail, AsyncMail am)
{
    if (ctk_guard_bad(am.scal ? counters : nullptr)) {
        if (blockIdx.x == 0) async_mail_write(am, counters, 0u, __hip_atomic_load(&counters[CTK_CNT_NOPS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));      // (a poisoned pass: the op slots the shared tail was asked for, so that the host can size the next pass)
        return;
    }
    const int64_t nl = am.scal ? (int64_t)*am.nlab_ptr : n_labels;
    uint32_t v = 0;
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; l <= nl; l += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = ext[l], hi = ext[n_labels + 1 + l];
        v += (hi >= lo && hi - lo + 1 >= persistence) ? 1u : 0u;
    }
    uint32_t s = wave_sum_u32(v);
    if (lane_id() == 0 && s) atomicAdd(&counters[CTK_CNT_ALIVE], s);
    if (am.scal) {                                          // the mail's sums over the timesteps: every workgroup its share
        uint32_t nc = 0, no = 0, np = 0;
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < am.T; t += (int64_t)gridDim.x * blockDim.x) { nc += am.rec_cnt[t]; no += am.t_nops[t]; np += am.pair_cnt[t]; }
        nc = wave_sum_u32(nc); no = wave_sum_u32(no); np = wave_sum_u32(np);
        if (lane_id() == 0) {
            if (nc) atomicAdd(&counters[CTK_CNT_SUM_NC], nc);
            if (no) atomicAdd(&counters[CTK_CNT_SUM_NOPS], no);
            if (np) atomicAdd(&counters[CTK_CNT_SUM_NP], np);
        }
    }
    // the last workgroup to finish publishes the results in pinned host memory (no copy command)
    __shared__ bool last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&counters[CTK_CNT_TICKET], 1u) == gridDim.x - 1;
    __syncthreads();
    if (last) {
        __shared__ uint32_t zw[4];
        const uint32_t zv = ctk_zf_mine(counters, (int)threadIdx.x, 256);
        const bool zany = __ballot(zv != 0u) != 0ull;
        if (lane_id() == 0) zw[threadIdx.x >> 6] = zany ? 1u : 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t z = zw[0] | zw[1] | zw[2] | zw[3];
            counters[CTK_CNT_WROTE_ZERO] = z;
            mail[0] = __hip_atomic_load(&counters[CTK_CNT_ALIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mail[1] = z;
        }
        __syncthreads();
    }
    if (last && am.scal)
        async_mail_write(am, counters, __hip_atomic_load(&counters[CTK_CNT_SUM_NC], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                         __hip_atomic_load(&counters[CTK_CNT_SUM_NOPS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                         __hip_atomic_load(&counters[CTK_CNT_SUM_NP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// the same in ONE workgroup (up to a few hundred thousand ids): no atomics, no ticket
__global__ __launch_bounds__(1024) void k_count_alive_1(const int32_t *__restrict__ ext, int64_t n_labels, int persistence,
                                                        uint32_t *counters, uint32_t *mail, AsyncMail am)
{
    if (ctk_guard_bad(am.scal ? counters : nullptr)) {
        async_mail_write(am, counters, 0u, __hip_atomic_load(&counters[CTK_CNT_NOPS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));      // (a poisoned pass: the op slots the shared tail was asked for, so that the host can size the next pass)
        return;
    }
    uint32_t v = 0, nc = 0, no = 0, np = 0;
    if (am.scal) {
        for (int64_t t = threadIdx.x; t < am.T; t += 1024) { nc += am.rec_cnt[t]; no += am.t_nops[t]; np += am.pair_cnt[t]; }
    }
    const int64_t nl = am.scal ? (int64_t)*am.nlab_ptr : n_labels;
    if (am.scal && am.t_alive) { for (int64_t t = threadIdx.x; t < am.T; t += 1024) v += am.t_alive[t]; }
    else for (int64_t l = threadIdx.x + 1; l <= nl; l += 1024) {
        const int64_t lo = ext[l], hi = ext[n_labels + 1 + l];
        v += (hi >= lo && hi - lo + 1 >= persistence) ? 1u : 0u;
    }
    __shared__ uint32_t sm[16], sn[16], so[16], sp[16];
    const uint32_t s = wave_sum_u32(v), s2 = wave_sum_u32(nc), s3 = wave_sum_u32(no), s4 = wave_sum_u32(np);
    if (lane_id() == 0) { sm[threadIdx.x >> 6] = s; sn[threadIdx.x >> 6] = s2; so[threadIdx.x >> 6] = s3; sp[threadIdx.x >> 6] = s4; }
    __syncthreads();
    uint32_t tot = 0, ncand = 0, nops = 0, npairs = 0;
    for (int i = 0; i < 16; i++) { tot += sm[i]; ncand += sn[i]; nops += so[i]; npairs += sp[i]; }
    __shared__ uint32_t zw[16];
    {
        const uint32_t zv = ctk_zf_mine(counters, (int)threadIdx.x, 1024);
        const bool zany = __ballot(zv != 0u) != 0ull;
        if (lane_id() == 0) zw[threadIdx.x >> 6] = zany ? 1u : 0u;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t z = 0;
        for (int i = 0; i < 16; i++) z |= zw[i];
        counters[CTK_CNT_ALIVE] = tot;
        counters[CTK_CNT_WROTE_ZERO] = z;
        mail[0] = tot;
        mail[1] = z;
    }
    if (am.scal) {
        __syncthreads();                                   // (counters[CTK_CNT_ALIVE] is one of the words mailed)
        async_mail_write(am, counters, ncand, nops, npairs);
    }
}

// k_count_alive_1 for the fused one-call pass (round 6): the same block of scalars, but every global load of the kernel is issued
// before the first one is used -- the per-timestep sums, the zero flags, the device counters, the filter's change words were four
// dependent rounds of loads behind three barriers (6.8-7.8 us for a few KB; it is the last kernel of the pass: the host waits for its
// stamp).  One barrier.  (The ids that survive were counted per timestep slice by k_run_values: t_alive.)
__global__ __launch_bounds__(1024) void k_count_alive_f(uint32_t *counters, uint32_t *mail, AsyncMail am)
{
    if (ctk_guard_bad(counters)) {
        async_mail_write(am, counters, 0u, __hip_atomic_load(&counters[CTK_CNT_NOPS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));      // (a poisoned pass: the op slots the shared tail was asked for)
        return;
    }
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- all loads ------------------------------------------------------------------------------------------
    uint32_t v = 0, nc = 0, no = 0, np = 0;
    for (int64_t t = tid; t < am.T; t += 1024) { v += am.t_alive[t]; nc += am.rec_cnt[t]; no += am.t_nops[t]; np += am.pair_cnt[t]; }
    const uint32_t zv = ctk_zf_mine(counters, tid, 1024);
    // filter pass k changed something?  word j of pass k is looked at by thread k * 64 + j (passes <= 32: two per thread)
    const int np1 = am.passes;
    const uint32_t ch0 = (tid < np1 * 64) ? am.changed[tid] : 0u, ch1 = (tid + 1024 < np1 * 64) ? am.changed[tid + 1024] : 0u;
    uint32_t cval = 0;                                       // wave 0: what lane `lane` mails besides the sums
    if (wv == 0) {
        if (lane < CTK_CNT_N) cval = __hip_atomic_load(&counters[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (lane == CTK_AM_NC) cval = *am.nc_ptr;
        else if (lane == CTK_AM_NLAB) cval = *am.nlab_ptr;
        else if (lane == CTK_AM_AMBIG) cval = *am.ambig;
    }
    // ---- reductions --------------------------------------------------------------------------------------------
    __shared__ uint32_t sm[16], sn[16], so[16], sp[16], sz[16], sc[32];
    const uint32_t s1 = wave_sum_u32(v), s2 = wave_sum_u32(nc), s3 = wave_sum_u32(no), s4 = wave_sum_u32(np);
    const bool zany = __ballot(zv != 0u) != 0ull, c0 = __ballot(ch0 != 0u) != 0ull, c1 = __ballot(ch1 != 0u) != 0ull;
    if (lane == 0) { sm[wv] = s1; sn[wv] = s2; so[wv] = s3; sp[wv] = s4; sz[wv] = zany ? 1u : 0u; sc[wv] = c0 ? 1u : 0u; sc[wv + 16] = c1 ? 1u : 0u; }
    __syncthreads();
    if (wv != 0) return;
    uint32_t tot = 0, ncand = 0, nops = 0, npairs = 0, z = 0;
    for (int i = 0; i < 16; i++) { tot += sm[i]; ncand += sn[i]; nops += so[i]; npairs += sp[i]; z |= sz[i]; }
    uint32_t conv = 0;                                       // first pass that changed nothing, + 1; 0 = none of those launched
    for (int k = np1 - 1; k >= 0; k--) if (!sc[k]) conv = (uint32_t)k + 1u;
    if (np1 > 32) conv = 0;                                  // (never launched with more: the synchronous path takes over)
    if (lane == 0) { counters[CTK_CNT_ALIVE] = tot; counters[CTK_CNT_WROTE_ZERO] = z; mail[0] = tot; mail[1] = z; }
    // ---- the block of scalars (async_mail_write's layout), one word per lane ---------------------------------------------
    uint32_t out = cval;
    bool has = lane < CTK_CNT_N || lane == CTK_AM_NC || lane == CTK_AM_NLAB || lane == CTK_AM_AMBIG;
    if (lane == CTK_CNT_ALIVE) out = tot;
    else if (lane == CTK_CNT_WROTE_ZERO) out = z;
    else if (lane == CTK_CNT_NOPS) out = nops;
    else if (lane == CTK_AM_NCAND) { out = ncand; has = true; }
    else if (lane == CTK_AM_NPAIRS) { out = npairs; has = true; }
    else if (lane == CTK_AM_CONV) { out = conv; has = true; }
    if (has) am.scal[lane] = out;
    __threadfence_system();
    if (lane == CTK_AM_DONE) __hip_atomic_store(&am.scal[CTK_AM_DONE], am.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// time extents of the ids start empty; the counters of the write stage start at zero
__global__ void k_fill_ext(int32_t *ext, int64_t n_labels, uint32_t *counters)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_labels; i += (int64_t)gridDim.x * blockDim.x) {
        ext[i] = INT32_MAX; ext[n_labels + 1 + i] = INT32_MIN;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < CTK_ZF_SLOTS; i += (int64_t)gridDim.x * blockDim.x) ctk_zf_reset(counters, i);
    if (blockIdx.x == 0 && threadIdx.x == 0) { counters[CTK_CNT_WROTE_ZERO] = 0; counters[CTK_CNT_ALIVE] = 0; counters[CTK_CNT_TICKET] = 0; }
}

// op staging block [CtkOp ops[n]] [int32 next[n]] [int32 hi_label[n], nf used] [int32 first_op[n], nf used] in pinned host memory ->
// device copy of ops + next (same layout) and first[label]
// ... and what k_fill_ext does (one launch on the critical path between the host seam driver and the relabel)
__global__ void k_ops_ingest(const int32_t *__restrict__ staging, int64_t nops, int32_t nf, int32_t *__restrict__ dst, int32_t *__restrict__ op_first,
                             int32_t *__restrict__ ext, int64_t n_labels, uint32_t *counters)
{
    const int64_t words = nops * 9;
    const int32_t *lab = staging + words, *first = lab + nops;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words + nf; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < words) dst[i] = staging[i];
        else op_first[lab[i - words]] = first[i - words];
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_labels; i += (int64_t)gridDim.x * blockDim.x) {
        ext[i] = INT32_MAX; ext[n_labels + 1 + i] = INT32_MIN;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < CTK_ZF_SLOTS; i += (int64_t)gridDim.x * blockDim.x) ctk_zf_reset(counters, i);
    if (blockIdx.x == 0 && threadIdx.x == 0) { counters[CTK_CNT_WROTE_ZERO] = 0; counters[CTK_CNT_ALIVE] = 0; counters[CTK_CNT_TICKET] = 0; }
}

__global__ void k_scatter_i32(const int32_t *__restrict__ where, const int32_t *__restrict__ what, int n, int32_t *__restrict__ dst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[where[i]] = what[i];
}

// mask words -> one byte per pixel (debug / staged parity)
__global__ void k_expand_mask(const uint64_t *__restrict__ mask, int64_t nrows, int nx, int W, uint8_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows * nx; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t row = i / nx;
        int x = (int)(i - row * nx);
        out[i] = (uint8_t)((mask[row * W + (x >> 6)] >> (x & 63)) & 1ull);
    }
}

// halo export: pack {mask row words, rowstart, run_comp} of the LAST timestep into one blob
__global__ void k_copy_u32(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// deterministic synthetic slab for throughput runs (bench only): a sum of 24 drifting smooth waves, tuned
// (numpy emulation) to std ~ 94, ~9 % of the pixels above 160 and ~40 components per 1 deg timestep.
// ------------------------------------------------------------------------------------------------
__global__ void k_synth(float *__restrict__ out, int64_t T, int ny, int nx, uint64_t seed, int64_t t_first)
{
    // grid-stride: a HIP grid carries at most 2^32-1 work-items per dimension
    const int64_t n = T * (int64_t)ny * nx;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int x = (int)(i % nx);
    int y = (int)((i / nx) % ny);
    int64_t t = i / ((int64_t)nx * ny) + t_first;
    float lon = 6.2831853f * (float)x / (float)nx, lat = 3.1415927f * ((float)y / (float)(ny - 1) - 0.5f);
    float tt = (float)t;
    float s = 0.f;
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
#pragma unroll 1
    for (int k = 0; k < 24; k++) {
        z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        int kx = 1 + (int)(z % 12), ky = 1 + (int)((z >> 8) % 9);
        float ph = (float)((z >> 16) % 6283) * 1e-3f, om = ((float)((z >> 32) % 2001) - 1000.f) * 2e-4f;
        float amp = 40.f / (float)(1 + (k >> 3));
        float drift = om * tt;                                      // keep the fast-math arguments small: reduce mod 2 pi
        drift -= 6.2831853f * floorf(drift * 0.15915494f);
        s += amp * __sinf((float)kx * lon + ph + drift) * __cosf((float)ky * lat * 2.f + ph * 0.7f - 0.5f * drift);
    }
    float slow = 0.01f * tt;
    slow -= 6.2831853f * floorf(slow * 0.15915494f);
    out[i] = 35.f + 1.7f * (s * __cosf(lat) + 40.f * __sinf(lat * 2.f + slow));
    }
}


// ------------------------------------------------------------------------------------------------
// What a PLAIN stream of the two streaming kernels' shape reaches on this board (ctk_debug_stream_ceiling; bench.py puts it next to
// the roofline): 16-byte non-temporal stores of zeros over a buffer / 16-byte non-temporal loads ORed into a register.  The write
// kernel k_relabel_v5 cannot be faster than the first, k_threshold_v7 not faster than the second.
// ------------------------------------------------------------------------------------------------
// xcd: 0 the chunks in launch order, 1 one contiguous eighth of the buffer per XCD (xcd_chunk: +10 % for a store stream on every size
// timed, 705 MB ... 8 GB, tools/store_stream.hip)
__global__ __launch_bounds__(256) void k_stream_store(i32x4 *__restrict__ dst, int64_t n16, int xcd)
{
    constexpr int U = 8;
    const int64_t base = (int64_t)xcd_chunk(blockIdx.x, gridDim.x, xcd) * (256 * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; u++) { const int64_t i = base + u * 256; if (i < n16) __builtin_nontemporal_store((i32x4)(0), dst + i); }
}
__global__ __launch_bounds__(256) void k_stream_load(const i32x4 *__restrict__ src, int64_t n16, int32_t *__restrict__ sink)
{
    constexpr int U = 8;
    const int64_t base = (int64_t)blockIdx.x * (256 * U) + threadIdx.x;
    i32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(src + min(base + u * 256, n16 - 1));
    int32_t acc = 0;
#pragma unroll
    for (int u = 0; u < U; u++) acc |= v[u].x | v[u].y | v[u].z | v[u].w;
    if (acc == 0x5a5a5a5a) *sink = acc;                       // (never, in practice: keeps the loads alive)
}

// position-weighted checksum of an int32 array (ctk_checksum_i32_dev): out[0] += sum (uint32)p[i] * (((index0 + i) * golden) | 1),
// out[1] += nonzero elements.  One 64-bit atomic pair per workgroup (integer: the result does not depend on the order).
__global__ __launch_bounds__(256) void k_checksum_i32(const int32_t *__restrict__ p, int64_t n, int64_t index0, unsigned long long *__restrict__ out)
{
    unsigned long long s = 0, nz = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)p[i];
        if (v) { s += (unsigned long long)v * ((((unsigned long long)(index0 + i)) * 0x9E3779B97F4A7C15ull) | 1ull); nz++; }
    }
    __shared__ unsigned long long sm[2][4];
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); nz += __shfl_down(nz, o); }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s; sm[1][threadIdx.x >> 6] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long a = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3], b = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
        if (a) atomicAdd(&out[0], a);
        if (b) atomicAdd(&out[1], b);
    }
}


// ------------------------------------------------------------------------------------------------
// Size-independent properties of a result, checked on the device (ctk_check_flag_dev; slabs no host can hold):
//   (1) flag != 0 only where the compare of contrack.py:665 holds -- evaluated here in float64, independently of the float32
//       threshold adjustment k_threshold uses;
//   (2) ids lie in [1, max_id];
//   (3) time extent of every id (contrack.py:765-772: survivors span >= persistence steps) -- tmin / tmax per id, one look-before
//       atomic pair per horizontal run of an id.
// One wave per row, grid-stride over the T * ny rows.  out: [0] pixels violating (1), [1] pixels violating (2), [2] nonzero pixels.
// ------------------------------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void k_check_flag(const float *__restrict__ anom, const int32_t *__restrict__ flag, int64_t nrows, int ny, int nx,
                                                    const double *__restrict__ thr, int64_t max_id, int32_t *__restrict__ tmin, int32_t *__restrict__ tmax,
                                                    unsigned long long *__restrict__ out)
{
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    unsigned long long v1 = 0, v2 = 0, nz = 0;
    for (int64_t row = wave; row < nrows; row += nwaves) {
        const int32_t t = (int32_t)(row / ny);
        const double th = thr[t];
        const float *a = anom + row * (int64_t)nx;
        const int32_t *f = flag + row * (int64_t)nx;
        for (int x = lane; x < nx; x += WAVE) {
            const int32_t id = f[x];
            if (id == 0) continue;
            nz++;
            if (!cmp_op<OP, double>((double)a[x], th)) v1++;
            if (id < 0 || (int64_t)id > max_id) { v2++; continue; }
            if (x > 0 && f[x - 1] == id) continue;                       // not the head of a run of this id
            if (tmin[id] > t) atomicMin(&tmin[id], t);                   // (a stale look is only looser: at worst a superfluous atomic)
            if (tmax[id] < t) atomicMax(&tmax[id], t);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { v1 += __shfl_down(v1, o); v2 += __shfl_down(v2, o); nz += __shfl_down(nz, o); }
    if (lane == 0) {
        if (v1) atomicAdd(&out[0], v1);
        if (v2) atomicAdd(&out[1], v2);
        if (nz) atomicAdd(&out[2], nz);
    }
}

// out: [3] ids present, [4] ids present with a time extent below `persistence`, [5] largest id present
__global__ __launch_bounds__(256) void k_check_ids(const int32_t *__restrict__ tmin, const int32_t *__restrict__ tmax, int64_t max_id, int persistence,
                                                   unsigned long long *__restrict__ out)
{
    unsigned long long n = 0, nshort = 0, big = 0;
    for (int64_t i = 1 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= max_id; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t a = tmin[i], b = tmax[i];
        if (b < a) continue;
        n++;
        if ((int64_t)b - a + 1 < (int64_t)persistence) nshort++;
        big = (unsigned long long)i;
    }
    for (int o = 32; o > 0; o >>= 1) { n += __shfl_down(n, o); nshort += __shfl_down(nshort, o); const unsigned long long ob = __shfl_down(big, o); big = ob > big ? ob : big; }
    if (lane_id() == 0) {
        if (n) atomicAdd(&out[3], n);
        if (nshort) atomicAdd(&out[4], nshort);
        if (big) atomicMax(&out[5], big);
    }
}

__global__ __launch_bounds__(256) void k_fill_minmax(int32_t *__restrict__ tmin, int32_t *__restrict__ tmax, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { tmin[i] = 0x7fffffff; tmax[i] = -1; }
}
