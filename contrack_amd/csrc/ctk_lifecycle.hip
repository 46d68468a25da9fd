// ctk_lifecycle.hip -- per (time step, flag id) reductions of contrack.run_lifecycle (contrack/contrack.py:798-906):
// size, intensity numerator and the centre-of-mass sums of every flagged contour, one workgroup per time step.
// Included by ctk_api.hip (uses dev_limbs_to_double of ctk_resolve_dev.hip).
//
// The reference loops over time steps and, inside, over np.unique(flag[t]); per label it evaluates
//   areacon      = np.sum(weight_grid[flag == label])                                   (:874)
//   intensitycon = np.sum(weight_grid[...] * variable[...]) / areacon                   (:875-876)
//   center_of_mass(variable * weight_grid, flag, [label])                               (:892)
// and, when the label touches both x = 0 and x = nx-1, rolls the plane so that the contour's western edge
// (the column right of the widest gap in the occupied columns, :882-883) becomes column 0 before taking the
// centre of mass (:884-889).  All of that is four sums per label plus a column-occupancy bit set:
//   area  (exact: integer limbs of the float32 row weights, rounded once)
//   swv   = sum w*v          swvy = sum (w*v)*y          swvx = sum (w*v)*x'      x' = (x - shift) mod nx
// in float64 (LDS atomics: the summation order differs from numpy's, the products are the same).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LC_HASH 1024      // hash slots for the flag ids of one time step (<= LC_NL ids: load factor <= 0.5)
#define LC_NL 512         // distinct flag ids one time step may hold
#define LC_THREADS 512
#define LC_ERR_LABELS 1u  // more than LC_NL ids in one time step
#define LC_ERR_SEAM 2u    // more seam-crossing ids in one time step than column bit sets fit

struct CtkLifeRowDev {
    int32_t t, label, shift, pad;    // shift: roll applied before the centre of mass (-1: none, -2: undefined, single column)
                                     // pad: rows that may hold the id, first | last << 16 (bounds the scans of the exact kernels)
    double area, swv, swvy, swvx;
};

__device__ inline uint32_t lc_hash(int32_t label) { return ((uint32_t)label * 2654435761u) >> 22; }   // 10 bits

// slot of `label` in the table (insert = false: the label is known to be present)
template <bool INSERT>
__device__ inline int lc_slot(int32_t *hkey, int32_t label)
{
    uint32_t s = lc_hash(label) & (LC_HASH - 1);
    for (int probe = 0; probe < LC_HASH; ++probe) {
        const int32_t k = hkey[s];
        if (k == label) return (int)s;
        if (INSERT && k == 0) {
            const int32_t old = atomicCAS(&hkey[s], 0, label);
            if (old == 0 || old == label) return (int)s;
        }
        s = (s + 1) & (LC_HASH - 1);
    }
    return -1;
}

template <typename VT>
__global__ __launch_bounds__(LC_THREADS) void k_lifecycle(const int32_t *__restrict__ flag, const VT *__restrict__ field, int ny, int nx, int nxw,
                                                          int ks, const int64_t *__restrict__ wlo, const int64_t *__restrict__ whi,
                                                          const float *__restrict__ wrow, int wshift, int limb_bits, CtkLifeRowDev *rows,
                                                          unsigned long long cap_rows, unsigned long long *counters,
                                                          // A time step with more ids than the LDS tables hold is redone in several passes, each
                                                          // taking the ids of one residue class: work item i = {t, P, j} -> ids with id mod P == j.
                                                          // work == nullptr: item = time step, all ids.  ovf[item] = 1: the item did not fit.
                                                          const int32_t *__restrict__ work, unsigned char *__restrict__ ovf)
{
    __shared__ int32_t hkey[LC_HASH];
    __shared__ unsigned hedge[LC_HASH / 4];      // per slot one byte: bit 0: id seen at x = 0, bit 1: at x = nx-1
    __shared__ uint16_t hidx[LC_HASH];
    __shared__ int32_t dlabel[LC_NL], dshift[LC_NL], dseam[LC_NL];
    __shared__ long long alo[LC_NL], ahi[LC_NL];
    __shared__ double swv[LC_NL], swvy[LC_NL], swvx[LC_NL];
    __shared__ int seam_idx[64];
    __shared__ int nlab, nseam;
    __shared__ unsigned err;
    __shared__ unsigned long long base;
    extern __shared__ unsigned colbits[];        // [ks][nxw] occupied columns of the seam-crossing ids

    const int tid = threadIdx.x;
    const int64_t t = work ? work[3 * blockIdx.x] : (int64_t)blockIdx.x;
    const uint32_t cP = work ? (uint32_t)work[3 * blockIdx.x + 1] : 1u, cj = work ? (uint32_t)work[3 * blockIdx.x + 2] : 0u;
    const uint32_t npx = (uint32_t)ny * (uint32_t)nx;            // ny, nx <= 65535 (checked by the host)
    const int32_t *fp = flag + t * (int64_t)npx;
    const VT *vp = field + t * (int64_t)npx;
    const bool vec = (npx & 3u) == 0 && (((uintptr_t)flag) & 15u) == 0;      // every plane starts 16-byte aligned

    for (int s = tid; s < LC_HASH; s += LC_THREADS) { hkey[s] = 0; if (s < LC_HASH / 4) hedge[s] = 0; }
    if (tid == 0) { nlab = 0; nseam = 0; err = 0; }
    __syncthreads();

    // four consecutive pixels of the plane (zeros beyond its end)
    auto load4 = [&](uint32_t p0, int32_t l[4]) {
        if (vec) {
            const int4 q = *(const int4 *)(fp + p0);
            l[0] = q.x; l[1] = q.y; l[2] = q.z; l[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) l[k] = (p0 + k < npx) ? fp[p0 + k] : 0;
        }
        if (cP > 1u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if ((uint32_t)l[k] % cP != cj) l[k] = 0;
        }
    };

    // ---- A: the ids present, and which of them touch the two seam columns
    for (uint32_t p0 = (uint32_t)tid * 4; p0 < npx; p0 += LC_THREADS * 4) {
        int32_t l[4];
        load4(p0, l);
        if ((l[0] | l[1] | l[2] | l[3]) == 0) continue;
        int32_t prev = 0;
        int prev_slot = -1;
        uint32_t x = p0 % (uint32_t)nx;
#pragma unroll
        for (int k = 0; k < 4; ++k, x = (x + 1 == (uint32_t)nx) ? 0 : x + 1) {
            if (l[k] == 0) continue;
            const int s = (l[k] == prev) ? prev_slot : lc_slot<true>(hkey, l[k]);
            if (s < 0) { err = LC_ERR_LABELS; continue; }
            prev = l[k]; prev_slot = s;
            const unsigned e = (x == 0 ? 1u : 0u) | (x == (uint32_t)nx - 1 ? 2u : 0u);
            if (e && ((hedge[s >> 2] >> (8 * (s & 3))) & e) != e) atomicOr(&hedge[s >> 2], e << (8 * (s & 3)));
        }
    }
    __syncthreads();

    // ---- B: dense index per id, accumulators
    for (int s = tid; s < LC_HASH; s += LC_THREADS) {
        if (hkey[s] == 0) continue;
        const int i = atomicAdd(&nlab, 1);
        if (i >= LC_NL) continue;
        hidx[s] = (uint16_t)i;
        dlabel[i] = hkey[s];
        dshift[i] = -1;
        dseam[i] = -1;
        alo[i] = 0; ahi[i] = 0;
        swv[i] = 0.0; swvy[i] = 0.0; swvx[i] = 0.0;
        if (((hedge[s >> 2] >> (8 * (s & 3))) & 3u) == 3u) {
            const int q = atomicAdd(&nseam, 1);
            if (q < ks) { dseam[i] = q; seam_idx[q] = i; }
        }
    }
    for (int w = tid; w < ks * nxw; w += LC_THREADS) colbits[w] = 0;
    __syncthreads();
    if (nlab > LC_NL || nseam > ks || err) {
        if (tid == 0) { ovf[blockIdx.x] = 1; counters[1] = 1ull; }        // (plain stores: any writer says the same)
        return;
    }
    const int n = nlab;
    if (n == 0) return;

    // ---- C: western edge of the ids that cross the seam
    if (nseam > 0) {
        for (uint32_t p0 = (uint32_t)tid * 4; p0 < npx; p0 += LC_THREADS * 4) {
            int32_t l[4];
            load4(p0, l);
            if ((l[0] | l[1] | l[2] | l[3]) == 0) continue;
            uint32_t x = p0 % (uint32_t)nx;
#pragma unroll
            for (int k = 0; k < 4; ++k, x = (x + 1 == (uint32_t)nx) ? 0 : x + 1) {
                if (l[k] == 0) continue;
                const int q = dseam[hidx[lc_slot<false>(hkey, l[k])]];
                if (q < 0) continue;
                const unsigned bit = 1u << (x & 31);
                unsigned *wp = &colbits[q * nxw + (x >> 5)];
                if (!(*wp & bit)) atomicOr(wp, bit);
            }
        }
        __syncthreads();
        if (tid < nseam) {
            const unsigned *cb = &colbits[tid * nxw];
            int prev = -1, best = 0, sh = -2;            // np.argmax(np.diff(cols)): first largest gap (:883)
            for (int x = 0; x < nx; ++x) {
                if (!((cb[x >> 5] >> (x & 31)) & 1u)) continue;
                if (prev >= 0 && x - prev > best) { best = x - prev; sh = x; }
                prev = x;
            }
            dshift[seam_idx[tid]] = sh;
        }
        __syncthreads();
    }

    // ---- D: the sums.  A wave covers 256 consecutive pixels per step: the contributions of its lanes are combined per
    // flag id (usually one or two ids) before they reach the LDS accumulators -- same-address LDS atomics serialise.
    const int lane = tid & 63;
    const uint32_t per_step = LC_THREADS * 4, steps = (npx + per_step - 1) / per_step;
    for (uint32_t st = 0; st < steps; ++st) {                 // uniform loop: whole waves take part in the ballots below
        const uint32_t p0 = st * per_step + (uint32_t)tid * 4;
        int32_t l[4] = {0, 0, 0, 0};
        if (p0 < npx) load4(p0, l);
        const bool anyfg = (l[0] | l[1] | l[2] | l[3]) != 0;
        if (__ballot(anyfg) == 0ull) continue;
        // this lane's contribution to ONE id (s_*); a second id within its four pixels (rare) goes to LDS directly
        int s_ci = -1;
        long long s_lo = 0, s_hi = 0;
        double s_wv = 0.0, s_wvy = 0.0, s_wvx = 0.0;
        if (anyfg) {
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (l[k] != 0) ? (double)vp[p0 + k] : 0.0;
            int32_t cur = 0;
            int ci = -1, cy = -1, cnt = 0, csh = 0;
            double a_wv = 0.0, a_wvy = 0.0, a_wvx = 0.0, wy = 0.0;
            auto flush = [&]() {
                if (ci < 0) return;
                const long long lo = (long long)cnt * wlo[cy], hi = (long long)cnt * whi[cy];
                if (s_ci < 0 || s_ci == ci) {
                    s_ci = ci; s_lo += lo; s_hi += hi; s_wv += a_wv; s_wvy += a_wvy; s_wvx += a_wvx;
                } else {
                    atomicAdd((unsigned long long *)&alo[ci], (unsigned long long)lo);
                    atomicAdd((unsigned long long *)&ahi[ci], (unsigned long long)hi);
                    atomicAdd(&swv[ci], a_wv);
                    atomicAdd(&swvy[ci], a_wvy);
                    atomicAdd(&swvx[ci], a_wvx);
                }
                ci = -1;
            };
            int y = (int)(p0 / (uint32_t)nx), x = (int)(p0 - (uint32_t)y * (uint32_t)nx);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (l[k] == 0) {
                    flush(); cur = 0;
                } else {
                    if (l[k] != cur || y != cy) {
                        flush();
                        cur = l[k]; cy = y; cnt = 0;
                        ci = hidx[lc_slot<false>(hkey, cur)];
                        csh = dshift[ci];
                        wy = (double)wrow[y];
                        a_wv = a_wvy = a_wvx = 0.0;
                    }
                    const double wv = v[k] * wy;                        // variable * weight_grid (:892), float64
                    int xr = x;
                    if (csh > 0) { xr = x - csh; if (xr < 0) xr += nx; }
                    cnt += 1;
                    // partial sums are formed in registers and added once: same products as the reference, another order
                    a_wv += wv;
                    a_wvy += wv * (double)y;
                    a_wvx += wv * (double)xr;
                }
                if (++x == nx) { x = 0; ++y; }
            }
            flush();
        }
        // combine across the wave, one id at a time
        uint64_t todo = __ballot(s_ci >= 0);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            const int L = __shfl(s_ci, src);
            const bool mine = s_ci == L;
            double r_wv = mine ? s_wv : 0.0, r_wvy = mine ? s_wvy : 0.0, r_wvx = mine ? s_wvx : 0.0;
            long long r_lo = mine ? s_lo : 0ll, r_hi = mine ? s_hi : 0ll;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                r_wv += __shfl_xor(r_wv, o); r_wvy += __shfl_xor(r_wvy, o); r_wvx += __shfl_xor(r_wvx, o);
                r_lo += __shfl_xor(r_lo, o); r_hi += __shfl_xor(r_hi, o);
            }
            if (lane == 0) {
                atomicAdd((unsigned long long *)&alo[L], (unsigned long long)r_lo);
                atomicAdd((unsigned long long *)&ahi[L], (unsigned long long)r_hi);
                atomicAdd(&swv[L], r_wv);
                atomicAdd(&swvy[L], r_wvy);
                atomicAdd(&swvx[L], r_wvx);
            }
            todo &= ~__ballot(mine);
        }
    }
    __syncthreads();

    if (tid == 0) base = atomicAdd(&counters[0], (unsigned long long)n);
    __syncthreads();
    if (base + (unsigned long long)n > cap_rows) return;
    for (int i = tid; i < n; i += LC_THREADS) {
        CtkLifeRowDev r;
        r.t = (int32_t)t; r.label = dlabel[i]; r.shift = dshift[i]; r.pad = (int32_t)((uint32_t)(ny - 1) << 16);      // (rows 0 .. ny-1)
        r.area = dev_limbs_to_double(alo[i], ahi[i], wshift, limb_bits);
        r.swv = swv[i]; r.swvy = swvy[i]; r.swvx = swvx[i];
        rows[base + i] = r;
    }
}


// ------------------------------------------------------------------------------------------------
// Strip form: the plane of one time step is cut into strips of LB_SW columns x 4 * rw rows, one workgroup each (one wave per
// rw rows), so that a few hundred time steps of a 0.25-degree grid (10^6 pixels per plane) fill the chip, and every byte of
// flag / field is read once.
//   k_life_seam   per time step: the ids present in BOTH seam columns (x = 0, x = nx-1)               -> cross[t][LB_KS]
//   k_life_strips a lane owns four columns and walks down the rows: inside a contour it adds to registers (no table look-up, no
//                 cross-lane traffic); the sums go to the workgroup's LDS table when another id enters the lane's columns and
//                 at the end, and from there into the time step's global table (LB_GH slots; integer / float64 atomics).  The
//                 x-sum is taken on the UNROLLED axis; for the crossing ids the lanes also hand in the occupancy of their
//                 columns and the per-column sums of p = w*v
//   k_life_finish per time step: western edge of the crossing ids from the occupancy (as phase C above), then
//                   sum p*x' = sum p*x - shift * sum p + nx * sum_{x < shift} p          x' = (x - shift) mod nx
//                 and one row per id.
// A time step with more ids than the tables hold raises ovf[t]; the host redoes those with k_lifecycle (which splits further).
// ------------------------------------------------------------------------------------------------
#define LB_THREADS 256
#define LB_SW 256         // columns per strip: one wave, four per lane
// (rows per wave: a launch parameter -- life_rows_per_wave in ctk_api.hip; a workgroup = 4 waves = 4 x that many rows of one strip)
#define LB_LH 128         // LDS hash slots per chunk (<= LB_LN ids)
#define LB_LN 64
#define LB_GH 256         // global hash slots per time step (<= LB_GN ids)
#define LB_GN 128
#define LB_KS 4           // seam-crossing ids per time step
#ifndef LB_PIPE
#define LB_PIPE 1         // 1: three-stage load pipeline (k_life_strips), 0: one batch at a time
#endif
#ifndef LB_BATCH
#define LB_BATCH 1        // rows per pipeline stage
#endif

struct CtkLifeAcc {
    unsigned long long lo, hi;      // area limbs
    double swv, swvy, swvx;         // sum p, sum p*y, sum p*x (x unrolled)
    unsigned ytop, ybot;            // 65536 - first row, last row + 1 (0 = none yet: the table starts zeroed, both grow by atomicMax)
};

template <int SLOTS>
__device__ inline int lb_slot_insert(int32_t *hkey, int32_t label)
{
    uint32_t s = (((uint32_t)label * 2654435761u) >> 16) & (SLOTS - 1);
    for (int probe = 0; probe < SLOTS; ++probe) {
        const int32_t k = hkey[s];
        if (k == label) return (int)s;
        if (k == 0) {
            const int32_t old = atomicCAS(&hkey[s], 0, label);
            if (old == 0 || old == label) return (int)s;
        }
        s = (s + 1) & (SLOTS - 1);
    }
    return -1;
}

__global__ __launch_bounds__(64) void k_life_seam(const int32_t *__restrict__ flag, int ny, int nx, int32_t *__restrict__ cross,
                                                  unsigned char *__restrict__ ovf)
{
    __shared__ int32_t hkey[LC_HASH];
    __shared__ unsigned hedge[LC_HASH / 4];
    __shared__ int ncr, bad;
    const int tid = threadIdx.x;
    const int64_t t = blockIdx.x;
    const int32_t *fp = flag + t * (int64_t)ny * nx;
    for (int s = tid; s < LC_HASH; s += 64) { hkey[s] = 0; if (s < LC_HASH / 4) hedge[s] = 0; }
    if (tid == 0) { ncr = 0; bad = 0; }
    __syncthreads();
    for (int y = tid; y < ny; y += 64) {
        const int32_t a = fp[(int64_t)y * nx], b = fp[(int64_t)y * nx + nx - 1];
        if (a) { const int s = lc_slot<true>(hkey, a); if (s < 0) bad = 1; else atomicOr(&hedge[s >> 2], 1u << (8 * (s & 3))); }
        if (b) { const int s = lc_slot<true>(hkey, b); if (s < 0) bad = 1; else atomicOr(&hedge[s >> 2], 2u << (8 * (s & 3))); }
    }
    __syncthreads();
    for (int s = tid; s < LC_HASH; s += 64) {
        if (hkey[s] == 0 || ((hedge[s >> 2] >> (8 * (s & 3))) & 3u) != 3u) continue;
        const int q = atomicAdd(&ncr, 1);
        if (q < LB_KS) cross[t * (LB_KS + 1) + 1 + q] = hkey[s];
    }
    __syncthreads();
    if (tid == 0) {
        cross[t * (LB_KS + 1)] = ncr <= LB_KS ? ncr : 0;
        if (ncr > LB_KS || bad) ovf[t] = 1;
    }
}

// inclusive sum along each row of 16 lanes (DPP row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0): lane 15 of a
// row ends up with the row's total
template <int CTRL>
__device__ inline long long dpp_mov_i64(long long v)
{
    int lo = (int)(unsigned)(unsigned long long)v, hi = (int)(unsigned)((unsigned long long)v >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ inline long long row16_sum(long long v)
{
    v += dpp_mov_i64<0x111>(v); v += dpp_mov_i64<0x112>(v); v += dpp_mov_i64<0x114>(v); v += dpp_mov_i64<0x118>(v);
    return v;
}
__device__ inline double row16_sum(double v)
{
    v += __longlong_as_double(dpp_mov_i64<0x111>(__double_as_longlong(v)));
    v += __longlong_as_double(dpp_mov_i64<0x112>(__double_as_longlong(v)));
    v += __longlong_as_double(dpp_mov_i64<0x114>(__double_as_longlong(v)));
    v += __longlong_as_double(dpp_mov_i64<0x118>(__double_as_longlong(v)));
    return v;
}

// VEC: nx a multiple of 4 and both slabs aligned for one 16- / 32-byte request per lane and row (the host checks); a compile-time
// choice so that no branch stands between the loads of the pipeline below
// (93-100 VGPRs: five waves per SIMD.  Room for six / seven / eight -- 80 / 72 / 63 VGPRs, 10 / 17 / 68 values in scratch -- costs more than it
// brings: device part of the frame 0.81 -> 0.90 / 0.98 / 1.25 ms at 2707 x 181 x 360; round 6)
template <typename VT, bool VEC>
__global__ __launch_bounds__(LB_THREADS) void k_life_strips(const int32_t *__restrict__ flag, const VT *__restrict__ field, int ny, int nx, int nxw, int nsx, int nby, int rw,
                                                            const int64_t *__restrict__ wlo, const int64_t *__restrict__ whi, const float *__restrict__ wrow,
                                                            const int32_t *__restrict__ cross, int32_t *__restrict__ gkey, CtkLifeAcc *__restrict__ gacc,
                                                            unsigned *__restrict__ occ, double *__restrict__ cp, unsigned char *__restrict__ ovf)
{
    __shared__ int32_t hkey[LB_LH];
    __shared__ long long alo[LB_LH], ahi[LB_LH];
    __shared__ double swv[LB_LH], swvy[LB_LH], swvx[LB_LH];
    __shared__ unsigned ytop[LB_LH], ybot[LB_LH];
    __shared__ int bad, skip;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned per_t = (unsigned)nsx * (unsigned)nby;
    const int64_t t = blockIdx.x / per_t;
    const unsigned rem = blockIdx.x % per_t;
    const int x0 = (int)(rem % (unsigned)nsx) * LB_SW + lane * 4;          // this lane's four columns, the same in every row
    const int ya = ((int)(rem / (unsigned)nsx) * (LB_THREADS / 64) + wave) * rw, yb = min(ny, ya + rw);
    for (int s = tid; s < LB_LH; s += LB_THREADS) { hkey[s] = 0; alo[s] = 0; ahi[s] = 0; swv[s] = 0.0; swvy[s] = 0.0; swvx[s] = 0.0; ytop[s] = 0u; ybot[s] = 0u; }
    if (tid == 0) { bad = 0; skip = ovf[t]; }                              // the seam kernel (or a sibling) already gave the time step up
    __syncthreads();
    if (skip) return;
    const uint32_t npx = (uint32_t)ny * (uint32_t)nx;
    const int32_t *fp = flag + t * (int64_t)npx;
    const VT *vp = field + t * (int64_t)npx;
    constexpr bool vec = VEC, vecv = VEC;
    const int32_t *cr = cross + t * (LB_KS + 1);
    const int ncr = cr[0];
    int32_t cid[LB_KS];
#pragma unroll
    for (int q = 0; q < LB_KS; ++q) cid[q] = q < ncr ? cr[1 + q] : 0;
    auto cross_of = [&](int32_t lab) { int cq = -1;
#pragma unroll
        for (int q = 0; q < LB_KS; ++q) if (cid[q] == lab) cq = q;
        return cq; };

    // The id this lane currently collects (in registers): its columns stay the same from row to row, so inside a contour a lane
    // keeps adding to registers; only when another id enters its columns the collected sums go to the LDS tables.
    int32_t h = 0;
    int hs = -1, hq = -1;
    long long h_lo = 0, h_hi = 0;
    double h_wv = 0.0, h_wvy = 0.0, h_wvx = 0.0, hc[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned hoc = 0;                                                      // which of the four columns hold a pixel of h
    int h_y0 = 0, h_y1 = 0;                                                // first / last row with a pixel of h in this lane
    auto flush_columns = [&]() {                                            // seam-crossing id: occupied columns, column sums of p
        if (hq < 0 || !hoc) return;
        const size_t row = (size_t)t * LB_KS + (size_t)hq;
        unsigned *wp = &occ[row * nxw + (x0 >> 5)];
        const unsigned bits = hoc << (x0 & 31);                             // (x0 is a multiple of 4: one word)
        if ((*wp & bits) != bits) atomicOr(wp, bits);
#pragma unroll
        for (int k = 0; k < 4; ++k) if ((hoc >> k) & 1u) unsafeAtomicAdd(&cp[row * nx + x0 + k], hc[k]);
    };
    auto flush_rows = [&](int slot, int ya_, int yb_) {                    // (look before the atomic: most lanes of a contour say the same)
        if (ytop[slot] < 65536u - (unsigned)ya_) atomicMax(&ytop[slot], 65536u - (unsigned)ya_);
        if (ybot[slot] < (unsigned)yb_ + 1u) atomicMax(&ybot[slot], (unsigned)yb_ + 1u);
    };
    auto flush_held = [&]() {
        if (!h) return;
        flush_rows(hs, h_y0, h_y1);
        atomicAdd((unsigned long long *)&alo[hs], (unsigned long long)h_lo);
        atomicAdd((unsigned long long *)&ahi[hs], (unsigned long long)h_hi);
        atomicAdd(&swv[hs], h_wv);
        atomicAdd(&swvy[hs], h_wvy);
        atomicAdd(&swvx[hs], h_wvx);
        flush_columns();
        h = 0; hs = -1; hq = -1; h_lo = 0; h_hi = 0; h_wv = h_wvy = h_wvx = 0.0; hoc = 0;
        hc[0] = hc[1] = hc[2] = hc[3] = 0.0;
    };

    // uniform loops: whole waves take part in the ballots.  A three-stage pipeline over batches of LB_BATCH rows: while batch i is
    // added up, the field values of batch i + 1 and the flags of batch i + 2 are in flight.  Nothing around the loads is
    // conditional (a branch makes hipcc wait for every outstanding load at the join): rows beyond the wave's last read row yb - 1
    // again, lanes beyond the grid read column 0, and a lane whose four flags are all background reads the plane's first field
    // values -- one cached line for the whole wave -- instead of its own (9 in 10 lanes: the field costs a fraction of its 4 B per
    // pixel in HBM traffic).
    const int xl = x0 < nx ? x0 : 0;
    auto load_flags = [&](int4 (&q)[LB_BATCH], int y0) {
#pragma unroll
        for (int u = 0; u < LB_BATCH; ++u) {
            const int yy = min(y0 + u, yb - 1);
            const int32_t *rp = fp + (size_t)yy * nx + xl;
            if (vec) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const i32x4 qq = __builtin_nontemporal_load((const i32x4 *)rp);
                q[u] = make_int4(qq.x, qq.y, qq.z, qq.w);
            } else {
                q[u].x = rp[0];
                q[u].y = (xl + 1 < nx) ? rp[1] : 0;
                q[u].z = (xl + 2 < nx) ? rp[2] : 0;
                q[u].w = (xl + 3 < nx) ? rp[3] : 0;
            }
        }
    };
    // (what was read for rows beyond yb or columns beyond nx is masked where it is USED: overwriting a register a load is still
    // filling would make the wave wait for that load right there)
    const bool lane_in = x0 < nx;
    auto load_fields = [&](VT (&v)[LB_BATCH][4], const int4 (&q)[LB_BATCH], int y0) {
#pragma unroll
        for (int u = 0; u < LB_BATCH; ++u) {
            const bool any = lane_in && y0 + u < yb && (q[u].x | q[u].y | q[u].z | q[u].w) != 0;
            if (vecv) {
                const VT *rp = any ? vp + (size_t)(y0 + u) * nx + x0 : vp;
                __builtin_memcpy(v[u], (const void *)rp, sizeof(v[u]));                    // one 16- / 32-byte request
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = (VT)0;
                if (any) {
                    const VT *rp = vp + (size_t)(y0 + u) * nx + x0;
                    if (q[u].x) v[u][0] = rp[0];
                    if (q[u].y) v[u][1] = rp[1];
                    if (q[u].z) v[u][2] = rp[2];
                    if (q[u].w) v[u][3] = rp[3];
                }
            }
        }
    };
#if LB_PIPE
    int4 qb[LB_BATCH], qn[LB_BATCH], qnn[LB_BATCH];
    VT vb[LB_BATCH][4], vn[LB_BATCH][4];
    load_flags(qb, ya);
    load_flags(qn, ya + LB_BATCH);
    load_fields(vb, qb, ya);
#else
    int4 qb[LB_BATCH];
    VT vb[LB_BATCH][4];
#endif
    for (int yq = ya; yq < yb; yq += LB_BATCH) {
#if LB_PIPE
        load_flags(qnn, yq + 2 * LB_BATCH);
        load_fields(vn, qn, yq + LB_BATCH);
#else
        load_flags(qb, yq);
        load_fields(vb, qb, yq);
#endif
#pragma unroll
        for (int u = 0; u < LB_BATCH; ++u) {
            const int y = yq + u;                                          // (wave-uniform)
            if (y >= yb) continue;
            const int32_t l[4] = {lane_in ? qb[u].x : 0, lane_in ? qb[u].y : 0, lane_in ? qb[u].z : 0, lane_in ? qb[u].w : 0};
            const bool anyfg = (l[0] | l[1] | l[2] | l[3]) != 0;
            if (__ballot(anyfg) == 0ull) continue;
            if (!anyfg) continue;
            const double wy = (double)wrow[y];
            const long long lo_y = wlo[y], hi_y = whi[y];
            bool own = h != 0 && (l[0] == h || l[1] == h || l[2] == h || l[3] == h);
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (l[k] == 0) continue;
                const double wv = (double)vb[u][k] * wy;                   // variable * weight_grid (contrack.py:892), float64
                if (l[k] != h) {
                    if (!own) {                                            // the held id is not in this row's columns: hand it in, take this one
                        flush_held();
                        h = l[k];
                        hs = lb_slot_insert<LB_LH>(hkey, h);
                        if (hs < 0) { bad = 1; hs = 0; }
                        hq = cross_of(h);
                        h_y0 = y;
                        own = true;
                    } else {                                               // a second id in the lane's columns (rare): straight to the tables
                        int s2 = lb_slot_insert<LB_LH>(hkey, l[k]);
                        if (s2 < 0) { bad = 1; s2 = 0; }
                        atomicAdd((unsigned long long *)&alo[s2], (unsigned long long)lo_y);
                        atomicAdd((unsigned long long *)&ahi[s2], (unsigned long long)hi_y);
                        atomicAdd(&swv[s2], wv);
                        atomicAdd(&swvy[s2], wv * (double)y);
                        atomicAdd(&swvx[s2], wv * (double)(x0 + k));
                        flush_rows(s2, y, y);
                        const int q2 = cross_of(l[k]);
                        if (q2 >= 0) {
                            const size_t row = (size_t)t * LB_KS + (size_t)q2;
                            unsigned *wp = &occ[row * nxw + ((x0 + k) >> 5)];
                            const unsigned bit = 1u << ((x0 + k) & 31);
                            if (!(*wp & bit)) atomicOr(wp, bit);
                            unsafeAtomicAdd(&cp[row * nx + x0 + k], wv);
                        }
                        continue;
                    }
                }
                cnt += 1;
                h_wv += wv;
                h_wvy += wv * (double)y;
                h_wvx += wv * (double)(x0 + k);
                hc[k] += wv;
                hoc |= 1u << k;
            }
            h_lo += (long long)cnt * lo_y;
            h_hi += (long long)cnt * hi_y;
            if (cnt) h_y1 = y;
        }
#if LB_PIPE
#pragma unroll
        for (int u = 0; u < LB_BATCH; ++u) {
            qb[u] = qn[u]; qn[u] = qnn[u];
#pragma unroll
            for (int k = 0; k < 4; ++k) vb[u][k] = vn[u][k];
        }
#endif
    }
    // what the lanes still hold: column data lane by lane, the five sums combined per row of 16 lanes first (same-address LDS
    // atomics serialise; DPP moves run at VALU speed)
    flush_columns();
    if (h) flush_rows(hs, h_y0, h_y1);
    uint64_t todo = __ballot(h != 0);
    while (todo) {
        const int L = __builtin_amdgcn_readlane(hs, __builtin_ctzll(todo));
        const bool mine = h != 0 && hs == L;
        double r_wv = mine ? h_wv : 0.0, r_wvy = mine ? h_wvy : 0.0, r_wvx = mine ? h_wvx : 0.0;
        long long r_lo = mine ? h_lo : 0ll, r_hi = mine ? h_hi : 0ll;
        r_wv = row16_sum(r_wv); r_wvy = row16_sum(r_wvy); r_wvx = row16_sum(r_wvx);
        r_lo = row16_sum(r_lo); r_hi = row16_sum(r_hi);
        if ((lane & 15) == 15) {
            if ((r_lo | r_hi) != 0) {
                atomicAdd((unsigned long long *)&alo[L], (unsigned long long)r_lo);
                atomicAdd((unsigned long long *)&ahi[L], (unsigned long long)r_hi);
            }
            if (r_wv != 0.0 || r_wvy != 0.0 || r_wvx != 0.0) {
                atomicAdd(&swv[L], r_wv);
                atomicAdd(&swvy[L], r_wvy);
                atomicAdd(&swvx[L], r_wvx);
            }
        }
        todo &= ~__ballot(mine);
    }
    __syncthreads();
    if (bad) { if (tid == 0) ovf[t] = 1; return; }
    // merge the workgroup's ids into the time step's table
    for (int s = tid; s < LB_LH; s += LB_THREADS) {
        const int32_t lab = hkey[s];
        if (lab == 0) continue;
        int32_t *gk = gkey + t * LB_GH;
        const int g = lb_slot_insert<LB_GH>(gk, lab);
        if (g < 0) { ovf[t] = 1; continue; }
        CtkLifeAcc *a = gacc + t * LB_GH + g;
        atomicAdd(&a->lo, (unsigned long long)alo[s]);
        atomicAdd(&a->hi, (unsigned long long)ahi[s]);
        unsafeAtomicAdd(&a->swv, swv[s]);
        unsafeAtomicAdd(&a->swvy, swvy[s]);
        unsafeAtomicAdd(&a->swvx, swvx[s]);
        if (a->ytop < ytop[s]) atomicMax(&a->ytop, ytop[s]);
        if (a->ybot < ybot[s]) atomicMax(&a->ybot, ybot[s]);
    }
}

__global__ __launch_bounds__(LB_GH) void k_life_finish(int nx, int nxw, const int32_t *__restrict__ cross, const int32_t *__restrict__ gkey,
                                                        const CtkLifeAcc *__restrict__ gacc, const unsigned *__restrict__ occ, const double *__restrict__ cp,
                                                        int wshift, int limb_bits, CtkLifeRowDev *rows, unsigned long long cap_rows,
                                                        unsigned long long *counters, unsigned char *__restrict__ ovf)
{
    __shared__ int cshift[LB_KS];
    __shared__ double cleft[LB_KS];
    __shared__ int nlab;
    __shared__ unsigned long long base;
    const int tid = threadIdx.x;
    const int64_t t = blockIdx.x;
    if (tid == 0) nlab = 0;
    __syncthreads();
    const int32_t lab = gkey[t * LB_GH + tid];
    int mine = -1;
    if (lab != 0) mine = atomicAdd(&nlab, 1);
    const int32_t *cr = cross + t * (LB_KS + 1);
    const int ncr = cr[0];
    {   // wave q: western edge of crossing id q, np.argmax(np.diff(cols)) + 1 = column right of the FIRST largest gap (contrack.py:883)
        const int q = tid >> 6, lane = tid & 63;
        if (q < ncr) {                                       // (uniform per wave)
            const unsigned *cb = occ + ((size_t)t * LB_KS + q) * nxw;
            int prev = -1, best = 0, sh = -2;
            for (int w0 = 0; w0 < nxw; w0 += 64) {
                const unsigned mine_w = (w0 + lane < nxw) ? cb[w0 + lane] : 0u;
                // inside the lane's word: first largest gap between neighbouring occupied columns (all lanes at once) ...
                int in_best = 0, in_at = 0;
                {
                    int p = -1;
                    for (unsigned bb = mine_w; bb; bb &= bb - 1) {
                        const int x = __builtin_ctz(bb);
                        if (p >= 0 && x - p > in_best) { in_best = x - p; in_at = x; }
                        p = x;
                    }
                }
                // ... then the words in order (wave-uniform scalars): gap to the previous word's last column, the word's own gap
                uint64_t nz = __ballot(mine_w != 0u);
                while (nz) {
                    const int wl = __builtin_ctzll(nz);
                    nz &= nz - 1;
                    const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)mine_w, wl);
                    const int wb = __builtin_amdgcn_readlane(in_best, wl), wa = __builtin_amdgcn_readlane(in_at, wl);
                    const int xbase = (w0 + wl) * 32, first = xbase + __builtin_ctz(bits), last = xbase + 31 - __builtin_clz(bits);
                    if (prev >= 0 && first - prev > best) { best = first - prev; sh = first; }
                    if (wb > best) { best = wb; sh = xbase + wa; }
                    prev = last;
                }
            }
            if (lane == 0) { cshift[q] = sh; cleft[q] = 0.0; }
        }
    }
    __syncthreads();
    for (int q = 0; q < ncr; ++q) {                          // sum of the columns left of the edge: all lanes, independent loads
        const double *cc = cp + ((size_t)t * LB_KS + q) * nx;
        const int sh = cshift[q];
        double left = 0.0;
        for (int x = tid; x < sh; x += LB_GH) left += cc[x];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) left += __shfl_xor(left, o);
        if ((tid & 63) == 0 && left != 0.0) atomicAdd(&cleft[q], left);
    }
    __syncthreads();
    if (ovf[t] != 0 || nlab > LB_GN) {                          // uniform: ovf[t] was written by earlier launches only
        if (tid == 0) { ovf[t] = 1; counters[1] = 1ull; }
        return;
    }
    if (nlab == 0) return;
    if (tid == 0) base = atomicAdd(&counters[0], (unsigned long long)nlab);
    __syncthreads();
    if (base + (unsigned long long)nlab > cap_rows || mine < 0) return;
    const CtkLifeAcc a = gacc[t * LB_GH + tid];
    CtkLifeRowDev r;
    r.t = (int32_t)t; r.label = lab; r.shift = -1;
    r.pad = (int32_t)((65536u - a.ytop) | ((a.ybot - 1u) << 16));
    r.area = dev_limbs_to_double((long long)a.lo, (long long)a.hi, wshift, limb_bits);
    r.swv = a.swv; r.swvy = a.swvy; r.swvx = a.swvx;
    for (int q = 0; q < ncr; ++q) {
        if (cr[1 + q] != lab) continue;
        r.shift = cshift[q];
        if (r.shift > 0) r.swvx = (a.swvx - (double)r.shift * a.swv) + (double)nx * cleft[q];
    }
    rows[base + mine] = r;
}


// ------------------------------------------------------------------------------------------------
// Exact rows.  The sums above are float64 but taken in another order than the reference's: np.sum (pairwise) for the area and
// the intensity numerator (contrack.py:874-875), and np.bincount -- strictly sequential, raster order of the ROLLED plane --
// inside ndimage.center_of_mass (:886 / :892).  That matters only for rows on a rounding boundary (a centre of mass that is an
// integer up to rounding, a value at the edge of two decimals): the host picks those (about 1 %) and this kernel re-evaluates
// them in the reference's own orders.  One wave per row; every lane executes the same scalar recurrence on values broadcast from
// the lane that holds the pixel.
//   out[i] = {np.sum(w), np.sum(w * v), sequential sum of p, of p * y, of p * x'}   with p = v * w,  x' = (x - shift) mod nx
// ------------------------------------------------------------------------------------------------
struct CtkLifeExact {
    double area, swv, s, sy, sx;
};

// numpy's pairwise float64 add.reduce over a contiguous array (ctk_np_sum / np_pairwise in ctk_resolve.cpp), without recursion
__device__ inline double dev_np_leaf(const double *a, uint32_t n)
{
    if (n < 8u) {
        double r = 0.0;
        for (uint32_t i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    uint32_t i = 8;
    for (; i + 8 <= n; i += 8) { r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3]; r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7]; }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}
// post-order walk of numpy's halving tree over a block of cn <= 8192 elements; leaf(offset, n) is called for the leaves
// (n <= 128) in order and returns their sums.  The frame stack lives where the caller puts it (LDS: a dynamically indexed
// private array would sit in scratch memory, i.e. one HBM round trip per access)
struct NpFrame { uint32_t off, n, st, pad; double l; };       // {offset, n, state, value of the left child}
template <typename Leaf>
__device__ inline double np_walk(uint32_t cn, Leaf leaf, NpFrame *f)
{
    int sp = 1;
    f[0].off = 0; f[0].n = cn; f[0].st = 0;
    double val = 0.0;
    bool have = false;
    while (sp > 0) {
        const int k = sp - 1;
        if (!have) {
            if (f[k].n <= 128u) { val = leaf(f[k].off, f[k].n); have = true; sp--; continue; }
            uint32_t n2 = f[k].n / 2; n2 -= n2 % 8u;
            f[k].st = 1;
            f[sp].off = f[k].off; f[sp].n = n2; f[sp].st = 0; sp++;
            continue;
        }
        // a child of frame k has just finished with `val`
        if (f[k].st == 1) {
            uint32_t n2 = f[k].n / 2; n2 -= n2 % 8u;
            f[k].l = val; f[k].st = 2; have = false;
            f[sp].off = f[k].off + n2; f[sp].n = f[k].n - n2; f[sp].st = 0; sp++;
        } else { val = f[k].l + val; sp--; }
    }
    return val;
}
// The sum over a long list by (the first 256 threads of) a workgroup: thread 0 lists the leaves of a block of 8192, the threads sum one leaf each, thread 0
// combines them in the tree's order.  lo / ln: LDS scratch for 128 leaves.
// Two such sums over lists of the same length at once (k_life_exact: the area and the weighted sum of one contour): the tree, and
// so the list of leaves, depends on the length only -- it is listed once per chunk LENGTH (every chunk but the last is 8192 long),
// the leaves of both lists are summed side by side (threads 0..63 / 64..127 of a group: its first two waves; the other two take no part -- k_life_exact
// gives them other work) and combined by two waves at the same time.
// lv: 256 doubles, frames: 2 x 16.  Results valid in thread 0 (ra) and thread 64 (rb).
// Round 5: a workgroup of 1024 threads takes FOUR blocks per round (groups of 256 threads, each with its own leaf list, leaf sums and
// the two threads that walk the trees); the block sums are added in the order of the blocks by thread 0 / thread 64, as numpy's
// buffered reduction adds its 8192-element buffers one after the other.
// lo / ln: [G][128], lv: [G][256], nleaf: [G], frames: [G][32], cres: [2][4]   (G = blockDim / 256 <= 4)
// sync: the barrier between the steps, made by every thread that runs this function (the first two waves of every group at least)
template <typename Sync>
__device__ inline void wg_np_sum2(const double *a, const double *b, size_t n, uint32_t *lo_all, uint32_t *ln_all, double *lv_all, int *nleaf_all, NpFrame *frames_all,
                                  double *cres, double &ra, double &rb, Sync sync)
{
    double acc = 0.0;
    const int tid = (int)threadIdx.x, G = (int)(blockDim.x >> 8) > 0 ? (int)(blockDim.x >> 8) : 1, grp = tid >> 8, t8 = tid & 255;
    uint32_t *lo = lo_all + grp * 128, *ln = ln_all + grp * 128;
    double *lv = lv_all + grp * 256;
    NpFrame *frames = frames_all + grp * 32;
    uint32_t listed = 0;                                                     // (of this thread's group; uniform within the group)
    for (size_t r0 = 0; r0 < n; r0 += (size_t)8192 * G) {
        const size_t c0 = r0 + (size_t)8192 * grp;
        const uint32_t cn = c0 < n ? (uint32_t)(n - c0 < 8192 ? n - c0 : 8192) : 0u;
        if (cn && cn != listed) {
            if (t8 == 0) {
                int k = 0;
                (void)np_walk(cn, [&](uint32_t off, uint32_t m) { lo[k] = off; ln[k] = m; k++; return 0.0; }, frames);
                nleaf_all[grp] = k;
            }
            listed = cn;
        }
        sync();
        const int nl = cn ? nleaf_all[grp] : 0;
        if (t8 < nl) lv[t8] = dev_np_leaf(a + c0 + lo[t8], ln[t8]);
        else if (t8 >= 64 && t8 - 64 < nl) lv[64 + t8] = dev_np_leaf(b + c0 + lo[t8 - 64], ln[t8 - 64]);
        sync();
        if (cn && t8 == 0) { int k = 0; cres[grp] = np_walk(cn, [&](uint32_t, uint32_t) { return lv[k++]; }, frames); }
        else if (cn && t8 == 64) { int k = 128; cres[4 + grp] = np_walk(cn, [&](uint32_t, uint32_t) { return lv[k++]; }, frames + 16); }
        sync();
        if (tid == 0) { for (int g = 0; g < G; g++) if (r0 + (size_t)8192 * g < n) acc += cres[g]; }
        else if (tid == 64) { for (int g = 0; g < G; g++) if (r0 + (size_t)8192 * g < n) acc += cres[4 + g]; }
        sync();
    }
    if (tid == 0) ra = acc;
    if (tid == 64) rb = acc;
}

struct CtkLifeKey {
    int32_t t, label, shift, pad;
};

// ------------------------------------------------------------------------------------------------
// The compact lists of the listed rows (round 6: three kernels over ALL rows of ALL listed contours; they were the first three phases
// of k_life_exact -- one workgroup per contour, 0.08 + 0.21 ms on one CU for a contour of 105 000 pixels -- and k_life_count, a scan of
// its own over the same rows, 75 us).
// rowtab (per key, 3 * nrows words at roffs[key]): pixels of the id per row | those left of the roll edge | their offset in the lists.
//   k_life_rows  <<<(keys, rows / 4), 256>>>  one wave per row of the contour's row extent: the two counts
//   k_life_rowscan <<<keys, 256>>>            exclusive scan of the counts, the contour's pixel count -> counts[key]  (the host sums them: list offsets)
//   k_life_lists <<<(keys, rows / 4), 256>>>  one wave per row: the five lists.  sw / sp: the weights and products in raster order of the plane
//       (what np.sum sees, contrack.py:874-875); sq / sqy / sqx: p, p*y, p*x' in raster order of the ROLLED plane (np.bincount's order inside
//       ndimage.center_of_mass, :886 / :892): of a row the pixels right of the edge (x >= shift) first, then those left of it.
// ------------------------------------------------------------------------------------------------
#define LR_U 12                                                  // flag loads in flight per lane (one wave per row: a scan bound by the trips to memory)
__global__ __launch_bounds__(256) void k_life_rows(const int32_t *__restrict__ flag, const CtkLifeKey *__restrict__ keys, const uint64_t *__restrict__ roffs,
                                                   int ny, int nx, uint32_t *__restrict__ rowtab)
{
    const CtkLifeKey k = keys[blockIdx.x];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int ya = (int)((uint32_t)k.pad & 0xffffu), yb = min(ny - 1, (int)((uint32_t)k.pad >> 16)), nrows = yb - ya + 1;
    const int r = (int)blockIdx.y * 4 + wave;
    if (r >= nrows) return;
    const int shift = k.shift > 0 ? k.shift : 0;
    const int32_t *rp = flag + (int64_t)k.t * ((int64_t)ny * nx) + (size_t)(ya + r) * nx;
    uint32_t *rcnt = rowtab + roffs[blockIdx.x], *rleft = rcnt + nrows;
    uint32_t c = 0, cl = 0;
    for (int x0 = 0; x0 < nx; x0 += LR_U * 64) {
        int32_t v[LR_U];
#pragma unroll
        for (int u = 0; u < LR_U; ++u) { const int x = x0 + u * 64 + lane; v[u] = x < nx ? rp[x] : 0; }
#pragma unroll
        for (int u = 0; u < LR_U; ++u) {
            const int x = x0 + u * 64 + lane;
            const bool m = x < nx && v[u] == k.label;
            c += (uint32_t)__popcll(__ballot(m));
            cl += (uint32_t)__popcll(__ballot(m && x < shift));
        }
    }
    if (lane == 0) { rcnt[r] = c; rleft[r] = cl; }
}

__global__ __launch_bounds__(256) void k_life_rowscan(const CtkLifeKey *__restrict__ keys, const uint64_t *__restrict__ roffs, int ny, uint32_t *__restrict__ rowtab,
                                                      uint32_t *__restrict__ counts)
{
    __shared__ uint32_t part[256];
    const CtkLifeKey k = keys[blockIdx.x];
    const int tid = (int)threadIdx.x;
    const int ya = (int)((uint32_t)k.pad & 0xffffu), yb = min(ny - 1, (int)((uint32_t)k.pad >> 16)), nrows = yb - ya + 1;
    uint32_t *rcnt = rowtab + roffs[blockIdx.x], *roff = rcnt + 2 * (size_t)nrows;
    const int per = (nrows + 255) / 256, r0 = tid * per, r1 = min(nrows, r0 + per);
    uint32_t sum = 0;
    for (int r = r0; r < r1; ++r) sum += rcnt[r];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; }
        counts[blockIdx.x] = run;
    }
    __syncthreads();
    uint32_t run = part[tid];
    for (int r = r0; r < r1; ++r) { roff[r] = run; run += rcnt[r]; }
}

template <typename VT>
__global__ __launch_bounds__(256) void k_life_lists(const int32_t *__restrict__ flag, const VT *__restrict__ field, const float *__restrict__ wrow,
                                                    const CtkLifeKey *__restrict__ keys, const uint64_t *__restrict__ offs, const uint64_t *__restrict__ roffs,
                                                    int ny, int nx, double *__restrict__ sw, double *__restrict__ sp_, double *__restrict__ sq,
                                                    double *__restrict__ sqy, double *__restrict__ sqx, const uint32_t *__restrict__ rowtab)
{
    const CtkLifeKey k = keys[blockIdx.x];
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int ya = (int)((uint32_t)k.pad & 0xffffu), yb = min(ny - 1, (int)((uint32_t)k.pad >> 16)), nrows = yb - ya + 1;
    const int r = (int)blockIdx.y * 4 + wave;
    if (r >= nrows) return;
    const uint32_t *rcnt = rowtab + roffs[blockIdx.x], *rleft = rcnt + nrows, *roff = rleft + nrows;
    if (rcnt[r] == 0) return;                                              // (wave-uniform)
    const int shift = k.shift > 0 ? k.shift : 0;
    const int y = ya + r;
    const int64_t npx = (int64_t)ny * nx;
    const int32_t *rp = flag + (int64_t)k.t * npx + (size_t)y * nx;
    const VT *rv = field + (int64_t)k.t * npx + (size_t)y * nx;
    const uint64_t o = offs[blockIdx.x];
    double *gw = sw + o, *gp = sp_ + o, *gq = sq + o, *gqy = sqy + o, *gqx = sqx + o;
    const double w = (double)wrow[y];
    const uint32_t base = roff[r], nleft = rleft[r], nright = rcnt[r] - nleft;
    uint32_t seen = 0;
    for (int xq = 0; xq < nx; xq += LR_U * 64) {
        int32_t v4[LR_U];
        VT f4[LR_U];
#pragma unroll
        for (int u = 0; u < LR_U; ++u) { const int x = xq + u * 64 + lane; v4[u] = x < nx ? rp[x] : 0; }
        // (the values of all members of the batch in one trip: a load inside the branch below would be a trip of its own per u)
#pragma unroll
        for (int u = 0; u < LR_U; ++u) { const int x = xq + u * 64 + lane; f4[u] = (x < nx && v4[u] == k.label) ? rv[x] : (VT)0; }
#pragma unroll
        for (int u = 0; u < LR_U; ++u) {
            const int x = xq + u * 64 + lane;
            const bool m = x < nx && v4[u] == k.label;
            const uint64_t bal = __ballot(m);
            if (!bal) continue;
            if (m) {
                const uint32_t idx = seen + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                const double p = (double)f4[u] * w;                       // variable * weight_grid (:886 / :892) = weight_grid * variable (:875)
                gw[base + idx] = w;
                gp[base + idx] = p;
                int xr = x - shift;
                uint32_t pos;
                if (xr < 0) { xr += nx; pos = nright + idx; } else pos = idx - nleft;
                gq[base + pos] = p;
                gqy[base + pos] = p * (double)y;
                gqx[base + pos] = p * (double)xr;
            }
            seen += (uint32_t)__popcll(bal);
        }
    }
}

// One workgroup per listed row, on the lists of k_life_lists: out[i] = the five sums in the reference's orders.
// History: round 5 launched it with 1024 threads when a listed contour is large (one contour of 105 000 pixels at 0.25 deg: 2.8 -> 1.6 ms:
// row scans, pairwise sums, sequential sums one after the other, three waves walking one chain each from LDS).  Round 6: G = threads / 256
// groups of the pairwise sums, a ring of four blocks of 256 (one group) or 512 values of the sequential sums (~31 KB of LDS for the common
// 256-thread launch, ~66 KB for the large one); the pairwise sums, the sequential sums and their loads on different waves at the same time
// (see "D and E" below); the lists built by kernels of their own (above): 1.60 -> 0.46 (+ 0.11) ms for that contour.
template <int G>
__global__ __launch_bounds__(256 * G) void k_life_exact(const uint64_t *__restrict__ offs, const uint32_t *__restrict__ counts,
                                                    const double *__restrict__ sw, const double *__restrict__ sp_, const double *__restrict__ sq,
                                                    const double *__restrict__ sqy, const double *__restrict__ sqx,
                                                    CtkLifeExact *__restrict__ out, uint32_t *__restrict__ fail /* zeroed; != 0: a wait inside a workgroup expired */)
{
    constexpr int LX_EB = G == 1 ? 256 : 512;                   // values of a list per block of the sequential sums
    constexpr int LX_NSLOT = 4;                                 // blocks in LDS
    constexpr int LX_NF = G;                                    // waves that feed them (the third wave of every group of four)
    constexpr uint64_t LX_SPIN = 20000000ull;                   // 0.2 s of the 100 MHz clock: no wait inside this kernel is longer than microseconds
    __shared__ uint32_t d_arrived, e_consumed, e_filled[LX_NSLOT], lx_bad;
    __shared__ uint32_t lo[G * 128], ln[G * 128];
    __shared__ double lv[G * 256], cres[8];
    __shared__ int nleaf[G];
    __shared__ NpFrame frames[G * 32];
    __shared__ __attribute__((aligned(16))) double stage[LX_NSLOT][3][LX_EB + 2];      // (+2: the three lists of a block start in different banks)
    __shared__ double res[5];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)(blockDim.x >> 6);
    const double *gw = sw + offs[blockIdx.x], *gp = sp_ + offs[blockIdx.x], *gq = sq + offs[blockIdx.x], *gqy = sqy + offs[blockIdx.x], *gqx = sqx + offs[blockIdx.x];
    const size_t total = counts[blockIdx.x];
#ifdef CTK_PHASE_TIMING
#define LX_MARK(k) do { if (total > 50000 && tid == 0) g_phase_t[k] = wall_clock64(); } while (0)
#else
#define LX_MARK(k) do { } while (0)
#endif
    LX_MARK(1);
    // D and E at the same time (round 6; they were 0.2-0.34 + 0.9 ms of 1.6 for a contour of 105 000 pixels), by three kinds of waves:
    // D, waves 4g and 4g+1: np.sum over the raster-order lists (wg_np_sum2).  Their barrier is a counter in LDS -- the hardware barrier would
    //    stop the other waves too.
    // E, wave 3: np.bincount's strictly sequential sums over the rolled lists.  Lane l walks list l % 3 (lanes 0..2 keep the results), so the
    //    three chains advance with ONE v_add_f64 per step; a full block is one basic block of LX_EB adds and LX_EB / 2 LDS loads (hipcc waits
    //    for ALL outstanding LDS loads -- lgkmcnt(0) -- in front of the first use of a value that was loaded before a loop edge, i.e. also for
    //    the loads just issued: 1.6 of the old loop's 4.3 ns per add; tools/f64chain.hip: 2.7 ns for adds from registers, 3.5 in this form;
    //    the old form -- three waves, lane 0 of each -- ran at 8.5 ns inside this kernel).
    // F, waves 4g+2: they carry the blocks of the three lists from memory into a ring of LX_NSLOT blocks in LDS, each its own blocks
    //    (f, f + LX_NF, ...), one block in registers while the previous one waits for its slot.  (The chain wave feeding itself had one block
    //    of loads in flight and waited ~1 us of every 2.9 for it.)
    // Hand-over through LDS words: e_filled[slot] = block + 1 once a block is stored, e_consumed = blocks the chain has finished.  LDS
    // executes the operations of a wave in order, so a word written behind the data is seen behind the data.
    const bool role_d = (wave & 3) < 2, role_f = (wave & 3) == 2, role_e = wave == 3;
    const uint32_t d_waves = (uint32_t)nw / 2u;
    const size_t nblk = (total + LX_EB - 1) / LX_EB;
    if (tid == 0) { d_arrived = 0u; e_consumed = 0u; lx_bad = 0u; }
    if (tid < LX_NSLOT) e_filled[tid] = 0u;
    __syncthreads();
    auto lds_ld = [&](uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto give_up = [&]() { if (lane == 0) __hip_atomic_store(&lx_bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    double acc = 0.0;
    if (role_d) {
        uint32_t d_phase = 0;
        double area = 0.0, swv = 0.0;
        wg_np_sum2(gw, gp, total, lo, ln, lv, nleaf, frames, cres, area, swv, [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            d_phase++;
            if (lane == 0) atomicAdd(&d_arrived, 1u);
            SpinGuard sg;
            while (lds_ld(&d_arrived) < d_waves * d_phase && !lds_ld(&lx_bad)) {
                __builtin_amdgcn_s_sleep(2);
                if (spin_expired(sg, LX_SPIN)) { give_up(); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        });
        if (tid == 0) res[0] = area;
        if (tid == 64) res[1] = swv;
        LX_MARK(2);                                                          // (the pairwise sums are done; [4]: the chains)
    } else if (role_f) {
        constexpr int PER = LX_EB / 64;                                      // values per lane and list of a block
        double aq[PER], ay[PER], ax[PER], bq[PER], by[PER], bx[PER];
        auto fetch = [&](double (&hq)[PER], double (&hy)[PER], double (&hx)[PER], size_t blk) {      // (zeros behind the end of the lists)
            const size_t c0 = blk * LX_EB;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const size_t i = c0 + (size_t)u * 64 + lane;
                const bool in = i < total;
                hq[u] = in ? gq[i] : 0.0; hy[u] = in ? gqy[i] : 0.0; hx[u] = in ? gqx[i] : 0.0;
            }
        };
        auto put = [&](const double (&hq)[PER], const double (&hy)[PER], const double (&hx)[PER], size_t blk) -> bool {
            const int slot = (int)(blk % LX_NSLOT);
            SpinGuard sg;
            while ((size_t)lds_ld(&e_consumed) + LX_NSLOT <= blk) {          // the block that sits in this slot is not finished yet
                if (lds_ld(&lx_bad)) return false;
                __builtin_amdgcn_s_sleep(4);
                if (spin_expired(sg, LX_SPIN)) { give_up(); return false; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int u = 0; u < PER; ++u) { stage[slot][0][u * 64 + lane] = hq[u]; stage[slot][1][u * 64 + lane] = hy[u]; stage[slot][2][u * 64 + lane] = hx[u]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&e_filled[slot], (uint32_t)(blk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return true;
        };
        size_t blk = (size_t)(wave >> 2);
        if (blk < nblk) fetch(aq, ay, ax, blk);
        while (blk < nblk) {
            if (blk + LX_NF < nblk) fetch(bq, by, bx, blk + LX_NF);          // (in flight while the block in hand waits for its slot)
            if (!put(aq, ay, ax, blk)) break;
            blk += LX_NF;
            if (blk >= nblk) break;
            if (blk + LX_NF < nblk) fetch(aq, ay, ax, blk + LX_NF);
            if (!put(bq, by, bx, blk)) break;
            blk += LX_NF;
        }
    } else if (role_e) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const int l3 = lane % 3;
        for (size_t blk = 0; blk < nblk; ++blk) {
            const int slot = (int)(blk % LX_NSLOT);
            const int cn = (int)min((size_t)LX_EB, total - blk * LX_EB);
            SpinGuard sg;
            bool ok = true;
            while (lds_ld(&e_filled[slot]) != (uint32_t)(blk + 1)) {
                if (lds_ld(&lx_bad)) { ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
                if (spin_expired(sg, LX_SPIN)) { give_up(); ok = false; break; }
            }
            if (!ok) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const double *sv = stage[slot][l3];                              // (behind the end of the list: zeros are NOT added -- cn bounds the loop)
            if (cn == LX_EB) {
#define LX_LD(o) (*reinterpret_cast<const d2 *>(sv + (o)))
#define LX_ADD(q0, q1, q2, q3) do { acc += q0.x; acc += q0.y; acc += q1.x; acc += q1.y; acc += q2.x; acc += q2.y; acc += q3.x; acc += q3.y; } while (0)
                // sixteen values ahead of the adds.  (The empty asm statements -- the chain's value passes through them -- keep the loads in front of
                // the adds that are meant to cover them.)
                d2 a0 = LX_LD(0), a1 = LX_LD(2), a2 = LX_LD(4), a3 = LX_LD(6), a4 = LX_LD(8), a5 = LX_LD(10), a6 = LX_LD(12), a7 = LX_LD(14);
#pragma unroll
                for (int j = 16; j < LX_EB; j += 16) {
                    const d2 b0 = LX_LD(j), b1 = LX_LD(j + 2), b2 = LX_LD(j + 4), b3 = LX_LD(j + 6), b4 = LX_LD(j + 8), b5 = LX_LD(j + 10), b6 = LX_LD(j + 12), b7 = LX_LD(j + 14);
                    asm volatile("" : "+v"(acc) : : "memory");
                    LX_ADD(a0, a1, a2, a3); LX_ADD(a4, a5, a6, a7);
                    a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
                }
                LX_ADD(a0, a1, a2, a3); LX_ADD(a4, a5, a6, a7);
#undef LX_LD
#undef LX_ADD
            } else {
                for (int i = 0; i < cn; ++i) acc += sv[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&e_consumed, (uint32_t)(blk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#ifdef CTK_PHASE_TIMING
        if (total > 50000 && lane == 0) g_phase_t[4] = wall_clock64();
#endif
    }
    if (role_e && lane < 3) res[2 + lane] = acc;
    __syncthreads();
    LX_MARK(3);
    if (tid == 0) {
        CtkLifeExact r;
        r.area = res[0]; r.swv = res[1]; r.s = res[2]; r.sy = res[3]; r.sx = res[4];
        out[blockIdx.x] = r;
        if (lx_bad) atomicOr(fail, 1u);
    }
}
