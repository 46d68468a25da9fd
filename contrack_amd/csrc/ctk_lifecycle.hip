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
        r.t = (int32_t)t; r.label = dlabel[i]; r.shift = dshift[i]; r.pad = 0;
        r.area = dev_limbs_to_double(alo[i], ahi[i], wshift, limb_bits);
        r.swv = swv[i]; r.swvy = swvy[i]; r.swvx = swvx[i];
        rows[base + i] = r;
    }
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
__device__ inline double dev_np_sum(const double *a, size_t n)
{
    double acc = 0.0;
    for (size_t c0 = 0; c0 < n; c0 += 8192) {
        const uint32_t cn = (uint32_t)(n - c0 < 8192 ? n - c0 : 8192);
        // post-order walk of the halving tree: frames {offset, n, state, value of the left child}
        uint32_t f_off[16], f_n[16], f_st[16];
        double f_l[16];
        int sp = 1;
        f_off[0] = 0; f_n[0] = cn; f_st[0] = 0;
        double val = 0.0;
        bool have = false;
        while (sp > 0) {
            const int k = sp - 1;
            if (!have) {
                if (f_n[k] <= 128u) { val = dev_np_leaf(a + c0 + f_off[k], f_n[k]); have = true; sp--; continue; }
                uint32_t n2 = f_n[k] / 2; n2 -= n2 % 8u;
                f_st[k] = 1;
                f_off[sp] = f_off[k]; f_n[sp] = n2; f_st[sp] = 0; sp++;
                continue;
            }
            // a child of frame k has just finished with `val`
            if (f_st[k] == 1) {
                uint32_t n2 = f_n[k] / 2; n2 -= n2 % 8u;
                f_l[k] = val; f_st[k] = 2; have = false;
                f_off[sp] = f_off[k] + n2; f_n[sp] = f_n[k] - n2; f_st[sp] = 0; sp++;
            } else { val = f_l[k] + val; sp--; }
        }
        acc += val;
    }
    return acc;
}

struct CtkLifeKey {
    int32_t t, label, shift, pad;
};

// members of every listed (time step, id)
__global__ __launch_bounds__(64) void k_life_count(const int32_t *__restrict__ flag, const CtkLifeKey *__restrict__ keys, int ny, int nx, uint32_t *__restrict__ counts)
{
    const CtkLifeKey k = keys[blockIdx.x];
    const uint32_t npx = (uint32_t)ny * (uint32_t)nx;
    const int32_t *fp = flag + (int64_t)k.t * npx;
    uint32_t c = 0;
    for (uint32_t p = threadIdx.x; p < npx; p += 64) c += fp[p] == k.label ? 1u : 0u;
    c = wave_sum_u32(c);
    if (threadIdx.x == 0) counts[blockIdx.x] = c;
}

template <typename VT>
__global__ __launch_bounds__(64) void k_life_exact(const int32_t *__restrict__ flag, const VT *__restrict__ field, const float *__restrict__ wrow,
                                                   const CtkLifeKey *__restrict__ keys, const uint64_t *__restrict__ offs, int ny, int nx,
                                                   double *__restrict__ sw, double *__restrict__ sp_, CtkLifeExact *__restrict__ out)
{
    const CtkLifeKey k = keys[blockIdx.x];
    const uint32_t npx = (uint32_t)ny * (uint32_t)nx;
    const int32_t *fp = flag + (int64_t)k.t * npx;
    const VT *vp = field + (int64_t)k.t * npx;
    const int lane = (int)threadIdx.x;
    double *gw = sw + offs[blockIdx.x], *gp = sp_ + offs[blockIdx.x];
    // A: the row weights and the products of the id's pixels, raster order -> scratch (what weight_grid[mask] and
    //    weight_grid[mask] * variable[mask] hand to np.sum, contrack.py:874-875)
    uint32_t pos = 0;
    for (uint32_t p0 = 0; p0 < npx; p0 += 64) {
        const uint32_t p = p0 + lane;
        const bool m = p < npx && fp[p] == k.label;
        const uint64_t bal = __ballot(m);
        if (!bal) continue;
        if (m) {
            const int y = (int)(p / (uint32_t)nx);
            const double w = (double)wrow[y];
            const uint32_t at = pos + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            gw[at] = w;
            gp[at] = w * (double)vp[p];
        }
        pos += (uint32_t)__popcll(bal);
    }
    // B: ndimage.center_of_mass = np.bincount: strictly sequential over the (rolled) plane in raster order
    const int shift = k.shift > 0 ? k.shift : 0;
    double s = 0.0, sy = 0.0, sx = 0.0;
    for (int y = 0; y < ny; y++) {
        const double w = (double)wrow[y];
        for (int x0 = 0; x0 < nx; x0 += 64) {
            const int xr = x0 + lane;                                  // column in the rolled frame
            int x = xr + shift;
            if (x >= nx) x -= nx;
            const bool m = xr < nx && fp[(uint32_t)y * (uint32_t)nx + (uint32_t)x] == k.label;
            uint64_t bal = __ballot(m);
            if (!bal) continue;
            const double pv = m ? (double)vp[(uint32_t)y * (uint32_t)nx + (uint32_t)x] * w : 0.0;      // variable * weight_grid (:886 / :892)
            while (bal) {
                const int b = __builtin_ctzll(bal);
                bal &= bal - 1;
                const double q = __shfl(pv, b);
                s += q;
                sy += q * (double)y;
                sx += q * (double)(x0 + b);
            }
        }
    }
    __threadfence();
    if (lane == 0) {
        CtkLifeExact r;
        r.area = dev_np_sum(gw, pos);
        r.swv = dev_np_sum(gp, pos);
        r.s = s; r.sy = sy; r.sx = sx;
        out[blockIdx.x] = r;
    }
}
