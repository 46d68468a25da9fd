// ctk_anom.hip -- SURVEY.md section 8(f) rows N2 / N3 on the device (included by ctk_api.hip):
//   N2  contrack.calc_clim / calc_anom (contrack/contrack.py:458-581): group means (day-of-year climatology), centred rolling
//       mean over the groups with the reference's fill, anomaly, centred rolling mean over time.  The anomaly slab can stay
//       resident in HBM for run_contrack (ctk_track_resident): the producer of the hot path's input no longer pays PCIe twice.
//   N3  the percentile threshold of README.rst:150-151: per grid point of a latitude band the q-quantile over time (exact
//       order statistics by radix selection, numpy's linear interpolation), then the mean over the band.
// The reference evaluates these with xarray, which cannot be installed in the build container: what is implemented is the
// numpy restatement in oracle/anom_port.py (PARITY UNPINNED, see its header); sums in float64, results rounded to the input's
// dtype where xarray keeps it.
#pragma once

// gfx950 only.  k_quantile alone keeps 66 KB of static LDS per workgroup (64 pixels x 257 histogram bins): more than the 64 KB a
// workgroup gets on gfx90a / gfx942.  The Makefile's ARCH is overridable for gfx950 variants (xnack / sramecc suffixes), not for
// other parts.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libcontrack_hip.so is written for gfx950 (MI355X): build with ARCH=gfx950"
#endif

template <typename VT>
__device__ __forceinline__ bool an_isnan(VT v) { return v != v; }

// clim_raw[g][p] = mean over the timesteps of group g, NaNs skipped (contrack.py:483)
template <typename VT>
__global__ __launch_bounds__(256) void k_clim_raw(const VT *__restrict__ x, const int32_t *__restrict__ tlist, const int32_t *__restrict__ goff, int64_t npix,
                                                  VT *__restrict__ raw)
{
    const int g = (int)blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    double s = 0.0;
    int c = 0;
    for (int k = goff[g]; k < goff[g + 1]; k++) {
        const VT v = x[(int64_t)tlist[k] * npix + p];
        if (!an_isnan(v)) { s += (double)v; c++; }
    }
    raw[(int64_t)g * npix + p] = c ? (VT)(s / c) : (VT)__builtin_nanf("");
}

// centred rolling mean over the groups, NaN -> mean of the last `window` groups (contrack.py:487-489)
template <typename VT>
__global__ __launch_bounds__(256) void k_clim_roll(const VT *__restrict__ raw, int G, int window, int64_t npix, VT *__restrict__ clim)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    double fs = 0.0;
    int fc = 0;
    for (int g = max(0, G - window); g < G; g++) {
        const VT v = raw[(int64_t)g * npix + p];
        if (!an_isnan(v)) { fs += (double)v; fc++; }
    }
    const VT fill = fc ? (VT)(fs / fc) : (VT)__builtin_nanf("");
    for (int g = (int)blockIdx.y; g < G; g += (int)gridDim.y) {
        const int lo = g - window / 2, hi = g + (window - 1) / 2;
        VT r = fill;
        if (lo >= 0 && hi < G) {
            double s = 0.0;
            for (int j = lo; j <= hi; j++) s += (double)raw[(int64_t)j * npix + p];        // (a NaN makes the window NaN)
            const VT m = (VT)(s / window);
            if (!an_isnan(m)) r = m;
        }
        clim[(int64_t)g * npix + p] = r;
    }
}

// anomaly + centred rolling mean over time (contrack.py:566-570); NaN where the window leaves the axis or holds a NaN
template <typename VT>
__global__ __launch_bounds__(256) void k_anom(const VT *__restrict__ x, const VT *__restrict__ clim, const int32_t *__restrict__ group, int64_t T, int64_t npix,
                                              int smooth, int tseg, VT *__restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int64_t t0 = (int64_t)blockIdx.y * tseg, t1 = min(T, t0 + tseg);
    for (int64_t t = t0; t < t1; t++) {
        const int64_t lo = t - smooth / 2, hi = t + (smooth - 1) / 2;
        VT r = (VT)__builtin_nanf("");
        if (lo >= 0 && hi < T) {
            double s = 0.0;
            for (int64_t j = lo; j <= hi; j++) s += (double)(VT)((double)x[j * npix + p] - (double)clim[(int64_t)group[j] * npix + p]);
            r = (VT)(s / smooth);
        }
        out[t * npix + p] = r;
    }
}

// ---- N3: q-quantile over time per grid point of rows [y0, y1), exact (radix selection on order-preserving integer keys) -----
__device__ __forceinline__ uint32_t an_key(float v) { const uint32_t u = __float_as_uint(v); return (u >> 31) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ uint64_t an_key(double v) { const uint64_t u = (uint64_t)__double_as_longlong(v); return (u >> 63) ? ~u : (u | 0x8000000000000000ull); }
__device__ __forceinline__ float an_unkey(uint32_t k) { return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ double an_unkey(uint64_t k) { return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k)); }

template <typename VT, typename KT>
__global__ __launch_bounds__(256) void k_quantile(const VT *__restrict__ x, int64_t T, int64_t npix, int64_t p0, int64_t nband, double q, double *__restrict__ out)
{
    __shared__ uint32_t hist[64][257];
    __shared__ uint32_t s_n[64], s_k[64];
    __shared__ KT s_prefix[64], s_next[64];
    __shared__ uint32_t s_le[64];
    const int px = (int)threadIdx.x & 63, tq = (int)threadIdx.x >> 6;
    const int64_t pb = (int64_t)blockIdx.x * 64 + px;
    const bool live = pb < nband;
    const VT *col = x + p0 + pb;
    if (threadIdx.x < 64) { s_n[px] = 0; s_prefix[px] = 0; s_next[px] = ~(KT)0; s_le[px] = 0; }
    __syncthreads();
    // values that count (NaNs are skipped, np.nanquantile)
    {
        uint32_t c = 0;
        if (live) for (int64_t t = tq; t < T; t += 4) c += an_isnan(col[t * npix]) ? 0u : 1u;
        if (c) atomicAdd(&s_n[px], c);
    }
    __syncthreads();
    if (threadIdx.x < 64) { const double h = ((double)s_n[px] - 1.0) * q; s_k[px] = s_n[px] ? (uint32_t)floor(h) : 0u; }
    constexpr int NB = (int)sizeof(KT);
    for (int b = NB - 1; b >= 0; b--) {
        for (int i = (int)threadIdx.x; i < 64 * 257; i += 256) (&hist[0][0])[i] = 0;
        __syncthreads();
        if (live && s_n[px]) {
            const KT pre = s_prefix[px];
            for (int64_t t = tq; t < T; t += 4) {
                const VT v = col[t * npix];
                if (an_isnan(v)) continue;
                const KT k = an_key(v);
                if (b == NB - 1 || (k >> (8 * (b + 1))) == pre) atomicAdd(&hist[px][(uint32_t)(k >> (8 * b)) & 255u], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64 && s_n[px]) {
            uint32_t k = s_k[px], cum = 0;
            int bin = 0;
            for (; bin < 256; bin++) { if (cum + hist[px][bin] > k) break; cum += hist[px][bin]; }
            s_prefix[px] = (s_prefix[px] << 8) | (KT)min(bin, 255);
            s_k[px] = k - cum;
        }
        __syncthreads();
    }
    // the next larger value and the number of values <= the selected one
    if (live && s_n[px]) {
        const KT sel = s_prefix[px];
        uint32_t le = 0;
        KT nx = ~(KT)0;
        for (int64_t t = tq; t < T; t += 4) {
            const VT v = col[t * npix];
            if (an_isnan(v)) continue;
            const KT k = an_key(v);
            if (k <= sel) le++; else if (k < nx) nx = k;
        }
        if (le) atomicAdd(&s_le[px], le);
        if (nx != ~(KT)0) atomicMin(&s_next[px], nx);
    }
    __syncthreads();
    if (threadIdx.x < 64 && live) {
        double r = __builtin_nan("");
        if (s_n[px]) {
            const double h = ((double)s_n[px] - 1.0) * q;
            const uint32_t lo = (uint32_t)floor(h);
            const double t = h - (double)lo;
            const double a = (double)an_unkey(s_prefix[px]);
            const double bb = (s_le[px] > lo + 1 || lo + 1 >= s_n[px]) ? a : (double)an_unkey(s_next[px]);
            // numpy's _lerp: a + (b - a) * t, taken from the other end for t >= 0.5
            const double d = bb - a;
            r = a + d * t;
            if (t >= 0.5) r = bb - d * (1.0 - t);
            if (t == 0.0 || d == 0.0) r = a;
        }
        out[pb] = r;
    }
}

// mean of the non-NaN entries, fixed order (one workgroup, pairwise tree)
__global__ __launch_bounds__(1024) void k_nanmean(const double *__restrict__ v, int64_t n, double *__restrict__ out)
{
    __shared__ double ss[1024];
    __shared__ unsigned long long sc[1024];
    double s = 0.0;
    unsigned long long c = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) { const double a = v[i]; if (a == a) { s += a; c++; } }
    ss[threadIdx.x] = s; sc[threadIdx.x] = c;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) { ss[threadIdx.x] += ss[threadIdx.x + d]; sc[threadIdx.x] += sc[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sc[0] ? ss[0] / (double)sc[0] : __builtin_nan("");
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
template <typename VT>
static int anom_impl(ctk_handle *h, const VT *x_host, int64_t T, int ny, int nx, const int32_t *group, int ngroups, int window, int smooth,
                     const VT *clim_in, VT *anom_out, VT *clim_out, int keep_resident)
{
    if (!h || !x_host || !group || T < 1 || ny < 1 || nx < 1 || ngroups < 1 || window < 1 || smooth < 1 || (!anom_out && !clim_out && !keep_resident))
        return ctk_set_error(CTK_E_INVALID, "ctk_anom: bad arguments");
    for (int64_t t = 0; t < T; t++) if (group[t] < 0 || group[t] >= ngroups) return ctk_set_error(CTK_E_INVALID, "ctk_anom: group[%lld] = %d outside 0..%d", (long long)t, group[t], ngroups - 1);
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int64_t npix = (int64_t)ny * nx;
    const size_t esz = sizeof(VT), n = (size_t)T * (size_t)npix;
    CTKCHK(ensure(h, h->io_in, n * esz));
    CTKCHK(ensure(h, h->an_out, n * esz));
    CTKCHK(ensure(h, h->an_clim, (size_t)ngroups * npix * esz));
    CTKCHK(ensure(h, h->an_raw, (size_t)ngroups * npix * esz));
    CTKCHK(ensure(h, h->an_idx, ((size_t)2 * T + ngroups + 2) * 4));
    HIPCHK(hipMemcpy(h->io_in.p, x_host, n * esz, hipMemcpyHostToDevice));
    // timesteps sorted by group, group offsets, group of every timestep
    std::vector<int32_t> idx((size_t)2 * T + ngroups + 1);
    int32_t *tlist = idx.data(), *goff = tlist + T, *grp = goff + ngroups + 1;
    std::fill(goff, goff + ngroups + 1, 0);
    for (int64_t t = 0; t < T; t++) goff[group[t] + 1]++;
    for (int g = 0; g < ngroups; g++) goff[g + 1] += goff[g];
    {
        std::vector<int32_t> cur(goff, goff + ngroups);
        for (int64_t t = 0; t < T; t++) tlist[cur[(size_t)group[t]]++] = (int32_t)t;
    }
    memcpy(grp, group, (size_t)T * 4);
    HIPCHK(hipMemcpy(h->an_idx.p, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    const int32_t *d_tlist = P<int32_t>(h->an_idx), *d_goff = d_tlist + T, *d_grp = d_goff + ngroups + 1;
    const unsigned gx = (unsigned)((npix + 255) / 256);
    if (clim_in) {
        HIPCHK(hipMemcpy(h->an_clim.p, clim_in, (size_t)ngroups * npix * esz, hipMemcpyHostToDevice));
    } else {
        k_clim_raw<VT><<<dim3(gx, (unsigned)ngroups), 256, 0, s>>>((const VT *)h->io_in.p, d_tlist, d_goff, npix, (VT *)h->an_raw.p);
        k_clim_roll<VT><<<dim3(gx, (unsigned)std::min(ngroups, 64)), 256, 0, s>>>((const VT *)h->an_raw.p, ngroups, window, npix, (VT *)h->an_clim.p);
        HIPCHK(hipGetLastError());
    }
    if (clim_out) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipMemcpy(clim_out, h->an_clim.p, (size_t)ngroups * npix * esz, hipMemcpyDeviceToHost)); }
    if (anom_out || keep_resident) {
        h->an_T = -1; h->an_gen++;                                     // the resident slab (if any) is being overwritten
        const int tseg = 32;
        k_anom<VT><<<dim3(gx, (unsigned)((T + tseg - 1) / tseg)), 256, 0, s>>>((const VT *)h->io_in.p, (const VT *)h->an_clim.p, d_grp, T, npix, smooth, tseg,
                                                                             (VT *)h->an_out.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        h->an_T = T; h->an_ny = ny; h->an_nx = nx; h->an_f64 = sizeof(VT) == 8;
        if (anom_out) {
            if (!h->bounce) h->bounce = new (std::nothrow) BouncePool();
            if (!h->bounce || !bounce_copy(*h->bounce, h->device, h->an_out.p, anom_out, n * esz, false))
                HIPCHK(hipMemcpy(anom_out, h->an_out.p, n * esz, hipMemcpyDeviceToHost));
        }
    }
    if ((anom_out || keep_resident) && !keep_resident) h->an_T = -1;      // (a climatology-only call leaves an earlier resident slab alone)
    return CTK_OK;
}

extern "C" int ctk_anom_f32(ctk_handle *h, const float *x, int64_t T, int ny, int nx, const int32_t *group, int ngroups, int window, int smooth,
                            const float *clim_in, float *anom_out, float *clim_out, int keep_resident)
{
    return anom_impl<float>(h, x, T, ny, nx, group, ngroups, window, smooth, clim_in, anom_out, clim_out, keep_resident);
}
extern "C" int ctk_anom_f64(ctk_handle *h, const double *x, int64_t T, int ny, int nx, const int32_t *group, int ngroups, int window, int smooth,
                            const double *clim_in, double *anom_out, double *clim_out, int keep_resident)
{
    return anom_impl<double>(h, x, T, ny, nx, group, ngroups, window, smooth, clim_in, anom_out, clim_out, keep_resident);
}

// WHICH slab is resident: a number that changes whenever the resident slab is written (any ctk_anom_* call that produces
// anomalies) or dropped (ctk_release_io).  A caller that left a slab resident remembers the number and runs on the slab only
// while it is unchanged -- two class instances sharing one handle cannot take each other's anomalies for their own.
extern "C" int ctk_resident_anom_generation(ctk_handle *h, uint64_t *generation)
{
    if (!h || !generation) return ctk_set_error(CTK_E_INVALID, "null argument");
    *generation = h->an_gen;
    return CTK_OK;
}

// shape of the anomaly slab kept in HBM by the last ctk_anom_* call with keep_resident (T = -1: none)
extern "C" int ctk_resident_anom(ctk_handle *h, int64_t *T, int *ny, int *nx, int *is_f64)
{
    if (!h) return ctk_set_error(CTK_E_INVALID, "null handle");
    if (T) *T = h->an_T;
    if (ny) *ny = h->an_ny;
    if (nx) *nx = h->an_nx;
    if (is_f64) *is_f64 = h->an_f64 ? 1 : 0;
    return CTK_OK;
}

// run_contrack on the resident anomaly slab: no H2D; flag goes to the host
extern "C" int ctk_track_resident(ctk_handle *h, const double *thr, int cmp_op, const float *wrow, double overlap, int persistence, int twosided,
                                  int32_t *flag, int64_t *n_tracked)
{
    if (!h || !flag) return ctk_set_error(CTK_E_INVALID, "null argument");
    if (h->an_T < 1) return ctk_set_error(CTK_E_STATE, "ctk_track_resident: no anomaly slab is resident (ctk_anom_* with keep_resident)");
    HIPCHK(hipSetDevice(h->device));
    return track_to_host(h, h->an_out.p, h->an_f64, h->an_T, h->an_ny, h->an_nx, thr, cmp_op, wrow, overlap, persistence, twosided, flag, n_tracked, nullptr);
}

template <typename VT, typename KT>
static int percentile_impl(ctk_handle *h, const VT *x_host, int64_t T, int ny, int nx, int y0, int y1, double q, double *out)
{
    if (!h || !out || T < 1 || ny < 1 || nx < 1 || y0 < 0 || y1 > ny || y0 >= y1 || !(q >= 0.0 && q <= 1.0)) return ctk_set_error(CTK_E_INVALID, "ctk_percentile: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int64_t npix = (int64_t)ny * nx, nband = (int64_t)(y1 - y0) * nx;
    const VT *x_dev;
    if (x_host) {
        CTKCHK(ensure(h, h->io_in, (size_t)T * npix * sizeof(VT)));
        HIPCHK(hipMemcpy(h->io_in.p, x_host, (size_t)T * npix * sizeof(VT), hipMemcpyHostToDevice));
        x_dev = (const VT *)h->io_in.p;
    } else {
        if (h->an_T != T || h->an_ny != ny || h->an_nx != nx || h->an_f64 != (sizeof(VT) == 8)) return ctk_set_error(CTK_E_STATE, "ctk_percentile: no matching anomaly slab is resident");
        x_dev = (const VT *)h->an_out.p;
    }
    CTKCHK(ensure(h, h->an_raw, ((size_t)nband + 8) * 8));
    double *qv = P<double>(h->an_raw);
    k_quantile<VT, KT><<<(unsigned)((nband + 63) / 64), 256, 0, s>>>(x_dev, T, npix, (int64_t)y0 * nx, nband, q, qv);
    k_nanmean<<<1, 1024, 0, s>>>(qv, nband, qv + nband);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, qv + nband, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CTK_OK;
}
extern "C" int ctk_percentile_f32(ctk_handle *h, const float *x, int64_t T, int ny, int nx, int y0, int y1, double q, double *out)
{
    return percentile_impl<float, uint32_t>(h, x, T, ny, nx, y0, y1, q, out);
}
extern "C" int ctk_percentile_f64(ctk_handle *h, const double *x, int64_t T, int ny, int nx, int y0, int y1, double q, double *out)
{
    return percentile_impl<double, uint64_t>(h, x, T, ny, nx, y0, y1, q, out);
}
