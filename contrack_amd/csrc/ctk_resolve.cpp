// ctk_resolve.cpp -- host-side, GPU-free resolution of the sequential parts of run_contrack on
// component tables (no pixels are touched here).
//
//   overlap filter recurrence      contrack/contrack.py:706-742   (two-sided rule reads the ALREADY
//                                  filtered previous step, :719 after :729-737)
//   3-D labelling ids              contrack/contrack.py:747-751   (scipy numbering: first pixel in C
//                                  raster order; 8-conn in plane + same pixel at t+-1)
//   bbox-confined seam merges      contrack/contrack.py:753-763   (boxes computed once, :753)
//
// The HIP stages deliver, per timestep, the 2-D components WITHOUT longitude wrap (these are the
// nodes of the 3-D labelling, which does not wrap either), the id of the seam-merged component each
// belongs to (contrack.py:691-698), exact integer area sums, the (component@t, component@t-1) pixel
// co-occurrence areas, and the rows whose two seam pixels are both set.
#include "ctk_tables.h"
#include "ctk_seam.h"
#include "../../include/contrack_hip_debug.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

// ---------------------------------------------------------------------------------------------
// thread-local error text (shared by every translation unit of the library)
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
extern "C" const char *ctk_last_error(void) { return g_err; }
int ctk_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---------------------------------------------------------------------------------------------
// exact limb sums -> float64.  value = (hi * 2^lb + lo) * 2^-wshift, rounded ONCE to nearest-even.
// numpy evaluates the same sum pairwise in float64 (contrack.py:717-719); both agree whenever the
// exact sum is representable, which holds for every component whose fraction can tie with the
// overlap threshold in practice (DESIGN.md "exact areas").  *inexact reports a rounded result.
// ---------------------------------------------------------------------------------------------
static double limbs_to_double(int64_t lo, int64_t hi, int wshift, int lb, bool *inexact)
{
    __int128 v = (__int128)hi * ((__int128)1 << lb) + (__int128)lo;
    if (v == 0) return 0.0;
    bool neg = v < 0;
    unsigned __int128 a = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    uint64_t top = (uint64_t)(a >> 64), bot = (uint64_t)a;
    int msb = top ? 127 - __builtin_clzll(top) : 63 - __builtin_clzll(bot);
    double d;
    if (msb <= 52) {
        d = (double)bot;
    } else {
        int sh = msb - 52;
        unsigned __int128 q = a >> sh;
        unsigned __int128 rem = a & ((((unsigned __int128)1) << sh) - 1);
        unsigned __int128 half = ((unsigned __int128)1) << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) q += 1;
        if (rem != 0 && inexact) *inexact = true;
        d = std::ldexp((double)(uint64_t)q, sh);
    }
    d = std::ldexp(d, -wshift);
    return neg ? -d : d;
}

namespace {

struct Box3 {
    int32_t t0, t1, y0, y1, x0, x1;
};

struct Uf {
    std::vector<int64_t> p;
    explicit Uf(int64_t n) : p((size_t)n) { for (int64_t i = 0; i < n; i++) p[(size_t)i] = i; }
    int64_t find(int64_t i)
    {
        int64_t r = i;
        while (p[(size_t)r] != r) r = p[(size_t)r];
        while (p[(size_t)i] != r) { int64_t n = p[(size_t)i]; p[(size_t)i] = r; i = n; }
        return r;
    }
    void unite(int64_t a, int64_t b)
    {
        a = find(a); b = find(b);
        if (a < b) p[(size_t)b] = a; else if (b < a) p[(size_t)a] = b;   // smaller index wins => root = first raster pixel's component
    }
};

struct View {   // one shard's blob, parsed
    const CtkBlobHeader *h;
    const uint32_t *ncomp;
    const uint32_t *mrep;
    const uint16_t *box;
    const int64_t *area;
    const CtkPair *pairs;
    const CtkSeam *seams;
};

bool parse(const void *blob, size_t nbytes, View &v)
{
    if (nbytes < sizeof(CtkBlobHeader)) return false;
    const char *p = (const char *)blob;
    v.h = (const CtkBlobHeader *)p;
    if (v.h->magic != CTK_BLOB_MAGIC || v.h->T < 0 || v.h->ncomps < 0 || v.h->npairs < 0 || v.h->nseams < 0) return false;
    if (ctk_blob_bytes(v.h->T, v.h->ncomps, v.h->npairs, v.h->nseams) > nbytes) return false;
    p += sizeof(CtkBlobHeader);
    v.ncomp = (const uint32_t *)p; p += ctk_align8((size_t)v.h->T * 4);
    v.mrep = (const uint32_t *)p;  p += ctk_align8((size_t)v.h->ncomps * 4);
    v.box = (const uint16_t *)p;   p += ctk_align8((size_t)v.h->ncomps * 8);
    v.area = (const int64_t *)p;   p += (size_t)v.h->ncomps * 16;
    v.pairs = (const CtkPair *)p;  p += (size_t)v.h->npairs * sizeof(CtkPair);
    v.seams = (const CtkSeam *)p;
    return true;
}

inline bool in_box(const CtkOp &o, int32_t t, int32_t y, int32_t x)
{
    return t >= o.t0 && t <= o.t1 && y >= o.y0 && y <= o.y1 && x >= o.x0 && x <= o.x1;
}

}  // namespace

// numpy's float64 add.reduce over a contiguous 1-D array (numpy/_core/src/umath/loops_utils.h.src, *_pairwise_sum, and
// the 8192-element inner-loop chunks of the reduction machinery), restated: eight running sums over blocks of up to 128
// elements, halving above that, chunk results added to the accumulator in order.
static double np_pairwise(const double *a, size_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (size_t i = 0; i < n; i++) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        size_t i = 8;
        for (; i + 8 <= n; i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}
double ctk_np_sum(const double *a, size_t n)
{
    double res = 0.0;
    for (size_t i = 0; i < n; i += 8192) res += np_pairwise(a + i, std::min<size_t>(8192, n - i));
    return res;
}

int ctk_resolve_ex(const void *const *blobs, const size_t *nbytes, int nshards, double overlap, int twosided, CtkExactAreas *exact,
                   ctk_result **out);

// test hook (GPU-free): the numpy-order sum used for decisions on rounded area sums
extern "C" double ctk_debug_np_sum(const double *a, size_t n) { return ctk_np_sum(a, n); }

extern "C" int ctk_resolve(const void *const *blobs, const size_t *nbytes, int nshards, double overlap,
                           int twosided, ctk_result **out)
{
    return ctk_resolve_ex(blobs, nbytes, nshards, overlap, twosided, nullptr, out);
}

int ctk_resolve_ex(const void *const *blobs, const size_t *nbytes, int nshards, double overlap, int twosided, CtkExactAreas *exact,
                   ctk_result **out)
{
    if (!blobs || !nbytes || nshards <= 0 || !out) return ctk_set_error(CTK_E_INVALID, "ctk_resolve: bad arguments");
    *out = nullptr;
    try {
        std::vector<View> sv((size_t)nshards);
        int64_t T = 0, NC = 0, NP = 0, NS = 0;
        for (int s = 0; s < nshards; s++) {
            if (!parse(blobs[s], nbytes[s], sv[(size_t)s])) return ctk_set_error(CTK_E_INVALID, "ctk_resolve: blob %d is malformed", s);
            const CtkBlobHeader *h = sv[(size_t)s].h;
            if (h->ny != sv[0].h->ny || h->nx != sv[0].h->nx || h->wshift != sv[0].h->wshift || h->limb_bits != sv[0].h->limb_bits)
                return ctk_set_error(CTK_E_INVALID, "ctk_resolve: shards disagree on grid / weight scale");
            T += h->T; NC += h->ncomps; NP += h->npairs; NS += h->nseams;
        }
        const int nx = sv[0].h->nx;
        const int wshift = sv[0].h->wshift, lb = sv[0].h->limb_bits;
        if (lb < CTK_LIMB_BITS_MIN || lb > CTK_LIMB_BITS_MAX) return ctk_set_error(CTK_E_INVALID, "ctk_resolve: limb width %d out of range", lb);
        if (NC >= ((int64_t)1 << 31) - 2) return ctk_set_error(CTK_E_RANGE, "ctk_resolve: %lld components exceed int32 ids", (long long)NC);

        // ---- flatten: global timestep index, component offsets ---------------------------------
        std::vector<int64_t> coff((size_t)T + 1, 0);          // components of global step t: [coff[t], coff[t+1])
        std::vector<uint32_t> mrep((size_t)NC);
        std::vector<uint16_t> box((size_t)NC * 4);
        std::vector<int64_t> area((size_t)NC * 2);
        std::vector<int64_t> shard_t((size_t)nshards + 1, 0), shard_c((size_t)nshards + 1, 0);
        {
            int64_t t = 0, c = 0;
            for (int s = 0; s < nshards; s++) {
                const View &v = sv[(size_t)s];
                shard_t[(size_t)s] = t; shard_c[(size_t)s] = c;
                int64_t csum = 0;
                for (int64_t k = 0; k < v.h->T; k++) { coff[(size_t)(t + k)] = c + csum; csum += v.ncomp[k]; }
                if (csum != v.h->ncomps) return ctk_set_error(CTK_E_INVALID, "ctk_resolve: blob %d component counts inconsistent", s);
                if (v.h->ncomps) {
                    memcpy(&mrep[(size_t)c], v.mrep, (size_t)v.h->ncomps * 4);
                    memcpy(&box[(size_t)c * 4], v.box, (size_t)v.h->ncomps * 8);
                    memcpy(&area[(size_t)c * 2], v.area, (size_t)v.h->ncomps * 16);
                }
                t += v.h->T; c += v.h->ncomps;
            }
            coff[(size_t)T] = c; shard_t[(size_t)nshards] = t; shard_c[(size_t)nshards] = c;
        }
        auto ncomp_at = [&](int64_t t) { return coff[(size_t)t + 1] - coff[(size_t)t]; };

        // pairs grouped by global t (counting sort); a pair at t links t and t-1
        std::vector<int64_t> poff((size_t)T + 2, 0);
        std::vector<CtkPair> pairs((size_t)NP);
        {
            for (int s = 0; s < nshards; s++) {
                const View &v = sv[(size_t)s];
                for (int64_t k = 0; k < v.h->npairs; k++) {
                    int64_t t = shard_t[(size_t)s] + v.pairs[k].t;
                    if (v.pairs[k].t >= (uint64_t)v.h->T || t < 1 || (v.pairs[k].t == 0 && !v.h->has_prev))
                        return ctk_set_error(CTK_E_INVALID, "ctk_resolve: blob %d pair %lld has a bad timestep", s, (long long)k);
                    poff[(size_t)t + 1]++;
                }
            }
            for (int64_t t = 0; t <= T; t++) poff[(size_t)t + 1] += poff[(size_t)t];
            std::vector<int64_t> cur(poff.begin(), poff.end() - 1);
            for (int s = 0; s < nshards; s++) {
                const View &v = sv[(size_t)s];
                for (int64_t k = 0; k < v.h->npairs; k++) {
                    CtkPair p = v.pairs[k];
                    int64_t t = shard_t[(size_t)s] + p.t;
                    if (p.c >= (uint64_t)ncomp_at(t) || p.d >= (uint64_t)ncomp_at(t - 1))
                        return ctk_set_error(CTK_E_INVALID, "ctk_resolve: blob %d pair %lld refers to a missing component", s, (long long)k);
                    p.t = (uint32_t)t;
                    pairs[(size_t)cur[(size_t)t]++] = p;
                }
            }
        }
        // seams sorted by (t, y)
        std::vector<CtkSeam> seams((size_t)NS);
        {
            int64_t n = 0;
            for (int s = 0; s < nshards; s++) {
                const View &v = sv[(size_t)s];
                for (int64_t k = 0; k < v.h->nseams; k++) {
                    CtkSeam q = v.seams[k];
                    if (q.t >= (uint64_t)v.h->T) return ctk_set_error(CTK_E_INVALID, "ctk_resolve: blob %d seam %lld has a bad timestep", s, (long long)k);
                    int64_t t = shard_t[(size_t)s] + q.t;
                    if (q.cl >= (uint64_t)ncomp_at(t) || q.cr >= (uint64_t)ncomp_at(t))
                        return ctk_set_error(CTK_E_INVALID, "ctk_resolve: blob %d seam %lld refers to a missing component", s, (long long)k);
                    q.t = (uint32_t)t;
                    seams[(size_t)n++] = q;
                }
            }
            std::sort(seams.begin(), seams.end(), [](const CtkSeam &a, const CtkSeam &b) { return a.t != b.t ? a.t < b.t : a.y < b.y; });
        }

        // ---- step 3: overlap filter on seam-merged components (contrack.py:706-742) ------------
        // per merged component (indexed by its representative's global index): area, forward overlap
        std::vector<int64_t> A((size_t)NC * 2, 0), F((size_t)NC * 2, 0), B((size_t)NC * 2, 0);
        for (int64_t t = 0; t < T; t++)
            for (int64_t g = coff[(size_t)t]; g < coff[(size_t)t + 1]; g++) {
                if (mrep[(size_t)g] >= (uint64_t)ncomp_at(t)) return ctk_set_error(CTK_E_INVALID, "ctk_resolve: bad merged-component id");
                int64_t r = coff[(size_t)t] + mrep[(size_t)g];
                A[(size_t)r * 2] += area[(size_t)g * 2];
                A[(size_t)r * 2 + 1] += area[(size_t)g * 2 + 1];
            }
        // forward overlap of a component at t = sum of its co-occurrences with ANY component at t+1
        // (plane t+1 is not filtered yet when t is visited, contrack.py:718)
        for (int64_t t = 1; t < T; t++)
            for (int64_t k = poff[(size_t)t]; k < poff[(size_t)t + 1]; k++) {
                const CtkPair &p = pairs[(size_t)k];
                int64_t r = coff[(size_t)t - 1] + mrep[(size_t)(coff[(size_t)t - 1] + p.d)];
                F[(size_t)r * 2] += p.lo;
                F[(size_t)r * 2 + 1] += p.hi;
            }
        std::vector<uint8_t> keep((size_t)NC, 1);          // per merged representative; members look it up
        int64_t n_ambiguous = 0, n_exact_fixups = 0;
        auto rep_of = [&](int64_t t, uint32_t c) { return coff[(size_t)t] + mrep[(size_t)(coff[(size_t)t] + c)]; };
        for (int64_t t = 1; t < T - 1; t++) {
            // backward overlap: co-occurrences with components of t-1 that SURVIVED the filter (contrack.py:719)
            for (int64_t k = poff[(size_t)t]; k < poff[(size_t)t + 1]; k++) {
                const CtkPair &p = pairs[(size_t)k];
                if (!keep[(size_t)rep_of(t - 1, p.d)]) continue;
                int64_t r = rep_of(t, p.c);
                B[(size_t)r * 2] += p.lo;
                B[(size_t)r * 2 + 1] += p.hi;
            }
            for (int64_t g = coff[(size_t)t]; g < coff[(size_t)t + 1]; g++) {
                if ((int64_t)(coff[(size_t)t] + mrep[(size_t)g]) != g) continue;       // representatives only
                bool inexact = false;
                double areacon = limbs_to_double(A[(size_t)g * 2], A[(size_t)g * 2 + 1], wshift, lb, &inexact);
                double fwd = limbs_to_double(F[(size_t)g * 2], F[(size_t)g * 2 + 1], wshift, lb, &inexact);
                double bwd = limbs_to_double(B[(size_t)g * 2], B[(size_t)g * 2 + 1], wshift, lb, &inexact);
                double inv = 1.0 / areacon;                 // contrack.py:721-722: reciprocal, then multiply
                double fb = inv * bwd;
                double ff = inv * fwd;
                if (exact && !inexact) {
                    // numpy can round INSIDE its reduction even when the total is representable: any sum that spans more than
                    // 53 bits above the smallest weight bit (components with pole-row pixels) counts as rounded
                    __int128 v = (__int128)A[(size_t)g * 2 + 1] * ((__int128)1 << lb) + (__int128)A[(size_t)g * 2];
                    unsigned __int128 m = v < 0 ? (unsigned __int128)(-v) : (unsigned __int128)v;
                    int bl = 0;
                    while (m) { bl++; m >>= 1; }
                    if (bl + 1 - exact->min_lsb() > 53) inexact = true;
                }
                if (inexact) {
                    // a rounded sum can differ from numpy's pairwise result by a few ulp: for decisions that sit that close
                    // to the threshold, take numpy's sums from the provider (single shard), else report them
                    // (DESIGN.md "exact areas"; a zero sum is exact)
                    double tol = CTK_AMBIG_ULPS * 2.220446049250313e-16 * std::fabs(overlap);
                    if ((ff != 0 && std::fabs(ff - overlap) <= tol) || (twosided && fb != 0 && std::fabs(fb - overlap) <= tol)) {
                        double s3[3];
                        if (exact && nshards == 1 &&
                            exact->sums(t, (uint32_t)(g - coff[(size_t)t]), [&](uint32_t d) { return keep[(size_t)rep_of(t - 1, d)] != 0; }, s3)) {
                            areacon = s3[0]; fwd = s3[1]; bwd = s3[2];
                            inv = 1.0 / areacon; fb = inv * bwd; ff = inv * fwd;
                            n_exact_fixups++;
                        } else n_ambiguous++;
                    }
                }
                bool kill = false;
                if (twosided) {
                    if (fb != 0 && ff != 0) { if (fb < overlap || ff < overlap) kill = true; }
                    if (fb != 0 && ff == 0) { if (fb < overlap) kill = true; }
                    if (fb == 0 && ff != 0) { if (ff < overlap) kill = true; }
                } else {
                    if (ff < overlap) kill = true;
                }
                if (kill) keep[(size_t)g] = 0;
            }
        }
        auto kept = [&](int64_t t, uint32_t c) { return keep[(size_t)rep_of(t, c)] != 0; };

        // ---- step 4: 3-D labelling of the surviving components (contrack.py:747-751) -----------
        Uf uf(NC);
        for (int64_t t = 1; t < T; t++)
            for (int64_t k = poff[(size_t)t]; k < poff[(size_t)t + 1]; k++) {
                const CtkPair &p = pairs[(size_t)k];
                if (kept(t, p.c) && kept(t - 1, p.d)) uf.unite(coff[(size_t)t] + p.c, coff[(size_t)t - 1] + p.d);
            }
        std::vector<int32_t> lab((size_t)NC, 0);
        int32_t nlab = 0;
        for (int64_t t = 0; t < T; t++)
            for (int64_t g = coff[(size_t)t]; g < coff[(size_t)t + 1]; g++) {
                if (!keep[(size_t)(coff[(size_t)t] + mrep[(size_t)g])]) continue;
                int64_t r = uf.find(g);
                if (r == g) lab[(size_t)g] = ++nlab; else lab[(size_t)g] = lab[(size_t)r];
            }
        // boxes of the fresh labels: find_objects ONCE (contrack.py:753)
        std::vector<Box3> bx((size_t)nlab + 1, Box3{INT32_MAX, -1, INT32_MAX, -1, INT32_MAX, -1});
        for (int64_t t = 0; t < T; t++)
            for (int64_t g = coff[(size_t)t]; g < coff[(size_t)t + 1]; g++) {
                int32_t l = lab[(size_t)g];
                if (!l) continue;
                Box3 &b = bx[(size_t)l];
                const uint16_t *q = &box[(size_t)g * 4];
                b.t0 = std::min<int32_t>(b.t0, (int32_t)t); b.t1 = std::max<int32_t>(b.t1, (int32_t)t);
                b.y0 = std::min<int32_t>(b.y0, q[0]); b.y1 = std::max<int32_t>(b.y1, q[1]);
                b.x0 = std::min<int32_t>(b.x0, q[2]); b.x1 = std::max<int32_t>(b.x1, q[3]);
            }

        // ---- step 4b: sequential seam merges confined to the ORIGINAL box of the larger label -----
        // Pixels are never touched: an ordered op list is kept; the current label of a pixel is the fold
        // of the ops over its fresh label.
        std::vector<CtkOp> ops;
        std::vector<std::vector<int32_t>> ops_of((size_t)nlab + 1);     // op indices by `hi`, ascending
        std::vector<int32_t> inflow((size_t)nlab + 1, -1);              // last recorded op whose `lo` is the label
        auto fold_pixel = [&](int32_t l, int32_t t, int32_t y, int32_t x) {
            int32_t s = 0;
            for (;;) {
                const std::vector<int32_t> &lst = ops_of[(size_t)l];
                bool moved = false;
                for (int32_t idx : lst) {
                    if (idx < s) continue;
                    if (in_box(ops[(size_t)idx], t, y, x)) { l = ops[(size_t)idx].lo; s = idx + 1; moved = true; break; }
                }
                if (!moved) return l;
            }
        };
        for (const CtkSeam &q : seams) {
            int64_t t = q.t;
            if (!kept(t, q.cl)) continue;                    // both seam pixels belong to one merged component
            int32_t p0 = fold_pixel(lab[(size_t)(coff[(size_t)t] + q.cl)], (int32_t)t, (int32_t)q.y, 0);
            int32_t p1 = fold_pixel(lab[(size_t)(coff[(size_t)t] + q.cr)], (int32_t)t, (int32_t)q.y, nx - 1);
            if (p0 == p1) continue;
            int32_t hi = std::max(p0, p1), lo = std::min(p0, p1);
            // nothing has flowed into hi since its last op: no pixel labelled hi is left inside box[hi], the
            // relabel of contrack.py:759/763 finds nothing -- not recorded (keeps the chains short)
            if (!ops_of[(size_t)hi].empty() && inflow[(size_t)hi] < ops_of[(size_t)hi].back()) continue;
            const Box3 &b = bx[(size_t)hi];
            inflow[(size_t)lo] = (int32_t)ops.size();
            ops_of[(size_t)hi].push_back((int32_t)ops.size());
            ops.push_back(CtkOp{hi, lo, b.t0, b.t1, b.y0, b.y1, b.x0, b.x1});
        }

        // ---- final id per component: fold with the component's box; mixed containment => per pixel --
        ctk_result *res = new ctk_result();
        memset(res, 0, sizeof(*res));
        res->nshards = nshards; res->T = T; res->ncomps = NC; res->n_labels = nlab; res->n_ambiguous = n_ambiguous; res->n_exact = n_exact_fixups;
        res->shard_comp_off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nshards + 1));
        res->shard_t_off = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nshards + 1));
        res->comp_label = (int32_t *)malloc(sizeof(int32_t) * (size_t)(NC > 0 ? NC : 1));
        res->nops = (int64_t)ops.size();
        res->ops = (CtkOp *)malloc(sizeof(CtkOp) * (ops.size() ? ops.size() : 1));
        if (!res->shard_comp_off || !res->shard_t_off || !res->comp_label || !res->ops) {
            ctk_result_free(res);
            return ctk_set_error(CTK_E_NOMEM, "ctk_resolve: out of memory");
        }
        memcpy(res->shard_comp_off, shard_c.data(), sizeof(int64_t) * ((size_t)nshards + 1));
        memcpy(res->shard_t_off, shard_t.data(), sizeof(int64_t) * ((size_t)nshards + 1));
        if (!ops.empty()) memcpy(res->ops, ops.data(), sizeof(CtkOp) * ops.size());
        int64_t n_complex = 0;
        for (int64_t t = 0; t < T; t++)
            for (int64_t g = coff[(size_t)t]; g < coff[(size_t)t + 1]; g++) {
                int32_t l = lab[(size_t)g];
                if (!l || ops.empty()) { res->comp_label[(size_t)g] = l; continue; }
                const uint16_t *q = &box[(size_t)g * 4];
                int32_t cur = l, s = 0;
                bool complex_ = false;
                for (bool again = true; again && !complex_;) {
                    again = false;
                    for (int32_t idx : ops_of[(size_t)cur]) {
                        if (idx < s) continue;
                        const CtkOp &o = ops[(size_t)idx];
                        bool t_in = (int32_t)t >= o.t0 && (int32_t)t <= o.t1;
                        bool inside = t_in && q[0] >= o.y0 && q[1] <= o.y1 && q[2] >= o.x0 && q[3] <= o.x1;
                        bool disjoint = !t_in || q[1] < o.y0 || q[0] > o.y1 || q[3] < o.x0 || q[2] > o.x1;
                        if (inside) { cur = o.lo; s = idx + 1; again = true; break; }
                        if (!disjoint) { complex_ = true; break; }
                    }
                }
                if (complex_) { res->comp_label[(size_t)g] = -l; n_complex++; }
                else res->comp_label[(size_t)g] = cur;
            }
        res->n_complex = n_complex;
        *out = res;
        return CTK_OK;
    } catch (const std::bad_alloc &) {
        return ctk_set_error(CTK_E_NOMEM, "ctk_resolve: out of memory");
    }
}

extern "C" void ctk_result_free(ctk_result *r)
{
    if (!r) return;
    free(r->shard_comp_off); free(r->shard_t_off); free(r->comp_label); free(r->ops);
    delete r;
}

extern "C" int ctk_result_arrays(const ctk_result *r, const int32_t **comp_label, int64_t *ncomps, const void **ops, int64_t *nops,
                                 const int64_t **shard_comp_off, const int64_t **shard_t_off)
{
    if (!r) return ctk_set_error(CTK_E_INVALID, "ctk_result_arrays: null result");
    if (comp_label) *comp_label = r->comp_label;
    if (ncomps) *ncomps = r->ncomps;
    if (ops) *ops = r->ops;
    if (nops) *nops = r->nops;
    if (shard_comp_off) *shard_comp_off = r->shard_comp_off;
    if (shard_t_off) *shard_t_off = r->shard_t_off;
    return CTK_OK;
}

extern "C" int ctk_result_nshards(const ctk_result *r)
{
    return r ? r->nshards : ctk_set_error(CTK_E_INVALID, "ctk_result_nshards: null result");
}

extern "C" int ctk_result_info(const ctk_result *r, int64_t *n_labels, int64_t *n_ops, int64_t *n_complex,
                               int64_t *n_ambiguous, int64_t *n_components)
{
    if (!r) return ctk_set_error(CTK_E_INVALID, "ctk_result_info: null result");
    if (n_labels) *n_labels = r->n_labels;
    if (n_ops) *n_ops = r->nops;
    if (n_complex) *n_complex = r->n_complex;
    if (n_ambiguous) *n_ambiguous = r->n_ambiguous;
    if (n_components) *n_components = r->ncomps;
    return CTK_OK;
}

// ---------------------------------------------------------------------------------------------
// Row weights -> exact integers.  Every finite float32 w[y] is m * 2^e with a 24-bit integer m; with
// S = -min e all weights become integers W[y] = w[y] * 2^S, split into two signed limbs
// (W = lo + hi * 2^L, both limbs carry the sign).  L = 31 whenever the weights span at most 62 bits: sums
// sum_y n(y) * W[y]  of up to 2^31 pixels then fit int64 per limb and are exact.  Latitudes held in float64 with exact
// poles give pole-row weights of ~1e-13 next to ~1e4 (cos(pi/2) = 6e-17 in float64): ~78 bits.  Then L = ceil(span / 2),
// which is possible because no area sum ever covers more than npix = ny * nx pixels: L + ceil(log2 npix) <= 62.
// The float32 values are the ones the host computed as contrack.py:703-704 does (including slightly negative pole rows).
// ---------------------------------------------------------------------------------------------
extern "C" int ctk_weights_to_limbs(const float *wrow, int ny, int64_t npix, int64_t *wlo, int64_t *whi, int32_t *wshift, int32_t *limb_bits)
{
    if (!wrow || !wlo || !whi || !wshift || !limb_bits || ny < 1 || npix < 1) return ctk_set_error(CTK_E_INVALID, "ctk_weights_to_limbs: bad arguments");
    std::vector<uint32_t> mant((size_t)ny);
    std::vector<int> ex((size_t)ny);
    int emin = INT32_MAX;
    for (int y = 0; y < ny; y++) {
        uint32_t bits;
        memcpy(&bits, &wrow[y], 4);
        uint32_t efield = (bits >> 23) & 0xffu, frac = bits & 0x7fffffu;
        if (efield == 0xffu) return ctk_set_error(CTK_E_INVALID, "row weight %d is not finite", y);
        uint32_t m;
        int e;
        if (efield == 0) { m = frac; e = -149; } else { m = frac | 0x800000u; e = (int)efield - 150; }
        if (m == 0) { mant[(size_t)y] = 0; ex[(size_t)y] = 0; continue; }
        int tz = __builtin_ctz(m);
        m >>= tz; e += tz;
        mant[(size_t)y] = m; ex[(size_t)y] = e;
        emin = std::min(emin, e);
    }
    if (emin == INT32_MAX) emin = 0;                     // all weights zero
    int span = 0, span_row = 0;
    for (int y = 0; y < ny; y++) {
        if (mant[(size_t)y] == 0) continue;
        const int bitlen = 32 - __builtin_clz(mant[(size_t)y]) + (ex[(size_t)y] - emin);
        if (bitlen > span) { span = bitlen; span_row = y; }
    }
    int lb = CTK_LIMB_BITS_MIN;
    if (span > 2 * CTK_LIMB_BITS_MIN) {
        int lg = 0;
        while (((int64_t)1 << lg) < npix) lg++;
        lb = (span + 1) / 2;
        if (lb > CTK_LIMB_BITS_MAX || lb + lg > 62)
            return ctk_set_error(CTK_E_RANGE, "row weights span %d bits (row %d): two %d-bit limbs cannot sum %lld pixels exactly in int64",
                                 span, span_row, lb, (long long)npix);
    }
    for (int y = 0; y < ny; y++) {
        if (mant[(size_t)y] == 0) { wlo[y] = 0; whi[y] = 0; continue; }
        const unsigned __int128 W = (unsigned __int128)mant[(size_t)y] << (ex[(size_t)y] - emin);
        const int64_t lo = (int64_t)(uint64_t)(W & ((((unsigned __int128)1) << lb) - 1)), hi = (int64_t)(uint64_t)(W >> lb);
        const bool neg = std::signbit(wrow[y]);
        wlo[y] = neg ? -lo : lo;
        whi[y] = neg ? -hi : hi;
    }
    *wshift = -emin;
    *limb_bits = lb;
    return CTK_OK;
}

// ---------------------------------------------------------------------------------------------
// test hook (GPU-free): the label numbering across time-shard boundaries (boundary_resolve, ctk_seam.h) on flat arrays.
// last_flat / halo_flat: the per-rank records one after the other (nlast[q] / nh[q] entries each).
// ---------------------------------------------------------------------------------------------
extern "C" int ctk_debug_boundary_resolve(int world, const int32_t *nlast, const int32_t *nh, const int32_t *nroots, const int32_t *last_flat,
                                          const int32_t *halo_flat, int64_t *off, int32_t *last_label_flat, int32_t *halo_label_flat,
                                          int32_t *n_absorbed)
{
    if (world < 1 || !nlast || !nh || !nroots || !off) return ctk_set_error(CTK_E_INVALID, "ctk_debug_boundary_resolve: bad arguments");
    std::vector<BoundaryIn> in((size_t)world);
    size_t lo = 0, ho = 0;
    for (int q = 0; q < world; q++) {
        in[(size_t)q].nlast = nlast[q]; in[(size_t)q].nh = nh[q]; in[(size_t)q].nroots = nroots[q];
        in[(size_t)q].last = last_flat + lo; in[(size_t)q].halo = halo_flat + ho;
        lo += (size_t)nlast[q]; ho += (size_t)nh[q];
    }
    BoundaryOut out;
    if (!boundary_resolve(in, out)) return ctk_set_error(CTK_E_INVALID, "ctk_debug_boundary_resolve: contradictory records");
    memcpy(off, out.off.data(), sizeof(int64_t) * ((size_t)world + 1));
    lo = ho = 0;
    for (int q = 0; q < world; q++) {
        if (last_label_flat && nlast[q]) memcpy(last_label_flat + lo, out.last_label[(size_t)q].data(), (size_t)nlast[q] * 4);
        if (halo_label_flat && nh[q]) memcpy(halo_label_flat + ho, out.halo_label[(size_t)q].data(), (size_t)nh[q] * 4);
        if (n_absorbed) n_absorbed[q] = (int32_t)out.absorbed[(size_t)q].size();
        lo += (size_t)nlast[q]; ho += (size_t)nh[q];
    }
    return CTK_OK;
}
