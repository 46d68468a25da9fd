/*
 * contrack_oracle.c -- CPU restatement of ConTrack's run_contrack hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path; it is never linked
 * into, imported by or called from the product (contrack_amd/).  Only tests/, the smoke test
 * in __graft_entry__.py and bench.py's cpu_baseline leg may load it.
 *
 * Pinning: validated bit-for-bit against the *imported, unmodified* reference module
 * (/root/reference/contrack/contrack.py run under tests/minixr.py) on the reference's own
 * test slab and on seeded synthetic slabs -- see tests/golden/make_golden.py and
 * tests/test_oracle_golden.py.
 *
 * It works on pixels, exactly as the reference does, and follows the reference line by line
 * in *behaviour* (not in code: the reference is numpy/scipy Python):
 *
 *   step 1  threshold                 contrack/contrack.py:646-674
 *   step 2  2-D labelling (8-conn)    contrack/contrack.py:684-687  (scipy.ndimage.label, structure with
 *                                      only the middle plane set; labels numbered by first pixel in C
 *                                      raster order -- scipy/ndimage/_measurements.py:43-236)
 *   step 2b longitude seam merge      contrack/contrack.py:691-698
 *   step 3  area-overlap filter       contrack/contrack.py:703-742  (np.sum == numpy pairwise summation,
 *                                      numpy/_core/src/umath/loops_utils.h.src DOUBLE_pairwise_sum)
 *   step 4  binarise + 3-D labelling  contrack/contrack.py:747-751
 *   step 4b seam merge inside bbox    contrack/contrack.py:753-763  (scipy.ndimage.find_objects once)
 *   step 4c persistence               contrack/contrack.py:765-772
 *
 * Third-party arithmetic restated here: scipy.ndimage.label / find_objects (scipy is unpinned in
 * the reference's requirements.txt:2; docs/environment.yml pins 1.5.2; validated here against
 * scipy 1.15.3) and numpy's pairwise float64 sum (numpy 2.2.6).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -o liboracle.so contrack_oracle.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---------------------------------------------------------------------------------------------
 * numpy pairwise summation of a contiguous double vector (what np.sum does on the 1-D gathered
 * weight array at contrack.py:717-719).
 * ------------------------------------------------------------------------------------------- */
static double np_pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* np.sum / np.add.reduce over a 1-D array (numpy 2.2.6): the reduction iterator hands the inner
 * loop at most `bufsize` = 8192 elements at a time; each chunk is summed pairwise and added to the
 * running result, which starts at 0 (probe: tests/test_oracle_golden.py). */
#define NP_BUFSIZE 8192
static double np_sum(const double *a, int64_t n)
{
    double res = 0.0;
    for (int64_t i = 0; i < n; i += NP_BUFSIZE) {
        int64_t m = n - i < NP_BUFSIZE ? n - i : NP_BUFSIZE;
        res += np_pairwise_sum(a + i, m);
    }
    return res;
}

double orc_np_sum(const double *a, int64_t n) { return np_sum(a, n); }

/* ---------------------------------------------------------------------------------------------
 * step 1: threshold.  cmp_op: 0 '>=', 1 '<=', 2 '>', 3 '<'.  The compare is done in double on
 * (double)anom vs thr[t]; the caller rounds thr to float32 first when the reference would have
 * compared in float32 (Python-number threshold against a float32 array, contrack.py:665).  NaN
 * compares false for every operator.
 * ------------------------------------------------------------------------------------------- */
int orc_threshold(const float *anom, int64_t T, int ny, int nx, const double *thr, int cmp_op,
                  uint8_t *mask)
{
    int64_t npl = (int64_t)ny * nx;
    for (int64_t t = 0; t < T; t++) {
        double th = thr[t];
        const float *a = anom + t * npl;
        uint8_t *m = mask + t * npl;
        for (int64_t i = 0; i < npl; i++) {
            double v = (double)a[i];
            int b;
            switch (cmp_op) {
            case 0: b = v >= th; break;
            case 1: b = v <= th; break;
            case 2: b = v > th; break;
            case 3: b = v < th; break;
            default: return -1;
            }
            m[i] = (uint8_t)b;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * scipy.ndimage.label restated: connected components of a (T,ny,nx) binary array under
 *   temporal == 0 : 8-connectivity inside each plane, no link across planes (contrack.py:684-686)
 *   temporal == 1 : the same plus the same pixel at t-1 / t+1               (contrack.py:748-750)
 * Labels are 1..N in the order in which each component's first pixel is met in C raster order.
 * Implementation: union-find with "smaller index wins", so the root IS the first raster pixel.
 * ------------------------------------------------------------------------------------------- */
static int64_t uf_find(int64_t *p, int64_t i)
{
    int64_t r = i;
    while (p[r] != r) r = p[r];
    while (p[i] != r) { int64_t n = p[i]; p[i] = r; i = n; }
    return r;
}
static void uf_union(int64_t *p, int64_t a, int64_t b)
{
    a = uf_find(p, a); b = uf_find(p, b);
    if (a < b) p[b] = a; else if (b < a) p[a] = b;
}

int64_t orc_label(const uint8_t *mask, int64_t T, int ny, int nx, int temporal, int32_t *lab)
{
    int64_t npl = (int64_t)ny * nx, n = T * npl;
    int64_t *p = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!p) return -1;
    for (int64_t i = 0; i < n; i++) p[i] = i;
    for (int64_t t = 0; t < T; t++)
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++) {
                int64_t i = t * npl + (int64_t)y * nx + x;
                if (!mask[i]) continue;
                if (x > 0 && mask[i - 1]) uf_union(p, i, i - 1);
                if (y > 0) {
                    if (mask[i - nx]) uf_union(p, i, i - nx);
                    if (x > 0 && mask[i - nx - 1]) uf_union(p, i, i - nx - 1);
                    if (x < nx - 1 && mask[i - nx + 1]) uf_union(p, i, i - nx + 1);
                }
                if (temporal && t > 0 && mask[i - npl]) uf_union(p, i, i - npl);
            }
    int64_t next = 0;
    for (int64_t i = 0; i < n; i++) {
        if (!mask[i]) { lab[i] = 0; continue; }
        int64_t r = uf_find(p, i);
        if (r == i) lab[i] = (int32_t)(++next);
        else lab[i] = lab[r];            /* r < i, already numbered */
    }
    free(p);
    return next;
}

/* ---------------------------------------------------------------------------------------------
 * step 2b: 2-D longitude seam merge (contrack.py:691-698).  Row by row; when both seam pixels of
 * a row are labelled and differ, the larger label is rewritten to the smaller in the WHOLE
 * timestep.
 * ------------------------------------------------------------------------------------------- */
void orc_seam2d(int32_t *lab, int64_t T, int ny, int nx)
{
    int64_t npl = (int64_t)ny * nx;
    for (int64_t t = 0; t < T; t++) {
        int32_t *f = lab + t * npl;
        for (int y = 0; y < ny; y++) {
            int32_t a = f[(int64_t)y * nx], b = f[(int64_t)y * nx + nx - 1];
            if (a > 0 && b > 0 && a != b) {
                int32_t hi = a > b ? a : b, lo = a > b ? b : a;
                for (int64_t i = 0; i < npl; i++) if (f[i] == hi) f[i] = lo;
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * step 3: overlap filter (contrack.py:703-742).  wrow[y] is the float32 row weight the host
 * computed exactly as contrack.py:703-704 does; weight_grid holds those values as float64.
 * For tt = 1..T-2 (ascending), every label present in plane tt is visited in ascending label
 * order (find_objects order); the three sums run over the label's bounding box in raster order
 * through np.sum (chunked pairwise sum); plane tt-1 has already been filtered, plane tt+1 has not.
 * ------------------------------------------------------------------------------------------- */
typedef struct { int y0, y1, x0, x1; } box2;

int orc_overlap_filter(int32_t *lab, int64_t T, int ny, int nx, const float *wrow,
                       double overlap, int twosided)
{
    int64_t npl = (int64_t)ny * nx;
    double *buf_c = (double *)malloc(sizeof(double) * (size_t)npl);
    double *buf_f = (double *)malloc(sizeof(double) * (size_t)npl);
    double *buf_b = (double *)malloc(sizeof(double) * (size_t)npl);
    if (!buf_c || !buf_f || !buf_b) return -1;
    for (int64_t tt = 1; tt < T - 1; tt++) {
        int32_t *f = lab + tt * npl;
        const int32_t *fn = lab + (tt + 1) * npl, *fp = lab + (tt - 1) * npl;
        int32_t lmin = INT32_MAX, lmax = 0;
        for (int64_t i = 0; i < npl; i++)
            if (f[i] > 0) { if (f[i] < lmin) lmin = f[i]; if (f[i] > lmax) lmax = f[i]; }
        if (lmax == 0) continue;
        int64_t nl = (int64_t)lmax - lmin + 1;
        box2 *bx = (box2 *)malloc(sizeof(box2) * (size_t)nl);
        if (!bx) return -1;
        for (int64_t k = 0; k < nl; k++) { bx[k].y0 = ny; bx[k].y1 = -1; bx[k].x0 = nx; bx[k].x1 = -1; }
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++) {
                int32_t l = f[(int64_t)y * nx + x];
                if (l <= 0) continue;
                box2 *b = &bx[l - lmin];
                if (y < b->y0) b->y0 = y;
                if (y > b->y1) b->y1 = y;
                if (x < b->x0) b->x0 = x;
                if (x > b->x1) b->x1 = x;
            }
        for (int64_t k = 0; k < nl; k++) {
            box2 b = bx[k];
            if (b.y1 < 0) continue;                      /* slice_ is None */
            int32_t label = (int32_t)(lmin + k);
            int64_t nc = 0, nf = 0, nb = 0;
            for (int y = b.y0; y <= b.y1; y++) {
                double w = (double)wrow[y];
                for (int x = b.x0; x <= b.x1; x++) {
                    int64_t i = (int64_t)y * nx + x;
                    if (f[i] != label) continue;
                    buf_c[nc++] = w;
                    if (fn[i] >= 1) buf_f[nf++] = w;
                    if (fp[i] >= 1) buf_b[nb++] = w;
                }
            }
            double areacon = np_sum(buf_c, nc);
            double area_fwd = np_sum(buf_f, nf);
            double area_bwd = np_sum(buf_b, nb);
            double inv = 1.0 / areacon;
            double fb = inv * area_bwd;
            double ff = inv * area_fwd;
            int kill = 0;
            if (twosided) {
                if (fb != 0 && ff != 0) { if (fb < overlap || ff < overlap) kill = 1; }
                if (fb != 0 && ff == 0) { if (fb < overlap) kill = 1; }
                if (fb == 0 && ff != 0) { if (ff < overlap) kill = 1; }
            } else {
                if (ff < overlap) kill = 1;
            }
            if (kill)
                for (int y = b.y0; y <= b.y1; y++)
                    for (int x = b.x0; x <= b.x1; x++) {
                        int64_t i = (int64_t)y * nx + x;
                        if (f[i] == label) f[i] = 0;
                    }
        }
        free(bx);
    }
    free(buf_c); free(buf_f); free(buf_b);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * scipy.ndimage.find_objects restated for a (T,ny,nx) label array: box[k] for label k+1,
 * empty (t1 < 0) when the label does not occur.
 * ------------------------------------------------------------------------------------------- */
typedef struct { int64_t t0, t1; int y0, y1, x0, x1; } box3;

static box3 *find_objects3(const int32_t *lab, int64_t T, int ny, int nx, int64_t *nlab)
{
    int64_t npl = (int64_t)ny * nx, n = T * npl;
    int32_t mx = 0;
    for (int64_t i = 0; i < n; i++) if (lab[i] > mx) mx = lab[i];
    *nlab = mx;
    box3 *bx = (box3 *)malloc(sizeof(box3) * (size_t)(mx > 0 ? mx : 1));
    if (!bx) return NULL;
    for (int64_t k = 0; k < mx; k++) { bx[k].t0 = T; bx[k].t1 = -1; bx[k].y0 = ny; bx[k].y1 = -1; bx[k].x0 = nx; bx[k].x1 = -1; }
    for (int64_t t = 0; t < T; t++)
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++) {
                int32_t l = lab[t * npl + (int64_t)y * nx + x];
                if (l <= 0) continue;
                box3 *b = &bx[l - 1];
                if (t < b->t0) b->t0 = t;
                if (t > b->t1) b->t1 = t;
                if (y < b->y0) b->y0 = y;
                if (y > b->y1) b->y1 = y;
                if (x < b->x0) b->x0 = x;
                if (x > b->x1) b->x1 = x;
            }
    return bx;
}

static void relabel_in_box(int32_t *lab, int ny, int nx, const box3 *b, int32_t from, int32_t to)
{
    int64_t npl = (int64_t)ny * nx;
    for (int64_t t = b->t0; t <= b->t1; t++)
        for (int y = b->y0; y <= b->y1; y++)
            for (int x = b->x0; x <= b->x1; x++) {
                int64_t i = t * npl + (int64_t)y * nx + x;
                if (lab[i] == from) lab[i] = to;
            }
}

/* step 4b (contrack.py:753-763): boxes computed ONCE on the fresh 3-D labelling; sequential over
 * (t, y); the relabel is confined to the original box of the larger label.  Returns the number of
 * merge operations performed (diagnostic). */
int64_t orc_seam3d(int32_t *lab, int64_t T, int ny, int nx)
{
    int64_t npl = (int64_t)ny * nx, nlab = 0, nops = 0;
    box3 *bx = find_objects3(lab, T, ny, nx, &nlab);
    if (!bx) return -1;
    for (int64_t t = 0; t < T; t++)
        for (int y = 0; y < ny; y++) {
            int64_t i0 = t * npl + (int64_t)y * nx, i1 = i0 + nx - 1;
            if (lab[i0] > 0 && lab[i1] > 0 && lab[i0] > lab[i1]) {
                relabel_in_box(lab, ny, nx, &bx[lab[i0] - 1], lab[i0], lab[i1]);
                nops++;
            }
            if (lab[i0] > 0 && lab[i1] > 0 && lab[i0] < lab[i1]) {
                relabel_in_box(lab, ny, nx, &bx[lab[i1] - 1], lab[i1], lab[i0]);
                nops++;
            }
        }
    free(bx);
    return nops;
}

/* step 4c (contrack.py:765-772): boxes recomputed; a label whose time extent is shorter than
 * `persistence` is erased. */
int orc_persistence(int32_t *lab, int64_t T, int ny, int nx, int persistence)
{
    int64_t nlab = 0;
    box3 *bx = find_objects3(lab, T, ny, nx, &nlab);
    if (!bx) return -1;
    for (int64_t k = 0; k < nlab; k++) {
        if (bx[k].t1 < 0) continue;
        if ((bx[k].t1 + 1 - bx[k].t0) < persistence)
            relabel_in_box(lab, ny, nx, &bx[k], (int32_t)(k + 1), 0);
    }
    free(bx);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Whole path (contrack.py:646-796) on a C-contiguous (T,ny,nx) float32 slab.
 *   flag_out : int32 (T,ny,nx); n_tracked = number of distinct non-zero ids (contrack.py:793).
 *   stage_out: optional (may be NULL) int32 (T,ny,nx) receiving the 2-D labels after step 2b
 *              (before the overlap filter), for staged parity tests.
 * ------------------------------------------------------------------------------------------- */
int orc_run_contrack(const float *anom, int64_t T, int ny, int nx, const double *thr, int cmp_op,
                     const float *wrow, double overlap, int persistence, int twosided,
                     int32_t *flag_out, int64_t *n_tracked, int32_t *stage_out)
{
    int64_t n = T * (int64_t)ny * nx;
    uint8_t *mask = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    if (!mask) return -1;
    if (orc_threshold(anom, T, ny, nx, thr, cmp_op, mask)) { free(mask); return -2; }
    if (orc_label(mask, T, ny, nx, 0, flag_out) < 0) { free(mask); return -1; }
    orc_seam2d(flag_out, T, ny, nx);
    if (stage_out) memcpy(stage_out, flag_out, sizeof(int32_t) * (size_t)n);
    if (orc_overlap_filter(flag_out, T, ny, nx, wrow, overlap, twosided)) { free(mask); return -1; }
    for (int64_t i = 0; i < n; i++) mask[i] = flag_out[i] >= 1;
    if (orc_label(mask, T, ny, nx, 1, flag_out) < 0) { free(mask); return -1; }
    if (orc_seam3d(flag_out, T, ny, nx) < 0) { free(mask); return -1; }
    if (orc_persistence(flag_out, T, ny, nx, persistence)) { free(mask); return -1; }
    /* len(np.unique(flag)) - 1  (contrack.py:793; note: off by one when no background pixel exists) */
    int32_t mx = 0;
    for (int64_t i = 0; i < n; i++) if (flag_out[i] > mx) mx = flag_out[i];
    uint8_t *seen = (uint8_t *)calloc((size_t)mx + 1, 1);
    if (!seen) { free(mask); return -1; }
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; i++) if (!seen[flag_out[i]]) { seen[flag_out[i]] = 1; cnt++; }
    if (n_tracked) *n_tracked = cnt - 1;
    free(seen); free(mask);
    return 0;
}
