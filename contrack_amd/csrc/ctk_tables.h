// ctk_tables.h -- component / pair / seam tables exchanged between the HIP stages and the host
// resolve step (internal; the public ABI is include/contrack_hip.h).
#pragma once
#include <stdint.h>
#include <stddef.h>

#define CTK_BLOB_MAGIC 0x314b544e4f43ull /* "CONTK1" */
#define CTK_LIMB_BITS_MIN 31     /* limb width when the row weights span <= 62 bits (every float32-latitude grid) */
#define CTK_LIMB_BITS_MAX 46     /* widest limb: n * limb with n <= 65535 pixels of a run must stay below 2^62 */
#define CTK_AMBIG_ULPS 64       /* a rounded area sum whose fraction lies this close to `overlap` is re-evaluated in numpy's order */

// One record per (component at t, component at t-1) co-occurrence; duplicates of the same (t,c,d) may
// occur (partial sums) and are additive.  lo/hi are the two limb sums of  sum_y n(y) * W[y]  with
// W[y] = wlo[y] + (whi[y] << limb_bits) the row weight scaled to an integer (see ctk_weights_to_limbs: limb_bits is 31
// unless the weights span more than 62 bits -- float64 latitudes with exact poles -- and is then chosen per grid).
struct CtkPair {
    uint32_t t;      // timestep of c (shard-local in blobs, global inside the resolver)
    uint32_t c;      // no-wrap 2-D component id at t   (0-based, raster order within the timestep)
    uint32_t d;      // no-wrap 2-D component id at t-1
    uint32_t pad;
    int64_t lo, hi;
};

// One record per (t, y) whose two seam pixels (x = 0 and x = nx-1) are both foreground.
struct CtkSeam {
    uint32_t t, y;
    uint32_t cl, cr; // no-wrap component ids of the pixel at x = 0 / x = nx-1
};

// bbox-confined relabel operation of contrack.py:753-763, in execution order.
struct CtkOp {
    int32_t hi, lo;            // pixels labelled hi inside the box become lo
    int32_t t0, t1;            // box of label hi on the fresh 3-D labelling (inclusive, GLOBAL t)
    int32_t y0, y1, x0, x1;
};

// Candidate record of the seam driver: rows y0..y1 (yy = y0 | y1 << 16) of timestep t whose seam pixels carry the pair of
// fresh labels (ll at x = 0, lr at x = nx-1) -- as dense ids of the labels that occur in such records.
struct CtkCand {
    int32_t t, yy, ll, lr;
};

struct CtkBlobHeader {
    uint64_t magic;
    int64_t T;
    int32_t ny, nx;
    int32_t wshift;
    int32_t has_prev;
    int32_t limb_bits;         // width of the low limb of every area sum in this blob
    int32_t pad0;
    int64_t ncomps, npairs, nseams;
    int64_t npairs_grouped;   // the first npairs_grouped pair records are grouped per timestep (pair_base / pair_cnt)
    // followed (each section 8-byte aligned) by
    //   uint32_t ncomp[T]
    //   uint32_t comp_mrep[ncomps]     id (within the timestep) of the smallest member of the seam-merged component
    //   uint16_t comp_box[ncomps][4]   y0, y1, x0, x1 (inclusive)
    //   int64_t  comp_area[ncomps][2]  limb sums of the component's own area
    //   CtkPair  pairs[npairs]
    //   CtkSeam  seams[nseams]         (t, y) order
    //   uint32_t pair_base[T], pair_cnt[T]   records of timestep t: pairs[pair_base[t] .. +pair_cnt[t]) (grouped part)
};

static inline size_t ctk_align8(size_t n) { return (n + 7) & ~(size_t)7; }

static inline size_t ctk_blob_bytes(int64_t T, int64_t ncomps, int64_t npairs, int64_t nseams)
{
    return sizeof(CtkBlobHeader) + ctk_align8((size_t)T * 4) + ctk_align8((size_t)ncomps * 4) +
           ctk_align8((size_t)ncomps * 8) + (size_t)ncomps * 16 + (size_t)npairs * sizeof(CtkPair) +
           (size_t)nseams * sizeof(CtkSeam) + 2 * ctk_align8((size_t)T * 4);
}

// Provider of numpy-order area sums for ONE seam-merged component (host resolver, single shard): used for the rare overlap
// decisions whose exactly accumulated sums had to be rounded AND land within rounding distance of the threshold
// (exact areas).  `comp` is the component's representative id within timestep t; kept_prev(d) tells whether the
// (no-wrap) component d of timestep t-1 survived the filter.  out = {areacon, areaover_forward, areaover_backward}
// as np.sum returns them (contrack.py:717-719).
#ifdef __cplusplus
#include <functional>
struct CtkExactAreas {
    virtual ~CtkExactAreas() {}
    virtual bool sums(int64_t t, uint32_t comp, const std::function<bool(uint32_t)> &kept_prev, double out[3]) = 0;
    // lowest set bit over the integer row weights (ctk_weights_to_limbs scale): a sum whose integer needs more than
    // 53 bits above it can be rounded somewhere inside numpy's reduction even if its total is representable
    virtual int min_lsb() const = 0;
};
#endif

struct ctk_result {
    int nshards;
    int64_t T;                 // total timesteps
    int64_t ncomps;            // total no-wrap components
    int64_t n_labels;          // labels of the fresh 3-D labelling (contrack.py:748)
    int64_t n_complex, n_ambiguous;
    int64_t n_exact;           // decisions re-evaluated with numpy-order sums (CtkExactAreas)
    int64_t *shard_comp_off;   // [nshards+1] offsets into comp_label
    int64_t *shard_t_off;      // [nshards+1]
    int32_t *comp_label;       // [ncomps]  0 = filtered out; L>0 = final id; -L = needs per-pixel fold from 3-D label L
    CtkOp *ops;                // [nops] execution order
    int64_t nops;
};
