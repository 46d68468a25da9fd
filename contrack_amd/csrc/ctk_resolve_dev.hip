// ctk_resolve_dev.hip -- device-side resolution of the component tables (included by ctk_api.hip).
//
// Same semantics as the GPU-free reference implementation in ctk_resolve.cpp (which stays the
// specification and is what the CPU tests exercise); here every data-parallel part runs as small
// grid-stride kernels on the tables already resident in HBM, and only the inherently sequential seam-merge
// driver (contrack.py:753-763, a few thousand candidate rows) runs on the host between two tiny copies.
//
//   R1  pair preparation        rep/global ids of both ends, forward overlap F (contrack.py:718)
//   R2  overlap filter          Jacobi iteration of keep[t] = f(keep[t-1]) (contrack.py:706-742): every pass
//                               recomputes the backward overlap from the previous pass' keep bits; a pass
//                               that changes nothing proves the fixed point = the sequential result.
//   R3  3-D labelling           lock-free union-find over surviving (c@t, d@t-1) links, roots ranked by a
//                               prefix sum = scipy's raster-order numbering (contrack.py:748-751)
//   R4  boxes of the fresh labels (find_objects once, contrack.py:753) + surviving seam rows in (t, y) order
//   --  host: sequential seam driver on {label pair, box} records  ->  ordered op list
//   R5  final id of every component (fold with box containment; mixed containment -> per-pixel fold later)
#pragma once

struct ResolveDev {
    // inputs (dense, (t, c) order)
    const uint32_t *ncomp, *cprefix;      // [T], [T+1]
    const uint32_t *mrep, *comp_t;        // [NC]
    const uint16_t *box;                  // [NC][4]
    const int64_t *A;                     // [NC][2]  merged area at representatives
    const CtkPair *pairs;
    const uint32_t *counters;             // pairs count at CTK_CNT_PAIRS
    uint32_t pair_cap;
    int64_t T;
    int32_t wshift, limb_bits;
    double overlap;
    int twosided;
    // work
    uint32_t *p_rc, *p_rd, *p_gc, *p_gd;  // [pair_cap]
    int64_t *F, *B;                       // [NC][2]
    double *inv, *ff;                     // [NC] 1/areacon and forward fraction of every representative (contrack.py:721-722)
    uint8_t *keep0, *keep1;               // [NC]
    uint32_t *changed;                    // [CTK_MAX_JACOBI + 1][CTK_CHG_SLOTS]: some keep bit flipped in that pass
    uint32_t *parent;                     // [NC]
    uint32_t *isroot, *rank;              // [NC], [NC+1]
    int32_t *lab;                         // [NC] fresh 3-D label of every component (0 = filtered out)
    int32_t *lab_root;                    // [NC] or nullptr: the root index k_rs_roots found (kept for k_fz_rank_mark)
    uint8_t *mark;                        // [NC+1] label occurs in a seam row with two different labels
    // labels that occur in candidate records get dense ids (claim order), so that the host seam driver works on
    // tables of a few thousand entries instead of one slot per fresh label
    uint32_t *dmap;                       // [NC+1] 0 = not a candidate label, else dense id + 1
    int32_t *dorig;                       // [dense] fresh label of the dense id
    int32_t *dbox;                        // [dense][6] its box
    uint32_t *dcount;                     // number of dense ids
    int32_t *op_first;                    // [NC+1] first op per label: reset to -1 here, filled by k_ops_ingest
    uint8_t *inex;                        // [NC] the component's own area or forward overlap is a rounded sum
    uint32_t *ambig;                      // set when a decision with a rounded sum lies within rounding distance of the threshold
    int minlsb;                           // lowest set bit over the integer row weights (numpy can round inside its reduction when a sum
                                          // spans more than 53 bits above it, even if the total is representable)
    const int32_t *next_tiny;             // [ny+1] first row >= y whose weight has bits so low that a float64 partial sum of it with other
                                          // rows can be inexact (pole rows); ny if none.  Components that touch no such row sum exactly in
                                          // ANY order: only the others are subject to the minlsb rule.
    uint32_t *touch;                      // [NC] at representatives: some member's box holds such a row
    // time shards (ctk_sharded.hip): components [0, *nh_ptr) are the HALO -- the previous shard's last timestep, "timestep -1"
    // (cprefix[-1] = 0, cprefix[0] = *nh_ptr); their keep bits are imported, they are nodes of the 3-D union-find but never
    // roots that take a number.  nh_ptr == nullptr: no halo.  The filter visits the local timesteps t_lo .. t_hi.
    const uint32_t *nh_ptr;
    int t_lo, t_hi;
    // Decisions on rounded area sums that land within rounding distance of the threshold (exact ties on components with
    // pole-row pixels) are re-evaluated with numpy-order sums (ctk_sharded.hip, "exact fix-up"): the pass records such
    // components once (ovr_slot[g] = 0x40000000 | list index) and, once the host has supplied {areacon, forward, backward}
    // as np.sum returns them (ovr_slot[g] = 0x80000000 | index), decides with those.  ovr_slot == nullptr: only *ambig is set.
    uint32_t *ovr_slot;                   // [NC]
    const double *ovr_val;                // [amb_cap][3]
    uint32_t *amb_cnt, *amb_list;
    uint32_t amb_cap;
    // fused one-call path (no host hand-off, ctk_seam_dev.hip); all nullptr elsewhere
    uint32_t *cl_parent;                  // [labels + 1] clusters of labels that share seam rows: union-find (reset here, k_rs_roots)
    int32_t *cl_tmin, *cl_tmax;           // [labels + 1] at cluster roots: timesteps that hold records of the cluster
    uint32_t *cl_nops;                    // [labels + 1] operations of the cluster (k_seam_driver)
    int32_t *lbox;                        // [labels + 1][6] boxes of the marked labels (k_fz_groups)
    int32_t *ext;                         // time extents of the ids, reset by k_rs_roots: ext[l] = min t, ext[ext_off + l] = max t
    int64_t ext_off;
    uint32_t *counters_w;                 // the write-stage counters are reset by k_rs_roots as well
    uint32_t *pstate;                     // [T + 1] k_rs_pass_sys: per timestep (iterations done) << 24 | changed bits; zeroed by k_rs_init
    // Inter-workgroup waits of the systolic filter kernels are BOUNDED (HIP promises nothing about dispatch order; a wait for a
    // workgroup that is not resident would be a hung GPU): a wait that lasts longer than spin_limit ticks of the 100 MHz wall clock
    // gives up -- CTK_POISON_SPIN in *poison, the wave publishes a "poisoned" state word so that nobody waits for IT, and the pass is
    // repeated with one launch per filter pass (no waits).  dbg_stall (test hook): 1 = the first workgroup arrives late (by
    // spin_limit / 4), 2 = it never publishes.
    uint64_t spin_limit;
    uint32_t *poison;
    int dbg_stall;
};
#define CTK_POISON_SPIN    4u    // an inter-workgroup wait of the systolic filter pass gave up (ResolveDev::spin_limit)
#define CTK_SPIN_LIMIT_TICKS 20000000ull      // 0.2 s of the 100 MHz wall clock: ~10^4 times the longest legitimate wait measured

// one poll of a bounded wait went by: true when the wait has lasted longer than `limit` ticks.  The clock is read once per 1024
// polls (the first time after ~1 ms of waiting: a pass that runs normally never reads it).
struct SpinGuard { uint64_t t0 = 0; uint32_t polls = 0; };
__device__ __forceinline__ bool spin_expired(SpinGuard &g, uint64_t limit)
{
    if ((++g.polls & 1023u) != 0u) return false;
    const uint64_t now = wall_clock64();
    if (g.t0 == 0) { g.t0 = now; return false; }
    return now - g.t0 > limit;
}
#define CTK_PSTATE_POISONED 0xffffffffu       // "published everything" for whoever waits: the pass is invalid anyway

#define CTK_PSTATE_STRIDE 32        // words between the per-timestep state words of k_rs_pass_sys: one 128-byte line each (the words are
                                    // polled with device-scope loads: neighbours in one line would all hit the same memory channel)
#define CTK_CHG_SLOTS 64            // 'changed' words per filter pass (= wave width: one ballot reads them)
#define CTK_MAX_JACOBI 240          // hard cap of filter passes on the device (then: host resolver)
#define CTK_JACOBI_ROUND 10         // passes launched per round before convergence is checked

// pair records: k in [0, ng) are the grouped ones pairs[k]; k in [ng, ng+nu) the ungrouped ones stored from the
// end of the buffer downwards (k_overlap)
__device__ __forceinline__ uint32_t dev_ngrouped(const ResolveDev &r)
{
    uint32_t n = r.counters[CTK_CNT_PAIRS];
    return n < r.pair_cap ? n : r.pair_cap;
}
__device__ __forceinline__ uint32_t dev_nungrouped(const ResolveDev &r)
{
    uint32_t n = r.counters[CTK_CNT_UPAIRS];
    return n < r.pair_cap ? n : r.pair_cap;
}
// The pair table overflowed (records were dropped or collided): its contents must not be interpreted.  Every
// resolver kernel returns at once; the host sees the same condition at its next sync and regrows the table.
__device__ __forceinline__ bool dev_tables_bad(const ResolveDev &r)
{
    return (r.counters[CTK_CNT_OVERFLOW] & CTK_OVF_PAIRS) != 0u ||
           (uint64_t)r.counters[CTK_CNT_PAIRS] + (uint64_t)r.counters[CTK_CNT_UPAIRS] > (uint64_t)r.pair_cap;
}
__device__ __forceinline__ uint32_t dev_npairs(const ResolveDev &r)
{
    if (dev_tables_bad(r)) return 0;
    return dev_ngrouped(r) + dev_nungrouped(r);
}
__device__ __forceinline__ const CtkPair &pair_at(const ResolveDev &r, uint32_t k, uint32_t ng)
{
    return k < ng ? r.pairs[k] : r.pairs[r.pair_cap - 1u - (k - ng)];
}
// index of pair record k (0 <= k < ng + nu) in the pair arrays p_rc / p_rd / p_gc / p_gd: where the record itself sits
__device__ __forceinline__ uint32_t pair_slot(const ResolveDev &r, uint32_t k, uint32_t ng) { return k < ng ? k : r.pair_cap - 1u - (k - ng); }
__device__ __forceinline__ uint32_t dev_ncomps(const ResolveDev &r) { return r.cprefix[r.T]; }

// exact limb sums -> float64, rounded once to nearest-even (identical to limbs_to_double in ctk_resolve.cpp)
__device__ inline double dev_limbs_to_double(int64_t lo, int64_t hi, int wshift, int lb, bool *inexact = nullptr)
{
    __int128 v = (__int128)hi * ((__int128)1 << lb) + (__int128)lo;
    if (v == 0) return 0.0;
    const bool neg = v < 0;
    unsigned __int128 a = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    const uint64_t top = (uint64_t)(a >> 64), bot = (uint64_t)a;
    const int msb = top ? 127 - __builtin_clzll(top) : 63 - __builtin_clzll(bot);
    double d;
    if (msb <= 52) {
        d = (double)bot;
    } else {
        const int sh = msb - 52;
        unsigned __int128 q = a >> sh;
        const unsigned __int128 rem = a & ((((unsigned __int128)1) << sh) - 1);
        const unsigned __int128 half = ((unsigned __int128)1) << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) q += 1;
        if (inexact && rem != 0) *inexact = true;
        d = ldexp((double)(uint64_t)q, sh);
    }
    d = ldexp(d, -wshift);
    return neg ? -d : d;
}

__device__ __forceinline__ uint32_t gfind(uint32_t *p, uint32_t i)
{
    for (;;) {
        uint32_t q = __hip_atomic_load(&p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q == i) return i;
        i = q;
    }
}

__global__ void k_rs_init(ResolveDev r)
{
    const uint32_t nc = dev_ncomps(r);
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nc; g += gridDim.x * blockDim.x) {
        r.F[2 * (int64_t)g] = 0; r.F[2 * (int64_t)g + 1] = 0;
        r.B[2 * (int64_t)g] = 0; r.B[2 * (int64_t)g + 1] = 0;
        r.keep0[g] = 1; r.keep1[g] = 1;
        r.touch[g] = 0;
        if (r.ovr_slot) r.ovr_slot[g] = 0;
        r.parent[g] = g;                                   // (k_rs_parent_init, for the first round)
    }
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < (CTK_MAX_JACOBI + 1) * CTK_CHG_SLOTS; i += blockDim.x) r.changed[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *r.ambig = 0; *r.dcount = 0; if (r.amb_cnt) *r.amb_cnt = 0; }
    if (r.pstate) for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= r.T; t += (int64_t)gridDim.x * blockDim.x) r.pstate[(size_t)t * CTK_PSTATE_STRIDE] = 0u;
}

__global__ void k_rs_parent_init(ResolveDev r)
{
    const uint32_t nc = dev_ncomps(r);
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nc; g += gridDim.x * blockDim.x) r.parent[g] = g;
}

__global__ void k_rs_pairs(ResolveDev r)
{
    const uint32_t np = dev_npairs(r), ng = dev_ngrouped(r);     // (the fused path does all of this in k_overlap / k_compact_init)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
        const CtkPair p = pair_at(r, i, ng);
        const uint32_t cb = r.cprefix[p.t], db = r.cprefix[(int32_t)p.t - 1];       // (timestep -1: the halo of a time shard)
        const uint32_t gc = cb + p.c, gd = db + p.d;
        const uint32_t rc = cb + r.mrep[gc], rd = db + r.mrep[gd];
        const uint32_t q = pair_slot(r, i, ng);
        r.p_gc[q] = gc; r.p_gd[q] = gd; r.p_rc[q] = rc; r.p_rd[q] = rd;
        // forward overlap of the EARLIER component: plane t is unfiltered when t-1 is visited (contrack.py:718)
        atomicAdd((unsigned long long *)&r.F[2 * (int64_t)rd], (unsigned long long)p.lo);
        atomicAdd((unsigned long long *)&r.F[2 * (int64_t)rd + 1], (unsigned long long)p.hi);
    }
    // seam-merged components that hold a row with very low weight bits (see ResolveDev::next_tiny)
    const uint32_t nc = dev_tables_bad(r) ? 0u : dev_ncomps(r);
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nc; g += gridDim.x * blockDim.x) {
        const uint16_t *q = r.box + 4 * (int64_t)g;
        if (r.next_tiny[q[0]] <= (int32_t)q[1]) {
            const uint32_t rep = r.cprefix[(int32_t)r.comp_t[g]] + r.mrep[g];
            if (r.touch[rep] == 0u) atomicOr(&r.touch[rep], 1u);
        }
    }
}

// k_rs_pairs on pair records in fixed per-timestep slots (k_overlap with pslot, time-shard path): thread = slot, + the ungrouped ones
__global__ void k_rs_pairs_slots(ResolveDev r, const uint32_t *__restrict__ pair_cnt, uint32_t pslot)
{
    if (!dev_tables_bad(r)) {
        const uint64_t nslots = (uint64_t)r.T * pslot;
        const uint32_t nu = dev_nungrouped(r);
        for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nslots + nu; k += (uint64_t)gridDim.x * blockDim.x) {
            uint32_t i;
            if (k < nslots) {
                const uint32_t t = (uint32_t)(k / pslot), j = (uint32_t)(k - (uint64_t)t * pslot);
                if (j >= pair_cnt[t]) continue;
                i = (uint32_t)k;
            } else i = r.pair_cap - 1u - (uint32_t)(k - nslots);
            const CtkPair p = r.pairs[i];
            const uint32_t cb = r.cprefix[p.t], db = r.cprefix[(int32_t)p.t - 1];
            const uint32_t gc = cb + p.c, gd = db + p.d;
            const uint32_t rc = cb + r.mrep[gc], rd = db + r.mrep[gd];
            r.p_gc[i] = gc; r.p_gd[i] = gd; r.p_rc[i] = rc; r.p_rd[i] = rd;
            atomicAdd((unsigned long long *)&r.F[2 * (int64_t)rd], (unsigned long long)p.lo);
            atomicAdd((unsigned long long *)&r.F[2 * (int64_t)rd + 1], (unsigned long long)p.hi);
        }
    }
    const uint32_t nc = dev_tables_bad(r) ? 0u : dev_ncomps(r);
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nc; g += gridDim.x * blockDim.x) {
        const uint16_t *q = r.box + 4 * (int64_t)g;
        if (r.next_tiny[q[0]] <= (int32_t)q[1]) {
            const uint32_t rep = r.cprefix[(int32_t)r.comp_t[g]] + r.mrep[g];
            if (r.touch[rep] == 0u) atomicOr(&r.touch[rep], 1u);
        }
    }
}

// 1/areacon and the forward fraction do not change between passes
__device__ __forceinline__ void dev_prep_comp(const ResolveDev &r, uint32_t g, double *inv_out, double *ff_out, bool *inex_out = nullptr)
{
        bool inexact = false;
        const double areacon = dev_limbs_to_double(r.A[2 * (int64_t)g], r.A[2 * (int64_t)g + 1], r.wshift, r.limb_bits, &inexact);
        const double fwd = dev_limbs_to_double(r.F[2 * (int64_t)g], r.F[2 * (int64_t)g + 1], r.wshift, r.limb_bits, &inexact);
        if (r.touch[g]) {
            const __int128 v = (__int128)r.A[2 * (int64_t)g + 1] * ((__int128)1 << r.limb_bits) + (__int128)r.A[2 * (int64_t)g];
            const unsigned __int128 m = v < 0 ? (unsigned __int128)(-v) : (unsigned __int128)v;
            const uint64_t top = (uint64_t)(m >> 64), bot = (uint64_t)m;
            const int bl = top ? 128 - __builtin_clzll(top) : (bot ? 64 - __builtin_clzll(bot) : 0);
            if (bl + 1 - r.minlsb > 53) inexact = true;
        }
        r.inex[g] = inexact ? 1 : 0;
        const double inv = 1.0 / areacon;                     // reciprocal, then multiply -- as the reference does
        r.inv[g] = inv;
        r.ff[g] = inv * fwd;
        *inv_out = inv; *ff_out = inv * fwd;
        if (inex_out) *inex_out = inexact;
}
__global__ void k_rs_prep(ResolveDev r)
{
    const uint32_t nc = dev_ncomps(r);
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nc; g += gridDim.x * blockDim.x) {
        double a, b;
        dev_prep_comp(r, g, &a, &b);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused filter pass: one 64-thread workgroup per timestep.  keep[] is updated IN PLACE (chaotic iteration:
// a workgroup may read its predecessor's bits from this pass or the previous one -- both are valid iterates,
// and a pass that changes nothing anywhere is the fixed point of keep[t] = f(keep[t-1]), i.e. the sequential
// result).  A timestep whose predecessor did not change in the previous pass is skipped.
//   tdirty[(it&1)][t] = keep bits of t changed in pass `it`
// Backward overlaps are accumulated in LDS (<= CTK_PASS_COMPS components per timestep) or, for larger
// timesteps, in the global scratch r.B.
// ------------------------------------------------------------------------------------------------
#define CTK_PASS_COMPS 512
// The workgroup's work is a chain of dependent loads; independent ones are issued together (three global-memory
// round trips instead of nine): [everything indexed by t] -> [pair records and component constants of the first 64
// pairs / components] -> [keep bits of the pairs' predecessors].
__global__ __launch_bounds__(64) void k_rs_pass(ResolveDev r, int it, const uint32_t *__restrict__ pair_base, const uint32_t *__restrict__ pair_cnt,
                                                uint8_t *__restrict__ tdirty)
{
    if (dev_tables_bad(r)) return;
    const int t = (int)blockIdx.x + r.t_lo;                    // one slab: timesteps 1 .. T-2 are filtered
    const int64_t T = r.T;
    // entry [t + 1] of two buffers of T + 1 flags ([0] = the halo timestep of a time shard)
    uint8_t *dcur = tdirty + (size_t)(it & 1) * (size_t)(T + 1) + 1, *dprev = tdirty + (size_t)((it & 1) ^ 1) * (size_t)(T + 1) + 1;
    const int lane = (int)threadIdx.x;
    // round trip 1
    const uint32_t ch_prev = it > 0 ? (__ballot(r.changed[(it - 1) * CTK_CHG_SLOTS + lane] != 0u) != 0ull ? 1u : 0u) : 1u;
    const uint8_t d_prev = it > 0 ? dprev[t - 1] : (uint8_t)1;
    const uint32_t cb = r.cprefix[t], ce = r.cprefix[t + 1];
    const uint32_t pb = pair_base[t], pn = pair_cnt[t];
    const uint32_t nu = dev_nungrouped(r);
    if (ch_prev == 0) return;                                  // fixed point reached in an earlier pass
    if (!d_prev) { if (lane == 0) dcur[t] = 0; return; }
    const uint32_t nct = ce - cb;
    __shared__ long long Bl[2 * CTK_PASS_COMPS];
    const bool lds = nct <= CTK_PASS_COMPS;
    long long *B = lds ? Bl : (long long *)(r.B + 2 * (int64_t)cb);
    if (lds) for (uint32_t c = lane; c < 2 * nct; c += 64) Bl[c] = 0;
    // round trip 2: first pair and first component of this lane
    uint8_t *keep = r.keep0;
    const bool has_p = (uint32_t)lane < pn, has_c = (uint32_t)lane < nct;
    const uint32_t k0 = pb + lane, g0 = cb + lane;
    const uint32_t rd0 = has_p ? r.p_rd[k0] : 0u, rc0 = has_p ? r.p_rc[k0] : 0u;
    CtkPair p0;
    if (has_p) p0 = r.pairs[k0]; else { p0.lo = 0; p0.hi = 0; }
    const uint32_t mrep0 = has_c ? r.mrep[g0] : 0xffffffffu;
    const double inv0 = has_c ? r.inv[g0] : 0.0, ff0 = has_c ? r.ff[g0] : 0.0;
    const uint8_t kold0 = has_c ? __hip_atomic_load(&keep[g0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
    // round trip 3
    const uint8_t kd0 = has_p ? __hip_atomic_load(&keep[rd0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
    __syncthreads();
    if (has_p && kd0) {
        const uint32_t c = rc0 - cb;
        atomicAdd((unsigned long long *)&B[2 * c], (unsigned long long)p0.lo);
        atomicAdd((unsigned long long *)&B[2 * c + 1], (unsigned long long)p0.hi);
    }
    for (uint32_t i = lane + 64; i < pn; i += 64) {            // timesteps with more than 64 pair records
        const uint32_t k = pb + i;
        if (!__hip_atomic_load(&keep[r.p_rd[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
        const CtkPair p = r.pairs[k];
        const uint32_t c = r.p_rc[k] - cb;
        atomicAdd((unsigned long long *)&B[2 * c], (unsigned long long)p.lo);
        atomicAdd((unsigned long long *)&B[2 * c + 1], (unsigned long long)p.hi);
    }
    if (nu) {                                                   // records that bypassed the hash table (rare)
        for (uint32_t i = lane; i < nu; i += 64) {
            const CtkPair p = r.pairs[r.pair_cap - 1u - i];
            if ((int)p.t != t) continue;
            if (!__hip_atomic_load(&keep[r.p_rd[r.pair_cap - 1u - i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
            const uint32_t c = r.p_rc[r.pair_cap - 1u - i] - cb;
            atomicAdd((unsigned long long *)&B[2 * c], (unsigned long long)p.lo);
            atomicAdd((unsigned long long *)&B[2 * c + 1], (unsigned long long)p.hi);
        }
    }
    __syncthreads();
    bool any = false;
    for (uint32_t c = lane; c < nct; c += 64) {
        const uint32_t g = cb + c;
        const bool first = c == (uint32_t)lane;
        long long blo, bhi;
        if (lds) { blo = Bl[2 * c]; bhi = Bl[2 * c + 1]; }
        else {
            blo = __hip_atomic_load(&B[2 * c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bhi = __hip_atomic_load(&B[2 * c + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&B[2 * c], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&B[2 * c + 1], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((first ? mrep0 : r.mrep[g]) != c) continue;         // representatives only
        bool inexact = r.inex[g] != 0;
        const double bwd = dev_limbs_to_double(blo, bhi, r.wshift, r.limb_bits, &inexact);
        double fb = (first ? inv0 : r.inv[g]) * bwd, ff = first ? ff0 : r.ff[g];
        const uint32_t os = r.ovr_slot ? r.ovr_slot[g] : 0u;
        if (os & 0x80000000u) {
            // numpy's own sums (contrack.py:717-719) for this component, for the current keep bits of its predecessors
            const double *v = r.ovr_val + 3 * (size_t)(os & 0x3fffffffu);
            const double inv = 1.0 / v[0];
            fb = inv * v[2]; ff = inv * v[1];
        } else if (inexact) {
            // numpy sums these float64 values pairwise; a rounded sum can differ from this exact-then-rounded one by a few
            // ulp: flag decisions that sit that close to the threshold (same rule as ctk_resolve.cpp, DESIGN.md "exact areas")
            const double tol = CTK_AMBIG_ULPS * 2.220446049250313e-16 * fabs(r.overlap);
            if ((ff != 0 && fabs(ff - r.overlap) <= tol) || (r.twosided && fb != 0 && fabs(fb - r.overlap) <= tol)) {    // (a zero sum is exact)
                *r.ambig = 1u;
                if (r.ovr_slot && os == 0u) {
                    const uint32_t idx = atomicAdd(r.amb_cnt, 1u);
                    if (idx < r.amb_cap) { r.amb_list[idx] = g; r.ovr_slot[g] = 0x40000000u | idx; }
                }
            }
        }
        bool kill = false;
        if (r.twosided) {
            if (fb != 0 && ff != 0) { if (fb < r.overlap || ff < r.overlap) kill = true; }
            if (fb != 0 && ff == 0) { if (fb < r.overlap) kill = true; }
            if (fb == 0 && ff != 0) { if (ff < r.overlap) kill = true; }
        } else {
            if (ff < r.overlap) kill = true;
        }
        const uint8_t k = kill ? 0 : 1;
        const uint8_t kold = first ? kold0 : keep[g];           // only this workgroup writes the bits of timestep t
        if (k != kold) { __hip_atomic_store(&keep[g], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); any = true; }
    }
    const bool wave_any = __ballot(any) != 0ull;
    if (lane == 0) {
        dcur[t] = wave_any ? 1 : 0;
        // "some bit flipped in pass it": a plain store into one of 64 slots.  (One atomicOr per workgroup on a single word
        // serialised in L2: 2705 of them were 32 of pass 0's 35 us.)
        if (wave_any) r.changed[it * CTK_CHG_SLOTS + (t & (CTK_CHG_SLOTS - 1))] = 1u;
    }
}

// ------------------------------------------------------------------------------------------------
// The same iteration in ONE launch ("systolic"): the workgroup of timestep t runs its iterations 0 .. K-1 itself; iteration k
// needs the predecessor's bits of iteration k-1 and nothing else, so it waits for ONE word -- pstate[t-1], written by the
// workgroup of t-1 after each of its iterations: (iterations done) << 24 | bit k = "my bits changed in iteration k" -- instead of
// a kernel boundary.  Workgroups are dispatched in index order, so the one waited for is always running or done.  The pair
// records and component constants of the timestep stay in registers across the iterations (a launch per pass reloaded them:
// three dependent round trips + the launch, 4.7 us a pass, 47-56 us for the ten to twelve passes of the bench slab).  An
// iteration whose predecessor did not change in the previous one does nothing but publish its word.  K <= 24.
// changed[] (per pass, read by the host / the mailbox) is kept as k_rs_pass keeps it.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_rs_pass_sys(ResolveDev r, int it0, int K, const uint32_t *__restrict__ pair_base, const uint32_t *__restrict__ pair_cnt,
                                                    uint32_t *__restrict__ pstate /* [T + 1], zeroed */, int prep_inline /* k_rs_prep's work for this timestep first */,
                                                    int do_unite /* k_rs_unite's work for the pairs of this timestep last; the grid then also covers t_hi + 1 .. T - 1 */)
{
    if (dev_tables_bad(r)) return;
    const int Kfull = K;
    const int t = (int)blockIdx.x + r.t_lo;
    const int lane = (int)threadIdx.x;
    // round trip 1
    const uint32_t cb = r.cprefix[t], ce = r.cprefix[t + 1];
    const uint32_t pb = pair_base[t], pn = pair_cnt[t];
    const uint32_t nu = dev_nungrouped(r);
    const uint32_t nct = ce - cb;
    const bool filtered = t <= r.t_hi;                     // (workgroups behind t_hi only unite their pairs)
    if (!filtered) K = 0;
    __shared__ long long Bl[2 * CTK_PASS_COMPS];
    const bool lds = nct <= CTK_PASS_COMPS;
    long long *B = lds ? Bl : (long long *)(r.B + 2 * (int64_t)cb);
    // round trip 2: first pair and first component of this lane (constant over the iterations)
    uint8_t *keep = r.keep0;
    const bool has_p = (uint32_t)lane < pn, has_c = (uint32_t)lane < nct;
    const uint32_t k0 = pb + lane, g0 = cb + lane;
    const uint32_t rd0 = has_p ? r.p_rd[k0] : 0u, rc0 = has_p ? r.p_rc[k0] : 0u;
    CtkPair p0;
    if (has_p) p0 = r.pairs[k0]; else { p0.lo = 0; p0.hi = 0; }
    const uint32_t mrep0 = has_c ? r.mrep[g0] : 0xffffffffu;
    double inv0 = 0.0, ff0 = 0.0;
    if (prep_inline && filtered) {
        for (uint32_t c = lane; c < nct; c += 64) {
            double a, b;
            dev_prep_comp(r, cb + c, &a, &b);
            if (c == (uint32_t)lane) { inv0 = a; ff0 = b; }
        }
        if (nct > 64) __syncthreads();                     // (components beyond the first 64 are re-read from memory below)
    } else if (has_c) { inv0 = r.inv[g0]; ff0 = r.ff[g0]; }
    uint8_t kold0 = has_c ? __hip_atomic_load(&keep[g0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
    const bool dyn_pred = t > r.t_lo;                      // the predecessor is filtered by this launch too
    uint32_t mybits = 0;
    for (int k = 0; k < K; k++) {
        const int it = it0 + k;
        bool evaluate = k == 0;
        if (k > 0 && dyn_pred) {
            uint32_t st;
            // relaxed polls (an acquire load invalidates the caches on EVERY poll: 2705 waves doing that made an iteration cost 95 us);
            // the bits are read with device-scope loads, which need no invalidation
            SpinGuard sg;
            bool gave_up = false;
            while (((st = __hip_atomic_load(&pstate[(size_t)(t - 1) * CTK_PSTATE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 24) < (uint32_t)k) {
                __builtin_amdgcn_s_sleep(2);
                if (spin_expired(sg, r.spin_limit)) { gave_up = true; break; }
            }
            if (gave_up) {
                if (lane == 0) { atomicOr(r.poison, CTK_POISON_SPIN); __hip_atomic_store(&pstate[(size_t)t * CTK_PSTATE_STRIDE], CTK_PSTATE_POISONED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                return;
            }
            evaluate = (st >> (k - 1)) & 1u;
        }
        bool wave_any = false;
        if (evaluate) {
            if (lds) for (uint32_t c = lane; c < 2 * nct; c += 64) Bl[c] = 0;
            const uint8_t kd0 = has_p ? __hip_atomic_load(&keep[rd0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
            __syncthreads();
            if (has_p && kd0) {
                const uint32_t c = rc0 - cb;
                atomicAdd((unsigned long long *)&B[2 * c], (unsigned long long)p0.lo);
                atomicAdd((unsigned long long *)&B[2 * c + 1], (unsigned long long)p0.hi);
            }
            for (uint32_t i = lane + 64; i < pn; i += 64) {            // timesteps with more than 64 pair records
                const uint32_t kk = pb + i;
                if (!__hip_atomic_load(&keep[r.p_rd[kk]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
                const CtkPair p = r.pairs[kk];
                const uint32_t c = r.p_rc[kk] - cb;
                atomicAdd((unsigned long long *)&B[2 * c], (unsigned long long)p.lo);
                atomicAdd((unsigned long long *)&B[2 * c + 1], (unsigned long long)p.hi);
            }
            if (nu) {                                                   // records that bypassed the hash table (rare)
                for (uint32_t i = lane; i < nu; i += 64) {
                    const CtkPair p = r.pairs[r.pair_cap - 1u - i];
                    if ((int)p.t != t) continue;
                    if (!__hip_atomic_load(&keep[r.p_rd[r.pair_cap - 1u - i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;
                    const uint32_t c = r.p_rc[r.pair_cap - 1u - i] - cb;
                    atomicAdd((unsigned long long *)&B[2 * c], (unsigned long long)p.lo);
                    atomicAdd((unsigned long long *)&B[2 * c + 1], (unsigned long long)p.hi);
                }
            }
            __syncthreads();
            bool any = false;
            for (uint32_t c = lane; c < nct; c += 64) {
                const uint32_t g = cb + c;
                const bool first = c == (uint32_t)lane;
                long long blo, bhi;
                if (lds) { blo = Bl[2 * c]; bhi = Bl[2 * c + 1]; }
                else {
                    blo = __hip_atomic_load(&B[2 * c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bhi = __hip_atomic_load(&B[2 * c + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&B[2 * c], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&B[2 * c + 1], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if ((first ? mrep0 : r.mrep[g]) != c) continue;         // representatives only
                bool inexact = r.inex[g] != 0;
                const double bwd = dev_limbs_to_double(blo, bhi, r.wshift, r.limb_bits, &inexact);
                double fb = (first ? inv0 : r.inv[g]) * bwd, ff = first ? ff0 : r.ff[g];
                const uint32_t os = r.ovr_slot ? r.ovr_slot[g] : 0u;
                if (os & 0x80000000u) {
                    const double *v = r.ovr_val + 3 * (size_t)(os & 0x3fffffffu);
                    const double inv = 1.0 / v[0];
                    fb = inv * v[2]; ff = inv * v[1];
                } else if (inexact) {
                    const double tol = CTK_AMBIG_ULPS * 2.220446049250313e-16 * fabs(r.overlap);
                    if ((ff != 0 && fabs(ff - r.overlap) <= tol) || (r.twosided && fb != 0 && fabs(fb - r.overlap) <= tol)) {
                        *r.ambig = 1u;
                        if (r.ovr_slot && os == 0u) {
                            const uint32_t idx = atomicAdd(r.amb_cnt, 1u);
                            if (idx < r.amb_cap) { r.amb_list[idx] = g; r.ovr_slot[g] = 0x40000000u | idx; }
                        }
                    }
                }
                bool kill = false;
                if (r.twosided) {
                    if (fb != 0 && ff != 0) { if (fb < r.overlap || ff < r.overlap) kill = true; }
                    if (fb != 0 && ff == 0) { if (fb < r.overlap) kill = true; }
                    if (fb == 0 && ff != 0) { if (ff < r.overlap) kill = true; }
                } else {
                    if (ff < r.overlap) kill = true;
                }
                const uint8_t kn = kill ? 0 : 1;
                // only this workgroup writes the bits of timestep t (device-scope load: what an earlier iteration of this launch stored)
                const uint8_t kold = first ? kold0 : __hip_atomic_load(&keep[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (kn != kold) { __hip_atomic_store(&keep[g], kn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); any = true; if (first) kold0 = kn; }
            }
            wave_any = __ballot(any) != 0ull;
            __syncthreads();                                            // (Bl is zeroed again by the next evaluation)
        }
        if (wave_any) mybits |= 1u << k;
        if (lane == 0) {
            if (wave_any) r.changed[it * CTK_CHG_SLOTS + (t & (CTK_CHG_SLOTS - 1))] = 1u;
            // the bits (device-scope stores: written through to the coherence point) before the word that announces them: all
            // stores of this wave acknowledged, then the word.  (A release fence would write back the whole L2 of this XCD.)
            __builtin_amdgcn_s_waitcnt(0);
            __hip_atomic_store(&pstate[(size_t)t * CTK_PSTATE_STRIDE], ((uint32_t)(k + 1) << 24) | mybits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!do_unite) return;
    // 3-D links of this timestep (contrack.py:748-750): its kept components with the kept components of t-1 they overlap.  The
    // predecessor's bits are final once it has published all of its iterations.
    if (t - 1 >= r.t_lo && t - 1 <= r.t_hi) {
        const uint32_t need = (uint32_t)Kfull;
        SpinGuard sg;
        while ((__hip_atomic_load(&pstate[(size_t)(t - 1) * CTK_PSTATE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 24) < need) {
            __builtin_amdgcn_s_sleep(2);
            if (spin_expired(sg, r.spin_limit)) { if (lane == 0) atomicOr(r.poison, CTK_POISON_SPIN); return; }
        }
    }
    auto link = [&](uint32_t slot) {
        if (!__hip_atomic_load(&keep[r.p_rc[slot]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
            !__hip_atomic_load(&keep[r.p_rd[slot]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        uint32_t a = r.p_gc[slot], b = r.p_gd[slot];
        for (;;) {
            a = gfind(r.parent, a);
            b = gfind(r.parent, b);
            if (a == b) break;
            if (a < b) { const uint32_t q = a; a = b; b = q; }
            const uint32_t old = atomicMin(&r.parent[a], b);
            if (old == a) break;
            a = old;
        }
    };
    for (uint32_t i = lane; i < pn; i += 64) link(pb + i);
    for (uint32_t i = lane; i < nu; i += 64) { if ((int)r.pairs[r.pair_cap - 1u - i].t == t) link(r.pair_cap - 1u - i); }
}

// ------------------------------------------------------------------------------------------------
// k_rs_pass_sys with PB_G consecutive timesteps per workgroup, one WAVE each ("blocked systolic").  The hand-shake between
// neighbouring timesteps -- a word that says how many iterations the predecessor has published and in which of them its bits
// changed -- costs ~2.5 us through memory (device-scope store, acknowledgement, device-scope poll from another CU) and is paid
// once per iteration: 12 iterations = 31 us of the bench pass, nearly all of it waiting.  Inside a workgroup the word and the
// predecessor's keep bits travel through LDS (~0.2 us); only the first wave of a workgroup listens to memory and only the last
// one publishes there.  A chain of K iterations crosses ceil(K / PB_G) workgroup boundaries.
// Same iteration, same fixed point (contrack.py:706-742), same outputs (keep bits in memory, changed[], pstate of the
// workgroups' last timesteps).  Waves synchronise only with themselves after the start-up barrier.
// ------------------------------------------------------------------------------------------------
#define PB_G 16
#define PB_COMPS 128         // components of a timestep whose backward sums and keep bits live in LDS (more: memory, as before)
// the lanes of ONE wave hand data to each other through LDS (executed in order for a wave) -- or, for timesteps whose sums live
// in the global scratch, through memory: then the operations have to be complete first
__device__ __forceinline__ void pb_wave_sync(bool through_memory)
{
    if (through_memory) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!through_memory) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifdef CTK_PHASE_TIMING
__device__ unsigned long long g_pb_t[4 * 1024];      // per workgroup: first entry | after the start-up barrier (last wave) | iterations done (last wave) | end (last wave)
#define PB_MARK_MIN(k) do { if (lane == 0 && blockIdx.x < 1024) atomicMin(&g_pb_t[4 * blockIdx.x + (k)], wall_clock64()); } while (0)
#define PB_MARK_MAX(k) do { __builtin_amdgcn_s_waitcnt(0); if (lane == 0 && blockIdx.x < 1024) atomicMax(&g_pb_t[4 * blockIdx.x + (k)], wall_clock64()); } while (0)
#else
#define PB_MARK_MIN(k) do { } while (0)
#define PB_MARK_MAX(k) do { } while (0)
#endif
__device__ __forceinline__ void rs_pass_blk_body(const ResolveDev &r, int it0, int K, const uint32_t *__restrict__ pair_base, const uint32_t *__restrict__ pair_cnt,
                                                 uint32_t *__restrict__ pstate /* [T + 1], zeroed */, int prep_inline, int do_unite)
{
    if (dev_tables_bad(r)) return;
    __shared__ long long Bl_all[PB_G][2 * PB_COMPS];
    __shared__ uint8_t kb_all[PB_G][PB_COMPS];                 // current keep bits of the wave's timestep (components < PB_COMPS)
    __shared__ uint32_t lstate[PB_G];                          // the wave's pstate word
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    PB_MARK_MIN(0);
    const int Kfull = K;
    const int t = (int)blockIdx.x * PB_G + w + r.t_lo;
    const bool live = t < (int)r.T;
    if (lane == 0) lstate[w] = 0u;
    uint8_t *keep = r.keep0;
    long long *Bl = Bl_all[w];
    uint8_t *kb = kb_all[w];
    // round trip 1
    const uint32_t cb = live ? r.cprefix[t] : 0u, ce = live ? r.cprefix[t + 1] : 0u, cbp = live ? r.cprefix[t - 1] : 0u;
    const uint32_t pb = live ? pair_base[t] : 0u, pn = live ? pair_cnt[t] : 0u;
    const uint32_t nu = dev_nungrouped(r);
    const uint32_t nct = ce - cb;
    const bool filtered = live && t <= r.t_hi;                 // (waves behind t_hi only unite their pairs)
    if (!filtered) K = 0;
    const bool lds = nct <= PB_COMPS;
    long long *B = lds ? Bl : (long long *)(r.B + 2 * (int64_t)cb);
    // the predecessor's bits and word: through LDS if it is a wave of this workgroup and its components fit there
    const bool pred_here = w > 0, pred_lds = pred_here && (cb - cbp) <= PB_COMPS;
    const uint8_t *kbp = kb_all[w > 0 ? w - 1 : 0];
    // round trip 2: first pair and first component of this lane (constant over the iterations)
    const bool has_p = (uint32_t)lane < pn, has_c = (uint32_t)lane < nct;
    const uint32_t k0 = pb + lane, g0 = cb + lane;
    const uint32_t rd0 = has_p ? r.p_rd[k0] : 0u, rc0 = has_p ? r.p_rc[k0] : 0u;
    CtkPair p0;
    if (has_p) p0 = r.pairs[k0]; else { p0.lo = 0; p0.hi = 0; }
    const uint32_t mrep0 = has_c ? r.mrep[g0] : 0xffffffffu;
    const uint32_t gc0 = (has_p && do_unite) ? r.p_gc[k0] : 0u, gd0 = (has_p && do_unite) ? r.p_gd[k0] : 0u;      // (for the unions at the end)
    double inv0 = 0.0, ff0 = 0.0;
    bool inex0 = false;
    if (prep_inline && filtered) {
        for (uint32_t c = lane; c < nct; c += 64) {
            double a, b;
            bool ix;
            dev_prep_comp(r, cb + c, &a, &b, &ix);
            if (c == (uint32_t)lane) { inv0 = a; ff0 = b; inex0 = ix; }
        }
        if (nct > 64) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }      // (re-read from memory below)
    } else if (has_c) { inv0 = r.inv[g0]; ff0 = r.ff[g0]; inex0 = r.inex[g0] != 0; }
    uint8_t kold0 = has_c ? __hip_atomic_load(&keep[g0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
    if (lds) for (uint32_t c = lane; c < nct; c += 64) kb[c] = (c == (uint32_t)lane) ? kold0 : __hip_atomic_load(&keep[cb + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                           // lstate = 0 and the initial bits of every wave are in LDS
    PB_MARK_MAX(1);
    if (!live) return;
    if (r.dbg_stall && blockIdx.x == 0) {                      // test hook: the head of the chain is late / never publishes
        if (r.dbg_stall == 2) return;
        const uint64_t t_in = wall_clock64();
        while (wall_clock64() - t_in < r.spin_limit / 4) __builtin_amdgcn_s_sleep(8);
    }
    auto pred_keep = [&](uint32_t rd) -> uint8_t {             // keep bit of a representative of timestep t-1
        if (pred_lds) return __hip_atomic_load(&kbp[rd - cbp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __hip_atomic_load(&keep[rd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto pred_state = [&]() -> uint32_t {
        if (pred_here) return __hip_atomic_load(&lstate[w - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __hip_atomic_load(&pstate[(size_t)(t - 1) * CTK_PSTATE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const bool publish_mem = w == PB_G - 1 || t == (int)r.T - 1 || !lds;      // somebody outside the workgroup (or without my LDS bits) listens
    const bool dyn_pred = t > r.t_lo;                          // the predecessor is filtered by this launch too
    uint32_t mybits = 0, st_seen = 0;
    for (int k = 0; k < K; k++) {
        const int it = it0 + k;
        bool evaluate = k == 0;
        if (k > 0 && dyn_pred) {
            // The predecessor's word holds the FINAL change bits of all the iterations it counts: a word read earlier that already
            // counts k is as good as a new one.  (Every read of a word of another workgroup is a trip to L2, ~1 us; wave 0 used to
            // make one per iteration even when its predecessor was iterations ahead, and the waves behind it wait for wave 0.)
            uint32_t st = st_seen;
            SpinGuard sg;
            bool gave_up = false;
            if ((st >> 24) < (uint32_t)k) {
                while (((st = pred_state()) >> 24) < (uint32_t)k) {
                    __builtin_amdgcn_s_sleep(1);
                    if (spin_expired(sg, r.spin_limit)) { gave_up = true; break; }
                }
                st_seen = st;
            }
            if (gave_up) {
                // nobody may wait for this wave either: the poisoned word says "everything published" to the waves behind
                if (lane == 0) {
                    atomicOr(r.poison, CTK_POISON_SPIN);
                    __hip_atomic_store(&pstate[(size_t)t * CTK_PSTATE_STRIDE], CTK_PSTATE_POISONED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&lstate[w], CTK_PSTATE_POISONED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                return;
            }
            evaluate = (st >> (k - 1)) & 1u;
        }
        bool wave_any = false;
        if (evaluate) {
            if (lds) for (uint32_t c = lane; c < 2 * nct; c += 64) Bl[c] = 0;
            const uint8_t kd0 = has_p ? pred_keep(rd0) : (uint8_t)0;
            pb_wave_sync(!lds);
            if (has_p && kd0) {
                const uint32_t c = rc0 - cb;
                atomicAdd((unsigned long long *)&B[2 * c], (unsigned long long)p0.lo);
                atomicAdd((unsigned long long *)&B[2 * c + 1], (unsigned long long)p0.hi);
            }
            for (uint32_t i = lane + 64; i < pn; i += 64) {            // timesteps with more than 64 pair records
                const uint32_t kk = pb + i;
                if (!pred_keep(r.p_rd[kk])) continue;
                const CtkPair p = r.pairs[kk];
                const uint32_t c = r.p_rc[kk] - cb;
                atomicAdd((unsigned long long *)&B[2 * c], (unsigned long long)p.lo);
                atomicAdd((unsigned long long *)&B[2 * c + 1], (unsigned long long)p.hi);
            }
            if (nu) {                                                   // records that bypassed the hash table (rare)
                for (uint32_t i = lane; i < nu; i += 64) {
                    const CtkPair p = r.pairs[r.pair_cap - 1u - i];
                    if ((int)p.t != t) continue;
                    if (!pred_keep(r.p_rd[r.pair_cap - 1u - i])) continue;
                    const uint32_t c = r.p_rc[r.pair_cap - 1u - i] - cb;
                    atomicAdd((unsigned long long *)&B[2 * c], (unsigned long long)p.lo);
                    atomicAdd((unsigned long long *)&B[2 * c + 1], (unsigned long long)p.hi);
                }
            }
            pb_wave_sync(!lds);
            bool any = false;
            for (uint32_t c = lane; c < nct; c += 64) {
                const uint32_t g = cb + c;
                const bool first = c == (uint32_t)lane;
                long long blo, bhi;
                if (lds) { blo = Bl[2 * c]; bhi = Bl[2 * c + 1]; }
                else {
                    blo = __hip_atomic_load(&B[2 * c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bhi = __hip_atomic_load(&B[2 * c + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&B[2 * c], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&B[2 * c + 1], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if ((first ? mrep0 : r.mrep[g]) != c) continue;         // representatives only
                bool inexact = first ? inex0 : (r.inex[g] != 0);
                const double bwd = dev_limbs_to_double(blo, bhi, r.wshift, r.limb_bits, &inexact);
                double fb = (first ? inv0 : r.inv[g]) * bwd, ff = first ? ff0 : r.ff[g];
                const uint32_t os = r.ovr_slot ? r.ovr_slot[g] : 0u;
                if (os & 0x80000000u) {
                    const double *v = r.ovr_val + 3 * (size_t)(os & 0x3fffffffu);
                    const double inv = 1.0 / v[0];
                    fb = inv * v[2]; ff = inv * v[1];
                } else if (inexact) {
                    const double tol = CTK_AMBIG_ULPS * 2.220446049250313e-16 * fabs(r.overlap);
                    if ((ff != 0 && fabs(ff - r.overlap) <= tol) || (r.twosided && fb != 0 && fabs(fb - r.overlap) <= tol)) {
                        *r.ambig = 1u;
                        if (r.ovr_slot && os == 0u) {
                            const uint32_t idx = atomicAdd(r.amb_cnt, 1u);
                            if (idx < r.amb_cap) { r.amb_list[idx] = g; r.ovr_slot[g] = 0x40000000u | idx; }
                        }
                    }
                }
                bool kill = false;
                if (r.twosided) {
                    if (fb != 0 && ff != 0) { if (fb < r.overlap || ff < r.overlap) kill = true; }
                    if (fb != 0 && ff == 0) { if (fb < r.overlap) kill = true; }
                    if (fb == 0 && ff != 0) { if (ff < r.overlap) kill = true; }
                } else {
                    if (ff < r.overlap) kill = true;
                }
                const uint8_t kn = kill ? 0 : 1;
                const uint8_t kold = first ? kold0 : (lds ? kb[c] : __hip_atomic_load(&keep[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (kn != kold) {
                    __hip_atomic_store(&keep[g], kn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lds) __hip_atomic_store(&kb[c], kn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    any = true;
                    if (first) kold0 = kn;
                }
            }
            wave_any = __ballot(any) != 0ull;
            pb_wave_sync(!lds);                                         // (Bl is zeroed again by the next evaluation)
        }
        if (wave_any) mybits |= 1u << k;
        const uint32_t word = ((uint32_t)(k + 1) << 24) | mybits;
        if (lane == 0) {
            if (wave_any) r.changed[it * CTK_CHG_SLOTS + (t & (CTK_CHG_SLOTS - 1))] = 1u;
            if (publish_mem) {
                // the bits (device-scope stores) before the word that announces them: all stores of this wave acknowledged first
                __builtin_amdgcn_s_waitcnt(0);
                __hip_atomic_store(&pstate[(size_t)t * CTK_PSTATE_STRIDE], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (LDS executes a wave's operations in order: the bits above are there before the word)
            __hip_atomic_store(&lstate[w], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    PB_MARK_MAX(2);
    if (!do_unite) return;
    // 3-D links of this timestep (contrack.py:748-750): its kept components with the kept components of t-1 they overlap.  The
    // predecessor's bits are final once it has published all of its iterations.
    if (t - 1 >= r.t_lo && t - 1 <= r.t_hi) {
        const uint32_t need = (uint32_t)Kfull;
        SpinGuard sg;
        while ((st_seen >> 24) < need && (pred_state() >> 24) < need) {
            __builtin_amdgcn_s_sleep(1);
            if (spin_expired(sg, r.spin_limit)) { if (lane == 0) atomicOr(r.poison, CTK_POISON_SPIN); return; }
        }
    }
    auto link = [&](uint32_t slot, bool first) {
        const uint32_t rc = first ? rc0 : r.p_rc[slot], rd = first ? rd0 : r.p_rd[slot];
        const uint8_t kc = lds ? kb[rc - cb] : __hip_atomic_load(&keep[rc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!kc || !pred_keep(rd)) return;
        uint32_t a = first ? gc0 : r.p_gc[slot], b = first ? gd0 : r.p_gd[slot];
        for (;;) {
            a = gfind(r.parent, a);
            b = gfind(r.parent, b);
            if (a == b) break;
            if (a < b) { const uint32_t q = a; a = b; b = q; }
            const uint32_t old = atomicMin(&r.parent[a], b);
            if (old == a) break;
            a = old;
        }
    };
    for (uint32_t i = lane; i < pn; i += 64) link(pb + i, i == (uint32_t)lane);
    for (uint32_t i = lane; i < nu; i += 64) { if ((int)r.pairs[r.pair_cap - 1u - i].t == t) link(r.pair_cap - 1u - i, false); }
    PB_MARK_MAX(3);
}
// Two builds of the same body.  With the 106 SGPRs it takes when left alone a SIMD holds six of its waves: ONE workgroup of sixteen per CU
// instead of two (CTK_SGPR_8WAVES, ctk_kernels.hip) -- that is the build for launches with more workgroups than CUs (438 000 x 192 x 288: 2.5 ->
// 1.9 ms).  A launch that leaves CUs empty anyway (2707 steps: 170 workgroups) is one chain per workgroup, and the ~110 values the limit moves
// into VGPR lanes sit in that chain: 29.8 -> 32.6 us; it keeps all its SGPRs.
__global__ __launch_bounds__(64 * PB_G) void k_rs_pass_blk(ResolveDev r, int it0, int K, const uint32_t *__restrict__ pair_base, const uint32_t *__restrict__ pair_cnt,
                                                           uint32_t *__restrict__ pstate, int prep_inline, int do_unite)
{
    rs_pass_blk_body(r, it0, K, pair_base, pair_cnt, pstate, prep_inline, do_unite);
}
__global__ __launch_bounds__(64 * PB_G) CTK_SGPR_8WAVES void k_rs_pass_blk_2pc(ResolveDev r, int it0, int K, const uint32_t *__restrict__ pair_base, const uint32_t *__restrict__ pair_cnt,
                                                                               uint32_t *__restrict__ pstate, int prep_inline, int do_unite)
{
    rs_pass_blk_body(r, it0, K, pair_base, pair_cnt, pstate, prep_inline, do_unite);
}

__global__ void k_rs_unite(ResolveDev r)
{
    const uint32_t np = dev_npairs(r), ng = dev_ngrouped(r);
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < np; k += gridDim.x * blockDim.x) {
        const uint32_t i = pair_slot(r, k, ng);
        if (!r.keep0[r.p_rc[i]] || !r.keep0[r.p_rd[i]]) continue;
        uint32_t a = r.p_gc[i], b = r.p_gd[i];
        for (;;) {
            a = gfind(r.parent, a);
            b = gfind(r.parent, b);
            if (a == b) break;
            if (a < b) { uint32_t s = a; a = b; b = s; }
            uint32_t old = atomicMin(&r.parent[a], b);
            if (old == a) break;
            a = old;
        }
    }
}

// the same on pair records in fixed per-timestep slots (k_overlap with pslot): thread = slot, + the ungrouped records
__global__ void k_rs_unite_slots(ResolveDev r, const uint32_t *__restrict__ pair_cnt, uint32_t pslot)
{
    if (dev_tables_bad(r)) return;
    const uint64_t nslots = (uint64_t)r.T * pslot;
    const uint32_t nu = dev_nungrouped(r);
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nslots + nu; k += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t i;
        if (k < nslots) {
            const uint32_t t = (uint32_t)(k / pslot), j = (uint32_t)(k - (uint64_t)t * pslot);
            if (j >= pair_cnt[t]) continue;
            i = (uint32_t)k;
        } else i = r.pair_cap - 1u - (uint32_t)(k - nslots);
        if (!r.keep0[r.p_rc[i]] || !r.keep0[r.p_rd[i]]) continue;
        uint32_t a = r.p_gc[i], b = r.p_gd[i];
        for (;;) {
            a = gfind(r.parent, a);
            b = gfind(r.parent, b);
            if (a == b) break;
            if (a < b) { uint32_t s = a; a = b; b = s; }
            uint32_t old = atomicMin(&r.parent[a], b);
            if (old == a) break;
            a = old;
        }
    }
}

// one component per thread; bsum[block] = surviving roots among the block's components (first half of the rank scan)
__global__ __launch_bounds__(256) void k_rs_roots(ResolveDev r, uint32_t *__restrict__ bsum)
{
    const uint32_t nc = dev_ncomps(r);
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    uint32_t isr = 0;
    if (g < nc) {
        const int32_t t = (int32_t)r.comp_t[g];            // -1: halo component
        const bool kept = r.keep0[r.cprefix[t] + r.mrep[g]] != 0;
        const uint32_t root = gfind(r.parent, g);
        r.lab[g] = kept ? (int32_t)root : -1;               // temporarily: root index, -1 = filtered out
        if (r.lab_root) r.lab_root[g] = kept ? (int32_t)root : -1;      // (fused path: kept while k_fz_rank_mark writes the labels into lab)
        isr = (kept && root == g && g >= (r.nh_ptr ? *r.nh_ptr : 0u)) ? 1u : 0u;       // (a halo root is numbered by an earlier shard)
        r.isroot[g] = isr;
        // labels <= components: candidate mark / dense-id slot g+1 are initialised here
        r.mark[g + 1] = 0;
        r.dmap[g + 1] = 0;
        r.op_first[g + 1] = -1;
        if (g == 0) { *r.dcount = 0; r.op_first[0] = -1; }
        if (r.cl_parent) {                                  // fused one-call path: label-indexed tables of the device seam driver
            const uint32_t l = g + 1u;
            r.cl_parent[l] = l; r.cl_tmin[l] = INT32_MAX; r.cl_tmax[l] = -1; r.cl_nops[l] = 0xffffffffu;
            int32_t *b = r.lbox + 6 * (int64_t)l;
            b[0] = INT32_MAX; b[1] = -1; b[2] = INT32_MAX; b[3] = -1; b[4] = INT32_MAX; b[5] = -1;
        }
        if (r.ext) {                                        // fused one-call path: what k_ops_ingest / k_fill_ext do elsewhere
            r.ext[g + 1] = INT32_MAX; r.ext[r.ext_off + g + 1] = INT32_MIN;
            if (g == 0) {
                r.ext[0] = INT32_MAX; r.ext[r.ext_off] = INT32_MIN;
                r.counters_w[CTK_CNT_WROTE_ZERO] = 0; r.counters_w[CTK_CNT_ALIVE] = 0; r.counters_w[CTK_CNT_TICKET] = 0;
            }
        }
    }
    if (r.ext) for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < CTK_ZF_SLOTS; i += (int64_t)gridDim.x * 256) ctk_zf_reset(r.counters_w, i);
    __shared__ uint32_t sm[8];
    uint32_t tot;
    const uint32_t ex = block_excl_scan(isr, sm, &tot);
    if (g < nc) r.rank[g] = ex;                             // roots in front of g inside its block (k_rs_rank overwrites it with the global rank)
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// rank[g] = surviving roots before g (second half: every block sums the block counts in front of it itself -- a few hundred
// values -- instead of waiting for a separate scan launch); rank[n] and *total = their number.  Same partition as k_rs_roots.
__global__ __launch_bounds__(256) void k_rs_rank(const uint32_t *__restrict__ isroot, const uint32_t *n_ptr, const uint32_t *__restrict__ bsum,
                                                 uint32_t *__restrict__ rank, uint32_t *__restrict__ total)
{
    __shared__ uint32_t sm[8];
    const uint32_t n = *n_ptr, b = blockIdx.x;
    const uint32_t nblk = (n + 255u) / 256u;
    if (b >= nblk && b != 0) return;
    const uint32_t upto = (b == 0) ? nblk : b;               // block 0 also delivers the grand total
    uint32_t s = 0;
    for (uint32_t j = threadIdx.x; j < upto; j += 256) s += bsum[j];
    uint32_t front;
    (void)block_excl_scan(s, sm, &front);
    __syncthreads();
    if (b == 0) {
        if (threadIdx.x == 0) { rank[n] = front; *total = front; }
        front = 0;
    }
    const uint32_t i = b * 256u + threadIdx.x;
    const uint32_t v = (i < n) ? isroot[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, sm, &tot);
    if (i < n) rank[i] = front + ex;
}

// k_rs_rank + k_rs_labels in one launch (while the block sums fit LDS): every workgroup builds the exclusive prefix of ALL block
// counts itself (a few hundred values), then label = 1 + prefix[block of the root] + roots in front of it inside that block.
// rank[] is not written (the time-shard path, which reads it, keeps the two kernels); *total = number of labels.
#define CTK_RL_BLOCKS 8192
__global__ __launch_bounds__(256) void k_rs_rank_labels(ResolveDev r, const uint32_t *__restrict__ bsum, uint32_t nsb, uint32_t *__restrict__ total)
{
    extern __shared__ uint32_t pre[];                       // [nsb]
    __shared__ uint32_t sm[8];
    const uint32_t nc = dev_ncomps(r);
    const uint32_t nblk = min((nc + 255u) / 256u, nsb);       // blocks that hold components (the grid is sized for the run count)
    if (blockIdx.x >= nblk && blockIdx.x != 0) return;
    uint32_t carry = 0;
    for (uint32_t j0 = 0; j0 < nblk; j0 += 256) {
        const uint32_t j = j0 + threadIdx.x;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(j < nblk ? bsum[j] : 0u, sm, &tot);
        if (j < nblk) pre[j] = carry + ex;
        carry += tot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *total = carry;
    __syncthreads();
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nc; g += gridDim.x * blockDim.x) {
        const int32_t root = r.lab[g];
        int32_t l = 0;
        if (root >= 0) {
            l = (int32_t)(pre[(uint32_t)root >> 8] + r.rank[root]) + 1;      // (r.rank: roots in front of it inside its block, k_rs_roots)
        }
        r.lab[g] = l;
    }
}

// fresh labels: 1 + rank of the root among surviving roots (raster order)
__global__ void k_rs_labels(ResolveDev r)
{
    const uint32_t nc = dev_ncomps(r);
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nc; g += gridDim.x * blockDim.x) {
        const int32_t root = r.lab[g];
        r.lab[g] = root < 0 ? 0 : (int32_t)r.rank[root] + 1;
    }
}

// Dense ids for the labels on surviving seam rows (claim order).  A label sits on many consecutive rows and
// timesteps: only the first row of a stretch tries (compare with the previous lane), the CAS on dmap[l] elects one
// winner per label, and the winners of a wave take their ids with ONE atomicAdd (same-address atomics serialise).
__device__ inline bool cand_try_claim(const ResolveDev &r, int32_t l)
{
    return r.dmap[l] == 0u && atomicCAS(&r.dmap[l], 0u, 0xffffffffu) == 0u;
}
__device__ inline void cand_publish(const ResolveDev &r, int32_t l, uint32_t id)
{
    r.dorig[id] = l;
    int32_t *d = r.dbox + 6 * (int64_t)id;                // box of the label (find_objects ONCE, contrack.py:753): filled
    d[0] = INT32_MAX; d[1] = -1; d[2] = INT32_MAX; d[3] = -1; d[4] = INT32_MAX; d[5] = -1;    // by k_rs_cand_groups
    r.dmap[l] = id + 1;                                   // read by later launches
}

// Only labels that occur in some seam row with two DIFFERENT labels can ever take part in a relabel
// operation (the first op needs such a row; every later `hi`/`lo` is a label of such a row or the `lo` of an
// earlier op).  Rows with equal labels that are not marked can therefore be dropped before the host driver.
__global__ __launch_bounds__(256) void k_rs_cand_mark(ResolveDev r, const CtkSeam *__restrict__ seams, const uint32_t *__restrict__ seam_cnt,
                                                      const uint32_t *__restrict__ seam_off, int ny, uint8_t *__restrict__ mark,
                                                      int2 *__restrict__ res /* [T][ny]: fresh labels of the row's two seam pixels, x < 0: filtered out */)
{
    const int t = (int)blockIdx.x;
    const uint32_t n = seam_cnt[t], cb = r.cprefix[t];
    const CtkSeam *scratch = seams + seam_off[t];
    const int lane = lane_id();
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {                     // whole waves enter together (shuffles / ballots below)
        const uint32_t i = i0 + threadIdx.x;
        int2 v = make_int2(-1, -1);
        if (i < n) {
            const CtkSeam q = scratch[i];
            if (r.keep0[cb + r.mrep[cb + q.cl]]) {
                v.x = r.lab[cb + q.cl]; v.y = r.lab[cb + q.cr];
                if (v.x != v.y) { mark[v.x] = 1; mark[v.y] = 1; }
            }
            res[(int64_t)t * ny + i] = v;
        }
        const int px = __shfl_up(v.x, 1), py = __shfl_up(v.y, 1);
        const bool try_x = v.x >= 0 && !(lane > 0 && px == v.x);
        const bool try_y = v.x >= 0 && v.y != v.x && !(lane > 0 && py == v.y);
        const bool won_x = try_x && cand_try_claim(r, v.x);
        const bool won_y = try_y && cand_try_claim(r, v.y);
        const uint64_t bx = __ballot(won_x), by = __ballot(won_y);
        const uint32_t nxw = (uint32_t)__popcll(bx), nyw = (uint32_t)__popcll(by);
        if (nxw + nyw) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(r.dcount, nxw + nyw);
            base = (uint32_t)__shfl((int)base, 0);
            const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            if (won_x) cand_publish(r, v.x, base + (uint32_t)__popcll(bx & below));
            if (won_y) cand_publish(r, v.y, base + nxw + (uint32_t)__popcll(by & below));
        }
    }
}

// Reductions onto HOT addresses (the box of a label that lives for hundreds of timesteps is updated by every one of them, from
// all eight XCDs; same-address device-scope operations serialise at ~20-80 ns each): the kernels below give a workgroup FZ_TW
// consecutive timesteps, one wave each, reduce into an LDS hash first and touch global memory once per label and workgroup.
// (Looking before the atomic does not help by itself: the L2s of the XCDs are not coherent with each other, so the look has to be
// a device-scope load -- just as hot as the atomic.  2000 x 721 x 1440: 0.4-0.7 ms per kernel before, see profiles/NOTES.md.)
#define FZ_TW 16             // timesteps (waves) of one workgroup
#define FZ_HS 512            // slots of the label-box hash
#define FZ_CS 256            // slots of the cluster-range hash
#define FZ_PROBES 8

__device__ __forceinline__ int fz_slot(int32_t *keys, int mask, int32_t key)
{
    uint32_t s = ((uint32_t)key * 2654435761u) >> 16;
    for (int p = 0; p < FZ_PROBES; p++, s++) {
        const int32_t old = atomicCAS(&keys[s & mask], 0, key);
        if (old == 0 || old == key) return (int)(s & mask);
    }
    return -1;
}
// global min / max after a device-scope look (six loads in flight together, then only the atomics that still change something)
__device__ __forceinline__ void fz_box_merge(int32_t *b, const int32_t *v)
{
    int32_t cur[6];
#pragma unroll
    for (int k = 0; k < 6; k++) cur[k] = __hip_atomic_load(&b[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k & 1) { if (v[k] > cur[k]) atomicMax(&b[k], v[k]); }
        else if (v[k] < cur[k]) atomicMin(&b[k], v[k]);
    }
}

// surviving seam rows of timestep t, run-length grouped: consecutive rows (y, y+1, ...) with the same pair of
// labels become ONE record {t, y0 | y1 << 16, label at x=0, label at x=nx-1}; (t, y) order.  One wave per
// timestep, one row per lane, group boundaries from ballots; the groups go into a row-indexed scratch, a scan +
// gather makes them dense.  The same wave adds the boxes of the timestep's components to the boxes of the labels
// that have a dense id (the only boxes anyone needs), through the workgroup's LDS hash.
__global__ __launch_bounds__(64 * FZ_TW) void k_rs_cand_groups(ResolveDev r, const CtkSeam *__restrict__ seams, const uint32_t *__restrict__ seam_cnt,
                                                               const uint32_t *__restrict__ seam_off, const int2 *__restrict__ res,
                                                               const uint8_t *__restrict__ mark, int ny, int64_t t_begin, uint32_t *__restrict__ cand_cnt,
                                                               CtkCand *__restrict__ scratch /* [T][ny] */)
{
    __shared__ int32_t hk[FZ_HS], hv[FZ_HS][6];
    for (int s = (int)threadIdx.x; s < FZ_HS; s += 64 * FZ_TW) {
        hk[s] = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) hv[s][k] = (k & 1) ? INT32_MIN : INT32_MAX;
    }
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * FZ_TW + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63);
    const bool live = t < r.T;
    const uint32_t cb = live ? r.cprefix[t] : 0u, nct = live ? r.cprefix[t + 1] - cb : 0u;
    const int32_t tt = (int32_t)(t_begin + t);
    for (uint32_t c = lane; c < nct; c += 64) {
        const uint32_t g = cb + c;
        const int32_t l = r.lab[g];
        if (l <= 0) continue;
        const uint32_t d = r.dmap[l];
        if (d == 0) continue;
        const uint16_t *q = r.box + 4 * (int64_t)g;
        const int32_t v[6] = {tt, tt, (int32_t)q[0], (int32_t)q[1], (int32_t)q[2], (int32_t)q[3]};
        const int s = fz_slot(hk, FZ_HS - 1, (int32_t)d);
        if (s < 0) { fz_box_merge(r.dbox + 6 * (int64_t)(d - 1), v); continue; }        // (a crowded hash: straight to memory)
#pragma unroll
        for (int k = 0; k < 6; k++) { if (k & 1) atomicMax(&hv[s][k], v[k]); else atomicMin(&hv[s][k], v[k]); }
    }
    const uint32_t n = live ? seam_cnt[t] : 0u;
    const CtkSeam *sc = seams + (live ? seam_off[t] : 0u);
    const int2 *rs = res + t * ny;
    CtkCand *dst = scratch + t * ny;                      // at most one group per seam row
    // 64 rows per step, one per lane.  A row STARTS a group unless the previous row is valid, carries the same pair of
    // labels and is the row right above; the group's record is written by its first row (y1 = y0) and its last row
    // overwrites the y1 half.  State carried between steps: the last row of the previous step.
    uint32_t ng = 0;
    bool c_valid = false;
    int32_t c_ll = 0, c_lr = 0, c_y = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        int2 v = make_int2(-1, -1);
        int32_t y = 0;
        if (i < n) {
            v = rs[i];
            if (v.x >= 0 && v.x == v.y && !mark[v.x]) v.x = -1;      // can never take part in an op
            y = (int32_t)sc[i].y;
        }
        const bool valid = v.x >= 0;
        int32_t pll = __shfl_up(v.x, 1), plr = __shfl_up(v.y, 1), py = __shfl_up(y, 1);
        bool pvalid = pll >= 0;
        if (lane == 0) { pll = c_ll; plr = c_lr; py = c_y; pvalid = c_valid; }
        const bool start = valid && !(pvalid && pll == v.x && plr == v.y && y == py + 1);
        const uint64_t S = __ballot(start), V = __ballot(valid);
        const uint64_t upto = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        const uint32_t idx = ng + (uint32_t)__popcll(S & upto) - 1u;          // record of this row's group
        if (start) { CtkCand g; g.t = tt; g.yy = y | (y << 16); g.ll = v.x; g.lr = v.y; dst[idx] = g; }
        // the previous step's last row ended its group if this step's first row does not continue it
        if (lane == 0 && c_valid && (!valid || start)) reinterpret_cast<uint16_t *>(&dst[ng - 1u].yy)[1] = (uint16_t)c_y;
        if (valid && lane < 63) {
            const bool next_valid = (V >> (lane + 1)) & 1ull, next_start = (S >> (lane + 1)) & 1ull;
            if ((!next_valid || next_start) && !start) reinterpret_cast<uint16_t *>(&dst[idx].yy)[1] = (uint16_t)y;
        }
        ng += (uint32_t)__popcll(S);
        c_valid = (V >> 63) & 1ull;
        c_ll = __shfl(v.x, 63); c_lr = __shfl(v.y, 63); c_y = __shfl(y, 63);
    }
    if (lane == 0 && live) {
        if (c_valid) reinterpret_cast<uint16_t *>(&dst[ng - 1u].yy)[1] = (uint16_t)c_y;
        cand_cnt[t] = ng;
    }
    __syncthreads();
    for (int s = (int)threadIdx.x; s < FZ_HS; s += 64 * FZ_TW) if (hk[s]) fz_box_merge(r.dbox + 6 * (int64_t)(hk[s] - 1), hv[s]);
}

// What the host needs after the resolver kernels, written by the device straight into pinned host memory (no copy
// commands, one stream synchronisation): scalars, the candidate records, the dense label tables.
struct CandMail {
    uint32_t *scal;                 // [CTK_MAIL_SCALARS]
    CtkCand *cand;                  // [cap_c]
    int32_t *dorig, *dbox;          // [cap_d], [cap_d][6]
    uint32_t cap_c, cap_d;
};
#define CTK_MAIL_SCALARS 64
#define CTK_MAIL_NC 8
#define CTK_MAIL_NCAND 9
#define CTK_MAIL_ND 10
#define CTK_MAIL_NLAB 11
#define CTK_MAIL_CHANGED 12         // .. + passes of the round (<= 32)
#define CTK_MAIL_AMBIG 50

// dense (t, y)-ordered group records from the row-indexed scratch, labels replaced by their dense ids; mailbox
__global__ __launch_bounds__(64) void k_compact_cands(ResolveDev r, const CtkCand *__restrict__ scratch, const uint32_t *__restrict__ cand_cnt,
                                                      const uint32_t *__restrict__ cand_off, int ny, CtkCand *__restrict__ out,
                                                      const uint32_t *__restrict__ nlab_ptr, int it0, int round, CandMail m)
{
    const int64_t t = blockIdx.x;
    if (t < r.T) {
        const uint32_t n = cand_cnt[t], o = cand_off[t];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            CtkCand c = scratch[t * ny + i];
            c.ll = (int32_t)r.dmap[c.ll] - 1;
            c.lr = (int32_t)r.dmap[c.lr] - 1;
            out[o + i] = c;
            if (o + i < m.cap_c) m.cand[o + i] = c;
        }
    }
    const uint32_t nd = min(*r.dcount, m.cap_d);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += gridDim.x * blockDim.x) {
        m.dorig[i] = r.dorig[i];
#pragma unroll
        for (int k = 0; k < 6; k++) m.dbox[6 * (int64_t)i + k] = r.dbox[6 * (int64_t)i + k];
    }
    if (blockIdx.x == 0 && threadIdx.x < CTK_CNT_N) m.scal[threadIdx.x] = r.counters[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        m.scal[CTK_MAIL_NC] = r.cprefix[r.T];
        m.scal[CTK_MAIL_NCAND] = r.T > 0 ? cand_off[r.T] : 0u;
        m.scal[CTK_MAIL_ND] = *r.dcount;
        m.scal[CTK_MAIL_NLAB] = *nlab_ptr;
        m.scal[CTK_MAIL_AMBIG] = *r.ambig;
    }
    if (blockIdx.x == 0) {                                   // (the workgroup is one wave)
        for (int k = 0; k < round; k++) {
            const bool any = __ballot(r.changed[(it0 + k) * CTK_CHG_SLOTS + threadIdx.x] != 0u) != 0ull;
            if (threadIdx.x == 0) m.scal[CTK_MAIL_CHANGED + k] = any ? 1u : 0u;
        }
    }
}

__global__ void k_iota_mul(uint32_t *p, uint32_t n, uint32_t mul)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i * mul;
}
