// ctk_seam_dev.hip -- the bbox-confined seam merge (contrack/contrack.py:753-763) driven ON THE DEVICE (included by ctk_api.hip).
//
// The merge is sequential in (t, y) and order-dependent, but only INSIDE a cluster of labels: an operation connects the two labels
// of one seam row, so labels that never share a seam row (directly or through a chain of rows) never influence each other.  The
// candidate records of a slab fall into ~10^3 clusters of a few dozen records each (2707 x 181 x 360: 1016 clusters, at most 26
// records and 4 operations in one).  One wave per cluster runs ctk_seam.h's SeamDriver on its records with the cluster's labels
// and operations in LDS; the clusters run side by side.  No host hand-off: the pass has no synchronisation between the run scan
// of stage 1 and its end.
//
//   k_rs_cand_groups   (ctk_resolve_dev.hip) unites the two dense label ids of every group record      -> cl_parent
//   k_seam_clusters    cluster root of every record, record labels -> dense ids, time range of every cluster
//   k_seam_driver      one wave per cluster: ops[], op_next[], op_first[label] -- what k_ops_ingest produced from the host's list
//
// Capacities are per cluster (labels / operations held in LDS) and global (operation slots); a cluster or a slab that exceeds
// them raises a bit in *poison and the host repeats the resolution on its own path (SeamDriver in ctk_seam.h).
#pragma once

#define SD_LAB 64            // labels of one cluster (one per lane: the slot search is a ballot)
#define SD_OPS_OWN 8         // op slots every candidate label id owns (clusters with more: shared tail)
#define CTK_POISON_OPCAP   1u    // more operations than the op arrays hold
#define CTK_POISON_CLUSTER 2u    // a cluster with more labels / operations than the LDS tables hold
#define CTK_POISON_TABLES  4u    // the co-occurrence table overflowed (k_overlap)
#define CTK_POISON_DENSE   8u    // more candidate labels than the dense tables hold

struct SeamDev {
    uint32_t *cl_parent;           // [dense] union-find over dense label ids (init: cand_publish)
    int32_t *cl_tmin, *cl_tmax;    // [dense] at cluster roots: local timesteps that hold records of the cluster
    uint32_t *rec_root;            // [T][ny] cluster root of every group record
    CtkCand *recs;                 // [T][ny] group records (k_rs_cand_groups' scratch); labels become dense ids in k_seam_clusters
    const uint32_t *rec_cnt;       // [T]
    const uint32_t *dcount;        // number of dense ids
    const int32_t *dorig, *dbox;   // [dense], [dense][6]
    const uint32_t *dmap;          // [labels + 1]
    CtkOp *ops;                    // [op_cap] out
    int32_t *op_next;              // [op_cap]
    int32_t *op_first;             // [labels + 1] (reset to -1 by k_rs_roots)
    uint32_t *op_count;            // operations placed in the shared tail of the op arrays
    uint32_t *cl_nops;             // [dense] at cluster roots: operations of the cluster (the count kernel adds them up)
    uint32_t op_cap;               // slots of ops / op_next: own_ids * SD_OPS_OWN + shared tail
    uint32_t own_ids;              // cluster roots below this id own SD_OPS_OWN slots each
    uint32_t dense_cap;
    uint32_t *poison;
    int ny, nx;
    int64_t T;
    int dbg;                       // experiments (CTK_SD_DBG): stop after a stage
    int lab_cap, ops_cap;          // labels / operations of one cluster (<= 64; a test hook lowers them)
};

__global__ __launch_bounds__(64) void k_seam_clusters(SeamDev a)
{
    if (*a.poison) return;
    const int t = (int)blockIdx.x, lane = (int)threadIdx.x;
    const uint32_t n = a.rec_cnt[t];
    for (uint32_t i = lane; i < n; i += 64) {
        CtkCand c = a.recs[(int64_t)t * a.ny + i];
        const uint32_t dl = a.dmap[c.ll] - 1u, dr = a.dmap[c.lr] - 1u;
        const uint32_t root = gfind(a.cl_parent, dl);
        c.ll = (int32_t)dl; c.lr = (int32_t)dr;
        a.recs[(int64_t)t * a.ny + i] = c;
        a.rec_root[(int64_t)t * a.ny + i] = root;
        // look first: same-address atomics serialise.  Device-scope loads: the L2s of the eight XCDs are not coherent with each
        // other, a plain load would keep returning the bound this XCD saw first.
        if (t < __hip_atomic_load(&a.cl_tmin[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&a.cl_tmin[root], t);
        if (t > __hip_atomic_load(&a.cl_tmax[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&a.cl_tmax[root], t);
    }
}

// wave-wide min / max of one int per lane (every lane gets the result): DPP inside the rows of 16 lanes (quad swaps, half-row and
// row mirrors: full-rate VALU), the four row results through v_readlane.  (__shfl_xor = ds_bpermute costs an LDS round trip per
// step: twelve of them per fold step were most of a large cluster's time.)
template <int CTRL>
__device__ __forceinline__ int32_t sd_dpp(int32_t v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int32_t sd_wave_min(int32_t v)
{
    v = min(v, sd_dpp<0xB1>(v)); v = min(v, sd_dpp<0x4E>(v)); v = min(v, sd_dpp<0x141>(v)); v = min(v, sd_dpp<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int32_t sd_wave_max(int32_t v)
{
    v = max(v, sd_dpp<0xB1>(v)); v = max(v, sd_dpp<0x4E>(v)); v = max(v, sd_dpp<0x141>(v)); v = max(v, sd_dpp<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
#define SD_RL(v, i) __builtin_amdgcn_readlane((int)(v), (int)(i))
#define SD_U(x) __builtin_amdgcn_readfirstlane((int)(x))          // wave-uniform by construction: keep it in a scalar register

#define SD_BATCH 512         // records of one 64-timestep batch staged in LDS (more: the cluster goes to the host path)

// One wave (= one workgroup) per cluster.  The cluster's state lives in REGISTERS, one item per lane: lane k holds operation k
// (box, `hi` / `lo` as label slots) and label slot k (dense id, fresh label, box); the sequential driver reads them with
// v_readlane and asks its questions with ballots -- "is slot s `hi` of some operation", "the last operation with hi == s", the fold of
// the operations over a seam pixel (first lane >= s whose box holds the pixel) -- so that a row step costs a few dozen scalar
// instructions instead of a chain of LDS round trips.  The records are gathered 64 timesteps at a time (a lane per timestep: all
// loads of a batch in flight together) and worked off in (t, y) order.  At most 64 labels and 64 operations per cluster.
__global__ __launch_bounds__(64) void k_seam_driver(SeamDev a, int64_t t_begin)
{
    if (*a.poison) return;
    __shared__ CtkCand brec[SD_BATCH];
    const int lane = (int)threadIdx.x;
    const uint32_t nd = min(*a.dcount, a.dense_cap);
    for (uint32_t R = blockIdx.x; R < nd; R += gridDim.x) {
        const uint32_t par = a.cl_parent[R];
        const int32_t tmin = a.cl_tmin[R], tmax = a.cl_tmax[R];            // (one round trip for the three)
        if (par != R || tmax < tmin) continue;                             // not a cluster root / a boundary label without records
        if (a.dbg == 1) continue;
        int32_t o_hi = -1, o_lo = -1, o_t0 = 0, o_t1 = -1, o_y0 = 0, o_y1 = -1, o_x0 = 0, o_x1 = -1;      // operation `lane`
        int32_t l_id = -1, l_orig = 0, l_b0 = 0, l_b1 = 0, l_b2 = 0, l_b3 = 0, l_b4 = 0, l_b5 = 0;        // label slot `lane`
        int nl = 0, nops = 0;
        bool bad = false;
        uint32_t dbg_steps = 0, dbg_fold = 0; unsigned long long dbg_tp = 0, dbg_t0 = wall_clock64(), dbg_tf = 0;
        // Rows ya..yb of (global) timestep tg carry the pair of label slots (sl, sr): SeamDriver::run's inner loop (ctk_seam.h).
        // The folds of the two seam pixels are independent chains and advance together in one loop (a lone dependent chain issues
        // an instruction every ~8 cycles on this machine: a row step of ~150 of them was 0.7 us).  Only the upper end of the row
        // interval over which both answers provably stay the same is needed (the lower end served the host version's memo).
        auto record = [&](int32_t tg, int32_t ya, int32_t yb, int sl, int sr) {
            nops = SD_U(nops);
            bool tl = __ballot(lane < nops && o_hi == sl) != 0ull;                         // is `hi` of some op
            bool tr = (sr == sl) ? tl : (__ballot(lane < nops && o_hi == sr) != 0ull);
            sl = SD_U(sl); sr = SD_U(sr); ya = SD_U(ya); yb = SD_U(yb); tg = SD_U(tg);
            for (int32_t y = ya; y <= yb;) {
                dbg_steps++;
                y = SD_U(y);
                if (sl == sr && !tl) break;                                  // same label, never relabelled: nothing can differ
                int32_t p0 = sl, p1 = sr, yhi = INT32_MAX;
                const unsigned long long tf0 = a.dbg >= 10 ? clock64() : 0;
                if (tl || tr) {
                    const bool live = lane < nops && tg >= o_t0 && tg <= o_t1;
                    const bool y_in = y >= o_y0 && y <= o_y1, below = y < o_y0;
                    const bool xa = live && o_x0 <= 0, xb = live && o_x1 >= a.nx - 1;      // the box holds x = 0 / x = nx - 1
                    int32_t sa = 0, sb = 0;
                    bool ga = tl, gb = tr;                                   // chain still going
                    while (ga || gb) {
                        dbg_fold++;
                        const bool ca = ga && xa && o_hi == p0 && lane >= sa, cb = gb && xb && o_hi == p1 && lane >= sb;
                        const uint64_t ia = __ballot(ca && y_in), ib = __ballot(cb && y_in);
                        const int ka = ia ? (int)__builtin_ctzll(ia) : 64, kb = ib ? (int)__builtin_ctzll(ib) : 64;
                        // ops examined before the hit (or all, without one) and found outside because of y only: they stay outside
                        // while y stays on the same side of their rows
                        const bool va = ca && lane < ka && below, vb = cb && lane < kb && below;
                        if (__ballot(va || vb)) yhi = min(yhi, sd_wave_min((va || vb) ? o_y0 - 1 : INT32_MAX));
                        if (ka < 64) { yhi = min(yhi, SD_RL(o_y1, ka)); p0 = SD_RL(o_lo, ka); sa = ka + 1; } else ga = false;
                        if (kb < 64) { yhi = min(yhi, SD_RL(o_y1, kb)); p1 = SD_RL(o_lo, kb); sb = kb + 1; } else gb = false;
                        yhi = SD_U(yhi); p0 = SD_U(p0); p1 = SD_U(p1); sa = SD_U(sa); sb = SD_U(sb);
                    }
                }
                if (a.dbg >= 10) dbg_tf += clock64() - tf0;
                const int32_t same_until = min(yb, yhi);
                if (p0 == p1) { y = same_until + 1; continue; }              // nothing happens on these rows
                const bool p0_hi = SD_RL(l_orig, p0) > SD_RL(l_orig, p1);    // the larger FRESH label becomes the smaller (:759/763)
                const int32_t hi = p0_hi ? p0 : p1, lo = p0_hi ? p1 : p0;
                const uint64_t hb = __ballot(lane < nops && o_hi == hi), fb = __ballot(lane < nops && o_lo == hi);
                const int lh = hb ? 63 - (int)__builtin_clzll(hb) : -1;      // last op with this `hi`; last op that moved pixels INTO it
                const int inflow = fb ? 63 - (int)__builtin_clzll(fb) : -1;
                if (lh >= 0 && inflow < lh) { y = same_until + 1; continue; }      // nothing to move until an op is recorded
                if (nops >= a.ops_cap) { bad = true; return; }
                const int32_t b0 = SD_RL(l_b0, hi), b1 = SD_RL(l_b1, hi), b2 = SD_RL(l_b2, hi), b3 = SD_RL(l_b3, hi), b4 = SD_RL(l_b4, hi), b5 = SD_RL(l_b5, hi);
                if (lane == nops) { o_hi = hi; o_lo = lo; o_t0 = b0; o_t1 = b1; o_y0 = b2; o_y1 = b3; o_x0 = b4; o_x1 = b5; }
                nops++;
                tl = tl || hi == sl; tr = tr || hi == sr;
                y++;                                                         // the next row sees the new op
            }
        };
        for (int32_t t0 = tmin; t0 <= tmax && !bad; t0 += 64) {
            // gather: the records of the 64 timesteps, flattened in (t, i) order, 64 at a time -- every load of a step is independent
            // of the others (a loop over i per timestep made each of its loads wait for the previous one: ~2 us apiece from L2 / HBM,
            // 50 of the kernel's 55 us).  Matching records go to LDS in order.
            const int32_t t = t0 + lane;
            const uint32_t cnt = (t <= tmax) ? a.rec_cnt[t] : 0u;
            const uint32_t inc = wave_incl_scan_u32(cnt), exc = inc - cnt;
            const uint32_t win = (uint32_t)__shfl((int)inc, 63);            // records of the window (all clusters)
            __syncthreads();                                                 // (the previous batch has been read)
            uint32_t total = 0;
            for (uint32_t j0 = 0; j0 < win; j0 += 64) {
                const uint32_t j = j0 + lane;
                int lo = 0, hi = 63;                                         // lane q whose timestep holds flattened record j
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const int mid = (lo + hi) >> 1;
                    if (j >= (uint32_t)__shfl((int)inc, mid)) lo = mid + 1; else hi = mid;
                }
                const int64_t at = (int64_t)(t0 + lo) * a.ny + (int64_t)(j - (uint32_t)__shfl((int)exc, lo));
                uint32_t root = 0xffffffffu;
                CtkCand c;
                c.t = 0; c.yy = 0; c.ll = -1; c.lr = -1;
                if (j < win) { root = a.rec_root[at]; c = a.recs[at]; }
                const bool match = j < win && root == R;
                const uint64_t mb = __ballot(match);
                const uint32_t pos = total + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull));
                if (match && pos < SD_BATCH) { c.t = (int32_t)(t_begin + t0 + lo); brec[pos] = c; }
                total += (uint32_t)__popcll(mb);
            }
            if (total > SD_BATCH) { bad = true; break; }
            __syncthreads();
            if (a.dbg == 2) continue;
            for (uint32_t c0 = 0; c0 < total && !bad; c0 += 64) {
                const int nchunk = (int)min(64u, total - c0);
                CtkCand c;
                c.t = 0; c.yy = 0; c.ll = -1; c.lr = -1;
                if (lane < nchunk) c = brec[c0 + lane];
                // label slots for the labels this chunk brings: found in parallel, their tables loaded in one round trip
                bool need_l = lane < nchunk, need_r = lane < nchunk && c.lr != c.ll;
                for (int sidx = 0; sidx < nl; sidx++) { const int32_t id = SD_RL(l_id, sidx); need_l = need_l && c.ll != id; need_r = need_r && c.lr != id; }
                const int nl0 = nl;
                for (;;) {
                    const uint64_t bl = __ballot(need_l), br = __ballot(need_r);
                    if (!(bl | br)) break;
                    if (nl >= a.lab_cap) { bad = true; break; }
                    const int32_t d = bl ? SD_RL(c.ll, __builtin_ctzll(bl)) : SD_RL(c.lr, __builtin_ctzll(br));
                    if (lane == nl) l_id = d;
                    nl++;
                    need_l = need_l && c.ll != d; need_r = need_r && c.lr != d;
                }
                if (bad) break;
                if (lane >= nl0 && lane < nl) {
                    l_orig = a.dorig[l_id];
                    const int32_t *bx = a.dbox + 6 * (int64_t)l_id;
                    l_b0 = bx[0]; l_b1 = bx[1]; l_b2 = bx[2]; l_b3 = bx[3]; l_b4 = bx[4]; l_b5 = bx[5];
                }
                if (a.dbg == 3) continue;
                const unsigned long long tp0 = wall_clock64();
                for (int k = 0; k < nchunk && !bad; k++) {
                    const int32_t ll = SD_RL(c.ll, k), lr = SD_RL(c.lr, k), yy = SD_RL(c.yy, k), tg = SD_RL(c.t, k);
                    const int sl = (int)__builtin_ctzll(__ballot(lane < nl && l_id == ll));
                    const int sr = (lr == ll) ? sl : (int)__builtin_ctzll(__ballot(lane < nl && l_id == lr));
                    record(tg, yy & 0xffff, (int32_t)((uint32_t)yy >> 16), sl, sr);
                }
                dbg_tp += wall_clock64() - tp0;
            }
        }
        if (a.dbg >= 10 && lane == 0) { atomicMax(&a.poison[2], dbg_steps); atomicMax(&a.poison[3], dbg_fold); atomicMax(&a.poison[4], (uint32_t)dbg_tp); atomicMax(&a.poison[5], (uint32_t)(wall_clock64() - dbg_t0)); atomicMax(&a.poison[6], (uint32_t)dbg_tf); }
        if (bad) { if (lane == 0) atomicOr(a.poison, CTK_POISON_CLUSTER); continue; }
        if (nops == 0) continue;
        // The cluster's operations take a contiguous range of the op arrays (chain order = index order inside the cluster): the
        // first SD_OPS_OWN in the slots that belong to its root id, larger clusters a range of the shared tail.  (One counter for
        // all clusters: a thousand same-address atomics from eight XCDs, ~80 ns each -- 47 of this kernel's 54 us.)
        a.cl_nops[R] = (uint32_t)nops;
        uint32_t base = R * SD_OPS_OWN;
        if (nops > SD_OPS_OWN || R >= a.own_ids) {
            if (lane == 0) base = a.own_ids * SD_OPS_OWN + atomicAdd(a.op_count, (uint32_t)nops);
            base = (uint32_t)__shfl((int)base, 0);
        }
        if ((uint64_t)base + (uint64_t)nops > (uint64_t)a.op_cap) { if (lane == 0) atomicOr(a.poison, CTK_POISON_OPCAP); continue; }
        // chains: next op with the same `hi` (per op lane), first op with hi == slot (per label lane)
        int32_t nxt = -1, fst = -1;
        for (int j = nops - 1; j >= 0; j--) {
            const int32_t hj = SD_RL(o_hi, j);
            if (j > lane && hj == o_hi) nxt = j;
            if (hj == lane) fst = j;
        }
        const int32_t hi_orig = __shfl(l_orig, o_hi >= 0 ? o_hi : 0), lo_orig = __shfl(l_orig, o_lo >= 0 ? o_lo : 0);
        if (lane < nops) {
            CtkOp o;
            o.hi = hi_orig; o.lo = lo_orig; o.t0 = o_t0; o.t1 = o_t1; o.y0 = o_y0; o.y1 = o_y1; o.x0 = o_x0; o.x1 = o_x1;
            a.ops[base + lane] = o;
            a.op_next[base + lane] = nxt < 0 ? -1 : (int32_t)(base + (uint32_t)nxt);
        }
        if (lane < nl && fst >= 0) a.op_first[l_orig] = (int32_t)(base + (uint32_t)fst);
    }
}
