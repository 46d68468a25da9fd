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
#define SD_OPS_OWN 8         // op slots every label id owns (clusters with more: shared tail)
#define CTK_POISON_OPCAP   1u    // more operations than the op arrays hold
#define CTK_POISON_CLUSTER 2u    // a cluster with more labels / operations than the driver's per-cluster tables hold

// Everything is indexed by the FRESH 3-D label itself (ids <= components): no dense renumbering of the candidate labels, hence
// no claim counter -- a counter that every claiming wave of eight XCDs adds to costs ~80 ns per add.
struct SeamDev {
    const ResolveDev *dummy;       // (unused; keeps the layout explicit)
    uint32_t *cl_parent;           // [labels + 1] union-find over labels that share seam rows (init: k_rs_roots; unions: k_fz_mark)
    int32_t *cl_tmin, *cl_tmax;    // [labels + 1] at cluster roots: local timesteps that hold records of the cluster
    uint32_t *cl_nops;             // [labels + 1] claim word of a cluster root (0xffffffff: nobody drives the cluster yet)
    uint32_t *t_nops;              // [T] operations of the clusters whose first record sits in timestep t (the count kernel adds them up)
    int32_t *lbox;                 // [labels + 1][6] box of every marked label on the fresh labelling (find_objects ONCE, contrack.py:753)
    uint8_t *mark;                 // [labels + 1] label occurs in a seam row with two different labels
    uint32_t *rec_root;            // [T][ny] cluster root of every group record
    CtkCand *recs;                 // [T][ny] group records, row-indexed scratch (labels = fresh labels)
    uint32_t *rec_cnt;             // [T]
    CtkOp *ops;                    // [op_cap] out
    int32_t *op_next;              // [op_cap]
    int32_t *op_first;             // [labels + 1] (reset to -1 by k_rs_roots)
    uint32_t *op_count;            // operations placed in the shared tail of the op arrays
    uint32_t op_cap;               // slots of ops / op_next: own_ids * SD_OPS_OWN + shared tail
    uint32_t own_ids;              // cluster roots below this label own SD_OPS_OWN slots each
    uint32_t *poison;
    int ny, nx;
    int64_t T;
    int dbg;                       // experiments (CTK_SD_DBG): stop after a stage
    int lab_cap, ops_cap;          // labels / operations of one cluster (<= 64; a test hook lowers them)
    // time-shard path (ctk_sharded.hip): labels are GLOBAL ids; the clusters driven here hold only ids this shard numbered,
    // (own_base, own_base + own_ids) -- their op slots are indexed by id - own_base -- and a cluster that holds an id reaching a
    // shard boundary is "shared": all-gathered and driven identically on every rank (host), skipped here.  One-call path: 0 / nullptr.
    uint32_t own_base;
    const uint8_t *cl_shared;      // [labels + 1] at cluster roots, or nullptr
};

// fresh labels of the two seam pixels of every seam row; labels that meet a different label on a row are marked (only they can
// ever take part in an operation) and united into one cluster
__global__ __launch_bounds__(64) void k_fz_mark(ResolveDev r, SeamDev a, const CtkSeam *__restrict__ seams, const uint32_t *__restrict__ seam_cnt,
                                                const uint32_t *__restrict__ seam_off, int2 *__restrict__ res)
{
    if (dev_tables_bad(r)) return;
    const int t = (int)blockIdx.x, lane = (int)threadIdx.x;
    const uint32_t n = seam_cnt[t], cb = r.cprefix[t];
    const CtkSeam *scratch = seams + seam_off[t];
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        int2 v = make_int2(-1, -1);
        if (i < n) {
            const CtkSeam q = scratch[i];
            if (r.keep0[cb + r.mrep[cb + q.cl]]) { v.x = r.lab[cb + q.cl]; v.y = r.lab[cb + q.cr]; }
            res[(int64_t)t * a.ny + i] = v;
        }
        const int px = __shfl_up(v.x, 1), py = __shfl_up(v.y, 1);
        if (v.x >= 0 && v.x != v.y && !(lane > 0 && px == v.x && py == v.y)) {      // first row of a stretch with this pair
            a.mark[v.x] = 1; a.mark[v.y] = 1;
            uint32_t p = (uint32_t)v.x, q = (uint32_t)v.y;
            for (;;) {
                p = gfind(a.cl_parent, p);
                q = gfind(a.cl_parent, q);
                if (p == q) break;
                if (p < q) { const uint32_t s = p; p = q; q = s; }
                const uint32_t old = atomicMin(&a.cl_parent[p], q);
                if (old == p) break;
                p = old;
            }
        }
    }
}

// k_rs_rank_labels + k_fz_mark in ONE launch.  The marks need the labels of the two seam components of every seam row -- and a
// label is a cheap function of what k_rs_roots left behind: 1 + (roots in the blocks in front of the root's block) + (roots in
// front of it inside its block).  So the workgroups that number the components (part A, one per 256 components) and the waves
// that mark the seam rows (part B, one per timestep) do not wait for each other: every workgroup builds the prefix of the block
// counts in LDS and the seam waves derive the labels they need from the ROOT indices (lab_root, a copy k_rs_roots keeps while
// part A overwrites r.lab with the labels).
__global__ __launch_bounds__(256) void k_fz_rank_mark(ResolveDev r, SeamDev a, const CtkSeam *__restrict__ seams, const uint32_t *__restrict__ seam_cnt,
                                                      const uint32_t *__restrict__ seam_off, int2 *__restrict__ res, const uint32_t *__restrict__ bsum,
                                                      uint32_t nsb, uint32_t *__restrict__ total)
{
    if (dev_tables_bad(r)) return;
    extern __shared__ uint32_t pre[];                       // [nsb]
    __shared__ uint32_t sm[8];
    const uint32_t nc = dev_ncomps(r);
    const uint32_t nblk = min((nc + 255u) / 256u, nsb);
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
    auto label_of = [&](int32_t root) -> int32_t { return root < 0 ? 0 : (int32_t)(pre[(uint32_t)root >> 8] + r.rank[root]) + 1; };
    // part A: labels of the components
    if (blockIdx.x < nblk)
        for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < nc; g += nblk * 256u) r.lab[g] = label_of(r.lab_root[g]);
    // part B: seam rows of timestep 4 b + wave
    const int t = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    if (t >= (int)a.T) return;
    const uint32_t n = seam_cnt[t], cb = r.cprefix[t];
    const CtkSeam *scratch = seams + seam_off[t];
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        int2 v = make_int2(-1, -1);
        if (i < n) {
            const CtkSeam q = scratch[i];
            if (r.keep0[cb + r.mrep[cb + q.cl]]) { v.x = label_of(r.lab_root[cb + q.cl]); v.y = label_of(r.lab_root[cb + q.cr]); }
            res[(int64_t)t * a.ny + i] = v;
        }
        const int px = __shfl_up(v.x, 1), py = __shfl_up(v.y, 1);
        if (v.x >= 0 && v.x != v.y && !(lane > 0 && px == v.x && py == v.y)) {      // first row of a stretch with this pair
            a.mark[v.x] = 1; a.mark[v.y] = 1;
            uint32_t p = (uint32_t)v.x, q = (uint32_t)v.y;
            for (;;) {
                p = gfind(a.cl_parent, p);
                q = gfind(a.cl_parent, q);
                if (p == q) break;
                if (p < q) { const uint32_t s = p; p = q; q = s; }
                const uint32_t old = atomicMin(&a.cl_parent[p], q);
                if (old == p) break;
                p = old;
            }
        }
    }
}

// k_rs_cand_groups for the fused path: boxes of the marked labels, surviving seam rows run-length grouped, and for every group
// record the root of its cluster and the cluster's range of timesteps (the unions are complete: k_fz_mark is a launch of its own).
//
// A workgroup takes FZ_TW consecutive timesteps, one wave each.  The label boxes and the cluster ranges are min / max reductions
// whose addresses are HOT: a label that lives for hundreds of timesteps is updated by every one of them, from all eight XCDs,
// and same-address device-scope operations serialise at ~20-80 ns each (2000 x 721 x 1440: 0.4 ms of this kernel).  The waves of
// a workgroup therefore reduce into an LDS hash first and the workgroup touches global memory once per label: FZ_TW times fewer
// hot operations.  (Looking before the atomic does not help by itself: the L2s of the XCDs are not coherent with each other, so
// the look has to be a device-scope load -- just as hot as the atomic.)
__device__ __forceinline__ void fz_range_merge(SeamDev &a, uint32_t root, int32_t lo, int32_t hi)
{
    if (lo < __hip_atomic_load(&a.cl_tmin[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&a.cl_tmin[root], lo);
    if (hi > __hip_atomic_load(&a.cl_tmax[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&a.cl_tmax[root], hi);
}

__global__ __launch_bounds__(64 * FZ_TW) void k_fz_groups(ResolveDev r, SeamDev a, const CtkSeam *__restrict__ seams, const uint32_t *__restrict__ seam_cnt,
                                                          const uint32_t *__restrict__ seam_off, const int2 *__restrict__ res, int64_t t_begin)
{
    if (dev_tables_bad(r)) return;
    __shared__ int32_t hk[FZ_HS], hv[FZ_HS][6], ck[FZ_CS], cv[FZ_CS][2];
    for (int s = (int)threadIdx.x; s < FZ_HS; s += 64 * FZ_TW) {
        hk[s] = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) hv[s][k] = (k & 1) ? INT32_MIN : INT32_MAX;
    }
    for (int s = (int)threadIdx.x; s < FZ_CS; s += 64 * FZ_TW) { ck[s] = 0; cv[s][0] = INT32_MAX; cv[s][1] = INT32_MIN; }
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * FZ_TW + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63);
    const int ny = a.ny;
    const bool live = t < a.T;
    const uint32_t cb = live ? r.cprefix[t] : 0u, nct = live ? r.cprefix[t + 1] - cb : 0u;
    const int32_t tt = (int32_t)(t_begin + t);
    for (uint32_t c = lane; c < nct; c += 64) {
        const uint32_t g = cb + c;
        const int32_t l = r.lab[g];
        if (l <= 0 || !a.mark[l]) continue;
        const uint16_t *q = r.box + 4 * (int64_t)g;
        const int32_t v[6] = {tt, tt, (int32_t)q[0], (int32_t)q[1], (int32_t)q[2], (int32_t)q[3]};
        const int s = fz_slot(hk, FZ_HS - 1, l);
        if (s < 0) { fz_box_merge(a.lbox + 6 * (int64_t)l, v); continue; }        // (a crowded hash: straight to memory)
#pragma unroll
        for (int k = 0; k < 6; k++) { if (k & 1) atomicMax(&hv[s][k], v[k]); else atomicMin(&hv[s][k], v[k]); }
    }
    const uint32_t n = live ? seam_cnt[t] : 0u;
    const CtkSeam *sc = seams + (live ? seam_off[t] : 0u);
    const int2 *rs = res + t * ny;
    CtkCand *dst = a.recs + t * ny;                        // at most one group per seam row
    uint32_t ng = 0;
    bool c_valid = false;
    int32_t c_ll = 0, c_lr = 0, c_y = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        int2 v = make_int2(-1, -1);
        int32_t y = 0;
        if (i < n) {
            v = rs[i];
            if (v.x >= 0 && v.x == v.y && !a.mark[v.x]) v.x = -1;      // can never take part in an op
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
        if (start) {
            CtkCand g; g.t = tt; g.yy = y | (y << 16); g.ll = v.x; g.lr = v.y; dst[idx] = g;
            const uint32_t root = gfind(a.cl_parent, (uint32_t)v.x);
            a.rec_root[t * ny + idx] = root;
            const int cs = fz_slot(ck, FZ_CS - 1, (int32_t)root);                  // (roots are labels: >= 1)
            if (cs < 0) fz_range_merge(a, root, (int32_t)t, (int32_t)t);
            else { atomicMin(&cv[cs][0], (int32_t)t); atomicMax(&cv[cs][1], (int32_t)t); }
        }
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
        a.rec_cnt[t] = ng;
    }
    __syncthreads();
    for (int s = (int)threadIdx.x; s < FZ_HS; s += 64 * FZ_TW) if (hk[s]) fz_box_merge(a.lbox + 6 * (int64_t)hk[s], hv[s]);
    for (int s = (int)threadIdx.x; s < FZ_CS; s += 64 * FZ_TW) if (ck[s]) fz_range_merge(a, (uint32_t)ck[s], cv[s][0], cv[s][1]);
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
    if (ctk_guard_bad(a.poison - CTK_CNT_POISON)) return;         // (the kernels in front returned early: nothing here is valid)
    __shared__ CtkCand brec[SD_BATCH];
    const int lane = (int)threadIdx.x;
    // A cluster is driven by the workgroup of the FIRST timestep that holds one of its records (its "home"): the records of a
    // timestep are few, the clusters spread evenly over the timesteps, and nothing has to be scanned for roots.  The lane whose
    // record wins the claim word of the root (cl_nops: 0xffffffff = unclaimed) brings the cluster in.
    for (int32_t th = (int32_t)blockIdx.x; th < (int32_t)a.T; th += (int32_t)gridDim.x) {
      const uint32_t nrec = a.rec_cnt[th];
      uint32_t home_ops = 0;
      for (uint32_t i0 = 0; i0 < nrec; i0 += 64) {
      const uint32_t irec = i0 + lane;
      bool isr = false;
      uint32_t my_root = 0;
      int32_t my_tmax = -1;
      if (irec < nrec) {
          my_root = a.rec_root[(int64_t)th * a.ny + irec];
          const int32_t rt_min = a.cl_tmin[my_root];
          my_tmax = a.cl_tmax[my_root];                                    // (requested together)
          isr = rt_min == th;
      }
      if (nrec <= 64) {                                                    // the first record of its root in this timestep brings the cluster in
          bool dup = false;
          for (uint32_t j = 0; j + 1 < nrec; j++) { const uint32_t rj = (uint32_t)__shfl((int)my_root, (int)j); dup = dup || ((uint32_t)lane > j && rj == my_root); }
          isr = isr && !dup;
      } else if (isr) isr = atomicCAS(&a.cl_nops[my_root], 0xffffffffu, 0u) == 0xffffffffu;       // many records: a claim word per root
      if (isr && a.cl_shared && a.cl_shared[my_root]) isr = false;         // (time shards: driven with the other ranks' records)
      uint64_t roots = __ballot(isr);
      while (roots) {
        const int src = (int)__builtin_ctzll(roots);
        roots &= roots - 1;
        const uint32_t R = (uint32_t)SD_RL(my_root, src);
        const int32_t tmin = th, tmax = SD_RL(my_tmax, src);
        if (a.dbg == 1) continue;
        int32_t o_hi = -1, o_lo = -1, o_t0 = 0, o_t1 = -1, o_y0 = 0, o_y1 = -1, o_x0 = 0, o_x1 = -1;      // operation `lane`
        bool o_succ = false;                                               // a LATER operation has this one's `lo` as `hi`: a pixel it moved may move again
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
                        const uint64_t sm = __ballot(o_succ);
                        if (ka < 64) { yhi = min(yhi, SD_RL(o_y1, ka)); p0 = SD_RL(o_lo, ka); sa = ka + 1; ga = (sm >> ka) & 1ull; } else ga = false;
                        if (kb < 64) { yhi = min(yhi, SD_RL(o_y1, kb)); p1 = SD_RL(o_lo, kb); sb = kb + 1; gb = (sm >> kb) & 1ull; } else gb = false;
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
                if (lane < nops && o_lo == hi) o_succ = true;
                nops++;
                tl = tl || hi == sl; tr = tr || hi == sr;
                y++;                                                         // the next row sees the new op
            }
        };
        // Lane-parallel screen of the records of a chunk: true if, with the operations recorded so far, some row of the lane's
        // record leaves the two seam pixels with different labels (only then can the sequential step record anything).  The
        // fold of a pixel is one pass over the operations in order -- an operation applies if it is alive at the timestep, its box
        // holds the pixel and its `hi` is the label the pixel carries by then; `yhi` = last row for which the pass provably runs
        // the same way (as in `record`).  No operations yet: the pixels differ iff their labels do.
        auto screen = [&](bool mine, int32_t tlo, int32_t thi /* timesteps of the records screened */, int32_t tg, int32_t ya, int32_t yb, int sl, int sr) -> bool {
            const int n = SD_U(nops);
            if (n == 0) return mine && sl != sr;
            const uint64_t alive_ops = __ballot(lane < n && o_t0 <= thi && o_t1 >= tlo);      // (the others apply to none of them)
            bool act = false, open = mine;
            int32_t y = ya;
            while (__ballot(open)) {
                int32_t p0 = sl, p1 = sr, yhi = INT32_MAX;
                for (uint64_t m = alive_ops; m; m &= m - 1) {
                    const int j = (int)__builtin_ctzll(m);
                    const int32_t jh = SD_RL(o_hi, j), jl = SD_RL(o_lo, j), jt0 = SD_RL(o_t0, j), jt1 = SD_RL(o_t1, j);
                    const int32_t jy0 = SD_RL(o_y0, j), jy1 = SD_RL(o_y1, j), jx0 = SD_RL(o_x0, j), jx1 = SD_RL(o_x1, j);
                    const bool alive = tg >= jt0 && tg <= jt1, y_in = y >= jy0 && y <= jy1, below = y < jy0;
                    const bool ca = alive && jx0 <= 0 && jh == p0, cb = alive && jx1 >= a.nx - 1 && jh == p1;
                    if (ca && y_in) { p0 = jl; yhi = min(yhi, jy1); }
                    if (cb && y_in) { p1 = jl; yhi = min(yhi, jy1); }
                    if ((ca || cb) && below) yhi = min(yhi, jy0 - 1);
                }
                if (open) {
                    if (p0 != p1) { act = true; open = false; }
                    else { y = min(yb, yhi) + 1; open = y <= yb; }
                }
            }
            return act;
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
                int my_sl = 0, my_sr = 0;                                    // label slots of this lane's record
                for (int sidx = 0; sidx < nl; sidx++) {
                    const int32_t id = SD_RL(l_id, sidx);
                    if (c.ll == id) { need_l = false; my_sl = sidx; }
                    if (c.lr == id) { need_r = false; my_sr = sidx; }
                }
                const int nl0 = nl;
                for (;;) {
                    const uint64_t bl = __ballot(need_l), br = __ballot(need_r);
                    if (!(bl | br)) break;
                    if (nl >= a.lab_cap) { bad = true; break; }
                    const int32_t d = bl ? SD_RL(c.ll, __builtin_ctzll(bl)) : SD_RL(c.lr, __builtin_ctzll(br));
                    if (lane == nl) l_id = d;
                    if (c.ll == d) { need_l = false; my_sl = nl; }
                    if (c.lr == d) { need_r = false; my_sr = nl; }
                    nl++;
                }
                if (bad) break;
                if (lane >= nl0 && lane < nl) {
                    l_orig = l_id;                                          // (slots are keyed by the fresh label itself)
                    const int32_t *bx = a.lbox + 6 * (int64_t)l_id;
                    l_b0 = bx[0]; l_b1 = bx[1]; l_b2 = bx[2]; l_b3 = bx[3]; l_b4 = bx[4]; l_b5 = bx[5];
                }
                if (a.dbg == 3) continue;
                const unsigned long long tp0 = wall_clock64();
                // Most records do nothing (the rows of a pair whose operation is already recorded, rows with one label on both sides):
                // every lane first runs ITS record against the operations recorded so far -- `screen`, the same fold with the roles
                // swapped: records in the lanes, the operations read one after the other -- and only the records on which the two
                // seam pixels end up with different labels go through the sequential step, in order.  A new operation changes the
                // answers of the records behind it: they are screened again.  (A sequential step is ~0.8 us of dependent scalar
                // work; a cluster of 467 records needed 0.34 ms that way.)
                int cursor = 0;
                while (cursor < nchunk && !bad) {
                    const bool mine = lane >= cursor && lane < nchunk;
                    const bool acts = a.dbg == 7 ? mine : screen(mine, SD_RL(c.t, cursor), SD_RL(c.t, nchunk - 1), c.t, (int32_t)(c.yy & 0xffff), (int32_t)((uint32_t)c.yy >> 16), my_sl, my_sr);
                    uint64_t fl = __ballot(acts);
                    const int nops0 = SD_U(nops);
                    while (fl) {
                        const int k = (int)__builtin_ctzll(fl);
                        fl &= fl - 1;
                        const int32_t yy = SD_RL(c.yy, k);
                        record(SD_RL(c.t, k), yy & 0xffff, (int32_t)((uint32_t)yy >> 16), SD_RL(my_sl, k), SD_RL(my_sr, k));
                        cursor = k + 1;
                        if (nops != nops0 || bad) break;
                    }
                    if (nops == nops0) break;                                // every record that could act has been looked at
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
        home_ops += (uint32_t)nops;
        uint32_t base = (R - a.own_base) * SD_OPS_OWN;
        if (nops > SD_OPS_OWN || R < a.own_base || R - a.own_base >= a.own_ids) {
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
      if (lane == 0) a.t_nops[th] = home_ops;
    }
}
