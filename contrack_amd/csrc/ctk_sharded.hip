// ctk_sharded.hip -- the whole path on a TIME SHARD, one rank per GPU (included by ctk_api.hip).
//
// Rank r owns the timesteps [t_begin, t_begin + T) of a slab of T_total steps.  Everything that touches pixels is
// local.  What couples neighbouring shards is one timestep wide, and what couples all of them is a few hundred records:
//
//   X1  neighbour exchange (one RCCL group): my labelled LAST timestep -> rank r+1 (the "halo": bit mask, run prefixes,
//       run -> component ids: the compressed one-timestep label map), my FIRST timestep's bit mask -> rank r-1 (its
//       last timestep's forward overlap, contrack.py:718, needs nothing else)
//       overlap histogram and resolver tables are local; the halo's components are "timestep -1" of the local tables
//   X3  overlap filter (contrack.py:706-742): every rank iterates keep[t] = f(keep[t-1]) on its shard with the halo's
//       keep bits as boundary condition (first guess: all kept); all-gather of {converged?, keep bits of my last
//       timestep}; repeat while any rank changed or received changed bits.  The fixed point is the sequential result.
//   X4  3-D labelling (contrack.py:748-751): local union-find (halo components are nodes), own roots ranked locally;
//       all-gather of the sets that touch a shard boundary; boundary_resolve (ctk_seam.h) turns them into scipy's ids.
//   X5  seam merges (contrack.py:753-763): candidate groups whose labels never leave the shard are driven locally;
//       only the groups connected to a boundary-crossing label are all-gathered and driven identically on every rank.
//   X6  time extents (persistence, contrack.py:765-772): all-gather + min/max of the boundary-crossing ids only.
//   X7  counts.
// Bulk pixel data never crosses the fabric; the replicated work is the (small) boundary and shared-seam part.
#pragma once

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
struct HaloHeader {
    uint32_t nruns, ncomp, pad0, pad1;
};

// halo blob of the shard's last timestep: [HaloHeader][mask ny*W u64][wstart ny*W u16, padded][rowstart ny u32, padded][run_comp]
__global__ void k_halo_pack(const uint64_t *__restrict__ mask, const uint16_t *__restrict__ wstart, const uint32_t *__restrict__ rowstart,
                            const uint32_t *__restrict__ run_base, const uint32_t *__restrict__ run_comp, const uint32_t *__restrict__ ncomp,
                            int64_t T, int ny, int W, size_t off_w, size_t off_r, size_t off_c, size_t cap_runs, unsigned char *__restrict__ out)
{
    const int64_t t = T - 1;
    const size_t nw = (size_t)ny * W;
    const uint32_t rb = run_base[t], n = run_base[t + 1] - rb;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    if (i0 == 0) { HaloHeader hd; hd.nruns = n; hd.ncomp = ncomp[t]; hd.pad0 = 0; hd.pad1 = 0; *(HaloHeader *)out = hd; }
    uint64_t *om = (uint64_t *)(out + sizeof(HaloHeader));
    uint16_t *ow = (uint16_t *)(out + off_w);
    uint32_t *orow = (uint32_t *)(out + off_r), *oc = (uint32_t *)(out + off_c);
    for (size_t i = i0; i < nw; i += st) { om[i] = mask[(size_t)t * nw + i]; ow[i] = wstart[(size_t)t * nw + i]; }
    for (size_t i = i0; i < (size_t)ny; i += st) orow[i] = rowstart[(size_t)t * ny + i];
    for (size_t i = i0; i < (size_t)n && i < cap_runs; i += st) oc[i] = run_comp[rb + i];
}

// table rows of the halo components: each is its own (already seam-resolved) representative at "timestep -1"
__global__ void k_halo_comps_init(const uint32_t *__restrict__ nh_ptr, uint32_t *__restrict__ mrep, uint32_t *__restrict__ comp_t,
                                  uint16_t *__restrict__ box, int64_t *__restrict__ area)
{
    const uint32_t nh = *nh_ptr;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < nh; h += gridDim.x * blockDim.x) {
        mrep[h] = h; comp_t[h] = 0xffffffffu;
        box[4 * (size_t)h] = 0; box[4 * (size_t)h + 1] = 0; box[4 * (size_t)h + 2] = 0; box[4 * (size_t)h + 3] = 0;
        area[2 * (size_t)h] = 0; area[2 * (size_t)h + 1] = 0;
    }
}

// forward overlap of the shard's LAST timestep (contrack.py:718) from the next shard's first bit mask
__global__ __launch_bounds__(256) void k_sh_fwd_last(ResolveDev r, const uint64_t *__restrict__ mask, const uint16_t *__restrict__ wstart,
                                                     const uint32_t *__restrict__ rowstart, const uint32_t *__restrict__ run_base,
                                                     const uint32_t *__restrict__ run_comp, const uint64_t *__restrict__ mask_next,
                                                     const int64_t *__restrict__ wlo, const int64_t *__restrict__ whi, int ny, int W)
{
    const int64_t t = r.T - 1;
    const size_t nw = (size_t)ny * W;
    const uint64_t *mc = mask + (size_t)t * nw;
    const uint16_t *ws = wstart + (size_t)t * nw;
    const uint32_t *rs = rowstart + (size_t)t * ny;
    const uint32_t *rc = run_comp + run_base[t];
    const uint32_t cb = r.cprefix[t];
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nw; idx += (size_t)gridDim.x * blockDim.x) {
        const uint64_t c = mc[idx];
        uint64_t o = c & mask_next[idx];
        if (o == 0ull) continue;
        const int y = (int)(idx / W), w = (int)(idx - (size_t)y * W);
        const uint64_t cin = (w > 0) ? (mc[idx - 1] >> 63) : 0ull;
        const uint64_t sc = c & ~((c << 1) | cin);
        const uint32_t ec = rs[y] + ws[idx];
        while (o) {
            const int b = __builtin_ctzll(o);
            // all set bits of o inside ONE run of c: walk run by run
            const uint64_t below = (b + 1 >= 64) ? FULL64 : ((1ull << (b + 1)) - 1ull);
            const uint32_t run = ec + (uint32_t)__popcll(sc & below) - 1u;
            // the run's bits in this word: from its start (or bit 0) to the first clear bit of c at or after b
            const uint64_t cs = c >> b;
            const int len = (~cs == 0ull) ? 64 - b : __builtin_ctzll(~cs);
            const uint64_t seg = (len >= 64) ? FULL64 : (((1ull << len) - 1ull) << b);
            const int n = __popcll(o & seg);
            const uint32_t rep = cb + r.mrep[cb + rc[run]];
            atomicAdd((unsigned long long *)&r.F[2 * (int64_t)rep], (unsigned long long)((int64_t)n * wlo[y]));
            atomicAdd((unsigned long long *)&r.F[2 * (int64_t)rep + 1], (unsigned long long)((int64_t)n * whi[y]));
            o &= ~seg;
        }
    }
}

// X3 payload of one rank: [KeepHeader][keep bit (one byte) of every component of my last timestep, capB bytes]
struct KeepHeader {
    uint32_t not_conv, nlast, ambig /* components recorded for the exact fix-up */, tables_bad;
    uint32_t nc_own, passes, hint_c, hint_d;      // hint_*: capacities this rank would like for the shared seam records (X5)
};
struct BoundHeader;
__device__ __forceinline__ void dev_pack_boundary(const ResolveDev &r, uint32_t capB, unsigned char *__restrict__ out);
__global__ void k_sh_pack_keep(ResolveDev r, int it_first, int it_count, uint32_t capB, uint32_t hint_c, uint32_t hint_d, uint32_t fix_changed,
                               unsigned char *__restrict__ out, unsigned char *__restrict__ bound_out /* speculative X4: the boundary record too, or nullptr */)
{
    if (bound_out) dev_pack_boundary(r, capB, bound_out);
    const int64_t T = r.T;
    const uint32_t cb = r.cprefix[T - 1], nlast = r.cprefix[T] - cb;
    unsigned char *bits = out + sizeof(KeepHeader);
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < capB; c += gridDim.x * blockDim.x)
        bits[c] = (c < nlast) ? r.keep0[cb + r.mrep[cb + c]] : (unsigned char)0;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        // converged = the last pass launched changed nothing (passes after a fixed point return at once and record nothing)
        const bool any = it_count > 0 && __ballot(r.changed[(it_first + it_count - 1) * CTK_CHG_SLOTS + threadIdx.x] != 0u) != 0ull;
        if (threadIdx.x == 0) {
            KeepHeader hd;
            hd.not_conv = (any || fix_changed) ? 1u : 0u; hd.nlast = nlast; hd.ambig = *r.amb_cnt; hd.tables_bad = (dev_tables_bad(r) ? 1u : 0u) | ((__hip_atomic_load(r.poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & CTK_POISON_SPIN) ? 2u : 0u);
            hd.nc_own = r.cprefix[T] - (r.nh_ptr ? *r.nh_ptr : 0u); hd.passes = (uint32_t)(it_first + it_count); hd.hint_c = hint_c; hd.hint_d = hint_d;
            *(KeepHeader *)out = hd;
        }
    }
}

// After the all-gather: import the predecessor's bits into the halo components; decide -- identically on every rank --
// whether another round is needed: some rank has not converged, or some rank's bits differ from the previous round's
// (then its successor must re-evaluate).  prev = last round's gathered payloads (first round: "all kept").
#define CTK_SHM_CONTINUE 0
#define CTK_SHM_MAXNLAST 1
#define CTK_SHM_NCSUM_LO 2
#define CTK_SHM_NCSUM_HI 3
#define CTK_SHM_AMBIG    4
#define CTK_SHM_BAD      5
#define CTK_SHM_MYDIFF   6
#define CTK_SHM_HINT_C   7
#define CTK_SHM_HINT_D   8
#define CTK_SHM_MYFLAGGED 9
#define CTK_SHM_MAXFLAGGED 10
#define CTK_SHM_NPAIRS 11
#define CTK_SHM_STAMP 12         // written last: the host polls for it instead of draining the stream
__global__ __launch_bounds__(256) void k_sh_unpack_keep(ResolveDev r, const unsigned char *__restrict__ gathered, unsigned char *__restrict__ prev,
                                                        int first_round, int redo, size_t slot, uint32_t capB, int rank, int world, int it_next,
                                                        uint8_t *__restrict__ tdirty, uint32_t *__restrict__ mail,
                                                        // speculative X4: the boundary records travel with the bits (offset bound_off inside a
                                                        // rank's payload, bslot bytes) and go straight into pinned host memory
                                                        size_t bound_off, size_t bslot, uint32_t *__restrict__ bound_pinned,
                                                        const uint32_t *__restrict__ pair_cnt /* slot mode: grouped records per timestep, else nullptr */,
                                                        uint32_t stamp)
{
    if (pair_cnt) {                                             // how many grouped pair records the shard holds (statistics)
        __shared__ uint32_t s_np;
        if (threadIdx.x == 0) s_np = 0;
        __syncthreads();
        uint32_t v = 0;
        for (int64_t t = threadIdx.x; t < r.T; t += blockDim.x) v += pair_cnt[t];
        atomicAdd(&s_np, v);
        __syncthreads();
        if (threadIdx.x == 0) mail[CTK_SHM_NPAIRS] = s_np;
    }
    if (bound_off)
        for (int q = 0; q < world; q++) {
            const uint32_t *src = (const uint32_t *)(gathered + (size_t)q * slot + bound_off);
            for (size_t i = threadIdx.x; i < bslot / 4; i += blockDim.x) bound_pinned[(size_t)q * (bslot / 4) + i] = src[i];
        }
    if (r.pstate) for (int64_t t = threadIdx.x; t <= r.T; t += blockDim.x) r.pstate[(size_t)t * CTK_PSTATE_STRIDE] = 0u;      // (the next round of k_rs_pass_sys counts from zero)
    __shared__ uint32_t s_diff_any, s_my_diff;
    if (threadIdx.x == 0) { s_diff_any = 0; s_my_diff = 0; }
    __syncthreads();
    const uint32_t nh = r.nh_ptr ? *r.nh_ptr : 0u;
    // my halo
    if (rank > 0) {
        const unsigned char *pb = gathered + (size_t)(rank - 1) * slot + sizeof(KeepHeader);
        bool d = false;
        for (uint32_t hh = threadIdx.x; hh < nh && hh < capB; hh += blockDim.x) {
            const unsigned char nb = pb[hh];
            if (r.keep0[hh] != nb) { r.keep0[hh] = nb; d = true; }
        }
        if (d) s_my_diff = 1;
    }
    // every boundary: bits of rank q this round vs last round
    bool d = false;
    for (int q = 0; q + 1 < world; q++) {
        const unsigned char *nb = gathered + (size_t)q * slot + sizeof(KeepHeader), *ob = prev + (size_t)q * slot + sizeof(KeepHeader);
        const uint32_t n = min(((const KeepHeader *)(gathered + (size_t)q * slot))->nlast, capB);
        for (uint32_t c = threadIdx.x; c < n; c += blockDim.x) {
            const unsigned char o = first_round ? (unsigned char)1 : ob[c];
            if (nb[c] != o) d = true;
        }
    }
    if (d) s_diff_any = 1;
    __syncthreads();
    {
        const size_t kb = sizeof(KeepHeader) + (size_t)capB;                 // (the bits are all the next round compares)
        for (int q = 0; q < world; q++)
            for (size_t i = threadIdx.x; i < kb; i += blockDim.x) prev[(size_t)q * slot + i] = gathered[(size_t)q * slot + i];
    }
    if (threadIdx.x == 0) {
        uint32_t nc_any = 0, mx = 0, amb = 0, bad = 0, hc = 0, hd_ = 0, mxf = 0;
        uint64_t ncs = 0;
        for (int q = 0; q < world; q++) {
            const KeepHeader *hd = (const KeepHeader *)(gathered + (size_t)q * slot);
            nc_any |= hd->not_conv; mx = max(mx, hd->nlast); ncs += hd->nc_own; amb |= hd->ambig; bad |= hd->tables_bad; mxf = max(mxf, hd->ambig);
            hc = max(hc, hd->hint_c); hd_ = max(hd_, hd->hint_d);
        }
        mail[CTK_SHM_HINT_C] = hc; mail[CTK_SHM_HINT_D] = hd_; mail[CTK_SHM_MAXFLAGGED] = mxf;
        mail[CTK_SHM_MYFLAGGED] = ((const KeepHeader *)(gathered + (size_t)rank * slot))->ambig;
        mail[CTK_SHM_CONTINUE] = (nc_any || s_diff_any) ? 1u : 0u;
        mail[CTK_SHM_MAXNLAST] = mx;
        mail[CTK_SHM_NCSUM_LO] = (uint32_t)ncs; mail[CTK_SHM_NCSUM_HI] = (uint32_t)(ncs >> 32);
        mail[CTK_SHM_AMBIG] = amb; mail[CTK_SHM_BAD] = bad; mail[CTK_SHM_MYDIFF] = s_my_diff;
        // the next pass (it_next) must look at timestep 0 again iff the halo's bits changed
        const int64_t T = r.T;
        uint8_t *dprev = tdirty + (size_t)(((it_next - 1) & 1)) * (size_t)(T + 1), *dother = tdirty + (size_t)((it_next & 1)) * (size_t)(T + 1);
        // (redo: the exchange is repeated with a larger capacity; what the first attempt imported stays pending)
        dprev[0] = (s_my_diff || (redo && dprev[0])) ? 1 : 0;
        dother[0] = 0;
        if (s_my_diff && it_next > 0) r.changed[(it_next - 1) * CTK_CHG_SLOTS] = 1u;
    }
    // everything this kernel wrote into pinned host memory (the scalars above, the boundary records) is complete before the stamp
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&mail[CTK_SHM_STAMP], stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// X4 payload of one rank: [BoundHeader][int32 last[capB]][int32 halo[capB]]  (see boundary_resolve in ctk_seam.h)
struct BoundHeader {
    int32_t nlast, nh, nroots, pad;
};
__device__ __forceinline__ void dev_pack_boundary(const ResolveDev &r, uint32_t capB, unsigned char *__restrict__ out)
{
    const int64_t T = r.T;
    const uint32_t nh = r.nh_ptr ? *r.nh_ptr : 0u;
    const uint32_t cb = r.cprefix[T - 1], nlast = r.cprefix[T] - cb;
    int32_t *last = (int32_t *)(out + sizeof(BoundHeader)), *halo = last + capB;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < capB; c += gridDim.x * blockDim.x) {
        int32_t v = -1;
        if (c < nlast) {
            const int32_t root = r.lab[cb + c];                     // k_rs_roots: root index, -1 = filtered out
            if (root >= 0) v = ((uint32_t)root < nh) ? 2 * root + 1 : 2 * (int32_t)r.rank[root];
        }
        last[c] = v;
        halo[c] = (c < nh) ? r.lab[c] : -1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        BoundHeader hd;
        hd.nlast = (int32_t)nlast; hd.nh = (int32_t)nh; hd.nroots = (int32_t)r.rank[r.cprefix[T]]; hd.pad = 0;
        *(BoundHeader *)out = hd;
    }
}
__global__ void k_sh_pack_boundary(ResolveDev r, uint32_t capB, unsigned char *__restrict__ out) { dev_pack_boundary(r, capB, out); }

// the tables indexed by GLOBAL ids start empty (one launch instead of three fill commands, ~5 us of stream time each)
__global__ void k_sh_clear_tables(uint8_t *__restrict__ mark, uint32_t *__restrict__ dmap, int32_t *__restrict__ op_first, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { mark[i] = 0; dmap[i] = 0u; op_first[i] = -1; }
}

// scipy's ids from the local root ranks and the boundary resolution (pinned staging written by the host):
//   st[0..1] = off (int64), st[2] = nA, st[3] = nmark, then A[nA], Alab[nA], halo_label[nh], mark_labels[nmark]
__global__ void k_rs_labels_sh(ResolveDev r, const int32_t *__restrict__ st, uint8_t *__restrict__ mark,
                               // device seam path: the time extents and the write-stage counters are reset here (k_sh_seam_init's second half)
                               int32_t *__restrict__ ext = nullptr, int64_t n_labels = 0, uint32_t *__restrict__ counters = nullptr, uint32_t *__restrict__ scal = nullptr)
{
    if (ext) {
        const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stp = (int64_t)gridDim.x * blockDim.x;
        for (int64_t i = i0; i <= n_labels; i += stp) { ext[i] = INT32_MAX; ext[n_labels + 1 + i] = INT32_MIN; }
        for (int64_t i = i0; i < CTK_ZF_SLOTS; i += stp) ctk_zf_reset(counters, i);
        if (i0 == 0) {
            counters[CTK_CNT_WROTE_ZERO] = 0; counters[CTK_CNT_ALIVE] = 0; counters[CTK_CNT_TICKET] = 0; counters[CTK_CNT_NOPS] = 0;
            scal[0] = 0; scal[1] = 0; scal[2] = 0; scal[3] = 0;
        }
    }
    const uint32_t nc = dev_ncomps(r);
    const uint32_t nh = r.nh_ptr ? *r.nh_ptr : 0u;
    const int64_t off = *(const int64_t *)st;
    const int32_t nA = st[2], nmark = st[3];
    const int32_t *A = st + 4, *Alab = A + nA, *hlab = Alab + nA, *ml = hlab + nh;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nc; g += gridDim.x * blockDim.x) {
        const int32_t root = r.lab[g];
        int32_t l = 0;
        if (root >= 0) {
            if ((uint32_t)root < nh) l = hlab[root];
            else {
                const int32_t k = (int32_t)r.rank[root];
                int lo = 0, hi = nA;                                  // first absorbed root >= k
                while (lo < hi) { const int m = (lo + hi) >> 1; if (A[m] < k) lo = m + 1; else hi = m; }
                l = (lo < nA && A[lo] == k) ? Alab[lo] : (int32_t)(off + k - lo + 1);
            }
        }
        r.lab[g] = l;
    }
    // labels that reach a shard boundary always count as "can take part in a relabel operation": another shard may hold
    // the seam row that says so
    // ... and always get a dense id: their boxes (find_objects over ALL timesteps, contrack.py:753) are the union of what
    // every shard holding them contributes, whether or not this shard has a seam row with them
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nmark; i += gridDim.x * blockDim.x) {
        const int32_t l = ml[i];
        mark[l] = 1;
        if (cand_try_claim(r, l)) cand_publish(r, l, atomicAdd(r.dcount, 1u));
    }
}

// ------------------------------------------------------------------------------------------------
// X5 on the device (seam merges, contrack.py:753-763).  The clusters of labels that never leave the shard are driven by
// k_seam_driver exactly as in the one-call pass -- labels are the GLOBAL ids here, the label-indexed tables cover all NL of them --
// without the host: no synchronisation between the boundary resolution and the end of the pass.  Clusters that hold a label
// reaching a shard boundary ("shared") are packed by the device, all-gathered, and driven on every rank's HOST from the gathered
// records while its GPU drives the local clusters.
// ------------------------------------------------------------------------------------------------
struct ShSeamTabs {
    uint8_t *mark, *cl_shared, *cl_sent;
    uint32_t *dmap, *cl_parent, *cl_nops;
    int32_t *op_first, *cl_tmin, *cl_tmax, *lbox, *ext;
};
// label-indexed tables [0, n) with n = NL + 2, time extents [0, NL], the write-stage counters
// (the tables alone -- n_labels < 0 -- can be initialised before the boundary resolution says how many labels there are: for as
// many as the buffers hold; the extents, whose layout depends on that number, and the counters then follow in k_rs_labels_sh)
__global__ void k_sh_seam_init(ShSeamTabs tb, int64_t n, int64_t n_labels, uint32_t *__restrict__ counters, uint32_t *__restrict__ scal /* [4] */)
{
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = i0; i < n; i += st) {
        tb.mark[i] = 0; tb.cl_shared[i] = 0; tb.cl_sent[i] = 0; tb.dmap[i] = 0u; tb.op_first[i] = -1;
        tb.cl_parent[i] = (uint32_t)i; tb.cl_nops[i] = 0xffffffffu; tb.cl_tmin[i] = INT32_MAX; tb.cl_tmax[i] = -1;
        int32_t *b = tb.lbox + 6 * i;
        b[0] = INT32_MAX; b[1] = -1; b[2] = INT32_MAX; b[3] = -1; b[4] = INT32_MAX; b[5] = -1;
    }
    if (n_labels < 0) return;
    for (int64_t i = i0; i <= n_labels; i += st) { tb.ext[i] = INT32_MAX; tb.ext[n_labels + 1 + i] = INT32_MIN; }
    for (int64_t i = i0; i < CTK_ZF_SLOTS; i += st) ctk_zf_reset(counters, i);
    if (i0 == 0) {
        counters[CTK_CNT_WROTE_ZERO] = 0; counters[CTK_CNT_ALIVE] = 0; counters[CTK_CNT_TICKET] = 0; counters[CTK_CNT_NOPS] = 0;
        scal[0] = 0; scal[1] = 0; scal[2] = 0; scal[3] = 0;
    }
}
// a cluster that holds a label reaching a shard boundary is shared (st: the staging block of k_rs_labels_sh, whose last list
// holds those labels)
__global__ void k_sh_shared_mark(ResolveDev r, const int32_t *__restrict__ st, uint32_t *__restrict__ cl_parent, uint8_t *__restrict__ cl_shared)
{
    const uint32_t nh = r.nh_ptr ? *r.nh_ptr : 0u;
    const int32_t nA = st[2], nmark = st[3];
    const int32_t *ml = st + 4 + 2 * nA + nh;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nmark; i += gridDim.x * blockDim.x) cl_shared[gfind(cl_parent, (uint32_t)ml[i])] = 1;
}
// The shared clusters' group records (in (t, y) order) and labels (with this shard's part of their boxes) -> one payload of the
// X5 all-gather: [SeamHeader][CtkCand recs[capC]][{label, box[6]} labs[capD]].  The header carries the true counts: when any
// rank's exceed the capacities, everybody enlarges them and the exchange is repeated.  ONE workgroup (the order matters).
// scal[0] = group records of the shard in all (statistics).
struct ShSeamHeader { uint32_t ncand, nlab, pad0, pad1; };
__global__ __launch_bounds__(1024) void k_sh_pack_shared(ResolveDev r, SeamDev a, const int32_t *__restrict__ st, uint8_t *__restrict__ cl_sent, uint32_t capC,
                                                         uint32_t capD, unsigned char *__restrict__ out, uint32_t *__restrict__ toff /* [T] scratch */,
                                                         uint32_t *__restrict__ scal, int redo /* the claims of a first attempt are cleared first */)
{
    __shared__ uint32_t sm[17];
    __shared__ uint32_t nlab_s, nrec_all;
    const int tid = (int)threadIdx.x;
    const int64_t T = a.T;
    const int ny = a.ny;
    CtkCand *oc = (CtkCand *)(out + sizeof(ShSeamHeader));
    int32_t *ol = (int32_t *)(out + sizeof(ShSeamHeader) + (size_t)capC * sizeof(CtkCand));
    if (tid == 0) { nlab_s = 0; nrec_all = 0; }
    const uint32_t nh = r.nh_ptr ? *r.nh_ptr : 0u;
    const int32_t nA = st[2], nmark = st[3];
    const int32_t *ml = st + 4 + 2 * nA + nh;
    if (redo) {
        for (int64_t t = tid; t < T; t += 1024) {
            const uint32_t n = a.rec_cnt[t];
            for (uint32_t i = 0; i < n; i++) { const CtkCand c = a.recs[t * ny + i]; cl_sent[c.ll] = 0; cl_sent[c.lr] = 0; }
        }
        for (int32_t i = tid; i < nmark; i += 1024) cl_sent[ml[i]] = 0;
    }
    __syncthreads();
    uint32_t carry = 0, all = 0;
    for (int64_t t0 = 0; t0 < T; t0 += 1024) {
        const int64_t t = t0 + tid;
        uint32_t cnt = 0;
        if (t < T) {
            const uint32_t n = a.rec_cnt[t];
            all += n;
            for (uint32_t i = 0; i < n; i++) cnt += a.cl_shared[a.rec_root[t * ny + i]] ? 1u : 0u;
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan(cnt, sm, &tot);
        if (t < T) toff[t] = carry + ex;
        carry += tot;
    }
    all = wave_sum_u32(all);
    if ((tid & 63) == 0 && all) atomicAdd(&nrec_all, all);
    __syncthreads();
    auto send_label = [&](int32_t l) {
        // one claim per label: the byte is set by whoever comes first (a byte-wide atomic OR through the containing word)
        uint32_t *w = (uint32_t *)(cl_sent + ((size_t)l & ~(size_t)3));
        const uint32_t bit = 1u << (8 * ((uint32_t)l & 3u));
        if (atomicOr(w, bit) & bit) return;
        const uint32_t idx = atomicAdd(&nlab_s, 1u);
        if (idx < capD) {
            int32_t *q = ol + 7 * (size_t)idx;
            q[0] = l;
            const int32_t *b = a.lbox + 6 * (int64_t)l;
#pragma unroll
            for (int k = 0; k < 6; k++) q[1 + k] = b[k];
        }
    };
    for (int64_t t = tid; t < T; t += 1024) {
        const uint32_t n = a.rec_cnt[t];
        uint32_t j = toff[t];
        for (uint32_t i = 0; i < n; i++) {
            if (!a.cl_shared[a.rec_root[t * ny + i]]) continue;
            const CtkCand c = a.recs[t * ny + i];
            if (j < capC) oc[j] = c;
            j++;
            send_label(c.ll);
            if (c.lr != c.ll) send_label(c.lr);
        }
    }
    // labels that reach a shard boundary travel even without a record here: another shard may hold the rows, and their boxes
    // (find_objects over ALL timesteps, contrack.py:753) are the union of every shard's part
    for (int32_t i = tid; i < nmark; i += 1024) send_label(ml[i]);
    __syncthreads();
    if (tid == 0) {
        ShSeamHeader hd; hd.ncand = carry; hd.nlab = nlab_s; hd.pad0 = 0; hd.pad1 = 0;
        *(ShSeamHeader *)out = hd;
        scal[0] = nrec_all;
    }
}
// gathered payloads -> pinned host memory, then a stamp the host polls for (an event or a stream drain would also wait for the
// kernels enqueued BEHIND this one: the local clusters are driven while the host works on the shared ones).  One workgroup.
__global__ __launch_bounds__(1024) void k_sh_copy_stamp(const uint4 *__restrict__ src, uint4 *__restrict__ dst_pinned, size_t n16, uint32_t *__restrict__ word,
                                                        uint32_t stamp)
{
    for (size_t i = threadIdx.x; i < n16; i += 1024) dst_pinned[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(word, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the shared clusters' operations (driven on the host, identical on every rank) join the device's: staging block in pinned host
// memory [CtkOp ops[ng]] [int32 next[ng]] [int32 label[ng], nf used] [int32 first[ng], nf used], placed at slot `base` of the op arrays
__global__ void k_sh_ops_append(const int32_t *__restrict__ staging, int32_t ng, int32_t nf, uint32_t base, CtkOp *__restrict__ ops, int32_t *__restrict__ op_next,
                                int32_t *__restrict__ op_first)
{
    const int32_t *sn = staging + 8 * (int64_t)ng, *sl = sn + ng, *sf = sl + ng;      // (CtkOp = 8 words)
    int32_t *dst = (int32_t *)(ops + base);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 8 * (int64_t)ng; i += (int64_t)gridDim.x * blockDim.x) dst[i] = staging[i];
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += gridDim.x * blockDim.x) op_next[base + i] = sn[i] < 0 ? -1 : (int32_t)(base + (uint32_t)sn[i]);
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += gridDim.x * blockDim.x) op_first[sl[i]] = (int32_t)(base + (uint32_t)sf[i]);
}

// X6 (+ X7): time extents of the ids shared between shards, and the counts.  Payload of one rank:
//   [alive_own, zero_seen][lo, hi of every shared id]
// alive_own = ids numbered by THIS shard (l0 < id <= l1) that no other shard knows (not in elist), are present and survive
// persistence -- their extents are final before the exchange; the shared ids are counted by everybody from the reduced extents.
// zero_seen = a background pixel in a sample of the shard's mask: the output then certainly holds a 0 (len(np.unique) counts it,
// contrack.py:793).  Only when NO rank has seen one do the flags of the write pass have to be exchanged afterwards (k_sh_count).
// SH_PE_BLOCKS workgroups share the ids (a single one walked 33 000 ids in 15 us); each keeps the sorted list of shared ids in
// LDS for its searches; partial counts meet in two counters (zeroed by k_ops_ingest), the last workgroup writes the header.
// Lists beyond SH_PE_LDS ids: one workgroup, the list in global memory.
#define SH_PE_BLOCKS 32
#define SH_PE_LDS 2048
__global__ __launch_bounds__(256) void k_sh_pack_ext(const int32_t *__restrict__ elist_pinned, int32_t ne, const int32_t *__restrict__ ext, int64_t n_labels,
                                                     int64_t l0, int64_t l1, int persistence, const uint64_t *__restrict__ mask, int64_t nsample, int W,
                                                     uint64_t last_full /* valid bits of a row's last word */, int32_t *__restrict__ elist, int32_t *__restrict__ out,
                                                     uint32_t *__restrict__ counters, int lds_cap /* <= SH_PE_LDS (a test hook lowers it) */)
{
    __shared__ int32_t el[SH_PE_LDS];
    const int tid = (int)threadIdx.x;
    const bool in_lds = ne <= lds_cap;                                     // (else the launch has one workgroup)
    for (int32_t i = tid; i < ne; i += 256) { const int32_t l = elist_pinned[i]; if (in_lds) el[i] = l; if (!in_lds || blockIdx.x == 0) elist[i] = l; }
    __threadfence_block();
    __syncthreads();
    const int32_t *E = in_lds ? el : elist;
    const int64_t gtid = (int64_t)blockIdx.x * 256 + tid, gsize = (int64_t)gridDim.x * 256;
    for (int64_t i = gtid; i < ne; i += gsize) {
        const int32_t l = E[i];
        out[2 + 2 * i] = ext[l]; out[2 + 2 * i + 1] = ext[n_labels + 1 + l];
    }
    uint32_t v = 0;
    for (int64_t l = l0 + 1 + gtid; l <= l1; l += gsize) {
        const int64_t lo = ext[l], hi = ext[n_labels + 1 + l];
        if (!(hi >= lo && hi - lo + 1 >= persistence)) continue;
        int a = 0, b = ne;                                                   // first shared id >= l
        while (a < b) { const int m = (a + b) >> 1; if (E[m] < (int32_t)l) a = m + 1; else b = m; }
        if (!(a < ne && E[a] == (int32_t)l)) v++;
    }
    bool bg = false;
    for (int64_t k = gtid; k < nsample; k += gsize) bg = bg || mask[k] != (((int)(k % W) == W - 1) ? last_full : ~0ull);
    __shared__ uint32_t sm[4], sz[4];
    __shared__ bool last;
    const uint32_t sv = wave_sum_u32(v);
    const bool zany = __ballot(bg) != 0ull;
    if (lane_id() == 0) { sm[tid >> 6] = sv; sz[tid >> 6] = zany ? 1u : 0u; }
    __syncthreads();
    if (tid == 0) {
        const uint32_t tot = sm[0] + sm[1] + sm[2] + sm[3], z = sz[0] | sz[1] | sz[2] | sz[3];
        if (tot) atomicAdd(&counters[CTK_CNT_ALIVE], tot);
        if (z) atomicOr(&counters[CTK_CNT_WROTE_ZERO], 1u);
        __threadfence();
        last = atomicAdd(&counters[CTK_CNT_TICKET], 1u) == gridDim.x - 1;
        if (last) {
            out[0] = (int32_t)__hip_atomic_load(&counters[CTK_CNT_ALIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // bit 0: a background pixel was seen; bit 1: this rank's device seam driver gave up (cluster / op slots beyond its tables)
            out[1] = (int32_t)((__hip_atomic_load(&counters[CTK_CNT_WROTE_ZERO], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u) |
                               ((__hip_atomic_load(&counters[CTK_CNT_POISON], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (CTK_POISON_OPCAP | CTK_POISON_CLUSTER)) ? 2u : 0u));
        }
    }
}
// after the all-gather: extents of the shared ids = min / max over the shards; the job's count = everybody's own ids + the shared
// ids that survive; both go straight into pinned host memory (mail[0] = surviving ids, mail[1] = some rank has seen a 0)
__global__ __launch_bounds__(1024) void k_sh_reduce_ext(const int32_t *__restrict__ elist, int32_t ne, const int32_t *__restrict__ gathered, int world,
                                                        int32_t *__restrict__ ext, int64_t n_labels, int persistence, uint32_t *__restrict__ mail,
                                                        const uint32_t *__restrict__ counters = nullptr, const uint32_t *__restrict__ scal = nullptr,
                                                        const uint32_t *__restrict__ nc_ptr = nullptr, const uint32_t *__restrict__ t_nops = nullptr,
                                                        const uint32_t *__restrict__ rec_cnt = nullptr, int64_t T = 0)
{
    const size_t sw = 2 + 2 * (size_t)ne;                                  // words of one rank's payload
    uint32_t v = 0;
    for (int32_t i = (int32_t)threadIdx.x; i < ne; i += 1024) {
        int32_t lo = INT32_MAX, hi = INT32_MIN;
        for (int q = 0; q < world; q++) {
            lo = min(lo, gathered[(size_t)q * sw + 2 + 2 * i]);
            hi = max(hi, gathered[(size_t)q * sw + 2 + 2 * i + 1]);
        }
        const int32_t l = elist[i];
        ext[l] = lo; ext[n_labels + 1 + l] = hi;
        v += (hi >= lo && (int64_t)hi - (int64_t)lo + 1 >= persistence) ? 1u : 0u;
    }
    uint32_t own = 0, z = 0;
    for (int q = (int)threadIdx.x; q < world; q += 1024) { own += (uint32_t)gathered[(size_t)q * sw]; z |= (uint32_t)gathered[(size_t)q * sw + 1]; }
    // (statistics of the device seam driver: operations and group records of this shard)
    __shared__ uint32_t s_ops, s_recs;
    if (threadIdx.x == 0) { s_ops = 0; s_recs = 0; }
    __syncthreads();
    if (t_nops) {
        uint32_t a = 0, b = 0;
        for (int64_t t = threadIdx.x; t < T; t += 1024) { a += t_nops[t]; b += rec_cnt[t]; }
        a = wave_sum_u32(a); b = wave_sum_u32(b);
        if (lane_id() == 0) { if (a) atomicAdd(&s_ops, a); if (b) atomicAdd(&s_recs, b); }
    }
    __shared__ uint32_t sm[16], sz[16];
    const uint32_t sv = wave_sum_u32(v + own);
    const bool zany = __ballot((z & 1u) != 0u) != 0ull, pany = __ballot((z & 2u) != 0u) != 0ull;
    if (lane_id() == 0) { sm[threadIdx.x >> 6] = sv; sz[threadIdx.x >> 6] = (zany ? 1u : 0u) | (pany ? 2u : 0u); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0, zz = 0;
        for (int i = 0; i < 16; i++) { tot += sm[i]; zz |= sz[i]; }
        // [0] surviving ids of the job, [1] some rank has seen a background pixel, [2] some rank's device seam driver gave up,
        // [3] this rank's poison bits, [4] its operations, [5] its group records
        mail[0] = tot; mail[1] = zz & 1u; mail[2] = (zz >> 1) & 1u;
        // [6] its components, [7] / [8] its ungrouped / counted co-occurrence records
        mail[3] = counters ? counters[CTK_CNT_POISON] : 0u; mail[4] = t_nops ? s_ops : 0u; mail[5] = t_nops ? s_recs : (scal ? scal[0] : 0u);
        mail[6] = nc_ptr ? *nc_ptr : 0u; mail[7] = counters ? counters[CTK_CNT_UPAIRS] : 0u; mail[8] = counters ? counters[CTK_CNT_PAIRS] : 0u;
    }
}
// ids numbered by THIS shard (l0 < id <= l1) that are present and survive persistence; + "a zero was written"
__global__ __launch_bounds__(1024) void k_sh_count(const int32_t *__restrict__ ext, int64_t n_labels, int64_t l0, int64_t l1, int persistence,
                                                   const uint32_t *__restrict__ counters, uint32_t *__restrict__ out)
{
    uint32_t v = 0;
    for (int64_t l = l0 + 1 + threadIdx.x; l <= l1; l += 1024) {
        const int64_t lo = ext[l], hi = ext[n_labels + 1 + l];
        v += (hi >= lo && hi - lo + 1 >= persistence) ? 1u : 0u;
    }
    __shared__ uint32_t sm[16];
    const uint32_t s = wave_sum_u32(v);
    if (lane_id() == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    __shared__ uint32_t zw[16];
    const uint32_t zv = ctk_zf_mine(counters, (int)threadIdx.x, 1024);
    const bool zany = __ballot(zv != 0u) != 0ull;
    if (lane_id() == 0) zw[threadIdx.x >> 6] = zany ? 1u : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0, z = 0;
        for (int i = 0; i < 16; i++) { tot += sm[i]; z |= zw[i]; }
        out[0] = tot;
        out[1] = z;
    }
}

// ------------------------------------------------------------------------------------------------
// exact fix-up: per-row pixel counts of the components whose decision sits on a rounding boundary.  np.sum over
// weight_grid[slice][mask] (contrack.py:717-719) adds the row weights of the selected pixels in raster order: the array is
// fully described by how many pixels every row contributes.  counts[k][0|1|2][y] = pixels of flagged component k in row y:
// all / also set at t+1 / whose pixel at t-1 belongs to a component that is currently kept.
// ------------------------------------------------------------------------------------------------
struct PlaneRef {
    const uint64_t *mask;            // [ny][W]
    const uint16_t *wstart;
    const uint32_t *rowstart;
    const uint32_t *run_comp;        // runs of this timestep
};
__device__ __forceinline__ int32_t plane_comp_at(const PlaneRef &p, int W, int y, int w, int b)
{
    const uint64_t m = p.mask[(size_t)y * W + w];
    if (!((m >> b) & 1ull)) return -1;
    const uint64_t carry = w > 0 ? p.mask[(size_t)y * W + w - 1] >> 63 : 0ull;
    const uint64_t starts = m & ~((m << 1) | carry);
    const uint64_t below = b == 63 ? FULL64 : ((1ull << (b + 1)) - 1ull);
    return (int32_t)p.run_comp[p.rowstart[y] + p.wstart[(size_t)y * W + w] + (uint32_t)__popcll(starts & below) - 1u];
}

__global__ __launch_bounds__(256) void k_exact_counts(ResolveDev r, const uint32_t *__restrict__ list, const uint64_t *__restrict__ mask,
                                                      const uint16_t *__restrict__ wstart, const uint32_t *__restrict__ rowstart,
                                                      const uint32_t *__restrict__ run_base, const uint32_t *__restrict__ run_comp, PlaneRef halo,
                                                      const uint64_t *__restrict__ mask_next, int has_prev, int has_next, int ny, int W,
                                                      uint32_t *__restrict__ counts /* [n][3][ny], zeroed */)
{
    const uint32_t k = blockIdx.x, g = list[k];
    const int64_t T = r.T;
    const int t = (int)r.comp_t[g];
    const uint32_t cb = r.cprefix[t], c = g - cb;
    const size_t nw = (size_t)ny * W;
    PlaneRef cur = {mask + (size_t)t * nw, wstart + (size_t)t * nw, rowstart + (size_t)t * ny, run_comp + run_base[t]};
    const bool prev_local = t > 0, prev_halo = t == 0 && has_prev;
    PlaneRef prv = halo;
    if (prev_local) prv = {mask + (size_t)(t - 1) * nw, wstart + (size_t)(t - 1) * nw, rowstart + (size_t)(t - 1) * ny, run_comp + run_base[t - 1]};
    const uint32_t pb = r.cprefix[t - 1];                           // (t = 0: the halo components start at cprefix[-1] = 0)
    const uint64_t *nxt = (t + 1 < T) ? mask + (size_t)(t + 1) * nw : (has_next ? mask_next : nullptr);
    uint32_t *ca = counts + (size_t)k * 3 * ny, *cf = ca + ny, *cbk = cf + ny;
    for (size_t idx = threadIdx.x; idx < nw; idx += blockDim.x) {
        uint64_t m = cur.mask[idx];
        if (!m) continue;
        const int y = (int)(idx / W), w = (int)(idx - (size_t)y * W);
        uint32_t na = 0, nf = 0, nb = 0;
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int32_t cc = plane_comp_at(cur, W, y, w, b);
            if (cc < 0 || r.mrep[cb + cc] != c) continue;
            na++;
            if (nxt && ((nxt[idx] >> b) & 1ull)) nf++;
            if (prev_local || prev_halo) {
                const int32_t d = plane_comp_at(prv, W, y, w, b);
                if (d >= 0 && r.keep0[pb + r.mrep[pb + d]]) nb++;
            }
        }
        if (na) atomicAdd(&ca[y], na);
        if (nf) atomicAdd(&cf[y], nf);
        if (nb) atomicAdd(&cbk[y], nb);
    }
}

// the host's numpy-order sums are in place (r.ovr_val): switch the listed components over and make the next pass look at them
__global__ void k_exact_apply(ResolveDev r, uint32_t n, int it_next, uint8_t *__restrict__ tdirty)
{
    const int64_t T = r.T;
    uint8_t *dprev = tdirty + (size_t)((it_next - 1) & 1) * (size_t)(T + 1) + 1;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t g = r.amb_list[k];
        r.ovr_slot[g] = 0x80000000u | k;
        const int t = (int)r.comp_t[g];
        dprev[t - 1] = 1;                                            // "the predecessor changed": timestep t is evaluated again
        if (it_next > 0) r.changed[(it_next - 1) * CTK_CHG_SLOTS + (t & (CTK_CHG_SLOTS - 1))] = 1u;
    }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static size_t halo2_off_wstart(const ctk_handle *h) { return sizeof(HaloHeader) + (size_t)h->ny * h->W * 8; }
static size_t halo2_off_rowstart(const ctk_handle *h) { return halo2_off_wstart(h) + ctk_align8((size_t)h->ny * h->W * 2); }
static size_t halo2_off_runcomp(const ctk_handle *h) { return halo2_off_rowstart(h) + ctk_align8((size_t)h->ny * 4); }
static size_t halo2_max_runs(const ctk_handle *h) { return (size_t)h->ny * ((size_t)h->nx / 2 + 1); }
static size_t halo2_bytes(const ctk_handle *h) { return halo2_off_runcomp(h) + halo2_max_runs(h) * 4; }

// (halo_off_* in ctk_api.hip are the offsets behind the header)
static int shard_overlap_v2(ctk_handle *h) { return ctk_shard_overlap(h); }

struct ShardScratch {
    // host vectors kept between calls
    std::vector<unsigned char> hbuf;
    std::vector<BoundaryIn> bin;
    BoundaryOut bout;
    std::vector<int32_t> uf, elist, glabel, gbox, gfirst_lab, marks;
    std::vector<CtkCand> gcand, lcand;
    std::vector<CtkOp> ops_g, ops_l;
    std::vector<uint8_t> isglob;
    std::vector<int32_t> lmap, lorig, lbox;
    std::vector<double> ovr, ovr_prev, wsum;
    std::vector<uint32_t> cnt;
};

static void shard_scratch_free(ShardScratch *s) { delete s; }

// CTK_SHDEBUG=1: synchronise and report after every stage (finds the stage a fault belongs to)
static int g_shdbg = -1;
#define SHDBG(name) do { if (g_shdbg < 0) g_shdbg = getenv("CTK_SHDEBUG") ? 1 : 0; if (g_shdbg) { hipError_t e_ = hipStreamSynchronize(s); \
    fprintf(stderr, "[shard %d/%d] %-24s %s\n", rank, world, name, hipGetErrorString(e_)); } } while (0)

// X5 + X6 with the seam merges of the shard's own clusters driven on the DEVICE (see k_sh_seam_init).  Everything up to the extent
// exchange is enqueued here; the only wait is for the gathered shared clusters, and it ends as soon as their copy has reached
// pinned memory -- k_seam_driver, enqueued behind that copy, runs underneath the host's work on them.  mail2[64..69] are written
// by k_sh_reduce_ext at the end (the caller waits for the stream after the write pass).
#define CTK_SH_GOPS 16384            // op slots reserved for the shared clusters' operations (more: every rank falls back alike)
static int sharded_seam_device(ctk_handle *h, ctk_comm *c, ShardScratch &S, ResolveDev &r, const ResolveIn &in, ResolvePlan &pl, int64_t T, int64_t t_begin,
                               int ny, int nx, int W, int persistence, int64_t NL, int64_t lab0, int64_t lab1, bool any_boundary, uint32_t hint_c,
                               uint32_t hint_d, uint32_t *mail2, bool *too_many_shared_ops, size_t pre_inited, const void *const *pre_ptrs)
{
    hipStream_t s = h->stream;
    const int rank = c->rank, world = c->world;
    (void)rank; (void)W;
    *too_many_shared_ops = false;
    const size_t NT = (size_t)NL + 2;
    CTKCHK(ensure(h, h->rv_mark, NT)); CTKCHK(ensure(h, h->rv_dmap, NT * 4)); CTKCHK(ensure(h, h->op_first, NT * 4));
    CTKCHK(ensure(h, h->ext, ((size_t)NL + 1) * 8));
    CTKCHK(ensure(h, h->sd_parent, NT * 4)); CTKCHK(ensure(h, h->sd_tmin, NT * 4)); CTKCHK(ensure(h, h->sd_tmax, NT * 4));
    CTKCHK(ensure(h, h->sd_nops, NT * 4)); CTKCHK(ensure(h, h->sd_lbox, NT * 24));
    CTKCHK(ensure(h, h->sh_cl_shared, NT + 8)); CTKCHK(ensure(h, h->sh_cl_sent, NT + 8));
    CTKCHK(ensure(h, h->sd_root, (size_t)std::max<int64_t>(T * ny, 1) * 4));
    CTKCHK(ensure(h, h->seam_off, (size_t)(T + 1) * 4));
    r.mark = P<uint8_t>(h->rv_mark); r.dmap = P<uint32_t>(h->rv_dmap); r.op_first = P<int32_t>(h->op_first);
    // op slots: SD_OPS_OWN per id this shard numbered, a shared tail, and a reserve for the shared clusters' operations
    const uint32_t own_ids = (uint32_t)std::max<int64_t>(lab1 - lab0, 0);
    const uint64_t cap_dev = (uint64_t)own_ids * SD_OPS_OWN + h->op_cap_hint, cap_all = cap_dev + CTK_SH_GOPS;
    if (cap_all > 0x7ffffff0ull) return ctk_set_error(CTK_E_RANGE, "ctk_track_sharded: more op slots than 2^31");
    CTKCHK(ensure(h, h->ops, (size_t)cap_all * (sizeof(CtkOp) + 4)));
    uint32_t *scal = P<uint32_t>(h->rv_scalars) + 8;
    SeamDev sd;
    sd.dummy = nullptr;
    sd.cl_parent = P<uint32_t>(h->sd_parent); sd.cl_tmin = P<int32_t>(h->sd_tmin); sd.cl_tmax = P<int32_t>(h->sd_tmax); sd.cl_nops = P<uint32_t>(h->sd_nops);
    sd.lbox = P<int32_t>(h->sd_lbox); sd.mark = P<uint8_t>(h->rv_mark);
    sd.rec_root = P<uint32_t>(h->sd_root); sd.recs = P<CtkCand>(h->rv_cand_scratch); sd.rec_cnt = P<uint32_t>(h->rv_cand_cnt);
    sd.t_nops = P<uint32_t>(h->rv_cand_off);
    sd.ops = P<CtkOp>(h->ops); sd.op_next = (int32_t *)(P<CtkOp>(h->ops) + cap_all); sd.op_first = r.op_first;
    sd.op_count = P<uint32_t>(h->counters) + CTK_CNT_NOPS; sd.op_cap = (uint32_t)cap_dev; sd.own_ids = own_ids; sd.own_base = (uint32_t)(lab0 + 1);
    sd.cl_shared = any_boundary ? P<uint8_t>(h->sh_cl_shared) : nullptr;
    sd.poison = P<uint32_t>(h->counters) + CTK_CNT_POISON; sd.ny = ny; sd.nx = nx; sd.T = T;
    sd.dbg = 0;
    sd.lab_cap = h->debug_sd_lab ? std::min(h->debug_sd_lab, SD_LAB) : SD_LAB; sd.ops_cap = h->debug_sd_ops ? std::min(h->debug_sd_ops, 64) : 64;
    h->d_op_next = sd.op_next;
    h->nops = 1;                                                       // (nonzero: the folds look at the chains)
    const int32_t *st = (const int32_t *)h->h_lab;
    {
        Timer tm(h, CTK_K_RESOLVE);
        ShSeamTabs tb;
        tb.mark = P<uint8_t>(h->rv_mark); tb.cl_shared = P<uint8_t>(h->sh_cl_shared); tb.cl_sent = P<uint8_t>(h->sh_cl_sent);
        tb.dmap = P<uint32_t>(h->rv_dmap); tb.cl_parent = sd.cl_parent; tb.cl_nops = sd.cl_nops; tb.op_first = r.op_first;
        tb.cl_tmin = sd.cl_tmin; tb.cl_tmax = sd.cl_tmax; tb.lbox = sd.lbox; tb.ext = P<int32_t>(h->ext);
        // the tables were initialised ahead of the boundary resolution (seam_tables_preinit) unless they were too small or moved
        const void *now_ptrs[10] = {h->rv_mark.p, h->rv_dmap.p, h->op_first.p, h->sd_parent.p, h->sd_tmin.p, h->sd_tmax.p, h->sd_nops.p, h->sd_lbox.p,
                                    h->sh_cl_shared.p, h->sh_cl_sent.p};
        bool pre_ok = NT <= pre_inited;
        for (int k = 0; k < 10 && pre_ok; k++) pre_ok = pre_ptrs && pre_ptrs[k] == now_ptrs[k];
        if (!pre_ok) k_sh_seam_init<<<(int)std::min<size_t>((NT + 255) / 256, 2048), 256, 0, s>>>(tb, (int64_t)NT, -1, P<uint32_t>(h->counters), scal);
        k_rs_labels_sh<<<pl.gc, 256, 0, s>>>(r, st, P<uint8_t>(h->rv_mark), P<int32_t>(h->ext), NL, P<uint32_t>(h->counters), scal);
        k_fz_mark<<<(int)T, 64, 0, s>>>(r, sd, in.seams, in.seam_cnt, in.seam_off, P<int2>(h->rv_seam_res));
        k_fz_groups<<<(int)((T + FZ_TW - 1) / FZ_TW), 64 * FZ_TW, 0, s>>>(r, sd, in.seams, in.seam_cnt, in.seam_off, P<int2>(h->rv_seam_res), t_begin);
        HIPCHK(hipGetLastError());
    }
    auto launch_driver = [&]() -> int {
        Timer tm(h, CTK_K_RESOLVE);
        k_seam_driver<<<(int)std::min<int64_t>(T, 65536), 64, 0, s>>>(sd, t_begin);
        HIPCHK(hipGetLastError());
        return CTK_OK;
    };
    S.glabel.clear(); S.gbox.clear(); S.gcand.clear(); S.ops_g.clear();
    if (!any_boundary) {
        CTKCHK(launch_driver());
        if (h->debug_fail_stage == 5) { h->debug_fail_stage = 0; return ctk_set_error(CTK_E_INTERNAL, "injected failure at stage 5 (test hook)"); }
    } else {
        k_sh_shared_mark<<<8, 256, 0, s>>>(r, st, sd.cl_parent, P<uint8_t>(h->sh_cl_shared));
        HIPCHK(hipGetLastError());
        uint32_t capC = std::max<uint32_t>(hint_c, 256), capD = (std::max<uint32_t>(hint_d, 256) + 3u) & ~3u;
        size_t sslot = 0;
        bool driver_launched = false;
        for (int redo = 0;; redo = 1) {
            sslot = sizeof(ShSeamHeader) + (size_t)capC * sizeof(CtkCand) + (size_t)capD * 28;      // (a multiple of 16: capD is one of 4)
            CTKCHK(ensure_host(&h->h_seam, &h->h_seam_cap, sslot * (size_t)(world + 1), true));
            CTKCHK(ensure(h, h->sh_send, sslot));
            CTKCHK(ensure(h, h->sh_recv, sslot * (size_t)world));
            unsigned char *sb = (unsigned char *)h->h_seam;
            const uint32_t stamp = (uint32_t)((h->pass_no << 4) | (uint32_t)(redo ? 2 + (capC & 7u) : 1)) | 0x80000000u;
            mail2[100] = 0;
            k_sh_pack_shared<<<1, 1024, 0, s>>>(r, sd, st, P<uint8_t>(h->sh_cl_sent), capC, capD, (unsigned char *)h->sh_send.p, P<uint32_t>(h->seam_off), scal, redo);
            HIPCHK(hipGetLastError());
            CTKCHK(ctk_comm_allgather(c, h->sh_send.p, h->sh_recv.p, sslot));
            k_sh_copy_stamp<<<1, 1024, 0, s>>>((const uint4 *)h->sh_recv.p, (uint4 *)(sb + sslot), sslot * (size_t)world / 16, mail2 + 100, stamp);
            HIPCHK(hipGetLastError());
            if (!driver_launched) { driver_launched = true; CTKCHK(launch_driver()); }      // runs while the host works on the shared clusters
            CTKCHK(ctk_comm_wait_word(c, mail2 + 100, stamp));
            uint32_t mc = 0, md = 0;
            for (int q = 0; q < world; q++) {
                const ShSeamHeader *qh = (const ShSeamHeader *)(sb + sslot * (size_t)(q + 1));
                mc = std::max(mc, qh->ncand); md = std::max(md, qh->nlab);
            }
            if (mc <= capC && md <= capD) break;
            capC = std::max(capC, mc + mc / 2 + 64); capD = (std::max(capD, md + md / 2 + 64) + 3u) & ~3u;       // same on every rank
        }
        h->sh_capC = capC; h->sh_capD = capD;
        if (h->debug_fail_stage == 5) { h->debug_fail_stage = 0; return ctk_set_error(CTK_E_INTERNAL, "injected failure at stage 5 (test hook)"); }
        // merged table of the shared labels (boxes: union over the shards) and the shared candidate groups in (t, y) order
        const double t_host = now_ms();
        const unsigned char *gb = (const unsigned char *)h->h_seam + sslot;
        std::vector<std::pair<int32_t, int32_t>> &tmp = h->sh_pairs;      // (label, position) for the merge
        tmp.clear();
        for (int q = 0; q < world; q++) {
            const unsigned char *p = gb + sslot * (size_t)q;
            const ShSeamHeader *qh = (const ShSeamHeader *)p;
            const int32_t *ql = (const int32_t *)(p + sizeof(ShSeamHeader) + (size_t)capC * sizeof(CtkCand));
            for (uint32_t i = 0; i < qh->nlab; i++) tmp.emplace_back(ql[7 * i], (int32_t)(q * (int64_t)capD + i));
        }
        std::sort(tmp.begin(), tmp.end());
        for (size_t i = 0; i < tmp.size(); i++) {
            const int q = tmp[i].second / (int32_t)capD, k = tmp[i].second % (int32_t)capD;
            const int32_t *ql = (const int32_t *)(gb + sslot * (size_t)q + sizeof(ShSeamHeader) + (size_t)capC * sizeof(CtkCand)) + 7 * (size_t)k;
            if (S.glabel.empty() || S.glabel.back() != tmp[i].first) {
                S.glabel.push_back(tmp[i].first);
                S.gbox.insert(S.gbox.end(), ql + 1, ql + 7);
            } else {
                int32_t *b = &S.gbox[S.gbox.size() - 6];
                b[0] = std::min(b[0], ql[1]); b[1] = std::max(b[1], ql[2]); b[2] = std::min(b[2], ql[3]);
                b[3] = std::max(b[3], ql[4]); b[4] = std::min(b[4], ql[5]); b[5] = std::max(b[5], ql[6]);
            }
        }
        auto gid = [&](int32_t l) { return (int32_t)(std::lower_bound(S.glabel.begin(), S.glabel.end(), l) - S.glabel.begin()); };
        for (int q = 0; q < world; q++) {                                 // rank order = time order
            const unsigned char *p = gb + sslot * (size_t)q;
            const ShSeamHeader *qh = (const ShSeamHeader *)p;
            const CtkCand *qc = (const CtkCand *)(p + sizeof(ShSeamHeader));
            for (uint32_t i = 0; i < qh->ncand; i++) { CtkCand v = qc[i]; v.ll = gid(v.ll); v.lr = gid(v.lr); S.gcand.push_back(v); }
        }
        h->sd_glob.run(S.gcand.data(), (int64_t)S.gcand.size(), S.glabel.data(), S.gbox.data(), (int64_t)S.glabel.size(), nx, S.ops_g);
        h->ms[CTK_T_HOST_RESOLVE] += now_ms() - t_host;
    }
    h->stats[9] = h->sd_glob.loop_ns; h->stats[10] = h->sd_glob.nfold;
    h->stats[CTK_S_SHARED_ROWS] = (int64_t)S.gcand.size();
    // ids whose time extent is shared between shards: everything that reaches a boundary + the shared seam labels
    S.elist = S.bout.crossing;
    S.elist.insert(S.elist.end(), S.glabel.begin(), S.glabel.end());
    std::sort(S.elist.begin(), S.elist.end());
    S.elist.erase(std::unique(S.elist.begin(), S.elist.end()), S.elist.end());
    const int32_t ne = (int32_t)S.elist.size();
    const int64_t ng = (int64_t)S.ops_g.size();
    if (ng > CTK_SH_GOPS) { *too_many_shared_ops = true; return CTK_OK; }      // (the same list on every rank: everybody takes the host-driven form)
    // the shared clusters' operations + the list of shared ids -> pinned staging read by the kernels
    const size_t bytes = (size_t)std::max<int64_t>(ng, 1) * (sizeof(CtkOp) + 4 + 8) + (size_t)ne * 4 + 64;
    CTKCHK(ensure_host(&h->h_ops, &h->h_ops_cap, bytes));
    CtkOp *s_ops = (CtkOp *)h->h_ops;
    int32_t *s_next = (int32_t *)(s_ops + ng), *s_label = s_next + ng, *s_first = s_label + ng, *s_el = s_first + ng;
    int32_t nf = 0;
    if (ng) memcpy(s_ops, S.ops_g.data(), (size_t)ng * sizeof(CtkOp));
    for (int64_t i = 0; i < ng; i++) s_next[i] = h->sd_glob.next[(size_t)i];
    if (ng)
        for (size_t d = 0; d < S.glabel.size(); d++)
            if (h->sd_glob.first[d] >= 0) { s_label[nf] = S.glabel[d]; s_first[nf] = h->sd_glob.first[d]; nf++; }
    if (ne) memcpy(s_el, S.elist.data(), (size_t)ne * 4);
    h->stats[CTK_S_OPS] = ng;                                          // (+ the device's own, added when the pass has ended)
    if (ng) {
        k_sh_ops_append<<<(int)std::min<int64_t>((ng * 8 + 255) / 256, 256), 256, 0, s>>>((const int32_t *)h->h_ops, (int32_t)ng, nf, (uint32_t)cap_dev, sd.ops, sd.op_next, r.op_first);
        HIPCHK(hipGetLastError());
    }
    h->state = ST_TABLES;
    CTKCHK(launch_extents(h, true, true));                             // + the final id of every own component
    {
        const size_t eslot = 8 + (size_t)ne * 8;
        CTKCHK(ensure(h, h->sh_send, eslot));
        CTKCHK(ensure(h, h->sh_recv, eslot * (size_t)world));
        CTKCHK(ensure(h, h->sh_elist, (size_t)std::max(ne, 1) * 4));
        const int64_t nsample = std::min<int64_t>((int64_t)T * ny * h->W, 16384);
        const uint64_t last_full = (nx & 63) ? ((1ull << (nx & 63)) - 1ull) : ~0ull;
        const int pe_lds = h->debug_mail_d ? (int)std::min<uint32_t>(h->debug_mail_d, SH_PE_LDS) : SH_PE_LDS;      // (ctk_debug_set_mailbox)
        k_sh_pack_ext<<<ne <= pe_lds ? SH_PE_BLOCKS : 1, 256, 0, s>>>(s_el, ne, P<int32_t>(h->ext), NL, lab0, lab1, persistence, P<uint64_t>(h->mask), nsample, h->W,
                                                                      last_full, P<int32_t>(h->sh_elist), P<int32_t>(h->sh_send), P<uint32_t>(h->counters), pe_lds);
        HIPCHK(hipGetLastError());
        void *gath = world == 1 ? h->sh_send.p : h->sh_recv.p;
        CTKCHK(ctk_comm_allgather(c, h->sh_send.p, gath, eslot));
        k_sh_reduce_ext<<<1, 1024, 0, s>>>(P<int32_t>(h->sh_elist), ne, (const int32_t *)gath, world, P<int32_t>(h->ext), NL, persistence, mail2 + 64,
                                           P<uint32_t>(h->counters), scal, in.cprefix + T, sd.t_nops, sd.rec_cnt, T);
        HIPCHK(hipGetLastError());
    }
    h->state = ST_EXTENTS;
    return CTK_OK;
}

static int track_sharded_impl(ctk_handle *h, ctk_comm *c, const void *anom_dev, bool f64, int64_t T, int64_t t_begin, int64_t T_total, int ny, int nx,
                              const double *thr, int cmp_op, const float *wrow, double overlap, int persistence, int twosided, int32_t *flag_dev,
                              int64_t *n_tracked)
{
    if (!h || !c) return ctk_set_error(CTK_E_INVALID, "ctk_track_sharded: null handle or communicator");
    const int rank = c->rank, world = c->world;
    if (T < 1) return ctk_set_error(CTK_E_INVALID, "ctk_track_sharded: rank %d owns no timestep (every rank needs at least one: use at most T ranks)", rank);
    if (t_begin < 0 || t_begin + T > T_total || (rank == 0) != (t_begin == 0) || (rank == world - 1) != (t_begin + T == T_total))
        return ctk_set_error(CTK_E_INVALID, "ctk_track_sharded: shard [%lld, %lld) of %lld steps does not fit rank %d of %d", (long long)t_begin,
                             (long long)(t_begin + T), (long long)T_total, rank, world);
    if (c->stream != h->stream) return ctk_set_error(CTK_E_INVALID, "ctk_track_sharded: the communicator belongs to another handle");
    if (world > 448) return ctk_set_error(CTK_E_RANGE, "ctk_track_sharded: at most 448 ranks");         // (pinned scalar block of the counts)
    if (c->dead) return ctk_set_error(CTK_E_COMM, "ctk_track_sharded: the communicator was aborted by an earlier failure; create a new one");
    // from here on every buffer growth (hipFree waits for the device) and every wait goes through the communicator's guarded
    // wait, and every error return that the other ranks cannot see by themselves is published to them (ctk_comm_abort in the
    // C entry): no rank is left inside a collective
    ActiveComm active(h, c);
    // a failure decided on the SAME gathered data by every rank (all return the same code; the communicator stays usable)
#define COLLECTIVE_FAIL(...) do { h->sh_collective_err = true; return ctk_set_error(__VA_ARGS__); } while (0)
#define INJECT(k) do { if (h->debug_fail_stage == (k)) { h->debug_fail_stage = 0; return ctk_set_error(CTK_E_INTERNAL, "injected failure at stage %d (test hook)", (k)); } } while (0)
    const double t_call = now_ms();
    hipStream_t s = h->stream;
    const bool has_prev = rank > 0, has_next = rank + 1 < world;
    if (!h->shard) h->shard = new (std::nothrow) ShardScratch();
    if (!h->shard) return ctk_set_error(CTK_E_NOMEM, "out of memory");
    ShardScratch &S = *h->shard;
    S.ovr_prev.clear();

    // ---- stage 1: threshold, runs, 2-D labelling (compaction of the component tables waits for the halo) ----------------
    CTKCHK(shard_label2d_impl(h, anom_dev, f64, T, ny, nx, thr, cmp_op, wrow, has_prev ? 1 : 0, /*defer_compact=*/true));
    const int W = h->W;
    const size_t nw = (size_t)ny * W;
    SHDBG("label2d");
    INJECT(1);

    // ---- X1: halo forward, first mask plane backward ------------------------------------------------------------------
    const size_t hb = halo2_bytes(h);
    CTKCHK(ensure(h, h->halo_out, hb));
    CTKCHK(ensure(h, h->halo_in, hb));
    CTKCHK(ensure(h, h->sh_mask_next, nw * 8));
    if (has_next) {
        k_halo_pack<<<64, 256, 0, s>>>(P<uint64_t>(h->mask), P<uint16_t>(h->wstart), P<uint32_t>(h->rowstart), P<uint32_t>(h->run_base),
                                       P<uint32_t>(h->run_comp), P<uint32_t>(h->ncomp), T, ny, W, halo2_off_wstart(h), halo2_off_rowstart(h),
                                       halo2_off_runcomp(h), halo2_max_runs(h), (unsigned char *)h->halo_out.p);
        HIPCHK(hipGetLastError());
    }
    // (the zeroed header is remembered by address AND capacity: a regrown buffer may come back at the same address; every other
    // writer of halo_in -- ctk_shard_halo_import, a call with has_prev -- clears the flag)
    if (!has_prev && !(h->halo_in_zero && h->halo_in_zero_p == h->halo_in.p && h->halo_in_zero_cap == h->halo_in.cap)) {       // no halo: zero components (kept from call to call)
        HIPCHK(hipMemsetAsync(h->halo_in.p, 0, sizeof(HaloHeader), s));
        h->halo_in_zero = true; h->halo_in_zero_p = h->halo_in.p; h->halo_in_zero_cap = h->halo_in.cap;
    }
    if (has_prev) h->halo_in_zero = false;
    CTKCHK(ctk_comm_shift(c, +1, h->halo_out.p, hb, h->halo_in.p, hb));
    CTKCHK(ctk_comm_shift(c, -1, h->mask.p, nw * 8, h->sh_mask_next.p, nw * 8));
    const uint32_t *nh_ptr = &((const HaloHeader *)h->halo_in.p)->ncomp;
    h->halo_valid = has_prev;
    SHDBG("X1");
    INJECT(2);

    // ---- component tables: halo components first ("timestep -1"), then the shard's own in (t, c) order ------------------
    // One launch (k_compact_init, as in the one-call pass): every timestep's workgroup sums the component counts in front of it
    // itself, compacts its tables and initialises the resolver's per-component arrays; k_overlap then prepares every pair record
    // as it writes it.  (Before: scan, compaction, halo rows, k_rs_init and k_rs_pairs_slots -- five launches, ~29 us at 1 deg.)
    const bool sys_pass = !ctk_env().pass_launches && !h->no_sys;
    const bool one_init = !ctk_env().sh_no_slots;
    {
        Timer tm(h, CTK_K_SCAN);
        if (one_init) {
            const size_t R0 = (size_t)(h->total_runs ? h->total_runs : 1) + halo2_max_runs(h);
            CTKCHK(ensure(h, h->rv_F, R0 * 16)); CTKCHK(ensure(h, h->rv_B, R0 * 16));
            CTKCHK(ensure(h, h->rv_keep0, R0)); CTKCHK(ensure(h, h->rv_keep1, R0));
            CTKCHK(ensure(h, h->rv_touch, R0 * 4)); CTKCHK(ensure(h, h->rv_parent, R0 * 4));
            CTKCHK(ensure(h, h->rv_changed, (size_t)(CTK_MAX_JACOBI + 8) * CTK_CHG_SLOTS * 4));
            CTKCHK(ensure(h, h->rv_scalars, 64));
            CTKCHK(ensure(h, h->sh_ovr_slot, R0 * 4));
            if (sys_pass) CTKCHK(ensure(h, h->rv_pstate, (size_t)(T + 1) * 4 * CTK_PSTATE_STRIDE));
            CompInit ci;
            ci.F = P<int64_t>(h->rv_F); ci.B = P<int64_t>(h->rv_B); ci.keep0 = P<uint8_t>(h->rv_keep0); ci.keep1 = P<uint8_t>(h->rv_keep1);
            ci.touch = P<uint32_t>(h->rv_touch); ci.parent = P<uint32_t>(h->rv_parent); ci.changed = P<uint32_t>(h->rv_changed);
            ci.ambig = P<uint32_t>(h->rv_scalars) + 1; ci.pstate = sys_pass ? P<uint32_t>(h->rv_pstate) : nullptr;
            ci.next_tiny = (const int32_t *)(P<int64_t>(h->wlo) + 2 * (size_t)h->ny);
            ci.nchanged = (CTK_MAX_JACOBI + 1) * CTK_CHG_SLOTS; ci.pstride = CTK_PSTATE_STRIDE; ci.T = T;
            ci.base_ptr = nh_ptr; ci.ovr_slot = P<uint32_t>(h->sh_ovr_slot); ci.amb_cnt = P<uint32_t>(h->rv_scalars) + 2; ci.dcount = P<uint32_t>(h->rv_scalars);
            ci.bsum = nullptr;
            if (T > 4 * CTK_CI_BLOCK) {
                const int nb = (int)((T + CTK_CI_BLOCK - 1) / CTK_CI_BLOCK);
                CTKCHK(ensure(h, h->ci_bsum, (size_t)nb * 4));
                k_sum_blocks<<<nb, CTK_CI_BLOCK, 0, s>>>(P<uint32_t>(h->ncomp), T, P<uint32_t>(h->ci_bsum));
                ci.bsum = P<uint32_t>(h->ci_bsum);
            }
            k_compact_init<<<(int)T, 256, 0, s>>>(P<uint32_t>(h->run_base), P<uint32_t>(h->ncomp), CPX(h), P<uint32_t>(h->cs_mrep), P<uint32_t>(h->cs_box),
                                                  P<int64_t>(h->cs_area), P<uint32_t>(h->d_mrep), P<uint16_t>(h->d_box), P<int64_t>(h->d_area),
                                                  P<uint32_t>(h->d_comp_t), ci);
            h->fz_init = true;                        // (k_overlap: the resolver's view of every pair record is written with the record)
        } else {
            k_scan_u32<<<1, 1024, 0, s>>>(P<uint32_t>(h->ncomp), T, CPX(h), P<uint32_t>(h->counters) + CTK_CNT_OVERFLOW, nullptr, nh_ptr);
            k_compact_comps<<<(int)T, 256, 0, s>>>(P<uint32_t>(h->run_base), P<uint32_t>(h->ncomp), CPX(h), P<uint32_t>(h->cs_mrep), P<uint32_t>(h->cs_box),
                                                   P<int64_t>(h->cs_area), P<uint32_t>(h->d_mrep), P<uint16_t>(h->d_box), P<int64_t>(h->d_area),
                                                   P<uint32_t>(h->d_comp_t));
            k_halo_comps_init<<<16, 256, 0, s>>>(nh_ptr, P<uint32_t>(h->d_mrep), P<uint32_t>(h->d_comp_t), P<uint16_t>(h->d_box), P<int64_t>(h->d_area));
        }
        HIPCHK(hipGetLastError());
    }
    h->state = ST_LABELLED;
    SHDBG("compact");

    // ---- stage 2: co-occurrence histogram (the first local timestep against the halo) ------------------------------------
    // (fixed per-timestep slots for the pair records instead of one global counter that every timestep's workgroup adds to:
    // same-address atomics from eight XCDs, 11 of that kernel's 50 us at 1 deg)
    h->sh_slots = !ctk_env().sh_no_slots;
    const int rc_ov = shard_overlap_v2(h);
    h->sh_slots = false;
    const bool tables_ready = h->fz_init;             // (k_compact_init + k_overlap did what k_rs_init and k_rs_pairs* do)
    h->fz_init = false;
    CTKCHK(rc_ov);
    const uint32_t pslot = h->fz_pslot;
    SHDBG("overlap");

    // ---- resolver tables ---------------------------------------------------------------------------------------------
    const size_t HB = halo2_max_runs(h);
    const size_t R = (size_t)(h->total_runs ? h->total_runs : 1) + HB;
    CTKCHK(ensure(h, h->comp_label, R * 4));
    CTKCHK(ensure(h, h->seam_rowoff, (size_t)(T + 1) * 4));
    if (h->rowoff_T != T || h->rowoff_ny != ny || h->rowoff_p != h->seam_rowoff.p) {
        k_iota_mul<<<(int)((T + 255) / 256), 256, 0, s>>>(P<uint32_t>(h->seam_rowoff), (uint32_t)T, (uint32_t)ny);
        h->rowoff_T = T; h->rowoff_ny = ny; h->rowoff_p = h->seam_rowoff.p;
    }
    ResolveIn in;
    in.T = T; in.R = R;
    in.ncomp = P<uint32_t>(h->ncomp); in.cprefix = CPX(h); in.mrep = P<uint32_t>(h->d_mrep); in.comp_t = P<uint32_t>(h->d_comp_t);
    in.box = P<uint16_t>(h->d_box); in.area = P<int64_t>(h->d_area);
    in.pairs = P<CtkPair>(h->pairs); in.pair_cap = h->pair_cap; in.counters = P<uint32_t>(h->counters);
    in.pair_base = P<uint32_t>(h->pair_base); in.pair_cnt = P<uint32_t>(h->pair_cnt);
    in.seams = P<CtkSeam>(h->seams); in.seam_cnt = P<uint32_t>(h->seam_cnt); in.seam_off = P<uint32_t>(h->seam_rowoff);
    in.seam_cap = T * ny;
    in.comp_label = P<int32_t>(h->comp_label);
    in.extra_dense = 2 * HB;                          // boundary labels always get a dense id
    ResolvePlan pl;
    CTKCHK(rs_prepare(h, in, overlap, twosided, pl));
    ResolveDev &r = pl.r;
    r.nh_ptr = nh_ptr;
    CTKCHK(ensure(h, h->rv_lab_root, R * 4));
    r.lab_root = P<int32_t>(h->rv_lab_root);            // (k_rs_roots keeps a copy of the root indices: a second attempt of X5 starts from them)
    r.t_lo = has_prev ? 0 : 1;                       // global timesteps 1 .. T_total-2 are filtered
    r.t_hi = has_next ? (int)T - 1 : (int)T - 2;
    const uint32_t AMB_CAP = 1u << 16;
    CTKCHK(ensure(h, h->sh_ovr_slot, R * 4));
    CTKCHK(ensure(h, h->sh_ovr_val, (size_t)AMB_CAP * 24));
    CTKCHK(ensure(h, h->sh_amb_list, (size_t)AMB_CAP * 4));
    r.ovr_slot = P<uint32_t>(h->sh_ovr_slot); r.ovr_val = P<double>(h->sh_ovr_val); r.amb_list = P<uint32_t>(h->sh_amb_list); r.amb_cap = AMB_CAP;
    const int npass_grid = r.t_hi - r.t_lo + 1;
    if (sys_pass) { CTKCHK(ensure(h, h->rv_pstate, (size_t)(T + 1) * 4 * CTK_PSTATE_STRIDE)); r.pstate = P<uint32_t>(h->rv_pstate); }
    const int gc = pl.gc, gp = pl.gp, nsb = pl.nsb;

    // pinned scalars the device writes for the host
    if (!h->h_mail2) { HIPCHK(hipHostMalloc((void **)&h->h_mail2, 4096, hipHostMallocDefault)); memset(h->h_mail2, 0, 4096); }
    uint32_t *mail2 = h->h_mail2;

    // ---- X3: overlap filter with boundary exchange --------------------------------------------------------------------
    // Capacities of the exchanged records must be the SAME on every rank, whatever each handle has seen before: the boundary
    // component capacity starts at a fixed value and grows by consensus inside the call; for the shared seam records every
    // rank sends what it remembers and all take the maximum.
    uint32_t capB = 256;
    int it_done = 0, rounds = 0;
    uint64_t nc_sum = 0;
    bool first_round = true;
    {
        Timer tm(h, CTK_K_RESOLVE);
        if (!tables_ready) {
            k_rs_init<<<gc, 256, 0, s>>>(r);                               // (also zeroes the resolver's scalars)
            const int gps = (int)std::min<uint64_t>(((uint64_t)T * std::max<uint32_t>(pslot, 1u) + 255) / 256 + 16, 4096);
            if (pslot) k_rs_pairs_slots<<<gps, 256, 0, s>>>(r, in.pair_cnt, pslot);
            else k_rs_pairs<<<gp, 256, 0, s>>>(r);
        }
        if (has_next)
            k_sh_fwd_last<<<(int)std::min<size_t>((nw + 255) / 256, 1024), 256, 0, s>>>(r, P<uint64_t>(h->mask), P<uint16_t>(h->wstart), P<uint32_t>(h->rowstart),
                                                                                        P<uint32_t>(h->run_base), P<uint32_t>(h->run_comp),
                                                                                        P<uint64_t>(h->sh_mask_next), P<int64_t>(h->wlo),
                                                                                        P<int64_t>(h->wlo) + ny, ny, W);
        HIPCHK(hipGetLastError());
    }
    // Device seam path (X5): its label-indexed tables are initialised NOW -- for as many labels as the buffers held after the previous
    // call -- so that nothing but the labelling itself has to be launched behind the boundary resolution (a 4 us kernel there left
    // the GPU idle for the 12 us until the next launch arrived).  Buffers that turn out too small are initialised again there.
    size_t seam_pre_n = 0;
    const void *seam_pre_ptrs[10] = {h->rv_mark.p, h->rv_dmap.p, h->op_first.p, h->sd_parent.p, h->sd_tmin.p, h->sd_tmax.p, h->sd_nops.p, h->sd_lbox.p,
                                     h->sh_cl_shared.p, h->sh_cl_sent.p};
    if (!ctk_env().sh_host_seam && !(h->sh_dev_off_ny == ny && h->sh_dev_off_nx == nx) && T <= 65536) {
        size_t n = std::min(h->rv_mark.cap, std::min(h->sh_cl_shared.cap, h->sh_cl_sent.cap));
        for (const DevBuf *b : {&h->rv_dmap, &h->op_first, &h->sd_parent, &h->sd_tmin, &h->sd_tmax, &h->sd_nops}) n = std::min(n, b->cap / 4);
        n = std::min(n, h->sd_lbox.cap / 24);
        bool have = n >= 1024;
        for (int k = 0; k < 10; k++) have = have && seam_pre_ptrs[k] != nullptr;
        if (have) {
            ShSeamTabs tb;
            tb.mark = P<uint8_t>(h->rv_mark); tb.cl_shared = P<uint8_t>(h->sh_cl_shared); tb.cl_sent = P<uint8_t>(h->sh_cl_sent);
            tb.dmap = P<uint32_t>(h->rv_dmap); tb.cl_parent = P<uint32_t>(h->sd_parent); tb.cl_nops = P<uint32_t>(h->sd_nops); tb.op_first = P<int32_t>(h->op_first);
            tb.cl_tmin = P<int32_t>(h->sd_tmin); tb.cl_tmax = P<int32_t>(h->sd_tmax); tb.lbox = P<int32_t>(h->sd_lbox); tb.ext = nullptr;
            k_sh_seam_init<<<(int)std::min<size_t>((n + 255) / 256, 2048), 256, 0, s>>>(tb, (int64_t)n, -1, nullptr, nullptr);
            HIPCHK(hipGetLastError());
            seam_pre_n = n;
        }
    }
    bool prepped = false;                             // 1/areacon and the forward fractions: by the first filter launch itself where it can
    SHDBG("rs P1");
    bool fix_changed = false, last_was_fixup = false;
    bool spec = false, parent_dirty = false;         // see "Speculative X4" below
    int n_fixups = 0;
    for (;;) {
        const int npass = first_round ? h->filter_round : std::max(2, h->filter_round / 2);
        if (it_done + npass > CTK_MAX_JACOBI) COLLECTIVE_FAIL(CTK_E_RANGE, "ctk_track_sharded: overlap filter did not converge within %d passes", CTK_MAX_JACOBI);      // (the rounds are in lockstep)
        // (decided before the passes: a speculating round lets the filter kernel unite the pairs itself)
        spec = first_round ? world == 1 : true;
        if (ctk_env().no_spec_x4) spec = false;                      // (experiments: set it for every rank or for none)
        bool united = false;
        {
            Timer tm(h, CTK_K_RESOLVE);
            if (npass_grid > 0) {
                if (sys_pass && npass <= 24) {
                    // all passes of the round in one launch (neighbour hand-shake through LDS / pstate, zeroed by k_rs_init / by the
                    // previous round's k_sh_unpack_keep); with `spec` also the 3-D unions of the surviving pairs
                    if (spec && parent_dirty) k_rs_parent_init<<<gc, 256, 0, s>>>(r);
                    const int nb = (int)((T - r.t_lo + PB_G - 1) / PB_G);
                    if (nb > h->n_cus) k_rs_pass_blk_2pc<<<nb, 64 * PB_G, 0, s>>>(r, it_done, npass, in.pair_base, in.pair_cnt, r.pstate, prepped ? 0 : 1, spec ? 1 : 0);
                    else k_rs_pass_blk<<<nb, 64 * PB_G, 0, s>>>(r, it_done, npass, in.pair_base, in.pair_cnt, r.pstate, prepped ? 0 : 1, spec ? 1 : 0);
                    united = spec;
                    prepped = true;
                } else {
                    if (!prepped) { k_rs_prep<<<gc, 256, 0, s>>>(r); prepped = true; }
                    for (int it = it_done; it < it_done + npass; it++)
                        k_rs_pass<<<npass_grid, 64, 0, s>>>(r, it, in.pair_base, in.pair_cnt, P<uint8_t>(h->rv_tdirty));
                }
            }
            HIPCHK(hipGetLastError());
        }
        // Speculative X4: if this round turns out to be the last one (nobody changed anything), the 3-D labelling of the shard --
        // unions, roots, local ranks -- is what X4 would compute next, and its boundary record can travel with the bits: one
        // exchange less.  A round that is not the last wasted the labelling (~50 us at 1 deg).  The rule has to give the same answer
        // on every rank (the payload size depends on it), so it uses nothing a handle remembers: with one shard the first
        // round is the last unless the filter itself needs more passes; with several shards a dropped component on any boundary
        // means a second round, so the first one does not speculate and every later one does.
        if (spec) {
            Timer tm(h, CTK_K_RESOLVE);
            if (!united) {
                if (parent_dirty) k_rs_parent_init<<<gc, 256, 0, s>>>(r);
                { if (pslot) k_rs_unite_slots<<<(int)std::min<uint64_t>(((uint64_t)T * pslot + 255) / 256 + 16, 4096), 256, 0, s>>>(r, in.pair_cnt, pslot); else k_rs_unite<<<gp, 256, 0, s>>>(r); }
            }
            k_rs_roots<<<nsb, 256, 0, s>>>(r, P<uint32_t>(h->rv_bsum));                       // (nsb blocks of 256 components)
            k_rs_rank<<<nsb, 256, 0, s>>>(r.isroot, in.cprefix + T, P<uint32_t>(h->rv_bsum), r.rank, P<uint32_t>(h->rv_boff) + nsb);
            HIPCHK(hipGetLastError());
            parent_dirty = true;
        }
        for (int redo = 0;; redo = 1) {                                  // (repeated only when capB has to grow)
            const size_t kslot = sizeof(KeepHeader) + ctk_align8(capB), bslot = sizeof(BoundHeader) + (size_t)capB * 8;
            const size_t slot = kslot + bslot;
            CTKCHK(ensure(h, h->sh_send, slot));
            CTKCHK(ensure(h, h->sh_recv, slot * (size_t)world));
            CTKCHK(ensure_host(&h->h_shard, &h->h_shard_cap, bslot * (size_t)world + 4096, true));
            if (h->sh_prev.cap < slot * (size_t)world) { CTKCHK(ensure(h, h->sh_prev, slot * (size_t)world)); if (!first_round) return ctk_set_error(CTK_E_INTERNAL, "boundary buffer grew between rounds"); }
            const uint32_t keep_stamp = (++h->sh_stamp_seq) | 0x80000000u;
            mail2[CTK_SHM_STAMP] = 0;
            k_sh_pack_keep<<<8, 256, 0, s>>>(r, it_done, npass_grid > 0 ? npass : 0, capB, h->sh_capC, h->sh_capD, fix_changed ? 1u : 0u, (unsigned char *)h->sh_send.p,
                                             spec ? (unsigned char *)h->sh_send.p + kslot : nullptr);
            HIPCHK(hipGetLastError());
            // (one rank: what was "gathered" is what was packed -- no copy of the buffer onto itself)
            void *gath = world == 1 ? h->sh_send.p : h->sh_recv.p;
            CTKCHK(ctk_comm_allgather(c, h->sh_send.p, gath, slot));
            k_sh_unpack_keep<<<1, 256, 0, s>>>(r, (const unsigned char *)gath, (unsigned char *)h->sh_prev.p, first_round ? 1 : 0, redo, slot, capB, rank, world,
                                               it_done + npass, P<uint8_t>(h->rv_tdirty), mail2, spec ? kslot : 0, bslot, (uint32_t *)h->h_shard,
                                               pslot ? in.pair_cnt : nullptr, keep_stamp);
            HIPCHK(hipGetLastError());
            // (not a drain of the stream: the kernel's stamp in pinned memory arrives several microseconds before the completion signal)
            CTKCHK(ctk_comm_wait_word(c, mail2 + CTK_SHM_STAMP, keep_stamp));
            if (mail2[CTK_SHM_MAXNLAST] <= capB) break;
            if (!first_round) return ctk_set_error(CTK_E_INTERNAL, "boundary component count changed between rounds");
            capB = mail2[CTK_SHM_MAXNLAST] + mail2[CTK_SHM_MAXNLAST] / 2 + 64;      // same decision on every rank; nothing was imported
                                                                                   // beyond capB, the bits are simply exchanged again
        }
        it_done += npass;
        rounds++;
        first_round = false;
        INJECT(3);
        if (mail2[CTK_SHM_BAD] & 2u) {
            // An inter-workgroup wait of some rank's systolic filter pass gave up (bounded spins, ResolveDev::spin_limit).  Every rank
            // reads that from the same gathered headers: all repeat the call with one launch per filter pass (no waits), once.
            h->no_sys = true;
            h->stats[CTK_S_HOST_REASON] |= 8;
            if (h->sh_retrying) COLLECTIVE_FAIL(CTK_E_INTERNAL, "ctk_track_sharded: a filter pass reported a wait that gave up although none was launched");
            h->sh_retrying = true;
            const int rc2 = track_sharded_impl(h, c, anom_dev, f64, T, t_begin, T_total, ny, nx, thr, cmp_op, wrow, overlap, persistence, twosided, flag_dev, n_tracked);
            h->sh_retrying = false;
            h->stats[CTK_S_HOST_REASON] |= 8;
            return rc2;
        }
        if (mail2[CTK_SHM_BAD]) COLLECTIVE_FAIL(CTK_E_RANGE, "ctk_track_sharded: the co-occurrence table of some rank overflowed");
        nc_sum = (uint64_t)mail2[CTK_SHM_NCSUM_LO] | ((uint64_t)mail2[CTK_SHM_NCSUM_HI] << 32);
        fix_changed = false;
        if (mail2[CTK_SHM_CONTINUE]) { last_was_fixup = false; continue; }
        if (!mail2[CTK_SHM_AMBIG]) break;                               // converged, no decision on a rounding boundary anywhere
        if (last_was_fixup) break;                                      // ... or the numpy-order sums were just confirmed: nothing changed
        // ---- exact fix-up: some rank holds decisions on rounded area sums within rounding distance of the threshold --------
        const uint32_t nflag = mail2[CTK_SHM_MYFLAGGED];
        if (mail2[CTK_SHM_MAXFLAGGED] > AMB_CAP) COLLECTIVE_FAIL(CTK_E_RANGE, "ctk_track_sharded: more than %u overlap decisions on rounding boundaries on one rank", AMB_CAP);
        if (nflag) {
            const size_t cw = (size_t)nflag * 3 * (size_t)ny;
            CTKCHK(ensure(h, h->sh_counts, cw * 4));
            HIPCHK(hipMemsetAsync(h->sh_counts.p, 0, cw * 4, s));
            const char *hl = (const char *)h->halo_in.p + sizeof(HaloHeader);
            PlaneRef halo = {(const uint64_t *)hl, (const uint16_t *)(hl + halo_off_wstart(h)), (const uint32_t *)(hl + halo_off_rowstart(h)),
                             (const uint32_t *)(hl + halo_off_runcomp(h))};
            k_exact_counts<<<nflag, 256, 0, s>>>(r, r.amb_list, P<uint64_t>(h->mask), P<uint16_t>(h->wstart), P<uint32_t>(h->rowstart), P<uint32_t>(h->run_base),
                                                P<uint32_t>(h->run_comp), halo, P<uint64_t>(h->sh_mask_next), has_prev ? 1 : 0, has_next ? 1 : 0, ny, W,
                                                P<uint32_t>(h->sh_counts));
            HIPCHK(hipGetLastError());
            S.cnt.resize(cw);
            HIPCHK(hipMemcpyAsync(S.cnt.data(), h->sh_counts.p, cw * 4, hipMemcpyDeviceToHost, s));
            CTKCHK(ctk_comm_wait(c));
            S.ovr.assign((size_t)nflag * 3, 0.0);
            for (uint32_t k = 0; k < nflag; k++)
                for (int q = 0; q < 3; q++) {
                    const uint32_t *cy = &S.cnt[((size_t)k * 3 + q) * ny];
                    S.wsum.clear();
                    for (int y = 0; y < ny; y++) S.wsum.insert(S.wsum.end(), cy[y], (double)h->c_w[(size_t)y]);       // raster order
                    S.ovr[(size_t)k * 3 + q] = ctk_np_sum(S.wsum.data(), S.wsum.size());
                }
            fix_changed = S.ovr_prev.size() != S.ovr.size() || memcmp(S.ovr_prev.data(), S.ovr.data(), S.ovr.size() * 8) != 0;
            if (fix_changed) {
                CTKCHK(ensure_host(&h->h_lab, &h->h_lab_cap, S.ovr.size() * 8 + 64));
                memcpy(h->h_lab, S.ovr.data(), S.ovr.size() * 8);
                HIPCHK(hipMemcpyAsync(h->sh_ovr_val.p, h->h_lab, S.ovr.size() * 8, hipMemcpyHostToDevice, s));
                k_exact_apply<<<(int)((nflag + 255) / 256), 256, 0, s>>>(r, nflag, it_done, P<uint8_t>(h->rv_tdirty));
                HIPCHK(hipGetLastError());
                CTKCHK(ctk_comm_wait(c));                                 // (h_lab is reused below)
                S.ovr_prev = S.ovr;
            }
            n_fixups = (int)nflag;
        }
        last_was_fixup = true;
    }
    h->stats[CTK_S_EXACT_FIXUPS] = n_fixups;
    h->sh_capB = capB;
    const uint32_t hint_c = mail2[CTK_SHM_HINT_C], hint_d = mail2[CTK_SHM_HINT_D];
    SHDBG("X3");
    h->stats[CTK_S_FILTER_PASSES] = it_done; h->stats[CTK_S_FILTER_ROUNDS] = rounds;
    h->stats[CTK_S_AMBIGUOUS] = 0;                    // (decisions on rounding boundaries are resolved here: CTK_S_EXACT_FIXUPS)
    if (nc_sum > 0x7ffffff0ull) COLLECTIVE_FAIL(CTK_E_RANGE, "ctk_track_sharded: more than 2^31 components over all shards");

    // ---- X4: 3-D labelling -------------------------------------------------------------------------------------------
    const size_t bslot = sizeof(BoundHeader) + (size_t)capB * 8;
    CTKCHK(ensure(h, h->sh_send, bslot));
    CTKCHK(ensure(h, h->sh_recv, bslot * (size_t)world));
    CTKCHK(ensure_host(&h->h_shard, &h->h_shard_cap, bslot * (size_t)world + 4096, true));
    h->stats[CTK_S_X4_SPECULATED] = spec ? 1 : 0;
    if (!spec) {                                      // (the last round did not carry the boundary records: X4 as an exchange of its own)
        {
            Timer tm(h, CTK_K_RESOLVE);
            if (parent_dirty) k_rs_parent_init<<<gc, 256, 0, s>>>(r);
            { if (pslot) k_rs_unite_slots<<<(int)std::min<uint64_t>(((uint64_t)T * pslot + 255) / 256 + 16, 4096), 256, 0, s>>>(r, in.pair_cnt, pslot); else k_rs_unite<<<gp, 256, 0, s>>>(r); }
            const uint32_t *ncp = in.cprefix + T;
            k_rs_roots<<<nsb, 256, 0, s>>>(r, P<uint32_t>(h->rv_bsum));                       // (nsb blocks of 256 components)
            k_rs_rank<<<nsb, 256, 0, s>>>(r.isroot, ncp, P<uint32_t>(h->rv_bsum), r.rank, P<uint32_t>(h->rv_boff) + nsb);
            k_sh_pack_boundary<<<8, 256, 0, s>>>(r, capB, (unsigned char *)h->sh_send.p);
            HIPCHK(hipGetLastError());
        }
        SHDBG("pack boundary");
        CTKCHK(ctk_comm_allgather(c, h->sh_send.p, h->sh_recv.p, bslot));
        HIPCHK(hipMemcpyAsync(h->h_shard, h->sh_recv.p, bslot * (size_t)world, hipMemcpyDeviceToHost, s));
        CTKCHK(ctk_comm_wait(c));
    }
    S.bin.assign((size_t)world, BoundaryIn());
    for (int q = 0; q < world; q++) {
        const unsigned char *p = (const unsigned char *)h->h_shard + (size_t)q * bslot;
        const BoundHeader *hd = (const BoundHeader *)p;
        BoundaryIn &b = S.bin[(size_t)q];
        b.nlast = hd->nlast; b.nh = hd->nh; b.nroots = hd->nroots;
        b.last = (const int32_t *)(p + sizeof(BoundHeader)); b.halo = b.last + capB;
        if (b.nlast < 0 || b.nh < 0 || (uint32_t)b.nlast > capB || (uint32_t)b.nh > capB) COLLECTIVE_FAIL(CTK_E_INTERNAL, "boundary record of rank %d is malformed", q);
    }
    INJECT(4);
    if (!boundary_resolve(S.bin, S.bout)) COLLECTIVE_FAIL(CTK_E_INTERNAL, "ctk_track_sharded: the shards' boundary records contradict each other");
    const int64_t NL = S.bout.off[(size_t)world];
    if (NL > 0x7ffffff0ll) COLLECTIVE_FAIL(CTK_E_RANGE, "ctk_track_sharded: more than 2^31 - 16 ids");
    h->n_labels = NL; h->t_begin = t_begin;
    const int64_t lab0 = S.bout.off[(size_t)rank], lab1 = S.bout.off[(size_t)rank + 1];
    // labels to mark on this rank: those of my halo components and of my last timestep's components (if a shard follows)
    S.marks.clear();
    for (int32_t l : S.bout.halo_label[(size_t)rank]) if (l > 0) S.marks.push_back(l);
    if (has_next) for (int32_t l : S.bout.last_label[(size_t)rank]) if (l > 0) S.marks.push_back(l);
    std::sort(S.marks.begin(), S.marks.end());
    S.marks.erase(std::unique(S.marks.begin(), S.marks.end()), S.marks.end());
    // staging block read by k_rs_labels_sh straight from pinned memory
    {
        const std::vector<int32_t> &A = S.bout.absorbed[(size_t)rank], &AL = S.bout.absorbed_label[(size_t)rank], &HL = S.bout.halo_label[(size_t)rank];
        const size_t words = 4 + 2 * A.size() + HL.size() + S.marks.size();
        CTKCHK(ensure_host(&h->h_lab, &h->h_lab_cap, words * 4 + 64));
        int32_t *st = (int32_t *)h->h_lab;
        *(int64_t *)st = lab0; st[2] = (int32_t)A.size(); st[3] = (int32_t)S.marks.size();
        int32_t *p = st + 4;
        if (!A.empty()) { memcpy(p, A.data(), A.size() * 4); p += A.size(); memcpy(p, AL.data(), A.size() * 4); p += A.size(); }
        if (!HL.empty()) { memcpy(p, HL.data(), HL.size() * 4); p += HL.size(); }
        if (!S.marks.empty()) memcpy(p, S.marks.data(), S.marks.size() * 4);
    }
    // ---- X5 .. X7.  First attempt: seam merges of the shard's own clusters ON THE DEVICE (k_seam_driver, as in the one-call pass),
    // nothing between the boundary resolution and the end of the pass waits for the host; only the clusters shared between shards
    // -- gathered records, driven identically on every rank -- pass through it, underneath the device's work on the local ones.
    // A cluster beyond the device driver's tables (or beyond the op slots) poisons the attempt; the flag travels with the extent
    // exchange, so EVERY rank repeats X5 .. X7 on the host-driven form below (second attempt), and the grid stays there.
    const bool any_boundary_label_all = [&]() {
        if (ctk_env().sh_force_split) return true;        // (experiments: the split / exchange even without shared groups; every rank or none)
        for (int q = 0; q < world; q++) {
            for (int32_t l : S.bout.halo_label[(size_t)q]) if (l > 0) return true;
            if (q + 1 < world) for (int32_t l : S.bout.last_label[(size_t)q]) if (l > 0) return true;
        }
        return false;
    }();
    int64_t alive = 0;
    bool zero = false;
    for (int attempt = 0;; attempt++) {
    const bool dev_seam = attempt == 0 && !ctk_env().sh_host_seam && !(h->sh_dev_off_ny == ny && h->sh_dev_off_nx == nx) && T <= 65536 &&
                          NL + 2 < 0x7fffffffll;
    if (dev_seam) {
        bool too_many = false;
        CTKCHK(sharded_seam_device(h, c, S, r, in, pl, T, t_begin, ny, nx, W, persistence, NL, lab0, lab1, any_boundary_label_all, hint_c, hint_d, mail2, &too_many,
                                   seam_pre_n, seam_pre_ptrs));
        if (too_many) { h->stats[CTK_S_HOST_REASON] |= 16; continue; }       // (decided alike on every rank, nothing of X6 launched yet)
    } else {
    if (attempt > 0) {
        // the first attempt turned the root indices in r.lab into labels and handed out dense ids to the boundary labels: start over
        HIPCHK(hipMemcpyAsync(r.lab, r.lab_root, R * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemsetAsync(r.dcount, 0, 4, s));
    }
    // label-indexed tables are indexed by GLOBAL ids here
    CTKCHK(ensure(h, h->rv_mark, (size_t)NL + 2));
    CTKCHK(ensure(h, h->rv_dmap, ((size_t)NL + 2) * 4));
    CTKCHK(ensure(h, h->op_first, ((size_t)NL + 2) * 4));
    CTKCHK(ensure(h, h->ext, ((size_t)NL + 1) * 8));
    r.mark = P<uint8_t>(h->rv_mark); r.dmap = P<uint32_t>(h->rv_dmap); r.op_first = P<int32_t>(h->op_first);
    // mailbox of the candidate records (same scheme as the single-GPU path)
    CandMail &mail = pl.mail;
    {
        Timer tm(h, CTK_K_RESOLVE);
        k_sh_clear_tables<<<(int)std::min<int64_t>((NL + 2 + 255) / 256, 2048), 256, 0, s>>>(P<uint8_t>(h->rv_mark), P<uint32_t>(h->rv_dmap), P<int32_t>(h->op_first), NL + 2);
        k_rs_labels_sh<<<gc, 256, 0, s>>>(r, (const int32_t *)h->h_lab, P<uint8_t>(h->rv_mark));
        k_rs_cand_mark<<<(int)T, 256, 0, s>>>(r, in.seams, in.seam_cnt, in.seam_off, ny, P<uint8_t>(h->rv_mark), P<int2>(h->rv_seam_res));
        k_rs_cand_groups<<<(int)((T + FZ_TW - 1) / FZ_TW), 64 * FZ_TW, 0, s>>>(r, in.seams, in.seam_cnt, in.seam_off, P<int2>(h->rv_seam_res), P<uint8_t>(h->rv_mark), ny, t_begin,
                                               P<uint32_t>(h->rv_cand_cnt), P<CtkCand>(h->rv_cand_scratch));
        CTKCHK(launch_scan_u32(h, P<uint32_t>(h->rv_cand_cnt), T, P<uint32_t>(h->rv_cand_off)));
        k_compact_cands<<<(int)T, 64, 0, s>>>(r, P<CtkCand>(h->rv_cand_scratch), P<uint32_t>(h->rv_cand_cnt), P<uint32_t>(h->rv_cand_off), ny,
                                              P<CtkCand>(h->rv_cand), P<uint32_t>(h->rv_boff) + nsb, 0, 0, mail);
        HIPCHK(hipGetLastError());
        CTKCHK(ctk_comm_wait(c));
    }
    uint32_t hs[CTK_MAIL_SCALARS];
    memcpy(hs, mail.scal, sizeof(hs));
    SHDBG("labels+cands");
    const int64_t ncand = hs[CTK_MAIL_NCAND];
    const size_t nd = hs[CTK_MAIL_ND];
    h->stats[CTK_S_COMPONENTS] = (int64_t)hs[CTK_MAIL_NC]; h->stats[CTK_S_PAIRS] = (int64_t)(pslot ? mail2[CTK_SHM_NPAIRS] : hs[CTK_CNT_PAIRS]) + hs[CTK_CNT_UPAIRS];
    h->stats[CTK_S_SEAM_ROWS] = ncand; h->stats[CTK_S_LABELS] = NL; h->stats[CTK_S_UPAIRS] = hs[CTK_CNT_UPAIRS];
    h->mail_want_c = std::max<size_t>(h->mail_want_c, (size_t)ncand + (size_t)ncand / 2);
    h->mail_want_d = std::max<size_t>(h->mail_want_d, nd + nd / 2);
    // candidate records + dense label tables -> pageable copies
    const size_t cb = (size_t)ncand * sizeof(CtkCand);
    h->sd_cand.resize(cb + nd * 28 + 64);
    char *dst = (char *)h->sd_cand.data();
    if (ncand || nd) {                     // (boundary labels have dense ids even without a candidate record)
        if ((size_t)ncand <= mail.cap_c && nd <= mail.cap_d) {
            memcpy(dst, mail.cand, cb); memcpy(dst + cb, mail.dorig, nd * 4); memcpy(dst + cb + nd * 4, mail.dbox, nd * 24);
        } else {
            if (cb) HIPCHK(hipMemcpyAsync(dst, h->rv_cand.p, cb, hipMemcpyDeviceToHost, s));
            if (nd) {
                HIPCHK(hipMemcpyAsync(dst + cb, h->rv_dorig.p, nd * 4, hipMemcpyDeviceToHost, s));
                HIPCHK(hipMemcpyAsync(dst + cb + nd * 4, h->rv_dbox.p, nd * 24, hipMemcpyDeviceToHost, s));
            }
            CTKCHK(ctk_comm_wait(c));
        }
    }
    const CtkCand *hc = (const CtkCand *)dst;
    const int32_t *ho = (const int32_t *)(dst + cb), *hbx = ho + nd;

    static const bool hostprof = getenv("CTK_HOSTPROF") != nullptr;
    static double hp_acc[8] = {0}; static int hp_n = 0;
    double hp_t = now_ms();
    auto HP = [&](int k) { if (hostprof) { const double n = now_ms(); hp_acc[k] += n - hp_t; hp_t = n; } };
    // ---- X5: seam merges.  Candidate groups connected (through shared rows) to a label that reaches a shard boundary are
    // shared: all-gathered and driven identically everywhere; the others are this shard's own. ------------------------------
    const double t_host = now_ms();
    // No label reaches a shard boundary anywhere (always so with one shard): nobody has shared groups -- every rank knows it from
    // the boundary resolution, which all ranks computed from the same gathered records -- and the exchange is left out; the
    // candidate records go to the driver as they are.
    bool any_boundary_label = ctk_env().sh_force_split;      // (experiments: the split / exchange even without shared groups; every rank or none)
    for (int q = 0; q < world && !any_boundary_label; q++) {
        for (int32_t l : S.bout.halo_label[(size_t)q]) if (l > 0) { any_boundary_label = true; break; }
        if (q + 1 < world) for (int32_t l : S.bout.last_label[(size_t)q]) if (l > 0) { any_boundary_label = true; break; }
    }
    size_t nGd = 0;
    int64_t nGc = 0;
    if (any_boundary_label) {
        // clusters of labels connected through candidate rows; a cluster holding a boundary label is shared
        S.uf.resize(nd);
        for (size_t i = 0; i < nd; i++) S.uf[i] = (int32_t)i;
        auto find = [&](int32_t i) { while (S.uf[(size_t)i] != i) { S.uf[(size_t)i] = S.uf[(size_t)S.uf[(size_t)i]]; i = S.uf[(size_t)i]; } return i; };
        for (int64_t k = 0; k < ncand; k++) {
            const int32_t a = find(hc[k].ll), b = find(hc[k].lr);
            if (a != b) S.uf[(size_t)std::max(a, b)] = std::min(a, b);
        }
        for (size_t i = 0; i < nd; i++) S.uf[i] = find((int32_t)i);                 // flat from here on: S.uf[i] is the root
        S.isglob.assign(nd, 0);
        for (size_t i = 0; i < nd; i++)
            if (std::binary_search(S.marks.begin(), S.marks.end(), ho[i])) S.isglob[(size_t)S.uf[i]] = 1;
        for (size_t i = 0; i < nd; i++) if (S.isglob[(size_t)S.uf[i]]) nGd++;
        for (int64_t k = 0; k < ncand; k++) if (S.isglob[(size_t)S.uf[(size_t)hc[k].ll]]) nGc++;
    }
    auto shared = [&](int32_t dense) { return S.isglob[(size_t)S.uf[(size_t)dense]] != 0; };       // (only with any_boundary_label)
    const int32_t *lorig = ho;                         // labels of the groups driven here (all of them without shared groups)
    size_t n_lorig = nd;
    HP(0);
    uint32_t capC = std::max<uint32_t>(hint_c, 256), capD = std::max<uint32_t>(hint_d, 256);
    struct SeamHeader { uint32_t ncand, nlab, pad0, pad1; };
    size_t sslot = 0;
    bool local_done = false;
    auto drive_local = [&]() {
        // this shard's own groups (dense ids renumbered without the shared labels), driven here
        // (sized once, filled by index: a push_back / insert per label was a third of this step's time)
        S.lmap.resize(nd); S.lorig.resize(nd); S.lbox.resize(6 * nd); S.lcand.resize((size_t)ncand);
        size_t nl = 0, nk = 0;
        for (size_t i = 0; i < nd; i++) {
            if (shared((int32_t)i)) { S.lmap[i] = -1; continue; }
            S.lmap[i] = (int32_t)nl; S.lorig[nl] = ho[i]; memcpy(&S.lbox[6 * nl], hbx + 6 * i, 24); nl++;
        }
        for (int64_t k = 0; k < ncand; k++) {
            if (shared(hc[k].ll)) continue;
            CtkCand v = hc[k]; v.ll = S.lmap[(size_t)v.ll]; v.lr = S.lmap[(size_t)v.lr]; S.lcand[nk++] = v;
        }
        h->sd.run(S.lcand.data(), (int64_t)nk, S.lorig.data(), S.lbox.data(), (int64_t)nl, nx, S.ops_l);
        lorig = S.lorig.data(); n_lorig = nl;
    };
    if (!any_boundary_label) h->sd.run(hc, ncand, ho, hbx, (int64_t)nd, nx, S.ops_l);
    for (; any_boundary_label;) {
        sslot = sizeof(SeamHeader) + (size_t)capC * sizeof(CtkCand) + (size_t)capD * 28;
        CTKCHK(ensure_host(&h->h_seam, &h->h_seam_cap, sslot * (size_t)(world + 1), true));
        CTKCHK(ensure(h, h->sh_send, sslot));
        CTKCHK(ensure(h, h->sh_recv, sslot * (size_t)world));
        unsigned char *sb = (unsigned char *)h->h_seam;
        SeamHeader sh; sh.ncand = (uint32_t)nGc; sh.nlab = (uint32_t)nGd; sh.pad0 = 0; sh.pad1 = 0;
        memcpy(sb, &sh, sizeof(sh));
        if ((uint32_t)nGc <= capC && nGd <= capD) {
            CtkCand *oc = (CtkCand *)(sb + sizeof(SeamHeader));
            int32_t *ol = (int32_t *)(sb + sizeof(SeamHeader) + (size_t)capC * sizeof(CtkCand));
            size_t j = 0;
            for (int64_t k = 0; k < ncand; k++)
                if (shared(hc[k].ll)) { CtkCand q = hc[k]; q.ll = ho[q.ll]; q.lr = ho[q.lr]; oc[j++] = q; }      // GLOBAL labels
            j = 0;
            for (size_t i = 0; i < nd; i++)
                if (shared((int32_t)i)) { ol[7 * j] = ho[i]; memcpy(ol + 7 * j + 1, hbx + 6 * i, 24); j++; }
        }
        HIPCHK(hipMemcpyAsync(h->sh_send.p, sb, sslot, hipMemcpyHostToDevice, s));
        CTKCHK(ctk_comm_allgather(c, h->sh_send.p, h->sh_recv.p, sslot));
        HIPCHK(hipMemcpyAsync(sb + sslot, h->sh_recv.p, sslot * (size_t)world, hipMemcpyDeviceToHost, s));
        if (!local_done) { local_done = true; drive_local(); }          // (while the shared groups travel)
        CTKCHK(ctk_comm_wait(c));
        uint32_t mc = 0, md = 0;
        for (int q = 0; q < world; q++) {
            const SeamHeader *qh = (const SeamHeader *)(sb + sslot * (size_t)(q + 1));
            mc = std::max(mc, qh->ncand); md = std::max(md, qh->nlab);
        }
        if (mc <= capC && md <= capD) break;
        capC = std::max(capC, mc + mc / 2 + 64); capD = std::max(capD, md + md / 2 + 64);       // same on every rank
    }
    h->sh_capC = capC; h->sh_capD = capD;
    HP(1);
    INJECT(5);
    // merged table of the shared labels (boxes: union over the shards) and the shared candidate groups in (t, y) order
    S.glabel.clear(); S.gbox.clear(); S.gcand.clear();
    if (any_boundary_label) {
        const unsigned char *gb = (const unsigned char *)h->h_seam + sslot;
        std::vector<std::pair<int32_t, int32_t>> &tmp = h->sh_pairs;      // (label, position) for the merge
        tmp.clear();
        for (int q = 0; q < world; q++) {
            const unsigned char *p = gb + sslot * (size_t)q;
            const SeamHeader *qh = (const SeamHeader *)p;
            const int32_t *ql = (const int32_t *)(p + sizeof(SeamHeader) + (size_t)capC * sizeof(CtkCand));
            for (uint32_t i = 0; i < qh->nlab; i++) tmp.emplace_back(ql[7 * i], (int32_t)(q * (int64_t)capD + i));
        }
        std::sort(tmp.begin(), tmp.end());
        for (size_t i = 0; i < tmp.size(); i++) {
            const int q = tmp[i].second / (int32_t)capD, k = tmp[i].second % (int32_t)capD;
            const int32_t *ql = (const int32_t *)(gb + sslot * (size_t)q + sizeof(SeamHeader) + (size_t)capC * sizeof(CtkCand)) + 7 * (size_t)k;
            if (S.glabel.empty() || S.glabel.back() != tmp[i].first) {
                S.glabel.push_back(tmp[i].first);
                S.gbox.insert(S.gbox.end(), ql + 1, ql + 7);
            } else {
                int32_t *b = &S.gbox[S.gbox.size() - 6];
                b[0] = std::min(b[0], ql[1]); b[1] = std::max(b[1], ql[2]); b[2] = std::min(b[2], ql[3]);
                b[3] = std::max(b[3], ql[4]); b[4] = std::min(b[4], ql[5]); b[5] = std::max(b[5], ql[6]);
            }
        }
        auto gid = [&](int32_t l) { return (int32_t)(std::lower_bound(S.glabel.begin(), S.glabel.end(), l) - S.glabel.begin()); };
        for (int q = 0; q < world; q++) {                                 // rank order = time order
            const unsigned char *p = gb + sslot * (size_t)q;
            const SeamHeader *qh = (const SeamHeader *)p;
            const CtkCand *qc = (const CtkCand *)(p + sizeof(SeamHeader));
            for (uint32_t i = 0; i < qh->ncand; i++) { CtkCand v = qc[i]; v.ll = gid(v.ll); v.lr = gid(v.lr); S.gcand.push_back(v); }
        }
    }
    h->sd_glob.run(S.gcand.data(), (int64_t)S.gcand.size(), S.glabel.data(), S.gbox.data(), (int64_t)S.glabel.size(), nx, S.ops_g);
    h->stats[9] = h->sd.loop_ns + h->sd_glob.loop_ns; h->stats[10] = h->sd.nfold + h->sd_glob.nfold;
    h->stats[CTK_S_OPS] = (int64_t)(S.ops_g.size() + S.ops_l.size());
    h->stats[CTK_S_SHARED_ROWS] = (int64_t)S.gcand.size();
    // ids whose time extent is shared between shards: everything that reaches a boundary + the shared seam labels
    S.elist = S.bout.crossing;
    S.elist.insert(S.elist.end(), S.glabel.begin(), S.glabel.end());
    std::sort(S.elist.begin(), S.elist.end());
    S.elist.erase(std::unique(S.elist.begin(), S.elist.end()), S.elist.end());
    const int32_t ne = (int32_t)S.elist.size();
    h->ms[CTK_T_HOST_RESOLVE] += now_ms() - t_host;
    HP(2);

    // ---- ops -> device (one pinned staging block read by k_ops_ingest), then extents ---------------------------------------
    {
        const int64_t ng = (int64_t)S.ops_g.size(), nl = (int64_t)S.ops_l.size(), nops = ng + nl;
        h->nops = (int32_t)nops;
        const size_t bytes = (size_t)std::max<int64_t>(nops, 1) * (sizeof(CtkOp) + 4 + 8) + (size_t)ne * 4 + 64;
        CTKCHK(ensure_host(&h->h_ops, &h->h_ops_cap, bytes));
        CTKCHK(ensure(h, h->ops, bytes));
        CtkOp *s_ops = (CtkOp *)h->h_ops;
        int32_t *s_next = (int32_t *)(s_ops + nops), *s_label = s_next + nops, *s_first = s_label + nops, *s_el = s_first + nops;
        int32_t nf = 0;
        if (ng) memcpy(s_ops, S.ops_g.data(), (size_t)ng * sizeof(CtkOp));
        if (nl) memcpy(s_ops + ng, S.ops_l.data(), (size_t)nl * sizeof(CtkOp));
        for (int64_t i = 0; i < ng; i++) s_next[i] = h->sd_glob.next[(size_t)i];
        for (int64_t i = 0; i < nl; i++) s_next[ng + i] = h->sd.next[(size_t)i] < 0 ? -1 : (int32_t)(h->sd.next[(size_t)i] + ng);
        for (size_t d = 0; d < S.glabel.size(); d++)
            if (h->sd_glob.first[d] >= 0) { s_label[nf] = S.glabel[d]; s_first[nf] = h->sd_glob.first[d]; nf++; }
        for (size_t d = 0; d < n_lorig; d++)
            if (h->sd.first[d] >= 0) { s_label[nf] = lorig[d]; s_first[nf] = (int32_t)(h->sd.first[d] + ng); nf++; }
        if (ne) memcpy(s_el, S.elist.data(), (size_t)ne * 4);
        int32_t *d_next = (int32_t *)(P<CtkOp>(h->ops) + nops);
        h->d_op_next = d_next;
        const int64_t work = std::max<int64_t>(nops * 9, NL + 1);
        k_ops_ingest<<<(int)std::min<int64_t>((work + 255) / 256, 1024), 256, 0, s>>>((const int32_t *)h->h_ops, nops, nf, P<int32_t>(h->ops), P<int32_t>(h->op_first),
                                                                                      P<int32_t>(h->ext), NL, P<uint32_t>(h->counters));
        HIPCHK(hipGetLastError());
        h->state = ST_TABLES;
        h->total_comps = (uint32_t)hs[CTK_MAIL_NC];
        HP(3);
        if (hostprof && (++hp_n % 16) == 0) fprintf(stderr, "HOSTPROF uf+glob %.1f  local drive(+exchange) %.1f  merge+glob drive %.1f  staging+ingest %.1f us\n", hp_acc[0] / hp_n * 1e3, hp_acc[1] / hp_n * 1e3, hp_acc[2] / hp_n * 1e3, hp_acc[3] / hp_n * 1e3);
        CTKCHK(launch_extents(h, true, true));                           // + the final id of every own component
    SHDBG("extents");
        // ---- X6 + X7: extents of the shared ids and the counts, one exchange ----------------------------------------------
        {
            const size_t eslot = 8 + (size_t)ne * 8;
            CTKCHK(ensure(h, h->sh_send, eslot));
            CTKCHK(ensure(h, h->sh_recv, eslot * (size_t)world));
            CTKCHK(ensure(h, h->sh_elist, (size_t)std::max(ne, 1) * 4));
            const int64_t nsample = std::min<int64_t>((int64_t)T * ny * W, 16384);
            const uint64_t last_full = (nx & 63) ? ((1ull << (nx & 63)) - 1ull) : ~0ull;
            const int pe_lds = h->debug_mail_d ? (int)std::min<uint32_t>(h->debug_mail_d, SH_PE_LDS) : SH_PE_LDS;      // (ctk_debug_set_mailbox)
            k_sh_pack_ext<<<ne <= pe_lds ? SH_PE_BLOCKS : 1, 256, 0, s>>>(s_el, ne, P<int32_t>(h->ext), NL, lab0, lab1, persistence, P<uint64_t>(h->mask), nsample, W,
                                                                          last_full, P<int32_t>(h->sh_elist), P<int32_t>(h->sh_send), P<uint32_t>(h->counters), pe_lds);
            HIPCHK(hipGetLastError());
            CTKCHK(ctk_comm_allgather(c, h->sh_send.p, h->sh_recv.p, eslot));
            k_sh_reduce_ext<<<1, 1024, 0, s>>>(P<int32_t>(h->sh_elist), ne, P<int32_t>(h->sh_recv), world, P<int32_t>(h->ext), NL, persistence, mail2 + 64);
            HIPCHK(hipGetLastError());
        }
        h->state = ST_EXTENTS;
    }

    }   // (host-driven X5 / X6)

    // ---- persistence + write ----------------------------------------------------------------------------------------
    int cv_rows = 0;
    int32_t *cv = chunk_vals_for(h, flag_dev, &cv_rows);
    {
        Timer tm(h, CTK_K_RUNLABEL);
        k_run_values<<<(int)T, 256, 0, s>>>(P<uint32_t>(h->run_base), P<uint32_t>(h->run_comp), CPX(h), P<int32_t>(h->comp_label), P<int32_t>(h->ext), NL,
                                            persistence, P<uint32_t>(h->d_mrep), 0, 0, P<int32_t>(h->run_val), P<uint32_t>(h->rowstart), ny, cv_rows, cv);
        HIPCHK(hipGetLastError());
    }
    {
        Timer tm(h, CTK_K_RELABEL);
        CTKCHK((flag_dev || !h->sio) ? launch_relabel(h, persistence, flag_dev, true, cv) : stream_out(h, persistence, cv));
    }
    SHDBG("relabel");
    INJECT(6);
    // ---- counts: written into pinned memory by k_sh_reduce_ext.  A 0 in the output: certain when some rank has seen a background
    // pixel; otherwise (slabs that are foreground everywhere) the write pass' own flags are exchanged -- same decision on every rank
    CTKCHK(ctk_comm_wait(c));
    alive = mail2[64];
    zero = mail2[65] != 0;
    if (!zero) {
        CTKCHK(ensure(h, h->sh_send, 64));
        CTKCHK(ensure(h, h->sh_recv, 64 * (size_t)world));
        k_sh_count<<<1, 1024, 0, s>>>(P<int32_t>(h->ext), NL, lab0, lab1, persistence, P<uint32_t>(h->counters), P<uint32_t>(h->sh_send));
        HIPCHK(hipGetLastError());
        CTKCHK(ctk_comm_allgather(c, h->sh_send.p, h->sh_recv.p, 8));
        HIPCHK(hipMemcpyAsync(mail2 + 128, h->sh_recv.p, 8 * (size_t)world, hipMemcpyDeviceToHost, s));
        CTKCHK(ctk_comm_wait(c));
        for (int q = 0; q < world; q++) zero = zero || mail2[128 + 2 * q + 1] != 0;
    }
    if (dev_seam && mail2[66]) {
        // some rank's device driver gave up (every rank reads the same gathered word): the whole of X5 .. X7 again, host-driven
        const uint32_t pz = mail2[67];                                    // this rank's own poison bits
        if (pz & CTK_POISON_OPCAP) h->op_cap_hint = std::max<uint32_t>(h->op_cap_hint * 2, mail2[68] + mail2[68] / 2 + 1024);
        if (pz & CTK_POISON_CLUSTER) { h->sh_dev_off_ny = ny; h->sh_dev_off_nx = nx; }
        h->stats[CTK_S_HOST_REASON] |= 16;
        continue;
    }
    if (dev_seam) {
        h->stats[CTK_S_OPS] += (int64_t)mail2[68]; h->stats[CTK_S_SEAM_ROWS] = (int64_t)mail2[69]; h->stats[CTK_S_LABELS] = NL;
        h->stats[CTK_S_COMPONENTS] = (int64_t)mail2[70]; h->stats[CTK_S_UPAIRS] = (int64_t)mail2[71];
        h->stats[CTK_S_PAIRS] = (int64_t)(pslot ? mail2[CTK_SHM_NPAIRS] : mail2[72]) + mail2[71];
        h->total_comps = mail2[70];
        h->op_cap_hint = std::max<uint32_t>(h->op_cap_hint, mail2[68] * 2 + 1024);
    }
    h->stats[CTK_S_FUSED] = dev_seam ? 1 : 0;
    break;
    }   // attempts
    h->last_alive = alive;
    if (n_tracked) *n_tracked = alive + (zero ? 1 : 0) - 1;              // len(np.unique(flag)) - 1, contrack.py:793
    collect_event_times(h);
    h->state = ST_TABLES;
    h->ms[CTK_T_TOTAL] += now_ms() - t_call;
    return CTK_OK;
}

// C entries: an error the other ranks cannot see by themselves (anything but a COLLECTIVE_FAIL) is published to them and retires
// the communicator -- they return CTK_E_COMM from their next wait instead of sitting in a collective for ever
static int track_sharded_entry(ctk_handle *h, ctk_comm *c, const void *anom_dev, bool f64, int64_t T_local, int64_t t_begin, int64_t T_total, int ny, int nx,
                               const double *thr, int cmp_op, const float *wrow, double overlap, int persistence, int twosided, int32_t *flag_dev,
                               int64_t *n_tracked)
{
    if (h) h->sh_collective_err = false;
    const int rc = track_sharded_impl(h, c, anom_dev, f64, T_local, t_begin, T_total, ny, nx, thr, cmp_op, wrow, overlap, persistence, twosided, flag_dev, n_tracked);
    if (rc != CTK_OK && c && h && !h->sh_collective_err) {
        std::string msg = ctk_last_error();                          // (the abort drains the stream: keep the message of the cause)
        ctk_comm_abort(c, rc);
        ctk_set_error(rc, "%s", msg.c_str());
    }
    return rc;
}

extern "C" int ctk_debug_fail_at(ctk_handle *h, int stage)
{
    if (!h || stage < 0) return ctk_set_error(CTK_E_INVALID, "ctk_debug_fail_at: null handle or negative stage");
    h->debug_fail_stage = stage;
    return CTK_OK;
}

extern "C" int ctk_track_sharded_f32_dev(ctk_handle *h, ctk_comm *c, const float *anom_dev, int64_t T_local, int64_t t_begin, int64_t T_total, int ny, int nx,
                                         const double *thr, int cmp_op, const float *wrow, double overlap, int persistence, int twosided,
                                         int32_t *flag_dev, int64_t *n_tracked)
{
    return track_sharded_entry(h, c, anom_dev, false, T_local, t_begin, T_total, ny, nx, thr, cmp_op, wrow, overlap, persistence, twosided, flag_dev, n_tracked);
}
extern "C" int ctk_track_sharded_f64_dev(ctk_handle *h, ctk_comm *c, const double *anom_dev, int64_t T_local, int64_t t_begin, int64_t T_total, int ny, int nx,
                                         const double *thr, int cmp_op, const float *wrow, double overlap, int persistence, int twosided,
                                         int32_t *flag_dev, int64_t *n_tracked)
{
    return track_sharded_entry(h, c, anom_dev, true, T_local, t_begin, T_total, ny, nx, thr, cmp_op, wrow, overlap, persistence, twosided, flag_dev, n_tracked);
}
