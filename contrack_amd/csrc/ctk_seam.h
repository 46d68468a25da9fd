// ctk_seam.h -- host side of the bbox-confined seam merge (contrack/contrack.py:753-763): the sequential driver on
// candidate records, and what the time-sharded path adds around it (GPU-free; internal).
#pragma once
#include "ctk_tables.h"

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <vector>

// contrack.py:753-763 on {t, y, label at x=0, label at x=nx-1} records in (t, y) order.  Labels are DENSE ids of the
// labels that occur in the records (orig[id] = fresh label, box[id] = its box): every table of the driver has a few
// thousand entries and stays in the CPU's L1/L2, whatever the number of fresh labels.  Flat arrays: per label the
// chain of ops that have it as `hi`, in execution order.  `ops` (cleared first) receives the ops with the fresh labels.
struct SeamDriver {
    std::vector<int32_t> first, last, inflow, next, lo_d;      // first/last/inflow: per dense id; next/lo_d: per op
    int64_t nfold = 0, loop_ns = 0;

    void run(const CtkCand *cand, int64_t ncand, const int32_t *orig, const int32_t *box, int64_t nd, int nx, std::vector<CtkOp> &ops)
    {
        ops.clear();
        first.assign((size_t)nd + 1, -1);
        last.assign((size_t)nd + 1, -1);
        next.clear();
        lo_d.clear();                                             // dense id of ops[i].lo
        // fold of the ops over a seam pixel.  Consecutive seam rows of one blob ask the same question with y+1; the
        // answer is reused while it provably cannot change: same label / timestep / side, no op recorded since, and
        // y inside the interval over which every box test taken on the way gives the same outcome.
        struct Memo { int32_t l = -1, t = -1, ylo = 0, yhi = -1, res = 0; size_t epoch = (size_t)-1; };
        Memo memo[2];
        auto fold = [&](int side, int32_t l0, int32_t t, int32_t y, int32_t x) {
            Memo &m = memo[side];
            if (m.l == l0 && m.t == t && m.epoch == ops.size() && y >= m.ylo && y <= m.yhi) return m.res;
            int32_t l = l0, s = 0, ylo = INT32_MIN, yhi = INT32_MAX;
            for (;;) {
                bool moved = false;
                for (int32_t idx = first[(size_t)l]; idx >= 0; idx = next[(size_t)idx]) {
                    if (idx < s) continue;
                    const CtkOp &o = ops[(size_t)idx];
                    const bool tx_in = t >= o.t0 && t <= o.t1 && x >= o.x0 && x <= o.x1;
                    if (!tx_in) continue;                                   // outside for every y
                    if (y >= o.y0 && y <= o.y1) {                           // inside: stays inside for y in [y0, y1]
                        ylo = std::max(ylo, o.y0); yhi = std::min(yhi, o.y1);
                        l = lo_d[(size_t)idx]; s = idx + 1; moved = true; break;
                    }
                    if (y < o.y0) yhi = std::min(yhi, o.y0 - 1); else ylo = std::max(ylo, o.y1 + 1);   // outside because of y only
                }
                if (!moved) break;
            }
            m.l = l0; m.t = t; m.ylo = ylo; m.yhi = yhi; m.res = l; m.epoch = ops.size();
            return l;
        };
        // An op (hi -> lo) moves the pixels labelled hi inside box[hi].  Right after one, no such pixel is left, and
        // new ones can only arrive through a later op whose `lo` is hi.  A seam row that asks for hi -> anything
        // while nothing has flowed into hi since hi's last op therefore changes no pixel (the reference runs the
        // same relabel and finds nothing, contrack.py:759/763): it is not recorded.  This keeps the per-label
        // chains short where a stranded fragment sits on the seam for many rows.
        inflow.assign((size_t)nd + 1, -1);                       // index of the last recorded op with lo == label
        const auto t_loop = std::chrono::steady_clock::now();
        nfold = 0;
        for (int64_t k = 0; k < ncand; k++) {
            const CtkCand &c = cand[k];
            const int32_t y_last = (int32_t)((uint32_t)c.yy >> 16);
            // rows y0..y_last of timestep c.t carry the same pair of fresh labels; visit them in order, skipping the rows
            // for which the previous evaluation provably still holds
            for (int32_t y = c.yy & 0xffff; y <= y_last;) {
                const bool tl = first[(size_t)c.ll] >= 0, tr = first[(size_t)c.lr] >= 0;    // is `hi` of some op
                if (c.ll == c.lr && !tl) break;                                // same label, never relabelled: nothing can differ
                int32_t same_until = y_last;
                const int32_t p0 = tl ? fold(0, c.ll, c.t, y, 0) : c.ll;
                const int32_t p1 = tr ? fold(1, c.lr, c.t, y, nx - 1) : c.lr;
                nfold += (tl ? 1 : 0) + (tr ? 1 : 0);
                if (tl) same_until = std::min(same_until, memo[0].yhi);
                if (tr) same_until = std::min(same_until, memo[1].yhi);
                if (p0 == p1) { y = same_until + 1; continue; }                // nothing happens on these rows
                const bool p0_hi = orig[p0] > orig[p1];                        // the larger FRESH label becomes the smaller (:759/763)
                const int32_t hi = p0_hi ? p0 : p1, lo = p0_hi ? p1 : p0;
                if (last[(size_t)hi] >= 0 && inflow[(size_t)hi] < last[(size_t)hi]) { y = same_until + 1; continue; }   // nothing to move, and
                                                                               // nothing changes until an op is recorded
                const int32_t *b = box + 6 * (int64_t)hi;
                const int32_t idx = (int32_t)ops.size();
                ops.push_back(CtkOp{orig[hi], orig[lo], b[0], b[1], b[2], b[3], b[4], b[5]});
                lo_d.push_back(lo);
                next.push_back(-1);
                if (last[(size_t)hi] >= 0) next[(size_t)last[(size_t)hi]] = idx; else first[(size_t)hi] = idx;
                last[(size_t)hi] = idx;
                inflow[(size_t)lo] = idx;
                y++;                                                           // the next row sees the new op
            }
        }
        loop_ns = (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_loop).count();
    }
};

// ------------------------------------------------------------------------------------------------
// Time shards: 3-D label numbering across shard boundaries (contrack.py:748-751, scipy's raster-order ids).
//
// Every rank labels its own shard (union-find over its components plus the components of the previous shard's last
// timestep, the "halo", as extra nodes with the smallest indices) and ranks its OWN roots 0..nroots-1 in raster order.
// What couples the shards is small: the sets that touch a shard boundary.  Per rank q, in time order:
//   last[c]  for every component c of its last timestep:  -1 filtered out;  2k   its set is rooted at own root number k;
//                                                          2h+1 its set is rooted at halo component h
//   halo[h]  for every halo component h (= component h of rank q-1's last timestep):  -1 filtered out;  else the
//            smallest halo component of its set (own components of the first timestep can join two halo components)
// Halo component h of rank q+1 IS last-timestep component h of rank q: uniting through these identities gives the
// global sets.  The true root of a global set is its own-root node in the earliest rank (smallest k there); every
// other own root in the set is "absorbed": it does not take a number, and the numbers behind it move up by one.
//   label(rank q, own root k) = off[q] + k - #{absorbed a in rank q, a < k} + 1,   off[q] = true roots of ranks < q
// ------------------------------------------------------------------------------------------------
struct BoundaryIn {
    int32_t nlast = 0, nh = 0, nroots = 0;
    const int32_t *last = nullptr, *halo = nullptr;
};
struct BoundaryOut {
    std::vector<int64_t> off;                    // [world+1] labels before rank q; off[world] = number of labels
    std::vector<std::vector<int32_t>> absorbed;  // per rank: absorbed own-root numbers, ascending
    std::vector<std::vector<int32_t>> absorbed_label;   // ... and the label of the set that absorbed them
    std::vector<std::vector<int32_t>> halo_label;       // per rank: label of every halo component (0 = filtered out)
    std::vector<std::vector<int32_t>> last_label;       // per rank: label of every last-timestep component (0 = filtered out)
    std::vector<int32_t> crossing;               // labels of all sets that touch a shard boundary, ascending, unique
};

// returns false if the records contradict each other (a kept halo component whose twin was filtered out, ...)
inline bool boundary_resolve(const std::vector<BoundaryIn> &in, BoundaryOut &out)
{
    const int W = (int)in.size();
    // nodes: per rank [last comps | halo comps]
    std::vector<int64_t> base((size_t)W + 1, 0);
    for (int q = 0; q < W; q++) base[(size_t)q + 1] = base[(size_t)q] + in[(size_t)q].nlast + in[(size_t)q].nh;
    const int64_t N = base[(size_t)W];
    std::vector<int64_t> par((size_t)N);
    for (int64_t i = 0; i < N; i++) par[(size_t)i] = i;
    auto find = [&](int64_t i) { while (par[(size_t)i] != i) { par[(size_t)i] = par[(size_t)par[(size_t)i]]; i = par[(size_t)i]; } return i; };
    auto unite = [&](int64_t a, int64_t b) { a = find(a); b = find(b); if (a != b) par[(size_t)std::max(a, b)] = std::min(a, b); };
    auto nodeL = [&](int q, int c) { return base[(size_t)q] + c; };
    auto nodeH = [&](int q, int h) { return base[(size_t)q] + in[(size_t)q].nlast + h; };
    std::vector<std::pair<int32_t, int32_t>> byroot;             // (own root k, last comp) of one rank
    for (int q = 0; q < W; q++) {
        const BoundaryIn &b = in[(size_t)q];
        if (q == 0 && b.nh != 0) return false;
        if (q > 0 && b.nh != in[(size_t)q - 1].nlast) return false;
        for (int h = 0; h < b.nh; h++) {
            const int32_t r = b.halo[h];
            if (r < 0) continue;
            if (r > h || b.halo[r] != r) return false;
            unite(nodeH(q, h), nodeH(q, r));
            if (in[(size_t)q - 1].last[h] < 0) return false;      // kept here, filtered out there
            unite(nodeH(q, h), nodeL(q - 1, h));
        }
        byroot.clear();
        for (int c = 0; c < b.nlast; c++) {
            const int32_t v = b.last[c];
            if (v < 0) continue;
            if (v & 1) {
                const int32_t h = v >> 1;
                if (h >= b.nh || b.halo[h] < 0) return false;
                unite(nodeL(q, c), nodeH(q, h));
            } else {
                if ((v >> 1) >= b.nroots) return false;
                byroot.emplace_back(v >> 1, c);
            }
        }
        std::sort(byroot.begin(), byroot.end());
        for (size_t i = 1; i < byroot.size(); i++)
            if (byroot[i].first == byroot[i - 1].first) unite(nodeL(q, byroot[i].second), nodeL(q, byroot[i - 1].second));
    }
    // true root of every global set: smallest (rank, own root) among its own-root nodes
    std::vector<int64_t> best((size_t)N, INT64_MAX);             // at set representatives: q << 32 | k
    for (int q = 0; q < W; q++) {
        const BoundaryIn &b = in[(size_t)q];
        for (int c = 0; c < b.nlast; c++) {
            const int32_t v = b.last[c];
            if (v < 0 || (v & 1)) continue;
            const int64_t s = find(nodeL(q, c));
            best[(size_t)s] = std::min(best[(size_t)s], ((int64_t)q << 32) | (int64_t)(v >> 1));
        }
    }
    out.absorbed.assign((size_t)W, {});
    out.absorbed_label.assign((size_t)W, {});
    for (int q = 0; q < W; q++) {
        const BoundaryIn &b = in[(size_t)q];
        std::vector<int32_t> &A = out.absorbed[(size_t)q];
        for (int c = 0; c < b.nlast; c++) {
            const int32_t v = b.last[c];
            if (v < 0 || (v & 1)) continue;
            const int64_t s = find(nodeL(q, c));
            if (best[(size_t)s] == INT64_MAX) return false;
            if (best[(size_t)s] != (((int64_t)q << 32) | (int64_t)(v >> 1))) A.push_back(v >> 1);
        }
        std::sort(A.begin(), A.end());
        A.erase(std::unique(A.begin(), A.end()), A.end());
    }
    out.off.assign((size_t)W + 1, 0);
    for (int q = 0; q < W; q++) out.off[(size_t)q + 1] = out.off[(size_t)q] + in[(size_t)q].nroots - (int64_t)out.absorbed[(size_t)q].size();
    auto label_of_set = [&](int64_t s) -> int64_t {
        const int64_t bq = best[(size_t)s];
        if (bq == INT64_MAX) return -1;
        const int q = (int)(bq >> 32);
        const int32_t k = (int32_t)(bq & 0xffffffff);
        const std::vector<int32_t> &A = out.absorbed[(size_t)q];
        const int64_t before = std::lower_bound(A.begin(), A.end(), k) - A.begin();
        return out.off[(size_t)q] + k - before + 1;
    };
    out.halo_label.assign((size_t)W, {});
    out.last_label.assign((size_t)W, {});
    out.crossing.clear();
    for (int q = 0; q < W; q++) {
        const BoundaryIn &b = in[(size_t)q];
        out.halo_label[(size_t)q].assign((size_t)b.nh, 0);
        out.last_label[(size_t)q].assign((size_t)b.nlast, 0);
        for (int h = 0; h < b.nh; h++) {
            if (b.halo[h] < 0) continue;
            const int64_t l = label_of_set(find(nodeH(q, h)));
            if (l <= 0 || l > INT32_MAX) return false;
            out.halo_label[(size_t)q][(size_t)h] = (int32_t)l;
            out.crossing.push_back((int32_t)l);
        }
        for (int c = 0; c < b.nlast; c++) {
            if (b.last[c] < 0) continue;
            const int64_t l = label_of_set(find(nodeL(q, c)));
            if (l <= 0 || l > INT32_MAX) return false;
            out.last_label[(size_t)q][(size_t)c] = (int32_t)l;
            if (q + 1 < W) out.crossing.push_back((int32_t)l);         // (nothing follows the last rank's last timestep)
        }
        const std::vector<int32_t> &A = out.absorbed[(size_t)q];
        out.absorbed_label[(size_t)q].assign(A.size(), 0);
        for (int c = 0; c < b.nlast; c++) {
            const int32_t v = b.last[c];
            if (v < 0 || (v & 1)) continue;
            const auto it = std::lower_bound(A.begin(), A.end(), v >> 1);
            if (it != A.end() && *it == (v >> 1)) out.absorbed_label[(size_t)q][(size_t)(it - A.begin())] = out.last_label[(size_t)q][(size_t)c];
        }
    }
    std::sort(out.crossing.begin(), out.crossing.end());
    out.crossing.erase(std::unique(out.crossing.begin(), out.crossing.end()), out.crossing.end());
    return true;
}
