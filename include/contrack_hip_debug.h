/* contrack_hip_debug.h -- test hooks, staged-parity accessors, experiment knobs and measurement support of libcontrack_hip.so.
 *
 * NOT part of the drop-in boundary (include/contrack_hip.h is): nothing a caller of the reference's run_contrack / run_lifecycle /
 * calc_anom replacement needs is declared here.  The parity tests (tests/), the measurement scripts (bench.py, tools/) and the
 * probes bind these through ctypes; they are exported by the same shared object and may change between rounds.
 *   ctk_debug_*                      test hooks (forced overflows, failures, stalls), staged outputs (mask, 2-D labels), GPU-free
 *                                    pieces of the resolver, placement / launch-shape experiments
 *   ctk_set_filter_round / _fused_pass / _device_resolve     which form of the pass runs (the defaults are the product)
 *   ctk_synth_fill*, ctk_checksum_i32_dev, ctk_check_flag_dev   synthetic slabs and size-independent result checks on the device
 *   ctk_expand_runs_host             the decoder of the run-table result transfer on its own (host only)
 */
#ifndef CONTRACK_HIP_DEBUG_H
#define CONTRACK_HIP_DEBUG_H

#include "contrack_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test hook: the next ctk_track_sharded_* call on this handle fails at stage 1..6 (between two collectives), once */
int  ctk_debug_fail_at(ctk_handle *h, int stage);

int ctk_debug_mask(ctk_handle *h, uint8_t *mask /* (T,ny,nx) 0/1 */);

/* 2-D labels exactly as scipy numbers them at contrack.py:684 (before_seam=1) or after the seam merge
 * of contrack.py:691-698 (before_seam=0): ids are global over time, 1-based, raster order.        */
int ctk_debug_label2d(ctk_handle *h, int before_seam, int32_t *lab /* (T,ny,nx) */);

/* test hook: the NEXT call behaves as if the pair table held only `records` entries (exercises regrowth) */
int ctk_debug_set_pair_capacity(ctk_handle *h, uint32_t records);

/* test hook, GPU-free: numpy's float64 add.reduce order (what np.sum(weight_grid[...]) computes, contrack.py:717-719), used to
 * re-evaluate overlap decisions whose exactly accumulated area sums had to be rounded */
double ctk_debug_np_sum(const double *a, size_t n);

/* test hook, GPU-free: scipy's 3-D ids across time-shard boundaries from the per-rank boundary records of ctk_track_sharded_*
 * (contrack_amd/csrc/ctk_seam.h explains the records); flat arrays, rank after rank */
int ctk_debug_boundary_resolve(int world, const int32_t *nlast, const int32_t *nh, const int32_t *nroots, const int32_t *last_flat,
                               const int32_t *halo_flat, int64_t *off /* [world+1] */, int32_t *last_label_flat, int32_t *halo_label_flat,
                               int32_t *n_absorbed /* [world] */);

/* test hook: labels / operations of one seam cluster the device seam driver accepts (0 = its limits, 64 each): clusters beyond
 * send the pass to the synchronous path with the host driver, and the grid stays there */
int ctk_debug_set_seam_caps(ctk_handle *h, int labels, int ops);

/* test hook: cap the device-written mailbox of the resolver hand-off (0 = no cap), so that the explicit-copy path runs; `labels`
 * also caps the list of shared ids the time-shard path's extent exchange keeps in LDS (longer lists: its one-workgroup form) */
int ctk_debug_set_mailbox(ctk_handle *h, uint32_t cand_records, uint32_t labels);

/* Test hook for the BOUNDED inter-workgroup waits of the one-launch ("systolic") filter pass: a wave that has waited longer than
 * limit_ms (0 = the default, 200 ms) for its predecessor gives up, the pass is marked invalid and the call repeats the resolution
 * with one launch per filter pass (CTK_S_HOST_REASON bit 3; the handle keeps doing so).  stall_mode 1: the first workgroup of the
 * chain arrives late (limit / 4); 2: it never publishes; 0: normal.  Also clears the handle's "no one-launch pass" state. */
int ctk_debug_set_spin(ctk_handle *h, double limit_ms, int stall_mode);

/* experiments: which chunk of the slab the workgroups of the two streaming kernels take (0 = in launch order, 1 = one contiguous eighth
 * per XCD, k > 1 = tiles of k chunks per XCD); -1 = the default */
int ctk_debug_set_xcd(ctk_handle *h, int thr_mode, int rel_mode);

/* experiments: threads (0 = default, 256 / 512 / 1024) and rows (0 = default) per workgroup of the write kernel k_relabel_v5 */
int ctk_debug_set_relabel(ctk_handle *h, int threads, int rows);

/* measurement support (bench.py, next to the roofline): best-of-`reps` time in ms of a PLAIN stream over a 16-byte-aligned device buffer --
 * mode 1: 16-byte non-temporal stores of zeros, 32 KB per workgroup in launch order; mode 2: the same with one contiguous eighth of the
 * buffer per XCD (the faster store stream on every size timed; the better of the two bounds the write kernel); mode 0: 16-byte
 * non-temporal loads (what bounds the threshold kernel).  Overwrites the buffer in modes 1 and 2. */
int ctk_debug_stream_ceiling(ctk_handle *h, void *p_dev, size_t nbytes, int mode, int reps, double *best_ms);

/* measurement support: the write kernel launched `reps` times on the finished tables of the last one-call pass into flag_dev (the same flags
 * again), every launch timed by events -> ms[reps].  variant: 0 k_relabel_v5, 1 the same without its SGPR limit; xcd: chunk -> XCD order (0 launch order, 1 one eighth of the launch per XCD, 16: tiles of 16; -1: the handle's). */
int ctk_debug_time_relabel(ctk_handle *h, int32_t *flag_dev, int persistence, int variant, int xcd, int reps, double *ms);

/* experiments: threads per workgroup (0 = default, 64 / 128 / 256) of the one-workgroup-per-timestep kernels k_extent, k_run_values,
 * k_compact_init of the one-call pass; extent = 1024: the sixteen-timesteps-per-workgroup form k_extent_blk whatever the shard's length
 * (test hook: by default it serves shards of more than 2048 timesteps on grids narrower than 1024) */
int ctk_debug_set_small_threads(ctk_handle *h, int extent, int run_values, int compact_init);

/* placement experiment: the bit mask `off` bytes (a multiple of 256, up to 64 MB) into a larger allocation from the next call on; -1: plain */
int ctk_debug_set_mask_offset(ctk_handle *h, int64_t off);

/* placement experiment: frees one work-space buffer (0 mask, 1 wstart, 2 rowstart, 3 chunk_vals, 4 run_val, 5 run_base); the next call allocates it anew */
int ctk_debug_drop_buffer(ctk_handle *h, int which);

/* filter passes launched per round before convergence is checked on the host (default 10, 1..32)   */
int ctk_set_filter_round(ctk_handle *h, int passes);

/* 1 (default; CTK_ASYNC=0 in the environment turns it off): the one-call entries run the whole pass without a host hand-off (device
 * seam driver, one synchronisation at the end, validated from a device-written block of scalars; CTK_S_FUSED) and repeat the
 * resolution on the synchronous path below only if the validation says so; 0: always the synchronous path (host seam driver) */
int ctk_set_fused_pass(ctk_handle *h, int enable);

/* 1 (default): ctk_track_* resolve the tables on the device; 0: download + ctk_resolve on the host */
int ctk_set_device_resolve(ctk_handle *h, int enable);

/* deterministic on-device synthetic slab for throughput runs (bench only; not part of the path) */
int ctk_synth_fill(ctk_handle *h, float *anom_dev, int64_t T, int ny, int nx, uint64_t seed);

/* the window [t0, t0 + T) of the slab that ctk_synth_fill(seed) generates for any T >= t0 + T: time shards of one synthetic slab */
int ctk_synth_fill_window(ctk_handle *h, float *anom_dev, int64_t t0, int64_t T, int ny, int nx, uint64_t seed);

/* position-weighted checksum of an int32 device array (bench.py's in-run parity check of the time-shard path: the shards'
 * checksums against those of the one-call result).  out[0] = sum over i of (uint32)p[i] * (((index0 + i) * 0x9E3779B97F4A7C15) | 1)
 * mod 2^64, out[1] = number of nonzero elements.  Equal for two arrays iff (up to 2^-64 collisions) the arrays are equal. */
int ctk_checksum_i32_dev(ctk_handle *h, const int32_t *p_dev, int64_t n, int64_t index0, uint64_t *out2);

/* Size-independent properties of a result that lives in device memory (slabs no host holds: BASELINE.json configs[2], 60.6 GB each
 * way), for the parity tests: out6 = { pixels with flag != 0 where (double)anom <op> thr[t] is false (contrack.py:665 -- evaluated
 * in float64, independently of the kernels' float32 form), pixels whose id lies outside [1, max_id], nonzero pixels, distinct ids,
 * ids whose time extent stop - start is below `persistence` (contrack.py:765-772: none may survive), largest id }. */
int ctk_check_flag_dev(ctk_handle *h, const float *anom_dev, const int32_t *flag_dev, int64_t T, int ny, int nx, const double *thr, int cmp_op,
                       int persistence, int64_t max_id, uint64_t *out6);

/* The decoder of that transfer on its own, on tables in host memory (no device call; for tests): mask u64 [T][ny][ceil(nx/64)],
 * rowstart u32 [T][ny] (first run of the row, relative to its time step), run_base u32 [T + 1], run_val i32 [runs] -> flag
 * [T][ny][nx]; *wrote_background: a zero was written; *complex_runs: a negative run value was met (its pixels are not decoded). */
int ctk_expand_runs_host(const uint64_t *mask, const uint32_t *rowstart, const uint32_t *run_base, const int32_t *run_val, int64_t T, int ny, int nx,
                         int32_t *flag, int *wrote_background, int *complex_runs);

#ifdef __cplusplus
}
#endif

#endif /* CONTRACK_HIP_DEBUG_H */
