/*
 * contrack_hip.h -- C ABI of libcontrack_hip.so, the MI355X (gfx950) implementation of ConTrack's
 * run_contrack hot path.
 *
 * What this replaces.  The reference (steidani/ConTrack v0.4.1) has NO FFI/plugin interface: its hot
 * path is the Python method contrack.run_contrack (contrack/contrack.py:583-796), which calls
 * scipy.ndimage.label (:684, :748), scipy.ndimage.find_objects (:708, :753, :766) and numpy reductions
 * (:717-719) on a (time, lat, lon) array.  The entry points below are what a ctypes binding inside
 * run_contrack binds instead of those calls (INTEGRATION.md shows the stub).  Plain pointers and
 * sizes only; all buffers are caller-owned and borrowed for the duration of the call; the library owns
 * its device workspace (inside the opaque handle).  Every function returns 0 on success or a negative
 * CTK_E_* code; ctk_last_error() returns a thread-local message.  Nothing throws or exits.
 *
 * Conventions shared by all entry points
 *   anom      C-contiguous float32 (T, ny, nx) -- the slab after contrack.py:677-681 put it in
 *             (time, lat, lon) order.
 *   thr       T doubles; the compare evaluated is (double)anom[t,y,x] <op> thr[t].  The caller rounds a
 *             Python-number threshold to float32 first (that is what contrack.py:665 compares against
 *             for a float32 array); a float64 threshold vector is passed unrounded (contrack.py:650).
 *   cmp_op    0 '>='/'ge', 1 '<='/'le', 2 '>'/'gt', 3 '<'/'lt'     (contrack.py:649-656, :664-671)
 *   wrow      ny float32 row weights, computed by the host exactly as contrack.py:703-704 does.
 *   overlap, persistence, twosided                                   (contrack.py:587-589)
 *   flag      int32 (T, ny, nx): the ids of contrack.py:776-791, identical to the reference's
 *             (identity permutation), 0 = background.
 *   n_tracked len(np.unique(flag)) - 1                               (contrack.py:793)
 *
 * This header is the drop-in boundary: create / destroy / errors, the track entries (host arrays, device-resident, time-sharded,
 * streaming, resident anomaly slab), the staged time-shard protocol, the communicator, run_lifecycle, calc_anom / percentile, the
 * thin memory helpers a ctypes host needs, timing and workload statistics.  Test hooks (ctk_debug_*), experiment knobs and the
 * measurement support of the tests and of bench.py are declared in contrack_hip_debug.h.
 */
#ifndef CONTRACK_HIP_H
#define CONTRACK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTK_OK            0
#define CTK_E_INVALID    -1   /* bad argument (shape, cmp_op, NULL, non-finite weight ...)      */
#define CTK_E_NODEVICE   -2   /* no usable HIP device / HIP runtime error                        */
#define CTK_E_NOMEM      -3   /* host or device allocation failed                                */
#define CTK_E_RANGE      -4   /* weight dynamic range or problem size beyond what the kernels carry */
#define CTK_E_INTERNAL   -5
#define CTK_E_STATE      -6   /* staged calls out of order                                       */
#define CTK_E_COMM       -7   /* time-shard path: another rank gave up, died or did not arrive in time (the message names it);
                                 the communicator is retired -- create a new one                      */

typedef struct ctk_handle ctk_handle;

/* ---- library / device ------------------------------------------------------------------------ */
int         ctk_version(void);
const char *ctk_last_error(void);
int         ctk_device_count(void);                       /* <0 on HIP error, 0 if no GPU          */
int         ctk_create(ctk_handle **h, int device);       /* binds the handle to one GPU + stream  */
void        ctk_destroy(ctk_handle *h);

/* ---- whole path, one call (replaces contrack.py:646-772 on the (time,lat,lon) slab) ----------- */
/* host buffers in, host buffers out (H2D + kernels + D2H) */
int ctk_track_f32(ctk_handle *h, const float *anom, int64_t T, int ny, int nx, const double *thr,
                  int cmp_op, const float *wrow, double overlap, int persistence, int twosided,
                  int32_t *flag, int64_t *n_tracked);
/* device buffers in, device buffers out (anom_dev / flag_dev are HIP device pointers) */
int ctk_track_f32_dev(ctk_handle *h, const float *anom_dev, int64_t T, int ny, int nx,
                      const double *thr, int cmp_op, const float *wrow, double overlap,
                      int persistence, int twosided, int32_t *flag_dev, int64_t *n_tracked);

/* The host-array entries keep device copies of the slab and of the result in the handle between calls (grow-only, so that
 * repeated calls do not reallocate).  ctk_release_io frees them (and the copy lanes): call it after a one-off large slab. */
int ctk_release_io(ctk_handle *h);

/* float64 slabs (xarray often hands float64 anomalies): the compare is evaluated in float64, exactly as
 * numpy does for a float64 array at contrack.py:665; everything downstream is identical.            */
int ctk_track_f64(ctk_handle *h, const double *anom, int64_t T, int ny, int nx, const double *thr,
                  int cmp_op, const float *wrow, double overlap, int persistence, int twosided,
                  int32_t *flag, int64_t *n_tracked);
int ctk_track_f64_dev(ctk_handle *h, const double *anom_dev, int64_t T, int ny, int nx,
                      const double *thr, int cmp_op, const float *wrow, double overlap,
                      int persistence, int twosided, int32_t *flag_dev, int64_t *n_tracked);

/* ---- staged path: the stages of the path one by one, with the component tables resolved on the HOST (ctk_resolve).
 *      Kept for stage-level parity tests and as the table-level specification of the resolver; the multi-GPU product path
 *      is ctk_track_sharded_* above.  Each shard owns timesteps [t_begin, t_begin+T). ------ */
/* stage 1: threshold -> bit mask -> 2-D labelling with longitude wrap (contrack.py:646-698) + per
 *          component areas (contrack.py:717).  has_prev != 0 means a previous shard exists and
 *          its last timestep will be imported before stage 2.                                    */
int ctk_shard_label2d(ctk_handle *h, const float *anom_dev, int64_t T, int ny, int nx,
                      const double *thr, int cmp_op, const float *wrow, int has_prev);
int ctk_shard_label2d_f64(ctk_handle *h, const double *anom_dev, int64_t T, int ny, int nx,
                          const double *thr, int cmp_op, const float *wrow, int has_prev);
/* halo: the labelled LAST timestep of this shard, as an opaque device blob for the next rank
 *       (bit mask + run->component ids; the compressed form of the one-timestep label map).      */
int ctk_shard_halo_size(ctk_handle *h, size_t *max_bytes);           /* upper bound, same on all ranks */
int ctk_shard_halo_export(ctk_handle *h, void **blob_dev, size_t *nbytes);
int ctk_shard_halo_import(ctk_handle *h, const void *blob_dev, size_t nbytes);
/* stage 2: label co-occurrence histogram between consecutive timesteps (the overlap areas of
 *          contrack.py:718-719 and the temporal links of contrack.py:748-750).                   */
int ctk_shard_overlap(ctk_handle *h);
/* tables: serialised component / pair / seam tables of this shard (host memory owned by the
 *         handle, valid until the next staged call on it).                                       */
int ctk_shard_tables(ctk_handle *h, const void **blob, size_t *nbytes);

/* resolve: host-side, GPU-free.  Takes the table blobs of ALL shards in time order and evaluates the
 *          sequential parts of the reference on component tables: overlap filter recurrence
 *          (contrack.py:706-742), 3-D labelling ids (contrack.py:748-751), bbox-confined seam merges
 *          (contrack.py:753-763).  Returns an opaque result (free with ctk_result_free).          */
typedef struct ctk_result ctk_result;
int  ctk_resolve(const void *const *blobs, const size_t *nbytes, int nshards, double overlap,
                 int twosided, ctk_result **out);
void ctk_result_free(ctk_result *r);
int  ctk_result_info(const ctk_result *r, int64_t *n_labels, int64_t *n_ops, int64_t *n_complex,
                     int64_t *n_ambiguous, int64_t *n_components);

/* read-only views into a result (valid until ctk_result_free): comp_label[ncomps] in (shard, t, c)
 * order -- 0 = removed by the overlap filter, L > 0 = final id of every pixel of the component, -L =
 * the component must be folded pixel by pixel starting from 3-D label L; ops[nops] = the bbox-confined
 * relabel operations in execution order, 8 int32 each: hi, lo, t0, t1, y0, y1, x0, x1 (inclusive). */
int  ctk_result_arrays(const ctk_result *r, const int32_t **comp_label, int64_t *ncomps, const void **ops,
                       int64_t *nops, const int64_t **shard_comp_off, const int64_t **shard_t_off);
int  ctk_result_nshards(const ctk_result *r);          /* length-1 of the two offset arrays above */
/* exact integer limbs of the float32 row weights: w[y] = (wlo[y] + whi[y] * 2^limb_bits) * 2^-wshift.  limb_bits is 31
 * unless the weights span more than 62 bits (float64 latitudes with exact poles: cos(pi/2) = 6e-17 next to 1); then it
 * is ceil(span / 2), which needs limb_bits + ceil(log2 npix) <= 62 with npix = ny * nx, the most pixels any area sum
 * covers (CTK_E_RANGE otherwise). */
int  ctk_weights_to_limbs(const float *wrow, int ny, int64_t npix, int64_t *wlo, int64_t *whi, int32_t *wshift, int32_t *limb_bits);

/* stage 3: apply the result to this shard (index `shard` of the blobs given to ctk_resolve):
 *          per-label time extents (persistence, contrack.py:765-772), then the relabel pass that
 *          writes flag.  ext_dev: device int32 [2*(n_labels+1)] (min t | max t); when ranks > 1 the
 *          caller all-reduces the first half with MIN and the second with MAX between _extents and
 *          _write.                                                                                */
int ctk_shard_extents(ctk_handle *h, const ctk_result *r, int shard, int64_t t_begin,
                      int32_t **ext_dev, int64_t *n_labels);
int ctk_shard_write(ctk_handle *h, int persistence, int32_t *flag_dev, int64_t *n_alive_local,
                    int *wrote_background);
/* number of labels that survive persistence (identical on all ranks after the all-reduce)        */
int ctk_shard_count_tracked(ctk_handle *h, int64_t *n_alive);

/* ---- time-sharded path, one rank per GPU --------------------------------------------------------------------
 * Rank r owns the timesteps [t_begin, t_begin + T_local) of a slab of T_total steps (time order = rank order; every rank
 * owns at least one step).  ctk_track_sharded_* is ctk_track_*_dev on that shard: bulk data stays local, the ranks
 * exchange a one-timestep label-map halo with their neighbours and a few hundred boundary records through the
 * communicator (contrack_amd/csrc/ctk_sharded.hip).  flag / n_tracked are identical to what one call on the whole slab
 * returns; n_tracked is the same on every rank.  All ranks must make the call (it contains collectives).
 *
 * Communicator: RCCL over xGMI (one process per GPU; librccl.so is loaded on first use), an in-process group (several
 * handles driven by one host thread each), or a shared-memory transport for several processes on one node without RCCL. */
typedef struct ctk_comm ctk_comm;
typedef struct ctk_comm_group ctk_comm_group;
#define CTK_COMM_ID_BYTES 128
int  ctk_comm_unique_id(void *id /* CTK_COMM_ID_BYTES, made on rank 0 (ncclGetUniqueId), handed to the other ranks by the launcher */);
int  ctk_comm_init_rccl(ctk_handle *h, const void *id, int rank, int world, ctk_comm **out);
int  ctk_comm_group_create(int world, ctk_comm_group **out);
void ctk_comm_group_destroy(ctk_comm_group *g);
int  ctk_comm_init_local(ctk_handle *h, ctk_comm_group *g, int rank, ctk_comm **out);
int  ctk_comm_init_shm(ctk_handle *h, const char *segment_name, int rank, int world, ctk_comm **out);
void ctk_comm_destroy(ctk_comm *c);
int  ctk_comm_rank(const ctk_comm *c);
int  ctk_comm_world(const ctk_comm *c);
int  ctk_comm_barrier(ctk_comm *c);
int  ctk_comm_allgather_host(ctk_comm *c, const void *send, void *recv /* world * nbytes */, size_t nbytes /* <= 4096 */);
int  ctk_comm_ops(const ctk_comm *c, int64_t *shifts, int64_t *allgathers);      /* operations issued so far */
/* the librccl file the RCCL transport took its symbols from ("" before the first ctk_comm_init_rccl).  Order of choice: CTK_RCCL_LIB; a
 * copy already mapped into the process (one RCCL per process); $ROCM_PATH/lib, /opt/rocm/lib; the loader path. */
const char *ctk_comm_rccl_library(void);
/* Failure behaviour.  No rank is left waiting for one that gave up or died: the ranks of an rccl / shm communicator share a small
 * control segment in POSIX shared memory (single node) with a failure word and every rank's pid; every wait of the path polls it,
 * checks that the peers' processes still exist, and carries a deadline (seconds; default 120 or CTK_COMM_TIMEOUT_S).  A call that
 * fails on one rank makes every other rank's call return CTK_E_COMM (the message names the rank and its code) -- within
 * milliseconds, not at the deadline; the RCCL communicator is aborted (ncclCommAbort) so that kernels in flight return.  Errors
 * that every rank derives from the same gathered data (table overflow, id range) are returned by all ranks with their own code
 * and leave the communicator usable. */
int  ctk_comm_set_timeout(ctk_comm *c, double seconds);
int  ctk_comm_failed(const ctk_comm *c, int *code /* 0: fine */, int *rank);
int  ctk_comm_abort_rank(ctk_comm *c, int code /* < 0 */);                      /* this rank gives up (e.g. its caller failed elsewhere) */
int  ctk_track_sharded_f32_dev(ctk_handle *h, ctk_comm *c, const float *anom_dev, int64_t T_local, int64_t t_begin, int64_t T_total,
                               int ny, int nx, const double *thr /* T_local */, int cmp_op, const float *wrow, double overlap,
                               int persistence, int twosided, int32_t *flag_dev, int64_t *n_tracked);
int  ctk_track_sharded_f64_dev(ctk_handle *h, ctk_comm *c, const double *anom_dev, int64_t T_local, int64_t t_begin, int64_t T_total,
                               int ny, int nx, const double *thr, int cmp_op, const float *wrow, double overlap,
                               int persistence, int twosided, int32_t *flag_dev, int64_t *n_tracked);

/* ---- next row N4: streaming entries (xr.open_dataset contrack.py:176 -> run_contrack -> to_netcdf README.rst:154) ------------
 * ctk_track_f32 / _f64 for slabs that should not (or cannot) sit in HBM twice: the slab passes through two chunk-sized device
 * buffers per direction; only the bit mask (1/32 of the slab) and the run / component tables stay resident between the
 * two pixel passes.
 *     chunk k+1 arrives (reader callback / host array -> H2D)   ||   k_threshold(chunk k) -> mask
 *     labelling, overlap filter, 3-D ids, seam merges, persistence on mask and tables          (nothing of the slab needed)
 *     k_relabel(chunk k+1)                                      ||   D2H(chunk k) -> writer callback / host array
 * ctk_track_stream_f32 / _f64: source and sink are host arrays (any size the host holds; device footprint 4 chunks + slab/32).
 * ctk_track_stream_cb: source and sink are callbacks, e.g. a netCDF variable read / written slice by slice.  The reader fills
 * `dst` (pinned host memory owned by the library, nt * ny * nx elements of elem_bytes) with timesteps [t0, t0 + nt) and returns 0;
 * the writer receives the int32 flag values of [t0, t0 + nt) in pinned memory valid during the call and returns 0; a nonzero
 * return aborts the call with CTK_E_INVALID.  Readers are called in increasing t0, once per chunk -- unless the call meets a
 * decision on a rounding boundary (CTK_S_EXACT_FIXUPS), in which case the input is streamed a second time.  Writers are called
 * in increasing t0, once.  chunk_steps = 0: about 256 MB of input per chunk.  Results: identical to ctk_track_*. */
typedef int (*ctk_read_chunk_fn)(void *user, int64_t t0, int64_t nt, void *dst);
typedef int (*ctk_write_chunk_fn)(void *user, int64_t t0, int64_t nt, const int32_t *src);
int ctk_track_stream_f32(ctk_handle *h, const float *anom, int64_t T, int ny, int nx, const double *thr, int cmp_op, const float *wrow,
                         double overlap, int persistence, int twosided, int32_t *flag, int64_t *n_tracked, int64_t chunk_steps);
int ctk_track_stream_f64(ctk_handle *h, const double *anom, int64_t T, int ny, int nx, const double *thr, int cmp_op, const float *wrow,
                         double overlap, int persistence, int twosided, int32_t *flag, int64_t *n_tracked, int64_t chunk_steps);
int ctk_track_stream_cb(ctk_handle *h, int elem_bytes /* 4: float32, 8: float64 */, int64_t T, int ny, int nx, ctk_read_chunk_fn reader,
                        void *reader_user, const double *thr, int cmp_op, const float *wrow, double overlap, int persistence, int twosided,
                        ctk_write_chunk_fn writer, void *writer_user, int64_t *n_tracked, int64_t chunk_steps);
/* times of the last streaming call: {reader callbacks, writer callbacks, input phase, output phase} in ms */
int ctk_stream_times(ctk_handle *h, double *ms4);

/* ---- timing (HIP events on the handle's stream) ----------------------------------------------- */
#define CTK_K_THRESHOLD 0
#define CTK_K_SCAN      1
#define CTK_K_LABEL2D   2
#define CTK_K_OVERLAP   3
#define CTK_K_EXTENT    4
#define CTK_K_RUNLABEL  5
#define CTK_K_RELABEL   6
#define CTK_K_RESOLVE   7      /* device resolver: filter passes, 3-D union-find, ids, boxes */
#define CTK_K_RESOLVE2  8      /* device resolver: final id per component                     */
#define CTK_K_COUNT     9
#define CTK_T_HOST_RESOLVE 10  /* host: sequential seam driver (or ctk_resolve on the host path) */
#define CTK_T_D2H          11  /* downloads                                            */
#define CTK_T_H2D          12  /* uploads                                              */
#define CTK_T_TOTAL        13
#define CTK_NTIMERS        14
/* level 0: no HIP events; 1: events around ONE of the two pixel-streaming kernels in every second pass (k_threshold in passes
 * 0, 4, 8 ... of the handle, k_relabel in passes 2, 6, 10 ...; whatever was not timed reads 0 for that pass) -- what bench.py keeps
 * on during its timed region; 2: events around every kernel group (each record is a ~5 us command on the stream) */
int ctk_set_timing(ctk_handle *h, int level);
/* event times summed over the calls since the last reset, and the number of calls that measured each group ([CTK_NTIMERS] each;
 * either may be NULL): what a loop that times many passes reads ONCE at its end instead of ctk_get_timings after every call */
int ctk_get_timing_sums(ctk_handle *h, double *sums, int64_t *counts, int reset);
/* workload statistics of the last call (for bench reports: cost depends on them) */
#define CTK_S_RUNS          0   /* foreground runs in the shard                         */
#define CTK_S_MAX_RUNS_STEP 1   /* most runs in one timestep                            */
#define CTK_S_COMPONENTS    2   /* 2-D components (no wrap)                             */
#define CTK_S_PAIRS         3   /* co-occurrence records                                */
#define CTK_S_SEAM_ROWS     4   /* seam rows handed to the sequential driver            */
#define CTK_S_LABELS        5   /* fresh 3-D labels                                     */
#define CTK_S_OPS           6   /* recorded bbox-confined relabel operations            */
#define CTK_S_FILTER_PASSES 7   /* passes of the overlap-filter iteration that ran      */
#define CTK_S_HOST_PATH     8   /* 1 if the call fell back to the host resolver         */
#define CTK_S_UPAIRS        12  /* co-occurrence records that bypassed the LDS hash table   */
#define CTK_S_PAIR_REGROW   13  /* times the pair table had to be regrown (host path)       */
#define CTK_S_FILTER_ROUNDS 14  /* rounds of filter passes (convergence is checked per round) */
#define CTK_S_EXACT_FIXUPS  16  /* overlap decisions re-evaluated with numpy-order sums on the host (see CTK_S_AMBIGUOUS)  */
#define CTK_S_AMBIGUOUS     15  /* > 0: an overlap decision used an area sum that had to be ROUNDED (components touching a pole row,
                                   whose weight is ~2^-20 of the others) and came out within 8 ulp of the threshold; numpy's pairwise
                                   float64 summation may land on the other side there (only with exact ties: blocky test fields,
                                   overlap = 1.0).  Device path: 0/1 flag; host resolver: the number of such decisions. */
#define CTK_S_HOST_REASON   18  /* why a one-call track left the fused device path: bit 0 the co-occurrence table had to be regrown, bit 1
                                   the overlap filter needed more than 240 passes (both: host resolver, CTK_S_HOST_PATH = 1); 4 = decisions
                                   on rounding boundaries, re-evaluated on the device through the time-shard path with one rank;
                                   bit 3 (8) = a bounded inter-workgroup wait gave up, the resolution was repeated with one launch per
                                   filter pass */
#define CTK_S_SHARED_ROWS   17  /* time-sharded path: seam candidate groups shared between shards (driven on every rank)        */
#define CTK_S_RELABEL_KERNEL 19    /* which write kernel ran: 6 = k_relabel_sparse behind k_flag_zero, 5 = k_relabel_v5, 4 = k_relabel_v4, 0 = generic k_relabel, -1 = none (run transfer) */
#define CTK_S_FUSED         20    /* 1: the one-call pass ran without a host hand-off (device seam driver, one synchronisation) */
#define CTK_S_X4_SPECULATED 21    /* time-sharded path: 1 if the boundary records of the 3-D labelling travelled with the last round of
                                   * the overlap filter's exchange (one all-gather less) */
#define CTK_S_RLE_OUT       22    /* host-array entries: 0 = the result was written by k_relabel and copied densely; n > 0 = it travelled as
                                   * run tables and was expanded on the host, n - 1 blocks of timesteps (those holding complex
                                   * components) went through the write kernel */
#define CTK_S_MASK_TRIES    23    /* allocations of the bit mask that were checked against the slab when it was last (re)allocated (0: not checked); sticky */
#define CTK_S_MASK_RATIO    24    /* 1000 x (threshold kernel on the kept mask / the same kernel without its stores), from that check; sticky */
#define CTK_NSTATS          25    /* what ctk_get_stats writes: FROZEN at 25 (a host built against this header is never overrun by a newer library) */
int ctk_get_stats(ctk_handle *h, int64_t *out /* [CTK_NSTATS] */);
/* statistics added after that are read with an explicit length: writes min(n, CTK_NSTATS_ALL) entries, returns CTK_OK */
#define CTK_S_MASK_CHECK_US 25    /* host time of the last mask placement check, microseconds (bounded: at most one other allocation); sticky */
#define CTK_S_MASK_SPACER_MB 26   /* device memory that check held as a spacer while it ran (freed before the call went on), MB; sticky */
#define CTK_NSTATS_ALL      32
int ctk_get_stats_n(ctk_handle *h, int64_t *out, int n);
int ctk_get_timings(ctk_handle *h, double *ms /* [CTK_NTIMERS] */);

/* ---- thin device-memory helpers so that a ctypes host needs no other HIP binding -------------- */
int ctk_dev_malloc(ctk_handle *h, void **p, size_t nbytes);
int ctk_dev_free(ctk_handle *h, void *p);
int ctk_memcpy_h2d(ctk_handle *h, void *dst_dev, const void *src, size_t nbytes);
int ctk_memcpy_d2h(ctk_handle *h, void *dst, const void *src_dev, size_t nbytes);
int ctk_sync(ctk_handle *h);
/* Result buffers without first-touch page faults for the host-array entries (ctk_track_f32 / _f64 / ctk_track_resident): a `flag`
 * pointer into memory from ctk_host_alloc (pinned, CPU-cacheable) or into a caller array registered with ctk_host_register is
 * recognised and filled with ONE DMA at PCIe rate; an ordinary (pageable, usually fresh) array goes through eight threads draining
 * pinned bounce buffers, bound by the page faults of its first touch.  What replaces np.empty at contrack.py:776-791 when results
 * are produced in a loop (ensemble members): contrack_amd/_native.py recycles such blocks. */
int ctk_host_alloc(ctk_handle *h, void **p, size_t nbytes);
int ctk_host_free(ctk_handle *h, void *p);
/* How the result of the host-array entries crosses PCIe.  1 (default; CTK_RLE_OUT=0 in the environment turns it off): as the
 * pass's own run tables -- the bit mask, the first run of every row and the final value of every foreground run (k_run_values),
 * 1/23 of the dense int32 slab at 2707 x 181 x 360 --, expanded into `flag` by sixteen host threads; the write kernel does not
 * run (blocks of timesteps that hold a "complex" component, whose pixels are folded one by one, still go through it).  Every
 * value is computed on the device; the host only decodes.  0: k_relabel writes the dense slab in HBM and it is copied (the
 * scheme above).  -1: back to the environment's choice.  The device-resident entries always write `flag_dev` densely.
 * CTK_S_RLE_OUT reports what a call did.  (contrack.py:776-791: where the reference materialises `flag`) */
int ctk_set_result_transfer(ctk_handle *h, int mode);
int ctk_host_register(ctk_handle *h, void *p, size_t nbytes);
int ctk_host_unregister(ctk_handle *h, void *p);
void *ctk_stream(ctk_handle *h);                          /* hipStream_t */
int ctk_dev_memset(ctk_handle *h, void *p_dev, int byte, size_t nbytes);

/* ---- next row N1: contrack.run_lifecycle reductions (contrack/contrack.py:798-906) --------------------------
 * One row per (time step, flag id != 0) of an int32 flag slab (time, lat, lon) and a field of the same shape:
 *   area  = np.sum(weight_grid[flag == id])                       contrack.py:874   (exact, rounded once)
 *   swv   = np.sum(weight_grid[...] * field[...])                 contrack.py:875   (float64)
 *   swvy, swvx = the two numerators of ndimage.center_of_mass(field * weight_grid, flag, [id])   contrack.py:892
 *   shift = column that becomes x = 0 when the id touches both seam columns (np.roll by -shift,
 *           contrack.py:880-889; swvx is taken in the rolled frame); -1 if not rolled; -2 if the id occupies a
 *           single column (the reference's argmax of an empty diff raises there)
 * The host finishes with  intensity = swv / area,  com = (swvy / swv, swvx / swv),  int() and the coordinate
 * look-ups (contrack.py:886-895).  ctk_lifecycle_* computes and keeps the rows in the handle (sorted by
 * (label, t) like the reference's frame, contrack.py:906) and returns their number; ctk_lifecycle_rows copies
 * them out.  wrow: float32 row weights, contrack.py:847-848.  Every byte of flag / field is read once (strips of 256 columns x
 * 64 rows per workgroup); a time step with more ids than the tables of that form hold (128, or 4 that cross the seam) is redone
 * by a one-workgroup-per-time-step kernel in passes over residue classes of the ids: no limit.
 * pad: the rows that hold the id, first | last << 16 (internal: bounds the scans of ctk_lifecycle_exact).
 * Host entries with field = NULL take the anomaly slab that ctk_anom_* left resident in HBM (same shape and type). */
typedef struct ctk_life_row {
    int32_t t, label, shift, pad;
    double area, swv, swvy, swvx;
} ctk_life_row;
int ctk_lifecycle_f32_dev(ctk_handle *h, const int32_t *flag_dev, const float *field_dev, int64_t T, int ny, int nx, const float *wrow, int64_t *nrows);
int ctk_lifecycle_f64_dev(ctk_handle *h, const int32_t *flag_dev, const double *field_dev, int64_t T, int ny, int nx, const float *wrow, int64_t *nrows);
int ctk_lifecycle_f32(ctk_handle *h, const int32_t *flag, const float *field, int64_t T, int ny, int nx, const float *wrow, int64_t *nrows);
int ctk_lifecycle_f64(ctk_handle *h, const int32_t *flag, const double *field, int64_t T, int ny, int nx, const float *wrow, int64_t *nrows);
int ctk_lifecycle_rows(ctk_handle *h, ctk_life_row *rows, int64_t cap);
/* The sums above are float64 but not in the reference's summation order; that shows only where a result sits on a rounding
 * boundary (a centre of mass that is an integer up to rounding, a value at the edge of two decimals).  The caller picks those rows
 * (indices into the sorted rows) and gets them in the reference's own orders: np.sum (pairwise) for weight_grid[mask] and
 * weight_grid[mask] * variable[mask] (contrack.py:874-875), np.bincount (sequential, raster order of the rolled plane) inside
 * ndimage.center_of_mass (:886 / :892).  Valid until the next ctk_lifecycle_* call on the handle; for the *_dev entries the
 * caller's device slabs must still be alive. */
typedef struct ctk_life_exact {
    double area, swv, s, sy, sx;       /* np.sum(w), np.sum(w * v), sum p, sum p * y, sum p * x'   (p = v * w) */
} ctk_life_exact;
int ctk_lifecycle_exact(ctk_handle *h, const int64_t *row_idx, int64_t n, ctk_life_exact *out);

/* ---- next rows N2 / N3: the producer of the slab on the device ----------------------------------------------------------
 * ctk_anom_*: contrack.calc_clim / calc_anom (contrack/contrack.py:458-581) on a host slab x (T, ny, nx):
 *   clim_raw[g] = mean over the timesteps t with group[t] == g, NaNs skipped           (groupby(...).mean, :483)
 *   clim[g]     = mean of clim_raw over the centred window of `window` groups; NaN (window beyond the axis, or NaN data) ->
 *                 mean of the last `window` groups of clim_raw                         (rolling(center=True).mean().fillna, :487-489)
 *   anom[t]     = mean over the centred window of `smooth` timesteps of x[j] - clim[group[j]]; NaN where the window leaves the
 *                 axis or holds a NaN                                                  (:568-570)
 *   centred window of w around i: [i - w / 2, i + (w - 1) / 2].
 * group: T ids in [0, ngroups) (the class maps the time coordinate's dayofyear / month ... to them).  clim_in (optional,
 * [ngroups][ny][nx]): use this climatology instead of computing one (the `clim=` argument).  anom_out / clim_out: optional host
 * outputs.  keep_resident != 0: the anomaly slab stays in HBM and ctk_track_resident / ctk_percentile_* (x = NULL) run on it
 * without another host-to-device copy.  Sums in float64, results in the slab's dtype (xarray's dtype rules).  The xarray calls
 * themselves cannot be run in the build container: checked against the numpy restatement oracle/anom_port.py (parity unpinned). */
int ctk_anom_f32(ctk_handle *h, const float *x, int64_t T, int ny, int nx, const int32_t *group, int ngroups, int window, int smooth,
                 const float *clim_in, float *anom_out, float *clim_out, int keep_resident);
int ctk_anom_f64(ctk_handle *h, const double *x, int64_t T, int ny, int nx, const int32_t *group, int ngroups, int window, int smooth,
                 const double *clim_in, double *anom_out, double *clim_out, int keep_resident);
int ctk_resident_anom(ctk_handle *h, int64_t *T, int *ny, int *nx, int *is_f64);       /* T = -1: nothing resident */
/* identity of the resident slab: changes whenever a ctk_anom_* call writes anomalies or ctk_release_io drops them.  Remember it
 * after the call that left YOUR slab resident and use the slab only while it is unchanged. */
int ctk_resident_anom_generation(ctk_handle *h, uint64_t *generation);
/* ctk_track_f32 / _f64 on the resident anomaly slab (flag: host int32 (T, ny, nx)) */
int ctk_track_resident(ctk_handle *h, const double *thr, int cmp_op, const float *wrow, double overlap, int persistence, int twosided,
                       int32_t *flag, int64_t *n_tracked);
/* README.rst:150-151: anom.sel(latitude=rows y0..y1-1).quantile(q, dim='time').mean() -- per grid point the exact q-quantile over
 * time (numpy's linear interpolation, NaNs skipped), then the mean over the band.  x = NULL: the resident anomaly slab. */
int ctk_percentile_f32(ctk_handle *h, const float *x, int64_t T, int ny, int nx, int y0, int y1, double q, double *out);
int ctk_percentile_f64(ctk_handle *h, const double *x, int64_t T, int ny, int nx, int y0, int y1, double q, double *out);

#ifdef __cplusplus
}
#endif
#endif /* CONTRACK_HIP_H */
