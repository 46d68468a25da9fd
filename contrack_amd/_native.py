"""ctypes binding of libcontrack_hip.so (include/contrack_hip.h).  No torch, no other HIP binding.

The library is the product's only compute path: if it is missing or no GPU is visible the calls raise
-- there is no CPU fallback (the CPU restatement lives in oracle/ and is test infrastructure).
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTK_LIB", os.path.join(_HERE, "libcontrack_hip.so"))
_lib = None

CMP_OPS = {">=": 0, "ge": 0, "<=": 1, "le": 1, ">": 2, "gt": 2, "<": 3, "lt": 3}
GORL_ERRMSG = ' Please select from [>, >=, <, >=] for gorl'      # contrack.py:658

TIMER_NAMES = ["k_threshold", "k_scan", "k_label2d", "k_overlap", "k_extent", "k_run_values", "k_relabel",
               "k_resolve", "k_resolve_final", "k_count", "host_seam_driver", "d2h", "h2d", "total"]

EXPORTS = [
    "ctk_version", "ctk_last_error", "ctk_device_count", "ctk_create", "ctk_destroy", "ctk_track_f32",
    "ctk_track_f32_dev", "ctk_track_f64", "ctk_track_f64_dev", "ctk_release_io", "ctk_shard_label2d", "ctk_shard_label2d_f64", "ctk_shard_halo_size", "ctk_shard_halo_export",
    "ctk_shard_halo_import", "ctk_shard_overlap", "ctk_shard_tables", "ctk_resolve", "ctk_result_free",
    "ctk_result_info", "ctk_result_arrays", "ctk_result_nshards", "ctk_weights_to_limbs", "ctk_shard_extents", "ctk_shard_write",
    "ctk_shard_count_tracked", "ctk_debug_mask", "ctk_debug_label2d", "ctk_debug_set_pair_capacity", "ctk_debug_set_mailbox", "ctk_debug_set_seam_caps", "ctk_debug_set_spin", "ctk_debug_set_xcd", "ctk_debug_set_relabel", "ctk_debug_set_small_threads", "ctk_debug_np_sum", "ctk_debug_boundary_resolve", "ctk_set_timing", "ctk_get_timings", "ctk_get_timing_sums", "ctk_set_device_resolve", "ctk_set_fused_pass", "ctk_set_result_transfer", "ctk_debug_set_mask_offset", "ctk_debug_drop_buffer", "ctk_expand_runs_host", "ctk_set_filter_round", "ctk_get_stats", "ctk_get_stats_n", "ctk_debug_stream_ceiling", "ctk_debug_time_relabel",
    "ctk_dev_malloc", "ctk_dev_free", "ctk_host_alloc", "ctk_host_free", "ctk_host_register", "ctk_host_unregister", "ctk_memcpy_h2d", "ctk_memcpy_d2h", "ctk_sync", "ctk_stream",
    "ctk_synth_fill",
    "ctk_comm_unique_id", "ctk_comm_init_rccl", "ctk_comm_group_create", "ctk_comm_group_destroy", "ctk_comm_init_local", "ctk_comm_init_shm",
    "ctk_comm_destroy", "ctk_comm_rank", "ctk_comm_world", "ctk_comm_barrier", "ctk_comm_allgather_host", "ctk_comm_ops",
    "ctk_comm_set_timeout", "ctk_comm_rccl_library", "ctk_comm_failed", "ctk_comm_abort_rank", "ctk_debug_fail_at", "ctk_synth_fill_window", "ctk_checksum_i32_dev", "ctk_dev_memset", "ctk_check_flag_dev",
    "ctk_track_sharded_f32_dev", "ctk_track_sharded_f64_dev",
    "ctk_anom_f32", "ctk_anom_f64", "ctk_resident_anom", "ctk_resident_anom_generation", "ctk_track_resident", "ctk_percentile_f32", "ctk_percentile_f64",
    "ctk_lifecycle_f32", "ctk_lifecycle_f64", "ctk_lifecycle_f32_dev", "ctk_lifecycle_f64_dev", "ctk_lifecycle_rows", "ctk_lifecycle_exact",
    "ctk_track_stream_f32", "ctk_track_stream_f64", "ctk_track_stream_cb", "ctk_stream_times",
]

READ_CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)       # ctk_read_chunk_fn
WRITE_CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p)      # ctk_write_chunk_fn

# ctk_life_row (include/contrack_hip.h)
LIFE_ROW = np.dtype([("t", "<i4"), ("label", "<i4"), ("shift", "<i4"), ("pad", "<i4"),
                     ("area", "<f8"), ("swv", "<f8"), ("swvy", "<f8"), ("swvx", "<f8")])


# ctk_life_exact
LIFE_EXACT = np.dtype([("area", "<f8"), ("swv", "<f8"), ("s", "<f8"), ("sy", "<f8"), ("sx", "<f8")])


class ContrackHipError(RuntimeError):
    pass


class CommError(ContrackHipError):
    """CTK_E_COMM: another rank of the time-shard path gave up, died or did not arrive in time; the communicator is retired"""


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ContrackHipError(
            "libcontrack_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C contrack_amd/csrc`; the HIP path has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    p, i64, i32, dbl, sz = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_size_t
    pp = C.POINTER(C.c_void_p)
    L.ctk_last_error.restype = C.c_char_p
    L.ctk_create.argtypes = [pp, i32]
    L.ctk_destroy.argtypes = [p]
    L.ctk_destroy.restype = None
    track_args = [p, p, i64, i32, i32, p, i32, p, dbl, i32, i32, p, C.POINTER(i64)]
    L.ctk_track_f32.argtypes = track_args
    L.ctk_track_f32_dev.argtypes = track_args
    L.ctk_track_f64.argtypes = track_args
    L.ctk_track_f64_dev.argtypes = track_args
    L.ctk_release_io.argtypes = [p]
    L.ctk_track_stream_f32.argtypes = track_args + [i64]
    L.ctk_track_stream_f64.argtypes = track_args + [i64]
    L.ctk_track_stream_cb.argtypes = [p, i32, i64, i32, i32, READ_CHUNK_FN, p, p, i32, p, dbl, i32, i32, WRITE_CHUNK_FN, p, C.POINTER(i64), i64]
    L.ctk_stream_times.argtypes = [p, C.POINTER(dbl)]
    L.ctk_shard_label2d.argtypes = [p, p, i64, i32, i32, p, i32, p, i32]
    L.ctk_shard_label2d_f64.argtypes = [p, p, i64, i32, i32, p, i32, p, i32]
    L.ctk_shard_halo_size.argtypes = [p, C.POINTER(sz)]
    L.ctk_shard_halo_export.argtypes = [p, pp, C.POINTER(sz)]
    L.ctk_shard_halo_import.argtypes = [p, p, sz]
    L.ctk_shard_overlap.argtypes = [p]
    L.ctk_shard_tables.argtypes = [p, pp, C.POINTER(sz)]
    L.ctk_resolve.argtypes = [pp, C.POINTER(sz), i32, dbl, i32, pp]
    L.ctk_result_free.argtypes = [p]
    L.ctk_result_free.restype = None
    L.ctk_result_info.argtypes = [p] + [C.POINTER(i64)] * 5
    L.ctk_result_arrays.argtypes = [p, pp, C.POINTER(i64), pp, C.POINTER(i64), pp, pp]
    L.ctk_result_nshards.argtypes = [p]
    L.ctk_weights_to_limbs.argtypes = [p, i32, i64, p, p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ctk_shard_extents.argtypes = [p, p, i32, i64, pp, C.POINTER(i64)]
    L.ctk_shard_write.argtypes = [p, i32, p, C.POINTER(i64), C.POINTER(i32)]
    L.ctk_shard_count_tracked.argtypes = [p, C.POINTER(i64)]
    L.ctk_debug_mask.argtypes = [p, p]
    L.ctk_debug_label2d.argtypes = [p, i32, p]
    L.ctk_debug_set_pair_capacity.argtypes = [p, C.c_uint32]
    L.ctk_debug_set_mailbox.argtypes = [p, C.c_uint32, C.c_uint32]
    L.ctk_debug_set_seam_caps.argtypes = [p, i32, i32]
    L.ctk_debug_set_spin.argtypes = [p, dbl, i32]
    L.ctk_debug_set_xcd.argtypes = [p, i32, i32]
    L.ctk_debug_set_relabel.argtypes = [p, i32, i32]
    L.ctk_debug_set_small_threads.argtypes = [p, i32, i32, i32]
    L.ctk_debug_np_sum.argtypes = [p, sz]
    L.ctk_debug_np_sum.restype = dbl
    L.ctk_debug_boundary_resolve.argtypes = [i32, p, p, p, p, p, p, p, p, p]
    L.ctk_set_timing.argtypes = [p, i32]
    L.ctk_get_timings.argtypes = [p, p]
    L.ctk_get_timing_sums.argtypes = [p, p, p, i32]
    L.ctk_set_device_resolve.argtypes = [p, i32]
    L.ctk_set_fused_pass.argtypes = [p, i32]
    L.ctk_set_result_transfer.argtypes = [p, i32]
    L.ctk_debug_set_mask_offset.argtypes = [p, i64]
    L.ctk_debug_drop_buffer.argtypes = [p, i32]
    L.ctk_expand_runs_host.argtypes = [p, p, p, p, i64, i32, i32, p, p, p]
    L.ctk_get_stats.argtypes = [p, p]
    L.ctk_get_stats_n.argtypes = [p, p, i32]
    L.ctk_debug_stream_ceiling.argtypes = [p, p, sz, i32, i32, p]
    L.ctk_debug_time_relabel.argtypes = [p, p, i32, i32, i32, i32, p]
    L.ctk_set_filter_round.argtypes = [p, i32]
    L.ctk_dev_malloc.argtypes = [p, pp, sz]
    L.ctk_dev_free.argtypes = [p, p]
    L.ctk_host_alloc.argtypes = [p, pp, sz]
    L.ctk_host_free.argtypes = [p, p]
    L.ctk_host_register.argtypes = [p, p, sz]
    L.ctk_host_unregister.argtypes = [p, p]
    L.ctk_memcpy_h2d.argtypes = [p, p, p, sz]
    L.ctk_memcpy_d2h.argtypes = [p, p, p, sz]
    L.ctk_sync.argtypes = [p]
    L.ctk_stream.argtypes = [p]
    L.ctk_stream.restype = p
    L.ctk_synth_fill.argtypes = [p, p, i64, i32, i32, C.c_uint64]
    L.ctk_synth_fill_window.argtypes = [p, p, i64, i64, i32, i32, C.c_uint64]
    L.ctk_checksum_i32_dev.argtypes = [p, p, i64, i64, p]
    L.ctk_dev_memset.argtypes = [p, p, i32, sz]
    L.ctk_check_flag_dev.argtypes = [p, p, p, i64, i32, i32, p, i32, i32, i64, p]
    L.ctk_comm_set_timeout.argtypes = [p, dbl]
    L.ctk_comm_failed.argtypes = [p, C.POINTER(i32), C.POINTER(i32)]
    L.ctk_comm_abort_rank.argtypes = [p, i32]
    L.ctk_debug_fail_at.argtypes = [p, i32]
    for name in ("ctk_lifecycle_f32", "ctk_lifecycle_f64", "ctk_lifecycle_f32_dev", "ctk_lifecycle_f64_dev"):
        getattr(L, name).argtypes = [p, p, p, i64, i32, i32, p, C.POINTER(i64)]
    L.ctk_lifecycle_rows.argtypes = [p, p, i64]
    L.ctk_lifecycle_exact.argtypes = [p, p, i64, p]
    for name in ("ctk_anom_f32", "ctk_anom_f64"):
        getattr(L, name).argtypes = [p, p, i64, i32, i32, p, i32, i32, i32, p, p, p, i32]
    L.ctk_resident_anom.argtypes = [p, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.ctk_resident_anom_generation.argtypes = [p, C.POINTER(C.c_uint64)]
    L.ctk_track_resident.argtypes = [p, p, i32, p, dbl, i32, i32, p, C.POINTER(i64)]
    for name in ("ctk_percentile_f32", "ctk_percentile_f64"):
        getattr(L, name).argtypes = [p, p, i64, i32, i32, i32, i32, dbl, C.POINTER(dbl)]
    L.ctk_comm_unique_id.argtypes = [p]
    L.ctk_comm_init_rccl.argtypes = [p, p, i32, i32, pp]
    L.ctk_comm_group_create.argtypes = [i32, pp]
    L.ctk_comm_group_destroy.argtypes = [p]
    L.ctk_comm_group_destroy.restype = None
    L.ctk_comm_init_local.argtypes = [p, p, i32, pp]
    L.ctk_comm_init_shm.argtypes = [p, C.c_char_p, i32, i32, pp]
    L.ctk_comm_destroy.argtypes = [p]
    L.ctk_comm_destroy.restype = None
    L.ctk_comm_rank.argtypes = [p]
    L.ctk_comm_world.argtypes = [p]
    L.ctk_comm_barrier.argtypes = [p]
    L.ctk_comm_allgather_host.argtypes = [p, p, p, sz]
    L.ctk_comm_ops.argtypes = [p, C.POINTER(i64), C.POINTER(i64)]
    L.ctk_comm_rccl_library.argtypes = []
    L.ctk_comm_rccl_library.restype = C.c_char_p
    sharded_args = [p, p, p, i64, i64, i64, i32, i32, p, i32, p, dbl, i32, i32, p, C.POINTER(i64)]
    L.ctk_track_sharded_f32_dev.argtypes = sharded_args
    L.ctk_track_sharded_f64_dev.argtypes = sharded_args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().ctk_last_error().decode("utf-8", "replace")
        if rc == -3:
            raise MemoryError(msg)
        if rc in (-1, -4):
            raise ValueError(msg)
        if rc == -7:
            raise CommError("libcontrack_hip rc=-7: %s" % msg)
        raise ContrackHipError("libcontrack_hip rc=%d: %s" % (rc, msg))


def device_count():
    return int(lib().ctk_device_count())


def weights_to_limbs(wrow, npix=1 << 16, with_bits=False):
    """exact integer limbs of float32 row weights: w[y] = (lo[y] + hi[y] * 2**bits) / 2**shift.  npix = ny * nx of the grid
    (only matters when the weights span more than 62 bits).  Returns (lo, hi, shift) or (lo, hi, shift, bits)."""
    wrow = np.ascontiguousarray(wrow, dtype=np.float32)
    lo = np.empty(wrow.shape[0], dtype=np.int64)
    hi = np.empty(wrow.shape[0], dtype=np.int64)
    sh, lb = C.c_int32(0), C.c_int32(0)
    check(lib().ctk_weights_to_limbs(wrow.ctypes.data, wrow.shape[0], int(npix), lo.ctypes.data, hi.ctypes.data, C.byref(sh), C.byref(lb)))
    return (lo, hi, int(sh.value), int(lb.value)) if with_bits else (lo, hi, int(sh.value))


def expand_runs_host(mask, rowstart, run_base, run_val, nx):
    """the host-side decoder of the run-table result transfer on its own (no device call): mask uint64 (T, ny, ceil(nx / 64)),
    rowstart uint32 (T, ny), run_base uint32 (T + 1,), run_val int32 (runs,) -> (flag int32 (T, ny, nx), a zero was written,
    a negative run value was met)"""
    mask = np.ascontiguousarray(mask, dtype=np.uint64)
    rowstart = np.ascontiguousarray(rowstart, dtype=np.uint32)
    run_base = np.ascontiguousarray(run_base, dtype=np.uint32)
    run_val = np.ascontiguousarray(run_val, dtype=np.int32)
    T, ny, W = mask.shape
    if W != (nx + 63) // 64 or rowstart.shape != (T, ny) or run_base.shape != (T + 1,) or (T and run_val.shape[0] < int(run_base[-1])):
        raise ValueError("table shapes do not fit (T, ny, nx)")
    flag = np.empty((T, ny, nx), dtype=np.int32)
    z, cx = C.c_int(0), C.c_int(0)
    check(lib().ctk_expand_runs_host(mask.ctypes.data, rowstart.ctypes.data, run_base.ctypes.data, run_val.ctypes.data if run_val.size else None,
                                     T, ny, nx, flag.ctypes.data, C.byref(z), C.byref(cx)))
    return flag, bool(z.value), bool(cx.value)


class Result:
    """Owner of a ctk_result (output of the host-side resolver)."""

    def __init__(self, ptr):
        self._p = ptr

    def __del__(self):
        self.free()

    def free(self):
        if getattr(self, "_p", None):
            lib().ctk_result_free(self._p)
            self._p = None

    @property
    def ptr(self):
        return self._p

    def info(self):
        v = [C.c_int64(0) for _ in range(5)]
        check(lib().ctk_result_info(self._p, *[C.byref(x) for x in v]))
        return dict(zip(("n_labels", "n_ops", "n_complex", "n_ambiguous", "n_components"), (int(x.value) for x in v)))

    def arrays(self):
        cl, ops, sco, sto = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nc, nops = C.c_int64(0), C.c_int64(0)
        check(lib().ctk_result_arrays(self._p, C.byref(cl), C.byref(nc), C.byref(ops), C.byref(nops), C.byref(sco), C.byref(sto)))
        n = int(nc.value)
        comp_label = np.ctypeslib.as_array(C.cast(cl, C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy()
        k = int(nops.value)
        opsa = np.ctypeslib.as_array(C.cast(ops, C.POINTER(C.c_int32)), shape=(max(k, 1) * 8,))[:k * 8].copy().reshape(k, 8)
        return comp_label, opsa

    def shard_offsets(self):
        """(component offsets, timestep offsets) of every shard inside comp_label: two int64 arrays of nshards+1"""
        cl, ops, sco, sto = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nc, nops = C.c_int64(0), C.c_int64(0)
        check(lib().ctk_result_arrays(self._p, C.byref(cl), C.byref(nc), C.byref(ops), C.byref(nops), C.byref(sco), C.byref(sto)))
        k = int(lib().ctk_result_nshards(self._p)) + 1
        co = np.ctypeslib.as_array(C.cast(sco, C.POINTER(C.c_int64)), shape=(k,)).copy()
        to = np.ctypeslib.as_array(C.cast(sto, C.POINTER(C.c_int64)), shape=(k,)).copy()
        return co, to


def resolve(blobs, overlap, twosided):
    """blobs: list of bytes-like / (address, nbytes) table blobs in time order."""
    L = lib()
    n = len(blobs)
    ptrs = (C.c_void_p * n)()
    sizes = (C.c_size_t * n)()
    keep = []
    for i, b in enumerate(blobs):
        if isinstance(b, tuple):
            ptrs[i], sizes[i] = b
        else:
            arr = np.frombuffer(b, dtype=np.uint8)
            keep.append(arr)
            ptrs[i], sizes[i] = arr.ctypes.data, arr.size
    out = C.c_void_p()
    check(L.ctk_resolve(ptrs, sizes, n, float(overlap), int(bool(twosided)), C.byref(out)))
    return Result(out)


COMM_ID_BYTES = 128


def rccl_library():
    """the librccl file the RCCL transport loaded ("" before the first RCCL communicator)"""
    return (lib().ctk_comm_rccl_library() or b"").decode()


def comm_unique_id():
    """128 opaque bytes (ncclGetUniqueId) made on rank 0; every rank passes them to Comm.rccl"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(lib().ctk_comm_unique_id(buf))
    return buf.raw


class CommGroup:
    """Several ranks inside one process (one host thread per rank): ctk_comm_group"""

    def __init__(self, world):
        self._g = C.c_void_p()
        self.world = int(world)
        check(lib().ctk_comm_group_create(self.world, C.byref(self._g)))

    def close(self):
        if getattr(self, "_g", None):
            lib().ctk_comm_group_destroy(self._g)
            self._g = None

    def __del__(self):
        self.close()


class Comm:
    """The communicator of the time-sharded path (ctk_comm), bound to one Tracker."""

    def __init__(self, ptr, keep=None):
        self._c, self._keep = ptr, keep

    @classmethod
    def rccl(cls, tracker, unique_id, rank, world):
        c = C.c_void_p()
        check(lib().ctk_comm_init_rccl(tracker.handle, unique_id, int(rank), int(world), C.byref(c)))
        return cls(c)

    @classmethod
    def local(cls, tracker, group, rank):
        c = C.c_void_p()
        check(lib().ctk_comm_init_local(tracker.handle, group._g, int(rank), C.byref(c)))
        return cls(c, keep=group)

    @classmethod
    def shm(cls, tracker, name, rank, world):
        c = C.c_void_p()
        check(lib().ctk_comm_init_shm(tracker.handle, name.encode(), int(rank), int(world), C.byref(c)))
        return cls(c)

    @property
    def ptr(self):
        return self._c

    @property
    def rank(self):
        return int(lib().ctk_comm_rank(self._c))

    @property
    def world(self):
        return int(lib().ctk_comm_world(self._c))

    def barrier(self):
        check(lib().ctk_comm_barrier(self._c))

    def allgather(self, arr):
        """small host array (<= 4096 bytes) from every rank -> array (world, ...)"""
        arr = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + arr.shape, dtype=arr.dtype)
        check(lib().ctk_comm_allgather_host(self._c, arr.ctypes.data, out.ctypes.data, arr.nbytes))
        return out

    def set_timeout(self, seconds):
        """deadline of every wait of the time-shard path on this communicator (default 120 s / CTK_COMM_TIMEOUT_S)"""
        check(lib().ctk_comm_set_timeout(self._c, float(seconds)))

    def failed(self):
        """None, or (code, rank) of the failure published by the rank that gave up first"""
        a, b = C.c_int(0), C.c_int(-1)
        check(lib().ctk_comm_failed(self._c, C.byref(a), C.byref(b)))
        return None if a.value == 0 else (int(a.value), int(b.value))

    def abort(self, code=-5):
        """this rank gives up: the other ranks' calls return CommError instead of waiting for it"""
        check(lib().ctk_comm_abort_rank(self._c, int(code)))

    def ops(self):
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib().ctk_comm_ops(self._c, C.byref(a), C.byref(b)))
        return dict(neighbour_exchanges=int(a.value), allgathers=int(b.value))

    def close(self):
        if getattr(self, "_c", None):
            lib().ctk_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        self.close()


class _ResultPool:
    """Recycles the memory of result arrays (the `flag` slabs the host-array entries return).

    A fresh np.empty array consists of pages that do not exist yet: the device -> host copy then runs behind first-touch page
    faults (47 GB/s through eight bounce threads at best; measured, tools/d2h_probe*.py).  Memory that HAS been touched can be
    registered with HIP in ~1 ms and takes the result in ONE DMA at PCIe rate (57 GB/s) -- so the blocks of results the caller has
    dropped are kept (a few, CTK_RESULT_POOL_MB in all, default 2048), registered on their first reuse, and handed out again.
    A loop over ensemble members that writes each result out and drops it gets every result but the first at DMA speed; a
    caller that keeps every result sees exactly the old behaviour.  The arrays handed out are ordinary numpy arrays; a block
    returns to the pool when the last view of it is gone (weakref on the buffer all views share)."""

    def __init__(self, tracker):
        import threading
        import collections
        self._trk = weakref.ref(tracker)
        self._lock = threading.RLock()       # re-entrant: belt and braces, see _release
        self._free = []                      # blocks: dict(mem=np.uint8 array, registered=bool)
        self._leased = {}                    # id(block) -> block
        self._returned = collections.deque()  # blocks whose last view died; filed under the lock by _drain
        self._closed = False
        self.cap = int(float(os.environ.get("CTK_RESULT_POOL_MB", "2048")) * (1 << 20))
        self.hits = self.misses = 0

    def take(self, shape, dtype=np.int32):
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        if nbytes < (8 << 20) or self.cap <= 0:            # small results: nothing to gain
            return np.empty(shape, dtype=dtype)
        blk = None
        with self._lock:
            self._drain()
            i = 0
            while i < len(self._free):
                b = self._free[i]
                if nbytes <= b["mem"].nbytes <= nbytes + nbytes // 4:
                    blk = self._free.pop(i)
                    break
                i += 1
        if blk is None:
            self.misses += 1
            blk = dict(mem=np.empty(nbytes, dtype=np.uint8), registered=False)
        else:
            self.hits += 1
            trk = self._trk()
            if not blk["registered"] and trk is not None and trk.handle and not trk.result_as_runs:
                # (touched by its first use: registration is cheap now; a failure just leaves the block pageable.  With the
                # result travelling as run tables -- the default -- host threads write the block: nothing to register)
                blk["registered"] = lib().ctk_host_register(trk.handle, blk["mem"].ctypes.data, blk["mem"].nbytes) == 0
        carr = (C.c_ubyte * nbytes).from_address(blk["mem"].ctypes.data)
        with self._lock:
            self._leased[id(blk)] = blk
        weakref.finalize(carr, self._release, blk)
        return np.frombuffer(carr, dtype=dtype).reshape(shape)

    def _unregister(self, blk):
        trk = self._trk()
        if blk["registered"] and trk is not None and trk.handle:
            lib().ctk_host_unregister(trk.handle, blk["mem"].ctypes.data)
        blk["registered"] = False

    def _release(self, blk):
        """weakref.finalize callback: runs wherever the last reference dies -- possibly inside a cyclic-GC pass that a
        thread triggered while it held the pool's lock.  So it takes no lock and allocates nothing GC-tracked: the block goes
        onto a deque (append is atomic) and is filed by the next take() / close()."""
        self._returned.append(blk)

    def _drain(self):
        """file the returned blocks (caller holds the lock)"""
        drop = []
        while True:
            try:
                blk = self._returned.popleft()
            except IndexError:
                break
            self._leased.pop(id(blk), None)
            held = 0
            for b in self._free:
                held += b["mem"].nbytes
            if not self._closed and self._trk() is not None and held + blk["mem"].nbytes <= self.cap:
                self._free.append(blk)
            else:
                drop.append(blk)
        for blk in drop:
            self._unregister(blk)

    def close(self):
        """the handle goes away: nothing stays registered (arrays still held by the caller remain valid, pageable memory)
        and nothing that comes back afterwards is kept"""
        with self._lock:
            self._closed = True
            self._drain()
            blocks = self._free + list(self._leased.values())
            self._free = []
            self._leased = {}
        for b in blocks:
            self._unregister(b)


class Tracker:
    """One GPU + stream + reusable device workspace (ctk_handle)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().ctk_create(C.byref(self._h), int(device)))
        self._pool = _ResultPool(self)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_pool", None) is not None:
                self._pool.close()
            lib().ctk_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    # ---- one call, host numpy in / out ------------------------------------------------------
    def track(self, anom, thr, cmp_op, wrow, overlap, persistence, twosided=True, f64=False, out=None):
        anom = np.ascontiguousarray(anom, dtype=np.float64 if f64 else np.float32)
        T, ny, nx = anom.shape
        thr = np.ascontiguousarray(thr, dtype=np.float64)
        wrow = np.ascontiguousarray(wrow, dtype=np.float32)
        if thr.shape != (T,) or wrow.shape != (ny,):
            raise ValueError("thr must have shape (T,) and wrow (ny,)")
        if out is not None:
            if out.dtype != np.int32 or out.shape != (T, ny, nx) or not out.flags.c_contiguous or not out.flags.writeable:
                raise ValueError("out must be a writable C-contiguous int32 array of the slab's shape")
            flag = out
        else:
            flag = self._pool.take((T, ny, nx))
        n = C.c_int64(0)
        fn = lib().ctk_track_f64 if f64 else lib().ctk_track_f32
        check(fn(self._h, anom.ctypes.data, T, ny, nx, thr.ctypes.data, int(cmp_op), wrow.ctypes.data,
                                  float(overlap), int(persistence), int(bool(twosided)), flag.ctypes.data, C.byref(n)))
        return flag, int(n.value)

    # ---- streaming entries (next row N4) ------------------------------------------------------------------------
    def track_stream(self, source, thr, cmp_op, wrow, overlap, persistence, twosided=True, sink=None, shape=None, dtype=None, chunk_steps=0):
        """ctk_track_* with the slab passing through chunk-sized device buffers (device footprint: 4 chunks + slab / 32).

        source: a (T, ny, nx) float32 / float64 array (np.memmap included), or a callable reader(t0, nt, out) that fills
                `out` (a (nt, ny, nx) view of pinned memory) with the timesteps [t0, t0 + nt) -- then `shape` = (T, ny, nx)
                and `dtype` are required;
        sink:   None (a new int32 array is returned), an int32 array (T, ny, nx), or a callable writer(t0, nt, flags) that
                receives each flag chunk as a (nt, ny, nx) int32 view valid during the call.
        Returns (flag array or None, n_tracked)."""
        L = lib()
        if callable(source):
            if shape is None or dtype is None:
                raise ValueError("a reader callback needs shape=(T, ny, nx) and dtype")
            T, ny, nx = (int(v) for v in shape)
            dt = np.dtype(dtype)
        else:
            source = np.ascontiguousarray(source) if not isinstance(source, np.memmap) else source
            if source.dtype not in (np.float32, np.float64):
                source = np.ascontiguousarray(source, dtype=np.float64)
            T, ny, nx = source.shape
            dt = source.dtype
        if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("the slab must be float32 or float64")
        thr = np.ascontiguousarray(thr, dtype=np.float64)
        wrow = np.ascontiguousarray(wrow, dtype=np.float32)
        if thr.shape != (T,) or wrow.shape != (ny,):
            raise ValueError("thr must have shape (T,) and wrow (ny,)")
        out = None
        if sink is None:
            out = sink = np.empty((T, ny, nx), dtype=np.int32)
        elif not callable(sink):
            if sink.dtype != np.int32 or sink.shape != (T, ny, nx) or not sink.flags.c_contiguous:
                raise ValueError("the sink array must be C-contiguous int32 (T, ny, nx)")
            out = sink
        n = C.c_int64(0)
        tail = (thr.ctypes.data, int(cmp_op), wrow.ctypes.data, float(overlap), int(persistence), int(bool(twosided)))
        if not callable(source) and not callable(sink):
            fn = L.ctk_track_stream_f64 if dt == np.float64 else L.ctk_track_stream_f32
            check(fn(self._h, source.ctypes.data, T, ny, nx, *tail, sink.ctypes.data, C.byref(n), int(chunk_steps)))
            return out, int(n.value)
        errors = []

        def rd(_user, t0, nt, dst):
            try:
                view = np.ctypeslib.as_array(C.cast(dst, C.POINTER(C.c_float if dt == np.float32 else C.c_double)), shape=(nt, ny, nx))
                if callable(source):
                    source(int(t0), int(nt), view)
                else:
                    view[...] = source[t0:t0 + nt]
                return 0
            except BaseException as e:                    # an exception must not cross the C frames
                errors.append(e)
                return 1

        def wr(_user, t0, nt, src):
            try:
                view = np.ctypeslib.as_array(C.cast(src, C.POINTER(C.c_int32)), shape=(nt, ny, nx))
                if callable(sink):
                    sink(int(t0), int(nt), view)
                else:
                    sink[t0:t0 + nt] = view
                return 0
            except BaseException as e:
                errors.append(e)
                return 1
        rcb, wcb = READ_CHUNK_FN(rd), WRITE_CHUNK_FN(wr)
        rc = L.ctk_track_stream_cb(self._h, dt.itemsize, T, ny, nx, rcb, None, *tail, wcb, None, C.byref(n), int(chunk_steps))
        if errors:
            raise errors[0]
        check(rc)
        return out, int(n.value)

    def stream_times(self):
        """ms of the last streaming call: reader callbacks, writer callbacks, input phase, output phase"""
        ms = (C.c_double * 4)()
        check(lib().ctk_stream_times(self._h, ms))
        return dict(zip(("reader", "writer", "input_phase", "output_phase"), (float(v) for v in ms)))

    # ---- calc_anom / percentile threshold on the device ---------------------------------------------------------
    def anomalies(self, x, group, ngroups, window=1, smooth=1, clim=None, want_anom=True, want_clim=False, keep_resident=False):
        """x (T, ny, nx) float32 / float64; group: T ids in [0, ngroups).  Returns (anom or None, clim or None)."""
        x = np.ascontiguousarray(x)
        f64 = x.dtype != np.float32
        if f64:
            x = np.ascontiguousarray(x, dtype=np.float64)
        T, ny, nx = x.shape
        group = np.ascontiguousarray(group, dtype=np.int32)
        if group.shape != (T,):
            raise ValueError("group must hold one id per timestep")
        cin = None if clim is None else np.ascontiguousarray(clim, dtype=x.dtype)
        if cin is not None and cin.shape != (ngroups, ny, nx):
            raise ValueError("clim must have shape (ngroups, ny, nx)")
        anom = np.empty_like(x) if want_anom else None
        cout = np.empty((ngroups, ny, nx), dtype=x.dtype) if want_clim else None
        fn = lib().ctk_anom_f64 if f64 else lib().ctk_anom_f32
        check(fn(self._h, x.ctypes.data, T, ny, nx, group.ctypes.data, int(ngroups), int(window), int(smooth),
                 None if cin is None else cin.ctypes.data, None if anom is None else anom.ctypes.data,
                 None if cout is None else cout.ctypes.data, int(bool(keep_resident))))
        return anom, cout

    def resident_anom(self):
        T, ny, nx, f = C.c_int64(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().ctk_resident_anom(self._h, C.byref(T), C.byref(ny), C.byref(nx), C.byref(f)))
        return None if T.value < 0 else (int(T.value), int(ny.value), int(nx.value), bool(f.value))

    def resident_generation(self):
        """identity of the resident anomaly slab (changes whenever one is written or dropped)"""
        g = C.c_uint64(0)
        check(lib().ctk_resident_anom_generation(self._h, C.byref(g)))
        return int(g.value)

    def track_resident(self, thr, cmp_op, wrow, overlap, persistence, twosided=True):
        shape = self.resident_anom()
        if shape is None:
            raise ContrackHipError("no anomaly slab is resident on the device")
        T, ny, nx, _ = shape
        thr = np.ascontiguousarray(thr, dtype=np.float64)
        wrow = np.ascontiguousarray(wrow, dtype=np.float32)
        flag = self._pool.take((T, ny, nx))
        n = C.c_int64(0)
        check(lib().ctk_track_resident(self._h, thr.ctypes.data, int(cmp_op), wrow.ctypes.data, float(overlap), int(persistence),
                                       int(bool(twosided)), flag.ctypes.data, C.byref(n)))
        return flag, int(n.value)

    def percentile(self, x, y0, y1, q):
        """mean over rows [y0, y1) of the per-grid-point q-quantile over time; x = None: the resident anomaly slab"""
        out = C.c_double(0.0)
        if x is None:
            shape = self.resident_anom()
            if shape is None:
                raise ContrackHipError("no anomaly slab is resident on the device")
            T, ny, nx, f64 = shape
            ptr = None
        else:
            x = np.ascontiguousarray(x)
            f64 = x.dtype != np.float32
            if f64:
                x = np.ascontiguousarray(x, dtype=np.float64)
            T, ny, nx = x.shape
            ptr = x.ctypes.data
        fn = lib().ctk_percentile_f64 if f64 else lib().ctk_percentile_f32
        check(fn(self._h, ptr, T, ny, nx, int(y0), int(y1), float(q), C.byref(out)))
        return float(out.value)

    def release_io(self):
        """free the device copies of slab / result that the host-array calls keep in the handle"""
        check(lib().ctk_release_io(self._h))

    # ---- device-resident --------------------------------------------------------------------------
    def malloc(self, nbytes):
        p = C.c_void_p()
        check(lib().ctk_dev_malloc(self._h, C.byref(p), int(nbytes)))
        return p

    def free(self, p):
        check(lib().ctk_dev_free(self._h, p))

    def h2d(self, dst, arr):
        arr = np.ascontiguousarray(arr)
        check(lib().ctk_memcpy_h2d(self._h, dst, arr.ctypes.data, arr.nbytes))

    def d2h(self, arr, src):
        assert arr.flags.c_contiguous
        check(lib().ctk_memcpy_d2h(self._h, arr.ctypes.data, src, arr.nbytes))

    def sync(self):
        check(lib().ctk_sync(self._h))

    def synth_fill(self, dst, T, ny, nx, seed=0, t0=0):
        """deterministic synthetic slab (measurements and tests only); t0 > 0: the window [t0, t0 + T) of the same slab"""
        check(lib().ctk_synth_fill_window(self._h, dst, int(t0), T, ny, nx, int(seed)))

    def checksum_i32(self, ptr, n, index0=0):
        """(position-weighted 64-bit checksum, number of nonzero elements) of an int32 device array"""
        out = np.zeros(2, dtype=np.uint64)
        check(lib().ctk_checksum_i32_dev(self._h, ptr, int(n), int(index0), out.ctypes.data))
        return int(out[0]), int(out[1])

    def memset(self, ptr, byte, nbytes):
        check(lib().ctk_dev_memset(self._h, ptr, int(byte), int(nbytes)))

    def check_flag(self, anom_dev, flag_dev, T, ny, nx, thr, cmp_op, persistence, max_id):
        """device-side properties of a result (ctk_check_flag_dev): dict of counts"""
        thr = np.ascontiguousarray(thr, dtype=np.float64)
        if thr.shape != (T,):
            raise ValueError("thr must have shape (T,)")
        out = np.zeros(6, dtype=np.uint64)
        check(lib().ctk_check_flag_dev(self._h, anom_dev, flag_dev, int(T), int(ny), int(nx), thr.ctypes.data, int(cmp_op), int(persistence), int(max_id),
                                       out.ctypes.data))
        return dict(zip(("flag_outside_mask", "ids_out_of_range", "nonzero", "ids", "ids_below_persistence", "max_id"), (int(v) for v in out)))

    def debug_fail_at(self, stage):
        check(lib().ctk_debug_fail_at(self._h, int(stage)))

    def _thr_w_ptrs(self, thr, wrow):
        """addresses of the per-step thresholds (float64) and row weights (float32); remembered while the caller passes the very same
        arrays in the right layout (a loop over passes: the conversions and two ctypes objects per call were ~3 us with the GPU idle)"""
        c = getattr(self, "_pc", None)
        if c is not None and c[0] is thr and c[1] is wrow:
            return c[2], c[3], thr, wrow
        t = np.ascontiguousarray(thr, dtype=np.float64)
        w = np.ascontiguousarray(wrow, dtype=np.float32)
        tp, wp = t.ctypes.data, w.ctypes.data
        self._pc = (thr, wrow, tp, wp) if (t is thr and w is wrow) else None      # (never the address of a converted copy)
        return tp, wp, t, w

    def track_dev(self, anom_dev, T, ny, nx, thr, cmp_op, wrow, overlap, persistence, twosided, flag_dev):
        tp, wp, thr, wrow = self._thr_w_ptrs(thr, wrow)
        n = C.c_int64(0)
        check(lib().ctk_track_f32_dev(self._h, anom_dev, T, ny, nx, tp, int(cmp_op), wp,
                                      float(overlap), int(persistence), int(bool(twosided)), flag_dev, C.byref(n)))
        return int(n.value)

    def track_sharded_dev(self, comm, anom_dev, T_local, t_begin, T_total, ny, nx, thr, cmp_op, wrow, overlap, persistence, twosided, flag_dev,
                          f64=False):
        """the whole path on the time shard [t_begin, t_begin + T_local) of T_total steps; every rank of `comm` must call"""
        tp, wp, thr, wrow = self._thr_w_ptrs(thr, wrow)
        if thr.shape != (T_local,):
            raise ValueError("thr must hold one value per local timestep")
        n = C.c_int64(0)
        fn = lib().ctk_track_sharded_f64_dev if f64 else lib().ctk_track_sharded_f32_dev
        check(fn(self._h, comm.ptr, anom_dev, int(T_local), int(t_begin), int(T_total), ny, nx, tp, int(cmp_op), wp,
                 float(overlap), int(persistence), int(bool(twosided)), flag_dev, C.byref(n)))
        return int(n.value)

    # ---- run_lifecycle reductions ------------------------------------------------------------------------
    def _life_rows(self, n):
        rows = np.empty(n, dtype=LIFE_ROW)
        check(lib().ctk_lifecycle_rows(self._h, rows.ctypes.data, n))
        return rows

    def lifecycle(self, flag, field, wrow, resident_f64=None):
        """flag (T, ny, nx) int32, field float32/float64 of the same shape -> LIFE_ROW records sorted by (label, t).
        field = None: the anomaly slab left resident by `anomalies(..., keep_resident=True)` (resident_f64: its type)"""
        flag = np.ascontiguousarray(flag, dtype=np.int32)
        if field is None:
            f64 = bool(resident_f64)
        else:
            f64 = np.asarray(field).dtype == np.float64
            field = np.ascontiguousarray(field, dtype=np.float64 if f64 else np.float32)
        if flag.ndim != 3 or (field is not None and field.shape != flag.shape):
            raise ValueError("flag and field must share one (time, lat, lon) shape")
        T, ny, nx = flag.shape
        wrow = np.ascontiguousarray(wrow, dtype=np.float32)
        if wrow.shape != (ny,):
            raise ValueError("wrow must have shape (ny,)")
        n = C.c_int64(0)
        fn = lib().ctk_lifecycle_f64 if f64 else lib().ctk_lifecycle_f32
        check(fn(self._h, flag.ctypes.data, None if field is None else field.ctypes.data, T, ny, nx, wrow.ctypes.data, C.byref(n)))
        return self._life_rows(int(n.value))

    def lifecycle_exact(self, row_idx):
        """rows of the LAST lifecycle call (indices into its sorted rows) in the reference's own summation orders: LIFE_EXACT"""
        idx = np.ascontiguousarray(row_idx, dtype=np.int64)
        out = np.empty(len(idx), dtype=LIFE_EXACT)
        check(lib().ctk_lifecycle_exact(self._h, idx.ctypes.data, len(idx), out.ctypes.data))
        return out

    def lifecycle_dev(self, flag_dev, field_dev, T, ny, nx, wrow, f64=False):
        wrow = np.ascontiguousarray(wrow, dtype=np.float32)
        n = C.c_int64(0)
        fn = lib().ctk_lifecycle_f64_dev if f64 else lib().ctk_lifecycle_f32_dev
        check(fn(self._h, flag_dev, field_dev, T, ny, nx, wrow.ctypes.data, C.byref(n)))
        return self._life_rows(int(n.value))

    def set_timing(self, level=2):
        """0: off; 1: HIP events around k_threshold / k_relabel only; 2 (or True): around every kernel group"""
        check(lib().ctk_set_timing(self._h, 2 if level is True else int(level)))

    def set_fused(self, enable=True):
        """the one-call entries without a host hand-off (default) or always on the synchronous path"""
        check(lib().ctk_set_fused_pass(self._h, int(bool(enable))))

    def set_result_transfer(self, mode=-1):
        """how the host-array entries bring the result over PCIe: 1 run tables expanded by host threads (default), 0 the dense
        slab written by k_relabel, -1 the environment's choice (CTK_RLE_OUT); 2: test hook -- run tables wanted but made unavailable (the
        library then repeats the pass with the dense copy)"""
        check(lib().ctk_set_result_transfer(self._h, int(mode)))
        self._transfer_mode = int(mode)

    @property
    def result_as_runs(self):
        m = getattr(self, "_transfer_mode", -1)
        if m >= 0:
            return m == 1
        import re
        v = os.environ.get("CTK_RLE_OUT")                      # (as the library reads it: atoi)
        mt = re.match(r"\s*[+-]?\d+", v) if v is not None else None
        return v is None or (mt is not None and int(mt.group(0)) != 0)

    def stats(self):
        v = np.zeros(27, dtype=np.int64)
        check(lib().ctk_get_stats_n(self._h, v.ctypes.data, 27))
        names = ["runs", "max_runs_per_step", "components", "pairs", "seam_rows_to_driver", "labels_3d", "seam_ops",
                 "filter_passes", "host_path", "seam_loop_ns", "seam_folds", "seam_copy_ns", "ungrouped_pairs", "pair_table_regrows", "filter_rounds",
                 "ambiguous_decisions", "exact_fixups", "shared_seam_rows", "off_fused_path_reason", "relabel_kernel", "fused_pass", "x4_speculated", "result_as_runs", "mask_allocations_tried", "mask_ratio_x1000",
                 "mask_check_us", "mask_spacer_mb"]
        return dict(zip(names, v.tolist()))

    def debug_set_pair_capacity(self, records):
        check(lib().ctk_debug_set_pair_capacity(self._h, int(records)))

    def debug_set_seam_caps(self, labels, ops):
        check(lib().ctk_debug_set_seam_caps(self._h, int(labels), int(ops)))

    def debug_set_spin(self, limit_ms=0.0, stall_mode=0):
        """bounded inter-workgroup waits of the one-launch filter pass: limit (0 = default 200 ms), 1 = head of the chain late, 2 = never"""
        check(lib().ctk_debug_set_spin(self._h, float(limit_ms), int(stall_mode)))

    def debug_set_mailbox(self, cand_records, labels):
        check(lib().ctk_debug_set_mailbox(self._h, int(cand_records), int(labels)))

    def set_filter_round(self, passes):
        check(lib().ctk_set_filter_round(self._h, int(passes)))

    def set_device_resolve(self, on=True):
        check(lib().ctk_set_device_resolve(self._h, int(bool(on))))

    def timing_sums(self, reset=True):
        """(mean ms per measured call, calls that measured it) per kernel group since the last reset"""
        sums = np.zeros(len(TIMER_NAMES), dtype=np.float64)
        cnt = np.zeros(len(TIMER_NAMES), dtype=np.int64)
        check(lib().ctk_get_timing_sums(self._h, sums.ctypes.data, cnt.ctypes.data, int(bool(reset))))
        return ({k: (float(v) / int(n) if n else 0.0) for k, v, n in zip(TIMER_NAMES, sums, cnt)}, dict(zip(TIMER_NAMES, cnt.tolist())))

    def time_relabel(self, flag_dev, persistence, variant, xcd=-1, reps=5):
        """ms of `reps` launches of the write kernel on the finished tables of the last track_dev pass (variant 0 k_relabel_v5, 1 without its
        SGPR limit; xcd: chunk -> XCD order, -1 the handle's)"""
        ms = np.zeros(reps, dtype=np.float64)
        check(lib().ctk_debug_time_relabel(self._h, flag_dev, int(persistence), int(variant), int(xcd), int(reps), ms.ctypes.data))
        return ms

    def stream_ceiling(self, ptr, nbytes, write, reps=5):
        """best-of-reps ms of a plain 16-byte non-temporal store (write; 2: one contiguous eighth of the buffer per XCD) / load stream
        over a device buffer (overwritten when write)"""
        ms = C.c_double(0.0)
        check(lib().ctk_debug_stream_ceiling(self._h, ptr, int(nbytes), int(write), int(reps), C.byref(ms)))
        return ms.value

    def timings(self):
        ms = np.zeros(len(TIMER_NAMES), dtype=np.float64)
        check(lib().ctk_get_timings(self._h, ms.ctypes.data))
        return dict(zip(TIMER_NAMES, ms.tolist()))

    # ---- staged (time-sharded) -------------------------------------------------------------------
    def shard_label2d(self, anom_dev, T, ny, nx, thr, cmp_op, wrow, has_prev):
        thr = np.ascontiguousarray(thr, dtype=np.float64)
        wrow = np.ascontiguousarray(wrow, dtype=np.float32)
        check(lib().ctk_shard_label2d(self._h, anom_dev, T, ny, nx, thr.ctypes.data, int(cmp_op), wrow.ctypes.data, int(bool(has_prev))))

    def halo_size(self):
        s = C.c_size_t(0)
        check(lib().ctk_shard_halo_size(self._h, C.byref(s)))
        return int(s.value)

    def halo_export(self):
        p, s = C.c_void_p(), C.c_size_t(0)
        check(lib().ctk_shard_halo_export(self._h, C.byref(p), C.byref(s)))
        return p, int(s.value)

    def halo_import(self, blob_dev, nbytes):
        check(lib().ctk_shard_halo_import(self._h, blob_dev, int(nbytes)))

    def shard_overlap(self):
        check(lib().ctk_shard_overlap(self._h))

    def shard_tables(self):
        p, s = C.c_void_p(), C.c_size_t(0)
        check(lib().ctk_shard_tables(self._h, C.byref(p), C.byref(s)))
        return C.string_at(p, s.value)

    def shard_extents(self, result, shard, t_begin):
        p, n = C.c_void_p(), C.c_int64(0)
        check(lib().ctk_shard_extents(self._h, result.ptr, int(shard), int(t_begin), C.byref(p), C.byref(n)))
        return p, int(n.value)

    def shard_write(self, persistence, flag_dev):
        n, z = C.c_int64(0), C.c_int(0)
        check(lib().ctk_shard_write(self._h, int(persistence), flag_dev, C.byref(n), C.byref(z)))
        return int(n.value), bool(z.value)

    def debug_mask(self, T, ny, nx):
        m = np.empty((T, ny, nx), dtype=np.uint8)
        check(lib().ctk_debug_mask(self._h, m.ctypes.data))
        return m

    def debug_label2d(self, T, ny, nx, before_seam):
        lab = np.empty((T, ny, nx), dtype=np.int32)
        check(lib().ctk_debug_label2d(self._h, int(bool(before_seam)), lab.ctypes.data))
        return lab
