"""_native._ResultPool on the CPU (no device call is made: the stand-in tracker has no handle).  The blocks of dropped result
arrays return through weakref.finalize callbacks, which may run inside a cyclic-GC pass started while the SAME thread holds the
pool's lock (round-4 advisor finding): the callback must not take that lock."""
import gc

import numpy as np

from contrack_amd import _native


class _Trk:
    handle = None
    result_as_runs = True


def test_blocks_are_recycled_and_dropped_after_close():
    trk = _Trk()
    pool = _native._ResultPool(trk)
    shape = (3, 1024, 1024)                                   # 12 MB: above the 8 MB floor
    a = pool.take(shape)
    assert a.shape == shape and a.dtype == np.int32 and pool.misses == 1
    addr = a.ctypes.data
    a[...] = 7
    del a
    b = pool.take(shape)
    assert pool.hits == 1 and b.ctypes.data == addr
    small = pool.take((4, 4, 4))
    assert small.shape == (4, 4, 4) and pool.hits == 1 and pool.misses == 1       # small results bypass the pool
    pool.close()
    del b
    c = pool.take(shape)                                      # a block that returns after close() is not kept
    assert pool.hits == 1 and pool.misses == 2 and len(pool._free) == 0
    del c
    pool.close()
    assert len(pool._returned) == 0 and len(pool._leased) == 0


def test_release_inside_a_gc_pass_while_the_lock_is_held():
    """a pooled array in cyclic garbage, collected while this thread holds the lock: with a plain Lock taken in the finalizer
    this hung"""
    trk = _Trk()                                              # (the pool holds its tracker weakly)
    pool = _native._ResultPool(trk)
    shape = (3, 1024, 1024)

    class Node:
        pass

    n = Node()
    n.me = n
    n.flag = pool.take(shape)
    del n
    with pool._lock:
        gc.collect()                                          # runs the finalizer here
    assert len(pool._returned) == 1
    again = pool.take(shape)
    assert pool.hits == 1 and len(pool._returned) == 0
    del again
