// ctk_comm.h -- the small communicator the time-sharded path needs (internal; public entry points in
// include/contrack_hip.h).  One rank per GPU.  Three transports behind the same two primitives:
//   rccl   one process per GPU, RCCL (ncclSend/ncclRecv, ncclAllGather) over xGMI, enqueued on the handle's stream;
//          librccl.so is dlopen'ed on first use (it is a 570 MB library: a single-GPU user never loads it)
//   local  several ranks inside ONE process, one host thread per rank (single-process multi-GPU, and the in-process
//          tests that drive N handles on one GPU): device-to-device copies + a thread barrier
//   shm    one process per rank on one node WITHOUT RCCL (several ranks sharing one GPU -- RCCL refuses that --, e.g.
//          bench.py's N > 1 leg on a one-GPU box): host staging through a POSIX shared-memory segment
// Primitives (both ordered after everything enqueued on `stream` so far; their results are visible to everything
// enqueued afterwards):
//   shift      rank r sends `sbytes` to rank r+dir and receives `rbytes` from rank r-dir (dir = +1 / -1, no wrap-around:
//              the ends only send or only receive).  The two sizes are known on both sides by construction.
//   allgather  every rank contributes `nbytes`; recv holds world * nbytes in rank order.
//
// Failure behaviour (every transport): no rank is ever left waiting for one that gave up or died.
//   * The ranks of a process-per-rank communicator (rccl, shm) share a small CONTROL SEGMENT in POSIX shared memory
//     (single node): {failure code + rank, pid of every rank, barrier words}.  ctk_comm_abort() publishes a failure there;
//     every wait of the path -- ctk_comm_wait() instead of hipStreamSynchronize, the barriers of the shm transport --
//     polls it, polls whether the peers' processes still exist, and carries a deadline (CTK_COMM_TIMEOUT_S, default
//     120 s).  A rank that sees a failure aborts its RCCL communicator (ncclCommAbort: the collective kernels in flight
//     return) and fails with CTK_E_COMM naming the rank that gave up and its error code.
//   * An in-process group has the same flag in the group object.
//   * A failed communicator stays failed: every later operation on it returns CTK_E_COMM at once.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

struct ctk_comm_group;
struct CtkCtlSeg;

struct ctk_comm {
    int rank = 0, world = 1;
    int kind = 0;                       // 0 local, 1 shm, 2 rccl
    int device = 0;
    hipStream_t stream = nullptr;
    // local
    ctk_comm_group *group = nullptr;
    // shm
    void *shm = nullptr;                // data slots (kind 1)
    size_t shm_bytes = 0, shm_slot = 0;
    char shm_name[96] = {0};
    // control segment (kinds 1 and 2; nullptr: a communicator of one rank, or the segment could not be created)
    CtkCtlSeg *ctl = nullptr;
    char ctl_name[96] = {0};
    // rccl
    void *nccl = nullptr;               // ncclComm_t
    uint64_t n_shift = 0, n_allgather = 0;      // operations issued (bench reports them)
    double timeout_s = 120.0;
    bool dead = false;                  // aborted: nothing can be done with it any more
    void *scratch = nullptr;            // small device buffer of ctk_comm_allgather_host
    size_t scratch_cap = 0;
};

int ctk_comm_shift(ctk_comm *c, int dir, const void *send, size_t sbytes, void *recv, size_t rbytes);
int ctk_comm_allgather(ctk_comm *c, const void *send, void *recv, size_t nbytes);
// the stream has drained -- or a rank failed / died / did not arrive before the deadline (CTK_E_COMM; the RCCL communicator
// is aborted so that the stream does drain).  The only way the time-sharded path waits for its stream.
int ctk_comm_wait(ctk_comm *c);
int ctk_comm_wait_word(ctk_comm *c, const volatile uint32_t *word, uint32_t stamp);     // a stamp written to pinned memory by a kernel on the stream
// this rank gives up with `code`: tells every other rank (control segment / group flag) and retires the communicator
void ctk_comm_abort(ctk_comm *c, int code);
