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
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

struct ctk_comm_group;

struct ctk_comm {
    int rank = 0, world = 1;
    int kind = 0;                       // 0 local, 1 shm, 2 rccl
    int device = 0;
    hipStream_t stream = nullptr;
    // local
    ctk_comm_group *group = nullptr;
    // shm
    void *shm = nullptr;
    size_t shm_bytes = 0, shm_slot = 0;
    char shm_name[96] = {0};
    // rccl
    void *nccl = nullptr;               // ncclComm_t
    uint64_t n_shift = 0, n_allgather = 0;      // operations issued (bench reports them)
};

int ctk_comm_shift(ctk_comm *c, int dir, const void *send, size_t sbytes, void *recv, size_t rbytes);
int ctk_comm_allgather(ctk_comm *c, const void *send, void *recv, size_t nbytes);
void ctk_comm_abort(ctk_comm *c);          // this rank gives up: releases the ranks of an in-process group that wait for it
