// ctk_comm.hip -- transports of the time-sharded path's communicator (see ctk_comm.h).  Host code only.
#include "ctk_comm.h"
#include "../../include/contrack_hip_debug.h"

#include <rccl/rccl.h>          // types and prototypes only: librccl.so is dlopen'ed, never linked

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cerrno>
#include <cstddef>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

int ctk_set_error(int code, const char *fmt, ...);            // ctk_resolve.cpp

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return ctk_set_error(e_ == hipErrorOutOfMemory ? CTK_E_NOMEM : CTK_E_NODEVICE, "%s failed: %s (%s:%d)", \
                                 #expr, hipGetErrorString(e_), __FILE__, __LINE__);                           \
    } while (0)

namespace {
double mono_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
double default_timeout_s()
{
    const char *e = getenv("CTK_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 120.0;
}
void nap_us(long us)
{
    struct timespec ts = {0, us * 1000};
    nanosleep(&ts, nullptr);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
std::string g_rccl_path;               // the file the symbols came from (dladdr)

int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.ok) return CTK_OK;
    // WHICH librccl (round-5 verdict: "state which one is intended and pin it").  In this order:
    //   1. CTK_RCCL_LIB, if set: the caller's explicit choice;
    //   2. a copy that is ALREADY mapped into the process (a host that imported torch has torch's bundled librccl loaded): one RCCL per
    //      process -- a second copy would keep its own bootstrap state, its own IPC handles and its own set of proxy threads;
    //   3. the ROCm installation's ($ROCM_PATH/lib, /opt/rocm/lib): the build that belongs to the HIP runtime this library is linked
    //      against -- NOT whatever "librccl.so.1" resolves to first on the loader path (on the round-5 test box: torch's bundled copy);
    //   4. the loader path, as a last resort.
    // ctk_comm_rccl_library() reports the file that was taken; bench_dist.py prints it in its line.
    std::string rocm1, rocm2;
    if (const char *rp = getenv("ROCM_PATH")) { rocm1 = std::string(rp) + "/lib/librccl.so.1"; rocm2 = std::string(rp) + "/lib/librccl.so"; }
    if (const char *e = getenv("CTK_RCCL_LIB")) { if (*e) g_rccl.lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL); }
    if (!g_rccl.lib) for (const char *n : {"librccl.so.1", "librccl.so"}) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD); if (g_rccl.lib) break; }
    const char *names[] = {rocm1.c_str(), rocm2.c_str(), "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        if (g_rccl.lib) break;
        if (!n || !*n) continue;
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!g_rccl.lib) return ctk_set_error(CTK_E_NODEVICE, "librccl.so not found (%s): set CTK_RCCL_LIB", dlerror());
#define SYM(field, name)                                                                                       \
    g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.lib, name);                                            \
    if (!g_rccl.field) return ctk_set_error(CTK_E_NODEVICE, "librccl.so lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(g_rccl.lib, "ncclCommAbort");       // (optional: without it a failed communicator is only abandoned)
    {
        Dl_info di;
        if (dladdr((void *)g_rccl.GetUniqueId, &di) && di.dli_fname) g_rccl_path = di.dli_fname;
    }
    g_rccl.ok = true;
    return CTK_OK;
}
#define NCCLCHK(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return ctk_set_error(CTK_E_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)
}  // namespace

extern "C" const char *ctk_comm_rccl_library(void)
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    return g_rccl_path.c_str();                                // ("" until a communicator of the RCCL transport has been created)
}

// ------------------------------------------------------------------------------------------------
// control segment: what the ranks of a process-per-rank communicator (shm, rccl) share beside the data path
// ------------------------------------------------------------------------------------------------
#define CTK_CTL_MAXWORLD 2048
struct CtkCtlSeg {
    uint64_t magic;
    int32_t world, pad;
    uint64_t slot;                      // bytes per rank data slot (shm transport), 0 otherwise
    // 0 = fine; else set ONCE by a compare-and-swap of the whole 64-bit word {code (low half, < 0), rank (high half)}: a reader
    // never sees a code without its rank (ctl_failed)
    int32_t failed_code;
    int32_t failed_rank;
    uint32_t bar_count, bar_gen;        // central barrier of the shm transport
    uint32_t attached;                  // ranks that have mapped the segment
    int32_t pid[CTK_CTL_MAXWORLD];      // process of every rank (0: not there yet)
    uint32_t pidns[CTK_CTL_MAXWORLD];   // ... and the PID namespace it lives in (inode of /proc/self/ns/pid; 0: unknown): a pid means
                                        // something only to a process of the same namespace (one container per GPU sharing /dev/shm)
};
namespace {
constexpr uint64_t kCtlMagic = 0x334c54434b5443ull;      // "CTKCTL3": bumped whenever the layout or size of CtkCtlSeg changes (round 4 changed both under "CTKCTL2")
constexpr size_t kCtlBytes = 32768;
static_assert(sizeof(CtkCtlSeg) <= kCtlBytes, "control segment header");
static_assert(offsetof(CtkCtlSeg, failed_code) % 8 == 0 && offsetof(CtkCtlSeg, failed_rank) == offsetof(CtkCtlSeg, failed_code) + 4, "failure word");
constexpr size_t kShmSlot = (size_t)4 << 20;

inline int32_t ld32(const int32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline uint32_t ldu32(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

bool process_gone(int32_t pid)
{
    if (pid <= 0) return false;
    if (kill((pid_t)pid, 0) != 0) return errno == ESRCH;
    char path[64], buf[256];                                         // killed but not reaped yet: a zombie answers kill(pid, 0)
    snprintf(path, sizeof(path), "/proc/%d/stat", (int)pid);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *q = strrchr(buf, ')');
    return q && q[1] == ' ' && (q[2] == 'Z' || q[2] == 'X');
}

// inode of this process' PID namespace (0 if it cannot be read)
uint32_t my_pidns()
{
    static const uint32_t v = [] {
        struct stat st;
        return stat("/proc/self/ns/pid", &st) == 0 ? (uint32_t)st.st_ino : 0u;
    }();
    return v;
}

// publish a failure (first one wins); code < 0
void ctl_publish(CtkCtlSeg *s, int rank, int code)
{
    if (!s) return;
    if (code == 0) code = CTK_E_COMM;
    uint64_t expect = 0;
    const uint64_t word = (uint64_t)(uint32_t)(int32_t)code | ((uint64_t)(uint32_t)(int32_t)rank << 32);
    (void)__atomic_compare_exchange_n((uint64_t *)&s->failed_code, &expect, word, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
}
// the published failure: code (0 = none) and the rank that gave up first, read together
inline int32_t ctl_failed(const CtkCtlSeg *s, int32_t *rank)
{
    const uint64_t w = __atomic_load_n((const uint64_t *)&s->failed_code, __ATOMIC_ACQUIRE);
    if (rank) *rank = w ? (int32_t)(uint32_t)(w >> 32) : -1;
    return (int32_t)(uint32_t)w;
}

struct WaitState {
    double t0, t_live;
    uint64_t spins = 0;
    WaitState() { t0 = t_live = mono_s(); }
};
// one poll of "should this rank stop waiting?": CTK_OK = keep waiting, else the error (already set).  `what`: for the message.
int ctl_poll(ctk_comm *c, WaitState &w, const char *what)
{
    CtkCtlSeg *s = c->ctl;
    w.spins++;
    if (s) {
        int32_t fr = -1;
        const int32_t fc = ctl_failed(s, &fr);
        if (fc != 0) {
            return ctk_set_error(CTK_E_COMM, "rank %d: %s abandoned -- rank %d gave up with error %d", c->rank, what, (int)fr, (int)fc);
        }
    }
    if ((w.spins & 255u) != 0) return CTK_OK;
    const double now = mono_s();
    if (s && now - w.t_live > 0.05) {
        w.t_live = now;
        for (int r = 0; r < c->world; r++) {
            if (r == c->rank) continue;
            const int32_t pid = ld32(&s->pid[r]);
            // (a pid of another PID namespace names nothing -- or somebody else -- here: such a rank is covered by the deadline only)
            const uint32_t ns = ldu32(&s->pidns[r]);
            if (ns == 0u || ns != my_pidns()) continue;
            if (process_gone(pid)) {
                ctl_publish(s, r, CTK_E_COMM);
                return ctk_set_error(CTK_E_COMM, "rank %d: %s abandoned -- the process of rank %d (pid %d) is gone", c->rank, what, r, (int)pid);
            }
        }
    }
    if (now - w.t0 > c->timeout_s) {
        ctl_publish(s, c->rank, CTK_E_COMM);
        return ctk_set_error(CTK_E_COMM, "rank %d: %s did not complete within %.0f s (CTK_COMM_TIMEOUT_S)", c->rank, what, c->timeout_s);
    }
    return CTK_OK;
}
inline void wait_backoff(const WaitState &w)
{
    if (w.spins > 20000) nap_us(50);
    else if (w.spins > 2000) sched_yield();
}

// map (rank 0: create) the segment `name` of `bytes`; every rank registers its pid; rank 0 removes the NAME once all ranks
// hold the mapping (nothing is left in /dev/shm if a rank dies later).  Doubles as the arrival check of the ranks.
int ctl_open(ctk_comm *c, const char *name, size_t bytes, uint64_t slot, void **base)
{
    *base = nullptr;
    int fd = -1;
    const double t0 = mono_s();
    if (c->rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) close(fd); return ctk_set_error(CTK_E_COMM, "shm_open(%s) failed: %s", name, strerror(errno)); }
    } else {
        while (fd < 0) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size > bytes) {      // a segment of another library build (or a stale one of this name)
                close(fd);
                return ctk_set_error(CTK_E_COMM, "rank %d: shared-memory segment %s has %lld bytes, this library maps %zu: another build of the library, or a stale segment", c->rank, name, (long long)st.st_size, bytes);
            }
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes)) { close(fd); fd = -1; }
            if (fd < 0) {
                if (mono_s() - t0 > c->timeout_s) return ctk_set_error(CTK_E_COMM, "rank %d: shared-memory segment %s did not appear within %.0f s", c->rank, name, c->timeout_s);
                nap_us(2000);
            }
        }
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (c->rank == 0) shm_unlink(name); return ctk_set_error(CTK_E_NOMEM, "mmap of %s failed", name); }
    CtkCtlSeg *s = (CtkCtlSeg *)p;
    if (c->rank == 0) {
        memset(s, 0, sizeof(CtkCtlSeg));
        s->world = c->world; s->slot = slot;
        s->pidns[0] = my_pidns();
        s->pid[0] = (int32_t)getpid();
        __atomic_store_n(&s->magic, kCtlMagic, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n(&s->magic, __ATOMIC_ACQUIRE) != kCtlMagic) {
            if (mono_s() - t0 > c->timeout_s) { munmap(p, bytes); return ctk_set_error(CTK_E_COMM, "rank %d: segment %s was never initialised", c->rank, name); }
            nap_us(1000);
        }
        if (s->world != c->world) { munmap(p, bytes); return ctk_set_error(CTK_E_COMM, "segment %s belongs to a communicator of %d ranks, not %d", name, s->world, c->world); }
        __atomic_store_n(&s->pidns[c->rank], my_pidns(), __ATOMIC_RELEASE);
        __atomic_store_n(&s->pid[c->rank], (int32_t)getpid(), __ATOMIC_RELEASE);
    }
    __atomic_fetch_add(&s->attached, 1u, __ATOMIC_ACQ_REL);
    c->ctl = s;
    snprintf(c->ctl_name, sizeof(c->ctl_name), "%s", name);
    // everybody waits until all ranks are attached (deadline, failure flag, peers alive)
    WaitState w;
    while (ldu32(&s->attached) < (uint32_t)c->world) {
        if (int rc = ctl_poll(c, w, "the rendezvous of the ranks")) {
            if (c->rank == 0) shm_unlink(name);
            c->ctl = nullptr;
            munmap(p, bytes);
            return rc;
        }
        nap_us(200);
    }
    if (c->rank == 0) shm_unlink(name);
    *base = p;
    return CTK_OK;
}

// central barrier in the control segment (shm transport)
int ctl_barrier(ctk_comm *c)
{
    CtkCtlSeg *s = c->ctl;
    const uint32_t gen = ldu32(&s->bar_gen);
    const uint32_t n = __atomic_fetch_add(&s->bar_count, 1u, __ATOMIC_ACQ_REL) + 1u;
    if (n == (uint32_t)c->world) {
        __atomic_store_n(&s->bar_count, 0u, __ATOMIC_RELEASE);
        __atomic_fetch_add(&s->bar_gen, 1u, __ATOMIC_ACQ_REL);
        return CTK_OK;
    }
    WaitState w;
    while (ldu32(&s->bar_gen) == gen) {
        if (int rc = ctl_poll(c, w, "a barrier of the shared-memory transport")) return rc;
        wait_backoff(w);
    }
    return CTK_OK;
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// local: ranks are threads of this process
// ------------------------------------------------------------------------------------------------
struct ctk_comm_group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::atomic<bool> failed{false};
    std::atomic<int> failed_rank{-1}, failed_code{0};
    const void **pub = nullptr;         // [world] what each rank currently offers
    int attached = 0;
    double timeout_s = 120.0;
};

namespace {
void group_fail(ctk_comm_group *g, int rank, int code)
{
    std::lock_guard<std::mutex> lk(g->mu);
    if (!g->failed.load()) { g->failed_rank = rank; g->failed_code = code; }
    g->failed = true;
    g->cv.notify_all();
}
// returns false if some rank reported a failure or did not arrive in time (every waiter is released)
bool group_barrier(ctk_comm_group *g, int rank)
{
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->failed) return false;
    const uint64_t my = g->gen;
    if (++g->arrived == g->world) { g->arrived = 0; g->gen++; g->cv.notify_all(); return !g->failed; }
    const bool ok = g->cv.wait_for(lk, std::chrono::duration<double>(g->timeout_s), [&] { return g->gen != my || g->failed.load(); });
    if (!ok) {                                                        // deadline: some rank never came
        if (!g->failed.load()) { g->failed_rank = rank; g->failed_code = CTK_E_COMM; }
        g->failed = true;
        g->cv.notify_all();
        return false;
    }
    return !g->failed;
}
int group_error(ctk_comm *c)
{
    return ctk_set_error(CTK_E_COMM, "rank %d: rank %d of the in-process group gave up with error %d (or did not arrive in time)", c->rank,
                         c->group->failed_rank.load(), c->group->failed_code.load());
}
#define LOCALCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { group_fail(c->group, c->rank, CTK_E_NODEVICE); \
    return ctk_set_error(CTK_E_NODEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
#define LOCALBAR() do { if (!group_barrier(c->group, c->rank)) return group_error(c); } while (0)

inline char *shm_slot(ctk_comm *c, int r) { return (char *)c->shm + kCtlBytes + (size_t)r * c->shm_slot; }

int comm_dead_error(const ctk_comm *c)
{
    return ctk_set_error(CTK_E_COMM, "rank %d: the communicator was aborted by an earlier failure; create a new one", c->rank);
}

// after a failure: the communicator is never used again; collective kernels still in flight are made to return
void comm_retire(ctk_comm *c)
{
    if (c->dead) return;
    c->dead = true;
    if (c->kind == 2 && c->nccl) {
        if (g_rccl.ok && g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)c->nccl);
        c->nccl = nullptr;                                            // (aborted = destroyed)
    }
    if (c->stream && c->kind != 0) {                                  // bounded drain: what was enqueued behind the collective
        const double t0 = mono_s();
        while (hipStreamQuery(c->stream) == hipErrorNotReady && mono_s() - t0 < 10.0) nap_us(200);
    }
}
}  // namespace

int ctk_comm_wait(ctk_comm *c)
{
    if (!c) return ctk_set_error(CTK_E_INVALID, "ctk_comm_wait: null communicator");
    if (c->dead) return comm_dead_error(c);
    if (c->world == 1) { HIPCHK(hipStreamSynchronize(c->stream)); return CTK_OK; }
    WaitState w;
    for (;;) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) {
            ctk_comm_abort(c, CTK_E_NODEVICE);
            return ctk_set_error(CTK_E_NODEVICE, "rank %d: hipStreamQuery: %s", c->rank, hipGetErrorString(e));
        }
        int rc = CTK_OK;
        if (c->kind == 0) {
            if (c->group->failed.load()) rc = group_error(c);
            else if ((++w.spins & 1023u) == 0 && mono_s() - w.t0 > c->timeout_s) {
                group_fail(c->group, c->rank, CTK_E_COMM);
                rc = ctk_set_error(CTK_E_COMM, "rank %d: the stream did not drain within %.0f s", c->rank, c->timeout_s);
            }
        } else {
            rc = ctl_poll(c, w, "a collective of the time-shard path");
        }
        if (rc != CTK_OK) { comm_retire(c); return rc; }
        wait_backoff(w);
    }
    // a failure published while this rank's stream happened to drain (e.g. its last collective completed before the other rank
    // gave up): report it at the next wait, not here -- what this rank computed so far is consistent.
    return CTK_OK;
}

// Guarded wait for a word in pinned host memory that a kernel on the communicator's stream writes (a stamp): returns as soon as
// the word is there -- kernels enqueued BEHIND the writer keep running -- or with the error of whoever gave up (same rules as
// ctk_comm_wait).  A stream that drains without the stamp is an error.
int ctk_comm_wait_word(ctk_comm *c, const volatile uint32_t *word, uint32_t stamp)
{
    if (!c || !word) return ctk_set_error(CTK_E_INVALID, "ctk_comm_wait_word: null argument");
    if (c->dead) return comm_dead_error(c);
    WaitState w;
    for (uint64_t spins = 1;; spins++) {
        if (*word == stamp) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return CTK_OK; }
        if ((spins & 0x3fffu) != 0) continue;
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) {
            if (*word == stamp) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return CTK_OK; }
            return ctk_set_error(CTK_E_INTERNAL, "rank %d: the stream drained without the expected stamp", c->rank);
        }
        if (e != hipErrorNotReady) {
            ctk_comm_abort(c, CTK_E_NODEVICE);
            return ctk_set_error(CTK_E_NODEVICE, "rank %d: hipStreamQuery: %s", c->rank, hipGetErrorString(e));
        }
        if (c->world == 1) continue;
        int rc = CTK_OK;
        if (c->kind == 0) {
            if (c->group->failed.load()) rc = group_error(c);
            else if (mono_s() - w.t0 > c->timeout_s) {
                group_fail(c->group, c->rank, CTK_E_COMM);
                rc = ctk_set_error(CTK_E_COMM, "rank %d: a kernel of the time-shard path did not report within %.0f s", c->rank, c->timeout_s);
            }
        } else {
            w.spins |= 255u;                                            // (every call here is already one in 2^14 polls: look at everything)
            rc = ctl_poll(c, w, "a collective of the time-shard path");
        }
        if (rc != CTK_OK) { comm_retire(c); return rc; }
    }
}

void ctk_comm_abort(ctk_comm *c, int code)
{
    if (!c) return;
    if (code >= 0) code = CTK_E_INTERNAL;
    if (c->kind == 0) { if (c->group) group_fail(c->group, c->rank, code); c->dead = c->world > 1; return; }
    if (c->world == 1) return;
    ctl_publish(c->ctl, c->rank, code);
    comm_retire(c);
}

// ------------------------------------------------------------------------------------------------
// the two primitives
// ------------------------------------------------------------------------------------------------
static int comm_shift_impl(ctk_comm *c, int dir, const void *send, size_t sbytes, void *recv, size_t rbytes)
{
    if (!c || (dir != 1 && dir != -1)) return ctk_set_error(CTK_E_INVALID, "ctk_comm_shift: bad arguments");
    if (c->dead) return comm_dead_error(c);
    const int dst = c->rank + dir, src = c->rank - dir;
    const bool has_dst = dst >= 0 && dst < c->world && sbytes > 0, has_src = src >= 0 && src < c->world && rbytes > 0;
    c->n_shift++;
    if (c->world == 1) return CTK_OK;
    if (c->kind == 2) {
        NCCLCHK(g_rccl.GroupStart());
        if (has_dst) NCCLCHK(g_rccl.Send(send, sbytes, ncclChar, dst, (ncclComm_t)c->nccl, c->stream));
        if (has_src) NCCLCHK(g_rccl.Recv(recv, rbytes, ncclChar, src, (ncclComm_t)c->nccl, c->stream));
        NCCLCHK(g_rccl.GroupEnd());
        return CTK_OK;
    }
    if (c->kind == 0) {
        LOCALCHK(hipStreamSynchronize(c->stream));                      // what is offered is complete
        c->group->pub[c->rank] = send;
        LOCALBAR();
        if (has_src) {
            LOCALCHK(hipMemcpyAsync(recv, c->group->pub[src], rbytes, hipMemcpyDefault, c->stream));
            LOCALCHK(hipStreamSynchronize(c->stream));
        }
        LOCALBAR();                                                     // everybody has read: buffers may be reused
        return CTK_OK;
    }
    // shm: chunks of one slot
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t slot = c->shm_slot;
    size_t so = 0, ro = 0;
    // the number of chunks is agreed through the slot header (first 8 bytes of every slot = bytes still to come)
    for (;;) {
        const size_t sn = has_dst ? (sbytes - so < slot - 8 ? sbytes - so : slot - 8) : 0;
        char *mine = shm_slot(c, c->rank);
        *(uint64_t *)mine = has_dst ? (uint64_t)(sbytes - so) : 0;
        if (sn) HIPCHK(hipMemcpy(mine + 8, (const char *)send + so, sn, hipMemcpyDefault));
        so += sn;
        if (int rc = ctl_barrier(c)) return rc;
        if (has_src && ro < rbytes) {
            const char *peer = shm_slot(c, src);
            const size_t left = (size_t) * (const uint64_t *)peer;
            const size_t rn = left < slot - 8 ? left : slot - 8;
            if (rn) HIPCHK(hipMemcpy((char *)recv + ro, peer + 8, rn < rbytes - ro ? rn : rbytes - ro, hipMemcpyDefault));
            ro += rn;
        }
        // continue while ANY rank still has data: everyone publishes its remaining bytes, everyone reads all of them
        uint64_t any = 0;
        for (int r = 0; r < c->world; r++) { const uint64_t left = *(const uint64_t *)shm_slot(c, r); if (left > slot - 8) any = 1; }
        if (int rc = ctl_barrier(c)) return rc;
        if (!any) break;
    }
    return CTK_OK;
}

static int comm_allgather_impl(ctk_comm *c, const void *send, void *recv, size_t nbytes)
{
    if (!c || (nbytes && (!send || !recv))) return ctk_set_error(CTK_E_INVALID, "ctk_comm_allgather: bad arguments");
    if (c->dead) return comm_dead_error(c);
    c->n_allgather++;
    if (nbytes == 0) return CTK_OK;
    if (c->world == 1) {
        if ((const char *)recv != (const char *)send) HIPCHK(hipMemcpyAsync(recv, send, nbytes, hipMemcpyDefault, c->stream));
        return CTK_OK;
    }
    if (c->kind == 2) {
        NCCLCHK(g_rccl.AllGather(send, recv, nbytes, ncclChar, (ncclComm_t)c->nccl, c->stream));
        return CTK_OK;
    }
    if (c->kind == 0) {
        LOCALCHK(hipStreamSynchronize(c->stream));
        c->group->pub[c->rank] = send;
        LOCALBAR();
        for (int r = 0; r < c->world; r++) {
            char *dst = (char *)recv + (size_t)r * nbytes;
            if (dst != (const char *)c->group->pub[r]) LOCALCHK(hipMemcpyAsync(dst, c->group->pub[r], nbytes, hipMemcpyDefault, c->stream));
        }
        LOCALCHK(hipStreamSynchronize(c->stream));
        LOCALBAR();
        return CTK_OK;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t slot = c->shm_slot;
    for (size_t off = 0; off < nbytes; off += slot) {
        const size_t n = nbytes - off < slot ? nbytes - off : slot;
        HIPCHK(hipMemcpy(shm_slot(c, c->rank), (const char *)send + off, n, hipMemcpyDefault));
        if (int rc = ctl_barrier(c)) return rc;
        for (int r = 0; r < c->world; r++) HIPCHK(hipMemcpy((char *)recv + (size_t)r * nbytes + off, shm_slot(c, r), n, hipMemcpyDefault));
        if (int rc = ctl_barrier(c)) return rc;
    }
    return CTK_OK;
}

// "A failed communicator stays failed": whatever made a primitive give up on this rank (a HIP error inside the shared-memory
// transport, a barrier that reported another rank's failure, an RCCL error) is published to the other ranks and retires the
// communicator -- not only when the caller is the time-shard entry.  Argument errors leave it alone.
static int comm_fail_sticky(ctk_comm *c, int rc)
{
    if (rc != CTK_OK && rc != CTK_E_INVALID && c && !c->dead && c->world > 1) {
        // (the abort may set its own message: keep the one that explains the failure)
        std::string keep = ctk_last_error();
        ctk_comm_abort(c, rc);
        ctk_set_error(rc, "%s", keep.c_str());
    }
    return rc;
}
int ctk_comm_shift(ctk_comm *c, int dir, const void *send, size_t sbytes, void *recv, size_t rbytes)
{
    return comm_fail_sticky(c, comm_shift_impl(c, dir, send, sbytes, recv, rbytes));
}
int ctk_comm_allgather(ctk_comm *c, const void *send, void *recv, size_t nbytes)
{
    return comm_fail_sticky(c, comm_allgather_impl(c, send, recv, nbytes));
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int ctk_device_of(ctk_handle *h);      // ctk_api.hip

static int comm_new(ctk_handle *h, int rank, int world, int kind, ctk_comm **out)
{
    if (!h || !out || world < 1 || rank < 0 || rank >= world) return ctk_set_error(CTK_E_INVALID, "ctk_comm_init: bad handle, rank %d or world %d", rank, world);
    if (kind != 0 && world > CTK_CTL_MAXWORLD) return ctk_set_error(CTK_E_RANGE, "ctk_comm_init: at most %d ranks", CTK_CTL_MAXWORLD);
    ctk_comm *c = new (std::nothrow) ctk_comm();
    if (!c) return ctk_set_error(CTK_E_NOMEM, "ctk_comm_init: out of memory");
    c->rank = rank; c->world = world; c->kind = kind;
    c->device = ctk_device_of(h);
    c->stream = (hipStream_t)ctk_stream(h);
    c->timeout_s = default_timeout_s();
    *out = c;
    return CTK_OK;
}

extern "C" int ctk_comm_unique_id(void *id)
{
    if (!id) return ctk_set_error(CTK_E_INVALID, "ctk_comm_unique_id: null pointer");
    if (int rc = rccl_load()) return rc;
    static_assert(sizeof(ncclUniqueId) == CTK_COMM_ID_BYTES, "CTK_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return CTK_OK;
}

extern "C" int ctk_comm_init_rccl(ctk_handle *h, const void *id, int rank, int world, ctk_comm **out)
{
    if (!id) return ctk_set_error(CTK_E_INVALID, "ctk_comm_init_rccl: null id");
    if (out) *out = nullptr;
    if (int rc = rccl_load()) return rc;
    ctk_comm *c = nullptr;
    if (int rc = comm_new(h, rank, world, 2, &c)) return rc;
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    if (world > 1) {
        // control segment, named after the unique id (one per communicator by construction).  Attaching to it is the arrival
        // check: ncclCommInitRank is only entered once every rank is known to be there.
        uint64_t hsh = 1469598103934665603ull;
        for (size_t i = 0; i < sizeof(u); i++) hsh = (hsh ^ ((const unsigned char *)&u)[i]) * 1099511628211ull;
        char name[96];
        snprintf(name, sizeof(name), "/ctk_ctl_%016llx", (unsigned long long)hsh);
        void *base = nullptr;
        if (int rc = ctl_open(c, name, kCtlBytes, 0, &base)) { delete c; return rc; }
    }
    // ncclCommInitRank has no deadline of its own: it runs on a helper thread, this thread watches the clock and the ranks
    struct Job { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t r = ncclSuccess; ncclComm_t comm = nullptr; };
    auto job = std::make_shared<Job>();
    const int device = c->device;
    std::thread([job, u, rank, world, device]() {
        ncclComm_t cm = nullptr;
        ncclResult_t r = hipSetDevice(device) == hipSuccess ? g_rccl.CommInitRank(&cm, world, u, rank) : ncclUnhandledCudaError;
        std::lock_guard<std::mutex> lk(job->mu);
        job->r = r; job->comm = cm; job->done = true;
        job->cv.notify_all();
    }).detach();
    {
        WaitState w;
        std::unique_lock<std::mutex> lk(job->mu);
        while (!job->done) {
            job->cv.wait_for(lk, std::chrono::milliseconds(20));
            if (job->done) break;
            w.spins |= 255u;                                            // (every poll looks at the clock and the peers)
            lk.unlock();
            const int rc = world > 1 ? ctl_poll(c, w, "ncclCommInitRank") : CTK_OK;
            lk.lock();
            if (rc != CTK_OK && !job->done) {
                if (c->ctl) munmap(c->ctl, kCtlBytes);
                delete c;                                               // (the helper thread is abandoned with its call)
                return rc;
            }
        }
    }
    if (job->r != ncclSuccess) {
        ctl_publish(c->ctl, rank, CTK_E_COMM);
        const int rc = ctk_set_error(CTK_E_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(job->r));
        if (c->ctl) munmap(c->ctl, kCtlBytes);
        delete c;
        return rc;
    }
    c->nccl = job->comm;
    *out = c;
    return CTK_OK;
}

extern "C" int ctk_comm_group_create(int world, ctk_comm_group **out)
{
    if (!out || world < 1 || world > 4096) return ctk_set_error(CTK_E_INVALID, "ctk_comm_group_create: world %d", world);
    ctk_comm_group *g = new (std::nothrow) ctk_comm_group();
    if (!g) return ctk_set_error(CTK_E_NOMEM, "out of memory");
    g->world = world;
    g->timeout_s = default_timeout_s();
    g->pub = (const void **)calloc((size_t)world, sizeof(void *));
    if (!g->pub) { delete g; return ctk_set_error(CTK_E_NOMEM, "out of memory"); }
    *out = g;
    return CTK_OK;
}

extern "C" void ctk_comm_group_destroy(ctk_comm_group *g)
{
    if (!g) return;
    free(g->pub);
    delete g;
}

extern "C" int ctk_comm_init_local(ctk_handle *h, ctk_comm_group *g, int rank, ctk_comm **out)
{
    if (!g) return ctk_set_error(CTK_E_INVALID, "ctk_comm_init_local: null group");
    ctk_comm *c = nullptr;
    if (int rc = comm_new(h, rank, g->world, 0, &c)) return rc;
    c->group = g;
    c->timeout_s = g->timeout_s;
    *out = c;
    return CTK_OK;
}

extern "C" int ctk_comm_init_shm(ctk_handle *h, const char *name, int rank, int world, ctk_comm **out)
{
    if (!name || !*name || strlen(name) > 80) return ctk_set_error(CTK_E_INVALID, "ctk_comm_init_shm: bad segment name");
    if (out) *out = nullptr;
    ctk_comm *c = nullptr;
    if (int rc = comm_new(h, rank, world, 1, &c)) return rc;
    *out = nullptr;
    snprintf(c->shm_name, sizeof(c->shm_name), "%s%s", name[0] == '/' ? "" : "/", name);
    c->shm_slot = kShmSlot;
    c->shm_bytes = kCtlBytes + (size_t)world * kShmSlot;
    void *base = nullptr;
    if (int rc = ctl_open(c, c->shm_name, c->shm_bytes, kShmSlot, &base)) { delete c; return rc; }
    c->shm = base;
    *out = c;
    return CTK_OK;
}

extern "C" void ctk_comm_destroy(ctk_comm *c)
{
    if (!c) return;
    if (c->kind == 2 && c->nccl && g_rccl.ok && !c->dead) (void)g_rccl.CommDestroy((ncclComm_t)c->nccl);
    if (c->kind == 1 && c->shm) munmap(c->shm, c->shm_bytes);
    else if (c->kind == 2 && c->ctl) munmap(c->ctl, kCtlBytes);
    if (c->scratch) (void)hipFree(c->scratch);
    delete c;
}

extern "C" int ctk_comm_rank(const ctk_comm *c) { return c ? c->rank : -1; }
extern "C" int ctk_comm_world(const ctk_comm *c) { return c ? c->world : -1; }
extern "C" int ctk_comm_ops(const ctk_comm *c, int64_t *shifts, int64_t *allgathers)
{
    if (!c) return ctk_set_error(CTK_E_INVALID, "null communicator");
    if (shifts) *shifts = (int64_t)c->n_shift;
    if (allgathers) *allgathers = (int64_t)c->n_allgather;
    return CTK_OK;
}

extern "C" int ctk_comm_set_timeout(ctk_comm *c, double seconds)
{
    if (!c || !(seconds > 0)) return ctk_set_error(CTK_E_INVALID, "ctk_comm_set_timeout: null communicator or non-positive time");
    c->timeout_s = seconds;
    if (c->kind == 0 && c->group) c->group->timeout_s = seconds;
    return CTK_OK;
}

// 0: fine; else the published failure (code of the rank that gave up first, and that rank)
extern "C" int ctk_comm_failed(const ctk_comm *c, int *code, int *rank)
{
    if (!c) return ctk_set_error(CTK_E_INVALID, "null communicator");
    int fc = 0, fr = -1;
    if (c->kind == 0 && c->group) { if (c->group->failed.load()) { fc = c->group->failed_code.load(); fr = c->group->failed_rank.load(); if (!fc) fc = CTK_E_COMM; } }
    else if (c->ctl) { int32_t r32 = -1; fc = ctl_failed(c->ctl, &r32); fr = r32; }
    if (code) *code = fc;
    if (rank) *rank = fr;
    return CTK_OK;
}

extern "C" int ctk_comm_abort_rank(ctk_comm *c, int code)
{
    if (!c) return ctk_set_error(CTK_E_INVALID, "null communicator");
    ctk_comm_abort(c, code);
    return CTK_OK;
}

// small host payloads (timings, counts, checksums): staged through a device scratch so that every transport can carry them
static int allgather_host_impl(ctk_comm *c, const void *send, void *recv, size_t nbytes)
{
    if (!c || !send || !recv || nbytes == 0 || nbytes > 4096) return ctk_set_error(CTK_E_INVALID, "ctk_comm_allgather_host: 1..4096 bytes per rank");
    if (c->dead) return comm_dead_error(c);
    HIPCHK(hipSetDevice(c->device));
    const size_t need = (size_t)4096 * (size_t)(c->world + 1);
    if (c->scratch_cap < need) {
        if (int rc = ctk_comm_wait(c)) return rc;                       // (hipMalloc / hipFree wait for the device)
        if (c->scratch) (void)hipFree(c->scratch);
        c->scratch = nullptr; c->scratch_cap = 0;
        HIPCHK(hipMalloc(&c->scratch, need));
        c->scratch_cap = need;
    }
    char *d = (char *)c->scratch;
    HIPCHK(hipMemcpyAsync(d, send, nbytes, hipMemcpyHostToDevice, c->stream));
    if (int rc = ctk_comm_wait(c)) return rc;
    if (int rc = ctk_comm_allgather(c, d, d + 4096, nbytes)) return rc;
    if (int rc = ctk_comm_wait(c)) return rc;                           // guarded: only then the (blocking) copy to pageable memory
    HIPCHK(hipMemcpy(recv, d + 4096, nbytes * (size_t)c->world, hipMemcpyDeviceToHost));
    return CTK_OK;
}

extern "C" int ctk_comm_allgather_host(ctk_comm *c, const void *send, void *recv, size_t nbytes)
{
    return comm_fail_sticky(c, allgather_host_impl(c, send, recv, nbytes));
}

extern "C" int ctk_comm_barrier(ctk_comm *c)
{
    uint64_t x = 1, all[CTK_CTL_MAXWORLD];
    if (!c || c->world > CTK_CTL_MAXWORLD) return ctk_set_error(CTK_E_INVALID, "ctk_comm_barrier: bad communicator");
    return ctk_comm_allgather_host(c, &x, all, 8);
}
