// ctk_comm.hip -- transports of the time-sharded path's communicator (see ctk_comm.h).  Host code only.
#include "ctk_comm.h"
#include "../../include/contrack_hip.h"

#include <rccl/rccl.h>          // types and prototypes only: librccl.so is dlopen'ed, never linked

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
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

// ------------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
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

int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.ok) return CTK_OK;
    const char *names[] = {getenv("CTK_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return ctk_set_error(CTK_E_NODEVICE, "librccl.so not found (%s): set CTK_RCCL_LIB", dlerror());
#define SYM(field, name)                                                                                       \
    g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.lib, name);                                            \
    if (!g_rccl.field) return ctk_set_error(CTK_E_NODEVICE, "librccl.so lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllGather, "ncclAllGather"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.ok = true;
    return CTK_OK;
}
#define NCCLCHK(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return ctk_set_error(CTK_E_NODEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)
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
    bool failed = false;
    const void **pub = nullptr;         // [world] what each rank currently offers
    int attached = 0;
};

namespace {
// returns false if some rank reported a failure (every waiter is released)
bool group_barrier(ctk_comm_group *g)
{
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->failed) return false;
    const uint64_t my = g->gen;
    if (++g->arrived == g->world) { g->arrived = 0; g->gen++; g->cv.notify_all(); return !g->failed; }
    g->cv.wait(lk, [&] { return g->gen != my || g->failed; });
    return !g->failed;
}
void group_fail(ctk_comm_group *g)
{
    std::lock_guard<std::mutex> lk(g->mu);
    g->failed = true;
    g->cv.notify_all();
}
#define LOCALCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { group_fail(c->group); \
    return ctk_set_error(CTK_E_NODEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
#define LOCALBAR() do { if (!group_barrier(c->group)) return ctk_set_error(CTK_E_STATE, "another rank of the in-process group failed"); } while (0)
}  // namespace

// ------------------------------------------------------------------------------------------------
// shm: ranks are processes of this node, data staged through a shared-memory segment
// ------------------------------------------------------------------------------------------------
namespace {
struct ShmHeader {
    uint64_t magic;
    int32_t world, pad;
    uint64_t slot;                      // bytes per rank slot
    pthread_barrier_t bar;
};
constexpr uint64_t kShmMagic = 0x4d48534b5443ull;   // "CTKSHM"
constexpr size_t kShmSlot = (size_t)4 << 20;
inline char *shm_slot(ctk_comm *c, int r) { return (char *)c->shm + 4096 + (size_t)r * c->shm_slot; }
inline int shm_barrier(ctk_comm *c)
{
    const int rc = pthread_barrier_wait(&((ShmHeader *)c->shm)->bar);
    return (rc == 0 || rc == PTHREAD_BARRIER_SERIAL_THREAD) ? CTK_OK : ctk_set_error(CTK_E_INTERNAL, "pthread_barrier_wait: %d", rc);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// the two primitives
// ------------------------------------------------------------------------------------------------
int ctk_comm_shift(ctk_comm *c, int dir, const void *send, size_t sbytes, void *recv, size_t rbytes)
{
    if (!c || (dir != 1 && dir != -1)) return ctk_set_error(CTK_E_INVALID, "ctk_comm_shift: bad arguments");
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
        if (int rc = shm_barrier(c)) return rc;
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
        if (int rc = shm_barrier(c)) return rc;
        if (!any) break;
    }
    return CTK_OK;
}

int ctk_comm_allgather(ctk_comm *c, const void *send, void *recv, size_t nbytes)
{
    if (!c || (nbytes && (!send || !recv))) return ctk_set_error(CTK_E_INVALID, "ctk_comm_allgather: bad arguments");
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
        if (int rc = shm_barrier(c)) return rc;
        for (int r = 0; r < c->world; r++) HIPCHK(hipMemcpy((char *)recv + (size_t)r * nbytes + off, shm_slot(c, r), n, hipMemcpyDefault));
        if (int rc = shm_barrier(c)) return rc;
    }
    return CTK_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int ctk_device_of(ctk_handle *h);      // ctk_api.hip

static int comm_new(ctk_handle *h, int rank, int world, int kind, ctk_comm **out)
{
    if (!h || !out || world < 1 || rank < 0 || rank >= world) return ctk_set_error(CTK_E_INVALID, "ctk_comm_init: bad handle, rank %d or world %d", rank, world);
    ctk_comm *c = new (std::nothrow) ctk_comm();
    if (!c) return ctk_set_error(CTK_E_NOMEM, "ctk_comm_init: out of memory");
    c->rank = rank; c->world = world; c->kind = kind;
    c->device = ctk_device_of(h);
    c->stream = (hipStream_t)ctk_stream(h);
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
    if (int rc = rccl_load()) return rc;
    ctk_comm *c = nullptr;
    if (int rc = comm_new(h, rank, world, 2, &c)) return rc;
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return ctk_set_error(CTK_E_NODEVICE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r)); }
    c->nccl = comm;
    *out = c;
    return CTK_OK;
}

extern "C" int ctk_comm_group_create(int world, ctk_comm_group **out)
{
    if (!out || world < 1 || world > 4096) return ctk_set_error(CTK_E_INVALID, "ctk_comm_group_create: world %d", world);
    ctk_comm_group *g = new (std::nothrow) ctk_comm_group();
    if (!g) return ctk_set_error(CTK_E_NOMEM, "out of memory");
    g->world = world;
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
    *out = c;
    return CTK_OK;
}

extern "C" int ctk_comm_init_shm(ctk_handle *h, const char *name, int rank, int world, ctk_comm **out)
{
    if (!name || !*name || strlen(name) > 80) return ctk_set_error(CTK_E_INVALID, "ctk_comm_init_shm: bad segment name");
    ctk_comm *c = nullptr;
    if (int rc = comm_new(h, rank, world, 1, &c)) return rc;
    snprintf(c->shm_name, sizeof(c->shm_name), "%s%s", name[0] == '/' ? "" : "/", name);
    c->shm_slot = kShmSlot;
    c->shm_bytes = 4096 + (size_t)world * kShmSlot;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(c->shm_name);
        fd = shm_open(c->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->shm_bytes) != 0) { if (fd >= 0) close(fd); delete c; return ctk_set_error(CTK_E_NOMEM, "shm_open(%s) failed", name); }
    } else {
        for (int tries = 0; tries < 6000 && fd < 0; tries++) {          // up to 60 s for rank 0 to create the segment
            fd = shm_open(c->shm_name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->shm_bytes)) { close(fd); fd = -1; }
            if (fd < 0) { struct timespec ts = {0, 10000000}; nanosleep(&ts, nullptr); }
        }
        if (fd < 0) { delete c; return ctk_set_error(CTK_E_STATE, "shared-memory segment %s did not appear", name); }
    }
    c->shm = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->shm == MAP_FAILED) { c->shm = nullptr; delete c; return ctk_set_error(CTK_E_NOMEM, "mmap of %s failed", name); }
    ShmHeader *hd = (ShmHeader *)c->shm;
    if (rank == 0) {
        pthread_barrierattr_t at;
        pthread_barrierattr_init(&at);
        pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&hd->bar, &at, (unsigned)world);
        pthread_barrierattr_destroy(&at);
        hd->world = world; hd->slot = kShmSlot;
        __atomic_store_n(&hd->magic, kShmMagic, __ATOMIC_RELEASE);
    } else {
        for (int tries = 0; tries < 6000 && __atomic_load_n(&hd->magic, __ATOMIC_ACQUIRE) != kShmMagic; tries++) {
            struct timespec ts = {0, 10000000};
            nanosleep(&ts, nullptr);
        }
        if (__atomic_load_n(&hd->magic, __ATOMIC_ACQUIRE) != kShmMagic || hd->world != world) {
            munmap(c->shm, c->shm_bytes); delete c;
            return ctk_set_error(CTK_E_STATE, "shared-memory segment %s was not initialised for %d ranks", name, world);
        }
    }
    *out = c;
    return CTK_OK;
}

extern "C" void ctk_comm_destroy(ctk_comm *c)
{
    if (!c) return;
    if (c->kind == 2 && c->nccl && g_rccl.ok) (void)g_rccl.CommDestroy((ncclComm_t)c->nccl);
    if (c->kind == 1 && c->shm) {
        munmap(c->shm, c->shm_bytes);
        if (c->rank == 0) shm_unlink(c->shm_name);
    }
    delete c;
}

// a rank gives up (error outside the communicator's own operations): ranks of an in-process group waiting for it are released
void ctk_comm_abort(ctk_comm *c)
{
    if (c && c->kind == 0 && c->group) group_fail(c->group);
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

// small host payloads (timings, counts): staged through a device scratch so that every transport can carry them
extern "C" int ctk_comm_allgather_host(ctk_comm *c, const void *send, void *recv, size_t nbytes)
{
    if (!c || !send || !recv || nbytes == 0 || nbytes > 4096) return ctk_set_error(CTK_E_INVALID, "ctk_comm_allgather_host: 1..4096 bytes per rank");
    HIPCHK(hipSetDevice(c->device));
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, nbytes * (size_t)(c->world + 1)));
    int rc = CTK_OK;
    hipError_t e = hipMemcpyAsync(d, send, nbytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) rc = ctk_comm_allgather(c, d, (char *)d + nbytes, nbytes);
    if (e == hipSuccess && rc == CTK_OK) e = hipMemcpyAsync(recv, (char *)d + nbytes, nbytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && rc == CTK_OK) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return ctk_set_error(CTK_E_NODEVICE, "ctk_comm_allgather_host: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int ctk_comm_barrier(ctk_comm *c)
{
    uint64_t x = 1, all[4096 / 8];
    if (!c || c->world > (int)(sizeof(all) / 8)) return ctk_set_error(CTK_E_INVALID, "ctk_comm_barrier: bad communicator");
    return ctk_comm_allgather_host(c, &x, all, 8);
}
