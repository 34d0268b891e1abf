// ba_comm.cpp -- RCCL inside the library: the one exchange step of the sharded LM trial (all-reduce of the packed reduced
// camera system over xGMI) is issued by the C layer itself on the solver's stream, not by a host-language
// callback; the distributed factor (BA_DIST_FACTOR) adds ncclReduceScatter and ncclBroadcast.  The reference has no counterpart (single process, no communication); north_star asks for "a thin C-ABI host layer
// ... with an RCCL all-reduce over xGMI on the reduced camera blocks".
//
// librccl.so.1 is opened on first use (dlopen), so that single-GPU users -- the executables, the tests -- neither load nor link
// it; in a process that already holds an RCCL (PyTorch) the loader hands back that copy (same soname).
#include "ba_internal.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>
#include <cstdlib>
#include <string>
#include <thread>
#include <unistd.h>

namespace {

struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi &api()
{
    static RcclApi a;
    if (a.h || a.ok) return a;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        a.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (a.h) break;
    }
    if (!a.h) {
        fprintf(stderr, "ba_mi355x: cannot open librccl.so.1: %s\n", dlerror());
        return a;
    }
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.h, "ncclCommInitRank");
    a.AllReduce = (decltype(a.AllReduce))dlsym(a.h, "ncclAllReduce");
    a.Broadcast = (decltype(a.Broadcast))dlsym(a.h, "ncclBroadcast");
    a.ReduceScatter = (decltype(a.ReduceScatter))dlsym(a.h, "ncclReduceScatter");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.h, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.h, "ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.Broadcast && a.ReduceScatter && a.CommDestroy && a.GetErrorString;
    if (!a.ok) fprintf(stderr, "ba_mi355x: librccl lacks an expected symbol\n");
    return a;
}

int fail(const char *what, ncclResult_t r)
{
    fprintf(stderr, "ba_mi355x: %s failed: %s\n", what, api().GetErrorString ? api().GetErrorString(r) : "?");
    return BA_ERR_COMM;
}

} // namespace

static_assert(sizeof(ncclUniqueId) == BA_COMM_ID_BYTES, "BA_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

int ba_rccl_init(void **comm_out, const void *id128, int rank, int world)
{
    RcclApi &a = api();
    if (!a.ok) return BA_ERR_COMM;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t r = a.CommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) return fail("ncclCommInitRank", r);
    *comm_out = (void *)c;
    return BA_OK;
}

void ba_rccl_destroy(void *comm)
{
    if (comm && api().ok) (void)api().CommDestroy((ncclComm_t)comm);
}

// in place, on `stream`; f64: 1 = double, 0 = float; op 0 sum, 1 max
int ba_rccl_allreduce(void *comm, void *buf, size_t count, int f64, int op, void *stream)
{
    RcclApi &a = api();
    if (!a.ok || !comm) return BA_ERR_COMM;
    const ncclResult_t r = a.AllReduce(buf, buf, count, f64 ? ncclDouble : ncclFloat, op == 0 ? ncclSum : ncclMax, (ncclComm_t)comm, (hipStream_t)stream);
    return r == ncclSuccess ? BA_OK : fail("ncclAllReduce", r);
}

// in place, on `stream`: `count` scalars from rank `root` to every rank (the distributed factor's panels)
int ba_rccl_broadcast(void *comm, void *buf, size_t count, int f64, int root, void *stream)
{
    RcclApi &a = api();
    if (!a.ok || !comm) return BA_ERR_COMM;
    const ncclResult_t r = a.Broadcast(buf, buf, count, f64 ? ncclDouble : ncclFloat, root, (ncclComm_t)comm, (hipStream_t)stream);
    return r == ncclSuccess ? BA_OK : fail("ncclBroadcast", r);
}

// in place (ncclReduceScatter's in-place form: recvbuff = sendbuff + rank * count): buf holds world chunks of `count` scalars; on
// return chunk `rank` of THIS rank's buffer is the sum over the ranks of their chunk `rank` (the other chunks are unspecified)
int ba_rccl_reduce_scatter(void *comm, void *buf, size_t count, int f64, int rank, void *stream)
{
    RcclApi &a = api();
    if (!a.ok || !comm) return BA_ERR_COMM;
    const size_t sz = f64 ? 8 : 4;
    const ncclResult_t r = a.ReduceScatter(buf, (char *)buf + (size_t)rank * count * sz, count, f64 ? ncclDouble : ncclFloat, ncclSum, (ncclComm_t)comm,
                                           (hipStream_t)stream);
    return r == ncclSuccess ? BA_OK : fail("ncclReduceScatter", r);
}

extern "C" {

int ba_comm_unique_id(void *id_out)
{
    if (!id_out) return BA_ERR_ARG;
    RcclApi &a = api();
    if (!a.ok) return BA_ERR_COMM;
    ncclUniqueId id;
    const ncclResult_t r = a.GetUniqueId(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    memcpy(id_out, &id, sizeof id);
    return BA_OK;
}

// Rendezvous through a file for processes started by hand (the executables with BA_WORLD / BA_RANK): rank 0 creates the id and
// publishes it with an atomic rename, the others wait for the file (BA_COMM_WAIT_S seconds, default 60).  File = the 128-byte id +
// an 8-byte launch nonce (a hash of the environment variable BA_COMM_NONCE; 0 when unset).
// A file left behind by an earlier run must never be taken for this run's: rank 0 removes whatever is there before it creates the
// id and removes its own file again once the communicator stands (ba_comm_id_file_done: ncclCommInitRank is collective, every rank
// has read the id by then), so a stale file only survives a run that died in between.  A reader refuses a file whose nonce is not
// its own.  Without a nonce it takes a file that was written after THIS PROCESS started (less two seconds; the start = the moment
// the library was loaded -- not the moment of this call, which comes after the problem has been loaded and the solver created,
// seconds later on a big shard: ADVICE r3) at once; an older file -- rank 0 started earlier by hand, or a dead run's leftover --
// only after it has stayed unchanged for BA_COMM_GRACE_S (default 5) seconds of polling, since this launch's rank 0 would have
// removed a leftover at its own start.  What remains: a leftover AND a rank 0 that starts more than the grace period after a
// reader -- set BA_COMM_NONCE per launch to rule that out (INTEGRATION.md does).  Temporary file: O_CREAT | O_EXCL | O_NOFOLLOW, 0600.
static unsigned long long ba_comm_nonce()
{
    const char *e = getenv("BA_COMM_NONCE");
    if (!e || !*e) return 0;
    unsigned long long h = 1469598103934665603ull; // FNV-1a
    for (; *e; e++) { h ^= (unsigned char)*e; h *= 1099511628211ull; }
    return h ? h : 1;
}

static struct timespec ba_now_realtime()
{
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    return t;
}
static const struct timespec g_process_start = ba_now_realtime(); // (static initialiser: runs when the library is loaded)

static double env_seconds(const char *name, double dflt)
{
    const char *e = getenv(name);
    const double v = e && *e ? atof(e) : dflt;
    return v > 0 ? v : dflt;
}

int ba_comm_id_via_file(const char *path, int rank, void *id_out)
{
    if (!path || !id_out || rank < 0) return BA_ERR_ARG;
    const unsigned long long nonce = ba_comm_nonce();
    if (rank == 0) {
        (void)unlink(path); // an earlier run's id
        int rc = ba_comm_unique_id(id_out);
        if (rc) return rc;
        const std::string tmp = std::string(path) + ".tmp";
        (void)unlink(tmp.c_str());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
        if (fd < 0) return BA_ERR_FILE;
        unsigned char buf[BA_COMM_ID_BYTES + 8];
        memcpy(buf, id_out, BA_COMM_ID_BYTES);
        memcpy(buf + BA_COMM_ID_BYTES, &nonce, 8);
        const bool ok = write(fd, buf, sizeof buf) == (ssize_t)sizeof buf;
        (void)close(fd);
        if (!ok || rename(tmp.c_str(), path) != 0) { (void)unlink(tmp.c_str()); return BA_ERR_FILE; }
        return BA_OK;
    }
    const double wait_s = env_seconds("BA_COMM_WAIT_S", 60.0), grace_s = env_seconds("BA_COMM_GRACE_S", 5.0);
    const struct timespec t0 = g_process_start;
    const auto begin = std::chrono::steady_clock::now();
    unsigned char seen[BA_COMM_ID_BYTES + 8];
    struct timespec seen_mtime = {0, 0};
    auto seen_since = begin;
    bool have_seen = false;
    for (;;) {
        const auto now = std::chrono::steady_clock::now();
        const int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
        bool got = false;
        unsigned char buf[BA_COMM_ID_BYTES + 8];
        struct stat sb;
        if (fd >= 0) {
            got = fstat(fd, &sb) == 0 && read(fd, buf, sizeof buf) == (ssize_t)sizeof buf;
            (void)close(fd);
        }
        if (got) {
            unsigned long long fn = 0;
            memcpy(&fn, buf + BA_COMM_ID_BYTES, 8);
            if (fn == nonce) {
                const double age_at_start = (double)(t0.tv_sec - sb.st_mtim.tv_sec) + 1e-9 * (double)(t0.tv_nsec - sb.st_mtim.tv_nsec);
                bool take = nonce != 0 || age_at_start < 2.0;
                if (!take) { // written before this process started: only once it has stayed the same for the grace period
                    const bool same = have_seen && seen_mtime.tv_sec == sb.st_mtim.tv_sec && seen_mtime.tv_nsec == sb.st_mtim.tv_nsec &&
                                      memcmp(seen, buf, sizeof buf) == 0;
                    if (!same) {
                        memcpy(seen, buf, sizeof buf);
                        seen_mtime = sb.st_mtim;
                        seen_since = now;
                        have_seen = true;
                    } else if (std::chrono::duration<double>(now - seen_since).count() >= grace_s)
                        take = true;
                }
                if (take) {
                    memcpy(id_out, buf, BA_COMM_ID_BYTES);
                    return BA_OK;
                }
            }
        } else
            have_seen = false; // (rank 0 has removed it: whatever comes next is new)
        if (std::chrono::duration<double>(now - begin).count() >= wait_s) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    fprintf(stderr, "ba_mi355x: no communicator id of this launch appeared in %s within %.0f s (rank %d)\n", path, wait_s, rank);
    return BA_ERR_COMM;
}

// Call after ba_solver_comm_init has returned (collective): rank 0 removes the rendezvous file, the others do nothing.
int ba_comm_id_file_done(const char *path, int rank)
{
    if (!path || rank < 0) return BA_ERR_ARG;
    if (rank == 0) (void)unlink(path);
    return BA_OK;
}

} // extern "C"
