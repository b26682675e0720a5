// comm.hip -- the exchange step of the multi-GPU matching path behind the C ABI (RCCL over xGMI).
//
// The reference moves descriptors between robots over ROS 2 topics (GlobalDescriptors messages,
// cslam/global_descriptor_loop_closure_detection.py:198-227 publish, :407-422 receive) and the per-robot top-k lists
// never leave the robot.  Inside one 8-GPU node, with one robot bank (or one row shard of the metric's bank) per GPU,
// that transport is one all-gather of the step's new descriptors and, for a row-sharded bank, one all-to-all of the
// (rows, scores, count) lists (Python twin: cslam_amd/sharded.py over torch.distributed).  These entry points
// give a host that is not Python the same two collectives: one process per GPU, rank 0 creates the 128-byte id and hands
// it to the others by whatever channel the host has (file, socket, launcher environment).
//
// RCCL is resolved at run time (dlopen of the librccl the process already has -- PyTorch-ROCm ships one -- or the system
// one), so libcslam_hip.so carries no link-time dependency on it and loads on hosts without RCCL; the entry points then
// fail with CSLAM_E_UNSUPPORTED.
#include <dlfcn.h>
#include <new>
#include "common.h"

typedef struct { char internal[128]; } rcclUniqueId;                 // = ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
typedef void *rcclComm_t;
enum { RCCL_INT8 = 0 };                                              // ncclInt8 / ncclChar
typedef int (*fn_get_id)(rcclUniqueId *);
typedef int (*fn_init_rank)(rcclComm_t *, int, rcclUniqueId, int);
typedef int (*fn_destroy)(rcclComm_t);
typedef int (*fn_allgather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t);
typedef int (*fn_alltoall)(const void *, void *, size_t, int, rcclComm_t, hipStream_t);
typedef const char *(*fn_errstr)(int);

static struct {
    void *h;
    fn_get_id get_id; fn_init_rank init_rank; fn_destroy destroy; fn_allgather allgather; fn_alltoall alltoall;
    fn_errstr errstr;
    bool tried;
} g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false};

static int rccl_load() {
    if (g_rccl.h) return CSLAM_OK;
    if (!g_rccl.tried) {
        g_rccl.tried = true;
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_rccl.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
            g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
            g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
            g_rccl.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
            g_rccl.alltoall = (fn_alltoall)dlsym(h, "ncclAllToAll");
            g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
            if (g_rccl.get_id && g_rccl.init_rank && g_rccl.destroy && g_rccl.allgather && g_rccl.alltoall) { g_rccl.h = h; break; }
            dlclose(h);
        }
    }
    if (!g_rccl.h) { cslam_set_error("RCCL (librccl.so) not found: the multi-GPU exchange needs it"); return CSLAM_E_UNSUPPORTED; }
    return CSLAM_OK;
}

#define RCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        int _r = (expr);                                                                           \
        if (_r != 0) {                                                                             \
            cslam_set_error("%s failed: %s", #expr, g_rccl.errstr ? g_rccl.errstr(_r) : "RCCL error"); \
            return CSLAM_E_HIP;                                                                    \
        }                                                                                          \
    } while (0)

struct cslam_comm { rcclComm_t comm; int world, rank, device; };

CSLAM_API int cslam_comm_unique_id(void *id128) {
    ARG_CHECK(id128, "id buffer is NULL");
    int rc = rccl_load();
    if (rc) return rc;
    rcclUniqueId id;
    RCCL_TRY(g_rccl.get_id(&id));
    memcpy(id128, id.internal, 128);
    return CSLAM_OK;
}

CSLAM_API int cslam_comm_init(int world, int rank, const void *id128, int device, cslam_comm_t **out) {
    ARG_CHECK(out && id128, "NULL argument");
    ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "rank outside [0, world)");
    int rc = rccl_load();
    if (rc) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) { cslam_set_error("hipSetDevice(%d) failed", device); return CSLAM_E_HIP; }
    cslam_comm *c = new (std::nothrow) cslam_comm();
    if (!c) { cslam_set_error("out of host memory"); return CSLAM_E_NOMEM; }
    c->comm = nullptr; c->world = world; c->rank = rank; c->device = device;
    rcclUniqueId id;
    memcpy(id.internal, id128, 128);
    int r = g_rccl.init_rank(&c->comm, world, id, rank);
    if (r != 0) {
        cslam_set_error("ncclCommInitRank failed: %s", g_rccl.errstr ? g_rccl.errstr(r) : "RCCL error");
        delete c;
        return CSLAM_E_HIP;
    }
    *out = c;
    return CSLAM_OK;
}

CSLAM_API int cslam_comm_destroy(cslam_comm_t *c) {
    if (!c) return CSLAM_OK;
    if (c->comm && g_rccl.destroy) {
        DeviceGuard guard(c->device);
        (void)g_rccl.destroy(c->comm);
    }
    delete c;
    return CSLAM_OK;
}

CSLAM_API int cslam_comm_info(const cslam_comm_t *c, int *world, int *rank) {
    ARG_CHECK(c, "communicator is NULL");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return CSLAM_OK;
}

// every rank contributes `rows` descriptors of `row_bytes` bytes; d_all [world * rows][row_bytes], rank-major
CSLAM_API int cslam_allgather_queries_dev(cslam_comm_t *c, const void *d_local, int64_t rows, int64_t row_bytes, void *d_all,
                                          void *stream) {
    ARG_CHECK(c && (d_local || rows == 0) && (d_all || rows == 0), "NULL argument");
    ARG_CHECK(rows >= 0 && row_bytes >= 1, "bad sizes");
    if (rows == 0) return CSLAM_OK;
    DeviceGuard guard(c->device);
    if (!guard.ok) { cslam_set_error("hipSetDevice(%d) failed", c->device); return CSLAM_E_HIP; }
    RCCL_TRY(g_rccl.allgather(d_local, d_all, (size_t)(rows * row_bytes), RCCL_INT8, c->comm, (hipStream_t)stream));
    return CSLAM_OK;
}

// d_send [world][bytes_per_rank]: slice r goes to rank r; d_recv [world][bytes_per_rank]: slice s came from rank s
CSLAM_API int cslam_exchange_lists_dev(cslam_comm_t *c, const void *d_send, void *d_recv, int64_t bytes_per_rank, void *stream) {
    ARG_CHECK(c && (d_send || bytes_per_rank == 0) && (d_recv || bytes_per_rank == 0), "NULL argument");
    ARG_CHECK(bytes_per_rank >= 0, "bad size");
    if (bytes_per_rank == 0) return CSLAM_OK;
    DeviceGuard guard(c->device);
    if (!guard.ok) { cslam_set_error("hipSetDevice(%d) failed", c->device); return CSLAM_E_HIP; }
    RCCL_TRY(g_rccl.alltoall(d_send, d_recv, (size_t)bytes_per_rank, RCCL_INT8, c->comm, (hipStream_t)stream));
    return CSLAM_OK;
}
