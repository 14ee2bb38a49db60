// diasss_amd/csrc/dsss_comm.hip -- the communicator of a context: one process per GPU, sums across ranks.
// north_star's multi-GPU split is "image pairs shard over the GPUs of one node, RCCL all-reduce over xGMI of the reduced
// Hessian": dsss_pg.hip calls dsss_comm_allreduce once per LM trial on [interface values | update matrices | right-hand side]
// and once more on three scalars.  Two transports:
//   RCCL      ncclAllReduce on device buffers, on the context's stream (dsss_comm_init).  The RCCL entry points are resolved
//             at run time (dlsym on what the process already loaded -- PyTorch ships its own librccl next to its own HIP
//             runtime -- else dlopen of librccl.so), for the same reason libdsss.so leaves the HIP runtime undefined: two
//             HIP runtimes in one process cannot share the GPU.
//   callback  the caller sums a HOST buffer in place (dsss_comm_init_callback): lets the test-suite run two ranks over gloo
//             on a box with one GPU, where RCCL refuses two ranks on the same device.
#include "dsss_internal.h"
#include <dlfcn.h>

namespace {
typedef struct { char internal[128]; } rccl_uid;                       // ncclUniqueId (rccl.h:43)
typedef int (*fn_get_uid)(rccl_uid*);
typedef int (*fn_init_rank)(void** comm, int nranks, rccl_uid id, int rank);
typedef int (*fn_destroy)(void* comm);
typedef int (*fn_allreduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t st);
typedef int (*fn_allgather)(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t st);
typedef const char* (*fn_errstr)(int);
struct rccl_api { fn_get_uid get_uid = nullptr; fn_init_rank init_rank = nullptr; fn_destroy destroy = nullptr; fn_allreduce allreduce = nullptr; fn_allgather allgather = nullptr; fn_errstr errstr = nullptr; bool tried = false; };
rccl_api g_rccl;
const int RCCL_FLOAT64 = 8, RCCL_SUM = 0, RCCL_UINT8 = 1;              // ncclFloat64, ncclSum, ncclUint8 (rccl.h:448,460,467)

bool rccl_load(std::string* why)
{
    if (g_rccl.allreduce) return true;
    if (g_rccl.tried) { if (why) *why = "librccl not available"; return false; }
    g_rccl.tried = true;
    void* h = nullptr;
    if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
        const char* names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (h) break; }      // already in the process (PyTorch's copy)
        if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) { if (why) *why = std::string("dlopen(librccl): ") + dlerror(); return false; }
    }
    auto sym = [&](const char* n) { void* p = h ? dlsym(h, n) : nullptr; return p ? p : dlsym(RTLD_DEFAULT, n); };
    g_rccl.get_uid = (fn_get_uid)sym("ncclGetUniqueId"); g_rccl.init_rank = (fn_init_rank)sym("ncclCommInitRank");
    g_rccl.destroy = (fn_destroy)sym("ncclCommDestroy"); g_rccl.allreduce = (fn_allreduce)sym("ncclAllReduce"); g_rccl.allgather = (fn_allgather)sym("ncclAllGather"); g_rccl.errstr = (fn_errstr)sym("ncclGetErrorString");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.allreduce) { g_rccl.allreduce = nullptr; if (why) *why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce"; return false; }
    return true;
}
} // namespace

struct dsss_comm {
    int rank = 0, world = 1;
    void* nccl = nullptr;                                              // ncclComm_t
    dsss_comm_fn cb = nullptr; void* cb_user = nullptr;
    dsss_comm_dev_fn dcb = nullptr;                                    // transport that works on the DEVICE buffer, ordered on the stream
    double* h_stage = nullptr; size_t h_cap = 0;                       // pinned staging of the callback transport
    double bytes = 0; long long calls = 0;
};

void dsss_comm_free(dsss_ctx* c)
{
    dsss_comm* m = c->comm;
    if (!m) return;
    if (m->nccl && g_rccl.destroy) g_rccl.destroy(m->nccl);
    if (m->h_stage) hipHostFree(m->h_stage);
    delete m; c->comm = nullptr;
}
int dsss_comm_rank(const dsss_ctx* c) { return c->comm ? c->comm->rank : 0; }
int dsss_comm_world(const dsss_ctx* c) { return c->comm ? c->comm->world : 1; }

// in-place sum over the ranks of n doubles in device memory, ordered on stream st; every rank ends with identical bits
int dsss_comm_allreduce(dsss_ctx* c, double* dev, size_t n, hipStream_t st)
{
    dsss_comm* m = c->comm;
    if (!m || n == 0) return DSSS_OK;
    m->bytes += (double)n * 8; m->calls++;
    if (m->nccl) {
        const int rc = g_rccl.allreduce(dev, dev, n, RCCL_FLOAT64, RCCL_SUM, m->nccl, st);
        if (rc != 0) DSSS_FAIL(c, DSSS_E_COMM, "ncclAllReduce: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "error");
        return DSSS_OK;
    }
    if (m->dcb) {
        if (m->dcb(m->cb_user, 0, dev, n, (void*)st) != 0) DSSS_FAIL(c, DSSS_E_COMM, "all-reduce device callback failed");
        return DSSS_OK;
    }
    if (m->cb) {
        if (m->h_cap < n) { if (m->h_stage) hipHostFree(m->h_stage); m->h_stage = nullptr; m->h_cap = 0;
                            HIPCHK(c, hipHostMalloc(&m->h_stage, n * sizeof(double), hipHostMallocDefault)); m->h_cap = n; }
        HIPCHK(c, hipMemcpyAsync(m->h_stage, dev, n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (m->cb(m->cb_user, 0, m->h_stage, n) != 0) DSSS_FAIL(c, DSSS_E_COMM, "all-reduce callback failed");
        HIPCHK(c, hipMemcpyAsync(dev, m->h_stage, n * sizeof(double), hipMemcpyHostToDevice, st));
        return DSSS_OK;
    }
    return DSSS_OK;                                                    // world 1 without a transport: nothing to sum
}

// every rank contributes `bytes` bytes (device, at recv + rank * bytes); afterwards recv holds world x bytes on every rank
int dsss_comm_allgather(dsss_ctx* c, void* recv_dev, size_t bytes, hipStream_t st)
{
    dsss_comm* m = c->comm;
    if (!m || m->world == 1 || bytes == 0) return DSSS_OK;
    m->bytes += (double)bytes * m->world; m->calls++;
    if (m->nccl) {
        if (!g_rccl.allgather) DSSS_FAIL(c, DSSS_E_COMM, "librccl lacks ncclAllGather");
        const int rc = g_rccl.allgather((const char*)recv_dev + (size_t)m->rank * bytes, recv_dev, bytes, RCCL_UINT8, m->nccl, st);
        if (rc != 0) DSSS_FAIL(c, DSSS_E_COMM, "ncclAllGather: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "error");
        return DSSS_OK;
    }
    if (m->dcb) {
        if (m->dcb(m->cb_user, 1, recv_dev, bytes, (void*)st) != 0) DSSS_FAIL(c, DSSS_E_COMM, "all-gather device callback failed");
        return DSSS_OK;
    }
    if (m->cb) {
        const size_t tot = bytes * m->world, nd = (tot + 7) / 8;
        if (m->h_cap < nd) { if (m->h_stage) hipHostFree(m->h_stage); m->h_stage = nullptr; m->h_cap = 0;
                             HIPCHK(c, hipHostMalloc(&m->h_stage, nd * sizeof(double), hipHostMallocDefault)); m->h_cap = nd; }
        HIPCHK(c, hipMemcpyAsync((char*)m->h_stage + (size_t)m->rank * bytes, (const char*)recv_dev + (size_t)m->rank * bytes, bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (m->cb(m->cb_user, 1, m->h_stage, bytes) != 0) DSSS_FAIL(c, DSSS_E_COMM, "all-gather callback failed");
        HIPCHK(c, hipMemcpyAsync(recv_dev, m->h_stage, tot, hipMemcpyHostToDevice, st));
    }
    return DSSS_OK;
}

extern "C" {

int dsss_comm_unique_id(void* id128)
{
    if (!id128) return DSSS_E_ARG;
    std::string why;
    if (!rccl_load(&why)) return DSSS_E_COMM;
    rccl_uid id;
    if (g_rccl.get_uid(&id) != 0) return DSSS_E_COMM;
    memcpy(id128, id.internal, 128);
    return DSSS_OK;
}

int dsss_comm_init(dsss_ctx* c, const void* id128, int rank, int world)
{
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    std::string why;
    if (!rccl_load(&why)) DSSS_FAIL(c, DSSS_E_COMM, "%s", why.c_str());
    dsss_comm_free(c);
    dsss_comm* m = new dsss_comm(); m->rank = rank; m->world = world;
    rccl_uid id; memcpy(id.internal, id128, 128);
    const int rc = g_rccl.init_rank(&m->nccl, world, id, rank);
    if (rc != 0) { delete m; DSSS_FAIL(c, DSSS_E_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.errstr ? g_rccl.errstr(rc) : "error"); }
    c->comm = m;
    return DSSS_OK;
}

int dsss_comm_init_callback(dsss_ctx* c, int rank, int world, dsss_comm_fn fn, void* user)
{
    if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) return DSSS_E_ARG;
    dsss_comm_free(c);
    dsss_comm* m = new dsss_comm(); m->rank = rank; m->world = world; m->cb = fn; m->cb_user = user;
    c->comm = m;
    return DSSS_OK;
}

int dsss_comm_init_device_callback(dsss_ctx* c, int rank, int world, dsss_comm_dev_fn fn, void* user)
{
    if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) return DSSS_E_ARG;
    dsss_comm_free(c);
    dsss_comm* m = new dsss_comm(); m->rank = rank; m->world = world; m->dcb = fn; m->cb_user = user;
    c->comm = m;
    return DSSS_OK;
}

int dsss_comm_destroy(dsss_ctx* c) { if (!c) return DSSS_E_ARG; hipStreamSynchronize(c->stream); dsss_comm_free(c); return DSSS_OK; }

int dsss_comm_stats(dsss_ctx* c, int* rank, int* world, double* allreduce_bytes, int64_t* allreduce_calls)
{
    if (!c) return DSSS_E_ARG;
    if (rank) *rank = dsss_comm_rank(c);
    if (world) *world = dsss_comm_world(c);
    if (allreduce_bytes) *allreduce_bytes = c->comm ? c->comm->bytes : 0.0;
    if (allreduce_calls) *allreduce_calls = c->comm ? c->comm->calls : 0;
    return DSSS_OK;
}

} // extern "C"
