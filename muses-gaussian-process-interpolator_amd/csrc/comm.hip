// RCCL over xGMI behind the C-ABI: communicators and the few collectives the 2-D block-cyclic Cholesky needs
// (SURVEY.md section 8e): panel broadcasts along process rows / columns, small reductions of the replicated
// right-hand side, scalar all-reduces (logdet, failing pivot, the benchmark's max-over-ranks clock).
//
// librccl.so.1 is 570 MB; it is opened lazily with dlopen the first time a communicator entry point is called, so the
// single-GPU path never maps it.  Every collective is enqueued on the stream of the ctx passed with it (no host
// synchronisation); RCCL errors come back as MGPI_ERR_COMM with ncclGetErrorString in mgpi_last_error().
//
// The reference has no communication of any kind ("Distributed communication backend: absent");
// what these calls move are the blocks of K = L L^T that take the role of np.linalg.inv's operand
// (mgpi/interpolators.py:92,179,343) when it is spread over the GPUs of one node.
#include "mgpi_internal.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <new>
#include <stdlib.h>

struct rccl_api {
    void* handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
    const char* (*GetErrorString)(ncclResult_t);
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
};

static rccl_api g_rccl;
static int g_rccl_state = 0;     // 0 not tried, 1 loaded, -1 failed

static int rccl_load() {
    if (g_rccl_state == 1) return 0;
    if (g_rccl_state == -1) {
        mgpi_set_error("librccl could not be loaded earlier in this process");
        return MGPI_ERR_COMM;
    }
    // MGPI_RCCL_LIB names THE library to use (nothing else is tried when it is set); otherwise the usual places
    const char* forced = getenv("MGPI_RCCL_LIB");
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = nullptr;
    if (forced != nullptr && *forced != 0) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    } else {
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
    }
    if (!h) {
        g_rccl_state = -1;
        const char* why = dlerror();
        mgpi_set_error("dlopen(%s) failed: %s", (forced != nullptr && *forced != 0) ? forced : "librccl.so.1", why ? why : "?");
        return MGPI_ERR_COMM;
    }
    g_rccl.handle = h;
    bool ok = true;
#define RCCL_SYM(field, name)                                                   \
    do {                                                                        \
        *(void**)(&g_rccl.field) = dlsym(h, name);                              \
        if (!g_rccl.field) { ok = false; mgpi_set_error("librccl has no symbol %s", name); } \
    } while (0)
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommSplit, "ncclCommSplit");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(CommCount, "ncclCommCount");
    RCCL_SYM(CommUserRank, "ncclCommUserRank");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
    RCCL_SYM(Broadcast, "ncclBroadcast");
    RCCL_SYM(Reduce, "ncclReduce");
    RCCL_SYM(AllReduce, "ncclAllReduce");
    RCCL_SYM(AllGather, "ncclAllGather");
    RCCL_SYM(Send, "ncclSend");
    RCCL_SYM(Recv, "ncclRecv");
    RCCL_SYM(GroupStart, "ncclGroupStart");
    RCCL_SYM(GroupEnd, "ncclGroupEnd");
#undef RCCL_SYM
    if (!ok) {
        g_rccl_state = -1;
        return MGPI_ERR_COMM;
    }
    g_rccl_state = 1;
    return 0;
}

#define MGPI_RCCL_TRY(expr)                                                                          \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess) {                                                                     \
            mgpi_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return MGPI_ERR_COMM;                                                                    \
        }                                                                                            \
    } while (0)

struct mgpi_comm {
    ncclComm_t comm;
    int rank, size;
    int device;
    int fanout;     // broadcasts as grouped point-to-point sends from the root (see mgpi_comm_bcast)
};

static_assert(MGPI_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "mgpi_hip.h and rccl.h disagree on the unique-id size");

extern "C" int mgpi_comm_unique_id(void* out) {
    MGPI_REQUIRE(out != nullptr, "mgpi_comm_unique_id: null pointer");
    MGPI_TRY(rccl_load());
    ncclUniqueId id;
    MGPI_RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

static int comm_wrap(ncclComm_t c, int rank, int size, int device, mgpi_comm** out) {
    mgpi_comm* m = new (std::nothrow) mgpi_comm();
    if (!m) { mgpi_set_error("mgpi_comm: out of host memory"); return MGPI_ERR_ALLOC; }
    m->comm = c;
    m->rank = rank;
    m->size = size;
    m->device = device;
    // xGMI is point to point (7 links per GPU, every peer of the node one hop away): a root that sends the panel to
    // its 1-3 row / column peers directly drives that many links at once, where a ring broadcast is bound by one.
    // MGPI_BCAST=rccl selects ncclBroadcast instead.
    const char* env = getenv("MGPI_BCAST");
    m->fanout = !(env && strcmp(env, "rccl") == 0);
    *out = m;
    return 0;
}

extern "C" int mgpi_comm_create(mgpi_ctx* ctx, const void* unique_id, int rank, int world, mgpi_comm** out) {
    MGPI_REQUIRE(ctx && unique_id && out, "mgpi_comm_create: null pointer");
    MGPI_REQUIRE(world >= 1 && rank >= 0 && rank < world, "mgpi_comm_create: rank %d outside [0, %d)", rank, world);
    *out = nullptr;
    MGPI_TRY(rccl_load());
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    MGPI_RCCL_TRY(g_rccl.CommInitRank(&c, world, id, rank));
    return comm_wrap(c, rank, world, ctx->device, out);
}

// ranks with the same color form one new communicator, ordered by key (ncclCommSplit); collective over `comm`
extern "C" int mgpi_comm_split(mgpi_comm* comm, int color, int key, mgpi_comm** out) {
    MGPI_REQUIRE(comm && out, "mgpi_comm_split: null pointer");
    MGPI_REQUIRE(color >= 0 && key >= 0, "mgpi_comm_split: color %d / key %d must be non-negative", color, key);
    *out = nullptr;
    MGPI_HIP_TRY(hipSetDevice(comm->device));
    ncclComm_t c = nullptr;
    MGPI_RCCL_TRY(g_rccl.CommSplit(comm->comm, color, key, &c, nullptr));
    int nr = 0, ns = 0;     // rank and size inside the new communicator (members are ordered by key)
    MGPI_RCCL_TRY(g_rccl.CommCount(c, &ns));
    MGPI_RCCL_TRY(g_rccl.CommUserRank(c, &nr));
    return comm_wrap(c, nr, ns, comm->device, out);
}

extern "C" int mgpi_comm_destroy(mgpi_comm* comm) {
    if (!comm) return 0;
    (void)hipSetDevice(comm->device);
    if (g_rccl_state == 1 && comm->comm) (void)g_rccl.CommDestroy(comm->comm);
    delete comm;
    return 0;
}

extern "C" int mgpi_comm_info(const mgpi_comm* comm, int* rank, int* size) {
    MGPI_REQUIRE(comm != nullptr, "mgpi_comm_info: null communicator");
    if (rank) *rank = comm->rank;
    if (size) *size = comm->size;
    return 0;
}

#define COMM_ARGS_OK(name)                                                                                      \
    MGPI_REQUIRE(comm && ctx && buf, name ": null pointer");                                                    \
    MGPI_REQUIRE(count >= 0, name ": negative count");                                                          \
    MGPI_REQUIRE(ctx->device == comm->device, name ": ctx is on device %d, the communicator on %d", ctx->device, \
                 comm->device);                                                                                 \
    if (count == 0) return 0;                                                                                   \
    MGPI_HIP_TRY(hipSetDevice(ctx->device))

// buf (count doubles, device) on rank `root` of this communicator -> the same buffer on every other rank
extern "C" int mgpi_comm_bcast(mgpi_comm* comm, mgpi_ctx* ctx, double* buf, int64_t count, int root) {
    COMM_ARGS_OK("mgpi_comm_bcast");
    MGPI_REQUIRE(root >= 0 && root < comm->size, "mgpi_comm_bcast: root %d outside [0, %d)", root, comm->size);
    if (comm->size == 1) return 0;
    if (!comm->fanout) {
        MGPI_RCCL_TRY(g_rccl.Broadcast(buf, buf, (size_t)count, ncclDouble, root, comm->comm, ctx->stream));
        return 0;
    }
    MGPI_RCCL_TRY(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    if (comm->rank == root) {
        for (int peer = 0; peer < comm->size && r == ncclSuccess; ++peer)
            if (peer != root) r = g_rccl.Send(buf, (size_t)count, ncclDouble, peer, comm->comm, ctx->stream);
    } else {
        r = g_rccl.Recv(buf, (size_t)count, ncclDouble, root, comm->comm, ctx->stream);
    }
    ncclResult_t e = g_rccl.GroupEnd();
    MGPI_RCCL_TRY(r);
    MGPI_RCCL_TRY(e);
    return 0;
}

// in place: on `root`, buf <- sum over ranks of buf; the other ranks' buffers are unchanged
extern "C" int mgpi_comm_reduce_sum(mgpi_comm* comm, mgpi_ctx* ctx, double* buf, int64_t count, int root) {
    COMM_ARGS_OK("mgpi_comm_reduce_sum");
    MGPI_REQUIRE(root >= 0 && root < comm->size, "mgpi_comm_reduce_sum: root %d outside [0, %d)", root, comm->size);
    if (comm->size == 1) return 0;
    MGPI_RCCL_TRY(g_rccl.Reduce(buf, buf, (size_t)count, ncclDouble, ncclSum, root, comm->comm, ctx->stream));
    return 0;
}

// in place on every rank; op 0 = sum, 1 = max
extern "C" int mgpi_comm_allreduce(mgpi_comm* comm, mgpi_ctx* ctx, double* buf, int64_t count, int op) {
    COMM_ARGS_OK("mgpi_comm_allreduce");
    MGPI_REQUIRE(op == 0 || op == 1, "mgpi_comm_allreduce: op %d (0 = sum, 1 = max)", op);
    if (comm->size == 1) return 0;
    MGPI_RCCL_TRY(g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, op == 0 ? ncclSum : ncclMax, comm->comm, ctx->stream));
    return 0;
}

// recv (size * count doubles) <- the send buffers (count doubles each) of all ranks, in rank order
extern "C" int mgpi_comm_allgather(mgpi_comm* comm, mgpi_ctx* ctx, const double* send, double* recv, int64_t count) {
    const double* buf = send;
    COMM_ARGS_OK("mgpi_comm_allgather");
    MGPI_REQUIRE(recv != nullptr, "mgpi_comm_allgather: null receive buffer");
    if (comm->size == 1) {
        if (recv != send) MGPI_HIP_TRY(hipMemcpyAsync(recv, send, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    MGPI_RCCL_TRY(g_rccl.AllGather(send, recv, (size_t)count, ncclDouble, comm->comm, ctx->stream));
    return 0;
}

// every rank's queued work on ctx's stream is complete and every rank has arrived: all-reduce of one scalar + sync
extern "C" int mgpi_comm_barrier(mgpi_comm* comm, mgpi_ctx* ctx) {
    MGPI_REQUIRE(comm && ctx, "mgpi_comm_barrier: null pointer");
    MGPI_REQUIRE(ctx->device == comm->device, "mgpi_comm_barrier: ctx is on device %d, the communicator on %d", ctx->device,
                 comm->device);
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    if (comm->size > 1) {
        double* token = ctx->d_scal + 8000;          // scratch scalar nothing else uses
        MGPI_HIP_TRY(hipMemsetAsync(token, 0, sizeof(double), ctx->stream));
        MGPI_RCCL_TRY(g_rccl.AllReduce(token, token, 1, ncclDouble, ncclSum, comm->comm, ctx->stream));
    }
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}
