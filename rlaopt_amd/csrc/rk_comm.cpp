// RCCL transport behind the C ABI (rk_comm_*): one communicator per rank (one rank per GPU), the
// three collectives the path uses.  librk_hip.so does NOT link RCCL: the entry points bind the library at first
// use - the copy already mapped into the process if there is one (PyTorch ships its own librccl.so; two copies
// in one process would each bring their own bootstrap state), otherwise the system's.  The host-side package
// moves its buffers with torch.distributed (the same RCCL underneath); these exports are the transport a
// maintainer binding rk.h into the reference would use in place of its worker queues
// (rlaopt/linops/base.py:202-267, linops/distributed.py:40-50,81-91).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "rk_common.h"

namespace rk {
namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

void bind_rccl() {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) {  // a copy that is already mapped (PyTorch's) wins
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        if (void* sym = dlsym(RTLD_DEFAULT, "ncclGetUniqueId")) {  // mapped under another name
            (void)sym;
            g_rccl.handle = RTLD_DEFAULT;
        }
    }
    if (!g_rccl.handle) {
        for (const char* n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
    }
    if (!g_rccl.handle) return;
#define RK_BIND(field, sym) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, sym))
    RK_BIND(GetUniqueId, "ncclGetUniqueId");
    RK_BIND(CommInitRank, "ncclCommInitRank");
    RK_BIND(CommDestroy, "ncclCommDestroy");
    RK_BIND(AllReduce, "ncclAllReduce");
    RK_BIND(AllGather, "ncclAllGather");
    RK_BIND(ReduceScatter, "ncclReduceScatter");
    RK_BIND(GetErrorString, "ncclGetErrorString");
#undef RK_BIND
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce &&
                g_rccl.AllGather && g_rccl.ReduceScatter && g_rccl.GetErrorString;
}

int need_rccl(const char* fn) {
    std::call_once(g_once, bind_rccl);
    if (!g_rccl.ok) {
        set_error("%s: RCCL is not available in this process (librccl.so could not be bound)", fn);
        return RK_ERR_LAUNCH;
    }
    return RK_OK;
}
int nccl_status(const char* fn, ncclResult_t r) {
    if (r == ncclSuccess) return RK_OK;
    set_error("%s: RCCL error: %s", fn, g_rccl.GetErrorString(r));
    return RK_ERR_LAUNCH;
}
bool dtype_of(int dtype, ncclDataType_t* out) {
    if (dtype == RK_F32) { *out = ncclFloat32; return true; }
    if (dtype == RK_F64) { *out = ncclFloat64; return true; }
    return false;
}
}  // namespace
}  // namespace rk

using namespace rk;

extern "C" int rk_comm_unique_id(void* id128) {
    clear_error();
    RK_REQUIRE(id128 != nullptr, "rk_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
    int rc = need_rccl("rk_comm_unique_id");
    if (rc != RK_OK) return rc;
    ncclUniqueId id;
    rc = nccl_status("rk_comm_unique_id", g_rccl.GetUniqueId(&id));
    if (rc == RK_OK) memcpy(id128, &id, sizeof(id));
    return rc;
}

extern "C" int rk_comm_create(void** comm, int nranks, int rank, const void* id128) {
    clear_error();
    RK_REQUIRE(comm != nullptr && id128 != nullptr, "rk_comm_create: null pointer");
    RK_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rk_comm_create: rank %d outside [0, %d)", rank, nranks);
    int rc = need_rccl("rk_comm_create");
    if (rc != RK_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    rc = nccl_status("rk_comm_create", g_rccl.CommInitRank(&c, nranks, id, rank));
    *comm = rc == RK_OK ? static_cast<void*>(c) : nullptr;
    return rc;
}

extern "C" int rk_comm_destroy(void* comm) {
    clear_error();
    if (comm == nullptr) return RK_OK;
    int rc = need_rccl("rk_comm_destroy");
    if (rc != RK_OK) return rc;
    return nccl_status("rk_comm_destroy", g_rccl.CommDestroy(static_cast<ncclComm_t>(comm)));
}

extern "C" int rk_comm_allreduce_sum(void* comm, void* buf, int64_t count, int dtype, void* stream) {
    clear_error();
    ncclDataType_t dt;
    RK_REQUIRE(comm && buf && count >= 0, "rk_comm_allreduce_sum: bad arguments");
    RK_REQUIRE(dtype_of(dtype, &dt), "rk_comm_allreduce_sum: bad dtype %d", dtype);
    int rc = need_rccl("rk_comm_allreduce_sum");
    if (rc != RK_OK) return rc;
    return nccl_status("rk_comm_allreduce_sum",
                       g_rccl.AllReduce(buf, buf, (size_t)count, dt, ncclSum, static_cast<ncclComm_t>(comm),
                                        static_cast<hipStream_t>(stream)));
}

extern "C" int rk_comm_allgather(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype,
                                 void* stream) {
    clear_error();
    ncclDataType_t dt;
    RK_REQUIRE(comm && send && recv && count_per_rank >= 0, "rk_comm_allgather: bad arguments");
    RK_REQUIRE(dtype_of(dtype, &dt), "rk_comm_allgather: bad dtype %d", dtype);
    int rc = need_rccl("rk_comm_allgather");
    if (rc != RK_OK) return rc;
    return nccl_status("rk_comm_allgather",
                       g_rccl.AllGather(send, recv, (size_t)count_per_rank, dt, static_cast<ncclComm_t>(comm),
                                        static_cast<hipStream_t>(stream)));
}

extern "C" int rk_comm_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype,
                                          void* stream) {
    clear_error();
    ncclDataType_t dt;
    RK_REQUIRE(comm && send && recv && count_per_rank >= 0, "rk_comm_reduce_scatter_sum: bad arguments");
    RK_REQUIRE(dtype_of(dtype, &dt), "rk_comm_reduce_scatter_sum: bad dtype %d", dtype);
    int rc = need_rccl("rk_comm_reduce_scatter_sum");
    if (rc != RK_OK) return rc;
    return nccl_status("rk_comm_reduce_scatter_sum",
                       g_rccl.ReduceScatter(send, recv, (size_t)count_per_rank, dt, ncclSum,
                                            static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
}
