// mcx_comm.hip — the multi-GPU exchange of the C ABI (include/mcx.h "Multi-GPU exchange"): RCCL over xGMI.
//
// paths are independent units sharded over the GPUs of one node (one process per GPU); what crosses GPUs is
// a few KB per pass — accumulator records, LSM normal-equation moments, radix-select histograms.  RCCL is resolved with
// dlopen at the first mcx_comm_* call, so libmcx_hip.so loads (and every single-GPU entry point works) on a box without it,
// and a process that already carries torch's RCCL shares that copy (same soname).
#include "mcx_internal.h"

#include <dlfcn.h>

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x)
typedef struct { char internal[MCX_COMM_ID_BYTES]; } rccl_unique_id;
typedef void* rccl_comm;
enum { RCCL_SUCCESS = 0 };
enum { RCCL_FLOAT64 = 8 };      // ncclDataType_t: ncclFloat64 / ncclDouble
enum { RCCL_SUM = 0 };          // ncclRedOp_t

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(rccl_comm*, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

RcclApi* rccl(mcx_handle* h)
{
    static RcclApi api;
    if (api.lib) return &api;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { h->err = std::string("RCCL not found: ") + dlerror(); return nullptr; }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(lib, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(lib, "ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))dlsym(lib, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.AllGather) {
        h->err = "RCCL library lacks a required symbol";
        dlclose(lib);
        return nullptr;
    }
    api.lib = lib;
    return &api;
}

#define MCX_RCCL(h, api, expr)                                                                                        \
    do {                                                                                                              \
        const int _r = (expr);                                                                                        \
        if (_r != RCCL_SUCCESS) MCX_FAIL(h, -200 - _r, "%s failed: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(_r) : "?"); \
    } while (0)

}  // namespace

extern "C" int mcx_comm_unique_id(mcx_handle* h, void* out_id)
{
    if (!h || !out_id) return -1;
    RcclApi* api = rccl(h);
    if (!api) return -2;
    rccl_unique_id id;
    MCX_RCCL(h, api, api->GetUniqueId(&id));
    memcpy(out_id, &id, MCX_COMM_ID_BYTES);
    return 0;
}

extern "C" int mcx_comm_init(mcx_handle* h, int32_t n_ranks, int32_t rank, const void* id)
{
    if (!h || !id) return -1;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) MCX_FAIL(h, -2, "mcx_comm_init: rank %d of %d", rank, n_ranks);
    if (h->comm) MCX_FAIL(h, -3, "mcx_comm_init: the handle already has a communicator");
    RcclApi* api = rccl(h);
    if (!api) return -2;
    MCX_HIP(h, hipSetDevice(h->device));
    rccl_unique_id uid;
    memcpy(&uid, id, MCX_COMM_ID_BYTES);
    rccl_comm c = nullptr;
    MCX_RCCL(h, api, api->CommInitRank(&c, n_ranks, uid, rank));
    h->comm = c; h->comm_ranks = n_ranks; h->comm_rank = rank;
    return 0;
}

extern "C" int mcx_comm_destroy(mcx_handle* h)
{
    if (!h) return -1;
    if (!h->comm) return 0;
    RcclApi* api = rccl(h);
    if (api) api->CommDestroy((rccl_comm)h->comm);
    h->comm = nullptr; h->comm_ranks = 1; h->comm_rank = 0;
    return 0;
}

extern "C" int mcx_allreduce_f64(mcx_handle* h, double* d_buf, int64_t n, void* stream)
{
    if (!h || !d_buf || n < 0) return -1;
    if (!h->comm) MCX_FAIL(h, -3, "mcx_allreduce_f64: call mcx_comm_init first");
    if (n == 0) return 0;
    RcclApi* api = rccl(h);
    if (!api) return -2;
    MCX_RCCL(h, api, api->AllReduce(d_buf, d_buf, (size_t)n, RCCL_FLOAT64, RCCL_SUM, (rccl_comm)h->comm, (hipStream_t)stream));
    return 0;
}

extern "C" int mcx_allgather_f64(mcx_handle* h, const double* d_in, double* d_out, int64_t n, void* stream)
{
    if (!h || !d_in || !d_out || n < 0) return -1;
    if (!h->comm) MCX_FAIL(h, -3, "mcx_allgather_f64: call mcx_comm_init first");
    if (n == 0) return 0;
    RcclApi* api = rccl(h);
    if (!api) return -2;
    MCX_RCCL(h, api, api->AllGather(d_in, d_out, (size_t)n, RCCL_FLOAT64, (rccl_comm)h->comm, (hipStream_t)stream));
    return 0;
}
