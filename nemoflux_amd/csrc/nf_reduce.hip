// nf_reduce.hip -- the ONE collective of the multi-GPU path, behind the C ABI: the per-rank partial rows
// (nt, nseg + ntransect) float64 are summed over all ranks with a single ncclAllReduce over RCCL / xGMI.
//
// The reference has no distributed code; the serial loop this shards is fluxplot.py:51-59 (time steps) around
// field.py:161 (the z contraction): every (t,z) slab adds independently to every transect total, so a rank integrates its
// own slabs (nf_field_set_slab_range) and one reduce of the small rows finishes the job.
//
// librccl is NOT a link-time dependency (the image holds two copies with the SONAME librccl.so.1: /opt/rocm/lib and the
// one PyTorch ships): the entry points are resolved at first use from the copy the process already holds
// (dlopen RTLD_NOLOAD), so a Python client shares torch's RCCL and a plain-C client gets the system's.
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

#include <mutex>

#include "nf_common.h"

namespace nf {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string origin;
    bool ok = false;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static void rccl_load()
{
    RcclApi &a = g_rccl;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)                       // the copy this process already mapped, if any
        if ((a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!a.handle)
        for (const char *n : names)
            if ((a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!a.handle) return;
    auto sym = [&](const char *s) { return dlsym(a.handle, s); };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
    a.CommCuDevice = (decltype(a.CommCuDevice))sym("ncclCommCuDevice");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.CommCount && a.CommUserRank && a.CommCuDevice &&
           a.AllReduce && a.GetErrorString;
    Dl_info info;
    if (a.AllReduce && dladdr((void *)a.AllReduce, &info) && info.dli_fname) a.origin = info.dli_fname;
}

static int rccl_api(RcclApi **out)
{
    std::call_once(g_rccl_once, rccl_load);
    if (!g_rccl.ok) {
        set_error("RCCL is not loadable (librccl.so.1: dlopen / dlsym failed); the multi-GPU reduce needs it");
        return NF_ERR_HOST;
    }
    *out = &g_rccl;
    return NF_OK;
}

static int rccl_fail(const RcclApi &a, ncclResult_t r, const char *what)
{
    char buf[400];
    snprintf(buf, sizeof buf, "RCCL error %d (%s) in %s", (int)r, a.GetErrorString(r), what);
    set_error(buf);
    return NF_ERR_HIP;
}

#define NF_RCCL(api, call)                                           \
    do {                                                             \
        ncclResult_t r_ = (call);                                    \
        if (r_ != ncclSuccess) return rccl_fail(*(api), r_, #call);  \
    } while (0)

static int need_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no usable AMD GPU (hipGetDeviceCount); nemoflux_amd has no CPU fallback");
        return NF_ERR_NO_DEVICE;
    }
    return NF_OK;
}

}  // namespace nf

using namespace nf;
extern "C" {

int nf_rccl_unique_id(void *id128)
try {
    NF_REQUIRE(id128, NF_ERR_ARG, "nf_rccl_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == NF_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    RcclApi *a = nullptr;
    if (int rc = rccl_api(&a)) return rc;
    ncclUniqueId id;
    NF_RCCL(a, a->GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return NF_OK;
} catch (...) {
    set_error("nf_rccl_unique_id: internal error");
    return NF_ERR_HOST;
}

// Everything nf_rccl_comm_init needs that can be checked WITHOUT the other ranks: librccl resolves and the calling thread
// has a usable HIP device.  ncclCommInitRank is collective -- a rank that fails before it leaves the others waiting
// inside it -- so the ranks call this first, agree on the outcome by whatever means they already share (nemoflux_amd.dist:
// one MIN all-reduce over torch.distributed) and only then create the communicator together.
int nf_rccl_preflight(int *device)
try {
    if (int rc = need_device()) return rc;
    RcclApi *a = nullptr;
    if (int rc = rccl_api(&a)) return rc;
    int dev = -1;
    NF_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    NF_HIP(hipGetDeviceProperties(&prop, dev));
    if (device) *device = dev;
    return NF_OK;
} catch (...) {
    set_error("nf_rccl_preflight: internal error");
    return NF_ERR_HOST;
}

int nf_rccl_comm_init(void **comm, int nranks, const void *id128, int rank)
try {
    NF_REQUIRE(comm && id128, NF_ERR_ARG, "nf_rccl_comm_init: null argument");
    NF_REQUIRE(nranks > 0 && rank >= 0 && rank < nranks, NF_ERR_ARG, "nf_rccl_comm_init: rank outside [0, nranks)");
    if (int rc = need_device()) return rc;
    RcclApi *a = nullptr;
    if (int rc = rccl_api(&a)) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    NF_RCCL(a, a->CommInitRank(&c, nranks, id, rank));   // binds to the calling thread's current HIP device
    *comm = c;
    return NF_OK;
} catch (...) {
    set_error("nf_rccl_comm_init: internal error");
    return NF_ERR_HOST;
}

int nf_rccl_comm_destroy(void *comm)
try {
    if (!comm) return NF_OK;
    RcclApi *a = nullptr;
    if (int rc = rccl_api(&a)) return rc;
    NF_RCCL(a, a->CommDestroy((ncclComm_t)comm));
    return NF_OK;
} catch (...) {
    set_error("nf_rccl_comm_destroy: internal error");
    return NF_ERR_HOST;
}

int nf_rccl_comm_info(void *comm, int *nranks, int *rank, int *device)
try {
    NF_REQUIRE(comm, NF_ERR_ARG, "nf_rccl_comm_info: null communicator");
    RcclApi *a = nullptr;
    if (int rc = rccl_api(&a)) return rc;
    if (nranks) NF_RCCL(a, a->CommCount((ncclComm_t)comm, nranks));
    if (rank) NF_RCCL(a, a->CommUserRank((ncclComm_t)comm, rank));
    if (device) NF_RCCL(a, a->CommCuDevice((ncclComm_t)comm, device));
    return NF_OK;
} catch (...) {
    set_error("nf_rccl_comm_info: internal error");
    return NF_ERR_HOST;
}

int nf_rccl_library(char *buf, int buflen)
try {
    NF_REQUIRE(buf && buflen > 0, NF_ERR_ARG, "nf_rccl_library: null or empty buffer");
    RcclApi *a = nullptr;
    if (int rc = rccl_api(&a)) return rc;
    snprintf(buf, (size_t)buflen, "%s", a->origin.c_str());
    return NF_OK;
} catch (...) {
    set_error("nf_rccl_library: internal error");
    return NF_ERR_HOST;
}

int nf_rows_allreduce(void *rccl_comm, double *rows_dev, size_t n, void *hip_stream)
try {
    NF_REQUIRE(rccl_comm, NF_ERR_ARG, "nf_rows_allreduce: null communicator");
    NF_REQUIRE(rows_dev || n == 0, NF_ERR_ARG, "nf_rows_allreduce: null rows");
    if (n == 0) return NF_OK;
    if (int rc = need_device()) return rc;
    RcclApi *a = nullptr;
    if (int rc = rccl_api(&a)) return rc;
    NF_RCCL(a, a->AllReduce(rows_dev, rows_dev, n, ncclDouble, ncclSum, (ncclComm_t)rccl_comm, (hipStream_t)hip_stream));
    return NF_OK;
} catch (...) {
    set_error("nf_rows_allreduce: internal error");
    return NF_ERR_HOST;
}

}  // extern "C"
