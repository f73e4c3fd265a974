// nf_capi_util.hip -- the C ABI of libnemoflux_amd.so, part 1 of 3: plumbing (errors, devices, memory, tuning knobs), the
// host-side helpers of the file ingest and the entry points of the on-device generator.  Declared in include/nemoflux_amd.h.
// (Part 2: nf_capi_mint.hip = Level 1, the mint-shaped surface; part 3: nf_capi_field.hip = Level 2, the Field engine.)
#include "nf_capi.h"

namespace nf {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_err = buf;
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? NF_ERR_NO_DEVICE : NF_ERR_HIP;
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no usable AMD GPU (hipGetDeviceCount); nemoflux_amd has no CPU fallback");
        return NF_ERR_NO_DEVICE;
    }
    return NF_OK;
}

int trap_exception() noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        try { g_err = "out of host memory"; } catch (...) {}
    } catch (const std::exception &e) {
        try { g_err = std::string("internal error: ") + e.what(); } catch (...) {}
    } catch (...) {
        try { g_err = "internal error: unknown C++ exception"; } catch (...) {}
    }
    return NF_ERR_HOST;
}

}  // namespace nf

using namespace nf;

// =============================================================================================== plumbing
extern "C" {

const char *nf_last_error(void) { return g_err.c_str(); }
int nf_version(void) { return 100; }

int nf_device_count(int *count)
try {
    NF_REQUIRE(count, NF_ERR_ARG, "nf_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return NF_OK;
}
NF_API_CATCH
int nf_set_device(int device)
try {
    NF_NEED_DEVICE();
    NF_HIP(hipSetDevice(device));
    return NF_OK;
}
NF_API_CATCH
int nf_device_name(char *buf, int buflen)
try {
    NF_REQUIRE(buf && buflen > 0, NF_ERR_ARG, "nf_device_name: null or empty buffer");
    NF_NEED_DEVICE();
    int dev = 0;
    NF_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    NF_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(buf, buflen, "%s:%s:%dCU", p.gcnArchName, p.name, p.multiProcessorCount);
    return NF_OK;
}
NF_API_CATCH
int nf_malloc(void **dev, size_t bytes)
try {
    NF_REQUIRE(dev, NF_ERR_ARG, "nf_malloc: null argument");
    NF_NEED_DEVICE();
    NF_HIP(hipMalloc(dev, bytes ? bytes : 16));
    return NF_OK;
}
NF_API_CATCH
int nf_free(void *dev)
try {
    if (dev) NF_HIP(hipFree(dev));
    return NF_OK;
}
NF_API_CATCH
int nf_host_alloc(void **host, size_t bytes)
try {
    NF_REQUIRE(host, NF_ERR_ARG, "nf_host_alloc: null argument");
    NF_NEED_DEVICE();
    NF_HIP(hipHostMalloc(host, bytes ? bytes : 16, hipHostMallocDefault));
    return NF_OK;
}
NF_API_CATCH
int nf_host_free(void *host)
try {
    if (host) NF_HIP(hipHostFree(host));
    return NF_OK;
}
NF_API_CATCH
int nf_memcpy_h2d(void *dev, const void *host, size_t bytes)
try {
    NF_NEED_DEVICE();
    NF_HIP(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    return NF_OK;
}
NF_API_CATCH
int nf_memcpy_d2h(void *host, const void *dev, size_t bytes)
try {
    NF_NEED_DEVICE();
    NF_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return NF_OK;
}
NF_API_CATCH
int nf_memset(void *dev, int value, size_t bytes)
try {
    NF_NEED_DEVICE();
    NF_HIP(hipMemset(dev, value, bytes));
    return NF_OK;
}
NF_API_CATCH
int nf_tuning_set(const char *name, int value)
try {
    NF_REQUIRE(name, NF_ERR_ARG, "nf_tuning_set: null name");
    if (!strcmp(name, "batch_cellsteps_m"))
        NF_REQUIRE(value >= 0 && value <= 2047, NF_ERR_ARG, "nf_tuning_set: batch_cellsteps_m must be in [0, 2047]");
    if (field_tuning_set(name, value) == NF_OK) return NF_OK;   // batch_steps, batch_cellsteps_m, partial_step_planes, graph
    if (!strcmp(name, "edge_weights")) {   // K3 on the engine's planes: 1 = unique-edge entries (built by the next
                                           // nf_field_build_weights), 0 = (cell, 4 weights) records (default)
        integral_use_edges(value);
        return NF_OK;
    }
    if (!strcmp(name, "datagen_rows")) {   // generator: 1 = the row kernel (default), 0 = one cell per lane with plain division
        datagen_use_rows(value);
        return NF_OK;
    }
    int rc = tuning_set(name, value);
    NF_REQUIRE(rc == NF_OK, NF_ERR_ARG, std::string("nf_tuning_set: unknown knob ") + name);
    return NF_OK;
}
NF_API_CATCH
int nf_release_scratch(void)
try {
    weights_trim_scratch();
    return NF_OK;
}
NF_API_CATCH
int nf_synchronize(void)
try {
    NF_NEED_DEVICE();
    NF_HIP(hipDeviceSynchronize());
    return NF_OK;
}
NF_API_CATCH

}  // extern "C"

extern "C" {

// ------------------------------------------------------------------------------------------- file decode helper
// Inverse of HDF5's shuffle filter on the HOST (file decoding, like zlib's inflate next to it -- not a compute path):
// src holds the es byte planes of n elements one after the other, dst receives the n elements.  Called by
// nemoflux_amd/hdf5min.py from its inflate threads (ctypes releases the GIL); ten times faster than numpy's strided copies.
int nf_host_unshuffle(const void *src, void *dst, size_t n, int es)
try {
    NF_REQUIRE(src && dst && es > 0 && es <= 16, NF_ERR_ARG, "nf_host_unshuffle: bad arguments");
    const unsigned char *s = (const unsigned char *)src;
    unsigned char *d = (unsigned char *)dst;
    if (es == 4) {
        const unsigned char *p0 = s, *p1 = s + n, *p2 = s + 2 * n, *p3 = s + 3 * n;
        uint32_t *o = (uint32_t *)d;
        if (((uintptr_t)d & 3) == 0) {
            for (size_t i = 0; i < n; ++i)
                o[i] = (uint32_t)p0[i] | ((uint32_t)p1[i] << 8) | ((uint32_t)p2[i] << 16) | ((uint32_t)p3[i] << 24);
            return NF_OK;
        }
    }
    for (int j = 0; j < es; ++j) {
        const unsigned char *pj = s + (size_t)j * n;
        for (size_t i = 0; i < n; ++i) d[i * es + j] = pj[i];
    }
    return NF_OK;
}
NF_API_CATCH

// Gather n byte ranges into a staging buffer with `nthreads` native threads (file ingest: the compressed chunks of a group
// of time steps, copied out of the mapped file into pinned memory).  One call, no interpreter lock between the copies: the
// Python thread pool this replaces took the GIL twice per chunk and stalled for hundreds of milliseconds whenever the
// caller's thread was busy in the interpreter.  Ranges are dealt to the threads in contiguous runs of about equal bytes.
int nf_host_gather(const unsigned long long *src_addr, const unsigned long long *dst_addr, const long long *len, long long n,
                   int nthreads)
try {
    NF_REQUIRE(n == 0 || (src_addr && dst_addr && len), NF_ERR_ARG, "nf_host_gather: null argument");
    NF_REQUIRE(n >= 0 && nthreads >= 1 && nthreads <= 256, NF_ERR_ARG, "nf_host_gather: bad counts");
    long long total = 0;
    for (long long i = 0; i < n; ++i) {
        NF_REQUIRE(len[i] >= 0, NF_ERR_ARG, "nf_host_gather: negative length");
        total += len[i];
    }
    if (total == 0) return NF_OK;
    const int nt = (int)std::min<long long>(nthreads, n);
    auto work = [&](long long lo, long long hi) {
        for (long long i = lo; i < hi; ++i)
            if (len[i]) memcpy((void *)(uintptr_t)dst_addr[i], (const void *)(uintptr_t)src_addr[i], (size_t)len[i]);
    };
    if (nt <= 1) {
        work(0, n);
        return NF_OK;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt);
    const long long share = (total + nt - 1) / nt;
    long long lo = 0;
    try {
        for (int t = 0; t < nt && lo < n; ++t) {
            long long hi = lo, acc = 0;
            while (hi < n && (acc < share || t == nt - 1)) acc += len[hi++];
            pool.emplace_back(work, lo, hi);
            lo = hi;
        }
    } catch (...) {   // a thread could not be started: the caller's thread finishes the rest, the started ones are joined
        for (auto &th : pool) th.join();   // (destroying a joinable std::thread would call std::terminate)
        work(lo, n);
        return NF_OK;
    }
    for (auto &th : pool) th.join();
    return NF_OK;
}
NF_API_CATCH

// ------------------------------------------------------------------------------------------- datagen
int nf_datagen_bounds(double *bounds_lon_dev, double *bounds_lat_dev, long ny, long nx, double xmin, double xmax,
                      double ymin, double ymax, double delta_lon_deg, double delta_lat_deg, int lat_uses_dx,
                      void *hip_stream)
try {
    NF_REQUIRE(bounds_lon_dev && bounds_lat_dev, NF_ERR_ARG, "nf_datagen_bounds: null argument");
    NF_NEED_DEVICE();
    NF_TRY(launch_datagen_bounds(bounds_lon_dev, bounds_lat_dev, ny, nx, xmin, xmax, ymin, ymax, delta_lon_deg,
                                 delta_lat_deg, lat_uses_dx, (hipStream_t)hip_stream));
    NF_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    return NF_OK;
}
NF_API_CATCH
int nf_datagen_uv(void *u_dev, void *v_dev, int dtype, long t_begin, long t_end, long nt, long nz, long ny, long nx,
                  double xmin, double xmax, double ymin, double ymax, double zmin, double zmax, int lat_uses_dx,
                  int psi, void *hip_stream)
try {
    NF_REQUIRE(u_dev && v_dev, NF_ERR_ARG, "nf_datagen_uv: null argument");
    NF_REQUIRE(dtype == NF_F64 || dtype == NF_F32, NF_ERR_ARG, "nf_datagen_uv: dtype must be NF_F64/NF_F32");
    NF_NEED_DEVICE();
    return launch_datagen_uv(u_dev, v_dev, dtype, t_begin, t_end, nt, nz, ny, nx, xmin, xmax, ymin, ymax, zmin, zmax,
                             lat_uses_dx, psi, (hipStream_t)hip_stream);
}
NF_API_CATCH

}  // extern "C"
