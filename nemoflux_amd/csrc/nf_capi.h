// nf_capi.h -- what the three files of the C ABI share (nf_capi_util.hip: plumbing, nf_capi_mint.hip: Level 1,
// nf_capi_field.hip: Level 2): the exception barrier, per-call device scratch, the sparse host staging, and the Grid_t that
// Level 1 owns and a Field lends (nf_field_grid).  Internal: nothing here is exported.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <thread>
#include <vector>

#include "nf_common.h"

namespace nf {

// defined in nf_capi_util.hip (the per-thread error text lives there)
int require_device();
int trap_exception() noexcept;
// knobs of the Field layer (nf_capi_field.hip) that nf_tuning_set reaches: NF_OK, or -1 when the name is not one of them
int field_tuning_set(const char *name, int value);
#define NF_NEED_DEVICE()                      \
    do {                                      \
        int rc_ = nf::require_device();       \
        if (rc_ != NF_OK) return rc_;         \
    } while (0)
// Exception barrier of the C ABI: every entry point is a function-try-block ending in NF_API_CATCH, so a
// std::bad_alloc (or any other C++ exception) raised by the host-side containers becomes NF_ERR_HOST.
#define NF_API_CATCH catch (...) { return nf::trap_exception(); }

// device scratch that lives for one call: freed on every return path
struct DevTmp {
    void *p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        NF_HIP(hipMalloc(&p, bytes ? bytes : 16));
        return NF_OK;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

template <typename T>
static int dev_alloc(T **p, size_t count)
{
    NF_HIP(hipMalloc((void **)p, sizeof(T) * (count ? count : 1)));
    return NF_OK;
}
template <typename T>
static void dev_free(T *&p)
{
    if (p) (void)hipFree((void *)p);
    p = nullptr;
}

// Sparse staging of a caller's HOST (ncell,4) array for the Level-1 entry points: the 32 bytes of every cell an object
// touches (the records of a PolylineIntegral, the located cells of a VectorInterp) are gathered into a pinned buffer in
// record / point order, so that what crosses PCIe is n x 32 B instead of the whole array (207 MB at ORCA12 size) -- mint's
// own getIntegral is a sparse dot over the same entries (field.py:102).  Native threads from 32 Ki rows on; ids < 0 (a
// point outside the grid) give a row of zeros.
template <typename I>
static void host_gather_rows4(const double *data, const I *ids, long n, double *out)
{
    auto work = [=](long lo, long hi) {
        for (long k = lo; k < hi; ++k) {
            if (k + 16 < hi && ids[k + 16] >= 0) __builtin_prefetch(data + 4 * (long)ids[k + 16]);
            if (ids[k] >= 0) memcpy(out + 4 * k, data + 4 * (long)ids[k], 32);
            else memset(out + 4 * k, 0, 32);
        }
    };
    const long per = 1l << 15;
    long nthr = std::min<long>(std::min<long>(8, (long)std::thread::hardware_concurrency()), n / per);
    if (nthr <= 1) {
        work(0, n);
        return;
    }
    std::vector<std::thread> pool;
    const long chunk = (n + nthr - 1) / nthr;
    for (long t = 1; t < nthr; ++t) pool.emplace_back(work, t * chunk, std::min(n, (t + 1) * chunk));
    work(0, std::min(n, chunk));
    for (auto &th : pool) th.join();
}

// pinned host buffer + its HBM twin, sized once per weight build / point search
struct GatherStage {
    double *h = nullptr, *d = nullptr;
    long rows = 0;
    void release()
    {
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        h = d = nullptr;
        rows = 0;
    }
    int resize(long n)
    {
        release();
        if (n <= 0) return NF_OK;
        NF_HIP(hipHostMalloc((void **)&h, sizeof(double) * 4 * (size_t)n, hipHostMallocDefault));
        NF_HIP(hipMalloc((void **)&d, sizeof(double) * 4 * (size_t)n));
        rows = n;
        return NF_OK;
    }
    template <typename I>
    int upload(const double *data, const I *ids)   // gather on the host, one copy of rows x 32 B; complete at return
    {
        if (rows == 0) return NF_OK;
        host_gather_rows4(data, ids, rows, h);
        NF_HIP(hipMemcpy(d, h, sizeof(double) * 4 * (size_t)rows, hipMemcpyHostToDevice));
        return NF_OK;
    }
};

}  // namespace nf

struct Grid_t {
    long ncell = 0;
    double *host_points = nullptr;  // borrowed (ncell,4,3)
    double *d_xy = nullptr;         // corner table (ncell,4,2)
    bool owns_xy = true;
    long version = 0;               // bumped by every build: weights / located points of an older build are refused
    nf::LocatorBoxes boxes;             // the locator of this grid (filled by the first computeWeights, dropped when the points change)
    long row_length = 0;            // mnt_grid_setRowLength: the cells are rows of this many (0 = a flat list, like mint's)
};

// (x0, y0, dx, dy) and the counterclock flag of every segment of a polyline (npoints, 3); returns the number of segments
inline int polyline_segments(const double *xyz, int npoints, int counterclock, std::vector<double> &segs,
                             std::vector<int> &cc)
{
    for (int s = 0; s + 1 < npoints; ++s) {
        segs.push_back(xyz[3 * s]);
        segs.push_back(xyz[3 * s + 1]);
        segs.push_back(xyz[3 * (s + 1)] - xyz[3 * s]);
        segs.push_back(xyz[3 * (s + 1) + 1] - xyz[3 * s + 1]);
        cc.push_back(counterclock ? 1 : 0);
    }
    return npoints > 1 ? npoints - 1 : 0;
}
