// nf_geom.hip -- K0: cell-bounds assembly and great-circle edge lengths on gfx950.
//
// Replaces  nemoflux/horizgrid.py:17-22  (points (ncell,4,3) = lon,lat,0 per corner)
//           nemoflux/field.py:170-181 + nemoflux/geo.py:14-27  (arc lengths on the unit sphere)
//           nemoflux/field.py:27-30  (lon/lat box)
//
// Data layout in HBM (all float64):
//   corner table xy  (ncell,4,2)  lon,lat of corners 0=SW,1=SE,2=NE,3=NW   64 B/cell   (feeds K2)
//   arc4             (ncell,4)    edge e joins corner e -> (e+1)%4          32 B/cell   (API parity)
//   arcE, arcN       (ncell)      columns 1 and 2 of arc4 as SoA            16 B/cell   (feeds K1: only
//                                 these two columns are consumed, field.py:195-196)
//
// One workgroup = 256 cells.  The (ny,nx,4) bounds arrays are AoS per cell, so the 4 corners of the 64 cells
// of a wavefront are 256 CONTIGUOUS values: they are read coalesced (lane k <- element k), staged in LDS
// and re-read per cell (lane c <- its 4 lon + 4 lat).  The arithmetic follows geo.py statement by
// statement (same operation order, no fma contraction: built with -ffp-contract=off); device sin/cos/acos
// differ from glibc by ulps, which acos amplifies by 1/sin(angle) -- see test_geometry in tests/test_gpu_parity.py.
#include "nf_common.h"

namespace nf {

// order-preserving map double -> u64 (for atomicMin/atomicMax on signed doubles)
__host__ __device__ inline unsigned long long dkey(double x)
{
    unsigned long long b;
    __builtin_memcpy(&b, &x, 8);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
double box_key_to_double(unsigned long long k)
{
    unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    double x;
    __builtin_memcpy(&x, &b, 8);
    return x;
}

__device__ inline double wave_min(double x)
{
    for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o, kWave));
    return x;
}
__device__ inline double wave_max(double x)
{
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, kWave));
    return x;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_geometry(const T *__restrict__ blon, const T *__restrict__ blat,
                                                     long ncell, double *__restrict__ xy,
                                                     double *__restrict__ arc4, double *__restrict__ arcE,
                                                     double *__restrict__ arcN, unsigned long long *box)
{
    __shared__ double s_lon[kBlock * 4];
    __shared__ double s_lat[kBlock * 4];
    const int tid = threadIdx.x;
    const long c0 = (long)blockIdx.x * kBlock;
    const long nval = ncell * 4;
    // coalesced stage of this workgroup's 1024 lon + 1024 lat values
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        long g = c0 * 4 + tid + r * kBlock;
        if (g < nval) {
            s_lon[tid + r * kBlock] = (double)blon[g];
            s_lat[tid + r * kBlock] = (double)blat[g];
        }
    }
    __syncthreads();
    const long c = c0 + tid;
    double lomin = 1e300, lomax = -1e300, lamin = 1e300, lamax = -1e300;
    if (c < ncell) {
        double lon[4], lat[4], X[4], Y[4], Z[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            lon[v] = s_lon[tid * 4 + v];
            lat[v] = s_lat[tid * 4 + v];
            // geo.py:15-21 with radius = geo.EARTH_RADIUS = 1.0
            double lam = lon[v] * kDeg2Rad;
            double the = lat[v] * kDeg2Rad;
            double sl, cl, st, ct;  // sincos shares the argument reduction; same values as sin() and cos()
            sincos(lam, &sl, &cl);
            sincos(the, &st, &ct);
            double rho = 1.0 * ct;
            X[v] = rho * cl;
            Y[v] = rho * sl;
            Z[v] = 1.0 * st;
            lomin = fmin(lomin, lon[v]);
            lomax = fmax(lomax, lon[v]);
            lamin = fmin(lamin, lat[v]);
            lamax = fmax(lamax, lat[v]);
        }
        double a[4];
#pragma unroll
        for (int i0 = 0; i0 < 4; ++i0) {  // field.py:179-181
            const int i1 = (i0 + 1) & 3;
            double dot = (X[i0] * X[i1] + Y[i0] * Y[i1]) + Z[i0] * Z[i1];  // geo.py:26
            a[i0] = fabs(1.0 * acos(dot / 1.0));                            // geo.py:26-27
        }
        double2 *pxy = reinterpret_cast<double2 *>(xy + c * 8);
#pragma unroll
        for (int v = 0; v < 4; ++v) pxy[v] = make_double2(lon[v], lat[v]);
        double2 *pa = reinterpret_cast<double2 *>(arc4 + c * 4);
        pa[0] = make_double2(a[0], a[1]);
        pa[1] = make_double2(a[2], a[3]);
        arcE[c] = a[1];
        arcN[c] = a[2];
    }
    // lon/lat box: one atomic per wavefront and bound
    lomin = wave_min(lomin);
    lomax = wave_max(lomax);
    lamin = wave_min(lamin);
    lamax = wave_max(lamax);
    // The bounds only move outwards, so each is read first (device scope) and the atomic is issued only by a wave that would
    // move it: a stale read costs a spare atomic, never a missed one.  Unconditional, the 4 x 101 250 atomics of the
    // ORCA12-like grid queue up on four addresses and ARE the kernel's time (4.6 ms; tools/geom_timing.py).
    if ((tid & (kWave - 1)) == 0 && lomin <= lomax) {
        const unsigned long long k0 = dkey(lomin), k1 = dkey(lomax), k2 = dkey(lamin), k3 = dkey(lamax);
        if (k0 < __hip_atomic_load(&box[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&box[0], k0);
        if (k1 > __hip_atomic_load(&box[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&box[1], k1);
        if (k2 < __hip_atomic_load(&box[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&box[2], k2);
        if (k3 > __hip_atomic_load(&box[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&box[3], k3);
    }
}

__global__ void k_box_init(unsigned long long *box)
{
    box[0] = ~0ull;
    box[1] = 0ull;
    box[2] = ~0ull;
    box[3] = 0ull;
}

int launch_geometry(const void *blon, const void *blat, int dtype, long ncell, double *xy, double *arc4,
                    double *arcE, double *arcN, unsigned long long *box_keys, hipStream_t s)
{
    NF_REQUIRE(ncell > 0 && ncell < (1l << 31), NF_ERR_ARG, "geometry: ncell out of range");
    unsigned nb = (unsigned)((ncell + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_box_init, dim3(1), dim3(1), 0, s, box_keys);
    if (dtype == NF_F64)
        hipLaunchKernelGGL(k_geometry<double>, dim3(nb), dim3(kBlock), 0, s, (const double *)blon,
                           (const double *)blat, ncell, xy, arc4, arcE, arcN, box_keys);
    else if (dtype == NF_F32)
        hipLaunchKernelGGL(k_geometry<float>, dim3(nb), dim3(kBlock), 0, s, (const float *)blon,
                           (const float *)blat, ncell, xy, arc4, arcE, arcN, box_keys);
    else
        NF_REQUIRE(false, NF_ERR_ARG, "geometry: dtype must be NF_F64 or NF_F32");
    NF_HIP(hipGetLastError());
    return NF_OK;
}

// mint.Grid.setPoints path: points (ncell,4,3) -> corner table (ncell,4,2); one lane per corner.
__global__ __launch_bounds__(kBlock) void k_points_to_xy(const double *__restrict__ points, long ncorner,
                                                         double *__restrict__ xy)
{
    long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k < ncorner) {
        xy[2 * k] = points[3 * k];
        xy[2 * k + 1] = points[3 * k + 1];
    }
}
// horizgrid.py:19-22: points[..., 0] = lon; points[..., 1] = lat; z = 0
__global__ __launch_bounds__(kBlock) void k_xy_to_points(const double *__restrict__ xy, long ncorner,
                                                         double *__restrict__ points)
{
    long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k < ncorner) {
        points[3 * k] = xy[2 * k];
        points[3 * k + 1] = xy[2 * k + 1];
        points[3 * k + 2] = 0.0;
    }
}

int launch_corner_table_from_points(const double *points, long ncell, double *xy, hipStream_t s)
{
    long n = ncell * 4;
    hipLaunchKernelGGL(k_points_to_xy, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, points,
                       n, xy);
    NF_HIP(hipGetLastError());
    return NF_OK;
}
int launch_points_from_corner_table(const double *xy, long ncell, double *points, hipStream_t s)
{
    long n = ncell * 4;
    hipLaunchKernelGGL(k_xy_to_points, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, xy, n,
                       points);
    NF_HIP(hipGetLastError());
    return NF_OK;
}

}  // namespace nf
