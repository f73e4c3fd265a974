// nf_flux.hip -- K1: vertical integration of uo/vo over the owned z-levels of one time step, fused with
// the edge-flux assembly.  This is the bandwidth-bound kernel the roofline is quoted on.
//
// Replaces  nemoflux/field.py:145-163  readField: missing -> 0 (:157), sum_z thickness[z]*f[z,j,i] (:161)
//           nemoflux/field.py:183-234  computeIntegratedFlux: eU = +U*arc[:,1], eV = -V*arc[:,2] (:195-196),
//                                      cell-by-cell 4-edge array incl. the neighbour copies (:209-223),
//                                      Sverdrup scaling (:225-228), |eU|,|eV| (:231-232), running max (:234)
//
// Access pattern.  uo/vo are (nt,nz,ny,nx) x-fastest.  A lane owns VEC consecutive cells (16 B: 2 x f64 or
// 4 x f32) and walks z with stride ncell; a wavefront therefore reads 1 KiB contiguous per (z, field) with
// global_load_dwordx4, UZ levels x 2 fields in flight per lane before the first use.  Nothing is reused, so
// nothing is staged in LDS; the thickness vector is wave-uniform (scalar loads).
//
// Resident output layout (HBM, float64) -- SoA planes instead of the reference's (ncell,4) AoS, so that every
// store is a dense 16 B/lane stream and the neighbour copies become SHIFTED dense stores:
//   iV planes  [4][ncell]   plane 1 = eU, plane 2 = eV,
//                           plane 0[c+nx] = eV[c]           (south slot of the row above, field.py:219;
//                                                            row 0 is never written and stays 0)
//                           plane 3[j, (i+1)%nx] = eU[j,i]  (west slot incl. the periodic wrap, :221-223)
//   abs planes [2][ncell]   |eU|, |eV|
// nf_field_read_step() re-packs the planes into the reference's (ncell,4) layout on demand.
//
// Algorithmic bytes per (t,z,j,i) unit: 2*sizeof(T) read + (16 arc + 32 iV + 16 abs)/nz  (SURVEY 8d).
#include "nf_common.h"

namespace nf {

template <typename T, int VEC> struct vec_t;
typedef double dvec2 __attribute__((ext_vector_type(2)));  // clang vectors: accepted by the nontemporal builtin
typedef float fvec4 __attribute__((ext_vector_type(4)));
template <> struct vec_t<double, 2> { using type = dvec2; };
template <> struct vec_t<float, 4> { using type = fvec4; };
template <> struct vec_t<double, 1> { using type = double; };
template <> struct vec_t<float, 1> { using type = float; };

template <typename T, int VEC> struct Lanes {
    T x[VEC];
};

template <typename T, int VEC, bool NT>
__device__ inline Lanes<T, VEC> load_cells(const T *p)
{
    using V = typename vec_t<T, VEC>::type;
    Lanes<T, VEC> r;
    V v = NT ? __builtin_nontemporal_load(reinterpret_cast<const V *>(p)) : *reinterpret_cast<const V *>(p);
    __builtin_memcpy(&r, &v, sizeof(V));
    return r;
}

// missing -> 0 (field.py:157): NaN, or equal to the variable's _FillValue (compared in the file's dtype)
template <typename T>
__device__ inline double fixed(T x, T fill)
{
    return (x != x || x == fill) ? 0.0 : (double)x;
}

__device__ inline void store2(double *p, double a, double b, bool aligned)
{
    if (aligned) {
        *reinterpret_cast<double2 *>(p) = make_double2(a, b);
    } else {
        p[0] = a;
        p[1] = b;
    }
}

template <typename T, int VEC, int UZ, bool NT>
__global__ __launch_bounds__(kBlock) void k_flux(const T *__restrict__ u, const T *__restrict__ v, long ncell,
                                                 unsigned ny, unsigned nx, int z0, int z1,
                                                 const double *__restrict__ thickness,
                                                 const double *__restrict__ arcE,
                                                 const double *__restrict__ arcN, T fill, double scale,
                                                 int sverdrup, double *__restrict__ iV,
                                                 double *__restrict__ absUV, unsigned long long *maxbits,
                                                 unsigned ntiles, int xcd_map)
{
    const unsigned tile = xcd_map ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
    double tmax = 0.0;
    if (tile < ntiles) {
        const long c0 = ((long)tile * kBlock + threadIdx.x) * VEC;
        if (c0 < ncell) {  // ncell % VEC == 0 is guaranteed by the launcher, so the lane is full
            double accU[VEC], accV[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) accU[k] = accV[k] = 0.0;
            const T *pu = u + (long)z0 * ncell + c0;
            const T *pv = v + (long)z0 * ncell + c0;
            int z = z0;
            for (; z + UZ <= z1; z += UZ) {
                Lanes<T, VEC> lu[UZ], lv[UZ];
#pragma unroll
                for (int q = 0; q < UZ; ++q) {
                    lu[q] = load_cells<T, VEC, NT>(pu + (long)q * ncell);
                    lv[q] = load_cells<T, VEC, NT>(pv + (long)q * ncell);
                }
#pragma unroll
                for (int q = 0; q < UZ; ++q) {
                    const double th = thickness[z + q];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        accU[k] = fma(th, fixed<T>(lu[q].x[k], fill), accU[k]);
                        accV[k] = fma(th, fixed<T>(lv[q].x[k], fill), accV[k]);
                    }
                }
                pu += (long)UZ * ncell;
                pv += (long)UZ * ncell;
            }
            for (; z < z1; ++z) {
                Lanes<T, VEC> lu = load_cells<T, VEC, NT>(pu);
                Lanes<T, VEC> lv = load_cells<T, VEC, NT>(pv);
                const double th = thickness[z];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    accU[k] = fma(th, fixed<T>(lu.x[k], fill), accU[k]);
                    accV[k] = fma(th, fixed<T>(lv.x[k], fill), accV[k]);
                }
                pu += ncell;
                pv += ncell;
            }
            // edge fluxes (field.py:195-196, 225-228)
            double eU[VEC], eV[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                eU[k] = +accU[k] * arcE[c0 + k];
                eV[k] = -accV[k] * arcN[c0 + k];
                if (sverdrup) {
                    eU[k] *= scale;
                    eV[k] *= scale;
                }
                tmax = fmax(tmax, fmax(fabs(eU[k]), fabs(eV[k])));
            }
            double *p0 = iV, *p1 = iV + ncell, *p2 = iV + 2 * ncell, *p3 = iV + 3 * ncell;
            double *aU = absUV, *aV = absUV + ncell;
            const unsigned j0 = (unsigned)(c0 / nx);
            const unsigned i0 = (unsigned)(c0 - (long)j0 * nx);
            if (VEC == 1) {
                p1[c0] = eU[0];
                p2[c0] = eV[0];
                aU[c0] = fabs(eU[0]);
                aV[c0] = fabs(eV[0]);
                if (j0 + 1 < ny) p0[c0 + nx] = eV[0];
                p3[(i0 + 1 < nx) ? c0 + 1 : c0 + 1 - nx] = eU[0];
            } else {
                // own slots and |.|: dense 16 B/lane stores (c0 is a multiple of VEC)
#pragma unroll
                for (int k = 0; k < VEC; k += 2) {
                    store2(p1 + c0 + k, eU[k], eU[k + 1], true);
                    store2(p2 + c0 + k, eV[k], eV[k + 1], true);
                    store2(aU + c0 + k, fabs(eU[k]), fabs(eU[k + 1]), true);
                    store2(aV + c0 + k, fabs(eV[k]), fabs(eV[k + 1]), true);
                }
                if (i0 + VEC <= nx) {
                    // lane's cells sit in one row: south slots of the row above = the same stream shifted by nx
                    if (j0 + 1 < ny) {
                        const bool al = (nx & 1u) == 0;
#pragma unroll
                        for (int k = 0; k < VEC; k += 2) store2(p0 + c0 + nx + k, eV[k], eV[k + 1], al);
                    }
                    // west slots of the cells to the right: shifted by one (8 B stores; the row's last cell
                    // wraps to column 0, field.py:223)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) p3[(i0 + k + 1 < nx) ? c0 + k + 1 : c0 + k + 1 - nx] = eU[k];
                } else {
                    // lane straddles a row end (nx % VEC != 0): per-cell bookkeeping
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const long c = c0 + k;
                        const unsigned j = (unsigned)(c / nx);
                        const unsigned i = (unsigned)(c - (long)j * nx);
                        if (j + 1 < ny) p0[c + nx] = eV[k];
                        p3[(i + 1 < nx) ? c + 1 : c + 1 - nx] = eU[k];
                    }
                }
            }
        }
    }
    // running max (field.py:234): wavefront butterfly, then one atomic per workgroup.  All values are
    // non-negative doubles, whose bit patterns order like unsigned integers.
    for (int o = 32; o > 0; o >>= 1) tmax = fmax(tmax, __shfl_xor(tmax, o, kWave));
    __shared__ double s_max[kBlock / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) s_max[threadIdx.x / kWave] = tmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
        if (m > 0.0) {
            unsigned long long b;
            __builtin_memcpy(&b, &m, 8);
            atomicMax(maxbits, b);
        }
    }
}

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <typename T, int VEC, int UZ, bool NT>
static int launch_flux_t(const FluxArgs &a, hipStream_t s)
{
    const long per_tile = (long)kBlock * VEC;
    const unsigned ntiles = (unsigned)((a.ncell + per_tile - 1) / per_tile);
    static const int xcd_map = env_int("NF_XCD_MAP", 1);
    const unsigned grid = xcd_map ? xcd_grid(ntiles) : ntiles;
    hipLaunchKernelGGL((k_flux<T, VEC, UZ, NT>), dim3(grid), dim3(kBlock), 0, s, (const T *)a.u, (const T *)a.v,
                       a.ncell, (unsigned)a.ny, (unsigned)a.nx, a.z0, a.z1, a.thickness, a.arcE, a.arcN,
                       (T)a.fill, a.scale, a.sverdrup, a.iV, a.absU, a.maxbits, ntiles, xcd_map);
    NF_HIP(hipGetLastError());
    return NF_OK;
}

template <typename T, int VEC>
static int launch_flux_v(const FluxArgs &a, hipStream_t s)
{
    static const int uz = env_int("NF_FLUX_UZ", 4);
    static const int nt = env_int("NF_FLUX_NT", 1);
    if (VEC == 1) return launch_flux_t<T, VEC, 4, false>(a, s);
    if (nt) {
        if (uz == 8) return launch_flux_t<T, VEC, 8, true>(a, s);
        if (uz == 4) return launch_flux_t<T, VEC, 4, true>(a, s);
        if (uz == 3) return launch_flux_t<T, VEC, 3, true>(a, s);
        return launch_flux_t<T, VEC, 5, true>(a, s);
    }
    if (uz == 8) return launch_flux_t<T, VEC, 8, false>(a, s);
    if (uz == 4) return launch_flux_t<T, VEC, 4, false>(a, s);
    if (uz == 3) return launch_flux_t<T, VEC, 3, false>(a, s);
    return launch_flux_t<T, VEC, 5, false>(a, s);
}

int launch_flux(const FluxArgs &a, hipStream_t s)
{
    NF_REQUIRE(a.ncell > 0 && a.ncell == a.ny * a.nx && a.ncell < (1l << 31), NF_ERR_ARG, "flux: bad grid sizes");
    NF_REQUIRE(a.z1 > a.z0 && a.z0 >= 0, NF_ERR_ARG, "flux: empty z range");
    NF_REQUIRE(a.absV == a.absU + a.ncell, NF_ERR_ARG, "flux: abs planes must be contiguous");
    const bool al16 = ((uintptr_t)a.u % 16 == 0) && ((uintptr_t)a.v % 16 == 0);
    if (a.dtype == NF_F64) {
        if (al16 && a.ncell % 2 == 0) return launch_flux_v<double, 2>(a, s);
        return launch_flux_v<double, 1>(a, s);
    } else if (a.dtype == NF_F32) {
        if (al16 && a.ncell % 4 == 0) return launch_flux_v<float, 4>(a, s);
        return launch_flux_v<float, 1>(a, s);
    }
    NF_REQUIRE(false, NF_ERR_ARG, "flux: dtype must be NF_F64 or NF_F32");
}

// ---- re-pack of the resident planes into the reference's (ncell,4) AoS (field.py:62) -----------------
__global__ __launch_bounds__(kBlock) void k_planes_to_aos(const double *__restrict__ planes, long ncell,
                                                          double *__restrict__ aos)
{
    long c = (long)blockIdx.x * kBlock + threadIdx.x;
    if (c < ncell) {
        double2 *o = reinterpret_cast<double2 *>(aos + 4 * c);
        o[0] = make_double2(planes[c], planes[ncell + c]);
        o[1] = make_double2(planes[2 * ncell + c], planes[3 * ncell + c]);
    }
}
int launch_planes_to_aos(const double *planes, long ncell, double *aos, hipStream_t s)
{
    hipLaunchKernelGGL(k_planes_to_aos, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       planes, ncell, aos);
    NF_HIP(hipGetLastError());
    return NF_OK;
}

}  // namespace nf
