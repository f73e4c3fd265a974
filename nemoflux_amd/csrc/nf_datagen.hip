// nf_datagen.hip -- on-device counterpart of nemoflux/datagen.py.
//
// Replaces  datagen.py:42-66   buildUniformHorizontal: uniform lon/lat mesh + (ny,nx,4) cell bounds
//           datagen.py:116-166 rotatePole: displaced-pole curvilinear grid incl. the date-line fix
//           datagen.py:69-113  applyStreamFunction + computeUVFromPotential:
//                              u = (psi2 - psi1)/ds21,  v = -(psi2 - psi3)/ds23,  ds23 >= 1e-12
// The configs of BASELINE.json beyond 360x180 cannot be generated on the host (93 GB for ORCA12 x 12 steps;
// the reference's rotatePole is a Python triple loop), so the generator lives on the GPU.  There is no
// eval() on the device: psi comes from the fixed menu of include/nemoflux_amd.h, each entry evaluated in the
// SAME operation order as the Python expression it names (no fma contraction), psi = post(g(z,t) * h(x,y)).
#include <vector>

#include "nf_common.h"

namespace nf {

constexpr double kPi = 3.14159265358979323846;

__device__ inline double psi_h(int psi, double x, double y)
{
    switch (psi) {
        case 0: return x;                                                     // "x"
        case 1: return atan2(y, x + 180.0) / (2.0 * kPi);                     // "arctan2(y, x+180)/(2*pi)"
        case 2:
        case 3: return cos(2.0 * kPi * y / 360.0) + sin(2.0 * kPi * x / 360.0);
        case 4: { double q = y / 180.0; return 0.5 * (q * q) + sin(2.0 * kPi * x / 360.0); }
        case 5: return atan2(y, x + 180.0);
    }
    return 0.0;
}
__host__ __device__ inline double psi_g(int psi, double z, long t, long nt)
{
    switch (psi) {
        case 3:
        case 5: return (1.0 + 10.0 * z) * (double)(t + 1);
        case 4: return cos((double)(t * 2) * kPi / (double)nt) + 2.0;
    }
    return 1.0;
}
__device__ inline double psi_pot(int psi, double g, double h)
{
    switch (psi) {
        case 3:
        case 4: return g * h;
        case 5: return g * h / (2.0 * kPi);
    }
    return h;
}

struct Rot {
    double m[9];
    int on;
};

// datagen.py:138-166 for one corner
__device__ inline void rotate_corner(const Rot &R, double &lon, double &lat)
{
    const double the = kPi * lat / 180.0;
    const double lam = kPi * lon / 180.0;
    const double cos_the = cos(the), sin_the = sin(the);
    const double rho = cos_the;
    const double cos_lam = cos(lam), sin_lam = sin(lam);
    const double xo = rho * cos_lam, yo = rho * sin_lam, zo = sin_the;
    const double xn = (R.m[0] * xo + R.m[1] * yo) + R.m[2] * zo;
    const double yn = (R.m[3] * xo + R.m[4] * yo) + R.m[5] * zo;
    double zn = (R.m[6] * xo + R.m[7] * yo) + R.m[8] * zo;
    zn = fmin(1.0, fmax(-1.0, zn));
    lat = 180.0 * asin(zn) / kPi;
    lon = 180.0 * atan2(yn, xn) / kPi;
}

__global__ __launch_bounds__(kBlock) void k_bounds(double *__restrict__ blon, double *__restrict__ blat,
                                                   unsigned ny, unsigned nx, double xmin, double ymin, double dx,
                                                   double dyy, Rot R)
{
    const long c = (long)blockIdx.x * kBlock + threadIdx.x;
    if (c >= (long)ny * nx) return;
    const unsigned j = (unsigned)(c / nx), i = (unsigned)(c - (long)j * nx);
    const unsigned di[4] = {0, 1, 1, 0}, dj[4] = {0, 0, 1, 1};  // datagen.py:56-66
    double lon[4], lat[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        lon[v] = xmin + (double)(i + di[v]) * dx;   // datagen.py:48
        lat[v] = ymin + (double)(j + dj[v]) * dyy;  // datagen.py:49 (dyy = dx in the reference)
        if (R.on) {
            rotate_corner(R, lon[v], lat[v]);
            if (v > 0) {  // date line fix relative to corner 0 (datagen.py:162-166)
                const double dLon = lon[v] - lon[0];
                if (dLon > +270.0) lon[v] -= 360.0;
                if (dLon < -270.0) lon[v] += 360.0;
            }
        }
    }
    double2 *pl = reinterpret_cast<double2 *>(blon + 4 * c);
    double2 *pa = reinterpret_cast<double2 *>(blat + 4 * c);
    pl[0] = make_double2(lon[0], lon[1]);
    pl[1] = make_double2(lon[2], lon[3]);
    pa[0] = make_double2(lat[0], lat[1]);
    pa[1] = make_double2(lat[2], lat[3]);
}

int launch_datagen_bounds(double *blon, double *blat, long ny, long nx, double xmin, double xmax, double ymin,
                          double ymax, double dlon, double dlat, int lat_uses_dx, hipStream_t s)
{
    NF_REQUIRE(ny > 0 && nx > 0 && ny * nx < (1l << 31), NF_ERR_ARG, "datagen: bad sizes");
    const double dy = (ymax - ymin) / (double)ny, dx = (xmax - xmin) / (double)nx;  // datagen.py:45
    Rot R{};
    R.on = (dlon != 0.0 || dlat != 0.0);
    if (R.on) {  // datagen.py:121-135; every entry of rot_bet.rot_alp is a single product
        const double alpha = kPi * dlat / 180.0, beta = kPi * dlon / 180.0;
        const double ca = cos(alpha), sa = sin(alpha), cb = cos(beta), sb = sin(beta);
        const double m[9] = {cb * ca, sb, cb * sa, -sb * ca, cb, -sb * sa, -sa, 0.0, ca};
        for (int k = 0; k < 9; ++k) R.m[k] = m[k];
    }
    const long ncell = ny * nx;
    hipLaunchKernelGGL(k_bounds, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, blon, blat,
                       (unsigned)ny, (unsigned)nx, xmin, ymin, dx, lat_uses_dx ? dx : dy, R);
    NF_HIP(hipGetLastError());
    return NF_OK;
}

// h(x,y) at the (ny+1) x (nx+1) mesh nodes
__global__ __launch_bounds__(kBlock) void k_node_h(double *__restrict__ h, unsigned ny1, unsigned nx1, double xmin,
                                                   double ymin, double dx, double dyy, int psi)
{
    const long n = (long)blockIdx.x * kBlock + threadIdx.x;
    if (n >= (long)ny1 * nx1) return;
    const unsigned jj = (unsigned)(n / nx1), ii = (unsigned)(n - (long)jj * nx1);
    h[n] = psi_h(psi, xmin + (double)ii * dx, ymin + (double)jj * dyy);
}

__device__ inline void lonlat_xyz(double lon, double lat, double &x, double &y, double &z)
{
    const double lam = lon * kDeg2Rad, the = lat * kDeg2Rad;  // geo.py:15-21
    const double rho = 1.0 * cos(the);
    x = rho * cos(lam);
    y = rho * sin(lam);
    z = 1.0 * sin(the);
}

// ds21, ds23 per cell on the LOGICAL (un-rotated) mesh (datagen.py:89-104)
__global__ __launch_bounds__(kBlock) void k_ds(double *__restrict__ ds21, double *__restrict__ ds23, unsigned ny,
                                               unsigned nx, double xmin, double ymin, double dx, double dyy)
{
    const long c = (long)blockIdx.x * kBlock + threadIdx.x;
    if (c >= (long)ny * nx) return;
    const unsigned j = (unsigned)(c / nx), i = (unsigned)(c - (long)j * nx);
    const double x0 = xmin + (double)i * dx, x1 = xmin + (double)(i + 1) * dx;
    const double y0 = ymin + (double)j * dyy, y1 = ymin + (double)(j + 1) * dyy;
    double ax, ay, az, bx, by, bz, cx, cy, cz;
    lonlat_xyz(x1, y0, ax, ay, az);  // pp1
    lonlat_xyz(x1, y1, bx, by, bz);  // pp2
    lonlat_xyz(x0, y1, cx, cy, cz);  // pp3
    const double d21 = fabs(1.0 * acos(((bx * ax + by * ay) + bz * az) / 1.0));
    double d23 = fabs(1.0 * acos(((bx * cx + by * cy) + bz * cz) / 1.0));
    if (d23 < 1.e-12) d23 = 1.e-12;  // datagen.py:104
    ds21[c] = d21;
    ds23[c] = d23;
}

// ---- u, v of all slabs: the write stream of the generator (93 GB for the bench workload) -----------------------------
// Division is the arithmetic of this kernel (u = dpsi/ds21, v = -dpsi/ds23, psi = g h / (2 pi)), and its divisors do not
// depend on the slab: ds21, ds23 belong to the cell, 2 pi is a constant.  The compiler's own float64 division on gfx950 is
//      r = rcp(d); e = fma(-d, r, 1); r = fma(r, e, r); e = fma(-d, r, 1); r = fma(r, e, r);      <- divisor only
//      q = n * r;  rem = fma(-d, q, n);  n / d = fma(rem, r, q)                                   <- per quotient
// (wrapped in v_div_scale / v_div_fmas / v_div_fixup, which change nothing unless an operand sits near the ends of the
// exponent range or the numerator is zero).  So the divisor half is done ONCE per cell and kept in registers and every slab
// pays three operations per quotient instead of eleven plus a quarter-rate reciprocal -- the same operations on the same
// operands, hence the same bits (test_device_datagen_vs_reference, test_datagen_fast_path_equals_plain_division).
// Operands outside [2^-400, 2^400] (never the case for the menu's stream functions) take the plain division.
struct Divisor {
    double d, r;
    bool plain;   // divisor outside the range in which the hardware sequence runs unscaled
};
__device__ inline Divisor make_divisor(double d)
{
    Divisor x;
    x.d = d;
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    x.r = r;
    x.plain = !(d >= 0x1p-400 && d <= 0x1p400);
    return x;
}
__device__ inline double divide(double n, const Divisor &x)
{
    const double q = n * x.r;
    const double rem = fma(-x.d, q, n);
    double res = fma(rem, x.r, q);
    const double an = fabs(n);
    if (x.plain || !(an >= 0x1p-400 && an <= 0x1p400)) {   // zero, tiny, huge, NaN: rare
        double nn = n;
        asm volatile("" : "+v"(nn));   // keeps this a real branch (skipped by the whole wave): without it the compiler
        res = nn / x.d;                // computes BOTH forms for every quotient and selects
    }
    return res;
}
// POT: how psi follows from g(z,t) and h(x,y) -- 0: h, 1: g*h, 2: g*h/(2 pi)  (psi_pot, as a compile-time choice)
__host__ __device__ inline int psi_pot_kind(int psi) { return psi == 5 ? 2 : (psi == 3 || psi == 4) ? 1 : 0; }
template <int POT>
__device__ inline double psi_pot_div(double g, double h, const Divisor &twopi)
{
    if (POT == 2) return divide(g * h, twopi);   // g * h / (2.0 * kPi)
    if (POT == 1) return g * h;
    return h;
}

// one cell per lane: any shape, any alignment (and the reference for the row kernel below)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_uv(T *__restrict__ u, T *__restrict__ v,
                                               const double *__restrict__ h, const double *__restrict__ ds21,
                                               const double *__restrict__ ds23, const double *__restrict__ gtab,
                                               unsigned ny, unsigned nx, int psi)
{
    const long c = (long)blockIdx.x * kBlock + threadIdx.x;
    const long ncell = (long)ny * nx;
    if (c >= ncell) return;
    const long slab = blockIdx.y;  // (t - t0) * nz + k
    const double g = gtab[slab];
    const unsigned j = (unsigned)(c / nx), i = (unsigned)(c - (long)j * nx);
    const long nx1 = (long)nx + 1;
    const double p1 = psi_pot(psi, g, h[(long)j * nx1 + i + 1]);        // corner 1
    const double p2 = psi_pot(psi, g, h[(long)(j + 1) * nx1 + i + 1]);  // corner 2
    const double p3 = psi_pot(psi, g, h[(long)(j + 1) * nx1 + i]);      // corner 3
    u[slab * ncell + c] = (T)((p2 - p1) / ds21[c]);   // datagen.py:107,110
    v[slab * ncell + c] = (T)(-(p2 - p3) / ds23[c]);  // datagen.py:108,113
}

// A lane owns VEC consecutive cells of ONE grid row (16 bytes of output: 2 x float64 or 4 x float32; nx % VEC == 0) and
// walks `chunk` slabs: node values h and the divisors' reciprocals are loaded / refined once and stay in registers; per
// slab the wavefront stores 1 KiB contiguous to u and to v with non-temporal 16-byte stores (a write-once stream).
template <typename T, int VEC, int POT>
__global__ __launch_bounds__(kBlock) void k_uv_rows(T *__restrict__ u, T *__restrict__ v,
                                                    const double *__restrict__ h, const double *__restrict__ ds21,
                                                    const double *__restrict__ ds23, const double *__restrict__ gtab,
                                                    unsigned ny, unsigned nx, long nslab, int chunk)
{
    typedef T vecT __attribute__((ext_vector_type(VEC)));
    const long ncell = (long)ny * nx;
    const long c0 = ((long)blockIdx.x * kBlock + threadIdx.x) * VEC;
    if (c0 >= ncell) return;
    const unsigned j = (unsigned)(c0 / nx), i = (unsigned)(c0 - (long)j * nx);
    const long nx1 = (long)nx + 1;
    double hlo[VEC], hhi[VEC + 1];          // nodes (j, i+1 .. i+VEC) and (j+1, i .. i+VEC)
    Divisor d21[VEC], d23[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        hlo[k] = h[(long)j * nx1 + i + 1 + k];
        d21[k] = make_divisor(ds21[c0 + k]);
        d23[k] = make_divisor(ds23[c0 + k]);
    }
#pragma unroll
    for (int k = 0; k <= VEC; ++k) hhi[k] = h[(long)(j + 1) * nx1 + i + k];
    const Divisor twopi = make_divisor(2.0 * kPi);
    const long s0 = (long)blockIdx.y * chunk;
    const long s1 = s0 + chunk < nslab ? s0 + chunk : nslab;
    T *pu = u + s0 * ncell + c0, *pv = v + s0 * ncell + c0;
    for (long s = s0; s < s1; ++s) {
        const double g = gtab[s];           // wave-uniform: a scalar load
        double plo[VEC], phi[VEC + 1];
#pragma unroll
        for (int k = 0; k < VEC; ++k) plo[k] = psi_pot_div<POT>(g, hlo[k], twopi);
#pragma unroll
        for (int k = 0; k <= VEC; ++k) phi[k] = psi_pot_div<POT>(g, hhi[k], twopi);
        vecT ou, ov;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            ou[k] = (T)divide(phi[k + 1] - plo[k], d21[k]);       // (p2 - p1) / ds21      datagen.py:107,110
            ov[k] = (T)divide(-(phi[k + 1] - phi[k]), d23[k]);    // -(p2 - p3) / ds23     datagen.py:108,113
        }
        __builtin_nontemporal_store(ou, reinterpret_cast<vecT *>(pu));
        __builtin_nontemporal_store(ov, reinterpret_cast<vecT *>(pv));
        pu += ncell;
        pv += ncell;
    }
}

// 1 = the row kernel where shape and alignment allow (default); 0 = always one cell per lane (the test's reference path)
static int g_uv_rows = 1;
void datagen_use_rows(int on) { g_uv_rows = on; }

int launch_datagen_uv(void *u, void *v, int dtype, long t0, long t1, long nt, long nz, long ny, long nx,
                      double xmin, double xmax, double ymin, double ymax, double zmin, double zmax,
                      int lat_uses_dx, int psi, hipStream_t s)
{
    NF_REQUIRE(psi >= 0 && psi < NF_PSI_COUNT, NF_ERR_ARG, "datagen: unknown stream function id");
    NF_REQUIRE(ny > 0 && nx > 0 && nz > 0 && t1 >= t0 && ny * nx < (1l << 31), NF_ERR_ARG, "datagen: bad sizes");
    NF_REQUIRE((t1 - t0) * nz < 65536, NF_ERR_ARG, "datagen: more than 65535 slabs per call");
    const double dy = (ymax - ymin) / (double)ny, dx = (xmax - xmin) / (double)nx;
    const double dyy = lat_uses_dx ? dx : dy;
    const double dz = (zmax - zmin) / (double)nz;  // datagen.py:35
    const long ncell = ny * nx, nnode = (ny + 1) * (nx + 1);
    double *h = nullptr, *ds21 = nullptr, *ds23 = nullptr, *gtab = nullptr;
    struct FreeOnExit {       // every return path, the early ones of NF_HIP included (round-4 advisor)
        double *&a, *&b, *&c, *&d;
        ~FreeOnExit()
        {
            for (double *p : {a, b, c, d})
                if (p) (void)hipFree(p);
        }
    } cleanup{h, ds21, ds23, gtab};
    NF_HIP(hipMalloc((void **)&h, sizeof(double) * nnode));
    NF_HIP(hipMalloc((void **)&ds21, sizeof(double) * ncell));
    NF_HIP(hipMalloc((void **)&ds23, sizeof(double) * ncell));
    hipLaunchKernelGGL(k_node_h, dim3((unsigned)((nnode + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, h,
                       (unsigned)(ny + 1), (unsigned)(nx + 1), xmin, ymin, dx, dyy, psi);
    const unsigned nb = (unsigned)((ncell + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_ds, dim3(nb), dim3(kBlock), 0, s, ds21, ds23, (unsigned)ny, (unsigned)nx, xmin, ymin, dx,
                       dyy);
    // g(z, t) of every slab: a table (the lanes of a slab share it, and psi 4's cosine is then evaluated once per slab)
    const long nslab = (t1 - t0) * nz;
    std::vector<double> gh((size_t)(nslab ? nslab : 1));
    for (long sl = 0; sl < nslab; ++sl) {
        const long t = t0 + sl / nz, k = sl % nz;
        gh[(size_t)sl] = psi_g(psi, zmin + ((double)k + 0.5) * dz, t, nt);   // z: datagen.py:38
    }
    NF_HIP(hipMalloc((void **)&gtab, sizeof(double) * gh.size()));
    NF_HIP(hipMemcpyAsync(gtab, gh.data(), sizeof(double) * gh.size(), hipMemcpyHostToDevice, s));
    if (nslab > 0) {
        const int vec = dtype == NF_F64 ? 2 : 4;
        const bool rows = g_uv_rows && nx % vec == 0 && (uintptr_t)u % 16 == 0 && (uintptr_t)v % 16 == 0;
        if (rows) {
            const int chunk = 25;   // slabs per workgroup: the per-cell prologue (7 loads, 2 reciprocals) is paid once per chunk
            dim3 grid((unsigned)((ncell / vec + kBlock - 1) / kBlock), (unsigned)((nslab + chunk - 1) / chunk));
            const int pot = psi_pot_kind(psi);
#define NF_UV_ROWS(T, VEC, POT)                                                                                          \
    hipLaunchKernelGGL((k_uv_rows<T, VEC, POT>), grid, dim3(kBlock), 0, s, (T *)u, (T *)v, h, ds21, ds23, gtab, (unsigned)ny, \
                       (unsigned)nx, nslab, chunk)
            if (dtype == NF_F64) {
                if (pot == 2) NF_UV_ROWS(double, 2, 2);
                else if (pot == 1) NF_UV_ROWS(double, 2, 1);
                else NF_UV_ROWS(double, 2, 0);
            } else {
                if (pot == 2) NF_UV_ROWS(float, 4, 2);
                else if (pot == 1) NF_UV_ROWS(float, 4, 1);
                else NF_UV_ROWS(float, 4, 0);
            }
#undef NF_UV_ROWS
        } else {
            dim3 grid(nb, (unsigned)nslab);      // nslab < 65536: checked on entry
            if (dtype == NF_F64)
                hipLaunchKernelGGL(k_uv<double>, grid, dim3(kBlock), 0, s, (double *)u, (double *)v, h, ds21, ds23, gtab,
                                   (unsigned)ny, (unsigned)nx, psi);
            else
                hipLaunchKernelGGL(k_uv<float>, grid, dim3(kBlock), 0, s, (float *)u, (float *)v, h, ds21, ds23, gtab,
                                   (unsigned)ny, (unsigned)nx, psi);
        }
    }
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(s);     // the kernels are done before the tables go (gh, too, is read by a copy)
    NF_HIP(e);
    NF_HIP(e2);
    return NF_OK;
}

}  // namespace nf
