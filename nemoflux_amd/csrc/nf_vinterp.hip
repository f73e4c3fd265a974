// nf_vinterp.hip -- point location and W2 (face / Piola) vector interpolation on the target line.
//
// Replaces  mint.VectorInterp.setGrid / buildLocator / findPoints / getFaceVectors as nemoflux drives them
//           (nemoflux/field.py:90-95 at construction, :119-120 at every update) -- python-mint >= 1.24.4, third-party,
//           not vendored: "parity unpinned" beyond README.md:36 (psi = x -> the arrows point down in y).
//
//   findPoints:  every target point (tried at x - periodX, x, x + periodX) is located in the cell with the lowest id
//                whose bilinear parameters (xi, eta) lie in [-tol, 1+tol]^2, tol = sqrt(tol2).  One wavefront owns 64
//                consecutive cells (corner rows staged through LDS, as in K2); the point list is wave-uniform; a wave
//                bounding box rejects almost every (tile, point) pair; winners are resolved with a 64-bit atomicMin
//                on the key cell*4 + shift, which is order-independent, so the result is deterministic.
//   getFaceVectors: one lane per located point,
//                V = [ (d3 (1-xi) + d1 xi) r_xi - (d0 (1-eta) + d2 eta) r_eta ] / J,  J = r_xi x r_eta
//                with d0..d3 = the cell's S,E,N,W edge data ((ncell,4) AoS or the engine's [4][ncell] planes).
#include "nf_common.h"

namespace nf {

constexpr int kNewtonMaxV = 16;
constexpr unsigned long long kNotFound = ~0ull;

__device__ inline void inv_bilinear_v(const double *v, double px, double py, double &xi0, double &xi1)
{
    const double ax = v[0], ay = v[1];
    const double e1x = v[2] - v[0], e1y = v[3] - v[1];
    const double e3x = v[6] - v[0], e3y = v[7] - v[1];
    const double hx = (v[0] - v[2]) + (v[4] - v[6]), hy = (v[1] - v[3]) + (v[5] - v[7]);
    double s = 0.5, t = 0.5;
    for (int it = 0; it < kNewtonMaxV; ++it) {
        const double fx = ((ax + s * e1x) + t * e3x) + (s * t) * hx - px;
        const double fy = ((ay + s * e1y) + t * e3y) + (s * t) * hy - py;
        const double j00 = e1x + t * hx, j01 = e3x + s * hx;
        const double j10 = e1y + t * hy, j11 = e3y + s * hy;
        const double det = j00 * j11 - j01 * j10;
        if (!(det != 0.0)) break;
        const double ds = (fx * j11 - fy * j01) / det;
        const double dt = (fy * j00 - fx * j10) / det;
        s -= ds;
        t -= dt;
        if (fabs(ds) + fabs(dt) < 1.e-15) break;
    }
    xi0 = s;
    xi1 = t;
}

__device__ inline double vmin64(double x)
{
    for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o, kWave));
    return x;
}
__device__ inline double vmax64(double x)
{
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, kWave));
    return x;
}

// targets are sorted by y (order[] gives the caller's index), so a wave only visits the points whose y falls in
// its tile's bounding box: two wave-uniform binary searches instead of a scan of all points
__global__ __launch_bounds__(kBlock) void k_find_points(const double *__restrict__ xy, long ncell,
                                                        const double *__restrict__ targets,
                                                        const long *__restrict__ order, long npts, int nshift,
                                                        double periodX, double tol, unsigned long long *best)
{
    __shared__ double s_xy[kBlock * 8];
    const int tid = threadIdx.x;
    const long c0 = (long)blockIdx.x * kBlock;
    const long nval = ncell * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        long g = c0 * 8 + tid + r * kBlock;
        if (g < nval) s_xy[tid + r * kBlock] = xy[g];
    }
    __syncthreads();
    const long c = c0 + tid;
    bool valid = c < ncell;
    double v[8];
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = valid ? s_xy[tid * 8 + k] : 0.0;
    valid = valid && quad_is_finite(v);            // NaN / infinite corners: not a cell
    unwrap_quad(v, nshift == 3 ? periodX : 0.0);   // date-line cells, as in K2 (nf_common.h)
    if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xmin = fmin(xmin, v[2 * i]);
            xmax = fmax(xmax, v[2 * i]);
            ymin = fmin(ymin, v[2 * i + 1]);
            ymax = fmax(ymax, v[2 * i + 1]);
        }
    }
    const double slack = valid ? 1.e-6 * (fabs(xmin) + fabs(xmax) + fabs(ymin) + fabs(ymax) + 1.0) : 0.0;
    const bool unusable = valid && quad_is_nonconvex(v);
    const double wslack = vmax64(slack);
    const double wxmin = vmin64(xmin) - wslack, wxmax = vmax64(xmax) + wslack;
    const double wymin = vmin64(ymin) - wslack, wymax = vmax64(ymax) + wslack;
    long lo = 0, hi = npts;  // first point with y >= wymin
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (targets[3 * mid + 1] < wymin) lo = mid + 1;
        else hi = mid;
    }
    for (long q = lo; q < npts; ++q) {
        const double tx = targets[3 * q], ty = targets[3 * q + 1];
        if (ty > wymax) break;  // wave-uniform
        const long p = order[q];
        for (int k = 0; k < nshift; ++k) {
            const double px = tx + (nshift == 3 ? k - 1 : 0) * periodX;
            if (px < wxmin || px > wxmax) continue;  // wave-uniform
            if (!valid || px < xmin - slack || px > xmax + slack || ty < ymin - slack || ty > ymax + slack) continue;
            if (unusable) continue;   // no inverse bilinear map in a non-convex / pole-vertex cell: the point is "not found" there
            double xi, eta;
            inv_bilinear_v(v, px, ty, xi, eta);
            if (xi >= -tol && xi <= 1.0 + tol && eta >= -tol && eta <= 1.0 + tol) {
                // Newton must have converged onto the point
                const double mx = ((v[0] + xi * (v[2] - v[0])) + eta * (v[6] - v[0])) + (xi * eta) * ((v[0] - v[2]) + (v[4] - v[6])) - px;
                const double my = ((v[1] + xi * (v[3] - v[1])) + eta * (v[7] - v[1])) + (xi * eta) * ((v[1] - v[3]) + (v[5] - v[7])) - ty;
                if (fabs(mx) + fabs(my) <= 1.e-9 * ((xmax - xmin) + (ymax - ymin)))
                    atomicMin(&best[p], (unsigned long long)c * 4 + (unsigned long long)k);
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_locate_finish(const double *__restrict__ xy,
                                                          const double *__restrict__ targets, long npts, int nshift,
                                                          double periodX, const unsigned long long *__restrict__ best,
                                                          long *__restrict__ cell, double *__restrict__ pcoords)
{
    const long p = (long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= npts) return;
    const unsigned long long key = best[p];
    if (key == kNotFound) {
        cell[p] = -1;
        pcoords[2 * p] = pcoords[2 * p + 1] = 0.0;
        return;
    }
    const long c = (long)(key >> 2);
    const int k = (int)(key & 3);
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = xy[c * 8 + i];
    unwrap_quad(v, nshift == 3 ? periodX : 0.0);
    double xi, eta;
    inv_bilinear_v(v, targets[3 * p] + (nshift == 3 ? k - 1 : 0) * periodX, targets[3 * p + 1], xi, eta);
    cell[p] = c;
    pcoords[2 * p] = xi;
    pcoords[2 * p + 1] = eta;
}

__global__ __launch_bounds__(kBlock) void k_face_vectors(const double *__restrict__ xy, const long *__restrict__ cell,
                                                         const double *__restrict__ pcoords, long npts,
                                                         const double *__restrict__ data, long ncell, int planes,
                                                         double periodX, double *__restrict__ vectors)
{
    const long p = (long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= npts) return;
    const long c = cell[p];
    double vx = 0.0, vy = 0.0;
    if (c >= 0) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = xy[c * 8 + i];
        unwrap_quad(v, periodX);   // the tangent vectors of a date-line cell are those of the unwrapped quad
        const double xi = pcoords[2 * p], eta = pcoords[2 * p + 1];
        double d0, d1, d2, d3;
        if (planes == 2) {          // rows gathered per target point on the host: (npts,4)
            d0 = data[4 * p];
            d1 = data[4 * p + 1];
            d2 = data[4 * p + 2];
            d3 = data[4 * p + 3];
        } else if (planes) {
            d0 = data[c];
            d1 = data[ncell + c];
            d2 = data[2 * ncell + c];
            d3 = data[3 * ncell + c];
        } else {
            d0 = data[4 * c];
            d1 = data[4 * c + 1];
            d2 = data[4 * c + 2];
            d3 = data[4 * c + 3];
        }
        const double rxx = (1.0 - eta) * (v[2] - v[0]) + eta * (v[4] - v[6]);
        const double rxy = (1.0 - eta) * (v[3] - v[1]) + eta * (v[5] - v[7]);
        const double rex = (1.0 - xi) * (v[6] - v[0]) + xi * (v[4] - v[2]);
        const double rey = (1.0 - xi) * (v[7] - v[1]) + xi * (v[5] - v[3]);
        const double jac = rxx * rey - rxy * rex;
        const double fx = d3 * (1.0 - xi) + d1 * xi;
        const double fe = d0 * (1.0 - eta) + d2 * eta;
        vx = (fx * rxx - fe * rex) / jac;
        vy = (fx * rxy - fe * rey) / jac;
    }
    vectors[3 * p] = vx;
    vectors[3 * p + 1] = vy;
    vectors[3 * p + 2] = 0.0;
}

int launch_find_points(const double *xy, long ncell, const double *targets_dev, const double *sorted_dev,
                       const long *order_dev, long npts, double periodX, double tol2, unsigned long long *best_dev,
                       long *cell_dev, double *pcoords_dev, hipStream_t s)
{
    if (npts == 0) return NF_OK;
    const int nshift = periodX > 0.0 ? 3 : 1;
    NF_HIP(hipMemsetAsync(best_dev, 0xff, sizeof(unsigned long long) * npts, s));
    hipLaunchKernelGGL(k_find_points, dim3((unsigned)((ncell + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, xy, ncell,
                       sorted_dev, order_dev, npts, nshift, periodX, sqrt(tol2), best_dev);
    hipLaunchKernelGGL(k_locate_finish, dim3((unsigned)((npts + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, xy,
                       targets_dev, npts, nshift, periodX, best_dev, cell_dev, pcoords_dev);
    NF_HIP(hipGetLastError());
    return NF_OK;
}

int launch_face_vectors(const double *xy, const long *cell_dev, const double *pcoords_dev, long npts, const double *data,
                        long ncell, int planes, double periodX, double *vectors_dev, hipStream_t s)
{
    if (npts == 0) return NF_OK;
    hipLaunchKernelGGL(k_face_vectors, dim3((unsigned)((npts + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, xy, cell_dev,
                       pcoords_dev, npts, data, ncell, planes, periodX, vectors_dev);
    NF_HIP(hipGetLastError());
    return NF_OK;
}

}  // namespace nf
