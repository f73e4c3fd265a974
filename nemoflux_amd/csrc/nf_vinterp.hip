// nf_vinterp.hip -- point location and W2 (face / Piola) vector interpolation on the target line.
//
// Replaces  mint.VectorInterp.setGrid / buildLocator / findPoints / getFaceVectors as nemoflux drives them
//           (nemoflux/field.py:90-95 at construction, :119-120 at every update) -- python-mint >= 1.24.4, third-party,
//           not vendored: "parity unpinned" beyond README.md:36 (psi = x -> the arrows point down in y).
//
//   findPoints:  every target point (tried at x - periodX, x, x + periodX) is located in the cell with the lowest id
//                whose bilinear parameters (xi, eta) lie in [-tol, 1+tol]^2, tol = sqrt(tol2).  The candidates come from
//                the grid's locator (nf_locator.h: the box hierarchy of the weight build, walked by all point images at
//                once, one lane per (group, point, child)); one lane per (cell, point image) candidate then applies the
//                exact test; winners are resolved with a 64-bit atomicMin on the key cell*4 + shift, which is
//                order-independent, so the result is deterministic.  (Until round 5 every 64-cell tile scanned the points
//                of its latitude range: 746 ms for 2 M points on the ORCA12-like grid; tools/findpoints_timing.py.)
//   getFaceVectors: one lane per located point,
//                V = [ (d3 (1-xi) + d1 xi) r_xi - (d0 (1-eta) + d2 eta) r_eta ] / J,  J = r_xi x r_eta
//                with d0..d3 = the cell's S,E,N,W edge data ((ncell,4) AoS or the engine's [4][ncell] planes).
#include "nf_common.h"
#include "nf_locator.h"

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

// the walk's test for points: lane t = child (t % 16) of pair (t / 16).  An image is point img / nshift moved by one of the
// nshift periods.  A point within the per-cell slack of the exact test below (1e-6 x the cell's coordinates) of a cell is
// within this margin of every box that holds the cell.
__global__ __launch_bounds__(kBlock) void k_walk_count_points(const int *__restrict__ pnode, const int *__restrict__ pimg, long np,
                                                              const Box4 *__restrict__ box, Layout lay,
                                                              const double *__restrict__ targets, int nshift, double periodX,
                                                              unsigned long long *__restrict__ wmask, int *__restrict__ bcnt)
{
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    const long p = t / kFan;
    bool pass = false;
    if (p < np) {
        const long child = child_of(lay, pnode ? pnode[p] : 0, (int)(t & (kFan - 1)));
        if (child >= 0) {
            const int img = pimg ? pimg[p] : (int)p;
            const int q = img / nshift, k = img - q * nshift;
            const double px = targets[3 * (long)q] + (nshift == 3 ? k - 1 : 0) * periodX, py = targets[3 * (long)q + 1];
            const Box4 b = box[child];
            const double m = 1.e-6 * (2.0 * fmax(fabs((double)b.xmin), fabs((double)b.xmax)) +
                                      2.0 * fmax(fabs((double)b.ymin), fabs((double)b.ymax)) + 1.0);
            pass = px >= b.xmin - m && px <= b.xmax + m && py >= b.ymin - m && py <= b.ymax + m;
        }
    }
    block_count(__ballot(pass), wmask, bcnt);
}

// the exact test, one lane per (cell, point image) candidate
__global__ __launch_bounds__(kBlock) void k_locate_pairs(const double *__restrict__ xy, const int *__restrict__ ccell,
                                                         const int *__restrict__ cimg, long nc,
                                                         const double *__restrict__ targets, int nshift, double periodX,
                                                         double tol, unsigned long long *best)
{
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= nc) return;
    const long c = ccell[t];
    const int img = cimg[t];
    const int q = img / nshift, k = img - q * nshift;
    const double px = targets[3 * (long)q] + (nshift == 3 ? k - 1 : 0) * periodX, ty = targets[3 * (long)q + 1];
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = xy[c * 8 + i];
    if (!quad_is_finite(v)) return;                // NaN / infinite corners: not a cell
    unwrap_quad(v, nshift == 3 ? periodX : 0.0);   // date-line cells, as in K2 (nf_common.h)
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        xmin = fmin(xmin, v[2 * i]);
        xmax = fmax(xmax, v[2 * i]);
        ymin = fmin(ymin, v[2 * i + 1]);
        ymax = fmax(ymax, v[2 * i + 1]);
    }
    const double slack = 1.e-6 * (fabs(xmin) + fabs(xmax) + fabs(ymin) + fabs(ymax) + 1.0);
    if (px < xmin - slack || px > xmax + slack || ty < ymin - slack || ty > ymax + slack) return;
    if (quad_is_nonconvex(v)) return;   // no inverse bilinear map in a non-convex / pole-vertex cell: the point is "not found" there
    double xi, eta;
    inv_bilinear_v(v, px, ty, xi, eta);
    if (xi >= -tol && xi <= 1.0 + tol && eta >= -tol && eta <= 1.0 + tol) {
        // Newton must have converged onto the point
        const double mx = ((v[0] + xi * (v[2] - v[0])) + eta * (v[6] - v[0])) + (xi * eta) * ((v[0] - v[2]) + (v[4] - v[6])) - px;
        const double my = ((v[1] + xi * (v[3] - v[1])) + eta * (v[7] - v[1])) + (xi * eta) * ((v[1] - v[3]) + (v[5] - v[7])) - ty;
        if (fabs(mx) + fabs(my) <= 1.e-9 * ((xmax - xmin) + (ymax - ymin)))
            atomicMin(&best[q], (unsigned long long)c * 4 + (unsigned long long)k);
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

int launch_find_points(const double *xy, long ncell, long row_length, LocatorBoxes *keep, const double *targets_dev, long npts,
                       double periodX, double tol2, unsigned long long *best_dev, long *cell_dev, double *pcoords_dev,
                       hipStream_t s)
{
    if (npts == 0) return NF_OK;
    const int nshift = periodX > 0.0 ? 3 : 1;
    NF_REQUIRE(npts * nshift < (1l << 31), NF_ERR_ARG, "findPoints: too many target points for one call (2^31 / 3)");
    NF_HIP(hipMemsetAsync(best_dev, 0xff, sizeof(unsigned long long) * npts, s));
    ScratchLease lease;
    Walker wk(*lease.sc, s);
    NF_TRY(wk.prepare(xy, ncell, nshift == 3 ? periodX : 0.0, row_length, keep));
    int *c_cell = nullptr, *c_img = nullptr;
    long nc = 0;
    NF_TRY(wk.walk(npts * nshift,
                   [&](int, const int *pn, const int *pi, long npairs, const Box4 *child_boxes, Layout lay, unsigned nb,
                       unsigned long long *mask, int *cnt) {
                       hipLaunchKernelGGL(k_walk_count_points, dim3(nb), dim3(kBlock), 0, s, pn, pi, npairs, child_boxes, lay,
                                          targets_dev, nshift, periodX, mask, cnt);
                   },
                   &c_cell, &c_img, &nc));
    if (nc > 0)
        hipLaunchKernelGGL(k_locate_pairs, dim3((unsigned)((nc + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, xy, (const int *)c_cell,
                           (const int *)c_img, nc, targets_dev, nshift, periodX, sqrt(tol2), best_dev);
    hipLaunchKernelGGL(k_locate_finish, dim3((unsigned)((npts + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, xy,
                       targets_dev, npts, nshift, periodX, best_dev, cell_dev, pcoords_dev);
    NF_HIP(hipGetLastError());
    NF_HIP(hipStreamSynchronize(s));
    lease.drained = true;
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
