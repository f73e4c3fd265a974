/*
 * nf_oracle.c -- CPU restatement of the nemoflux transect-flux hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in nemoflux_amd/ (the product) may include, link, import or
 * execute this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference):
 *   A1  nfo_assemble_points      nemoflux/horizgrid.py:17-22
 *   A3  nfo_arc_lengths          nemoflux/field.py:170-181, nemoflux/geo.py:14-27 (R = 1, geo.py:3)
 *   A4  nfo_vertical_integral    nemoflux/field.py:145-163 (fillna(0) :157, tensordot :161)
 *   A5  nfo_edge_flux            nemoflux/field.py:183-234
 *   A6  nfo_polyline_weights     mint.PolylineIntegral.computeWeights as driven by field.py:45-48
 *   A7  nfo_get_integral         mint.PolylineIntegral.getIntegral as driven by field.py:102
 *
 * Parity pinning:
 *   A1,A3,A4,A5 are pinned by tests/golden (npz files), which were produced by RUNNING the reference's own
 *   datagen.py / field.py / geo.py (oracle/gen_golden.py).
 *   A6,A7 restate a THIRD-PARTY dependency that is absent from /root/reference and from this image:
 *   conda-forge python-mint >= 1.24.4 (README.md:12; no lock file, no vendored source).  The
 *   reference holds no tests for it.  The restatement follows the published algorithm (SURVEY.md
 *   section 8a row A6: planar lon/lat, per-cell clip of every target segment, inverse bilinear map,
 *   four edge weights, 1/n multiplicity for sub-segments shared by n cells, periodicity in x) and is
 *   pinned by the reference's README known answers only (360, 0.5, 0, ~1e-15, ~1e-11:
 *   tests/golden/known_answers.json) plus the path-independence property README.md:45,58 states.
 *   Edge-case behaviour beyond those (tolerances, non-convex cells) is "parity unpinned".
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off).  fma() is used explicitly where the
 * HIP kernels use it so the two can be compared bit-for-bit; no other contraction is allowed.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define NFO_DEG2RAD (3.14159265358979323846 / 180.0) /* geo.py:4 numpy.pi/180. */
#define NFO_EARTH_RADIUS_GEO 1.0                     /* geo.py:3 */
#define NFO_EARTH_RADIUS_SV 6371000.0                /* field.py:12 */

/* ---- A1: horizgrid.py:17-22 : points[c][v] = (lon, lat, 0) ------------------------------------ */
void nfo_assemble_points(const double *bounds_lon, const double *bounds_lat, long ncell, double *points)
{
    for (long k = 0; k < ncell * 4; ++k) {
        points[3 * k + 0] = bounds_lon[k];
        points[3 * k + 1] = bounds_lat[k];
        points[3 * k + 2] = 0.0;
    }
}

/* ---- A3: field.py:170-181 + geo.py:14-27 ------------------------------------------------------ */
static void lonlat2xyz(const double *p, double *xyz)
{
    /* geo.py:15-21 */
    double lam = p[0] * NFO_DEG2RAD;
    double the = p[1] * NFO_DEG2RAD;
    double rho = NFO_EARTH_RADIUS_GEO * cos(the);
    xyz[0] = rho * cos(lam);
    xyz[1] = rho * sin(lam);
    xyz[2] = NFO_EARTH_RADIUS_GEO * sin(the);
}

void nfo_arc_lengths(const double *points, long ncell, double *arc)
{
    const double r2 = NFO_EARTH_RADIUS_GEO * NFO_EARTH_RADIUS_GEO;
    for (long c = 0; c < ncell; ++c) {
        double xyz[4][3];
        for (int v = 0; v < 4; ++v) lonlat2xyz(points + (c * 4 + v) * 3, xyz[v]);
        for (int i0 = 0; i0 < 4; ++i0) { /* field.py:179-181 */
            int i1 = (i0 + 1) % 4;
            /* geo.py:26 numpy.sum(xyzA*xyzB, axis=-1): ((x + y) + z) */
            double dot = (xyz[i0][0] * xyz[i1][0] + xyz[i0][1] * xyz[i1][1]) + xyz[i0][2] * xyz[i1][2];
            double angle = acos(dot / r2);
            arc[c * 4 + i0] = fabs(NFO_EARTH_RADIUS_GEO * angle); /* geo.py:27 */
        }
    }
}

/* ---- A4: field.py:145-163 ---------------------------------------------------------------------
 * f: (nz, ncell) x-fastest, float64 (is_f32 = 0) or float32 (is_f32 = 1).  Missing -> 0 (:157): xarray
 * decodes values equal to _FillValue into NaN and fillna(0.0) zeroes them; fill = NaN means "no
 * _FillValue attribute".  Vertical integral (:161) as a sequential fma chain over z (numpy hands the
 * contraction to BLAS, whose summation order is unspecified; see tests for the tolerance). */
void nfo_vertical_integral(const void *f, int is_f32, long nz, long ncell, const double *thickness,
                           double fill, double *out)
{
    const double *fd = (const double *)f;
    const float *ff = (const float *)f;
    const float fillf = (float)fill;
    /* columns are independent: with -fopenmp (and OMP_NUM_THREADS > 1) they are split over the host cores; every
     * column still runs the same sequential fma chain over z, so the result does not depend on the thread count */
#pragma omp parallel for schedule(static)
    for (long c = 0; c < ncell; ++c) out[c] = 0.0;
    for (long z = 0; z < nz; ++z) {
        double th = thickness[z];
#pragma omp parallel for schedule(static)
        for (long c = 0; c < ncell; ++c) {
            double x;
            if (is_f32) {
                float xf = ff[z * ncell + c];
                x = (xf != xf || xf == fillf) ? 0.0 : (double)xf;
            } else {
                x = fd[z * ncell + c];
                x = (x != x || x == fill) ? 0.0 : x;
            }
            out[c] = fma(th, x, out[c]);
        }
    }
}

/* ---- A5: field.py:183-234 ---------------------------------------------------------------------
 * iV must be zero-initialised ONCE by the caller (field.py:62) and then carried across calls: row 0's
 * south slot is never written (:219).  max_abs is the running maximum (:234). */
void nfo_edge_flux(const double *uInt, const double *vInt, const double *arc /* (ncell,4) */, long ny,
                   long nx, int sverdrup, double *iV, double *eU, double *eV, double *max_abs)
{
    long ncell = ny * nx;
    for (long c = 0; c < ncell; ++c) {
        eU[c] = +uInt[c] * arc[c * 4 + 1]; /* :195 */
        eV[c] = -vInt[c] * arc[c * 4 + 2]; /* :196 */
        iV[c * 4 + 1] = eU[c];             /* :209 */
        iV[c * 4 + 2] = eV[c];             /* :211 */
    }
    for (long j = 1; j < ny; ++j)
        for (long i = 0; i < nx; ++i) iV[(j * nx + i) * 4 + 0] = eV[(j - 1) * nx + i]; /* :219 */
    for (long j = 0; j < ny; ++j) {
        for (long i = 1; i < nx; ++i) iV[(j * nx + i) * 4 + 3] = eU[j * nx + i - 1]; /* :221 */
        iV[(j * nx) * 4 + 3] = eU[j * nx + nx - 1];                                 /* :223 */
    }
    if (sverdrup) { /* :225-228 */
        const double s = NFO_EARTH_RADIUS_SV / 1.e6;
        for (long c = 0; c < ncell; ++c) { eU[c] *= s; eV[c] *= s; }
        for (long k = 0; k < ncell * 4; ++k) iV[k] *= s;
    }
    double m = *max_abs;
    for (long c = 0; c < ncell; ++c) { /* :231-234 */
        eU[c] = fabs(eU[c]);
        eV[c] = fabs(eV[c]);
        if (eU[c] > m) m = eU[c];
        if (eV[c] > m) m = eV[c];
    }
    *max_abs = m;
}

/* ---- A6: mint.PolylineIntegral.computeWeights (field.py:45-48) --------------------------------- */
#define NFO_EPS_PAR 1.e-12     /* |sin(angle)| below which a cell edge and the target are parallel */
#define NFO_TOL_DIST_REL 1.e-12 /* on-the-edge distance tolerance, relative to max |coordinate| */
#define NFO_TOL_T 1.e-10        /* min sub-segment length / interval matching tolerance, in t */
#define NFO_NEWTON_MAX 16
#define NFO_COVER_TOL 1.e-8     /* a target segment covered more than 1 + this is counted twice somewhere: an error ... */
#define NFO_COVER_LEN_TOL 1.e-9 /* ... provided the excess, as a length, exceeds this x max(1, |coordinates|) degrees (the rounding
                                   of t on target segments of ~1e-8 degrees and shorter is not an overlap of cells) */

typedef struct {
    int seg;
    int shift;
    long cell;
    double ta, tb;
    double w[4];
    double coef;
} nfo_rec;

static double max2(double a, double b) { return a > b ? a : b; }

/* Clip the segment q + t d, t in [0,1], against the convex quad v (x0,y0,...,x3,y3). */
static int clip_cell(const double *v, double qx, double qy, double dx, double dy, double *ta, double *tb)
{
    double area2 = ((v[2] - v[0]) * (v[5] - v[1]) - (v[4] - v[0]) * (v[3] - v[1])) +
                   ((v[4] - v[0]) * (v[7] - v[1]) - (v[6] - v[0]) * (v[5] - v[1]));
    if (!(area2 != 0.0)) return 0;
    double sgn = area2 > 0.0 ? 1.0 : -1.0;
    double M = max2(max2(fabs(qx), fabs(qy)), max2(fabs(qx + dx), fabs(qy + dy)));
    for (int k = 0; k < 8; ++k) M = max2(M, fabs(v[k]));
    double told = NFO_TOL_DIST_REL * M;
    double dd = dx * dx + dy * dy;
    double t0 = 0.0, t1 = 1.0;
    for (int e = 0; e < 4; ++e) {
        int e1 = (e + 1) & 3;
        double ax = v[2 * e], ay = v[2 * e + 1];
        double gx = v[2 * e1] - ax, gy = v[2 * e1 + 1] - ay;
        double gg = gx * gx + gy * gy;
        if (gg <= told * told) continue; /* collapsed edge (pole): no constraint */
        double nx = -gy * sgn, ny = gx * sgn; /* inward normal */
        double num = nx * (qx - ax) + ny * (qy - ay);
        double den = nx * dx + ny * dy;
        if (den * den <= (NFO_EPS_PAR * NFO_EPS_PAR) * gg * dd) {
            if (num < 0.0 && num * num > told * told * gg) return 0; /* parallel and outside */
        } else {
            double t = -num / den;
            if (den > 0.0) { if (t > t0) t0 = t; }
            else           { if (t < t1) t1 = t; }
        }
    }
    if (!(t1 - t0 > NFO_TOL_T)) return 0;
    *ta = t0;
    *tb = t1;
    return 1;
}

/* Inverse of the bilinear map of quad v at point p (Newton from the cell centre). */
static int inv_bilinear(const double *v, double px, double py, double *xi0, double *xi1)
{
    double ax = v[0], ay = v[1];
    double e1x = v[2] - v[0], e1y = v[3] - v[1];
    double e3x = v[6] - v[0], e3y = v[7] - v[1];
    double hx = (v[0] - v[2]) + (v[4] - v[6]), hy = (v[1] - v[3]) + (v[5] - v[7]);
    double s = 0.5, t = 0.5;
    for (int it = 0; it < NFO_NEWTON_MAX; ++it) {
        double fx = ((ax + s * e1x) + t * e3x) + (s * t) * hx - px;
        double fy = ((ay + s * e1y) + t * e3y) + (s * t) * hy - py;
        double j00 = e1x + t * hx, j01 = e3x + s * hx;
        double j10 = e1y + t * hy, j11 = e3y + s * hy;
        double det = j00 * j11 - j01 * j10;
        if (!(det != 0.0)) break;
        double ds = (fx * j11 - fy * j01) / det;
        double dt = (fy * j00 - fx * j10) / det;
        s -= ds;
        t -= dt;
        if (fabs(ds) + fabs(dt) < 1.e-15) break;
    }
    *xi0 = s;
    *xi1 = t;
    /* converged?  residual of the map against the size of the cell */
    double fx = ((ax + s * e1x) + t * e3x) + (s * t) * hx - px;
    double fy = ((ay + s * e1y) + t * e3y) + (s * t) * hy - py;
    double size = max2(max2(fabs(e1x), fabs(e1y)), max2(fabs(e3x), fabs(e3y)));
    size = max2(size, max2(fabs(v[4] - v[0]), fabs(v[5] - v[1])));
    return (fabs(fx) + fabs(fy) <= 1.e-9 * size) && s > -1.e-6 && s < 1.0 + 1.e-6 && t > -1.e-6 && t < 1.0 + 1.e-6;
}

/* ---- cells the algorithm is not defined on ---------------------------------------------------------------
 * The clip assumes a convex quad and the weights need the inverse of the cell's bilinear map, which is not one-to-one
 * in a quad with a reflex corner or a bow-tie (the lon-lat images of the cells that touch a geographic pole on a
 * rotated grid are such quads: datagen.py:116-166 leaves the pole's longitude arbitrary).  mint's behaviour there is
 * not pinned by anything in the reference, so the restatement refuses: a target segment that overlaps such a cell over
 * a positive length is an ERROR (status 1), never a silent number; a non-convex cell the line does not touch is
 * ignored.  Status 2: the Newton iteration of the inverse map did not converge in a (convex) cell. */
static int quad_is_nonconvex(const double *v)
{
    /* a corner AT a geographic pole that is not the end of an edge lying on the pole line (|lat| = 90 along a whole edge,
     * as in the top row of an un-rotated lon-lat grid, is fine): the pole's longitude is arbitrary, so the planar quad is
     * not the image of the cell, convex or not */
    int npole = 0, first = -1;
    for (int k = 0; k < 4; ++k)
        if (fabs(v[2 * k + 1]) >= 90.0 - 1.e-9) { ++npole; if (first < 0) first = k; }
    if (npole == 1 || npole == 3) return 1;
    if (npole == 2 && !(fabs(v[2 * ((first + 1) & 3) + 1]) >= 90.0 - 1.e-9 ||
                        (first == 0 && fabs(v[2 * 3 + 1]) >= 90.0 - 1.e-9))) return 1;   /* opposite corners */
    /* a geographic pole INSIDE the cell (rotated grid whose pole is not a mesh node): the longitude winds once around the
     * globe going round the corners -- the differences, each taken the short way, add up to +-360 instead of 0 */
    double turn = 0.0;
    for (int k = 0; k < 4; ++k) {
        double d = v[2 * ((k + 1) & 3)] - v[2 * k];
        d -= 360.0 * rint(d / 360.0);
        turn += d;
    }
    if (fabs(turn) > 180.0) return 1;
    double cmin = 0.0, cmax = 0.0, scale = 0.0;
    for (int k = 0; k < 4; ++k) {
        int k1 = (k + 1) & 3, k2 = (k + 2) & 3;
        double ex = v[2 * k1] - v[2 * k], ey = v[2 * k1 + 1] - v[2 * k + 1];
        double fx = v[2 * k2] - v[2 * k1], fy = v[2 * k2 + 1] - v[2 * k1 + 1];
        double cr = ex * fy - ey * fx;
        if (cr < cmin) cmin = cr;
        if (cr > cmax) cmax = cr;
        scale = max2(scale, ex * ex + ey * ey);
    }
    return cmin < -1.e-12 * scale && cmax > 1.e-12 * scale;
}

/* Date-line unwrap (round 4).  A global file stores bounds_lon wrapped into one period (e.g. [-180,180]), so the cell that
 * straddles the cut has corners 350 degrees apart: taken as a planar quad it is a clockwise sliver across the whole domain
 * that the clip accepts and counts a second time.  With a periodic locator (periodX > 0) every corner is therefore brought
 * to within periodX/2 of corner 0 before the cell is used -- the rule the reference's own generator applies to its rotated
 * grids (datagen.py:161-166, with 270 degrees there).  Only the copy the weights / the point location work on is changed:
 * getPoints() and the arc lengths keep the file's values.  mint's behaviour on such cells is parity unpinned. */
static void unwrap_quad(double *v, double periodX)
{
    if (!(periodX > 0.0)) return;
    for (int k = 1; k < 4; ++k) {
        double n = rint((v[2 * k] - v[0]) / periodX);
        if (n != 0.0) v[2 * k] -= n * periodX;
    }
}

/* a cell with a corner that is not a finite number (NaN / infinite bounds on land-only subdomains) is no cell at all */
static int quad_is_finite(const double *v)
{
    for (int k = 0; k < 8; ++k) if (!(fabs(v[k]) <= DBL_MAX)) return 0;
    return 1;
}

static int point_in_quad_evenodd(const double *v, double px, double py)
{
    int in = 0;
    for (int k = 0; k < 4; ++k) {
        int k1 = (k + 1) & 3;
        double ax = v[2 * k], ay = v[2 * k + 1], bx = v[2 * k1], by = v[2 * k1 + 1];
        if ((ay > py) != (by > py)) {
            double xc = ax + (py - ay) * (bx - ax) / (by - ay);
            if (px < xc) in = !in;
        }
    }
    return in;
}

/* does q + t d, t in [0,1], overlap the (possibly non-convex) quad over a length of more than NFO_TOL_T in t? */
static int segment_overlaps_quad(const double *v, double qx, double qy, double dx, double dy)
{
    double ts[6];
    int n = 0;
    ts[n++] = 0.0;
    ts[n++] = 1.0;
    for (int k = 0; k < 4; ++k) {
        int k1 = (k + 1) & 3;
        double ax = v[2 * k], ay = v[2 * k + 1];
        double gx = v[2 * k1] - ax, gy = v[2 * k1 + 1] - ay;
        double den = dx * gy - dy * gx;
        if (den == 0.0) continue;
        double t = ((ax - qx) * gy - (ay - qy) * gx) / den;
        double u = ((ax - qx) * dy - (ay - qy) * dx) / den;
        if (t > 0.0 && t < 1.0 && u >= 0.0 && u <= 1.0) ts[n++] = t;
    }
    for (int i = 1; i < n; ++i)   /* insertion sort of at most 6 values */
        for (int j = i; j > 0 && ts[j] < ts[j - 1]; --j) { double x = ts[j]; ts[j] = ts[j - 1]; ts[j - 1] = x; }
    for (int i = 0; i + 1 < n; ++i) {
        if (!(ts[i + 1] - ts[i] > NFO_TOL_T)) continue;
        double tm = 0.5 * (ts[i] + ts[i + 1]);
        if (point_in_quad_evenodd(v, qx + tm * dx, qy + tm * dy)) return 1;
    }
    return 0;
}

static int rec_cmp(const void *a, const void *b)
{
    const nfo_rec *x = (const nfo_rec *)a, *y = (const nfo_rec *)b;
    if (x->seg != y->seg) return x->seg < y->seg ? -1 : 1;
    if (x->ta != y->ta) return x->ta < y->ta ? -1 : 1;
    if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
    if (x->shift != y->shift) return x->shift < y->shift ? -1 : 1;
    return 0;
}

/*
 * points: (ncell,4,3) as handed to mint.Grid.setPoints (horizgrid.py:24); xyz: (npts,3) target polyline
 * (field.py:48); periodX: buildLocator's periodX (field.py:47; 0 = not periodic); counterclock:
 * computeWeights' flag (field.py:48 passes False = all edges oriented in +xi).
 * Output (sorted by segment, ta, cell): cell_edge[k] = cell*4 + edge, weight[k], seg[k] (0-based target
 * segment).  Returns the number of entries (4 per crossed cell), or -(needed) if cap is too small.
 * status[0]: 0 = fine; 1 = a target segment overlaps a non-convex cell; 2 = the inverse bilinear map did not converge;
 * 3 = a target segment is covered more than once (overlapping cells; status[1] = -1, coverage[] is filled in);
 * status[1] = the smallest offending cell id, status[2] = its first segment.  On an error no weights are returned (0).
 */
long nfo_polyline_weights(const double *points, long ncell, const double *xyz, int npts, double periodX,
                          int counterclock, long cap, int64_t *cell_edge, double *weight, int *seg, long *status,
                          double *coverage /* npts-1 values or NULL: sum of coef*(tb-ta) per segment */,
                          int skip_unsupported /* 1: non-convex cells contribute nothing instead of being an error */)
{
    long nrec = 0, rcap = 1024;
    long err = 0, err_cell = -1, err_seg = -1;     /* kind 1: a target segment overlaps a cell the weights are not defined on */
    long err2_cell = -1, err2_seg = -1;             /* kind 2: the inverse bilinear map did not converge (reported only when no kind 1) */
    nfo_rec *recs = (nfo_rec *)malloc(rcap * sizeof(nfo_rec));
    int nshift = periodX > 0.0 ? 3 : 1;
    for (int s = 0; s + 1 < npts; ++s) {
        double p0x = xyz[3 * s], p0y = xyz[3 * s + 1];
        double dx = xyz[3 * (s + 1)] - p0x, dy = xyz[3 * (s + 1) + 1] - p0y;
        if (dx == 0.0 && dy == 0.0) continue;
        for (int k = 0; k < nshift; ++k) {
            int shift = nshift == 3 ? k - 1 : 0;
            double qx = p0x + shift * periodX, qy = p0y;
            double sxmin = qx < qx + dx ? qx : qx + dx, sxmax = qx < qx + dx ? qx + dx : qx;
            double symin = qy < qy + dy ? qy : qy + dy, symax = qy < qy + dy ? qy + dy : qy;
            for (long c = 0; c < ncell; ++c) {
                double v[8];
                double cxmin = 1e300, cxmax = -1e300, cymin = 1e300, cymax = -1e300;
                for (int i = 0; i < 4; ++i) {
                    v[2 * i] = points[(c * 4 + i) * 3];
                    v[2 * i + 1] = points[(c * 4 + i) * 3 + 1];
                }
                if (!quad_is_finite(v)) continue;
                unwrap_quad(v, periodX);
                for (int i = 0; i < 4; ++i) {
                    if (v[2 * i] < cxmin) cxmin = v[2 * i];
                    if (v[2 * i] > cxmax) cxmax = v[2 * i];
                    if (v[2 * i + 1] < cymin) cymin = v[2 * i + 1];
                    if (v[2 * i + 1] > cymax) cymax = v[2 * i + 1];
                }
                /* cheap reject (pure speed-up; one part in 1e9 slack so it never decides anything) */
                double slack = 1.e-9 * (fabs(cxmin) + fabs(cxmax) + fabs(cymin) + fabs(cymax) + 1.0);
                if (cxmin > sxmax + slack || cxmax < sxmin - slack || cymin > symax + slack || cymax < symin - slack)
                    continue;
                if (quad_is_nonconvex(v)) {
                    if (!skip_unsupported && segment_overlaps_quad(v, qx, qy, dx, dy) && (err_cell < 0 || c < err_cell)) {
                        err = 1; err_cell = c; err_seg = s;
                    }
                    continue;
                }
                double ta, tb;
                if (!clip_cell(v, qx, qy, dx, dy, &ta, &tb)) continue;
                double a0, a1, b0, b1;
                int ok = inv_bilinear(v, qx + ta * dx, qy + ta * dy, &a0, &a1);
                ok &= inv_bilinear(v, qx + tb * dx, qy + tb * dy, &b0, &b1);
                if (!ok && (err2_cell < 0 || c < err2_cell)) { err2_cell = c; err2_seg = s; }
                double d0 = b0 - a0, d1 = b1 - a1;
                double m0 = 0.5 * (a0 + b0), m1 = 0.5 * (a1 + b1);
                if (nrec == rcap) {
                    rcap *= 2;
                    recs = (nfo_rec *)realloc(recs, rcap * sizeof(nfo_rec));
                }
                nfo_rec *r = &recs[nrec++];
                r->seg = s;
                r->shift = shift;
                r->cell = c;
                r->ta = ta;
                r->tb = tb;
                r->w[0] = d0 * (1.0 - m1); /* south  0->1 */
                r->w[1] = d1 * m0;         /* east   1->2 */
                r->w[2] = d0 * m1;         /* north  3->2 */
                r->w[3] = d1 * (1.0 - m0); /* west   0->3 */
                if (counterclock) { r->w[2] = -r->w[2]; r->w[3] = -r->w[3]; } /* edges 2->3, 3->0 */
                r->coef = 1.0;
            }
        }
    }
    /* the device finds kind 1 in its counting pass and stops there; kind 2 can only show in the pass that solves for xi */
    if (!err && err2_cell >= 0) { err = 2; err_cell = err2_cell; err_seg = err2_seg; }
    if (status) { status[0] = err; status[1] = err_cell; status[2] = err_seg; }
    if (err) { free(recs); return 0; }
    qsort(recs, nrec, sizeof(nfo_rec), rec_cmp);
    /* multiplicity: a sub-segment found with the same [ta,tb] in n cells counts 1/n in each */
    for (long i = 0; i < nrec; ++i) {
        int n = 0;
        for (long j = i; j >= 0 && recs[j].seg == recs[i].seg && recs[i].ta - recs[j].ta <= NFO_TOL_T; --j)
            if (fabs(recs[j].tb - recs[i].tb) <= NFO_TOL_T) ++n;
        for (long j = i + 1; j < nrec && recs[j].seg == recs[i].seg && recs[j].ta - recs[i].ta <= NFO_TOL_T; ++j)
            if (fabs(recs[j].tb - recs[i].tb) <= NFO_TOL_T) ++n;
        recs[i].coef = 1.0 / (double)n;
    }
    {   /* coverage of every target segment; more than 1 = some stretch of it was found in two cells that are not the
         * same sub-segment (overlapping cells): that would be counted twice, so it is an error (status 3), never a number */
        double *cov = (double *)calloc(npts > 1 ? npts - 1 : 1, sizeof(double));
        for (long i = 0; i < nrec; ++i) cov[recs[i].seg] += recs[i].coef * (recs[i].tb - recs[i].ta);
        for (int q = 0; q + 1 < npts; ++q)   /* a zero-length segment has nothing to cover */
            if (xyz[3 * (q + 1)] == xyz[3 * q] && xyz[3 * (q + 1) + 1] == xyz[3 * q + 1]) cov[q] = 1.0;
        if (coverage) for (int q = 0; q + 1 < npts; ++q) coverage[q] = cov[q];
        for (int q = 0; q + 1 < npts; ++q)
            if (cov[q] > 1.0 + NFO_COVER_TOL) {
                const double ddx = xyz[3 * (q + 1)] - xyz[3 * q], ddy = xyz[3 * (q + 1) + 1] - xyz[3 * q + 1];
                double m = 1.0;
                const double cs[4] = {xyz[3 * q], xyz[3 * q + 1], xyz[3 * q] + ddx, xyz[3 * q + 1] + ddy};
                for (int k = 0; k < 4; ++k) if (fabs(cs[k]) > m) m = fabs(cs[k]);
                if (!((cov[q] - 1.0) * sqrt(ddx * ddx + ddy * ddy) > NFO_COVER_LEN_TOL * m)) continue;
                if (status) { status[0] = 3; status[1] = -1; status[2] = q; }
                free(cov); free(recs);
                return 0;
            }
        free(cov);
    }
    long need = nrec * 4;
    if (need > cap) { free(recs); return -need; }
    for (long i = 0; i < nrec; ++i)
        for (int e = 0; e < 4; ++e) {
            cell_edge[4 * i + e] = (int64_t)recs[i].cell * 4 + e;
            weight[4 * i + e] = recs[i].w[e] * recs[i].coef;
            seg[4 * i + e] = recs[i].seg;
        }
    free(recs);
    return need;
}

/* ---- A7: mint.PolylineIntegral.getIntegral(data, CELL_BY_CELL_DATA) (field.py:102) -------------
 * data: flat (ncell*4).  seg_totals (nseg, may be NULL) receives the per-target-segment sums. */
double nfo_get_integral(const double *data, long n, const int64_t *cell_edge, const double *weight,
                        const int *seg, int nseg, double *seg_totals)
{
    double total = 0.0;
    if (seg_totals) for (int s = 0; s < nseg; ++s) seg_totals[s] = 0.0;
    long k = 0;
    while (k < n) {
        int s = seg[k];
        double acc = 0.0;
        for (; k < n && seg[k] == s; ++k) acc += weight[k] * data[cell_edge[k]];
        if (seg_totals && s < nseg) seg_totals[s] = acc;
        total += acc;
    }
    return total;
}

/* One whole time step the way Field.update()+getFluxText() run it (field.py:112-120,98-103), for the
 * CPU-baseline timing of bench.py: A4 (u and v) -> A5 -> A7. */
double nfo_step(const void *u, const void *v, int is_f32, long nz, long ny, long nx, const double *thickness,
                double fill, const double *arc, int sverdrup, double *uInt, double *vInt, double *iV,
                double *eU, double *eV, double *max_abs, long nw, const int64_t *cell_edge,
                const double *weight, const int *seg)
{
    nfo_vertical_integral(u, is_f32, nz, ny * nx, thickness, fill, uInt);
    nfo_vertical_integral(v, is_f32, nz, ny * nx, thickness, fill, vInt);
    nfo_edge_flux(uInt, vInt, arc, ny, nx, sverdrup, iV, eU, eV, max_abs);
    return nfo_get_integral(iV, nw, cell_edge, weight, seg, 0, NULL);
}

/* ---- VectorInterp: mint.VectorInterp.findPoints + getFaceVectors (field.py:90-95,119-120) -----------------
 * python-mint (absent) again: "parity unpinned".  Restated from the W2 (face / Piola) interpolation the reference
 * asks for: for a target point inside cell c with bilinear parameters (xi, eta),
 *      V = [ (d3 (1-xi) + d1 xi) r_xi  -  (d0 (1-eta) + d2 eta) r_eta ] / J ,   J = r_xi x r_eta
 * with d0..d3 the cell's edge data (S,E,N,W; all oriented in +xi: counterclock = False) and r_xi, r_eta the tangent
 * vectors of the bilinear map.  With nemoflux's data (d1 = +U arc, d2 = -V arc: field.py:195-196) this is the
 * vertically integrated velocity per unit length in grid units; the only pin in the reference is README.md:36
 * ("the velocity is uniform and points down in the y direction" for psi = x).  A point is located in the cell with
 * the LOWEST id whose parameters lie in [-tol, 1+tol]^2 (tol = sqrt(tol2)), trying x, x-periodX, x+periodX;
 * points outside every cell get the zero vector and cell id -1. */
void nfo_vector_interp(const double *points, long ncell, const double *targets, long npts, double periodX, double tol2,
                       const double *data /* (ncell,4) */, double *vectors /* (npts,3) */, long *cell_ids /* npts or NULL */)
{
    const double tol = sqrt(tol2);
    const int nshift = periodX > 0.0 ? 3 : 1;
    for (long p = 0; p < npts; ++p) {
        vectors[3 * p] = vectors[3 * p + 1] = vectors[3 * p + 2] = 0.0;
        if (cell_ids) cell_ids[p] = -1;
        int found = 0;
        for (long c = 0; c < ncell && !found; ++c) {
            double v[8];
            double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = points[(c * 4 + i) * 3];
                v[2 * i + 1] = points[(c * 4 + i) * 3 + 1];
            }
            if (!quad_is_finite(v)) continue;
            unwrap_quad(v, periodX);   /* date-line cells: as in A6 */
            for (int i = 0; i < 4; ++i) {
                if (v[2 * i] < xmin) xmin = v[2 * i];
                if (v[2 * i] > xmax) xmax = v[2 * i];
                if (v[2 * i + 1] < ymin) ymin = v[2 * i + 1];
                if (v[2 * i + 1] > ymax) ymax = v[2 * i + 1];
            }
            const double slack = 1.e-6 * (fabs(xmin) + fabs(xmax) + fabs(ymin) + fabs(ymax) + 1.0);
            for (int k = 0; k < nshift && !found; ++k) {
                const double px = targets[3 * p] + (nshift == 3 ? k - 1 : 0) * periodX, py = targets[3 * p + 1];
                if (px < xmin - slack || px > xmax + slack || py < ymin - slack || py > ymax + slack) continue;
                if (quad_is_nonconvex(v)) continue;   /* no inverse map there (see A6): the point is "not found" in such a cell */
                double xi, eta;
                inv_bilinear(v, px, py, &xi, &eta);
                if (!(xi >= -tol && xi <= 1.0 + tol && eta >= -tol && eta <= 1.0 + tol)) continue;
                {   /* Newton must have converged onto the point */
                    const double mx = ((v[0] + xi * (v[2] - v[0])) + eta * (v[6] - v[0])) + (xi * eta) * ((v[0] - v[2]) + (v[4] - v[6])) - px;
                    const double my = ((v[1] + xi * (v[3] - v[1])) + eta * (v[7] - v[1])) + (xi * eta) * ((v[1] - v[3]) + (v[5] - v[7])) - py;
                    if (!(fabs(mx) + fabs(my) <= 1.e-9 * ((xmax - xmin) + (ymax - ymin)))) continue;
                }
                const double rxx = (1.0 - eta) * (v[2] - v[0]) + eta * (v[4] - v[6]);
                const double rxy = (1.0 - eta) * (v[3] - v[1]) + eta * (v[5] - v[7]);
                const double rex = (1.0 - xi) * (v[6] - v[0]) + xi * (v[4] - v[2]);
                const double rey = (1.0 - xi) * (v[7] - v[1]) + xi * (v[5] - v[3]);
                const double jac = rxx * rey - rxy * rex;
                const double *d = data + 4 * c;
                const double fx = d[3] * (1.0 - xi) + d[1] * xi;    /* flux in +xi  (west .. east) */
                const double fe = d[0] * (1.0 - eta) + d[2] * eta;  /* stream-function difference along xi (south .. north) */
                vectors[3 * p] = (fx * rxx - fe * rex) / jac;
                vectors[3 * p + 1] = (fx * rxy - fe * rey) / jac;
                if (cell_ids) cell_ids[p] = c;
                found = 1;
            }
        }
    }
}
