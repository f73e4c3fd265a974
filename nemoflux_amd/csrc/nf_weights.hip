// nf_weights.hip -- K2: target-line / cell-edge intersection weights for a BATCH of polylines on gfx950.
//
// Replaces  mint.PolylineIntegral.buildLocator + computeWeights as nemoflux drives them
//           (nemoflux/field.py:45-48: numCellsPerBucket=128, periodX=360., enableFolding=False,
//            counterclock=False) -- python-mint >= 1.24.4, third-party, not vendored (README.md:12).
//
// Algorithm: in the planar (lon,lat) plane, every target segment q + t d, t in [0,1]
// (tried at x-shifts -periodX, 0, +periodX) is clipped against every convex quad cell (Cyrus-Beck); the
// two ends of the sub-segment [ta,tb] are mapped to the cell's bilinear parameters xi (Newton), giving the
// four edge weights
//      w_S = dxi0 (1 - xim1)   w_E = dxi1 xim0   w_N = dxi0 xim1   w_W = dxi1 (1 - xim0)
// (all edges oriented in +xi: counterclock = False), times 1/n when the same [ta,tb] is found in n cells
// (a sub-segment running along a shared edge is counted once).
//
// Mapping to the hardware (round 5; until round 4 one wavefront owned 64 cells and walked ALL segment images, two lanes of 64
// busy in the clip -- 16 / 95 / 723 ms for 65 / 512 / 4096 transects on the ORCA12-like grid, now 2.9 / 12 / 68 ms with the
// same bits: profiles/r05_weights_scaling.txt).  Everything is one lane per unit of work, compacted with ballot / popcount
// and a scan between a count pass and a fill pass, so every list has a fixed order and the result is bitwise reproducible:
//   1. locator (buildLocator): bounding boxes of the cells and of groups of 16, 256, 4096 ... of them, 16 bytes each, from
//      one pass over the corner table: 4 x 4 blocks of cells (then of blocks) when the cells are known to be rows of nx -- a
//      Field's grid --, 16 consecutive cells for a flat mint.Grid;
//   2. walk: (group, segment image) pairs from the root down, one lane per (pair, child): boxes apart? box corners on one
//      side of the target line? -- no divisions; the last level leaves (cell, image) candidates;
//   3. clip: one lane per candidate (Cyrus-Beck against the cell's four edges); hits become records (key, cell, image, ta,
//      tb); a candidate cell the weights are not defined on is refused here;
//   4. records are stably radix-sorted (rocPRIM) by the 64-bit key (global segment id, ta quantised to 2^-40), runs of equal
//      keys are put into the engine's historical (64-cell tile, image, cell) order, the multiplicity is resolved per record
//      against its neighbours in that order (same segment, |dta|,|dtb| <= 1e-10), both ends of the piece are mapped into the
//      cell (Newton) and each record is written out as (cell, 4 edge weights, segment) -- already in the order K3's wavefront
//      segmented reduction wants.
// numCellsPerBucket (field.py:47: 128) has no counterpart to tune: the groups are 16-fold, the leaves single cells.
// No atomics except the error word.
#include <cstring>  // rocprim's texture iterator needs host memset declared first
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <mutex>
#include <vector>

#include "nf_common.h"
#include "nf_locator.h"

namespace nf {

// tolerances: same values as the oracle (oracle/nf_oracle.c) -- they are part of the algorithm's definition
constexpr double kEpsPar = 1.e-12;
constexpr double kTolDistRel = 1.e-12;
constexpr double kTolT = 1.e-10;
constexpr int kNewtonMax = 16;

__device__ inline double dmax2(double a, double b) { return a > b ? a : b; }

__device__ inline bool clip_cell(const double *v, double qx, double qy, double dx, double dy, double &ta,
                                 double &tb)
{
    double area2 = ((v[2] - v[0]) * (v[5] - v[1]) - (v[4] - v[0]) * (v[3] - v[1])) +
                   ((v[4] - v[0]) * (v[7] - v[1]) - (v[6] - v[0]) * (v[5] - v[1]));
    if (!(area2 != 0.0)) return false;
    const double sgn = area2 > 0.0 ? 1.0 : -1.0;
    double M = dmax2(dmax2(fabs(qx), fabs(qy)), dmax2(fabs(qx + dx), fabs(qy + dy)));
#pragma unroll
    for (int k = 0; k < 8; ++k) M = dmax2(M, fabs(v[k]));
    const double told = kTolDistRel * M;
    const double dd = dx * dx + dy * dy;
    double t0 = 0.0, t1 = 1.0;
    bool outside = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int e1 = (e + 1) & 3;
        const double ax = v[2 * e], ay = v[2 * e + 1];
        const double gx = v[2 * e1] - ax, gy = v[2 * e1 + 1] - ay;
        const double gg = gx * gx + gy * gy;
        if (gg <= told * told) continue;  // collapsed edge (pole): no constraint
        const double nx = -gy * sgn, ny = gx * sgn;  // inward normal
        const double num = nx * (qx - ax) + ny * (qy - ay);
        const double den = nx * dx + ny * dy;
        if (den * den <= (kEpsPar * kEpsPar) * gg * dd) {
            if (num < 0.0 && num * num > told * told * gg) outside = true;  // parallel and outside
        } else {
            const double t = -num / den;
            if (den > 0.0) {
                if (t > t0) t0 = t;
            } else {
                if (t < t1) t1 = t;
            }
        }
    }
    if (outside) return false;
    if (!(t1 - t0 > kTolT)) return false;
    ta = t0;
    tb = t1;
    return true;
}

// true when the Newton iteration converged to a point of the cell that maps onto p
__device__ inline bool inv_bilinear(const double *v, double px, double py, double &xi0, double &xi1)
{
    const double ax = v[0], ay = v[1];
    const double e1x = v[2] - v[0], e1y = v[3] - v[1];
    const double e3x = v[6] - v[0], e3y = v[7] - v[1];
    const double hx = (v[0] - v[2]) + (v[4] - v[6]), hy = (v[1] - v[3]) + (v[5] - v[7]);
    double s = 0.5, t = 0.5;
    for (int it = 0; it < kNewtonMax; ++it) {
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
    const double fx = ((ax + s * e1x) + t * e3x) + (s * t) * hx - px;
    const double fy = ((ay + s * e1y) + t * e3y) + (s * t) * hy - py;
    double size = dmax2(dmax2(fabs(e1x), fabs(e1y)), dmax2(fabs(e3x), fabs(e3y)));
    size = dmax2(size, dmax2(fabs(v[4] - v[0]), fabs(v[5] - v[1])));
    return (fabs(fx) + fabs(fy) <= 1.e-9 * size) && s > -1.e-6 && s < 1.0 + 1.e-6 && t > -1.e-6 && t < 1.0 + 1.e-6;
}

// ---- cells the algorithm is not defined on ----------------------------------------------------------------------
// The clip assumes a convex quad and the weights need the inverse of the cell's bilinear map, which is not one-to-one in a
// quad with a reflex corner or a bow-tie (the lon-lat images of the cells that touch a geographic pole on a rotated grid
// are such quads: datagen.py:116-166 leaves the pole's longitude arbitrary).  mint's behaviour there is pinned by nothing
// in the reference, so the engine refuses: a target segment that overlaps such a cell over a positive length makes
// computeWeights fail with NF_ERR_ARG (never a silent number); a non-convex cell the line does not touch is ignored.
__device__ inline bool point_in_quad_evenodd(const double *v, double px, double py)
{
    bool in = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int k1 = (k + 1) & 3;
        const double ax = v[2 * k], ay = v[2 * k + 1], bx = v[2 * k1], by = v[2 * k1 + 1];
        if ((ay > py) != (by > py)) {
            const double xc = ax + (py - ay) * (bx - ax) / (by - ay);
            if (px < xc) in = !in;
        }
    }
    return in;
}

// does q + t d, t in [0,1], overlap the (possibly non-convex) quad over more than kTolT in t?
__device__ inline bool segment_overlaps_quad(const double *v, double qx, double qy, double dx, double dy)
{
    double ts[6];
    int n = 0;
    ts[n++] = 0.0;
    ts[n++] = 1.0;
    for (int k = 0; k < 4; ++k) {
        const int k1 = (k + 1) & 3;
        const double ax = v[2 * k], ay = v[2 * k + 1];
        const double gx = v[2 * k1] - ax, gy = v[2 * k1 + 1] - ay;
        const double den = dx * gy - dy * gx;
        if (den == 0.0) continue;
        const double t = ((ax - qx) * gy - (ay - qy) * gx) / den;
        const double u = ((ax - qx) * dy - (ay - qy) * dx) / den;
        if (t > 0.0 && t < 1.0 && u >= 0.0 && u <= 1.0) ts[n++] = t;
    }
    for (int i = 1; i < n; ++i)
        for (int j = i; j > 0 && ts[j] < ts[j - 1]; --j) {
            const double x = ts[j];
            ts[j] = ts[j - 1];
            ts[j - 1] = x;
        }
    for (int i = 0; i + 1 < n; ++i) {
        if (!(ts[i + 1] - ts[i] > kTolT)) continue;
        const double tm = 0.5 * (ts[i] + ts[i + 1]);
        if (point_in_quad_evenodd(v, qx + tm * dx, qy + tm * dy)) return true;
    }
    return false;
}

// error word of a weight build: the smallest offending (cell, kind, segment), ~0 = none
__device__ inline void flag_cell(unsigned long long *err, long cell, int kind, int seg)
{
    atomicMin(err, ((unsigned long long)cell << 32) | ((unsigned long long)kind << 24) | (unsigned)(seg & 0xffffff));
}

struct Records {  // SoA, device
    unsigned long long *key;  // (segment << 40) | floor(ta * 2^40)
    int *cell;
    unsigned char *kshift;    // which periodic image of the target segment (0 .. nshift-1) found the cell
    double *ta, *tb;          // the piece of the target segment inside the cell
};
constexpr int kTaBits = 40;
constexpr unsigned long long kTaOne = 1ull << kTaBits;
constexpr unsigned long long kTaWindow = 112;  // > kTolT * 2^40 + 1


struct SegImage {   // one periodic image of a target segment: q + t d, t in [0, 1]
    double qx, qy, dx, dy;
    int s, k;
};
// NSHIFT: 1 or 3 at compile time (the division by 3 becomes a multiplication), 0 = whatever nshift says
template <int NSHIFT = 0>
__device__ inline SegImage load_image(const double *__restrict__ segs, int img, int nshift, double periodX)
{
    SegImage g;
    if (NSHIFT) nshift = NSHIFT;
    g.s = img / nshift;
    g.k = img - g.s * nshift;
    g.dx = segs[4 * g.s + 2];
    g.dy = segs[4 * g.s + 3];
    g.qx = segs[4 * g.s] + (nshift == 3 ? g.k - 1 : 0) * periodX;
    g.qy = segs[4 * g.s + 1];
    return g;
}

// Can the segment touch anything inside the (slack-expanded) box with corners (x0,y0) .. (x1,y1) given as 4 points p?  No, when
// the bounding boxes are apart, or when the four points lie on one side of the target LINE by more than 1e-9 x the size of
// the coordinates -- a thousand times the clip's own distance tolerance: whatever lies in the hull of the points is then
// missed by the clip too.  Conservative, no divisions.
__device__ inline bool line_may_touch(const SegImage &g, const double *p /* 4 (x,y) pairs */, double xmin, double xmax,
                                      double ymin, double ymax)
{
    const double sxmin = g.dx < 0.0 ? g.qx + g.dx : g.qx, sxmax = g.dx < 0.0 ? g.qx : g.qx + g.dx;
    const double symin = g.dy < 0.0 ? g.qy + g.dy : g.qy, symax = g.dy < 0.0 ? g.qy : g.qy + g.dy;
    if (xmin > sxmax || xmax < sxmin || ymin > symax || ymax < symin) return false;
    double M = dmax2(dmax2(fabs(g.qx), fabs(g.qy)), dmax2(fabs(g.qx + g.dx), fabs(g.qy + g.dy)));
#pragma unroll
    for (int i = 0; i < 8; ++i) M = dmax2(M, fabs(p[i]));
    const double m2 = (1.e-9 * M) * (1.e-9 * M) * (g.dx * g.dx + g.dy * g.dy);
    bool all_pos = true, all_neg = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double sd = g.dx * (p[2 * i + 1] - g.qy) - g.dy * (p[2 * i] - g.qx);
        const bool far = sd * sd > m2;
        all_pos = all_pos && far && sd > 0.0;
        all_neg = all_neg && far && sd < 0.0;
    }
    return !(all_pos || all_neg);
}

// One level of the walk, count pass: lane t tests child (t % 16) of pair (t / 16) -- the box of a group of the level below or,
// at the last level, of a cell.  pnode == nullptr: the pairs are (root, image p) for every image.  Writes the ballot mask per
// wavefront and the number of set bits per workgroup.
template <int NSHIFT>
__global__ __launch_bounds__(kBlock) void k_walk_count(const int *__restrict__ pnode, const int *__restrict__ pimg, long np,
                                                       const Box4 *__restrict__ box, Layout lay,
                                                       const double *__restrict__ segs, int nshift, double periodX,
                                                       unsigned long long *__restrict__ wmask, int *__restrict__ bcnt)
{
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    const long p = t / kFan;
    bool pass = false;
    if (p < np) {
        const long child = child_of(lay, pnode ? pnode[p] : 0, (int)(t & (kFan - 1)));
        if (child >= 0) {
            const SegImage g = load_image<NSHIFT>(segs, pimg ? pimg[p] : (int)p, nshift, periodX);
            if (!(g.dx == 0.0 && g.dy == 0.0)) {
                const Box4 b = box[child];
                const double c4[8] = {b.xmin, b.ymin, b.xmax, b.ymin, b.xmax, b.ymax, b.xmin, b.ymax};
                pass = line_may_touch(g, c4, b.xmin, b.xmax, b.ymin, b.ymax);
            }
        }
    }
    block_count(__ballot(pass), wmask, bcnt);
}


// The clip stage: one lane per (cell, image) candidate, in (image, cell) order.  Count pass (FILL = false): clip, ballot mask
// and count per wavefront; a candidate cell the weights are not defined on is refused here (flag_cell) or dropped.  Fill pass:
// the lanes of the mask clip again (same inputs, same bits) and write the record (key, cell, image, ta, tb); the weights
// themselves are formed after the sort (k_expand), straight into their final place.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_clip_pairs(const double *__restrict__ xy, const int *__restrict__ ccell,
                                                       const int *__restrict__ cimg, long nc, double period,
                                                       const double *__restrict__ segs, const int *__restrict__ seg_cc,
                                                       int nshift, double periodX, unsigned long long *__restrict__ wmask,
                                                       int *__restrict__ bcnt, const long *__restrict__ boff, Records rec,
                                                       unsigned long long *__restrict__ err, int skip_unsupported)
{
    const long t = (long)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    bool hit = false;
    if (FILL) {
        if (t < nc) hit = (wmask[t / kWave] >> lane) & 1ull;
        if (!hit) return;
    }
    double v[8], ta = 0.0, tb = 0.0;
    SegImage g{};
    long c = 0;
    if (t < nc) {
        c = ccell[t];
        g = load_image(segs, cimg[t], nshift, periodX);
        double cxmin, cxmax, cymin, cymax, slack;
        cell_geometry(xy, c, period, v, cxmin, cxmax, cymin, cymax, slack);   // a candidate is a finite cell
        if (FILL) {
            clip_cell(v, g.qx, g.qy, g.dx, g.dy, ta, tb);
        } else if (g.dx == 0.0 && g.dy == 0.0) {
            // a zero-length piece (a repeated vertex) crosses nothing; the walk never lets one through, a one-cell grid has no walk
        } else if (quad_is_nonconvex(v)) {   // not a cell the weights are defined on: refuse if the line really crosses it
            // (skip policy: the cell contributes nothing and the segment's coverage says so)
            if (segment_overlaps_quad(v, g.qx, g.qy, g.dx, g.dy)) {
                if (skip_unsupported) atomicAdd(err + 1, 1ull);     // counted: the caller's warning says how many were dropped
                else flag_cell(err, c, 1, g.s);
            }
        } else {
            hit = clip_cell(v, g.qx, g.qy, g.dx, g.dy, ta, tb);
        }
    }
    if (!FILL) {
        block_count(__ballot(hit), wmask, bcnt);
        return;
    }
    const unsigned long long mask = wmask[t / kWave];
    const long pos = wave_offset(wmask, boff, blockIdx.x, threadIdx.x / kWave) + __popcll(mask & ((1ull << lane) - 1ull));
    unsigned long long q = (unsigned long long)(ta * (double)kTaOne);
    if (q >= kTaOne) q = kTaOne - 1;
    rec.key[pos] = ((unsigned long long)g.s << kTaBits) | q;
    rec.cell[pos] = (int)c;
    rec.kshift[pos] = (unsigned char)g.k;
    rec.ta[pos] = ta;
    rec.tb[pos] = tb;
}

// Records with the same sort key (same target segment, same quantised ta: a sub-segment along an edge that two cells share, or
// found through two periodic images) keep, through the stable sort, the order the clip stage wrote them in: (image, cell).
// The engine has always listed them in (64-cell tile, image, cell) order -- the order its first weight kernel worked in --
// and the order of the records is the order K3 adds them up in: this pass puts every run of equal keys into that order, so
// that weights, rows and their checksums stay bit for bit what they were.  Runs are two to four records long.
__global__ __launch_bounds__(kBlock) void k_tie_order(const unsigned long long *__restrict__ keys,
                                                      const unsigned *__restrict__ perm_in, long nrec, Records rec,
                                                      unsigned *__restrict__ perm_out)
{
    const long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nrec) return;
    const unsigned long long key = keys[i];
    long a = i, b = i + 1;
    while (a > 0 && keys[a - 1] == key) --a;
    while (b < nrec && keys[b] == key) ++b;
    const unsigned r = perm_in[i];
    if (b - a == 1) {
        perm_out[i] = r;
        return;
    }
    auto order = [&](unsigned q) {
        const unsigned long long cell = (unsigned)rec.cell[q];
        return ((cell / kWave) << 8) | ((unsigned long long)rec.kshift[q] << 6) | (cell & (kWave - 1));
    };
    const unsigned long long mine = order(r);
    long rank = 0;
    for (long j = a; j < b; ++j) rank += order(perm_in[j]) < mine;
    perm_out[a + rank] = r;
}

__global__ __launch_bounds__(kBlock) void k_iota(unsigned *p, long n)
{
    long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k < n) p[k] = (unsigned)k;
}

// lower bound of every segment id in the sorted key list -> CSR over records
__global__ __launch_bounds__(kBlock) void k_seg_bounds(const unsigned long long *__restrict__ keys, long nrec,
                                                       int nseg, int *__restrict__ rec_start)
{
    int s = blockIdx.x * kBlock + threadIdx.x;
    if (s > nseg) return;
    long lo = 0, hi = nrec;
    while (lo < hi) {
        long mid = (lo + hi) >> 1;
        if ((keys[mid] >> kTaBits) < (unsigned long long)s) lo = mid + 1;
        else hi = mid;
    }
    rec_start[s] = (int)lo;
}

// multiplicity, the four edge weights of the record (both ends of the piece mapped into the cell: Newton) and the output
// arrays, one lane per record in sorted order
__global__ __launch_bounds__(kBlock) void k_expand(const unsigned long long *__restrict__ keys,
                                                   const unsigned *__restrict__ perm, long nrec, Records rec,
                                                   const double *__restrict__ xy, double period,
                                                   const double *__restrict__ segs, const int *__restrict__ seg_cc, int nshift,
                                                   double periodX, unsigned long long *__restrict__ err,
                                                   int *__restrict__ cell_out, double *__restrict__ w4_out,
                                                   int *__restrict__ seg_out, double *__restrict__ len_out)
{
    long i = (long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nrec) return;
    const unsigned long long key = keys[i];
    const unsigned long long s = key >> kTaBits;
    const unsigned r = perm[i];
    const double ta = rec.ta[r], tb = rec.tb[r];
    int n = 1;
    // neighbours in (segment, ta) order: only records within the ta window can match
    for (long j = i - 1; j >= 0 && key - keys[j] <= kTaWindow; --j) {
        const unsigned rj = perm[j];
        if (fabs(rec.ta[rj] - ta) <= kTolT && fabs(rec.tb[rj] - tb) <= kTolT) ++n;
    }
    for (long j = i + 1; j < nrec && keys[j] - key <= kTaWindow; ++j) {
        const unsigned rj = perm[j];
        if (fabs(rec.ta[rj] - ta) <= kTolT && fabs(rec.tb[rj] - tb) <= kTolT) ++n;
    }
    const double coef = 1.0 / (double)n;
    const long c = rec.cell[r];
    const SegImage g = load_image(segs, (int)s * nshift + rec.kshift[r], nshift, periodX);
    double v[8], cxmin, cxmax, cymin, cymax, slack;
    cell_geometry(xy, c, period, v, cxmin, cxmax, cymin, cymax, slack);
    double a0, a1, b0, b1;
    bool ok = inv_bilinear(v, g.qx + ta * g.dx, g.qy + ta * g.dy, a0, a1);
    ok = inv_bilinear(v, g.qx + tb * g.dx, g.qy + tb * g.dy, b0, b1) && ok;
    if (!ok) flag_cell(err, c, 2, g.s);
    const double d0 = b0 - a0, d1 = b1 - a1;
    const double m0 = 0.5 * (a0 + b0), m1 = 0.5 * (a1 + b1);
    double w0 = d0 * (1.0 - m1), w1 = d1 * m0, w2 = d0 * m1, w3 = d1 * (1.0 - m0);
    if (seg_cc[g.s]) {
        w2 = -w2;
        w3 = -w3;
    }
    double2 *po = reinterpret_cast<double2 *>(w4_out + 4 * i);
    po[0] = make_double2(w0 * coef, w1 * coef);
    po[1] = make_double2(w2 * coef, w3 * coef);
    cell_out[i] = (int)c;
    seg_out[i] = (int)s;
    len_out[i] = coef * (tb - ta);   // the piece of the target segment this record accounts for
}

// coverage of every target segment: sum of coef * (tb - ta) over its records = the fraction of the segment that lies in
// cells of the grid (1 when it is inside, counted once).  One wavefront per segment, fixed summation order.
__global__ __launch_bounds__(kBlock) void k_seg_coverage(const double *__restrict__ len, const int *__restrict__ rec_start,
                                                         int nseg, double *__restrict__ cov)
{
    const int s = (blockIdx.x * kBlock + threadIdx.x) / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (s >= nseg) return;
    double acc = 0.0;
    for (long k = rec_start[s] + lane; k < rec_start[s + 1]; k += kWave) acc += len[k];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
    if (lane == 0) cov[s] = acc;
}


// ---- unique-edge folding (see WeightSet::EdgeEntry) ---------------------------------------------------------------
constexpr unsigned kNoElem = 0xffffffffu;   // row-0 south slot: never written, carries no flux (field.py:219)

// 4 (key, weight) pairs per record: key = (segment << 32) | element of [eU | eV] that carries the slot
__global__ __launch_bounds__(kBlock) void k_fold_keys(const int *__restrict__ cell, const double *__restrict__ w4,
                                                      const int *__restrict__ seg, long nrec, long ncell, unsigned nx,
                                                      unsigned long long *__restrict__ key, double *__restrict__ val)
{
    const long idx = (long)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= 4 * nrec) return;
    const long i = idx >> 2;
    const int e = (int)(idx & 3);
    const long c = cell[i];
    const unsigned j = (unsigned)(c / nx), col = (unsigned)(c - (long)j * nx);
    unsigned elem;
    if (e == 0) elem = j > 0 ? (unsigned)(ncell + c - nx) : kNoElem;        // south = eV of the row below
    else if (e == 1) elem = (unsigned)c;                                     // east  = eU
    else if (e == 2) elem = (unsigned)(ncell + c);                           // north = eV
    else elem = (unsigned)(col > 0 ? c - 1 : c - 1 + nx);                    // west  = eU of the left neighbour (periodic)
    key[idx] = ((unsigned long long)(unsigned)seg[i] << 32) | elem;
    val[idx] = w4[idx];
}

__global__ __launch_bounds__(kBlock) void k_fold_heads(const unsigned long long *__restrict__ key, long n,
                                                       int *__restrict__ head)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const unsigned long long me = key[k];
    head[k] = ((unsigned)me != kNoElem) && (k == 0 || key[k - 1] != me);
}

// the first pair of every run of equal keys adds up its run (in sorted order: a fixed summation order) and writes the entry
__global__ __launch_bounds__(kBlock) void k_fold_merge(const unsigned long long *__restrict__ key,
                                                       const double *__restrict__ val, const int *__restrict__ head,
                                                       const int *__restrict__ pos, long n,
                                                       WeightSet::EdgeEntry *__restrict__ ent)
{
    const long k = (long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n || !head[k]) return;
    const unsigned long long me = key[k];
    double acc = val[k];
    for (long m = k + 1; m < n && key[m] == me; ++m) acc += val[m];
    WeightSet::EdgeEntry o;
    o.elem = (int)(unsigned)me;
    o.seg = (int)(me >> 32);
    o.w = acc;
    ent[pos[k]] = o;
}

__global__ __launch_bounds__(kBlock) void k_ent_bounds(const WeightSet::EdgeEntry *__restrict__ ent, long nent, int nseg,
                                                       int *__restrict__ ent_start)
{
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (s > nseg) return;
    long lo = 0, hi = nent;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (ent[mid].seg < s) lo = mid + 1;
        else hi = mid;
    }
    ent_start[s] = (int)lo;
}

void WeightSet::release()
{
    if (ent) (void)hipFree(ent);
    if (ent_start) (void)hipFree(ent_start);
    ent = nullptr;
    ent_start = nullptr;
    nent = 0;
    if (cell) (void)hipFree(cell);
    if (w4) (void)hipFree(w4);
    if (seg) (void)hipFree(seg);
    if (seg_start) (void)hipFree(seg_start);
    coverage.clear();
    over_seg = -1;
    dropped = 0;
    cell = nullptr;
    w4 = nullptr;
    seg = nullptr;
    seg_start = nullptr;
    nrec = 0;
    nseg = 0;
}

int weights_to_host(const WeightSet &ws, int64_t *cell_edge, double *weight, int *seg)
{
    if (ws.nrec == 0) return NF_OK;
    std::vector<int> c((size_t)ws.nrec), sg((size_t)ws.nrec);
    NF_HIP(hipMemcpy(c.data(), ws.cell, sizeof(int) * ws.nrec, hipMemcpyDeviceToHost));
    NF_HIP(hipMemcpy(sg.data(), ws.seg, sizeof(int) * ws.nrec, hipMemcpyDeviceToHost));
    if (weight) NF_HIP(hipMemcpy(weight, ws.w4, sizeof(double) * 4 * ws.nrec, hipMemcpyDeviceToHost));
    for (long i = 0; i < ws.nrec; ++i)  // pure re-indexing of the device result for the caller
        for (int e = 0; e < 4; ++e) {
            if (cell_edge) cell_edge[4 * i + e] = (int64_t)c[i] * 4 + e;
            if (seg) seg[4 * i + e] = sg[i];
        }
    return NF_OK;
}

namespace {
std::mutex g_pool_mtx;
std::vector<BuildScratch *> g_pool;      // idle scratches; never freed at process exit: the HIP runtime may be gone by then
}  // namespace

BuildScratch *scratch_checkout(int device)
{
    std::lock_guard<std::mutex> lock(g_pool_mtx);
    for (size_t k = g_pool.size(); k-- > 0;)
        if (g_pool[k]->device == device) {
            BuildScratch *sc = g_pool[k];
            g_pool.erase(g_pool.begin() + (long)k);
            return sc;
        }
    return nullptr;
}

bool scratch_checkin(BuildScratch *sc)
{
    std::lock_guard<std::mutex> lock(g_pool_mtx);
    if (g_pool.size() >= kScratchPool) return false;
    g_pool.push_back(sc);
    return true;
}

void weights_trim_scratch()
{
    std::vector<BuildScratch *> idle;
    {
        std::lock_guard<std::mutex> lock(g_pool_mtx);
        idle.swap(g_pool);
    }
    for (BuildScratch *sc : idle) delete sc;     // scratches checked out by a running build are not in the pool: untouched
}

void LocatorBoxes::release()
{
    for (void *p : level)
        if (p) (void)hipFree(p);
    level.clear();
    count.clear();
    xy = nullptr;
    ncell = 0;
    period = -1.0;
}

namespace {
struct DevBuf {  // frees on scope exit
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};
}  // namespace

int fold_weights(WeightSet *ws, long ncell, long nx, hipStream_t s)
{
    if (ws->ent) (void)hipFree(ws->ent);
    if (ws->ent_start) (void)hipFree(ws->ent_start);
    ws->ent = nullptr;
    ws->ent_start = nullptr;
    ws->nent = 0;
    NF_REQUIRE(nx > 0 && ncell > 0 && ncell % nx == 0 && 2 * ncell < (long)kNoElem, NF_ERR_ARG, "fold_weights: bad grid sizes");
    NF_REQUIRE(ws->nrec < (1l << 29), NF_ERR_ARG, "fold_weights: too many records");
    NF_HIP(hipMalloc((void **)&ws->ent_start, sizeof(int) * (size_t)(ws->nseg + 1)));
    if (ws->nrec == 0) {
        NF_HIP(hipMemsetAsync(ws->ent_start, 0, sizeof(int) * (size_t)(ws->nseg + 1), s));
        return NF_OK;
    }
    const long n = 4 * ws->nrec;
    const unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
    DevBuf k_in, k_out, v_in, v_out, tmp, head, pos;
    NF_HIP(k_in.alloc(sizeof(unsigned long long) * n));
    NF_HIP(k_out.alloc(sizeof(unsigned long long) * n));
    NF_HIP(v_in.alloc(sizeof(double) * n));
    NF_HIP(v_out.alloc(sizeof(double) * n));
    hipLaunchKernelGGL(k_fold_keys, dim3(nb), dim3(kBlock), 0, s, ws->cell, ws->w4, ws->seg, ws->nrec, ncell, (unsigned)nx,
                       k_in.as<unsigned long long>(), v_in.as<double>());
    int bits = 1;
    while ((1l << bits) < (long)ws->nseg + 1 && bits < 24) ++bits;
    size_t tmp_bytes = 0;
    NF_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const unsigned long long *)k_in.p, k_out.as<unsigned long long>(),
                                     (const double *)v_in.p, v_out.as<double>(), (size_t)n, 0u, (unsigned)(32 + bits), s));
    NF_HIP(tmp.alloc(tmp_bytes));
    NF_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, (const unsigned long long *)k_in.p, k_out.as<unsigned long long>(),
                                     (const double *)v_in.p, v_out.as<double>(), (size_t)n, 0u, (unsigned)(32 + bits), s));
    NF_HIP(head.alloc(sizeof(int) * n));
    NF_HIP(pos.alloc(sizeof(int) * n));
    hipLaunchKernelGGL(k_fold_heads, dim3(nb), dim3(kBlock), 0, s, k_out.as<unsigned long long>(), n, head.as<int>());
    DevBuf tmp2;
    size_t tmp2_bytes = 0;
    NF_HIP(rocprim::exclusive_scan(nullptr, tmp2_bytes, head.as<int>(), pos.as<int>(), 0, (size_t)n, rocprim::plus<int>(), s));
    NF_HIP(tmp2.alloc(tmp2_bytes));
    NF_HIP(rocprim::exclusive_scan(tmp2.p, tmp2_bytes, head.as<int>(), pos.as<int>(), 0, (size_t)n, rocprim::plus<int>(), s));
    int last_pos = 0, last_head = 0;
    NF_HIP(hipMemcpyAsync(&last_pos, pos.as<int>() + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    NF_HIP(hipMemcpyAsync(&last_head, head.as<int>() + (n - 1), sizeof(int), hipMemcpyDeviceToHost, s));
    NF_HIP(hipStreamSynchronize(s));
    ws->nent = (long)last_pos + last_head;
    NF_HIP(hipMalloc((void **)&ws->ent, sizeof(WeightSet::EdgeEntry) * (size_t)(ws->nent ? ws->nent : 1)));
    hipLaunchKernelGGL(k_fold_merge, dim3(nb), dim3(kBlock), 0, s, k_out.as<unsigned long long>(), v_out.as<double>(),
                       head.as<int>(), pos.as<int>(), n, ws->ent);
    hipLaunchKernelGGL(k_ent_bounds, dim3((unsigned)((ws->nseg + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, ws->ent,
                       ws->nent, ws->nseg, ws->ent_start);
    NF_HIP(hipGetLastError());
    NF_HIP(hipStreamSynchronize(s));
    return NF_OK;
}

int build_weights(const double *xy, long ncell, const double *segs_host, const int *seg_cc_host, int nseg,
                  double periodX, WeightSet *out, hipStream_t s, int skip_unsupported, int overlap_warn, LocatorBoxes *keep,
                  long row_length)
{
    out->release();
    out->nseg = nseg;
    out->coverage.assign((size_t)nseg, 0.0);
    NF_REQUIRE(ncell > 0 && ncell < (1l << 31), NF_ERR_ARG, "weights: ncell out of range");
    NF_REQUIRE(nseg >= 0 && nseg < (1 << 23), NF_ERR_ARG, "weights: segment count out of range");
    const int nshift = periodX > 0.0 ? 3 : 1;
    const double period = nshift == 3 ? periodX : 0.0;

    ScratchLease lease;      // the arenas: rewound and kept for the next build when this one ends drained, freed otherwise
    Arena &misc = lease.sc->misc;
    Arena(&level)[2] = lease.sc->level;
    double *d_segs = nullptr;
    int *d_cc = nullptr;
    unsigned long long *d_err = nullptr;
    NF_HIP(misc.take(&d_err, 2));      // [0]: the error word (~0 = none); [1]: crossings of unsupported cells dropped ('skip')
    NF_HIP(hipMemsetAsync(d_err, 0xff, sizeof(unsigned long long), s));
    NF_HIP(hipMemsetAsync(d_err + 1, 0, sizeof(unsigned long long), s));
    unsigned long long err_word = ~0ull, err_pair[2] = {~0ull, 0ull};
    auto refuse = [&](unsigned long long w) {
        char buf[256];
        const long cell = (long)(w >> 32);
        const int kind = (int)((w >> 24) & 0xff), seg = (int)(w & 0xffffff);
        snprintf(buf, sizeof buf,
                 kind == 1 ? "computeWeights: target segment %d crosses cell %ld, which is not convex in the (lon,lat) plane "
                             "(a reflex corner or a bow-tie, e.g. a cell touching the pole of a rotated grid, or a cell that contains the pole): the weights "
                             "are not defined there (setUnsupportedCells('skip') drops such cells instead: coverage < 1)"
                           : "computeWeights: target segment %d: the inverse bilinear map did not converge in cell %ld",
                 seg, cell);
        set_error(buf);
        out->release();
        return NF_ERR_ARG;
    };
    NF_HIP(misc.take(&d_segs, 4 * (size_t)nseg));
    NF_HIP(misc.take(&d_cc, (size_t)nseg));
    if (nseg > 0) {
        NF_HIP(hipMemcpyAsync(d_segs, segs_host, sizeof(double) * 4 * (size_t)nseg, hipMemcpyHostToDevice, s));
        NF_HIP(hipMemcpyAsync(d_cc, seg_cc_host, sizeof(int) * (size_t)nseg, hipMemcpyHostToDevice, s));
    }

    // ---- the locator (box hierarchy over the cells) and the walk: (group, image) pairs from the root down to (cell, image)
    // candidates, in (image, then walk) order -- nf_locator.h; the test of a pair is k_walk_count's
    Walker wk(*lease.sc, s);
    NF_TRY(wk.prepare(xy, ncell, period, row_length, keep));
    int *p_node = nullptr, *p_img = nullptr;
    long np = 0;
    NF_TRY(wk.walk((long)nseg * nshift,
                   [&](int, const int *pn, const int *pi, long npairs, const Box4 *child_boxes, Layout lay, unsigned nb,
                       unsigned long long *mask, int *cnt) {
                       if (nshift == 3)
                           hipLaunchKernelGGL(k_walk_count<3>, dim3(nb), dim3(kBlock), 0, s, pn, pi, npairs, child_boxes, lay,
                                              (const double *)d_segs, nshift, periodX, mask, cnt);
                       else
                           hipLaunchKernelGGL(k_walk_count<1>, dim3(nb), dim3(kBlock), 0, s, pn, pi, npairs, child_boxes, lay,
                                              (const double *)d_segs, nshift, periodX, mask, cnt);
                   },
                   &p_node, &p_img, &np));
    const long ncand = np;

    // ---- the clip stage over the candidates: count, scan, fill
    long nrec = 0;
    const long cw = Walker::lanes_to_waves(ncand);
    const unsigned cb = (unsigned)(cw / (kBlock / kWave));
    Records none{};
    if (ncand > 0) {
        NF_REQUIRE(ncand / kBlock < (1l << 31), NF_ERR_ARG, "weights: too many (cell, segment) candidates; split the transect set");
        NF_TRY(wk.reserve_waves(cw));
        hipLaunchKernelGGL(k_clip_pairs<false>, dim3(cb), dim3(kBlock), 0, s, xy, (const int *)p_node, (const int *)p_img, ncand,
                           period, (const double *)d_segs, (const int *)d_cc, nshift, periodX, wk.w_mask, wk.w_cnt,
                           (const long *)nullptr, none, d_err, skip_unsupported);
        NF_TRY(wk.scan_waves(cw, &nrec));
    }
    NF_HIP(hipMemcpyAsync(err_pair, d_err, sizeof err_pair, hipMemcpyDeviceToHost, s));
    NF_HIP(hipStreamSynchronize(s));
    err_word = err_pair[0];
    if (err_word != ~0ull) {
        lease.drained = true;      // nothing of this build is running any more: the scratch may serve the next one
        return refuse(err_word);
    }
    out->dropped = (long)err_pair[1];
    NF_REQUIRE(nrec < (1l << 31), NF_ERR_ARG, "weights: more than 2^31 (segment, cell) records; split the transect set");

    NF_HIP(hipMalloc((void **)&out->seg_start, sizeof(int) * (size_t)(nseg + 1)));
    if (nrec == 0) {
        NF_HIP(hipMemsetAsync(out->seg_start, 0, sizeof(int) * (size_t)(nseg + 1), s));
        NF_HIP(hipStreamSynchronize(s));
        lease.drained = true;
        for (int q = 0; q < nseg; ++q)
            if (segs_host[4 * q + 2] == 0.0 && segs_host[4 * q + 3] == 0.0) out->coverage[(size_t)q] = 1.0;
        return NF_OK;
    }

    Records rec{};
    NF_HIP(misc.take(&rec.key, (size_t)nrec));
    NF_HIP(misc.take(&rec.cell, (size_t)nrec));
    NF_HIP(misc.take(&rec.kshift, (size_t)nrec));
    NF_HIP(misc.take(&rec.ta, (size_t)nrec));
    NF_HIP(misc.take(&rec.tb, (size_t)nrec));
    hipLaunchKernelGGL(k_clip_pairs<true>, dim3(cb), dim3(kBlock), 0, s, xy, (const int *)p_node, (const int *)p_img, ncand, period,
                       (const double *)d_segs, (const int *)d_cc, nshift, periodX, wk.w_mask, wk.w_cnt, (const long *)wk.w_off, rec, d_err,
                       skip_unsupported);
    NF_HIP(hipGetLastError());

    // Stable sort of record indices by (target segment, ta).  (A segmented sort over the 40 bits of ta -- the records leave the
    // clip stage grouped by segment -- was measured: 3.1 ms against 2.2 ms for these seven device-wide passes at 44 M records.)
    // The walk's pairs are dead once the fill pass above has run, and everything below is ordered behind it on the same
    // stream: the sort's buffers re-use the two level arenas.
    level[0].rewind();
    level[1].rewind();
    unsigned long long *k_out = nullptr;
    unsigned *v_in = nullptr, *v_out = nullptr, *v_tie = nullptr;
    void *tmp = nullptr;
    int *rstart = nullptr;
    NF_HIP(level[0].take(&k_out, (size_t)nrec));
    NF_HIP(level[1].take(&v_in, (size_t)nrec));
    NF_HIP(level[1].take(&v_out, (size_t)nrec));
    const unsigned nb_rec = (unsigned)((nrec + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_iota, dim3(nb_rec), dim3(kBlock), 0, s, v_in, nrec);
    int bits = 1;
    while ((1l << bits) < (long)nseg + 1 && bits < 24) ++bits;
    const unsigned end_bit = (unsigned)(kTaBits + bits);
    size_t tmp_bytes = 0;
    NF_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const unsigned long long *)rec.key, k_out, (const unsigned *)v_in, v_out,
                                     (size_t)nrec, 0u, end_bit, s));
    NF_HIP(level[0].take(&tmp, tmp_bytes));
    NF_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, (const unsigned long long *)rec.key, k_out, (const unsigned *)v_in, v_out,
                                     (size_t)nrec, 0u, end_bit, s));
    // runs of equal keys into the engine's historical (tile, image, cell) order: see k_tie_order
    NF_HIP(level[1].take(&v_tie, (size_t)nrec));
    hipLaunchKernelGGL(k_tie_order, dim3(nb_rec), dim3(kBlock), 0, s, (const unsigned long long *)k_out, (const unsigned *)v_out, nrec,
                       rec, v_tie);
    NF_HIP(misc.take(&rstart, (size_t)(nseg + 1)));
    hipLaunchKernelGGL(k_seg_bounds, dim3((unsigned)((nseg + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       (const unsigned long long *)k_out, nrec, nseg, rstart);

    out->nrec = nrec;
    NF_HIP(hipMalloc((void **)&out->cell, sizeof(int) * (size_t)nrec));
    NF_HIP(hipMalloc((void **)&out->w4, sizeof(double) * 4 * (size_t)nrec));
    NF_HIP(hipMalloc((void **)&out->seg, sizeof(int) * (size_t)nrec));
    double *d_len = nullptr, *d_cov = nullptr;
    NF_HIP(level[0].take(&d_len, (size_t)nrec));
    NF_HIP(misc.take(&d_cov, (size_t)(nseg + 1)));
    hipLaunchKernelGGL(k_expand, dim3(nb_rec), dim3(kBlock), 0, s, (const unsigned long long *)k_out, (const unsigned *)v_tie, nrec,
                       rec, xy, period, (const double *)d_segs, (const int *)d_cc, nshift, periodX, d_err, out->cell, out->w4,
                       out->seg, d_len);
    if (nseg > 0) {
        hipLaunchKernelGGL(k_seg_coverage, dim3((unsigned)(((long)nseg * kWave + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                           (const double *)d_len, (const int *)rstart, nseg, d_cov);
        NF_HIP(hipMemcpyAsync(out->coverage.data(), d_cov, sizeof(double) * (size_t)nseg, hipMemcpyDeviceToHost, s));
    }
    NF_HIP(hipMemcpyAsync(out->seg_start, rstart, sizeof(int) * (size_t)(nseg + 1), hipMemcpyDeviceToDevice, s));
    NF_HIP(hipMemcpyAsync(&err_word, d_err, sizeof err_word, hipMemcpyDeviceToHost, s));
    NF_HIP(hipGetLastError());
    NF_HIP(hipStreamSynchronize(s));
    lease.drained = true;      // the last kernel of the build has finished; nothing below launches anything
    if (err_word != ~0ull) return refuse(err_word);   // Newton did not converge somewhere (fill pass)
    for (int q = 0; q < nseg; ++q)                     // a zero-length segment has nothing to cover
        if (segs_host[4 * q + 2] == 0.0 && segs_host[4 * q + 3] == 0.0) out->coverage[(size_t)q] = 1.0;
    // A stretch of a target segment found in two cells that do not hold the SAME sub-segment (overlapping cells) would be
    // counted twice: refuse, naming the segment.  The coverage stays readable (getCoverage) so the caller can see how much.
    for (int q = 0; q < nseg; ++q)
        if (over_covered(out->coverage[(size_t)q], segs_host + 4 * q)) {
            out->over_seg = q;
            if (overlap_warn) break;    // policy 'warn': the numbers stand, the coverage and over_seg tell the caller
            char buf[320];
            snprintf(buf, sizeof buf,
                     "computeWeights: target segment %d is covered %.9g times by the cells of the grid: cells overlap along it "
                     "(a cell wrapped across the date line with a non-periodic locator, or duplicated / folded cells that "
                     "are not identical), so part of the line would be counted twice", q, out->coverage[(size_t)q]);
            set_error(buf);
            std::vector<double> keep = out->coverage;
            out->release();                 // no records, no segments: nothing a later getIntegral could launch on
            out->coverage.swap(keep);       // ... but the coverage stays readable (its size says how many segments it is for)
            out->over_seg = q;
            return NF_ERR_ARG;
        }
    return NF_OK;
}

}  // namespace nf
