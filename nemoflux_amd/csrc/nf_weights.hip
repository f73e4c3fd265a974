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
// Mapping to the hardware.  One wavefront owns 64 consecutive cells.  Their corner table rows (64 x 64 B =
// 4 KiB contiguous) are read coalesced and staged in LDS, then each lane keeps its own 4 corners in
// registers for the whole kernel.  The locator has three levels: the 256 lanes of a workgroup clip 256 segment
// images at a time against the workgroup's box (exact segment-vs-box test) and compact the survivors into an
// LDS list; each wave then checks the survivors against its own 64-cell box (numCellsPerBucket -> the wave tile)
// with a uniform branch, and each lane against its cell.  Hits are compacted with ballot/popcount into a deterministic order
// (tile, segment, shift, lane): pass 1 counts per wave, a single-workgroup scan turns counts into offsets,
// pass 2 recomputes and writes records.  Records are then stably radix-sorted (rocPRIM) by the 64-bit key
// (global segment id, ta quantised to 2^-40), the multiplicity is resolved per record against its
// neighbours in that order (same segment, |dta|,|dtb| <= 1e-10), and each record is written out as
// (cell, 4 edge weights, segment) -- already in the order K3's wavefront segmented reduction wants.
// No atomics, so the result is bitwise reproducible run to run.
#include <cstring>  // rocprim's texture iterator needs host memset declared first
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <vector>

#include "nf_common.h"

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

__device__ inline double wmin(double x)
{
    for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o, kWave));
    return x;
}
__device__ inline double wmax(double x)
{
    for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, kWave));
    return x;
}

struct Records {  // SoA, device
    unsigned long long *key;  // (segment << 40) | floor(ta * 2^40)
    int *cell;
    double *ta, *tb, *w;  // w: 4 per record
};
constexpr int kTaBits = 40;
constexpr unsigned long long kTaOne = 1ull << kTaBits;
constexpr unsigned long long kTaWindow = 112;  // > kTolT * 2^40 + 1

// The clip stage of k_clip for up to 64 queued (image, cell-of-this-wave's-tile) pairs, one per lane, in queue order -- which
// is (image, lane) order, the order the records of a tile have always been written in.  tile_xy: the tile's 64 unwrapped
// corner rows in LDS; tile_c0: its first cell.  Returns the number of records of the batch (wave-uniform).
template <bool FILL>
__device__ inline int clip_pairs(volatile int *wq, int q_head, int n, const double *tile_xy, long tile_c0,
                                 const double *__restrict__ segs, const int *__restrict__ seg_cc, int nshift, double periodX,
                                 long base, Records rec, unsigned long long *__restrict__ err, int lane,
                                 unsigned long long lt_mask)
{
    bool hit = false;
    double ta = 0.0, tb = 0.0, qx = 0.0, qy = 0.0, dx = 0.0, dy = 0.0;
    double vv[8];
    int s = 0, cl = 0;
    if (lane < n) {
        const int e = wq[(q_head + lane) & (2 * kWave - 1)];
        const int im = e >> 6;
        cl = e & (kWave - 1);
        s = im / nshift;
        const int k = im - s * nshift;
#pragma unroll
        for (int i = 0; i < 8; ++i) vv[i] = tile_xy[cl * 8 + i];
        dx = segs[4 * s + 2];
        dy = segs[4 * s + 3];
        qx = segs[4 * s] + (nshift == 3 ? k - 1 : 0) * periodX;
        qy = segs[4 * s + 1];
        hit = clip_cell(vv, qx, qy, dx, dy, ta, tb);
    }
    const unsigned long long mask = __ballot(hit);
    if (FILL && hit) {
        const long pos = base + __popcll(mask & lt_mask);
        const long c = tile_c0 + cl;
        double a0, a1, b0, b1;
        bool ok = inv_bilinear(vv, qx + ta * dx, qy + ta * dy, a0, a1);
        ok = inv_bilinear(vv, qx + tb * dx, qy + tb * dy, b0, b1) && ok;
        if (!ok) flag_cell(err, c, 2, s);
        const double d0 = b0 - a0, d1 = b1 - a1;
        const double m0 = 0.5 * (a0 + b0), m1 = 0.5 * (a1 + b1);
        double w0 = d0 * (1.0 - m1), w1 = d1 * m0, w2 = d0 * m1, w3 = d1 * (1.0 - m0);
        if (seg_cc[s]) {
            w2 = -w2;
            w3 = -w3;
        }
        unsigned long long q = (unsigned long long)(ta * (double)kTaOne);
        if (q >= kTaOne) q = kTaOne - 1;
        rec.key[pos] = ((unsigned long long)s << kTaBits) | q;
        rec.cell[pos] = (int)c;
        rec.ta[pos] = ta;
        rec.tb[pos] = tb;
        double2 *pw = reinterpret_cast<double2 *>(rec.w + 4 * pos);
        pw[0] = make_double2(w0, w1);
        pw[1] = make_double2(w2, w3);
    }
    return __popcll(mask);
}

template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_clip(const double *__restrict__ xy, long ncell,
                                                 const double *__restrict__ segs,
                                                 const int *__restrict__ seg_cc, int nseg, int nshift,
                                                 double periodX, const int *__restrict__ wave_off,
                                                 int *__restrict__ wave_cnt, Records rec,
                                                 unsigned long long *__restrict__ err, int skip_unsupported)
{
    __shared__ double s_xy[kBlock * 8];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const long c0 = (long)blockIdx.x * kBlock;
    const long nval = ncell * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {  // coalesced stage of 256 cells x 4 corners x (lon,lat)
        long g = c0 * 8 + tid + r * kBlock;
        if (g < nval) s_xy[tid + r * kBlock] = xy[g];
    }
    __syncthreads();
    const long c = c0 + tid;
    bool valid = c < ncell;
    double v[8];
    double cxmin = 1e300, cxmax = -1e300, cymin = 1e300, cymax = -1e300;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = valid ? s_xy[tid * 8 + k] : 0.0;
    valid = valid && quad_is_finite(v);            // NaN / infinite corners: not a cell
    unwrap_quad(v, nshift == 3 ? periodX : 0.0);   // date-line cells (nf_common.h)
#pragma unroll
    for (int k = 0; k < 8; ++k) s_xy[tid * 8 + k] = v[k];   // the clip stage below reads ANY cell of the wave's tile from LDS
    if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cxmin = fmin(cxmin, v[2 * i]);
            cxmax = fmax(cxmax, v[2 * i]);
            cymin = fmin(cymin, v[2 * i + 1]);
            cymax = fmax(cymax, v[2 * i + 1]);
        }
    }
    const double slack = valid ? 1.e-9 * (fabs(cxmin) + fabs(cxmax) + fabs(cymin) + fabs(cymax) + 1.0) : 0.0;
    const bool nonconvex = valid && quad_is_nonconvex(v);
    // wave tile bounding box (the locator bucket)
    const double wslack = wmax(slack);
    const double wxmin = wmin(cxmin) - wslack, wxmax = wmax(cxmax) + wslack;
    const double wymin = wmin(cymin) - wslack, wymax = wmax(cymax) + wslack;
    const long wave_id = (c0 + tid) / kWave;
    int count = 0;
    const int base = (FILL && wave_id * kWave < ncell) ? wave_off[wave_id] : 0;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // Locator, level 1 (workgroup): the 256 lanes test 256 segment images at a time against the workgroup's box
    // with an exact segment-vs-box clip and compact the survivors, in image order, into an LDS list; level 2 (wave)
    // and level 3 (cell) then only look at those.  A workgroup is a 256-cell strip of one grid row, so almost every
    // image is rejected here: tens of candidates instead of nseg*nshift loop trips per wave.
    __shared__ int s_hits[kBlock];
    __shared__ int s_wcount[kBlock / kWave];
    __shared__ double s_box[4][kBlock / kWave];
    if (lane == 0) {
        s_box[0][tid / kWave] = wxmin;
        s_box[1][tid / kWave] = wxmax;
        s_box[2][tid / kWave] = wymin;
        s_box[3][tid / kWave] = wymax;
    }
    __syncthreads();
    double bxmin = s_box[0][0], bxmax = s_box[1][0], bymin = s_box[2][0], bymax = s_box[3][0];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; ++w) {
        bxmin = fmin(bxmin, s_box[0][w]);
        bxmax = fmax(bxmax, s_box[1][w]);
        bymin = fmin(bymin, s_box[2][w]);
        bymax = fmax(bymax, s_box[3][w]);
    }
    // per-wave queue of (image << 6 | lane) pairs waiting for the clip (ring of kQueue entries, wave-private: no barrier)
    constexpr int kQueue = 2 * kWave;
    __shared__ int s_queue[kBlock / kWave][kQueue];
    volatile int *wq = s_queue[tid / kWave];
    int q_head = 0, q_tail = 0;
    const int nimg = nseg * nshift;
    for (int chunk = 0; chunk < nimg; chunk += kBlock) {
        const int img = chunk + tid;
        bool cand = false;
        if (img < nimg) {
            const int s = img / nshift, k = img - s * nshift;
            const double dx = segs[4 * s + 2], dy = segs[4 * s + 3];
            if (!(dx == 0.0 && dy == 0.0)) {
                const double qx = segs[4 * s] + (nshift == 3 ? k - 1 : 0) * periodX, qy = segs[4 * s + 1];
                // Liang-Barsky clip of q + t d, t in [0,1], against the (slack-expanded) box; conservative
                double t0 = 0.0, t1 = 1.0;
                cand = true;
                if (dx == 0.0) cand = qx >= bxmin && qx <= bxmax;
                else {
                    double ta = (bxmin - qx) / dx, tb = (bxmax - qx) / dx;
                    if (ta > tb) { const double tt = ta; ta = tb; tb = tt; }
                    t0 = fmax(t0, ta);
                    t1 = fmin(t1, tb);
                }
                if (dy == 0.0) cand = cand && qy >= bymin && qy <= bymax;
                else {
                    double ta = (bymin - qy) / dy, tb = (bymax - qy) / dy;
                    if (ta > tb) { const double tt = ta; ta = tb; tb = tt; }
                    t0 = fmax(t0, ta);
                    t1 = fmin(t1, tb);
                }
                cand = cand && t0 <= t1 + 1.e-9;
            }
        }
        const unsigned long long cmask = __ballot(cand);
        if (lane == 0) s_wcount[tid / kWave] = __popcll(cmask);
        __syncthreads();
        int before = 0, nhit = 0;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) {
            if (w < tid / kWave) before += s_wcount[w];
            nhit += s_wcount[w];
        }
        if (cand) s_hits[before + __popcll(cmask & lt_mask)] = img;
        __syncthreads();
        for (int h = 0; h < nhit; ++h) {  // workgroup-uniform trip count, image order preserved
            const int im = s_hits[h];
            const int s = im / nshift, k = im - s * nshift;
            const double p0x = segs[4 * s], p0y = segs[4 * s + 1];
            const double dx = segs[4 * s + 2], dy = segs[4 * s + 3];
            const int shift = nshift == 3 ? k - 1 : 0;
            const double qx = p0x + shift * periodX, qy = p0y;
            const double sxmin = qx < qx + dx ? qx : qx + dx, sxmax = qx < qx + dx ? qx + dx : qx;
            const double symin = qy < qy + dy ? qy : qy + dy, symax = qy < qy + dy ? qy + dy : qy;
            if (wxmin > sxmax || wxmax < sxmin || wymin > symax || wymax < symin) continue;  // wave-uniform
            bool pass = valid && !(cxmin > sxmax + slack || cxmax < sxmin - slack || cymin > symax + slack ||
                                   cymax < symin - slack);
            // Level 3a, one lane per cell: is the cell's corner set entirely on one side of the target LINE, by more than
            // 1e-9 x the size of the coordinates (a thousand times the clip's own distance tolerance)?  Then the clip below
            // finds nothing either -- whatever the quad's shape: it lies in the hull of its corners -- and the cell drops
            // out here, for four cross products.  A long segment's bounding box holds whole tiles of cells it never touches.
            if (pass) {
                double M = dmax2(dmax2(fabs(qx), fabs(qy)), dmax2(fabs(qx + dx), fabs(qy + dy)));
#pragma unroll
                for (int i = 0; i < 8; ++i) M = dmax2(M, fabs(v[i]));
                const double m2 = (1.e-9 * M) * (1.e-9 * M) * (dx * dx + dy * dy);
                bool all_pos = true, all_neg = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double sd = dx * (v[2 * i + 1] - qy) - dy * (v[2 * i] - qx);
                    const bool far = sd * sd > m2;
                    all_pos = all_pos && far && sd > 0.0;
                    all_neg = all_neg && far && sd < 0.0;
                }
                pass = !(all_pos || all_neg);
            }
            if (pass && nonconvex) {   // not a cell the weights are defined on: refuse if the line really crosses it
                // (skip policy: the cell contributes nothing and the segment's coverage says so)
                if (!FILL && !skip_unsupported && segment_overlaps_quad(v, qx, qy, dx, dy)) flag_cell(err, c, 1, s);
                pass = false;
            }
            // Level 3b, one lane per (cell, image) PAIR: the cells that are left -- one to three of a tile's 64 -- are queued
            // in (image, lane) order, and the expensive part (the clip's four divisions, in the fill pass two Newton
            // solves) runs on 64 queued pairs at a time with every lane busy, instead of once per image with two lanes busy.
            const unsigned long long pmask = __ballot(pass);
            if (pmask == 0ull) continue;
            if (pass) wq[(q_tail + __popcll(pmask & lt_mask)) & (kQueue - 1)] = (im << 6) | lane;
            q_tail += __popcll(pmask);
            while (q_tail - q_head >= kWave) {
                count += clip_pairs<FILL>(wq, q_head, kWave, s_xy + (tid - lane) * 8, c0 + (tid - lane), segs, seg_cc, nshift,
                                          periodX, base + count, rec, err, lane, lt_mask);
                q_head += kWave;
            }
        }
        __syncthreads();  // s_hits is reused by the next chunk
    }
    if (q_tail > q_head)
        count += clip_pairs<FILL>(wq, q_head, q_tail - q_head, s_xy + (tid - lane) * 8, c0 + (tid - lane), segs, seg_cc, nshift,
                                  periodX, base + count, rec, err, lane, lt_mask);
    if (!FILL && lane == 0 && wave_id * kWave < ncell) wave_cnt[wave_id] = count;
}

// exclusive scan of n ints by ONE workgroup (n ~ ncell/64: 1e5 for ORCA12); total -> out[n]
__global__ __launch_bounds__(1024) void k_scan(const int *__restrict__ in, long n, int *__restrict__ out)
{
    __shared__ long s_sum[1024];
    const int tid = threadIdx.x;
    const long chunk = (n + 1023) / 1024;
    const long lo = tid * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    long acc = 0;
    for (long k = lo; k < hi; ++k) acc += in[k];
    s_sum[tid] = acc;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive
        long t = (tid >= o) ? s_sum[tid - o] : 0;
        __syncthreads();
        s_sum[tid] += t;
        __syncthreads();
    }
    long run = s_sum[tid] - acc;
    for (long k = lo; k < hi; ++k) {
        out[k] = (int)run;
        run += in[k];
    }
    if (tid == 1023) out[n] = s_sum[1023] > 0x7fffffffl ? -1 : (int)s_sum[1023];  // -1: too many records
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

// multiplicity + expansion into 4 entries per record
__global__ __launch_bounds__(kBlock) void k_expand(const unsigned long long *__restrict__ keys,
                                                   const unsigned *__restrict__ perm, long nrec, Records rec,
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
    const double2 *pw = reinterpret_cast<const double2 *>(rec.w + 4 * (long)r);
    const double2 a = pw[0], b = pw[1];
    double2 *po = reinterpret_cast<double2 *>(w4_out + 4 * i);
    po[0] = make_double2(a.x * coef, a.y * coef);
    po[1] = make_double2(b.x * coef, b.y * coef);
    cell_out[i] = rec.cell[r];
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
struct DevBuf {  // frees on scope exit
    void *p = nullptr;
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
                  double periodX, WeightSet *out, hipStream_t s, int skip_unsupported, int overlap_warn)
{
    out->release();
    out->nseg = nseg;
    out->coverage.assign((size_t)nseg, 0.0);
    NF_REQUIRE(ncell > 0 && ncell < (1l << 31), NF_ERR_ARG, "weights: ncell out of range");
    NF_REQUIRE(nseg >= 0 && nseg < (1 << 23), NF_ERR_ARG, "weights: segment count out of range");
    const int nshift = periodX > 0.0 ? 3 : 1;
    const long nwaves = (ncell + kWave - 1) / kWave;
    const unsigned nblocks = (unsigned)((ncell + kBlock - 1) / kBlock);

    DevBuf d_segs, d_cc, d_cnt, d_off, d_err;
    NF_HIP(d_err.alloc(sizeof(unsigned long long)));
    NF_HIP(hipMemsetAsync(d_err.p, 0xff, sizeof(unsigned long long), s));
    unsigned long long err_word = ~0ull;
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
    NF_HIP(d_segs.alloc(sizeof(double) * 4 * (size_t)nseg));
    NF_HIP(d_cc.alloc(sizeof(int) * (size_t)nseg));
    NF_HIP(d_cnt.alloc(sizeof(int) * (size_t)nwaves));
    NF_HIP(d_off.alloc(sizeof(int) * (size_t)(nwaves + 1)));
    if (nseg > 0) {
        NF_HIP(hipMemcpyAsync(d_segs.p, segs_host, sizeof(double) * 4 * (size_t)nseg, hipMemcpyHostToDevice, s));
        NF_HIP(hipMemcpyAsync(d_cc.p, seg_cc_host, sizeof(int) * (size_t)nseg, hipMemcpyHostToDevice, s));
    }
    NF_HIP(hipMemsetAsync(d_cnt.p, 0, sizeof(int) * (size_t)nwaves, s));

    Records none{};
    hipLaunchKernelGGL(k_clip<false>, dim3(nblocks), dim3(kBlock), 0, s, xy, ncell, d_segs.as<double>(),
                       d_cc.as<int>(), nseg, nshift, periodX, (const int *)nullptr, d_cnt.as<int>(), none,
                       d_err.as<unsigned long long>(), skip_unsupported);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, d_cnt.as<int>(), nwaves, d_off.as<int>());
    int nrec_i = 0;
    NF_HIP(hipMemcpyAsync(&nrec_i, d_off.as<int>() + nwaves, sizeof(int), hipMemcpyDeviceToHost, s));
    NF_HIP(hipMemcpyAsync(&err_word, d_err.p, sizeof err_word, hipMemcpyDeviceToHost, s));
    NF_HIP(hipStreamSynchronize(s));
    if (err_word != ~0ull) return refuse(err_word);
    NF_REQUIRE(nrec_i >= 0, NF_ERR_ARG, "weights: more than 2^31 (segment, cell) records; split the transect set");
    const long nrec = nrec_i;

    NF_HIP(hipMalloc((void **)&out->seg_start, sizeof(int) * (size_t)(nseg + 1)));
    if (nrec == 0) {
        NF_HIP(hipMemsetAsync(out->seg_start, 0, sizeof(int) * (size_t)(nseg + 1), s));
        NF_HIP(hipStreamSynchronize(s));
        for (int q = 0; q < nseg; ++q)
            if (segs_host[4 * q + 2] == 0.0 && segs_host[4 * q + 3] == 0.0) out->coverage[(size_t)q] = 1.0;
        return NF_OK;
    }

    DevBuf r_key, r_cell, r_ta, r_tb, r_w, k_out, v_in, v_out, tmp, rstart;
    NF_HIP(r_key.alloc(sizeof(unsigned long long) * nrec));
    NF_HIP(r_cell.alloc(sizeof(int) * nrec));
    NF_HIP(r_ta.alloc(sizeof(double) * nrec));
    NF_HIP(r_tb.alloc(sizeof(double) * nrec));
    NF_HIP(r_w.alloc(sizeof(double) * 4 * nrec));
    Records rec{r_key.as<unsigned long long>(), r_cell.as<int>(), r_ta.as<double>(), r_tb.as<double>(),
                r_w.as<double>()};
    hipLaunchKernelGGL(k_clip<true>, dim3(nblocks), dim3(kBlock), 0, s, xy, ncell, d_segs.as<double>(),
                       d_cc.as<int>(), nseg, nshift, periodX, d_off.as<int>(), (int *)nullptr, rec,
                       d_err.as<unsigned long long>(), skip_unsupported);
    NF_HIP(hipGetLastError());

    // stable sort of record indices by global segment id
    NF_HIP(k_out.alloc(sizeof(unsigned long long) * nrec));
    NF_HIP(v_in.alloc(sizeof(unsigned) * nrec));
    NF_HIP(v_out.alloc(sizeof(unsigned) * nrec));
    const unsigned nb_rec = (unsigned)((nrec + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_iota, dim3(nb_rec), dim3(kBlock), 0, s, v_in.as<unsigned>(), nrec);
    int bits = 1;
    while ((1l << bits) < (long)nseg + 1 && bits < 24) ++bits;
    const unsigned end_bit = (unsigned)(kTaBits + bits);
    size_t tmp_bytes = 0;
    NF_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const unsigned long long *)r_key.p,
                                     k_out.as<unsigned long long>(), v_in.as<unsigned>(), v_out.as<unsigned>(),
                                     (size_t)nrec, 0u, end_bit, s));
    NF_HIP(tmp.alloc(tmp_bytes));
    NF_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, (const unsigned long long *)r_key.p,
                                     k_out.as<unsigned long long>(), v_in.as<unsigned>(), v_out.as<unsigned>(),
                                     (size_t)nrec, 0u, end_bit, s));
    NF_HIP(rstart.alloc(sizeof(int) * (size_t)(nseg + 1)));
    hipLaunchKernelGGL(k_seg_bounds, dim3((unsigned)((nseg + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       k_out.as<unsigned long long>(), nrec, nseg, rstart.as<int>());

    out->nrec = nrec;
    NF_HIP(hipMalloc((void **)&out->cell, sizeof(int) * (size_t)nrec));
    NF_HIP(hipMalloc((void **)&out->w4, sizeof(double) * 4 * (size_t)nrec));
    NF_HIP(hipMalloc((void **)&out->seg, sizeof(int) * (size_t)nrec));
    DevBuf d_len, d_cov;
    NF_HIP(d_len.alloc(sizeof(double) * (size_t)nrec));
    NF_HIP(d_cov.alloc(sizeof(double) * (size_t)(nseg + 1)));
    hipLaunchKernelGGL(k_expand, dim3(nb_rec), dim3(kBlock), 0, s, k_out.as<unsigned long long>(),
                       v_out.as<unsigned>(), nrec, rec, out->cell, out->w4, out->seg, d_len.as<double>());
    if (nseg > 0) {
        hipLaunchKernelGGL(k_seg_coverage, dim3((unsigned)(((long)nseg * kWave + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                           d_len.as<double>(), rstart.as<int>(), nseg, d_cov.as<double>());
        NF_HIP(hipMemcpyAsync(out->coverage.data(), d_cov.p, sizeof(double) * (size_t)nseg, hipMemcpyDeviceToHost, s));
    }
    NF_HIP(hipMemcpyAsync(out->seg_start, rstart.p, sizeof(int) * (size_t)(nseg + 1), hipMemcpyDeviceToDevice, s));
    NF_HIP(hipMemcpyAsync(&err_word, d_err.p, sizeof err_word, hipMemcpyDeviceToHost, s));
    NF_HIP(hipGetLastError());
    NF_HIP(hipStreamSynchronize(s));
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
