// nf_common.h -- shared declarations of the gfx950 transect-flux engine (internal; the public surface is
// include/nemoflux_amd.h).  Written for CDNA4 only: 64-lane wavefronts, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <cmath>
#include <initializer_list>
#include <mutex>
#include <vector>

#include "../../include/nemoflux_amd.h"

namespace nf {

// [flux-fingerprint-begin]  (bench.flux_source_sha16 hashes nf_flux.hip and the marked parts of this header: what K1 is built from)
constexpr int kWave = 64;        // CDNA wavefront
constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kXcds = 8;         // MI355X: 8 XCDs, workgroups are dealt round-robin over them
constexpr double kDeg2Rad = 3.14159265358979323846 / 180.0;  // geo.py:4
constexpr double kEarthRadiusSv = 6371000.0;                 // field.py:12
// [flux-fingerprint-end]

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define NF_HIP(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) return nf::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

#define NF_REQUIRE(cond, code, msg)            \
    do {                                       \
        if (!(cond)) {                         \
            nf::set_error(msg);                \
            return (code);                     \
        }                                      \
    } while (0)

#define NF_TRY(call)                  \
    do {                              \
        int rc_ = (call);             \
        if (rc_ != NF_OK) return rc_; \
    } while (0)

// [flux-fingerprint-begin]
// ---- XCD-aware tile mapping -------------------------------------------------------------------------
// Workgroup b lands on XCD b % 8.  Give every XCD one CONTIGUOUS band of logical tiles so that (a) the
// neighbour-slot stores of the edge-flux kernel (row j -> row j+1, column i -> i+1) meet the owning row's
// stores in the SAME L2 and leave it as whole lines, and (b) each L2 streams one band of every slab.
// Grid must be launched with xcd_grid(ntiles) workgroups; tiles >= ntiles exit.
__host__ __device__ inline unsigned xcd_grid(unsigned ntiles) { return ((ntiles + kXcds - 1) / kXcds) * kXcds; }
__device__ inline unsigned xcd_tile(unsigned b, unsigned grid) { return (b % kXcds) * (grid / kXcds) + b / kXcds; }
// [flux-fingerprint-end]

// ---- cells the weights / the point location are not defined on (docs/PARITY.md) --------------------------------
// v = (x0,y0,...,x3,y3) of a quad in the (lon,lat) plane
__device__ inline bool quad_is_nonconvex(const double *v)
{
    // a corner AT a geographic pole that is not the end of an edge lying on the pole line (|lat| = 90 along a whole edge,
    // as in the top row of an un-rotated lon-lat grid, is fine): the pole's longitude is arbitrary, so the planar quad is
    // not the image of the cell, convex or not
    int npole = 0, first = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (fabs(v[2 * k + 1]) >= 90.0 - 1.e-9) {
            ++npole;
            if (first < 0) first = k;
        }
    if (npole == 1 || npole == 3) return true;
    if (npole == 2 && !(fabs(v[2 * ((first + 1) & 3) + 1]) >= 90.0 - 1.e-9 || (first == 0 && fabs(v[2 * 3 + 1]) >= 90.0 - 1.e-9)))
        return true;   // opposite corners
    // a geographic pole INSIDE the cell (a rotated grid whose pole is not a mesh node): going round the four corners the
    // longitude winds once around the globe -- the differences, each taken the short way, add up to +-360 instead of 0 --
    // and the planar quad is a sliver along the pole line, not the image of the cell
    double turn = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double d = v[2 * ((k + 1) & 3)] - v[2 * k];
        d -= 360.0 * rint(d / 360.0);
        turn += d;
    }
    if (fabs(turn) > 180.0) return true;
    double cmin = 0.0, cmax = 0.0, scale = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int k1 = (k + 1) & 3, k2 = (k + 2) & 3;
        const double ex = v[2 * k1] - v[2 * k], ey = v[2 * k1 + 1] - v[2 * k + 1];
        const double fx = v[2 * k2] - v[2 * k1], fy = v[2 * k2 + 1] - v[2 * k1 + 1];
        const double cr = ex * fy - ey * fx;
        cmin = fmin(cmin, cr);
        cmax = fmax(cmax, cr);
        scale = fmax(scale, ex * ex + ey * ey);
    }
    return cmin < -1.e-12 * scale && cmax > 1.e-12 * scale;
}


// ---- date-line unwrap (round 4) -----------------------------------------------------------------------------------------
// A global file stores bounds_lon wrapped into one period (say [-180,180]), so the cell that straddles the cut has corners
// ~350 degrees apart: as a planar quad it is a clockwise sliver across the whole domain that the clip would accept and count
// a second time.  With a periodic locator (periodX > 0) every corner is brought to within periodX/2 of corner 0 in the
// lane's REGISTER copy of the cell -- the rule the reference's own generator applies to its rotated grids
// (datagen.py:161-166, with 270 degrees there).  The corner table in HBM, getPoints() and the arc lengths keep the file's
// values.  mint's own behaviour on such cells is parity unpinned.
__device__ inline void unwrap_quad(double *v, double periodX)
{
    if (!(periodX > 0.0)) return;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const double n = rint((v[2 * k] - v[0]) / periodX);
        if (n != 0.0) v[2 * k] -= n * periodX;
    }
}
// a cell with a corner that is not a finite number (NaN or infinite bounds on land-only subdomains) is no cell at all: it
// takes part in nothing and the coverage of a line through it says so
__device__ inline bool quad_is_finite(const double *v)
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) ok = ok && (fabs(v[k]) <= 1.7976931348623157e308);
    return ok;
}
// A target segment covered more than once is counted twice somewhere (overlapping cells): an error.  Two conditions, both
// needed (round-4 advisor): the excess in the segment's parameter, cov - 1 > kCoverTol, AND the same excess as a LENGTH,
// (cov - 1) |d| > kCoverLenTol max(1, |coordinates|) degrees.  The parameter of a sub-segment end carries a rounding error of
// about eps |coordinates| / |d|: on a target segment of 1e-9 degrees across a cell edge that is 1e-6 in t -- the two cells'
// pieces then no longer match within kTolT and the shared stretch counts twice, an excess of up to 5e-7 in t but 1e-15
// degrees of line.  Real overlaps (a date-line cell with periodX = 0, duplicated cells) double whole cell crossings.
constexpr double kCoverTol = 1.e-8;
constexpr double kCoverLenTol = 1.e-9;
inline bool over_covered(double cov, const double *seg4 /* x0, y0, dx, dy */)
{
    if (!(cov > 1.0 + kCoverTol)) return false;
    const double len = std::sqrt(seg4[2] * seg4[2] + seg4[3] * seg4[3]);
    double m = 1.0;
    for (double c : {seg4[0], seg4[1], seg4[0] + seg4[2], seg4[1] + seg4[3]}) m = std::fabs(c) > m ? std::fabs(c) : m;
    return (cov - 1.0) * len > kCoverLenTol * m;
}

// ---- launchers (defined in the .hip files) ----------------------------------------------------------
// K0: geometry.  bounds (ncell,4) of T -> corner table xy (ncell,4,2), arc (ncell,4), arcE/arcN (ncell),
// lon/lat box (4 doubles: lonmin, lonmax, latmin, latmax as order-preserving keys; see nf_geom.hip).
int launch_geometry(const void *blon, const void *blat, int dtype, long ncell, double *xy, double *arc4,
                    double *arcE, double *arcN, unsigned long long *box_keys, hipStream_t s);
int launch_corner_table_from_points(const double *points, long ncell, double *xy, hipStream_t s);
int launch_points_from_corner_table(const double *xy, long ncell, double *points, hipStream_t s);
double box_key_to_double(unsigned long long k);

// [flux-fingerprint-begin]
// several time steps in one launch (launch-bound small grids): step tb reads u,v + tb*in_stride, integrates levels
// [zr[2tb], zr[2tb+1]) and writes planes iV + tb*4*ncell, abs + tb*2*ncell.  zr == nullptr: one step (z0, z1).
struct StepBatch {
    int nsteps = 0;
    long in_stride = 0;       // elements of the field dtype between consecutive time steps
    const int *zr = nullptr;  // device, 2*nsteps
};

// K1: vertical integral of one time step's slabs [z0,z1) + edge fluxes (the bandwidth-bound kernel).
struct FluxArgs {
    const void *u, *v;        // base of the time step: (nz, ncell)
    int dtype;                // NF_F64 / NF_F32
    long ncell, ny, nx;
    int z0, z1;
    const double *thickness;  // device, nz
    const double *arcE, *arcN;
    double fill;              // NaN = none
    double fill2 = __builtin_nan("");   // a second missing marker (CF missing_value that differs from _FillValue); NaN = none
    double scale;             // 1 or 6371000/1e6
    int sverdrup;
    double *iV, *absU, *absV; // resident outputs
    unsigned long long *maxbits;  // running max as the bits of a non-negative double
    StepBatch batch;
    int signed_only = 0;      // 1: store only planes 1 (eU) and 2 (eV); launch_expand_planes derives the other four
    // timing: when the default step runs as flux kernel + expansion, mid_event is recorded between the two launches
    hipEvent_t mid_event = nullptr;
    bool *mid_recorded = nullptr;
};
int launch_flux(const FluxArgs &a, hipStream_t s);
bool flux_supports_signed_only(const FluxArgs &a);
// planes 0 (south copies), 3 (west copies incl. the periodic wrap) and |eU|, |eV| from planes 1 and 2 (field.py:209-232)
int launch_expand_planes(double *iV, double *absUV, long ncell, long ny, long nx, hipStream_t s);
int tuning_set(const char *name, int value);
long tuning_version();
int launch_planes_to_aos(const double *planes, long ncell, double *aos, hipStream_t s);
// [flux-fingerprint-end]

// K2: batched polyline weights.
struct WeightSet {  // device-resident result: one record per (target segment, crossed cell), sorted by segment, ta
    long nrec = 0;
    int *cell = nullptr;       // cell id of the record
    double *w4 = nullptr;      // 4 edge weights per record (S,E,N,W), multiplicity applied
    int *seg = nullptr;        // global segment id of the record
    int nseg = 0;              // total target segments
    int *seg_start = nullptr;  // (nseg+1) CSR over records
    // host: fraction of every target segment that lies inside cells of the grid (sum of coef*(tb-ta) over its records);
    // 1 = inside the grid, each point counted once; < 1 = part of the segment is outside (contributes 0, like mint)
    std::vector<double> coverage;
    int over_seg = -1;         // first target segment build_weights found covered more than once (over_covered), or -1
    long dropped = 0;          // skip policy: (cell, segment image) crossings of unsupported cells that were left out
    // Unique-edge form for the engine's own planes (fold_weights): the south / west slots of integratedVelocity are copies
    // of the neighbours' north / east values (field.py:219-223), so every (cell, edge) weight is folded onto the element
    // of the two signed planes that really carries it and duplicates are merged per target segment (adjacent cells
    // share an edge): about half as many entries as 4 per record, ONE gather each.
    struct EdgeEntry {
        int elem;   // index into [eU | eV] = planes 1 and 2 taken as one array of 2*ncell values
        int seg;    // global target segment
        double w;   // summed weight
    };
    long nent = 0;
    EdgeEntry *ent = nullptr;    // sorted by (segment, elem)
    int *ent_start = nullptr;    // (nseg+1) CSR over entries
    long entries() const { return 4 * nrec; }  // mint's view: (cell*4+edge, weight) entries
    void release();
};
// builds ws.ent / ws.ent_start from the records, for a grid of nx columns (row-0 south slots carry no flux: dropped)
int fold_weights(WeightSet *ws, long ncell, long nx, hipStream_t s);
// expands records into mint-style entries (host arrays): cell_edge = cell*4+edge, weight, seg
int weights_to_host(const WeightSet &ws, int64_t *cell_edge, double *weight, int *seg);
// segs_host: (nseg,4) = x0,y0,dx,dy ; seg_cc_host: counterclock flag per segment
// skip_unsupported: 0 = a target segment that overlaps a non-convex / pole-vertex cell is an error (default); 1 = such
// cells contribute nothing and the segment's coverage is < 1
// The locator of a grid (nf_weights.hip: bounding boxes of the cells and of groups of 16, 256, ... consecutive cells) for one
// period, kept by whoever owns the corner table -- mint's buildLocator builds it once per PolylineIntegral; here a Grid_t keeps
// it for all the PolylineIntegral objects made on it (fluxviz / fluxplot make one per transect).  build_weights fills it when
// it does not match the grid it is called with.
struct LocatorBoxes {
    std::vector<void *> level;   // level[l]: HBM array of the boxes of level l (0 = cells)
    std::vector<long> count;
    const double *xy = nullptr;  // the corner table the boxes were built from
    long ncell = 0;
    double period = -1.0;
    // Objects that share a Grid share this cache.  Whoever may (re)build it or walks it holds `mtx` for the whole call
    // (computeWeights / findPoints are synchronous: the kernels that read the boxes have finished when the call returns), and
    // so does whoever releases it (new points, new row length, grid deletion): two host threads driving their own
    // PolylineIntegral objects on one Grid are serialised here instead of racing on release() (round-5 advisor).
    std::mutex mtx;
    void release();
    ~LocatorBoxes() { release(); }
};
// overlap_warn: 0 = a target segment covered more than once (over_covered) is an error (default); 1 = the build goes through,
// out->over_seg names the first such segment and the coverage says how much (the caller warns)
// boxes: nullptr = the locator lives for this build only (the batched build of a Field)
// row_length: the cells are rows of this many (a Field's (ny, nx) grid: the locator groups them in 4 x 4 blocks); 0 = unknown
int build_weights(const double *xy, long ncell, const double *segs_host, const int *seg_cc_host, int nseg,
                  double periodX, WeightSet *out, hipStream_t s, int skip_unsupported = 0, int overlap_warn = 0,
                  LocatorBoxes *boxes = nullptr, long row_length = 0);
// gives the idle scratch memory that weight builds / point searches keep between calls (a process-wide pool of at most 4
// scratches of at most 1 GiB each) back to the system
void weights_trim_scratch();

// K3: gather + wavefront segmented reduction -> per-segment sums, then per-transect sums.
// row: (nseg + ntransect) doubles in HBM; tr_offsets_dev: (ntransect+1) segment offsets.
// scratch: at least ws.nrec doubles.
// data: (ncell,4) AoS (planes = 0), [4][ncell] planes (planes = 1), or the engine's own planes read through their two
// signed members only (planes = 2; needs nx for the neighbour indexing).
// nsteps > 1: step tb gathers from data + tb*data_stride and writes row + tb*row_stride (scratch: nsteps*ws.nrec).
// planes = 2 uses the unique-edge entries when fold_weights built them (scratch: nsteps * max(ws.nrec, ws.nent)).
void integral_use_edges(int on);
int integral_uses_edges();
int launch_integral(const WeightSet &ws, const double *data, long ncell, int planes, long nx,
                    const int *tr_offsets_dev, int ntransect, double *scratch, double *row, hipStream_t s,
                    int nsteps = 1, long data_stride = 0, long row_stride = 0);

// VectorInterp (field.py:90-95,119-120)
// targets_dev: caller order (n,3); sorted_dev: the same points sorted by y; order_dev: caller index of sorted point q
int launch_find_points(const double *xy, long ncell, long row_length, LocatorBoxes *keep, const double *targets_dev, long npts,
                       double periodX, double tol2, unsigned long long *best_dev, long *cell_dev, double *pcoords_dev,
                       hipStream_t s);
int launch_face_vectors(const double *xy, const long *cell_dev, const double *pcoords_dev, long npts, const double *data,
                        long ncell, int planes, double periodX, double *vectors_dev, hipStream_t s);

// datagen
int launch_datagen_bounds(double *blon, double *blat, long ny, long nx, double xmin, double xmax, double ymin,
                          double ymax, double dlon, double dlat, int lat_uses_dx, hipStream_t s);
void datagen_use_rows(int on);   // 0: always the one-cell-per-lane generator kernel (plain division): the row kernel's reference
int launch_datagen_uv(void *u, void *v, int dtype, long t0, long t1, long nt, long nz, long ny, long nx,
                      double xmin, double xmax, double ymin, double ymax, double zmin, double zmax,
                      int lat_uses_dx, int psi, hipStream_t s);

}  // namespace nf
