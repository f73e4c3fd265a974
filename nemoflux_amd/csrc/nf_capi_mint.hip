// nf_capi_mint.hip -- the C ABI of libnemoflux_amd.so, part 2 of 3: Level 1, the entry points nemoflux reaches in the mint C
// library through python-mint's ctypes wrapper (mnt_grid_*: horizgrid.py:23-24,30,43; mnt_polylineintegral_*: field.py:45-48,
// 102, fluxplot.py:56; mnt_vectorinterp_*: field.py:90-95,119-120).  Host-side orchestration only: every number is produced by
// the HIP kernels of nf_geom.hip / nf_weights.hip / nf_integral.hip / nf_vinterp.hip.  There is no CPU path.
#include "nf_capi.h"

using namespace nf;

// =============================================================================================== Level 1

struct PolylineIntegral_t {
    Grid_t *grid = nullptr;
    bool locator = false;
    double periodX = 0.0;
    WeightSet ws;
    int *d_tr_off = nullptr;
    double *d_scratch = nullptr;
    double *d_row = nullptr;
    // getIntegral on a HOST array stages only the cells the weights touch (GatherStage): the record cell ids stay on the
    // host after computeWeights, the gathered (nrec,4) rows are indexed by RECORD NUMBER on the device (d_iota = 0..nrec-1)
    std::vector<int> h_cell;
    GatherStage stage;
    int *d_iota = nullptr;
    long grid_version = -1;     // the grid build the weights belong to
    int nseg = 0;
    int skip_unsupported = 1;   // mnt_polylineintegral_setUnsupportedCells (default 'skip': mint's computeWeights never fails there)
    int overlap_warn = 0;       // mnt_polylineintegral_setOverlappingCells
};

extern "C" {

int mnt_grid_new(Grid_t **self)
try {
    NF_REQUIRE(self, NF_ERR_ARG, "mnt_grid_new: null argument");
    *self = new Grid_t();
    return NF_OK;
}
NF_API_CATCH
int mnt_grid_del(Grid_t **self)
try {
    if (self && *self) {
        { std::lock_guard<std::mutex> lock((*self)->boxes.mtx); (*self)->boxes.release(); }   // waits for a build that walks them
        if ((*self)->owns_xy) dev_free((*self)->d_xy);
        delete *self;
        *self = nullptr;
    }
    return NF_OK;
}
NF_API_CATCH
int mnt_grid_setPointsPtr(Grid_t **self, double *points)
try {
    NF_REQUIRE(self && *self && points, NF_ERR_ARG, "mnt_grid_setPointsPtr: null argument");
    (*self)->host_points = points;
    return NF_OK;
}
NF_API_CATCH
int mnt_grid_build(Grid_t **self, int nVertsPerCell, long long ncells)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "mnt_grid_build: null grid");
    Grid_t *g = *self;
    NF_REQUIRE(nVertsPerCell == 4, NF_ERR_ARG, "mnt_grid_build: only quad cells (4 vertices) are supported");
    NF_REQUIRE(g->host_points, NF_ERR_STATE, "mnt_grid_build: setPointsPtr first");
    NF_REQUIRE(ncells > 0 && ncells < (1ll << 31), NF_ERR_ARG, "mnt_grid_build: bad cell count");
    NF_NEED_DEVICE();
    std::lock_guard<std::mutex> lock(g->boxes.mtx);   // no computeWeights / findPoints of another thread is inside the locator
    g->boxes.release();             // they describe the old points
    if (g->owns_xy) dev_free(g->d_xy);
    g->owns_xy = true;
    g->ncell = (long)ncells;
    DevTmp points;
    NF_TRY(points.alloc(sizeof(double) * 12 * (size_t)ncells));
    NF_TRY(dev_alloc(&g->d_xy, (size_t)ncells * 8));
    NF_HIP(hipMemcpy(points.p, g->host_points, sizeof(double) * 12 * (size_t)ncells, hipMemcpyHostToDevice));
    NF_TRY(launch_corner_table_from_points(points.as<double>(), g->ncell, g->d_xy, nullptr));
    NF_HIP(hipDeviceSynchronize());
    ++g->version;
    return NF_OK;
}
NF_API_CATCH
int mnt_grid_setRowLength(Grid_t **self, long long rowLength)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "mnt_grid_setRowLength: null grid");
    NF_REQUIRE(rowLength >= 0, NF_ERR_ARG, "mnt_grid_setRowLength: negative row length");
    std::lock_guard<std::mutex> lock((*self)->boxes.mtx);
    if ((*self)->row_length != (long)rowLength) (*self)->boxes.release();    // the groups of the locator follow the layout
    (*self)->row_length = (long)rowLength;
    return NF_OK;
}
NF_API_CATCH
int mnt_grid_getNumberOfCells(Grid_t **self, size_t *numCells)
try {
    NF_REQUIRE(self && *self && numCells, NF_ERR_ARG, "mnt_grid_getNumberOfCells: null argument");
    *numCells = (size_t)(*self)->ncell;
    return NF_OK;
}
NF_API_CATCH
int mnt_grid_dump(Grid_t **self, const char *fileName)
try {
    NF_REQUIRE(self && *self && fileName, NF_ERR_ARG, "mnt_grid_dump: null argument");
    Grid_t *g = *self;
    NF_REQUIRE(g->ncell > 0, NF_ERR_STATE, "mnt_grid_dump: grid not built");
    std::vector<double> pts;
    const double *p = g->host_points;
    if (!p) {  // grid view of a Field: rebuild (lon,lat,0) from the corner table
        NF_NEED_DEVICE();
        DevTmp points;
        NF_TRY(points.alloc(sizeof(double) * 12 * (size_t)g->ncell));
        NF_TRY(launch_points_from_corner_table(g->d_xy, g->ncell, points.as<double>(), nullptr));
        pts.resize((size_t)g->ncell * 12);
        NF_HIP(hipMemcpy(pts.data(), points.p, sizeof(double) * pts.size(), hipMemcpyDeviceToHost));
        p = pts.data();
    }
    FILE *f = fopen(fileName, "w");
    NF_REQUIRE(f, NF_ERR_ARG, std::string("mnt_grid_dump: cannot open ") + fileName);
    fprintf(f, "# vtk DataFile Version 3.0\nnemoflux_amd grid\nASCII\nDATASET UNSTRUCTURED_GRID\n");
    fprintf(f, "POINTS %ld double\n", g->ncell * 4);
    for (long k = 0; k < g->ncell * 4; ++k) fprintf(f, "%.17g %.17g %.17g\n", p[3 * k], p[3 * k + 1], p[3 * k + 2]);
    fprintf(f, "CELLS %ld %ld\n", g->ncell, g->ncell * 5);
    for (long c = 0; c < g->ncell; ++c) fprintf(f, "4 %ld %ld %ld %ld\n", 4 * c, 4 * c + 1, 4 * c + 2, 4 * c + 3);
    fprintf(f, "CELL_TYPES %ld\n", g->ncell);
    for (long c = 0; c < g->ncell; ++c) fprintf(f, "9\n");  // VTK_QUAD
    const bool bad = ferror(f) != 0;
    NF_REQUIRE(fclose(f) == 0 && !bad, NF_ERR_HOST, std::string("mnt_grid_dump: error writing ") + fileName);
    return NF_OK;
}
NF_API_CATCH

int mnt_polylineintegral_new(PolylineIntegral_t **self)
try {
    NF_REQUIRE(self, NF_ERR_ARG, "mnt_polylineintegral_new: null argument");
    *self = new PolylineIntegral_t();
    return NF_OK;
}
NF_API_CATCH
int mnt_polylineintegral_del(PolylineIntegral_t **self)
try {
    if (self && *self) {
        PolylineIntegral_t *p = *self;
        p->ws.release();
        dev_free(p->d_tr_off);
        dev_free(p->d_scratch);
        dev_free(p->d_row);
        dev_free(p->d_iota);
        p->stage.release();
        delete p;
        *self = nullptr;
    }
    return NF_OK;
}
NF_API_CATCH
int mnt_polylineintegral_setGrid(PolylineIntegral_t **self, Grid_t *grid)
try {
    NF_REQUIRE(self && *self && grid, NF_ERR_ARG, "mnt_polylineintegral_setGrid: null argument");
    NF_REQUIRE(grid->ncell > 0 && grid->d_xy, NF_ERR_STATE, "mnt_polylineintegral_setGrid: grid not built");
    (*self)->grid = grid;
    return NF_OK;
}
NF_API_CATCH
int mnt_polylineintegral_buildLocator(PolylineIntegral_t **self, int numCellsPerBucket, double periodX,
                                      int enableFolding)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "mnt_polylineintegral_buildLocator: null argument");
    NF_REQUIRE((*self)->grid, NF_ERR_STATE, "mnt_polylineintegral_buildLocator: setGrid first");
    NF_REQUIRE(numCellsPerBucket > 0, NF_ERR_ARG, "mnt_polylineintegral_buildLocator: numCellsPerBucket <= 0");
    NF_REQUIRE(periodX >= 0.0, NF_ERR_ARG, "mnt_polylineintegral_buildLocator: negative periodX");
    NF_REQUIRE(!enableFolding, NF_ERR_ARG, "mnt_polylineintegral_buildLocator: enableFolding is not supported");
    (*self)->periodX = periodX;
    (*self)->locator = true;  // the box hierarchy itself is built (and kept by the grid) at the first computeWeights
    return NF_OK;
}
NF_API_CATCH

int mnt_polylineintegral_setUnsupportedCells(PolylineIntegral_t **self, int skip)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "mnt_polylineintegral_setUnsupportedCells: null argument");
    NF_REQUIRE(skip == 0 || skip == 1, NF_ERR_ARG, "mnt_polylineintegral_setUnsupportedCells: policy must be 0 (refuse) or 1 (skip)");
    (*self)->skip_unsupported = skip;
    return NF_OK;
}
NF_API_CATCH

int mnt_polylineintegral_getNumberOfDroppedCrossings(PolylineIntegral_t **self, size_t *n)
try {
    NF_REQUIRE(self && *self && n, NF_ERR_ARG, "mnt_polylineintegral_getNumberOfDroppedCrossings: null argument");
    *n = (size_t)(*self)->ws.dropped;
    return NF_OK;
}
NF_API_CATCH

int mnt_polylineintegral_setOverlappingCells(PolylineIntegral_t **self, int warn)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "mnt_polylineintegral_setOverlappingCells: null argument");
    NF_REQUIRE(warn == 0 || warn == 1, NF_ERR_ARG, "mnt_polylineintegral_setOverlappingCells: policy must be 0 (refuse) or 1 (warn)");
    (*self)->overlap_warn = warn;
    return NF_OK;
}
NF_API_CATCH

int mnt_polylineintegral_computeWeights(PolylineIntegral_t **self, int npoints, const double xyz[],
                                        int counterclock)
try {
    NF_REQUIRE(self && *self && xyz, NF_ERR_ARG, "mnt_polylineintegral_computeWeights: null argument");
    PolylineIntegral_t *p = *self;
    NF_REQUIRE(p->grid && p->locator, NF_ERR_STATE, "mnt_polylineintegral_computeWeights: setGrid/buildLocator first");
    NF_REQUIRE(npoints >= 2, NF_ERR_ARG, "mnt_polylineintegral_computeWeights: need at least 2 points");
    NF_NEED_DEVICE();
    std::vector<double> segs;
    std::vector<int> cc;
    p->nseg = polyline_segments(xyz, npoints, counterclock, segs, cc);
    // whatever a previous computeWeights left is gone first: after a refused build getIntegral must say "computeWeights
    // first", not launch on buffers sized for another polyline
    dev_free(p->d_tr_off);
    dev_free(p->d_scratch);
    dev_free(p->d_row);
    dev_free(p->d_iota);
    p->stage.release();
    p->h_cell.clear();
    {
        // objects on one Grid share its locator cache: (re)built and walked under the grid's lock, which is held until the
        // build's kernels have finished (build_weights returns drained) -- two host threads, each with its own
        // PolylineIntegral on a shared Grid, are serialised here instead of racing on the cache
        std::lock_guard<std::mutex> lock(p->grid->boxes.mtx);
        NF_TRY(build_weights(p->grid->d_xy, p->grid->ncell, segs.data(), cc.data(), p->nseg, p->periodX, &p->ws, nullptr,
                             p->skip_unsupported, p->overlap_warn, &p->grid->boxes, p->grid->row_length));
    }
    NF_TRY(dev_alloc(&p->d_tr_off, 2));
    NF_TRY(dev_alloc(&p->d_scratch, (size_t)p->ws.nrec));
    const int off[2] = {0, p->nseg};
    NF_HIP(hipMemcpy(p->d_tr_off, off, sizeof off, hipMemcpyHostToDevice));
    // what getIntegral needs to stage a host array sparsely: the cells of the records, here; their row numbers, there
    p->h_cell.resize((size_t)p->ws.nrec);
    NF_TRY(dev_alloc(&p->d_iota, (size_t)p->ws.nrec));
    if (p->ws.nrec > 0) {
        NF_HIP(hipMemcpy(p->h_cell.data(), p->ws.cell, sizeof(int) * (size_t)p->ws.nrec, hipMemcpyDeviceToHost));
        std::vector<int> iota((size_t)p->ws.nrec);
        for (long k = 0; k < p->ws.nrec; ++k) iota[(size_t)k] = (int)k;
        NF_HIP(hipMemcpy(p->d_iota, iota.data(), sizeof(int) * iota.size(), hipMemcpyHostToDevice));
    }
    NF_TRY(p->stage.resize(p->ws.nrec));
    NF_TRY(dev_alloc(&p->d_row, (size_t)p->nseg + 1));   // last: its presence means "weights are ready"
    p->grid_version = p->grid->version;
    return NF_OK;
}
NF_API_CATCH

// the reduction of one object: gather + wavefront segmented scan + the two finalize kernels, then the row comes back
static int pli_reduce(PolylineIntegral_t *p, const WeightSet &ws, const double *data_dev, long nrows, double *result,
                      double *seg_totals_host)
{
    NF_TRY(launch_integral(ws, data_dev, nrows, 0, 0, p->d_tr_off, 1, p->d_scratch, p->d_row, nullptr));
    std::vector<double> row((size_t)p->nseg + 1);
    NF_HIP(hipMemcpy(row.data(), p->d_row, sizeof(double) * row.size(), hipMemcpyDeviceToHost));
    *result = row[p->nseg];
    if (seg_totals_host) memcpy(seg_totals_host, row.data(), sizeof(double) * p->nseg);
    return NF_OK;
}

static int pli_ready(PolylineIntegral_t *p, int placement)
{
    NF_REQUIRE(p->d_row, NF_ERR_STATE, "mnt_polylineintegral_getIntegral: computeWeights first");
    NF_REQUIRE(p->grid && p->grid_version == p->grid->version, NF_ERR_STATE,
               "mnt_polylineintegral_getIntegral: the grid was rebuilt after computeWeights (the weights index the old cells): computeWeights again");
    NF_REQUIRE(placement == MNT_CELL_BY_CELL_DATA, NF_ERR_ARG,
               "mnt_polylineintegral_getIntegral: only CELL_BY_CELL_DATA is supported (field.py:102)");
    return NF_OK;
}

int mnt_polylineintegral_getIntegralDev(PolylineIntegral_t **self, const double *data_dev, int placement,
                                        double *result, double *seg_totals_host)
try {
    NF_REQUIRE(self && *self && data_dev && result, NF_ERR_ARG, "mnt_polylineintegral_getIntegral: null argument");
    PolylineIntegral_t *p = *self;
    NF_TRY(pli_ready(p, placement));
    NF_NEED_DEVICE();
    return pli_reduce(p, p->ws, data_dev, p->grid->ncell, result, seg_totals_host);
}
NF_API_CATCH

// Host data: mint's getIntegral is a sparse dot over the K = 4 x (cells crossed) entries (field.py:102; fluxplot.py:55-58
// calls it once per transect per time step), so the cost here must not depend on the size of the grid either: the 32
// bytes of every record's cell are gathered on the host into a pinned buffer, nrec x 32 B go to HBM and the SAME kernels
// run on them with the record number as the cell index -- the same products summed in the same tree, hence the bits of
// mnt_polylineintegral_getIntegralDev on the whole array (tests/test_gpu_parity.py::test_level1_host_data_is_staged_sparsely).
int mnt_polylineintegral_getIntegral(PolylineIntegral_t **self, const double data[], int placement, double *result)
try {
    NF_REQUIRE(self && *self && data && result, NF_ERR_ARG, "mnt_polylineintegral_getIntegral: null argument");
    PolylineIntegral_t *p = *self;
    NF_REQUIRE(p->grid, NF_ERR_STATE, "mnt_polylineintegral_getIntegral: setGrid first");
    NF_TRY(pli_ready(p, placement));
    NF_NEED_DEVICE();
    NF_TRY(p->stage.upload(data, p->h_cell.data()));
    WeightSet rows;              // a view of the object's records whose cell index is the record number
    rows.nrec = p->ws.nrec;
    rows.cell = p->d_iota;
    rows.w4 = p->ws.w4;
    rows.seg = p->ws.seg;
    rows.nseg = p->ws.nseg;
    rows.seg_start = p->ws.seg_start;
    return pli_reduce(p, rows, p->stage.d, p->ws.nrec, result, nullptr);
}
NF_API_CATCH

int mnt_polylineintegral_getCoverage(PolylineIntegral_t **self, double *coverage)
try {
    NF_REQUIRE(self && *self && coverage, NF_ERR_ARG, "mnt_polylineintegral_getCoverage: null argument");
    // also readable after a build that was refused for over-coverage (the message names one segment; this has them all)
    NF_REQUIRE((*self)->d_row || !(*self)->ws.coverage.empty(), NF_ERR_STATE,
               "mnt_polylineintegral_getCoverage: computeWeights first");
    const std::vector<double> &c = (*self)->ws.coverage;
    if (!c.empty()) memcpy(coverage, c.data(), sizeof(double) * c.size());
    return NF_OK;
}
NF_API_CATCH
int mnt_polylineintegral_getNumberOfWeights(PolylineIntegral_t **self, size_t *n)
try {
    NF_REQUIRE(self && *self && n, NF_ERR_ARG, "mnt_polylineintegral_getNumberOfWeights: null argument");
    *n = (size_t)(*self)->ws.entries();
    return NF_OK;
}
NF_API_CATCH
int mnt_polylineintegral_getWeights(PolylineIntegral_t **self, int64_t *cell_edge, double *weight, int *seg)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "mnt_polylineintegral_getWeights: null argument");
    if ((*self)->ws.nrec == 0) return NF_OK;
    NF_NEED_DEVICE();
    return weights_to_host((*self)->ws, cell_edge, weight, seg);
}
NF_API_CATCH

}  // extern "C"

// ----------------------------------------------------------------------------------------------- VectorInterp
struct VectorInterp_t {
    Grid_t *grid = nullptr;
    bool locator = false;
    double periodX = 0.0;
    long npts = 0;
    double *d_targets = nullptr, *d_pcoords = nullptr, *d_vectors = nullptr;
    long *d_cell = nullptr;
    unsigned long long *d_best = nullptr;
    // getFaceVectors on a HOST array stages only the located cells, one (4) row per target point (GatherStage)
    std::vector<long> h_cell;
    GatherStage stage;
    long grid_version = -1;     // the grid build the located cells belong to
};

static void vi_free_points(VectorInterp_t *v)
{
    dev_free(v->d_targets);
    dev_free(v->d_pcoords);
    dev_free(v->d_vectors);
    dev_free(v->d_cell);
    dev_free(v->d_best);
    v->stage.release();
    v->h_cell.clear();
    v->npts = 0;
}

extern "C" {

int mnt_vectorinterp_new(VectorInterp_t **self)
try {
    NF_REQUIRE(self, NF_ERR_ARG, "mnt_vectorinterp_new: null argument");
    *self = new VectorInterp_t();
    return NF_OK;
}
NF_API_CATCH
int mnt_vectorinterp_del(VectorInterp_t **self)
try {
    if (self && *self) {
        vi_free_points(*self);
        delete *self;
        *self = nullptr;
    }
    return NF_OK;
}
NF_API_CATCH
int mnt_vectorinterp_setGrid(VectorInterp_t **self, Grid_t *grid)
try {
    NF_REQUIRE(self && *self && grid, NF_ERR_ARG, "mnt_vectorinterp_setGrid: null argument");
    NF_REQUIRE(grid->ncell > 0 && grid->d_xy, NF_ERR_STATE, "mnt_vectorinterp_setGrid: grid not built");
    (*self)->grid = grid;
    return NF_OK;
}
NF_API_CATCH
int mnt_vectorinterp_buildLocator(VectorInterp_t **self, int numCellsPerBucket, double periodX, int enableFolding)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "mnt_vectorinterp_buildLocator: null argument");
    NF_REQUIRE((*self)->grid, NF_ERR_STATE, "mnt_vectorinterp_buildLocator: setGrid first");
    NF_REQUIRE(numCellsPerBucket > 0 && periodX >= 0.0, NF_ERR_ARG, "mnt_vectorinterp_buildLocator: bad arguments");
    NF_REQUIRE(!enableFolding, NF_ERR_ARG, "mnt_vectorinterp_buildLocator: enableFolding is not supported");
    (*self)->periodX = periodX;
    (*self)->locator = true;
    return NF_OK;
}
NF_API_CATCH
int mnt_vectorinterp_findPoints(VectorInterp_t **self, size_t numPoints, const double targetPoints[], double tol2,
                                size_t *numNotFound)
try {
    NF_REQUIRE(self && *self && (targetPoints || numPoints == 0), NF_ERR_ARG, "mnt_vectorinterp_findPoints: null argument");
    VectorInterp_t *v = *self;
    NF_REQUIRE(v->grid && v->locator, NF_ERR_STATE, "mnt_vectorinterp_findPoints: setGrid/buildLocator first");
    NF_REQUIRE(tol2 >= 0.0, NF_ERR_ARG, "mnt_vectorinterp_findPoints: negative tolerance");
    NF_NEED_DEVICE();
    vi_free_points(v);
    v->npts = (long)numPoints;
    if (numNotFound) *numNotFound = 0;
    if (numPoints == 0) return NF_OK;
    NF_TRY(dev_alloc(&v->d_targets, numPoints * 3));
    NF_TRY(dev_alloc(&v->d_pcoords, numPoints * 2));
    NF_TRY(dev_alloc(&v->d_vectors, numPoints * 3));
    NF_TRY(dev_alloc(&v->d_cell, numPoints));
    NF_TRY(dev_alloc(&v->d_best, numPoints));
    NF_HIP(hipMemcpy(v->d_targets, targetPoints, sizeof(double) * 3 * numPoints, hipMemcpyHostToDevice));
    v->h_cell.resize(numPoints);      // the located cells stay on the host too: they address the caller's host arrays
    {
        std::lock_guard<std::mutex> lock(v->grid->boxes.mtx);   // the grid's locator cache: see computeWeights
        NF_TRY(launch_find_points(v->grid->d_xy, v->grid->ncell, v->grid->row_length, &v->grid->boxes, v->d_targets, v->npts,
                                  v->periodX, tol2, v->d_best, v->d_cell, v->d_pcoords, nullptr));
        // (the blocking copy on the null stream orders itself behind the search: the walk is over when it returns)
        NF_HIP(hipMemcpy(v->h_cell.data(), v->d_cell, sizeof(long) * numPoints, hipMemcpyDeviceToHost));
    }
    NF_TRY(v->stage.resize((long)numPoints));
    if (numNotFound) {
        size_t n = 0;
        for (long c : v->h_cell) n += (c < 0);
        *numNotFound = n;
    }
    v->grid_version = v->grid->version;
    return NF_OK;
}
NF_API_CATCH
/* layout: 0 = (ncell,4) AoS, 1 = [4][ncell] planes (the engine's resident layout), 2 = (npts,4) rows gathered per point */
static int vi_vectors(VectorInterp_t *v, const double *data_dev, int layout, double vectors[])
{
    NF_REQUIRE(vectors, NF_ERR_ARG, "mnt_vectorinterp_getFaceVectors: null output");
    NF_REQUIRE(v->grid_version == v->grid->version, NF_ERR_STATE,
               "mnt_vectorinterp_getFaceVectors: the grid was rebuilt after findPoints (the located cells are the old grid's): findPoints again");
    NF_NEED_DEVICE();
    NF_TRY(launch_face_vectors(v->grid->d_xy, v->d_cell, v->d_pcoords, v->npts, data_dev, v->grid->ncell, layout,
                               v->periodX, v->d_vectors, nullptr));
    NF_HIP(hipMemcpy(vectors, v->d_vectors, sizeof(double) * 3 * v->npts, hipMemcpyDeviceToHost));
    return NF_OK;
}
int mnt_vectorinterp_getFaceVectorsDev(VectorInterp_t **self, const double *data_dev, int layout, double vectors[])
try {
    NF_REQUIRE(self && *self && data_dev, NF_ERR_ARG, "mnt_vectorinterp_getFaceVectors: null argument");
    VectorInterp_t *v = *self;
    NF_REQUIRE(v->grid, NF_ERR_STATE, "mnt_vectorinterp_getFaceVectors: setGrid first");
    NF_REQUIRE(layout == 0 || layout == 1, NF_ERR_ARG, "mnt_vectorinterp_getFaceVectorsDev: layout must be 0 ((ncell,4)) or 1 ([4][ncell] planes)");
    if (v->npts == 0) return NF_OK;
    return vi_vectors(v, data_dev, layout, vectors);
}
NF_API_CATCH
// Host data: only the rows of the located cells travel (npts x 32 B, not ncell x 32 B): field.py:119 calls this at every
// update() of the viewer.  Same arithmetic on the same values as the resident-data call: same bits.
int mnt_vectorinterp_getFaceVectors(VectorInterp_t **self, const double data[], int placement, double vectors[])
try {
    NF_REQUIRE(self && *self && data, NF_ERR_ARG, "mnt_vectorinterp_getFaceVectors: null argument");
    VectorInterp_t *v = *self;
    NF_REQUIRE(v->grid, NF_ERR_STATE, "mnt_vectorinterp_getFaceVectors: setGrid first");
    NF_REQUIRE(placement == MNT_CELL_BY_CELL_DATA, NF_ERR_ARG,
               "mnt_vectorinterp_getFaceVectors: only CELL_BY_CELL_DATA (placement=0) is supported (field.py:94-95)");
    if (v->npts == 0) return NF_OK;
    NF_NEED_DEVICE();
    NF_TRY(v->stage.upload(data, v->h_cell.data()));
    return vi_vectors(v, v->stage.d, 2, vectors);
}
NF_API_CATCH
int mnt_vectorinterp_getCells(VectorInterp_t **self, long long *cell_ids, double *pcoords)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "mnt_vectorinterp_getCells: null argument");
    VectorInterp_t *v = *self;
    if (v->npts == 0) return NF_OK;
    NF_NEED_DEVICE();
    if (cell_ids) NF_HIP(hipMemcpy(cell_ids, v->d_cell, sizeof(long) * v->npts, hipMemcpyDeviceToHost));
    if (pcoords) NF_HIP(hipMemcpy(pcoords, v->d_pcoords, sizeof(double) * 2 * v->npts, hipMemcpyDeviceToHost));
    return NF_OK;
}
NF_API_CATCH

}  // extern "C"
