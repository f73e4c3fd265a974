// nf_capi_field.hip -- the C ABI of libnemoflux_amd.so, part 3 of 3: Level 2, the Field-shaped engine (field.py:15-234;
// BASELINE north_star's computeFlux(tIndex)): nf_field_*.  Host-side orchestration only: every number is produced by the HIP
// kernels of nf_geom.hip / nf_flux.hip / nf_weights.hip / nf_integral.hip.  There is no CPU path.
#include "nf_capi.h"

using namespace nf;

// 1 = compute_all may put all time steps of a small grid into one launch per kernel ("batch_steps" tuning knob)
static int g_batch_steps = 1;
// nf_tuning_set knobs of this file (the library reads no environment variable):
//   "batch_cellsteps_m"    the all-steps-in-one-launch form is used while nt*ncell stays under this many Mi cell-steps (32)
//   "partial_step_planes"  1 = a rank's PARTIAL time steps keep the six-plane epilogue (the code before round 4: the before
//                          leg of profiles/r04_rank_emulation.txt); 0 = signed planes only (default)
//   "graph"                0 = nf_field_compute_all_async never replays a captured graph of the pass (default 1)
static long g_batch_cellsteps = 32l << 20;
static int g_partial_full = 0;
static int g_use_graph = 1;

namespace nf {
int field_tuning_set(const char *name, int value)
{
    if (!strcmp(name, "batch_steps")) {
        g_batch_steps = value;
        return NF_OK;
    }
    if (!strcmp(name, "batch_cellsteps_m")) {
        if (!(value >= 0 && value <= 2047)) return -1;   // nf_tuning_set then reports the knob as unknown / out of range
        g_batch_cellsteps = (long)value << 20;
        return NF_OK;
    }
    if (!strcmp(name, "partial_step_planes")) {
        g_partial_full = value != 0;
        return NF_OK;
    }
    if (!strcmp(name, "graph")) {
        g_use_graph = value != 0;
        return NF_OK;
    }
    return -1;
}
}  // namespace nf

// =============================================================================================== Level 2
struct nf_field {
    hipStream_t stream = nullptr;
    long ny = 0, nx = 0, ncell = 0, nz = 0, nt = 0;
    // geometry
    double *d_xy = nullptr, *d_arc4 = nullptr, *d_arcE = nullptr, *d_arcN = nullptr;
    unsigned long long *d_box = nullptr;
    double box[4] = {0, 0, 0, 0};
    double *d_thick = nullptr;
    // velocity fields
    const void *u = nullptr, *v = nullptr;
    int uv_dtype = NF_F64, uv_on_device = 1;
    double fill = std::numeric_limits<double>::quiet_NaN();
    double fill2 = std::numeric_limits<double>::quiet_NaN();   // nf_field_set_missing_value
    void *d_stage_u = nullptr, *d_stage_v = nullptr;
    int sverdrup = 0;
    long s_begin = 0, s_end = -1;
    // resident per-step outputs
    double *d_iV = nullptr;   // [4][ncell]
    double *d_abs = nullptr;  // [2][ncell]
    double *d_aos = nullptr;  // (ncell,4) re-pack buffer for read_step, allocated on first use
    // compact resident mode (nf_field_set_compact): the flux kernel stores only eU and eV; the neighbour-copy planes and
    // the two |.| planes are derived when somebody asks for them (read_step, device_ptr)
    int compact = 0;
    bool derived_stale = false;
    // multi-step launches for small grids (compute_all): per-step planes, scratch and z ranges
    double *d_iVb = nullptr, *d_absb = nullptr, *d_scratchb = nullptr;
    int *d_zr = nullptr;
    long batch_steps = 0, batch_version = -1;
    unsigned long long *d_maxbits = nullptr;
    // transects
    std::vector<std::vector<double>> polylines;
    std::vector<int> poly_cc;
    std::vector<int> tr_off;
    WeightSet ws;
    bool weights_built = false;
    int skip_unsupported = 0;   // nf_field_set_unsupported_cells
    int overlap_warn = 0;       // nf_field_set_overlapping_cells
    int *d_tr_off = nullptr;
    double *d_scratch = nullptr, *d_row = nullptr;
    Grid_t grid_view;
    // timing
    bool timing = false;
    struct TimedLaunch {         // events around one flux launch (+ expansion); mid sits between the two kernels;
        hipEvent_t e0 = nullptr, mid = nullptr, e1 = nullptr, e2 = nullptr;   // e2 closes the transect reduction (K3)
        bool has_mid = false, has_k3 = false;
    };
    std::vector<TimedLaunch> ev;     // pool: created once (nf_field_timing reserves), re-used after every timing_read
    size_t ev_used = 0;              // launches recorded since the last timing_read
    long ev_dropped = 0;             // launches not recorded because the pool was at its cap
    double last_flux_ms = 0.0, last_expand_ms = 0.0, last_k3_ms = 0.0;   // split of the last timing_read
    // hipGraph of one compute_all pass (launch-bound small grids: 4 launches per time step)
    hipGraphExec_t graph_exec = nullptr;
    double *graph_rows = nullptr;
    long graph_version = -1, version = 0;  // version is bumped by every call that changes what a pass launches
};

static void field_drop_graph(nf_field *f)
{
    if (f->graph_exec) (void)hipGraphExecDestroy(f->graph_exec);
    f->graph_exec = nullptr;
    f->graph_rows = nullptr;
    f->graph_version = -1;
}

static int field_free_geometry(nf_field *f)
{
    dev_free(f->d_xy);
    dev_free(f->d_arc4);
    dev_free(f->d_arcE);
    dev_free(f->d_arcN);
    dev_free(f->d_box);
    dev_free(f->d_iV);
    dev_free(f->d_abs);
    dev_free(f->d_aos);
    dev_free(f->d_iVb);
    dev_free(f->d_absb);
    dev_free(f->d_scratchb);
    dev_free(f->d_zr);
    f->batch_steps = 0;
    f->batch_version = -1;
    dev_free(f->d_maxbits);
    // the locator cache and the lent grid describe the corner table that was just freed: gone with it, on EVERY path out of
    // set_bounds (an early return used to leave boxes keyed on the freed pointer, which hipMalloc often hands out again:
    // round-5 advisor)
    {
        std::lock_guard<std::mutex> lock(f->grid_view.boxes.mtx);
        f->grid_view.boxes.release();
    }
    f->grid_view.d_xy = nullptr;
    f->grid_view.ncell = 0;
    ++f->grid_view.version;
    f->weights_built = false;
    return NF_OK;
}

static int elem_size(int dtype) { return dtype == NF_F32 ? 4 : 8; }

static void field_drop_events(nf_field *f)
{
    for (auto &t : f->ev)
        for (hipEvent_t e : {t.e0, t.mid, t.e1, t.e2})
            if (e) (void)hipEventDestroy(e);
    f->ev.clear();
    f->ev_used = 0;
    f->ev_dropped = 0;
}

// event triples are created outside the timed region (nf_field_timing(n) reserves n) and re-used; the pool never grows
// past kMaxTimedLaunches, so a caller that never reads the timing does not leak events
constexpr size_t kMaxTimedLaunches = 1 << 16;
static int field_reserve_events(nf_field *f, size_t n)
{
    if (n > kMaxTimedLaunches) n = kMaxTimedLaunches;
    while (f->ev.size() < n) {   // an entry joins the pool only when all of its events exist
        nf_field::TimedLaunch t;
        hipError_t err = hipSuccess;
        for (hipEvent_t *e : {&t.e0, &t.mid, &t.e1, &t.e2})
            if (err == hipSuccess) err = hipEventCreate(e);
        if (err != hipSuccess) {
            for (hipEvent_t e : {t.e0, t.mid, t.e1, t.e2})
                if (e) (void)hipEventDestroy(e);
            NF_HIP(err);
        }
        f->ev.push_back(t);
    }
    return NF_OK;
}

// one flux launch (and, in the default step, the expansion behind it) bracketed by events on the field's stream
static int field_timed_flux(nf_field *f, FluxArgs &a)
{
    if (f->ev_used >= kMaxTimedLaunches) {   // nobody reads the timing: keep computing, stop recording
        ++f->ev_dropped;
        return launch_flux(a, f->stream);
    }
    NF_TRY(field_reserve_events(f, f->ev_used + 1));   // no-op when nf_field_timing reserved enough
    const size_t k = f->ev_used++;
    bool mid = false;
    a.mid_event = f->ev[k].mid;
    a.mid_recorded = &mid;
    NF_HIP(hipEventRecord(f->ev[k].e0, f->stream));
    const int rc = launch_flux(a, f->stream);
    a.mid_event = nullptr;
    a.mid_recorded = nullptr;
    f->ev[k].has_mid = mid;
    f->ev[k].has_k3 = false;
    NF_TRY(rc);
    NF_HIP(hipEventRecord(f->ev[k].e1, f->stream));
    return NF_OK;
}

// the transect reduction launched right behind a timed flux launch: e1 .. e2 of the same entry
static int field_timed_k3_end(nf_field *f)
{
    if (!f->timing || f->ev_used == 0 || f->ev_dropped) return NF_OK;
    auto &t = f->ev[f->ev_used - 1];
    NF_HIP(hipEventRecord(t.e2, f->stream));
    t.has_k3 = true;
    return NF_OK;
}

// compact mode: bring planes 0, 3 and the |.| planes up to date with planes 1, 2 of the latest step
static int field_ensure_derived(nf_field *f)
{
    if (!f->derived_stale) return NF_OK;
    NF_TRY(launch_expand_planes(f->d_iV, f->d_abs, f->ncell, f->ny, f->nx, f->stream));
    f->derived_stale = false;
    return NF_OK;
}

static int field_row_length(const nf_field *f) { return f->ws.nseg + (int)f->polylines.size(); }

// one time step on the field's stream; row_dev receives [segments | transects]
static int field_step_async(nf_field *f, long t, double *row_dev)
{
    NF_REQUIRE(f->d_arcE && f->d_thick && f->u && f->v, NF_ERR_STATE,
               "compute: set_bounds, set_thickness and set_uv first");
    NF_REQUIRE(t >= 0 && t < f->nt, NF_ERR_ARG, "compute: time index out of range");
    const long s_end = f->s_end < 0 ? f->nt * f->nz : f->s_end;
    long lo = t * f->nz, hi = (t + 1) * f->nz;
    if (lo < f->s_begin) lo = f->s_begin;
    if (hi > s_end) hi = s_end;
    const int rowlen = field_row_length(f);
    if (hi <= lo) {  // this rank owns no slab of step t: contributes zeros
        if (row_dev && rowlen > 0) NF_HIP(hipMemsetAsync(row_dev, 0, sizeof(double) * rowlen, f->stream));
        return NF_OK;
    }
    const int z0 = (int)(lo - t * f->nz), z1 = (int)(hi - t * f->nz);
    const size_t es = elem_size(f->uv_dtype);
    const size_t step_bytes = (size_t)f->nz * f->ncell * es;
    const void *ut, *vt;
    if (f->uv_on_device) {
        ut = (const char *)f->u + (size_t)t * step_bytes;
        vt = (const char *)f->v + (size_t)t * step_bytes;
    } else {  // host-resident fields: stage the owned slabs of this step (PCIe-inclusive path)
        if (!f->d_stage_u) {
            NF_HIP(hipMalloc(&f->d_stage_u, step_bytes));
            NF_HIP(hipMalloc(&f->d_stage_v, step_bytes));
        }
        const size_t off = (size_t)z0 * f->ncell * es, len = (size_t)(z1 - z0) * f->ncell * es;
        NF_HIP(hipMemcpyAsync((char *)f->d_stage_u + off, (const char *)f->u + (size_t)t * step_bytes + off, len,
                              hipMemcpyHostToDevice, f->stream));
        NF_HIP(hipMemcpyAsync((char *)f->d_stage_v + off, (const char *)f->v + (size_t)t * step_bytes + off, len,
                              hipMemcpyHostToDevice, f->stream));
        ut = f->d_stage_u;
        vt = f->d_stage_v;
    }
    FluxArgs a{};
    a.u = ut;
    a.v = vt;
    a.dtype = f->uv_dtype;
    a.ncell = f->ncell;
    a.ny = f->ny;
    a.nx = f->nx;
    a.z0 = z0;
    a.z1 = z1;
    a.thickness = f->d_thick;
    a.arcE = f->d_arcE;
    a.arcN = f->d_arcN;
    a.fill = f->fill;
    a.fill2 = f->fill2;
    a.scale = kEarthRadiusSv / 1.e6;  // field.py:226
    a.sverdrup = f->sverdrup;
    a.iV = f->d_iV;
    a.absU = f->d_abs;
    a.absV = f->d_abs + f->ncell;
    a.maxbits = f->d_maxbits;
    // A rank of a sharded run that owns only PART of this step's levels (slab sharding cuts inside a time step) produces
    // partial sums: its south / west copies and |.| planes mean nothing (only a step owned whole has full-field outputs),
    // the transect reduction reads the two signed planes only, so the step runs in the signed-only form and the four derived
    // planes are written on demand (read_step / device_ptr), exactly as in the compact mode.  0.116 ms per such launch at
    // the C4 size; the rows are bit-identical (test_slab_sharding_sums_to_full).  nf_tuning_set("partial_step_planes", 1) keeps the
    // six-plane epilogue on partial steps (the before / after measurement of profiles/r04_rank_emulation.txt).
    const bool partial = (z0 > 0 || z1 < (int)f->nz) && !g_partial_full;
    a.signed_only = (f->compact || partial) && flux_supports_signed_only(a);
    f->derived_stale = a.signed_only != 0;
    if (f->timing) {
        NF_TRY(field_timed_flux(f, a));
    } else {
        NF_TRY(launch_flux(a, f->stream));
    }
    if (row_dev && rowlen > 0) {
        NF_REQUIRE(f->weights_built, NF_ERR_STATE, "compute: build_weights first");
        NF_TRY(launch_integral(f->ws, f->d_iV, f->ncell, 2, f->nx, f->d_tr_off, (int)f->polylines.size(),
                               f->d_scratch, row_dev, f->stream));
        NF_TRY(field_timed_k3_end(f));
    }
    return NF_OK;
}

// All nt steps in FOUR launches (flux kernel with blockIdx.y = step, then the three reduction kernels): small grids
// are launch-bound (4 launches of a few microseconds per step otherwise).  Needs HBM-resident fields and per-step
// planes (nt x 48 B per cell), so it is used while nt*ncell stays under 32 Mi cell-steps ("batch_cellsteps_m").
static long batch_cell_steps() { return g_batch_cellsteps; }

static bool field_can_batch(const nf_field *f)
{
    // launch-bound grids only: from about a million cells on, one launch per step (with the one-field form of the flux
    // kernel for its few wavefronts) is as fast or faster -- 1440 x 1021: 700 vs 733 us per 4-step pass at float32, 1166 vs
    // 1144 at float64; 2160 x 1080: 1035 vs 1187 and 1863 vs 1928 (tools/size_sweep.py, profiles/r04_size_sweep.txt)
    return g_batch_steps && f->uv_on_device && f->nt >= 2 && f->nt < 65536 && f->nt * f->ncell <= batch_cell_steps() &&
           f->ncell <= (1l << 20) && f->weights_built;
}

// every time step of a pass, one after the other.  A rank of a multi-GPU run owns a contiguous range of steps (slab
// sharding): the rows of all the others are zeroed with two memsets instead of one per step (at 8 ranks that is 84 tiny
// launches per pass saved), and only the owned steps are walked.
static int field_all_steps_direct(nf_field *f, double *rows_dev)
{
    const int rowlen = field_row_length(f);
    const long total = f->nt * f->nz;
    const long s_end = f->s_end < 0 ? total : std::min(f->s_end, total);
    const long s_begin = std::min(f->s_begin, s_end);
    const long ta = s_end > s_begin ? s_begin / f->nz : 0;                    // first step with an owned slab
    const long tb = s_end > s_begin ? (s_end - 1) / f->nz + 1 : 0;            // one past the last
    if (rowlen > 0) {
        if (ta > 0) NF_HIP(hipMemsetAsync(rows_dev, 0, sizeof(double) * rowlen * (size_t)ta, f->stream));
        if (tb < f->nt)
            NF_HIP(hipMemsetAsync(rows_dev + (size_t)tb * rowlen, 0, sizeof(double) * rowlen * (size_t)(f->nt - tb),
                                  f->stream));
    }
    for (long t = ta; t < tb; ++t) NF_TRY(field_step_async(f, t, rows_dev + (size_t)t * rowlen));
    return NF_OK;
}

static int field_all_steps_batched(nf_field *f, double *rows_dev)
{
    const int rowlen = field_row_length(f);
    const size_t n = (size_t)f->ncell;
    if (f->batch_steps != f->nt) {
        dev_free(f->d_iVb);
        dev_free(f->d_absb);
        dev_free(f->d_zr);
        NF_TRY(dev_alloc(&f->d_iVb, n * 4 * f->nt));
        NF_TRY(dev_alloc(&f->d_absb, n * 2 * f->nt));
        NF_TRY(dev_alloc(&f->d_zr, (size_t)2 * f->nt));
        // south slots of row 0 are never written (field.py:219): every step's planes start as zeros
        NF_HIP(hipMemsetAsync(f->d_iVb, 0, sizeof(double) * n * 4 * f->nt, f->stream));
        f->batch_steps = f->nt;
        f->batch_version = -1;
    }
    if (f->batch_version != f->version) {  // scratch follows the weight set, z ranges follow the slab ownership
        dev_free(f->d_scratchb);
        NF_TRY(dev_alloc(&f->d_scratchb, (size_t)std::max(f->ws.nrec, f->ws.nent) * f->nt));
        std::vector<int> zr((size_t)2 * f->nt);
        const long s_end = f->s_end < 0 ? f->nt * f->nz : f->s_end;
        for (long t = 0; t < f->nt; ++t) {
            long lo = t * f->nz, hi = (t + 1) * f->nz;
            if (lo < f->s_begin) lo = f->s_begin;
            if (hi > s_end) hi = s_end;
            if (hi < lo) hi = lo;
            zr[2 * t] = (int)(lo - t * f->nz);
            zr[2 * t + 1] = (int)(hi - t * f->nz);
        }
        NF_HIP(hipMemcpy(f->d_zr, zr.data(), sizeof(int) * zr.size(), hipMemcpyHostToDevice));
        f->batch_version = f->version;
    }
    FluxArgs a{};
    a.u = f->u;
    a.v = f->v;
    a.dtype = f->uv_dtype;
    a.ncell = f->ncell;
    a.ny = f->ny;
    a.nx = f->nx;
    a.z0 = 0;
    a.z1 = (int)f->nz;
    a.thickness = f->d_thick;
    a.arcE = f->d_arcE;
    a.arcN = f->d_arcN;
    a.fill = f->fill;
    a.fill2 = f->fill2;
    a.scale = kEarthRadiusSv / 1.e6;
    a.sverdrup = f->sverdrup;
    a.iV = f->d_iVb;
    a.absU = f->d_absb;
    a.absV = f->d_absb + f->ncell;
    a.maxbits = f->d_maxbits;
    a.batch.nsteps = (int)f->nt;
    a.batch.in_stride = f->nz * f->ncell;
    a.batch.zr = f->d_zr;
    if (f->timing) {
        NF_TRY(field_timed_flux(f, a));
    } else {
        NF_TRY(launch_flux(a, f->stream));
    }
    if (rowlen > 0) {
        NF_TRY(launch_integral(f->ws, f->d_iVb, f->ncell, 2, f->nx, f->d_tr_off, (int)f->polylines.size(), f->d_scratchb,
                               rows_dev, f->stream, (int)f->nt, (long)(4 * n), rowlen));
        NF_TRY(field_timed_k3_end(f));
    }
    // the resident single-step arrays keep their meaning: they hold the LAST step (what read_step returns)
    NF_HIP(hipMemcpyAsync(f->d_iV, f->d_iVb + (size_t)(f->nt - 1) * 4 * n, sizeof(double) * 4 * n,
                          hipMemcpyDeviceToDevice, f->stream));
    NF_HIP(hipMemcpyAsync(f->d_abs, f->d_absb + (size_t)(f->nt - 1) * 2 * n, sizeof(double) * 2 * n,
                          hipMemcpyDeviceToDevice, f->stream));
    f->derived_stale = false;
    return NF_OK;
}

extern "C" {

int nf_field_new(nf_field **self)
try {
    NF_REQUIRE(self, NF_ERR_ARG, "nf_field_new: null argument");
    *self = new nf_field();
    (*self)->grid_view.owns_xy = false;
    return NF_OK;
}
NF_API_CATCH
int nf_field_del(nf_field **self)
try {
    if (self && *self) {
        nf_field *f = *self;
        field_free_geometry(f);
        dev_free(f->d_thick);
        if (f->d_stage_u) (void)hipFree(f->d_stage_u);
        if (f->d_stage_v) (void)hipFree(f->d_stage_v);
        f->ws.release();
        dev_free(f->d_tr_off);
        dev_free(f->d_scratch);
        dev_free(f->d_row);
        field_drop_events(f);
        field_drop_graph(f);
        delete f;
        *self = nullptr;
    }
    return NF_OK;
}
NF_API_CATCH
int nf_field_set_stream(nf_field **self, void *hip_stream)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_set_stream: null field");
    (*self)->stream = (hipStream_t)hip_stream;
    ++(*self)->version;
    return NF_OK;
}
NF_API_CATCH

int nf_field_set_bounds(nf_field **self, const void *bounds_lon, const void *bounds_lat, long ny, long nx,
                        int dtype, int on_device)
try {
    NF_REQUIRE(self && *self && bounds_lon && bounds_lat, NF_ERR_ARG, "nf_field_set_bounds: null argument");
    NF_REQUIRE(ny > 0 && nx > 0 && ny * nx < (1l << 31), NF_ERR_ARG, "nf_field_set_bounds: bad (ny, nx)");
    NF_REQUIRE(dtype == NF_F64 || dtype == NF_F32, NF_ERR_ARG, "nf_field_set_bounds: dtype must be NF_F64/NF_F32");
    NF_NEED_DEVICE();
    nf_field *f = *self;
    field_free_geometry(f);
    f->ny = ny;
    f->nx = nx;
    f->ncell = ny * nx;
    const size_t n = (size_t)f->ncell;
    NF_TRY(dev_alloc(&f->d_xy, n * 8));
    NF_TRY(dev_alloc(&f->d_arc4, n * 4));
    NF_TRY(dev_alloc(&f->d_arcE, n));
    NF_TRY(dev_alloc(&f->d_arcN, n));
    NF_TRY(dev_alloc(&f->d_box, 4));
    NF_TRY(dev_alloc(&f->d_iV, n * 4));
    NF_TRY(dev_alloc(&f->d_abs, n * 2));
    NF_TRY(dev_alloc(&f->d_maxbits, 1));
    // field.py:59-63: the per-step arrays start as zeros (row 0's south slot stays zero for ever)
    NF_HIP(hipMemsetAsync(f->d_iV, 0, sizeof(double) * n * 4, f->stream));
    NF_HIP(hipMemsetAsync(f->d_abs, 0, sizeof(double) * n * 2, f->stream));
    NF_HIP(hipMemsetAsync(f->d_maxbits, 0, sizeof(unsigned long long), f->stream));
    const size_t bytes = n * 4 * elem_size(dtype);
    DevTmp d_lon, d_lat;   // hipFree waits for the stream's work before releasing
    const void *plon = bounds_lon, *plat = bounds_lat;
    if (!on_device) {
        NF_TRY(d_lon.alloc(bytes));
        NF_TRY(d_lat.alloc(bytes));
        NF_HIP(hipMemcpyAsync(d_lon.p, bounds_lon, bytes, hipMemcpyHostToDevice, f->stream));
        NF_HIP(hipMemcpyAsync(d_lat.p, bounds_lat, bytes, hipMemcpyHostToDevice, f->stream));
        plon = d_lon.p;
        plat = d_lat.p;
    }
    NF_TRY(launch_geometry(plon, plat, dtype, f->ncell, f->d_xy, f->d_arc4, f->d_arcE, f->d_arcN, f->d_box, f->stream));
    unsigned long long keys[4] = {0, 0, 0, 0};
    NF_HIP(hipMemcpyAsync(keys, f->d_box, sizeof keys, hipMemcpyDeviceToHost, f->stream));
    NF_HIP(hipStreamSynchronize(f->stream));
    for (int k = 0; k < 4; ++k) f->box[k] = box_key_to_double(keys[k]);
    f->grid_view.ncell = f->ncell;
    f->grid_view.d_xy = f->d_xy;
    f->grid_view.row_length = f->nx;  // (field_free_geometry above dropped the locator of the old corner table)
    ++f->grid_view.version;
    f->grid_view.owns_xy = false;
    f->weights_built = false;
    ++f->version;
    return NF_OK;
}
NF_API_CATCH

int nf_field_set_thickness(nf_field **self, const double *thickness, long nz)
try {
    NF_REQUIRE(self && *self && thickness, NF_ERR_ARG, "nf_field_set_thickness: null argument");
    NF_REQUIRE(nz > 0 && nz < (1l << 30), NF_ERR_ARG, "nf_field_set_thickness: bad nz");
    NF_NEED_DEVICE();
    nf_field *f = *self;
    dev_free(f->d_thick);
    NF_TRY(dev_alloc(&f->d_thick, (size_t)nz));
    NF_HIP(hipMemcpy(f->d_thick, thickness, sizeof(double) * nz, hipMemcpyHostToDevice));
    f->nz = nz;
    ++f->version;
    return NF_OK;
}
NF_API_CATCH

int nf_field_set_uv(nf_field **self, const void *u, const void *v, long nt, int dtype, int on_device,
                    double fill_value)
try {
    NF_REQUIRE(self && *self && u && v, NF_ERR_ARG, "nf_field_set_uv: null argument");
    NF_REQUIRE(nt > 0, NF_ERR_ARG, "nf_field_set_uv: nt must be positive");
    NF_REQUIRE(dtype == NF_F64 || dtype == NF_F32, NF_ERR_ARG, "nf_field_set_uv: dtype must be NF_F64/NF_F32");
    nf_field *f = *self;
    if (f->uv_dtype != dtype) {  // the staging slabs of host-resident fields are sized by the dtype
        if (f->d_stage_u) (void)hipFree(f->d_stage_u);
        if (f->d_stage_v) (void)hipFree(f->d_stage_v);
        f->d_stage_u = f->d_stage_v = nullptr;
    }
    f->u = u;
    f->v = v;
    f->nt = nt;
    f->uv_dtype = dtype;
    f->uv_on_device = on_device;
    f->fill = fill_value;
    ++f->version;
    return NF_OK;
}
NF_API_CATCH

int nf_field_set_missing_value(nf_field **self, double missing_value)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_set_missing_value: null field");
    (*self)->fill2 = missing_value;
    ++(*self)->version;
    return NF_OK;
}
NF_API_CATCH

int nf_field_set_sverdrup(nf_field **self, int sverdrup)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_set_sverdrup: null field");
    (*self)->sverdrup = sverdrup ? 1 : 0;
    ++(*self)->version;
    return NF_OK;
}
NF_API_CATCH

int nf_field_set_compact(nf_field **self, int compact)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_set_compact: null field");
    nf_field *f = *self;
    if (!compact && f->d_iV) NF_TRY(field_ensure_derived(f));   // leaving the mode: the planes become whole again
    f->compact = compact ? 1 : 0;
    ++f->version;
    return NF_OK;
}
NF_API_CATCH

int nf_field_set_slab_range(nf_field **self, long s_begin, long s_end)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_set_slab_range: null field");
    NF_REQUIRE(s_begin >= 0 && s_end >= s_begin, NF_ERR_ARG, "nf_field_set_slab_range: need 0 <= begin <= end");
    (*self)->s_begin = s_begin;
    (*self)->s_end = s_end;
    ++(*self)->version;
    return NF_OK;
}
NF_API_CATCH

int nf_field_add_transect(nf_field **self, const double *xyz, int npts, int counterclock, int *transect_id)
try {
    NF_REQUIRE(self && *self && xyz, NF_ERR_ARG, "nf_field_add_transect: null argument");
    NF_REQUIRE(npts >= 2, NF_ERR_ARG, "nf_field_add_transect: need at least 2 points");
    nf_field *f = *self;
    f->polylines.emplace_back(xyz, xyz + 3 * (size_t)npts);
    f->poly_cc.push_back(counterclock ? 1 : 0);
    f->weights_built = false;
    if (transect_id) *transect_id = (int)f->polylines.size() - 1;
    return NF_OK;
}
NF_API_CATCH

int nf_field_set_unsupported_cells(nf_field **self, int skip)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_set_unsupported_cells: null field");
    NF_REQUIRE(skip == 0 || skip == 1, NF_ERR_ARG, "nf_field_set_unsupported_cells: policy must be 0 (refuse) or 1 (skip)");
    (*self)->skip_unsupported = skip;
    return NF_OK;
}
NF_API_CATCH

int nf_field_num_dropped_crossings(nf_field **self, size_t *n)
try {
    NF_REQUIRE(self && *self && n, NF_ERR_ARG, "nf_field_num_dropped_crossings: null argument");
    *n = (size_t)(*self)->ws.dropped;
    return NF_OK;
}
NF_API_CATCH
int nf_field_set_overlapping_cells(nf_field **self, int warn)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_set_overlapping_cells: null field");
    NF_REQUIRE(warn == 0 || warn == 1, NF_ERR_ARG, "nf_field_set_overlapping_cells: policy must be 0 (refuse) or 1 (warn)");
    (*self)->overlap_warn = warn;
    return NF_OK;
}
NF_API_CATCH

int nf_field_build_weights(nf_field **self, int numCellsPerBucket, double periodX)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_build_weights: null field");
    nf_field *f = *self;
    NF_REQUIRE(f->d_xy, NF_ERR_STATE, "nf_field_build_weights: set_bounds first");
    NF_REQUIRE(numCellsPerBucket > 0 && periodX >= 0.0, NF_ERR_ARG, "nf_field_build_weights: bad locator arguments");
    NF_NEED_DEVICE();
    std::vector<double> segs;
    std::vector<int> cc;
    f->tr_off.assign(1, 0);
    for (size_t p = 0; p < f->polylines.size(); ++p) {
        polyline_segments(f->polylines[p].data(), (int)(f->polylines[p].size() / 3), f->poly_cc[p], segs, cc);
        f->tr_off.push_back((int)cc.size());
    }
    f->weights_built = false;
    const int bw = build_weights(f->d_xy, f->ncell, segs.data(), cc.data(), (int)cc.size(), periodX, &f->ws, f->stream,
                                 f->skip_unsupported, f->overlap_warn, nullptr, f->nx);
    if (bw != NF_OK) {
        // over-covered segment (overlapping cells): name the transect and its own segment index, not the batch's
        for (size_t q = 0; q < f->ws.coverage.size(); ++q)
            if ((long)q == (long)f->ws.over_seg) {
                size_t p = 0;
                while (p + 2 < f->tr_off.size() && (size_t)f->tr_off[p + 1] <= q) ++p;
                char buf[320];
                snprintf(buf, sizeof buf,
                         "nf_field_build_weights: transect %zu, target segment %zu is covered %.9g times by the cells of the "
                         "grid: cells overlap along it (a cell wrapped across the date line with periodX = 0, or duplicated "
                         "/ folded cells that are not identical), so part of the line would be counted twice",
                         p, q - (size_t)f->tr_off[p], f->ws.coverage[q]);
                set_error(buf);
                break;
            }
        return bw;
    }
    // the engine reduces its own planes: fold the (cell, edge) weights onto the unique edges of (eU, eV) (field.py:219-223)
    // (only on request -- nf_tuning_set("edge_weights", 1) -- because the records measure faster: see nf_integral.hip)
    if (integral_uses_edges()) NF_TRY(fold_weights(&f->ws, f->ncell, f->nx, f->stream));
    dev_free(f->d_tr_off);
    dev_free(f->d_scratch);
    dev_free(f->d_row);
    NF_TRY(dev_alloc(&f->d_tr_off, f->tr_off.size()));
    NF_TRY(dev_alloc(&f->d_scratch, (size_t)std::max(f->ws.nrec, f->ws.nent)));
    NF_TRY(dev_alloc(&f->d_row, (size_t)field_row_length(f)));
    NF_HIP(hipMemcpy(f->d_tr_off, f->tr_off.data(), sizeof(int) * f->tr_off.size(), hipMemcpyHostToDevice));
    f->weights_built = true;
    ++f->version;
    return NF_OK;
}
NF_API_CATCH

int nf_field_num_transects(nf_field **self, int *n)
try {
    NF_REQUIRE(self && *self && n, NF_ERR_ARG, "nf_field_num_transects: null argument");
    *n = (int)(*self)->polylines.size();
    return NF_OK;
}
NF_API_CATCH
int nf_field_num_segments(nf_field **self, int *nseg_total)
try {
    NF_REQUIRE(self && *self && nseg_total, NF_ERR_ARG, "nf_field_num_segments: null argument");
    NF_REQUIRE((*self)->weights_built, NF_ERR_STATE, "nf_field_num_segments: build_weights first");
    *nseg_total = (*self)->ws.nseg;
    return NF_OK;
}
NF_API_CATCH
int nf_field_segment_offsets(nf_field **self, int *offsets)
try {
    NF_REQUIRE(self && *self && offsets, NF_ERR_ARG, "nf_field_segment_offsets: null argument");
    NF_REQUIRE((*self)->weights_built, NF_ERR_STATE, "nf_field_segment_offsets: build_weights first");
    memcpy(offsets, (*self)->tr_off.data(), sizeof(int) * (*self)->tr_off.size());
    return NF_OK;
}
NF_API_CATCH
int nf_field_num_weights(nf_field **self, size_t *n)
try {
    NF_REQUIRE(self && *self && n, NF_ERR_ARG, "nf_field_num_weights: null argument");
    *n = (size_t)(*self)->ws.entries();
    return NF_OK;
}
NF_API_CATCH
int nf_field_get_weights(nf_field **self, int64_t *cell_edge, double *weight, int *seg_global)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_get_weights: null field");
    if ((*self)->ws.nrec == 0) return NF_OK;
    NF_NEED_DEVICE();
    return weights_to_host((*self)->ws, cell_edge, weight, seg_global);
}
NF_API_CATCH
int nf_field_get_coverage(nf_field **self, double *coverage)
try {
    NF_REQUIRE(self && *self && coverage, NF_ERR_ARG, "nf_field_get_coverage: null argument");
    NF_REQUIRE((*self)->weights_built || !(*self)->ws.coverage.empty(), NF_ERR_STATE,
               "nf_field_get_coverage: build_weights first");
    const std::vector<double> &c = (*self)->ws.coverage;
    if (!c.empty()) memcpy(coverage, c.data(), sizeof(double) * c.size());
    return NF_OK;
}
NF_API_CATCH
int nf_field_num_edge_weights(nf_field **self, size_t *n)
try {
    NF_REQUIRE(self && *self && n, NF_ERR_ARG, "nf_field_num_edge_weights: null argument");
    *n = (size_t)(*self)->ws.nent;
    return NF_OK;
}
NF_API_CATCH
int nf_field_get_edge_weights(nf_field **self, int *elem, int *seg_global, double *weight)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_get_edge_weights: null field");
    const WeightSet &ws = (*self)->ws;
    if (ws.nent == 0) return NF_OK;
    NF_NEED_DEVICE();
    std::vector<WeightSet::EdgeEntry> h((size_t)ws.nent);
    NF_HIP(hipMemcpy(h.data(), ws.ent, sizeof(WeightSet::EdgeEntry) * h.size(), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < h.size(); ++k) {   // pure re-indexing of the device result for the caller
        if (elem) elem[k] = h[k].elem;
        if (seg_global) seg_global[k] = h[k].seg;
        if (weight) weight[k] = h[k].w;
    }
    return NF_OK;
}
NF_API_CATCH
int nf_field_row_length(nf_field **self, int *n)
try {
    NF_REQUIRE(self && *self && n, NF_ERR_ARG, "nf_field_row_length: null argument");
    *n = field_row_length(*self);
    return NF_OK;
}
NF_API_CATCH

int nf_field_compute_flux(nf_field **self, long tIndex, double *row_host)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_compute_flux: null field");
    NF_NEED_DEVICE();
    nf_field *f = *self;
    const int rowlen = field_row_length(f);
    NF_TRY(field_step_async(f, tIndex, (f->weights_built && rowlen > 0) ? f->d_row : nullptr));
    if (row_host && rowlen > 0 && f->weights_built)
        NF_HIP(hipMemcpyAsync(row_host, f->d_row, sizeof(double) * rowlen, hipMemcpyDeviceToHost, f->stream));
    NF_HIP(hipStreamSynchronize(f->stream));
    return NF_OK;
}
NF_API_CATCH

int nf_field_compute_all_async(nf_field **self, double *rows_dev)
try {
    NF_REQUIRE(self && *self && rows_dev, NF_ERR_ARG, "nf_field_compute_all_async: null argument");
    NF_NEED_DEVICE();
    nf_field *f = *self;
    NF_REQUIRE(f->weights_built, NF_ERR_STATE, "nf_field_compute_all_async: build_weights first");
    if (field_can_batch(f)) return field_all_steps_batched(f, rows_dev);
    // Replay a captured graph of the whole pass when nothing changed since it was captured.  Capture needs a real
    // (non-null) stream, resident fields, and no per-launch timing events.
    const bool can_graph = g_use_graph && f->stream != nullptr && f->uv_on_device && !f->timing;
    if (can_graph && f->graph_exec && f->graph_rows == rows_dev && f->graph_version == f->version + tuning_version()) {
        NF_HIP(hipGraphLaunch(f->graph_exec, f->stream));
        return NF_OK;
    }
    if (can_graph) {
        field_drop_graph(f);
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(f->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int rc = field_all_steps_direct(f, rows_dev);
            hipError_t e = hipStreamEndCapture(f->stream, &graph);
            if (rc == NF_OK && e == hipSuccess && graph &&
                hipGraphInstantiate(&f->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                (void)hipGraphDestroy(graph);
                f->graph_rows = rows_dev;
                f->graph_version = f->version + tuning_version();
                NF_HIP(hipGraphLaunch(f->graph_exec, f->stream));
                return NF_OK;
            }
            if (graph) (void)hipGraphDestroy(graph);
            f->graph_exec = nullptr;
            (void)hipGetLastError();  // fall through to direct launches (they report any real error)
        } else {
            (void)hipGetLastError();
        }
    }
    return field_all_steps_direct(f, rows_dev);
}
NF_API_CATCH

int nf_field_read_step(nf_field **self, double *iV_host, double *eU_host, double *eV_host, double *max_abs)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_read_step: null field");
    nf_field *f = *self;
    NF_REQUIRE(f->d_iV, NF_ERR_STATE, "nf_field_read_step: set_bounds first");
    NF_NEED_DEVICE();
    if (iV_host || eU_host || eV_host) NF_TRY(field_ensure_derived(f));
    const size_t n = (size_t)f->ncell;
    if (iV_host) {  // re-pack the planes into the reference's (ncell,4) layout, then one D2H into the caller's array
        if (!f->d_aos) NF_TRY(dev_alloc(&f->d_aos, n * 4));
        NF_TRY(launch_planes_to_aos(f->d_iV, f->ncell, f->d_aos, f->stream));
        NF_HIP(hipMemcpyAsync(iV_host, f->d_aos, sizeof(double) * n * 4, hipMemcpyDeviceToHost, f->stream));
    }
    if (eU_host) NF_HIP(hipMemcpyAsync(eU_host, f->d_abs, sizeof(double) * n, hipMemcpyDeviceToHost, f->stream));
    if (eV_host) NF_HIP(hipMemcpyAsync(eV_host, f->d_abs + n, sizeof(double) * n, hipMemcpyDeviceToHost, f->stream));
    if (max_abs) {
        unsigned long long b = 0;
        NF_HIP(hipMemcpyAsync(&b, f->d_maxbits, sizeof b, hipMemcpyDeviceToHost, f->stream));
        NF_HIP(hipStreamSynchronize(f->stream));
        memcpy(max_abs, &b, 8);
    }
    NF_HIP(hipStreamSynchronize(f->stream));
    return NF_OK;
}
NF_API_CATCH

int nf_field_reset_max(nf_field **self)
try {
    NF_REQUIRE(self && *self && (*self)->d_maxbits, NF_ERR_STATE, "nf_field_reset_max: set_bounds first");
    NF_HIP(hipMemsetAsync((*self)->d_maxbits, 0, sizeof(unsigned long long), (*self)->stream));
    return NF_OK;
}
NF_API_CATCH

int nf_field_get_arclengths(nf_field **self, double *arc_host)
try {
    NF_REQUIRE(self && *self && arc_host, NF_ERR_ARG, "nf_field_get_arclengths: null argument");
    NF_REQUIRE((*self)->d_arc4, NF_ERR_STATE, "nf_field_get_arclengths: set_bounds first");
    NF_HIP(hipMemcpy(arc_host, (*self)->d_arc4, sizeof(double) * 4 * (size_t)(*self)->ncell, hipMemcpyDeviceToHost));
    return NF_OK;
}
NF_API_CATCH

int nf_field_get_points(nf_field **self, double *points_host)
try {
    NF_REQUIRE(self && *self && points_host, NF_ERR_ARG, "nf_field_get_points: null argument");
    nf_field *f = *self;
    NF_REQUIRE(f->d_xy, NF_ERR_STATE, "nf_field_get_points: set_bounds first");
    DevTmp points;
    NF_TRY(points.alloc(sizeof(double) * 12 * (size_t)f->ncell));
    NF_TRY(launch_points_from_corner_table(f->d_xy, f->ncell, points.as<double>(), f->stream));
    NF_HIP(hipMemcpyAsync(points_host, points.p, sizeof(double) * 12 * (size_t)f->ncell, hipMemcpyDeviceToHost,
                          f->stream));
    NF_HIP(hipStreamSynchronize(f->stream));
    return NF_OK;
}
NF_API_CATCH

int nf_field_get_box(nf_field **self, double *lonmin, double *lonmax, double *latmin, double *latmax)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_get_box: null field");
    NF_REQUIRE(lonmin && lonmax && latmin && latmax, NF_ERR_ARG, "nf_field_get_box: null argument");
    NF_REQUIRE((*self)->d_xy, NF_ERR_STATE, "nf_field_get_box: set_bounds first");
    *lonmin = (*self)->box[0];
    *lonmax = (*self)->box[1];
    *latmin = (*self)->box[2];
    *latmax = (*self)->box[3];
    return NF_OK;
}
NF_API_CATCH

int nf_field_device_ptr(nf_field **self, int which, void **dev)
try {
    NF_REQUIRE(self && *self && dev, NF_ERR_ARG, "nf_field_device_ptr: null argument");
    nf_field *f = *self;
    if (which >= 0 && which <= 2) NF_TRY(field_ensure_derived(f));   // compact mode: the derived planes on demand
    switch (which) {
        case 0: *dev = f->d_iV; break;
        case 1: *dev = f->d_abs; break;
        case 2: *dev = f->d_abs ? f->d_abs + f->ncell : nullptr; break;
        case 3: *dev = f->d_arc4; break;
        case 4: *dev = f->d_xy; break;
        default: NF_REQUIRE(false, NF_ERR_ARG, "nf_field_device_ptr: unknown array id");
    }
    return NF_OK;
}
NF_API_CATCH

int nf_field_grid(nf_field **self, Grid_t **grid)
try {
    NF_REQUIRE(self && *self && grid, NF_ERR_ARG, "nf_field_grid: null argument");
    NF_REQUIRE((*self)->d_xy, NF_ERR_STATE, "nf_field_grid: set_bounds first");
    *grid = &(*self)->grid_view;
    return NF_OK;
}
NF_API_CATCH

int nf_field_timing(nf_field **self, int enable)
try {
    NF_REQUIRE(self && *self, NF_ERR_ARG, "nf_field_timing: null field");
    nf_field *f = *self;
    if (!enable) field_drop_events(f);
    f->ev_used = 0;
    f->ev_dropped = 0;
    f->timing = enable != 0;
    if (enable > 1) NF_TRY(field_reserve_events(f, (size_t)enable));   // event creation stays out of the timed region
    ++f->version;
    return NF_OK;
}
NF_API_CATCH
int nf_field_timing_read(nf_field **self, long *launches, double *total_ms)
try {
    NF_REQUIRE(self && *self && launches && total_ms, NF_ERR_ARG, "nf_field_timing_read: null argument");
    nf_field *f = *self;
    NF_HIP(hipStreamSynchronize(f->stream));
    double tot = 0.0, flux = 0.0, expand = 0.0, k3 = 0.0;
    for (size_t k = 0; k < f->ev_used; ++k) {
        const auto &t = f->ev[k];
        float ms = 0.f, part = 0.f;
        if (t.has_k3) {
            NF_HIP(hipEventElapsedTime(&part, t.e1, t.e2));
            k3 += part;
            part = 0.f;
        }
        NF_HIP(hipEventElapsedTime(&ms, t.e0, t.e1));
        tot += ms;
        if (t.has_mid) {
            NF_HIP(hipEventElapsedTime(&part, t.e0, t.mid));
            flux += part;
            expand += ms - part;
        } else {
            flux += ms;
        }
    }
    *launches = (long)f->ev_used;
    *total_ms = tot;
    f->last_flux_ms = flux;
    f->last_expand_ms = expand;
    f->last_k3_ms = k3;
    f->ev_used = 0;
    return NF_OK;
}
NF_API_CATCH
int nf_field_timing_split(nf_field **self, double *flux_ms, double *expand_ms)
try {
    NF_REQUIRE(self && *self && flux_ms && expand_ms, NF_ERR_ARG, "nf_field_timing_split: null argument");
    *flux_ms = (*self)->last_flux_ms;
    *expand_ms = (*self)->last_expand_ms;
    return NF_OK;
}
NF_API_CATCH
int nf_field_timing_k3(nf_field **self, double *k3_ms)
try {
    NF_REQUIRE(self && *self && k3_ms, NF_ERR_ARG, "nf_field_timing_k3: null argument");
    *k3_ms = (*self)->last_k3_ms;
    return NF_OK;
}
NF_API_CATCH

}  // extern "C"
