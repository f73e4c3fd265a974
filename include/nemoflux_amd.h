/*
 * nemoflux_amd.h -- C ABI of libnemoflux_amd.so, the MI355X (gfx950) transect-flux engine.
 *
 * This is the drop-in boundary for nemoflux's hot path.  Two levels:
 *
 *   Level 1  mnt_grid_* / mnt_polylineintegral_*  -- the entry points nemoflux reaches in the
 *            un-vendored `mint` C library through its ctypes wrapper (python-mint >= 1.24.4,
 *            /root/reference/README.md:12).  Same names, argument order, handle convention
 *            (opaque object passed as T**) and error convention (int return, 0 = OK) as mint's
 *            C API, so nemoflux/horizgrid.py:23-24,30,43 and nemoflux/field.py:45-48,102 run
 *            unchanged on top of nemoflux_amd/mint.py.
 *   Level 2  nf_field_*  -- the Field-shaped engine (nemoflux/field.py:15-234): geometry set-up,
 *            per-time-step vertical integration + edge flux (the bandwidth-bound kernel), batched
 *            transect weights and the per-segment / per-transect reduction, with U/V resident in HBM
 *            and (t,z) slab ownership for multi-GPU runs.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in any signature (streams travel as void*).
 *   - every function returns 0 on success, non-zero on error; nf_last_error() gives the message.
 *   - "host" pointers are ordinary CPU memory owned by the caller; "dev" pointers are HBM addresses
 *     of the current device (e.g. torch.Tensor.data_ptr() or nf_malloc()).
 *   - all entry points are synchronous at return unless the name ends in _async.
 *   - there is NO CPU fallback: every compute entry point fails with NF_ERR_NO_DEVICE when no
 *     gfx950 device is usable.
 *   - threads: the reference drives mint from one thread (fluxviz.py:20-23,351) and so may a client.  What holds beyond
 *     that: the error text is per thread; one OBJECT (a PolylineIntegral, a VectorInterp, a field, an inflater) is used by
 *     one thread at a time; different objects may be driven from different host threads even when they share a Grid_t --
 *     the grid's locator cache is built, walked and released under the grid's own lock; the weight-build scratch is a
 *     process-wide pool under a lock.  Giving a Grid_t new points (mnt_grid_build) while another thread still integrates
 *     with objects made on it is the caller's race, as with mint.  nf_tuning_set is NOT thread-safe (see there).
 */
#ifndef NEMOFLUX_AMD_H
#define NEMOFLUX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NF_OK 0
#define NF_ERR_ARG 1        /* bad argument / shape (the reference raises RuntimeError: field.py:135,154) */
#define NF_ERR_STATE 2      /* call order violated (e.g. computeWeights before setGrid) */
#define NF_ERR_HIP 3        /* a HIP runtime call failed */
#define NF_ERR_NO_DEVICE 4  /* no usable GPU: the engine never falls back to the CPU */
#define NF_ERR_HOST 5       /* host-side failure (out of host memory, file write error); no C++ exception ever
                               crosses this ABI */

#define NF_F64 0
#define NF_F32 1

/* mint.CELL_BY_CELL_DATA / mint.UNIQUE_EDGE_DATA (field.py:102 uses the former) */
#define MNT_CELL_BY_CELL_DATA 0
#define MNT_UNIQUE_EDGE_DATA 1

/* ------------------------------------------------------------------ library / device plumbing */
const char *nf_last_error(void);
int nf_version(void);
int nf_device_count(int *count);
int nf_set_device(int device);
int nf_device_name(char *buf, int buflen); /* e.g. "gfx950:..." */
int nf_malloc(void **dev, size_t bytes);
int nf_free(void *dev);
int nf_host_alloc(void **host, size_t bytes); /* pinned host memory (fast D2H into numpy views) */
int nf_host_free(void *host);
int nf_memcpy_h2d(void *dev, const void *host, size_t bytes);
int nf_memcpy_d2h(void *host, const void *dev, size_t bytes);
int nf_memset(void *dev, int value, size_t bytes);
int nf_synchronize(void);
/* Weight builds and point searches keep their scratch memory between calls while it is small (a process-wide pool of at most
 * 4 idle scratches of at most 1 GiB of HBM each: a viewer makes one PolylineIntegral per transect and each build would
 * otherwise pay a dozen hipMalloc / hipFree pairs).  A build checks a scratch out of the pool and hands it back, whichever
 * host thread runs it, so threads that end leave nothing behind.  This call frees every idle scratch of the process
 * (the Python package calls it at interpreter exit). */
int nf_release_scratch(void);
/* host-side file decoding helper: undo HDF5's shuffle filter (es byte planes of n elements -> n elements) */
int nf_host_unshuffle(const void *src, void *dst, size_t n, int es);
/* host-side file staging helper: copy n byte ranges (addresses as integers) with nthreads native threads -- the compressed
 * chunks of a group of time steps out of the mapped file into a pinned staging buffer (field.py:149 reads them one by one) */
int nf_host_gather(const unsigned long long *src_addr, const unsigned long long *dst_addr, const long long *len, long long n,
                   int nthreads);
/* tuning knobs for A/B measurements inside one process: "flux_variant" (0 = default store form, 5 = the other one; see nf_flux.hip),
 * "xcd_map" (1 = on), "batch_steps" (1 = small grids run all time steps in one launch), "field_split" (-1 = one-step launches
 * of fewer than 20 000 wavefronts integrate uo and vo in different wavefronts, 0 = never, 1 = always), "edge_weights", "datagen_rows"
 * (0 = the generator's one-cell-per-lane kernel with plain division, the reference of its row kernel), "west_shift" (0 = the
 * west slots of integratedVelocity as 8-byte stores, the form before round 5), "batch_cellsteps_m" (all-steps-in-one-launch
 * limit in Mi cell-steps, 32), "partial_step_planes" (1 = six-plane epilogue on a rank's partial time steps), "graph" (0 =
 * no graph replay of a pass).  The library reads NO environment variable: these calls are the only switches.
 * NOT thread-safe: the knobs are plain process-wide variables read by every later launch of every thread; set them before the
 * objects they affect are used and from one thread only (the reference drives mint from a single thread; A/B tools
 * and the bench's --knob do the same).  A product run never needs to call this. */
int nf_tuning_set(const char *name, int value);

/* ------------------------------------------------------------------ Level 1: mint-shaped API */
typedef struct Grid_t Grid_t;
typedef struct PolylineIntegral_t PolylineIntegral_t;

/* mint.Grid()                                   horizgrid.py:23 */
int mnt_grid_new(Grid_t **self);
int mnt_grid_del(Grid_t **self);
/* mint.Grid.setPoints(points (ncell,4,3) float64)  horizgrid.py:24.  `points` is BORROWED (host) and
 * must outlive the grid, as in mint; the corner (lon,lat) pairs are uploaded to HBM by mnt_grid_build. */
int mnt_grid_setPointsPtr(Grid_t **self, double *points);
int mnt_grid_build(Grid_t **self, int nVertsPerCell, long long ncells);
/* extension (not in mint): the cells handed to setPoints are the rows of a (ny, nx) grid, nx = rowLength -- what
 * horizgrid.py:17-22 flattens.  Only a hint: the weights do not depend on it, the locator groups the cells in 4 x 4 blocks
 * instead of 16 consecutive ones and long target lines are located several times faster.  0 = unknown (default). */
int mnt_grid_setRowLength(Grid_t **self, long long rowLength);
/* mint.Grid.getNumberOfCells()                  horizgrid.py:30 */
int mnt_grid_getNumberOfCells(Grid_t **self, size_t *numCells);
/* mint.Grid.dump(fileName): legacy-VTK unstructured grid  horizgrid.py:43 */
int mnt_grid_dump(Grid_t **self, const char *fileName);

/* mint.PolylineIntegral()                       field.py:45 */
int mnt_polylineintegral_new(PolylineIntegral_t **self);
int mnt_polylineintegral_del(PolylineIntegral_t **self);
/* .setGrid(grid)                                field.py:46 */
int mnt_polylineintegral_setGrid(PolylineIntegral_t **self, Grid_t *grid);
/* .buildLocator(numCellsPerBucket=128, periodX=360., enableFolding=False)   field.py:47
 * The locator is a 16-fold hierarchy of bounding boxes over the cells, built at the first computeWeights / findPoints on the
 * grid and kept by the grid (nf_locator.h); numCellsPerBucket must be positive and has nothing to tune; enableFolding != 0
 * is rejected (nemoflux never enables it).  periodX > 0: target lines are also tried one period to the west and to the east,
 * and every cell's corners are brought to within periodX/2 of its corner 0 before the cell is used (a global file stores
 * bounds_lon wrapped into one period, so the cells on the cut have corners ~355 degrees apart: datagen.py:161-166 is the
 * reference's own form of this rule); the grid's points themselves are not touched. */
int mnt_polylineintegral_buildLocator(PolylineIntegral_t **self, int numCellsPerBucket, double periodX,
                                      int enableFolding);
/* .computeWeights(xyz (npoints,3), counterclock=False)   field.py:48
 * A cell the weights are not defined on -- a quad that is not convex in the (lon,lat) plane, one with a corner AT a
 * geographic pole (the cells around the pole of a rotated grid) or one that CONTAINS a pole -- that a target segment
 * overlaps over a positive length is left out (policy 'skip', the default of this mint-shaped level since round 6: mint's
 * computeWeights has no error path there, field.py:44-49): the rest of the line is integrated,
 * mnt_polylineintegral_getCoverage reports < 1 for the segment and mnt_polylineintegral_getNumberOfDroppedCrossings says
 * how many such crossings were left out -- never a silent number.  With mnt_polylineintegral_setUnsupportedCells(0) the
 * call returns NF_ERR_ARG (with the cell id in nf_last_error) instead.  NF_ERR_ARG naming
 * the segment, too, when some stretch of a target segment lies in two cells that do not hold the same sub-segment (coverage
 * > 1 + 1e-8 AND the excess, as a length, > 1e-9 max(1, |coordinates|) degrees: overlapping cells, e.g. a date-line-wrapped
 * grid with periodX = 0): that stretch would be counted twice.  (The length condition keeps rounding noise on target
 * segments of ~1e-8 degrees and shorter -- near-duplicate vertices on a grid node -- from refusing a whole line.)
 * mnt_polylineintegral_getCoverage still answers after that error.  Cells with a corner that is not a finite number take
 * part in nothing. */
int mnt_polylineintegral_computeWeights(PolylineIntegral_t **self, int npoints, const double xyz[],
                                        int counterclock);
/* extension (not in mint): what computeWeights does with such a cell.  skip = 1 (default): the cell contributes nothing --
 * the rest of the line is integrated and mnt_polylineintegral_getCoverage reports the fraction of every target segment
 * that was (real ORCA grids with a few distorted polar cells far from the transect's physics; mint itself returns a number
 * there, pinned by nothing in the reference).  skip = 0: NF_ERR_ARG naming the cell. */
int mnt_polylineintegral_setUnsupportedCells(PolylineIntegral_t **self, int skip);
/* extension: (cell, target-segment image) crossings of unsupported cells the last computeWeights left out (0 unless the
 * policy is 'skip' and the line meets such cells) */
int mnt_polylineintegral_getNumberOfDroppedCrossings(PolylineIntegral_t **self, size_t *n);
/* extension (not in mint): what computeWeights does with a target segment that is covered more than once (see above).
 * warn = 0 (default): NF_ERR_ARG.  warn = 1: the weights are built as mint would build them -- the stretch counts twice --
 * and mnt_polylineintegral_getCoverage reports > 1 for the segment (the Python wrapper warns). */
int mnt_polylineintegral_setOverlappingCells(PolylineIntegral_t **self, int warn);
/* .getIntegral(data (ncell,4) float64 HOST, placement) -> *result   field.py:102, fluxplot.py:56
 * Host data is staged SPARSELY: only the 32 bytes of every cell the weights touch are gathered (pinned buffer) and sent
 * to HBM -- the cost follows the number of weights, not the size of the grid, like mint's own sparse dot -- and the same
 * kernels run on them: the result has the bits of ...getIntegralDev on the whole array (see that for resident data). */
int mnt_polylineintegral_getIntegral(PolylineIntegral_t **self, const double data[], int placement,
                                     double *result);
/* extensions (not in mint): device-resident data, per-target-segment sums, weight read-back */
int mnt_polylineintegral_getIntegralDev(PolylineIntegral_t **self, const double *data_dev, int placement,
                                        double *result, double *seg_totals_host /* nseg or NULL */);
/* coverage[s] = fraction of target segment s that lies inside cells of the grid (1 = inside, counted once; < 1 = part of it
 * lies in no cell and contributes nothing; > 1 is refused by computeWeights); see nf_field_get_coverage.  npoints-1 values. */
int mnt_polylineintegral_getCoverage(PolylineIntegral_t **self, double *coverage);
int mnt_polylineintegral_getNumberOfWeights(PolylineIntegral_t **self, size_t *n);
int mnt_polylineintegral_getWeights(PolylineIntegral_t **self, int64_t *cell_edge, double *weight, int *seg);

/* mint.VectorInterp: the arrows on the target line                field.py:90-95, 119-120 */
typedef struct VectorInterp_t VectorInterp_t;
int mnt_vectorinterp_new(VectorInterp_t **self);
int mnt_vectorinterp_del(VectorInterp_t **self);
/* .setGrid(grid)                                field.py:91 */
int mnt_vectorinterp_setGrid(VectorInterp_t **self, Grid_t *grid);
/* .buildLocator(numCellsPerBucket=128, periodX=360.)   field.py:92 */
int mnt_vectorinterp_buildLocator(VectorInterp_t **self, int numCellsPerBucket, double periodX, int enableFolding);
/* .findPoints(targetPoints (n,3) host, tol2=1.e-12)    field.py:93; *numNotFound (may be NULL) counts points outside */
int mnt_vectorinterp_findPoints(VectorInterp_t **self, size_t numPoints, const double targetPoints[], double tol2,
                                size_t *numNotFound);
/* .getFaceVectors(data (ncell,4) host, placement=0) -> vectors (n,3) host   field.py:94-95,119
 * Only the rows of the located cells travel to HBM (n x 32 B); same bits as ...getFaceVectorsDev on the whole array. */
int mnt_vectorinterp_getFaceVectors(VectorInterp_t **self, const double data[], int placement, double vectors[]);
/* extensions: data resident in HBM (layout 0 = (ncell,4), 1 = the engine's [4][ncell] planes); located cells */
int mnt_vectorinterp_getFaceVectorsDev(VectorInterp_t **self, const double *data_dev, int layout, double vectors[]);
int mnt_vectorinterp_getCells(VectorInterp_t **self, long long *cell_ids, double *pcoords /* (n,2) */);

/* ------------------------------------------------------------------ Level 2: Field-shaped engine */
typedef struct nf_field nf_field;

int nf_field_new(nf_field **self);
int nf_field_del(nf_field **self);
/* HIP stream all of this field's kernels and copies are issued on (NULL = the null stream, which is
 * also torch's default current stream). */
int nf_field_set_stream(nf_field **self, void *hip_stream);
/* Cell bounds (field.py:22-24, horizgrid.py:12-24): bounds_lon/lat (ny,nx,4), dtype NF_F64 (datagen) or
 * NF_F32 (real NEMO), host (on_device=0) or HBM (1).  Runs the geometry kernel: corner table, great-circle
 * edge lengths (field.py:170-181), lon/lat box (field.py:27-30). */
int nf_field_set_bounds(nf_field **self, const void *bounds_lon, const void *bounds_lat, long ny, long nx,
                        int dtype, int on_device);
/* Layer thickness = deptht_bounds[:,1]-deptht_bounds[:,0] (field.py:51), host, nz values. */
int nf_field_set_thickness(nf_field **self, const double *thickness, long nz);
/* uo/vo (nt,nz,ny,nx) x-fastest (field.py:34-35,122-136).  on_device=1: HBM pointers, used in place (no
 * copy).  on_device=0: host arrays, one time step is staged through HBM per compute call.  fill_value:
 * the variable's _FillValue (datagen.py:191,204: 1e20) -- NaN and fill_value both count as missing ->
 * 0 (field.py:157); pass NaN for "no _FillValue". */
int nf_field_set_uv(nf_field **self, const void *u, const void *v, long nt, int dtype, int on_device,
                    double fill_value);
/* A second value that counts as missing -> 0: the CF attribute missing_value when it differs from _FillValue (xarray's
 * decode_cf, which the reference relies on at field.py:34-35, masks both).  Compared in the fields' dtype; NaN = none. */
int nf_field_set_missing_value(nf_field **self, double missing_value);
/* field.py:19,225-228: scale fluxes by 6371000/1e6 */
int nf_field_set_sverdrup(nf_field **self, int sverdrup);
/* Compact resident mode (default off).  The reference stores, per time step, the (ncell,4) array whose slots 0 and 3
 * are copies of the neighbours' slots 2 and 1 (field.py:209-223) and the two |.| arrays (field.py:231-232).  With
 * compact != 0 the flux kernel keeps only the two signed arrays (eU, eV) resident -- all the transect reduction reads --
 * and the other four are derived, bit-identically, when nf_field_read_step / nf_field_device_ptr ask for them:
 * 104 instead of 311 MB of stores per step on the ORCA12-like grid.  Batch drivers (fluxplot.py:51-59) never ask. */
int nf_field_set_compact(nf_field **self, int compact);

/* Multi-GPU ownership: this rank integrates the flattened slabs s = t*nz + z in [s_begin, s_end)
 * (SURVEY.md section 8e).  Default: all of them. */
int nf_field_set_slab_range(nf_field **self, long s_begin, long s_end);
/* Transects (field.py:43-49): add polylines, then build all weights in one batched pass.
 * xyz: (npts,3) host.  *transect_id receives the index. */
int nf_field_add_transect(nf_field **self, const double *xyz, int npts, int counterclock, int *transect_id);
/* same policy switch as mnt_polylineintegral_setUnsupportedCells, for the batched build below; call before build_weights.
 * The default of THIS level is 0 (refuse): a batch driver should hear about a transect it cannot integrate in full. */
int nf_field_set_unsupported_cells(nf_field **self, int skip);
/* crossings of unsupported cells the last nf_field_build_weights left out (policy 'skip'), over all transects */
int nf_field_num_dropped_crossings(nf_field **self, size_t *n);
/* same policy switch as mnt_polylineintegral_setOverlappingCells, for the batched build below; call before build_weights */
int nf_field_set_overlapping_cells(nf_field **self, int warn);
int nf_field_build_weights(nf_field **self, int numCellsPerBucket, double periodX);
int nf_field_num_transects(nf_field **self, int *n);
int nf_field_num_segments(nf_field **self, int *nseg_total);             /* over all transects */
int nf_field_segment_offsets(nf_field **self, int *offsets /* ntransect+1 */);
int nf_field_num_weights(nf_field **self, size_t *n);
int nf_field_get_weights(nf_field **self, int64_t *cell_edge, double *weight, int *seg_global);
/* The same weights as the engine's own reduction uses them: folded onto the unique edges of the resident signed planes.
 * The south / west slots of integratedVelocity are copies of the neighbours' north / east values (field.py:219-223, row 0's
 * south slot is never written), so each (cell, edge) weight belongs to one element of [eU | eV] (elem in [0, 2*ncell):
 * eU[c] = c, eV[c] = ncell + c) and the weights that meet on an element are summed per target segment.  Sorted by
 * (segment, elem).  mint has no counterpart; nf_field_get_weights above stays the mint-shaped view. */
/* Fraction of every target segment (nseg_total values, transect after transect) that lies inside cells of the grid: 1 =
 * inside, each point counted once; less = part of the segment is outside the grid and contributes nothing (mint warns
 * when its own sum of coefficient * (tb - ta) is not 1 [recall]; here the caller can look). */
int nf_field_get_coverage(nf_field **self, double *coverage);
int nf_field_num_edge_weights(nf_field **self, size_t *n);
int nf_field_get_edge_weights(nf_field **self, int *elem, int *seg_global, double *weight);
/* Length of one output row: nseg_total + ntransect doubles = [per-segment sums | per-transect sums]. */
int nf_field_row_length(nf_field **self, int *n);

/* One time step (Field.update + getFluxText's integrals: field.py:112-120,98-103): vertical integral of
 * this rank's slabs of step tIndex, edge fluxes into the resident integratedVelocity, then the transect
 * reduction.  row_host (row_length doubles) may be NULL. */
int nf_field_compute_flux(nf_field **self, long tIndex, double *row_host);
/* All nt steps back to back, asynchronously on the field's stream; rows_dev: HBM (nt, row_length),
 * fully overwritten (zeros where this rank owns no slab).  This is the timed "step" of bench.py. */
int nf_field_compute_all_async(nf_field **self, double *rows_dev);
/* Read-back of the resident per-step arrays into caller-owned HOST arrays, in place (fluxviz.py aliases
 * them: fluxviz.py:148,160,168): integratedVelocity (ncell,4), edgeFluxesU/V (ncell) = |flux|; any may be
 * NULL.  max_abs: running max (field.py:234). */
int nf_field_read_step(nf_field **self, double *iV_host, double *eU_host, double *eV_host, double *max_abs);
int nf_field_reset_max(nf_field **self);
int nf_field_get_arclengths(nf_field **self, double *arc_host /* (ncell,4) */);
int nf_field_get_points(nf_field **self, double *points_host /* (ncell,4,3) */);
int nf_field_get_box(nf_field **self, double *lonmin, double *lonmax, double *latmin, double *latmax);
/* HBM addresses of resident arrays (for zero-copy consumers / RCCL): which = 0 integratedVelocity,
 * 1 |eU|, 2 |eV|, 3 arcLengths (ncell,4), 4 corner table (ncell,4,2) */
int nf_field_device_ptr(nf_field **self, int which, void **dev);
/* A Grid_t view of the field's corner table (so mint.PolylineIntegral objects can share it). */
int nf_field_grid(nf_field **self, Grid_t **grid);
/* Kernel timing with HIP events on the field's stream, around the vertical-integral+edge-flux launches
 * (bench.py's roofline leg): enable, run, then read (launch count, total ms).  enable > 1 also creates that many
 * event triples up front, so that no event is created inside a timed region; the events are re-used after every read
 * (at most 65536 launches are recorded between two reads). */
int nf_field_timing(nf_field **self, int enable);
int nf_field_timing_read(nf_field **self, long *launches, double *total_ms);
/* how the total of the last nf_field_timing_read splits between the flux kernel and the expansion kernel behind it */
int nf_field_timing_split(nf_field **self, double *flux_ms, double *expand_ms);
/* time of the transect reductions (K3: gather + segmented scan + the two finalize kernels) launched behind the timed flux
 * launches of the last nf_field_timing_read -- measured with the same events, not part of its total */
int nf_field_timing_k3(nf_field **self, double *k3_ms);

/* ------------------------------------------------------------------ multi-GPU: the one collective */
/* The reference walks the time steps serially (fluxplot.py:51-59) and contracts z with one tensordot (field.py:161); here
 * every rank integrates its own (t,z) slabs (nf_field_set_slab_range above) into partial rows (nt, row_length) and ONE
 * all-reduce(sum, float64) over RCCL / xGMI gives every rank the totals.  One process per GPU.  librccl is resolved at
 * first use from the copy the process already holds (PyTorch's, or the system's librccl.so.1): no link-time dependency.
 *   nf_rccl_unique_id : rank 0 creates the 128-byte id and hands it to the other ranks by any means (file, socket, MPI,
 *                       torch.distributed store)
 *   nf_rccl_preflight : NOT collective: checks what nf_rccl_comm_init needs that does not involve the other ranks (librccl
 *                       resolves, the calling thread has a usable HIP device; *device = its index).  ncclCommInitRank is
 *                       collective, so a rank that failed before it would leave the others waiting inside it: call this
 *                       on every rank, agree on the outcome over the channel that carried the id, then init together
 *   nf_rccl_comm_init : every rank, after nf_set_device(its GPU); collective over the nranks callers
 *   nf_rows_allreduce : rows_dev (HBM, n doubles) summed in place over all ranks, asynchronous on hip_stream.  rccl_comm may
 *                       also be a ncclComm_t the caller created itself with the same librccl
 *   nf_rccl_comm_info : what the communicator says about itself (number of ranks, this rank, HIP device index)
 *   nf_rccl_library   : path of the librccl the entry points were resolved from */
#define NF_RCCL_UNIQUE_ID_BYTES 128
int nf_rccl_unique_id(void *id128);
int nf_rccl_preflight(int *device);
int nf_rccl_comm_init(void **comm, int nranks, const void *id128, int rank);
int nf_rccl_comm_destroy(void *comm);
int nf_rccl_comm_info(void *comm, int *nranks, int *rank, int *device);
int nf_rccl_library(char *buf, int buflen);
int nf_rows_allreduce(void *rccl_comm, double *rows_dev, size_t n, void *hip_stream);

/* ------------------------------------------------------------------ file ingest straight to HBM (field.py:149) */
/* Real NEMO files are NetCDF-4 = HDF5 with uo / vo stored as byte-shuffled, deflated chunks; the reference has netCDF4 /
 * xarray inflate one time step on the host at every update (field.py:149: nc[name][timeIndex, :, :, :]).  Here the
 * compressed chunks are handed over as they sit in the file and are inflated on the device, one wavefront per chunk
 * (RFC 1950/1951 decoder with the Adler-32 check, then the inverse of HDF5's shuffle filter), into out_dev.
 *   comp_host  : host buffer (pinned for an asynchronous copy) holding the compressed chunks, comp_bytes long; NULL = the
 *                bytes a preceding nf_inflater_upload of the same comp_bytes put in HBM
 *   in_off/in_len[i] : where chunk i's zlib stream sits in comp_host
 *   chunk_bytes      : decoded size of every chunk = cz*cy*cx*elem_size (checked against what each stream inflates to)
 *   elem_size        : 4 or 8 (1 for raw bytes); shuffled != 0: the chunks went through HDF5's shuffle filter
 *   chunk_dims (3)   : (cz, cy, cx) of a chunk; slab_dims (3): (nz, ny, nx) of the time step; origin (nchunks x 3): where
 *                      each chunk starts in the slab -- chunks may tile y and x, edge chunks may hang over the slab
 *   out_dev          : the slab in HBM, nz*ny*nx*elem_size bytes
 *   status_host (nchunks ints or NULL): 0 = fine, else the decoder's error code per chunk
 * Synchronous on hip_stream.  NF_ERR_ARG (message names the first bad chunk) if any stream is malformed -- the slab is
 * then undefined; nothing outside it is ever written. */
typedef struct nf_inflater nf_inflater;
int nf_inflater_new(nf_inflater **self);
int nf_inflater_del(nf_inflater **self);
/* `self` uses owner's decode scratch (the decoded group, job and status arrays) from now on instead of its own: for two
 * inflaters whose nf_inflater_run calls never overlap in time (a run is synchronous) but which each need their own
 * compressed buffer -- the two staging slots of a file-backed field.  owner must outlive self. */
int nf_inflater_share_scratch(nf_inflater **self, nf_inflater **owner);
/* how many chunks the device decodes at once (resident decoder wavefronts): callers batch that many per nf_inflater_run */
int nf_inflater_capacity(int *streams);
/* Early upload: copy comp_bytes of compressed chunks to HBM on the inflater's own stream, complete at return.  Meant for a
 * staging thread that has just gathered the NEXT group while the GPU still decodes this one (the copy then runs under the
 * decode); a following nf_inflater_run with comp_host = NULL and the same comp_bytes decodes what was uploaded. */
int nf_inflater_upload(nf_inflater **self, const void *comp_host, size_t comp_bytes);
/* The same straight from the mapped file, without a staging copy: range i (src_addr[i], len[i] bytes of ordinary host memory)
 * goes to byte dst_off[i] of the compressed buffer of comp_bytes bytes; what lies between the ranges reads as zeros. */
int nf_inflater_upload_ranges(nf_inflater **self, const unsigned long long *src_addr, const long long *dst_off,
                              const long long *len, long long n, size_t comp_bytes);
int nf_inflater_run(nf_inflater **self, const void *comp_host, size_t comp_bytes, const long long *in_off,
                    const long long *in_len, int nchunks, long long chunk_bytes, int elem_size, int shuffled,
                    const long long *chunk_dims, const long long *slab_dims, const long long *origin, void *out_dev,
                    void *hip_stream, int *status_host);

/* ------------------------------------------------------------------ synthetic data (datagen.py) */
/* Stream functions offered on device (no eval on the GPU): psi = g(z,t) * h(x,y)
 *   0 "x"                                                    README.md:26
 *   1 "arctan2(y, x+180)/(2*pi)"                             README.md:50
 *   2 "cos(2*pi*y/360) + sin(2*pi*x/360)"                    README.md:65
 *   3 "(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))"   README.md:89
 *   4 "(cos(t*2*pi/nt)+2)*(0.5*(y/180)**2 + sin(2*pi*x/360))" datagen.py:211 (default)
 *   5 "(1+10*z)*(t+1)*arctan2(y, x+180)/(2*pi)"              (config C4's modulated singular case) */
#define NF_PSI_COUNT 6
/* datagen.py:42-66 (+ rotatePole :116-166 when deltaDeg != (0,0)): writes bounds_lon/lat (ny,nx,4) f64
 * into HBM.  lat_uses_dx=1 reproduces datagen.py:49 (latitude spaced with dx). */
int nf_datagen_bounds(double *bounds_lon_dev, double *bounds_lat_dev, long ny, long nx, double xmin,
                      double xmax, double ymin, double ymax, double delta_lon_deg, double delta_lat_deg,
                      int lat_uses_dx, void *hip_stream);
/* datagen.py:69-113 for time steps [t_begin, t_end): writes u,v ((t_end-t_begin),nz,ny,nx) of dtype into
 * HBM; nt is the series length (enters psi 4).  zhalf_k = zmin+(k+0.5)dz. */
int nf_datagen_uv(void *u_dev, void *v_dev, int dtype, long t_begin, long t_end, long nt, long nz, long ny,
                  long nx, double xmin, double xmax, double ymin, double ymax, double zmin, double zmax,
                  int lat_uses_dx, int psi, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* NEMOFLUX_AMD_H */
