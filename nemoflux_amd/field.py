"""Field: the flux engine, same surface as nemoflux/field.py:15-234, running on the gfx950 kernels.

What fluxviz.py / fluxplot.py read stays available under the same names: timeIndex, nt, nz,
ny, nx, dx, lonmin..latmax, sverdrup, maxAbsFlux, lonlat, edgeFluxesUArray, edgeFluxesVArray,
integratedVelocity, vectorPoints, vectorValues, plis, timeObj, gr, thickness, arcLengths; update(),
getFluxText(); plus computeFlux(tIndex) / computeAll() (BASELINE.json north_star).  The host arrays that
VTK aliases (fluxviz.py:148,160,168) are written IN PLACE at every update().
"""
import ctypes
import re

import numpy

from . import _lib, mint
from ._lib import lib, check, NF_F64, NF_F32
from .horizgrid import HorizGrid
from .io import open_tfile, open_uvfile
from .timeobj import TimeObj

EARTH_RADIUS = 6371000.0  # field.py:12


def _dtype_code(a):
    import torch
    dt = a.dtype
    if isinstance(dt, numpy.dtype):
        dt = dt.type
    if dt in (numpy.float64, torch.float64):
        return NF_F64
    if dt in (numpy.float32, torch.float32):
        return NF_F32
    raise RuntimeError(f'ERROR: unsupported dtype {dt} (need float64 or float32)')


def _native(a):
    """C-contiguous host array in native byte order (files may be big-endian: NetCDF classic, some HDF5)."""
    a = numpy.ascontiguousarray(a)
    return a if a.dtype.isnative else a.astype(a.dtype.newbyteorder('='))


def _geometry_only(bounds_lon, bounds_lat):
    """Run the geometry kernel alone: {'points': (ncell,4,3), 'arcLengths': (ncell,4), 'box': 4 floats}."""
    h = ctypes.c_void_p()
    check(lib.nf_field_new(ctypes.byref(h)))
    try:
        blon = _native(bounds_lon)
        blat = _native(bounds_lat)
        if blon.dtype != blat.dtype:
            blat = blat.astype(blon.dtype)
        ny, nx, _ = blon.shape
        check(lib.nf_field_set_bounds(ctypes.byref(h), blon.ctypes.data, blat.ctypes.data, ny, nx, _dtype_code(blon), 0))
        pts = numpy.empty((ny * nx, 4, 3), numpy.float64)
        arc = numpy.empty((ny * nx, 4), numpy.float64)
        check(lib.nf_field_get_points(ctypes.byref(h), _lib.dptr(pts)))
        check(lib.nf_field_get_arclengths(ctypes.byref(h), _lib.dptr(arc)))
        b = [ctypes.c_double() for _ in range(4)]
        check(lib.nf_field_get_box(ctypes.byref(h), *[ctypes.byref(x) for x in b]))
        return dict(points=pts, arcLengths=arc, box=[x.value for x in b])
    finally:
        lib.nf_field_del(ctypes.byref(h))


class _TimeObj(object):
    """timeobj.TimeObj stand-in that tolerates a missing time axis."""

    def __init__(self, timeValues=None):
        self.timeVarName = 'time_counter' if timeValues is not None else ''
        self.timeValues = timeValues

    def getSize(self):
        return 0 if self.timeValues is None else len(self.timeValues)

    def getValues(self):
        return self.timeValues

    def getTimeAsDate(self, timeIndex):
        return timeIndex if self.timeValues is None else self.timeValues[timeIndex]

    def getTimeAsString(self, timeIndex):
        return f'{timeIndex}' if self.timeValues is None else f'{self.timeValues[timeIndex]}'


class _PinnedBlock(object):
    """Owner of one pinned host allocation; freed when the last numpy view of it is gone (VTK may outlive the Field)."""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        check(lib.nf_host_alloc(ctypes.byref(p), max(int(nbytes), 8)))
        self.ptr = p.value

    def __del__(self):
        try:
            if self.ptr:
                lib.nf_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class _Transect(object):
    """Entry of Field.plis: quacks like mint.PolylineIntegral.getIntegral (fluxplot.py:56).

    When asked about the Field's own integratedVelocity it returns the value the batched on-device
    reduction produced for the current step; any other data goes through a real PolylineIntegral."""

    def __init__(self, field, index, xyz):
        self._field, self._index, self._xyz = field, index, xyz
        self._pli = None

    def getIntegral(self, data, placement=mint.CELL_BY_CELL_DATA):
        f = self._field
        if data is f.integratedVelocity and f._row_valid:
            return float(f._row[f._nseg + self._index])
        if self._pli is None:
            self._pli = mint.PolylineIntegral()
            self._pli.setGrid(f.gr.getMintGrid())
            self._pli.buildLocator(numCellsPerBucket=128, periodX=f.periodX, enableFolding=False)
            self._pli.computeWeights(self._xyz, counterclock=False)
        return self._pli.getIntegral(data, placement)

    def getSegmentIntegrals(self):
        f = self._field
        o = f._tr_off
        return numpy.array(f._row[o[self._index]:o[self._index + 1]])


class Field(object):

    def __init__(self, tFile, uFile, vFile, lonLatZPoints, sverdrup=False, **kw):
        """Same positional signature as the reference (field.py:17).  tFile/uFile/vFile: NetCDF-4 files (read by
        nemoflux_amd.io / hdf5min; compressed uo/vo one time step at a time) or the .npz bundles of
        nemoflux_amd.datagen / subsetnemo; see fromArrays for in-memory / HBM data and _setup for the keywords
        (fill_value, periodX, slab_range, readback, compact, stream, unsupportedCells, overlappingCells, ...)."""
        t = open_tfile(tFile)
        if 'deptht_bounds' not in t:
            raise RuntimeError(f'ERROR: {tFile} has no variable deptht_bounds')
        uo, _, uvars = open_uvfile(uFile, 'uo', with_all=True)
        vo, _, vvars = open_uvfile(vFile, 'vo', with_all=True)
        # values that mean 'missing' (xarray's decode_cf masks _FillValue and missing_value, field.py:34-35, 157): the engine
        # compares every value of uo and vo with up to two markers
        markers = list(uvars['_markers_uo'])
        for m in vvars['_markers_vo']:
            if m not in markers:
                markers.append(m)
        if 'fill_value' in kw:
            markers = [float(kw.pop('fill_value'))] + [m for m in markers[1:2]]
        if len(markers) > 2:
            raise RuntimeError(f'ERROR: uo / vo carry {len(markers)} different _FillValue / missing_value markers '
                               f'({markers}); the engine masks at most two')
        kw.setdefault('timeObj', TimeObj.fromVariables(uvars))   # field.py:38: TimeObj(self.ncU)
        # The reference's constructor never raises over a target line that meets a distorted polar cell (field.py:44-49: mint's
        # computeWeights has no error path), so the drop-in signature leaves such cells out and warns (coverage + the number
        # of crossings dropped); Field.fromArrays -- the batch constructor of bench.py / fluxplot-style drivers -- refuses.
        kw.setdefault('unsupportedCells', 'skip')
        self._setup(t['bounds_lon'], t['bounds_lat'], t['deptht_bounds'], uo, vo, lonLatZPoints, sverdrup,
                    fill_value=markers[0] if markers else numpy.nan,
                    missing_value=markers[1] if len(markers) > 1 else numpy.nan, **kw)

    @classmethod
    def fromArrays(cls, bounds_lon, bounds_lat, deptht_bounds, uo, vo, lonLatZPoints, sverdrup=False, **kw):
        """bounds_*: (ny,nx,4) host arrays (f64/f32) or torch CUDA tensors; uo/vo: (nt,nz,ny,nx) [or (nz,ny,nx)]
        host arrays, torch CUDA tensors or DeviceArray (HBM-resident, used in place)."""
        self = cls.__new__(cls)
        self._setup(bounds_lon, bounds_lat, deptht_bounds, uo, vo, lonLatZPoints, sverdrup, **kw)
        return self

    # ------------------------------------------------------------------------------------------
    def _setup(self, bounds_lon, bounds_lat, deptht_bounds, uo, vo, lonLatZPoints, sverdrup,
               fill_value=numpy.nan, missing_value=numpy.nan, periodX=360., numCellsPerBucket=128, slab_range=None,
               readback=True,
               timeValues=None, stream=None, timeObj=None, compact=False, prefetch=True, gpu_decode=True,
               unsupportedCells='refuse', overlappingCells='refuse'):
        _lib.require_gpu()
        self.sverdrup = sverdrup
        self.periodX = periodX
        self._readback = readback
        self._h = ctypes.c_void_p()
        check(lib.nf_field_new(ctypes.byref(self._h)))
        if stream is not None:
            check(lib.nf_field_set_stream(ctypes.byref(self._h), ctypes.c_void_p(stream)))
        self._keep = [bounds_lon, bounds_lat, uo, vo]  # borrowed buffers must outlive the handle

        # --- cell bounds -> geometry kernel (field.py:22-31, 42, 56)
        plon, plat = _lib.device_pointer(bounds_lon), _lib.device_pointer(bounds_lat)
        if plon is None:
            bounds_lon = _native(bounds_lon)
            bounds_lat = _native(bounds_lat)
            if bounds_lat.dtype != bounds_lon.dtype:
                bounds_lat = bounds_lat.astype(bounds_lon.dtype)
            self._keep += [bounds_lon, bounds_lat]
            plon, plat, on_dev = bounds_lon.ctypes.data, bounds_lat.ctypes.data, 0
        else:
            on_dev = 1
        if len(bounds_lon.shape) != 3 or bounds_lon.shape[2] != 4 or tuple(bounds_lon.shape) != tuple(bounds_lat.shape):
            raise RuntimeError('ERROR: bounds_lat/bounds_lon must have shape (ny, nx, 4)')
        self._bounds = (bounds_lon, bounds_lat)
        ny, nx = int(bounds_lon.shape[0]), int(bounds_lon.shape[1])
        check(lib.nf_field_set_bounds(ctypes.byref(self._h), plon, plat, ny, nx, _dtype_code(bounds_lon), on_dev))
        b = [ctypes.c_double() for _ in range(4)]
        check(lib.nf_field_get_box(ctypes.byref(self._h), *[ctypes.byref(x) for x in b]))
        self.lonmin, self.lonmax, self.latmin, self.latmax = [x.value for x in b]
        print(f'lon-lat box: {self.lonmin}, {self.latmin} -> {self.lonmax}, {self.latmax}')  # field.py:31

        self.timeIndex = 0
        self.timeObj = timeObj if timeObj is not None else _TimeObj(timeValues)

        # --- uo / vo (field.py:34-35, 122-136)
        self.nt, self.nz, sy, sx = self.getSizes(tuple(uo.shape))
        if (sy, sx) != (ny, nx) or tuple(vo.shape) != tuple(uo.shape):
            raise RuntimeError("ERROR: uo/vo shapes do not match the (ny, nx) of the cell bounds")
        self.ny, self.nx = ny, nx
        pu, pv = _lib.device_pointer(uo), _lib.device_pointer(vo)
        self._lazy = None
        if hasattr(uo, 'read_step') or hasattr(vo, 'read_step'):
            # file-backed variables inflated one time step at a time (nemoflux_amd.hdf5min.LazyVariable)
            self._lazy = (uo, vo)
            self._lazy_dtype = numpy.dtype(uo.dtype).newbyteorder('=')
            self._uv_code, self._fill = _dtype_code(uo), float(fill_value)
            self._lazy_step = -1
            # two slots of staging buffers: while the GPU works on the steps of one, a background host thread prepares the
            # next ones in the other; deflated HDF5 chunks are inflated on the device (nemoflux_amd.staging / ingest)
            from .staging import StepStager
            self._stager = StepStager((uo, vo), self.nt, self.nz, ny, nx, self._lazy_dtype, self._host_array,
                                      prefetch=prefetch, gpu_decode=gpu_decode)
            pu = pv = None
            uv_dev = 0
        elif pu is None:
            uo = _native(uo)
            vo = _native(vo)
            if vo.dtype != uo.dtype:
                vo = vo.astype(uo.dtype)
            self._keep += [uo, vo]
            pu, pv, uv_dev = uo.ctypes.data, vo.ctypes.data, 0
        else:
            uv_dev = 1
        if self._lazy is None:
            check(lib.nf_field_set_uv(ctypes.byref(self._h), pu, pv, self.nt, _dtype_code(uo), uv_dev, float(fill_value)))
        if missing_value == missing_value:   # a second marker (CF missing_value that differs from _FillValue)
            check(lib.nf_field_set_missing_value(ctypes.byref(self._h), float(missing_value)))
        check(lib.nf_field_set_sverdrup(ctypes.byref(self._h), 1 if sverdrup else 0))
        if compact:   # keep only (eU, eV) resident; the (ncell,4) copies and |.| arrays are derived at read-back
            check(lib.nf_field_set_compact(ctypes.byref(self._h), 1))
        if slab_range is not None:
            check(lib.nf_field_set_slab_range(ctypes.byref(self._h), int(slab_range[0]), int(slab_range[1])))
        self.slab_range = slab_range

        # --- layer thickness (field.py:51)
        self.bounds_depth = numpy.asarray(deptht_bounds)
        self.thickness = numpy.ascontiguousarray(self.bounds_depth[:, 1] - self.bounds_depth[:, 0], dtype=numpy.float64)
        if self.thickness.shape[0] != self.nz:
            raise RuntimeError('ERROR: deptht_bounds does not match the number of levels of uo')
        check(lib.nf_field_set_thickness(ctypes.byref(self._h), _lib.dptr(self.thickness), self.nz))

        # --- grid + transects (field.py:42-49): all polylines in one batched weight build
        self.gr = HorizGrid(_field=self)
        self._polylines = [numpy.ascontiguousarray(numpy.array(p, dtype=numpy.float64)).reshape(-1, 3)
                           for p in lonLatZPoints]
        self.plis = []
        for i, xyz in enumerate(self._polylines):
            tid = ctypes.c_int()
            check(lib.nf_field_add_transect(ctypes.byref(self._h), _lib.dptr(xyz), xyz.shape[0], 0, ctypes.byref(tid)))
            self.plis.append(_Transect(self, i, xyz))
        if unsupportedCells not in ('refuse', 'skip'):
            raise RuntimeError("ERROR: unsupportedCells must be 'refuse' or 'skip'")
        if unsupportedCells == 'skip':   # non-convex / pole-vertex cells drop out; the coverage warning below reports it
            check(lib.nf_field_set_unsupported_cells(ctypes.byref(self._h), 1))
        if overlappingCells not in ('refuse', 'warn'):
            raise RuntimeError("ERROR: overlappingCells must be 'refuse' or 'warn'")
        if overlappingCells == 'warn':   # a stretch of a line found in two overlapping cells counts twice; the warning below says so
            check(lib.nf_field_set_overlapping_cells(ctypes.byref(self._h), 1))
        check(lib.nf_field_build_weights(ctypes.byref(self._h), int(numCellsPerBucket), float(periodX)))
        n = ctypes.c_int()
        check(lib.nf_field_num_segments(ctypes.byref(self._h), ctypes.byref(n)))
        self._nseg = n.value
        self._tr_off = numpy.zeros(len(self.plis) + 1, numpy.int32)
        check(lib.nf_field_segment_offsets(ctypes.byref(self._h), self._tr_off.ctypes.data_as(_lib.c_int_p)))
        check(lib.nf_field_row_length(ctypes.byref(self._h), ctypes.byref(n)))
        self._rowlen = n.value
        self._row = numpy.zeros(max(self._rowlen, 1), numpy.float64)
        self._row_valid = False
        # coverage > 1 (overlapping cells: a stretch of the line would be counted twice) was refused by build_weights above
        # unless overlappingCells='warn'; coverage < 1 means part of the line lies in no cell: mint only warns there [recall],
        # and so does this
        n = ctypes.c_size_t()
        check(lib.nf_field_num_dropped_crossings(ctypes.byref(self._h), ctypes.byref(n)))
        self.droppedCrossings = n.value
        if self.droppedCrossings:
            import warnings
            warnings.warn(f'{self.droppedCrossings} crossing(s) of cells the weights are not defined on (not convex in the lon-lat '
                          f'plane / a corner at a geographic pole) were left out of the transects (unsupportedCells=\'skip\'); '
                          f'the coverage warnings below name the segments', RuntimeWarning, stacklevel=3)
        for i, cov in enumerate(self.getCoverage()):
            xyz = self._polylines[i]
            over = numpy.array([q for q in range(cov.size) if overlappingCells == 'warn' and
                                _lib.over_covered(cov[q], xyz[q], xyz[q + 1])], dtype=int)
            if over.size:
                import warnings
                warnings.warn(f'transect {i}: {over.size} of {cov.size} target segments are covered more than once by the cells '
                              f'of the grid (up to {cov[over].max():.9g} times, first: segment {over[0]}): that part of the line '
                              f'is counted twice', RuntimeWarning, stacklevel=3)
            low = numpy.nonzero(cov < 1.0 - 1.e-8)[0]
            if low.size:
                import warnings
                warnings.warn(f'transect {i}: {low.size} of {cov.size} target segments are not fully inside the grid '
                              f'(covered fraction {cov[low].min():.6g} .. {cov[low].max():.6g}, first: segment {low[0]}); '
                              f'the parts outside contribute no flux', RuntimeWarning, stacklevel=3)

        numCells = self.ny * self.nx
        self.dx = min((self.lonmax - self.lonmin) / float(self.nx), (self.latmax - self.latmin) / float(self.ny))
        self._arc = None

        # --- host mirrors of the per-step arrays (field.py:59-63); pinned, updated in place
        self.edgeFluxesUArray = self._host_zeros((numCells,))
        self.edgeFluxesVArray = self._host_zeros((numCells,))
        self.integratedVelocity = self._host_zeros((numCells, 4))
        self.maxAbsFlux = 0.

        # first step (field.py:65-67)
        self._compute(self.timeIndex)
        print(f'max vertically integrated edge |flux|: {self.maxAbsFlux}')

        self._lonlat = None
        # arrow seed points along the target lines (field.py:71-87)
        vectorPoints = []
        uVectors = []
        # (batch drivers -- readback=False -- never look at the arrows: a batch of 4 096 transects on the ORCA12-like grid would
        # seed 2 x 10^8 of them)
        for lonlatpts in (self._polylines if self._readback else []):
            for i in range(len(lonlatpts) - 1):
                begPoint = numpy.array(lonlatpts[i])
                endPoint = numpy.array(lonlatpts[i + 1])
                u = endPoint - begPoint
                distance = numpy.sqrt(u.dot(u))
                if distance == 0 or self.dx <= 0:
                    continue
                u /= distance
                nvpts = max(2, int(distance / self.dx))
                vdx = distance / float(nvpts - 1)
                # begPoint + u*j*vdx for j in range(nvpts) (field.py:84-86), vectorised over j
                vectorPoints.append(begPoint + (u * numpy.arange(nvpts)[:, None]) * vdx)
                uVectors.append(numpy.broadcast_to(u, (nvpts, 3)))
        self.uVectors = numpy.concatenate(uVectors) if uVectors else numpy.zeros((0, 3))
        vectorPoints = numpy.concatenate(vectorPoints) if vectorPoints else []
        self.vectorPoints = numpy.ascontiguousarray(vectorPoints) if len(vectorPoints) else numpy.zeros((0, 3))
        # compute the vector at the target line (field.py:89-95), from the resident planes (no host round trip)
        self.vectorValues = numpy.zeros((self.vectorPoints.shape[0], 3), numpy.float64)
        self.vinterp = None
        if self._readback:   # batch drivers (readback=False) never look at the arrows
            self.vinterp = mint.VectorInterp()
            self.vinterp.setGrid(self.gr.getMintGrid())
            self.vinterp.buildLocator(numCellsPerBucket=128, periodX=periodX)
            self.vinterp.findPoints(self.vectorPoints, tol2=1.e-12)
            self._update_vectors()

    # ------------------------------------------------------------------------------------------
    def _host_array(self, shape, dtype=numpy.float64):
        """Pinned host array (fast D2H / H2D); the allocation lives as long as any view of the array."""
        dtype = numpy.dtype(dtype)
        n = int(numpy.prod(shape))
        nbytes = max(n, 1) * dtype.itemsize
        block = _PinnedBlock(nbytes)
        buf = (ctypes.c_ubyte * nbytes).from_address(block.ptr)
        buf._owner = block  # the ctypes object is the numpy array's base: it keeps the block alive
        return numpy.ctypeslib.as_array(buf).view(dtype)[:n].reshape(shape)

    def _host_zeros(self, shape):
        a = self._host_array(shape)
        a[...] = 0.0
        return a

    def __del__(self):
        try:
            if getattr(self, '_stager', None) is not None:
                self._stager.close()      # the prefetch thread writes into buffers this object owns
            if getattr(self, '_h', None):
                lib.nf_field_del(ctypes.byref(self._h))
        except Exception:
            pass

    def _download_points(self):
        pts = numpy.empty((self.ny * self.nx, 4, 3), numpy.float64)
        check(lib.nf_field_get_points(ctypes.byref(self._h), _lib.dptr(pts)))
        return pts

    @property
    def arcLengths(self):
        """(ncell, 4) great-circle edge lengths on the unit sphere (field.py:55-56, 170-181)."""
        if self._arc is None:
            self._arc = numpy.empty((self.ny * self.nx, 4), numpy.float64)
            check(lib.nf_field_get_arclengths(ctypes.byref(self._h), _lib.dptr(self._arc)))
        return self._arc

    @property
    def lonlat(self):
        """(ny, nx, 4, 3) corner coordinates for the VTK edge grids (field.py:138-143)."""
        if self._lonlat is None:
            self._lonlat = self.gr.getPoints().reshape((self.ny, self.nx, 4, 3))
        return self._lonlat

    def buildEdgeUVGrids(self, bounds_lon=None, bounds_lat=None):
        return self.lonlat

    def getSizes(self, shapeU=None):
        """field.py:122-136."""
        nt, nz, ny, nx = 1, 1, 0, 0
        if shapeU is None:
            return self.nt, self.nz, self.ny, self.nx
        if len(shapeU) == 4:
            nt, nz, ny, nx = shapeU
        elif len(shapeU) == 3:
            nz, ny, nx = shapeU
        elif len(shapeU) == 2:
            ny, nx = shapeU
        else:
            raise RuntimeError("ERROR: uo's shape does not match (t, z, y, x), (z, y, x) or (y, x)")
        return int(nt), int(nz), int(ny), int(nx)

    # ------------------------------------------------------------------------------------------
    def _compute(self, tIndex, readback=None, prefetch_next=None):
        if not (0 <= tIndex < self.nt):
            raise RuntimeError(f'ERROR: time index {tIndex} out of range [0, {self.nt})')
        if self._lazy is not None and self._lazy_step != tIndex:
            # one time step from the file(s): staged in pinned host memory or decoded into an HBM slab; the engine sees a
            # virtual (nt, nz, ny, nx) base that it only dereferences at step tIndex
            pu, pv, on_dev = self._stager.get(tIndex)
            self._lazy_step = tIndex
            off = tIndex * self._stager.step_bytes
            check(lib.nf_field_set_uv(ctypes.byref(self._h), pu - off, pv - off, self.nt, self._uv_code, on_dev, self._fill))
            # the steps that come next (fluxviz's 't' key, fluxplot's loop) are prepared on a host thread while the GPU works
            # on this one: the blocking C call below releases the GIL
            nxt = self._stager.next_after(tIndex)
            self._stager.prefetch(nxt % self.nt if prefetch_next is None else (nxt if nxt < self.nt else -1))
        check(lib.nf_field_compute_flux(ctypes.byref(self._h), int(tIndex), _lib.dptr(self._row)))
        self._row_valid = True
        if self._readback if readback is None else readback:
            m = ctypes.c_double()
            check(lib.nf_field_read_step(ctypes.byref(self._h), _lib.dptr(self.integratedVelocity),
                                         _lib.dptr(self.edgeFluxesUArray), _lib.dptr(self.edgeFluxesVArray),
                                         ctypes.byref(m)))
            self.maxAbsFlux = max(self.maxAbsFlux, m.value)  # field.py:234
        return self._row

    def _update_vectors(self):
        if self.vectorValues.shape[0] and self.vinterp is not None:
            p = ctypes.c_void_p()
            check(lib.nf_field_device_ptr(ctypes.byref(self._h), 0, ctypes.byref(p)))
            self.vinterp.getFaceVectors(p.value, out=self.vectorValues, _layout=1)   # in place (fluxviz.py:301)

    def update(self):
        """field.py:112-120: recompute the current time step; host arrays are refreshed in place."""
        self._compute(self.timeIndex, readback=True)
        self._update_vectors()   # field.py:119-120

    def computeFlux(self, tIndex, readback=False):
        """BASELINE north_star's computeFlux(tIndex): set the time index, run the step on the GPU and return
        the total flux of every transect (list of floats).  = fluxplot.py:51-59 for one step."""
        self.timeIndex = int(tIndex)
        row = self._compute(self.timeIndex, readback=readback)
        return [float(row[self._nseg + i]) for i in range(len(self.plis))]

    def getSegmentFluxes(self):
        """Per-target-segment sums of the last computed step, one array per transect."""
        return [numpy.array(self._row[self._tr_off[i]:self._tr_off[i + 1]]) for i in range(len(self.plis))]

    def computeAll(self, out=None):
        """All nt steps back to back on the GPU (no host round trip per step).  Returns (nt, ntransect)
        totals and (nt, nseg) per-segment sums.  `out`: optional torch CUDA tensor (nt, row_length) to
        receive the raw rows in HBM (for the RCCL reduce of nemoflux_amd.dist)."""
        import torch
        if self._lazy is not None:
            # file-backed: one step on the GPU, the next one inflating into the other pinned slot (no wrap-around prefetch
            # after the last step)
            rows = numpy.array([self._compute(t, readback=False, prefetch_next=True).copy() for t in range(self.nt)])
            if out is not None:
                out.copy_(torch.from_numpy(rows))
            return rows[:, self._nseg:self._nseg + len(self.plis)], rows[:, :self._nseg]
        if out is None:
            out = torch.empty((self.nt, max(self._rowlen, 1)), dtype=torch.float64, device='cuda')
        check(lib.nf_field_compute_all_async(ctypes.byref(self._h), ctypes.c_void_p(out.data_ptr())))
        rows = out.cpu().numpy()
        return rows[:, self._nseg:self._nseg + len(self.plis)], rows[:, :self._nseg]

    def getFluxText(self):
        """field.py:98-109."""
        txt = ""
        for pli in self.plis:
            totalFlux = pli.getIntegral(self.integratedVelocity, mint.CELL_BY_CELL_DATA)
            txt += f"{totalFlux:4.3g}, "
        if self.sverdrup:
            txt += "(Sv) "
        else:
            txt += "(A m^2/s) "
        txt = re.sub(r',\s*\(', ' (', txt)
        return txt

    def getWeights(self):
        """(cell*4+edge, weight, global segment id) of the batched transect set."""
        n = ctypes.c_size_t()
        check(lib.nf_field_num_weights(ctypes.byref(self._h), ctypes.byref(n)))
        ce = numpy.empty(n.value, numpy.int64)
        w = numpy.empty(n.value, numpy.float64)
        sg = numpy.empty(n.value, numpy.int32)
        check(lib.nf_field_get_weights(ctypes.byref(self._h), ce.ctypes.data_as(_lib.c_int64_p), _lib.dptr(w),
                                       sg.ctypes.data_as(_lib.c_int_p)))
        return ce, w, sg

    def getCoverage(self):
        """Per transect: the fraction of each of its target segments that lies inside cells of the grid (1 = inside, each
        point counted once; less = part of the line is outside the grid and contributes no flux)."""
        cov = numpy.zeros(max(self._nseg, 1), numpy.float64)
        check(lib.nf_field_get_coverage(ctypes.byref(self._h), _lib.dptr(cov)))
        return [cov[self._tr_off[i]:self._tr_off[i + 1]].copy() for i in range(len(self.plis))]

    def getEdgeWeights(self):
        """(element of [eU | eV], weight, global segment id): the weights folded onto the unique edges of the two signed
        planes, which is what the on-device reduction gathers (one value per entry)."""
        n = ctypes.c_size_t()
        check(lib.nf_field_num_edge_weights(ctypes.byref(self._h), ctypes.byref(n)))
        el = numpy.empty(n.value, numpy.int32)
        sg = numpy.empty(n.value, numpy.int32)
        w = numpy.empty(n.value, numpy.float64)
        check(lib.nf_field_get_edge_weights(ctypes.byref(self._h), el.ctypes.data_as(_lib.c_int_p),
                                            sg.ctypes.data_as(_lib.c_int_p), _lib.dptr(w)))
        return el, w, sg

    # timing hooks for bench.py
    def enableKernelTiming(self, on=True, reserve=0):
        """reserve: number of launches whose events are created now, outside the timed region"""
        check(lib.nf_field_timing(ctypes.byref(self._h), max(1, int(reserve)) if on else 0))

    def readKernelTiming(self, split=False):
        """(launches, total ms) of the timed steps since enableKernelTiming; split=True appends the flux-kernel and the
        expansion-kernel shares of that total."""
        n, ms = ctypes.c_long(), ctypes.c_double()
        check(lib.nf_field_timing_read(ctypes.byref(self._h), ctypes.byref(n), ctypes.byref(ms)))
        if not split:
            return n.value, ms.value
        a, b = ctypes.c_double(), ctypes.c_double()
        check(lib.nf_field_timing_split(ctypes.byref(self._h), ctypes.byref(a), ctypes.byref(b)))
        return n.value, ms.value, a.value, b.value

    def readTransectTiming(self):
        """ms spent in the transect reductions behind the launches of the last readKernelTiming"""
        k3 = ctypes.c_double()
        check(lib.nf_field_timing_k3(ctypes.byref(self._h), ctypes.byref(k3)))
        return k3.value
