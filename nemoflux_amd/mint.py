"""Drop-in for the subset of the `mint` Python package that nemoflux uses (python-mint >= 1.24.4,
/root/reference/README.md:12), backed by the gfx950 engine:

    mint.Grid().setPoints / getNumberOfCells / dump             horizgrid.py:23-24,30,43
    mint.PolylineIntegral().setGrid / buildLocator /
        computeWeights / getIntegral                            field.py:45-48,102; fluxplot.py:56
    mint.CELL_BY_CELL_DATA                                      field.py:102
    mint.VectorInterp().setGrid / buildLocator / findPoints /
        getFaceVectors                                          field.py:90-95,119-120

`import nemoflux_amd.mint as mint` in place of `import mint` is the whole integration.
Same names, argument meaning and error behaviour; all numbers come from HIP kernels (no CPU fallback).
"""
import ctypes

import numpy

from . import _lib
from ._lib import lib, check

CELL_BY_CELL_DATA = 0  # mint.CELL_BY_CELL_DATA
UNIQUE_EDGE_DATA = 1   # mint.UNIQUE_EDGE_DATA (unused by nemoflux; rejected by getIntegral)


class Grid(object):
    """mint.Grid: ncell independent quads in the planar (lon, lat) space."""

    def __init__(self):
        self.obj = ctypes.c_void_p()
        self._owner = True
        self.points = None
        check(lib.mnt_grid_new(ctypes.byref(self.obj)))

    @classmethod
    def _view(cls, handle, keepalive):
        """Wrap a Grid_t owned by a Field (shares the resident corner table)."""
        g = cls.__new__(cls)
        g.obj = ctypes.c_void_p(handle)
        g._owner = False
        g.points = None
        g._keepalive = keepalive
        return g

    def __del__(self):
        try:      # at interpreter shutdown the module globals may already be gone
            if getattr(self, '_owner', False) and self.obj:
                lib.mnt_grid_del(ctypes.byref(self.obj))
        except Exception:
            pass

    def setPoints(self, points):
        """points: float64 (ncell, 4, 3), C-contiguous; BORROWED like in mint (kept alive on self)."""
        pts = numpy.asarray(points)
        if pts.dtype != numpy.float64 or pts.ndim != 3 or pts.shape[1:] != (4, 3) or not pts.flags['C_CONTIGUOUS']:
            raise RuntimeError('ERROR: points must be a C-contiguous float64 array of shape (numCells, 4, 3)')
        self.points = pts
        check(lib.mnt_grid_setPointsPtr(ctypes.byref(self.obj), _lib.dptr(pts)))
        check(lib.mnt_grid_build(ctypes.byref(self.obj), 4, pts.shape[0]))

    def setRowLength(self, nx):
        """Extension (not in mint): the cells are the rows of a (ny, nx) grid.  A hint for the locator only -- long target
        lines are located faster; the weights do not change."""
        check(lib.mnt_grid_setRowLength(ctypes.byref(self.obj), int(nx)))

    def getNumberOfCells(self):
        n = ctypes.c_size_t()
        check(lib.mnt_grid_getNumberOfCells(ctypes.byref(self.obj), ctypes.byref(n)))
        return n.value

    def dump(self, fileName):
        check(lib.mnt_grid_dump(ctypes.byref(self.obj), str(fileName).encode('utf-8')))


class PolylineIntegral(object):
    """mint.PolylineIntegral: flux of cell-by-cell edge data across a target polyline."""

    def __init__(self):
        self.obj = ctypes.c_void_p()
        self.grid = None
        self.numSegments = 0
        check(lib.mnt_polylineintegral_new(ctypes.byref(self.obj)))

    def __del__(self):
        try:
            if self.obj:
                lib.mnt_polylineintegral_del(ctypes.byref(self.obj))
        except Exception:
            pass

    def setGrid(self, grid):
        self.grid = grid  # keep the grid (and its borrowed points) alive
        check(lib.mnt_polylineintegral_setGrid(ctypes.byref(self.obj), grid.obj))

    def buildLocator(self, numCellsPerBucket=128, periodX=360., enableFolding=False):
        check(lib.mnt_polylineintegral_buildLocator(ctypes.byref(self.obj), int(numCellsPerBucket), float(periodX),
                                                    1 if enableFolding else 0))

    def computeWeights(self, xyz, counterclock=False):
        xyz = numpy.ascontiguousarray(xyz, dtype=numpy.float64)
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise RuntimeError('ERROR: xyz must have shape (numPoints, 3)')
        self.numSegments = xyz.shape[0] - 1
        check(lib.mnt_polylineintegral_computeWeights(ctypes.byref(self.obj), xyz.shape[0], _lib.dptr(xyz),
                                                      1 if counterclock else 0))
        dropped = self.getNumberOfDroppedCrossings()
        if dropped:     # policy 'skip' (the default: mint's computeWeights has no error path, field.py:44-49) left cells out
            import warnings
            cov = self.getCoverage()
            low = numpy.nonzero(cov < 1.0 - 1.e-8)[0]
            warnings.warn(f'{dropped} crossing(s) of cells the weights are not defined on (not convex in the lon-lat plane / a '
                          f'corner at a geographic pole) were left out: {low.size} of {cov.size} target segments are covered '
                          f'only in part (down to {cov.min():.6g}); setUnsupportedCells(\'refuse\') raises instead',
                          RuntimeWarning, stacklevel=2)
        if getattr(self, '_overlap_warn', False):
            cov = self.getCoverage()
            over = numpy.array([q for q in range(cov.size) if _lib.over_covered(cov[q], xyz[q], xyz[q + 1])], dtype=int)
            if over.size:
                import warnings
                warnings.warn(f'{over.size} of {cov.size} target segments are covered more than once by the cells of the grid '
                              f'(up to {cov[over].max():.9g} times, first: segment {over[0]}): that part of the line is counted '
                              f'twice', RuntimeWarning, stacklevel=2)

    def getIntegral(self, data, placement=CELL_BY_CELL_DATA):
        """data: float64 (ncell, 4) [or flat]; host numpy (staged over PCIe) or HBM-resident
        (torch CUDA tensor / DeviceBuffer in the same (ncell,4) layout)."""
        res = ctypes.c_double()
        p = _lib.device_pointer(data)
        if p is not None:
            check(lib.mnt_polylineintegral_getIntegralDev(ctypes.byref(self.obj), p, int(placement),
                                                          ctypes.byref(res), None))
        else:
            d = numpy.ascontiguousarray(data, dtype=numpy.float64)
            if d.size != self.grid.getNumberOfCells() * 4:
                raise RuntimeError('ERROR: data must hold 4 edge values per cell')
            check(lib.mnt_polylineintegral_getIntegral(ctypes.byref(self.obj), _lib.dptr(d), int(placement),
                                                       ctypes.byref(res)))
        return res.value

    # ---- extensions beyond mint
    def setUnsupportedCells(self, policy='skip'):
        """What computeWeights does when the line overlaps a cell the weights are not defined on (not convex in the lon-lat
        plane / a corner at a geographic pole): 'skip' (the default of this mint-shaped class since round 6: mint's
        computeWeights never raises there, field.py:44-49) drops the cell, warns (RuntimeWarning with the number of
        crossings dropped) and getCoverage() reports < 1 for the segments concerned; 'refuse' raises, naming the cell."""
        if policy not in ('refuse', 'skip'):
            raise RuntimeError("ERROR: policy must be 'refuse' or 'skip'")
        check(lib.mnt_polylineintegral_setUnsupportedCells(ctypes.byref(self.obj), 1 if policy == 'skip' else 0))

    def setOverlappingCells(self, policy='refuse'):
        """What computeWeights does when a stretch of the line lies in two cells that do not hold the same sub-segment
        (overlapping cells: it would be counted twice): 'refuse' (default) raises, 'warn' builds the weights as they come
        and warns -- getCoverage() then reports > 1 for the segments concerned."""
        if policy not in ('refuse', 'warn'):
            raise RuntimeError("ERROR: policy must be 'refuse' or 'warn'")
        self._overlap_warn = policy == 'warn'
        check(lib.mnt_polylineintegral_setOverlappingCells(ctypes.byref(self.obj), 1 if policy == 'warn' else 0))

    def getNumberOfDroppedCrossings(self):
        """(cell, target-segment image) crossings of unsupported cells that the last computeWeights left out (policy 'skip')."""
        n = ctypes.c_size_t()
        check(lib.mnt_polylineintegral_getNumberOfDroppedCrossings(ctypes.byref(self.obj), ctypes.byref(n)))
        return n.value

    def getCoverage(self):
        """Fraction of every target segment that lies inside cells of the grid (1 = inside, each point counted once)."""
        cov = numpy.zeros(max(self.numSegments, 1), numpy.float64)
        check(lib.mnt_polylineintegral_getCoverage(ctypes.byref(self.obj), _lib.dptr(cov)))
        return cov[:self.numSegments]

    def getSegmentIntegrals(self, data):
        """Per-target-segment sums (numSegments,) and the total."""
        res = ctypes.c_double()
        seg = numpy.zeros(max(self.numSegments, 1), numpy.float64)
        p = _lib.device_pointer(data)
        buf = None
        if p is None:
            d = numpy.ascontiguousarray(data, dtype=numpy.float64)
            buf = _lib.DeviceBuffer(d.nbytes).upload(d)
            p = buf.ptr
        check(lib.mnt_polylineintegral_getIntegralDev(ctypes.byref(self.obj), p, CELL_BY_CELL_DATA,
                                                      ctypes.byref(res), _lib.dptr(seg)))
        if buf is not None:
            buf.free()
        return seg[:self.numSegments], res.value

    def getWeights(self):
        """(cell*4+edge int64, weight float64, segment int32) of every entry, sorted by segment."""
        n = ctypes.c_size_t()
        check(lib.mnt_polylineintegral_getNumberOfWeights(ctypes.byref(self.obj), ctypes.byref(n)))
        ce = numpy.empty(n.value, numpy.int64)
        w = numpy.empty(n.value, numpy.float64)
        sg = numpy.empty(n.value, numpy.int32)
        check(lib.mnt_polylineintegral_getWeights(ctypes.byref(self.obj), ce.ctypes.data_as(_lib.c_int64_p),
                                                  _lib.dptr(w), sg.ctypes.data_as(_lib.c_int_p)))
        return ce, w, sg


class VectorInterp(object):
    """mint.VectorInterp: vectors at target points from cell-by-cell edge data (W2 / face interpolation)."""

    def __init__(self):
        self.obj = ctypes.c_void_p()
        self.grid = None
        self.numTargetPoints = 0
        self.numNotFound = 0
        check(lib.mnt_vectorinterp_new(ctypes.byref(self.obj)))

    def __del__(self):
        try:
            if self.obj:
                lib.mnt_vectorinterp_del(ctypes.byref(self.obj))
        except Exception:
            pass

    def setGrid(self, grid):
        self.grid = grid
        check(lib.mnt_vectorinterp_setGrid(ctypes.byref(self.obj), grid.obj))

    def buildLocator(self, numCellsPerBucket=128, periodX=360., enableFolding=False):
        check(lib.mnt_vectorinterp_buildLocator(ctypes.byref(self.obj), int(numCellsPerBucket), float(periodX),
                                                1 if enableFolding else 0))

    def findPoints(self, targetPoints, tol2=1.e-12):
        """targetPoints: (n, 3); returns the number of points that fall outside every cell."""
        tp = numpy.ascontiguousarray(targetPoints, dtype=numpy.float64)
        if tp.size and (tp.ndim != 2 or tp.shape[1] != 3):
            raise RuntimeError('ERROR: targetPoints must have shape (numPoints, 3)')
        self.numTargetPoints = tp.shape[0] if tp.size else 0
        nf = ctypes.c_size_t()
        check(lib.mnt_vectorinterp_findPoints(ctypes.byref(self.obj), self.numTargetPoints,
                                              _lib.dptr(tp) if tp.size else None, float(tol2), ctypes.byref(nf)))
        self.numNotFound = nf.value
        return nf.value

    def getFaceVectors(self, data, placement=CELL_BY_CELL_DATA, out=None, _layout=0):
        """(n, 3) vectors; data: (ncell, 4) host array, or HBM-resident (torch CUDA tensor / DeviceBuffer / int)."""
        res = numpy.zeros((self.numTargetPoints, 3), numpy.float64) if out is None else out
        if self.numTargetPoints == 0:
            return res
        p = _lib.device_pointer(data)
        if p is not None:
            check(lib.mnt_vectorinterp_getFaceVectorsDev(ctypes.byref(self.obj), p, int(_layout), _lib.dptr(res)))
        else:
            d = numpy.ascontiguousarray(data, dtype=numpy.float64)
            if d.size != self.grid.getNumberOfCells() * 4:
                raise RuntimeError('ERROR: data must hold 4 edge values per cell')
            check(lib.mnt_vectorinterp_getFaceVectors(ctypes.byref(self.obj), _lib.dptr(d), int(placement),
                                                      _lib.dptr(res)))
        return res

    def getCells(self):
        """(cell id or -1, (xi, eta)) of every target point."""
        ids = numpy.zeros(self.numTargetPoints, numpy.int64)
        pc = numpy.zeros((self.numTargetPoints, 2), numpy.float64)
        if self.numTargetPoints:
            check(lib.mnt_vectorinterp_getCells(ctypes.byref(self.obj), ids.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)),
                                                _lib.dptr(pc)))
        return ids, pc
