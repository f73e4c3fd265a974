"""CPU oracle for the nemoflux transect-flux hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (nemoflux_amd/) never does.  Two layers:

  * ctypes wrappers around oracle/libnf_oracle.so (nf_oracle.c, the plain-C restatement), and
  * numpy restatements "as written" of the reference statements (used for the CPU baseline timing and
    for generating mid-size inputs): datagen.py:31-166, field.py:145-234, geo.py:14-27, fluxexact.py:21-46.

Pinning: A1/A3/A4/A5/datagen are checked against tests/golden/*.npz, produced by running the reference
itself (oracle/gen_golden.py).  A6/A7 restate python-mint (>=1.24.4, absent): pinned by the README known
answers only -- see the header of nf_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy
from numpy import pi, cos, sin, arctan2  # noqa: F401  (names the stream-function strings may use)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libnf_oracle.so')
_lib = None

DEG2RAD = numpy.pi / 180.  # geo.py:4
EARTH_RADIUS_SV = 6371000.0  # field.py:12


def build():
    """Compile the C restatement (gcc is in the image); idempotent."""
    src = os.path.join(_HERE, 'nf_oracle.c')
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'libnf_oracle.so'])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        L.nfo_assemble_points.argtypes = [dp, dp, ctypes.c_long, dp]
        L.nfo_arc_lengths.argtypes = [dp, ctypes.c_long, dp]
        L.nfo_vertical_integral.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_long, dp,
                                            ctypes.c_double, dp]
        L.nfo_edge_flux.argtypes = [dp, dp, dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, dp, dp, dp, dp]
        L.nfo_polyline_weights.argtypes = [dp, ctypes.c_long, dp, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                           ctypes.c_long, ctypes.POINTER(ctypes.c_int64), dp,
                                           ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_long), dp, ctypes.c_int]
        L.nfo_polyline_weights.restype = ctypes.c_long
        L.nfo_get_integral.argtypes = [dp, ctypes.c_long, ctypes.POINTER(ctypes.c_int64), dp,
                                       ctypes.POINTER(ctypes.c_int), ctypes.c_int, dp]
        L.nfo_get_integral.restype = ctypes.c_double
        L.nfo_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_long,
                               ctypes.c_long, dp, ctypes.c_double, dp, ctypes.c_int, dp, dp, dp, dp, dp, dp,
                               ctypes.c_long, ctypes.POINTER(ctypes.c_int64), dp, ctypes.POINTER(ctypes.c_int)]
        L.nfo_step.restype = ctypes.c_double
        L.nfo_vector_interp.argtypes = [dp, ctypes.c_long, dp, ctypes.c_long, ctypes.c_double, ctypes.c_double, dp, dp,
                                        ctypes.POINTER(ctypes.c_long)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c64(a):
    return numpy.ascontiguousarray(a, dtype=numpy.float64)


# ---------------------------------------------------------------- C restatement wrappers
def assemble_points(bounds_lon, bounds_lat):
    """A1 horizgrid.py:17-22."""
    blon, blat = _c64(bounds_lon), _c64(bounds_lat)
    ncell = blon.size // 4
    pts = numpy.empty((ncell, 4, 3), numpy.float64)
    lib().nfo_assemble_points(_dp(blon), _dp(blat), ncell, _dp(pts))
    return pts


def arc_lengths(points):
    """A3 field.py:170-181."""
    pts = _c64(points)
    ncell = pts.shape[0]
    arc = numpy.empty((ncell, 4), numpy.float64)
    lib().nfo_arc_lengths(_dp(pts), ncell, _dp(arc))
    return arc


def vertical_integral(f, thickness, fill=numpy.nan):
    """A4 field.py:145-163.  f: (nz, ny, nx) float64 or float32."""
    f = numpy.ascontiguousarray(f)
    assert f.dtype in (numpy.float64, numpy.float32)
    nz = f.shape[0]
    ncell = f.size // nz
    th = _c64(thickness)
    out = numpy.empty(f.shape[1:], numpy.float64)
    lib().nfo_vertical_integral(f.ctypes.data, int(f.dtype == numpy.float32), nz, ncell, _dp(th), float(fill),
                                _dp(out))
    return out


class EdgeFluxState(object):
    """The arrays Field carries across time steps (field.py:59-63)."""

    def __init__(self, ny, nx):
        n = ny * nx
        self.ny, self.nx = ny, nx
        self.integratedVelocity = numpy.zeros((n, 4), numpy.float64)
        self.edgeFluxesU = numpy.zeros((n,), numpy.float64)
        self.edgeFluxesV = numpy.zeros((n,), numpy.float64)
        self.maxAbsFlux = ctypes.c_double(0.0)


def edge_flux(state, uInt, vInt, arc, sverdrup=False):
    """A5 field.py:183-234 (updates state in place, like the reference)."""
    u, v, a = _c64(uInt), _c64(vInt), _c64(arc)
    lib().nfo_edge_flux(_dp(u), _dp(v), _dp(a), state.ny, state.nx, int(bool(sverdrup)),
                        _dp(state.integratedVelocity), _dp(state.edgeFluxesU), _dp(state.edgeFluxesV),
                        ctypes.cast(ctypes.pointer(state.maxAbsFlux), ctypes.POINTER(ctypes.c_double)))
    return state


class Weights(object):
    def __init__(self, cell_edge, weight, seg, nseg, coverage=None):
        self.cell_edge, self.weight, self.seg, self.nseg = cell_edge, weight, seg, nseg
        self.coverage = coverage    # fraction of every target segment found inside cells of the grid

    def as_dict(self):
        d = {}
        for ce, w, s in zip(self.cell_edge.tolist(), self.weight.tolist(), self.seg.tolist()):
            d[(s, ce)] = d.get((s, ce), 0.0) + w
        return d


class UnsupportedCell(ValueError):
    """A target segment overlaps a cell the algorithm is not defined on (non-convex quad / no inverse bilinear map)."""

    def __init__(self, kind, cell, seg):
        self.kind, self.cell, self.seg = int(kind), int(cell), int(seg)
        what = 'overlaps non-convex cell' if kind == 1 else 'inverse bilinear map did not converge in cell'
        super().__init__(f'target segment {seg} {what} {cell}')


class OverCovered(ValueError):
    """A target segment was found in overlapping cells over part of its length (coverage > 1): it would be counted twice."""

    def __init__(self, seg, coverage):
        self.seg, self.coverage = int(seg), coverage
        super().__init__(f'target segment {seg} is covered {coverage[seg]:.9g} times by the cells of the grid (overlapping cells)')


def polyline_weights(points, xyz, periodX=360., counterclock=False, skip_unsupported=False):
    """A6 mint.PolylineIntegral.computeWeights (field.py:45-48).  Raises UnsupportedCell instead of returning numbers
    for a line that crosses a non-convex cell; skip_unsupported=True drops such cells instead (coverage < 1)."""
    pts = _c64(points)
    ncell = pts.shape[0]
    xyz = _c64(xyz).reshape(-1, 3)
    cap = 4096
    while True:
        ce = numpy.empty(cap, numpy.int64)
        w = numpy.empty(cap, numpy.float64)
        sg = numpy.empty(cap, numpy.int32)
        status = (ctypes.c_long * 3)(0, -1, -1)
        cov = numpy.zeros(max(xyz.shape[0] - 1, 1), numpy.float64)
        n = lib().nfo_polyline_weights(_dp(pts), ncell, _dp(xyz), xyz.shape[0], float(periodX), int(counterclock),
                                       cap, ce.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _dp(w),
                                       sg.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), status, _dp(cov),
                                       1 if skip_unsupported else 0)
        if status[0] == 3:
            raise OverCovered(status[2], cov[:xyz.shape[0] - 1].copy())
        if status[0]:
            raise UnsupportedCell(status[0], status[1], status[2])
        if n >= 0:
            return Weights(ce[:n].copy(), w[:n].copy(), sg[:n].copy(), xyz.shape[0] - 1, cov[:xyz.shape[0] - 1])
        cap = -n


def get_integral(weights, data, with_segments=False):
    """A7 mint.PolylineIntegral.getIntegral(data, CELL_BY_CELL_DATA) (field.py:102)."""
    d = _c64(data).reshape(-1)
    segt = numpy.zeros(max(weights.nseg, 1), numpy.float64)
    tot = lib().nfo_get_integral(_dp(d), weights.weight.size,
                                 weights.cell_edge.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                 _dp(weights.weight), weights.seg.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                 weights.nseg, _dp(segt))
    return (tot, segt[:weights.nseg]) if with_segments else tot


def vector_interp(points, targets, data, periodX=360., tol2=1.e-12):
    """mint.VectorInterp.findPoints + getFaceVectors (field.py:90-95); returns (vectors (npts,3), cell ids)."""
    pts, tg, d = _c64(points), _c64(targets).reshape(-1, 3), _c64(data).reshape(-1, 4)
    vec = numpy.zeros((tg.shape[0], 3), numpy.float64)
    ids = numpy.zeros(tg.shape[0], numpy.int64)
    lib().nfo_vector_interp(_dp(pts), pts.shape[0], _dp(tg), tg.shape[0], float(periodX), float(tol2), _dp(d), _dp(vec),
                            ids.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))
    return vec, ids


# ---------------------------------------------------------------- numpy restatements, as written
def np_lonLat2XYZArray(p, radius=1.0):
    """geo.py:14-22."""
    lam = p[..., 0] * DEG2RAD
    the = p[..., 1] * DEG2RAD
    rho = radius * numpy.cos(the)
    xyz = numpy.zeros(p.shape, numpy.float64)
    xyz[..., 0] = rho * numpy.cos(lam)
    xyz[..., 1] = rho * numpy.sin(lam)
    xyz[..., 2] = radius * numpy.sin(the)
    return xyz


def np_getArcLengthArray(xyzA, xyzB, radius=1.0):
    """geo.py:24-27."""
    with numpy.errstate(invalid='ignore'):
        angle = numpy.arccos(numpy.sum(xyzA * xyzB, axis=-1) / (radius * radius))
    return numpy.fabs(radius * angle)


def np_arc_lengths(points):
    """field.py:170-181."""
    xyz = np_lonLat2XYZArray(points, radius=1.0)
    arc = numpy.zeros((points.shape[0], 4), numpy.float64)
    for i0 in range(4):
        i1 = (i0 + 1) % 4
        arc[:, i0] = np_getArcLengthArray(xyz[:, i0, :], xyz[:, i1, :], radius=1.0)
    return arc


def np_read_field(f, thickness, fill=1.e20):
    """field.py:157,161 (xarray turns _FillValue into NaN; fillna(0.0); tensordot over z)."""
    f = numpy.where(numpy.isnan(f) | (f == fill), 0.0, f)
    return numpy.tensordot(thickness, f, axes=(0, 0))


def np_edge_flux(state, uInt, vInt, arc, sverdrup=False):
    """field.py:183-234, statement by statement (state: EdgeFluxState; maxAbsFlux kept as float)."""
    ny, nx = state.ny, state.nx
    n = ny * nx
    state.edgeFluxesU[:] = + uInt.reshape((n,)) * arc[:, 1]
    state.edgeFluxesV[:] = - vInt.reshape((n,)) * arc[:, 2]
    state.integratedVelocity[:, 1] = state.edgeFluxesU
    state.integratedVelocity[:, 2] = state.edgeFluxesV
    eU = state.edgeFluxesU.reshape((ny, nx))
    eV = state.edgeFluxesV.reshape((ny, nx))
    iV = state.integratedVelocity.reshape((ny, nx, 4))
    iV[1:, :, 0] = eV[:-1, :]
    iV[:, 1:, 3] = eU[:, :-1]
    iV[:, 0, 3] = eU[:, -1]
    if sverdrup:
        eU *= EARTH_RADIUS_SV / 1.e6
        eV *= EARTH_RADIUS_SV / 1.e6
        iV *= EARTH_RADIUS_SV / 1.e6
    state.edgeFluxesU[:] = numpy.fabs(state.edgeFluxesU)
    state.edgeFluxesV[:] = numpy.fabs(state.edgeFluxesV)
    m = max(state.maxAbsFlux.value, state.edgeFluxesU.max(), state.edgeFluxesV.max())
    state.maxAbsFlux = ctypes.c_double(m)
    return state


class DataGen(object):
    """Vectorised restatement of datagen.DataGen (datagen.py:11-166); same attributes."""

    def __init__(self, nx, ny, nz, nt, xmin=-180., xmax=180., ymin=-90., ymax=90., zmin=0., zmax=1.,
                 lat_uses_dx=True):
        self.nx, self.ny, self.nz, self.nt = nx, ny, nz, nt
        self.xmin, self.xmax, self.ymin, self.ymax, self.zmin, self.zmax = xmin, xmax, ymin, ymax, zmin, zmax
        dz = (zmax - zmin) / float(nz)
        k = numpy.arange(nz)
        self.zhalf = zmin + (k + 0.5) * dz  # datagen.py:38
        self.ztop = zmin + (k + 1) * dz     # :39
        self.zbot = zmin + (k + 2) * dz     # :40
        dy, dx = (ymax - ymin) / float(ny), (xmax - xmin) / float(nx)  # :45
        x = xmin + numpy.arange(nx + 1) * dx                          # :48
        y = ymin + numpy.arange(ny + 1) * (dx if lat_uses_dx else dy)  # :49 (the reference uses dx)
        self.xx, self.yy = numpy.meshgrid(x, y, indexing='xy')
        self.bounds_lon = numpy.zeros((ny, nx, 4), numpy.float64)
        self.bounds_lat = numpy.zeros((ny, nx, 4), numpy.float64)
        for v, (sj, si) in enumerate([(slice(None, -1), slice(None, -1)), (slice(None, -1), slice(1, None)),
                                      (slice(1, None), slice(1, None)), (slice(1, None), slice(None, -1))]):
            self.bounds_lon[..., v] = self.xx[sj, si]  # :56-66
            self.bounds_lat[..., v] = self.yy[sj, si]

    def rotatePole(self, deltaDeg):
        """datagen.py:116-166, vectorised (same operation order per element)."""
        alpha = numpy.pi * deltaDeg[1] / 180.
        beta = numpy.pi * deltaDeg[0] / 180.
        ca, sa, cb, sb = numpy.cos(alpha), numpy.sin(alpha), numpy.cos(beta), numpy.sin(beta)
        rot_alp = numpy.array([[ca, 0., sa], [0., 1., 0.], [-sa, 0., ca]])
        rot_bet = numpy.array([[cb, sb, 0.], [-sb, cb, 0.], [0., 0., 1.]])
        M = numpy.dot(rot_bet, rot_alp)
        the = numpy.pi * self.bounds_lat / 180.
        lam = numpy.pi * self.bounds_lon / 180.
        rho = numpy.cos(the)
        xo, yo, zo = rho * numpy.cos(lam), rho * numpy.sin(lam), numpy.sin(the)
        xn = M[0, 0] * xo + M[0, 1] * yo + M[0, 2] * zo
        yn = M[1, 0] * xo + M[1, 1] * yo + M[1, 2] * zo
        zn = M[2, 0] * xo + M[2, 1] * yo + M[2, 2] * zo
        self.bounds_lat = 180. * numpy.arcsin(numpy.clip(zn, -1., 1.)) / numpy.pi
        lon = 180. * numpy.arctan2(yn, xn) / numpy.pi
        for v in range(1, 4):  # date line fix relative to vertex 0 (:162-166)
            d = lon[..., v] - lon[..., 0]
            lon[..., v] = numpy.where(d > 270., lon[..., v] - 360., numpy.where(d < -270., lon[..., v] + 360., lon[..., v]))
        self.bounds_lon = lon

    def potentialAtNodes(self, streamFunction, t, k):
        """datagen.py:74-78: pot = eval(streamFunction) with x, y = node lon/lat, z = zhalf[k]."""
        x, y, z, nt = self.xx, self.yy, self.zhalf[k], self.nt  # noqa: F841
        zmin, zmax = self.zmin, self.zmax  # noqa: F841
        with numpy.errstate(all='ignore'):
            return eval(streamFunction) * numpy.ones_like(self.xx)

    def computeUV(self, streamFunction):
        """datagen.py:69-113."""
        def xyz(xs, ys):
            p = numpy.zeros(xs.shape + (3,), numpy.float64)
            p[..., 0] = xs
            p[..., 1] = ys
            return np_lonLat2XYZArray(p, 1.0)
        xyz1 = xyz(self.xx[:-1, 1:], self.yy[:-1, 1:])
        xyz2 = xyz(self.xx[1:, 1:], self.yy[1:, 1:])
        xyz3 = xyz(self.xx[1:, :-1], self.yy[1:, :-1])
        ds21 = np_getArcLengthArray(xyz2, xyz1, 1.0)
        ds23 = np_getArcLengthArray(xyz2, xyz3, 1.0)
        numpy.clip(ds23, a_min=1.e-12, a_max=None, out=ds23)  # :104
        self.u = numpy.zeros((self.nt, self.nz, self.ny, self.nx), numpy.float64)
        self.v = numpy.zeros((self.nt, self.nz, self.ny, self.nx), numpy.float64)
        for t in range(self.nt):
            for k in range(self.nz):
                pot = self.potentialAtNodes(streamFunction, t, k)
                with numpy.errstate(all='ignore'):
                    self.u[t, k] = (pot[1:, 1:] - pot[:-1, 1:]) / ds21   # :107,110
                    self.v[t, k] = -(pot[1:, 1:] - pot[1:, :-1]) / ds23  # :108,113
        return self.u, self.v

    @property
    def thickness(self):
        return self.zbot - self.ztop  # field.py:51 applied to datagen.py:179-180


def fluxexact(potentialFunction, lonLatPoints, nz, nt, zmin=0., zmax=1.):
    """fluxexact.py:21-46 -> list of nt exact fluxes (arctan2 added to the eval namespace)."""
    xy = numpy.array(lonLatPoints, dtype=numpy.float64)
    dz = (zmax - zmin) / float(nz)
    zhalf = numpy.array([zmin + (k + 0.5) * dz for k in range(nz)])
    ztop = numpy.array([zmin + (k + 0) * dz for k in range(nz)])
    zbot = numpy.array([zmin + (k + 1) * dz for k in range(nz)])
    thickness = -(ztop - zbot)
    out = []
    for t in range(nt):
        flux = 0
        for k in range(nz):
            z = zhalf[k]  # noqa: F841
            x, y = xy[0, :2]
            phiA = eval(potentialFunction)
            x, y = xy[-1, :2]
            phiB = eval(potentialFunction)
            flux += (phiB - phiA) * thickness[k]
        out.append(float(flux))
    return out
