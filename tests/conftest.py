import ast
import json
import os
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a fresh checkout has no built artefacts (they are git-ignored): build the HIP library and the oracle once
    if not os.path.exists(os.path.join(ROOT, 'nemoflux_amd', 'libnemoflux_amd.so')):
        import __graft_entry__
        __graft_entry__.build()


def load_cases():
    with open(os.path.join(GOLDEN, 'cases.json')) as f:
        return json.load(f)


def load_golden(name):
    return numpy.load(os.path.join(GOLDEN, name + '.npz'))


def transect_xyz(points_str):
    xy = numpy.array(ast.literal_eval(points_str), dtype=numpy.float64)
    xyz = numpy.zeros((xy.shape[0], 3), numpy.float64)
    xyz[:, :2] = xy
    return xyz


@pytest.fixture(scope='session')
def oracle():
    import nf_oracle
    nf_oracle.build()
    return nf_oracle


@pytest.fixture(scope='session')
def cases():
    return load_cases()


FULL_CASES = ['c1_x', 'singular', 'cossin36', 'rot36_zt', 'def36_zt', 'sv36_land', 'reg16', 'wrap36_zt']


def case_box(m):
    """(xmin, xmax, ymin, ymax, zmin, zmax) of a golden case; the global box unless the case records another."""
    return tuple(m.get('box', (-180., 180., -90., 90., 0., 1.)))


def write_classic_triple(dirname, g, version=2):
    """T/U/V NetCDF-3 (classic / 64-bit-offset) files from a golden case, written by scipy.io.netcdf_file -- an
    independent implementation of the format -- the way IOIPSL-era NEMO wrote them: float32, big-endian, uo/vo record
    variables over an unlimited time axis, land as _FillValue (uo) and NaN (vo)."""
    from scipy.io import netcdf_file
    u, v = g['u'].astype(numpy.float32), g['v'].astype(numpy.float32)
    u[:, :, 4:9, 10:20] = numpy.float32(1.e20)
    v[:, :, 4:9, 10:20] = numpy.nan
    nt, nz, ny, nx = u.shape
    paths = {k: os.path.join(str(dirname), f'{k}.nc') for k in 'TUV'}
    f = netcdf_file(paths['T'], 'w', version=version)
    for n, s in (('y', ny), ('x', nx), ('nvertex', 4), ('deptht', nz), ('axis_nbounds', 2)):
        f.createDimension(n, s)
    for name in ('bounds_lon', 'bounds_lat'):
        f.createVariable(name, 'f4', ('y', 'x', 'nvertex'))[:] = g[name].astype(numpy.float32)
    f.createVariable('deptht_bounds', 'f4', ('deptht', 'axis_nbounds'))[:] = g['deptht_bounds'].astype(numpy.float32)
    f.close()
    for k, name, a in (('U', 'uo', u), ('V', 'vo', v)):
        f = netcdf_file(paths[k], 'w', version=version)
        for n, s in (('time_counter', None), ('depth', nz), ('y', ny), ('x', nx)):
            f.createDimension(n, s)
        tc = f.createVariable('time_counter', 'f8', ('time_counter',))
        tc.standard_name, tc.units, tc.calendar = 'time', 'seconds since 1900-01-01 00:00:00', 'noleap'
        var = f.createVariable(name, 'f4', ('time_counter', 'depth', 'y', 'x'))
        if k == 'U':
            var._FillValue = numpy.float32(1.e20)
        tc[:] = numpy.arange(nt) * 86400. * 30.5 + 1296000.
        var[:] = a
        f.close()
    return paths, u, v


def exact_segment_fluxes(psi, polys, nz, nt, zmin=0., zmax=1.):
    """Closed form per target SEGMENT: sum_k [psi(node s+1) - psi(node s)] dz_k for every time step --
    fluxexact.py:36-46 applied to each piece of each polyline, vectorised over levels and vertices with the same
    expression evaluator nemoflux_amd.fluxexact uses.  Returns one (nt, npts-1) array per polyline."""
    from nemoflux_amd import _expr
    code = _expr.compile_function(psi)
    dz = (zmax - zmin) / float(nz)
    zhalf = zmin + (numpy.arange(nz) + 0.5) * dz
    out = []
    for pts in polys:
        xy = numpy.array(pts, dtype=numpy.float64)
        seg = numpy.zeros((nt, xy.shape[0] - 1))
        for t in range(nt):
            phi = _expr.evaluate(code, x=xy[:, 0][:, None], y=xy[:, 1][:, None], z=zhalf[None, :], t=t, nt=nt)
            phi = phi + numpy.zeros((xy.shape[0], nz))
            seg[t] = ((phi[1:] - phi[:-1]) * dz).sum(axis=1)
        out.append(seg)
    return out


def deflated_dataset(a, name, chunk, level=4, shuffle=True, attrs=None, threads=None):
    """An in-memory NetCDF-4-style chunked variable (no HDF5 writer exists in this image): `a` (nt, nz, ny, nx) cut into
    chunks of shape `chunk` = (1, cz, cy, cx) -- edge chunks are stored whole, as HDF5 does, with arbitrary bytes beyond the
    edge -- each byte-shuffled and deflated exactly as the HDF5 filter pipeline stores it, behind a real
    nemoflux_amd.hdf5min.Dataset: the reader code that runs (chunk selection, device plan, gather) is the one files go
    through; only the metadata parsing is skipped.  Returns (LazyVariable, compressed bytes)."""
    import concurrent.futures
    import zlib
    from nemoflux_amd import hdf5min
    nt, nz, ny, nx = a.shape
    _, cz, cy, cx = chunk
    es = a.dtype.itemsize
    keys = [(t, z0, y0, x0) for t in range(nt) for z0 in range(0, nz, cz) for y0 in range(0, ny, cy) for x0 in range(0, nx, cx)]

    def pack(key):
        t, z0, y0, x0 = key
        sub = a[t, z0:z0 + cz, y0:y0 + cy, x0:x0 + cx]
        if sub.shape != (cz, cy, cx):
            blk = numpy.full((cz, cy, cx), 7.25, a.dtype)             # what lies beyond the edge is arbitrary
            blk[:sub.shape[0], :sub.shape[1], :sub.shape[2]] = sub
        else:
            blk = numpy.ascontiguousarray(sub)
        raw = blk.view(numpy.uint8).reshape(-1, es)
        return zlib.compress(numpy.ascontiguousarray(raw.T).tobytes() if shuffle else raw.tobytes(), level)
    with concurrent.futures.ThreadPoolExecutor(threads or hdf5min.io_threads()) as pool:
        blobs = list(pool.map(pack, keys))
    chunks, off = [], 0
    for key, b in zip(keys, blobs):
        chunks.append((key + (0,), len(b), 0, off))
        off += len(b)

    class MemFile(object):          # what hdf5min.Dataset needs of its File: the mapped bytes and the base address
        def __init__(self, blob):
            self._m, self._base = blob, 0
    filters = ([(2, [es])] if shuffle else []) + [(1, [level])]
    ds = hdf5min.Dataset(MemFile(b''.join(blobs)), name, a.shape, numpy.dtype(a.dtype).newbyteorder('<'),
                         ('chunked', None, (1, cz, cy, cx, es), None), filters, dict(attrs or {}))
    ds._chunks = chunks
    return hdf5min.LazyVariable(ds), off


# ---- date-line-wrapped cell bounds (round 4) ----------------------------------------------------------------------------
def wrap180(lon):
    """Longitudes the way a global NEMO T-file stores them: every corner on its own wrapped into [-180, 180)."""
    return (numpy.asarray(lon, numpy.float64) + 180.) % 360. - 180.


def wrapped_grid_case(oracle, kind, seed):
    """One geometry with its bounds_lon (a) as one continuous branch and (b) wrapped per corner into [-180, 180), so that the
    cells straddling the cut have corners ~355 degrees apart (what horizgrid.py:17-24 hands to mint for a real global file).
    kind: 'g0' = global 72 x 36 on [0, 360]; 'g73' = global on [73, 433] (ORCA's start longitude); 'sa150' = the real
    ORCA025 subset of data/sa/T.nc moved 150 degrees east (it then straddles the date line).
    Returns (points_plain, points_wrapped, data, transects, exact): data = cell-by-cell edge differences of a random node
    stream function (x-periodic on the global grids), transects = 20 seeded node-to-node polylines with free interior
    vertices, some crossing 180 E, exact[k] = psi(end) - psi(start) or None where the geometry has no shared nodes."""
    rng = numpy.random.default_rng(seed)
    if kind == 'sa150':
        b = load_golden('sa_T_bounds')
        blon = b['bounds_lon'].astype(numpy.float64) + 150.
        blat = b['bounds_lat'].astype(numpy.float64)
        ny, nx = blon.shape[:2]
        pts = oracle.assemble_points(blon, blat)
        f = [pts[:, :, 0], pts[:, :, 1]]     # planar lon / lat are bilinear per cell: their "flux" is the end-point difference
        data = [numpy.stack([g[:, 1] - g[:, 0], g[:, 2] - g[:, 1], g[:, 2] - g[:, 3], g[:, 3] - g[:, 0]], axis=1) for g in f]
        transects, exact = [], []
        for k in range(20):
            n = int(rng.integers(2, 7))
            x, y = rng.uniform(165., 185., n), rng.uniform(-39., -23., n)
            if k % 4 == 0:
                x[0], x[-1] = 171.3, 184.2        # certainly across 180 E
            if k % 4 == 1:
                x = wrap180(x)                       # the caller's own longitudes wrapped: pieces go the long way round
            xyz = numpy.zeros((n, 3))
            xyz[:, 0], xyz[:, 1] = x, y
            transects.append(xyz)
            exact.append(None)
        wr = pts.copy()
        wr[:, :, 0] = wrap180(pts[:, :, 0])
        return pts, wr, data, transects, exact
    x0 = {'g0': 0., 'g73': 73.}[kind]
    nx, ny = 72, 36
    o = oracle.DataGen(nx, ny, 1, 1, xmin=x0, xmax=x0 + 360.)
    pts = oracle.assemble_points(o.bounds_lon, o.bounds_lat)
    psi = rng.standard_normal((ny + 1, nx + 1))
    psi[:, -1] = psi[:, 0]
    p0, p1, p2, p3 = psi[:-1, :-1], psi[:-1, 1:], psi[1:, 1:], psi[1:, :-1]
    data = [numpy.stack([p1 - p0, p2 - p1, p2 - p3, p3 - p0], axis=-1).reshape(-1, 4)]
    xn, yn = o.xx[0], o.yy[:, 0]
    transects, exact = [], []
    for k in range(20):
        ia, ja, ib, jb = rng.integers(0, nx + 1), rng.integers(1, ny), rng.integers(0, nx + 1), rng.integers(1, ny)
        n = int(rng.integers(0, 5))
        x = numpy.concatenate([[xn[ia]], rng.uniform(x0, x0 + 360., n), [xn[ib]]])
        y = numpy.concatenate([[yn[ja]], rng.uniform(-80., 80., n), [yn[jb]]])
        if k % 4 == 0:      # straight across 180 E, end points on nodes either side of it
            ia, ib = int(numpy.argmin(abs(xn - 150.))), int(numpy.argmin(abs(xn - 215.)))
            x = numpy.array([xn[ia], xn[ib]])
            y = numpy.array([yn[ja], yn[jb]])
        if k % 4 == 1:      # the caller's longitudes in [-180, 180): the same nodes, pieces may go the long way round
            x = wrap180(x)
        if k % 4 == 2:      # the whole line moved by a period
            x = x - 360.
        xyz = numpy.zeros((x.size, 3))
        xyz[:, 0], xyz[:, 1] = x, y
        transects.append(xyz)
        exact.append(psi[jb, ib] - psi[ja, ia])
    wr = pts.copy()
    wr[:, :, 0] = wrap180(pts[:, :, 0])
    return pts, wr, data, transects, exact


DATELINE_LINES = ["(180,-60),(180,40)", "(-180,-60),(-180,40)", "(175,-60),(180,-60),(180,40),(185,40)", "(0,-60),(0,40)",
                  "(360,-60),(360,40)", "(170,-55),(190,-55)", "(170,-55),(-170,-55)", "(180,-60),(180,-60),(185,-55)"]


def orca_like_halo_grid(oracle, nx=72, ny=36, x0=73.):
    """A global grid that starts at 73 E like ORCA and carries two halo columns in front -- exact duplicates of its last two
    columns, one period to the west -- the way NEMO files before 4.2 store the east-west wrap.  Returns (points on continuous
    branches, the same with every corner's longitude wrapped into [-180, 180), node psi (ny+1, nx+1) periodic, and a function
    giving the cell-by-cell edge data of psi for this (ny, nx+2) layout)."""
    o = oracle.DataGen(nx, ny, 1, 1, xmin=x0, xmax=x0 + 360.)
    blon = numpy.ascontiguousarray(numpy.concatenate([o.bounds_lon[:, -2:] - 360., o.bounds_lon], axis=1))
    blat = numpy.ascontiguousarray(numpy.concatenate([o.bounds_lat[:, -2:], o.bounds_lat], axis=1))
    pts = oracle.assemble_points(blon, blat)
    wr = pts.copy()
    wr[:, :, 0] = wrap180(pts[:, :, 0])
    rng = numpy.random.default_rng(77)
    psi = rng.standard_normal((ny + 1, nx + 1))
    psi[:, -1] = psi[:, 0]
    p0, p1, p2, p3 = psi[:-1, :-1], psi[:-1, 1:], psi[1:, 1:], psi[1:, :-1]
    d = numpy.stack([p1 - p0, p2 - p1, p2 - p3, p3 - p0], axis=-1)           # (ny, nx, 4)
    data = numpy.ascontiguousarray(numpy.concatenate([d[:, -2:], d], axis=1)).reshape(-1, 4)
    return o, pts, wr, psi, data


# ---- irregular global grid + tiny target segments on its edges and nodes (round-5: over-coverage is measured as a length) ----
def irregular_wrapped_grid(oracle, seed=5, nx=72, ny=36):
    """A global mesh whose nodes are jittered by up to 1.2 degrees (cells abut exactly: they share the jittered nodes), seam
    column periodic, longitudes wrapped per corner into [-180, 180).  Returns (node lon, node lat, points (ncell,4,3))."""
    rng = numpy.random.default_rng(seed)
    xx, yy = numpy.meshgrid(numpy.linspace(0., 360., nx + 1), numpy.linspace(-90., 90., ny + 1))
    xx = xx + rng.uniform(-1.2, 1.2, xx.shape)
    yy = yy + rng.uniform(-1.2, 1.2, yy.shape)
    xx[:, -1] = xx[:, 0] + 360.
    yy[:, -1] = yy[:, 0]
    yy[0], yy[-1] = -90., 90.
    blon = numpy.stack([xx[:-1, :-1], xx[:-1, 1:], xx[1:, 1:], xx[1:, :-1]], axis=-1)
    blat = numpy.stack([yy[:-1, :-1], yy[:-1, 1:], yy[1:, 1:], yy[1:, :-1]], axis=-1)
    return xx, yy, oracle.assemble_points(wrap180(blon), blat)


def tiny_segment_lines(xx, yy, half, n, seed):
    """n four-point lines whose middle segment has half-length `half` degrees and is centred on a grid node (even k) or on a
    point of a cell edge (odd k), direction random; the two outer segments are ordinary ones."""
    rng = numpy.random.default_rng(seed)
    out = []
    for k in range(n):
        i, j = int(rng.integers(1, xx.shape[1] - 1)), int(rng.integers(2, xx.shape[0] - 2))
        ang = rng.uniform(0., 2. * numpy.pi)
        if k % 2 == 0:
            cx, cy = xx[j, i], yy[j, i]
        else:
            s = rng.uniform(0.2, 0.8)
            cx, cy = xx[j, i] + s * (xx[j, i + 1] - xx[j, i]), yy[j, i] + s * (yy[j, i + 1] - yy[j, i])
        d = half * numpy.array([numpy.cos(ang), numpy.sin(ang)])
        out.append(numpy.array([[cx - 7.3, cy - 3.1, 0.], [cx - d[0], cy - d[1], 0.], [cx + d[0], cy + d[1], 0.],
                                [cx + 5.2, cy + 6.4, 0.]]))
    return out


# ---- affine stream function: an oracle-free exact answer on curvilinear cells (round 6) ---------------------------------
def affine_edge_data(points, a, b, c=0.):
    """Cell-by-cell edge data of psi = a*lon + b*lat + c on cells `points` (ncell,4,3): nodal differences in the +xi
    orientation mint's counterclock=False uses: S psi1-psi0, E psi2-psi1, N psi2-psi3, W psi3-psi0.
    A bilinear cell reproduces an affine function exactly, so the flux across ANY polyline is a*dlon + b*dlat of its end
    points (end points need not be nodes) and every interpolated face vector is (b, -a, 0) -- with no reference to the
    oracle's or the device's clipping / inverse-bilinear code (field.py:45-48,90-95,102; README.md:45,58)."""
    psi = a * points[:, :, 0] + b * points[:, :, 1] + c
    return numpy.ascontiguousarray(numpy.stack([psi[:, 1] - psi[:, 0], psi[:, 2] - psi[:, 1], psi[:, 2] - psi[:, 3],
                                                psi[:, 3] - psi[:, 0]], axis=1))


def random_open_polylines(seed, n, box, nvert=(2, 7)):
    """n open polylines of 2..6 vertices drawn uniformly in box = (lonmin, lonmax, latmin, latmax): the end points (and
    every vertex) fall inside cells, not on nodes."""
    rng = numpy.random.default_rng(seed)
    out = []
    for _ in range(n):
        m = int(rng.integers(*nvert))
        xyz = numpy.zeros((m, 3))
        xyz[:, 0] = rng.uniform(box[0], box[1], m)
        xyz[:, 1] = rng.uniform(box[2], box[3], m)
        out.append(xyz)
    return out


def affine_expected(xyz, a, b):
    """(per-segment, total) flux of psi = a*lon + b*lat across the polyline xyz."""
    d = numpy.diff(xyz[:, :2], axis=0)
    seg = a * d[:, 0] + b * d[:, 1]
    return seg, float(a * (xyz[-1, 0] - xyz[0, 0]) + b * (xyz[-1, 1] - xyz[0, 1]))


def curl_of_bilinear_case(points, psi_cells, rng, n, ok):
    """Oracle-free expectation for mint.VectorInterp on curvilinear cells with NON-constant data (round 6): n target points
    made FORWARD from chosen (cell, xi, eta) -- p = F(xi, eta), no inverse map involved -- in cells where `ok`, and the vector
    the lowest-order face interpolation of edge data = nodal differences of psi must give there: the curl of the cell's
    bilinear psi_h, (d psi_h / d lat, -d psi_h / d lon, 0), from grad psi_h = DF^-T grad_ref psi_hat (the gradient form; the
    engine and the oracle evaluate the Piola form (psi_eta r_xi - psi_xi r_eta) / J).
    points (ncell,4,3); psi_cells (ncell,4): psi at the four corners of every cell.  Returns cells, xi, eta, targets (n,3), v."""
    cells = rng.choice(numpy.nonzero(ok)[0], n)
    xi, eta = rng.uniform(0.05, 0.95, n), rng.uniform(0.05, 0.95, n)
    P = points[cells][:, :, :2]
    N = numpy.stack([(1 - xi) * (1 - eta), xi * (1 - eta), xi * eta, (1 - xi) * eta], axis=1)
    dNx = numpy.stack([-(1 - eta), (1 - eta), eta, -eta], axis=1)
    dNe = numpy.stack([-(1 - xi), -xi, xi, (1 - xi)], axis=1)
    targets = numpy.zeros((n, 3))
    targets[:, :2] = (N[:, :, None] * P).sum(axis=1)
    rxi, reta = (dNx[:, :, None] * P).sum(axis=1), (dNe[:, :, None] * P).sum(axis=1)
    ps = psi_cells[cells]
    pxi, peta = (dNx * ps).sum(axis=1), (dNe * ps).sum(axis=1)
    J = rxi[:, 0] * reta[:, 1] - rxi[:, 1] * reta[:, 0]
    gx = (reta[:, 1] * pxi - rxi[:, 1] * peta) / J
    gy = (-reta[:, 0] * pxi + rxi[:, 0] * peta) / J
    return cells, xi, eta, targets, numpy.stack([gy, -gx, numpy.zeros(n)], axis=1)
