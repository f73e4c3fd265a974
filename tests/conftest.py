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


FULL_CASES = ['c1_x', 'singular', 'cossin36', 'rot36_zt', 'def36_zt', 'sv36_land', 'reg16']


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
