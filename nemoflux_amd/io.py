"""File access for the Field/HorizGrid constructors (nemoflux/field.py:22-25,34-35).

The reference reads NetCDF through xarray.  This image has neither xarray nor netCDF4 in its main interpreter, so:
  * `.npz` bundles with the same variable names are always accepted (nemoflux_amd.datagen.DataGen.save
    writes them): <prefix>T.npz, <prefix>U.npz, <prefix>V.npz;
  * NetCDF-4 / HDF5 files are parsed in-process by nemoflux_amd/hdf5min.py (numpy + zlib only): contiguous variables
    come back as views of the mapped file, so the engine stages each time step from the page cache straight to HBM;
    chunked / deflated variables (real NEMO output) are inflated one time step at a time (LazyVariable);
  * NetCDF classic / 64-bit-offset files (magic 'CDF\\x01' / 'CDF\\x02': older NEMO output) are opened with
    scipy.io.netcdf_file (memory-mapped, big-endian views; one time step is converted at a time);
  * what hdf5min does not understand falls back to xarray IF it is importable (same variable names: bounds_lat,
    bounds_lon, deptht_bounds, uo, vo; _FillValue kept, not decoded), and last to a one-off conversion by
    tools/nc2npz.py under any interpreter that has h5py (probed: the running one, /opt/conda/bin/python3.9, python3).
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy

from . import hdf5min
from ._lib import lib as _nflib


def _native_unshuffle(src, dst, n, es):
    rc = _nflib.nf_host_unshuffle(src.ctypes.data, dst.ctypes.data, n, es)
    if rc != 0:
        raise RuntimeError(_nflib.nf_last_error().decode('utf-8', 'replace'))


hdf5min.set_unshuffle(_native_unshuffle)     # HDF5 shuffle filter undone by the library's host helper (10x numpy)

_NC2NPZ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'nc2npz.py')
_H5_PYTHONS = [sys.executable, '/opt/conda/bin/python3.9', 'python3']


def _convert_with_h5py(path):
    """Child-process conversion; returns the dict of arrays or None if no interpreter with h5py is found."""
    if not os.path.exists(_NC2NPZ):
        return None
    for py in _H5_PYTHONS:
        with tempfile.TemporaryDirectory() as tmp:
            dst = os.path.join(tmp, 'converted.npz')
            try:
                r = subprocess.run([py, _NC2NPZ, path, dst], capture_output=True, timeout=3600)
            except (OSError, subprocess.TimeoutExpired):
                continue
            if r.returncode == 0 and os.path.exists(dst):
                return dict(numpy.load(dst, allow_pickle=False))
    return None


def _open_hdf5(path, lazy=()):
    """In-process NetCDF-4/HDF5 read; variables named in `lazy` are returned as hdf5min.LazyVariable."""
    out = {}
    f = hdf5min.File(path)
    for name, ds in f.datasets.items():
        if '/' in name:
            continue
        if isinstance(ds, hdf5min.Hdf5Error):
            if name in ('bounds_lat', 'bounds_lon', 'deptht_bounds', 'uo', 'vo'):
                raise ds
            continue
        if ds.dtype.kind not in 'fiu':
            continue
        # only a (t, z, y, x) variable is walked one time step at a time; anything smaller is read whole
        if name in lazy and len(ds.shape) == 4 and not (ds.is_contiguous() and ds.dtype.isnative):
            out[name] = hdf5min.LazyVariable(ds)
        else:
            a = ds.read()
            out[name] = a if a.dtype.isnative else a.astype(a.dtype.newbyteorder('='))
        if ds.fill_value is not None:
            out['_FillValue_' + name] = numpy.asarray(ds.fill_value)
        if len(ds.shape) == 1:     # coordinate variables: keep the CF attributes (time axis labelling, timeobj.py:9-13)
            out['_attrs_' + name] = dict(ds.attrs)
    out['_hdf5_file'] = f      # keeps the mapping alive for the views
    return out


class StepView(object):
    """A (nt, nz, ny, nx) array of a mapped file in the file's byte order, handed out one native-order time step at a
    time (same surface as hdf5min.LazyVariable: shape, dtype, read_step)."""

    def __init__(self, array):
        self._a, self.shape = array, tuple(array.shape)
        self.dtype = numpy.dtype(array.dtype.newbyteorder('='))

    def read_step(self, t, out=None):
        if out is None:
            return numpy.ascontiguousarray(self._a[t], dtype=self.dtype)
        numpy.copyto(out, self._a[t])      # converts the byte order
        return out


def _open_classic(path, lazy=()):
    """NetCDF-3 (CDF-1 / CDF-2) through scipy: variables are strided views of the mapped file, _FillValue kept as is."""
    import warnings
    from scipy.io import netcdf_file
    f = netcdf_file(path, 'r', mmap=True, maskandscale=False)
    out = {}
    for name, var in f.variables.items():
        if var.data.dtype.kind not in 'fiu':
            continue
        a = var.data
        if a.ndim == 4 and name in lazy:
            out[name] = StepView(a)                                  # big-endian on disk: converted step by step
        else:
            out[name] = a if a.dtype.isnative else a.astype(a.dtype.newbyteorder('='))
        attrs = dict(var._attributes)
        if '_FillValue' in attrs:
            out['_FillValue_' + name] = numpy.asarray(attrs['_FillValue'])
        if var.data.ndim == 1:
            out['_attrs_' + name] = attrs
    out['_classic_file'] = f       # keeps the mapping alive for the views
    warnings.filterwarnings('ignore', message='Cannot close a netcdf_file opened with mmap=True')
    return out


def _open(path, lazy=()):
    path = str(path)
    if path.endswith('.npz'):
        d = dict(numpy.load(path, allow_pickle=False))
        for k in [k for k in d if k.startswith('_attrs_')]:   # CF attributes kept as JSON text (subsetnemo.py)
            d[k] = json.loads(str(d[k]))
        return d
    if not os.path.exists(path):
        raise RuntimeError(f'ERROR: cannot read {path}: no such file')
    with open(path, 'rb') as fh:
        magic = fh.read(4)
    if magic in (b'CDF\x01', b'CDF\x02'):
        return _open_classic(path, lazy)
    try:
        return _open_hdf5(path, lazy)
    except hdf5min.Hdf5Error:
        pass
    try:
        import xarray
    except ImportError as e:
        d = _convert_with_h5py(path)
        if d is not None:
            return d
        raise RuntimeError(f'ERROR: cannot read {path}: neither xarray/netCDF4 nor an interpreter with h5py is '
                           'available; use the .npz bundles written by nemoflux_amd.datagen') from e
    out = {}
    with xarray.open_dataset(path, mask_and_scale=False) as nc:
        for k in nc.variables:
            out[k] = numpy.asarray(nc[k].values)
            fv = nc[k].attrs.get('_FillValue', None)
            if fv is not None:
                out['_FillValue_' + k] = numpy.asarray(fv)
    return out


def open_tfile(path):
    d = _open(path)
    for k in ('bounds_lat', 'bounds_lon'):
        if k not in d:
            raise RuntimeError(f'ERROR: {path} has no variable {k}')
    return d


def open_uvfile(path, name, with_all=False):
    d = _open(path, lazy=(name,))
    if name not in d:
        raise RuntimeError(f'ERROR: could not read {name} field')  # field.py:154
    fill = d.get('_FillValue_' + name, numpy.array(numpy.nan))
    return (d[name], float(fill), d) if with_all else (d[name], float(fill))
