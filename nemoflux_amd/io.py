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
import warnings

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


# ---------------------------------------------------------------------------------------------- CF decoding
# xarray.open_dataset (field.py:22-25, 34-35) decodes CF conventions by default (decode_cf / mask_and_scale): values equal to
# _FillValue OR missing_value become NaN (both, when both are given), then packed data are unpacked as
# raw * scale_factor + add_offset in a float type.  The attributes travel with the variables as '_<attr>_<name>' entries.
CF_ATTRS = ('_FillValue', 'missing_value', 'scale_factor', 'add_offset')


def _keep_cf(out, name, attrs):
    for k in CF_ATTRS:
        v = attrs.get(k)
        if v is None:
            continue
        v = numpy.asarray(v)
        if v.dtype.kind not in 'fiu' or v.size == 0:
            continue
        key = '_FillValue_' + name if k == '_FillValue' else f'_{k}_{name}'
        out[key] = v.reshape(-1) if (k == 'missing_value' and v.size > 1) else v.reshape(-1)[0]


def cf_markers(d, name):
    """Raw values of `name` that mean 'missing' (the variable's _FillValue and missing_value entries, duplicates merged)."""
    vals = []
    for key in ('_FillValue_' + name, '_missing_value_' + name):
        if key in d:
            for x in numpy.asarray(d[key]).reshape(-1):
                if not any(x == y or (x != x and y != y) for y in vals):
                    vals.append(x)
    return vals


def cf_float_dtype(raw_dtype, has_offset):
    """The float type xarray decodes into [recall of xarray.coding.variables._choose_float_dtype; xarray is not installed,
    parity unpinned]: float32 stays; integers of <= 16 bits without add_offset -> float32 (a scale factor vanishes into the
    mantissa, a large offset may not); everything else float64."""
    raw_dtype = numpy.dtype(raw_dtype)
    if raw_dtype.kind == 'f':
        return numpy.dtype(numpy.float32) if raw_dtype.itemsize <= 4 else numpy.dtype(numpy.float64)
    if raw_dtype.itemsize <= 2 and not has_offset:
        return numpy.dtype(numpy.float32)
    return numpy.dtype(numpy.float64)


def cf_is_packed(d, name, dtype):
    return numpy.dtype(dtype).kind in 'iu' or ('_scale_factor_' + name) in d or ('_add_offset_' + name) in d


def cf_decode_array(raw, markers, scale, offset, out_dtype, out=None):
    """mask (in the raw dtype), then unpack: the order xarray's coders run in when decoding"""
    raw = numpy.asarray(raw)
    if out is None:
        out = numpy.empty(raw.shape, out_dtype)
    numpy.copyto(out, raw, casting='unsafe')
    if scale is not None:
        out *= out.dtype.type(scale)
    if offset is not None:
        out += out.dtype.type(offset)
    for m in markers:
        if m == m:
            out[raw == numpy.asarray(m).astype(raw.dtype)] = numpy.nan
    return out


class CFDecodedVariable(object):
    """uo / vo stored packed (integers, scale_factor / add_offset): decoded one time step at a time on the host, the way
    xarray hands them to the reference -- masked values are NaN, the rest raw * scale_factor + add_offset.  Same surface as
    hdf5min.LazyVariable (shape, dtype, read_step) minus device_plan: packed integers take the host staging path."""

    def __init__(self, src, markers, scale, offset):
        self._src, self.shape = src, tuple(src.shape)
        self._markers, self._scale, self._offset = list(markers), scale, offset
        self.dtype = cf_float_dtype(src.dtype, offset is not None)

    def _raw_step(self, t):
        if hasattr(self._src, 'read_step'):
            return self._src.read_step(t)
        return self._src[t] if len(self.shape) == 4 else self._src

    def read_step(self, t, out=None):
        return cf_decode_array(self._raw_step(t), self._markers, self._scale, self._offset, self.dtype, out)


def _cf_apply_whole(d, name):
    """decode a small variable (bounds_lon / bounds_lat / deptht_bounds) in full, if the file encodes it"""
    if name not in d:
        return
    a = d[name]
    markers = cf_markers(d, name)
    if not cf_is_packed(d, name, a.dtype) and not markers:
        return
    if numpy.dtype(a.dtype).kind in 'iu' or cf_is_packed(d, name, a.dtype):
        dt = cf_float_dtype(a.dtype, ('_add_offset_' + name) in d)
    else:
        dt = numpy.dtype(a.dtype).newbyteorder('=')
    d[name] = cf_decode_array(a, markers, d.get('_scale_factor_' + name), d.get('_add_offset_' + name), dt)


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
        _keep_cf(out, name, ds.attrs)
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
        _keep_cf(out, name, attrs)
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
            _keep_cf(out, k, dict(nc[k].attrs))
    return out


def open_tfile(path):
    d = _open(path)
    for k in ('bounds_lat', 'bounds_lon'):
        if k not in d:
            raise RuntimeError(f'ERROR: {path} has no variable {k}')
    for k in ('bounds_lat', 'bounds_lon', 'deptht_bounds'):
        _cf_apply_whole(d, k)
    return d


def open_uvfile(path, name, with_all=False):
    """(variable, fill value as a float (NaN = none), [all variables of the file]).  The dict also carries
    '_markers_<name>': every value that means 'missing' (the _FillValue first, then a missing_value that differs from it).
    A packed variable (integers and / or scale_factor / add_offset) comes back as a CFDecodedVariable whose masked values
    are NaN (no markers left)."""
    d = _open(path, lazy=(name,))
    if name not in d:
        raise RuntimeError(f'ERROR: could not read {name} field')  # field.py:154
    var = d[name]
    markers = cf_markers(d, name)
    if cf_is_packed(d, name, var.dtype):
        var = CFDecodedVariable(var, markers, d.get('_scale_factor_' + name), d.get('_add_offset_' + name))
        markers = []
    markers = [float(m) for m in markers if m == m]       # NaN is always missing (field.py:157 fillna)
    d['_markers_' + name] = markers
    fill = markers[0] if markers else float('nan')
    return (var, fill, d) if with_all else (var, fill)
