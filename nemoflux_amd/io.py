"""File access for the Field/HorizGrid constructors (nemoflux/field.py:22-25,34-35).

The reference reads NetCDF through xarray.  This image has neither xarray nor netCDF4, so:
  * NetCDF files are read through xarray IF it is importable (same variable names: bounds_lat,
    bounds_lon, deptht_bounds, uo, vo; _FillValue decoded by hand), and
  * `.npz` bundles with the same variable names are always accepted (nemoflux_amd.datagen.DataGen.save
    writes them): <prefix>T.npz, <prefix>U.npz, <prefix>V.npz.
NetCDF/HDF5 ingest straight to HBM is SURVEY.md 8f rank 3 ("next").
"""
import numpy


def _open(path):
    path = str(path)
    if path.endswith('.npz'):
        return dict(numpy.load(path, allow_pickle=False))
    try:
        import xarray
    except ImportError as e:
        raise RuntimeError(f'ERROR: cannot read {path}: xarray/netCDF4 are not installed; '
                           'use the .npz bundles written by nemoflux_amd.datagen') from e
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


def open_uvfile(path, name):
    d = _open(path)
    if name not in d:
        raise RuntimeError(f'ERROR: could not read {name} field')  # field.py:154
    fill = d.get('_FillValue_' + name, numpy.array(numpy.nan))
    return d[name], float(fill)
