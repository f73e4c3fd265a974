#!/usr/bin/env python
"""NetCDF-4 (HDF5) -> .npz converter for the variables nemoflux reads (field.py:22-25,34-35,149).

Runs under ANY interpreter that has h5py + numpy (this image: /opt/conda/bin/python3.9); nemoflux_amd/io.py calls it
as a child process when xarray/netCDF4 are not importable.  Every dataset whose name nemoflux uses is copied with its
_FillValue (as _FillValue_<name>); nothing is decoded or scaled here (missing values are handled on the GPU).

    python nc2npz.py in.nc out.npz
"""
import sys

import h5py
import numpy

WANTED = ('bounds_lat', 'bounds_lon', 'deptht_bounds', 'uo', 'vo', 'time_counter', 'time_instant', 'time_centered')


def convert(src, dst):
    out = {}
    with h5py.File(src, 'r') as f:
        for name in f:
            ds = f[name]
            if not hasattr(ds, 'shape') or name not in WANTED:
                continue
            out[name] = ds[...]
            fv = ds.attrs.get('_FillValue', None)
            if fv is not None:
                out['_FillValue_' + name] = numpy.asarray(fv).reshape(-1)[0]
    numpy.savez(dst, **out)
    return sorted(out)


if __name__ == '__main__':
    print(' '.join(convert(sys.argv[1], sys.argv[2])))
