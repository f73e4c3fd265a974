"""Cut a (j, i) window out of a NEMO T/U/V file triple: the counterpart of nemoflux/subsetNEMO.py:6-93.

The reference copies through netCDF4 and writes T.nc / U.nc / V.nc.  This image has no NetCDF writer, so the window
is read with the engine's own NetCDF-4/HDF5 reader (nemoflux_amd/io.py; uo/vo one time step at a time, never the
whole variable) and written as the T.npz / U.npz / V.npz bundles that Field, HorizGrid and fluxplot open
(same variable names; `_FillValue` and the CF attributes of the time axis are kept).

    python -m nemoflux_amd.subsetnemo --tfile T.nc --ufile U.nc --vfile V.nc --outputdir sub --jmin 0 --jmax 100 \\
        --imin 200 --imax 300
"""
import argparse
import json
import os

import numpy

from . import io


def _text(v):
    if isinstance(v, bytes):
        return v.decode('utf-8', 'replace')
    if isinstance(v, numpy.ndarray):
        return _text(v.tobytes()) if v.dtype.kind == 'S' else v.tolist()
    if isinstance(v, numpy.generic):
        return v.item()
    return v


def _window(var, jmin, jmax, imin, imax):
    """var[..., jmin:jmax, imin:imax]; lazily read variables are cut step by step."""
    if hasattr(var, 'read_step'):
        return numpy.stack([var.read_step(t)[..., jmin:jmax, imin:imax] for t in range(var.shape[0])])
    return numpy.ascontiguousarray(var[..., jmin:jmax, imin:imax])


def main(*, tfile, ufile, vfile, outputdir='./', jmin, jmax, imin, imax):
    """Same keyword interface as subsetNEMO.py:6-9; writes <outputdir>/T.npz, U.npz, V.npz."""
    if not (0 <= jmin < jmax and 0 <= imin < imax):
        raise RuntimeError('ERROR: need 0 <= jmin < jmax and 0 <= imin < imax')
    os.makedirs(outputdir, exist_ok=True)
    print(f'T file: {tfile}')
    t = io.open_tfile(tfile)
    ny, nx = t['bounds_lon'].shape[:2]
    if jmax > ny or imax > nx:
        raise RuntimeError(f'ERROR: window [{jmin}:{jmax}, {imin}:{imax}] exceeds the (y, x) = ({ny}, {nx}) grid')
    out = {}
    for vname in 'bounds_lon', 'bounds_lat', 'deptht', 'deptht_bounds':   # subsetNEMO.py:39-51
        if vname not in t:
            continue
        print(f'creating variable {vname}')
        out[vname] = t[vname][jmin:jmax, imin:imax] if ('lon' in vname or 'lat' in vname) else numpy.asarray(t[vname])
    numpy.savez(os.path.join(outputdir, 'T.npz'), **out)

    for field, fname, path in ('U', 'uo', ufile), ('V', 'vo', vfile):     # subsetNEMO.py:56-91
        var, fill, d = io.open_uvfile(path, fname, with_all=True)
        out = {}
        for vname in 'time_counter', 'time_centered', 'time_centered_bounds':
            if vname in d:
                print(f'creating variable {vname}')
                out[vname] = numpy.asarray(d[vname])
                attrs = d.get('_attrs_' + vname)
                if attrs:
                    out['_attrs_' + vname] = numpy.array(json.dumps({k: _text(v) for k, v in attrs.items()}))
        print(f'creating variable {fname}')
        out[fname] = _window(var, jmin, jmax, imin, imax)
        if not numpy.isnan(fill):
            out['_FillValue_' + fname] = numpy.asarray(fill, dtype=out[fname].dtype)
        numpy.savez(os.path.join(outputdir, f'{field}.npz'), **out)
    return outputdir


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description='subset nemo data')
    ap.add_argument('--tfile', required=True, help='file containing the T cell grid data')
    ap.add_argument('--ufile', required=True, help='file containing u data')
    ap.add_argument('--vfile', required=True, help='file containing v data')
    ap.add_argument('--outputdir', default='./', help='the files are saved as T.npz, U.npz and V.npz there')
    ap.add_argument('--jmin', type=int, required=True)
    ap.add_argument('--jmax', type=int, required=True)
    ap.add_argument('--imin', type=int, required=True)
    ap.add_argument('--imax', type=int, required=True)
    main(**vars(ap.parse_args()))
