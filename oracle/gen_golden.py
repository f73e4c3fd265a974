#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING THE REFERENCE ITSELF (test infrastructure only).

This script imports /root/reference/nemoflux/{geo,datagen,field,fluxexact,latlonreader}.py
unmodified, with EMPTY stub modules for the third-party packages that are absent from this
image (netCDF4, defopt, xarray, mint, vtk), and dumps small input/output vectors into
tests/golden/.  Only the OUTPUTS travel (the reference itself never leaves this container).

What the reference executes here (unmodified code):
  * datagen.DataGen.{setSizes,setBoundingBox,build,rotatePole,applyStreamFunction,
    computeUVFromPotential}                          (datagen.py:16-166)
  * field.Field.computeArcLengths                    (field.py:170-181, geo.py:14-27)
  * field.Field.computeIntegratedFlux                (field.py:183-234)
  * fluxexact.main                                   (fluxexact.py:6-46; stdout captured)
  * latlonreader.LatLonReader                        (latlonreader.py:5-20)
What is restated in one line because xarray is absent:
  * field.Field.readField's arithmetic (field.py:157,161):
        numpy.tensordot(thickness, where(isnan(f)|f==fill, 0, f), axes=(0, 0))
mint (python-mint>=1.24.4, README.md:12) is NOT installed: nothing here pins A6/A7 except
the README known answers recorded in tests/golden/known_answers.json.

Run:  python oracle/gen_golden.py          (writes tests/golden/*.npz, *.json)
      python oracle/gen_golden.py --only=reg16        (one case; the others stay as they are)
      /opt/conda/bin/python3.9 oracle/gen_golden.py --tnc   (T.nc bounds via h5py)
"""
import sys, os, io, json, types, contextlib, glob

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def dump_tnc():
    """data/sa/T.nc (real ORCA025 subset, NetCDF-4/HDF5) -> npz; needs h5py (conda py3.9)."""
    import h5py, numpy
    with h5py.File(os.path.join(REF, 'data/sa/T.nc'), 'r') as f:
        numpy.savez_compressed(os.path.join(OUT, 'sa_T_bounds.npz'),
                               bounds_lon=f['bounds_lon'][:], bounds_lat=f['bounds_lat'][:],
                               deptht_bounds=f['deptht_bounds'][:])
    print('wrote sa_T_bounds.npz')


def main():
    import numpy
    for m in ['netCDF4', 'defopt', 'xarray', 'mint', 'vtk']:
        sys.modules[m] = types.ModuleType(m)
    sys.path.insert(0, os.path.join(REF, 'nemoflux'))
    import geo, datagen, field, fluxexact, latlonreader

    def run_fluxexact(psi, nz, nt, pts, zmin=0., zmax=1.):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
          try:
            fluxexact.main(potentialFunction=psi, zmin=zmin, zmax=zmax, nz=nz, nt=nt,
                           deltaDeg="(0.,0.)", lonLatPointsStr=pts)
          except NameError:
            # fluxexact.py:2 imports only pi, cos, sin: it cannot evaluate arctan2 (reference limitation)
            return None
        vals = []
        for line in buf.getvalue().splitlines():
            p = line.split()
            if len(p) == 2 and p[0].isdigit():
                vals.append(float(p[1]))
        return vals

    only = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--only=')]

    def case(name, psi, nx, ny, nz, nt, deltaDeg=(0., 0.), transects=(), sverdrup=False,
             full=True, land=None, box=(-180., 180., -90., 90., 0., 1.), wrap=False):
        if only and name not in only:
            return None
        with contextlib.redirect_stdout(io.StringIO()):
            dg = datagen.DataGen()
            dg.setSizes(nx, ny, nz, nt)
            dg.setBoundingBox(*box)
            dg.build()
            if deltaDeg[0] != 0 or deltaDeg[1] != 0:
                dg.rotatePole(deltaDeg=deltaDeg)
            dg.applyStreamFunction(psi)
            dg.computeUVFromPotential()
        if wrap:   # what a global T-file stores: every corner's longitude on its own wrapped into [-180, 180); the cells on the
            # cut then have corners ~350 degrees apart.  The reference's arc lengths / edge fluxes below run on THESE bounds
            dg.bounds_lon = (dg.bounds_lon + 180.) % 360. - 180.
        u, v = dg.u.copy(), dg.v.copy()
        fill = 1.e20
        if land is not None:  # (j0, j1, i0, i1): NaN block in u, 1e20 block in v (field.py:157)
            j0, j1, i0, i1 = land
            u[:, :, j0:j1, i0:i1] = numpy.nan
            v[:, :, j0:j1, i0:i1] = fill
        # T-file content (datagen.py:178-182)
        deptht_bounds = numpy.stack([dg.ztop, dg.zbot], axis=1)
        thickness = deptht_bounds[:, 1] - deptht_bounds[:, 0]          # field.py:51
        # HorizGrid.__init__ arithmetic (horizgrid.py:17-22)
        ncell = ny * nx
        points = numpy.zeros((ny, nx, 4, 3), numpy.float64)
        points[..., 0] = dg.bounds_lon
        points[..., 1] = dg.bounds_lat
        points = points.reshape((ncell, 4, 3))
        fld = object.__new__(field.Field)
        fld.gr = types.SimpleNamespace(getPoints=lambda: points)
        fld.ny, fld.nx, fld.sverdrup = ny, nx, sverdrup
        fld.arcLengths = numpy.zeros((ncell, 4), numpy.float64)
        fld.computeArcLengths()
        fld.edgeFluxesUArray = numpy.zeros((ncell,), numpy.float64)
        fld.edgeFluxesVArray = numpy.zeros((ncell,), numpy.float64)
        fld.integratedVelocity = numpy.zeros((ncell, 4), numpy.float64)
        fld.maxAbsFlux = 0.
        iVs, eUs, eVs, mx, uInts, vInts = [], [], [], [], [], []
        for t in range(nt):
            def rd(f):
                f = numpy.where(numpy.isnan(f) | (f == fill), 0.0, f)      # field.py:157
                return numpy.tensordot(thickness, f, axes=(0, 0))          # field.py:161
            uInt, vInt = rd(u[t]), rd(v[t])
            with numpy.errstate(all='ignore'):
                fld.computeIntegratedFlux(uInt, vInt)
            uInts.append(uInt.copy()); vInts.append(vInt.copy())
            iVs.append(fld.integratedVelocity.copy())
            eUs.append(fld.edgeFluxesUArray.copy()); eVs.append(fld.edgeFluxesVArray.copy())
            mx.append(float(fld.maxAbsFlux))
        iVs = numpy.array(iVs); eUs = numpy.array(eUs); eVs = numpy.array(eVs)
        # node potential of layer/time (for exact answers at nodes), datagen.py:69-82
        arrays = dict(bounds_lon=dg.bounds_lon, bounds_lat=dg.bounds_lat,
                      deptht_bounds=deptht_bounds, thickness=thickness,
                      arcLengths=fld.arcLengths, maxAbsFlux=numpy.array(mx))
        if full:
            arrays.update(u=u, v=v, uInt=numpy.array(uInts), vInt=numpy.array(vInts),
                          integratedVelocity=iVs, edgeFluxesU=eUs, edgeFluxesV=eVs)
        else:
            # samples + checksums for the larger case
            rows = [0, 1, ny // 2, ny - 2, ny - 1]
            cols = [0, 1, nx // 2, nx - 1]
            arrays.update(sample_rows=numpy.array(rows), sample_cols=numpy.array(cols),
                          u_t0_rows=u[0][:, rows, :], v_t0_rows=v[0][:, rows, :],
                          iV_rows=iVs.reshape(nt, ny, nx, 4)[:, rows], iV_cols=iVs.reshape(nt, ny, nx, 4)[:, :, cols],
                          eU_rows=eUs.reshape(nt, ny, nx)[:, rows], eV_rows=eVs.reshape(nt, ny, nx)[:, rows],
                          u_sum=numpy.array([u[t].sum() for t in range(nt)]),
                          v_sum_finite=numpy.array([v[t][:, :-1, :].sum() for t in range(nt)]),
                          iV_abs_sum=numpy.array([numpy.abs(iVs[t].reshape(ny, nx, 4)[:-1]).sum() for t in range(nt)]))
        numpy.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
        meta = dict(name=name, psi=psi, nx=nx, ny=ny, nz=nz, nt=nt, deltaDeg=list(deltaDeg),
                    sverdrup=sverdrup, fill_value=fill, land=land, transects={})
        for tname, pts in transects:
            meta['transects'][tname] = dict(points=pts, fluxexact=run_fluxexact(psi, nz, nt, pts, box[4], box[5]))
        meta['box'] = list(box)
        if wrap:
            meta['wrap'] = True
        print(name, 'maxAbsFlux', mx[-1])
        return meta

    metas = []
    T_README = "(-180,-70),(-160,-10),(-35,40),(20,-50),(60,50),(180,40)"
    T_SING = "(-180,-80), (-10, -80),(-10,80), (-180, 80)"
    T_TRI = "(-100,-80),(100,-80),(0,80),(-100,-80)"
    T_OPEN = "(-100,-80),(100,-80),(0,80)"
    PSI_CS = "cos(2*pi*y/360) + sin(2*pi*x/360)"
    PSI_ZT = "(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))"
    PSI_DEF = "(cos(t*2*pi/nt)+2)*(0.5*(y/180)**2 + sin(2*pi*x/360))"
    metas.append(case('c1_x', 'x', 36, 18, 1, 1, transects=[('readme', T_README), ('tri', T_TRI)]))
    metas.append(case('singular', 'arctan2(y, x+180)/(2*pi)', 36, 18, 1, 1, transects=[('sing', T_SING)]))
    metas.append(case('cossin36', PSI_CS, 36, 18, 1, 1, transects=[('tri', T_TRI), ('open', T_OPEN)]))
    # on this coarse rotated grid the cells that touch the geographic poles reach down to |lat| = 79 and their (lon,lat)
    # images are not quads the weights are defined on: the triangle stays clear of them
    metas.append(case('rot36_zt', PSI_ZT, 36, 18, 3, 2, deltaDeg=(20., 30.),
                      transects=[('tri', "(-100,-50),(100,-50),(0,50),(-100,-50)")]))
    metas.append(case('def36_zt', PSI_DEF, 36, 18, 2, 3, transects=[('open', T_OPEN), ('readme', T_README)]))
    metas.append(case('sv36_land', PSI_ZT, 36, 18, 3, 2, sverdrup=True, land=(4, 9, 10, 20),
                      transects=[('open', T_OPEN)]))
    metas.append(case('cossin360', PSI_CS, 360, 180, 1, 1, transects=[('tri', T_TRI)], full=False))
    metas.append(case('rot360_zt', PSI_ZT, 360, 180, 2, 2, deltaDeg=(20., 30.), transects=[('tri', T_TRI), ('open', T_OPEN)], full=False))
    # a regional box whose dx != dy: the reference spaces LATITUDE with dx (datagen.py:49), and a
    # depth range other than [0, 1]
    metas.append(case('reg16', PSI_ZT, 16, 10, 2, 2, box=(-60., 20., -50., 10., 0., 100.),
                      transects=[('tri', "(-50,-45),(-20,-45),(-35,-10),(-50,-45)"), ('open', "(-50,-45),(-20,-45),(-35,-10)")]))
    # a global grid on [0, 360] whose T-file bounds are wrapped into [-180, 180) (round 4: cells across the date line); one
    # transect far from the cut, one across 180 E, one closed loop around it -- end points on nodes, so fluxexact applies
    metas.append(case('wrap36_zt', PSI_ZT, 36, 18, 2, 2, box=(0., 360., -90., 90., 0., 1.), wrap=True,
                      transects=[('probe', "(20,-40),(100,30)"), ('across', "(150,-40),(210,30)"),
                                 ('loop', "(170,-60),(190,-60),(190,60),(170,60),(170,-60)")]))
    metas = [m for m in metas if m is not None]
    if only:      # regenerate single cases: merge into the existing index, leave everything else untouched
        with open(os.path.join(OUT, 'cases.json')) as f:
            old = json.load(f)
        metas = [m for m in old if m['name'] not in only] + metas
        with open(os.path.join(OUT, 'cases.json'), 'w') as f:
            json.dump(metas, f, indent=1)
        return
    with open(os.path.join(OUT, 'cases.json'), 'w') as f:
        json.dump(metas, f, indent=1)

    # transect station tables parsed by the reference's own reader (latlonreader.py:5-20)
    stations = {}
    for fn in sorted(glob.glob(os.path.join(REF, 'data', '**', '*.txt'), recursive=True)):
        ll = latlonreader.LatLonReader(fn).getLonLats()
        stations[os.path.relpath(fn, os.path.join(REF, 'data'))] = ll.tolist()
    with open(os.path.join(OUT, 'stations.json'), 'w') as f:
        json.dump(stations, f)
    print('stations:', {k: len(v) for k, v in stations.items()})

    # arc lengths of the real ORCA025 subset, by the reference's own routine
    p = os.path.join(OUT, 'sa_T_bounds.npz')
    if os.path.exists(p):
        d = numpy.load(p)
        blon, blat = d['bounds_lon'], d['bounds_lat']
        ny, nx, _ = blon.shape
        points = numpy.zeros((ny, nx, 4, 3), numpy.float64)
        points[..., 0] = blon; points[..., 1] = blat
        points = points.reshape((ny * nx, 4, 3))
        fld = object.__new__(field.Field)
        fld.gr = types.SimpleNamespace(getPoints=lambda: points)
        fld.arcLengths = numpy.zeros((ny * nx, 4), numpy.float64)
        with numpy.errstate(all='ignore'):
            fld.computeArcLengths()
        numpy.savez_compressed(os.path.join(OUT, 'sa_T_arc.npz'), arcLengths=fld.arcLengths)
        print('sa arc: nan count', int(numpy.isnan(fld.arcLengths).sum()), 'max', numpy.nanmax(fld.arcLengths))

    # README / screenshot known answers (the only things that pin mint's A6/A7 here)
    known = {
        "c1_x/readme": {"value": 360.0, "source": "README.md:32,39; pictures/simple.png"},
        "c1_x/tri": {"value": 0.0, "source": "pictures/closed.png"},
        "singular/sing": {"value": 0.5, "source": "README.md:50-51,56; pictures/singular.png"},
        "cossin360/tri": {"value": 0.0, "abs_tol_seen": 4.2e-15, "source": "README.md:65-68; pictures/closed2.png"},
        "rot360_amp1/tri": {"value": 0.0, "abs_tol_seen": 2.34e-11, "source": "README.md:77-79; pictures/rotatedPole.png"},
        "colorbar_max": {"c1_x": 10.0, "singular": 0.125, "cossin360": 0.0175,
                         "source": "pictures/simple.png, singular.png, closed2.png"},
    }
    with open(os.path.join(OUT, 'known_answers.json'), 'w') as f:
        json.dump(known, f, indent=1)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    if '--tnc' in sys.argv:
        dump_tnc()
    else:
        main()
