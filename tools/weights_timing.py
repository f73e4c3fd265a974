"""K2 (weight build) on the ORCA12-like grid with the 65-polyline batch: bounds on one continuous branch against bounds
wrapped into [-180, 180) (1800 cells across the cut; their workgroups' cull boxes span the whole domain)."""
import contextlib, ctypes, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
import bench
from nemoflux_amd._lib import lib, check
from nemoflux_amd.datagen import DataGen, STREAM_FUNCTIONS
from nemoflux_amd.field import Field

nx, ny = 3600, 1800
dg = DataGen(); dg.setSizes(nx, ny, 2, 1); dg.setBoundingBox(0., 360., -90., 90., 0., 1.); dg.build()
dg.applyStreamFunction(STREAM_FUNCTIONS[3]); u, v = dg.computeUVFromPotential()
polys = bench.make_transects(nx, ny, 0., 360., -90., 90., 64, seed=20260403, seam=True)
xyzs = [numpy.array([(x, y, 0.) for x, y in p]) for p in polys]
for label, blon in (('continuous branch', dg.bounds_lon), ('wrapped into [-180,180)', torch.remainder(dg.bounds_lon + 180., 360.) - 180.)):
    best = 1e30
    for rep in range(3):
        with contextlib.redirect_stdout(io.StringIO()):
            f = Field.fromArrays(blon, dg.bounds_lat, dg.deptht_bounds, u, v, xyzs, readback=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(lib.nf_field_build_weights(ctypes.byref(f._h), 128, ctypes.c_double(360.)))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        nrec = f.getWeights()[0].size // 4
        del f
    print(f'{label:28s}: build_weights {best * 1e3:.2f} ms for {len(xyzs)} polylines, {nrec} records')
