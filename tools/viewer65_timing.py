import contextlib, io, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy, torch, bench
from nemoflux_amd.datagen import DataGen, STREAM_FUNCTIONS
from nemoflux_amd.field import Field
dg = DataGen(); dg.setSizes(3600, 1800, 75, 2); dg.setBoundingBox(-180, 180, -90, 90, 0, 1); dg.build()
dg.applyStreamFunction(STREAM_FUNCTIONS[5]); dg.computeUVFromPotential()
polys = bench.make_transects(3600, 1800, -180., 180., -90., 90., 64)
tr = [numpy.array([(x, y, 0.) for x, y in p]) for p in polys]
for rep in range(2):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        f = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, dg.u, dg.v, tr)
    print(f'viewer Field, 65 transects: construction {time.perf_counter()-t0:.3f} s, {f.vectorPoints.shape[0]} arrow points, not found {f.vinterp.numNotFound}')
    for r in range(2): f.update()
    t0 = time.perf_counter()
    for r in range(5):
        f.timeIndex = r % 2; f.update()
    print(f'update(): {(time.perf_counter()-t0)/5*1e3:.2f} ms per step; text {f.getFluxText()[:60]!r}')
    del f
