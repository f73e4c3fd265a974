"""Viewer path timing: Field.update() on the ORCA12-like grid = flux kernel + transect reduction + re-pack + D2H of
integratedVelocity (207 MB) and the two |flux| arrays (104 MB) into the pinned host arrays VTK aliases."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from nemoflux_amd.datagen import DataGen, STREAM_FUNCTIONS
from nemoflux_amd.field import Field
dg = DataGen(); dg.setSizes(3600, 1800, 75, 2); dg.setBoundingBox(-180, 180, -90, 90, 0, 1); dg.build()
dg.applyStreamFunction(STREAM_FUNCTIONS[5]); dg.computeUVFromPotential()
tr = [numpy.array([(-180., -80., 0.), (-10., -80., 0.), (-10., 80., 0.), (-180., 80., 0.)])]
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    f = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, dg.u, dg.v, tr)
print(f'Field construction (geometry, weights, first step, locator of {f.vectorPoints.shape[0]} arrow points): {time.perf_counter()-t0:.3f} s')
for r in range(2):
    f.update()
t0 = time.perf_counter(); R = 5
for r in range(R):
    f.timeIndex = r % 2
    f.update()
dt = (time.perf_counter() - t0) / R
print(f'update(): {dt*1e3:.2f} ms per step ({(311e6+f.vectorValues.nbytes)/dt/1e9:.1f} GB/s D2H-equivalent); flux text {f.getFluxText()!r}')
for r in range(2):
    f.computeFlux(r)
t0 = time.perf_counter()
for r in range(R):
    f.computeFlux(r % 2)
print(f'computeFlux(t) without read-back: {(time.perf_counter()-t0)/R*1e3:.2f} ms per step')
