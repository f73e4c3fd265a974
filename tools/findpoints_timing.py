"""mint.VectorInterp.findPoints (field.py:93: the arrow seed points of all target lines) on the ORCA12-like grid: time against
the number of points.  usage: python tools/findpoints_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy, torch
from nemoflux_amd import mint
from nemoflux_amd.datagen import DataGen

nx, ny = 3600, 1800
dg = DataGen(); dg.setSizes(nx, ny, 1, 1); dg.setBoundingBox(-180., 180., -90., 90., 0., 1.); dg.build()
pts = numpy.zeros((ny * nx, 4, 3)); pts[:, :, 0] = dg.bounds_lon.cpu().numpy().reshape(-1, 4); pts[:, :, 1] = dg.bounds_lat.cpu().numpy().reshape(-1, 4)
gr = mint.Grid(); gr.setPoints(pts)
rng = numpy.random.default_rng(3)
for n in (100, 5000, 200000, 2000000):
    tp = numpy.zeros((n, 3)); tp[:, 0] = rng.uniform(-180, 180, n); tp[:, 1] = rng.uniform(-89.9, 89.9, n)
    vi = mint.VectorInterp(); vi.setGrid(gr); vi.buildLocator(numCellsPerBucket=128, periodX=360.)
    vi.findPoints(tp, tol2=1.e-12)
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter(); nf = vi.findPoints(tp, tol2=1.e-12); best = min(best, time.perf_counter() - t0)
    ids, pc = vi.getCells()
    j = numpy.floor((tp[:, 1] + 90.) / 0.1).astype(int); i = numpy.floor((tp[:, 0] + 180.) / 0.1).astype(int)
    ok = numpy.mean(numpy.abs(ids - (j * nx + i)) <= nx + 1)
    print(f'{n:8d} points: findPoints {best * 1e3:9.3f} ms ({n / best:12.0f} points/s), not found {nf}, in the expected cell or a neighbour {ok:.4f}', flush=True)
