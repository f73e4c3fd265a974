"""K2 (weight build) against the NUMBER of transects on the ORCA12-like grid (round-4 verdict W9): nf_field_build_weights for
65 (config C5's batch), 512 and 4096 seeded polylines of 8-64 vertices, with fingerprints of the result -- sha256 of the
mint-shaped weights (cell*4+edge, weight, segment) up to 512 transects, of the per-segment / per-transect rows of one time
step always -- so that two builds of the library can be compared bit for bit.

usage: python tools/weights_scaling.py [label] [counts, default 65,512,4096] [rot: the grid with its pole moved by (20, 30) degrees]
"""
import contextlib, ctypes, hashlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
import bench
from nemoflux_amd._lib import lib, check
from nemoflux_amd.datagen import DataGen, STREAM_FUNCTIONS
from nemoflux_amd.field import Field

label = sys.argv[1] if len(sys.argv) > 1 else ''
counts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '65,512,4096').split(',')]
nx, ny = 3600, 1800
rot = len(sys.argv) > 3 and sys.argv[3] == 'rot'
dg = DataGen(); dg.setSizes(nx, ny, 2, 1); dg.setBoundingBox(-180., 180., -90., 90., 0., 1.); dg.build()
if rot:     # curvilinear: the cells' boxes are no longer the cells, the ones around the moved pole are large and unusable
    dg.rotatePole((20., 30.))
dg.applyStreamFunction(STREAM_FUNCTIONS[3]); u, v = dg.computeUVFromPotential()
print(f'weight build vs number of transects {label}: {nx} x {ny} cells{" (rotated pole)" if rot else ""}, periodX = 360 (3 images per target segment)')
import warnings
warnings.simplefilter('ignore')
for n in counts:
    polys = bench.make_transects(nx, ny, -180., 180., -90., 90., n - 3, seed=20260402, seam=True)
    xyzs = [numpy.array([(x, y, 0.) for x, y in p]) for p in polys]
    nseg = sum(len(p) - 1 for p in polys)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        f = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, u, v, xyzs, readback=False,
                             unsupportedCells='skip' if rot else 'refuse')
    t_field = time.perf_counter() - t0
    best = 1e30
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(lib.nf_field_build_weights(ctypes.byref(f._h), 128, ctypes.c_double(360.)))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    nw = ctypes.c_size_t()
    check(lib.nf_field_num_weights(ctypes.byref(f._h), ctypes.byref(nw)))
    tot, segs = f.computeAll()
    h_rows = hashlib.sha256(numpy.ascontiguousarray(segs).tobytes() + numpy.ascontiguousarray(tot).tobytes()).hexdigest()[:16]
    h_w = '-'
    if n <= 512:
        ce, w, sg = f.getWeights()
        h_w = hashlib.sha256(ce.tobytes() + w.tobytes() + sg.tobytes()).hexdigest()[:16]
    cov = numpy.concatenate(f.getCoverage())
    print(f'{len(xyzs):5d} transects, {nseg:7d} target segments: build_weights {best * 1e3:10.2f} ms, {nw.value // 4:10d} records '
          f'({nseg / best:12.0f} segments/s); Field construction {t_field:.2f} s; coverage {cov.min():.9f}..{cov.max():.9f}; '
          f'sha256 weights {h_w} rows {h_rows}', flush=True)
    del f
    torch.cuda.empty_cache()
