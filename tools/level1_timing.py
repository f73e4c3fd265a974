"""Level-1 drop-in timing (round-4 verdict W4): what one call of the mint-shaped entry points costs when the caller hands
over HOST arrays, the way the reference does --

    mint.PolylineIntegral.getIntegral(integratedVelocity, CELL_BY_CELL_DATA)     field.py:102, fluxplot.py:55-58
    mint.VectorInterp.getFaceVectors(integratedVelocity, placement=0)            field.py:94-95, 119

-- on the ORCA025-like (C3, 1440 x 1021) and ORCA12-like (C4, 3600 x 1800) grids, for 1 and for 65 PolylineIntegral
objects (the reference's loop makes one call per transect per time step), next to the oracle's get_integral on the host
cores (the stand-in for mint's own sparse dot over the K ~ 4 x (cells crossed) entries; test infrastructure, timed here
as the yardstick only).  Also checks that the host-array call returns the bits of the call on HBM-resident data.

usage: python tools/level1_timing.py [label]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import nf_oracle as oracle  # noqa: E402
from nemoflux_amd import mint  # noqa: E402
from nemoflux_amd.datagen import DataGen  # noqa: E402

BOX = (-180., 180., -90., 90.)


def med(f, n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(numpy.median(ts))


def grid_points(nx, ny):
    dg = DataGen()
    dg.setSizes(nx, ny, 1, 1)
    dg.setBoundingBox(*BOX, 0., 1.)
    dg.build()
    blon, blat = dg.bounds_lon.cpu().numpy(), dg.bounds_lat.cpu().numpy()
    pts = numpy.zeros((ny * nx, 4, 3), numpy.float64)
    pts[:, :, 0] = blon.reshape(-1, 4)
    pts[:, :, 1] = blat.reshape(-1, 4)
    return pts


def one_grid(name, nx, ny, first):
    pts = grid_points(nx, ny)
    gr = mint.Grid()
    gr.setPoints(pts)
    ncell = gr.getNumberOfCells()
    rng = numpy.random.default_rng(nx)
    data = rng.standard_normal((ncell, 4))
    ddev = torch.from_numpy(data).cuda()
    polys = [first] + bench.make_transects(nx, ny, *BOX, 64)[1:]
    plis, ows = [], []
    t0 = time.perf_counter()
    for p in polys:
        xyz = numpy.array([(x, y, 0.) for x, y in p])
        pli = mint.PolylineIntegral()
        pli.setGrid(gr)
        pli.buildLocator(numCellsPerBucket=128, periodX=360., enableFolding=False)
        pli.computeWeights(xyz, counterclock=False)
        plis.append(pli)
    tbuild = time.perf_counter() - t0
    for pli in plis:
        ce, w, sg = pli.getWeights()
        ows.append(oracle.Weights(ce, w, sg, pli.numSegments))
    nent = [o.weight.size for o in ows]
    print(f'{name}: {nx} x {ny} = {ncell} cells; (ncell,4) host array = {data.nbytes / 1e6:.0f} MB; 65 PolylineIntegral objects built in '
          f'{tbuild:.2f} s; entries per object: first {nent[0]}, median {int(numpy.median(nent))}, max {max(nent)}, all {sum(nent)}')
    # same bits as the resident-data call
    same = all(pli.getIntegral(data) == pli.getIntegral(ddev) for pli in plis)
    print(f'  host-array getIntegral == HBM-resident getIntegral, bit for bit, on all 65: {same}')
    close = max(abs(pli.getIntegral(data) - oracle.get_integral(o, data)) / max(1e-300, numpy.abs(o.weight * data.reshape(-1)[o.cell_edge]).sum())
                for pli, o in zip(plis, ows))
    print(f'  vs the host sparse dot: max |difference| / sum |w d| = {close:.2e}')
    plis[0].getIntegral(data)
    t1 = med(lambda: plis[0].getIntegral(data), 9)
    t65 = med(lambda: [pli.getIntegral(data) for pli in plis], 3)
    o1 = med(lambda: oracle.get_integral(ows[0], data), 9)
    o65 = med(lambda: [oracle.get_integral(o, data) for o in ows], 3)
    d1 = med(lambda: plis[0].getIntegral(ddev), 9)
    print(f'  getIntegral(host array):  1 object {t1 * 1e3:9.3f} ms/call   65 objects {t65 * 1e3:9.2f} ms per step = {t65 / 65 * 1e3:8.3f} ms/call')
    print(f'  host sparse dot (oracle): 1 object {o1 * 1e3:9.3f} ms/call   65 objects {o65 * 1e3:9.2f} ms per step = {o65 / 65 * 1e3:8.3f} ms/call')
    print(f'  getIntegral(HBM-resident data): 1 object {d1 * 1e3:.3f} ms/call')
    # arrows: the reference seeds ~distance/dx points per target segment (field.py:71-87)
    dx = 360. / nx
    vp = []
    for a, b in zip(first[:-1], first[1:]):
        a, b = numpy.array(a + (0.,)), numpy.array(b + (0.,))
        n = max(2, int(numpy.sqrt(((b - a) ** 2).sum()) / dx))
        vp.append(a + (b - a) * numpy.linspace(0., 1., n)[:, None])
    vp = numpy.concatenate(vp)
    vi = mint.VectorInterp()
    vi.setGrid(gr)
    vi.buildLocator(numCellsPerBucket=128, periodX=360.)
    vi.findPoints(vp, tol2=1.e-12)
    a = vi.getFaceVectors(data, placement=mint.CELL_BY_CELL_DATA)
    b = vi.getFaceVectors(ddev)
    tv = med(lambda: vi.getFaceVectors(data, placement=mint.CELL_BY_CELL_DATA), 9)
    tvd = med(lambda: vi.getFaceVectors(ddev), 9)
    print(f'  getFaceVectors(host array), {vp.shape[0]} points: {tv * 1e3:.3f} ms/call (HBM-resident data: {tvd * 1e3:.3f}); same bits: {numpy.array_equal(a, b)}')
    return t1, t65 / 65


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else ''
    print(f'Level-1 call timing {label} on {torch.cuda.get_device_name(0)}; host cores {os.cpu_count()}')
    with open(os.path.join(ROOT, 'tests', 'golden', 'stations.json')) as f:      # the reference's own parse of data/S3_sta_bdep.txt
        first3 = [tuple(map(float, p)) for p in json.load(f)['S3_sta_bdep.txt']]
    a = one_grid('C3', 1440, 1021, first3)
    b = one_grid('C4', 3600, 1800, [(-180., -80.), (-10., -80.), (-10., 80.), (-180., 80.)])
    print(f'per-call cost C4 / C3 (cells 4.41 x): 1 object {b[0] / a[0]:.2f} x, 65 objects {b[1] / a[1]:.2f} x')


if __name__ == '__main__':
    main()
