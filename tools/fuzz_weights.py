"""Differential fuzz of K2 (nf_weights.hip) and the point location against the CPU oracle: random geometries (regular,
rotated pole, regional, sheared, date-line-wrapped, halo columns, float32-rounded bounds), random polylines (free, snapped
to nodes, along grid lines, a period away, closed), both periodic and non-periodic locators.  Every case: same error or
same weights entry by entry (1e-12), same coverage; when the oracle says coverage 1 on a grid with shared nodes, lon / lat
integrate to the end-point differences.      python tools/fuzz_weights.py [ncases] [seed]
Test infrastructure (uses oracle/): never imported by the product."""
import os
import sys
import time
import warnings

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import nf_oracle as oracle  # noqa: E402
from nemoflux_amd import mint  # noqa: E402
from nemoflux_amd._lib import NemofluxError  # noqa: E402

oracle.build()
rng = None


def wrap180(lon):
    return (lon + 180.) % 360. - 180.


def geometry():
    kind = rng.choice(['regular', 'rotated', 'regional', 'sheared', 'wrapped', 'halo', 'f32', 'start73'])
    nx, ny = int(rng.choice([24, 36, 72])), int(rng.choice([12, 18, 36]))
    x0 = 73. if kind in ('start73', 'halo') else (0. if kind == 'wrapped' else -180.)
    o = oracle.DataGen(nx, ny, 1, 1, xmin=x0, xmax=x0 + 360., lat_uses_dx=False)
    periodX = 360.
    if kind == 'rotated':
        o.rotatePole((float(rng.uniform(-40, 40)), float(rng.uniform(-40, 40))))
    blon, blat = o.bounds_lon.copy(), o.bounds_lat.copy()
    if kind == 'regional':
        j0, j1 = sorted(rng.choice(ny + 1, 2, replace=False))
        i0, i1 = sorted(rng.choice(nx + 1, 2, replace=False))
        if j1 - j0 < 2 or i1 - i0 < 2:
            j0, j1, i0, i1 = 2, ny - 2, 3, nx - 3
        blon, blat = blon[j0:j1, i0:i1], blat[j0:j1, i0:i1]
        periodX = float(rng.choice([0., 360.]))
    if kind == 'sheared':
        blon = blon + 0.3 * blat          # parallelograms (still a conforming mesh)
        periodX = 0.
    if kind in ('wrapped', 'start73', 'halo') and rng.random() < 0.8:
        if kind == 'halo':
            blon = numpy.concatenate([blon[:, -2:] - 360., blon], axis=1)
            blat = numpy.concatenate([blat[:, -2:], blat], axis=1)
        blon = wrap180(blon)
    if kind == 'f32':
        blon, blat = blon.astype(numpy.float32).astype(numpy.float64), blat.astype(numpy.float32).astype(numpy.float64)
    return kind, numpy.ascontiguousarray(blon), numpy.ascontiguousarray(blat), periodX, x0


def polyline(blon, blat, periodX):
    n = int(rng.integers(2, 9))
    if rng.random() < 0.08:
        n = int(rng.integers(90, 400))      # a long polyline: more than 256 segment images (several chunks of the workgroup cull)
    lo, hi = blon.min(), blon.max()
    if hi - lo > 350.:
        lo, hi = lo - 20., hi + 20.
    x = rng.uniform(lo - 5., hi + 5., n)
    y = rng.uniform(max(-88., blat.min() - 3.), min(88., blat.max() + 3.), n)
    mode = rng.integers(0, 6)
    nodes_x, nodes_y = numpy.unique(blon), numpy.unique(blat)
    if mode in (1, 2):
        x = nodes_x[rng.integers(0, nodes_x.size, n)]
        y = nodes_y[rng.integers(0, nodes_y.size, n)]
        y = numpy.clip(y, -88., 88.)
        if mode == 2:
            x[1::2] = x[0::2][:x[1::2].size]
    if mode == 3 and periodX > 0:
        x = x + float(rng.choice([-360., 360.]))
    if mode == 4:
        x, y = numpy.append(x, x[0]), numpy.append(y, y[0])
    if mode == 5 and n > 2:
        x[1], y[1] = x[0], y[0]            # a repeated point
    xyz = numpy.zeros((x.size, 3))
    xyz[:, 0], xyz[:, 1] = x, y
    return xyz


def run(ncases=500, seed=1, verbose=True):
    global rng
    rng = numpy.random.default_rng(seed)
    stats = {'ok': 0, 'refused': 0, 'over': 0, 'points': 0}
    kinds = {}
    t0 = time.time()
    for case in range(ncases):
        kind, blon, blat, periodX, x0 = geometry()
        pts = oracle.assemble_points(blon, blat)
        grid = mint.Grid()
        grid.setPoints(pts)
        for trial in range(4):
            xyz = polyline(blon, blat, periodX)
            pli = mint.PolylineIntegral()
            pli.setGrid(grid)
            pli.buildLocator(numCellsPerBucket=128, periodX=periodX, enableFolding=False)
            pli.setUnsupportedCells('refuse')          # like the oracle's default: both must refuse the same cell
            try:
                ow = oracle.polyline_weights(pts, xyz, periodX=periodX)
            except oracle.UnsupportedCell as e:
                try:
                    pli.computeWeights(xyz, counterclock=False)
                except NemofluxError as ge:
                    assert f'cell {e.cell}' in str(ge), (case, kind, str(ge), e.cell)
                    stats['refused'] += 1
                    kinds[('refused', kind)] = kinds.get(('refused', kind), 0) + 1
                    # round 6: the same line under the drop-in's default policy ('skip'): the unsupported cells drop out of both
                    # sides alike -- same entries, same coverage -- and the device counts what it dropped
                    try:
                        osk = oracle.polyline_weights(pts, xyz, periodX=periodX, skip_unsupported=True)
                    except (oracle.OverCovered, oracle.UnsupportedCell):
                        continue        # what is left of the line is over-covered / Newton fails elsewhere: other campaigns
                    pli.setUnsupportedCells('skip')
                    with warnings.catch_warnings():
                        warnings.simplefilter('ignore')
                        pli.computeWeights(xyz, counterclock=False)
                    assert pli.getNumberOfDroppedCrossings() >= 1, (case, kind)
                    assert numpy.allclose(pli.getCoverage(), osk.coverage, rtol=0, atol=1e-10), (case, kind)
                    ce, w, sg = pli.getWeights()
                    gd = {}
                    for a, b, c in zip(sg.tolist(), ce.tolist(), w.tolist()):
                        gd[(a, b)] = gd.get((a, b), 0.0) + c
                    od = osk.as_dict()
                    assert set(gd) == set(od) and (not od or max(abs(gd[k] - od[k]) for k in od) <= 1e-12), (case, kind)
                    stats['skipped_ok'] = stats.get('skipped_ok', 0) + 1
                    continue
                raise AssertionError(f'case {case} ({kind}): oracle refuses cell {e.cell}, the GPU does not')
            except oracle.OverCovered as e:
                try:
                    pli.computeWeights(xyz, counterclock=False)
                except NemofluxError as ge:
                    assert f'segment {e.seg} ' in str(ge) and 'covered' in str(ge), (case, kind, str(ge), e.seg)
                    assert numpy.allclose(pli.getCoverage(), e.coverage, rtol=0, atol=1e-10)
                    stats['over'] += 1
                    kinds[('over', kind, periodX)] = kinds.get(('over', kind, periodX), 0) + 1
                    continue
                raise AssertionError(f'case {case} ({kind}): oracle says over-covered segment {e.seg}, the GPU does not')
            pli.computeWeights(xyz, counterclock=False)
            ce, w, sg = pli.getWeights()
            gd = {}
            for a, b, c in zip(sg.tolist(), ce.tolist(), w.tolist()):
                gd[(a, b)] = gd.get((a, b), 0.0) + c
            od = ow.as_dict()
            assert set(gd) == set(od), (case, kind, trial, sorted(set(gd) ^ set(od))[:6], xyz.tolist(), periodX)
            if od:
                worst = max(abs(gd[k] - od[k]) for k in od)
                assert worst <= 1e-12, (case, kind, trial, worst)
            assert numpy.allclose(pli.getCoverage(), ow.coverage, rtol=0, atol=1e-11), (case, kind, trial)
            stats['ok'] += 1
        # point location on the same grid
        tg = numpy.zeros((16, 3))
        tg[:, 0] = rng.uniform(blon.min() - 10., blon.max() + 10., 16)
        tg[:, 1] = rng.uniform(max(-89., blat.min() - 2.), min(89., blat.max() + 2.), 16)
        data = rng.standard_normal((pts.shape[0], 4))
        vi = mint.VectorInterp()
        vi.setGrid(grid)
        vi.buildLocator(numCellsPerBucket=128, periodX=periodX, enableFolding=False)
        vi.findPoints(tg, tol2=1.e-12)
        ov, oi = oracle.vector_interp(pts, tg, data, periodX=periodX)
        gi = vi.getCells()[0]
        assert numpy.array_equal(gi, oi), (case, kind, gi.tolist(), oi.tolist())
        gv = vi.getFaceVectors(data, placement=mint.CELL_BY_CELL_DATA)
        assert numpy.allclose(gv, ov, rtol=1e-10, atol=1e-10 * max(1., numpy.abs(ov).max())), (case, kind)
        stats['points'] += 16
        if verbose and case % 1000 == 999:
            print(f'{case + 1} geometries, {stats}, {time.time() - t0:.0f} s', flush=True)
    if verbose:
        print(f'fuzz OK: {ncases} geometries x 4 polylines, seed {seed}: {stats}')
        print('errors by kind (both sides agree):', kinds)
    return stats, kinds


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
