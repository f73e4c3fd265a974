"""GPU parity tests proper: HIP path (through the C ABI / ctypes) vs the CPU oracle and the golden
vectors produced by running the reference (oracle/gen_golden.py).

Bars (BASELINE north_star: "match ... to a stated fp64 tolerance"):
  * cell-bounds assembly (A1): bit-exact.
  * vertical integral + edge flux (A4+A5): bit-exact against the oracle given the same arc lengths (both are
    the same sequential fma chain); against the reference's numpy/BLAS tensordot: |d| <= 4 nz eps sum|th*f|.
  * arc lengths (A3): acos of a dot product amplifies the ulp differences between device and glibc sin/cos
    by 1/sin(angle): |d arc| <= 16 eps / max(arc, sqrt(eps)).
  * weights (A6): same algorithm, no transcendental: <= 1e-13 absolute per (segment, cell, edge).
  * transect totals (A7): <= 1e-12 * sum|w f|.
"""
import ast
import numpy
import pytest

from conftest import FULL_CASES, case_box, load_golden, transect_xyz

pytestmark = pytest.mark.gpu
EPS = numpy.finfo(numpy.float64).eps


def arc_tol(arc):
    return 16 * EPS / numpy.maximum(arc, numpy.sqrt(EPS))


def make_field(g, m, transects, **kw):
    from nemoflux_amd.field import Field
    return Field.fromArrays(g['bounds_lon'], g['bounds_lat'], g['deptht_bounds'], g['u'], g['v'], transects,
                            sverdrup=m['sverdrup'], fill_value=m['fill_value'], **kw)


@pytest.mark.parametrize('name', FULL_CASES + ['cossin360', 'rot360_zt'])
def test_geometry(name, oracle, cases):
    from nemoflux_amd.field import _geometry_only
    g = load_golden(name)
    geo = _geometry_only(g['bounds_lon'], g['bounds_lat'])
    ref_pts = oracle.assemble_points(g['bounds_lon'], g['bounds_lat'])
    assert numpy.array_equal(geo['points'], ref_pts)                       # A1 bit-exact
    arc = geo['arcLengths']
    ok = ~numpy.isnan(g['arcLengths'])
    assert numpy.all(numpy.abs(arc - g['arcLengths'])[ok] <= arc_tol(g['arcLengths'])[ok])  # vs reference
    assert numpy.all(numpy.abs(arc - oracle.arc_lengths(ref_pts))[ok] <= arc_tol(arc)[ok])  # vs oracle
    assert geo['box'][0] == g['bounds_lon'].min() and geo['box'][1] == g['bounds_lon'].max()
    assert geo['box'][2] == g['bounds_lat'].min() and geo['box'][3] == g['bounds_lat'].max()


def test_geometry_f32_real_nemo(oracle):
    """data/sa/T.nc: real ORCA025 subset, float32 bounds."""
    from nemoflux_amd.field import _geometry_only
    b = load_golden('sa_T_bounds')
    ref = load_golden('sa_T_arc')['arcLengths']
    geo = _geometry_only(b['bounds_lon'], b['bounds_lat'])
    assert b['bounds_lon'].dtype == numpy.float32
    assert numpy.array_equal(geo['points'][:, :, 0].reshape(b['bounds_lon'].shape), b['bounds_lon'].astype(numpy.float64))
    assert numpy.all(numpy.abs(geo['arcLengths'] - ref) <= arc_tol(ref))


@pytest.mark.parametrize('name', FULL_CASES)
def test_edge_flux_all_steps(name, oracle, cases):
    m = [c for c in cases if c['name'] == name][0]
    g = load_golden(name)
    trs = [transect_xyz(t['points']) for t in m['transects'].values()]
    fld = make_field(g, m, trs)
    arc = fld.arcLengths
    st = oracle.EdgeFluxState(m['ny'], m['nx'])
    for t in range(m['nt']):
        fld.timeIndex = t
        fld.update()
        # oracle: same fma chain, same arc lengths -> bit-exact
        U = oracle.vertical_integral(g['u'][t], g['thickness'], m['fill_value'])
        V = oracle.vertical_integral(g['v'][t], g['thickness'], m['fill_value'])
        oracle.edge_flux(st, U, V, arc, m['sverdrup'])
        assert numpy.array_equal(fld.integratedVelocity, st.integratedVelocity)
        assert numpy.array_equal(fld.edgeFluxesUArray, st.edgeFluxesU)
        assert numpy.array_equal(fld.edgeFluxesVArray, st.edgeFluxesV)
        assert fld.maxAbsFlux == st.maxAbsFlux.value
        # reference (numpy tensordot + its own arc lengths): tolerance
        ref = g['integratedVelocity'][t]
        scale = numpy.abs(ref).max() + 1e-300
        pole = numpy.zeros(ref.shape, bool)
        pole.reshape(m['ny'], m['nx'], 4)[-1, :, 2] = True   # pole row: 1e13 * acos(1 - ulp) garbage (quirk 4)
        pole.reshape(m['ny'], m['nx'], 4)[-1, :, :] |= m['deltaDeg'][0] != 0
        d = numpy.abs(fld.integratedVelocity - ref)
        assert d[~pole].max() <= 1e-10 * scale
        assert numpy.array_equal(fld.integratedVelocity.reshape(m['ny'], m['nx'], 4)[0, :, 0], numpy.zeros(m['nx']))


@pytest.mark.parametrize('name', FULL_CASES + ['cossin360', 'rot360_zt'])
def test_weights_vs_oracle(name, oracle, cases):
    from nemoflux_amd import mint
    m = [c for c in cases if c['name'] == name][0]
    g = load_golden(name)
    pts = oracle.assemble_points(g['bounds_lon'], g['bounds_lat'])
    grid = mint.Grid()
    grid.setPoints(pts)
    assert grid.getNumberOfCells() == pts.shape[0]
    rng = numpy.random.default_rng(7)
    data = rng.standard_normal((pts.shape[0], 4))
    for tn, tr in m['transects'].items():
        xyz = transect_xyz(tr['points'])
        pli = mint.PolylineIntegral()
        pli.setGrid(grid)
        pli.buildLocator(numCellsPerBucket=128, periodX=360., enableFolding=False)
        pli.computeWeights(xyz, counterclock=False)
        ce, w, sg = pli.getWeights()
        ow = oracle.polyline_weights(pts, xyz)
        assert ce.size == ow.weight.size
        gd = {}
        for a, b, c in zip(sg.tolist(), ce.tolist(), w.tolist()):
            gd[(a, b)] = gd.get((a, b), 0.0) + c
        od = ow.as_dict()
        assert set(gd) == set(od)
        assert max(abs(gd[k] - od[k]) for k in od) <= 1e-13
        assert numpy.all(numpy.diff(sg) >= 0)                       # sorted by target segment
        tot = pli.getIntegral(data, mint.CELL_BY_CELL_DATA)
        otot, osegs = oracle.get_integral(ow, data, True)
        bound = 1e-12 * numpy.abs(ow.weight * data.reshape(-1)[ow.cell_edge]).sum()
        assert abs(tot - otot) <= bound
        segs, tot2 = pli.getSegmentIntegrals(data)
        assert tot2 == tot and numpy.all(numpy.abs(segs - osegs) <= bound)


def test_known_answers_readme(oracle, cases):
    """The only things that pin mint's results: README.md:39 (360), :56 (0.5), closed loop (0)."""
    for name, tn, expect, tol in [('c1_x', 'readme', 360.0, 1e-12), ('c1_x', 'tri', 0.0, 1e-12),
                                  ('singular', 'sing', 0.5, 1e-14), ('cossin36', 'tri', 0.0, 1e-14)]:
        m = [c for c in cases if c['name'] == name][0]
        g = load_golden(name)
        fld = make_field(g, m, [transect_xyz(m['transects'][tn]['points'])])
        assert abs(fld.computeFlux(0)[0] - expect) <= tol, (name, tn)
    m = [c for c in cases if c['name'] == 'c1_x'][0]
    fld = make_field(load_golden('c1_x'), m, [transect_xyz(m['transects']['readme']['points'])])
    assert fld.getFluxText() == ' 360 (A m^2/s) '           # f"{360.0:4.3g}, " + unit, regex-tidied (field.py:103-108)
    assert abs(fld.maxAbsFlux - 10.0) <= 1e-14                # colour-bar max of pictures/simple.png


@pytest.mark.parametrize('name', ['def36_zt', 'cossin36', 'reg16', 'wrap36_zt'])
def test_field_vs_fluxexact(name, oracle, cases):
    m = [c for c in cases if c['name'] == name][0]
    g = load_golden(name)
    names = list(m['transects'])
    fld = make_field(g, m, [transect_xyz(m['transects'][n]['points']) for n in names])
    tot, segs = fld.computeAll()
    # a closed loop sums to 0 out of terms as large as the case's open-transect fluxes: errors scale with those
    amp = max([1.0] + [abs(x) for n in names for x in (m['transects'][n]['fluxexact'] or [])])
    for t in range(m['nt']):
        one = fld.computeFlux(t)
        for i, n in enumerate(names):
            exact = m['transects'][n]['fluxexact'][t]      # reference fluxexact.py prints %20.10g
            assert abs(one[i] - exact) <= 1e-9 * max(amp, abs(exact))
            assert one[i] == tot[t, i]
            ex2 = oracle.fluxexact(m['psi'], ast.literal_eval(m['transects'][n]['points']), m['nz'], m['nt'],
                                   *case_box(m)[4:])[t]
            assert abs(one[i] - ex2) <= 1e-13 * max(amp, abs(ex2))


@pytest.mark.parametrize('name', ['rot36_zt', 'cossin360', 'wrap36_zt'])
def test_level1_host_data_is_staged_sparsely(name, oracle, cases):
    """mint.PolylineIntegral.getIntegral / mint.VectorInterp.getFaceVectors on a HOST array (field.py:102,119; fluxplot.py:55-58:
    one call per transect per time step) move only the cells the object touches -- nrec x 32 B, not the whole (ncell,4) array
    (round-4 verdict W4) -- and return the BITS of the call on HBM-resident data: same products, same summation tree.  Proof
    that nothing else is read: NaN in every cell the weights / the located points do not touch changes nothing."""
    import torch
    from nemoflux_amd import mint
    m = [c for c in cases if c['name'] == name][0]
    g = load_golden(name)
    pts = oracle.assemble_points(g['bounds_lon'], g['bounds_lat'])
    grid = mint.Grid()
    grid.setPoints(pts)
    rng = numpy.random.default_rng(11)
    data = rng.standard_normal((pts.shape[0], 4))
    ddev = torch.from_numpy(data).cuda()
    for tn, tr in m['transects'].items():
        xyz = transect_xyz(tr['points'])
        pli = mint.PolylineIntegral()
        pli.setGrid(grid)
        pli.buildLocator(numCellsPerBucket=128, periodX=360., enableFolding=False)
        pli.computeWeights(xyz, counterclock=False)
        tot = pli.getIntegral(data, mint.CELL_BY_CELL_DATA)
        assert tot == pli.getIntegral(ddev, mint.CELL_BY_CELL_DATA)
        assert tot == pli.getIntegral(data.reshape(-1), mint.CELL_BY_CELL_DATA)          # flat form (field.py:102 passes either)
        ce, w, sg = pli.getWeights()
        holes = numpy.full_like(data, numpy.nan)
        touched = numpy.unique(ce // 4)
        holes[touched] = data[touched]
        assert pli.getIntegral(holes, mint.CELL_BY_CELL_DATA) == tot
        assert 0 < touched.size < pts.shape[0] // 2
        # a second set of weights on the same object re-sizes the staging
        pli.computeWeights(xyz[:2], counterclock=False)
        assert pli.getIntegral(data) == pli.getIntegral(ddev)
        # arrows on the same line
        vp = numpy.concatenate([a + (b - a) * numpy.linspace(0., 1., 7)[:, None] for a, b in zip(xyz[:-1], xyz[1:])])
        vp = numpy.concatenate([vp, [[500., 95., 0.]]])                                   # one point outside every cell
        vi = mint.VectorInterp()
        vi.setGrid(grid)
        vi.buildLocator(numCellsPerBucket=128, periodX=360.)
        assert vi.findPoints(vp, tol2=1.e-12) >= 1
        vec = vi.getFaceVectors(data, placement=mint.CELL_BY_CELL_DATA)
        assert numpy.array_equal(vec, vi.getFaceVectors(ddev)) and numpy.all(vec[-1] == 0.)
        ids, _ = vi.getCells()
        holes = numpy.full_like(data, numpy.nan)
        holes[ids[ids >= 0]] = data[ids[ids >= 0]]
        assert numpy.array_equal(vi.getFaceVectors(holes, placement=mint.CELL_BY_CELL_DATA), vec)
