"""GPU parity, oracle-free: affine stream functions on curvilinear cells (round-5 verdict W1).

Bilinear cells reproduce psi = a*lon + b*lat + c exactly.  With edge data = nodal differences of psi in the +xi orientation
(S: psi1-psi0, E: psi2-psi1, N: psi2-psi3, W: psi3-psi0), the flux across ANY open polyline is a*dlon + b*dlat of its end
points -- end points inside cells, not on nodes -- and every interpolated face vector is (b, -a, 0).  Neither statement
involves the oracle, whose clip / inverse-bilinear helpers are the same text as the device's: a red test here is a device bug.

Checked through the Level-1 C ABI (mnt_grid_*, mnt_polylineintegral_*, mnt_vectorinterp_*: one object per polyline, host
arrays and HBM-resident data) and through Field (the batched weight build, the flux kernel writing the cell-by-cell array from
uo / vo, the segmented reduction of computeAll, the arrows of update()), on
  * the C2 rotated grid 360 x 180, deltaDeg = (20, 30)                          (BASELINE configs[1]),
  * the real ORCA025 geometry of the reference's data/sa/T.nc (float32 bounds)   (tests/golden/sa_T_bounds.npz),
  * the ORCA12-size rotated grid 3600 x 1800                                     (BASELINE configs[3] geometry),
and on the date-line-wrapped 36 x 18 rotated golden grid with periodX = 360 (lines kept away from the seam, where an affine
psi jumps), in float64 and -- through Field -- float32 inputs.  periodX = 0 elsewhere: an affine psi is not periodic.

Reference: /root/reference/nemoflux/field.py:45-48 (weights), :90-95,119 (face vectors), :102 (getIntegral);
README.md:45,58 (the flux depends on the end points only).
"""
import contextlib
import io
import time
import warnings

import numpy
import pytest

from conftest import affine_edge_data, affine_expected, load_golden, random_open_polylines

pytestmark = pytest.mark.gpu

A, B, C = 1.7, -0.6, 3.0          # psi = A*lon + B*lat + C
NLINES = 200
NPOINTS = 100000


def _rotated_bounds(nx, ny):
    from nemoflux_amd.datagen import DataGen
    dg = DataGen()
    dg.setSizes(nx, ny, 1, 1)
    dg.setBoundingBox(-180., 180., -90., 90., 0., 1.)
    dg.build()
    dg.rotatePole((20., 30.))
    return dg.bounds_lon.cpu().numpy(), dg.bounds_lat.cpu().numpy()


def _case(kind):
    """(bounds_lon, bounds_lat, region the lines and points are drawn in).  The regions keep clear of the cells the weights
    are not defined on (the two that touch a geographic pole on the rotated grids) and, on the regional grid, of row 0 and
    column 0, whose south / west edges Field never fills with their own value (field.py:219,223: SURVEY 8a quirks 1, 2)."""
    if kind == 'c2_rotated':
        return _rotated_bounds(360, 180) + ((-170., 170., -85., 85.),)
    if kind == 'orca12_rotated':
        return _rotated_bounds(3600, 1800) + ((-170., 170., -85., 85.),)
    b = load_golden('sa_T_bounds')
    return b['bounds_lon'], b['bounds_lat'], (14., 36., -41., -21.5)


def _cell_condition(points):
    """(largest |coordinate| / shortest non-degenerate edge, longest / shortest edge) of every cell in the lon-lat plane.
    The first is the rounding amplification of an interpolated vector: the basis vectors are differences of corner coordinates
    divided by the cell size, and a coordinate near 170 degrees carries ulp = 2.8e-14 -- against a 0.1-degree ORCA12 cell that
    is 2.8e-13 relative per nodal difference, against the 2e-4-degree short side of a cell next to a rotated pole 1e-10."""
    xy = points[:, :, :2]
    e = numpy.linalg.norm(numpy.roll(xy, -1, axis=1) - xy, axis=2)
    longest = e.max(axis=1)
    shortest = numpy.where(e > 1e-9 * longest[:, None], e, numpy.inf).min(axis=1)
    return numpy.abs(xy).max(axis=(1, 2)) / shortest, longest / shortest


def _check_vectors(vec, ids, cond, label):
    """Every vector = (b, -a, 0) to 32 eps x the cell's amplification x (|a|+|b|); on grids of 0.25 degrees and coarser also
    to the flat 1e-12 (|a|+|b|) on every cell that is not slender (edge ratio <= 10)."""
    amp, ratio = cond
    found = ids >= 0
    assert found.all(), f'{label}: {(~found).sum()} of {ids.size} points inside the region were not located'
    err = numpy.abs(vec - numpy.array([B, -A, 0.])).max(axis=1)
    scale = abs(A) + abs(B)
    bound = 32 * numpy.finfo(float).eps * amp[ids] * scale
    compact = ratio[ids] <= 10.
    print(f'{label}: {ids.size} vectors, max err {err.max():.3g}, max err / bound {(err / bound).max():.3g}; cells with edge '
          f'ratio <= 10: {compact.sum()}, max err {err[compact].max():.3g}, median bound there {numpy.median(bound[compact]):.3g}')
    assert numpy.all(err <= bound), f'{label}: {(err / bound).max()}'
    assert compact.sum() > 0.9 * ids.size
    if numpy.median(amp[ids][compact]) <= 400.:         # 0.25-degree cells and coarser (ORCA12: ~900)
        assert err[compact].max() <= 1e-12 * scale, f'{label}: {err[compact].max()}'
    assert numpy.all(vec[:, 2] == 0.)


@pytest.mark.parametrize('kind', ['c2_rotated', 'orca025_real', 'orca12_rotated'])
def test_affine_psi_open_polylines_and_face_vectors(kind):
    from nemoflux_amd import _lib, mint
    from nemoflux_amd.field import _geometry_only
    t0 = time.time()
    blon, blat, box = _case(kind)
    ny, nx = blon.shape[:2]
    geo = _geometry_only(blon, blat)
    pts, arc = geo['points'], geo['arcLengths']
    data = affine_edge_data(pts, A, B, C)
    lines = random_open_polylines(20261005, NLINES, box)
    span = max(box[1] - box[0], box[3] - box[2])
    tol = 1e-12 * (abs(A) + abs(B)) * span
    cond = _cell_condition(pts)

    # ---- Level 1: the mint-shaped C ABI, one PolylineIntegral per line (field.py:43-49 pattern)
    grid = mint.Grid()
    grid.setPoints(pts)
    if kind == 'orca12_rotated':
        grid.setRowLength(nx)             # the locator hint; the other two cases run without it
    ddev = _lib.DeviceBuffer(data.nbytes).upload(data)
    worst = worst_seg = 0.
    for k, xyz in enumerate(lines):
        pli = mint.PolylineIntegral()
        pli.setGrid(grid)
        pli.setUnsupportedCells('skip')
        pli.buildLocator(numCellsPerBucket=128, periodX=0., enableFolding=False)
        pli.computeWeights(xyz, counterclock=False)
        cov = pli.getCoverage()
        assert numpy.all(numpy.abs(cov - 1.) <= 1e-9), (kind, k, cov)
        want_seg, want = affine_expected(xyz, A, B)
        got = pli.getIntegral(data, mint.CELL_BY_CELL_DATA)                     # host array, as field.py:102 passes it
        seg, tot = pli.getSegmentIntegrals(ddev)                                # HBM-resident data
        worst = max(worst, abs(got - want), abs(tot - want))
        worst_seg = max(worst_seg, numpy.abs(seg - want_seg).max())
        assert abs(got - want) <= tol and abs(tot - want) <= tol, (kind, k, got, tot, want)
        assert numpy.all(numpy.abs(seg - want_seg) <= tol), (kind, k)
    t1 = time.time()
    rng = numpy.random.default_rng(7)
    targets = numpy.zeros((NPOINTS, 3))
    targets[:, 0] = rng.uniform(box[0], box[1], NPOINTS)
    targets[:, 1] = rng.uniform(box[2], box[3], NPOINTS)
    vi = mint.VectorInterp()
    vi.setGrid(grid)
    vi.buildLocator(numCellsPerBucket=128, periodX=0., enableFolding=False)
    assert vi.findPoints(targets, tol2=1.e-12) == 0
    vec = vi.getFaceVectors(data, placement=mint.CELL_BY_CELL_DATA)             # host array (field.py:119)
    ids, _ = vi.getCells()
    _check_vectors(vec, ids, cond, f'{kind} level 1')
    assert numpy.array_equal(vi.getFaceVectors(ddev), vec)                      # HBM-resident data: same bits
    ddev.free()
    t2 = time.time()

    # ---- Field: uo / vo such that the flux kernel writes the affine edge data (field.py:195-196: eU = U arc_E, eV = -V arc_N)
    U, V = _field_uv(data, arc, ny, nx)
    fld = _quiet_field(blon, blat, numpy.array([[0., 1.]]), U, V, lines, periodX=0., unsupportedCells='skip')   # no coverage warning
    tot, segs = fld.computeAll()
    assert tot.shape == (1, NLINES)
    off = 0
    fworst = fworst_seg = 0.
    for k, xyz in enumerate(lines):
        want_seg, want = affine_expected(xyz, A, B)
        n = want_seg.size
        fworst = max(fworst, abs(tot[0, k] - want))
        fworst_seg = max(fworst_seg, numpy.abs(segs[0, off:off + n] - want_seg).max())
        off += n
    assert off == segs.shape[1]
    assert fworst <= tol and fworst_seg <= tol, (kind, fworst, fworst_seg, tol)
    # the per-step path fluxviz drives (update -> getIntegral on the host array, field.py:98-103) and the arrows (field.py:119)
    fld.update()
    for k in (0, 1, NLINES - 1):
        assert abs(fld.plis[k].getIntegral(fld.integratedVelocity) - affine_expected(lines[k], A, B)[1]) <= tol
    inner = numpy.ones(ny * nx, bool)
    if kind == 'orca025_real':
        inner.reshape(ny, nx)[0, :] = False       # quirks 1, 2: not this cell's own value
        inner.reshape(ny, nx)[:, 0] = False
    # (the few cells that reach around a geographic pole span > 90 degrees of longitude and do not sit on their neighbours'
    # branch: their slots, and the south / west slots copied from them, are not comparable -- no line goes there)
    wide = (numpy.ptp(pts[:, :, 0], axis=1) > 90.).reshape(ny, nx)
    assert wide.sum() <= 4 * (ny + nx)
    wide[1:, :] |= wide[:-1, :].copy()
    wide |= numpy.roll(wide, 1, axis=1)
    inner &= ~wide.reshape(-1)
    scale = numpy.abs(data[inner]).max()
    dif = numpy.where(inner[:, None], numpy.abs(fld.integratedVelocity - data), 0.)
    # column 0's west slot is column nx-1's east edge (field.py:223): the generator rotates that meridian once from lon = +180
    # and once from -180, which near a geographic pole differ by 1e-16 / cos(lat) in the rotated longitude (3e-12 degrees at 89.9)
    seam = dif.reshape(ny, nx, 4)[:, 0, 3].copy()
    dif.reshape(ny, nx, 4)[:, 0, 3] = 0.
    assert seam.max() <= 1e-10 * scale
    k = numpy.unravel_index(dif.argmax(), dif.shape)
    assert dif.max() <= 8 * numpy.finfo(float).eps * scale, (k, divmod(int(k[0]), nx), fld.integratedVelocity[k[0]], data[k[0]],
                                                              pts[k[0]], arc[k[0]])
    fids, _ = fld.vinterp.getCells()
    _check_vectors(fld.vectorValues, fids, cond, f'{kind} Field arrows')
    print(f'{kind}: {ny} x {nx}; level 1 worst |err| total {worst:.3g} segment {worst_seg:.3g}; Field total {fworst:.3g} '
          f'segment {fworst_seg:.3g}; tolerance {tol:.3g}; {fld.vectorValues.shape[0]} arrows; '
          f'seconds: level-1 lines {t1 - t0:.1f}, vectors {t2 - t1:.1f}, Field {time.time() - t2:.1f}')


def _field_uv(data, arc, ny, nx, dtype=numpy.float64):
    with numpy.errstate(all='ignore'):
        U = numpy.where(arc[:, 1] > 0, data[:, 1] / arc[:, 1], 0.).reshape(1, 1, ny, nx)
        V = numpy.where(arc[:, 2] > 0, -data[:, 2] / arc[:, 2], 0.).reshape(1, 1, ny, nx)
    return numpy.ascontiguousarray(U.astype(dtype)), numpy.ascontiguousarray(V.astype(dtype))


def _quiet_field(*a, **kw):
    from nemoflux_amd.field import Field
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        fld = Field.fromArrays(*a, **kw)
    assert not [w for w in caught if issubclass(w.category, RuntimeWarning)], [str(w.message) for w in caught]
    return fld


@pytest.mark.parametrize('kind', ['c2_rotated', 'orca025_real'])
def test_affine_psi_float32_inputs_through_field(kind):
    """The same closed form with uo / vo stored as float32 (what real NEMO files hold): the edge data then carry the float32
    rounding of U and V (6e-8 relative), so totals and per-segment sums agree to 5e-7 * (|a|+|b|) * span."""
    from nemoflux_amd.field import _geometry_only
    blon, blat, box = _case(kind)
    ny, nx = blon.shape[:2]
    geo = _geometry_only(blon, blat)
    data = affine_edge_data(geo['points'], A, B, C)
    lines = random_open_polylines(77, NLINES, box)
    tol = 5e-7 * (abs(A) + abs(B)) * max(box[1] - box[0], box[3] - box[2])
    U, V = _field_uv(data, geo['arcLengths'], ny, nx, numpy.float32)
    fld = _quiet_field(blon, blat, numpy.array([[0., 1.]]), U, V, lines, periodX=0., unsupportedCells='skip')
    tot, segs = fld.computeAll()
    want = [affine_expected(xyz, A, B) for xyz in lines]
    err = numpy.abs(tot[0] - numpy.array([w[1] for w in want])).max()
    err_seg = numpy.abs(segs[0] - numpy.concatenate([w[0] for w in want])).max()
    print(f'{kind} float32 uo/vo: worst |err| total {err:.3g} segment {err_seg:.3g} (tolerance {tol:.3g})')
    assert err <= tol and err_seg <= tol
    assert err > 0.            # it IS the float32 path (the float64 one is exact to 1e-11)


@pytest.mark.parametrize('kind', ['wrap36', 'c2_rotated_wrapped'])
def test_affine_psi_dateline_wrapped_periodic(kind):
    """Cell bounds wrapped per corner into [-180, 180) the way a global NEMO T-file stores them (horizgrid.py:17-24 hands
    them to mint as they are), periodX = 360: psi = a*lon + b*lat in the stored longitudes is continuous except across
    +-180, so the lines stay 15 degrees away from that seam and everything else is as above.  'wrap36' is the geometry of the
    reference-generated golden case wrap36_zt (global 36 x 18 on [0, 360]); 'c2_rotated_wrapped' is the C2 rotated grid with
    every corner wrapped.  Edge data are taken on each cell's own continuous branch (corners within 180 degrees of corner 0)."""
    from nemoflux_amd import mint
    from nemoflux_amd.field import _geometry_only
    from conftest import wrap180
    if kind == 'wrap36':
        g = load_golden('wrap36_zt')
        blon, blat = g['bounds_lon'].astype(numpy.float64), g['bounds_lat'].astype(numpy.float64)
        box = (-165., 165., -75., 75.)          # row 0's south edge is never filled by Field (quirk 1): keep off it
    else:
        blon, blat = _rotated_bounds(360, 180)
        blon = wrap180(blon)
        box = (-165., 165., -85., 85.)
    ny, nx = blon.shape[:2]
    geo = _geometry_only(blon, blat)
    pts, arc = geo['points'], geo['arcLengths']
    assert numpy.array_equal(pts[:, :, 0], blon.reshape(-1, 4))            # stored as given: the engine unwraps on the fly
    branch = pts.copy()
    branch[:, :, 0] -= 360. * numpy.round((pts[:, :, 0] - pts[:, :1, 0]) / 360.)
    assert (numpy.abs(branch[:, :, 0] - pts[:, :, 0]).max(axis=1) > 0).sum() >= ny // 2      # some cells do straddle the cut
    data = affine_edge_data(branch, A, B, C)
    lines = random_open_polylines(4242, NLINES, box)
    tol = 1e-12 * (abs(A) + abs(B)) * max(box[1] - box[0], box[3] - box[2])
    grid = mint.Grid()
    grid.setPoints(pts)
    worst = 0.
    for k, xyz in enumerate(lines):
        pli = mint.PolylineIntegral()
        pli.setGrid(grid)
        pli.setUnsupportedCells('skip')
        pli.buildLocator(numCellsPerBucket=128, periodX=360., enableFolding=False)
        pli.computeWeights(xyz, counterclock=False)
        assert numpy.all(numpy.abs(pli.getCoverage() - 1.) <= 1e-9), (kind, k)
        want_seg, want = affine_expected(xyz, A, B)
        seg, tot = pli.getSegmentIntegrals(data)
        got = pli.getIntegral(data)
        worst = max(worst, abs(got - want), numpy.abs(seg - want_seg).max())
        assert abs(got - want) <= tol and abs(tot - want) <= tol and numpy.all(numpy.abs(seg - want_seg) <= tol), (kind, k)
    rng = numpy.random.default_rng(8)
    targets = numpy.zeros((20000, 3))
    targets[:, 0], targets[:, 1] = rng.uniform(box[0], box[1], 20000), rng.uniform(box[2], box[3], 20000)
    vi = mint.VectorInterp()
    vi.setGrid(grid)
    vi.buildLocator(numCellsPerBucket=128, periodX=360., enableFolding=False)
    assert vi.findPoints(targets, tol2=1.e-12) == 0
    ids, _ = vi.getCells()
    cond = _cell_condition(branch)
    _check_vectors(vi.getFaceVectors(data), ids, cond, f'{kind} level 1')
    # Field, periodX = 360
    U, V = _field_uv(data, arc, ny, nx)
    fld = _quiet_field(blon, blat, numpy.array([[0., 1.]]), U, V, lines, periodX=360., unsupportedCells='skip')
    tot, segs = fld.computeAll()
    want = [affine_expected(xyz, A, B) for xyz in lines]
    err = numpy.abs(tot[0] - numpy.array([w[1] for w in want])).max()
    err_seg = numpy.abs(segs[0] - numpy.concatenate([w[0] for w in want])).max()
    assert err <= tol and err_seg <= tol, (kind, err, err_seg)
    fids, _ = fld.vinterp.getCells()
    _check_vectors(fld.vectorValues, fids, cond, f'{kind} Field arrows')
    print(f'{kind}: level 1 worst |err| {worst:.3g}; Field total {err:.3g} segment {err_seg:.3g}; tolerance {tol:.3g}')


def _nodal_case(nx, ny, seed):
    """Random stream function on the LOGICAL nodes of a rotated global grid -- single-valued on the sphere: periodic across the
    seam, one value along each pole row (all its nodes are one physical point) -- as cell-by-cell edge differences, plus the
    planar (lon, lat) of every node as the cells store it (corner 0 of its cell; last row / column from corners 3 / 1)."""
    blon, blat = _rotated_bounds(nx, ny)
    rng = numpy.random.default_rng(seed)
    psi = rng.standard_normal((ny + 1, nx + 1))
    psi[:, -1] = psi[:, 0]
    psi[0, :] = psi[0, 0]
    psi[-1, :] = psi[-1, 0]
    p0, p1, p2, p3 = psi[:-1, :-1], psi[:-1, 1:], psi[1:, 1:], psi[1:, :-1]
    data = numpy.ascontiguousarray(numpy.stack([p1 - p0, p2 - p1, p2 - p3, p3 - p0], axis=-1).reshape(-1, 4))
    return blon, blat, psi, data, rng


@pytest.mark.parametrize('nx,ny', [(72, 36), (360, 180)])
def test_random_nodal_psi_between_nodes_of_rotated_grids(nx, ny):
    """The second oracle-free closed form on curvilinear cells: ANY nodal psi is reproduced by the bilinear interpolants, so for
    edge data = nodal differences the flux across a polyline that starts and ends on grid NODES is psi(end) - psi(start),
    whatever its interior vertices (README.md:45,58) -- arbitrary conforming edge data, not only an affine field.  On the
    un-rotated grid this was tested since round 1; on the rotated grid only closed loops (-> 0) were, because the nodes
    were thought of in logical coordinates: but every node's planar (lon, lat) is a corner of a cell.  periodX = 360 (psi is
    periodic and the cells sit on their own branches); Level 1 and Field."""
    from nemoflux_amd import mint
    from nemoflux_amd.field import _geometry_only
    blon, blat, psi, data, rng = _nodal_case(nx, ny, 2026 + nx)
    geo = _geometry_only(blon, blat)
    pts = geo['points']
    cell = pts.reshape(ny, nx, 4, 3)
    lines, want = [], []
    while len(lines) < 100:
        ja, jb = rng.integers(ny // 12 + 1, ny - ny // 12, 2)       # clear of the rotated poles' own rows
        ia, ib = rng.integers(0, nx, 2)
        a, b = cell[ja, ia, 0, :2], cell[jb, ib, 0, :2]              # node (j, i) = corner 0 of cell (j, i)
        if max(abs(a[1]), abs(b[1])) > 84.:
            continue
        m = int(rng.integers(0, 4))
        xyz = numpy.zeros((m + 2, 3))
        xyz[0, :2], xyz[-1, :2] = a, b
        xyz[1:-1, 0], xyz[1:-1, 1] = rng.uniform(-175., 175., m), rng.uniform(-84., 84., m)
        lines.append(xyz)
        want.append(psi[jb, ib] - psi[ja, ia])
    want = numpy.array(want)
    # a line crosses up to ~(nx + ny) cells, each adding a rounding error of ~1e-13 |psi differences| (|psi| ~ 1): measured
    # 3.5e-13 on 72 x 36 and 8.6e-12 on 360 x 180
    tol = 1e-13 * (nx + ny)
    grid = mint.Grid()
    grid.setPoints(pts)
    worst = 0.
    for k, xyz in enumerate(lines):
        pli = mint.PolylineIntegral()
        pli.setGrid(grid)
        pli.buildLocator(numCellsPerBucket=128, periodX=360., enableFolding=False)
        pli.computeWeights(xyz, counterclock=False)
        assert numpy.all(numpy.abs(pli.getCoverage() - 1.) <= 1e-9), (k, pli.getCoverage())
        got = pli.getIntegral(data)
        worst = max(worst, abs(got - want[k]))
        assert abs(got - want[k]) <= tol, (k, got, want[k])
    U, V = _field_uv(data, geo['arcLengths'], ny, nx)
    fld = _quiet_field(blon, blat, numpy.array([[0., 1.]]), U, V, lines, periodX=360., unsupportedCells='skip')
    tot, _ = fld.computeAll()
    ferr = numpy.abs(tot[0] - want).max()
    print(f'rotated {nx} x {ny}, random nodal psi, 100 node-to-node lines: level 1 worst |err| {worst:.3g}, Field {ferr:.3g}')
    assert ferr <= tol


@pytest.mark.parametrize('kind', ['c2_rotated', 'orca025_real', 'orca12_rotated'])
def test_face_vectors_are_the_curl_of_the_bilinear_psi(kind):
    """Oracle-free, NON-constant data: for edge data = nodal differences of any nodal psi the face-vector interpolation must
    return the curl of the cell's bilinear psi_h at the target point.  The points are made FORWARD from chosen (cell, xi, eta),
    so findPoints has to return exactly that cell and those parametric coordinates (an inverse-bilinear check that shares no
    text with the device's Newton), and getFaceVectors the closed form of conftest.curl_of_bilinear_case (gradient form
    against the engine's Piola form).  field.py:90-95,119-120."""
    from conftest import curl_of_bilinear_case
    from nemoflux_amd import mint
    from nemoflux_amd.field import _geometry_only
    blon, blat, _ = _case(kind)
    ny, nx = blon.shape[:2]
    pts = _geometry_only(blon, blat)['points']
    rng = numpy.random.default_rng(31)
    rotated = kind != 'orca025_real'
    if rotated:       # nodal psi on the logical mesh, single-valued on the sphere
        psi = rng.standard_normal((ny + 1, nx + 1))
        psi[:, -1] = psi[:, 0]
        psi[0, :], psi[-1, :] = psi[0, 0], psi[-1, 0]
        pc = numpy.stack([psi[:-1, :-1], psi[:-1, 1:], psi[1:, 1:], psi[1:, :-1]], axis=-1).reshape(-1, 4)
    else:             # the real geometry is conforming too, but nothing here needs that: psi per cell corner
        pc = rng.standard_normal((ny * nx, 4))
    data = numpy.ascontiguousarray(numpy.stack([pc[:, 1] - pc[:, 0], pc[:, 2] - pc[:, 1], pc[:, 2] - pc[:, 3],
                                                pc[:, 3] - pc[:, 0]], axis=1))
    ok = numpy.ptp(pts[:, :, 0], axis=1) < 90.          # not the cells that reach around a geographic pole
    if rotated:
        ok.reshape(ny, nx)[0, :] = ok.reshape(ny, nx)[-1, :] = False      # the degenerate triangles at the rotated poles
    cells, xi, eta, targets, want = curl_of_bilinear_case(pts, pc, rng, NPOINTS, ok)
    grid = mint.Grid()
    grid.setPoints(pts)
    grid.setRowLength(nx)
    vi = mint.VectorInterp()
    vi.setGrid(grid)
    vi.buildLocator(numCellsPerBucket=128, periodX=360. if rotated else 0., enableFolding=False)
    assert vi.findPoints(targets, tol2=1.e-12) == 0
    ids, pcoords = vi.getCells()
    assert numpy.array_equal(ids, cells)
    perr = max(numpy.abs(pcoords[:, 0] - xi).max(), numpy.abs(pcoords[:, 1] - eta).max())
    vec = vi.getFaceVectors(data)
    amp, _ = _cell_condition(pts)
    err = numpy.abs(vec - want).max(axis=1)
    # rounding of the corner coordinates (as in _check_vectors) + what the error of the located (xi, eta) -- measured just above,
    # asserted <= 1e-9 below; 1.3e-10 in the slender cells next to the rotated poles of the ORCA12-size grid -- does to a
    # vector field whose relative variation across a cell is O(1): 4 x that error x |v|
    size = numpy.abs(want).max(axis=1).clip(min=numpy.abs(data[cells]).max(axis=1))
    bound = (64 * numpy.finfo(float).eps * amp[cells] + 4 * perr) * size
    print(f'{kind}: {NPOINTS} points made forward from (cell, xi, eta): cells identical, max |pcoord error| {perr:.3g}, '
          f'vectors max |err| {err.max():.3g} (max |v| {numpy.abs(want).max():.3g}), max err / bound {(err / bound).max():.3g}')
    assert perr <= 1e-9
    assert numpy.all(err <= bound)
