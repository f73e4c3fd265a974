"""GPU parity of the file-ingest path AT THE SIZE DESIGN.md quotes (section 4, "Ingest"): the C3 grid 1440 x 1021 x 75, float32,
one byte-shuffled + deflated chunk per level like XIOS output -- 5.9 MB per stream, hundreds of streams per launch, groups
of time steps, the pipelined gather.  The toy-size suites (tests/test_gpu_inflate.py, test_file_backed_field_against_the_
oracle) cover the format's corners; this one covers scale and occupancy: the decoder's phases are ordered by the issue
order of one wavefront, which is exactly what must hold with four streams resident per CU for hundreds of milliseconds.

Replaces nemoflux/field.py:149,157 (the lazy NetCDF read + fillna) for real NEMO files.  Everything is compared with zlib's
own inflate + a host un-shuffle, and with the CPU oracle on the decoded values."""
import zlib

import numpy
import pytest

from conftest import deflated_dataset, transect_xyz

pytestmark = pytest.mark.gpu

NX, NY, NZ = 1440, 1021, 75
PSI = "(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))"


def _c3_fields(nt, real='float32'):
    """u, v (nt, 75, 1021, 1440) of the C3 grid with a little noise (realistic entropy for zlib) and a land block"""
    from nemoflux_amd.datagen import DataGen
    dg = DataGen(real=real)
    dg.setSizes(NX, NY, NZ, nt)
    dg.setBoundingBox(-180., 180., -90., 90., 0., 1.)
    dg.build()
    dg.applyStreamFunction(PSI)
    dg.computeUVFromPotential()
    rng = numpy.random.default_rng(1)
    u, v = dg.u.cpu().numpy(), dg.v.cpu().numpy()
    dt = u.dtype
    u *= (1 + dt.type(1e-3) * rng.standard_normal(u.shape, dtype=dt))
    v *= (1 + dt.type(1e-3) * rng.standard_normal(v.shape, dtype=dt))
    v[:, :, -1, :] = 0                           # datagen's pole row is 1e13-sized garbage
    u[:, 30:, 200:400, 300:700] = dt.type(1.e20)  # land below level 30: _FillValue
    v[:, 30:, 200:400, 300:700] = dt.type(1.e20)
    return dg, u, v


@pytest.fixture(scope='module')
def c3():
    dg, u, v = _c3_fields(3)
    lu, su = deflated_dataset(u, 'uo', (1, 1, NY, NX), attrs={'_FillValue': numpy.float32(1.e20)})
    lv, sv = deflated_dataset(v, 'vo', (1, 1, NY, NX), attrs={'_FillValue': numpy.float32(1.e20)})
    assert 1.2 < (u.nbytes + v.nbytes) / (su + sv) < 4.0       # a compression ratio like real output, not a degenerate one
    return dg, u, v, lu, lv


def test_c3_levels_inflate_bit_identical_to_zlib(c3):
    """(a) every device-inflated slab of the 1440 x 1021 x 75 float32 grid -- 450 streams of 5.9 MB in flight at once, both
    variables of a group of three time steps in ONE launch -- is bit-identical to zlib.decompress + a host un-shuffle."""
    from nemoflux_amd._lib import DeviceBuffer
    from nemoflux_amd.ingest import ChunkDecoder
    dg, u, v, lu, lv = c3
    nt = u.shape[0]
    dec = ChunkDecoder()
    # zlib's own answer for a few chunks, un-shuffled on the host: the construction and the reference decoder agree
    for (var, a, t, z) in ((lu, u, 0, 0), (lu, u, 2, 74), (lv, v, 1, 37)):
        addr, size, origin = var.device_plan(t)['chunks'][z]
        raw = numpy.frombuffer(zlib.decompress(bytes(var.raw_bytes()[addr:addr + size])), numpy.uint8)
        assert origin == (z, 0, 0) and numpy.array_equal(raw.reshape(4, -1).T.copy().view('<f4').reshape(NY, NX), a[t, z])
    step_bytes = u[0].nbytes
    need = sum(ChunkDecoder.staging_bytes(x, nt) for x in (lu, lv)) * nt
    pinned = dec.new_pinned(need + 64)
    slab = DeviceBuffer(2 * nt * step_bytes)
    items = [(x.raw_bytes(), x.device_plan(t), (2 * t + k) * NZ) for t in range(nt) for k, x in enumerate((lu, lv))]
    staged = dec.gather_many(items, pinned, 2 * nt * NZ)
    assert len(staged) == 1 and len(staged[0].in_len) == 2 * nt * NZ == 450
    status = dec.decode(staged[0], slab.ptr)
    assert not status.any()
    got = slab.download((nt, 2, NZ, NY, NX), '<f4')
    for t in range(nt):
        assert numpy.array_equal(got[t, 0].view(numpy.uint32), u[t].view(numpy.uint32)), t
        assert numpy.array_equal(got[t, 1].view(numpy.uint32), v[t].view(numpy.uint32)), t
    slab.free()


@pytest.mark.parametrize('prefetch, group', [(True, None), (False, None), (True, 1), (True, 'all'), (False, 'all')])
def test_c3_file_backed_field_against_the_oracle(c3, prefetch, group, oracle, monkeypatch):
    """(b) the file-backed Field on that image (device inflate, groups of time steps, the pipelined gather when prefetch is
    on; with groups of ONE step the staging thread also uploads the next step's compressed bytes while the GPU decodes
    this one) against the CPU ORACLE on the decoded values: every step's full (ncell, 4) field bit for bit, the transect
    totals to rounding -- in file order, out of order, and through computeAll."""
    import contextlib
    import io as _io
    from nemoflux_amd.field import Field
    dg, u, v, lu, lv = c3
    nt = u.shape[0]
    if group == 'all':          # every step of the series in ONE launch of the decoder
        group = nt
    if group is not None:
        monkeypatch.setenv('NF_INFLATE_GROUP', str(group))
    tr = [transect_xyz("(-100,-80),(100,-80),(0,80)"), transect_xyz("(-170,10),(-20,-55),(135,62),(-170,10)")]
    blon, blat = dg.bounds_lon.cpu().numpy(), dg.bounds_lat.cpu().numpy()
    with contextlib.redirect_stdout(_io.StringIO()):
        ff = Field.fromArrays(blon, blat, dg.deptht_bounds, lu, lv, tr, fill_value=1.e20, prefetch=prefetch)
    st = ff._stager
    assert st.on_device and st.comp_bytes[0] is not None and st.comp_bytes[1] is not None and st.group == (group or -(-nt // 4))      # default: at least four groups per series
    pts = oracle.assemble_points(blon, blat)
    th = dg.zbot - dg.ztop
    ows = [oracle.polyline_weights(pts, xyz) for xyz in tr]
    state = oracle.EdgeFluxState(NY, NX)
    fill = float(numpy.float32(1.e20))
    want = {}
    for t in (0, 2, 1):
        oracle.edge_flux(state, oracle.vertical_integral(u[t], th, fill), oracle.vertical_integral(v[t], th, fill), ff.arcLengths)
        got = ff.computeFlux(t, readback=True)
        assert numpy.array_equal(ff.integratedVelocity, state.integratedVelocity), t
        want[t] = numpy.array([oracle.get_integral(w, state.integratedVelocity) for w in ows])
        bound = 1e-12 * max(numpy.abs(w.weight * state.integratedVelocity.reshape(-1)[w.cell_edge]).sum() for w in ows)
        assert numpy.abs(numpy.array(got) - want[t]).max() <= bound, t
    st.invalidate()                              # a second pass over the file: the staging buffers are re-used
    ff._lazy_step = -1
    tot, _ = ff.computeAll()
    if group == 1 and prefetch:                  # steps 1 and 2 were gathered AND uploaded by the staging thread
        assert st._slots[0]['early'] or st._slots[1]['early']
    for t in range(nt):
        assert numpy.abs(tot[t] - want[t]).max() <= 1e-12 * max(1.0, numpy.abs(want[t]).max()) * 10


@pytest.mark.parametrize('real, chunk', [('float32', (1, 26, 511, 724)), ('float64', (1, 13, 300, 362))])
def test_tiled_layouts_with_overhanging_edge_chunks_at_c3_size(real, chunk):
    """(c) chunks that tile z, y and x of the C3 grid with edge chunks hanging over the slab in every direction -- tens of
    megabytes per stream -- float32 (rows a multiple of four elements: the four-elements-per-lane placement) and float64
    (rows that are not: the one-element form): the placed slab equals the source bit for bit."""
    from nemoflux_amd._lib import DeviceBuffer
    from nemoflux_amd.ingest import ChunkDecoder
    dg, u, v = _c3_fields(1, real)
    a = u
    lz, _ = deflated_dataset(a, 'uo', chunk)
    plan = lz.device_plan(0)
    _, cz, cy, cx = chunk
    nchunks = -(-NZ // cz) * -(-NY // cy) * -(-NX // cx)
    assert plan is not None and len(plan['chunks']) == nchunks and plan['chunk_dims'] == (cz, cy, cx)
    assert NZ % cz and NY % cy and NX % cx                        # over-hanging in z, y and x
    dec = ChunkDecoder()
    pinned = dec.new_pinned(ChunkDecoder.staging_bytes(lz, 1) + 64)
    slab = DeviceBuffer(a[0].nbytes)
    check = numpy.uint32 if real == 'float32' else numpy.uint64
    staged = dec.gather(lz.raw_bytes(), plan, pinned)
    assert not dec.decode(staged, slab.ptr).any()
    assert numpy.array_equal(slab.download(a[0].shape, a.dtype).view(check), a[0].view(check))
    slab.free()


def test_more_megabyte_streams_than_resident_wavefronts():
    """(d) a group larger than nf_inflater_capacity: capacity + 300 streams of 1.47 MB in ONE launch -- the four byte planes
    of shuffled float32 levels (noise -> stored blocks, Huffman literals + short matches, long matches) and data that
    repeats with periods of 9 000 .. 30 000 bytes (long-distance matches: the copy's source wraps around the window ring)
    -- late workgroups start while early ones are mid-stream.  The shipped library is built with NFI_WINDOW = 32768 = the
    format's maximum distance (4 streams per CU, capacity 1024), so on the device every match source lies INSIDE the LDS
    window: the decoder's far path (sources already flushed to HBM, which a smaller window needs) is not taken here; it is
    covered by the host build of tests/test_inflate_cpu.py with -DNFI_WINDOW=8192 and, on the device, by the test-only 8 KiB
    build of tests/test_gpu_inflate.py::test_far_path_on_the_device."""
    from nemoflux_amd.ingest import ChunkDecoder
    dec = ChunkDecoder()
    cap = dec.capacity()
    assert cap >= 256
    rng = numpy.random.default_rng(7)
    y = numpy.linspace(-90, 90, NY)[:, None]
    x = numpy.linspace(-180, 180, NX)[None, :]
    datas, streams = [], []
    for k in range(12):
        f = ((numpy.cos(2 * numpy.pi * y / 360) + numpy.sin(2 * numpy.pi * x / 360)) * (1.3 + k)).astype('<f4')
        f *= (1 + 10.0 ** -(2 + k % 4) * rng.standard_normal(f.shape).astype('<f4'))
        sh = numpy.ascontiguousarray(f.view(numpy.uint8).reshape(-1, 4).T)
        for p in range(4):
            datas.append(sh[p].copy())
            streams.append(zlib.compress(sh[p].tobytes(), 1 + (k + p) % 9))
    for period in (9000, 16385, 23456, 30000, 32768):
        d = numpy.tile(rng.integers(0, 256, period, dtype=numpy.uint8), NY * NX // period + 1)[:NY * NX].copy()
        d[::4099] ^= 1                                   # break the matches now and then (literals between far matches)
        datas.append(d)
        streams.append(zlib.compress(d.tobytes(), 6))
    n = cap + 300
    order = [int(i) for i in rng.integers(0, len(streams), n)]
    order[:len(streams)] = range(len(streams))           # every kind at least once
    out = dec.decode_streams([streams[i] for i in order], datas[0].size)
    assert out.shape == (n, NY * NX)
    for j, i in enumerate(order):
        assert numpy.array_equal(out[j], datas[i]), (j, i)
