"""CPU suite, part 4: nemoflux_amd/hdf5min.py (the in-process NetCDF-4/HDF5 reader) against files written by the real
HDF5 library through h5py (oracle/gen_hdf5_fixtures.py, run under /opt/conda/bin/python3.9) and against the reference's
own data/sa/T.nc.  Expected values are recomputed from the generator's formula."""
import os

import numpy
import pytest

from conftest import GOLDEN, load_golden

H5 = os.path.join(GOLDEN, 'h5')


def field(shape, dtype, seed):
    n = int(numpy.prod(shape))
    a = numpy.sin(0.37 * numpy.arange(n, dtype=numpy.float64) + seed) * (1 + seed)
    if numpy.dtype(dtype).kind in 'iu':
        a = numpy.floor(a * 1000)
    return a.reshape(shape).astype(dtype)


EXPECTED = {
    'old_style': {'bounds_lon': ((5, 7, 4), '<f8', 1), 'uo': ((3, 4, 9, 11), '<f4', 2), 'vo': ((3, 4, 9, 11), '<f4', 3),
                  'time_counter': ((12,), '>f4', 4), 'index': ((6,), '<i4', 5), 'tiny': ((3,), '<f8', 6),
                  'sub/inner': ((2, 2), '<f8', 7), 'planes': ((2, 3, 4, 6), '<f4', 50)},
    'new_compact': {'bounds_lat': ((5, 7, 4), '<f4', 8), 'deptht_bounds': ((75, 2), '<f4', 9)},
    'new_dense': dict([(f'var{k:02d}', ((4, 3), '<f8', 10 + k)) for k in range(14)] + [('vo', ((2, 3, 8, 6), '<f4', 30))]),
    'latest': {'contig': ((6, 5), '<f8', 40), 'single_chunk': ((6, 5), '<f4', 41), 'many_chunks': ((6, 5), '<f4', 42),
               'fa_filtered': ((3, 2, 9, 7), '<f4', 43), 'fa_paged': ((40, 60), '<f8', 44), 'implicit': ((6, 5), '<f4', 46),
               'ea4': ((4, 3), '<f4', 47), 'ea100': ((100, 3), '<f4', 48), 'ea3000': ((3000, 2), '<f4', 49),
               'ea_filt': ((50, 4, 6), '<f4', 50), 'ea_mid': ((5, 30, 4), '<f4', 51),
               'bt2_small': ((4, 3), '<f4', 52), 'bt2_d1': ((30, 40), '<f4', 53), 'bt2_d2': ((100, 160), '<f4', 54),
               'bt2_filt': ((20, 6, 8), '<f4', 55)},
}


@pytest.mark.parametrize('fname', sorted(EXPECTED))
def test_every_variable_bit_exact(fname):
    from nemoflux_amd import hdf5min
    with hdf5min.File(os.path.join(H5, fname + '.h5')) as f:
        assert set(EXPECTED[fname]) <= set(f.datasets)
        for name, (shape, dt, seed) in EXPECTED[fname].items():
            ds = f.datasets[name]
            a = ds.read()
            want = field(shape, dt, seed)
            assert a.shape == shape and a.dtype == numpy.dtype(dt)
            assert numpy.array_equal(a.astype(a.dtype.newbyteorder('=')), want.astype(want.dtype.newbyteorder('=')))
            if len(shape) >= 2:   # one leading slab at a time == slicing the whole array
                for i in range(shape[0]):
                    assert numpy.array_equal(ds.read_leading(i), a[i])
            with pytest.raises(hdf5min.Hdf5Error):
                ds.read_leading(shape[0])


def test_fill_values_attributes_and_storage_kinds():
    from nemoflux_amd import hdf5min
    f = hdf5min.File(os.path.join(H5, 'old_style.h5'))
    assert f.datasets['uo'].fill_value == numpy.float32(1.e20) and f.datasets['vo'].fill_value is None
    assert f.datasets['bounds_lon'].is_contiguous() and not f.datasets['uo'].is_contiguous()
    a = f.datasets['bounds_lon'].read()
    assert not a.flags.owndata and not a.flags.writeable        # a view of the mapped file, no copy
    assert hdf5min.File(os.path.join(H5, 'new_compact.h5')).datasets['deptht_bounds'].fill_value == numpy.float32(-999.0)
    # > 8 attributes -> dense (fractal heap + v2 B-tree) storage, _FillValue created through a rename
    d = hdf5min.File(os.path.join(H5, 'new_dense.h5')).datasets['vo']
    assert d.fill_value == numpy.float32(1.e20)
    # HDF5 1.10 "latest" chunk indexes: single chunk, implicit, fixed array (plain, filtered, paged), extensible array (one
    # unlimited dimension) and version-2 B-tree (two); never-written chunks / pages read as the fill value
    g = hdf5min.File(os.path.join(H5, 'latest.h5'))
    sparse = numpy.full((40, 60), -1.5)
    sparse[39, 58:60] = field((2,), '<f8', 45)
    assert numpy.array_equal(g.datasets['fa_paged_sparse'].read(), sparse)
    sparse = numpy.full((5000, 2), -2.5)
    sparse[4321], sparse[7] = [1.0, 2.0], [3.0, 4.0]
    assert numpy.array_equal(g.datasets['ea_sparse'].read(), sparse)
    sparse = numpy.full((50, 50), 9.5)
    sparse[47, 3], sparse[0, 49] = 1.0, 2.0
    assert numpy.array_equal(g.datasets['bt2_sparse'].read(), sparse)
    assert not [k for k, v in g.datasets.items() if isinstance(v, hdf5min.Hdf5Error)]     # nothing in the file is refused
    with pytest.raises(hdf5min.Hdf5Error, match='not an HDF5 file'):
        hdf5min.File(os.path.join(GOLDEN, 'cases.json'))


def test_nemo_like_triple_through_io():
    """nemoflux_amd.io on a NetCDF-4-style T/U/V triple: float32, uo chunked+shuffled+deflated (read lazily, one time
    step at a time) with _FillValue among 10 attributes, vo contiguous (zero-copy view) with NaN land."""
    from nemoflux_amd import hdf5min, io
    g = load_golden('def36_zt')
    t = io.open_tfile(os.path.join(H5, 'nemo_T.h5'))
    for k in ('bounds_lon', 'bounds_lat', 'deptht_bounds'):
        assert numpy.array_equal(t[k], g[k].astype(numpy.float32))
    uo, fu = io.open_uvfile(os.path.join(H5, 'nemo_U.h5'), 'uo')
    vo, fv = io.open_uvfile(os.path.join(H5, 'nemo_V.h5'), 'vo')
    assert isinstance(uo, hdf5min.LazyVariable) and uo.shape == (3, 2, 18, 36) and fu == float(numpy.float32(1.e20))
    assert isinstance(vo, numpy.ndarray) and not vo.flags.owndata and numpy.isnan(fv)
    u = g['u'].astype(numpy.float32)
    u[:, :, 4:9, 10:20] = numpy.float32(1.e20)
    for tt in range(3):
        a = uo.read_step(tt)
        assert a.flags.c_contiguous and a.dtype == numpy.float32 and numpy.array_equal(a, u[tt])
    assert numpy.isnan(vo[:, :, 4:9, 10:20]).all()
    with pytest.raises(RuntimeError, match='could not read vo'):
        io.open_uvfile(os.path.join(H5, 'nemo_U.h5'), 'vo')


def _cf_expected(raw, markers, scale, offset, dtype):
    """CF decoding stated independently of nemoflux_amd.io: NaN where the raw value is a marker, else raw*scale + offset"""
    x = raw.astype(dtype)
    if scale is not None:
        x = x * dtype(scale)
    if offset is not None:
        x = x + dtype(offset)
    bad = numpy.zeros(raw.shape, bool)
    for m in markers:
        bad |= raw == m
    x[bad] = numpy.nan
    return x


def test_cf_decoding_like_xarray():
    """What xarray.open_dataset's default decode_cf gives the reference for free (field.py:22-25, 34-35): _FillValue AND a
    differing missing_value both mean 'missing'; packed integers are unpacked as raw*scale_factor + add_offset after the
    masking.  Raw values are pinned by h5py's own read-back (cf_*.raw.npy); xarray itself is not installed (the float type
    it decodes into is restated from memory: parity unpinned)."""
    from nemoflux_amd import hdf5min, io
    # packed int16 with both markers and scale/offset -> a CFDecodedVariable on the host path, masked values NaN
    raw = numpy.load(os.path.join(H5, 'cf_U.h5.uo.raw.npy'))
    assert raw.dtype == numpy.int16 and (raw == -32768).any() and (raw == -32767).any()
    uo, fill, d = io.open_uvfile(os.path.join(H5, 'cf_U.h5'), 'uo', with_all=True)
    assert isinstance(uo, io.CFDecodedVariable) and not hasattr(uo, 'device_plan')
    assert uo.shape == raw.shape and uo.dtype == numpy.float64          # add_offset present: float64
    assert numpy.isnan(fill) and d['_markers_uo'] == []
    want = _cf_expected(raw, (-32768, -32767), numpy.float32(0.002), numpy.float32(1.5), numpy.float64)
    for t in range(raw.shape[0]):
        a = uo.read_step(t)
        assert a.dtype == numpy.float64 and numpy.array_equal(a, want[t], equal_nan=True)
        buf = numpy.empty(raw.shape[1:], numpy.float64)
        assert uo.read_step(t, out=buf) is buf and numpy.array_equal(buf, want[t], equal_nan=True)
    g = load_golden('def36_zt')
    ok = ~numpy.isnan(want)
    assert numpy.abs(want[ok] - g['u'][ok]).max() <= 0.0011             # half a quantum of the packing
    assert io.cf_float_dtype(numpy.int16, False) == numpy.float32 and io.cf_float_dtype(numpy.int32, False) == numpy.float64
    assert io.cf_float_dtype(numpy.float32, True) == numpy.float32
    # float32 with _FillValue 1e20 and missing_value -9999: stays lazy (device-decodable), two markers for the engine
    rawv = numpy.load(os.path.join(H5, 'cf_V.h5.vo.raw.npy'))
    vo, fv, dv = io.open_uvfile(os.path.join(H5, 'cf_V.h5'), 'vo', with_all=True)
    assert isinstance(vo, hdf5min.LazyVariable) and vo.device_plan(0) is not None
    assert fv == float(numpy.float32(1.e20)) and dv['_markers_vo'] == [float(numpy.float32(1.e20)), -9999.0]
    assert numpy.array_equal(vo.read_step(1), rawv[1]) and (rawv == numpy.float32(-9999.)).any()
    # float64 contiguous, missing_value only (no _FillValue): the marker still reaches the engine
    v64, f64, d64 = io.open_uvfile(os.path.join(H5, 'cf_V64.h5'), 'vo', with_all=True)
    assert isinstance(v64, numpy.ndarray) and f64 == -9999.0 and d64['_markers_vo'] == [-9999.0]
    # small T-file variables are decoded whole
    dd = {'bounds_lon': numpy.array([[1, 2], [3, -5]], numpy.int16), '_FillValue_bounds_lon': numpy.int16(-5),
          '_scale_factor_bounds_lon': numpy.float64(0.5)}
    io._cf_apply_whole(dd, 'bounds_lon')
    assert dd['bounds_lon'].dtype == numpy.float32
    assert numpy.array_equal(dd['bounds_lon'], numpy.array([[0.5, 1.0], [1.5, numpy.nan]], numpy.float32), equal_nan=True)


def test_reference_t_file_in_process():
    """The reference's real NetCDF-4 file (superblock 0, v2 object headers, dense links): hdf5min == committed fixture."""
    from nemoflux_amd import hdf5min
    src = '/root/reference/data/sa/T.nc'
    if not os.path.exists(src):
        pytest.skip('reference tree not present')
    d = hdf5min.read_variables(src)
    g = load_golden('sa_T_bounds')
    for k in g.files:
        assert numpy.array_equal(d[k], g[k])
    assert d['time_counter'].shape == (12,) and d['deptht'].shape == (75,)
    with hdf5min.File(src) as f:
        assert f.datasets['deptht'].attrs['units'] == b'm' and f.datasets['deptht'].attrs['axis'] == b'Z'


def test_time_axis_labels():
    """nemoflux/timeobj.py:9-34 on the engine's own decoding: the time variable is found by standard_name / long_name,
    dates come out as year-month-day strings for the calendars NEMO writes; no time variable -> index labels."""
    from nemoflux_amd import io
    from nemoflux_amd.timeobj import TimeObj
    uo, fill, allv = io.open_uvfile(os.path.join(H5, 'nemo_U.h5'), 'uo', with_all=True)
    to = TimeObj.fromVariables(allv)
    assert to.timeVarName == 'time_counter' and to.getSize() == 3
    # 15 days, 45.5 days, 76 days after 1900-01-01 in a 365-day calendar
    assert [to.getTimeAsString(i) for i in range(3)] == ['1900-1-16', '1900-2-15', '1900-3-18']
    assert str(to.getTimeAsDate(0)) == '1900-01-16'
    T = TimeObj
    assert [T([425], 'days since 2000-01-01', c).getTimeAsString(0) for c in ('gregorian', 'noleap', '360_day', 'all_leap')] \
        == ['2001-3-1', '2001-3-2', '2001-3-6', '2001-2-29']
    assert T([3600 * 36], 'seconds since 1950-01-01 12:00:00', 'standard').getTimeAsString(0) == '1950-1-3'
    assert T([1.5], 'hours since 2020-02-28 23:00:00').getTimeAsString(0) == '2020-2-29'
    empty = T.fromVariables(io.open_tfile(os.path.join(H5, 'nemo_T.h5')))
    assert empty.getSize() == 0 and empty.getTimeAsString(4) == '4' and empty.getTimeAsDate(2) == 2
    assert T([59], 'days since 2001-01-01', '360_day').getTimeAsDate(0) == 0   # 30 February has no datetime.date


def test_threaded_and_serial_inflation_agree(monkeypatch):
    """Chunks of one time step are inflated by a thread pool (NF_IO_THREADS); the result does not depend on it."""
    from nemoflux_amd import hdf5min
    got = {}
    for n in ('1', '4'):
        monkeypatch.setenv('NF_IO_THREADS', n)
        assert hdf5min.io_threads() == int(n)
        with hdf5min.File(os.path.join(H5, 'old_style.h5')) as f:
            ds = f.datasets['uo']
            got[n] = (ds.read().copy(), [ds.read_leading(i).copy() for i in range(ds.shape[0])])
    assert numpy.array_equal(got['1'][0], got['4'][0])
    assert all(numpy.array_equal(a, b) for a, b in zip(got['1'][1], got['4'][1]))
    assert numpy.array_equal(got['4'][0], field((3, 4, 9, 11), '<f4', 2))


def test_never_written_chunks_read_as_the_fill_value():
    from nemoflux_amd import hdf5min
    with hdf5min.File(os.path.join(H5, 'old_style.h5')) as f:
        ds = f.datasets['sparse']
        want = numpy.full((4, 6), 7.5)
        want[2:4, 0:3] = field((2, 3), '<f8', 51)
        assert ds.h5fill == 7.5 and numpy.array_equal(ds.read(), want)
        assert numpy.array_equal(ds.read_leading(3), want[3]) and numpy.array_equal(ds.read_leading(0), want[0])


def test_subsetnemo_window_round_trip(tmp_path, capsys):
    """nemoflux_amd.subsetnemo (subsetNEMO.py:6-93): the (j, i) window of a NetCDF-4 style triple, written as the npz
    bundles the engine opens; fill value and the time axis survive."""
    from nemoflux_amd import io, subsetnemo
    from nemoflux_amd.timeobj import TimeObj
    src = {k: os.path.join(H5, f'nemo_{k}.h5') for k in 'TUV'}
    out = subsetnemo.main(tfile=src['T'], ufile=src['U'], vfile=src['V'], outputdir=str(tmp_path / 'sub'),
                          jmin=2, jmax=12, imin=5, imax=25)
    assert 'creating variable uo' in capsys.readouterr().out
    t_full, t_sub = io.open_tfile(src['T']), io.open_tfile(os.path.join(out, 'T.npz'))
    for k in ('bounds_lon', 'bounds_lat'):
        assert numpy.array_equal(t_sub[k], t_full[k][2:12, 5:25])
    assert numpy.array_equal(t_sub['deptht_bounds'], t_full['deptht_bounds'])
    for f, name in (('U', 'uo'), ('V', 'vo')):
        full, fill_full = io.open_uvfile(src[f], name)
        sub, fill_sub, d = io.open_uvfile(os.path.join(out, f + '.npz'), name, with_all=True)
        whole = numpy.stack([full.read_step(t) for t in range(full.shape[0])]) if hasattr(full, 'read_step') else full
        assert sub.dtype == numpy.float32 and numpy.array_equal(sub, whole[..., 2:12, 5:25], equal_nan=True)
        assert fill_sub == fill_full or (numpy.isnan(fill_sub) and numpy.isnan(fill_full))
    _, _, d = io.open_uvfile(os.path.join(out, 'U.npz'), 'uo', with_all=True)
    assert TimeObj.fromVariables(d).getTimeAsString(1) == '1900-2-15'
    with pytest.raises(RuntimeError, match='exceeds'):
        subsetnemo.main(tfile=src['T'], ufile=src['U'], vfile=src['V'], outputdir=str(tmp_path / 'bad'),
                        jmin=0, jmax=1000, imin=0, imax=4)


@pytest.mark.parametrize('version', [1, 2])
def test_netcdf_classic_files_through_io(tmp_path, version):
    """NetCDF-3 classic (CDF-1) and 64-bit-offset (CDF-2) files: big-endian record variables come back one native-order
    time step at a time, fill value and time axis included (nemoflux_amd.io -> scipy.io.netcdf_file)."""
    from conftest import write_classic_triple
    from nemoflux_amd import io
    from nemoflux_amd.timeobj import TimeObj
    g = load_golden('def36_zt')
    paths, u, v = write_classic_triple(tmp_path, g, version)
    t = io.open_tfile(paths['T'])
    for k in ('bounds_lon', 'bounds_lat', 'deptht_bounds'):
        assert t[k].dtype == numpy.float32 and t[k].dtype.isnative and numpy.array_equal(t[k], g[k].astype(numpy.float32))
    uo, fill, d = io.open_uvfile(paths['U'], 'uo', with_all=True)
    assert uo.shape == u.shape and uo.dtype == numpy.float32 and fill == float(numpy.float32(1.e20))
    for s in range(u.shape[0]):
        a = uo.read_step(s)
        assert a.dtype.isnative and a.flags.c_contiguous and numpy.array_equal(a, u[s])
    vo, vfill = io.open_uvfile(paths['V'], 'vo')
    assert numpy.isnan(vfill) and numpy.array_equal(vo.read_step(1), v[1], equal_nan=True)
    assert TimeObj.fromVariables(d).getTimeAsString(1) == '1900-2-15'


def test_paged_extensible_array_blocks(tmp_path):
    """Beyond 131 060 chunks the data blocks of an extensible-array index are paged (one bit per page in the super block).
    The file is too large to commit: it is written here by oracle/gen_hdf5_fixtures.py under an interpreter with h5py."""
    import subprocess
    from nemoflux_amd import hdf5min
    py = '/opt/conda/bin/python3.9'
    path = str(tmp_path / 'big.h5')
    gen = os.path.join(os.path.dirname(GOLDEN), '..', 'oracle', 'gen_hdf5_fixtures.py')
    if not os.path.exists(py) or subprocess.run([py, gen, '--big-extensible', path], capture_output=True).returncode != 0:
        pytest.skip('no interpreter with h5py')
    with hdf5min.File(path) as f:
        assert numpy.array_equal(f.datasets['big'].read()[:, 0], numpy.arange(140000, dtype=numpy.float32))
        want = numpy.full((600000, 1), -7.0, numpy.float32)
        for i, x in ((135000, 1.5), (199999, 2.5), (3, 3.5), (599999, 4.5), (300000, 5.5)):
            want[i] = x
        assert numpy.array_equal(f.datasets['big_sparse'].read(), want)


@pytest.mark.parametrize('libver,seed', [('latest', 11), ('earliest', 12)])
def test_random_datasets_against_h5py(tmp_path, libver, seed):
    """Differential test: 30 random chunked datasets per file (rank 1-4, ragged chunks, gzip / shuffle / fletcher32, both
    byte orders, integer types, 0-4 unlimited dimensions -> every chunk index kind, partly written ones), written by
    the HDF5 library through h5py under its own interpreter, read back by hdf5min: every value, and the last leading
    slab through read_leading.  Skipped where no interpreter with h5py exists."""
    import json
    import subprocess
    from nemoflux_amd import hdf5min
    py = '/opt/conda/bin/python3.9'
    path = str(tmp_path / f'rand_{libver}.h5')
    gen = os.path.join(os.path.dirname(GOLDEN), '..', 'oracle', 'gen_hdf5_fixtures.py')
    if not os.path.exists(py) or subprocess.run([py, gen, '--random', path, str(seed), libver, '30'],
                                                capture_output=True).returncode != 0:
        pytest.skip('no interpreter with h5py')
    with open(path + '.json') as f:
        names = json.load(f)
    native = lambda a: a.astype(a.dtype.newbyteorder('='))
    with hdf5min.File(path) as h5:
        for name in names:
            want = numpy.load(f'{path}.{name}.npy')
            ds = h5.datasets[name]
            assert not isinstance(ds, Exception), (name, ds)
            got = ds.read()
            assert got.shape == want.shape and numpy.array_equal(native(got), native(want)), name
            if want.ndim >= 2:
                assert numpy.array_equal(native(ds.read_leading(want.shape[0] - 1)), native(want[-1])), name
