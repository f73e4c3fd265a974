"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/nemoflux_amd.h declares; the
host-side argument checking works without a GPU; compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy
import pytest

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'nemoflux_amd.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    names = re.findall(r'\b((?:nf|mnt)_[A-Za-z0-9_]+)\s*\(', txt)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from nemoflux_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 55
    for s in syms:
        assert hasattr(_lib.lib, s), f'{s} declared in include/nemoflux_amd.h but not exported'


def test_header_compiles_as_plain_c(tmp_path):
    import subprocess
    src = tmp_path / 't.c'
    src.write_text('#include "nemoflux_amd.h"\nint main(void){ return NF_OK + MNT_CELL_BY_CELL_DATA; }\n')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), '-c', str(src),
                           '-o', str(tmp_path / 't.o')])


def test_no_gpu_means_loud_failure():
    """On a box without a GPU every compute path raises; on the GPU box this test is a no-op."""
    from nemoflux_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip('GPU present')
    from nemoflux_amd import mint
    from nemoflux_amd.field import Field
    g = mint.Grid()
    pts = numpy.zeros((4, 4, 3))
    with pytest.raises(_lib.NemofluxError, match='no usable AMD GPU'):
        g.setPoints(pts)
    with pytest.raises(_lib.NemofluxError, match='no CPU fallback'):
        Field.fromArrays(numpy.zeros((2, 2, 4)), numpy.zeros((2, 2, 4)), numpy.zeros((1, 2)), numpy.zeros((1, 1, 2, 2)),
                         numpy.zeros((1, 1, 2, 2)), [])
    # the file-ingest decoder runs on the device only: there is no host inflate behind this ABI
    from nemoflux_amd.ingest import ChunkDecoder
    with pytest.raises(_lib.NemofluxError, match='no CPU fallback'):
        ChunkDecoder()
    h, n = ctypes.c_void_p(), ctypes.c_int()
    assert _lib.lib.nf_inflater_new(ctypes.byref(h)) == 0
    ll = (ctypes.c_longlong * 3)(1, 1, 4)
    z = (ctypes.c_longlong * 3)(0, 0, 0)
    one = (ctypes.c_longlong * 1)(8)
    buf = (ctypes.c_ubyte * 16)()
    assert _lib.lib.nf_inflater_run(ctypes.byref(h), buf, 16, z, one, 1, 16, 4, 1, ll, ll, z, ctypes.c_void_p(16), None, None) == 4
    assert b'no CPU fallback' in _lib.lib.nf_last_error()
    assert _lib.lib.nf_inflater_capacity(ctypes.byref(n)) == 4
    assert _lib.lib.nf_inflater_del(ctypes.byref(h)) == 0 and not h.value


def test_argument_errors_without_gpu():
    from nemoflux_amd import _lib, mint
    lib = _lib.lib
    g = mint.Grid()
    with pytest.raises(RuntimeError):
        g.setPoints(numpy.zeros((3, 4, 2)))                     # wrong trailing shape
    with pytest.raises(RuntimeError):
        g.setPoints(numpy.zeros((3, 4, 3), numpy.float32))      # wrong dtype
    assert g.getNumberOfCells() == 0
    pli = mint.PolylineIntegral()
    with pytest.raises(_lib.NemofluxError, match='setGrid first'):
        pli.buildLocator()
    h = ctypes.c_void_p()
    assert lib.nf_field_new(ctypes.byref(h)) == 0
    assert lib.nf_field_set_slab_range(ctypes.byref(h), 5, 2) != 0
    assert b'begin <= end' in lib.nf_last_error()
    assert lib.nf_field_set_sverdrup(ctypes.byref(h), 1) == 0
    n = ctypes.c_int(-1)
    assert lib.nf_field_num_transects(ctypes.byref(h), ctypes.byref(n)) == 0 and n.value == 0
    xyz = numpy.zeros((1, 3))
    assert lib.nf_field_add_transect(ctypes.byref(h), _lib.dptr(xyz), 1, 0, None) != 0   # needs >= 2 points
    assert lib.nf_field_del(ctypes.byref(h)) == 0 and not h.value
    assert lib.nf_version() == 100


def test_null_arguments_are_errors_not_crashes():
    """Every out-parameter is checked: a NULL gives NF_ERR_ARG (1) and a message, never a segfault; no C++ exception
    crosses the ABI (include/nemoflux_amd.h: NF_ERR_HOST)."""
    from nemoflux_amd import _lib
    lib = _lib.lib
    hdr = open(os.path.join(ROOT, 'include', 'nemoflux_amd.h')).read()
    assert '#define NF_ERR_HOST 5' in hdr
    for fn in (lib.mnt_grid_new, lib.mnt_polylineintegral_new, lib.mnt_vectorinterp_new, lib.nf_field_new,
               lib.nf_device_count):
        assert fn(None) == 1 and b'null' in lib.nf_last_error()
    assert lib.nf_malloc(None, 16) == 1 and lib.nf_host_alloc(None, 16) == 1
    assert lib.nf_device_name(None, 0) == 1
    h = ctypes.c_void_p()
    assert lib.nf_field_new(ctypes.byref(h)) == 0
    assert lib.nf_field_get_box(ctypes.byref(h), None, None, None, None) == 1
    assert lib.nf_field_num_transects(ctypes.byref(h), None) == 1
    assert lib.nf_field_row_length(ctypes.byref(h), None) == 1
    assert lib.nf_field_del(ctypes.byref(h)) == 0
    assert lib.nf_field_del(ctypes.byref(h)) == 0          # deleting twice is harmless (handle was nulled)
    assert lib.nf_tuning_set(None, 0) == 1 and lib.nf_tuning_set(b'no_such_knob', 0) == 1
    assert lib.nf_tuning_set(b'batch_cellsteps_m', -1) == 1 and lib.nf_tuning_set(b'batch_cellsteps_m', 32) == 0
    # round-5 entry points: policy switches and hints check their arguments, the scratch release needs no GPU
    g, pl, f = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.mnt_grid_new(ctypes.byref(g)) == 0 and lib.mnt_polylineintegral_new(ctypes.byref(pl)) == 0
    assert lib.nf_field_new(ctypes.byref(f)) == 0
    assert lib.mnt_grid_setRowLength(None, 4) == 1 and lib.mnt_grid_setRowLength(ctypes.byref(g), -1) == 1
    assert lib.mnt_grid_setRowLength(ctypes.byref(g), 3600) == 0 and lib.mnt_grid_setRowLength(ctypes.byref(g), 0) == 0
    assert lib.mnt_polylineintegral_setOverlappingCells(None, 0) == 1
    assert lib.mnt_polylineintegral_setOverlappingCells(ctypes.byref(pl), 2) == 1 and b'0 (refuse) or 1 (warn)' in lib.nf_last_error()
    assert lib.mnt_polylineintegral_setOverlappingCells(ctypes.byref(pl), 1) == 0
    assert lib.nf_field_set_overlapping_cells(ctypes.byref(f), 3) == 1 and lib.nf_field_set_overlapping_cells(ctypes.byref(f), 0) == 0
    assert lib.nf_release_scratch() == 0
    assert lib.mnt_grid_del(ctypes.byref(g)) == 0 and lib.mnt_polylineintegral_del(ctypes.byref(pl)) == 0
    assert lib.nf_field_del(ctypes.byref(f)) == 0
    assert lib.nf_inflater_new(None) == 1 and lib.nf_inflater_capacity(None) == 1
    assert lib.nf_inflater_run(None, None, 0, None, None, 0, 0, 4, 0, None, None, None, None, None, None) == 1


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, 'nemoflux_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')) or fn == 'Makefile':
                txt = open(os.path.join(dirpath, fn)).read()
                assert 'nf_oracle' not in txt.replace('oracle/nf_oracle.c)', ''), fn
                assert 'import oracle' not in txt and 'from oracle' not in txt, fn


def test_stream_function_menu_and_fluxexact():
    from nemoflux_amd.datagen import STREAM_FUNCTIONS, streamFunctionId
    from nemoflux_amd.fluxexact import exactFlux
    assert streamFunctionId('x') == 0
    assert streamFunctionId('arctan2(y,x+180)/(2*pi)') == 1
    assert streamFunctionId(' cos(2*pi*y/360)+sin(2*pi*x/360) ') == 2
    with pytest.raises(RuntimeError):
        streamFunctionId('x**2')
    assert exactFlux('x', [(-180, -70), (180, 40)], 1, 1) == [360.0]
    assert abs(exactFlux(STREAM_FUNCTIONS[1], [(-180, -80), (-180, 80)], 1, 1)[0] - 0.5) < 1e-15


def test_get_sizes_and_flux_text_format():
    """field.py:122-136 shape fallbacks and field.py:103-108 text format, on the host logic alone."""
    import re as _re
    from nemoflux_amd.field import Field
    f = Field.__new__(Field)
    assert f.getSizes((20, 10, 180, 360)) == (20, 10, 180, 360)
    assert f.getSizes((10, 180, 360)) == (1, 10, 180, 360)
    assert f.getSizes((180, 360)) == (1, 1, 180, 360)
    with pytest.raises(RuntimeError):
        f.getSizes((1, 2, 3, 4, 5))
    txt = ''.join(f"{v:4.3g}, " for v in (360.0, -0.318)) + "(Sv) "
    assert _re.sub(r',\s*\(', ' (', txt) == ' 360, -0.318 (Sv) '


def test_netcdf4_ingest_of_the_reference_t_file():
    """nemoflux_amd.io reads the reference's real NetCDF-4 T file (data/sa/T.nc) when some interpreter with h5py exists
    (this image: /opt/conda/bin/python3.9) and agrees with the committed fixture.  Skipped where the reference tree or
    h5py is absent (the GPU box has no /root/reference)."""
    from conftest import load_golden
    from nemoflux_amd import io
    src = '/root/reference/data/sa/T.nc'
    if not os.path.exists(src):
        pytest.skip('reference tree not present')
    try:
        d = io.open_tfile(src)
    except RuntimeError as e:
        pytest.skip(str(e))
    g = load_golden('sa_T_bounds')
    for k in ('bounds_lon', 'bounds_lat', 'deptht_bounds'):
        assert d[k].dtype == numpy.float32 and numpy.array_equal(d[k], g[k])
    with pytest.raises(RuntimeError, match='no such file'):
        io.open_tfile('/nonexistent/T.nc')
    with pytest.raises(RuntimeError, match='could not read uo'):
        io.open_uvfile(src, 'uo')      # field.py:154: the T file holds no velocity


def test_command_line_expressions_are_parsed_not_evaluated():
    """The reference eval()s its -l / --potentialFunction strings; the engine parses literals and compiles only
    arithmetic on x, y, z, t, nt (nemoflux_amd/_expr.py)."""
    from nemoflux_amd import _expr
    from nemoflux_amd.fluxexact import exactFlux
    from nemoflux_amd.fluxplot import readTargets
    assert _expr.literal(' [(-180,-70),(180,40)] ') == [(-180, -70), (180, 40)]
    code = _expr.compile_function('(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))')
    assert _expr.evaluate(code, x=90., y=0., z=0.5, t=1, nt=2) == 6. * 2. * (numpy.cos(0.) + numpy.sin(numpy.pi / 2))
    for bad in ("__import__('os').system('true')", 'x.__class__', "open('f')", '[x for x in (1,)]', "'a'*3",
                'lambda: 1', 'q + 1', 'cos(x, out=y)'):
        with pytest.raises(RuntimeError):
            _expr.compile_function(bad)
        with pytest.raises(RuntimeError):
            exactFlux(bad, [(0, 0), (1, 1)], 1, 1)
    with pytest.raises(RuntimeError):
        _expr.literal("__import__('os')")
    # README.md:32 single polyline and fluxviz.py:378 list of polylines both work
    one, names = readTargets('(-180,-70),(-160,-10),(-35,40)')
    assert len(one) == 1 and one[0].shape == (3, 3) and names == ['line0']
    two, _ = readTargets('[(-180,-70),(-160,-10)],[(0,0),(10,10),(20,0)]')
    assert [a.shape for a in two] == [(2, 3), (3, 3)]
    with pytest.raises(RuntimeError):
        readTargets("__import__('os').getcwd()")


def build_c_client(tmp_path):
    """examples/c_client.c: plain C on the ABI, linked against the in-tree library (no Python, no torch)."""
    import subprocess
    exe = str(tmp_path / 'c_client')
    libdir = os.path.join(ROOT, 'nemoflux_amd')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'c_client.c'), '-L', libdir, '-lnemoflux_amd',
                           '-Wl,-rpath,' + libdir, '-lm', '-o', exe])
    return exe


def test_c_client_links_and_fails_loudly_without_gpu(tmp_path):
    import subprocess
    from nemoflux_amd import _lib
    exe = build_c_client(tmp_path)
    if _lib.device_count() > 0:
        pytest.skip('GPU present: tests/test_gpu_dist.py runs the client')
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and 'no usable AMD GPU' in r.stderr and 'no CPU fallback' in r.stderr


def test_host_unshuffle_helper_matches_numpy():
    """nf_host_unshuffle (file decoding on the host, no GPU involved): the inverse of HDF5's shuffle filter for element
    sizes 1..16, aligned and unaligned destinations."""
    from nemoflux_amd import _lib
    rng = numpy.random.default_rng(7)
    for es in (1, 2, 4, 8, 16, 3):
        for n in (1, 5, 64, 1001):
            elems = rng.integers(0, 256, size=(n, es), dtype=numpy.uint8)
            shuffled = numpy.ascontiguousarray(elems.T).reshape(-1)          # es byte planes of n bytes each
            for shift in (0, 1):
                buf = numpy.zeros(n * es + shift, numpy.uint8)
                dst = buf[shift:]
                assert _lib.lib.nf_host_unshuffle(shuffled.ctypes.data, dst.ctypes.data, n, es) == 0
                assert numpy.array_equal(dst.reshape(n, es), elems)
    assert _lib.lib.nf_host_unshuffle(None, None, 4, 4) == 1 and _lib.lib.nf_host_unshuffle(shuffled.ctypes.data, dst.ctypes.data, 4, 0) == 1


def test_host_gather_helper():
    """nf_host_gather (file staging on the host, no GPU involved): n byte ranges copied by native threads -- ragged lengths,
    empty ranges, more threads than ranges, one thread; bad arguments are errors."""
    import ctypes
    from nemoflux_amd._lib import lib
    rng = numpy.random.default_rng(11)
    src = rng.integers(0, 256, 1 << 20, dtype=numpy.uint8)
    for n, threads in ((1, 4), (7, 3), (900, 8), (900, 1), (3, 64)):
        lens = rng.integers(0, 3000, n).astype(numpy.int64)
        lens[rng.integers(0, n)] = 0
        offs = rng.integers(0, src.size - 3000, n)
        dst = numpy.full(int(lens.sum()) + 8 * n + 16, 0xAB, numpy.uint8)
        pos, dpos = 0, []
        for ln in lens:
            dpos.append(pos)
            pos += (int(ln) + 7) & ~7
        sa = (src.ctypes.data + offs).astype(numpy.uint64)
        da = (dst.ctypes.data + numpy.array(dpos)).astype(numpy.uint64)
        assert lib.nf_host_gather(sa.ctypes.data, da.ctypes.data, lens.ctypes.data, n, threads) == 0
        want = numpy.full_like(dst, 0xAB)
        for o, p, ln in zip(offs, dpos, lens):
            want[p:p + ln] = src[o:o + ln]
        assert numpy.array_equal(dst, want), (n, threads)
    assert lib.nf_host_gather(None, None, None, 0, 1) == 0
    assert lib.nf_host_gather(None, None, None, 5, 1) == 1 and lib.nf_host_gather(sa.ctypes.data, da.ctypes.data, lens.ctypes.data, n, 0) == 1


def test_python_overlap_predicate_is_the_engines():
    """nemoflux_amd._lib.over_covered = over_covered of csrc/nf_common.h (and of the oracle): coverage > 1 + 1e-8 AND the
    excess as a length > 1e-9 max(1, |coordinates|) degrees.  The Python warnings of policy 'warn' use it (round-5 advisor)."""
    from nemoflux_amd._lib import over_covered
    assert not over_covered(1.0 + 5e-9, (0., 0.), (10., 0.))                 # below the coverage tolerance
    assert over_covered(1.5, (0., 0.), (10., 0.))                            # half of a 10-degree segment counted twice
    assert not over_covered(1.00003, (170., 40.), (170. + 2e-9, 40.))        # 3e-5 of a 2e-9-degree segment: rounding noise
    assert over_covered(2.0, (170., 40.), (170. + 2e-6, 40.))                # a whole tiny segment twice: 2e-6 > 1e-9 * 170
    assert not over_covered(2.0, (170., 40.), (170. + 1e-8, 40.))            # 1e-8 of line < 1e-9 * 170 degrees
    assert not over_covered(float('nan'), (0., 0.), (1., 0.))
