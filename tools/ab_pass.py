"""In-process A/B of WHOLE passes (K1 + K3 for every time step, as bench.py times them) between k_flux variants."""
import contextlib, ctypes, io, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
import bench
# the tuning build of the library (make -C nemoflux_amd/csrc tuning): diagnostic and writer-wave variants exist only there
_tun = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'tuning', 'libnemoflux_amd_tuning.so')
if os.path.exists(_tun):
    os.environ.setdefault('NEMOFLUX_AMD_LIB', _tun)
from nemoflux_amd._lib import lib, check
from nemoflux_amd.datagen import DataGen, STREAM_FUNCTIONS
from nemoflux_amd.field import Field

# runs: "variant[:knob=value[:knob=value]]" separated by commas, e.g. "0:overlap=1,0:overlap=0,13"
variants = (sys.argv[1] if len(sys.argv) > 1 else '0:overlap=1,0:overlap=0').split(',')
nx, ny, nz, nt = 3600, 1800, 75, 6
dg = DataGen(); dg.setSizes(nx, ny, nz, nt); dg.setBoundingBox(-180, 180, -90, 90, 0, 1); dg.build()
dg.applyStreamFunction(STREAM_FUNCTIONS[5])
u, v = dg.computeUVFromPotential()
polys = bench.make_transects(nx, ny, -180., 180., -90., 90., 64)
xyzs = [numpy.array([(x, y, 0.) for x, y in p]) for p in polys]
with contextlib.redirect_stdout(io.StringIO()):
    fld = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, u, v, xyzs, readback=False,
                           stream=torch.cuda.current_stream().cuda_stream)
rows = torch.zeros((nt, fld._rowlen), dtype=torch.float64, device='cuda')
ref = None
res = {k: [] for k in variants}
k1 = {k: [] for k in variants}
for r in range(9):
    for k in (variants if r % 2 == 0 else variants[::-1]):
        parts = k.split(':')
        check(lib.nf_tuning_set(b'flux_variant', int(parts[0])))
        for kv in parts[1:]:
            name, val = kv.split('=')
            check(lib.nf_tuning_set(name.encode(), int(val)))
        fld.enableKernelTiming(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            check(lib.nf_field_compute_all_async(ctypes.byref(fld._h), ctypes.c_void_p(rows.data_ptr())))
        torch.cuda.synchronize()
        nl, kms = fld.readKernelTiming()
        if r:
            res[k].append((time.perf_counter() - t0) / 3 / nt * 1e3)
            k1[k].append(kms / nl)
        got = rows.cpu().numpy().copy()
        if ref is None:
            ref = got
        assert os.environ.get('AB_NOCHECK') or numpy.array_equal(ref, got), 'variants disagree'
for k in variants:
    print(f'{k:24s}: median {statistics.median(res[k]):.4f} ms per time step (K1+K3)  min {min(res[k]):.4f} max {max(res[k]):.4f}'
          f'   K1 events {statistics.median(k1[k]):.4f} ms')
