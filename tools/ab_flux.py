"""A/B timing of k_flux variants inside ONE process (interleaved rounds, medians): tuning aid, not product."""
import argparse, contextlib, ctypes, io, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
# the tuning build of the library (make -C nemoflux_amd/csrc tuning): diagnostic and writer-wave variants exist only there
_tun = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'build', 'tuning', 'libnemoflux_amd_tuning.so')
if os.path.exists(_tun):
    os.environ.setdefault('NEMOFLUX_AMD_LIB', _tun)
from nemoflux_amd._lib import lib, check
from nemoflux_amd.datagen import DataGen, STREAM_FUNCTIONS
from nemoflux_amd.field import Field

ap = argparse.ArgumentParser()
ap.add_argument('--variants', default='0,4,21')
ap.add_argument('--rounds', type=int, default=7)
ap.add_argument('--nt', type=int, default=4)
ap.add_argument('--dtype', default='float64')
ap.add_argument('--random', type=int, default=1)
ap.add_argument('--voff', type=int, default=0, help='extra byte offset of v relative to its allocation (16-B multiple)')
ap.add_argument('--nx', type=int, default=3600)
ap.add_argument('--ny', type=int, default=1800)
ap.add_argument('--nz', type=int, default=75)
ap.add_argument('--knobs', default='', help='semicolon list of knob settings per run, e.g. ww_blocks_per_cu=4,pipe_round_robin=1')
a = ap.parse_args()
nx, ny, nz = a.nx, a.ny, a.nz
dg = DataGen(real=a.dtype); dg.setSizes(nx, ny, nz, a.nt); dg.setBoundingBox(-180, 180, -90, 90, 0, 1); dg.build()
dg.applyStreamFunction(STREAM_FUNCTIONS[5])
if a.random:
    dt = torch.float64 if a.dtype == 'float64' else torch.float32
    u = torch.empty((a.nt, nz, ny, nx), dtype=dt, device='cuda').normal_()
    nelem = a.nt * nz * ny * nx
    esz = 8 if a.dtype == 'float64' else 4
    vbuf = torch.empty(nelem + a.voff // esz + 16, dtype=dt, device='cuda').normal_()
    v = vbuf[a.voff // esz: a.voff // esz + nelem].view(a.nt, nz, ny, nx)
    print('u ptr %x v ptr %x diff mod 2MiB %d' % (u.data_ptr(), v.data_ptr(), (v.data_ptr() - u.data_ptr()) % (2 << 20)))
else:
    u, v = dg.computeUVFromPotential()
with contextlib.redirect_stdout(io.StringIO()):
    fld = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, u, v, [], readback=False)
rows = torch.zeros((a.nt, 1), dtype=torch.float64, device='cuda')
variants = [int(x) for x in a.variants.split(',')]
knobsets = [k for k in a.knobs.split(';')] if a.knobs else ['']
runs = [(v, k) for v in variants for k in knobsets]
res = {r: [] for r in runs}
s = 8 if a.dtype == 'float64' else 4
gb = (2 * s + 64.0 / nz) * nz * ny * nx / 1e9
for r in range(a.rounds + 1):
    for vv, ks in runs:
        check(lib.nf_tuning_set(b'flux_variant', vv))
        for kv in [x for x in ks.split(',') if x]:
            name, val = kv.split('=')
            check(lib.nf_tuning_set(name.encode(), int(val)))
        fld.enableKernelTiming(True)
        for t in range(a.nt):
            check(lib.nf_field_compute_flux(ctypes.byref(fld._h), t, None))
        n, ms = fld.readKernelTiming()
        if r > 0:
            res[(vv, ks)].append(ms / n)
for run in runs:
    m = statistics.median(res[run])
    print(f'variant {run[0]:3d} {run[1]:40s}: median {m:.4f} ms  min {min(res[run]):.4f}  max {max(res[run]):.4f}  -> {gb / m * 1e3:.0f} GB/s algorithmic')
