"""Side measurements quoted in DESIGN.md (not the headline bench): hipGraph replay vs direct launches on the small C2
config, and the PCIe-inclusive rate when uo/vo are handed over as HOST arrays (staged one time step at a time)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from nemoflux_amd.datagen import DataGen
from nemoflux_amd.field import Field

PSI = "(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))"
TRI = numpy.array([(-100., -80., 0.), (100., -80., 0.), (0., 80., 0.), (-100., -80., 0.)])


def gen(nx, ny, nz, nt, delta=(0., 0.)):
    dg = DataGen(); dg.setSizes(nx, ny, nz, nt); dg.setBoundingBox(-180, 180, -90, 90, 0, 1); dg.build()
    if delta != (0., 0.):
        dg.rotatePole(delta)
    dg.applyStreamFunction(PSI); dg.computeUVFromPotential()
    return dg


def quiet(*a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return Field.fromArrays(*a, **k)


# ---- 1. C2: launch-bound small grid
dg = gen(360, 180, 10, 20, (20., 30.))
units = 360 * 180 * 10 * 20
import ctypes
from nemoflux_amd._lib import lib, check
for label, stream, batch in (('per-step launches (80 per pass)', None, 0), ('per-step launches in a hipGraph', torch.cuda.Stream(), 0),
                             ('all steps per launch (4 per pass)', None, 1)):
    check(lib.nf_tuning_set(b'batch_steps', batch))
    ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    with ctx:
        f = quiet(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, dg.u, dg.v, [TRI], readback=False,
                  stream=None if stream is None else stream.cuda_stream)
        out = torch.zeros((20, f._rowlen), dtype=torch.float64, device='cuda')
        for _ in range(3):
            check(lib.nf_field_compute_all_async(ctypes.byref(f._h), ctypes.c_void_p(out.data_ptr())))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 50
        for _ in range(R):
            check(lib.nf_field_compute_all_async(ctypes.byref(f._h), ctypes.c_void_p(out.data_ptr())))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / R
    print(f'C2 360x180x10x20 {label:34s}: {dt*1e3:.3f} ms per 20-step pass = {units/dt:.3e} integrals/s')
check(lib.nf_tuning_set(b'batch_steps', 1))

# ---- 2. host-resident fields (PCIe-inclusive)
nx, ny, nz, nt = 1440, 1021, 75, 2
dg = gen(nx, ny, nz, nt)
uh, vh = dg.u.cpu().numpy(), dg.v.cpu().numpy()
blon, blat = dg.bounds_lon.cpu().numpy(), dg.bounds_lat.cpu().numpy()
for label, (u, v) in (('HBM-resident', (dg.u, dg.v)), ('host arrays (pageable numpy)', (uh, vh))):
    f = quiet(blon, blat, dg.deptht_bounds, u, v, [TRI], readback=False)
    f.computeFlux(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 5
    for r in range(R):
        f.computeFlux(r % nt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    gb = 2 * nz * ny * nx * 8 / 1e9
    print(f'C3 1440x1021x75 f64 {label:30s}: {dt*1e3:.2f} ms/step = {nz*ny*nx/dt:.3e} integrals/s ({gb/dt:.1f} GB/s of u,v)')
