"""K1 (+K3) on grids smaller than the headline: how much of the HBM roofline does one time step reach at the sizes of
BASELINE's other configurations (C2 360x180x10, C3 1440x1021x75 = ORCA025) and in between?  HIP events per launch
(nf_field_timing), per-step launches (batch_steps = 0) and, for the small ones, the all-steps-in-one-launch form."""
import contextlib, ctypes, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from nemoflux_amd._lib import lib, check
from nemoflux_amd.datagen import DataGen
from nemoflux_amd.field import Field

PSI = "(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))"
TRI = numpy.array([(-100., -80., 0.), (100., -80., 0.), (0., 80., 0.), (-100., -80., 0.)])
sizes = [(360, 180, 10, 20), (720, 360, 31, 8), (1440, 1021, 75, 4), (2160, 1080, 75, 4), (3600, 1800, 75, 4)]
if len(sys.argv) > 1:
    sizes = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]]
for real, es in (('float64', 8), ('float32', 4)):
    for nx, ny, nz, nt in sizes:
        dg = DataGen(real=real); dg.setSizes(nx, ny, nz, nt); dg.setBoundingBox(-180, 180, -90, 90, 0, 1); dg.build()
        dg.applyStreamFunction(PSI); dg.computeUVFromPotential()
        for batch, fs in ((0, 0), (0, 1), (0, -1), (1, -1)):      # two fields per wave / one field per wave / the library's choice
            check(lib.nf_tuning_set(b'batch_steps', batch))
            check(lib.nf_tuning_set(b'field_split', fs))
            with contextlib.redirect_stdout(io.StringIO()):
                f = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, dg.u, dg.v, [TRI], readback=False)
            rows = torch.zeros((nt, f._rowlen), dtype=torch.float64, device='cuda')
            for _ in range(3):
                check(lib.nf_field_compute_all_async(ctypes.byref(f._h), ctypes.c_void_p(rows.data_ptr())))
            torch.cuda.synchronize()
            reps = 10
            f.enableKernelTiming(True, reserve=reps * nt)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                check(lib.nf_field_compute_all_async(ctypes.byref(f._h), ctypes.c_void_p(rows.data_ptr())))
            e1.record()
            torch.cuda.synchronize()
            nl, kms = f.readKernelTiming()
            f.enableKernelTiming(False)
            wall = e0.elapsed_time(e1) / reps
            bytes_step = (2 * es + 64.0 / nz) * nz * ny * nx
            launches_per_pass = nl / reps
            k1 = kms / nl                      # ms per launch (a launch = one step, or all steps when batched)
            steps_per_launch = nt / launches_per_pass
            frac = bytes_step * steps_per_launch / (k1 * 1e-3) / 8e12
            tag = ' (default)' if fs < 0 else ''
            print(f'{real} {nx}x{ny}x{nz}x{nt} {"one launch for all steps" if launches_per_pass < nt else "one launch per step":25s} field_split={fs}{tag}: '
                  f'K1 {k1 / steps_per_launch * 1e3:8.1f} us per step = {frac:.3f} of peak; pass {wall * 1e3:8.1f} us = '
                  f'{bytes_step * nt / (wall * 1e-3) / 8e12:.3f} of peak on the wall', flush=True)
            del f
        del dg
        torch.cuda.empty_cache()
check(lib.nf_tuning_set(b'batch_steps', 1))
check(lib.nf_tuning_set(b'field_split', -1))
