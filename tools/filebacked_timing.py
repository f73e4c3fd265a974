"""File-backed pass on a C3-sized float32 grid (1440 x 1021 x 75, real-NEMO layout: one shuffled + deflated chunk per level):
pipelined (the next step inflates on host threads while the GPU works) against serial.  The files are written by h5py
under /opt/conda (tools/write_nemo_h5.py) from device-generated data with a little noise so that zlib sees realistic
entropy.  Quoted in DESIGN.md; not part of the bench."""
import contextlib, io, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
from nemoflux_amd.datagen import DataGen
from nemoflux_amd.field import Field

nx, ny, nz, nt = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (1440, 1021, 75, 3)))
PSI = "(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))"
dg = DataGen(real='float32'); dg.setSizes(nx, ny, nz, nt); dg.setBoundingBox(-180, 180, -90, 90, 0, 1); dg.build()
dg.applyStreamFunction(PSI); dg.computeUVFromPotential()
tmp = tempfile.mkdtemp(prefix='nf_fb_', dir=os.environ.get('TMPDIR', '/tmp'))
rng = numpy.random.default_rng(1)
u = dg.u.cpu().numpy(); v = dg.v.cpu().numpy()
u *= (1 + 1e-3 * rng.standard_normal(u.shape).astype(numpy.float32))      # realistic entropy for zlib
v *= (1 + 1e-3 * rng.standard_normal(v.shape).astype(numpy.float32))
v[:, :, -1, :] = 0                                                        # the pole row of datagen is 1e13-sized garbage
numpy.save(os.path.join(tmp, 'u.npy'), u); numpy.save(os.path.join(tmp, 'v.npy'), v)
numpy.savez(os.path.join(tmp, 't.npz'), bounds_lon=dg.bounds_lon.cpu().numpy().astype(numpy.float32),
            bounds_lat=dg.bounds_lat.cpu().numpy().astype(numpy.float32), deptht_bounds=dg.deptht_bounds.astype(numpy.float32))
subprocess.check_call(['/opt/conda/bin/python', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'write_nemo_h5.py'), tmp])
sizes = {k: os.path.getsize(os.path.join(tmp, k)) for k in ('U.nc', 'V.nc')}
raw = u.nbytes + v.nbytes
print(f'files: {sizes}, raw {raw/1e6:.0f} MB, ratio {raw/sum(sizes.values()):.2f}')
tri = [numpy.array([(-100., -80., 0.), (100., -80., 0.), (0., 80., 0.)])]
for label, pf in (('serial (inflate, then H2D + kernels)', False), ('pipelined (next step inflates under the GPU work)', True)):
    with contextlib.redirect_stdout(io.StringIO()):
        f = Field(os.path.join(tmp, 'T.nc'), os.path.join(tmp, 'U.nc'), os.path.join(tmp, 'V.nc'), tri, prefetch=pf, readback=False)
    f.computeAll()
    t0 = time.perf_counter()
    tot, _ = f.computeAll()
    dt = time.perf_counter() - t0
    print(f'{label:52s}: {dt/nt*1e3:8.1f} ms per step = {nz*ny*nx*nt/dt:.3e} integrals/s  ({raw/nt/dt*nt/1e9:.2f} GB/s of decoded u,v)  flux {tot[:,0]}')
    del f
# resident reference
with contextlib.redirect_stdout(io.StringIO()):
    f = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda(), tri, readback=False)
f.computeAll(); t0 = time.perf_counter(); f.computeAll(); dt = time.perf_counter() - t0
print(f'{"HBM-resident":52s}: {dt/nt*1e3:8.2f} ms per step = {nz*ny*nx*nt/dt:.3e} integrals/s')
import shutil; shutil.rmtree(tmp)
