"""Does the relative placement of the u and v arrays matter for K1?  One process, several v views at different byte
offsets inside one over-sized buffer, interleaved timing rounds (tuning aid)."""
import contextlib, ctypes, io, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemoflux_amd._lib import lib, check
from nemoflux_amd.datagen import DataGen
from nemoflux_amd.field import Field
nx, ny, nz, nt = 3600, 1800, 75, 3
offs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '0,256,4096,65536,1048576,69888,0').split(',')]
dg = DataGen(); dg.setSizes(nx, ny, nz, nt); dg.setBoundingBox(-180, 180, -90, 90, 0, 1); dg.build()
n = nt * nz * ny * nx
u = torch.empty(n, dtype=torch.float64, device='cuda').normal_().view(nt, nz, ny, nx)
vbuf = torch.empty(n + (4 << 20), dtype=torch.float64, device='cuda').normal_()
fields = []
for off in offs:
    v = vbuf[off // 8: off // 8 + n].view(nt, nz, ny, nx)
    with contextlib.redirect_stdout(io.StringIO()):
        fields.append(Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, u, v, [], readback=False))
res = [[] for _ in offs]
for r in range(8):
    for i, f in enumerate(fields):
        f.enableKernelTiming(True)
        for t in range(nt):
            check(lib.nf_field_compute_flux(ctypes.byref(f._h), t, None))
        k, ms = f.readKernelTiming()
        if r:
            res[i].append(ms / k)
for off, rr in zip(offs, res):
    print(f'v offset {off:8d} B: median {statistics.median(rr):.4f} ms  min {min(rr):.4f} max {max(rr):.4f}')
