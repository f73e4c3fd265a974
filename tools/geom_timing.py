"""K0 timing on the ORCA12-like grid (one-off geometry kernel)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nemoflux_amd._lib import lib, check
from nemoflux_amd.datagen import DataGen
dg = DataGen(); dg.setSizes(3600, 1800, 1, 1); dg.setBoundingBox(-180, 180, -90, 90, 0, 1); dg.build()
h = ctypes.c_void_p(); check(lib.nf_field_new(ctypes.byref(h)))
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    check(lib.nf_field_set_bounds(ctypes.byref(h), dg.bounds_lon.data_ptr(), dg.bounds_lat.data_ptr(), 1800, 3600, 0, 1))
    torch.cuda.synchronize(); print(f'set_bounds (allocations + geometry kernel + box): {(time.perf_counter()-t0)*1e3:.2f} ms')
