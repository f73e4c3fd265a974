"""Time the generator's write kernel on the bench workload's shape (round-3 verdict W5: k_uv at 1.88 TB/s):
    python tools/datagen_timing.py [nt]      -> GB written, ms, TB/s for the row kernel and the one-cell-per-lane kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nemoflux_amd._lib import lib, check  # noqa: E402
from nemoflux_amd.datagen import DataGen, STREAM_FUNCTIONS  # noqa: E402

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for real, es in (('float64', 8), ('float32', 4)):
    for rows in (1, 0):
        check(lib.nf_tuning_set(b'datagen_rows', rows))
        dg = DataGen(real=real)
        dg.setSizes(3600, 1800, 75, nt)
        dg.setBoundingBox(-180., 180., -90., 90., 0., 1.)
        dg.build()
        dg.applyStreamFunction(STREAM_FUNCTIONS[5])
        best = 1e30
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            u, v = dg.computeUVFromPotential()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            del u, v
        gb = 2.0 * 3600 * 1800 * 75 * nt * es / 1e9
        print(f'{real} rows={rows}: {gb:.1f} GB in {best * 1e3:.2f} ms (whole call: h, ds, table, launch, sync) = {gb / best / 1e3:.2f} TB/s', flush=True)
        del dg
        torch.cuda.empty_cache()
check(lib.nf_tuning_set(b'datagen_rows', 1))
