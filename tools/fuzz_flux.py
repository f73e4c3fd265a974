"""Differential fuzz of K1 (nf_flux.hip: k_flux, k_flux_field, k_expand_planes, every store form and lane width) against the
CPU oracle, BIT FOR BIT: random shapes (1..70 x 1..40 cells, 1..23 levels, 1..4 steps: odd / even cell counts, rows that end
inside a lane's cells, single rows and columns), float64 / float32, NaN / _FillValue / a second missing marker, Sverdrup
units, compact mode, host-staged or HBM-resident fields, a sharded slab range, per-step launches or all steps in one, the
one-field form forced on or off.     python tools/fuzz_flux.py [ncases] [seed]
Test infrastructure (uses oracle/): never imported by the product."""
import contextlib
import io
import os
import sys

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import nf_oracle as oracle  # noqa: E402

oracle.build()


def run(ncases=200, seed=1, verbose=True):
    import torch
    from nemoflux_amd._lib import lib, check
    from nemoflux_amd.field import Field
    rng = numpy.random.default_rng(seed)
    forms = {}
    try:
        for case in range(ncases):
            nx, ny = int(rng.integers(1, 71)), int(rng.integers(1, 41))
            nz, nt = int(rng.integers(1, 24)), int(rng.integers(1, 5))
            dt = str(rng.choice(['float64', 'float32']))
            o = oracle.DataGen(nx, ny, nz, nt, lat_uses_dx=False)
            u = rng.standard_normal((nt, nz, ny, nx)).astype(dt)
            v = rng.standard_normal((nt, nz, ny, nx)).astype(dt)
            fill, miss = 1.e20, (-999. if rng.random() < 0.5 else float('nan'))
            u[rng.random(u.shape) < 0.05] = numpy.nan
            v[rng.random(v.shape) < 0.05] = fill
            if miss == miss:
                u[rng.random(u.shape) < 0.03] = miss
            th = rng.uniform(0.5, 2.0, nz)
            db = numpy.stack([numpy.zeros(nz), th], axis=1)
            sv = bool(rng.random() < 0.3)
            compact = bool(rng.random() < 0.3)
            resident = bool(rng.random() < 0.7)
            batch = int(rng.random() < 0.5)
            fs = int(rng.choice([-1, 0, 1]))
            sr = None
            if rng.random() < 0.3 and nt * nz > 2:
                a, b = sorted(rng.choice(nt * nz + 1, 2, replace=False).tolist())
                sr = (int(a), int(b))
            check(lib.nf_tuning_set(b'batch_steps', batch))
            check(lib.nf_tuning_set(b'field_split', fs))
            uu, vv = (torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()) if resident else (u, v)
            with contextlib.redirect_stdout(io.StringIO()):
                fld = Field.fromArrays(o.bounds_lon, o.bounds_lat, db, uu, vv, [], fill_value=fill, missing_value=miss,
                                       sverdrup=sv, compact=compact, slab_range=sr)
            if batch:
                fld.computeAll()                     # the one-launch path fills every step's planes first
            st = oracle.EdgeFluxState(ny, nx)
            # missing markers: the oracle takes one fill value; fold the second marker into NaN for it
            uo, vo = u.copy(), v.copy()
            if miss == miss:
                uo[uo == numpy.asarray(miss, dt)] = numpy.nan
            lo, hi = sr if sr is not None else (0, nt * nz)
            for t in range(nt):
                z0, z1 = max(lo, t * nz) - t * nz, min(hi, (t + 1) * nz) - t * nz
                fld.timeIndex = t
                fld.update()
                if z1 <= z0:
                    continue                          # the rank owns nothing of this step: planes keep the previous step
                U = oracle.vertical_integral(uo[t][z0:z1], fld.thickness[z0:z1], fill)
                V = oracle.vertical_integral(vo[t][z0:z1], fld.thickness[z0:z1], fill)
                oracle.edge_flux(st, U, V, fld.arcLengths, sv)
                key = (case, nx, ny, nz, nt, dt, sv, compact, resident, batch, fs, sr, t)
                if not numpy.array_equal(fld.integratedVelocity, st.integratedVelocity, equal_nan=True):   # (acos(1 + ulp) on one-column grids)
                    bad = numpy.argwhere(~((fld.integratedVelocity == st.integratedVelocity) | (numpy.isnan(fld.integratedVelocity) & numpy.isnan(st.integratedVelocity))))
                    print('MISMATCH', key, 'miss', miss, 'entries', bad[:8].tolist(), 'of', bad.shape[0])
                    for c, e in bad[:8].tolist():
                        print('   cell', c, 'edge', e, 'gpu', repr(fld.integratedVelocity[c, e]), 'oracle', repr(st.integratedVelocity[c, e]))
                    raise AssertionError(key)
                assert numpy.array_equal(fld.edgeFluxesUArray, st.edgeFluxesU, equal_nan=True), key
                assert numpy.array_equal(fld.edgeFluxesVArray, st.edgeFluxesV, equal_nan=True), key
                if sr is None and (not batch or t == nt - 1) and not numpy.isnan(st.maxAbsFlux.value):   # (a one-launch pass has already seen every step's maximum)
                    assert fld.maxAbsFlux == st.maxAbsFlux.value, key
            forms[(dt, batch, fs, compact, sr is not None)] = forms.get((dt, batch, fs, compact, sr is not None), 0) + 1
            del fld
            if verbose and case % 100 == 99:
                print(f'{case + 1} cases bit-identical', flush=True)
    finally:
        check(lib.nf_tuning_set(b'batch_steps', 1))
        check(lib.nf_tuning_set(b'field_split', -1))
    if verbose:
        print(f'fuzz OK: {ncases} random (shape, dtype, markers, units, mode) cases, seed {seed}: planes, |.| arrays and max '
              f'bit-identical to the oracle; {len(forms)} distinct (dtype, batch, field_split, compact, sharded) combinations')
    return forms


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
