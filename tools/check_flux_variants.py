"""Bit-identity of K1's alternative forms against the default kernel: `python tools/check_flux_variants.py 3 4 6 ...`.
Run against the shipped library it covers the store forms (0 / 5; any other number falls through to the default);
tests/test_gpu_configs.py::test_tuning_build_variants_bit_identical runs it against the tuning build
(NEMOFLUX_AMD_LIB=build/tuning/libnemoflux_amd_tuning.so, `make -C nemoflux_amd/csrc tuning`), which holds the measured
alternatives of the load loop (4 / 8 / 16 levels per batch, 2 chunks per lane, temporal loads, the nested loop)."""
import contextlib
import io
import os
import sys

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def check_variants(variants, real='float64'):
    from nemoflux_amd._lib import lib, check
    from nemoflux_amd.datagen import DataGen
    from nemoflux_amd.field import Field
    dg = DataGen(real=real)
    dg.setSizes(360, 180, 11, 2)
    dg.setBoundingBox(-180., 180., -90., 90., 0., 1.)
    dg.build()
    dg.rotatePole((20., 30.))
    dg.applyStreamFunction("(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))")
    dg.computeUVFromPotential()
    tri = numpy.array([(-100., -50., 0.), (100., -50., 0.), (0., 50., 0.), (-100., -50., 0.)])
    args = (dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, dg.u, dg.v, [tri])
    for variant in variants:
        try:
            with contextlib.redirect_stdout(io.StringIO()):      # a fresh pair: maxAbsFlux is monotone over a field's life
                ref = Field.fromArrays(*args)
            check(lib.nf_tuning_set(b'flux_variant', variant))
            with contextlib.redirect_stdout(io.StringIO()):
                alt = Field.fromArrays(*args)
            for t in range(2):
                check(lib.nf_tuning_set(b'flux_variant', 0))
                a = ref.computeFlux(t, readback=True)
                check(lib.nf_tuning_set(b'flux_variant', variant))
                b = alt.computeFlux(t, readback=True)
                assert a == b, (variant, t)
                assert numpy.array_equal(ref.integratedVelocity, alt.integratedVelocity), (variant, t)
                assert numpy.array_equal(ref.edgeFluxesUArray, alt.edgeFluxesUArray), (variant, t)
                assert numpy.array_equal(ref.edgeFluxesVArray, alt.edgeFluxesVArray), (variant, t)
                assert ref.maxAbsFlux == alt.maxAbsFlux, (variant, t)
        finally:
            check(lib.nf_tuning_set(b'flux_variant', 0))
    return len(variants)


if __name__ == '__main__':
    vs = [int(x) for x in sys.argv[1:]] or [5]
    for real in ('float64', 'float32'):
        check_variants(vs, real)
    print(f'flux variants {vs} bit-identical to the default (float64, float32) with {os.environ.get("NEMOFLUX_AMD_LIB", "the shipped library")}')
