#!/usr/bin/env python3
"""The worked examples of the reference's README.md, run end to end on the GPU engine:
datagen (on device) -> Field -> total flux, next to the number the README / fluxexact gives.

    python examples/readme_examples.py
"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy

from nemoflux_amd.datagen import DataGen
from nemoflux_amd.field import Field
from nemoflux_amd.fluxexact import exactFlux

EXAMPLES = [
    # (title, stream function, nx, ny, nz, nt, deltaDeg, transect, expected (None = use fluxexact))
    ("A simple example (README.md:26-39)", "x", 36, 18, 1, 1, (0., 0.),
     [(-180, -70), (-160, -10), (-35, 40), (20, -50), (60, 50), (180, 40)], 360.0),
    ("A singular example (README.md:50-56)", "arctan2(y, x+180)/(2*pi)", 36, 18, 1, 1, (0., 0.),
     [(-180, -80), (-10, -80), (-10, 80), (-180, 80)], 0.5),
    ("A more complex vector field, closed loop (README.md:65-68)", "cos(2*pi*y/360) + sin(2*pi*x/360)", 360, 180, 1, 1,
     (0., 0.), [(-100, -80), (100, -80), (0, 80), (-100, -80)], 0.0),
    ("A curvilinear grid, closed loop (README.md:77-79)", "cos(2*pi*y/360) + sin(2*pi*x/360)", 360, 180, 1, 1,
     (20., 30.), [(-100, -80), (100, -80), (0, 80), (-100, -80)], 0.0),
    ("Adding elevation and depth, un-rotated twin (README.md:89-91)",
     "(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))", 360, 180, 10, 20, (0., 0.),
     [(-100, -80), (100, -80), (0, 80)], None),
]


def run(title, psi, nx, ny, nz, nt, delta, pts, expected):
    dg = DataGen()
    dg.setSizes(nx, ny, nz, nt)
    dg.setBoundingBox(-180., 180., -90., 90., 0., 1.)
    dg.build()
    if delta != (0., 0.):
        dg.rotatePole(delta)
    dg.applyStreamFunction(psi)
    dg.computeUVFromPotential()
    xyz = numpy.array([(x, y, 0.) for x, y in pts], dtype=numpy.float64)
    with contextlib.redirect_stdout(io.StringIO()):
        fld = Field.fromArrays(dg.bounds_lon, dg.bounds_lat, dg.deptht_bounds, dg.u, dg.v, [xyz])
    totals, _ = fld.computeAll()
    exact = numpy.full(nt, expected) if expected is not None else numpy.array(exactFlux(psi, pts, nz, nt))
    err = numpy.abs(totals[:, 0] - exact).max()
    print(f'{title}\n    flux text: {fld.getFluxText()!r}   max |flux - expected| over {nt} step(s): {err:.3g}'
          f'   (expected {exact[0]:.6g}{" ..." if nt > 1 else ""})')
    return err, numpy.abs(exact).max()


def main():
    worst = 0.0
    for ex in EXAMPLES:
        err, scale = run(*ex)
        worst = max(worst, err / max(1.0, scale))
    print(f'worst relative deviation: {worst:.3g}')
    return worst


if __name__ == '__main__':
    main()
