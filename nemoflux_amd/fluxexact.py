"""Closed-form transect flux for stream-function data: same arithmetic as nemoflux/fluxexact.py:21-46
(host-side; it is the analytic oracle of the reference, not a kernel).  The potential is an arithmetic expression in
x, y, z, t, nt checked by nemoflux_amd._expr (no bare eval); arctan2 is available so the README's singular case can be
evaluated (the reference imports only pi, cos, sin)."""
import numpy

from . import _expr


def exactFlux(potentialFunction, lonLatPoints, nz, nt, zmin=0., zmax=1.0):
    """[sum_k (psi(end) - psi(beg)) * dz_k for t in range(nt)]  (fluxexact.py:36-46)."""
    xyVals = numpy.array(lonLatPoints, dtype=numpy.float64)
    dz = (zmax - zmin) / float(nz)
    zhalf = numpy.array([zmin + (k + 0.5) * dz for k in range(nz)])
    ztop = numpy.array([zmin + (k + 0) * dz for k in range(nz)])
    zbot = numpy.array([zmin + (k + 1) * dz for k in range(nz)])
    thickness = -(ztop - zbot)  # DEPTH HAS OPPOSITE SIGN TO Z
    xyBeg, xyEnd = xyVals[0, :], xyVals[-1, :]
    psi = _expr.compile_function(potentialFunction)
    out = []
    for t in range(nt):
        # psi at every level at once (same ufuncs element by element), then the reference's running sum over k
        phiA = _expr.evaluate(psi, x=xyBeg[0], y=xyBeg[1], z=zhalf, t=t, nt=nt) + numpy.zeros(nz)
        phiB = _expr.evaluate(psi, x=xyEnd[0], y=xyEnd[1], z=zhalf, t=t, nt=nt) + numpy.zeros(nz)
        flux = 0
        for k in range(nz):
            flux += (phiB[k] - phiA[k]) * thickness[k]
        out.append(float(flux))
    return out


def main(*, potentialFunction="(cos(t*2*pi/nt)+2)*(0.5*(y/180)**2 + sin(2*pi*x/360))", zmin=0., zmax=1.0, nz=5,
         nt=1, deltaDeg="(0.,0.)", lonLatPointsStr):
    """Prints the table of fluxexact.py:35,46."""
    xyVals = numpy.array(_expr.literal(lonLatPointsStr, 'lonLatPointsStr'), dtype=numpy.float64)
    print(f'zmin/zmax = {zmin}/{zmax}')
    print(f'beg/end target points: {xyVals[0, :]} {xyVals[-1, :]}')
    print('time_index                 flux')
    vals = exactFlux(potentialFunction, xyVals, nz, nt, zmin, zmax)
    for t, flux in enumerate(vals):
        print(f'{t:10d} {flux:20.10g}')
    return vals


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--potentialFunction', default="(cos(t*2*pi/nt)+2)*(0.5*(y/180)**2 + sin(2*pi*x/360))")
    ap.add_argument('--zmin', type=float, default=0.)
    ap.add_argument('--zmax', type=float, default=1.)
    ap.add_argument('--nz', type=int, default=5)
    ap.add_argument('--nt', type=int, default=1)
    ap.add_argument('--deltaDeg', default='(0.,0.)')
    ap.add_argument('--lonLatPointsStr', required=True)
    main(**vars(ap.parse_args()))
