"""Headless time series of the flux across one or more transects: the batch driver of nemoflux/fluxplot.py:18-76.

The reference loops `for itime in range(nt): fld.update(); pli.getIntegral(...)` (fluxplot.py:51-59), one host round
trip per step and per transect.  Here ALL time steps and ALL transects are integrated in one asynchronous pass on
the GPU (Field.computeAll) and only the (nt, ntransect) table comes back.  Plotting uses matplotlib (imported
only with --show): the table is always printed or written as CSV, which is what the plot shows.

    python -m nemoflux_amd.fluxplot -t T.npz -u U.npz -v V.npz -l "[(-100,-80),(100,-80),(0,80)],[...]" [-s] [-o out.csv]
    python -m nemoflux_amd.fluxplot -t T.npz -u U.npz -v V.npz -i "data/nz/*.txt"
"""
import argparse
import glob
import os

import numpy

from . import _expr
from .field import Field
from .latlonreader import LatLonReader


def readTargets(lonLatPoints='', iFiles=''):
    """fluxplot.py:27-46: a list of polylines from a Python-list string or from station files."""
    names = []
    if lonLatPoints:
        pts = _expr.literal(lonLatPoints, 'lonLatPoints')
        if len(pts) and not isinstance(pts[0][0], (list, tuple)):
            pts = [pts]  # README.md:32 passes a single polyline without the outer list
        lonLatZPoints = [numpy.array([(ll[0], ll[1], 0.0) for ll in llp]) for llp in pts]
        names = [f'line{i}' for i in range(len(lonLatZPoints))]
    elif iFiles:
        try:      # fluxplot.py:37 evaluates a Python list of names; a glob pattern or a single name is accepted too
            listOfFiles = _expr.literal(iFiles, 'iFiles')
            if isinstance(listOfFiles, str):
                listOfFiles = [listOfFiles]
        except RuntimeError:
            listOfFiles = sorted(glob.glob(iFiles)) or [iFiles]
        lonLatZPoints = []
        for iFile in listOfFiles:
            ll = LatLonReader(iFile).getLonLats()
            lonLatZPoints.append(numpy.array([(p[0], p[1], 0.) for p in ll]))
            names.append(os.path.basename(iFile))
    else:
        raise RuntimeError('ERROR must provide either iFiles (-i) or lonLatPoints (-l)!')  # fluxplot.py:48
    return lonLatZPoints, names


def fluxSeries(tFile, uFile, vFile, lonLatZPoints, sverdrup=False):
    """(nt, ntransect) total fluxes and the Field (one batched GPU pass over every time step)."""
    # only the totals are wanted: no read-back, and only the signed edge fluxes stay resident (compact mode)
    fld = Field(tFile, uFile, vFile, lonLatZPoints, sverdrup, readback=False, compact=True)
    totals, _ = fld.computeAll()
    return totals, fld


def main(*, tFile, uFile, vFile, lonLatPoints='', iFiles='', sverdrup=False, output='', show=False):
    lonLatZPoints, names = readTargets(lonLatPoints, iFiles)
    print(f'target points:\n {lonLatZPoints}')
    totals, fld = fluxSeries(tFile, uFile, vFile, lonLatZPoints, sverdrup)
    timeVals = [fld.timeObj.getTimeAsDate(t) for t in range(fld.nt)]
    unit = 'Sv' if sverdrup else 'A m^2/s'
    header = 'time,' + ','.join(names)
    lines = [header] + [f'{timeVals[t]},' + ','.join(f'{x:.15g}' for x in totals[t]) for t in range(fld.nt)]
    if output:
        with open(output, 'w') as f:
            f.write(f'# water flow [{unit}]\n' + '\n'.join(lines) + '\n')
    else:
        print(f'# water flow [{unit}]')
        print('\n'.join(lines))
    if show:
        try:
            import matplotlib.pyplot as plt
        except ImportError:
            print('# matplotlib is not installed: no plot')
            return totals
        lineTypes = ['b-', 'm--', 'c-.', 'r:', 'g-', 'k--']  # fluxplot.py:62
        for i, name in enumerate(names):
            plt.plot(timeVals, totals[:, i], lineTypes[i % len(lineTypes)])
        if len(names) > 1:
            plt.legend(names)
        plt.title('Water flow')
        plt.ylabel(unit)
        plt.show()
    return totals


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description='Flux time series across transects (GPU batch driver)')
    ap.add_argument('-t', '--tFile', required=True)
    ap.add_argument('-u', '--uFile', required=True)
    ap.add_argument('-v', '--vFile', required=True)
    ap.add_argument('-l', '--lonLatPoints', default='')
    ap.add_argument('-i', '--iFiles', default='')
    ap.add_argument('-s', '--sverdrup', action='store_true')
    ap.add_argument('-o', '--output', default='')
    ap.add_argument('--show', action='store_true')
    main(**vars(ap.parse_args()))
