"""HorizGrid: same surface as nemoflux/horizgrid.py:8-43, geometry assembled on the GPU."""
import ctypes

import numpy

from . import _lib, mint
from ._lib import lib, check
from .io import open_tfile


class HorizGrid(object):

    def __init__(self, tFile=None, bounds_lon=None, bounds_lat=None, _field=None):
        """tFile: T-grid file (see nemoflux_amd.io) or pass bounds arrays (ny,nx,4) directly."""
        if _field is not None:
            # share the corner table a Field already holds in HBM
            self._fieldref = _field
            h = ctypes.c_void_p()
            check(lib.nf_field_grid(ctypes.byref(_field._h), ctypes.byref(h)))
            self.grid = mint.Grid._view(h.value, _field)
            self.ny, self.nx = _field.ny, _field.nx
            self._points = None
            return
        if tFile is not None:
            t = open_tfile(tFile)
            bounds_lat, bounds_lon = t['bounds_lat'], t['bounds_lon']
        bounds_lon = numpy.ascontiguousarray(bounds_lon)
        bounds_lat = numpy.ascontiguousarray(bounds_lat)
        ny, nx, nvertex = bounds_lat.shape  # horizgrid.py:17
        if nvertex != 4 or bounds_lon.shape != bounds_lat.shape:
            raise RuntimeError('ERROR: bounds_lat/bounds_lon must have shape (ny, nx, 4)')
        self.ny, self.nx = ny, nx
        self._fieldref = None
        # cell-bounds assembly (horizgrid.py:19-22) on the device
        from .field import _geometry_only
        self._points = _geometry_only(bounds_lon, bounds_lat)['points']
        self.grid = mint.Grid()
        self.grid.setPoints(self._points)  # horizgrid.py:23-24
        self.grid.setRowLength(nx)         # hint for the locator (the reference's mint.Grid has no such call; optional)

    @property
    def points(self):
        if self._points is None:
            self._points = self._fieldref._download_points()
        return self._points

    def getMintGrid(self):
        return self.grid

    def getNumCells(self):
        return self.grid.getNumberOfCells()

    def getPoints(self):
        return self.points

    def getPoint(self, cellId, vertex):
        return self.points[cellId, vertex, :]

    def dump(self, fileName):
        """Dump the grid data to a VTK file (horizgrid.py:38-43)."""
        self.grid.dump(fileName)


def main(*, tFile):
    """Create the grid and dump it next to the T file as legacy VTK (horizgrid.py:45-52)."""
    import re
    gr = HorizGrid(tFile)
    vtkFile = re.sub(r'\.(nc|npz)$', '.vtk', str(tFile))
    gr.dump(vtkFile)
    return vtkFile


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description='Create grid')
    ap.add_argument('-t', '--tFile', required=True, help='file containing t grid data (NetCDF via xarray, or .npz)')
    main(**vars(ap.parse_args()))
