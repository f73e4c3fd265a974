"""On-device counterpart of nemoflux/datagen.py: same class surface, arrays live in HBM.

    dg = DataGen(prefix); dg.setSizes(nx, ny, nz, nt); dg.setBoundingBox(...); dg.build()
    dg.rotatePole(deltaDeg=(20., 30.)); dg.applyStreamFunction(psi); dg.computeUVFromPotential(); dg.save()

There is no eval() on the GPU: `streamFunction` must be one of the menu entries below (the README's examples
and datagen.py's default), each evaluated on the device in the same operation order as the Python expression.
torch is used for HBM allocation only.
"""
import ctypes
import re

import numpy
import torch

from . import _expr, _lib
from ._lib import lib, check, NF_F64, NF_F32

STREAM_FUNCTIONS = [
    "x",                                                           # README.md:26
    "arctan2(y, x+180)/(2*pi)",                                    # README.md:50
    "cos(2*pi*y/360) + sin(2*pi*x/360)",                           # README.md:65
    "(1+10*z)*(t+1)*(cos(2*pi*y/360) + sin(2*pi*x/360))",          # README.md:89
    "(cos(t*2*pi/nt)+2)*(0.5*(y/180)**2 + sin(2*pi*x/360))",       # datagen.py:211 (default)
    "(1+10*z)*(t+1)*arctan2(y, x+180)/(2*pi)",                     # BASELINE config C4 (modulated singular case)
]


def streamFunctionId(streamFunction):
    if isinstance(streamFunction, int):
        if not 0 <= streamFunction < len(STREAM_FUNCTIONS):
            raise RuntimeError(f'ERROR: stream function id must be in [0, {len(STREAM_FUNCTIONS)})')
        return streamFunction
    key = re.sub(r'\s+', '', streamFunction)
    for i, s in enumerate(STREAM_FUNCTIONS):
        if re.sub(r'\s+', '', s) == key:
            return i
    raise RuntimeError('ERROR: the device generator offers a fixed menu of stream functions:\n  ' +
                       '\n  '.join(STREAM_FUNCTIONS))


class DataGen(object):

    def __init__(self, prefix='', real='float64', lat_uses_dx=None):
        """real: dtype of uo/vo ('float64' like datagen.py:9, or 'float32' like real NEMO files).
        lat_uses_dx: None = reproduce datagen.py:49 (latitude spaced with dx) only when dx == dy, else use dy
        (SURVEY.md 8a quirk 6); True/False forces it."""
        self.prefix = prefix
        self.real = real
        self.lat_uses_dx = lat_uses_dx
        self.deltaDeg = (0., 0.)
        self.psi = None
        self.bounds_lon = self.bounds_lat = self.u = self.v = None

    def setBoundingBox(self, xmin, xmax, ymin, ymax, zmin, zmax):
        self.xmin, self.xmax, self.ymin, self.ymax, self.zmin, self.zmax = map(float, (xmin, xmax, ymin, ymax, zmin, zmax))

    def setSizes(self, nx, ny, nz, nt):
        self.nx, self.ny, self.nz, self.nt = int(nx), int(ny), int(nz), int(nt)

    def _latdx(self):
        if self.lat_uses_dx is not None:
            return 1 if self.lat_uses_dx else 0
        dy, dx = (self.ymax - self.ymin) / float(self.ny), (self.xmax - self.xmin) / float(self.nx)
        return 1 if dx == dy else 0

    def build(self):
        self.buildVertical()
        self.buildUniformHorizontal()

    def buildVertical(self):
        dz = (self.zmax - self.zmin) / float(self.nz)
        self.zhalf = numpy.array([self.zmin + (k + 0.5) * dz for k in range(self.nz)])  # datagen.py:38-40
        self.ztop = numpy.array([self.zmin + (k + 1) * dz for k in range(self.nz)])
        self.zbot = numpy.array([self.zmin + (k + 2) * dz for k in range(self.nz)])

    @property
    def deptht_bounds(self):
        return numpy.stack([self.ztop, self.zbot], axis=1)  # datagen.py:178-180

    def buildUniformHorizontal(self):
        _lib.require_gpu()
        shape = (self.ny, self.nx, 4)
        self.bounds_lon = torch.empty(shape, dtype=torch.float64, device='cuda')
        self.bounds_lat = torch.empty(shape, dtype=torch.float64, device='cuda')
        check(lib.nf_datagen_bounds(self.bounds_lon.data_ptr(), self.bounds_lat.data_ptr(), self.ny, self.nx,
                                    self.xmin, self.xmax, self.ymin, self.ymax, float(self.deltaDeg[0]),
                                    float(self.deltaDeg[1]), self._latdx(), None))

    def rotatePole(self, deltaDeg=(0., 0.)):
        """datagen.py:116-166 (bounds only; psi and ds stay on the logical mesh)."""
        self.deltaDeg = (float(deltaDeg[0]), float(deltaDeg[1]))
        self.buildUniformHorizontal()

    def applyStreamFunction(self, streamFunction):
        self.psi = streamFunctionId(streamFunction)

    def computeUVFromPotential(self, t_begin=0, t_end=None):
        """u, v for time steps [t_begin, t_end) of the nt-step series -> HBM tensors (t_end-t_begin, nz, ny, nx)."""
        _lib.require_gpu()
        if self.psi is None:
            raise RuntimeError('ERROR: applyStreamFunction first')
        t_end = self.nt if t_end is None else t_end
        dt = torch.float64 if self.real == 'float64' else torch.float32
        shape = (t_end - t_begin, self.nz, self.ny, self.nx)
        # one allocation for both fields: their relative placement in HBM (which decides how the two read streams
        # of the flux kernel share channels) is then the same in every process instead of an accident of the allocator
        self._uv = torch.empty((2,) + shape, dtype=dt, device='cuda')
        self.u, self.v = self._uv[0], self._uv[1]
        code = NF_F64 if self.real == 'float64' else NF_F32
        # at most 65535 slabs per launch
        per = max(1, 60000 // self.nz)
        for a in range(t_begin, t_end, per):
            b = min(t_end, a + per)
            off = (a - t_begin)
            check(lib.nf_datagen_uv(self.u[off:].data_ptr(), self.v[off:].data_ptr(), code, a, b, self.nt, self.nz,
                                    self.ny, self.nx, self.xmin, self.xmax, self.ymin, self.ymax, self.zmin,
                                    self.zmax, self._latdx(), self.psi, None))
        return self.u, self.v

    def save(self):
        """<prefix>T.npz, U.npz, V.npz with the variable names of datagen.py:168-208 (this image has no netCDF4)."""
        numpy.savez(self.prefix + 'T.npz', deptht_bounds=self.deptht_bounds,
                    bounds_lat=self.bounds_lat.cpu().numpy(), bounds_lon=self.bounds_lon.cpu().numpy())
        numpy.savez(self.prefix + 'U.npz', uo=self.u.cpu().numpy(), _FillValue_uo=numpy.array(1.e20))
        numpy.savez(self.prefix + 'V.npz', vo=self.v.cpu().numpy(), _FillValue_vo=numpy.array(1.e20))


def main(*, streamFunction="(cos(t*2*pi/nt)+2)*(0.5*(y/180)**2 + sin(2*pi*x/360))", prefix='',
         xmin=-180., xmax=180., ymin=-90., ymax=90., zmin=0., zmax=1.0, nx=36, ny=18, nz=1, nt=1,
         deltaDeg="(0.,0.)"):
    """Same keyword interface as datagen.py:211-242."""
    lldg = DataGen(prefix)
    lldg.setSizes(nx, ny, nz, nt)
    lldg.setBoundingBox(xmin=xmin, xmax=xmax, ymin=ymin, ymax=ymax, zmin=zmin, zmax=zmax)
    lldg.build()
    dd = _expr.literal(deltaDeg, 'deltaDeg') if isinstance(deltaDeg, str) else deltaDeg
    if dd[0] != 0 or dd[1] != 0:
        lldg.rotatePole(deltaDeg=dd)
    lldg.applyStreamFunction(streamFunction)
    lldg.computeUVFromPotential()
    lldg.save()
    return lldg


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser(description='Generate NEMO-like data on the GPU')
    ap.add_argument('--streamFunction', default=STREAM_FUNCTIONS[4])
    ap.add_argument('--prefix', default='')
    for k, d in [('xmin', -180.), ('xmax', 180.), ('ymin', -90.), ('ymax', 90.), ('zmin', 0.), ('zmax', 1.)]:
        ap.add_argument('--' + k, type=float, default=d)
    for k, d in [('nx', 36), ('ny', 18), ('nz', 1), ('nt', 1)]:
        ap.add_argument('--' + k, type=int, default=d)
    ap.add_argument('--deltaDeg', default='(0.,0.)')
    main(**vars(ap.parse_args()))
