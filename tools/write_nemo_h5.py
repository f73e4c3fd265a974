"""Runs under an interpreter with h5py (/opt/conda/bin/python in this image): writes <dir>/T.nc, U.nc, V.nc in the layout
XIOS gives NEMO output (NetCDF-4 = HDF5; float32; uo/vo chunked one level per chunk, shuffle + deflate 4, _FillValue 1e20)
from <dir>/u.npy, v.npy, t.npz.  Helper of tools/filebacked_timing.py."""
import os, sys
import h5py, numpy
d = sys.argv[1]
t = numpy.load(os.path.join(d, 't.npz'))
with h5py.File(os.path.join(d, 'T.nc'), 'w', libver='earliest') as f:
    for k in ('bounds_lon', 'bounds_lat', 'deptht_bounds'):
        f.create_dataset(k, data=t[k])
for name, var in (('U.nc', 'uo'), ('V.nc', 'vo')):
    a = numpy.load(os.path.join(d, var[0] + '.npy'), mmap_mode='r')
    with h5py.File(os.path.join(d, name), 'w', libver='earliest') as f:
        ds = f.create_dataset(var, shape=a.shape, dtype='<f4', chunks=(1, 1) + a.shape[2:], shuffle=True, compression='gzip',
                              compression_opts=4)
        ds.attrs['_FillValue'] = numpy.float32(1.e20)
        for t_ in range(a.shape[0]):
            ds[t_] = a[t_]
