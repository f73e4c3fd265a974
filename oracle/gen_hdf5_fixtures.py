#!/usr/bin/env python
"""Write the small HDF5 / NetCDF-4-like files that pin nemoflux_amd/hdf5min.py (test infrastructure).

Needs h5py (this image: /opt/conda/bin/python3.9 oracle/gen_hdf5_fixtures.py).  Every array is a deterministic
function of its shape (see `field`), so the tests recompute the expected values instead of storing them twice.
The files cover the HDF5 structures netCDF-4 / XIOS / h5py produce for NEMO-like data:
  old_style.h5      libver earliest: symbol-table groups, v1 object headers; contiguous f64, chunked+gzip+shuffle f32
                    with ragged edge chunks (4-D like uo), whole-plane chunks, a partly written chunked variable,
                    big-endian f4, int32, compact storage, _FillValue attributes
  new_compact.h5    creation-order tracking: v2 object headers, compact Link messages
  new_dense.h5      > 8 links and > 8 attributes: fractal-heap (dense) link and attribute storage, fletcher32
  latest.h5         libver latest: superblock v3, layout message v4: single-chunk, implicit and fixed-array chunk indexes
                    (plain, filtered, paged, partly written), extensible-array indexes (one unlimited dimension) and
                    version-2 B-tree indexes (two unlimited dimensions; depth 0, 1 and 2)
"""
import os
import sys

import h5py
import numpy

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'h5')


def field(shape, dtype, seed):
    n = int(numpy.prod(shape))
    a = numpy.sin(0.37 * numpy.arange(n, dtype=numpy.float64) + seed) * (1 + seed)
    if numpy.dtype(dtype).kind in 'iu':
        a = numpy.floor(a * 1000)
    return a.reshape(shape).astype(dtype)


def big_extensible(path):
    """140 000 + 600 000 one-element chunks: data blocks of the extensible array become PAGED beyond 131 060 elements
    (too large to commit: tests/test_hdf5min.py writes it to a temporary directory when an h5py interpreter exists)."""
    with h5py.File(path, 'w', libver='latest') as f:
        f.create_dataset('big', data=numpy.arange(140000, dtype='<f4').reshape(140000, 1), chunks=(1, 1), maxshape=(None, 1))
        d = f.create_dataset('big_sparse', shape=(600000, 1), dtype='<f4', chunks=(1, 1), maxshape=(None, 1), fillvalue=-7.0)
        for i, x in ((135000, 1.5), (199999, 2.5), (3, 3.5), (599999, 4.5), (300000, 5.5)):
            d[i] = x


def random_file(path, seed, libver, count):
    """`count` random chunked datasets (rank 1-4, random chunk shapes, filters, byte orders, fixed / one / several
    unlimited dimensions, some only partly written) plus <path>.<name>.npy with the values h5py itself reads back and
    <path>.json with the names: tests/test_hdf5min.py cross-checks the reader against them."""
    import json
    rng = numpy.random.default_rng(seed)
    names = []
    with h5py.File(path, 'w', libver=libver) as f:
        for k in range(count):
            rank = int(rng.integers(1, 5))
            shape = tuple(int(x) for x in rng.integers(1, 9, rank))
            chunks = tuple(int(rng.integers(1, s + 1)) for s in shape)
            dt = str(rng.choice(['<f4', '<f8', '>f4', '<i4', '<i2']))
            maxshape = list(shape)
            for i in rng.permutation(rank)[:[0, 1, 2, rank][int(rng.integers(0, 4))]]:
                maxshape[i] = None
            kw = {}
            if rng.random() < 0.5:
                kw.update(compression='gzip', compression_opts=int(rng.integers(1, 6)))
            if rng.random() < 0.4:
                kw['shuffle'] = True
            if rng.random() < 0.2:
                kw['fletcher32'] = True
            name = f'd{k}'
            a = (rng.standard_normal(shape) * 100).astype(dt)
            if rng.random() < 0.3:   # only a corner is ever written: the rest reads as the fill value
                d = f.create_dataset(name, shape=shape, dtype=dt, chunks=chunks, maxshape=tuple(maxshape),
                                     fillvalue=float(rng.integers(-5, 5)), **kw)
                sel = tuple(slice(0, max(1, s // 2)) for s in shape)
                d[sel] = a[sel]
            else:
                d = f.create_dataset(name, data=a, chunks=chunks, maxshape=tuple(maxshape), **kw)
            numpy.save(f'{path}.{name}.npy', d[...])
            names.append(name)
    with open(path + '.json', 'w') as f:
        json.dump(names, f)


def cf_files():
    """CF-encoded uo / vo, the decoding xarray's open_dataset does by default for the reference (field.py:34-35, decode_cf):
    cf_U.h5  uo int16 packed with scale_factor / add_offset, _FillValue AND a different missing_value (both masked),
             chunked + shuffled + deflated;  cf_V.h5  vo float32 with _FillValue 1e20 and missing_value -9999, whole-plane
             deflated chunks (the device-inflate layout);  cf_V64.h5  vo float64 contiguous, missing_value only.
    Values come from golden case def36_zt; <file>.<name>.raw.npy is what h5py reads back (undecoded)."""
    g = numpy.load(os.path.join(OUT, '..', 'def36_zt.npz'))
    u, v = g['u'].astype('<f8'), g['v'].astype('<f4')
    scale, offset = numpy.float32(0.002), numpy.float32(1.5)
    packed = numpy.clip(numpy.rint((u - float(offset)) / float(scale)), -32000, 32000).astype('<i2')
    packed[:, :, 4:9, 10:20] = -32768          # _FillValue
    packed[:, :, 12:14, 3:8] = -32767          # missing_value
    with h5py.File(os.path.join(OUT, 'cf_U.h5'), 'w', libver='earliest') as f:
        d = f.create_dataset('uo', data=packed, chunks=(1, 1, 9, 18), compression='gzip', shuffle=True)
        d.attrs.create('_FillValue', numpy.int16(-32768))
        d.attrs.create('missing_value', numpy.int16(-32767))
        d.attrs.create('scale_factor', scale)
        d.attrs.create('add_offset', offset)
        numpy.save(os.path.join(OUT, 'cf_U.h5.uo.raw.npy'), d[...])
    v[:, :, 4:9, 10:20] = numpy.float32(1.e20)
    v[:, :, 12:14, 3:8] = numpy.float32(-9999.)
    with h5py.File(os.path.join(OUT, 'cf_V.h5'), 'w', libver='earliest') as f:
        d = f.create_dataset('vo', data=v, chunks=(1, 1, 18, 36), compression='gzip', shuffle=True)
        d.attrs.create('_FillValue', numpy.float32(1.e20))
        d.attrs.create('missing_value', numpy.float32(-9999.))
        numpy.save(os.path.join(OUT, 'cf_V.h5.vo.raw.npy'), d[...])
    v64 = g['v'].astype('<f8')
    v64[:, :, 12:14, 3:8] = -9999.
    v64[:, :, 4:9, 10:20] = numpy.nan
    with h5py.File(os.path.join(OUT, 'cf_V64.h5'), 'w', libver='earliest') as f:
        d = f.create_dataset('vo', data=v64)
        d.attrs.create('missing_value', numpy.float64(-9999.))
    # the same float64 vo chunked + shuffled + deflated (device-decodable on its own): paired with a float32 uo it must NOT
    # take the device path of a float32 stager (8-byte elements into a 4-byte slab)
    with h5py.File(os.path.join(OUT, 'cf_V64c.h5'), 'w', libver='earliest') as f:
        d = f.create_dataset('vo', data=v64, chunks=(1, 1, 18, 36), compression='gzip', shuffle=True)
        d.attrs.create('missing_value', numpy.float64(-9999.))
    # a float32 uo with whole-plane deflated chunks to pair with cf_V.h5 on the device path (same masks, _FillValue only)
    u32 = g['u'].astype('<f4')
    u32[:, :, 4:9, 10:20] = numpy.float32(1.e20)
    with h5py.File(os.path.join(OUT, 'cf_U32.h5'), 'w', libver='earliest') as f:
        d = f.create_dataset('uo', data=u32, chunks=(1, 1, 18, 36), compression='gzip', shuffle=True)
        d.attrs.create('_FillValue', numpy.float32(1.e20))
    print('wrote the cf_* files')


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--cf':
        return cf_files()
    if len(sys.argv) > 2 and sys.argv[1] == '--big-extensible':
        return big_extensible(sys.argv[2])
    if len(sys.argv) > 5 and sys.argv[1] == '--random':
        return random_file(sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5]))
    os.makedirs(OUT, exist_ok=True)
    with h5py.File(os.path.join(OUT, 'old_style.h5'), 'w', libver='earliest') as f:
        f.create_dataset('bounds_lon', data=field((5, 7, 4), '<f8', 1))
        d = f.create_dataset('uo', data=field((3, 4, 9, 11), '<f4', 2), chunks=(1, 2, 4, 5), compression='gzip',
                             compression_opts=4, shuffle=True)
        d.attrs.create('_FillValue', numpy.float32(1.e20))
        d.attrs['units'] = 'm/s'
        f.create_dataset('vo', data=field((3, 4, 9, 11), '<f4', 3), chunks=(3, 4, 9, 11))     # one chunk, no filter
        f.create_dataset('time_counter', data=field((12,), '>f4', 4))
        f.create_dataset('index', data=field((6,), '<i4', 5))
        f.create_dataset('tiny', data=field((3,), '<f8', 6), chunks=None)
        g = f.create_group('sub')
        g.create_dataset('inner', data=field((2, 2), '<f8', 7))
        # one deflated + shuffled chunk per (t, level) plane, as XIOS writes uo/vo (whole-plane chunks)
        f.create_dataset('planes', data=field((2, 3, 4, 6), '<f4', 50), chunks=(1, 1, 4, 6), compression='gzip',
                         shuffle=True)
        # only one of the four chunks is ever written: the others read as the HDF5 fill value
        d = f.create_dataset('sparse', shape=(4, 6), dtype='<f8', chunks=(2, 3), fillvalue=7.5)
        d[2:4, 0:3] = field((2, 3), '<f8', 51)
    with h5py.File(os.path.join(OUT, 'new_compact.h5'), 'w', libver='earliest', track_order=True) as f:
        f.create_dataset('bounds_lat', data=field((5, 7, 4), '<f4', 8), track_order=True)
        d = f.create_dataset('deptht_bounds', data=field((75, 2), '<f4', 9), track_order=True)
        d.attrs.create('_FillValue', numpy.float32(-999.0))
    with h5py.File(os.path.join(OUT, 'new_dense.h5'), 'w', libver='earliest', track_order=True) as f:
        for k in range(14):
            f.create_dataset(f'var{k:02d}', data=field((4, 3), '<f8', 10 + k), track_order=True)
        d = f.create_dataset('vo', data=field((2, 3, 8, 6), '<f4', 30), chunks=(1, 3, 8, 6), compression='gzip',
                             fletcher32=True, track_order=True)
        for k in range(11):
            d.attrs[f'attr{k:02d}'] = f'value {k}'
        d.attrs.create('_FillValue', numpy.float32(1.e20))
    with h5py.File(os.path.join(OUT, 'latest.h5'), 'w', libver='latest') as f:
        f.create_dataset('contig', data=field((6, 5), '<f8', 40))
        f.create_dataset('single_chunk', data=field((6, 5), '<f4', 41), chunks=(6, 5), compression='gzip')
        f.create_dataset('many_chunks', data=field((6, 5), '<f4', 42), chunks=(2, 5))                 # fixed array
        f.create_dataset('fa_filtered', data=field((3, 2, 9, 7), '<f4', 43), chunks=(1, 1, 4, 7),       # fixed array of
                         compression='gzip', shuffle=True)                                              # filtered chunks
        f.create_dataset('fa_paged', data=field((40, 60), '<f8', 44), chunks=(1, 2))                    # 1200 > 1024 elements
        d = f.create_dataset('fa_paged_sparse', shape=(40, 60), dtype='<f8', chunks=(1, 2), fillvalue=-1.5)
        d[39, 58:60] = field((2,), '<f8', 45)                                                           # only the last page
        # chunks allocated at creation, no filter -> implicit index (low-level API: the high-level one cannot ask for it)
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_chunk((2, 5))
        dcpl.set_alloc_time(h5py.h5d.ALLOC_TIME_EARLY)
        dsid = h5py.h5d.create(f.id, b'implicit', h5py.h5t.IEEE_F32LE, h5py.h5s.create_simple((6, 5)), dcpl)
        dsid.write(h5py.h5s.ALL, h5py.h5s.ALL, field((6, 5), '<f4', 46))
        # one unlimited dimension -> extensible-array index: within the index block, through data blocks, through super
        # blocks, filtered, unlimited dimension not first (chunk numbering swizzled), partly written
        f.create_dataset('ea4', data=field((4, 3), '<f4', 47), chunks=(1, 3), maxshape=(None, 3))
        f.create_dataset('ea100', data=field((100, 3), '<f4', 48), chunks=(1, 3), maxshape=(None, 3))
        f.create_dataset('ea3000', data=field((3000, 2), '<f4', 49), chunks=(1, 2), maxshape=(None, 2))
        f.create_dataset('ea_filt', data=field((50, 4, 6), '<f4', 50), chunks=(1, 2, 6), maxshape=(None, 4, 6),
                         compression='gzip', shuffle=True)
        f.create_dataset('ea_mid', data=field((5, 30, 4), '<f4', 51), chunks=(2, 1, 4), maxshape=(5, None, 4))
        d = f.create_dataset('ea_sparse', shape=(5000, 2), dtype='<f8', chunks=(1, 2), maxshape=(None, 2), fillvalue=-2.5)
        d[4321] = [1.0, 2.0]
        d[7] = [3.0, 4.0]
        # two unlimited dimensions -> version-2 B-tree index: root leaf, depth 1, depth 2, filtered, partly written
        f.create_dataset('bt2_small', data=field((4, 3), '<f4', 52), chunks=(1, 3), maxshape=(None, None))
        f.create_dataset('bt2_d1', data=field((30, 40), '<f4', 53), chunks=(1, 2), maxshape=(None, None))
        f.create_dataset('bt2_d2', data=field((100, 160), '<f4', 54), chunks=(1, 2), maxshape=(None, None))
        f.create_dataset('bt2_filt', data=field((20, 6, 8), '<f4', 55), chunks=(1, 3, 8), maxshape=(None, None, 8),
                         compression='gzip', shuffle=True)
        d = f.create_dataset('bt2_sparse', shape=(50, 50), dtype='<f8', chunks=(5, 5), maxshape=(None, None), fillvalue=9.5)
        d[47, 3] = 1.0
        d[0, 49] = 2.0
    # a NEMO-like T/U/V triple built from the golden case def36_zt (reference datagen output), the way XIOS/netCDF-4
    # writes it: float32, uo chunked + shuffled + deflated with land as _FillValue, vo contiguous with NaN land
    g = numpy.load(os.path.join(OUT, '..', 'def36_zt.npz'))
    u, v = g['u'].astype('<f4'), g['v'].astype('<f4')
    u[:, :, 4:9, 10:20] = numpy.float32(1.e20)
    v[:, :, 4:9, 10:20] = numpy.nan
    with h5py.File(os.path.join(OUT, 'nemo_T.h5'), 'w', libver='earliest', track_order=True) as f:
        f.create_dataset('bounds_lon', data=g['bounds_lon'].astype('<f4'), track_order=True)
        f.create_dataset('bounds_lat', data=g['bounds_lat'].astype('<f4'), track_order=True)
        f.create_dataset('deptht_bounds', data=g['deptht_bounds'].astype('<f4'), track_order=True)
    with h5py.File(os.path.join(OUT, 'nemo_U.h5'), 'w', libver='earliest', track_order=True) as f:
        d = f.create_dataset('uo', data=u, chunks=(1, 1, 9, 18), compression='gzip', shuffle=True, track_order=True)
        d.attrs.create('_FillValue', numpy.float32(1.e20))
        for k in range(9):
            d.attrs[f'note{k}'] = numpy.float64(k)
        tc = f.create_dataset('time_counter', data=(numpy.arange(3) * 86400. * 30.5 + 1296000.).astype('>f8'),
                              track_order=True)
        tc.attrs.create('standard_name', numpy.string_('time'))
        tc.attrs.create('long_name', numpy.string_('Time axis'))
        tc.attrs.create('units', numpy.string_('seconds since 1900-01-01 00:00:00'))
        tc.attrs.create('calendar', numpy.string_('noleap'))
    with h5py.File(os.path.join(OUT, 'nemo_V.h5'), 'w', libver='earliest') as f:
        f.create_dataset('vo', data=v)
    print('wrote', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    main()
