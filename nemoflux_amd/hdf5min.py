"""hdf5min -- a small read-only HDF5 / NetCDF-4 reader (numpy + zlib only), enough for the files nemoflux opens.

The reference reads its T/U/V files through xarray + netCDF4 (nemoflux/field.py:22-25,34-35,149); neither is in this
image's main interpreter.  NetCDF-4 files ARE HDF5 files, and what nemoflux needs from them is small: a handful of
named n-d float arrays (bounds_lat, bounds_lon, deptht_bounds, uo, vo) and their `_FillValue` attribute.  This module
parses exactly that subset of the HDF5 file format (HDF5 File Format Specification v3):

  * superblock versions 0-3;
  * object headers version 1 and version 2 (with continuation blocks);
  * groups: old style (symbol table: v1 B-tree + local heap + SNOD nodes), new style with compact links (Link
    messages) and with dense links (fractal heap whose root is a direct block or a one-level indirect block);
  * datasets: IEEE float / integer datatypes of either byte order; contiguous, compact and chunked storage -- layout
    message v3 with the v1 B-tree chunk index (what netCDF-4 / XIOS write), layout v4 ("latest" format) with the
    single-chunk, implicit, fixed-array, extensible-array (paged blocks included) and version-2 B-tree indexes;
    deflate, shuffle and fletcher32 filters; never-written chunks read as
    the fill value; the chunks of a read are inflated on all host cores;
  * attributes in the object header or in dense storage (fractal heap + name-index B-tree), e.g. _FillValue, units.

Contiguous little-endian data is returned as a numpy memmap of the file itself -- no copy: handing such an array to
Field makes the engine stage each time step from the page cache straight to HBM.  Anything this reader does not
understand raises Hdf5Error (it never guesses); nemoflux_amd.io then falls back to xarray or an h5py interpreter.
"""
import concurrent.futures
import mmap
import os
import zlib

import numpy

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xffffffffffffffff


class Hdf5Error(RuntimeError):
    pass


# optional native byte un-shuffle: f(src_uint8_array, dst_uint8_array, n_elements, element_size); nemoflux_amd.io installs
# the library's nf_host_unshuffle here, hdf5min itself stays importable with numpy + zlib alone
_native_unshuffle = None


def set_unshuffle(func):
    global _native_unshuffle
    _native_unshuffle = func


def _unshuffle(a, dst, ne, es):
    """dst (uint8, ne*es bytes, C-contiguous) <- the ne elements whose es byte planes lie one after the other in a."""
    if _native_unshuffle is not None and dst.flags.c_contiguous and a.flags.c_contiguous:
        _native_unshuffle(a, dst, ne, es)
        return
    d2 = dst.reshape(ne, es)
    for j in range(es):     # byte-plane assignments: numpy releases the GIL for them
        d2[:, j] = a[j * ne:(j + 1) * ne]


def io_threads():
    """Worker threads for chunk inflation: NF_IO_THREADS, else the cores this process may run on (at most 32)."""
    n = os.environ.get('NF_IO_THREADS')
    if n:
        return max(1, int(n))
    try:
        return min(32, len(os.sched_getaffinity(0)))
    except AttributeError:
        return min(32, os.cpu_count() or 1)


class Dataset(object):
    def __init__(self, h5, name, shape, dtype, layout, filters, attrs, h5fill=None):
        self._h5, self.name, self.shape, self.dtype = h5, name, tuple(shape), dtype
        self._layout, self._filters, self.attrs = layout, filters, attrs
        self.h5fill = h5fill       # value of the HDF5 fill-value message: what never-written chunks read as

    @property
    def fill_value(self):
        fv = self.attrs.get('_FillValue')
        return None if fv is None else numpy.asarray(fv).reshape(-1)[0]

    def is_contiguous(self):
        return self._layout[0] == 'contiguous' and not self._filters

    def read(self):
        """The whole array.  Contiguous unfiltered data comes back as a (read-only) view of the mapped file."""
        kind = self._layout[0]
        n = int(numpy.prod(self.shape)) if self.shape else 1
        if kind == 'compact':
            return numpy.frombuffer(self._layout[1], dtype=self.dtype, count=n).reshape(self.shape)
        if kind == 'contiguous':
            addr = self._layout[1]
            if addr == UNDEF:
                return numpy.zeros(self.shape, self.dtype.newbyteorder('='))
            return numpy.frombuffer(self._h5._m, dtype=self.dtype, count=n, offset=self._h5._base + addr).reshape(self.shape)
        if kind == 'chunked':
            return self._read_chunked()
        raise Hdf5Error(f'{self.name}: unsupported layout {kind}')

    def read_leading(self, i, out=None):
        """The slab [i] of the leading axis (one time step of uo/vo) without touching the rest of the variable:
        a view of the mapped file for contiguous data, only the overlapping chunks are inflated otherwise.  `out`: a
        C-contiguous array of the slab's shape and this dataset's dtype to inflate into (a caller that walks the time
        steps re-uses one buffer instead of faulting in a fresh one per step)."""
        if not self.shape or not (0 <= i < self.shape[0]):
            raise Hdf5Error(f'{self.name}: leading index {i} out of range')
        if self._layout[0] != 'chunked':
            if out is None:
                return self.read()[i]
            numpy.copyto(out, self.read()[i])
            return out
        if out is not None:
            if out.shape != self.shape[1:] or out.dtype != self.dtype or not out.flags.c_contiguous:
                raise Hdf5Error(f'{self.name}: output buffer does not match the slab')
            return self._read_chunked(lead=i, out=out.reshape((1,) + self.shape[1:]))[0]
        return self._read_chunked(lead=i)[0]

    def _decode(self, raw, filter_mask, nbytes, dst=None):
        """Undo the filter pipeline (in reverse order of application).  When the last step is the un-shuffle and `dst`
        (a C-contiguous uint8 view of the chunk's place in the output) is given, that step writes straight into it and
        None is returned: no intermediate copy, and the strided numpy assignments run without the GIL."""
        for k in range(len(self._filters) - 1, -1, -1):
            fid, cd = self._filters[k]
            if filter_mask & (1 << k):
                continue
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:  # shuffle: bytes of each element were de-interleaved
                es = cd[0] if cd else self.dtype.itemsize
                a = numpy.frombuffer(raw, numpy.uint8)
                ne = a.size // es
                last = not any(not (filter_mask & (1 << j)) for j in range(k))
                if dst is not None and last and ne * es == a.size == dst.size:
                    _unshuffle(a, dst, ne, es)
                    return None
                tmp = numpy.empty(a.size, numpy.uint8)
                _unshuffle(a[:ne * es], tmp[:ne * es], ne, es)
                tmp[ne * es:] = a[ne * es:]
                raw = tmp
            elif fid == 3:  # fletcher32: checksum appended
                raw = raw[:-4]
            else:
                raise Hdf5Error(f'{self.name}: unsupported filter id {fid}')
        if len(raw) < nbytes:
            raise Hdf5Error(f'{self.name}: short chunk')
        return raw

    def _ensure_chunks(self):
        """self._chunks = [(offsets, compressed size, filter mask, file address)] of every chunk that was written."""
        if getattr(self, '_chunks', None) is None:
            _, btree, cdims, single = self._layout
            rank = len(self.shape)
            cshape = tuple(cdims[:rank])
            nbytes = int(numpy.prod(cshape)) * self.dtype.itemsize
            if single is not None:
                self._chunks = [single]
            elif isinstance(btree, tuple):      # layout version 4 indexes
                self._chunks = list(self._h5._chunk_index_v4(btree[0], btree[1], self.shape, cshape, nbytes,
                                                             btree[2] if len(btree) > 2 else None))
            else:
                self._chunks = list(self._h5._chunk_btree(btree, rank))
        return self._chunks

    def _chunks_of_lead(self, lead):
        """chunks whose leading offset covers `lead`, through an index built once (a year of daily output has tens of
        thousands of chunks: scanning them for every time step would be quadratic)"""
        chunks = self._ensure_chunks()
        idx = getattr(self, '_lead_index', None)
        if idx is None or idx[0] is not chunks:
            c0 = int(self._layout[2][0])
            table = {}
            for c in chunks:
                for k in range(c[0][0], min(c[0][0] + c0, self.shape[0])):
                    table.setdefault(k, []).append(c)
            idx = self._lead_index = (chunks, table)
        return idx[1].get(lead, [])

    def device_plan(self, lead):
        """How the slab [lead] of the leading axis can be decoded ON THE DEVICE (nemoflux_amd.ingest.ChunkDecoder), or None
        when it has to go through the host path: {'chunks': [(byte offset in the mapped file, compressed size, (z0, y0, x0)
        origin in the slab)], 'chunk_dims': (cz, cy, cx), 'slab_dims': (nz, ny, nx), 'chunk_bytes', 'elem_size', 'shuffled'}.
        Taken: what netCDF-4 / XIOS write for NEMO output -- deflate, optionally behind the shuffle filter, native 4- or
        8-byte elements, one leading index per chunk, every chunk of the slab written with the whole pipeline applied."""
        rank = len(self.shape)
        if self._layout[0] != 'chunked' or not (2 <= rank <= 4):
            return None
        fids = [f[0] for f in self._filters]
        if fids not in ([1], [2, 1]):             # order of application when writing: shuffle, then deflate
            return None
        es = self.dtype.itemsize
        if es not in (4, 8) or not self.dtype.isnative:
            return None
        shuffled = fids[0] == 2
        if shuffled:
            cd = self._filters[0][1]
            if (cd[0] if cd else es) != es:
                return None
        cshape = tuple(int(c) for c in self._layout[2][:rank])
        if cshape[0] != 1:
            return None
        pad = 4 - rank                            # slab seen as (nz, ny, nx) with leading ones
        slab = (1,) * pad + tuple(int(x) for x in self.shape[1:])
        cdim = (1,) * pad + cshape[1:]
        todo = self._chunks_of_lead(lead)
        expected = int(numpy.prod([-(-n // c) for n, c in zip(slab, cdim)]))
        if len(todo) != expected or any(c[2] != 0 or c[3] == UNDEF for c in todo):
            return None                           # chunks never written / stored unfiltered: the host path fills them in
        if int(numpy.prod(cdim)) * es >= 1 << 31 or any(c[1] >= 1 << 31 for c in todo):
            return None                           # the device decoder addresses a chunk with 32 bits: the host path takes these
        chunks = [(self._h5._base + c[3], c[1], (0,) * pad + tuple(int(o) for o in c[0][1:rank])) for c in todo]
        chunks.sort(key=lambda c: c[2])
        return dict(chunks=chunks, chunk_dims=cdim, slab_dims=slab, chunk_bytes=int(numpy.prod(cdim)) * es, elem_size=es,
                    shuffled=1 if shuffled else 0)

    def _read_chunked(self, lead=None, out=None):
        _, btree, cdims, single = self._layout
        rank = len(self.shape)
        cshape = tuple(cdims[:rank])
        shape = self.shape if lead is None else (1,) + self.shape[1:]
        nbytes = int(numpy.prod(cshape)) * self.dtype.itemsize
        self._ensure_chunks()
        todo = [c for c in (self._chunks if lead is None else self._chunks_of_lead(lead)) if c[3] != UNDEF]
        expected = int(numpy.prod([-(-s // c) for s, c in zip(shape, cshape)]))
        if len(todo) < expected:   # chunks that were never written read as the HDF5 fill value (0 when the file defines none)
            if out is None:
                out = numpy.empty(shape, self.dtype)
            out[...] = 0 if self.h5fill is None else self.h5fill
        elif out is None:
            out = numpy.empty(shape, self.dtype)

        def place(chunk):
            offs, size, mask, addr = chunk
            sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, self.shape))
            sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
            if lead is not None:
                sl_in = (slice(lead - offs[0], lead - offs[0] + 1),) + sl_in[1:]
                sl_out = (slice(0, 1),) + sl_out[1:]
            dst = out[sl_out]
            direct = dst.shape == cshape and dst.flags.c_contiguous     # whole chunk, one contiguous run of `out`
            raw = memoryview(self._h5._m)[self._h5._base + addr: self._h5._base + addr + size]   # no copy of the chunk
            raw = self._decode(raw, mask, nbytes, dst.reshape(-1).view(numpy.uint8) if direct else None)
            if raw is not None:
                blk = numpy.frombuffer(raw, self.dtype, count=nbytes // self.dtype.itemsize).reshape(cshape)
                dst[...] = blk[sl_in]     # chunks never overlap: the writes of concurrent workers are disjoint

        # zlib.decompress and the numpy copies release the GIL, so the chunks of one time step (real NEMO files: one
        # deflated chunk per level) inflate on all host cores at once
        nthreads = min(io_threads(), len(todo))
        if nthreads > 1:
            with concurrent.futures.ThreadPoolExecutor(nthreads) as pool:
                list(pool.map(place, todo))
        else:
            for chunk in todo:
                place(chunk)
        return out


class LazyVariable(object):
    """A (nt, ...) variable read one leading slab at a time (Field stages exactly one time step per compute call)."""

    def __init__(self, dataset):
        self.dataset, self.shape = dataset, dataset.shape
        self.dtype = numpy.dtype(dataset.dtype.newbyteorder('='))

    def device_plan(self, t):
        return self.dataset.device_plan(t)

    def raw_bytes(self):
        """The mapped file (chunk addresses of device_plan index into it)."""
        return self.dataset._h5._m

    def read_step(self, t, out=None):
        """Time step t in native byte order, C-contiguous; into `out` (same shape, native dtype) when given."""
        if out is not None and self.dataset.dtype.isnative:
            return self.dataset.read_leading(t, out=out)
        a = self.dataset.read_leading(t)
        if out is None:
            return numpy.ascontiguousarray(a, dtype=self.dtype)
        numpy.copyto(out, a)      # converts the byte order
        return out


class File(object):
    def __init__(self, path):
        self._f = open(path, 'rb')
        try:
            self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError as e:
            raise Hdf5Error(f'{path}: empty file') from e
        self._parse_superblock()
        self.datasets = {}
        self._walk_group(self._root, '', 0)

    def close(self):
        # views handed out by Dataset.read() keep the map alive; only the descriptor is closed here
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ------------------------------------------------------------------------------------------ primitives
    def _u(self, off, n):
        return int.from_bytes(self._m[off:off + n], 'little')

    def _addr(self, off):
        return self._u(off, self._O)

    def _len(self, off):
        return self._u(off, self._L)

    def _parse_superblock(self):
        m = self._m
        base = 0
        while base < len(m) and m[base:base + 8] != SIGNATURE:  # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
        if base >= len(m):
            raise Hdf5Error('not an HDF5 file')
        ver = m[base + 8]
        if ver in (0, 1):
            self._O, self._L = m[base + 13], m[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self._base = self._addr(p)
            p += 4 * self._O                       # base, free-space, end-of-file, driver-info addresses
            self._root = self._addr(p + self._O)   # root symbol table entry: link name offset, object header address
        elif ver in (2, 3):
            self._O, self._L = m[base + 9], m[base + 10]
            p = base + 12
            self._base = self._addr(p)
            self._root = self._addr(p + 3 * self._O)
        else:
            raise Hdf5Error(f'unsupported superblock version {ver}')
        if self._O != 8 or self._L != 8:
            raise Hdf5Error('only 8-byte offsets and lengths are supported')
        if self._base == UNDEF:
            self._base = 0

    # ------------------------------------------------------------------------------------------ object headers
    def _messages(self, addr):
        """[(type, flags, payload offset, payload size)] of the object header at file address addr."""
        m, p = self._m, self._base + addr
        out = []
        if m[p:p + 4] == b'OHDR':
            if m[p + 4] != 2:
                raise Hdf5Error('unsupported object header version')
            flags = m[p + 5]
            q = p + 6
            if flags & 0x20:
                q += 16
            if flags & 0x10:
                q += 4
            nsz = 1 << (flags & 3)
            size0 = self._u(q, nsz)
            q += nsz
            blocks = [(q, size0)]
            corder = 2 if flags & 0x04 else 0
            while blocks:
                q, size = blocks.pop(0)
                end = q + size
                while q + 4 + corder <= end:
                    mtype, msize, mflags = m[q], self._u(q + 1, 2), m[q + 3]
                    q += 4 + corder
                    if q + msize > end:
                        break
                    if mtype == 0x10:
                        caddr, clen = self._addr(q), self._len(q + self._O)
                        cp = self._base + caddr
                        if m[cp:cp + 4] != b'OCHK':
                            raise Hdf5Error('bad object header continuation')
                        blocks.append((cp + 4, clen - 8))     # minus signature and checksum
                    elif mtype != 0:
                        out.append((mtype, mflags, q, msize))
                    q += msize
            return out
        if m[p] != 1:
            raise Hdf5Error(f'unsupported object header at {addr:#x}')
        nmsgs, hsize = self._u(p + 2, 2), self._u(p + 8, 4)
        blocks = [(p + 16, hsize)]
        while blocks and len(out) < nmsgs + 64:
            q, size = blocks.pop(0)
            end = q + size
            while q + 8 <= end:
                mtype, msize, mflags = self._u(q, 2), self._u(q + 2, 2), m[q + 4]
                q += 8
                if mtype == 0x10:
                    blocks.append((self._base + self._addr(q), self._len(q + self._O)))
                elif mtype != 0:
                    out.append((mtype, mflags, q, msize))
                q += msize
        return out

    # ------------------------------------------------------------------------------------------ message decoders
    def _dataspace(self, q):
        m = self._m
        ver, rank, flags = m[q], m[q + 1], m[q + 2]
        q += 8 if ver == 1 else 4
        return [self._len(q + i * self._L) for i in range(rank)]

    def _unlimited_dims(self, q):
        """Indices of the dimensions whose maximum size is unlimited (needed by the extensible-array chunk index)."""
        m = self._m
        ver, rank, flags = m[q], m[q + 1], m[q + 2]
        if not flags & 1:
            return []
        q += (8 if ver == 1 else 4) + rank * self._L
        return [i for i in range(rank) if self._len(q + i * self._L) == UNDEF]

    def _datatype(self, q):
        """numpy dtype and the encoded size of the message."""
        m = self._m
        cls, ver = m[q] & 0x0f, m[q] >> 4
        bits0 = m[q + 1]
        size = self._u(q + 4, 4)
        order = '>' if bits0 & 1 else '<'
        if cls == 1:
            if size not in (4, 8):
                raise Hdf5Error('unsupported float size')
            return numpy.dtype(order + 'f' + str(size)), 8 + 12
        if cls == 0:
            kind = 'i' if bits0 & 0x08 else 'u'
            return numpy.dtype((order if size > 1 else '|') + kind + str(size)), 8 + 4
        if cls == 3:   # fixed-length string
            return numpy.dtype('S' + str(size)), 8
        raise Hdf5Error(f'unsupported datatype class {cls}')

    def _layout(self, q):
        m = self._m
        ver = m[q]
        if ver == 3:
            cls = m[q + 1]
            if cls == 0:
                n = self._u(q + 2, 2)
                return ('compact', bytes(m[q + 4:q + 4 + n]))
            if cls == 1:
                return ('contiguous', self._addr(q + 2), self._len(q + 2 + self._O))
            if cls == 2:
                nd = m[q + 2]
                bt = self._addr(q + 3)
                dims = [self._u(q + 3 + self._O + 4 * i, 4) for i in range(nd)]
                return ('chunked', bt, dims, None)
        elif ver == 4:
            cls = m[q + 1]
            if cls == 0:
                n = self._u(q + 2, 2)
                return ('compact', bytes(m[q + 4:q + 4 + n]))
            if cls == 1:
                return ('contiguous', self._addr(q + 2), self._len(q + 2 + self._O))
            if cls == 2:
                flags, nd, enc = m[q + 2], m[q + 3], m[q + 4]
                dims = [self._u(q + 5 + enc * i, enc) for i in range(nd)]
                p = q + 5 + enc * nd
                itype = m[p]
                p += 1
                if itype == 1:    # single chunk
                    size, mask = None, 0
                    if flags & 2:
                        size, mask = self._len(p), self._u(p + self._L, 4)
                        p += self._L + 4
                    addr = self._addr(p)
                    if size is None:
                        size = int(numpy.prod(dims[:-1])) * dims[-1]
                    return ('chunked', None, dims, ((0,) * (nd - 1), size, mask, addr))
                if itype == 2:    # implicit: unfiltered chunks laid out back to back in row-major chunk order
                    return ('chunked', ('implicit', self._addr(p)), dims, None)
                if itype == 3:    # fixed array (no unlimited dimension); one byte of creation parameters (page bits)
                    return ('chunked', ('farray', self._addr(p + 1)), dims, None)
                if itype == 4:    # extensible array (one unlimited dimension); five bytes of creation parameters
                    return ('chunked', ('earray', self._addr(p + 5)), dims, None)
                if itype == 5:    # version-2 B-tree (two or more unlimited dimensions); six bytes of creation parameters
                    return ('chunked', ('btree2', self._addr(p + 6)), dims, None)
                raise Hdf5Error(f'unsupported chunk index type {itype} (HDF5 1.10 "latest" format)')
        raise Hdf5Error(f'unsupported data layout message (version {ver})')

    def _filters(self, q):
        m = self._m
        ver, n = m[q], m[q + 1]
        q += 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = self._u(q, 2)
            q += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = self._u(q, 2)
                q += 2
            ncd = self._u(q + 2, 2)
            q += 4
            if nlen:
                q += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = [self._u(q + 4 * i, 4) for i in range(ncd)]
            q += 4 * ncd
            if ver == 1 and ncd % 2:
                q += 4
            out.append((fid, cd))
        return out

    def _attribute(self, q):
        m = self._m
        ver = m[q]
        nsz, dtsz, dssz = self._u(q + 2, 2), self._u(q + 4, 2), self._u(q + 6, 2)
        p = q + 8 + (1 if ver == 3 else 0)
        pad = (lambda x: (x + 7) // 8 * 8) if ver == 1 else (lambda x: x)
        name = bytes(m[p:p + nsz]).split(b'\0')[0].decode('utf-8', 'replace')
        p += pad(nsz)
        shape = self._dataspace(p + pad(dtsz)) if dssz >= 4 else []
        n = int(numpy.prod(shape)) if shape else 1
        esize = self._u(p + 4, 4)                       # element size of the attribute's datatype
        total = (p - q) + pad(dtsz) + pad(dssz) + n * esize
        try:
            dt, _ = self._datatype(p)
        except Hdf5Error:
            return name, None, total                    # e.g. variable-length strings: skipped
        p += pad(dtsz) + pad(dssz)
        val = numpy.frombuffer(m[p:p + n * dt.itemsize], dt, count=n)
        return name, (val.reshape(shape) if shape else val[0]), total

    # ------------------------------------------------------------------------------------------ groups
    def _walk_group(self, addr, prefix, depth):
        if depth > 4:
            return
        msgs = self._messages(addr)
        types = [t for t, _, _, _ in msgs]
        if 0x01 in types and 0x03 in types and 0x08 in types:   # a dataset at the root (not expected)
            return
        for mtype, _, q, size in msgs:
            if mtype == 0x11:      # old-style group: symbol table
                for name, child in self._symbol_table(self._addr(q), self._addr(q + self._O)):
                    self._visit(name, child, prefix, depth)
            elif mtype == 0x06:    # compact link
                link = self._link(q)
                if link:
                    self._visit(link[0], link[1], prefix, depth)
            elif mtype == 0x02:    # link info: dense storage in a fractal heap
                flags = self._m[q + 1]
                p = q + 2 + (8 if flags & 1 else 0)
                heap, btree = self._addr(p), self._addr(p + self._O)
                if heap != UNDEF:
                    # name-index records (type 5): hash (4), heap id (7)
                    for name, child in self._fractal_heap_objects(heap, self._link_item, btree, 4):
                        self._visit(name, child, prefix, depth)

    def _visit(self, name, addr, prefix, depth):
        try:
            msgs = self._messages(addr)
        except (Hdf5Error, IndexError):
            return
        info = {t: (q, s) for t, _, q, s in msgs}
        if 0x01 in info and 0x03 in info and 0x08 in info:
            try:
                shape = self._dataspace(info[0x01][0])
                dt, _ = self._datatype(info[0x03][0])
                layout = self._layout(info[0x08][0])
                if layout[0] == 'chunked' and isinstance(layout[1], tuple) and layout[1][0] == 'earray':
                    unl = self._unlimited_dims(info[0x01][0])
                    if len(unl) != 1:
                        raise Hdf5Error('extensible-array chunk index without exactly one unlimited dimension')
                    layout = ('chunked', ('earray', layout[1][1], unl[0]), layout[2], None)
                filters = self._filters(info[0x0b][0]) if 0x0b in info else []
            except Hdf5Error as e:
                self.datasets[prefix + name] = e     # reported when the variable is asked for
                return
            attrs = {}
            for t, _, q, s in msgs:
                if t == 0x0c:
                    k, v, _ = self._attribute(q)
                    if v is not None:
                        attrs[k] = v
                elif t == 0x15:    # attribute info: dense attribute storage (more than 8 attributes)
                    flags = self._m[q + 1]
                    p0 = q + 2 + (2 if flags & 1 else 0)
                    heap, btree = self._addr(p0), self._addr(p0 + self._O)
                    if heap != UNDEF:
                        # name-index records (type 8): heap id (8), flags (1), creation order (4), hash (4)
                        for k, v in self._fractal_heap_objects(heap, self._attribute_item, btree, 0):
                            if v is not None:
                                attrs[k] = v
            fv = self._fill_value(info, dt)
            if '_FillValue' not in attrs and fv is not None:   # netCDF-4 mirrors _FillValue in the fill-value message
                attrs['_FillValue'] = fv
            self.datasets[prefix + name] = Dataset(self, prefix + name, shape, dt, layout, filters, attrs, fv)
        elif 0x11 in info or 0x02 in info or 0x06 in info:
            self._walk_group(addr, prefix + name + '/', depth + 1)

    def _attribute_item(self, q):
        if self._m[q] not in (1, 2, 3):
            return None
        k, v, total = self._attribute(q)
        return (k, v), total

    def _link_item(self, q):
        if self._m[q] != 1:
            return None
        link = self._link(q)
        if not link:
            return None
        return (link[0], link[1]), link[2]

    def _fill_value(self, info, dt):
        m = self._m
        if 0x05 in info:
            q = info[0x05][0]
            ver = m[q]
            if ver in (1, 2):
                if ver == 1 or m[q + 3]:
                    n = self._u(q + 4, 4)
                    if n == dt.itemsize:
                        return numpy.frombuffer(m[q + 8:q + 8 + n], dt)[0]
            elif ver == 3 and m[q + 1] & 0x20:
                n = self._u(q + 2, 4)
                if n == dt.itemsize:
                    return numpy.frombuffer(m[q + 6:q + 6 + n], dt)[0]
        if 0x04 in info:
            q = info[0x04][0]
            n = self._u(q, 4)
            if n == dt.itemsize:
                return numpy.frombuffer(m[q + 4:q + 4 + n], dt)[0]
        return None

    def _link(self, q):
        m = self._m
        if m[q] != 1:
            return None
        flags = m[q + 1]
        p = q + 2
        ltype = 0
        if flags & 0x08:
            ltype = m[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nsz = 1 << (flags & 3)
        nlen = self._u(p, nsz)
        p += nsz
        name = bytes(m[p:p + nlen]).decode('utf-8', 'replace')
        p += nlen
        if ltype != 0:
            return None          # soft / external links are not followed
        return name, self._addr(p), p + self._O - q

    def _symbol_table(self, btree, heap):
        m = self._m
        hp = self._base + heap
        if m[hp:hp + 4] != b'HEAP':
            raise Hdf5Error('bad local heap')
        data = self._base + self._addr(hp + 8 + 2 * self._L)

        def name_at(off):
            e = m.find(b'\0', data + off)
            return bytes(m[data + off:e]).decode('utf-8', 'replace')

        def node(a):
            p = self._base + a
            if m[p:p + 4] == b'SNOD':
                n = self._u(p + 6, 2)
                q = p + 8
                for _ in range(n):
                    yield name_at(self._addr(q)), self._addr(q + self._O)
                    q += 2 * self._O + 24
                return
            if m[p:p + 4] != b'TREE' or m[p + 4] != 0:
                raise Hdf5Error('bad group B-tree node')
            n = self._u(p + 6, 2)
            q = p + 8 + 2 * self._O
            for i in range(n):
                q += self._L                      # key
                yield from node(self._addr(q))
                q += self._O
        yield from node(btree)

    def _fractal_heap(self, addr):
        """Geometry of a fractal heap: returns locate(offset) -> file position of the object at that heap offset, and
        blocks() -> [(file position of the first object, end)] of its direct blocks."""
        m, p = self._m, self._base + addr
        if m[p:p + 4] != b'FRHP' or m[p + 4] != 0:
            raise Hdf5Error('bad fractal heap header')
        id_len = self._u(p + 5, 2)
        filt_len = self._u(p + 7, 2)
        flags = m[p + 9]
        max_managed = self._u(p + 10, 4)
        q = p + 14 + self._L + self._O + self._L + self._O           # next huge id, huge btree, free space, fsm addr
        q += 8 * self._L                                             # managed space ... number of tiny objects
        width = self._u(q, 2)
        start_size = self._len(q + 2)
        max_direct = self._len(q + 2 + self._L)
        max_heap_bits = self._u(q + 2 + 2 * self._L, 2)
        root = self._addr(q + 2 + 2 * self._L + 4)
        nrows = self._u(q + 2 + 2 * self._L + 4 + self._O, 2)
        if filt_len:
            raise Hdf5Error('filtered fractal heaps are not supported')
        off_bytes = (max_heap_bits + 7) // 8
        hdr = 5 + self._O + off_bytes + (4 if flags & 2 else 0)      # direct block prefix
        direct = []                                                  # (heap offset, size, file position)
        if root != UNDEF:
            if nrows == 0:
                direct.append((0, start_size, self._base + root))
            else:
                b = self._base + root
                if m[b:b + 4] != b'FHIB':
                    raise Hdf5Error('bad fractal heap indirect block')
                q0 = b + 5 + self._O + off_bytes
                hoff = 0
                for r in range(nrows):
                    size = start_size * (1 << max(0, r - 1))
                    if size > max_direct:
                        break    # deeper indirect blocks: not needed for the handful of variables of a NEMO file
                    for _ in range(width):
                        a = self._addr(q0)
                        q0 += self._O
                        if a != UNDEF:
                            direct.append((hoff, size, self._base + a))
                        hoff += size
        len_bytes = min((max(max_direct.bit_length(), 1) + 7) // 8, (max(max_managed.bit_length(), 1) + 7) // 8)

        def locate(heap_id):
            """heap_id: bytes of a managed-object heap ID -> (file position, length) or None."""
            if (heap_id[0] >> 4) & 3 != 0:
                return None        # huge / tiny objects: not produced for links and small attributes
            off = int.from_bytes(heap_id[1:1 + off_bytes], 'little')
            ln = int.from_bytes(heap_id[1 + off_bytes:1 + off_bytes + len_bytes], 'little')
            for hoff, size, pos in direct:
                if hoff <= off < hoff + size:
                    return pos + (off - hoff), ln
            return None

        def blocks():
            return [(pos + hdr, pos + size) for _, size, pos in direct if m[pos:pos + 4] == b'FHDB']
        return locate, blocks, id_len

    def _btree2_heap_ids(self, addr, id_offset, id_len):
        """Heap IDs of the records of a version-2 B-tree whose root is a leaf (enough for tens of variables); None if
        the tree is deeper (the caller then scans the heap blocks instead)."""
        m = self._m
        if addr == UNDEF:
            return None
        p = self._base + addr
        if m[p:p + 4] != b'BTHD':
            return None
        rec_size, depth = self._u(p + 10, 2), self._u(p + 12, 2)
        root = self._addr(p + 16)
        nrec = self._u(p + 16 + self._O, 2)
        if depth != 0 or root == UNDEF:
            return None
        b = self._base + root
        if m[b:b + 4] != b'BTLF':
            return None
        return [bytes(m[b + 6 + i * rec_size + id_offset: b + 6 + i * rec_size + id_offset + id_len]) for i in range(nrec)]

    def _fractal_heap_objects(self, addr, item, btree=UNDEF, id_offset=0):
        """Objects (links / attributes) stored in a fractal heap: through the name-index B-tree when it is a single
        leaf (exact, immune to deleted objects), else by parsing each direct block front to back."""
        locate, blocks, id_len = self._fractal_heap(addr)
        ids = self._btree2_heap_ids(btree, id_offset, id_len)
        if ids is not None:
            for hid in ids:
                loc = locate(hid)
                if loc is None:
                    continue
                it = item(loc[0])
                if it is not None:
                    yield it[0]
            return
        for q0, end in blocks():
            while q0 + 4 < end:
                it = item(q0)
                if it is None:
                    break
                yield it[0]
                q0 += it[1]

    def _chunk_index_v4(self, kind, addr, shape, cshape, nbytes, unlimited=None):
        if kind == 'earray':
            return _chunk_index_earray(self, addr, shape, cshape, nbytes, unlimited)
        if kind == 'btree2':
            return _chunk_index_btree2(self, addr, shape, cshape, nbytes)
        return _chunk_index_v4_impl(self, kind, addr, shape, cshape, nbytes)

    def _chunk_btree(self, addr, rank):
        m = self._m
        if addr == UNDEF:
            return
        p = self._base + addr
        if m[p:p + 4] != b'TREE' or m[p + 4] != 1:
            raise Hdf5Error('bad chunk B-tree node')
        level, n = m[p + 5], self._u(p + 6, 2)
        q = p + 8 + 2 * self._O
        ksize = 8 + 8 * (rank + 1)
        for _ in range(n):
            size, mask = self._u(q, 4), self._u(q + 4, 4)
            offs = tuple(self._u(q + 8 + 8 * i, 8) for i in range(rank))
            child = self._addr(q + ksize)
            q += ksize + self._O
            if level == 0:
                yield offs, size, mask, child
            else:
                yield from self._chunk_btree(child, rank)


def _chunk_index_v4_impl(h5, kind, addr, shape, cshape, nbytes):
    m, O = h5._m, h5._O
    grid = [-(-s // c) for s, c in zip(shape, cshape)]
    nchunks = int(numpy.prod(grid))

    def offs_of(k):     # chunk number -> element offsets: row-major over the chunk grid
        o = []
        for g, c in zip(reversed(grid), reversed(cshape)):
            o.append((k % g) * c)
            k //= g
        return tuple(reversed(o))

    if addr == UNDEF:
        return
    if kind == 'implicit':
        for k in range(nchunks):
            yield offs_of(k), nbytes, 0, addr + k * nbytes
        return
    # ---- fixed array: header FAHD -> data block FADB (optionally paged)
    p = h5._base + addr
    if m[p:p + 4] != b'FAHD' or m[p + 4] != 0:
        raise Hdf5Error('bad fixed-array header')
    client, esize, page_bits = m[p + 5], m[p + 6], m[p + 7]
    nelm = h5._len(p + 8)
    dblk = h5._addr(p + 8 + h5._L)
    if dblk == UNDEF:
        return
    if nelm < nchunks or client not in (0, 1):
        raise Hdf5Error('fixed-array index does not match the dataset')
    q = h5._base + dblk
    if m[q:q + 4] != b'FADB' or m[q + 4] != 0:
        raise Hdf5Error('bad fixed-array data block')
    q += 6 + O                                   # signature, version, client id, header address
    szlen = esize - O - 4                        # filtered chunks: address, chunk size (szlen bytes), filter mask

    def element(r):
        a = h5._addr(r)
        if client == 1:
            return h5._u(r + O, szlen), h5._u(r + O + szlen, 4), a
        return nbytes, 0, a

    page_n = 1 << page_bits
    if nelm <= page_n:                           # one un-paged block: elements, checksum
        for k in range(nchunks):
            size, mask, a = element(q + k * esize)
            yield offs_of(k), size, mask, a
        return
    npages = -(-nelm // page_n)
    bitmap = q
    q += (npages + 7) // 8 + 4                   # page-initialised bitmap, checksum of the prefix
    for pg in range(npages):
        n_in = min(page_n, nelm - pg * page_n)
        if m[bitmap + pg // 8] & (0x80 >> (pg % 8)):    # most significant bit first; untouched pages hold no chunks
            for e in range(n_in):
                k = pg * page_n + e
                if k < nchunks:
                    size, mask, a = element(q + e * esize)
                    yield offs_of(k), size, mask, a
        q += n_in * esize + 4                    # every page carries its own checksum


def _chunk_index_earray(h5, addr, shape, cshape, nbytes, unlimited):
    """Extensible-array chunk index (HDF5 1.10 'latest' files, one unlimited dimension): header EAHD -> index block EAIB
    (first elements, data-block and super-block addresses) -> super blocks EASB -> data blocks EADB (paged when large).
    Chunks are numbered row-major with the unlimited dimension moved to the front."""
    m, O, base = h5._m, h5._O, h5._base
    if addr == UNDEF:
        return
    p = base + addr
    if m[p:p + 4] != b'EAHD' or m[p + 4] != 0:
        raise Hdf5Error('bad extensible-array header')
    client, esize, max_bits, idx_elmts, dmin, sbmin, page_bits = (m[p + 5 + i] for i in range(7))
    iblk = h5._addr(p + 12 + 6 * h5._L)
    if client not in (0, 1) or dmin == 0 or sbmin == 0 or dmin & (dmin - 1) or sbmin & (sbmin - 1):
        raise Hdf5Error('unsupported extensible-array parameters')
    if iblk == UNDEF:
        return
    log2 = lambda v: v.bit_length() - 1
    nsblks = 1 + (max_bits - log2(dmin))
    off_size = (max_bits + 7) // 8
    page_n = 1 << page_bits
    sb_ndblks = [1 << (u // 2) for u in range(nsblks)]
    sb_dnelm = [(1 << ((u + 1) // 2)) * dmin for u in range(nsblks)]
    sb_start_idx, sb_start_dblk, a, b = [], [], 0, 0
    for u in range(nsblks):
        sb_start_idx.append(a)
        sb_start_dblk.append(b)
        a += sb_ndblks[u] * sb_dnelm[u]
        b += sb_ndblks[u]
    ib_nsblks = 2 * log2(sbmin)
    ndblk_addrs = 2 * (sbmin - 1)
    szlen = esize - O - 4

    q = base + iblk
    if m[q:q + 4] != b'EAIB' or m[q + 4] != 0:
        raise Hdf5Error('bad extensible-array index block')
    ib_elems = q + 6 + O
    ib_dblks = ib_elems + idx_elmts * esize
    ib_sblks = ib_dblks + ndblk_addrs * O

    def element(r):
        a_ = h5._addr(r)
        if client == 1:
            return h5._u(r + O, szlen), h5._u(r + O + szlen, 4), a_
        return nbytes, 0, a_

    sblock_cache = {}

    def sblock(u):      # (address of the data-block address table, address of the page bitmaps or None)
        if u not in sblock_cache:
            sa = h5._addr(ib_sblks + (u - ib_nsblks) * O)
            if sa == UNDEF:
                sblock_cache[u] = None
            else:
                r = base + sa
                if m[r:r + 4] != b'EASB' or m[r + 4] != 0:
                    raise Hdf5Error('bad extensible-array super block')
                r += 6 + O + off_size
                bitmaps = None
                if sb_dnelm[u] > page_n:
                    # ONE bit array for the whole super block, bit (data block * pages per block + page), most significant
                    # bit first; its allocated size rounds every data block up to whole bytes
                    npages = sb_dnelm[u] // page_n
                    bitmaps = (r, npages)
                    r += sb_ndblks[u] * ((npages + 7) // 8)
                sblock_cache[u] = (r, bitmaps)
        return sblock_cache[u]

    def lookup(idx):    # -> (size, mask, addr) or None when the element was never written
        if idx < idx_elmts:
            return element(ib_elems + idx * esize)
        e = idx - idx_elmts
        u = log2(e // dmin + 1)
        e -= sb_start_idx[u]
        d, within = divmod(e, sb_dnelm[u])
        bitmaps = None
        if u < ib_nsblks:
            da = h5._addr(ib_dblks + (sb_start_dblk[u] + d) * O)
        else:
            sb = sblock(u)
            if sb is None:
                return None
            da = h5._addr(sb[0] + d * O)
            bitmaps = sb[1]
        if da == UNDEF:
            return None
        r = base + da
        if m[r:r + 4] != b'EADB' or m[r + 4] != 0:
            raise Hdf5Error('bad extensible-array data block')
        r += 6 + O + off_size
        if sb_dnelm[u] > page_n:      # paged: prefix checksum, then pages of page_n elements each with its own checksum
            pg, within = divmod(within, page_n)
            if bitmaps is not None:
                bit = d * bitmaps[1] + pg
                if not m[bitmaps[0] + bit // 8] & (0x80 >> (bit % 8)):
                    return None
            r += 4 + pg * (page_n * esize + 4)
        return element(r + within * esize)

    # chunk numbering: row-major with the unlimited dimension swizzled to the front
    rank = len(shape)
    grid = [-(-s_ // c) for s_, c in zip(shape, cshape)]
    order = [unlimited] + [i for i in range(rank) if i != unlimited]
    sgrid = [grid[i] for i in order]
    nchunks = int(numpy.prod(grid))
    for k in range(nchunks):
        got = lookup(k)
        if got is None or got[2] == UNDEF:
            continue
        coords, kk = [0] * rank, k
        for pos in range(rank - 1, -1, -1):
            coords[order[pos]] = (kk % sgrid[pos]) * cshape[order[pos]]
            kk //= sgrid[pos]
        yield tuple(coords), got[0], got[1], got[2]


def _chunk_index_btree2(h5, addr, shape, cshape, nbytes):
    """Version-2 B-tree chunk index (record types 10 / 11: plain / filtered chunks keyed by their scaled offsets).  The
    widths of the per-child record counters of the internal nodes follow from the node size, the record size and the
    depth, as in the HDF5 library."""
    m, O, base = h5._m, h5._O, h5._base
    if addr == UNDEF:
        return
    p = base + addr
    if m[p:p + 4] != b'BTHD' or m[p + 4] != 0:
        raise Hdf5Error('bad version-2 B-tree header')
    rtype, node_size, rec_size, depth = m[p + 5], h5._u(p + 6, 4), h5._u(p + 10, 2), h5._u(p + 12, 2)
    root, root_nrec = h5._addr(p + 16), h5._u(p + 16 + O, 2)
    rank = len(shape)
    if rtype not in (10, 11):
        raise Hdf5Error('unsupported version-2 B-tree record type for a chunk index')
    szlen = rec_size - O - 4 - 8 * rank if rtype == 11 else 0
    if rtype == 10 and rec_size != O + 8 * rank or rtype == 11 and not (1 <= szlen <= 8):
        raise Hdf5Error('version-2 B-tree record size does not match the dataset')
    nbytes_of = lambda v: max(1, (v.bit_length() + 7) // 8)
    # per level: max records of a node, width of the "records in child" / "records in subtree" counters pointing AT it
    max_nrec = [(node_size - 10) // rec_size]          # leaf
    cum_max = [max_nrec[0]]
    for d in range(1, depth + 1):
        ptr = O + nbytes_of(max_nrec[d - 1]) + (nbytes_of(cum_max[d - 1]) if d > 1 else 0)
        max_nrec.append((node_size - 10 - ptr) // (rec_size + ptr))
        cum_max.append((max_nrec[d] + 1) * cum_max[d - 1] + max_nrec[d])

    def record(r):
        a = h5._addr(r)
        if rtype == 11:
            size, mask, q = h5._u(r + O, szlen), h5._u(r + O + szlen, 4), r + O + szlen + 4
        else:
            size, mask, q = nbytes, 0, r + O
        offs = tuple(h5._u(q + 8 * i, 8) * cshape[i] for i in range(rank))
        return offs, size, mask, a

    def node(a, nrec, d):
        r = base + a
        if m[r:r + 4] != (b'BTIN' if d > 0 else b'BTLF') or m[r + 4] != 0 or m[r + 5] != rtype:
            raise Hdf5Error('bad version-2 B-tree node')
        recs = r + 6
        for i in range(nrec):
            yield record(recs + i * rec_size)
        if d > 0:
            q = recs + nrec * rec_size
            w1 = nbytes_of(max_nrec[d - 1])
            w2 = nbytes_of(cum_max[d - 1]) if d > 1 else 0
            for i in range(nrec + 1):
                child, cn = h5._addr(q), h5._u(q + O, w1)
                q += O + w1 + w2
                yield from node(child, cn, d - 1)

    if root != UNDEF:
        for offs, size, mask, a in node(root, root_nrec, depth):
            if a != UNDEF:
                yield offs, size, mask, a


def read_variables(path, wanted=None):
    """{name: array} (+ '_FillValue_<name>') of the top-level datasets of an HDF5 / NetCDF-4 file."""
    out = {}
    with File(path) as f:
        for name, ds in f.datasets.items():
            if wanted is not None and name not in wanted:
                continue
            if isinstance(ds, Hdf5Error):
                raise ds
            out[name] = ds.read()
            if ds.fill_value is not None:
                out['_FillValue_' + name] = numpy.asarray(ds.fill_value)
    return out
