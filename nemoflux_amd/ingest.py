"""File ingest straight to HBM (replaces the lazy NetCDF read of nemoflux/field.py:149): the
deflated, byte-shuffled HDF5 chunks of one time step are copied to the device as they sit in the file and inflated there
by nf_inflate.hip, one wavefront per chunk -- instead of zlib on the host cores, which is what bounds a file-backed pass.

    dec = ChunkDecoder()
    plan = lazy_variable.device_plan(t)            # nemoflux_amd.hdf5min: None when the layout needs the host path
    staged = dec.gather(lazy_variable.raw_bytes(), plan)     # host only (runs on the prefetch thread)
    dec.decode(staged, device_pointer, nbytes)     # H2D of the compressed bytes + inflate + un-shuffle, synchronous
"""
import ctypes

import numpy

from . import _lib
from ._lib import lib, check


class _Pinned(object):
    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        check(lib.nf_host_alloc(ctypes.byref(p), max(int(nbytes), 64)))
        self.ptr, self.nbytes = p.value, max(int(nbytes), 64)
        self.array = numpy.ctypeslib.as_array((ctypes.c_ubyte * self.nbytes).from_address(self.ptr))

    def __del__(self):
        try:
            if self.ptr:
                lib.nf_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class StagedChunks(object):
    """Compressed chunks of one slab, gathered into pinned host memory, with their placement."""

    def __init__(self, pinned, used, in_off, in_len, origin, plan):
        self.pinned, self.used, self.in_off, self.in_len, self.origin, self.plan = pinned, used, in_off, in_len, origin, plan


class ChunkDecoder(object):
    def __init__(self, threads=None):
        _lib.require_gpu()
        if threads is None:
            import os
            threads = int(os.environ.get('NF_GATHER_THREADS', 16))
        self._h = ctypes.c_void_p()
        check(lib.nf_inflater_new(ctypes.byref(self._h)))
        self._threads = threads

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                lib.nf_inflater_del(ctypes.byref(self._h))
        except Exception:
            pass

    def share_scratch_of(self, owner):
        """Decode into `owner`'s scratch (the decoded group, job and status arrays) instead of an own one: for decoders whose
        decode() calls never overlap in time but which each keep their own compressed buffer (the stager's two slots)."""
        check(lib.nf_inflater_share_scratch(ctypes.byref(self._h), ctypes.byref(owner._h)))
        self._scratch_owner = owner      # keeps it alive

    @staticmethod
    def capacity():
        """chunks the device decodes at once (resident decoder wavefronts)"""
        n = ctypes.c_int()
        check(lib.nf_inflater_capacity(ctypes.byref(n)))
        return n.value

    @staticmethod
    def new_pinned(nbytes):
        """Pinned staging buffer; allocate on the thread that owns the GPU context, fill (gather) on any thread."""
        return _Pinned(nbytes)

    def gather(self, raw, plan, pinned):
        """Copy the compressed chunks of `plan` out of the mapped file `raw` into `pinned`, back to back (8-byte aligned).
        Pure host work (memcpy releases the GIL): meant for the prefetch thread."""
        return self.gather_many([(raw, plan, 0)], pinned, plan['slab_dims'][0])[0]

    def upload_ranges(self, plan_ranges, used):
        """Early upload without the staging copy: `plan_ranges` = (source addresses, destination offsets, lengths) as
        gather_many(..., copy=False) returned them; the chunks go from the mapped file to HBM one hipMemcpyAsync each."""
        sa, do, ll = plan_ranges
        check(lib.nf_inflater_upload_ranges(ctypes.byref(self._h), sa.ctypes.data, do.ctypes.data, ll.ctypes.data, len(ll), int(used)))

    def gather_many(self, items, pinned, total_nz, threads=None, copy=True):
        """items: [(mapped file, device_plan of one slab, z offset of that slab in the group's slab)] -- e.g. uo and vo of
        several time steps, stacked along z into one (total_nz, ny, nx) slab.  The compressed chunks of all of them are
        copied into `pinned` back to back; slabs of the same chunk geometry are merged into ONE StagedChunks (= one launch,
        one wavefront per chunk); returns the list of StagedChunks (one per distinct geometry)."""
        groups = {}
        src_addr, dst_addr, lens = [], [], []
        keep = []
        pos = 0
        for raw, plan, zoff in items:
            key = (tuple(plan['chunk_dims']), tuple(plan['slab_dims'][1:]), plan['elem_size'], plan['shuffled'])
            g = groups.setdefault(key, dict(in_off=[], in_len=[], origin=[], plan=plan))
            src = numpy.frombuffer(raw, numpy.uint8)
            keep.append(src)
            base = src.ctypes.data
            for a, ln, org in plan['chunks']:
                if a < 0 or a + ln > src.size:
                    raise RuntimeError('ERROR: a chunk of the device plan lies outside the mapped file')
                g['in_off'].append(pos)
                g['in_len'].append(ln)
                g['origin'].append((org[0] + zoff, org[1], org[2]))
                src_addr.append(base + a)
                dst_addr.append(pinned.ptr + pos)
                lens.append(ln)
                pos += (ln + 7) & ~7
        if pos > pinned.nbytes:
            raise RuntimeError('ERROR: staging buffer too small for the compressed chunks of this group of time steps')
        # one native call (nf_host_gather: its own threads, no interpreter lock between the copies)
        threads = self._threads if threads is None else threads
        sa, da = numpy.array(src_addr, numpy.uint64), numpy.array(dst_addr, numpy.uint64)
        ll = numpy.array(lens, numpy.int64)
        if copy:
            check(lib.nf_host_gather(sa.ctypes.data, da.ctypes.data, ll.ctypes.data, len(lens), max(1, int(threads))))
        self.last_ranges = (sa, (da - numpy.uint64(pinned.ptr)).astype(numpy.int64), ll, keep)   # for upload_ranges
        out = []
        for key, g in groups.items():
            plan = dict(g['plan'])
            plan['slab_dims'] = (int(total_nz),) + tuple(plan['slab_dims'][1:])
            out.append(StagedChunks(pinned, pos, numpy.array(g['in_off'], numpy.int64), numpy.array(g['in_len'], numpy.int64),
                                    numpy.ascontiguousarray(numpy.array(g['origin'], numpy.int64).reshape(-1, 3)), plan))
        return out

    @staticmethod
    def staging_bytes(lazy, nt):
        """Largest compressed size of any time step of `lazy` (padded the way gather lays the chunks out), or None when
        some step cannot be decoded on the device."""
        worst = 0
        for t in range(nt):
            plan = lazy.device_plan(t)
            if plan is None:
                return None
            worst = max(worst, sum((c[1] + 7) & ~7 for c in plan['chunks']))
        return worst

    def upload(self, pinned, used):
        """Copy the gathered compressed bytes to HBM now, on the decoder's own stream (complete at return): what the staging
        thread does with the NEXT group while the GPU decodes this one.  decode(..., uploaded=True) then skips the copy."""
        check(lib.nf_inflater_upload(ctypes.byref(self._h), ctypes.c_void_p(pinned.ptr), int(used)))

    def decode(self, staged, out_ptr, stream=None, uploaded=False):
        """H2D of the gathered bytes (unless upload() already put them in HBM), inflate + un-shuffle + placement on the device
        into the slab at out_ptr.  Synchronous; raises NemofluxError naming the first malformed chunk."""
        n = len(staged.in_len)
        plan = staged.plan
        status = numpy.zeros(max(n, 1), numpy.int32)
        ll = _lib.c_ll_p
        cd = numpy.array(plan['chunk_dims'], numpy.int64)
        sd = numpy.array(plan['slab_dims'], numpy.int64)
        check(lib.nf_inflater_run(ctypes.byref(self._h), None if uploaded else ctypes.c_void_p(staged.pinned.ptr), int(staged.used),
                                  staged.in_off.ctypes.data_as(ll), staged.in_len.ctypes.data_as(ll), n,
                                  int(plan['chunk_bytes']), int(plan['elem_size']), int(plan['shuffled']),
                                  cd.ctypes.data_as(ll), sd.ctypes.data_as(ll), staged.origin.ctypes.data_as(ll),
                                  ctypes.c_void_p(int(out_ptr)), ctypes.c_void_p(stream) if stream else None,
                                  status.ctypes.data_as(_lib.c_int_p)))
        return status[:n]

    def decode_streams(self, streams, out_len, elem_size=1, shuffled=0):
        """Convenience for tests / tools: inflate a list of zlib streams that all decode to out_len bytes, one per row of a
        (n, 1, out_len / elem_size) slab; returns the decoded rows (host numpy uint8, shape (n, out_len))."""
        n = len(streams)
        total_in = sum((len(s) + 7) & ~7 for s in streams)
        pinned = _Pinned(total_in + 16)
        in_off = numpy.zeros(n, numpy.int64)
        pos = 0
        for i, s in enumerate(streams):
            in_off[i] = pos
            pinned.array[pos:pos + len(s)] = numpy.frombuffer(s, numpy.uint8)
            pos += (len(s) + 7) & ~7
        ne = out_len // elem_size
        plan = dict(chunk_dims=(1, 1, ne), slab_dims=(n, 1, ne), chunk_bytes=out_len, elem_size=elem_size, shuffled=shuffled)
        origin = numpy.zeros((n, 3), numpy.int64)
        origin[:, 0] = numpy.arange(n)
        staged = StagedChunks(pinned, pos, in_off, numpy.array([len(s) for s in streams], numpy.int64), origin, plan)
        buf = _lib.DeviceBuffer(max(n * out_len, 16))
        try:
            self.decode(staged, buf.ptr)
            flat = buf.download((max(n * out_len, 16),), numpy.uint8)
        finally:
            buf.free()
        return flat[:n * out_len].reshape(n, out_len)
