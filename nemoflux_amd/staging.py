"""Staging of file-backed uo / vo for the engine, one time step (host path) or one GROUP of time steps (device path) at a time.

Replaces the lazy NetCDF read of nemoflux/field.py:149 (`nc[name][timeIndex, :, :, :]`, inflated by netCDF4 on the host at
every update).  Two slots are kept, so that while the GPU works on the steps of one slot a background host thread prepares
the next ones in the other (fluxviz's 't' key and fluxplot's loop walk the steps in order):

* host path  -- a variable the device cannot decode (contiguous data, exotic chunking / filters, classic NetCDF): the step is
  read / inflated by nemoflux_amd.hdf5min on all host cores into a pinned buffer and handed to the engine as host memory
  (or copied into the slab when the other variable is on the device path).
* device path -- deflated (+ shuffled) HDF5 chunks, what netCDF-4 / XIOS write: the host thread only GATHERS the compressed
  chunks of a group of G steps into pinned memory and, when it runs in the background, copies them to HBM at once on the
  decoder's own stream (the copy then runs under the GPU's decode of the previous group); nf_inflate.hip inflates them
  there, one wavefront per chunk, all G x 2 x (chunks per step) of them in one launch.  The decoder is serial inside a
  chunk, so its throughput comes from the number of chunks in flight (4 per CU = 1024 on the chip): G is chosen to get
  there.
"""
import concurrent.futures
import ctypes
import os

import numpy

from . import _lib
from ._lib import lib, check


_TRACE = os.environ.get('NF_STAGE_TRACE', '0') != '0'


class StepStager(object):
    def __init__(self, sources, nt, nz, ny, nx, dtype, host_array, prefetch=True, gpu_decode=True, max_group_bytes=16 << 30):
        self.src = tuple(sources)
        self.nt, self.nz, self.ny, self.nx = nt, nz, ny, nx
        self.dtype = numpy.dtype(dtype)
        self.step_bytes = nz * ny * nx * self.dtype.itemsize
        self._host_array = host_array
        self._prefetch_on = bool(prefetch)
        self.decoder = None
        self._early_upload = os.environ.get('NF_EARLY_UPLOAD', '1') != '0'
        self._bg_threads = int(os.environ.get('NF_GATHER_THREADS_BG', 8))
        # early upload: gather into pinned memory, then ONE copy (default) -- or, NF_DIRECT_UPLOAD=1, one hipMemcpyAsync per
        # chunk straight from the mapped file (pageable source, no staging copy).  Both were measured again in round 6
        # (profiles/r06_group_pipeline.txt): 20.4 against 21.2 ms per C3 step at G = 6, 24.2 against 24.7 at G = 13, and the
        # per-chunk form is erratic under a running decode (62 to 309 ms for groups of the same size); round 3 had it 20.9 : 21.8
        self._direct_upload = os.environ.get('NF_DIRECT_UPLOAD', '0') != '0'
        self.comp_bytes = [None, None]            # staging size per step of a variable on the device path
        self.group = 1
        if gpu_decode and os.environ.get('NF_GPU_INFLATE', '1') != '0':
            from .ingest import ChunkDecoder
            # A variable takes the device path only when its chunks decode to exactly what the slab holds: elements of the
            # stager's dtype (uo's; a vo of another type is converted on the host path, like the reference's numpy would)
            # and slabs of (nz, ny, nx) -- the decoder writes elem_size bytes per element at element offsets of the slab
            need = [ChunkDecoder.staging_bytes(s, nt) if self._device_ok(s) else None for s in self.src]
            if any(n is not None for n in need):
                self.decoder = ChunkDecoder()
                # one decoder per slot: its compressed-bytes buffer in HBM is filled by the staging thread (early upload)
                # while the other slot's group is being decoded
                # (only the compressed buffer needs that: decode() is synchronous on the caller's thread, so the second decoder
                # borrows the first one's scratch for the decoded group -- one copy of it in HBM, not two)
                self._decoders = [self.decoder, ChunkDecoder()]
                self._decoders[1].share_scratch_of(self.decoder)
                self.comp_bytes = need
                per_step = sum(len(s.device_plan(0)['chunks']) for s, n in zip(self.src, need) if n is not None)
                # Resident decoder wavefronts: 4 per CU = 1024 streams.  A launch whose streams all start at once ends when its
                # slowest stream ends, with the slots of the faster ones idle; TWO waves of streams per launch (later chunks
                # start as earlier ones finish) measured 13.5 ms per C3 step against 16.2 with one (NF_INFLATE_WAVES)
                waves = max(1, int(os.environ.get('NF_INFLATE_WAVES', 2)))
                g = max(1, waves * ChunkDecoder.capacity() // max(per_step, 1))
                g = min(g, nt, max(1, int(max_group_bytes // (2 * self.step_bytes))))
                # ... but a pass needs groups to overlap: the first group's host and device halves run under nothing, so a
                # short series is cut into at least four groups (24 steps of the C3 image: 2 groups of 13 -> 24.7 ms per step,
                # 4 groups of 6 -> 20.4, although a launch of 900 chunks decodes at 16.2 ms per step against 13.5 for 1 950)
                g = min(g, max(1, -(-nt // 4)))
                self.group = int(os.environ.get('NF_INFLATE_GROUP', g))
        self.on_device = self.decoder is not None
        self._slots = [None, None]                # per slot: dict of buffers
        self._range = [(-1, -1), (-1, -1)]        # steps [g0, g1) staged in the slot (host half done)
        self._uploaded = [(-1, -1), (-1, -1)]     # ... and decoded / copied into its HBM slab
        self._cur = 1
        self._pending = None
        self._pending_slot = -1
        self._pool = None

    def _device_ok(self, src):
        if not hasattr(src, 'device_plan') or numpy.dtype(getattr(src, 'dtype', None)) != self.dtype:
            return False
        if tuple(getattr(src, 'shape', ())) != (self.nt, self.nz, self.ny, self.nx):
            return False
        plan = src.device_plan(0)
        return (plan is not None and plan['elem_size'] == self.dtype.itemsize and
                tuple(plan['slab_dims']) == (self.nz, self.ny, self.nx))

    # ------------------------------------------------------------------------------------------ buffers (caller's thread)
    def _alloc(self, slot):
        if self._slots[slot] is not None:
            return
        shp = (self.nz, self.ny, self.nx)
        G = self.group
        b = dict(staged=[], host=[None, None], comp=None, slab=None)
        for k in (0, 1):
            if self.comp_bytes[k] is None:        # decoded on the host: one pinned step buffer per step of the group
                b['host'][k] = [self._host_array(shp, self.dtype) for _ in range(G)]
        if self.on_device:
            from .ingest import ChunkDecoder
            total = sum(n for n in self.comp_bytes if n is not None) * G
            b['comp'] = ChunkDecoder.new_pinned(total + 64)
            b['slab'] = _lib.DeviceBuffer(2 * G * self.step_bytes)
        self._slots[slot] = b

    # ------------------------------------------------------------------------------------------ host half (any thread)
    def _read_host(self, src, t, buf):
        if hasattr(src, 'read_step'):
            if numpy.dtype(src.dtype) == self.dtype:
                return src.read_step(t, out=buf)
            numpy.copyto(buf, src.read_step(t))
            return buf
        numpy.copyto(buf, src[t] if len(src.shape) == 4 else src)
        return buf

    def _stage(self, g0, g1, slot):
        """zlib, the native un-shuffle, memcpy and numpy's copies all release the GIL"""
        import time
        t_start = time.perf_counter()
        b = self._slots[slot]
        self._range[slot] = (-1, -1)
        items = []
        for t in range(g0, g1):
            for k in (0, 1):
                if self.comp_bytes[k] is not None:
                    items.append((self.src[k].raw_bytes(), self.src[k].device_plan(t), ((t - g0) * 2 + k) * self.nz))
                else:
                    self._read_host(self.src[k], t, b['host'][k][t - g0])
        # in the background the gather shares the host's memory system with whatever the caller's thread copies; with the
        # early upload the caller's thread copies nothing in the steady state (NF_GATHER_THREADS_BG, default 8)
        import threading
        nthreads = None if threading.current_thread() is threading.main_thread() else self._bg_threads
        early = bool(items) and nthreads is not None and self._early_upload
        direct = early and self._direct_upload
        dec = self._decoders[slot] if self.decoder is not None else None
        b['staged'] = dec.gather_many(items, b['comp'], 2 * (g1 - g0) * self.nz, nthreads, copy=not direct) if items else []
        b['early'] = False
        t_gather = time.perf_counter()
        if direct:       # background thread: the chunks go from the mapped file to HBM without the staging copy -- the runtime
            dec.upload_ranges(dec.last_ranges[:3], b['staged'][0].used)   # pipelines its bounce buffers with the DMA
            b['early'] = True
        elif early:      # ... or gathered into pinned memory first, then one copy (NF_DIRECT_UPLOAD=0)
            dec.upload(b['comp'], b['staged'][0].used)
            b['early'] = True
        self._uploaded[slot] = (-1, -1)
        self._range[slot] = (g0, g1)
        if _TRACE:
            print(f'# staging: host half of steps [{g0},{g1}) {1e3 * (time.perf_counter() - t_start):.1f} ms (gather '
                  f'{1e3 * (t_gather - t_start):.1f} ms on {nthreads or self.decoder._threads} threads, early upload '
                  f'{1e3 * (time.perf_counter() - t_gather):.1f} ms)', flush=True)

    # ------------------------------------------------------------------------------------------ device half (caller's thread)
    def _upload(self, slot):
        g0, g1 = self._range[slot]
        if self._uploaded[slot] == (g0, g1):
            return
        b = self._slots[slot]
        import time
        t_start = time.perf_counter()
        for staged in b['staged']:
            self._decoders[slot].decode(staged, b['slab'].ptr, uploaded=b.get('early', False))
        if _TRACE:
            print(f'# staging: device half of steps [{g0},{g1}) {1e3 * (time.perf_counter() - t_start):.1f} ms '
                  f'({sum(len(x.in_len) for x in b["staged"])} chunks, {sum(int(x.used) for x in b["staged"][:1]) / 1e6:.0f} MB compressed)',
                  flush=True)
        for k in (0, 1):
            if self.comp_bytes[k] is None:
                for t in range(g0, g1):
                    buf = b['host'][k][t - g0]
                    check(lib.nf_memcpy_h2d(b['slab'].ptr + ((t - g0) * 2 + k) * self.step_bytes, buf.ctypes.data, buf.nbytes))
        self._uploaded[slot] = (g0, g1)

    # ------------------------------------------------------------------------------------------ interface
    def _wait(self):
        if self._pending is not None:
            fut = self._pending
            self._pending = None
            self._pending_slot = -1
            fut.result()        # re-raises a read error of the background thread here, in the caller

    def _group_of(self, t):
        g0 = (t // self.group) * self.group
        return g0, min(g0 + self.group, self.nt)

    def get(self, t):
        """(address of uo[t], address of vo[t], on_device) -- prefetched already, being prefetched, or staged now."""
        slot = None
        for s in (0, 1):          # a slot the worker is filling shows the empty range until it is done
            if self._range[s][0] <= t < self._range[s][1] and s != self._pending_slot:
                slot = s
        if slot is None:
            self._wait()
            for s in (0, 1):
                if self._range[s][0] <= t < self._range[s][1]:
                    slot = s
        if slot is None:
            slot = 1 - self._cur          # never the slot the engine may still be reading
            self._alloc(slot)
            self._stage(*self._group_of(t), slot)
        self._cur = slot
        b = self._slots[slot]
        g0 = self._range[slot][0]
        if self.on_device:
            # the NEXT group's host half (gather + early upload) starts now, on the staging thread, so that it runs under
            # this group's device half (copy + inflate, > 100 ms) and not only under its few milliseconds of flux kernels
            if self._uploaded[slot] != self._range[slot]:
                self.prefetch(self.next_after(t))
            self._upload(slot)
            base = b['slab'].ptr + (t - g0) * 2 * self.step_bytes
            return base, base + self.step_bytes, 1
        return b['host'][0][t - g0].ctypes.data, b['host'][1][t - g0].ctypes.data, 0

    def prefetch(self, t):
        """Start the host half of the group that holds step t on the background thread (no-op if it is staged already)."""
        if not self._prefetch_on or not (0 <= t < self.nt):
            return
        if any(r[0] <= t < r[1] for r in self._range) or self._pending is not None:
            return
        if self._pool is None:
            self._pool = concurrent.futures.ThreadPoolExecutor(1, thread_name_prefix='nf-prefetch')
        slot = 1 - self._cur
        self._alloc(slot)        # HIP calls stay on the caller's thread; the worker only fills host buffers
        self._pending_slot = slot
        self._pending = self._pool.submit(self._stage, *self._group_of(t), slot)

    def invalidate(self):
        """forget what is staged (the buffers stay): the next get() reads the file again -- measurements, re-opened files"""
        self._wait()
        self._range = [(-1, -1), (-1, -1)]
        self._uploaded = [(-1, -1), (-1, -1)]

    def next_after(self, t):
        """first step after the group of t (what is worth prefetching while t's group is being worked on)"""
        return self._group_of(t)[1]

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)      # the worker writes into buffers this object owns
            self._pool = None
