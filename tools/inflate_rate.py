"""Decode rate of the device inflate by kind of data (the four byte planes of a shuffled float32 level: two of noise that
zlib emits as stored blocks, one of Huffman-coded literals, one of long matches).  Run under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/inflate_rate.py
and read the nf::k_inflate durations of the trace in launch order (two launches per case: 4 streams to warm up, then 256)."""
import os, sys, zlib, numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemoflux_amd.ingest import ChunkDecoder
rng = numpy.random.default_rng(1)
ny, nx = 1021, 1440
y = numpy.linspace(-90, 90, ny)[:, None]; x = numpy.linspace(-180, 180, nx)[None, :]
f = ((numpy.cos(2*numpy.pi*y/360) + numpy.sin(2*numpy.pi*x/360)) * 3.1).astype('<f4')
f *= (1 + 1e-3 * rng.standard_normal(f.shape).astype('<f4'))
sh = numpy.ascontiguousarray(f.view(numpy.uint8).reshape(-1, 4).T)
dec = ChunkDecoder()
# calibration streams (1 MB each): Huffman-only literals with ~4-bit and ~8-bit codes, and matches of 8 bytes
lit4 = rng.integers(0, 16, 1 << 20, dtype=numpy.uint8).tobytes()
lit8 = rng.integers(0, 256, 1 << 20, dtype=numpy.uint8).tobytes()
words = rng.integers(0, 256, (4096, 8), dtype=numpy.uint8)
m8 = words[rng.integers(0, 4096, 1 << 17)].tobytes()
cases = [('plane0: noise -> stored blocks', sh[0].tobytes(), 4, zlib.Z_DEFAULT_STRATEGY),
         ('plane2: Huffman literals', sh[2].tobytes(), 4, zlib.Z_DEFAULT_STRATEGY),
         ('plane3: long matches', sh[3].tobytes(), 4, zlib.Z_DEFAULT_STRATEGY),
         ('whole level (4 planes)', sh.tobytes(), 4, zlib.Z_DEFAULT_STRATEGY),
         ('literals, 4-bit codes', lit4, 6, zlib.Z_HUFFMAN_ONLY), ('literals, 8-bit codes', lit8, 6, zlib.Z_HUFFMAN_ONLY),
         ('matches of 8 bytes', m8, 9, zlib.Z_DEFAULT_STRATEGY)]
for name, data, level, strategy in cases:
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    c = co.compress(data) + co.flush()
    whole = name.startswith('whole level')       # the real thing: 4-byte elements behind HDF5's shuffle filter -> k_place16
    kw = dict(elem_size=4, shuffled=1) if whole else {}
    dec.decode_streams([c] * 4, len(data), **kw)
    out = dec.decode_streams([c] * 256, len(data), **kw)
    assert bytes(out[5]) == (f.tobytes() if whole else data)
    print(f'{name}: {len(data)} bytes from {len(c)}')
