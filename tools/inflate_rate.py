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
cases = [('plane0: noise -> stored blocks', sh[0].tobytes()), ('plane2: Huffman literals', sh[2].tobytes()),
         ('plane3: long matches', sh[3].tobytes()), ('whole level (4 planes)', sh.tobytes())]
for name, data in cases:
    c = zlib.compress(data, 4)
    dec.decode_streams([c] * 4, len(data))
    out = dec.decode_streams([c] * 256, len(data))
    assert bytes(out[5]) == data
    print(f'{name}: {len(data)} bytes from {len(c)}')
