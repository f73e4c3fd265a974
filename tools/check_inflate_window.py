"""The device decoder against zlib on streams whose matches reach far back (periods of 9 000 .. 30 000 bytes: beyond an
8 KiB window, inside DEFLATE's 32 KiB) plus mixed content, more streams than are resident at once.  Run against the
`make window8k` build (NEMOFLUX_AMD_LIB=build/window8k/libnemoflux_amd_w8k.so) it exercises the decoder's FAR path on the
device; against the shipped library the same streams stay inside the LDS window."""
import os
import sys
import zlib

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nemoflux_amd.ingest import ChunkDecoder  # noqa: E402

dec = ChunkDecoder()
cap = dec.capacity()
rng = numpy.random.default_rng(5)
size = 120000
datas, streams = [], []
n = cap + 200
for i in range(n):
    kind = i % 3
    if kind == 0:       # every match is far for an 8 KiB window
        d = numpy.resize(rng.integers(0, 256, int(rng.integers(9000, 30000)), dtype=numpy.uint8), size)   # cyclic fill: `size` bytes, no more
    elif kind == 1:     # near and far mixed: a short period with a long one laid over it
        d = numpy.resize(rng.integers(0, 16, int(rng.integers(50, 400)), dtype=numpy.uint8), size).copy()
        d[::int(rng.integers(10000, 20000))] ^= 0x55
    else:               # shuffled floats: literals + short matches
        f = (numpy.cumsum(rng.standard_normal(size // 4)) * 1e-2).astype('<f4')
        d = numpy.ascontiguousarray(f.view(numpy.uint8).reshape(-1, 4).T).reshape(-1)
    d = numpy.ascontiguousarray(d[:size])
    datas.append(d)
    streams.append(zlib.compress(d.tobytes(), int(rng.integers(1, 10))))
out = dec.decode_streams(streams, size)
bad = [i for i in range(n) if not numpy.array_equal(out[i], datas[i])]
assert not bad, bad[:10]
print(f'inflate window check OK: {n} streams (capacity {cap}) bit-identical to zlib with {os.environ.get("NEMOFLUX_AMD_LIB", "the shipped library")}')
