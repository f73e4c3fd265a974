"""CPU suite: the wavefront DEFLATE decoder of the file-ingest path (nemoflux_amd/csrc/nf_inflate_core.h), built for the
host (one "lane", no barriers), against zlib's own streams: every block type (stored, fixed, dynamic), every compression
level and strategy, all four alignments of the stream's first byte, long-distance matches across the 32 KiB window,
overlapping copies, byte-shuffled float32 data like the HDF5 chunks of NEMO files, and malformed streams (which must end
in an error code, never in an out-of-bounds write).  The device build of the same source is checked by
tests/test_gpu_inflate.py."""
import ctypes
import os
import subprocess
import zlib

import numpy
import pytest

from conftest import ROOT


@pytest.fixture(scope='module', params=[32768, 8192], ids=['window32k', 'window8k'])
def host_inflate(request, tmp_path_factory):
    """the product's 32 KiB LDS window, and an 8 KiB one: matches that reach further back then take the far path of the
    copy (history read from the stream's flushed output instead of the window)"""
    so = str(tmp_path_factory.mktemp('nfi') / f'libnfi_host_{request.param}.so')
    subprocess.check_call(['g++', '-O2', '-Wall', '-Wno-unknown-pragmas', f'-DNFI_WINDOW={request.param}', '-shared', '-fPIC',
                           '-o', so, os.path.join(ROOT, 'tests', 'native', 'inflate_host.cpp')])
    lib = ctypes.CDLL(so)
    # one decoder state per wavefront in LDS: four fit a CU's 160 KiB (ten with the small window)
    assert lib.nfi_host_ctx_bytes() <= (40 if request.param == 32768 else 16) * 1024

    def inflate(comp, out_len, skip=0, readable_extra=16):
        buf = numpy.zeros(len(comp) + skip + 64, numpy.uint8)
        off = (-buf.ctypes.data) % 4 + skip
        buf[off:off + len(comp)] = numpy.frombuffer(comp, numpy.uint8)
        out = numpy.full(out_len + 16, 0xAB, numpy.uint8)
        oal = (-out.ctypes.data) % 4
        readable = ((skip + len(comp) + 3) // 4) * 4 + readable_extra
        rc = lib.nfi_host_inflate(ctypes.c_void_p(buf.ctypes.data + off), len(comp), readable,
                                  ctypes.c_void_p(out.ctypes.data + oal), out_len)
        assert (out[oal + out_len:oal + out_len + 8] == 0xAB).all(), 'wrote past the end of the output'
        return rc, bytes(out[oal:oal + out_len])
    return inflate


def payloads():
    rng = numpy.random.default_rng(0)
    f = (numpy.sin(numpy.arange(300000) * 1e-3) * (1 + 1e-3 * rng.standard_normal(300000))).astype('<f4')
    return {
        'empty': b'', 'one': b'a', 'short': b'hello hello hello hello hello', 'zeros': bytes(100000),
        'random': rng.integers(0, 256, 70000, dtype=numpy.uint8).tobytes(),
        'two_bit': rng.integers(0, 4, 200000, dtype=numpy.uint8).tobytes(),
        'f32_shuffled': numpy.ascontiguousarray(f.view(numpy.uint8).reshape(-1, 4).T).tobytes(),     # HDF5 shuffle filter
        'f32_plain': f.tobytes(), 'period3': b'abc' * 100000,
        'far_matches': (rng.integers(0, 256, 32768, dtype=numpy.uint8).tobytes()) * 4,                # distance 32768
    }


@pytest.mark.parametrize('name', list(payloads()))
def test_inflate_matches_zlib(name, host_inflate):
    data = payloads()[name]
    for level in (0, 1, 4, 6, 9):
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
            comp = co.compress(data) + co.flush()
            for skip in ((0, 1, 2, 3) if level in (0, 4) else (0,)):
                rc, out = host_inflate(comp, len(data), skip)
                assert rc == 0 and out == data, (name, level, strategy, skip, rc)


def test_inflate_multi_block_and_small_window_streams(host_inflate):
    rng = numpy.random.default_rng(3)
    data = rng.integers(0, 7, 400000, dtype=numpy.uint8).tobytes()
    co = zlib.compressobj(6, zlib.DEFLATED, 15)
    comp = b''.join(co.compress(data[i:i + 30000]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), 30000)) + co.flush()
    assert host_inflate(comp, len(data)) == (0, data)               # empty stored blocks between the Huffman blocks
    for wbits in (9, 12):
        co = zlib.compressobj(6, zlib.DEFLATED, wbits)
        comp = co.compress(data) + co.flush()
        assert host_inflate(comp, len(data)) == (0, data)


def test_malformed_streams_end_in_an_error(host_inflate):
    rng = numpy.random.default_rng(4)
    data = rng.integers(0, 16, 50000, dtype=numpy.uint8).tobytes()
    comp = zlib.compress(data, 6)
    assert host_inflate(comp, len(data))[0] == 0
    assert host_inflate(comp, len(data) - 1)[0] == 6               # more output than expected
    assert host_inflate(comp, len(data) + 1)[0] == 6               # less
    assert host_inflate(b'\x00\x00' + comp[2:], len(data))[0] == 1  # not a zlib header
    assert host_inflate(comp[:len(comp) // 2], len(data), readable_extra=0)[0] != 0   # truncated
    for trial in range(300):                # random bit flips: always noticed (structure, length or the Adler-32 trailer)
        c = bytearray(comp)
        for _ in range(int(rng.integers(1, 4))):
            c[int(rng.integers(2, len(c)))] ^= 1 << int(rng.integers(0, 8))
        if bytes(c) == comp:
            continue
        rc, out = host_inflate(bytes(c), len(data))
        assert rc != 0, trial
    for trial in range(100):                                        # pure noise behind a valid header
        noise = b'\x78\x9c' + rng.integers(0, 256, 2000, dtype=numpy.uint8).tobytes()
        host_inflate(noise, 10000)
