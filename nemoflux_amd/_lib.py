"""ctypes binding of libnemoflux_amd.so (the C ABI declared in include/nemoflux_amd.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises at import, and
every compute entry point returns NF_ERR_NO_DEVICE (-> RuntimeError) when no GPU is usable.

torch is imported first on purpose: PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7);
loading it before our library makes both share ONE HIP runtime in the process, so torch tensors'
data_ptr() can be handed to our kernels and our buffers to torch.distributed (RCCL).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below; see docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get('NEMOFLUX_AMD_LIB', os.path.join(_HERE, 'libnemoflux_amd.so'))  # override: sanitizer builds

if not os.path.exists(_SO):
    raise ImportError(
        f"{_SO} is missing: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C nemoflux_amd/csrc). "
        "nemoflux_amd has no CPU fallback.")

lib = ctypes.CDLL(_SO)

NF_F64, NF_F32 = 0, 1
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)
_h = ctypes.c_void_p  # opaque handle; passed as T** via byref


def _sig(name, argtypes, restype=ctypes.c_int):
    f = getattr(lib, name)
    f.argtypes = argtypes
    f.restype = restype
    return f


_sig('nf_last_error', [], ctypes.c_char_p)
_sig('nf_version', [])
_sig('nf_device_count', [c_int_p])
_sig('nf_set_device', [ctypes.c_int])
_sig('nf_device_name', [ctypes.c_char_p, ctypes.c_int])
_sig('nf_malloc', [c_void_pp, ctypes.c_size_t])
_sig('nf_free', [ctypes.c_void_p])
_sig('nf_host_alloc', [c_void_pp, ctypes.c_size_t])
_sig('nf_host_free', [ctypes.c_void_p])
_sig('nf_memcpy_h2d', [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t])
_sig('nf_memcpy_d2h', [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t])
_sig('nf_memset', [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t])
_sig('nf_synchronize', [])
_sig('nf_release_scratch', [])
_sig('nf_host_unshuffle', [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int])
_sig('nf_host_gather', [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int])
_sig('nf_tuning_set', [ctypes.c_char_p, ctypes.c_int])

_pp = ctypes.POINTER(_h)
_sig('mnt_grid_new', [_pp])
_sig('mnt_grid_del', [_pp])
_sig('mnt_grid_setPointsPtr', [_pp, c_double_p])
_sig('mnt_grid_build', [_pp, ctypes.c_int, ctypes.c_longlong])
_sig('mnt_grid_getNumberOfCells', [_pp, ctypes.POINTER(ctypes.c_size_t)])
_sig('mnt_grid_setRowLength', [_pp, ctypes.c_longlong])
_sig('mnt_grid_dump', [_pp, ctypes.c_char_p])
_sig('mnt_polylineintegral_new', [_pp])
_sig('mnt_polylineintegral_del', [_pp])
_sig('mnt_polylineintegral_setGrid', [_pp, _h])
_sig('mnt_polylineintegral_buildLocator', [_pp, ctypes.c_int, ctypes.c_double, ctypes.c_int])
_sig('mnt_polylineintegral_computeWeights', [_pp, ctypes.c_int, c_double_p, ctypes.c_int])
_sig('mnt_polylineintegral_setUnsupportedCells', [_pp, ctypes.c_int])
_sig('mnt_polylineintegral_setOverlappingCells', [_pp, ctypes.c_int])
_sig('mnt_polylineintegral_getNumberOfDroppedCrossings', [_pp, ctypes.POINTER(ctypes.c_size_t)])
_sig('mnt_polylineintegral_getIntegral', [_pp, c_double_p, ctypes.c_int, c_double_p])
_sig('mnt_polylineintegral_getIntegralDev', [_pp, ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p])
_sig('mnt_polylineintegral_getCoverage', [_pp, c_double_p])
_sig('mnt_polylineintegral_getNumberOfWeights', [_pp, ctypes.POINTER(ctypes.c_size_t)])
_sig('mnt_polylineintegral_getWeights', [_pp, c_int64_p, c_double_p, c_int_p])

_sig('mnt_vectorinterp_new', [_pp])
_sig('mnt_vectorinterp_del', [_pp])
_sig('mnt_vectorinterp_setGrid', [_pp, _h])
_sig('mnt_vectorinterp_buildLocator', [_pp, ctypes.c_int, ctypes.c_double, ctypes.c_int])
_sig('mnt_vectorinterp_findPoints', [_pp, ctypes.c_size_t, c_double_p, ctypes.c_double,
                                     ctypes.POINTER(ctypes.c_size_t)])
_sig('mnt_vectorinterp_getFaceVectors', [_pp, c_double_p, ctypes.c_int, c_double_p])
_sig('mnt_vectorinterp_getFaceVectorsDev', [_pp, ctypes.c_void_p, ctypes.c_int, c_double_p])
_sig('mnt_vectorinterp_getCells', [_pp, ctypes.POINTER(ctypes.c_longlong), c_double_p])

_sig('nf_field_new', [_pp])
_sig('nf_field_del', [_pp])
_sig('nf_field_set_stream', [_pp, ctypes.c_void_p])
_sig('nf_field_set_bounds', [_pp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                             ctypes.c_int])
_sig('nf_field_set_thickness', [_pp, c_double_p, ctypes.c_long])
_sig('nf_field_set_uv', [_pp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                         ctypes.c_double])
_sig('nf_field_set_missing_value', [_pp, ctypes.c_double])
_sig('nf_field_set_sverdrup', [_pp, ctypes.c_int])
_sig('nf_field_set_compact', [_pp, ctypes.c_int])
_sig('nf_field_set_slab_range', [_pp, ctypes.c_long, ctypes.c_long])
_sig('nf_field_add_transect', [_pp, c_double_p, ctypes.c_int, ctypes.c_int, c_int_p])
_sig('nf_field_set_unsupported_cells', [_pp, ctypes.c_int])
_sig('nf_field_set_overlapping_cells', [_pp, ctypes.c_int])
_sig('nf_field_num_dropped_crossings', [_pp, ctypes.POINTER(ctypes.c_size_t)])
_sig('nf_field_build_weights', [_pp, ctypes.c_int, ctypes.c_double])
_sig('nf_field_num_transects', [_pp, c_int_p])
_sig('nf_field_num_segments', [_pp, c_int_p])
_sig('nf_field_segment_offsets', [_pp, c_int_p])
_sig('nf_field_num_weights', [_pp, ctypes.POINTER(ctypes.c_size_t)])
_sig('nf_field_get_weights', [_pp, c_int64_p, c_double_p, c_int_p])
_sig('nf_field_get_coverage', [_pp, c_double_p])
_sig('nf_field_num_edge_weights', [_pp, ctypes.POINTER(ctypes.c_size_t)])
_sig('nf_field_get_edge_weights', [_pp, c_int_p, c_int_p, c_double_p])
_sig('nf_field_row_length', [_pp, c_int_p])
_sig('nf_field_compute_flux', [_pp, ctypes.c_long, c_double_p])
_sig('nf_field_compute_all_async', [_pp, ctypes.c_void_p])
_sig('nf_field_read_step', [_pp, c_double_p, c_double_p, c_double_p, c_double_p])
_sig('nf_field_reset_max', [_pp])
_sig('nf_field_get_arclengths', [_pp, c_double_p])
_sig('nf_field_get_points', [_pp, c_double_p])
_sig('nf_field_get_box', [_pp, c_double_p, c_double_p, c_double_p, c_double_p])
_sig('nf_field_device_ptr', [_pp, ctypes.c_int, c_void_pp])
_sig('nf_field_grid', [_pp, _pp])
_sig('nf_field_timing', [_pp, ctypes.c_int])
_sig('nf_field_timing_read', [_pp, ctypes.POINTER(ctypes.c_long), c_double_p])
_sig('nf_field_timing_split', [_pp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)])
_sig('nf_field_timing_k3', [_pp, ctypes.POINTER(ctypes.c_double)])
_sig('nf_rccl_unique_id', [ctypes.c_void_p])
_sig('nf_rccl_preflight', [c_int_p])
_sig('nf_rccl_comm_init', [c_void_pp, ctypes.c_int, ctypes.c_void_p, ctypes.c_int])
_sig('nf_rccl_comm_destroy', [ctypes.c_void_p])
_sig('nf_rccl_comm_info', [ctypes.c_void_p, c_int_p, c_int_p, c_int_p])
_sig('nf_rccl_library', [ctypes.c_char_p, ctypes.c_int])
_sig('nf_rows_allreduce', [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p])
c_ll_p = ctypes.POINTER(ctypes.c_longlong)
_sig('nf_inflater_new', [_pp])
_sig('nf_inflater_del', [_pp])
_sig('nf_inflater_share_scratch', [_pp, _pp])
_sig('nf_inflater_capacity', [c_int_p])
_sig('nf_inflater_upload', [_pp, ctypes.c_void_p, ctypes.c_size_t])
_sig('nf_inflater_upload_ranges', [_pp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_size_t])
_sig('nf_inflater_run', [_pp, ctypes.c_void_p, ctypes.c_size_t, c_ll_p, c_ll_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int,
                         ctypes.c_int, c_ll_p, c_ll_p, c_ll_p, ctypes.c_void_p, ctypes.c_void_p, c_int_p])
_sig('nf_datagen_bounds', [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long] + [ctypes.c_double] * 6 +
     [ctypes.c_int, ctypes.c_void_p])
_sig('nf_datagen_uv', [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_long] * 6 + [ctypes.c_double] * 6 +
     [ctypes.c_int, ctypes.c_int, ctypes.c_void_p])


def over_covered(cov, p0, p1):
    """The engine's own test for "this target segment is counted twice" (over_covered in csrc/nf_common.h, the same in the
    oracle): coverage > 1 + 1e-8 AND the excess as a length > 1e-9 max(1, |coordinates|) degrees -- rounding noise on target
    segments of ~1e-8 degrees that sit on a cell edge or node is not an overlap (round-5 advisor: the Python warnings used
    the first condition alone)."""
    if not cov > 1.0 + 1.e-8:
        return False
    dx, dy = float(p1[0]) - float(p0[0]), float(p1[1]) - float(p0[1])
    m = max(1.0, abs(float(p0[0])), abs(float(p0[1])), abs(float(p0[0]) + dx), abs(float(p0[1]) + dy))
    return (cov - 1.0) * (dx * dx + dy * dy) ** 0.5 > 1.e-9 * m


def _release_scratch_at_exit():
    """Give the pooled weight-build scratch (up to 4 GiB of HBM) back when the interpreter ends -- before the HIP runtime is
    torn down; nf_release_scratch frees every idle scratch of the process (round-5 verdict W8)."""
    try:
        lib.nf_release_scratch()
    except Exception:
        pass


import atexit  # noqa: E402
atexit.register(_release_scratch_at_exit)


class NemofluxError(RuntimeError):
    """A non-zero return code of the C ABI (the reference raises RuntimeError: field.py:135,154)."""


def check(rc):
    if rc != 0:
        msg = lib.nf_last_error().decode('utf-8', 'replace')
        raise NemofluxError(f'nemoflux_amd error {rc}: {msg}')


def device_count():
    n = ctypes.c_int(0)
    lib.nf_device_count(ctypes.byref(n))
    return n.value


def require_gpu():
    if device_count() <= 0:
        raise NemofluxError('nemoflux_amd: no usable AMD GPU; the engine has no CPU fallback')


def dptr(a):
    return a.ctypes.data_as(c_double_p)


class DeviceBuffer(object):
    """A block of HBM owned by the engine (nf_malloc/nf_free)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(lib.nf_malloc(ctypes.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, arr):
        import numpy
        a = numpy.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        check(lib.nf_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, shape, dtype):
        import numpy
        out = numpy.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib.nf_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if getattr(self, 'ptr', None):
            lib.nf_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceArray(object):
    """An HBM array described by (address, shape, dtype); `keepalive` owns the memory.  Used for the virtual
    global view of a rank's local slabs (nemoflux_amd.dist.virtual_base)."""

    def __init__(self, ptr, shape, dtype, keepalive=None):
        import numpy
        self.ptr, self.shape, self.dtype, self.keepalive = int(ptr), tuple(int(x) for x in shape), numpy.dtype(dtype), keepalive


def device_pointer(obj):
    """HBM address of a DeviceBuffer, a torch CUDA tensor or a raw int; None if obj lives on the host."""
    if isinstance(obj, (DeviceBuffer, DeviceArray)):
        return obj.ptr
    if isinstance(obj, int):
        return obj
    if isinstance(obj, torch.Tensor) and obj.is_cuda:
        assert obj.is_contiguous()
        return obj.data_ptr()
    return None
