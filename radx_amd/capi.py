"""ctypes binding of the C ABI in include/radx_hip.h.

The library is the product: there is no CPU fallback.  Importing this module loads
radx_amd/lib/libradx_hip.so (built in-tree by `make` / __graft_entry__.build()) and fails
loudly when it is missing.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libradx_hip.so")

# Every symbol include/radx_hip.h declares, with its ctypes signature.
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
SIGNATURES = {
    "radx_hip_version": (ctypes.c_int, []),
    "radx_hip_error_string": (ctypes.c_char_p, []),
    "radx_hip_scratch_bytes": (ctypes.c_size_t, [ctypes.c_uint64, ctypes.c_int]),
    "radx_hip_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(_vp)]),
    "radx_hip_destroy": (ctypes.c_int, [_vp]),
    "radx_hip_sort": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp]),
    "radx_hip_sort_noalloc": (ctypes.c_int, [ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_int,
                                             ctypes.c_int, _vp]),
    "radx_hip_sort_bits": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_int,
                                          ctypes.c_int, _vp]),
    "radx_hip_sort_bits_into": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_int,
                                               ctypes.c_int, _vp, _vp, _vp]),
    "radx_hip_msd_split": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_int, _vp]),
    "radx_hip_pass": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_int,
                                     ctypes.c_int, _vp]),
    "radx_hip_pass_fused": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]),
    "radx_hip_tune": (ctypes.c_int, [_vp, ctypes.c_int]),
    "radx_hip_set_key_order": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "radx_hip_geometry": (ctypes.c_int, [_vp, ctypes.c_uint64, _u32p, _u32p]),
    "radx_hip_copy_tables": (ctypes.c_int, [_vp, ctypes.c_uint64, _vp, _vp, _vp]),
    "radx_hip_digit_starts": (ctypes.c_int, [_vp, _u32p, _vp]),
    "radx_hip_digit_starts_async": (ctypes.c_int, [_vp, _vp, _vp]),
    "radx_hip_device_tables": (ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                              ctypes.POINTER(_vp), _u32p]),
    "radx_hip_profile": (ctypes.c_int, [_vp, ctypes.c_int]),
    "radx_hip_kernel_times": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double), _u64p]),
    "radx_hip_indiction": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint64, ctypes.c_size_t, _vp]),
    "radx_hip_permutation": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint64, ctypes.c_size_t, _vp]),
    "radx_hip_generate_u32": (ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_uint32, _vp]),
    "radx_hip_generate_u64": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64, ctypes.c_uint32, _vp]),
    "radx_hip_check_sorted": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64, ctypes.c_int, _u64p, _vp]),
    "radx_hip_checksum": (ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_int, _u64p, _vp]),
}


NOALLOC_MAX_N = 1 << 26  # RADX_HIP_NOALLOC_MAX_N


class RadxError(RuntimeError):
    """A C-ABI call returned a non-zero hipError_t."""

    def __init__(self, code, text):
        super().__init__("radx_hip error %d: %s" % (code, text))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make` (or __graft_entry__.build()); "
            "radx_amd has no fallback path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


def check(code):
    if code != 0:
        raise RadxError(code, lib.radx_hip_error_string().decode("utf-8", "replace"))
