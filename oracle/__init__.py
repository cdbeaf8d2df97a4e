"""ctypes binding of the CPU oracle (oracle/libradx_oracle.so, oracle/libradx_cpusort.so).

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load
these libraries.  Both are built by `make` from oracle/radx_oracle.c and oracle/cpu_sort.cpp.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.path.join(_HERE, "libradx_oracle.so")
CPUSORT_PATH = os.path.join(_HERE, "libradx_cpusort.so")


class Config(ctypes.Structure):
    """radx_ref_config: launch geometry of the reference kernels."""
    _fields_ = [("bits", ctypes.c_uint32), ("grid", ctypes.c_uint32),
                ("wave", ctypes.c_uint32), ("vec", ctypes.c_uint32)]


class Blocks(ctypes.Structure):
    _fields_ = [("count", ctypes.c_uint32), ("limit", ctypes.c_uint32), ("offset", ctypes.c_uint32)]


_vp = ctypes.c_void_p
_cfgp = ctypes.POINTER(Config)

ref = ctypes.CDLL(ORACLE_PATH)
ref.radx_ref_blocks_info.restype = Blocks
ref.radx_ref_blocks_info.argtypes = [_cfgp, ctypes.c_uint32, ctypes.c_uint32]
ref.radx_ref_digit.restype = ctypes.c_uint32
ref.radx_ref_digit.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]
ref.radx_ref_counting.restype = None
ref.radx_ref_counting.argtypes = [_cfgp, _vp, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, _vp]
ref.radx_ref_partition.restype = None
ref.radx_ref_partition.argtypes = [_cfgp, _vp, _vp]
ref.radx_ref_scattering.restype = None
ref.radx_ref_scattering.argtypes = [_cfgp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_uint32,
                                    ctypes.c_uint32, _vp]
ref.radx_ref_sort.restype = ctypes.c_int
ref.radx_ref_sort.argtypes = [_cfgp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32,
                              _vp, _vp]
ref.radx_ref_indiction.restype = None
ref.radx_ref_indiction.argtypes = [_vp, _vp, _vp, ctypes.c_uint64, ctypes.c_size_t]
ref.radx_ref_permutation.restype = None
ref.radx_ref_permutation.argtypes = [_vp, _vp, _vp, ctypes.c_uint64, ctypes.c_size_t]
ref.radx_ref_hash.restype = ctypes.c_uint32
ref.radx_ref_hash.argtypes = [ctypes.c_uint32]
ref.radx_ref_generate_u32.restype = None
ref.radx_ref_generate_u32.argtypes = [_vp, ctypes.c_uint64, ctypes.c_uint32]
ref.radx_ref_generate_u64.restype = None
ref.radx_ref_generate_u64.argtypes = [_vp, _vp, ctypes.c_uint64, ctypes.c_uint32]
ref.radx_ref_checksum.restype = None
ref.radx_ref_checksum.argtypes = [_vp, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
ref.radx_ref_violations.restype = ctypes.c_uint64
ref.radx_ref_violations.argtypes = [_vp, _vp, ctypes.c_uint64, ctypes.c_int]

cpu = ctypes.CDLL(CPUSORT_PATH)
cpu.radx_cpu_pstl_backend.restype = ctypes.c_char_p
cpu.radx_cpu_threads.restype = ctypes.c_int
for _name in ("radx_cpu_stable_sort_u32", "radx_cpu_stable_sort_u64",
              "radx_cpu_parallel_sort_u32", "radx_cpu_parallel_sort_u64"):
    getattr(cpu, _name).restype = None
    getattr(cpu, _name).argtypes = [_vp, ctypes.c_uint64]
for _name in ("radx_cpu_stable_sort_kv_u64", "radx_cpu_stable_sort_kv_u32"):
    getattr(cpu, _name).restype = None
    getattr(cpu, _name).argtypes = [_vp, _vp, ctypes.c_uint64]


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None else None


# Geometries.  HEAD: what the reference ships for AMD (2-bit digits, 108 workgroups, 64x4 tiles).
def head_config():
    return Config(bits=2, grid=108, wave=64, vec=4)


def engine_config(n, key_bytes=4, bits=4):
    """The HIP engine's decomposition: get_blocks_info with one workgroup per 16 KiB tile."""
    tile = 4096 if key_bytes == 4 else 2048
    return Config(bits=bits, grid=max(1, -(-n // tile)), wave=64, vec=tile // 64)


def key_dtype(key_bytes):
    return np.uint32 if key_bytes == 4 else np.uint64


def counting(cfg, keys, shift):
    counts = np.zeros((cfg.grid, 1 << cfg.bits), dtype=np.uint32)
    ref.radx_ref_counting(ctypes.byref(cfg), _p(keys), keys.itemsize, keys.size, shift, _p(counts))
    return counts


def partition(cfg, counts):
    parts = np.zeros_like(counts)
    ref.radx_ref_partition(ctypes.byref(cfg), _p(np.ascontiguousarray(counts)), _p(parts))
    return parts


def scattering(cfg, keys, shift, partitions, vals=None):
    out = np.zeros_like(keys)
    vout = np.zeros_like(vals) if vals is not None else None
    ref.radx_ref_scattering(ctypes.byref(cfg), _p(keys), _p(out), _p(vals), _p(vout), keys.itemsize,
                            keys.size, shift, _p(np.ascontiguousarray(partitions)))
    return (out, vout) if vals is not None else out


def one_pass(cfg, keys, shift, vals=None):
    """(counts, partitions, keys_out[, vals_out]) of one counting/partition/scattering pass."""
    c = counting(cfg, keys, shift)
    p = partition(cfg, c)
    out = scattering(cfg, keys, shift, p, vals)
    return (c, p) + (out if vals is not None else (out,))


def sort(cfg, keys, vals=None, passes=0, log=False):
    """Full pass loop on copies.  Returns sorted keys (and vals); with log=True also the per-pass tables."""
    k = np.array(keys, copy=True)
    swap = np.zeros_like(k)
    v = np.array(vals, copy=True) if vals is not None else None
    vswap = np.zeros_like(v) if v is not None else None
    npass = passes or (k.itemsize * 8) // cfg.bits
    clog = np.zeros((npass, cfg.grid, 1 << cfg.bits), dtype=np.uint32) if log else None
    plog = np.zeros_like(clog) if log else None
    where = ref.radx_ref_sort(ctypes.byref(cfg), _p(k), _p(swap), _p(v), _p(vswap), k.itemsize, k.size,
                              passes, _p(clog), _p(plog))
    rk, rv = (k, v) if where == 0 else (swap, vswap)
    res = (rk,) if vals is None else (rk, rv)
    if log:
        res = res + (clog, plog)
    return res if len(res) > 1 else res[0]


def generate_u32(n, seed):
    k = np.empty(n, dtype=np.uint32)
    ref.radx_ref_generate_u32(_p(k), n, seed & 0xFFFFFFFF)
    return k


def generate_u64(n, seed):
    k = np.empty(n, dtype=np.uint64)
    v = np.empty(n, dtype=np.uint32)
    ref.radx_ref_generate_u64(_p(k), _p(v), n, seed & 0xFFFFFFFF)
    return k, v


def checksum(keys):
    out = (ctypes.c_uint64 * 3)()
    ref.radx_ref_checksum(_p(keys), keys.size, keys.itemsize, out)
    return tuple(int(x) for x in out)


def violations(keys, vals=None):
    return int(ref.radx_ref_violations(_p(keys), _p(vals), keys.size, keys.itemsize))


def stable_sort(keys):
    """The reference's CPU comparator: std::stable_sort(std::execution::par, ...) (src/test/sort.cpp:218-219)."""
    k = np.array(keys, copy=True)
    fn = cpu.radx_cpu_stable_sort_u32 if k.itemsize == 4 else cpu.radx_cpu_stable_sort_u64
    fn(_p(k), k.size)
    return k


def parallel_sort(keys):
    k = np.array(keys, copy=True)
    fn = cpu.radx_cpu_parallel_sort_u32 if k.itemsize == 4 else cpu.radx_cpu_parallel_sort_u64
    fn(_p(k), k.size)
    return k


def stable_sort_kv(keys, vals):
    k = np.array(keys, copy=True)
    v = np.array(vals, copy=True)
    fn = cpu.radx_cpu_stable_sort_kv_u32 if k.itemsize == 4 else cpu.radx_cpu_stable_sort_kv_u64
    fn(_p(k), _p(v), k.size)
    return k, v


def indiction(src):
    """(ptrs, cache) of indiction.comp on rows src[i]."""
    n = src.shape[0]
    ptrs = np.empty(n, dtype=np.uint32)
    cache = np.empty_like(src)
    ref.radx_ref_indiction(_p(ptrs), _p(cache), _p(np.ascontiguousarray(src)), n, src.strides[0] if n else 4)
    return ptrs, cache


def permutation(cache, ptrs):
    out = np.empty_like(cache)
    ref.radx_ref_permutation(_p(out), _p(np.ascontiguousarray(cache)), _p(np.ascontiguousarray(ptrs)), ptrs.size,
                             cache.strides[0] if ptrs.size else 4)
    return out
