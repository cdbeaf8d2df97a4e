"""Thin Python driver over the C ABI: hands torch device buffers and the current HIP stream
to libradx_hip.so.  torch is used for device memory and streams only; every byte of sorting
work happens in the HIP kernels behind include/radx_hip.h.

The call sequence mirrors the reference's host API (include/radx/radx_internal.hpp:104-148):
a Sorter owns the scratch tables (Sort<Radix>::initialize -> Radix::createInternalMemory) and
sort() enqueues the pass loop on a stream (Sort::command -> Radix::command); the sorted
result is left in `keys`, `swap` is clobbered.
"""
import ctypes

import numpy as np

from . import capi


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(None)


def _stream(stream, device):
    if stream is None:
        import torch
        stream = torch.cuda.current_stream(device)
    return ctypes.c_void_p(getattr(stream, "cuda_stream", stream))


class Sorter:
    """One radx_hip_sorter handle.  key_bytes 4|8, value_bytes 0|4."""

    def __init__(self, max_n, key_bytes=4, value_bytes=0, device=0):
        self.max_n = int(max_n)
        self.key_bytes = key_bytes
        self.value_bytes = value_bytes
        self.device = device
        self._h = ctypes.c_void_p(None)
        capi.check(capi.lib.radx_hip_create(device, self.max_n, key_bytes, value_bytes,
                                            ctypes.byref(self._h)))

    def close(self):
        if self._h:
            capi.check(capi.lib.radx_hip_destroy(self._h))
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, t, elem, n, name):
        if t is None:
            return
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("%s must be a contiguous device tensor" % name)
        if t.element_size() != elem:
            raise ValueError("%s has %d-byte elements, sorter expects %d" % (name, t.element_size(), elem))
        if t.numel() < n:
            raise ValueError("%s holds %d elements, need %d" % (name, t.numel(), n))

    def _args(self, keys, swap, vals, vals_swap, n):
        n = keys.numel() if n is None else int(n)
        self._check(keys, self.key_bytes, n, "keys")
        self._check(swap, self.key_bytes, n, "swap")
        self._check(vals, 4, n, "vals")
        self._check(vals_swap, 4, n, "vals_swap")
        return n

    def sort(self, keys, swap, vals=None, vals_swap=None, n=None, stream=None):
        n = self._args(keys, swap, vals, vals_swap, n)
        capi.check(capi.lib.radx_hip_sort(self._h, _ptr(keys), _ptr(swap), _ptr(vals), _ptr(vals_swap),
                                          n, _stream(stream, self.device)))

    def sort_bits(self, keys, swap, begin_bit, end_bit, vals=None, vals_swap=None, n=None, stream=None):
        n = self._args(keys, swap, vals, vals_swap, n)
        capi.check(capi.lib.radx_hip_sort_bits(self._h, _ptr(keys), _ptr(swap), _ptr(vals),
                                               _ptr(vals_swap), n, begin_bit, end_bit,
                                               _stream(stream, self.device)))

    def sort_bits_into(self, keys, swap, keys_out, begin_bit, end_bit, vals=None, vals_swap=None,
                       vals_out=None, n=None, stream=None):
        """sort_bits with exactly ceil(bits/4) passes; the last one stores to keys_out (any key-aligned
        address that does not overlap the last pass's source)."""
        n = self._args(keys, swap, vals, vals_swap, n)
        self._check(keys_out, self.key_bytes, n, "keys_out")
        self._check(vals_out, 4, n, "vals_out")
        capi.check(capi.lib.radx_hip_sort_bits_into(self._h, _ptr(keys), _ptr(swap), _ptr(vals),
                                                    _ptr(vals_swap), n, begin_bit, end_bit, _ptr(keys_out),
                                                    _ptr(vals_out), _stream(stream, self.device)))

    def msd_split(self, keys_in, keys_out, log2_ranks, vals_in=None, vals_out=None, n=None, stream=None):
        """Stable partition on the top 4-bit digit, buckets in (sub-bucket, destination rank) order."""
        n = self._args(keys_in, keys_out, vals_in, vals_out, n)
        capi.check(capi.lib.radx_hip_msd_split(self._h, _ptr(keys_in), _ptr(keys_out), _ptr(vals_in),
                                               _ptr(vals_out), n, log2_ranks, _stream(stream, self.device)))

    def run_pass(self, keys_in, keys_out, bit, num_bits=4, vals_in=None, vals_out=None, n=None, stream=None):
        n = self._args(keys_in, keys_out, vals_in, vals_out, n)
        capi.check(capi.lib.radx_hip_pass(self._h, _ptr(keys_in), _ptr(keys_out), _ptr(vals_in),
                                          _ptr(vals_out), n, bit, num_bits, _stream(stream, self.device)))

    def run_pass_fused(self, keys_in, keys_out, bit, num_bits, next_bit, next_num_bits, use_fused_counts,
                       vals_in=None, vals_out=None, n=None, stream=None):
        """One pass as sort() chains them: optionally take the histogram the previous fused pass
        accumulated, and accumulate the next pass's histogram (next_bit < 0: do not)."""
        n = self._args(keys_in, keys_out, vals_in, vals_out, n)
        capi.check(capi.lib.radx_hip_pass_fused(self._h, _ptr(keys_in), _ptr(keys_out), _ptr(vals_in),
                                                _ptr(vals_out), n, bit, num_bits, next_bit, next_num_bits,
                                                1 if use_fused_counts else 0, _stream(stream, self.device)))

    def set_key_order(self, kind="unsigned", descending=False):
        """kind: 'unsigned' | 'signed' | 'float' (IEEE bits of the key width); stable either way."""
        capi.check(capi.lib.radx_hip_set_key_order(self._h, {"unsigned": 0, "signed": 1, "float": 2}[kind],
                                                   1 if descending else 0))

    def tune(self, fuse_histogram=True, scan_kernels=False, publish="all"):
        """fuse_histogram: scatter passes produce the next pass's histogram.  scan_kernels: run the
        scan kernels even for inputs small enough for the scan-free passes.  publish (tests):
        "none" - scan-free passes publish nothing, every tile workgroup sums the table itself;
        "bases" - no group sums are published, the scanning workgroup sums the table itself."""
        flags = (1 if fuse_histogram else 0) | (2 if scan_kernels else 0) | {"all": 0, "none": 4, "bases": 8}[publish]
        capi.check(capi.lib.radx_hip_tune(self._h, flags))

    def geometry(self, n):
        """(keys per tile, number of tiles) for n elements; one workgroup per tile."""
        tile, tiles = ctypes.c_uint32(0), ctypes.c_uint32(0)
        capi.check(capi.lib.radx_hip_geometry(self._h, int(n), ctypes.byref(tile), ctypes.byref(tiles)))
        return tile.value, tiles.value

    def tables(self, n, stream=None):
        """(counts, offsets) of the last pass as uint32 arrays of shape [num_tiles, 16]."""
        _, tiles = self.geometry(n)
        counts = np.empty((tiles, 16), dtype=np.uint32)
        offsets = np.empty((tiles, 16), dtype=np.uint32)
        capi.check(capi.lib.radx_hip_copy_tables(self._h, int(n), counts.ctypes.data_as(ctypes.c_void_p),
                                                 offsets.ctypes.data_as(ctypes.c_void_p),
                                                 _stream(stream, self.device)))
        return counts, offsets

    def digit_starts_into(self, dst, stream=None):
        """The 17 run starts of the last pass copied into dst[:17] (a 4-byte-element tensor, device
        or pinned host) in stream order; no synchronisation."""
        if dst.element_size() != 4 or dst.numel() < 17 or not dst.is_contiguous():
            raise ValueError("dst must be a contiguous tensor of at least 17 4-byte elements")
        capi.check(capi.lib.radx_hip_digit_starts_async(self._h, _ptr(dst), _stream(stream, self.device)))

    def device_tables(self, n):
        """Device addresses (counts, offsets, group_bases or None, tiles_per_group) of the last pass's tables."""
        c, o, gb = ctypes.c_void_p(None), ctypes.c_void_p(None), ctypes.c_void_p(None)
        t = ctypes.c_uint32(0)
        capi.check(capi.lib.radx_hip_device_tables(self._h, int(n), ctypes.byref(c), ctypes.byref(o), ctypes.byref(gb),
                                                   ctypes.byref(t)))
        return c.value, o.value, gb.value, t.value

    def digit_starts(self, stream=None):
        out = (ctypes.c_uint32 * 17)()
        capi.check(capi.lib.radx_hip_digit_starts(self._h, out, _stream(stream, self.device)))
        return np.array(out, dtype=np.uint32)

    def profile(self, enable=True):
        capi.check(capi.lib.radx_hip_profile(self._h, 1 if enable else 0))

    def kernel_times(self):
        """{'hist'|'scan'|'scatter': (total_ms, launches)} since the last call; synchronises."""
        ms = (ctypes.c_double * 3)()
        cnt = (ctypes.c_uint64 * 3)()
        capi.check(capi.lib.radx_hip_kernel_times(self._h, ms, cnt))
        return {k: (ms[i], int(cnt[i])) for i, k in enumerate(("hist", "scan", "scatter"))}


def sort_noalloc(keys, swap, vals=None, vals_swap=None, n=None, stream=None):
    """Full sort through radx_hip_sort_noalloc: no sorter handle, no allocation, tables in the
    library's resident device array; n <= capi.NOALLOC_MAX_N.  Calls on one GPU must not overlap."""
    n = keys.numel() if n is None else int(n)
    for t, name in ((keys, "keys"), (swap, "swap"), (vals, "vals"), (vals_swap, "vals_swap")):
        if t is not None and (not t.is_cuda or not t.is_contiguous() or t.numel() < n):
            raise ValueError("%s must be a contiguous device tensor of at least n elements" % name)
        if t is not None and t.device != keys.device:
            raise ValueError("%s is on %s, keys on %s" % (name, t.device, keys.device))
    if keys.element_size() not in (4, 8) or swap.dtype != keys.dtype:
        raise ValueError("keys must have 4- or 8-byte elements and swap the same dtype")
    if (vals is None) != (vals_swap is None):
        raise ValueError("vals and vals_swap go together")
    if vals is not None and (vals.element_size() != 4 or vals_swap.dtype != vals.dtype):
        raise ValueError("vals must have 4-byte elements and vals_swap the same dtype")
    device = keys.device.index if keys.device.index is not None else 0
    capi.check(capi.lib.radx_hip_sort_noalloc(device, _ptr(keys), _ptr(swap), _ptr(vals), _ptr(vals_swap), n,
                                              keys.element_size(), 0 if vals is None else 4,
                                              _stream(stream, keys.device)))


def generate_u32(keys, seed, n=None, stream=None):
    n = keys.numel() if n is None else int(n)
    capi.check(capi.lib.radx_hip_generate_u32(_ptr(keys), n, seed & 0xFFFFFFFF,
                                              _stream(stream, keys.device)))


def generate_u64(keys, vals, seed, n=None, stream=None):
    n = keys.numel() if n is None else int(n)
    capi.check(capi.lib.radx_hip_generate_u64(_ptr(keys), _ptr(vals), n, seed & 0xFFFFFFFF,
                                              _stream(stream, keys.device)))


def check_sorted(keys, vals=None, n=None, stream=None):
    """Number of adjacent order (and, with vals, stability) violations; synchronises."""
    n = keys.numel() if n is None else int(n)
    bad = ctypes.c_uint64(0)
    capi.check(capi.lib.radx_hip_check_sorted(_ptr(keys), _ptr(vals), n, keys.element_size(),
                                              ctypes.byref(bad), _stream(stream, keys.device)))
    return bad.value


def checksum(keys, n=None, stream=None):
    """(sum, xor, mixed sum) of the keys, each mod 2^64; synchronises."""
    n = keys.numel() if n is None else int(n)
    out = (ctypes.c_uint64 * 3)()
    capi.check(capi.lib.radx_hip_checksum(_ptr(keys), n, keys.element_size(), out,
                                          _stream(stream, keys.device)))
    return tuple(int(v) for v in out)


def indiction(ptrs, cache=None, src=None, n=None, stream=None):
    """ptrs[i] = i and, when given, cache[i] = src[i] (rows = src.shape[1:])."""
    n = ptrs.numel() if n is None else int(n)
    row = 4 if src is None else src[0].numel() * src.element_size() if src.dim() > 1 else src.element_size()
    capi.check(capi.lib.radx_hip_indiction(_ptr(ptrs), _ptr(cache), _ptr(src), n, row, _stream(stream, ptrs.device)))


def permutation(dst, cache, ptrs, n=None, stream=None):
    """dst[i] = cache[ptrs[i]] (rows = dst.shape[1:])."""
    n = ptrs.numel() if n is None else int(n)
    row = dst[0].numel() * dst.element_size() if dst.dim() > 1 else dst.element_size()
    capi.check(capi.lib.radx_hip_permutation(_ptr(dst), _ptr(cache), _ptr(ptrs), n, row, _stream(stream, dst.device)))
