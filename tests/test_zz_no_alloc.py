"""radx_hip_sort_noalloc — the entry point that allocates nothing (tables in a device array that
is part of the library) — against radx_hip_sort on the same inputs, through the C ABI, bit for bit.

Two calls in a row share the array; the second uses a smaller input, so anything the first call
left in the tables that the second wrongly relied on (or failed to overwrite) would show.
Collected last on purpose: the whole suite runs before it under `pytest -x`.
"""
import ctypes

import numpy as np
import pytest

from datasets import make_keys

pytestmark = pytest.mark.gpu


def to_dev(a, device):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a.view(np.int32 if a.itemsize == 4 else np.int64))).to(device)


def both_sorts(device, keys, vals=None):
    """(radx_hip_sort result, radx_hip_sort_noalloc result) as host arrays (keys, vals-or-None)."""
    import torch
    import radx_amd
    n, kb = keys.size, keys.itemsize
    out = []
    for noalloc in (False, True):
        k = to_dev(keys, device)
        swap = torch.empty_like(k)
        v = to_dev(vals, device) if vals is not None else None
        vswap = torch.empty_like(v) if v is not None else None
        if noalloc:
            radx_amd.sort_noalloc(k, swap, v, vswap)
        else:
            s = radx_amd.Sorter(n, kb, 4 if vals is not None else 0)
            s.sort(k, swap, v, vswap)
            torch.cuda.synchronize()
            s.close()
        torch.cuda.synchronize()
        out.append((k.cpu().numpy().view(keys.dtype), None if v is None else v.cpu().numpy().view(np.uint32)))
    return out


@pytest.mark.parametrize("key_bytes,with_vals", [(4, False), (8, True), (4, True), (8, False)])
def test_two_calls_in_a_row_match_the_sorter(device, key_bytes, with_vals):
    tile = 4096 if key_bytes == 4 else 2048
    # first: more than one scan group (> 256 tiles) and a ragged last tile; second: much smaller, ragged
    for n in (300 * tile + 77, 9 * tile + 5):
        keys = make_keys("uniform", n, key_bytes, seed=n)
        if with_vals:  # duplicates, so that the payload order checks stability
            keys &= np.array(0xFFFF00FF if key_bytes == 4 else 0xFFFF0000000000FF, dtype=keys.dtype)
        vals = np.arange(n, dtype=np.uint32) if with_vals else None
        (want_k, want_v), (got_k, got_v) = both_sorts(device, keys, vals)
        assert np.array_equal(want_k, np.sort(keys, kind="stable"))
        assert np.array_equal(got_k, want_k)
        if with_vals:
            assert np.array_equal(got_v, want_v)


def test_skewed_then_tiny(device):
    for n, dist in ((70 * 4096 + 1, "all_equal"), (3, "uniform"), (1, "uniform")):
        keys = make_keys(dist, n, 4, seed=5)
        (want_k, _), (got_k, _) = both_sorts(device, keys)
        assert np.array_equal(got_k, want_k)


def test_limit_is_enforced_without_touching_the_buffers(device):
    import torch
    import radx_amd
    from radx_amd import capi
    k = torch.zeros(16, dtype=torch.int32, device=device)
    swap = torch.zeros_like(k)
    rc = capi.lib.radx_hip_sort_noalloc(0, ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(swap.data_ptr()), None, None,
                                        capi.NOALLOC_MAX_N + 1, 4, 0, None)
    assert rc == 1  # hipErrorInvalidValue
    assert b"at most" in capi.lib.radx_hip_error_string()
    with pytest.raises(radx_amd.RadxError):
        capi.check(capi.lib.radx_hip_sort_noalloc(0, ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(swap.data_ptr()),
                                                  None, None, 16, 3, 0, None))
    # n == 0 is a no-op
    capi.check(capi.lib.radx_hip_sort_noalloc(0, None, None, None, None, 0, 4, 0, None))


def test_at_the_limit(device):
    import torch
    import radx_amd
    from radx_amd import capi
    n = capi.NOALLOC_MAX_N
    k = torch.empty(n, dtype=torch.int64, device=device)  # 8-byte keys: the layout with the most table rows
    radx_amd.generate_u64(k, None, seed=3)
    before = radx_amd.checksum(k)
    ref = k.clone()
    swap = torch.empty_like(k)
    radx_amd.sort_noalloc(k, swap)
    s = radx_amd.Sorter(n, 8, 0)
    s.sort(ref, swap)
    torch.cuda.synchronize()
    s.close()
    assert radx_amd.check_sorted(k) == 0 and radx_amd.checksum(k) == before
    assert bool(torch.equal(k, ref))


def test_two_calls_captured_in_one_graph(device):
    """Callers under HIP graph capture are who the entry point is for: two calls (the second on a
    smaller input, a different key width and with a payload) captured into one graph after a first
    call outside capture, replayed on new buffer contents and compared with numpy."""
    import torch
    import radx_amd
    n1, n2 = 70 * 4096 + 9, 11 * 2048 + 3
    k1 = torch.zeros(n1, dtype=torch.int32, device=device)
    s1 = torch.empty_like(k1)
    k2 = torch.zeros(n2, dtype=torch.int64, device=device)
    s2 = torch.empty_like(k2)
    v2 = torch.zeros(n2, dtype=torch.int32, device=device)
    w2 = torch.empty_like(v2)
    radx_amd.sort_noalloc(k1, s1)          # first call on this GPU: outside capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        radx_amd.sort_noalloc(k1, s1)
        radx_amd.sort_noalloc(k2, s2, v2, w2)
    for seed in (1, 2):
        a = make_keys("uniform", n1, 4, seed=seed)
        b = make_keys("few_distinct", n2, 8, seed=seed)
        k1.copy_(to_dev(a, device))
        k2.copy_(to_dev(b, device))
        v2.copy_(torch.arange(n2, dtype=torch.int32, device=device))
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(k1.cpu().numpy().view(np.uint32), np.sort(a))
        order = np.argsort(b, kind="stable")
        assert np.array_equal(k2.cpu().numpy().view(np.uint64), b[order])
        assert np.array_equal(v2.cpu().numpy().view(np.uint32), order.astype(np.uint32))


def test_wrapper_rejects_mismatched_tensors(device):
    import torch
    import radx_amd
    k = torch.zeros(64, dtype=torch.int32, device=device)
    with pytest.raises(ValueError):
        radx_amd.sort_noalloc(k, torch.zeros(64, dtype=torch.int64, device=device))            # swap dtype
    with pytest.raises(ValueError):
        radx_amd.sort_noalloc(k, torch.zeros_like(k), torch.zeros(64, dtype=torch.int64, device=device),
                              torch.zeros(64, dtype=torch.int64, device=device))               # 8-byte payload
    with pytest.raises(ValueError):
        radx_amd.sort_noalloc(k, torch.zeros_like(k), torch.zeros_like(k), None)               # half a payload
