"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/radx_hip.h declares, reports argument errors, and the harness's CPU path (the no-GPU benchmark
config 1: 1 Mi uniform uint32 keys, parallel std::sort, no GPU) runs and passes."""
import ctypes
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "radx_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(radx_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import radx_amd
    names = declared_symbols()
    assert len(names) >= 15
    raw = ctypes.CDLL(radx_amd.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), "libradx_hip.so does not export %s" % name
        assert name in radx_amd.capi.SIGNATURES, "radx_amd.capi has no signature for %s" % name
    assert sorted(radx_amd.capi.SIGNATURES) == names


def test_version_and_scratch_size_need_no_gpu():
    import radx_amd
    assert radx_amd.lib.radx_hip_version() >= 400
    small = radx_amd.lib.radx_hip_scratch_bytes(1 << 20, 4)
    big = radx_amd.lib.radx_hip_scratch_bytes(1 << 30, 4)
    assert 0 < small < big < (1 << 30) // 8       # tables are a fraction of a percent of the keys
    assert radx_amd.lib.radx_hip_scratch_bytes(1 << 20, 3) == 0
    assert radx_amd.lib.radx_hip_scratch_bytes(1 << 33, 4) == 0


def test_argument_errors_without_gpu():
    import radx_amd
    h = ctypes.c_void_p(None)
    assert radx_amd.lib.radx_hip_create(0, 1000, 5, 0, ctypes.byref(h)) == 1      # hipErrorInvalidValue
    assert b"key_bytes" in radx_amd.lib.radx_hip_error_string()
    assert radx_amd.lib.radx_hip_create(0, 1000, 4, 2, ctypes.byref(h)) == 1
    assert radx_amd.lib.radx_hip_create(0, 1 << 32, 4, 0, ctypes.byref(h)) == 1
    assert radx_amd.lib.radx_hip_sort(None, None, None, None, None, 10, None) == 1
    assert radx_amd.lib.radx_hip_destroy(None) == 0


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    import radx_amd.capi as capi
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        capi._load()
    importlib.reload(capi)


def test_harness_cpu_path_config1():
    exe = os.path.join(ROOT, "harness", "sort_test")
    assert os.path.exists(exe), "build the harness with `make`"
    for dist in ("uniform", "shuffle"):
        out = subprocess.run([exe, "--cpu-only", "--log2n", "20", "--dist", dist, "--iters", "1"],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "Sorting Finished: PASS" in out.stdout
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n"] == 1 << 20 and line["pass"] is True and line["threads"] >= 1


def test_reference_call_sequence_compiles_against_these_headers():
    """harness/reference_sequence.cpp is the reference harness's set-up and sort command
    (src/test/sort.cpp:304-338,411) with only the handle types swapped, plus the overloads and
    public members the reference declares and its own harness does not use.  Compiling it (no
    GPU needed) is the test; tests/test_gpu_harness.py runs it."""
    made = subprocess.run(["make", "-C", ROOT, "-W", "harness/reference_sequence.cpp", "harness/reference_sequence"], capture_output=True, text=True, timeout=600)
    assert made.returncode == 0, made.stdout + made.stderr
    assert "warning" not in made.stderr, made.stderr
    assert os.access(os.path.join(ROOT, "harness", "reference_sequence"), os.X_OK)


def test_header_is_plain_c():
    """The boundary is a C ABI: include/radx_hip.h compiles as C99 (and as C++) on its own, with
    nothing but <stddef.h> / <stdint.h> behind it."""
    header = os.path.join(ROOT, "include", "radx_hip.h")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", header],
                ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", header]):
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
