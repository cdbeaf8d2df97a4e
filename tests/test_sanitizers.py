"""Host-side AddressSanitizer / UndefinedBehaviorSanitizer run of everything that executes on the
CPU: the oracle restatement and the CPU comparator behind oracle/asan_driver.cpp (every geometry
and edge size of test_oracle.py on exact-size heap buffers), and the harness's --cpu-only leg.
The reference has no such check (its validation layers are commented out, src/test/sort.hpp:121-133);
GPU-side sanitizers do not exist on this platform, the kernels are covered by the per-pass parity
tests instead.  `make asan` builds both binaries (test infrastructure, not part of the product)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT_MARKS = ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:", "SUMMARY: UndefinedBehaviorSanitizer")


def run_clean(cmd, timeout):
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "abort_on_error=0:detect_leaks=1:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["OMP_NUM_THREADS"] = env.get("OMP_NUM_THREADS", "4")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    text = out.stdout + out.stderr
    for mark in REPORT_MARKS:
        assert mark not in text, text[-4000:]
    assert out.returncode == 0, text[-4000:]
    return out.stdout


def build():
    made = subprocess.run(["make", "-C", ROOT, "asan"], capture_output=True, text=True, timeout=600)
    assert made.returncode == 0, made.stdout + made.stderr


def test_oracle_and_cpu_comparator_under_sanitizers():
    build()
    out = run_clean([os.path.join(ROOT, "oracle", "asan_driver")], 600)
    assert "0 failed" in out


def test_harness_cpu_leg_under_sanitizers():
    build()
    exe = os.path.join(ROOT, "harness", "sort_test_asan")
    for args in (["--log2n", "16", "--dist", "uniform"], ["--n", "100003", "--dist", "shuffle"],
                 ["--n", "1", "--dist", "hash"]):
        out = run_clean([exe, "--cpu-only", "--iters", "1"] + args, 300)
        assert "Sorting Finished: PASS" in out
