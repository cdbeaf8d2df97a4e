"""The C++ harness (reference call sequence through include/radx/*.hpp) on the GPU."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "harness", "sort_test")


def run(*args):
    out = subprocess.run([EXE] + list(args), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Sorting Finished: PASS" in out.stdout
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("dist", ["shuffle", "uniform", "hash"])
def test_harness_u32_reference_size(device, dist):
    line = run("--log2n", "23", "--dist", dist, "--iters", "2")   # 2^23: src/test/sort.hpp:184
    assert line["pass"] is True and line["n"] == 1 << 23


@pytest.mark.parametrize("n", ["1", "1000", "4097", "1048576"])
def test_harness_u32_sizes(device, n):
    assert run("--n", n, "--dist", "uniform", "--iters", "1")["pass"] is True


@pytest.mark.parametrize("dist", ["shuffle", "hash"])
def test_harness_u64_key_value(device, dist):
    line = run("--type", "u64kv", "--log2n", "21", "--dist", dist, "--iters", "1")
    assert line["pass"] is True


def test_harness_staged_pipeline(device):
    line = run("--log2n", "22", "--dist", "uniform", "--batches", "5")
    assert line["pass"] is True and line["batches"] == 5
    assert line["overlapped_ms"] > 0 and line["serialised_ms"] > 0


SHARDED = os.path.join(ROOT, "harness", "sharded_test")


def run_sharded(*args):
    out = subprocess.run([SHARDED] + list(args), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Sorting Finished: PASS" in out.stdout
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_sharded_harness_plain_path(device):
    line = run_sharded("--gpus", "1", "--log2n", "20", "--iters", "1")
    assert line["pass"] is True and line["exchange"] is False
    line = run_sharded("--gpus", "1", "--log2n", "18", "--iters", "1", "--type", "u64kv")
    assert line["pass"] is True and line["type"] == "u64kv"


def test_reference_call_sequence_runs(device):
    """The transcription of the reference harness's own call sequence (compiled by the CPU test
    suite): sorts 2^20 + 16 shuffled keys through Sort<Radix>::command and reads the histogram /
    prefix-scan tables back through InternalInterface's public views."""
    out = subprocess.run([os.path.join(ROOT, "harness", "reference_sequence")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Sorting Finished: PASS" in out.stdout
