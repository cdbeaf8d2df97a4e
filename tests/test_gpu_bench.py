"""bench.py at a small size on the GPU: the one-GPU line (the N > 1 code path, rehearsed over RCCL
with a single rank, is in test_gpu_zz_rccl.py).  Checks the JSON contract, not the speed."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True,
                         text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly one JSON line"
    return json.loads(lines[0])


def check_contract(line, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["unit"] == "Gkeys/s" and line["dtype"] == "u32" and line["scaling"] == "weak"
    assert line["n_gpus"] == 1 and line["steps"] == steps and line["warmup"] == warmup
    assert line["vs_baseline"] is None and line["higher_is_better"] is True
    assert "workload" in line["config"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert line["verified"] is True and line["value"] > 0


def test_bench_single_gpu_line(device):
    line = run_bench("--log2n", "22", "--steps", "3", "--warmup", "1", "--no-extras", "--cpu-log2n", "18")
    check_contract(line, 3, 1)
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == "Gkeys/s"
    assert line["roofline"]["launches"] == 3 * 8
