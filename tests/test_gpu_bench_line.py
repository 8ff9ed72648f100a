"""bench.py's contract with the driver, end to end on the GPU: ONE JSON line, and it is the last line of stdout -- also
when RCCL has printed its version banner through C stdio (the chain leg makes a communicator at world size 1)."""
import json
import os
import subprocess
import sys

import pytest

import util

pytestmark = pytest.mark.gpu


def test_chain_bench_line_is_the_last_line_of_stdout():
    out = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--config", "chain8", "--steps", "3",
                          "--warmup", "1", "--repeats", "1"], capture_output=True, text=True, timeout=600, cwd=util.ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    line = json.loads(lines[-1])
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["parity"]["bit_identical"] is True
    d = line["distributed_path_world1"]
    assert d["bit_identical_to_the_one_launch_step"] is True and d["allreduce_calls_per_step"] == 14.0
