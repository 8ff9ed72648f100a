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


def test_two_rank_code_path_dry_run_on_one_gpu():
    """`bench.py --gpus 2` end to end on a one-GPU box (M3T_BENCH_SHARE_ONE_GPU=1: both ranks on device 0, gloo
    instead of RCCL, one workgroup per object): launcher, sharding, barriers, max over ranks, rank 0's parity check and
    its JSON line -- everything of the N-rank path but the transport.  The line carries the dry-run mark."""
    env = dict(os.environ, M3T_BENCH_SHARE_ONE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--objects", "8",
                          "--steps", "3", "--warmup", "1", "--repeats", "2", "--cpu-seconds", "0.5", "--n-divides", "2"],
                         capture_output=True, text=True, timeout=900, cwd=util.ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "dry_run" in line and line["metric"].startswith("[DRY RUN")
    assert line["scaling"] == "weak" and line["config"]["objects_per_gpu"] == 8 and line["config"]["ranks"] == 2
    assert line["parity"]["bit_identical"] is True
    # the ranks of this run talked over gloo: RCCL spanned none of them (the field is asked of the live process group,
    # not echoed from --gpus)
    assert line["config"]["rccl_ranks"] == 0
    assert line["roofline"]["workgroups_per_object"] == 1
    # weak scaling: value = the objects of both ranks over the slower rank's time
    assert abs(line["value"] - 2 * 8 * 3 / (line["ms_per_step"] * 3e-3)) / line["value"] < 0.01


def test_chain_two_rank_code_path_dry_run_on_one_gpu():
    """`bench.py --config chain8 --gpus 2` on a one-GPU box: both ranks on device 0, body i's modality on rank i mod 2,
    the link sums summed over gloo through m3t_hip_comm_set_reduce_callback (RCCL refuses two ranks on one device).
    Launcher, placement, the segment kernel with partial ownership, 14 reductions per step, rank 0's parity check
    against the oracle's single process: everything of the 8-GPU chain leg but the transport."""
    env = dict(os.environ, M3T_BENCH_SHARE_ONE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--config", "chain8", "--gpus", "2",
                          "--steps", "3", "--warmup", "1", "--repeats", "1"],
                         capture_output=True, text=True, timeout=600, cwd=util.ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "dry_run" in line and line["metric"].startswith("[DRY RUN")
    assert line["parity"]["bit_identical"] is True and line["parity"]["n"] == 8
    assert line["config"]["rccl_ranks"] == 0  # (no RCCL communicator: the host's transport)
    assert line["config"]["allreduce_calls_per_step"] == 14.0
    assert line["roofline"]["kernel"].startswith("whole step")  # not the one-launch kernel
    assert line["config"]["step_kernel"] == "tracking_step_tree_segment_kernel"
