"""bench.py --gpus N (the driver's entry point): N ranks or a loud failure, never a quiet single-rank run; and the
per-rank object / model placement of SURVEY 8(e) for N = 1, 2, 4, 8 (rbot_evaluator.cpp:144 runs its sequences as
independent units the same way)."""
import os
import subprocess
import sys

import pytest

import util

sys.path.insert(0, util.ROOT)
import bench  # noqa: E402


def test_gpus_2_without_two_gpus_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert "only 0 GPU(s) visible" in out.stderr and "--gpus 2" in out.stderr
    assert '"n_gpus"' not in out.stdout  # no line at all rather than one that says n_gpus 1


def test_launcher_command_line():
    cmd = bench.launch_ranks(4, ["--gpus", "4", "--config", "synth512", "--steps", "3"], n_visible=8)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    script = cmd.index(os.path.join(util.ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--config", "synth512", "--steps", "3"]
    with pytest.raises(SystemExit):
        bench.launch_ranks(8, [], n_visible=4)


def test_dry_run_switch_lets_n_ranks_share_one_gpu(monkeypatch):
    monkeypatch.setenv("M3T_BENCH_SHARE_ONE_GPU", "1")
    cmd = bench.launch_ranks(2, ["--gpus", "2"], n_visible=1)
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    with pytest.raises(SystemExit):
        bench.launch_ranks(2, ["--gpus", "2"], n_visible=0)  # (still needs a GPU)
    monkeypatch.delenv("M3T_BENCH_SHARE_ONE_GPU")
    with pytest.raises(SystemExit):
        bench.launch_ranks(2, ["--gpus", "2"], n_visible=1)


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "4"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "--gpus 4 but the launcher started 2 rank(s)" in out.stderr


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_weak_scaling_placement(world):
    plans = [bench.rank_plan("rbot64", r, world) for r in range(world)]
    seen = []
    for r, p in enumerate(plans):
        assert p["n_obj"] == 64 and p["total_objects"] == 64 * world
        assert p["global_ids"] == list(range(64 * r, 64 * r + 64))  # rank r owns [64 r, 64 r + 64)
        assert p["first_object"] == 64 * r                          # its own rendered streams
        assert set(p["model_of"]) == set(range(p["n_models"])) and p["n_models"] == 18
        seen += p["global_ids"]
    assert sorted(seen) == list(range(64 * world))


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("config,total", [("ycb21", 21), ("synth512", 512)])
def test_strong_scaling_placement(config, total, world):
    plans = [bench.rank_plan(config, r, world) for r in range(world)]
    seen = []
    for r, p in enumerate(plans):
        assert p["total_objects"] == total
        assert all(i % world == r for i in p["global_ids"])          # object i -> GPU i mod G
        assert p["n_obj"] == len(p["global_ids"]) == len(p["model_of"])
        # models only where used: every model a rank builds is looked at by one of ITS objects
        assert set(p["model_of"]) == set(range(p["n_models"]))
        assert p["n_models"] <= bench.CONFIGS[config]["models"]
        seen += p["global_ids"]
    assert sorted(seen) == list(range(total))
    if config == "synth512" and world == 8:
        assert all(p["n_obj"] == 64 for p in plans)                  # 512 / 8: the split-kernel regime per GPU


def test_chain_bodies_round_robin():
    pkg = util.pkg
    for world in (1, 2, 4, 8):
        ranks = pkg.sharding.place_bodies(8, world)
        assert ranks == [i % world for i in range(8)]


def test_rccl_ranks_come_from_the_live_process_group():
    """config.rccl_ranks answers "how many ranks did RCCL span" from the communicator, never from argv: 0 without a
    process group and 0 under gloo (the one-GPU dry run of the N-rank path runs over gloo), the group's size under nccl"""
    import torch.distributed as dist
    assert bench.live_rccl_ranks(None) == 0
    assert bench.live_rccl_ranks(dist) == 0  # no group yet
    port = util.free_port() if hasattr(util, "free_port") else 29631
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        assert dist.get_world_size() == 1 and bench.live_rccl_ranks(dist) == 0
    finally:
        dist.destroy_process_group()

    class FakeNccl:  # (no RCCL without a GPU: the rule itself)
        @staticmethod
        def is_initialized():
            return True

        @staticmethod
        def get_backend():
            return "nccl"

        @staticmethod
        def get_world_size():
            return 8
    assert bench.live_rccl_ranks(FakeNccl) == 8
