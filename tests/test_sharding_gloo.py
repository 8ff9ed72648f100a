"""N>1 path on CPU: world_size-2 gloo processes shard the objects, track their shard and
gather the poses; the result equals the single-process run object for object.  The tracking
engine in this CPU test is the oracle (checker), the sharding / gather code is the product's."""
import os
import sys

import numpy as np
import pytest

import util

ROOT = util.ROOT
N_OBJECTS, N_FRAMES = 6, 3


def _track(ids):
    import scenes
    ora = util.open_oracle()
    poses = []
    for i in ids:  # one context per object keeps global object seeds (1000 + i)
        inputs = scenes.Inputs(1, N_FRAMES, n_divides=1, first_object=int(i))
        inst = scenes.Instance(ora, inputs)
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
        for k in range(N_FRAMES):
            inst.upload_frame(k)
            assert inst.tracker.ExecuteTrackingStep(k)
        poses.append(inst.poses()[0])
        ora = util.open_oracle()
    return np.stack(poses) if poses else np.zeros((0, 4, 4), np.float32)


def _worker(rank, world, port, mode, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = util.pkg.sharding if hasattr(util.pkg, "sharding") else __import__("importlib").import_module(
        "3dobjecttracking_amd.sharding")
    ids = sh.shard_objects(N_OBJECTS, rank, world, mode)
    local = _track(ids)
    allp = sh.gather_poses(ids, local, N_OBJECTS, dist)
    t = sh.max_over_ranks(1.0 + rank, dist)
    assert t == float(world)
    np.save(os.path.join(out_dir, "poses_%s_%d.npy" % (mode, rank)), allp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["round_robin", "block"])
def test_two_rank_sharding_matches_single_process(tmp_path, mode):
    import importlib
    import torch.multiprocessing as mp
    sh = importlib.import_module("3dobjecttracking_amd.sharding")
    ids0, ids1 = sh.shard_objects(N_OBJECTS, 0, 2, mode), sh.shard_objects(N_OBJECTS, 1, 2, mode)
    assert sorted(list(ids0) + list(ids1)) == list(range(N_OBJECTS))
    port = 29500 + os.getpid() % 2000 + (0 if mode == "block" else 1)
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    ref = _track(range(N_OBJECTS))
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "poses_%s_%d.npy" % (mode, rank)))
        assert np.array_equal(got, ref)


def test_single_process_gather_is_identity():
    import importlib
    sh = importlib.import_module("3dobjecttracking_amd.sharding")
    poses = np.random.default_rng(0).normal(size=(4, 4, 4)).astype(np.float32)
    out = sh.gather_poses([3, 1, 0, 2], poses, 4)
    assert np.array_equal(out[[3, 1, 0, 2]], poses)
    assert sh.max_over_ranks(2.5) == 2.5
