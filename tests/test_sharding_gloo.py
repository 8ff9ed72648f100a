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


def _free_port():
    """a port nobody listens on right now (the rendezvous of a test must not depend on what else runs on the box)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


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
    port = _free_port()
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


# ---- one kinematic structure spread over two ranks: the single all-reduce of SURVEY §8e ----
def _chain_worker(rank, world, port, out_dir):
    import ctypes as C

    import torch
    import torch.distributed as dist
    from test_gpu_multibody import Chain, chain_inputs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    inputs, joint2parent, gt = chain_inputs(n_frames=3)
    start_a = util.syn.perturb_pose(gt[0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    api = util.open_oracle()
    owned = [0, 1] if world == 1 else [rank]  # body r's modality lives on rank r
    ch = Chain(api, inputs, joint2parent, start_a, gt[0][2] + 0.01, owned=owned)
    for i in owned:
        ch.cams[i].UpdateImage(inputs.color[i][0])
    assert ch.tracker.StartModalities(0)
    for k in range(len(gt)):
        for i in owned:
            ch.cams[i].UpdateImage(inputs.color[i][k])
        for c in range(7):
            assert ch.tracker.CalculateCorrespondences(k, c)
            for u in range(2):
                assert ch.tracker.CalculateGradientAndHessian(k, c, u)
                ptr, n = ch.tracker.CalculateOptimizationBegin()
                if world > 1:
                    buf = np.ctypeslib.as_array(ptr, shape=(n,))
                    t = torch.from_numpy(buf)  # shares memory with the library's buffer
                    dist.all_reduce(t)
                assert ch.tracker.CalculateOptimizationEnd()
        assert ch.tracker.CalculateResults(k)
    np.save(os.path.join(out_dir, "chain_%d_%d.npy" % (world, rank)), np.stack(ch.state()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_kinematic_structure_over_two_ranks(tmp_path):
    """every rank keeps the whole link tree but only its own bodies' modalities; one all-reduce of
    the stacked link sums per Newton step keeps all replicas identical and equal to the single-process result"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_chain_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    _chain_worker(0, 1, port + 1, str(tmp_path))
    ref = np.load(os.path.join(str(tmp_path), "chain_1_0.npy"))
    r0 = np.load(os.path.join(str(tmp_path), "chain_2_0.npy"))
    r1 = np.load(os.path.join(str(tmp_path), "chain_2_1.npy"))
    assert np.array_equal(r0, r1)       # replicas stay bit-identical
    assert np.array_equal(r0, ref)      # and equal the single-process run


# ---- four ranks: the sum over the ranks is exact (what is summed are the link sums) ----
def _four_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench_chain
    import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n_bodies, n_frames = 8, 2
    inputs, joints, gt = bench_chain.chain_inputs(scenes, util.syn, n_bodies, n_frames, 1)
    start_root = util.syn.perturb_pose(gt[0][0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    sh = __import__("importlib").import_module("3dobjecttracking_amd.sharding")
    owned = [i for i, r in enumerate(sh.place_bodies(n_bodies, world)) if r == rank]  # (two bodies per rank)
    ch = bench_chain.Chain(util.open_oracle(), util.host, util.syn, inputs, joints, start_root, gt[0][1] + 0.01, owned)
    ch.upload(inputs, 0)
    assert ch.tracker.StartModalities(0)
    for k in range(n_frames):
        ch.upload(inputs, k)
        for c in range(7):
            assert ch.tracker.CalculateCorrespondences(k, c)
            for u in range(2):
                assert ch.tracker.CalculateGradientAndHessian(k, c, u)
                ptr, n = ch.tracker.CalculateOptimizationBegin()
                assert n == n_bodies * 42
                if world > 1:
                    dist.all_reduce(torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(n,))))
                assert ch.tracker.CalculateOptimizationEnd()
        assert ch.tracker.CalculateResults(k)
    np.save(os.path.join(out_dir, "four_%d_%d.npy" % (world, rank)), ch.poses())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_four_ranks_equal_one_process_bit_for_bit(tmp_path):
    """the 8-body chain, two bodies' modalities per rank.  The ranks add up the LINK SUMS (6 + 36 floats per link): a
    link's modalities live on one rank, the other three add +0.0, so whatever order the all-reduce adds in, every rank
    holds the floats one process holds and solves the same system -- the poses are the single-process poses bit for
    bit.  (With the projected [dof x dof | dof] sums of round 3 the sum over the ranks was a reassociation of the sum
    over the links; the tracker's discrete decisions amplified it to 6e-3 on the poses within one frame.)"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_four_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    _four_worker(0, 1, port + 1, str(tmp_path))
    ref = np.load(os.path.join(str(tmp_path), "four_1_0.npy"))
    for r in range(4):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "four_4_%d.npy" % r)), ref)


# ---- soft constraints of a structure spread over ranks: every rank adds them after the sum, as one process does ----
def _soft_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from test_multibody_oracle import build_soft, random_pose
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    api = util.open_oracle()
    b1, b2 = random_pose(rng), random_pose(rng)
    link1, link2, opt = build_soft(api, b1, b2)
    d = random_pose(rng)
    d[:3, :3] = util.syn.rot_vec(rng.normal(size=3) * 0.3)
    d[:3, 3] *= 0.05
    link2.set_joint2parent_pose(np.linalg.inv(b1) @ d)
    tracker = util.host.Tracker(api, 1, 1)
    assert tracker.CalculateConsistentPoses()
    start = link2.joint2parent_pose()
    for it in range(5):
        ptr, n = tracker.CalculateOptimizationBegin()
        if world > 1:
            dist.all_reduce(torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(n,))))
        assert tracker.CalculateOptimizationEnd()
    assert np.max(np.abs(link2.joint2parent_pose() - start)) > 1e-3  # (the constraint did pull)
    state = np.stack([link1.link2world_pose(), link2.link2world_pose(), link2.joint2parent_pose()])
    np.save(os.path.join(out_dir, "soft_%d_%d.npy" % (world, rank)), state)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_soft_constraints_over_two_ranks_count_once(tmp_path):
    """every rank holds the whole structure, its soft constraints included.  Their g / H are added to the link sums
    in end(), after the ranks' sums have been added up, exactly where one process adds them (optimizer.cpp:281-286):
    they enter the system once, replicas identical and equal to the single process bit for bit.  (Round 3 summed the
    projected system, soft terms included: once per rank -- its advisor's finding.)"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_soft_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    _soft_worker(0, 1, port + 2, str(tmp_path))
    ref = np.load(os.path.join(str(tmp_path), "soft_1_0.npy"))
    r0 = np.load(os.path.join(str(tmp_path), "soft_2_0.npy"))
    r1 = np.load(os.path.join(str(tmp_path), "soft_2_1.npy"))
    assert np.array_equal(r0, r1) and np.array_equal(r0, ref)


def test_bodies_that_share_color_histograms_stay_on_one_rank():
    import importlib
    sh = importlib.import_module("3dobjecttracking_amd.sharding")
    assert sh.place_bodies(8, 4) == [0, 1, 2, 3, 0, 1, 2, 3]
    placed = sh.place_bodies(8, 4, shared_histograms=[(1, 5, 6), (2, 3)])
    assert placed[1] == placed[5] == placed[6] and placed[2] == placed[3]
    assert sorted(set(placed)) == [0, 1, 2, 3]  # every GPU still has work
    with pytest.raises(ValueError):
        sh.place_bodies(4, 2, shared_histograms=[(0, 1), (1, 2)])
