"""Launched by test_gpu_multigpu.py through torch.distributed.run, one rank per GPU: the 8-body chain of
bench_chain.py with body i on rank i mod N, the library's own RCCL communicator and ONE ncclAllReduce (of the link
sums) per Newton step.  Every rank must end on the same poses, and they must equal the oracle's single-process run,
bit for bit."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    # M3T_TEST_SHARE_ONE_GPU=1: every rank on device 0, the link sums over gloo through m3t_hip_comm_set_reduce_callback
    # (RCCL refuses two ranks on one device) -- the same worker, placement and checks on a one-GPU box
    share = os.environ.get("M3T_TEST_SHARE_ONE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    if share:
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=240))
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    pkg = importlib.import_module("3dobjecttracking_amd")
    import bench_chain as bc
    import scenes
    import util
    syn, host = pkg.synthetic, pkg.host
    n_bodies, n_frames = 8, 4
    inputs, joints, gt = bc.chain_inputs(scenes, syn, n_bodies, n_frames, 2)
    start_root = syn.perturb_pose(gt[0][0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    start_angles = gt[0][1] + 0.01
    hip = pkg.open_context(local)
    owned = [i for i, r in enumerate(pkg.sharding.place_bodies(n_bodies, world)) if r == rank]
    ch = bc.Chain(hip, host, syn, inputs, joints, start_root, start_angles, owned)
    reduce = None
    if share:
        reduce = pkg.sharding.HostReduce(hip, dist)
    else:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            buf = C.create_string_buffer(128)
            hip.call("comm_get_unique_id", buf, 128)
            uid = torch.tensor(list(buf.raw), dtype=torch.uint8, device="cuda")
        dist.broadcast(uid, 0)
        hip.call("comm_init_rank", C.create_string_buffer(bytes(uid.cpu().tolist()), 128), 128, world, rank)
    ch.upload(inputs, 0)
    assert ch.tracker.StartModalities(0)
    for k in range(n_frames):
        ch.upload(inputs, k)
        assert ch.tracker.ExecuteTrackingStep(k)
    name = C.create_string_buffer(96)
    hip.call("get_step_kernel", name, 96)
    assert name.value.decode() == "tracking_step_tree_segment_kernel", name.value  # one launch + one reduction per Newton step
    calls = C.c_longlong(0)
    hip.call("comm_get_allreduce_count", C.byref(calls))
    assert calls.value == 14 * n_frames, calls.value
    poses = torch.from_numpy(ch.poses())
    if not share:
        poses = poses.cuda()
    gathered = [torch.zeros_like(poses) for _ in range(world)]
    dist.all_gather(gathered, poses)
    if reduce is not None:
        assert reduce.error is None and reduce.calls == 14 * n_frames, (reduce.error, reduce.calls)
        reduce.close()
    else:
        hip.call("comm_destroy")
    if rank == 0:
        for r in range(1, world):
            assert torch.equal(gathered[0], gathered[r]), "replicas differ on rank %d" % r
        ora = util.open_oracle()
        oc = bc.Chain(ora, host, syn, inputs, joints, start_root, start_angles, range(n_bodies))
        oc.upload(inputs, 0)
        assert oc.tracker.StartModalities(0)
        for k in range(n_frames):
            oc.upload(inputs, k)
            assert oc.tracker.ExecuteTrackingStep(k)
        # what the all-reduce adds are the link sums: every link's modalities live on one rank, the others add
        # +0.0, the sum is exact in any order -- N ranks compute what one process computes, bit for bit
        err = np.abs(gathered[0].cpu().numpy() - oc.poses()).max()
        assert np.array_equal(gathered[0].cpu().numpy(), oc.poses()), err
        print("multigpu chain ok: %d ranks, max |pose - oracle| = %.3g" % (world, err))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
