"""Kinematic structures on the device (m3t_links.hip) against the oracle: Link tree with a 1-dof
joint tracked by two RegionModalities, hard constraints (constraint_convergence.cpp), and the
begin -> all-reduce -> end split of Optimizer::CalculateOptimization with RCCL (world size 1)."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
import util
from test_multibody_oracle import build, random_pose, run_convergence, run_soft
from util import host, syn

gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("seed", range(4))
def test_constraint_convergence_on_device(seed):
    errs_h, poses_h = run_convergence(util.open_hip(), seed)
    errs_o, poses_o = run_convergence(util.open_oracle(), seed)
    assert errs_h[-1][0] < 2e-5 and errs_h[-1][1] < 2e-5
    for ph, po in zip(poses_h, poses_o):  # atan2f / tanf / tan through csrc/m3t_exact_math.h on both sides (round 6)
        for a, b in zip(ph, po):
            assert np.array_equal(a, b)


@gpu
@pytest.mark.parametrize("kw", [dict(), dict(max_distance_rotation=0.1, max_distance_translation=0.005),
                                dict(directions=(1, 0, 1, 0, 1, 1), standard_deviation_rotation=0.05),
                                dict(directions=(1, 1, 1, 0, 0, 0), max_distance_rotation=0.2, root_free=False)])
def test_soft_constraints_on_device(kw):
    """SoftConstraint::AddGradientsAndHessiansToLinks in links_project_kernel against the oracle"""
    errs_h, poses_h = run_soft(util.open_hip(), 2, 25, **kw)
    errs_o, poses_o = run_soft(util.open_oracle(), 2, 25, **kw)
    assert errs_h[-1][0] < errs_h[0][0]
    for ph, po in zip(poses_h, poses_o):
        for a, b in zip(ph, po):
            assert np.array_equal(a, b)


class Chain:
    """body A (free root) -- revolute joint about the joint z axis -- body B, one camera each"""

    def __init__(self, api, inputs, joint2parent, start_a, start_angle, owned=(0, 1)):
        rp = dict(syn.RBOT_REGION_PARAMS)
        self.api = api
        self.models = [host.RegionModel(api, data_points=m[0], orientations=m[1], contour_lengths=m[2])
                       for m in inputs.region_models]
        self.bodies = [host.Body(api, np.eye(4)), host.Body(api, np.eye(4))]
        self.cams = [host.ColorCamera(api, **inputs.intr) for _ in range(2)]
        self.mods = {i: host.RegionModality(api, self.bodies[i], self.cams[i], self.models[i], **rp) for i in owned}
        self.link_a = host.Link(api, body=self.bodies[0])
        self.link_b = host.Link(api, body=self.bodies[1], parent=self.link_a,
                                joint2parent_pose=joint2parent @ syn.make_pose(syn.rot_vec([0, 0, start_angle]), [0, 0, 0]),
                                free_directions=(0, 0, 1, 0, 0, 0))
        if 0 in self.mods:
            self.link_a.AddModality(self.mods[0])
        if 1 in self.mods:
            self.link_b.AddModality(self.mods[1])
        self.opt = host.Optimizer(api, root_link=self.link_a)
        self.tracker = host.Tracker(api, 7, 2)
        self.bodies[0].set_body2world_pose(start_a)
        assert self.tracker.CalculateConsistentPoses()  # body B follows from the joint

    def upload(self, inputs, k):
        for i in range(2):
            self.cams[i].UpdateImage(inputs.color[i][k])

    def state(self):
        return [b.body2world_pose() for b in self.bodies] + [self.link_b.joint2parent_pose()]


def chain_inputs(n_frames=4):
    inputs = scenes.Inputs(2, 1, n_divides=2)
    rng = np.random.default_rng(11)
    joint2parent = syn.make_pose(syn.rot_vec([0.3, -0.2, 0.1]), [0.16, 0.02, 0.0])
    pose_a = inputs.gt[0][0].copy()
    gt, angle = [], 0.2
    inputs.color = [[], []]
    for k in range(n_frames):
        pose_a = syn.perturb_pose(pose_a, rng, rot_deg=0.7, trans=0.002)
        angle += rng.uniform(-0.03, 0.03)
        pose_b = pose_a @ joint2parent @ syn.make_pose(syn.rot_vec([0, 0, angle]), [0, 0, 0])
        gt.append((pose_a.copy(), pose_b.copy(), angle))
        inputs.color[0].append(inputs.scenes[0].render(pose_a))
        inputs.color[1].append(inputs.scenes[1].render(pose_b))
    return inputs, joint2parent, gt


@gpu
def test_kinematic_chain_tracking_matches_oracle():
    inputs, joint2parent, gt = chain_inputs()
    start_a = syn.perturb_pose(gt[0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    results = {}
    for name, api in (("hip", util.open_hip()), ("oracle", util.open_oracle())):
        ch = Chain(api, inputs, joint2parent, start_a, gt[0][2] + 0.01)
        ch.upload(inputs, 0)
        assert ch.tracker.StartModalities(0)
        states = []
        for k in range(len(gt)):
            ch.upload(inputs, k)
            assert ch.tracker.ExecuteTrackingStep(k)
            states.append(ch.state())
        results[name] = states
        # the joint keeps the structure consistent: B == A * joint2parent (body2joint = I)
        a, b, j = states[-1]
        assert np.max(np.abs(a.astype(np.float64) @ j.astype(np.float64) - b)) < 1e-5
        # and the tracker follows the ground truth of both bodies
        assert syn.pose_errors(b, gt[-1][1])[1] < 0.05
    # a pure tree uses + - * / only: with the reference's summation order the device is bit-exact
    for sh, so in zip(results["hip"], results["oracle"]):
        for x, y in zip(sh, so):
            assert np.array_equal(x, y)


@gpu
def test_begin_allreduce_end_with_rccl_world1():
    """the multi-GPU split: begin() exposes one device buffer, torch.distributed (RCCL) sums it,
    end() solves; with world size 1 the result equals the single call bit for bit"""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        out = []
        for split in (False, True):
            rng = np.random.default_rng(3)
            api = util.open_hip()
            b1, b2 = random_pose(rng), random_pose(rng)
            link1, link2, opt = build(api, b1, b2, directions=(1, 1, 0, 1, 1, 1))
            d = random_pose(rng)
            d[:3, :3] = syn.rot_vec(rng.normal(size=3) * 0.3)
            link2.set_joint2parent_pose(np.linalg.inv(b1) @ d)
            tracker = host.Tracker(api, 1, 1)
            assert tracker.CalculateConsistentPoses()
            stream_ptr = C.c_void_p()
            api.call("get_stream", C.byref(stream_ptr))
            ext = torch.cuda.ExternalStream(stream_ptr.value)
            for it in range(3):
                if split:
                    ptr, n = tracker.CalculateOptimizationBegin()
                    assert n == 2 * 42  # the link sums of both links
                    addr = C.cast(ptr, C.c_void_p).value

                    class _Buf:  # zero-copy view of the library's device buffer
                        __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (addr, False),
                                                    "version": 2}
                    with torch.cuda.stream(ext):
                        t = torch.as_tensor(_Buf(), device="cuda")
                        dist.all_reduce(t)  # one RCCL all-reduce over the stacked sums
                    assert tracker.CalculateOptimizationEnd()
                else:
                    assert tracker.CalculateOptimization(0, 0, 0)
            out.append((link1.link2world_pose(), link2.link2world_pose(), link2.joint2parent_pose()))
        for a, b in zip(*out):
            assert np.array_equal(a, b)
    finally:
        dist.destroy_process_group()


@gpu
def test_native_rccl_allreduce_world1():
    """the library's own communicator (m3t_hip_comm_get_unique_id / comm_init_rank, RCCL opened with dlopen) and its
    own ncclAllReduce call site: with a communicator set, CalculateOptimization runs begin -> all-reduce -> end by
    itself; with one rank the poses equal the plain call bit for bit"""
    out = []
    for with_comm in (False, True):
        rng = np.random.default_rng(3)
        api = util.open_hip()
        b1, b2 = random_pose(rng), random_pose(rng)
        link1, link2, opt = build(api, b1, b2, directions=(1, 1, 0, 1, 1, 1))
        d = random_pose(rng)
        d[:3, :3] = syn.rot_vec(rng.normal(size=3) * 0.3)
        link2.set_joint2parent_pose(np.linalg.inv(b1) @ d)
        tracker = host.Tracker(api, 1, 1)
        if with_comm:
            uid = C.create_string_buffer(128)
            api.call("comm_get_unique_id", uid, 128)
            api.call("comm_init_rank", uid, 128, 1, 0)
            assert api.raw("calculate_optimization_allreduce") == -2  # begin() first
        assert tracker.CalculateConsistentPoses()
        for it in range(3):
            assert tracker.CalculateOptimization(0, 0, 0)
        out.append((link1.link2world_pose(), link2.link2world_pose(), link2.joint2parent_pose()))
        if with_comm:
            api.call("comm_destroy")
    for a, b in zip(*out):
        assert np.array_equal(a, b)


@gpu
def test_tracking_step_all_reduces_with_a_communicator_set():
    """ExecuteTrackingStep of a kinematic structure while a communicator is set (m3t_hip.h: the step runs
    project -> ONE ncclAllReduce -> solve per Newton step by itself, optimizer.cpp:309-321 being the distributed sum):
    the library's own RCCL communicator at world size 1 on the 2-body chain.  The collective is really issued --
    7 x 2 calls per frame -- the launch is the one split at the all-reduce, not the one-launch tree kernel, and the
    poses equal the oracle's (= the single-GPU path's) bit for bit: a sum over one rank changes nothing."""
    inputs, joint2parent, gt = chain_inputs(3)
    start_a = syn.perturb_pose(gt[0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    results = {}
    for name, api in (("hip", util.open_hip()), ("oracle", util.open_oracle())):
        ch = Chain(api, inputs, joint2parent, start_a, gt[0][2] + 0.01)
        if name == "hip":
            uid = C.create_string_buffer(128)
            api.call("comm_get_unique_id", uid, 128)
            api.call("comm_init_rank", uid, 128, 1, 0)
        ch.upload(inputs, 0)
        assert ch.tracker.StartModalities(0)
        states = []
        for k in range(len(gt)):
            ch.upload(inputs, k)
            assert ch.tracker.ExecuteTrackingStep(k)
            states.append(ch.state())
        results[name] = states
        if name == "hip":
            count = C.c_longlong(-1)
            api.call("comm_get_allreduce_count", C.byref(count))
            assert count.value == len(gt) * 7 * 2
            kernel = C.create_string_buffer(64)
            api.call("get_step_kernel", kernel, 64)
            assert kernel.value.decode() != "tracking_step_tree_kernel"
            api.call("comm_destroy")
    for sh, so in zip(results["hip"], results["oracle"]):
        for x, y in zip(sh, so):
            assert np.array_equal(x, y)


@gpu
def test_soft_constraints_count_once_when_the_host_sums_the_buffers():
    """two contexts on this GPU stand in for two ranks that hold the same structure, soft constraint included: the
    host adds their begin() buffers (what an all-reduce does) and hands the sum to both.  What is summed are the link
    sums of the modalities; each context adds the soft constraint's terms afterwards, in end(), as one process does:
    both replicas equal the single-context run bit for bit (the round-3 protocol summed the projected system, soft
    terms included, and counted them once per rank)."""
    import torch
    from test_multibody_oracle import build_soft

    def structure(api):
        rng = np.random.default_rng(7)
        b1, b2 = random_pose(rng), random_pose(rng)
        link1, link2, opt = build_soft(api, b1, b2)
        d = random_pose(rng)
        d[:3, :3] = syn.rot_vec(rng.normal(size=3) * 0.3)
        d[:3, 3] *= 0.05
        link2.set_joint2parent_pose(np.linalg.inv(b1) @ d)
        tracker = host.Tracker(api, 1, 1)
        return tracker, (link1, link2)

    def state(links):
        return np.stack([links[0].link2world_pose(), links[1].link2world_pose(), links[1].joint2parent_pose()])

    def view(ptr, n):
        addr = C.cast(ptr, C.c_void_p).value

        class _Buf:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (addr, False), "version": 2}
        return torch.as_tensor(_Buf(), device="cuda")

    tracker, links = structure(util.open_hip())
    assert tracker.CalculateConsistentPoses()
    start_joint = links[1].joint2parent_pose()
    for it in range(5):
        assert tracker.CalculateOptimization(0, 0, 0)
    single = state(links)
    assert np.max(np.abs(single[2] - start_joint)) > 1e-3  # (the constraint did pull)
    ranks = [structure(util.open_hip()) for _ in range(2)]
    for tr, _ in ranks:
        assert tr.CalculateConsistentPoses()
    for it in range(5):
        bufs = [view(*tr.CalculateOptimizationBegin()) for tr, _ in ranks]
        torch.cuda.synchronize()
        total = bufs[0] + bufs[1]
        for b in bufs:
            b.copy_(total)
        torch.cuda.synchronize()
        for tr, _ in ranks:
            assert tr.CalculateOptimizationEnd()
    replicas = [state(l) for _, l in ranks]
    assert np.array_equal(replicas[0], replicas[1])
    assert np.array_equal(replicas[0], single)


@gpu
def test_chain_spread_over_two_contexts_equals_one_context():
    """SURVEY 8e's exchange step on the hardware, minus the transport: the 2-body chain held by two contexts of this
    GPU (each keeps the whole link tree and ONE body's modality -- what two ranks hold), the stacked link
    sums of begin() added by the host (what ncclAllReduce does) and handed to both.  links_gather_kernel /
    links_solve_sums_kernel run exactly as on two GPUs; replicas identical and equal to one context that owns both
    modalities, bit for bit (each rank adds exact zeros for the body it does not own), which in turn equals the
    oracle (test_kinematic_chain_tracking_matches_oracle)"""
    import torch

    def view(ptr, n):
        addr = C.cast(ptr, C.c_void_p).value

        class _Buf:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (addr, False), "version": 2}
        return torch.as_tensor(_Buf(), device="cuda")

    inputs, joint2parent, gt = chain_inputs(3)
    start_a = syn.perturb_pose(gt[0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)

    def run(owners):
        chains = [Chain(util.open_hip(), inputs, joint2parent, start_a, gt[0][2] + 0.01, owned=o) for o in owners]
        for ch, o in zip(chains, owners):
            for i in o:
                ch.cams[i].UpdateImage(inputs.color[i][0])
            assert ch.tracker.StartModalities(0)
        for k in range(len(gt)):
            for ch, o in zip(chains, owners):
                for i in o:
                    ch.cams[i].UpdateImage(inputs.color[i][k])
            for c in range(7):
                for ch in chains:
                    assert ch.tracker.CalculateCorrespondences(k, c)
                for u in range(2):
                    bufs = []
                    for ch in chains:
                        assert ch.tracker.CalculateGradientAndHessian(k, c, u)
                        bufs.append(view(*ch.tracker.CalculateOptimizationBegin()))
                    if len(chains) > 1:
                        torch.cuda.synchronize()
                        total = bufs[0] + bufs[1]
                        for b in bufs:
                            b.copy_(total)
                        torch.cuda.synchronize()
                    for ch in chains:
                        assert ch.tracker.CalculateOptimizationEnd()
            for ch in chains:
                assert ch.tracker.CalculateResults(k)
        return [np.stack(ch.state()) for ch in chains]

    one = run([[0, 1]])[0]
    two = run([[0], [1]])
    assert np.array_equal(two[0], two[1])
    assert np.array_equal(two[0], one)


@gpu
def test_eight_body_chain_over_four_contexts_equals_the_oracle():
    """BASELINE configs[4] spread the way four GPUs hold it (body i's modality in context i mod 4, the whole link tree
    everywhere), the link sums of begin() added by the host in an order of its own -- (c3 + c1) + (c0 + c2) -- and handed
    to all four: every replica ends on the poses of the oracle's single process, bit for bit, after two frames.  With
    the projected sums of round 3 this order-free identity did not hold (reassociation, amplified to 6e-3 within a
    frame by the tracker's discrete decisions)."""
    import torch
    import bench_chain as bc

    def view(ptr, n):
        addr = C.cast(ptr, C.c_void_p).value

        class _Buf:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (addr, False), "version": 2}
        return torch.as_tensor(_Buf(), device="cuda")

    n_bodies, n_frames, world = 8, 2, 4
    inputs, joints, gt = bc.chain_inputs(scenes, syn, n_bodies, n_frames, 2)
    start_root = syn.perturb_pose(gt[0][0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    start_angles = gt[0][1] + 0.01
    placed = util.pkg.sharding.place_bodies(n_bodies, world)
    chains = [bc.Chain(util.open_hip(), host, syn, inputs, joints, start_root, start_angles,
                       [i for i, r in enumerate(placed) if r == rank]) for rank in range(world)]
    oc = bc.Chain(util.open_oracle(), host, syn, inputs, joints, start_root, start_angles, range(n_bodies))
    for ch in chains + [oc]:
        ch.upload(inputs, 0)
        assert ch.tracker.StartModalities(0)
    for k in range(n_frames):
        oc.upload(inputs, k)
        assert oc.tracker.ExecuteTrackingStep(k)
        for ch in chains:
            ch.upload(inputs, k)
        for c in range(7):
            for ch in chains:
                assert ch.tracker.CalculateCorrespondences(k, c)
            for u in range(2):
                bufs = []
                for ch in chains:
                    assert ch.tracker.CalculateGradientAndHessian(k, c, u)
                    ptr, n = ch.tracker.CalculateOptimizationBegin()
                    assert n == n_bodies * 42
                    bufs.append(view(ptr, n))
                torch.cuda.synchronize()
                total = (bufs[3] + bufs[1]) + (bufs[0] + bufs[2])
                for b in bufs:
                    b.copy_(total)
                torch.cuda.synchronize()
                for ch in chains:
                    assert ch.tracker.CalculateOptimizationEnd()
        for ch in chains:
            assert ch.tracker.CalculateResults(k)
    ref = oc.poses()
    for ch in chains:
        assert np.array_equal(ch.poses(), ref)


@gpu
def test_rigid_context_switches_to_general_path_for_begin_end():
    """begin/end on a rigid-only context == the rigid fast path within one Newton-step tolerance"""
    inputs = scenes.Inputs(2, 2, n_divides=2)
    poses = []
    for split in (False, True):
        hip = util.open_hip()
        hip.call("set_fused_step", 0)
        a = scenes.Instance(hip, inputs)
        a.upload_frame(0)
        assert a.tracker.StartModalities(0)
        assert a.tracker.CalculateCorrespondences(0, 0)
        assert a.tracker.CalculateGradientAndHessian(0, 0, 0)
        if split:
            ptr, n = a.tracker.CalculateOptimizationBegin()
            assert n == 2 * 42
            assert a.tracker.CalculateOptimizationEnd()
        else:
            assert a.tracker.CalculateOptimization(0, 0, 0)
        poses.append(np.stack(a.poses()))
    assert np.array_equal(poses[0], poses[1])


def chain_state(ch):
    return [b.body2world_pose() for b in ch.bodies] + [l.joint2parent_pose() for l in ch.links[1:]]


@gpu
def test_eight_body_chain_matches_the_oracle():
    """BASELINE configs[4] on ONE GPU: bench.py --config chain8's own structure (bench_chain.chain_inputs: a free root
    and seven revolute joints, 13 dof, one RegionModality per body; optimizer.cpp:144-167, link.cpp:159-241) against
    the oracle over 5 frames: all 8 body2world and all 7 joint2parent poses after every frame, bit for bit"""
    import bench_chain
    inputs, joints, gt = bench_chain.chain_inputs(scenes, syn, 8, 5, 2)
    start_root = syn.perturb_pose(gt[0][0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    start_angles = gt[0][1] + 0.01
    states = {}
    for name, api in (("hip", util.open_hip()), ("oracle", util.open_oracle())):
        ch = bench_chain.Chain(api, host, syn, inputs, joints, start_root, start_angles, range(8))
        ch.upload(inputs, 0)
        assert ch.tracker.StartModalities(0)
        out = []
        for k in range(len(gt)):
            ch.upload(inputs, k)
            assert ch.tracker.ExecuteTrackingStep(k)
            out.append(chain_state(ch))
        states[name] = out
        if name == "hip":  # the whole loop nest of the structure in one launch, eight workgroups per body (round 5)
            kernel = C.create_string_buffer(64)
            api.call("get_step_kernel", kernel, 64)
            assert kernel.value.decode() == "tracking_step_tree_split_kernel"
            shape = (C.c_int * 4)()
            api.call("get_step_shape", shape)
            assert list(shape)[:3] == [8, 8, 512], list(shape)
    for k, (sh, so) in enumerate(zip(states["hip"], states["oracle"])):
        assert len(sh) == 15
        for x, y in zip(sh, so):
            assert np.array_equal(x, y), k
    # the chain is tracked: every body within 5 cm / 5 degrees of the ground truth (rbot_evaluator.cpp:416-433)
    for i in range(8):
        e = syn.pose_errors(states["hip"][-1][i], gt[-1][0][i])
        assert e[0] < np.deg2rad(5) and e[1] < 0.05


class DepthChain:
    """body A (free root) -- revolute joint -- body B, a RegionModality (measured occlusions) and a DepthModality on
    each link, YCB parameters, a colour and a depth camera per body"""

    def __init__(self, api, inputs, joint2parent, start_a, start_angle):
        # (32 bins: with 16 the pair table is staged in LDS and the structure takes the per-sub-step launches)
        rp, dp = dict(syn.YCB_REGION_PARAMS, n_histogram_bins=32), dict(syn.YCB_DEPTH_PARAMS)
        self.bodies = [host.Body(api, np.eye(4)), host.Body(api, np.eye(4))]
        self.cams = [host.ColorCamera(api, **inputs.intr) for _ in range(2)]
        self.dcams = [host.DepthCamera(api, depth_scale=inputs.depth_scale, **inputs.intr) for _ in range(2)]
        rmodels = [host.RegionModel(api, data_points=m[0], orientations=m[1], contour_lengths=m[2]) for m in inputs.region_models]
        dmodels = [host.DepthModel(api, data_points=m[0], orientations=m[1], surface_areas=m[2]) for m in inputs.depth_models]
        self.link_a = host.Link(api, body=self.bodies[0])
        self.link_b = host.Link(api, body=self.bodies[1], parent=self.link_a,
                                joint2parent_pose=joint2parent @ syn.make_pose(syn.rot_vec([0, 0, start_angle]), [0, 0, 0]),
                                free_directions=(0, 0, 1, 0, 0, 0))
        for i, link in enumerate((self.link_a, self.link_b)):
            link.AddModality(host.RegionModality(api, self.bodies[i], self.cams[i], rmodels[i], depth_camera=self.dcams[i], **rp))
            link.AddModality(host.DepthModality(api, self.bodies[i], self.dcams[i], dmodels[i], **dp))
        self.opt = host.Optimizer(api, root_link=self.link_a)
        self.tracker = host.Tracker(api, 4, 2)
        self.bodies[0].set_body2world_pose(start_a)
        assert self.tracker.CalculateConsistentPoses()

    def upload(self, frames, k):
        for i in range(2):
            self.cams[i].UpdateImage(frames[i][k][0])
            self.dcams[i].UpdateImage(frames[i][k][1])

    def state(self):
        return [b.body2world_pose() for b in self.bodies] + [self.link_b.joint2parent_pose()]


@gpu
@pytest.mark.parametrize("with_depth", [False, True])
def test_bodies_of_a_structure_split_over_workgroups(with_depth, monkeypatch):
    """tracking_step_tree_split_kernel (round 5): 2, 4 and 8 workgroups per tracked link -- the parts of a link exchange
    their lines' distributions (with measured occlusions: the lines' flags too; their points' correspondences) once per
    correspondence iteration, the link sums travel between the links as before -- against one workgroup per link and
    against the oracle: the same bits after every frame, Region links and Region + Depth links"""
    import bench_chain
    start_rng = np.random.default_rng(5)
    if with_depth:
        inputs = scenes.Inputs(2, 1, n_divides=2, with_depth=True)
        rng = np.random.default_rng(11)
        joint2parent = syn.make_pose(syn.rot_vec([0.3, -0.2, 0.1]), [0.16, 0.02, 0.0])
        pose_a, angle, frames, first = inputs.gt[0][0].copy(), 0.2, [[], []], None
        for k in range(3):
            pose_a = syn.perturb_pose(pose_a, rng, rot_deg=0.7, trans=0.002)
            angle += rng.uniform(-0.03, 0.03)
            pose_b = pose_a @ joint2parent @ syn.make_pose(syn.rot_vec([0, 0, angle]), [0, 0, 0])
            first = first or (pose_a.copy(), angle)
            frames[0].append(inputs.scenes[0].render(pose_a))
            frames[1].append(inputs.scenes[1].render(pose_b))
        start_a = syn.perturb_pose(first[0], start_rng, rot_deg=0.5, trans=0.001)
        build = lambda api: DepthChain(api, inputs, joint2parent, start_a, first[1] + 0.01)
        n, n_frames, feed = 2, 3, frames
    else:
        n = 4
        inputs, joints, gt = bench_chain.chain_inputs(scenes, syn, n, 4, 2)
        start_root = syn.perturb_pose(gt[0][0][0], start_rng, rot_deg=0.5, trans=0.001)
        build = lambda api: bench_chain.Chain(api, host, syn, inputs, joints, start_root, gt[0][1] + 0.01, range(n))
        n_frames, feed = len(gt), inputs
    states = {}
    for parts in ("oracle", 1, 2, 4, 8):
        if parts == "oracle":
            api = util.open_oracle()
        else:
            monkeypatch.setenv("M3T_HIP_TREE_PARTS", str(parts))
            api = util.open_hip()
        ch = build(api)
        ch.upload(feed, 0)
        assert ch.tracker.StartModalities(0)
        out = []
        for k in range(n_frames):
            ch.upload(feed, k)
            assert ch.tracker.ExecuteTrackingStep(k)
            out.append(ch.state() if with_depth else chain_state(ch))
        states[parts] = out
        if parts != "oracle":
            shape = (C.c_int * 4)()
            api.call("get_step_shape", shape)
            kernel = C.create_string_buffer(64)
            api.call("get_step_kernel", kernel, 64)
            assert list(shape)[:2] == [n, parts], (parts, list(shape))
            assert kernel.value.decode() == ("tracking_step_tree_kernel" if parts == 1 else "tracking_step_tree_split_kernel")
    for parts in (1, 2, 4, 8):
        for sa, sb in zip(states["oracle"], states[parts]):
            for x, y in zip(sa, sb):
                assert np.array_equal(x, y), parts


@gpu
def test_object_split_off_keeps_the_tree_kernel_out():
    """m3t_hip_set_object_split(ctx, 0) -- the switch of a context that shares its GPU -- also keeps
    tracking_step_tree_kernel (workgroups that wait for each other) out: per-sub-step launches, the same bits"""
    import bench_chain
    inputs, joints, gt = bench_chain.chain_inputs(scenes, syn, 4, 3, 2)
    start_root = syn.perturb_pose(gt[0][0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    states = {}
    for name in ("tree", "unfused"):
        api = util.open_hip()
        if name == "unfused":
            api.call("set_object_split", 0)
        ch = bench_chain.Chain(api, host, syn, inputs, joints, start_root, gt[0][1] + 0.01, range(4))
        ch.upload(inputs, 0)
        assert ch.tracker.StartModalities(0)
        out = []
        for k in range(len(gt)):
            ch.upload(inputs, k)
            assert ch.tracker.ExecuteTrackingStep(k)
            out.append(chain_state(ch))
        states[name] = out
        kernel = C.create_string_buffer(64)
        api.call("get_step_kernel", kernel, 64)
        assert kernel.value.decode().startswith("tracking_step_tree") == (name == "tree")
    for sa, sb in zip(states["tree"], states["unfused"]):
        for x, y in zip(sa, sb):
            assert np.array_equal(x, y)


@gpu
def test_closed_chain_with_a_hard_constraint_matches_the_oracle():
    """A -- revolute -- B -- revolute -- C with a Constraint (constraint.cpp:81-102, translation directions) that ties
    a point of C back to A: a closed kinematic loop, three RegionModalities.  The constraint rows make the system an
    indefinite KKT matrix.  atan2f / tanf / tan of the constraint Jacobians are one IEEE-only implementation on both
    sides since round 6 (csrc/m3t_exact_math.h; rounds 2-5 stated 2e-5 here): the tolerance is zero, as everywhere."""
    inputs = scenes.Inputs(3, 1, n_divides=2)
    rng = np.random.default_rng(3)
    j1 = syn.make_pose(syn.rot_vec([0.2, -0.1, 0.3]), [0.05, 0.01, 0.0])
    j2 = syn.make_pose(syn.rot_vec([-0.1, 0.25, 0.05]), [0.04, -0.02, 0.01])
    th1, th2 = 0.3, -0.2
    a_t_b = j1 @ syn.make_pose(syn.rot_vec([0, 0, th1]), [0, 0, 0])
    b_t_c = j2 @ syn.make_pose(syn.rot_vec([0, 0, th2]), [0, 0, 0])
    a_t_c = a_t_b @ b_t_c
    c2joint = syn.make_pose(syn.rot_vec([0.1, 0.2, -0.3]), [0.02, 0.03, -0.01])  # the closing joint seen from C
    a2joint = c2joint @ np.linalg.inv(a_t_c)                                      # ... and from A: closed at (th1, th2)
    pose_a = inputs.gt[0][0].copy()
    n_frames = 4
    gt = []
    inputs.color = [[], [], []]
    for k in range(n_frames):
        pose_a = syn.perturb_pose(pose_a, rng, rot_deg=0.7, trans=0.002)
        poses = [pose_a, pose_a @ a_t_b, pose_a @ a_t_c]
        gt.append([p.copy() for p in poses])
        for i in range(3):
            inputs.color[i].append(inputs.scenes[i].render(poses[i]))
    start_a = syn.perturb_pose(gt[0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    states = {}
    for name, api in (("hip", util.open_hip()), ("hip_comm", util.open_hip()), ("oracle", util.open_oracle())):
        if name == "hip_comm":  # the multi-GPU form of the step at world size 1: link sums -> ncclAllReduce ->
            uid = C.create_string_buffer(128)  # links_solve_sums_kernel (constraint rows included)
            api.call("comm_get_unique_id", uid, 128)
            api.call("comm_init_rank", uid, 128, 1, 0)
        rp = dict(syn.RBOT_REGION_PARAMS, measure_occlusions=0)
        models = [host.RegionModel(api, data_points=m[0], orientations=m[1], contour_lengths=m[2])
                  for m in inputs.region_models]
        bodies = [host.Body(api, np.eye(4)) for _ in range(3)]
        cams = [host.ColorCamera(api, **inputs.intr) for _ in range(3)]
        mods = [host.RegionModality(api, bodies[i], cams[i], models[i], **rp) for i in range(3)]
        la = host.Link(api, body=bodies[0])
        lb = host.Link(api, body=bodies[1], parent=la, free_directions=(0, 0, 1, 0, 0, 0),
                       joint2parent_pose=j1 @ syn.make_pose(syn.rot_vec([0, 0, th1 + 0.01]), [0, 0, 0]))
        lc = host.Link(api, body=bodies[2], parent=lb, free_directions=(0, 0, 1, 0, 0, 0),
                       joint2parent_pose=j2 @ syn.make_pose(syn.rot_vec([0, 0, th2 - 0.01]), [0, 0, 0]))
        for link, mod in zip((la, lb, lc), mods):
            link.AddModality(mod)
        opt = host.Optimizer(api, root_link=la)
        host.Constraint(api, opt, la, lc, body12joint1_pose=a2joint, body22joint2_pose=c2joint,
                        constraint_directions=(0, 0, 0, 1, 1, 1))
        tracker = host.Tracker(api, 7, 2)
        bodies[0].set_body2world_pose(start_a)
        assert tracker.CalculateConsistentPoses()
        for i in range(3):
            cams[i].UpdateImage(inputs.color[i][0])
        assert tracker.StartModalities(0)
        out = []
        for k in range(n_frames):
            for i in range(3):
                cams[i].UpdateImage(inputs.color[i][k])
            assert tracker.ExecuteTrackingStep(k)
            out.append([b.body2world_pose() for b in bodies] + [lb.joint2parent_pose(), lc.joint2parent_pose()])
        states[name] = out
        # the loop stays closed: the closing joint's two images coincide in translation
        a, b, c = [x.astype(np.float64) for x in out[-1][:3]]
        gap = (a @ np.linalg.inv(a2joint))[:3, 3] - (c @ np.linalg.inv(c2joint))[:3, 3]
        assert np.max(np.abs(gap)) < 1e-4
        if name == "hip_comm":
            api.call("comm_destroy")
    for sh, so in zip(states["hip"], states["oracle"]):  # no tolerance (round 6: atan2f / tan shared with the oracle)
        for x, y in zip(sh, so):
            assert np.array_equal(x, y)
    for sh, sc in zip(states["hip"], states["hip_comm"]):  # same device arithmetic on both paths
        for x, y in zip(sh, sc):
            assert np.array_equal(x, y)
