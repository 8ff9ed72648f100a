"""The device path N > 1 ranks of a kinematic structure really take -- tracking_step_tree_segment_kernel, one launch and one
reduction of the link sums per Newton step (optimizer.cpp:281-346, link.cpp:159-241 being the sum that is distributed)
-- run with PARTIAL ownership on one GPU: N contexts, one host thread each, body i's modality in context i mod N, the
collective supplied through m3t_hip_comm_set_reduce_callback (sharding.ThreadRanks: the threads meet at a barrier and
add their buffers in an order of the harness's own).  Every context must end on the oracle's single-process poses bit
for bit.  World size 1 (all earlier evidence for this kernel) cannot show a wrong link table hand-over, a link whose
sums come from another rank being skipped, or the copy-back after an odd number of Newton steps."""
import ctypes as C

import numpy as np
import pytest

import scenes
import util
from util import host, syn

gpu = pytest.mark.gpu
sharding = util.pkg.sharding


def step_kernel(api):
    name = C.create_string_buffer(96)
    api.call("get_step_kernel", name, 96)
    return name.value.decode()


def allreduce_count(api):
    n = C.c_longlong(-1)
    api.call("comm_get_allreduce_count", C.byref(n))
    return n.value


def chain_state(ch):
    return np.stack([b.body2world_pose() for b in ch.bodies] + [l.joint2parent_pose() for l in ch.links[1:]])


def run_chain_over_threads(world, n_bodies, n_frames, n_corr, n_update, order=None):
    """the open chain of bench.py --config chain8 over `world` contexts; returns per-rank states per frame + oracle's"""
    import bench_chain as bc
    inputs, joints, gt = bc.chain_inputs(scenes, syn, n_bodies, n_frames, 2)
    start_root = syn.perturb_pose(gt[0][0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    start_angles = gt[0][1] + 0.01
    placed = sharding.place_bodies(n_bodies, world)

    def build(api, owned):
        ch = bc.Chain(api, host, syn, inputs, joints, start_root, start_angles, owned)
        ch.tracker = host.Tracker(api, n_corr, n_update)
        return ch

    oc = build(util.open_oracle(), range(n_bodies))
    oc.upload(inputs, 0)
    assert oc.tracker.StartModalities(0)
    expected = []
    for k in range(n_frames):
        oc.upload(inputs, k)
        assert oc.tracker.ExecuteTrackingStep(k)
        expected.append(chain_state(oc))

    apis = [util.open_hip() for _ in range(world)]
    chains = [build(apis[r], [i for i, p in enumerate(placed) if p == r]) for r in range(world)]
    ranks = sharding.ThreadRanks(apis, order=order)

    def work(rank, api):
        ch = chains[rank]
        ch.upload(inputs, 0)
        assert ch.tracker.StartModalities(0)
        out = []
        for k in range(n_frames):
            ch.upload(inputs, k)
            assert ch.tracker.ExecuteTrackingStep(k)
            out.append(chain_state(ch))
        return out

    got = ranks.run(work)
    assert not ranks.errors, ranks.errors
    kernels = [step_kernel(a) for a in apis]
    counts = [allreduce_count(a) for a in apis]
    ranks.close()
    return got, expected, kernels, counts, gt


@gpu
@pytest.mark.parametrize("world", [2, 4])
def test_eight_body_chain_over_thread_ranks_takes_the_segment_kernel(world):
    """BASELINE configs[4] as `world` GPUs hold it (body i in context i mod world), 4 frames of 7 x 2 Newton steps:
    every context runs tracking_step_tree_segment_kernel with 8 / world workgroups, reduces 14 times per frame, and
    ends every frame on the oracle's poses (8 body2world + 7 joint2parent) bit for bit"""
    n_frames = 4
    got, expected, kernels, counts, gt = run_chain_over_threads(world, 8, n_frames, 7, 2,
                                                                order=[1, 0] if world == 2 else [3, 1, 0, 2])
    assert kernels == ["tracking_step_tree_segment_kernel"] * world, kernels
    assert counts == [14 * n_frames] * world, counts
    for rank in range(world):
        for k in range(n_frames):
            assert np.array_equal(got[rank][k], expected[k]), (rank, k)
    for i in range(8):  # and the chain is tracked (rbot_evaluator.cpp:416-433: 5 cm / 5 degrees)
        e = syn.pose_errors(expected[-1][i], gt[-1][0][i])
        assert e[0] < np.deg2rad(5) and e[1] < 0.05


@gpu
@pytest.mark.parametrize("n_corr,n_update", [(7, 1), (3, 3), (1, 1)])
def test_odd_numbers_of_newton_steps_over_thread_ranks(n_corr, n_update):
    """an odd n_corr x n_update ends the frame in the SECOND link table (the copy-back branch); 1 x 1 puts the seeding
    of the links from the bodies and the bodies' write-back into one launch (seeded from a snapshot).  Four bodies over
    two contexts, three frames"""
    n_frames = 3
    got, expected, kernels, counts, _ = run_chain_over_threads(2, 4, n_frames, n_corr, n_update)
    assert kernels == ["tracking_step_tree_segment_kernel"] * 2, kernels
    assert counts == [n_corr * n_update * n_frames] * 2, counts
    for rank in range(2):
        for k in range(n_frames):
            assert np.array_equal(got[rank][k], expected[k]), (rank, k)


@gpu
def test_a_rank_without_a_body_of_the_structure_still_solves_it():
    """three bodies over four contexts: the last one holds the link tree and no modality at all -- no workgroup could
    apply the summed link sums in the segment kernel, so that context takes the per-sub-step launches (same number of
    reductions) and still ends on the oracle's poses"""
    got, expected, kernels, counts, _ = run_chain_over_threads(4, 3, 2, 7, 2)
    assert kernels[:3] == ["tracking_step_tree_segment_kernel"] * 3 and kernels[3] == "", kernels
    assert counts == [14 * 2] * 4, counts
    for rank in range(4):
        for k in range(2):
            assert np.array_equal(got[rank][k], expected[k]), (rank, k)


def closed_chain(api, inputs, owned, j1, j2, th1, th2, a2joint, c2joint, start_a, soft):
    """A -- revolute -- B -- revolute -- C, a Constraint (constraint.cpp:81-102) or SoftConstraint
    (soft_constraint.cpp:220-351) tying a point of C back to A; RegionModalities of the bodies in `owned` only"""
    rp = dict(syn.RBOT_REGION_PARAMS, measure_occlusions=0)
    models = [host.RegionModel(api, data_points=m[0], orientations=m[1], contour_lengths=m[2]) for m in inputs.region_models]
    bodies = [host.Body(api, np.eye(4)) for _ in range(3)]
    cams = [host.ColorCamera(api, **inputs.intr) for _ in range(3)]
    la = host.Link(api, body=bodies[0])
    lb = host.Link(api, body=bodies[1], parent=la, free_directions=(0, 0, 1, 0, 0, 0),
                   joint2parent_pose=j1 @ syn.make_pose(syn.rot_vec([0, 0, th1 + 0.01]), [0, 0, 0]))
    lc = host.Link(api, body=bodies[2], parent=lb, free_directions=(0, 0, 1, 0, 0, 0),
                   joint2parent_pose=j2 @ syn.make_pose(syn.rot_vec([0, 0, th2 - 0.01]), [0, 0, 0]))
    links = (la, lb, lc)
    for i in owned:
        links[i].AddModality(host.RegionModality(api, bodies[i], cams[i], models[i], **rp))
    opt = host.Optimizer(api, root_link=la)
    if soft:
        host.SoftConstraint(api, opt, la, lc, body12joint1_pose=a2joint, body22joint2_pose=c2joint,
                            constraint_directions=(0, 0, 0, 1, 1, 1), max_distance_translation=0.0005,
                            standard_deviation_translation=0.002)
    else:
        host.Constraint(api, opt, la, lc, body12joint1_pose=a2joint, body22joint2_pose=c2joint,
                        constraint_directions=(0, 0, 0, 1, 1, 1))
    tracker = host.Tracker(api, 7, 2)
    bodies[0].set_body2world_pose(start_a)
    assert tracker.CalculateConsistentPoses()

    class S:
        pass
    s = S()
    s.bodies, s.cams, s.links, s.tracker, s.owned = bodies, cams, links, tracker, list(owned)
    return s


@gpu
@pytest.mark.parametrize("soft", [False, True], ids=["hard", "soft"])
@pytest.mark.parametrize("world", [2, 3])
def test_closed_chain_over_thread_ranks_equals_one_context(world, soft):
    """the _constrained_ twin of the segment kernel with partial ownership: the closed three-body loop (constraint rows
    in the system / soft-constraint terms onto the link sums AFTER the reduction: counted once) held by 2 and 3
    contexts.  Same device arithmetic as one context that owns everything and takes the one-launch constrained kernel:
    compared bit for bit (against the oracle the constrained structures carry the atan2f / tan tolerance of
    test_closed_chain_with_a_hard_constraint_matches_the_oracle, so the single context is the yardstick here)"""
    inputs = scenes.Inputs(3, 1, n_divides=2)
    rng = np.random.default_rng(3)
    j1 = syn.make_pose(syn.rot_vec([0.2, -0.1, 0.3]), [0.05, 0.01, 0.0])
    j2 = syn.make_pose(syn.rot_vec([-0.1, 0.25, 0.05]), [0.04, -0.02, 0.01])
    th1, th2 = 0.3, -0.2
    a_t_b = j1 @ syn.make_pose(syn.rot_vec([0, 0, th1]), [0, 0, 0])
    a_t_c = a_t_b @ j2 @ syn.make_pose(syn.rot_vec([0, 0, th2]), [0, 0, 0])
    c2joint = syn.make_pose(syn.rot_vec([0.1, 0.2, -0.3]), [0.02, 0.03, -0.01])
    a2joint = c2joint @ np.linalg.inv(a_t_c)
    pose_a = inputs.gt[0][0].copy()
    n_frames = 3
    frames = [[], [], []]
    for k in range(n_frames):
        pose_a = syn.perturb_pose(pose_a, rng, rot_deg=0.7, trans=0.002)
        poses = [pose_a, pose_a @ a_t_b, pose_a @ a_t_c]
        if k == 0:
            first = pose_a.copy()
        for i in range(3):
            frames[i].append(inputs.scenes[i].render(poses[i]))
    start_a = syn.perturb_pose(first, np.random.default_rng(5), rot_deg=0.5, trans=0.001)

    def track(s):
        for i in s.owned:
            s.cams[i].UpdateImage(frames[i][0])
        assert s.tracker.StartModalities(0)
        out = []
        for k in range(n_frames):
            for i in s.owned:
                s.cams[i].UpdateImage(frames[i][k])
            assert s.tracker.ExecuteTrackingStep(k)
            out.append(np.stack([b.body2world_pose() for b in s.bodies] + [l.joint2parent_pose() for l in s.links[1:]]))
        return out

    one_api = util.open_hip()
    one = track(closed_chain(one_api, inputs, range(3), j1, j2, th1, th2, a2joint, c2joint, start_a, soft))
    assert step_kernel(one_api) == "tracking_step_tree_constrained_kernel"
    apis = [util.open_hip() for _ in range(world)]
    structs = [closed_chain(apis[r], inputs, [i for i in range(3) if i % world == r], j1, j2, th1, th2, a2joint, c2joint,
                            start_a, soft) for r in range(world)]
    ranks = sharding.ThreadRanks(apis)
    got = ranks.run(lambda rank, api: track(structs[rank]))
    assert not ranks.errors, ranks.errors
    assert [step_kernel(a) for a in apis] == ["tracking_step_tree_segment_constrained_kernel"] * world
    assert [allreduce_count(a) for a in apis] == [14 * n_frames] * world
    ranks.close()
    for rank in range(world):
        for k in range(n_frames):
            assert np.array_equal(got[rank][k], one[k]), (rank, k)
    if not soft:  # the loop stays closed
        a, _, c = [x.astype(np.float64) for x in one[-1][:3]]
        gap = (a @ np.linalg.inv(a2joint))[:3, 3] - (c @ np.linalg.inv(c2joint))[:3, 3]
        assert np.max(np.abs(gap)) < 1e-4


@gpu
def test_reduce_callback_failure_is_reported():
    """a callback that returns non-zero: the step fails with M3T_ERR_DEVICE and names the code"""
    import bench_chain as bc
    inputs, joints, gt = bc.chain_inputs(scenes, syn, 2, 1, 2)
    api = util.open_hip()
    ch = bc.Chain(api, host, syn, inputs, joints, gt[0][0][0], gt[0][1], range(2))
    fn = util.pkg._capi.REDUCE_FN(lambda user, buf, count, stream: 7)
    api.call("comm_set_reduce_callback", C.cast(fn, C.c_void_p), None)
    ch.upload(inputs, 0)
    assert ch.tracker.StartModalities(0)
    rc = api.raw("execute_tracking_step", 0)
    assert rc == util.pkg._capi.M3T_ERR_DEVICE
    assert "code 7" in api.last_error()
    api.call("comm_set_reduce_callback", None, None)
    api.call("sync")
    assert ch.tracker.StartModalities(0)
    assert ch.tracker.ExecuteTrackingStep(0)  # without the callback: the one-launch step again
    assert step_kernel(api).startswith("tracking_step_tree")
