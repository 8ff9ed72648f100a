"""Multi-body rows (Link / Constraint / Optimizer of any dof) of the oracle, pinned by the
property the reference's own script examples/constraint_convergence.cpp:87-131 plots: a fully
constrained two-link system driven only by the constraint converges to zero joint error."""
import numpy as np
import pytest

import util
from util import host, syn


def random_pose(rng):
    ang = rng.uniform(-1, 1) * np.pi
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    t = rng.normal(size=3)
    t = t / np.linalg.norm(t) * rng.uniform(-1, 1)
    R = syn.rot_vec(axis * ang)
    return syn.make_pose(R, R @ t)  # pose.rotate(aa); pose.translate(t)


def build(api, b1, b2, directions=(1, 1, 1, 1, 1, 1)):
    link1 = host.Link(api)
    link2 = host.Link(api, parent=link1, body2joint_pose=b2, joint2parent_pose=np.linalg.inv(b1))
    opt = host.Optimizer(api, root_link=link1)
    host.Constraint(api, opt, link1, link2, body12joint1_pose=b1, body22joint2_pose=b2,
                    constraint_directions=directions)
    return link1, link2, opt


def run_convergence(api, seed, n_iterations=6):
    rng = np.random.default_rng(seed)
    b1, b2 = random_pose(rng), random_pose(rng)
    link1, link2, opt = build(api, b1, b2)
    # perturb the joint by a moderate random pose (the reference script draws the full range)
    d = random_pose(rng)
    d[:3, :3] = syn.rot_vec(rng.normal(size=3) * 0.5)
    link2.set_joint2parent_pose(np.linalg.inv(b1) @ d)
    tracker = host.Tracker(api, 1, 1)
    assert tracker.CalculateConsistentPoses()
    errs, poses = [], []
    for it in range(n_iterations):
        pose_error = b1.astype(np.float64) @ link2.joint2parent_pose().astype(np.float64)
        errs.append(syn.pose_errors(np.eye(4), pose_error))
        assert tracker.CalculateOptimization(0, 0, 0)
        poses.append((link1.link2world_pose(), link2.link2world_pose(), link2.joint2parent_pose()))
    return errs, poses


@pytest.mark.parametrize("seed", range(8))
def test_constraint_convergence(seed):
    errs, _ = run_convergence(util.open_oracle(), seed)
    rot = [e[0] for e in errs]
    trans = [e[1] for e in errs]
    assert rot[0] > 0.05
    assert rot[-1] < 2e-5 and trans[-1] < 2e-5, errs
    assert rot[2] < rot[0] * 0.2


def test_begin_end_equals_single_call():
    """calculate_optimization == begin + end (the multi-GPU split point), bit for bit"""
    poses = []
    for split in (False, True):
        rng = np.random.default_rng(3)
        api = util.open_oracle()
        b1, b2 = random_pose(rng), random_pose(rng)
        link1, link2, opt = build(api, b1, b2, directions=(1, 1, 0, 1, 1, 1))
        d = random_pose(rng)
        d[:3, :3] = syn.rot_vec(rng.normal(size=3) * 0.3)
        link2.set_joint2parent_pose(np.linalg.inv(b1) @ d)
        tracker = host.Tracker(api, 1, 1)
        tracker.CalculateConsistentPoses()
        for it in range(3):
            if split:
                ptr, n = tracker.CalculateOptimizationBegin()
                assert n == 2 * 42  # the link sums of both links
                assert tracker.CalculateOptimizationEnd()
            else:
                assert tracker.CalculateOptimization(0, 0, 0)
        poses.append((link1.link2world_pose(), link2.link2world_pose(), link2.joint2parent_pose()))
    for a, b in zip(*poses):
        assert np.array_equal(a, b)


# ---- soft constraints (soft_constraint.cpp) ------------------------------------------------------
def build_soft(api, b1, b2, directions=(1, 1, 1, 1, 1, 1), root_free=True, **kw):
    link1 = host.Link(api, free_directions=(1,) * 6 if root_free else (0,) * 6)
    link2 = host.Link(api, parent=link1, body2joint_pose=b2, joint2parent_pose=np.linalg.inv(b1))
    opt = host.Optimizer(api, root_link=link1)
    host.SoftConstraint(api, opt, link1, link2, body12joint1_pose=b1, body22joint2_pose=b2,
                        constraint_directions=directions, **kw)
    return link1, link2, opt


def run_soft(api, seed, n_iterations, rot_scale=0.3, **kw):
    rng = np.random.default_rng(seed)
    b1, b2 = random_pose(rng), random_pose(rng)
    link1, link2, opt = build_soft(api, b1, b2, **kw)
    d = random_pose(rng)
    d[:3, :3] = syn.rot_vec(rng.normal(size=3) * rot_scale)
    d[:3, 3] *= 0.05
    link2.set_joint2parent_pose(np.linalg.inv(b1) @ d)
    tracker = host.Tracker(api, 1, 1)
    assert tracker.CalculateConsistentPoses()
    errs, poses = [], []
    for it in range(n_iterations):
        pose_error = b1.astype(np.float64) @ link2.joint2parent_pose().astype(np.float64)
        errs.append(syn.pose_errors(np.eye(4), pose_error))
        assert tracker.CalculateOptimization(0, 0, 0)
        poses.append((link1.link2world_pose(), link2.link2world_pose(), link2.joint2parent_pose()))
    return errs, poses


@pytest.mark.parametrize("seed", range(4))
def test_soft_constraint_pulls_the_joint_together(seed):
    """with max_distance 0 a soft constraint is a (Tikhonov-damped) spring towards the joint: the error
    of a structure driven by nothing else decays monotonically to zero"""
    errs, _ = run_soft(util.open_oracle(), seed, 60)
    rot = np.asarray([e[0] for e in errs])
    trans = np.asarray([e[1] for e in errs])
    assert rot[0] > 0.05
    assert np.all(np.diff(rot) < 1e-6) and np.all(np.diff(trans) < 1e-6)
    assert rot[-1] < 1e-3 * rot[0] and trans[-1] < 1e-2 * max(trans[0], 1e-3)


def test_soft_constraint_dead_zone():
    """inside max_distance nothing moves; outside, the joint is pulled back to the edge of the zone"""
    inside, poses_inside = run_soft(util.open_oracle(), 1, 3, rot_scale=0.05, max_distance_rotation=1.0,
                                    max_distance_translation=1.0)
    assert inside[0][0] > 0.01
    assert np.array_equal(poses_inside[0][2], poses_inside[-1][2])
    # (root fixed: with a free root the reference's per-link Hessians, which carry no link1-link2
    # cross term, over-shoot into the zone -- restated as is)
    edge, _ = run_soft(util.open_oracle(), 1, 60, rot_scale=0.2, directions=(1, 1, 1, 0, 0, 0),
                       max_distance_rotation=0.2, root_free=False)
    assert edge[0][0] > 0.3
    assert abs(edge[-1][0] - 0.2) < 2e-3
