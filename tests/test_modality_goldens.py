"""The oracle (and, under -m gpu, the HIP library) against the reference's modality / optimizer /
tracker goldens (test/modality_test.cpp:180-193,222-248,280-316,433-456,486-502,534-550,
test/optimizer_test.cpp:97-105, test/tracker_test.cpp:164-178, test/refiner_test.cpp:96-105).  The
models those goldens were made with are not shipped; they are regenerated without OpenGL by
oracle/gl_model.py (validated against the reference's own model files in
test_model_generation.py) and committed as tests/golden/triangle_views.npz.

Everything except the refiner pose meets the reference's own criteria: gradients and Hessians 1e-3
element-wise relative, the visualisation images pixel for pixel, the tracker pose 1e-5 relative.
The refiner sequence (StartModalities before each of its 7 correspondence searches) is chaotic on
this fixture: moving the start pose by 1e-6 m moves its end pose by 2.5e-3 ... 6e-3
(test_refiner_sequence_amplifies_a_micrometre), so its golden is only approached (7e-3)."""
import numpy as np
import pytest

import golden_scene as gs
import util


def _region(api):
    f = gs.RegionFixture(api)
    assert f.tracker.StartModalities(0)
    assert f.tracker.CalculateCorrespondences(0, 0)
    return f


def check_region_goldens(api):
    f = _region(api)
    for opt_iteration, name in ((0, "global"), (1, "local")):  # RegionModalityTest.Calculate{Global,Local}GradientAndHessian
        assert f.tracker.CalculateGradientAndHessian(0, 0, opt_iteration)
        g, h = f.modality.gradient_hessian()
        gg = gs.golden("region_modality_%s_gradient.txt" % name)[:, 0]
        hg = gs.golden("region_modality_%s_hessian.txt" % name)
        assert np.max(np.abs((g - gg) / gg)) < 1e-3  # the reference's criterion (common_test.cpp:206-226)
        assert np.max(np.abs((h - hg) / hg)) < 1e-3


def check_region_visualisation(api):
    """RegionModalityTest.CalculateCorrespondences: region_modality.png"""
    f = _region(api)
    hf, hb = f.modality.histograms()
    lines = f.modality.data_lines()
    assert len(lines) == 179 and int(lines["valid"].sum()) == 179
    vis = gs.render_lines_visualisation(f.image, hf, hb, lines, n_bins=16, scale=6, distribution_length=12)
    gold = gs.load_png("modality_test/region_modality.png").astype(np.int32)
    diff = np.abs(vis - gold).max(axis=2)
    assert int((diff > 0).sum()) == 0  # CompareLoadedImages(..., 0, 0)


def check_measured_occlusion_goldens(api):
    """RegionModalityTest / DepthModalityTest CalculateCorrespondences[MeasuredOcclusions]
    (test/modality_test.cpp:222-248,433-456,486-502): which lines / points survive the measured
    occlusion test on depth frame 200, read off the visualisation goldens (the grey level of the
    depth background in those files follows 255/999 per raw unit instead of the shipped
    NormalizedDepthImage's 255/1000, camera.cpp:108-115, so only the drawn marks are compared)"""
    f = gs.RegionFixture(api, measure_occlusions=True)
    assert f.tracker.StartModalities(0) and f.tracker.CalculateCorrespondences(0, 0)
    lines = f.modality.data_lines()
    lines = lines[lines["valid"] != 0]
    assert len(lines) == 83  # of 179 without occlusion handling
    hf, hb = f.modality.histograms()
    vis = gs.render_lines_visualisation(f.image, hf, hb, lines, n_bins=16, scale=6, distribution_length=12)
    gold = gs.load_png("modality_test/region_modality_measured_occlusions.png").astype(np.int32)
    assert int((np.abs(vis - gold).max(axis=2) > 0).sum()) == 0
    w2c = np.linalg.inv(gs.mtv.DEPTH_CAMERA2WORLD).astype(np.float32)
    b2dc = (w2c @ gs.mtv.body2world()).astype(np.float32)
    mine = gs.point_mask((480, 848), lines["center_f_body"], b2dc,
                         gs.mtv.DEPTH_INTRINSICS)
    assert np.array_equal(mine, gs.golden_point_mask("modality_test/region_modality_depth_measured_occlusions.png",
                                                     (24, 184, 234)))
    for occlusions, name, n_valid in ((False, "depth_modality.png", 182),
                                      (True, "depth_modality_measured_occlusions.png", 110)):
        d = gs.DepthFixture(api, measure_occlusions=occlusions)
        assert d.tracker.CalculateCorrespondences(0, 0)
        pts = d.modality.data_points()
        pts = pts[pts["valid"] != 0]
        assert len(pts) == n_valid
        mine = gs.point_mask((480, 848), pts["center_f_body"], d.body2camera, gs.mtv.DEPTH_INTRINSICS)
        assert np.array_equal(mine, gs.golden_point_mask("modality_test/" + name, (187, 117, 0)))


def check_depth_goldens(api):
    """DepthModalityTest.CalculateGradientAndHessian, with the reference's own 1e-3 criterion"""
    f = gs.DepthFixture(api)
    assert f.tracker.CalculateCorrespondences(0, 0)
    assert f.tracker.CalculateGradientAndHessian(0, 0, 0)
    g, h = f.modality.gradient_hessian()
    gg, hg = gs.golden("depth_modality_gradient.txt")[:, 0], gs.golden("depth_modality_hessian.txt")
    assert np.max(np.abs((g - gg) / gg)) < 1e-3
    assert np.max(np.abs((h - hg) / hg)) < 1e-3


def check_optimizer_golden_full_chain(api):
    """OptimizerTest.Optimize (test/optimizer_test.cpp:17-41,97-105) without borrowing the golden g/H:
    frames -> correspondences -> g/H of both modalities -> one Tikhonov/LDLT solve -> pose.
    The reference asserts 1e-5 relative on 6-digit values (the smallest entry, -0.0042616, carries 4e-7 of
    print rounding alone); here 2e-5 absolute."""
    f = gs.TrackerFixture(api, measure_occlusions=False, tikhonov_rotation=5000.0, tikhonov_translation=500000.0)
    t = f.tracker
    assert t.StartModalities(0) and t.CalculateCorrespondences(0, 0)
    assert t.CalculateGradientAndHessian(0, 0, 0) and t.CalculateOptimization(0, 0, 0)
    golden = util.read_golden_matrix("optimizer_test/triangle_pose.txt")
    assert np.max(np.abs(f.body.body2world_pose() - golden)) < 2e-5


def check_tracker_and_refiner_goldens(api):
    """TrackerTest.OptimizePoseMatrix (test/tracker_test.cpp:164-178: StartModalities + one
    ExecuteTrackingStep of 7 x 2 iterations, Region + Depth with measured occlusions) with the reference's
    own criterion, and RefinerTest.OptimizePoseMatrix (test/refiner_test.cpp:96-105, refiner.cpp:98-117:
    7 x (StartModalities + correspondences + 3 updates)), which is chaotic on this fixture (see the
    module docstring) and only approached."""
    f = gs.TrackerFixture(api, measure_occlusions=True)
    assert f.tracker.StartModalities(0) and f.tracker.ExecuteTrackingStep(0)
    golden = util.read_golden_matrix("tracker_test/triangle_pose.txt")
    pose = f.body.body2world_pose()
    assert np.max(np.abs((pose - golden)[:3] / golden[:3])) < 1e-5  # CompareToLoadedMatrix(..., 1.0e-5f)

    f = gs.TrackerFixture(api, measure_occlusions=True, n_update_iterations=3)
    assert f.tracker.RefinePoses(7, 3)
    golden = util.read_golden_matrix("refiner_test/triangle_pose.txt")
    start = gs.mtv.body2world()
    pose = f.body.body2world_pose()
    assert np.max(np.abs(pose - golden)) < 1.5e-2
    assert np.max(np.abs(pose - golden)) < 0.5 * np.max(np.abs(start - golden))


def test_refiner_sequence_amplifies_a_micrometre():
    """The evidence behind "chaotic" (refiner.cpp:98-117: every one of the 7 searches re-initialises the histograms
    at the pose reached so far, test/refiner_test.cpp:96-105): the same RefinePoses(7, 3) from start poses that differ
    by ONE MICROMETRE ends on poses that differ by more than a millimetre / milliradian -- an amplification above
    1000, against the 1e-5 criterion of the reference's golden.  The oracle's distance from that golden (7e-3) lies
    inside the spread of these runs: at this sensitivity a golden made with OpenGL-generated models on other hardware
    cannot be reproduced to 1e-5 by any restatement whose model differs in a single contour point."""
    def end_pose(delta):
        f = gs.TrackerFixture(util.open_oracle(), measure_occlusions=True, n_update_iterations=3)
        start = gs.mtv.body2world().copy()
        start[:3, 3] += np.asarray(delta, np.float32)
        f.body.set_body2world_pose(start)
        assert f.tracker.RefinePoses(7, 3)
        return f.body.body2world_pose()

    base = end_pose([0, 0, 0])
    spread = [np.max(np.abs(end_pose(d) - base)) for d in ([1e-6, 0, 0], [0, 1e-6, 0], [0, 0, 1e-6], [-1e-6, 0, 0])]
    assert min(spread) > 1e-3, spread          # every micrometre perturbation: amplified more than 1000 times
    golden = util.read_golden_matrix("refiner_test/triangle_pose.txt")
    residual = np.max(np.abs(base - golden))
    assert residual < 2.0 * max(spread), (residual, spread)


def test_oracle_region_goldens():
    check_region_goldens(util.open_oracle())


def test_oracle_region_visualisation_golden():
    check_region_visualisation(util.open_oracle())


def test_oracle_depth_goldens():
    check_depth_goldens(util.open_oracle())


def test_oracle_measured_occlusion_goldens():
    check_measured_occlusion_goldens(util.open_oracle())


@pytest.mark.gpu
def test_hip_measured_occlusion_goldens():
    check_measured_occlusion_goldens(util.open_hip())


def test_oracle_optimizer_golden_full_chain():
    check_optimizer_golden_full_chain(util.open_oracle())


def test_oracle_tracker_and_refiner_goldens():
    check_tracker_and_refiner_goldens(util.open_oracle())


@pytest.mark.gpu
def test_hip_optimizer_golden_full_chain():
    check_optimizer_golden_full_chain(util.open_hip())


@pytest.mark.gpu
def test_hip_tracker_and_refiner_goldens():
    check_tracker_and_refiner_goldens(util.open_hip())


@pytest.mark.gpu
def test_hip_region_goldens():
    check_region_goldens(util.open_hip())


@pytest.mark.gpu
def test_hip_region_visualisation_golden():
    check_region_visualisation(util.open_hip())


@pytest.mark.gpu
def test_hip_depth_goldens():
    check_depth_goldens(util.open_hip())
