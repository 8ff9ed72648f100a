"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on
the same seeded inputs.  Stated tolerance: NONE.  Every quantity the path
produces -- view index, histograms, line / point state, gradient, Hessian, pose -- is compared
bit for bit, sub-step by sub-step and over free-running sequences, in every launch shape
(one workgroup per object, several workgroups per object, 256- and 512-thread workgroups,
unfused sub-step kernels): the device takes the g/H sums in the reference's order.
The only toleranced comparisons left are against the reference's own golden files
(6-digit text), with the reference's own criteria."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
import util
from util import host, syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs():
    return util.open_hip, util.open_oracle


def rel_fro(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_extension_loaded_and_device():
    hip = util.open_hip()
    import ctypes as C
    name = C.create_string_buffer(128)
    cus, mem = C.c_int(), C.c_size_t()
    hip.call("device_info", name, 128, C.byref(cus), C.byref(mem))
    assert b"gfx950" in name.value, name.value
    assert cus.value >= 200 and mem.value > 200e9


def test_closest_view_bit_exact():
    hip, ora = util.open_hip(), util.open_oracle()
    body = syn.Ellipsoid([0.08, 0.05, 0.03])
    dp, ori, cl = syn.make_region_model(body, n_divides=4, n_points=4)
    mh = host.RegionModel(hip, data_points=dp, orientations=ori, contour_lengths=cl)
    mo = host.RegionModel(ora, data_points=dp, orientations=ori, contour_lengths=cl)
    assert mh.n_views == 2562
    rng = np.random.default_rng(0)
    for _ in range(200):
        T = syn.make_pose(syn.random_rotation(rng), rng.normal(size=3))
        assert mh.GetClosestView(T) == mo.GetClosestView(T)
    # exact ties: orientation aligned with a vertex shared by symmetric neighbours; first max wins
    for v in (0, 5, 11):
        T = syn.make_pose(np.eye(3), ori[v].astype(np.float64))
        assert mh.GetClosestView(T) == mo.GetClosestView(T)
    assert mh.GetClosestView(np.eye(4)) == 0  # zero translation -> view 0
    path = os.path.join(util.GOLDEN, "model_test", "region_model.bin")
    gh, go = host.RegionModel(hip, path=path), host.RegionModel(ora, path=path)
    assert (gh.n_views, gh.n_points, gh.max_contour_length) == (go.n_views, go.n_points, go.max_contour_length)
    assert gh.GetClosestView(util.inv_pose_f32(util.SCHAUMA_WORLD2BODY)) == 128


def _pair(n_objects=3, n_frames=3, **kw):
    hip, ora = util.open_hip(), util.open_oracle()
    hip.call("set_fused_step", 0)
    inputs = scenes.Inputs(n_objects, n_frames, n_divides=2, **kw)
    return hip, ora, inputs


def _assert_lines_equal(la, lb):
    assert len(la) == len(lb)
    assert np.array_equal(la["model_point_index"], lb["model_point_index"])
    for f in ("center_f_body", "center_u", "center_v", "normal_u", "normal_v", "delta_r",
              "normal_component_to_scale", "continuous_distance", "distribution", "mean", "measured_variance"):
        assert np.array_equal(la[f], lb[f]), f


def test_region_substeps_rbot():
    """StartModality histograms, per-line state of every correspondence iteration, g/H and the
    pose after each optimisation, sub-step by sub-step (RBOT parameters, 32 bins)."""
    hip, ora, inputs = _pair()
    a = scenes.Instance(hip, inputs)
    b = scenes.Instance(ora, inputs)
    a.upload_frame(0)
    b.upload_frame(0)
    assert a.tracker.StartModalities(0) and b.tracker.StartModalities(0)
    for ra, rb in zip(a.region, b.region):
        fa, ba = ra.histograms()
        fb, bb = rb.histograms()
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)
        assert abs(fa.sum() - 1.0) < 1e-5
    for it in range(2):
        a.upload_frame(it)
        b.upload_frame(it)
        for c in range(7):
            a.set_poses(b.poses())  # keep discrete decisions comparable
            assert a.tracker.CalculateCorrespondences(it, c) and b.tracker.CalculateCorrespondences(it, c)
            for ra, rb in zip(a.region, b.region):
                _assert_lines_equal(ra.data_lines(), rb.data_lines())
                assert len(ra.data_lines()) > 150
            for u in range(2):
                a.set_poses(b.poses())
                assert a.tracker.CalculateGradientAndHessian(it, c, u)
                assert b.tracker.CalculateGradientAndHessian(it, c, u)
                for ra, rb in zip(a.region, b.region):
                    ga, ha = ra.gradient_hessian()
                    gb, hb = rb.gradient_hessian()
                    assert np.array_equal(ga, gb) and np.array_equal(ha, hb)
                    assert np.array_equal(ha, ha.T)
                assert a.tracker.CalculateOptimization(it, c, u) and b.tracker.CalculateOptimization(it, c, u)
                assert np.array_equal(np.stack(a.poses()), np.stack(b.poses())), (it, c, u)
        assert a.tracker.CalculateResults(it) and b.tracker.CalculateResults(it)
        for ra, rb in zip(a.region, b.region):
            ra_f, ra_b = ra.histograms()
            assert np.array_equal(ra_f, rb.histograms()[0]) and np.array_equal(ra_b, rb.histograms()[1])


def test_region_histogram_update_bit_exact():
    """CalculateResults on identical poses: counts, sums and the exponential blend are exact."""
    hip, ora, inputs = _pair(n_objects=2, n_frames=2)
    for bins in (8, 16, 32, 64):
        rp = dict(syn.RBOT_REGION_PARAMS, n_histogram_bins=bins)
        a = scenes.Instance(hip, inputs, region_params=rp)
        b = scenes.Instance(ora, inputs, region_params=rp)
        a.upload_frame(0)
        b.upload_frame(0)
        assert a.tracker.StartModalities(0) and b.tracker.StartModalities(0)
        a.upload_frame(1)
        b.upload_frame(1)
        a.set_poses(inputs.gt[i][1] for i in range(2))
        b.set_poses(inputs.gt[i][1] for i in range(2))
        assert a.tracker.CalculateResults(1) and b.tracker.CalculateResults(1)
        for ra, rb in zip(a.region, b.region):
            fa, ba = ra.histograms()
            fb, bb = rb.histograms()
            assert np.array_equal(fa, fb) and np.array_equal(ba, bb), bins
        hip = util.open_hip()
        hip.call("set_fused_step", 0)
        ora = util.open_oracle()


def test_tracking_step_fused_modes_bit_exact():
    """ExecuteTrackingStep, state re-synchronised per frame: the fused and the unfused device
    paths give the oracle's poses bit for bit; ADD-S as ycb_evaluator.cpp:816 is therefore 0."""
    inputs = scenes.Inputs(4, 4, n_divides=2)
    for mode in (0, 1, 2):
        hip, ora = util.open_hip(), util.open_oracle()
        hip.call("set_fused_step", mode)
        a = scenes.Instance(hip, inputs)
        b = scenes.Instance(ora, inputs)
        a.upload_frame(0)
        b.upload_frame(0)
        assert a.tracker.StartModalities(0) and b.tracker.StartModalities(0)
        for k in range(inputs.n_frames):
            a.upload_frame(k)
            b.upload_frame(k)
            a.set_poses(b.poses())
            for ra, rb in zip(a.region, b.region):
                ra.set_histograms(*rb.histograms())
            assert a.tracker.ExecuteTrackingStep(k) and b.tracker.ExecuteTrackingStep(k)
            pa, pb = a.poses(), b.poses()
            assert np.array_equal(np.stack(pa), np.stack(pb)), (mode, k)
            assert max(syn.add_s(inputs.vertices[i], pa[i], pb[i]) for i in range(inputs.n_objects)) == 0.0
            if mode == 1:
                assert hip.raw("region_modality_get_lines", a.region[0].id, None, 0, None) == -2
            if mode == 2:  # line state and g/H of the last iteration written back by the fused kernel
                for ra, rb in zip(a.region, b.region):
                    _assert_lines_equal(ra.data_lines(), rb.data_lines())
                    ga, ha = ra.gradient_hessian()
                    gb, hb = rb.gradient_hessian()
                    assert np.array_equal(ga, gb) and np.array_equal(ha, hb)


def test_tracking_free_running_50_frames_bit_exact():
    """no re-synchronisation for 50 frames: poses, histograms and therefore the RBOT success
    criterion (rbot_evaluator.cpp:416-433) are the oracle's, bit for bit, fused and unfused."""
    inputs = scenes.Inputs(6, 50, n_divides=2)
    ora = util.open_oracle()
    b = scenes.Instance(ora, inputs)
    b.upload_frame(0)
    assert b.tracker.StartModalities(0)
    ref = []
    for k in range(inputs.n_frames):
        b.upload_frame(k)
        assert b.tracker.ExecuteTrackingStep(k)
        ref.append(np.stack(b.poses()))
    ref_hist = [r.histograms() for r in b.region]
    tracked = 0
    for i in range(inputs.n_objects):
        e = syn.pose_errors(ref[-1][i], inputs.gt[i][-1])
        tracked += int(e[0] < np.deg2rad(5) and e[1] < 0.05)
    assert tracked == inputs.n_objects  # the sequence is a meaningful one: every object is still tracked
    for mode in (1, 0):
        hip = util.open_hip()
        hip.call("set_fused_step", mode)
        a = scenes.Instance(hip, inputs)
        a.upload_frame(0)
        assert a.tracker.StartModalities(0)
        for k in range(inputs.n_frames if mode else 12):
            a.upload_frame(k)
            assert a.tracker.ExecuteTrackingStep(k)
            assert np.array_equal(np.stack(a.poses()), ref[k]), (mode, k)
        if mode:
            for ra, (hf, hb) in zip(a.region, ref_hist):
                fa, ba = ra.histograms()
                assert np.array_equal(fa, hf) and np.array_equal(ba, hb)


@pytest.mark.parametrize("threads,parts", [(512, 0), (256, 0), (128, 0), (512, 4), (512, 8), (512, 16), (256, 8)])
def test_region_depth_sequence_bit_exact(threads, parts, monkeypatch):
    """Region + Depth (YCB parameters, measured occlusions), free running, in the launch shapes the library
    uses: 512-thread workgroups, the 256-thread workgroups of large batches (two per CU), and several
    workgroups per object (tracking_step_split_kernel: lines and points divided over 4 / 8 / 16 workgroups,
    their results exchanged once per correspondence iteration)."""
    monkeypatch.setenv("M3T_HIP_THREADS", str(threads))
    if parts:
        monkeypatch.setenv("M3T_HIP_SPLIT_PARTS", str(parts))
    else:
        monkeypatch.setenv("M3T_HIP_NO_SPLIT", "1")
    inputs = scenes.Inputs(3, 12, n_divides=2, with_depth=True)
    ora = util.open_oracle()
    b = scenes.Instance(ora, inputs, use_depth=True)
    b.upload_frame(0)
    assert b.tracker.StartModalities(0)
    hip = util.open_hip()
    a = scenes.Instance(hip, inputs, use_depth=True)
    a.upload_frame(0)
    assert a.tracker.StartModalities(0)
    for k in range(inputs.n_frames):
        a.upload_frame(k)
        b.upload_frame(k)
        assert a.tracker.ExecuteTrackingStep(k) and b.tracker.ExecuteTrackingStep(k)
        assert np.array_equal(np.stack(a.poses()), np.stack(b.poses())), k
    shape = (C.c_int * 4)()
    hip.call("get_step_shape", shape)
    assert list(shape)[:3] == [3, parts or 1, threads]
    for ra, rb in zip(a.region, b.region):
        assert np.array_equal(ra.histograms()[0], rb.histograms()[0])
        assert np.array_equal(ra.histograms()[1], rb.histograms()[1])


def test_optimizer_golden_on_device():
    """OptimizerTest.Optimize golden through the device solve (rigid_optimize_kernel)."""
    hip = util.open_hip()
    cam = host.ColorCamera(hip, **util.COLOR_INTR)
    dcam = host.DepthCamera(hip, depth_scale=0.001, **util.DEPTH_INTR)
    rm = host.RegionModel(hip, path=os.path.join(util.GOLDEN, "model_test", "region_model.bin"))
    dm = host.DepthModel(hip, path=os.path.join(util.GOLDEN, "model_test", "depth_model.bin"))
    body = host.Body(hip, util.inv_pose_f32(util.TRIANGLE_WORLD2BODY))
    region = host.RegionModality(hip, body, cam, rm, n_lines_max=10)
    depth = host.DepthModality(hip, body, dcam, dm, n_points_max=10)
    region.set_gradient_hessian(util.read_golden_matrix("modality_test/region_modality_global_gradient.txt")[:, 0],
                                util.read_golden_matrix("modality_test/region_modality_global_hessian.txt"))
    depth.set_gradient_hessian(util.read_golden_matrix("modality_test/depth_modality_gradient.txt")[:, 0],
                               util.read_golden_matrix("modality_test/depth_modality_hessian.txt"))
    host.Optimizer(hip, body=body, modalities=[region, depth], tikhonov_parameter_rotation=5000.0,
                   tikhonov_parameter_translation=500000.0)
    tracker = host.Tracker(hip, 1, 1)
    assert tracker.CalculateOptimization(0, 0, 0)
    golden = util.read_golden_matrix("optimizer_test/triangle_pose.txt")
    assert np.max(np.abs(body.body2world_pose() - golden)) < 1e-5


def test_nan_guard_keeps_pose():
    """NaN solve -> pose untouched, step still succeeds (optimizer.cpp:165-166)."""
    for api in (util.open_hip(), util.open_oracle()):
        cam = host.ColorCamera(api, **util.COLOR_INTR)
        rm = host.RegionModel(api, path=os.path.join(util.GOLDEN, "model_test", "region_model.bin"))
        start = util.inv_pose_f32(util.TRIANGLE_WORLD2BODY)
        body = host.Body(api, start)
        region = host.RegionModality(api, body, cam, rm, n_lines_max=10)
        g = np.full(6, np.nan, np.float32)
        region.set_gradient_hessian(g, -np.eye(6, dtype=np.float32))
        host.Optimizer(api, body=body, modalities=[region])
        assert host.Tracker(api, 1, 1).CalculateOptimization(0, 0, 0)
        assert np.array_equal(body.body2world_pose(), start)


def test_real_fixture_frames():
    """schauma bottle on the reference's own frames 200/201 (960x540) with the golden
    region_model.bin (10 lines), unit-test default parameters: lines bit-exact, pose parity."""
    hip, ora = util.open_hip(), util.open_oracle()
    hip.call("set_fused_step", 0)
    out = []
    for api in (hip, ora):
        cam = host.ColorCamera(api, **util.COLOR_INTR)
        rm = host.RegionModel(api, path=os.path.join(util.GOLDEN, "model_test", "region_model.bin"))
        body = host.Body(api, util.inv_pose_f32(util.SCHAUMA_WORLD2BODY))
        mod = host.RegionModality(api, body, cam, rm, n_lines_max=10, n_unoccluded_iterations=0)
        host.Optimizer(api, body=body, modalities=[mod])
        tr = host.Tracker(api, 7, 2)
        cam.UpdateImage(util.load_color_frame(200))
        assert tr.StartModalities(0)
        assert tr.CalculateCorrespondences(0, 0)
        lines = mod.data_lines().copy()
        assert tr.ExecuteTrackingStep(0)
        cam.UpdateImage(util.load_color_frame(201))
        assert tr.ExecuteTrackingStep(1)
        out.append((lines, body.body2world_pose(), mod.histograms()))
    _assert_lines_equal(out[0][0], out[1][0])
    assert len(out[0][0]) == 10
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][2][0], out[1][2][0]) and np.array_equal(out[0][2][1], out[1][2][1])


def test_depth_substeps_and_fused_region_depth():
    """DepthModality (YCB parameters, measured occlusions on both modalities): point state, g/H and
    poses bit-exact sub-step by sub-step; Region+Depth fused and unfused steps bit-exact."""
    hip, ora = util.open_hip(), util.open_oracle()
    hip.call("set_fused_step", 0)
    inputs = scenes.Inputs(3, 3, n_divides=2, with_depth=True)
    a = scenes.Instance(hip, inputs, use_depth=True)
    b = scenes.Instance(ora, inputs, use_depth=True)
    a.upload_frame(0)
    b.upload_frame(0)
    assert a.tracker.StartModalities(0) and b.tracker.StartModalities(0)
    for ra, rb in zip(a.region, b.region):
        assert np.array_equal(ra.histograms()[0], rb.histograms()[0])
    for c in range(4):
        a.set_poses(b.poses())
        assert a.tracker.CalculateCorrespondences(0, c) and b.tracker.CalculateCorrespondences(0, c)
        for da, db in zip(a.depth, b.depth):
            pa, pb = da.data_points(), db.data_points()
            assert len(pa) == len(pb) and len(pa) > 100
            for f in ("model_point_index", "center_u", "center_v", "depth", "correspondence_center_f_camera"):
                assert np.array_equal(pa[f], pb[f]), f
        for ra, rb in zip(a.region, b.region):
            _assert_lines_equal(ra.data_lines(), rb.data_lines())
        for u in range(2):
            a.set_poses(b.poses())
            assert a.tracker.CalculateGradientAndHessian(0, c, u) and b.tracker.CalculateGradientAndHessian(0, c, u)
            for da, db in zip(a.depth, b.depth):
                ga, ha = da.gradient_hessian()
                gb, hb = db.gradient_hessian()
                assert np.array_equal(ga, gb) and np.array_equal(ha, hb)
            assert a.tracker.CalculateOptimization(0, c, u) and b.tracker.CalculateOptimization(0, c, u)
            assert np.array_equal(np.stack(a.poses()), np.stack(b.poses())), (c, u)
    # fused Region+Depth step
    for mode in (1, 0):
        hip2, ora2 = util.open_hip(), util.open_oracle()
        hip2.call("set_fused_step", mode)
        worst = scenes.run_region_parity(hip2, ora2, inputs=inputs, use_depth=True)
        assert worst == (0.0, 0.0), (mode, worst)


def test_depth_only_and_occluder():
    """Depth-only tracking + an occluding plane in front of half of the object: the
    measured-occlusion window test rejects the same points / lines on both sides."""
    hip, ora = util.open_hip(), util.open_oracle()
    hip.call("set_fused_step", 0)
    inputs = scenes.Inputs(2, 2, n_divides=2, with_depth=True)
    for i in range(2):
        for k in range(2):
            d = inputs.depth[i][k]
            W = d.shape[1]
            col = int(inputs.gt[i][k][0, 3] * inputs.intr["fu"] / inputs.gt[i][k][2, 3] + inputs.intr["ppu"])
            col = min(max(col, 0), W - 1)
            d[:, col:] = np.minimum(d[:, col:], np.uint16(0.3 / inputs.depth_scale))  # occluder at 0.3 m
    a = scenes.Instance(hip, inputs, use_region=True, use_depth=True)
    b = scenes.Instance(ora, inputs, use_region=True, use_depth=True)
    a.upload_frame(0)
    b.upload_frame(0)
    assert a.tracker.StartModalities(0) and b.tracker.StartModalities(0)
    for ra, rb in zip(a.region, b.region):
        assert np.array_equal(ra.histograms()[0], rb.histograms()[0])
        assert np.array_equal(ra.histograms()[1], rb.histograms()[1])
    assert a.tracker.CalculateCorrespondences(0, 0) and b.tracker.CalculateCorrespondences(0, 0)
    for ra, rb, da, db in zip(a.region, b.region, a.depth, b.depth):
        la, lb = ra.data_lines(), rb.data_lines()
        _assert_lines_equal(la, lb)
        assert 20 < len(la) < 190  # roughly half of the lines are occluded
        assert np.array_equal(da.data_points()["model_point_index"], db.data_points()["model_point_index"])


def test_error_conventions_match_reference():
    hip = util.open_hip()
    inputs = scenes.Inputs(1, 1, n_divides=1)
    a = scenes.Instance(hip, inputs)
    assert not a.tracker.StartModalities(0)  # no image uploaded yet ("Set up ... first")
    assert "first" in hip.last_error()
    assert not a.tracker.ExecuteTrackingStep(0)
    with pytest.raises(util.pkg.M3TError) as e:
        # region checking needs its renderer: it is switched on with UseRegionChecking(renderer) afterwards
        host.RegionModality(hip, a.bodies[0], a.color_cams[0], a.region_models[0], use_region_checking=1)
    assert e.value.code == -1
    with pytest.raises(util.pkg.M3TError) as e:  # a renderer geometry needs bodies with a mesh
        host.RendererGeometry(hip).AddBody(a.bodies[0])
    assert e.value.code == -2
    with pytest.raises(util.pkg.M3TError):
        host.RegionModality(hip, a.bodies[0], a.color_cams[0], a.region_models[0], n_histogram_bins=12)
    with pytest.raises(util.pkg.M3TError):
        host.RegionModel(hip, path="/nonexistent/model.bin")
