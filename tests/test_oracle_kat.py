"""Pins the CPU oracle against the reference's own reachable known-answer tests
(SURVEY.md §8c).  CPU only."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import util
from util import host


def test_color_histograms_closed_form():
    """ColorHistogramsTest.TestHistogramCalculation (test/color_histograms_test.cpp:71-103),
    n_bins=32, learning rates 0.5 (data/color_histograms_test/color_histograms.yaml)."""
    api = util.open_oracle()
    lib = api.lib
    lib.m3t_oracle_histograms_create.restype = C.c_void_p
    lib.m3t_oracle_histograms_create.argtypes = [C.c_int, C.c_float, C.c_float]
    for n in ("clear_memory", "initialize", "update", "destroy"):
        getattr(lib, "m3t_oracle_histograms_" + n).argtypes = [C.c_void_p]
        getattr(lib, "m3t_oracle_histograms_" + n).restype = None
    for n in ("add_foreground", "add_background"):
        getattr(lib, "m3t_oracle_histograms_" + n).argtypes = [C.c_void_p, C.c_char_p]
        getattr(lib, "m3t_oracle_histograms_" + n).restype = None
    lib.m3t_oracle_histograms_get_probabilities.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float),
                                                            C.POINTER(C.c_float)]
    lib.m3t_oracle_histograms_get_probabilities.restype = None
    n_bins = 32
    h = lib.m3t_oracle_histograms_create(n_bins, 0.5, 0.5)
    assert h
    color1 = bytes([7, 100, 200])   # cv::Vec3b color1{7, 100, 200} style: two colours in different bins
    color2 = bytes([250, 1, 60])

    def prob(c):
        pf, pb = C.c_float(), C.c_float()
        lib.m3t_oracle_histograms_get_probabilities(h, c, C.byref(pf), C.byref(pb))
        return pf.value, pb.value

    lib.m3t_oracle_histograms_add_foreground(h, color1)
    lib.m3t_oracle_histograms_add_background(h, color2)
    lib.m3t_oracle_histograms_clear_memory(h)
    lib.m3t_oracle_histograms_update(h)
    u = np.float32(1.0) / np.float32(n_bins ** 3)
    assert prob(color1) == (u, u)
    lib.m3t_oracle_histograms_add_foreground(h, color1)
    lib.m3t_oracle_histograms_add_foreground(h, color2)
    lib.m3t_oracle_histograms_add_background(h, color2)
    lib.m3t_oracle_histograms_initialize(h)
    assert prob(color1) == (0.5, 0.0)
    assert prob(color2) == (0.5, 1.0)
    lib.m3t_oracle_histograms_add_foreground(h, color2)
    lib.m3t_oracle_histograms_add_background(h, color1)
    lib.m3t_oracle_histograms_update(h)
    assert prob(color1) == (0.25, 0.5)
    assert prob(color2) == (0.75, 0.5)
    lib.m3t_oracle_histograms_destroy(h)
    for bad in (3, 0, 128):
        assert not lib.m3t_oracle_histograms_create(bad, 0.5, 0.5)


def _parse_bin(path, region):
    """independent numpy parse of the .bin layout"""
    raw = open(path, "rb").read()
    off = 0
    model_type = raw[0:1]
    off = 1
    version, = struct.unpack_from("<i", raw, off); off += 4
    sphere_radius, n_divides, n_points, max_radius, stride = struct.unpack_from("<fiiff", raw, off); off += 20
    off += 1 + 4

    def body(o):
        n, = struct.unpack_from("<Q", raw, o)
        return o + 8 + n + 4 + 1 + 1 + 4 + 64
    off = body(off)
    n_assoc, = struct.unpack_from("<Q", raw, off); off += 8
    if region:
        for _ in range(4):
            n, = struct.unpack_from("<Q", raw, off); off += 8
            for _ in range(n):
                off = body(off)
    else:
        for _ in range(n_assoc):
            off = body(off)
    n_views, = struct.unpack_from("<Q", raw, off); off += 8
    pf = 38 if region else 36
    rec = n_points * pf + 4
    a = np.frombuffer(raw, np.float32, n_views * rec, off).reshape(n_views, rec)
    assert off + n_views * rec * 4 == len(raw)
    pts = a[:, :n_points * pf].reshape(n_views, n_points, pf)
    return model_type, version, n_points, pts, a[:, n_points * pf:n_points * pf + 3], a[:, -1], stride, max_radius


@pytest.mark.parametrize("region", [True, False])
def test_bin_golden_loader_and_closest_view(region):
    """data/model_test/{region,depth}_model.bin: loader + GetClosestView
    (region_model.cpp:105-130,259-307; depth_model.cpp:81-106,215-283)."""
    name = "region_model.bin" if region else "depth_model.bin"
    path = os.path.join(util.GOLDEN, "model_test", name)
    mt, version, n_points, pts, ori, ext, stride, max_radius = _parse_bin(path, region)
    assert mt == (b"r" if region else b"d") and version == (10 if region else 9)
    assert pts.shape[:2] == (162, 10)
    api = util.open_oracle()
    model = host.RegionModel(api, path=path) if region else host.DepthModel(api, path=path)
    assert (model.n_views, model.n_points) == (162, 10)
    me = model.max_contour_length if region else model.max_surface_area
    assert me == pytest.approx(float(ext.max()), rel=0, abs=0)
    # the model created from raw arrays behaves the same as the loaded one
    if region:
        model2 = host.RegionModel(api, data_points=pts, orientations=ori, contour_lengths=ext,
                                  stride_depth_offset=stride, max_radius_depth_offset=max_radius)
    else:
        model2 = host.DepthModel(api, data_points=pts, orientations=ori, surface_areas=ext,
                                 stride_depth_offset=stride, max_radius_depth_offset=max_radius)
    rng = np.random.default_rng(3)
    body2world = util.inv_pose_f32(util.SCHAUMA_WORLD2BODY)
    poses = [body2world] + [util.syn.make_pose(util.syn.random_rotation(rng), rng.normal(size=3)) for _ in range(50)]
    for T in poses:
        T = np.asarray(T, np.float32)
        o = np.linalg.inv(T[:3, :3].astype(np.float64)) @ (T[:3, 3] / np.linalg.norm(T[:3, 3]))
        dots = ori.astype(np.float64) @ o
        expect = int(np.argmax(dots))
        got = model.GetClosestView(T)
        if got != expect:  # only a float32 near-tie may differ
            assert abs(dots[got] - dots[expect]) < 1e-6
        assert model2.GetClosestView(T) == got
    # fixture pose: view 128 of 162 (dot 0.990), SURVEY §8c
    if region:
        assert model.GetClosestView(body2world) == 128
    # zero translation -> view 0
    Z = np.eye(4, dtype=np.float32)
    assert model.GetClosestView(Z) == 0


def _dummy_modalities(api):
    cam = host.ColorCamera(api, **util.COLOR_INTR)
    dcam = host.DepthCamera(api, depth_scale=0.001, **util.DEPTH_INTR)
    rm = host.RegionModel(api, path=os.path.join(util.GOLDEN, "model_test", "region_model.bin"))
    dm = host.DepthModel(api, path=os.path.join(util.GOLDEN, "model_test", "depth_model.bin"))
    body = host.Body(api, util.inv_pose_f32(util.TRIANGLE_WORLD2BODY))
    region = host.RegionModality(api, body, cam, rm, n_lines_max=10)
    depth = host.DepthModality(api, body, dcam, dm, n_points_max=10)
    return body, region, depth


def test_optimizer_closed_form_kat():
    """OptimizerTest.Optimize (test/optimizer_test.cpp:97-105): region(global)+depth golden g/H,
    lambda = 5000 / 500000, start pose = inverse of world2body (common_test.cpp:10-12)
    -> data/optimizer_test/triangle_pose.txt.  The goldens carry 6 significant digits and
    were regenerated within the 1e-3 modality tolerance, so compare with abs tol 1e-5
    (SURVEY §8c (3))."""
    api = util.open_oracle()
    body, region, depth = _dummy_modalities(api)
    gr = util.read_golden_matrix("modality_test/region_modality_global_gradient.txt")[:, 0]
    hr = util.read_golden_matrix("modality_test/region_modality_global_hessian.txt")
    gd = util.read_golden_matrix("modality_test/depth_modality_gradient.txt")[:, 0]
    hd = util.read_golden_matrix("modality_test/depth_modality_hessian.txt")
    region.set_gradient_hessian(gr, hr)
    depth.set_gradient_hessian(gd, hd)
    host.Optimizer(api, body=body, modalities=[region, depth], tikhonov_parameter_rotation=5000.0,
                   tikhonov_parameter_translation=500000.0)
    tracker = host.Tracker(api, 1, 1)
    assert tracker.CalculateOptimization(0, 0, 0)
    pose = body.body2world_pose()
    golden = util.read_golden_matrix("optimizer_test/triangle_pose.txt")
    assert np.max(np.abs(pose - golden)) < 1e-5
    # independent float64 solve of the same system
    A = -(hr + hd) + np.diag([5000.0] * 3 + [500000.0] * 3)
    theta = np.linalg.solve(A, gr + gd)
    T0 = util.inv_pose_f32(util.TRIANGLE_WORLD2BODY).astype(np.float64)
    dT = util.syn.make_pose(util.syn.rot_vec(theta[:3]), theta[3:])
    assert np.max(np.abs(pose - T0 @ dT)) < 2e-6


def test_static_detector_golden_is_start_pose():
    """triangle_static_detector.yaml link2world == inverse(world2body)"""
    txt = open(os.path.join(util.GOLDEN, "_body", "triangle_static_detector.yaml")).read()
    data = txt[txt.index("[") + 1:txt.index("]")]
    m = np.array([float(x) for x in data.replace("\n", " ").split(",")]).reshape(4, 4)
    assert np.max(np.abs(m - util.inv_pose_f32(util.TRIANGLE_WORLD2BODY))) < 2e-6


def test_not_set_up_returns_false():
    """reference error convention: steps return false before images exist (modality_test.cpp:150-158)"""
    api = util.open_oracle()
    body, region, depth = _dummy_modalities(api)
    tracker = host.Tracker(api, 7, 2)
    assert not tracker.StartModalities(0)
    assert not tracker.CalculateCorrespondences(0, 0)
    assert not tracker.ExecuteTrackingStep(0)
    assert "first" in api.last_error()


def test_oracle_parallel_step_equals_the_serial_step():
    """m3t_oracle_execute_tracking_step_parallel (OpenMP parallel-for over objects, bench.py's all-cores CPU
    baseline) walks every object's loop nest on its own: the poses and histograms equal the serial step's."""
    import ctypes as C
    import scenes
    inputs = scenes.Inputs(5, 3, n_divides=1)
    out = []
    for parallel in (False, True):
        ora = util.open_oracle()
        inst = scenes.Instance(ora, inputs)
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
        f = ora.lib.m3t_oracle_execute_tracking_step_parallel
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        buckets = (C.c_double * 4)()
        for k in range(3):
            inst.upload_frame(k)
            if parallel:
                assert f(ora.ctx, k, 3, buckets) == 0
            else:
                assert inst.tracker.ExecuteTrackingStep(k)
        if parallel:
            assert all(b > 0 for b in buckets)
        out.append((np.stack(inst.poses()), [np.concatenate(r.histograms()) for r in inst.region]))
    assert np.array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
