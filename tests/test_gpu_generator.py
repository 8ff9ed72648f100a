"""GenerateConfiguredTracker on the device: the reference's tracker_config.yaml end to end — meshes and
metafiles read, both sparse viewpoint models generated on the GPU and saved, poses detected, one tracking
step — against the golden pose of TrackerTest.OptimizePoseMatrixGeneratorSetUp (test/tracker_test.cpp:182-195)."""
import os

import ctypes as C

import numpy as np
import pytest

import util
from test_generator import reference_tree, run_generated_tracker

pytestmark = pytest.mark.gpu
cfg = util.pkg.config


def test_generated_tracker_with_generated_models(tmp_path):
    root = reference_tree(tmp_path)
    api = util.open_hip()
    tracker = run_generated_tracker(api, root)
    golden = util.read_golden_matrix("tracker_test/triangle_pose.txt")
    pose = tracker.body_ptrs()[0].body2world_pose()
    assert np.max(np.abs((pose - golden)[:3] / golden[:3])) < 1e-5  # CompareToLoadedMatrix(..., 1.0e-5f)

    # the models were saved where the metafiles say, in the reference's file format, for this body
    body = tracker.objects["Body"]["triangle"]
    for name, region in (("triangle_region_model", True), ("triangle_depth_model", False)):
        model = tracker.objects["RegionModel" if region else "DepthModel"][name]
        assert model.model_path == str(tmp_path / "temp" / (name + ".bin")) and model.n_views == 2562
        assert cfg.model_bin_matches(model.model_path, region, model.parameters, body.body_data())
        floats = (38 if region else 36) * 200 + 4
        assert os.path.getsize(model.model_path) > 2562 * floats * 4

    # a second tracker finds them and loads instead of generating: same pose, bit for bit
    api2 = util.open_hip()
    stamp = os.path.getmtime(tracker.objects["RegionModel"]["triangle_region_model"].model_path)
    tracker2 = run_generated_tracker(api2, root)
    assert os.path.getmtime(tracker2.objects["RegionModel"]["triangle_region_model"].model_path) == stamp
    assert np.array_equal(tracker2.body_ptrs()[0].body2world_pose(), pose)


@pytest.mark.parametrize("roi", [False, True])
def test_tracker_process_over_the_fixture_sequence(tmp_path, roi):
    """RunTrackerProcess-style loop: detect, start, then one step per loaded frame until the images run out
    (frames 200 and 201 exist); the object does not move between them, so the pose stays at the first step's.
    roi: the loader cameras hand their frames over as rectangles (enable_roi_ingest) -- the same poses"""
    root = reference_tree(tmp_path)
    api = util.open_hip()
    tracker = util.pkg.generator.GenerateConfiguredTracker(api, str(root / "tracker_test" / "tracker_config.yaml"))
    if roi:
        tracker.enable_roi_ingest(True, margin_px=16.0)
    assert tracker.RunTrackerProcess(5) is False  # not set up
    assert tracker.SetUp()
    assert tracker.RunTrackerProcess(5) == 2      # image 202 is missing: the process stops there
    golden = util.read_golden_matrix("tracker_test/triangle_pose.txt")  # the pose after the first step
    pose = tracker.body_ptrs()[0].body2world_pose()
    assert np.linalg.norm(pose[:3, 3] - golden[:3, 3]) < 5e-3


def test_loader_camera_overlapped_ingest(tmp_path):
    """LoaderColorCamera with the decode / upload pipeline (frame k + 2 decoded on a worker thread into a page-locked
    slab, frame k + 1 crossing PCIe on the copy stream while frame k is tracked; loader_camera.cpp:76-98 is the
    blocking step it replaces) against the blocking loader: the same poses after every frame, bit for bit, and
    UpdateImage fails the same way where the sequence ends"""
    from PIL import Image
    import scenes
    from util import host, syn
    n_frames = 9
    inputs = scenes.Inputs(1, n_frames, n_divides=2)
    for k in range(n_frames):
        Image.fromarray(np.ascontiguousarray(inputs.color[0][k][:, :, ::-1])).save(str(tmp_path / ("color_%04d.png" % k)))
    intr = inputs.intr
    results = {}
    # ("no read-back": nothing waits for a step until the sequence ends, the slabs are recycled on the strength of
    # camera_slot_sync alone; "roi": the frames go as the tracker's rectangle, m3t_hip_cameras_upload_batch_roi_async,
    # with adaptive margins -- "roi tight": with a margin of 1 pixel, so that steps are repeated on whole frames)
    for prefetch in (False, True, "no read-back", "roi", "roi tight", "roi no read-back"):
        api = util.open_hip()
        cam = util.pkg.generator.LoaderColorCamera(
            api, str(tmp_path), (intr["fu"], intr["fv"], intr["ppu"], intr["ppv"], intr["width"], intr["height"]),
            "color_", 0, 4, "")
        cam.enable_prefetch(bool(prefetch))
        if str(prefetch).startswith("roi"):
            cam.enable_roi_ingest(True, margin_px=1.0 if prefetch == "roi tight" else 24.0, adaptive=prefetch != "roi tight")
        body = host.Body(api, inputs.start[0])
        m = inputs.region_models[0]
        model = host.RegionModel(api, data_points=m[0], orientations=m[1], contour_lengths=m[2])
        region = host.RegionModality(api, body, cam, model, **dict(syn.RBOT_REGION_PARAMS, measure_occlusions=0))
        host.Optimizer(api, body=body, modalities=[region], tikhonov_parameter_rotation=1000.0,
                       tikhonov_parameter_translation=30000.0)
        tracker = host.Tracker(api, 7, 2)
        assert cam.UpdateImage(True)
        assert tracker.StartModalities(0)
        cam.set_load_index(0)
        poses = []
        for k in range(n_frames):
            assert cam.UpdateImage(True)
            assert np.array_equal(cam.image, inputs.color[0][k])
            assert tracker.ExecuteTrackingStep(k)
            if not str(prefetch).endswith("no read-back"):
                poses.append(body.body2world_pose())
        assert not cam.UpdateImage(True)  # frame 9 does not exist
        results[prefetch] = poses or [body.body2world_pose()]
        if str(prefetch).startswith("roi"):
            bodies, n, pulls = (C.c_int * 8)(), C.c_int(0), C.c_longlong(0)
            api.call("roi_get_status", bodies, 8, C.byref(n), C.byref(pulls))
            assert pulls.value >= n_frames - 3, pulls.value  # rectangles, not whole frames
            print(prefetch, "steps repeated on whole frames:", n.value)
            api.call("roi_get_unrecovered", bodies, 8, C.byref(n))
            assert n.value == 0
    for key in (True, "roi", "roi tight"):
        for a, b in zip(results[False], results[key]):
            assert np.array_equal(a, b), key
    assert np.array_equal(results["no read-back"][-1], results[True][-1])
    assert np.array_equal(results["roi no read-back"][-1], results[True][-1])
    e = syn.pose_errors(results[True][-1], inputs.gt[0][-1])
    assert e[0] < np.deg2rad(5) and e[1] < 0.05
