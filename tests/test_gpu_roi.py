"""ROI ingest (m3t_hip_set_roi_ingest, m3t_hip_cameras_upload_batch_roi_async): only the rectangle of
every frame that the trackers can read is pulled out of the page-locked host block, overlapped with the previous
tracking step; the poses of the sequence equal those of the blocking whole-frame hand-over bit for bit.  The guarded
kernels report no body that left its rectangle -- and when the margin is too small for the motion, the step of that
body is repeated on the whole frame inside the same call: still the whole-frame poses, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import scenes
import util

pytestmark = pytest.mark.gpu


def status(hip):
    bodies = (C.c_int * 64)()
    n = C.c_int(0)
    pulls = C.c_longlong(0)
    hip.call("roi_get_status", bodies, 64, C.byref(n), C.byref(pulls))
    return n.value, list(bodies[:min(n.value, 64)]), pulls.value


def unrecovered(hip):
    bodies = (C.c_int * 64)()
    n = C.c_int(0)
    hip.call("roi_get_unrecovered", bodies, 64, C.byref(n))
    return n.value


def run(inputs, mode, margin=24.0, n_frames=None, with_depth=False, reserve_cus=0, adaptive=False, expect_kernel=None,
        region_params=None):
    n_frames = n_frames or inputs.n_frames
    hip = util.open_hip()
    if reserve_cus:
        hip.call("reserve_ingest_cus", reserve_cus)
    inst = scenes.Instance(hip, inputs, use_depth=with_depth, region_params=region_params)
    if mode != "blocking":
        hip.call("set_roi_ingest", 2 if adaptive else 1, C.c_float(margin))
    inst.upload_frame(0)
    assert inst.tracker.StartModalities(0)
    out = []
    if mode == "blocking":
        for k in range(1, n_frames):
            inst.upload_frame(k)
            assert inst.tracker.ExecuteTrackingStep(k)
            out.append(np.stack(inst.poses()))
        return out, (0, [], 0)
    n = inputs.n_objects
    groups = [(inst.color_cams, inputs.color, 3, np.uint8)]
    if with_depth:
        groups.append((inst.depth_cams, inputs.depth, 1, np.uint16))
    rings = []
    for cams, frames, channels, dtype in groups:
        h, w = frames[0][0].shape[:2]
        blocks = []
        for k in range(n_frames):
            b = np.zeros((n, h, w * channels), dtype)
            for i in range(n):
                b[i] = frames[i][k].reshape(h, w * channels)
            inst.tracker.register_host_buffer(b)
            blocks.append(b)
        ids = (C.c_int * n)(*[cam.id for cam in cams])
        hip.call("cameras_set_ring", ids, n, 2)
        rings.append((ids, blocks))

    def upload(slot, k):
        for ids, blocks in rings:
            b = blocks[k]
            hip.call("cameras_upload_batch_roi_async", ids, n, slot, b.ctypes.data_as(C.c_void_p), b.strides[0], b.strides[1])

    upload(1, 1)  # (no step recorded yet: goes as whole frames)
    for k in range(1, n_frames):
        inst.tracker.select_slot(k % 2)
        assert inst.tracker.ExecuteTrackingStep(k)
        if k + 1 < n_frames:
            upload((k + 1) % 2, k + 1)
        out.append(np.stack(inst.poses()))  # (synchronises: the next upload is not overlapped here, the path is the same)
    inst.tracker.ingest_sync()
    assert unrecovered(hip) == 0
    kernel = C.create_string_buffer(128)
    hip.call("get_step_kernel", kernel, 128)
    assert b"_guard_kernel" in kernel.value, kernel.value  # the last step read rectangles
    assert expect_kernel is None or kernel.value.decode() == expect_kernel, kernel.value
    return out, status(hip)


def test_rectangles_track_like_whole_frames():
    inputs = scenes.Inputs(6, 7, n_divides=2)
    ref, _ = run(inputs, "blocking")
    got, (misses, bodies, pulls) = run(inputs, "roi")
    assert pulls >= inputs.n_frames - 3  # all but the first uploads went as rectangles
    assert misses == 0, bodies
    for k, (a, b) in enumerate(zip(got, ref)):
        assert np.array_equal(a, b), k


@pytest.mark.parametrize("shape", ["split", "one workgroup", "compact"])
def test_rectangles_region_and_depth(shape, monkeypatch):
    """Region + Depth objects (YCB parameters: measured occlusions, two cameras per object) through every guarded
    kernel: rectangles of the colour AND the depth frames, the poses of whole frames, nobody repeated"""
    if shape != "split":
        monkeypatch.setenv("M3T_HIP_NO_SPLIT", "1")
    if shape == "compact":
        monkeypatch.setenv("M3T_HIP_COMPACT", "1")
    inputs = scenes.Inputs(3, 6, n_divides=2, with_depth=True)
    ref, _ = run(inputs, "blocking", with_depth=True)
    got, (misses, bodies, pulls) = run(inputs, "roi", with_depth=True,
                                       expect_kernel={"split": "tracking_step_split_guard_kernel",
                                                      "one workgroup": "tracking_step_lds_guard_kernel",
                                                      "compact": "tracking_step_compact_guard_kernel"}[shape])
    assert pulls >= 2 * (inputs.n_frames - 3) and misses == 0, (pulls, bodies)
    for k, (a, b) in enumerate(zip(got, ref)):
        assert np.array_equal(a, b), k


def runaway_inputs():
    """object 1 runs away sideways by 2 cm (~24 pixels) per frame from frame 3 on -- which the tracker follows"""
    inputs = scenes.Inputs(3, 7, n_divides=2)
    runner = inputs.scenes[1]
    for k in range(3, inputs.n_frames):
        pose = inputs.gt[1][k].copy()
        pose[0, 3] -= 0.02 * (k - 2)
        inputs.gt[1][k] = pose
        inputs.color[1][k] = runner.render(pose)
    return inputs


@pytest.mark.parametrize("shape", ["split", "one workgroup", "one workgroup, pair table in LDS", "compact"])
def test_a_body_that_outruns_its_rectangle_is_repeated_on_the_whole_frame(shape, monkeypatch):
    """(every guarded kernel: the split one the planner picks for three objects, tracking_step_guard_kernel with the
    split switched off, tracking_step_compact_guard_kernel with the compact kernel forced)
    the runaway's rectangle (computed from the pose two frames back, with the margin that is enough for the ordinary
    motion of the other two bodies) does not hold what its step needs.  The guarded kernel drops that step, the
    library fetches the body's whole frame from the host block and repeats the step for that body alone: the poses
    of ALL bodies are those of the blocking whole-frame hand-over, bit for bit, and roi_get_status names the runaway,
    and only the runaway, as repeated"""
    if shape != "split":
        monkeypatch.setenv("M3T_HIP_NO_SPLIT", "1")
    if shape == "compact":
        monkeypatch.setenv("M3T_HIP_COMPACT", "1")
    inputs = runaway_inputs()
    params = dict(util.syn.RBOT_REGION_PARAMS, n_histogram_bins=16) if "LDS" in shape else None
    ref, _ = run(inputs, "blocking", region_params=params)
    got, (misses, bodies, pulls) = run(inputs, "roi", margin=24.0, region_params=params,
                                       expect_kernel={"split": "tracking_step_split_guard_kernel",
                                                      "one workgroup": "tracking_step_guard_kernel",
                                                      "one workgroup, pair table in LDS": "tracking_step_lds_guard_kernel",
                                                      "compact": "tracking_step_compact_guard_kernel"}[shape])
    assert pulls > 0 and misses >= 1
    assert set(bodies) == {1}, bodies  # body ids are creation order: object 1 is body 1
    for k, (a, b) in enumerate(zip(got, ref)):
        assert np.array_equal(a, b), k
    # the runaway was tracked (the repeated steps saw the frames)
    e = util.syn.pose_errors(got[-1][1], inputs.gt[1][-1])
    assert e[1] < 0.02, e


def test_adaptive_margins_track_like_whole_frames():
    """set_roi_ingest(2, cap): every body's margin follows what its rectangle moved over the last step (three times
    that, at least 4 pixels, at most the cap) -- fewer bytes per frame; a body that accelerates past it is repeated.
    Poses bit for bit those of whole frames, with and without the runaway"""
    for inputs in (scenes.Inputs(6, 8, n_divides=2), runaway_inputs()):
        ref, _ = run(inputs, "blocking")
        got, (misses, bodies, pulls) = run(inputs, "roi", margin=48.0, adaptive=True)
        assert pulls >= inputs.n_frames - 3
        for k, (a, b) in enumerate(zip(got, ref)):
            assert np.array_equal(a, b), (k, misses, bodies)


def test_pull_kernel_on_reserved_cus():
    """m3t_hip_reserve_ingest_cus: the tracking step on the CUs of one mask, the ROI pull kernel on the others (frame
    k + 1 crosses PCIe while step k runs).  Which CUs run what changes nothing in the results: the poses of the
    sequence equal the blocking whole-frame hand-over bit for bit, region + depth, no body outside its rectangle"""
    inputs = scenes.Inputs(3, 6, n_divides=2, with_depth=True)
    ref, _ = run(inputs, "blocking", with_depth=True)
    got, (misses, bodies, pulls) = run(inputs, "roi", with_depth=True, reserve_cus=32)
    assert pulls >= 2 * (inputs.n_frames - 3) and misses == 0, (pulls, bodies)
    for k, (a, b) in enumerate(zip(got, ref)):
        assert np.array_equal(a, b), k


def test_reserving_cus_replans_the_launch_and_replaces_the_stream():
    """64 objects x 4 workgroups need all 256 CUs; with 32 of them reserved the launch is planned for 224 (2 workgroups
    per object), the context's stream is a new one, the poses stay what they were; 0 gives everything back"""
    inputs = scenes.Inputs(64, 3, n_divides=2, n_models=4)
    poses, shapes, streams = [], [], []
    for reserve in (0, 32, -1):  # -1: reserve, then give back
        hip = util.open_hip()
        s0 = C.c_void_p()
        hip.call("get_stream", C.byref(s0))
        if reserve:
            hip.call("reserve_ingest_cus", 32)
        if reserve < 0:
            hip.call("reserve_ingest_cus", 0)
        s1 = C.c_void_p()
        hip.call("get_stream", C.byref(s1))
        streams.append((s0.value, s1.value))
        inst = scenes.Instance(hip, inputs)
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
        for k in range(1, inputs.n_frames):
            inst.upload_frame(k)
            assert inst.tracker.ExecuteTrackingStep(k)
        shape = (C.c_int * 4)()
        hip.call("get_step_shape", shape)
        shapes.append(list(shape))
        poses.append(np.stack(inst.poses()))
        assert hip.raw("reserve_ingest_cus", -8) < 0 and hip.raw("reserve_ingest_cus", 1 << 20) < 0
    assert shapes[0][1] == 4 and shapes[1][1] == 2 and shapes[2][1] == 4, shapes
    assert streams[0][0] == streams[0][1] and streams[1][0] != streams[1][1]
    assert np.array_equal(poses[0], poses[1]) and np.array_equal(poses[0], poses[2])
